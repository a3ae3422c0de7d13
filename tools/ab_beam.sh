#!/bin/bash
# A/B builds on the GPU box, beam-5 decode (row-major ring GEMM path): each argument is a hipcc flag set
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for flags in "$@"; do
  export CVC_EXTRA_HIPCC_FLAGS="$flags"
  python cyclical-visual-captioning_amd/build_hip.py --force > /dev/null 2>&1
  python bench.py --beam 5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ab_tmp.json 2>/dev/null
  python - <<PY
import json
j = json.load(open("gpurun_out/ab_tmp.json"))
print("flags='$flags'", j["value"], {k["kernel"]: k["avg_us"] for k in j["kernels"]})
PY
done
export CVC_EXTRA_HIPCC_FLAGS=""
python cyclical-visual-captioning_amd/build_hip.py --force > /dev/null 2>&1
