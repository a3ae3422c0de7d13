#!/bin/bash
# A/B builds on the GPU box: each argument is a hipcc flag set; prints decode throughput and kernel timings from bench.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for flags in "$@"; do
  export CVC_EXTRA_HIPCC_FLAGS="$flags"
  python cyclical-visual-captioning_amd/build_hip.py --force > /dev/null 2>&1
  for rep in 1 2; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab_tmp.json 2>/dev/null
  python - <<PY
import json
j = json.load(open("gpurun_out/ab_tmp.json"))
print("flags='$flags'", j["value"], {k["kernel"]: k["avg_us"] for k in j["kernels"]})
PY
  done
done
export CVC_EXTRA_HIPCC_FLAGS=""
python cyclical-visual-captioning_amd/build_hip.py --force > /dev/null 2>&1
