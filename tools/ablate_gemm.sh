#!/bin/bash
# Ablation of the ring GEMM kernel on the GPU box: full / no-DMA / no-MFMA builds, kernel timings from bench.py
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for abl in 0 1 2; do
  if [ "$abl" = "0" ]; then export CVC_EXTRA_HIPCC_FLAGS=""; else export CVC_EXTRA_HIPCC_FLAGS="-DCVC_ABL=$abl"; fi
  python cyclical-visual-captioning_amd/build_hip.py --force > /dev/null 2>&1
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/abl_$abl.json 2>/dev/null
  python - <<PY
import json
j = json.load(open("gpurun_out/abl_$abl.json"))
print("ABL=$abl", {k["kernel"]: k["avg_us"] for k in j["kernels"] if k["kernel"] in ("att_lstm", "lang_lstm", "logits", "h2attn", "gate_fc")})
PY
done
export CVC_EXTRA_HIPCC_FLAGS=""
python cyclical-visual-captioning_amd/build_hip.py --force > /dev/null 2>&1
