#!/bin/bash
# A/B of the several-queries score pass (csrc/attn_scores.h) on variant libraries built by tools/runs/build_variant.sh:
# prints attn_scores avg_us at beam 5 for cfg3 and cfg5
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in "" "$@"; do
  if [ -n "$v" ]; then export CVC_LIB=$PWD/cyclical-visual-captioning_amd/cvc/lib/variants/libcvc_$v.so; else unset CVC_LIB; fi
  for c in cfg3 cfg5; do
    python bench.py --config $c --beam 5 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_tmp.json 2>/dev/null
    python - <<PY
import json
j = json.load(open("gpurun_out/ab_tmp.json"))
k = {k["kernel"]: k["avg_us"] for k in j["kernels"]}
print("variant='${v:-product}' $c", j["value"], "attn_scores", k.get("attn_scores"), "attn_wsum", k.get("attn_wsum"))
PY
  done
done
