#!/bin/bash
# A/B of the several-queries weighted sum in the beam-5 decode: wave count (CVC_WSUM_MQ_WAVES=4|8) x hoisted softmax (CVC_WSUM_MQ_HOIST=0|1)
mkdir -p gpurun_out/wsum
for cfg in cfg3 cfg5; do
for rep in 1 2; do
for w in 4 8; do for h in 0 1; do
  CVC_WSUM_MQ_WAVES=$w CVC_WSUM_MQ_HOIST=$h python bench.py --config $cfg --beam 5 --no-cpu-baseline --no-secondary --steps 60 --warmup 5 2>/dev/null > gpurun_out/wsum/${cfg}_w${w}_h$h.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/wsum/${cfg}_w${w}_h$h.json").read().strip().splitlines()[-1])
k=[x for x in d["kernels"] if x["kernel"] in ("attn_wsum",)]
print("$cfg waves=$w hoist=$h", d["value"], d["ms_per_step"], [(x["kernel"],x["avg_us"],x.get("frac_hbm")) for x in k])
PY
done; done; done; done
