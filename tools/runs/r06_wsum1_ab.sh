#!/bin/bash
# A/B of the one-query weighted sum's hoisted softmax (CVC_WSUM_HOIST=0|1): greedy decode (the headline) and the training step
mkdir -p gpurun_out/wsum
for rep in 1 2; do for h in 0 1; do
  CVC_WSUM_HOIST=$h python bench.py --no-cpu-baseline --no-secondary 2>/dev/null > gpurun_out/wsum/greedy_h$h.json
  CVC_WSUM_HOIST=$h python bench.py --mode train --config cfg3 --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null > gpurun_out/wsum/train_h$h.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/wsum/greedy_h$h.json").read().strip().splitlines()[-1])
k=[x for x in d["kernels"] if x["kernel"] in ("attn_wsum",)]
print("greedy hoist=$h", d["value"], d["ms_per_step"], [(x["kernel"],x["avg_us"],x.get("frac_hbm")) for x in k])
d=json.loads(open("gpurun_out/wsum/train_h$h.json").read().strip().splitlines()[-1])
k=[x for x in d["kernels"] if "attn_wsum" in x["kernel"]]
print("train cfg3 hoist=$h", d["value"], d["ms_per_step"], [(x["kernel"],x["avg_us"],x.get("frac_hbm")) for x in k])
PY
done; done
