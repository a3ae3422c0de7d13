#!/bin/bash
# A/B of the beam selection summing the vocabulary GEMM slabs itself (default) against the separate finishing pass (CVC_BEAM_FINISH=1)
mkdir -p gpurun_out/beamsel
for cfg in cfg3 cfg5; do
for rep in 1 2; do
for f in "1 1" "0 1"; do
  set -- $f
  CVC_BEAM_FINISH=$1 CVC_BEAM_TWO_STAGE=$2 python bench.py --config $cfg --beam 5 --no-cpu-baseline --no-secondary --steps 60 --warmup 5 2>/dev/null > gpurun_out/beamsel/${cfg}_f$1_t$2.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/beamsel/${cfg}_f$1_t$2.json").read().strip().splitlines()[-1])
k=[x for x in d["kernels"] if x["kernel"] in ("word_select","logits_finish","logits")]
print("$cfg separate_finish=$1 two_stage=$2", d["value"], d["ms_per_step"], [(x["kernel"],x["avg_us"]) for x in k])
PY
done; done; done
