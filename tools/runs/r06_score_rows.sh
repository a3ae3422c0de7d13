#!/bin/bash
# the several-queries score pass at beam 5: rows per workgroup chosen per launch (product) against fixed values (CVC_SCORE_ROWS_RT)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for r in "" "$@"; do
  if [ -n "$r" ]; then export CVC_SCORE_ROWS_RT=$r; else unset CVC_SCORE_ROWS_RT; fi
  for c in cfg3 cfg5; do
    python bench.py --config $c --beam 5 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_tmp.json 2>/dev/null
    python - <<PY
import json
j = json.load(open("gpurun_out/ab_tmp.json"))
k = {k["kernel"]: k["avg_us"] for k in j["kernels"]}
print("rows='${r:-chosen}' $c", j["value"], "attn_scores", k.get("attn_scores"))
PY
  done
done
