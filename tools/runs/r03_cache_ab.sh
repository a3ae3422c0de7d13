#!/bin/bash
# A/B: which 200 MB of the step's 844 MB stay in the Infinity Cache -- the projected features (default) or the attention cell's weights
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03p; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-secondary > $O/b_$tag.json 2> $O/b_$tag.err; python - <<PY
import json
j = json.load(open("$O/b_$tag.json"))
print("$tag", j["value"], j["ms_per_step"], [(k["kernel"], k["avg_us"]) for k in j["kernels"]])
PY
}
run default X=1
run attw_budget50 CVC_ATT_W_CACHED=1 CVC_CACHE_BUDGET_MB=50
run attw_budget80 CVC_ATT_W_CACHED=1 CVC_CACHE_BUDGET_MB=80
run attw_budget130 CVC_ATT_W_CACHED=1 CVC_CACHE_BUDGET_MB=130
run noattw_budget130 CVC_CACHE_BUDGET_MB=130
run attw_budget208 CVC_ATT_W_CACHED=1
