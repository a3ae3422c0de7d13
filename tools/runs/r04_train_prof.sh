#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04a}; mkdir -p $O
python3 $R/bench.py --mode train --config cfg3 --steps 30 --warmup 3 --no-cpu-baseline > $O/train.json 2> $O/train.err
python3 - <<PY
import json
j = json.load(open("$O/train.json"))
print("train", j["value"], j["ms_per_step"], j.get("roofline"))
for k in j.get("kernels", [])[:25]: print(k)
PY
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --mode train --config cfg3 --steps 8 --warmup 2 --no-cpu-baseline > $O/kt.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py $DB > $O/train_kernel_stats.md 2>&1
rm -rf $O/kt
head -45 $O/train_kernel_stats.md
