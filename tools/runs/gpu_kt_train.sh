#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02r
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --mode train --config cfg3 --steps 8 --warmup 2 --no-cpu-baseline > $O/kt.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py $DB > $O/kt_train.md 2>&1
rm -rf $O/kt
head -45 $O/kt_train.md | cut -c1-150
