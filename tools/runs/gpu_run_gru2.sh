#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02k
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --mode encoder --steps 3 --warmup 1 > $O/kt.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py $DB > $O/kt_enc.md 2>&1
rm -rf $O/kt
head -24 $O/kt_enc.md
