#!/bin/bash
# kernel-trace timeline of the greedy decode step (graph replays): per-position duration + idle gap
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03l}; mkdir -p $O
shift
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary "$@" > $O/kt.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python3 $R/tools/rocpd_step_timeline.py $DB 7 > $O/timeline.md 2>&1
python3 $R/tools/rocpd_summary.py $DB > $O/kernel_stats.md 2>&1
rm -rf $O/kt
cat $O/timeline.md; head -12 $O/kernel_stats.md; tail -3 $O/kt.log
