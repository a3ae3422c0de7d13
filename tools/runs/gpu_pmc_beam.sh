#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_m -o m -- python3 $R/bench.py --beam 5 --no-graph --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_mfma.log 2>&1
MDB=$(find $O/pmc_m -name "*.db" | head -1)
python3 $R/tools/rocpd_pmc.py $MDB > $O/beam_pmc_mfma_busy.md 2>&1
rm -rf $O/pmc_m
head -20 $O/beam_pmc_mfma_busy.md | cut -c1-220
