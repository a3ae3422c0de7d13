#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02o
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*\|SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_WAIT[A-Z_0-9_]*\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_INSTS_VALU[A-Z_0-9]*" | sort -u > $O/counters.txt
cat $O/counters.txt | tr '\n' ' '
echo
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d $O/pmc_$tag -o p -- python3 $R/tools/bench_tile.py lang_lstm 1 2 > $O/pmc_$tag.log 2>&1
  DB=$(find $O/pmc_$tag -name "*.db" | head -1)
  python3 $R/tools/rocpd_pmc.py $DB 2>&1 | grep "tile_gemm" | cut -c1-160
  rm -rf $O/pmc_$tag
done
