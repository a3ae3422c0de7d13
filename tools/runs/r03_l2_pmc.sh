#!/bin/bash
# the L2-side counter passes of tools/gpu_profile.sh alone (gate GEMM forms, standalone launches)
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_READ_sum -d $O/pmc_l2a -o a -- python3 $R/tools/bench_ksx.py > $O/pmc_l2a.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc_l2b -o b -- python3 $R/tools/bench_ksx.py > $O/pmc_l2b.log 2>&1
{ echo "## TCP_TCC_READ_REQ_sum / TCC_READ_sum (requests per dispatch), tools/bench_ksx.py"; python3 $R/tools/rocpd_pmc.py $(find $O/pmc_l2a -name "*.db" | head -1);
  echo; echo "## TCC_HIT_sum / TCC_MISS_sum"; python3 $R/tools/rocpd_pmc.py $(find $O/pmc_l2b -name "*.db" | head -1); } > $O/${TAG}_pmc_l2_gate_gemm.md 2>&1
rm -rf $O/pmc_l2a $O/pmc_l2b
python3 $R/tools/bench_ksx.py > $O/${TAG}_gate_gemm_forms.log 2>&1
cat $O/${TAG}_pmc_l2_gate_gemm.md; cat $O/${TAG}_gate_gemm_forms.log
