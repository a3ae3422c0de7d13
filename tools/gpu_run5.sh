cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02e
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $R/gpurun_out/r02e/pmc_sq -o sq -- python3 $R/bench.py --beam 5 --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $R/gpurun_out/r02e/pmc_sq.log 2>&1
ls -R $R/gpurun_out/r02e/pmc_sq | head
DB=$(find $R/gpurun_out/r02e/pmc_sq -name "*.db" | head -1)
python3 $R/tools/rocpd_pmc.py $DB > $R/gpurun_out/r02e/pmc_sq.md 2>&1
grep -E "attn_scores|attn_wsum|tile_gemm" $R/gpurun_out/r02e/pmc_sq.md | head -40
rm -rf $R/gpurun_out/r02e/pmc_sq
