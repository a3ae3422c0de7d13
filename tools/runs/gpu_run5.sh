cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02e
cd $R && timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ksplit" 2>&1 | tail -2; cd /tmp
for PAD in 8 0 24; do
sed -i "s/^KS_PAD_QUADS = .*/KS_PAD_QUADS = $PAD/" $R/cyclical-visual-captioning_amd/cvc/decode.py
echo "== pad quads $PAD"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02e/kt -o kt -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r02e/kt.log 2>&1
DB=$(find $R/gpurun_out/r02e/kt -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py $DB | grep -E "packed_ks" | cut -c1-150
grep -o '"value": [0-9.]*' $R/gpurun_out/r02e/kt.log | head -1
rm -rf $R/gpurun_out/r02e/kt
done
