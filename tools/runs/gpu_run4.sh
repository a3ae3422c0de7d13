export TMPDIR=/tmp
mkdir -p gpurun_out/r02d
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r02d/t.log 2>&1; tail -8 gpurun_out/r02d/t.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "cfg3_beam5" 2>&1 | tail -3
for args in "--beam 5 --steps 20 --warmup 2" "--steps 20 --warmup 5" "--config cfg5 --beam 5 --steps 5 --warmup 1"; do
timeout 600 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'])
print([ (k['kernel'],k.get('avg_us')) for k in d['kernels']][:14])"
done
