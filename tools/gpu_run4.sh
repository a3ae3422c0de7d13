export TMPDIR=/tmp
mkdir -p gpurun_out/r02d
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -m gpu -x -q > gpurun_out/r02d/t.log 2>&1; tail -4 gpurun_out/r02d/t.log
for args in "--beam 5 --steps 20 --warmup 2" "--steps 20 --warmup 5" "--mode train --config cfg3 --steps 10 --warmup 3"; do
timeout 600 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'])
print([ (k['kernel'],k.get('avg_us')) for k in d['kernels']][:14])"
done
