export TMPDIR=/tmp
mkdir -p gpurun_out/r02h
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_encoder.py -m gpu -x -q > gpurun_out/r02h/t.log 2>&1; tail -8 gpurun_out/r02h/t.log
timeout 600 python bench.py --mode train --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'])
print([ (k['kernel'],k.get('ms_per_step')) for k in d['kernels']][:14])"
timeout 600 python bench.py --mode train --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline --train-graph 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('graph', d['value'], d['ms_per_step'])"
