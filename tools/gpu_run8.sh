export TMPDIR=/tmp
mkdir -p gpurun_out/r02h
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "backward_data or cyclical or a9" > gpurun_out/r02h/t.log 2>&1; tail -5 gpurun_out/r02h/t.log
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q > gpurun_out/r02h/t2.log 2>&1; tail -3 gpurun_out/r02h/t2.log
timeout 300 python tools/bench_nn.py 2>&1 | tail -6
timeout 600 python bench.py --mode train --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'])
print([ (k['kernel'],k.get('ms_per_step'), k.get('avg_us')) for k in d['kernels']][:8])"
timeout 300 python bench.py --mode encoder --steps 10 --warmup 2 2>&1 | tail -1 | cut -c1-1500
