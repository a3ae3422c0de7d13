export TMPDIR=/tmp
mkdir -p gpurun_out/r02h
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile_mm or backward_data or cyclical or a9" > gpurun_out/r02h/t.log 2>&1; tail -12 gpurun_out/r02h/t.log
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_encoder.py -m gpu -x -q > gpurun_out/r02h/t2.log 2>&1; tail -3 gpurun_out/r02h/t2.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "cyclical or ground" > gpurun_out/r02h/t3.log 2>&1; tail -3 gpurun_out/r02h/t3.log
timeout 600 python bench.py --mode train --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'])
print([ (k['kernel'],k.get('ms_per_step'), k.get('avg_us')) for k in d['kernels']][:16])"
