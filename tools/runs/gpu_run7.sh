export TMPDIR=/tmp
mkdir -p gpurun_out/r02g
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py tests/test_encoder.py -m gpu -x -q > gpurun_out/r02g/t.log 2>&1; tail -6 gpurun_out/r02g/t.log
timeout 300 python tools/bench_sample.py 2>&1 | tail -3
timeout 300 python tools/bench_sample.py rebind 2>&1 | tail -3
for args in "--steps 20 --warmup 5" "--beam 5 --steps 20 --warmup 2"; do
timeout 600 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'])"
done
