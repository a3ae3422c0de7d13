export TMPDIR=/tmp
mkdir -p gpurun_out/r02d
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile or beam or ragged" > gpurun_out/r02d/t.log 2>&1; tail -30 gpurun_out/r02d/t.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "cfg3_beam5" > gpurun_out/r02d/t2.log 2>&1; tail -5 gpurun_out/r02d/t2.log
timeout 600 python bench.py --beam 5 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r02d/bench_beam5.json 2> gpurun_out/r02d/bench_beam5.err; tail -3 gpurun_out/r02d/bench_beam5.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02d/bench_beam5.json'))
print(d['value'], d['ms_per_step'], d['config'].get('engine_path'))
for k in d['kernels']: print('   ', k['kernel'], k.get('avg_us'), k.get('share'), k.get('launches_per_decode'))
PY
