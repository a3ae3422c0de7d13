export TMPDIR=/tmp
mkdir -p gpurun_out/r02c
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile" 2>&1 | tail -3
timeout 300 python tools/bench_tile.py "" 0 1 0 1 2>/dev/null | tee gpurun_out/r02c/tile_loaders.log
timeout 600 python bench.py --beam 5 --steps 20 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'])
print([ (k['kernel'],k.get('avg_us')) for k in d['kernels']][:14])"
