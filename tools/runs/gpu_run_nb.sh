#!/bin/bash
mkdir -p gpurun_out/r02q
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r02q/t.log 2>&1
tail -3 gpurun_out/r02q/t.log
for k in 1 2 1 2; do
  timeout 300 python bench.py --lstm-blocks $k --no-cpu-baseline --steps 300 > gpurun_out/r02q/b_$k.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('gpurun_out/r02q/b_$k.json'))
ks={e['kernel']:e['avg_us'] for e in d['kernels']}
print('blocks=$k', d['value'], d['ms_per_step'], ks)
PY
done
