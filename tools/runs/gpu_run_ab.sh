#!/bin/bash
mkdir -p gpurun_out/r02n
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r02n/t.log 2>&1
tail -3 gpurun_out/r02n/t.log
for k in 0 1 2 0; do
  timeout 300 python bench.py --gate-ksplit $k --no-cpu-baseline --steps 300 > gpurun_out/r02n/b_$k.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('gpurun_out/r02n/b_$k.json'))
ks={e['kernel']:e['avg_us'] for e in d['kernels']}
print('ksplit=$k', d['value'], d['ms_per_step'], ks)
PY
done
