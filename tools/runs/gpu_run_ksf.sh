#!/bin/bash
mkdir -p gpurun_out/r02l
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ksplit_equals" > gpurun_out/r02l/t.log 2>&1
tail -5 gpurun_out/r02l/t.log
for k in 0 2 1 2 0; do
  timeout 300 python bench.py --gate-ksplit $k --no-cpu-baseline --steps 300 > gpurun_out/r02l/b_$k.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('gpurun_out/r02l/b_$k.json'))
ks={e['kernel']:e['avg_us'] for e in d['kernels']}
print('ksplit=$k', d['value'], d['ms_per_step'], 'att', ks.get('att_lstm'), 'lang', ks.get('lang_lstm'))
PY
done
