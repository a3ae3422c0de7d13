#!/bin/bash
# training-form packed LSTM: parity tests, then the training bench with the forward on either kernel
mkdir -p gpurun_out/r02j
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lstm_train_form" > gpurun_out/r02j/t.log 2>&1
tail -5 gpurun_out/r02j/t.log
python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r02j/t2.log 2>&1
tail -5 gpurun_out/r02j/t2.log
python bench.py --mode train --steps 30 --warmup 5 > gpurun_out/r02j/bench_train.json 2> gpurun_out/r02j/bench_train.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02j/bench_train.json'))
print(d['ms_per_step'], d['value'])
for k in d.get('kernels', [])[:8]: print(k['kernel'], k.get('launches_per_step'), k['ms_per_step'], k.get('avg_us'))
PY
