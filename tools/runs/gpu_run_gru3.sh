#!/bin/bash
mkdir -p gpurun_out/r02k
timeout 300 python tools/bench_gru.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02k/bench_gru.log
cat gpurun_out/r02k/bench_gru.log
timeout 600 python -m pytest tests/test_encoder.py -m gpu -x -q > gpurun_out/r02k/t.log 2>&1
tail -15 gpurun_out/r02k/t.log
timeout 300 python bench.py --mode encoder --steps 10 --warmup 2 > gpurun_out/r02k/bench_encoder.json 2> gpurun_out/r02k/bench_encoder.err
cat gpurun_out/r02k/bench_encoder.json
