#!/bin/bash
mkdir -p gpurun_out/r02k
python -m pytest tests/test_encoder.py -m gpu -x -q > gpurun_out/r02k/t.log 2>&1
tail -15 gpurun_out/r02k/t.log
python bench.py --mode encoder --steps 10 --warmup 2 > gpurun_out/r02k/bench_encoder.json 2> gpurun_out/r02k/bench_encoder.err
tail -3 gpurun_out/r02k/bench_encoder.err
cat gpurun_out/r02k/bench_encoder.json
