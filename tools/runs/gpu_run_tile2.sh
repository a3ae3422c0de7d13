#!/bin/bash
mkdir -p gpurun_out/r02o
timeout 300 python tools/bench_tile.py "" 1 2 1 2 2>&1 | grep -v amdgpu > gpurun_out/r02o/tile.log
cat gpurun_out/r02o/tile.log
