export TMPDIR=/tmp
mkdir -p gpurun_out/r02b
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile" > gpurun_out/r02b/tile_test.log 2>&1; tail -25 gpurun_out/r02b/tile_test.log
timeout 300 python tools/bench_tile.py > gpurun_out/r02b/bench_tile.log 2>&1; cat gpurun_out/r02b/bench_tile.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "a10 or nan or ground" > gpurun_out/r02b/a10.log 2>&1; tail -15 gpurun_out/r02b/a10.log
