export TMPDIR=/tmp
mkdir -p gpurun_out/r02h
timeout 1500 python -m pytest tests/test_gpu_train.py -m gpu -x -q > gpurun_out/r02h/t.log 2>&1; tail -25 gpurun_out/r02h/t.log
