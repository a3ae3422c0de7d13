export TMPDIR=/tmp
mkdir -p gpurun_out/r02i
timeout 900 python -m pytest tests/test_dataloader.py -m gpu -x -q > gpurun_out/r02i/t.log 2>&1; tail -30 gpurun_out/r02i/t.log
