#!/bin/bash
# bench.py --gpus 2 --mode train at BASELINE config 4 with two rank processes on ONE GPU (stand-in librccl, eager steps), call trace on
set -u
mkdir -p gpurun_out/two_rank
gcc -O2 -shared -fPIC -o /tmp/librccl.so tests/stub_rccl/stub_rccl.c -lpthread -lrt -ldl || exit 1
export CVC_RCCL_LIB=/tmp/librccl.so CVC_BENCH_DEVICE=0 CVC_STUB_TRACE=${CVC_STUB_TRACE:-1}
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 \
  bench.py --gpus 2 --mode train --config cfg4 --steps 5 --warmup 2 --no-train-graph --no-cpu-baseline --watchdog-seconds 150 \
  > gpurun_out/two_rank/train.out 2> gpurun_out/two_rank/train.err
echo rc=$?
grep -c '^{' gpurun_out/two_rank/train.out
grep -c stub_rccl gpurun_out/two_rank/train.err
grep "stub_rccl" gpurun_out/two_rank/train.err | tail -24
