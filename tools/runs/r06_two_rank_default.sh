#!/bin/bash
# The driver's N = 2 command on a 1-GPU box: both ranks on cuda:0, the stand-in librccl as the transport (eager steps: it cannot be
# captured).  Checks that the default decode bench with its all-ranks cfg4 training secondary FINISHES and prints one line.
set -u
mkdir -p gpurun_out/two_rank
gcc -O2 -shared -fPIC -o /tmp/librccl.so tests/stub_rccl/stub_rccl.c -lpthread -lrt -ldl || exit 1
export CVC_RCCL_LIB=/tmp/librccl.so CVC_BENCH_DEVICE=0 CVC_STUB_TRACE=0
timeout 700 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 5 --warmup 2 --no-train-graph --no-cpu-baseline --watchdog-seconds 600 > gpurun_out/two_rank/default.out 2> gpurun_out/two_rank/default.err
echo rc=$?
grep -c '^{' gpurun_out/two_rank/default.out
python - <<'PY'
import json
for l in open("gpurun_out/two_rank/default.out"):
    if l.startswith("{"):
        d = json.loads(l)
        print({k: d[k] for k in ("metric", "value", "n_gpus", "ranks_joined", "ranks_control_plane", "ms_per_step")})
        for s in d.get("secondary", []):
            print(s.get("name", "?")[:60], "|", s.get("error"), s.get("value"), s.get("exchange"))
PY
tail -8 gpurun_out/two_rank/default.err
