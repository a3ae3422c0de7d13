#!/usr/bin/env python3
"""Round-5 evidence (review item 2): the RCCL scenarios of tests/test_gpu_train.py, each in a fresh child interpreter, N times in a
row with NO retry -- the exchange runs on the package's own communicator (cvc.comm.RcclComm), so there is no c10d watchdog thread that
could die under the graph capture (round 4: about one capture in five, then 1 in 30, then slept around).
usage: python tools/runs/r05_rccl_scenarios_x50.py [N=50]  -> one line per scenario: passes / runs, wall seconds; exit code 1 on any failure"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
paths = [ROOT, os.path.join(ROOT, "cyclical-visual-captioning_amd"), os.path.join(ROOT, "tests")]
# (the config-4-size scenario runs the oracle's autograd on the host for ~10 s per run: a fifth of the repetitions)
scenarios = [("gradient_exchange_on_rccl_one_rank", N), ("rccl_allreduce_cabi_one_rank", N), ("cfg4_share_on_rccl_arenas", max(1, N // 5))]
bad = 0
for name, n in scenarios:
    ok, t0, fails = 0, time.time(), []
    for i in range(n):
        code = f"import sys; sys.path[:0] = {paths!r}; import test_gpu_train as t; t._scenario_{name}()"
        r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", code], capture_output=True, text=True, cwd=ROOT,
                           env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        if r.returncode == 0 and "SCENARIO-PASSED" in r.stdout:
            ok += 1
        else:
            fails.append((i, r.returncode, r.stderr[-600:]))
    print(f"{name}: {ok}/{n} passed, rc 0 and marker after teardown every time: {ok == n}; {time.time() - t0:.0f} s", flush=True)
    for f in fails[:3]:
        print("   FAILED run", f)
    bad += n - ok
sys.exit(1 if bad else 0)
