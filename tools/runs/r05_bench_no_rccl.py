"""bench.py's decode line on a box where the RCCL communicator cannot be brought up: the decode path has no collective, so the
line is still produced, says so ("rccl": ...) and counts its ranks on the gloo control plane.  Run under a launcher:
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/runs/r05_bench_no_rccl.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cyclical-visual-captioning_amd"))
import bench                      # noqa: E402
from cvc.comm import RcclComm     # noqa: E402


def broken(group=None):
    raise OSError("simulated: librccl cannot be opened")


RcclComm.from_process_group = staticmethod(broken)
sys.argv = ["bench.py", "--gpus", "1", "--steps", "20", "--warmup", "2", "--no-secondary"]
bench.main()
