"""timing of the two small training helpers against the library launches they replace"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cyclical-visual-captioning_amd")]
import torch
from cvc import hip
dev = torch.device("cuda:0")


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for n in (640, 1280, 1920, 7168):
    idx = torch.randint(0, 8000, (n,), device=dev)
    print(f"order n={n}: ours {t(lambda: hip.stable_order(idx)):.1f} us, argsort {t(lambda: torch.argsort(idx, stable=True)):.1f} us")
for S, n in ((64, 4096), (128, 4096), (1280, 512), (2560, 8192), (1280, 1024)):
    x = torch.randn(S, n, device=dev)
    o = torch.empty(n, device=dev)
    print(f"col_sum {S}x{n}: ours {t(lambda: hip.col_sum(x, o)):.1f} us, torch.sum {t(lambda: torch.sum(x, 0, out=o)):.1f} us")
