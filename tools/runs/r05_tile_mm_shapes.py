#!/usr/bin/env python3
"""Round-5 measurement: cvc.hip.tile_mm at the shapes of one cyclical training step (config 3: B = 64, T = 20, R = 2048, E = A = 1024,
V = 5000; S = 2 T B = 2560 sample rows), operands packed beforehand (the product alone), caches flushed between calls:
time, fp32-equivalent TFLOP/s and the fraction of the split-product roof (2.5 PFLOP/s bf16 / 6)."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cyclical-visual-captioning_amd")]
from cvc import hip
if len(sys.argv) > 1:
    hip.LIB_PATH = os.path.abspath(sys.argv[1])
hip.lib()
if os.environ.get("LOADERS"):          # A/B: 1 = 8 computing waves, 2 = 4 wide computing waves, 3 = per launch (library default)
    hip.lib().cvc_tile_gemm_loaders(int(os.environ["LOADERS"]))
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "64"))
S, n, R, E, V = 2 * 20 * B, 20 * B, 2048, 1024, 5000
shapes = [("dW lstm block (x6)", 4 * R, R, S, 6), ("dW emb block", 4 * R, E, S, 1), ("dW fc block", 4 * R, R, 2 * B, 1),
          ("hoisted emb (x2)", n, 4 * R, E, 2), ("hoisted ctx", n, 4 * R, R, 1), ("d_emb (x2)", n, E, 4 * R, 2), ("d_ctx", n, R, 4 * R, 1),
          ("head fwd (x2)", n, V, R, 2), ("head dX (x2)", n, R, V, 2), ("head dW (x2)", V, R, n, 2), ("h2attn dW", E, R, n, 1)]
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
tot = 0.0
for name, M, N, K, cnt in shapes:
    a, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    A, Bo = hip.TileOperand(a), hip.TileOperand(b)
    out = torch.empty(M, N, device=dev)
    ts = []
    for rep in range(7):
        flush.fill_(rep)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); hip.tile_mm(A, Bo, out=out); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    us = sorted(ts[2:])[2]
    tf = 2.0 * M * N * K / us / 1e6
    tot += us * cnt
    print(f"{name:22s} M={M:5d} N={N:5d} K={K:5d}: {us:7.1f} us  {tf:6.1f} TF  {tf / (2500 / 6):.2f} of the roof   x{cnt}")
print(f"sum over a step: {tot / 1e3:.2f} ms")
