#!/usr/bin/env python3
"""Micro-benchmark of the backward-data GEMM (csrc/gemm_nn.hip) against the library GEMMs it replaces.
  python tools/bench_nn.py            (on the GPU box)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cyclical-visual-captioning_amd"))
from cvc import hip  # noqa: E402


def timeit(fn, n=30):
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


def main():
    dev = torch.device("cuda:0")
    M, R, E = 64, 2048, 1024
    K = 4 * R
    for name, in_w, needed in (("lang_lstm all", 2 * R, [(0, R), (R, R)]), ("att_lstm all", 2 * R + E, [(0, R), (R, R), (2 * R, E)]),
                               ("att_lstm no-fc", 2 * R + E, [(0, R), (2 * R, E)])):
        w_ih = torch.randn(K, in_w, device=dev) / K ** 0.5
        w_hh = torch.randn(K, R, device=dev) / K ** 0.5
        dg = torch.randn(M, K, device=dev)
        dq = torch.zeros(K // 4, 64, 4, device=dev)
        dq.copy_(dg.view(M, K // 4, 4).permute(1, 0, 2))
        ranges = [(w_hh, 0, R)] + [(w_ih, c, n) for c, n in needed]
        for ks in (int(os.environ.get("KS", "0")) or None,) if os.environ.get("KS") else (8, 10):
            t = timeit(lambda: hip.linear_nn(dq, M, K, ranges, ksplit=ks))
            print("   ksplit", ks, "%.1f us" % t)
        t_nn = timeit(lambda: hip.linear_nn(dq, M, K, ranges))
        t_mm = timeit(lambda: [torch.mm(dg, w[:, c:c + n]) for (w, c, n) in ranges])
        byt = sum(n for _, _, n in ranges) * K * 4
        print("%-16s nn %.1f us (%.2f TB/s, %.1f TF)   torch.mm x%d %.1f us" % (
            name, t_nn, byt / t_nn / 1e6, 2 * M * byt / 4 / t_nn / 1e6, len(ranges), t_mm))


if __name__ == "__main__":
    main()
