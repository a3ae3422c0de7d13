#!/usr/bin/env python3
"""Packed GEMM: bf16x3-split products (6 bf16 MFMAs per 16 k) vs plain fp32 MFMA -- error against fp64 and time.  GPU box."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cyclical-visual-captioning_amd"))
from cvc import hip  # noqa: E402
from cvc.decode import pack_weights, to_quad  # noqa: E402


def timeit(fn, n=50):
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
    L = hip.lib()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(0)
    for (M, K, N, name) in ((64, 6144, 8192, "lang-LSTM gates"), (64, 5120, 8192, "att-LSTM gates"), (64, 2048, 5000, "logits"),
                            (64, 2048, 1024, "h2attn"), (17, 256, 96, "small")):
        w = torch.randn(N, K, device=dev) / K ** 0.5
        x = torch.randn(M, K, device=dev)
        x[:, ::7] *= 1e-3                                   # mixed magnitudes
        wp, xq = pack_weights(w), to_quad(x)
        b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev)
        ref = x.double() @ w.double().t() + b.double()
        line = "%-16s M=%d K=%d N=%d:" % (name, M, K, N)
        for mode, label in ((0, "fp32"), (1, "bf16x6/4w"), (2, "bf16x6/8w")):
            hip.lib().cvc_gemm_packed_split(mode)
            run = lambda: L.cvc_packed_linear_fwd(wp.data_ptr(), xq.data_ptr(), K, b.data_ptr(), M, N, 1, y.data_ptr(), N, None, st)
            run()
            torch.cuda.synchronize()
            err = (y.double() - ref).abs()
            t = timeit(run)
            line += "  %s: max %.2e rms %.2e %.1f us" % (label, err.max().item(), err.pow(2).mean().sqrt().item(), t)
        hip.lib().cvc_gemm_packed_split(1)
        print(line)


if __name__ == "__main__":
    main()
