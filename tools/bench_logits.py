#!/usr/bin/env python3
"""Micro-benchmark of the packed vocabulary projection (fused top-2 epilogue vs plain logits vs M halves).  GPU box."""
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
    for (M, K, V) in ((64, 2048, 5000), (32, 2048, 5000), (64, 2048, 8192), (64, 4096, 5000)):
        w = torch.randn(V, K, device=dev) / K ** 0.5
        wp = pack_weights(w)
        b = torch.randn(V, device=dev)
        xq = to_quad(torch.randn(M, K, device=dev))
        y = torch.empty(M, V, device=dev)
        part = torch.empty((V + 31) // 32, 64, 6, device=dev)
        t_y = timeit(lambda: L.cvc_packed_linear_fwd(wp.data_ptr(), xq.data_ptr(), K, b.data_ptr(), M, V, 1, y.data_ptr(), V, None, st))
        t_p = timeit(lambda: L.cvc_packed_linear_fwd(wp.data_ptr(), xq.data_ptr(), K, b.data_ptr(), M, V, 1, None, V, part.data_ptr(), st))
        y2 = torch.empty(4, M, V, device=dev)
        t_k = [timeit(lambda ks=ks: L.cvc_packed_linear_fwd(wp.data_ptr(), xq.data_ptr(), K, b.data_ptr(), M, V, ks, y2.data_ptr(), V, None, st))
               for ks in (2, 3, 4)]
        print("   split-K logits (slices summed by the consumer): ksplit 2 / 3 / 4 = %.1f / %.1f / %.1f us" % tuple(t_k))
        print("M=%d K=%d V=%d: logits only %.1f us, fused top-2 partials %.1f us  (weights %.0f MB -> %.2f TB/s)" % (
            M, K, V, t_y, t_p, V * K * 4 / 1e6, V * K * 4 / t_p / 1e6))


if __name__ == "__main__":
    main()
