#!/usr/bin/env python3
"""Round-5 measurement: the 128-row backward-data product (cvc_linear_nn_planes2_fwd) against two 64-row launches, at the two LSTM
cells' shapes of config 3 (K = 4R = 8192; language cell: 3 segments of R columns; attention cell: 2), caches flushed between calls.
usage: python tools/runs/r05_nn128_bench.py [lib.so ...]   (each library is measured in its own child process)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cyclical-visual-captioning_amd")]


def one(lib_path):
    import torch
    from cvc import hip
    if lib_path:
        hip.LIB_PATH = lib_path
    L = hip.lib()
    dev = torch.device("cuda:0")
    R, K = 2048, 8192
    g = torch.Generator().manual_seed(1)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    out = {}
    for name, nseg in (("lang", 3), ("att", 2)):
        W = (torch.randn(K, nseg * R, generator=g) * 0.02).to(dev)
        dA, dC = torch.randn(64, K, generator=g).to(dev), torch.randn(64, K, generator=g).to(dev)
        qA, qC = hip.pack_quad(dA), hip.pack_quad(dC)
        slabs = nseg * R // 128
        ks = max(1, min(K // 8 // 16, 256 // slabs))
        arr = (hip.NNSeg * nseg)()
        outs = [torch.empty(128, R, device=dev) for _ in range(nseg)]
        for i in range(nseg):
            arr[i] = hip.NNSeg(W.data_ptr() + 4 * i * R, outs[i].data_ptr(), W.stride(0), R, R)
        ws = torch.empty(ks * 128 * slabs * 128, device=dev)

        def f128():
            hip._check(L.cvc_linear_nn_planes2_fwd(qA.data_ptr(), qC.data_ptr(), K, 64, 64, arr, nseg, ks, ws.data_ptr(), 0, hip._stream()), "nn128")

        def f64():
            hip._check(L.cvc_linear_nn_planes_fwd(qA.data_ptr(), K, 64, arr, nseg, ks, ws.data_ptr(), hip._stream()), "nn64")
        for tag, fn in (("128", f128), ("64", f64)):
            ts = []
            for rep in range(12):
                flush.fill_(rep)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts = sorted(ts[2:])
            out[f"{name}.{tag}"] = round(ts[len(ts) // 2], 1)
    print(os.path.basename(lib_path or "default"), out, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        one(sys.argv[2] if len(sys.argv) > 2 else None)
    else:
        libs = sys.argv[1:] or [""]
        for lib in libs:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + ([os.path.abspath(lib)] if lib else []), check=False)
