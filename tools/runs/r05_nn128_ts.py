#!/usr/bin/env python3
"""Round-5 diagnostic: per-wave phase times inside skinny_gemm_nn_split128_kernel (library built with -DCVC_NN128_TS, see
tools/runs/build_variant.sh): K loop / cross-wave sums / stores in shader clocks, entry and exit in s_memrealtime (100 MHz)."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cyclical-visual-captioning_amd")]
from cvc import hip
hip.LIB_PATH = os.path.abspath(sys.argv[1])
L = hip.lib()
dev = torch.device("cuda:0")
R, K = 2048, 8192
g = torch.Generator().manual_seed(1)
for name, nseg in (("lang", 3), ("att", 2)):
    W = (torch.randn(K, nseg * R, generator=g) * 0.02).to(dev)
    qA, qC = hip.pack_quad(torch.randn(64, K, generator=g).to(dev)), hip.pack_quad(torch.randn(64, K, generator=g).to(dev))
    slabs = nseg * R // 128
    ks = max(1, min(K // 8 // 16, 256 // slabs))
    arr = (hip.NNSeg * nseg)()
    outs = [torch.empty(128, R, device=dev) for _ in range(nseg)]
    for i in range(nseg):
        arr[i] = hip.NNSeg(W.data_ptr() + 4 * i * R, outs[i].data_ptr(), W.stride(0), R, R)
    nwg = slabs * ks
    ws = torch.zeros(ks * 128 * slabs * 128 + nwg * 4 * 6 * 2 + 64, device=dev)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    for rep in range(3):
        flush.fill_(rep)
        hip._check(L.cvc_linear_nn_planes2_fwd(qA.data_ptr(), qC.data_ptr(), K, 64, 64, arr, nseg, ks, ws.data_ptr(), 0, hip._stream()), "nn128")
        torch.cuda.synchronize()
    rec = ws[ks * 128 * slabs * 128:].cpu().numpy().view(np.uint64)[: nwg * 4 * 6].reshape(nwg, 4, 6).astype(np.float64)
    t0 = rec[:, :, 0].min()
    start, end = (rec[:, :, 0] - t0) / 100.0, (rec[:, :, 4] - t0) / 100.0          # us
    print(f"{name}: {nwg} workgroups, ksplit {ks}, steps per wave {rec[:, :, 5].min():.0f}..{rec[:, :, 5].max():.0f}")
    print(f"  entry  us: min {start.min():.1f} median {np.median(start):.1f} max {start.max():.1f}")
    print(f"  exit   us: min {end.min():.1f} median {np.median(end):.1f} max {end.max():.1f}")
    for k, nm in ((1, "K loop"), (2, "cross-wave sums"), (3, "stores")):
        c = rec[:, :, k]
        print(f"  {nm:16s} shader clocks: min {c.min():.0f} median {np.median(c):.0f} max {c.max():.0f}   per step {np.median(c) / np.median(rec[:, :, 5]):.0f}" if k == 1 else
              f"  {nm:16s} shader clocks: min {c.min():.0f} median {np.median(c):.0f} max {c.max():.0f}")
    dur = end - start
    print(f"  wave lifetime us: median {np.median(dur):.1f}; implied shader clock {np.median(rec[:, :, 1:4].sum(2)) / np.median(dur):.0f} MHz")
