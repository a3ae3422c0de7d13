"""Where the vocabulary projection's ~20 us go (greedy decode, 64 rows x V = 5000 x K = 2048): the packed kernel alone, with its weights
hot (back-to-back launches), after a flush of the caches (a 1 GB streaming pass between launches), for K splits 1..8, for V = 5000
and V = 8192.  Usage (GPU box): python tools/runs/r06_logits_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cyclical-visual-captioning_amd"))
import torch
from cvc import hip
from cvc.decode import pack_weights, to_quad

dev = torch.device("cuda:0")
L = hip.lib()
st = torch.cuda.current_stream().cuda_stream
M, K = 64, 2048
big = torch.empty(256 << 20, device=dev)          # 1 GiB


def run(V, ks, flush, reps=40):
    g = torch.Generator().manual_seed(V + ks)
    w = (torch.randn(V, K, generator=g) / K ** 0.5).to(dev)
    wp, xq = pack_weights(w), to_quad(torch.randn(M, K, generator=g).to(dev))
    b = torch.randn(V, generator=g).to(dev)
    nblk = (V + 31) // 32
    top2 = torch.empty(nblk * M * 4, device=dev)
    parts = torch.empty(max(ks, 1), M, V, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i in range(reps + 3):
        if flush:
            big.add_(1.0)
        a, e = ev[max(i - 3, 0)]
        a.record()
        if ks == 0:
            rc = L.cvc_packed_linear_fwd(wp.data_ptr(), xq.data_ptr(), K, b.data_ptr(), M, V, 1, None, V, top2.data_ptr(), st)
        else:
            rc = L.cvc_packed_linear_fwd(wp.data_ptr(), xq.data_ptr(), K, None, M, V, ks, parts.data_ptr(), V, None, st)
        assert rc == 0, rc
        e.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(e) * 1e3 for a, e in ev)
    return t[len(t) // 2], t[0]


for mode in (2, 1):                 # cvc_gemm_packed_split: 2 = 8 waves per workgroup (default), 1 = 4 waves
    L.cvc_gemm_packed_split(mode)
    for V in (5000, 8192):
        for flush in (False, True):
            for ks in (0, 1, 2, 3, 4, 5, 6, 8):
                med, best = run(V, ks, flush)
                print(f"waves={8 if mode == 2 else 4} V={V} {'flushed' if flush else 'hot    '} {'top2 epilogue' if ks == 0 else 'ksplit=%d     ' % ks}: median {med:6.1f} us"
                      f"  best {best:6.1f} us  ({V * K * 4 / 1e6:.0f} MB of weights)", flush=True)
L.cvc_gemm_packed_split(2)
