"""Gate GEMM forms side by side at cfg2 size (R = 2048, 64 rows): full-K kernel, K-split + finishing launch, K-split with the
last-arriver finish, K-split with the exchange finish (cvc_packed_lstm_ksx_fwd).  Time per call (weights flushed out of the
Infinity Cache between calls, as in the decode step), result against the full-K kernel, error word, run-to-run bits."""
import sys, torch
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "cyclical-visual-captioning_amd"))
from cvc import hip
from cvc.decode import pack_weights, to_quad, from_quad
dev = torch.device("cuda:0")
L = hip.lib(); st = torch.cuda.current_stream().cuda_stream
R, M = 2048, 64
big = torch.empty(512 << 20, device=dev, dtype=torch.uint8)
g = torch.Generator().manual_seed(5)
for K, name, eg in ((3 * R, "lang", False), (2 * R, "att-embgate", True), (2 * R + 1024, "att", False)):
    w = (torch.randn(4 * R, K, generator=g) / K ** 0.5).to(dev)
    wp = pack_weights(w, R)
    b1 = (torch.randn(4 * R, generator=g) * 0.1).to(dev); b2 = (torch.randn(4 * R, generator=g) * 0.1).to(dev)
    gb = (torch.randn(M, 4 * R, generator=g) * 0.2).to(dev)
    x = torch.randn(M, K, generator=g).to(dev); c_prev = torch.randn(M, R, generator=g).to(dev)
    V = 300
    table = (torch.randn(V, 4 * R, generator=g) * 0.3).to(dev) if eg else None
    word = torch.randint(0, V, (M,), generator=g).to(dev) if eg else None
    xq, cq = to_quad(x), to_quad(c_prev)
    S = L.cvc_packed_lstm_ks_slices(K, R)
    slab = torch.empty(S * (R // 8) * 2048, device=dev)
    counters = torch.zeros(R // 64, dtype=torch.int32, device=dev)
    flags = torch.zeros(R // 8 + 1, dtype=torch.int32, device=dev)
    seq = [0]
    def outs():
        return [torch.zeros(R // 4, 64, 4, device=dev) for _ in range(3)]
    p = lambda t: None if t is None else t.data_ptr()
    def full(h1, h2, c2):
        if eg:
            return L.cvc_packed_lstm_embgate_fwd(p(wp), p(xq), K, p(b1), p(b2), p(gb), p(table), p(word), p(cq), M, R, p(h1), p(h2), p(c2), st)
        return L.cvc_packed_lstm_fwd(p(wp), p(xq), K, p(b1), p(b2), p(gb), p(cq), M, R, p(h1), p(h2), p(c2), st)
    def ks(h1, h2, c2):
        return L.cvc_packed_lstm_ks_fwd(p(wp), p(xq), K, p(b1), p(b2), p(gb), p(cq), M, R, p(h1), p(h2), p(c2), p(slab), wp.stride(0), st)
    def ksf(h1, h2, c2):
        return L.cvc_packed_lstm_ksf_fwd(p(wp), p(xq), K, p(b1), p(b2), p(gb), p(cq), M, R, p(h1), p(h2), p(c2), p(slab), p(counters), st)
    def ksx(h1, h2, c2):
        seq[0] += 1
        return L.cvc_packed_lstm_ksx_fwd(p(wp), p(xq), K, p(b1), p(b2), p(gb), p(table), p(word), p(cq), M, R, p(h1), p(h2), p(c2), p(slab),
                                         p(flags), seq[0], st)
    def ksx_mode(mode):
        def f(*o):
            L.cvc_packed_lstm_ksx_local(mode)
            rc = ksx(*o)
            L.cvc_packed_lstm_ksx_local(1)
            return rc
        return f
    forms = [("full-K", full), ("ksx-1", ksx_mode(1)), ("ksx-2", ksx_mode(2)), ("ksx-3", ksx_mode(3)), ("ksx-sys", ksx_mode(0))] + ([] if eg else [("ks+finish", ks)])
    ref = None
    for fname, fn in forms:
        o = outs(); assert fn(*o) == 0; torch.cuda.synchronize()
        got = [from_quad(t, M) for t in o]
        if ref is None:
            ref = got
        err = max(float((a - b).abs().max()) for a, b in zip(got, ref))
        bits = True
        for _ in range(3):
            o2 = outs(); assert fn(*o2) == 0; torch.cuda.synchronize()
            bits &= all(torch.equal(a, b) for a, b in zip(o, o2))
        tot, n = 0.0, 30
        for it in range(n + 3):
            big.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(*o); e1.record(); torch.cuda.synchronize()
            if it >= 3: tot += e0.elapsed_time(e1)
        print(f"{name:12s} K={K} {fname:10s}: {tot / n * 1e3:6.1f} us   max |diff| vs full-K {err:.2e}   same bits run to run: {bits}   "
              f"error word {int(flags[R // 8])}", flush=True)

if hasattr(L, "cvc_debug_ks_ts_read"):
    import ctypes as C, numpy as np
    buf = (C.c_ulonglong * (256 * 8))()
    L.cvc_debug_ks_ts_read.argtypes = [C.c_void_p]
    o = outs(); big.fill_(1); ksx(*o); torch.cuda.synchronize()
    assert L.cvc_debug_ks_ts_read(buf) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8).astype(np.int64)
    r = (t - t[:, 0].min()) / 100.0
    names = ["entry", "main loop done", "slab rows acknowledged", "tile's flags seen", "slab loads back", "end"]
    for k, nm in enumerate(names):
        print(f"  {nm:24s} min {r[:, k].min():6.2f} p50 {np.median(r[:, k]):6.2f} max {r[:, k].max():6.2f}")
    print("  per-phase p50:", " ".join(f"{np.median(r[:, k + 1] - r[:, k]):6.2f}" for k in range(5)))
