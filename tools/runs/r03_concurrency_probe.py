"""Can the vocabulary projection run NEXT TO the attention cell's gate GEMM?  Both in their 4-wave forms (cvc_gemm_packed_split(1),
build with -DCVC_PACKED_DEPTH=3: 232 + 212 VGPRs, 39 KB of LDS each, one wave per SIMD each -> they fit on one CU together).
Launches on two streams against the same two launches on one stream."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "cyclical-visual-captioning_amd"))
from cvc import hip
from cvc.decode import pack_weights, to_quad
dev = torch.device("cuda:0")
L = hip.lib()
R, M, V = 2048, 64, 5000
g = torch.Generator().manual_seed(1)
big = torch.empty(300 << 20, device=dev, dtype=torch.uint8)
for mode in (2, 1):
    hip.gemm_packed_split(mode)
    Ka = 2 * R
    w = (torch.randn(4 * R, Ka, generator=g) / Ka ** 0.5).to(dev)
    wa = pack_weights(w, R)
    xa = to_quad(torch.randn(M, Ka, generator=g).to(dev)); ca = to_quad(torch.randn(M, R, generator=g).to(dev))
    h1, h2, c2 = (torch.zeros(R // 4, 64, 4, device=dev) for _ in range(3))
    # the vocabulary head's pack: 32-row blocks in row order (cvc.decode.pack_linear)
    wo = pack_weights((torch.randn(V, R, generator=g) / R ** 0.5).to(dev))
    xo = to_quad(torch.randn(M, R, generator=g).to(dev))
    bo = torch.randn(V, generator=g).to(dev)
    top2 = torch.empty(((V + 31) // 32) * 64 * 6, device=dev)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    att = lambda st: L.cvc_packed_lstm_fwd(wa.data_ptr(), xa.data_ptr(), Ka, None, None, None, ca.data_ptr(), M, R, h1.data_ptr(), h2.data_ptr(), c2.data_ptr(), st)
    logits = lambda st: L.cvc_packed_linear_fwd(wo.data_ptr(), xo.data_ptr(), R, bo.data_ptr(), M, V, 1, None, V, top2.data_ptr(), st)
    def timed(fn, n=40):
        tot = 0.0
        for it in range(n + 3):
            big.fill_(it & 1)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record(); torch.cuda.synchronize()
            if it >= 3: tot += e0.elapsed_time(e1)
        return tot / n * 1e3
    cur = torch.cuda.current_stream()
    NP = 20            # pairs per timed region: the streams' fork / join (events) is paid once per region, not per pair
    def serial():
        for _ in range(NP):
            assert logits(cur.cuda_stream) == 0 and att(cur.cuda_stream) == 0
    def conc():
        sa.wait_stream(cur); sb.wait_stream(cur)
        for _ in range(NP):
            assert logits(sb.cuda_stream) == 0
            assert att(sa.cuda_stream) == 0
        cur.wait_stream(sa); cur.wait_stream(sb)
    def only(fn):
        def f():
            for _ in range(NP): fn(cur.cuda_stream)
        return f
    only_a = timed(only(att)) / NP; only_l = timed(only(logits)) / NP
    print(f"split mode {mode}: per pair -- att alone {only_a:.1f} us, logits alone {only_l:.1f} us, one stream {timed(serial) / NP:.1f} us, "
          f"two streams {timed(conc) / NP:.1f} us", flush=True)
