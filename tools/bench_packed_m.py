"""Gate GEMM (cvc_packed_lstm_fwd) time against the number of batch rows: how much of the launch is activation ingress."""
import sys, torch
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "cyclical-visual-captioning_amd"))
from cvc import hip
from cvc.decode import pack_weights, to_quad
dev = torch.device("cuda:0")
L = hip.lib(); st = torch.cuda.current_stream().cuda_stream
R = 2048
big = torch.empty(512 << 20, device=dev, dtype=torch.uint8)
for K, name in ((3 * R, "lang"), (2 * R + 1024, "att")):
    w = torch.randn(4 * R, K, device=dev) / K ** 0.5
    wp = pack_weights(w, R)
    b1 = torch.randn(4 * R, device=dev) * 0.1; b2 = torch.randn(4 * R, device=dev) * 0.1
    for M in (64, 48, 32, 16, 1):
        x = torch.randn(M, K, device=dev); c_prev = torch.randn(M, R, device=dev)
        xq, cq = to_quad(x), to_quad(c_prev)
        h1, h2, c2 = (torch.zeros(R // 4, 64, 4, device=dev) for _ in range(3))
        fn = lambda: L.cvc_packed_lstm_fwd(wp.data_ptr(), xq.data_ptr(), K, b1.data_ptr(), b2.data_ptr(), None, cq.data_ptr(), M, R, h1.data_ptr(), h2.data_ptr(), c2.data_ptr(), st)
        tot = 0.0; n = 30
        for _ in range(n + 3):
            big.fill_(1)                      # weights out of the Infinity Cache, as in the decode step
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            if _ >= 3: tot += e0.elapsed_time(e1)
        print(f"{name} K={K} M={M:2d}: {tot / n * 1e3:6.1f} us", flush=True)
