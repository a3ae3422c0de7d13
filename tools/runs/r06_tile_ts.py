"""Who waits at the tile GEMM's per-k-step barrier?  Needs a variant library built with -DCVC_TILE_TS
(tools/runs/build_variant.sh tilets gemm_tile.hip -DCVC_TILE_TS; CVC_LIB=.../variants/libcvc_tilets.so)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cyclical-visual-captioning_amd"))
import numpy as np
import torch
from cvc import hip
from cvc.decode import pack_weights_tile, to_frag

dev = torch.device("cuda:0")
L = hip.lib()
hip.lib().cvc_tile_gemm_loaders(int(os.environ.get("TILE_MODE", "2")))
for name, M, K, N, ks in [("lang_lstm beam 5", 320, 6144, 8192, 4), ("dW lstm block", 8192, 2560, 2048, 1)]:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    wb, xb = pack_weights_tile(w), to_frag(x, hip.tile_rows_alloc(M))
    parts = torch.empty(ks, M, N, device=dev)
    for _ in range(3):
        hip.tile_gemm(wb, xb, 0, K, M, N, ks, parts)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        hip.tile_gemm(wb, xb, 0, K, M, N, ks, parts)
    e1.record(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * (2 * 64 * 3))()
    fn = C.CDLL(hip.LIB_PATH).cvc_tile_ts_read
    assert fn(buf) == 0
    t = np.array(buf, dtype=np.int64).reshape(2, 64, 3).astype(np.float64) * 10.0        # ns
    c_arr, c_rel = t[0, :, 0], t[0, :, 1]
    l_data, l_rel, l_iss = t[1, :, 0], t[1, :, 1], t[1, :, 2]
    clk = (C.c_ulonglong * 4)()
    assert C.CDLL(hip.LIB_PATH).cvc_tile_clk_read(clk) == 0
    ghz = (clk[3] - clk[1]) / ((clk[2] - clk[0]) * 10.0)
    print(f"   shader clock over k steps 8 .. 71 of workgroup 0: {ghz:.2f} GHz (s_memtime / s_memrealtime)")
    n = slice(4, 60)
    step = np.diff(c_rel[n]).mean()
    print(f"{name}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch; k step {step:.0f} ns (pure MFMA 800 ns at 2.4 GHz)")
    print(f"   computing wave 0: waits at the barrier {np.mean(c_rel[n] - c_arr[n]):.0f} ns per k step (arrival -> release)")
    print(f"   loader wave 0:    data of its stage complete {np.mean(l_rel[n] - l_data[n]):.0f} ns BEFORE the barrier releases; issuing the next "
          f"stage's copies takes {np.mean(l_iss[n] - l_rel[n]):.0f} ns; then waits {np.mean(l_data[n][1:] - l_iss[n][:-1]):.0f} ns for the data")
