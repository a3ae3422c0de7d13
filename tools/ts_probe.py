"""Where the gate GEMM's time goes inside the launch (diagnostic build: CVC_EXTRA_HIPCC_FLAGS=-DCVC_TS): per-workgroup
timestamps (entry, first chunk multiplied, K loop done, end) of the last language / attention cell launch of a captured decode.
Prints, relative to the earliest entry of the launch: dispatch spread, first-data latency, loop end spread, epilogue length."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "cyclical-visual-captioning_amd"))
from cvc import synth, hip
from cvc.decode import DecodeEngine, DecodeWeights

dev = torch.device("cuda:0")
d = synth.CONFIGS["cfg2"]
sd = synth.hot_path_state_dict(d, 1234)
feats_np = synth.clip_features(d, 1234)
W = DecodeWeights({k: torch.from_numpy(v).to(dev) for k, v in sd.items()})
feats = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in feats_np.items()}
eng = DecodeEngine(W, feats, d.T, synth.UNK_IDX, beam=1)
eng.capture()
for _ in range(20):
    eng.run()
torch.cuda.synchronize()
L = hip.lib()
buf = (C.c_ulonglong * (2 * 256 * 8 * 4))()
L.cvc_debug_ts_read.argtypes = [C.c_void_p]
assert L.cvc_debug_ts_read(buf) == 0
ts = np.frombuffer(buf, dtype=np.uint64).reshape(2, 256, 8, 4).astype(np.int64)
import os
snaps = [ts.copy()]
for _ in range(3):                                         # further launches: is the pattern systematic or random?
    eng.run(); torch.cuda.synchronize()
    assert L.cvc_debug_ts_read(buf) == 0
    snaps.append(np.frombuffer(buf, dtype=np.uint64).reshape(2, 256, 8, 4).astype(np.int64).copy())
os.makedirs("gpurun_out/r03m", exist_ok=True)
np.save("gpurun_out/r03m/ts.npy", np.stack(snaps))
for slot, name in ((0, "lang"), (1, "att")):
    t = ts[slot]
    t0 = t[:, :, 0].min()
    r = (t - t0) / 100.0                                   # us (100 MHz clock)
    pr = lambda lab, a: print(f"  {lab:34s} min {a.min():6.2f}  p50 {np.median(a):6.2f}  p90 {np.percentile(a, 90):6.2f}  max {a.max():6.2f}")
    print(name)
    pr("entry (after earliest entry)", r[:, :, 0])
    pr("first chunk multiplied - entry", r[:, :, 1] - r[:, :, 0])
    pr("K loop done (abs)", r[:, :, 2])
    pr("K loop done, last wave per wg", r[:, :, 2].max(1))
    pr("K loop length per wave", r[:, :, 2] - r[:, :, 1])
    pr("end - own loop done (waves 0,1)", (r[:, :2, 3] - r[:, :2, 2]))
    pr("end - wg's last loop done", r[:, :2, 3].max(1) - r[:, :, 2].max(1))
    pr("end (abs, waves 0,1)", r[:, :2, 3])
    # per-XCD view of the loop end
    xe = [r[x::8, :, 2].max() for x in range(8)]
    print("  loop end per XCD (max):", " ".join(f"{v:6.2f}" for v in xe))
