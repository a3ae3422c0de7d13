import sys, numpy as np, torch
sys.path.insert(0, "cyclical-visual-captioning_amd")
from cvc import synth, hip
from cvc.decode import DecodeEngine, DecodeWeights
dev = torch.device("cuda:0")
d = synth.CONFIGS["cfg2"]
sd = synth.hot_path_state_dict(d, 1234); f_np = synth.clip_features(d, 1234)
W = DecodeWeights({k: torch.from_numpy(v).to(dev) for k, v in sd.items()})
feats = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in f_np.items()}
for mode in (2, 1):
    hip.gemm_packed_split(mode)
    eng = DecodeEngine(W, feats, d.T, synth.UNK_IDX)
    eng.capture()
    for _ in range(10): eng.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): eng.run()
    e1.record(); torch.cuda.synchronize()
    acc = {}
    eng.run_timed()
    for _ in range(3):
        for k, v in eng.run_timed().items(): acc.setdefault(k, []).extend(v)
    print("split mode", mode, "decode ms", e0.elapsed_time(e1) / 50, {k: round(float(np.mean(v)) * 1e3, 1) for k, v in acc.items()})
