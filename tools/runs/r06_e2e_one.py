"""one process: optional decode-graph capture first (argv[1] = none | captureonly), then the end-to-end training step (eager only when
argv[2] = eager); for rocprofv3 --kernel-trace comparisons (tools/runs/r06_e2e_after_decode.py is the timing harness)"""
import gc
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cyclical-visual-captioning_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import bench_e2e
from cvc import synth

pre = sys.argv[1] if len(sys.argv) > 1 else "none"
args = bench.parse(["--no-cpu-baseline"] + (["--no-train-graph"] if "eager" in sys.argv[2:] else []))
dev = torch.device("cuda:0")
if pre != "none":
    from cvc.decode import DecodeEngine, DecodeWeights
    d = synth.CONFIGS["cfg2"]
    W = DecodeWeights({k: torch.from_numpy(v).to(dev) for k, v in synth.hot_path_state_dict(d, 1).items()})
    feats = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth.clip_features(d, 1).items()}
    eng = DecodeEngine(W, feats, d.T, synth.UNK_IDX, beam=1)
    eng.run(); torch.cuda.synchronize()
    eng.capture(); torch.cuda.synchronize()
    del eng, W, feats
    gc.collect()
l = bench_e2e.run_e2e(args, bench_e2e.dims_of("cfg2"), dev, "train", steps=6, warmup=2, config_name="cfg2", cpu_baseline=False, probe=False)
print(pre, l["ms_per_step"], l["config"]["eager_ms_per_step"])
