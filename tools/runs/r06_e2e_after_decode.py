"""Diagnostic (round 6): why does the end-to-end training step measure 119 ms after the headline decode measurement in one process and
91 ms in a fresh one?  Runs bench_e2e's step after increasing pieces of run_decode, each in a fresh interpreter."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r"""
import sys, os, gc
sys.path.insert(0, "cyclical-visual-captioning_amd"); sys.path.insert(0, ".")
import numpy as np, torch, bench, bench_e2e
from cvc import synth, hip
args = bench.parse(["--no-cpu-baseline"])
dev = torch.device("cuda:0")
pre = os.environ.get("PRE", "none")
if pre.startswith("streams"):         # `streamsN`: N torch streams taken from the pool before the trainer takes its own
    KEEP = [torch.cuda.Stream() for _ in range(int(pre[7:]))]
    pre_run = False
else:
    pre_run = pre != "none"
if pre_run:
    from cvc.decode import DecodeEngine, DecodeWeights
    d = synth.CONFIGS["cfg2"]
    W = DecodeWeights({k: torch.from_numpy(v).to(dev) for k, v in synth.hot_path_state_dict(d, 1).items()})
    feats = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth.clip_features(d, 1).items()}
    eng = DecodeEngine(W, feats, d.T, synth.UNK_IDX, beam=1)
    eng.run(); torch.cuda.synchronize()
    if pre == "sidestream":
        s_ = torch.cuda.Stream(); s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_): eng._run_once()
        torch.cuda.current_stream().wait_stream(s_); torch.cuda.synchronize()
    if pre == "captureonly":
        eng.capture(); torch.cuda.synchronize()
    if pre == "capture1":
        eng.capture(); eng.run(); torch.cuda.synchronize()
    if pre == "dummygraph":            # any graph at all: one tiny torch op captured and replayed
        x = torch.zeros(1024, device=dev); g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_): x.add_(1.0)
        g_.replay(); torch.cuda.synchronize()
    if pre == "dummydestroy":          # ... and destroyed before the measurement
        x = torch.zeros(1024, device=dev); g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_): x.add_(1.0)
        g_.replay(); torch.cuda.synchronize()
        del g_; gc.collect()
    if pre == "capturekeep":
        eng.capture(); eng.run(); torch.cuda.synchronize()
        KEEP = eng
    if pre in ("capture", "timed", "all"):
        eng.capture()
        for _ in range(50): eng.run()
        torch.cuda.synchronize()
    if pre in ("timed", "all"):
        eng.run_timed(); torch.cuda.synchronize()
    if pre == "emptycache":
        del eng, W, feats
        gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    elif pre in ("keep", "capturekeep"):
        pass
    else:
        del eng, W, feats
        gc.collect()
    if pre == "all":
        torch.cuda.empty_cache()
if os.environ.get("NOPERSIST"):
    from cvc import gru
    gru.PERSISTENT = gru.BWD_PERSISTENT = False
l = bench_e2e.run_e2e(args, bench_e2e.dims_of("cfg2"), dev, "train", steps=5, warmup=2, config_name="cfg2", cpu_baseline=False, probe=False)
print(pre, l["ms_per_step"], l["config"]["eager_ms_per_step"], round(torch.cuda.memory_reserved() / 2**30, 1), "GiB reserved")
"""
for pre in sys.argv[1:] or ("none", "build", "capture", "timed", "all", "keep"):
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, PRE=pre), capture_output=True, text=True, cwd=ROOT)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-800:], flush=True)
