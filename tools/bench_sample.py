#!/usr/bin/env python3
"""End-to-end cost of the module-level greedy decode (model(..., lang_eval=True) -> captioner._sample), i.e. what an
evaluation loop pays per batch, next to the bare DecodeEngine replay that bench.py times.  GPU box."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cyclical-visual-captioning_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from cvc import synth
    from helpers import build_model, to_dev, model_call
    dev = torch.device("cuda:0")
    d = synth.CONFIGS["cfg2"]
    for graph in (True, False):
        model = build_model(d, synth.hot_path_state_dict(d, 1), dev, hip_graph=graph)
        feats = to_dev(synth.clip_features(d, 1), dev)
        batch = to_dev(synth.label_glue_batch(d, 1), dev)
        with torch.no_grad():
            for _ in range(3):
                model_call(model, feats, batch, True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                seq, att, _ = model_call(model, feats, batch, True)
            torch.cuda.synchronize()
        print("hip_graph=%s: %.2f ms per model(..., lang_eval=True) call (B=%d, T=%d)" % (graph, (time.perf_counter() - t0) / n * 1e3, d.B, d.T))


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def rebind():
    """Cost of binding a new batch SHAPE (new engine + plan + graph capture) -- what an evaluation loop pays when the
    number of proposals changes -- and of a decode through the C-ABI driver without a graph."""
    from cvc import synth
    from helpers import to_dev
    from cvc.decode import DecodeEngine, DecodeWeights
    dev = torch.device("cuda:0")
    d = synth.CONFIGS["cfg2"]
    W = DecodeWeights(to_dev(synth.hot_path_state_dict(d, 1), dev))
    f = to_dev(synth.clip_features(d, 1), dev)
    DecodeEngine(W, f, d.T, synth.UNK_IDX).capture().run()              # packs the weights once, warms everything
    torch.cuda.synchronize()
    for driver in (True, False):
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            e = DecodeEngine(W, f, d.T, synth.UNK_IDX, driver=driver)
            t1 = time.perf_counter()
            e.capture()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            ts.append((t1 - t0, t2 - t1))
        b, c = min(t[0] for t in ts) * 1e3, min(t[1] for t in ts) * 1e3
        e = DecodeEngine(W, f, d.T, synth.UNK_IDX, driver=driver)
        e.run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            e.run()
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 10 * 1e3
        print("driver=%s: bind %.2f ms, warm-up + graph capture %.2f ms, eager (no graph) decode %.2f ms" % (driver, b, c, eager))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "rebind":
    rebind()
