#!/usr/bin/env python3
"""Build-container check behind BASELINE.md section 3: the oracle's CPU throughput vs the reference's own modules on the
same synthetic config-2 workload (interleaved, best of 3 each).  Imports /root/reference: build container only."""
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "cyclical-visual-captioning_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import torch.nn.functional as F

torch.set_num_threads(os.cpu_count())
from cvc import synth
from oracle import ref_cpu as O
import make_golden as mk   # stubs tensorboardX and puts the reference on sys.path

torch.set_num_threads(os.cpu_count())

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
d = synth.CONFIGS[cfg]
sd, f = synth.hot_path_state_dict(d, 1), synth.clip_features(d, 1)
P, fo = O.to_torch(sd), O.to_torch(f)
ft = mk.feats_to_torch(f, False)
model = mk.build_reference_model(d, sd, ft)
core, embed, logit = model.decoder_core, model.embed, model.logit
mask = ft["pnt_mask"][:, 1:]


def ref_loop():
    state = (torch.zeros(2, d.B, d.R), torch.zeros(2, d.B, d.R))
    word = torch.zeros(d.B, dtype=torch.long)
    for _ in range(d.T):
        out, state, _, _, _ = core(embed(word), ft["fc_feats"], ft["conv_feats"], ft["p_conv_feats"], ft["pool_feats"],
                                   ft["p_pool_feats"], mask, state)
        top = torch.topk(F.log_softmax(logit(out), dim=1), 2, dim=1)[1]
        word = torch.where(top[:, 0] != 1, top[:, 0], top[:, 1])


with torch.no_grad():
    fns = {"reference modules": ref_loop, "oracle": lambda: O.greedy_sample(P, fo, d.T, 1)}
    best = {k: float("inf") for k in fns}
    for fn in fns.values():
        fn()
    for _ in range(5):                      # interleaved rounds: the container's neighbours are noisy
        for k, fn in fns.items():
            t0 = time.perf_counter()
            fn()
            best[k] = min(best[k], time.perf_counter() - t0)
    for k, v in best.items():
        print("%-18s %s: best of 5 %.3f s -> %.0f decode-steps/s (%d threads)" % (k, cfg, v, d.B * d.T / v, torch.get_num_threads()))
    print("oracle / reference = %.3f" % (best["reference modules"] / best["oracle"]))
