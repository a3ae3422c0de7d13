#!/usr/bin/env python3
"""Round-5 measurement (review item 3): an epoch of the PRODUCT's training loop -- cvc.trainer.Trainer.train(), the drop-in for the
reference's trainer.py:39-150: DevicePrefetcher, per-batch trimming to shape buckets, one captured HIP graph per shape, losses read
back once per display interval -- at config 3 size on synthetic clips, next to the step bench.py --mode train times
(Trainer.train_step_graphed on one static batch).  Batches of three different trimmed shapes, device-resident (the benchmark's input
contract: features in HBM).  usage: python tools/runs/r05_main_epoch.py [cfg3|cfg4] [steps=60]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cyclical-visual-captioning_amd")]
from cvc import synth, opts as cvc_opts, hip                                     # noqa: E402
from cvc.model.captioner import DecodeAndGroundCaptionerGVDROI, PrecomputedRegionFeatures   # noqa: E402
from cvc.trainer import Trainer, build_optimizer                                  # noqa: E402
from cvc.distributed import GradReducer                                           # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
d = synth.CONFIGS[cfg]
dev = torch.device("cuda:0")
hip.lib()


def make(disp):
    o = cvc_opts.parse_opt([])
    o.vocab_size, o.itow, o.wtoi = d.V, {str(i): "w%d" % i for i in range(d.V)}, {"UNK": synth.UNK_IDX}
    o.seq_length, o.rnn_size, o.input_encoding_size, o.att_hid_size = d.T, d.R, d.E, d.A
    o.detect_size, o.vis_encoding_size, o.train_decoder_only = d.DET, d.G, False
    o.xe_loss_weight, o.caption_consistency_loss_weight, o.learning_rate, o.batch_size = 0.5, 0.5, 1e-4, d.B
    o.disp_interval, o.hip_graph = disp, 1
    model = DecodeAndGroundCaptionerGVDROI(o, roi_extractor=PrecomputedRegionFeatures(d.DET, d.G))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.hot_path_state_dict(d, 1234).items()}, strict=False)
    model = model.to(dev).train()
    optim = build_optimizer(model, o, capturable=True)
    return o, model, optim, GradReducer(model.named_parameters())


def batch(seed, n_max, k_max):
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    f, b = synth.clip_features(d, seed), synth.label_glue_batch(d, seed)
    for key in ("pool_feats", "p_pool_feats", "g_pool_feats"):
        f[key][:, n_max:] = 0
    f["pnt_mask"][:, 1 + n_max:] = True
    b["box_mask"][:, 0, k_max:, :] = True
    b["num"][:, 1], b["num"][:, 2] = n_max, k_max
    f, b = {k: t(v) for k, v in f.items()}, {k: t(v) for k, v in b.items()}
    return (f, b["input_seq"], b["gt_seq"], b["num"].cpu(), b["proposals"], b["gt_bboxs"], b["box_mask"],
            ["v_x_segment_%02d" % i for i in range(d.B)], torch.zeros(d.B, d.N, 1), b["frm_mask"], b["sample_idx"], f["pnt_mask"][:, 1:].clone())


# ---- (a) the product loop: three shapes (full, 3/4 and 1/2 of the proposals), cycled
shapes = [(d.N, d.K), (3 * d.N // 4, max(1, d.K // 2)), (d.N // 2, d.K)]
batches = [batch(1234 + i, *shapes[i]) for i in range(3)]
o, model, optim, red = make(disp=20)
loader = [batches[i % 3] for i in range(steps + 1)]            # train() drops the last batch
tr = Trainer(o, None, model, optim, loader, None, grad_reducer=red)
assert tr.graph_capable()
tr.train(0)                                                     # epoch 0: eager first occurrences, captures
torch.cuda.synchronize()
t0 = time.perf_counter()
tr.train(1)
torch.cuda.synchronize()
epoch_ms = (time.perf_counter() - t0) / steps * 1e3
stats = dict(tr.graph_stats)
# per-shape replay time inside the loop (same trainer, its captured graphs)
per_shape = []
for i in range(3):
    b_ = tr._prepare(batches[i], True)
    for _ in range(3):
        tr.train_step_bucketed(b_)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        tr.train_step_bucketed(b_)
    torch.cuda.synchronize()
    per_shape.append(round((time.perf_counter() - t0) / 20 * 1e3, 3))
red.remove_hooks()
del tr, model, optim, red
torch.cuda.empty_cache()

# ---- (b) what bench.py --mode train times: train_step_graphed on one static full-shape batch
o, model, optim, red = make(disp=20)
tr = Trainer(o, None, model, optim, None, None, grad_reducer=red)
for _ in range(5):
    tr.train_step_graphed(batches[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    tr.train_step_graphed(batches[0])
torch.cuda.synchronize()
bench_ms = (time.perf_counter() - t0) / 30 * 1e3
print(f"{cfg}: Trainer.train() epoch of {steps} steps over 3 trimmed shapes {shapes}: {epoch_ms:.3f} ms per step "
      f"(graphs: {stats}); per shape inside the loop {per_shape} ms; bench-style static step (full shape) {bench_ms:.3f} ms; "
      f"loop / bench on the full shape = {per_shape[0] / bench_ms:.3f}; epoch mean / mean of its shapes = {epoch_ms / (sum(per_shape) / 3):.3f}")
