#!/usr/bin/env python3
"""Where do the small ATen kernels of one cyclical training step come from?  Runs one eager step at a
reduced config under torch.profiler (CPU-side op records with Python stacks) and prints, per ATen op,
the call counts grouped by the innermost frame inside this repo.   (GPU box)
  python tools/trace_small_ops.py [op ...]          default ops: fill_ zeros add cat copy_ zero_
"""
import collections
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cyclical-visual-captioning_amd"))
import bench  # noqa: E402


def main():
    ops = sys.argv[1:] or ["aten::fill_", "aten::zeros", "aten::zeros_like", "aten::add", "aten::add_", "aten::cat",
                           "aten::copy_", "aten::zero_", "aten::sum", "aten::mul", "aten::clone", "aten::sort", "aten::argsort",
                           "aten::_to_copy", "aten::where", "aten::clamp", "aten::index", "aten::gt", "aten::eq", "aten::ne", "aten::div",
                           "aten::neg", "aten::mean", "aten::stack", "aten::sub", "aten::arange", "aten::masked_fill", "aten::ones_like",
                           "aten::bitwise_or", "aten::bitwise_not", "aten::ge", "aten::index_put_", "aten::embedding",
                           "aten::embedding_dense_backward", "aten::relu", "aten::threshold_backward", "aten::mul_", "aten::div_"]
    from cvc import synth, opts as cvc_opts
    from cvc.model.captioner import DecodeAndGroundCaptionerGVDROI, PrecomputedRegionFeatures
    from cvc.trainer import Trainer, build_optimizer
    dev = torch.device("cuda:0")
    d = synth.CONFIGS["cfg1"]
    o = cvc_opts.parse_opt([])
    o.vocab_size, o.itow, o.wtoi = d.V, {str(i): "w%d" % i for i in range(d.V)}, {"UNK": synth.UNK_IDX}
    o.seq_length, o.rnn_size, o.input_encoding_size, o.att_hid_size = d.T, d.R, d.E, d.A
    o.detect_size, o.vis_encoding_size, o.train_decoder_only = d.DET, d.G, False
    o.xe_loss_weight, o.caption_consistency_loss_weight, o.learning_rate, o.batch_size = 0.5, 0.5, 1e-4, d.B
    model = DecodeAndGroundCaptionerGVDROI(o, roi_extractor=PrecomputedRegionFeatures(d.DET, d.G))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.hot_path_state_dict(d, 1).items()}, strict=False)
    model = model.to(dev).train()
    from cvc.distributed import GradReducer
    tr = Trainer(o, None, model, build_optimizer(model, o), None, None, grad_reducer=GradReducer(model.named_parameters()))
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    feats = {k: t(v) for k, v in synth.clip_features(d, 1).items()}
    b = {k: t(v) for k, v in synth.label_glue_batch(d, 1).items()}
    batch = (feats, b["input_seq"], b["gt_seq"], b["num"].cpu(), b["proposals"], b["gt_bboxs"], b["box_mask"],
             ["v_x_segment_%02d" % i for i in range(d.B)], torch.zeros(d.B, d.N, 1), b["frm_mask"], b["sample_idx"],
             feats["pnt_mask"][:, 1:])
    tr.train_step(batch)
    torch.cuda.synchronize()
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    by = {op: collections.Counter() for op in ops}

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = "aten::" + func.__name__.split(".")[0] if not str(func).startswith("aten::") else str(func)
            name = "aten::" + str(func).split(".")[1] if str(func).startswith("aten.") else name
            if name in by:
                where = "<autograd engine / no repo frame>"
                for fr in reversed(traceback.extract_stack()):
                    if "cyclical-visual-captioning_amd" in fr.filename:
                        where = "%s:%d %s" % (fr.filename.split("cyclical-visual-captioning_amd/")[-1], fr.lineno, fr.name)
                        break
                by[name][where] += 1
            return func(*args, **(kwargs or {}))

    with Spy():
        tr.train_step(batch)
        torch.cuda.synchronize()
    for op in ops:
        tot = sum(by[op].values())
        print("== %s: %d calls (T=%d)" % (op, tot, d.T))
        for w, n in by[op].most_common(12):
            print("   %5d  %s" % (n, w))


if __name__ == "__main__":
    main()
