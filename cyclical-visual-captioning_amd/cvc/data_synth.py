"""Synthetic stand-in for the ActivityNet-Entities loader (reference misc/dataloader_anet.py, out of scope: the
216 GB dataset is not available).  Items follow the reference's 12-tuple batch contract
with one difference by default: position 0 carries the clip's PRE-EXTRACTED features (a dict) instead of raw frame
features (the benchmark's input contract).  `raw=True` yields the reference's raw inputs instead -- frame features
[F, 3072] at position 0 and region features [N, att_feat_size] at position 8 -- for runs through the mirrored
once-per-clip encoder (cvc/model/backbone.py), together with the GloVe / Detectron tables its constructor needs."""
from __future__ import annotations

import dataclasses

import numpy as np
import torch
from torch.utils.data import Dataset

from . import synth


class SyntheticCaptionDataset(Dataset):
    def __init__(self, dims: synth.Dims, n_clips: int, seed: int = 0, split: str = "training", raw: bool = False):
        self.d = dataclasses.replace(dims, B=n_clips)
        self.split = split
        self.raw = raw
        if raw:
            self.glue = synth.encoder_inputs(self.d, seed)
            self.feats = {"pnt_mask": self.glue["pnt_mask_in"]}
            self.tables = synth.detectron_tables(self.d, seed)
            self.glove_clss, self.glove_vg_cls = self.tables["glove_clss"], self.tables["glove_vg_cls"]
            self.vg_cls = ["vg%d" % i for i in range(self.glove_vg_cls.shape[0])]
            self.detect_size = dims.DET
        else:
            self.feats = synth.clip_features(self.d, seed)
            self.glue = synth.label_glue_batch(self.d, seed)
        self.itow = {str(i): "w%d" % i for i in range(dims.V)}
        self.wtoi = {"UNK": synth.UNK_IDX}
        self.itod = {i: "cls%d" % i for i in range(1, dims.DET + 1)}
        self.ltow, self.itoc, self.wtod = {}, {}, {}
        self.vocab_size = dims.V
        # segment timestamps in the layout of the ANet-Entities annotation file the reference's eval reads
        # (opts.grd_reference -> ['annotations'][video]['segments'][segment]['timestamps'], trainer.py:162, 259-260)
        self.grd_reference = {"annotations": {
            "v_synth%05d" % i: {"segments": {"0": {"timestamps": [round(1.5 * i, 3), round(1.5 * i + 7.25, 3)]}}}
            for i in range(n_clips)}}

    def __len__(self):
        return self.d.B

    def __getitem__(self, i):
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x))
        g = self.glue
        if self.raw:
            f, region = t(g["segs_feat"][i]), t(g["region_feats"][i])
        else:
            f, region = {k: t(v[i]) for k, v in self.feats.items()}, torch.zeros(self.d.N, 1)
        return (f, t(g["input_seq"][i]), t(g["gt_seq"][i]), t(g["num"][i]), t(g["proposals"][i]), t(g["gt_bboxs"][i]),
                t(g["box_mask"][i]), "v_synth%05d_segment_%02d" % (i, 0), region, t(g["frm_mask"][i]),
                t(g["sample_idx"][i]), t(self.feats["pnt_mask"][i][1:]))


def collate(items):
    """default_collate for the tensors, dict-of-stacks for the feature dict, list for the ids."""
    cols = list(zip(*items))
    out = []
    for j, col in enumerate(cols):
        if isinstance(col[0], dict):
            out.append({k: torch.stack([c[k] for c in col]) for k in col[0]})
        elif isinstance(col[0], str):
            out.append(list(col))
        else:
            out.append(torch.stack(col))
    return tuple(out)
