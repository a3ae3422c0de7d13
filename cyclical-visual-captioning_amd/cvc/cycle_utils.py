"""Stage-2 warm start and small environment helpers -- the reference's cycle_utils.py.

`resume_decoder_roiextractor` reproduces the reference loader (cycle_utils.py:30-101) INCLUDING
its key-collision behaviour: every checkpoint key is reduced to the suffix after its first dotted
component and the LAST checkpoint entry with a given suffix wins, so the
stage-2 decoder's attention weights come from `attended_roi_decoder_core.soft_attn.*`.
`mode="corrected"` (new) instead matches on the full module prefix.
"""
from __future__ import annotations

import os
import pickle
import sys
from collections import OrderedDict

import torch


def is_code_development():
    return sys.platform == 'darwin'


def set_tb_logger(log_dir, exp_name, resume):
    from .misc.utils import set_tb_logger as _impl
    return _impl(log_dir, exp_name, resume)


def route_checkpoint(checkpoint, targets, mode="reference"):
    """checkpoint: full-model state_dict; targets: {"decoder_core": module_keys, "embed": ...,
    "logit": ..., "roi_feat_extractor": ...} -> {target: OrderedDict}."""
    routed = {name: OrderedDict() for name in targets}
    for key, value in checkpoint.items():
        head, _, tail = key.partition('.')
        new_key = key.split('.', 1)[-1]
        for name, keys in targets.items():
            if mode == "corrected" and head != name:
                continue
            if new_key in keys:
                routed[name][new_key] = value      # later entries overwrite earlier ones
    return routed


def resume_decoder_roiextractor(opts, exp_name, decoder, embed, logit, roi_extractor):
    file_extention = 'model-best.pth'
    info_path = os.path.join(opts.checkpoint_dir + exp_name + '/', 'infos_' + opts.id + '-best.pkl')
    resume_file_name = opts.checkpoint_dir + exp_name + '/' + file_extention
    if not os.path.isfile(resume_file_name):
        raise ValueError("=> no checkpoint found at '{}'".format(resume_file_name))

    with open(info_path, 'rb') as f:
        infos = pickle.load(f)
    opts.start_epoch = infos.get('epoch', 0)
    checkpoint = torch.load(resume_file_name, map_location='cpu')

    targets = {"decoder_core": list(decoder.state_dict().keys())}
    if opts.resume_embed:
        targets["embed"] = list(embed.state_dict().keys())
    if opts.resume_logit:
        targets["logit"] = list(logit.state_dict().keys())
    if opts.resume_roi_extractor:
        targets["roi_feat_extractor"] = list(roi_extractor.state_dict().keys())
    routed = route_checkpoint(checkpoint, targets, getattr(opts, "warm_start_mode", "reference"))

    assert set(routed["decoder_core"].keys()) == set(targets["decoder_core"])
    decoder.load_state_dict(routed["decoder_core"])
    for name, module in (("embed", embed), ("logit", logit), ("roi_feat_extractor", roi_extractor)):
        if name in targets:
            assert set(routed[name].keys()) == set(targets[name])
            print('resuming {} weights ...'.format(name))
            module.load_state_dict(routed[name])
    print("=> loaded pre-trained decoder{} from '{}' (epoch {})".format(
        " and ROI extractor" if opts.resume_roi_extractor else "", resume_file_name, opts.start_epoch))
    return decoder, embed, logit, roi_extractor
