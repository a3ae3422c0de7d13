"""Model factory -- the reference's model/create_model.py:11-37, with the once-per-clip encoder
injected (it is outside the hot path: pass `roi_extractor_factory`, default = pre-extracted
features)."""
from __future__ import annotations

import torch.nn as nn

from ..cycle_utils import resume_decoder_roiextractor
from .captioner import DecodeAndGroundCaptionerGVDROI, PrecomputedRegionFeatures
from .decoder_core import TopDownDecoderCore


def _default_extractor(opts):
    return PrecomputedRegionFeatures(opts.detect_size, opts.vis_encoding_size, opts.drop_prob_lm)


def build_model(opts, device, roi_extractor_factory=_default_extractor):
    pretrained_decoder, embed, logit = None, None, None
    roi_extractor = roi_extractor_factory(opts)
    if opts.resume_decoder_exp_name != '' and not opts.resume:
        # stage 2 of the cyclical regimen: warm-start decoder / embed / logit / encoder from stage 1
        pretrained_decoder = TopDownDecoderCore(opts)
        rows = opts.vocab_size + 1 if opts.embedding_vocab_plus_1 else opts.vocab_size
        embed = nn.Sequential(nn.Embedding(rows, opts.input_encoding_size), nn.ReLU(), nn.Dropout(opts.drop_prob_lm))
        logit = nn.Linear(opts.rnn_size, rows)
        pretrained_decoder, embed, logit, roi_extractor = resume_decoder_roiextractor(
            opts, opts.resume_decoder_exp_name, pretrained_decoder, embed, logit, roi_extractor)
    return DecodeAndGroundCaptionerGVDROI(opts, pretrained_decoder, embed, logit, roi_extractor).to(device)
