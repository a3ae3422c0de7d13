"""Model factory -- the reference's model/create_model.py:11-37.  The once-per-clip encoder is
`RegionalFeatureExtractorGVD` as in the reference whenever the data loader has put the GloVe tables on
`opts` (main.py:104-105 of the reference); runs on pre-extracted features (the benchmark's input
contract, `opts.precomputed_features`) get `PrecomputedRegionFeatures`.  `roi_extractor_factory`
injects anything else."""
from __future__ import annotations

import torch.nn as nn

from ..cycle_utils import resume_decoder_roiextractor
from .backbone import RegionalFeatureExtractorGVD
from .captioner import DecodeAndGroundCaptionerGVDROI, PrecomputedRegionFeatures
from .decoder_core import TopDownDecoderCore


def _default_extractor(opts):
    if getattr(opts, "precomputed_features", not hasattr(opts, "glove_clss")):
        return PrecomputedRegionFeatures(opts.detect_size, opts.vis_encoding_size, opts.drop_prob_lm)
    return RegionalFeatureExtractorGVD(opts)


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
