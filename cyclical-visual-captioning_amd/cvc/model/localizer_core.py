"""Cycle "localize" step -- drop-in for the reference's model/localizer_core.py: dot-product
attention over regions (masked) and frames with the embedded word as the query."""
from __future__ import annotations

import torch.nn as nn

from .modules import SoftAttention, _soft_attn_pair


class LocalizerNoLSTMCore(nn.Module):
    """reference model/localizer_core.py:7-40"""

    def __init__(self, opts):
        super().__init__()
        self.opts = opts
        self.soft_attn = SoftAttention(opts.input_encoding_size, opts.att_hid_size, temp=opts.localizer_softmax_temp)

    def forward(self, embedded_word, fc_feats, conv_feats, p_conv_feats, pool_feats, p_pool_feats, attn_mask, state,
                consistent_decoder_state, proposal_frame_mask=None, with_sentinel=False):
        # the frame-masked copy is computed by the reference and dropped (:36-37): skip it; with_sentinel reaches the region
        # attention only (:37), as -inf fill
        _, ((loc_feat, loc_prob, _), (loc_conv, _, _)) = _soft_attn_pair(
            self.soft_attn, embedded_word,
            [(p_pool_feats, pool_feats, attn_mask, None), (p_conv_feats, conv_feats, None, None)], with_sentinel=(with_sentinel, False))
        return loc_feat, loc_conv, loc_prob, state

    def forward_all_steps(self, embedded_words, conv_feats, p_conv_feats, pool_feats, p_pool_feats, attn_mask):
        """All T localizer steps at once (the loop at captioner.py:320-338 has no recurrence):
        embedded_words [B, T, E] -> loc_feat [B, T, R], loc_conv [B, T, R], prob [B, T, N]."""
        B, T, E = embedded_words.shape
        _, ((loc_feat, loc_prob, _), (loc_conv, _, _)) = _soft_attn_pair(
            self.soft_attn, embedded_words.reshape(B * T, E),
            [(p_pool_feats, pool_feats, attn_mask, None), (p_conv_feats, conv_feats, None, None)])
        return loc_feat.view(B, T, -1), loc_conv.view(B, T, -1), loc_prob.view(B, T, -1)

    def forward_all_steps_sum(self, embedded_words, conv_feats, p_conv_feats, pool_feats, p_pool_feats, attn_mask):
        """forward_all_steps returning only loc_feat + loc_conv [B, T, R] -- the one quantity the reconstruction loop reads
        (decoder_core.py:106: weighted_pool_feat + attn_conv), summed inside the attention kernel."""
        B, T, E = embedded_words.shape
        total, _sets = _soft_attn_pair(self.soft_attn, embedded_words.reshape(B * T, E),
                                       [(p_pool_feats, pool_feats, attn_mask, None), (p_conv_feats, conv_feats, None, None)])
        return total.view(B, T, -1)
