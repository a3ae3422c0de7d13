"""Soft attention modules -- drop-in for the reference's model/modules.py.

Same constructors, forward signatures, return tuples and parameter names
(`h2attn.{weight,bias}`, `alpha_net.{weight,bias}`); the arithmetic is the two-pass streaming
HIP attention of csrc/attn_fwd.hip (and its recomputing backward, csrc/attn_bwd.hip).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import functional as F_
from .. import hip


class SoftAttention(nn.Module):
    """Dot-product soft attention (reference model/modules.py:7-76)."""

    def __init__(self, rnn_hidden_size, attn_hidden_size, temp=1):
        super().__init__()
        self.h2attn = nn.Linear(rnn_hidden_size, attn_hidden_size)
        self.temp = temp
        self.min_value = -1e8   # finite: an all-masked row softmaxes to uniform, not NaN

    def forward(self, h, proj_context, context=None, mask=None, proposal_frame_mask=None, with_sentinel=False):
        """with_sentinel=True fills masked positions with -inf instead of min_value (reference modules.py:40-41; no reference
        caller uses it: decoder_core.py:55, localizer_core.py:37 pass False)."""
        q = F_.linear(h, self.h2attn.weight, self.h2attn.bias)
        src = proj_context if context is None else context
        _, ((ctx, attn, fm),) = F_.attention(hip.ATTN_DOT, q, None, None, 1.0 / float(self.temp),
                                             [(proj_context, src, mask, proposal_frame_mask)], with_sentinel=with_sentinel)
        return ctx, attn, fm


class AdditiveSoftAttention(nn.Module):
    """Additive (tanh) soft attention (reference model/modules.py:79-159).  The softmax
    temperature is stored but, as in the reference (:120), not applied."""

    def __init__(self, rnn_hidden_size, attn_hidden_size, temp=1):
        super().__init__()
        self.rnn_size = rnn_hidden_size
        self.att_hid_size = attn_hidden_size
        self.h2attn = nn.Linear(rnn_hidden_size, attn_hidden_size)
        self.alpha_net = nn.Linear(attn_hidden_size, 1)
        self.temp = temp
        self.min_value = -1e8

    def forward(self, h, proj_context, context=None, mask=None, proposal_frame_mask=None, with_sentinel=False):
        """with_sentinel=True: -inf fill (reference modules.py:123-124, 136-138)."""
        q = F_.linear(h, self.h2attn.weight, self.h2attn.bias)
        src = proj_context if context is None else context
        _, ((ctx, attn, fm),) = F_.attention(hip.ATTN_ADDITIVE, q, self.alpha_net.weight, self.alpha_net.bias, 1.0,
                                             [(proj_context, src, mask, proposal_frame_mask)], with_sentinel=with_sentinel)
        return ctx, attn, fm

    def forward_pair(self, h, sets, with_sentinel=False):
        """Both feature sets of a decoder step in one launch (decoder_core.py:54-56 calls the
        module twice with the same query): returns (ctx_sum, [(ctx, attn, fm)] per set).  with_sentinel: bool or one per set."""
        q = F_.linear(h, self.h2attn.weight, self.h2attn.bias)
        return F_.attention(hip.ATTN_ADDITIVE, q, self.alpha_net.weight, self.alpha_net.bias, 1.0, sets, with_sentinel=with_sentinel)


def _soft_attn_pair(mod, h, sets, with_sentinel=False):
    if isinstance(mod, AdditiveSoftAttention):
        return mod.forward_pair(h, sets, with_sentinel)
    q = F_.linear(h, mod.h2attn.weight, mod.h2attn.bias)
    return F_.attention(hip.ATTN_DOT, q, None, None, 1.0 / float(mod.temp), sets, with_sentinel=with_sentinel)


def proj_masking(feat, projector, mask=None):
    """Reference model/modules.py:162-176 (used by the once-per-clip encoder, kept for API
    completeness; plain torch)."""
    proj_feat = projector(feat.view(-1, feat.size(2))).view(feat.size(0), feat.size(1), -1)
    if mask is None:
        return proj_feat
    assert mask.sum() != 0
    return proj_feat * mask.unsqueeze(2).expand_as(proj_feat)
