"""Top-down two-LSTM decoder step and the reconstructor step -- drop-in for the reference's
model/decoder_core.py (same ctor/forward signatures and parameter names, including the dead
i2h_2 / h2h_2 / localied_fc parameters that reference checkpoints carry)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import dropout
from .. import functional as F_
from .modules import AdditiveSoftAttention, SoftAttention, _soft_attn_pair


def _cell(mod: nn.LSTMCell):
    return mod.weight_ih, mod.weight_hh, mod.bias_ih, mod.bias_hh


import os as _os

H_COPIES = _os.environ.get("CVC_H_COPIES", "1") != "0"   # False: one h' tensor per cell, autograd sums its consumers' gradients (A/B)


def _fused_drop(module: nn.Dropout, site, like):
    """(generator state, site id, p) when this step's output dropout runs inside the language cell's kernel, else None"""
    if site is None or not module.training or not (0 < module.p < 1) or not dropout.in_kernel(like):
        return None
    return dropout.rng_state(like.device), dropout.site_id(site), float(module.p)


class TopDownDecoderCore(nn.Module):
    """reference model/decoder_core.py:8-66"""

    def __init__(self, opts):
        super().__init__()
        self.opts = opts
        self.att_lstm = nn.LSTMCell(opts.input_encoding_size + opts.rnn_size * 2, opts.rnn_size)
        # never used in forward, kept because checkpoints list them
        self.i2h_2 = nn.Linear(opts.rnn_size * 2, opts.rnn_size)
        self.h2h_2 = nn.Linear(opts.rnn_size, opts.rnn_size)
        self.localied_fc = nn.Linear(opts.rnn_size, opts.att_hid_size)
        if opts.softattn_type == 'additive':
            self.soft_attn = AdditiveSoftAttention(opts.rnn_size, opts.att_hid_size, temp=opts.softmax_temp)
        else:
            self.soft_attn = SoftAttention(opts.rnn_size, opts.att_hid_size, temp=opts.softmax_temp)
        self.lang_lstm = nn.LSTMCell(opts.rnn_size * 2, opts.rnn_size)
        self.dropout = nn.Dropout(opts.drop_prob_lm)

    def forward(self, embedded_word, fc_feats, conv_feats, p_conv_feats, pool_feats, p_pool_feats, pnt_mask,
                state, proposal_frame_mask=None, with_sentinel=False):
        """The reference's call: state = (h [2, B, R], c [2, B, R]) in and out (layer 0 = attention LSTM, 1 = language LSTM)."""
        (h_att, h_lang), (c_att, c_lang) = state[0].unbind(0), state[1].unbind(0)
        output, st, roi_attn, frame_masked_attn, weighted_pool_feat = self.step(
            embedded_word, fc_feats, conv_feats, p_conv_feats, pool_feats, p_pool_feats, pnt_mask, (h_att, c_att, h_lang, c_lang),
            proposal_frame_mask, with_sentinel=with_sentinel)
        h_att, c_att, h_lang, c_lang = st[:4]
        return output, (torch.stack([h_att, h_lang]), torch.stack([c_att, c_lang])), roi_attn, frame_masked_attn, weighted_pool_feat

    def step(self, embedded_word, fc_feats, conv_feats, p_conv_feats, pool_feats, p_pool_feats, pnt_mask, state,
             proposal_frame_mask=None, drop_site=None, with_sentinel=False, gate_pre_att=None):
        """forward() on an unstacked state (h_att, c_att, h_lang, c_lang): what the captioner's T-step loops call --
        stacking the state every step costs ~17 small kernels per step in the backward pass (the selects' zero-filled
        gradients, their accumulation, the stack / unbind pairs)."""
        # h' of either cell has three consumers (this step's other cell / attention query / output, and the next step twice): under
        # autograd every consumer gets its own copy of the tensor (written by the cell's kernel), so that the backward pass
        # has no fan-out to accumulate; the state then carries h_lang twice (5 entries)
        h_att, c_att, h_lang, c_lang = state[:4]
        h_lang_s = state[4] if len(state) > 4 else h_lang
        k = 3 if (torch.is_grad_enabled() and H_COPIES) else 1
        xs = [h_lang, fc_feats, embedded_word] if self.opts.global_img_in_attn_lstm else [h_lang, embedded_word]
        # gate_pre_att: this step's slice of F_.hoisted_gates over (fc_feats, embedded word) -- multiplied for all T steps before
        # the loop, so the per-step launch streams only the h_lang columns of weight_ih and weight_hh
        *h_attn, c_attn = F_.lstm_cell(xs, h_att, c_att, *_cell(self.att_lstm), copies=k, gate_pre=gate_pre_att,
                                       hoisted=None if gate_pre_att is None else (False,) + (True,) * (len(xs) - 1))
        # regions (masked, optional frame-masked copy) and frames share the query: one launch
        ctx_sum, ((weighted_pool_feat, roi_attn, frame_masked_attn), _frames) = _soft_attn_pair(
            self.soft_attn, h_attn[0],
            [(p_pool_feats, pool_feats, pnt_mask, proposal_frame_mask), (p_conv_feats, conv_feats, None, None)],
            with_sentinel=(with_sentinel, False))          # the region set only, as the reference forwards it (decoder_core.py:55-56)
        drop = _fused_drop(self.dropout, drop_site, h_lang_s)
        if drop is not None:
            # output = dropout(h_lang) written by the cell's own kernel, mask generated there (cvc/dropout.py)
            *h_new, output, c_lang = F_.lstm_cell([ctx_sum, h_attn[1 % k]], h_lang_s, c_lang, *_cell(self.lang_lstm), copies=min(k, 2),
                                                  drop=drop)
            return output, (h_attn[2 % k], c_attn, h_new[0], c_lang, h_new[-1]), roi_attn, frame_masked_attn, weighted_pool_feat
        *h_new, c_lang = F_.lstm_cell([ctx_sum, h_attn[1 % k]], h_lang_s, c_lang, *_cell(self.lang_lstm), copies=k)
        return (dropout.apply(self.dropout, h_new[0], drop_site), (h_attn[2 % k], c_attn, h_new[1 % k], c_lang, h_new[2 % k]), roi_attn,
                frame_masked_attn, weighted_pool_feat)


class AttenedDecoderCore(nn.Module):
    """reference model/decoder_core.py:69-113: the reconstructor; shares the decoder's LSTM cells."""

    def __init__(self, opts, att_lstm, lang_lstm):
        super().__init__()
        self.opts = opts
        self.att_lstm = att_lstm
        if opts.softattn_type == 'additive':
            self.soft_attn = AdditiveSoftAttention(opts.rnn_size, opts.att_hid_size, temp=opts.softmax_temp)
        else:
            self.soft_attn = SoftAttention(opts.rnn_size, opts.att_hid_size, temp=opts.softmax_temp)
        self.lang_lstm = lang_lstm
        self.dropout = nn.Dropout(opts.drop_prob_lm)

    def forward(self, embedded_word, fc_feats, weighted_pool_feat, attn_conv, state, with_sentinel=False):
        (h_att, h_lang), (c_att, c_lang) = state[0].unbind(0), state[1].unbind(0)
        output, st = self.step(embedded_word, fc_feats, weighted_pool_feat, attn_conv, (h_att, c_att, h_lang, c_lang))
        h_att, c_att, h_lang, c_lang = st[:4]
        return output, (torch.stack([h_att, h_lang]), torch.stack([c_att, c_lang]))

    def step(self, embedded_word, fc_feats, weighted_pool_feat, attn_conv, state, drop_site=None, gate_pre_att=None, gate_pre_lang=None,
             ctx_sum=None):
        """forward() on an unstacked state (h_att, c_att, h_lang, c_lang), see TopDownDecoderCore.step"""
        h_att, c_att, h_lang, c_lang = state[:4]
        h_lang_s = state[4] if len(state) > 4 else h_lang
        grad = torch.is_grad_enabled() and H_COPIES
        xs = [h_lang, fc_feats, embedded_word] if self.opts.global_img_in_attn_lstm else [h_lang, embedded_word]
        *h_attn, c_attn = F_.lstm_cell(xs, h_att, c_att, *_cell(self.att_lstm), copies=2 if grad else 1, gate_pre=gate_pre_att,
                                       hoisted=None if gate_pre_att is None else (False,) + (True,) * (len(xs) - 1))
        # ctx_sum: weighted_pool_feat + attn_conv of this step, taken from the sum over all T steps the caller formed once;
        # gate_pre_lang: its hoisted product with weight_ih[:, :R] (the localized context does not depend on this loop's recurrence)
        ctx = ctx_sum if ctx_sum is not None else weighted_pool_feat + attn_conv
        drop = _fused_drop(self.dropout, drop_site, h_lang_s)
        if drop is not None:
            *h_new, output, c_lang = F_.lstm_cell([ctx, h_attn[0]], h_lang_s, c_lang, *_cell(self.lang_lstm), copies=2 if grad else 1,
                                                  gate_pre=gate_pre_lang, hoisted=None if gate_pre_lang is None else (True, False),
                                                  drop=drop)
            return output, (h_attn[-1], c_attn, h_new[0], c_lang, h_new[-1])
        *h_new, c_lang = F_.lstm_cell([ctx, h_attn[0]], h_lang_s, c_lang, *_cell(self.lang_lstm), copies=3 if grad else 1,
                                      gate_pre=gate_pre_lang, hoisted=None if gate_pre_lang is None else (True, False))
        return dropout.apply(self.dropout, h_new[0], drop_site), (h_attn[-1], c_attn, h_new[1 % len(h_new)], c_lang, h_new[2 % len(h_new)])
