"""Top-down two-LSTM decoder step and the reconstructor step -- drop-in for the reference's
model/decoder_core.py (same ctor/forward signatures and parameter names, including the dead
i2h_2 / h2h_2 / localied_fc parameters that reference checkpoints carry)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import functional as F_
from .modules import AdditiveSoftAttention, SoftAttention, _soft_attn_pair


def _cell(mod: nn.LSTMCell):
    return mod.weight_ih, mod.weight_hh, mod.bias_ih, mod.bias_hh


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
        if with_sentinel:
            raise NotImplementedError("with_sentinel=True is not part of the caption-decode hot path")
        prev_h = state[0][-1]                      # the language LSTM's h (:43)
        xs = [prev_h, fc_feats, embedded_word] if self.opts.global_img_in_attn_lstm else [prev_h, embedded_word]
        h_attn, c_attn = F_.lstm_cell(xs, state[0][0], state[1][0], *_cell(self.att_lstm))
        # regions (masked, optional frame-masked copy) and frames share the query: one launch
        ctx_sum, ((weighted_pool_feat, roi_attn, frame_masked_attn), _frames) = _soft_attn_pair(
            self.soft_attn, h_attn,
            [(p_pool_feats, pool_feats, pnt_mask, proposal_frame_mask), (p_conv_feats, conv_feats, None, None)])
        h_lang, c_lang = F_.lstm_cell([ctx_sum, h_attn], state[0][1], state[1][1], *_cell(self.lang_lstm))
        output = self.dropout(h_lang)
        state = (torch.stack([h_attn, h_lang]), torch.stack([c_attn, c_lang]))
        return output, state, roi_attn, frame_masked_attn, weighted_pool_feat


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
        prev_h = state[0][-1]
        xs = [prev_h, fc_feats, embedded_word] if self.opts.global_img_in_attn_lstm else [prev_h, embedded_word]
        h_attn, c_attn = F_.lstm_cell(xs, state[0][0], state[1][0], *_cell(self.att_lstm))
        h_lang, c_lang = F_.lstm_cell([weighted_pool_feat + attn_conv, h_attn], state[0][1], state[1][1],
                                      *_cell(self.lang_lstm))
        output = self.dropout(h_lang)
        state = (torch.stack([h_attn, h_lang]), torch.stack([c_attn, c_lang]))
        return output, state
