"""Row-major fallback for widths the packed / tile layouts do not take (odd R / A / E): checkpoint-layout weights through the
LDS-DMA ring kernels (csrc/gemm_skinny.hip).  Mixin of cvc.decode.engine.DecodeEngine."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch

from .. import hip
from .weights import *          # noqa: F401,F403  (packers, layouts, cache plan, switches)
from .weights import _segs


class RingPath:
    def _build(self):
        L, W = hip.lib(), self.W
        B, N, Fr, R, A, E, V, rows, beam = self.B, self.N, self.F, W.R, W.A, W.E, W.V, self.rows, self.beam
        fc, conv, pconv, pool, ppool = self.feats
        ptr = lambda t: None if t is None else t.data_ptr()
        out = []
        # fc_feats does not change over the T steps (decoder_core.py:46): its gate contribution and the
        # two bias vectors are computed once per decode, inside the timed/captured region
        seg_fc = _segs([(fc, self.clip_of_row if beam > 1 else None, W.w_ih_att[:, R:2 * R], False)])
        out.append(("gate_fc", L.cvc_linear_fwd, (seg_fc, 1, ptr(W.b_ih_att), ptr(W.b_hh_att), rows, 4 * R, ptr(self.gate_fc),
                                                  4 * R)))
        self._keep.append(seg_fc)
        if beam == 1 and rows <= 64:
            out.append(("embed_bos", L.cvc_embed_relu_fwd, (ptr(W.embed), ptr(self.words[0]), None, rows, E, ptr(self.emb))))
        for t in range(self.T):
            rd, wr = t & 1, (t + 1) & 1
            # att-LSTM: [h_lang(t-1) | relu(Emb[word_t])] x W_ih  +  h_att(t-1) x W_hh  + gate_fc
            fused_head = beam == 1 and rows <= 64
            seg_att = _segs([(self.h_lang[rd], None, W.w_ih_att[:, 0:R], False),
                             (self.emb, None, W.w_ih_att[:, 2 * R:2 * R + E], False) if fused_head else
                             (W.embed, self.words[t], W.w_ih_att[:, 2 * R:2 * R + E], True),
                             (self.h_att[rd], None, W.w_hh_att, False)])
            out.append(("att_lstm", L.cvc_lstm_cell_fwd, (seg_att, 3, None, None, ptr(self.gate_fc), ptr(self.c_att[rd]),
                                                          rows, R, ptr(self.h_att[wr]), ptr(self.c_att[wr]), None)))
            seg_q = _segs([(self.h_att[wr], None, W.w_h, False)])
            split_q = rows <= 64
            if split_q:
                out.append(("h2attn", L.cvc_linear_splitk_fwd, (seg_q, 1, None, rows, A, self.QSPLIT, ptr(self.q_parts))))
            else:
                out.append(("h2attn", L.cvc_linear_fwd, (seg_q, 1, ptr(W.b_h), None, rows, A, ptr(self.q), A)))
            sets = (hip.AttnSet * 2)()
            sets[0] = hip.AttnSet(ptr(ppool), ptr(pool), ptr(self.mask), None, ptr(self.scores_r), None,
                                  ptr(self.att_steps[t]), None, N, self.stream_r)
            sets[1] = hip.AttnSet(ptr(pconv), ptr(conv), None, None, ptr(self.scores_f), None, ptr(self.attn_f), None, Fr,
                                  self.stream_f)
            if split_q:
                out.append(("attn_scores", L.cvc_attn_scores_qparts, (W.kind, ptr(self.q_parts), self.QSPLIT, ptr(W.b_h),
                                                                      ptr(W.w_a), ptr(W.b_a), self.inv_temp, sets, 2, B, beam, A)))
            else:
                out.append(("attn_scores", L.cvc_attn_scores, (W.kind, ptr(self.q), ptr(W.w_a), ptr(W.b_a), self.inv_temp, sets,
                                                               2, B, beam, A)))
            out.append(("attn_wsum", L.cvc_attn_wsum, (sets, 2, B, beam, R, ptr(self.ctx_sum))))
            seg_lang = _segs([(self.ctx_sum, None, W.w_ih_lang[:, 0:R], False),
                              (self.h_att[wr], None, W.w_ih_lang[:, R:2 * R], False),
                              (self.h_lang[rd], None, W.w_hh_lang, False)])
            out.append(("lang_lstm", L.cvc_lstm_cell_fwd, (seg_lang, 3, ptr(W.b_ih_lang), ptr(W.b_hh_lang), None,
                                                           ptr(self.c_lang[rd]), rows, R, ptr(self.h_lang[wr]),
                                                           ptr(self.c_lang[wr]), None)))
            seg_o = _segs([(self.h_lang[wr], None, W.w_o, False)])
            if fused_head:
                # vocabulary projection with the top-2 / log-sum-exp partials in its epilogue ([B,V] logits are
                # never written), then merge + UNK rule + next step's embedded word
                out.append(("logits", L.cvc_linear_top2_fwd, (seg_o, 1, ptr(W.b_o), rows, V, None, ptr(self.top2_part))))
                out.append(("word_select", L.cvc_top2_final, (ptr(self.top2_part), (V + 31) // 32, rows, self.unk,
                                                              ptr(self.words[t + 1]), 1, ptr(self.logprob[t]), ptr(W.embed), E,
                                                              ptr(self.emb), E)))
            elif beam == 1:
                out.append(("logits", L.cvc_linear_fwd, (seg_o, 1, ptr(W.b_o), None, rows, V, ptr(self.logits), V)))
                out.append(("word_select", L.cvc_top2_unk, (ptr(self.logits), rows, V, self.unk, ptr(self.words[t + 1]), 1,
                                                            ptr(self.logprob[t]))))
            else:
                out.append(("logits", L.cvc_linear_fwd, (seg_o, 1, ptr(W.b_o), None, rows, V, ptr(self.logits), V)))
                srd, swr = t & 1, (t + 1) & 1
                out.append(("word_select", L.cvc_beam_select, (ptr(self.logits), ptr(self.score[srd]), ptr(self.done[srd]), B,
                                                               beam, V, self.unk, 1 if t == 0 else 0, ptr(self.parent[t]),
                                                               ptr(self.words[t + 1]), ptr(self.score[swr]),
                                                               ptr(self.done[swr]), ptr(self.beam_ws))))
                # reorder the freshly written state rows by parent (gather into tmp, copy back)
                for i, buf in enumerate((self.h_att[wr], self.c_att[wr], self.h_lang[wr], self.c_lang[wr])):
                    out.append(("beam_reorder", L.cvc_gather_rows, (ptr(buf), ptr(self.parent[t]), rows, beam, R,
                                                                    ptr(self.gather_tmp[i]))))
                    out.append(("beam_reorder", "copy", (buf, self.gather_tmp[i])))
            self._keep += [seg_att, seg_q, sets, seg_lang, seg_o]
        return out
