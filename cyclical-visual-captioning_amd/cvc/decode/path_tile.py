"""More than 64 live rows per step (beam search, big greedy batches): bf16 split-term fragments through the tile GEMM
(csrc/gemm_tile.hip), 12 launches per step.  Mixin of cvc.decode.engine.DecodeEngine."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch

from .. import hip
from .weights import *          # noqa: F401,F403  (packers, layouts, cache plan, switches)
from .weights import _segs


class TilePath:
    @staticmethod
    def _ksplit(n_out: int, K: int) -> int:
        """K slices of a tile GEMM: enough workgroups (128-row tiles x slices) to cover the 256 CUs, at least 8 k steps per
        slice; 4 / 8 slices preferred (they map onto whole XCDs)."""
        ntile = (n_out + 127) // 128
        ks = max(1, min(256 // ntile, (K // 16) // 8))
        for p in (8, 4, 2):
            if ks >= p and ks < 2 * p and ntile * p >= 192:
                return p
        return ks

    def _alloc_tile(self):
        """Weight fragments (once per checkpoint binding) and the activation fragment buffers of the tile path:
          XA = [h_lang(t-1) | relu(Emb[word_t]) | h_att(t-1)]       att-LSTM input, K = 2R + E
          XL = [ctx_regions + ctx_frames | h_att(t) | h_lang(t-1)]   lang-LSTM input, K = 3R (h2attn reads its middle segment)
          XH = h_lang(t)                                             vocabulary head input"""
        W, R, E, A, V = self.W, self.W.R, self.W.E, self.W.A, self.W.V
        dev = self.fc.device
        if self.embgate and not hasattr(W, "t_att2"):
            W.t_att2 = pack_weights_tile(torch.cat([W.w_ih_att[:, 0:R], W.w_hh_att], 1), R)     # K = 2R: [h_lang | h_att]
            if not hasattr(W, "t_embgate"):
                W.t_embgate = embgate_table(W)
        if not self.embgate and not hasattr(W, "t_att"):
            W.t_att = pack_weights_tile(torch.cat([W.w_ih_att[:, 0:R], W.w_ih_att[:, 2 * R:2 * R + E], W.w_hh_att], 1), R)
        if not hasattr(W, "t_lang"):
            W.t_lang = pack_weights_tile(torch.cat([W.w_ih_lang, W.w_hh_lang], 1), R)
            W.t_h = pack_weights_tile(W.w_h)
            W.t_o = pack_weights_tile(W.w_o)
            W.t_fc = pack_weights_tile(W.w_ih_att[:, R:2 * R].contiguous())
        rows, B = self.rows, self.B
        ra, rb = hip.tile_rows_alloc(rows), hip.tile_rows_alloc(B)
        zf = lambda r, k: torch.zeros(r // 32, k // 16, 3, 2, 32, 8, device=dev, dtype=torch.int16)
        self.ka_tile = 2 * R if self.embgate else 2 * R + E
        self.XAf, self.XLf, self.XHf, self.XFf = zf(ra, self.ka_tile), zf(ra, 3 * R), zf(ra, R), zf(rb, R)
        f32 = dict(device=dev, dtype=torch.float32)
        self.ks_gate, self.ks_q, self.ks_o, self.ks_fc = (self._ksplit(4 * R, min(self.ka_tile, 3 * R)), self._ksplit(A, R),
                                                          self._ksplit(V, R), self._ksplit(4 * R, R))
        self.parts_gate = torch.empty(self.ks_gate, rows, 4 * R, **f32)
        self.parts_q = torch.empty(self.ks_q, rows, A, **f32)
        self.parts_o = torch.empty(self.ks_o, rows, V, **f32)
        self.parts_fc = torch.empty(self.ks_fc, B, 4 * R, **f32)
        self.gate_fc_clip = torch.empty(B, 4 * R, **f32)
        z = lambda: torch.zeros(rows, R, **f32)
        self.t_h_att, self.t_c_att, self.t_h_lang, self.t_c_lang = z(), z(), z(), z()        # state of the current step
        self.t_c_att_prev, self.t_c_lang_prev, self.t_zero = z(), z(), z()

    def _build_tile(self):
        L, W = hip.lib(), self.W
        B, N, Fr, R, A, E, V, rows, beam = self.B, self.N, self.F, W.R, W.A, W.E, W.V, self.rows, self.beam
        fc, conv, pconv, pool, ppool = self.feats
        ptr = lambda t: None if t is None else t.data_ptr()
        fp = hip._frag_ptr
        out = []
        # ---- once per decode: hoisted fc gate term (+ both biases), one row per clip; step-0 operands from the zero state
        xf_p, xf_s = fp(self.XFf)
        out.append(("gate_fc", L.cvc_tile_pack_rows, (ptr(fc), fc.stride(0), None, 0, B, R, xf_p, xf_s)))
        out.append(("gate_fc", L.cvc_tile_gemm, (ptr(W.t_fc), xf_p, xf_s, R, B, 4 * R, self.ks_fc, ptr(self.parts_fc), 4 * R,
                                                 B * 4 * R)))
        out.append(("gate_fc", L.cvc_tile_linear_finish, (ptr(self.parts_fc), self.ks_fc, B * 4 * R, 4 * R, ptr(W.b_ih_att),
                                                          ptr(W.b_hh_att), B, 4 * R, ptr(self.gate_fc_clip), 4 * R)))
        xa_p, xa_s = fp(self.XAf)
        xl_p, xl_s = fp(self.XLf)
        xl_hatt, _ = fp(self.XLf, R)
        xl_hlang, _ = fp(self.XLf, 2 * R)
        xh_p, xh_s = fp(self.XHf)
        zero = ptr(self.t_zero)
        eg = self.embgate
        E_pack = 0 if eg else E                        # embedding-gate form: no embedding segment in XA, the word enters in the finish
        out.append(("beam_reorder", L.cvc_tile_reorder_pack, (None, ptr(self.words[0]), beam, zero, zero, zero, zero, ptr(W.embed), E_pack, V,
                                                              ptr(self.t_c_att_prev), ptr(self.t_c_lang_prev), xa_p, xa_s, xl_hlang,
                                                              xl_s, rows, R)))
        for t in range(self.T):
            out.append(("att_lstm", L.cvc_tile_gemm, (ptr(W.t_att2 if eg else W.t_att), xa_p, xa_s, self.ka_tile, rows, 4 * R, self.ks_gate,
                                                      ptr(self.parts_gate), 4 * R, rows * 4 * R)))
            if eg:
                out.append(("att_finish", L.cvc_tile_lstm_finish_embgate, (ptr(self.parts_gate), self.ks_gate, rows * 4 * R, None, None,
                                                                           ptr(self.gate_fc_clip), beam, ptr(W.t_embgate), ptr(self.words[t]), V,
                                                                           ptr(self.t_c_att_prev), rows, R, ptr(self.t_c_att),
                                                                           ptr(self.t_h_att), xl_hatt, xl_s, None, 0)))
            else:
                out.append(("att_finish", L.cvc_tile_lstm_finish, (ptr(self.parts_gate), self.ks_gate, rows * 4 * R, None, None,
                                                                   ptr(self.gate_fc_clip), beam, ptr(self.t_c_att_prev), rows, R,
                                                                   ptr(self.t_c_att), ptr(self.t_h_att), xl_hatt, xl_s, None, 0)))
            out.append(("h2attn", L.cvc_tile_gemm, (ptr(W.t_h), xl_hatt, xl_s, R, rows, A, self.ks_q, ptr(self.parts_q), A, rows * A)))
            sets = (hip.AttnSet * 2)()
            sets[0] = hip.AttnSet(ptr(ppool), ptr(pool), ptr(self.mask), None, ptr(self.scores_r), None,
                                  ptr(self.att_steps[t]), None, N, self.stream_r)
            sets[1] = hip.AttnSet(ptr(pconv), ptr(conv), None, None, ptr(self.scores_f), None, ptr(self.attn_f), None, Fr,
                                  self.stream_f)
            # the query slabs are summed ONCE here: every one of a clip's ~19 score workgroups would otherwise re-sum
            # ks_q x beam x A partials on its own (measured: 106 -> us of the score pass were that)
            out.append(("h2attn_finish", L.cvc_tile_linear_finish, (ptr(self.parts_q), self.ks_q, rows * A, A, ptr(W.b_h), None, rows, A,
                                                                    ptr(self.q), A)))
            out.append(("attn_scores", L.cvc_attn_scores, (W.kind, ptr(self.q), ptr(W.w_a), ptr(W.b_a), self.inv_temp, sets, 2, B,
                                                           beam, A)))
            out.append(("attn_wsum", L.cvc_attn_wsum_frag, (sets, 2, B, beam, R, xl_p, xl_s)))
            out.append(("lang_lstm", L.cvc_tile_gemm, (ptr(W.t_lang), xl_p, xl_s, 3 * R, rows, 4 * R, self.ks_gate,
                                                       ptr(self.parts_gate), 4 * R, rows * 4 * R)))
            out.append(("lang_finish", L.cvc_tile_lstm_finish, (ptr(self.parts_gate), self.ks_gate, rows * 4 * R, ptr(W.b_ih_lang),
                                                                ptr(W.b_hh_lang), None, 1, ptr(self.t_c_lang_prev), rows, R,
                                                                ptr(self.t_c_lang), ptr(self.t_h_lang), xh_p, xh_s, None, 0)))
            out.append(("logits", L.cvc_tile_gemm, (ptr(W.t_o), xh_p, xh_s, R, rows, V, self.ks_o, ptr(self.parts_o), V, rows * V)))
            # beams: the selection sums the K-slice slabs itself (in the finishing pass's order: same logits bit for bit), the
            # finished [rows, V] matrix is neither written nor read back; CVC_BEAM_FINISH=1 keeps the separate pass (A/B)
            fused_sel = beam > 1 and self.ks_o in (2, 4, 6, 8) and V % 4 == 0 and os.environ.get("CVC_BEAM_FINISH") != "1"
            if not fused_sel:
                out.append(("logits_finish", L.cvc_tile_linear_finish, (ptr(self.parts_o), self.ks_o, rows * V, V, ptr(W.b_o), None, rows, V,
                                                                        ptr(self.logits), V)))
            if beam == 1:
                out.append(("word_select", L.cvc_top2_unk, (ptr(self.logits), rows, V, self.unk, ptr(self.words[t + 1]), 1,
                                                            ptr(self.logprob[t]))))
                parent = None
            else:
                srd, swr = t & 1, (t + 1) & 1
                src = (ptr(self.parts_o), self.ks_o, rows * V, ptr(W.b_o)) if fused_sel else (ptr(self.logits), 1, 0, None)
                out.append(("word_select", L.cvc_beam_select_parts, src + (ptr(self.score[srd]), ptr(self.done[srd]), B,
                                                                           beam, V, self.unk, 1 if t == 0 else 0, ptr(self.parent[t]),
                                                                           ptr(self.words[t + 1]), ptr(self.score[swr]),
                                                                           ptr(self.done[swr]), ptr(self.beam_ws))))
                parent = ptr(self.parent[t])
            if t + 1 < self.T:
                out.append(("beam_reorder", L.cvc_tile_reorder_pack, (parent, ptr(self.words[t + 1]), beam, ptr(self.t_h_att),
                                                                      ptr(self.t_c_att), ptr(self.t_h_lang), ptr(self.t_c_lang),
                                                                      ptr(W.embed), E_pack, V, ptr(self.t_c_att_prev),
                                                                      ptr(self.t_c_lang_prev), xa_p, xa_s, xl_hlang, xl_s, rows, R)))
            self._keep.append(sets)
        return out

    # ------------------------------------------------------------------ launch list
