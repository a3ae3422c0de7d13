"""Greedy decode with at most 64 rows: both operands of every GEMM MFMA-fragment-native (csrc/gemm_packed.hip), 7 launches per
step; buffers + the eager launch list of the default schedule and of the embedding-gate schedule (the C driver csrc/decode_driver.hip
enqueues the same list from a bound descriptor).  Mixin of cvc.decode.engine.DecodeEngine."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch

from .. import hip
from .weights import *          # noqa: F401,F403  (packers, layouts, cache plan, switches)
from .weights import _segs


class PackedPath:
    def _alloc_packed(self):
        """Fragment-native operands: packed weight copies (once per checkpoint binding) and the
        recurrent activations as two ping-pong concat buffers in quad layout:
          XA = [h_lang(t-1) | relu(Emb[word_t]) | h_att(t-1)]   (att-LSTM input, K = 2R + E)
          XL = [ctx_regions + ctx_frames | h_att(t) | h_lang(t-1)]  (lang-LSTM input, K = 3R)"""
        W, R, E = self.W, self.W.R, self.W.E
        dev = self.fc.device
        # derived copies are built per schedule, on its first use with this checkpoint binding (the default embedding-gate schedule
        # never builds the full-K attention-cell pack: 168 MB at cfg2)
        if not self.embgate and not hasattr(W, "p_att"):
            W.p_att = pack_weights(torch.cat([W.w_ih_att[:, 0:R], W.w_ih_att[:, 2 * R:2 * R + E], W.w_hh_att], 1), R)
        if not hasattr(W, "p_lang"):
            W.p_lang = pack_weights(torch.cat([W.w_ih_lang, W.w_hh_lang], 1), R)
            W.p_h = pack_weights(W.w_h)
            W.p_o = pack_weights(W.w_o)
        if self.embgate and not hasattr(W, "p_att2"):
            W.p_att2 = pack_weights(torch.cat([W.w_ih_att[:, 0:R], W.w_hh_att], 1), R)        # K = 2R: [h_lang | h_att]
            W.t_embgate = embgate_table(W)
        zq = lambda k: torch.zeros(k // 4, 64, 4, device=dev, dtype=torch.float32)
        ka = 2 * R if self.embgate else 2 * R + E
        self.XA, self.XL = [zq(ka), zq(ka)], [zq(3 * R), zq(3 * R)]
        self.sel_counter = torch.zeros(4, device=dev, dtype=torch.int32)
        self.cA, self.cL = [zq(R), zq(R)], [zq(R), zq(R)]
        bos = torch.relu(W.embed[0]).view(1, E).expand(self.rows, E).contiguous()
        self.XA0_init = zq(ka)
        if not self.embgate:
            self.XA0_init[R // 4:(R + E) // 4] = to_quad(bos)
        L = hip.lib()
        self.ks_att = int(L.cvc_packed_lstm_ks_slices(2 * R + E, R)) if self.gate_ksplit else 0
        self.ks_lang = int(L.cvc_packed_lstm_ks_slices(3 * R, R)) if self.gate_ksplit else 0
        self.ks_pad = KS_PAD_QUADS if (self.ks_att or self.ks_lang) else 0
        if self.ks_pad and not hasattr(W, "p_att_ks"):
            W.p_att_ks = pack_weights(torch.cat([W.w_ih_att[:, 0:R], W.w_ih_att[:, 2 * R:2 * R + E], W.w_hh_att], 1), R, self.ks_pad)
            W.p_lang_ks = pack_weights(torch.cat([W.w_ih_lang, W.w_hh_lang], 1), R, self.ks_pad)
        if self.ks_att or self.ks_lang:
            self.gate_slab = torch.empty(max(self.ks_att, self.ks_lang) * (R // 8) * 2048, device=dev, dtype=torch.float32)
            self.gate_counters = torch.zeros(R // 64, device=dev, dtype=torch.int32)
        if self.gsk:
            # launch shapes of the stream-K schedule (host arithmetic, same call the C driver makes) and the partial-tile slabs
            A, V = W.A, W.V
            self.gsk_nwg = int(torch.cuda.get_device_properties(dev).multi_processor_count)
            nt_r, nt_v, nt_a = R // 64, ((V + 31) // 32 + 7) // 8, (A // 32 + 7) // 8
            self.plan_a = hip.gsk_plan([nt_r, nt_v], [2 * R // 32, R // 32], self.gsk_nwg)     # att-early || logits
            self.plan_o = hip.gsk_plan([nt_v], [R // 32], self.gsk_nwg)                        # logits alone (last step)
            self.plan_l = hip.gsk_plan([nt_r, nt_a], [2 * R // 32, R // 32], self.gsk_nwg)     # lang-early || h2attn
            slab = lambda ntile, maxseg: torch.zeros(ntile * maxseg * 16384, device=dev, dtype=torch.float32)
            self.slab_att = slab(nt_r, self.plan_a["maxseg"][0])
            self.slab_o = slab(nt_v, max(self.plan_a["maxseg"][1], self.plan_o["maxseg"][0]))
            self.slab_lang = slab(nt_r, self.plan_l["maxseg"][0])
            self.slab_q = slab(nt_a, self.plan_l["maxseg"][1])

    def _build_packed(self):
        L, W = hip.lib(), self.W
        B, N, Fr, R, A, E, V, rows = self.B, self.N, self.F, W.R, W.A, W.E, W.V, self.rows
        fc, conv, pconv, pool, ppool = self.feats
        ptr = lambda t: None if t is None else t.data_ptr()
        qoff = lambda buf, k0: buf.data_ptr() + (k0 // 4) * 64 * 4 * 4        # byte address of quad k0/4
        out = []
        seg_fc = _segs([(fc, None, W.w_ih_att[:, R:2 * R], False)])
        out.append(("gate_fc", L.cvc_linear_fwd, (seg_fc, 1, ptr(W.b_ih_att), ptr(W.b_hh_att), rows, 4 * R, ptr(self.gate_fc),
                                                  4 * R)))
        self._keep.append(seg_fc)
        nblk_v = (V + 31) // 32
        if self.gsk:
            return out + self._build_gsk_steps()
        if self.embgate:
            return out + self._build_embgate_steps()
        for t in range(self.T):
            rd, wr = t & 1, (t + 1) & 1
            XA_r, XA_w, XL_r, XL_w = self.XA[rd], self.XA[wr], self.XL[rd], self.XL[wr]
            if self.ks_att and self.gate_fused:
                out.append(("att_lstm", L.cvc_packed_lstm_ksf_fwd, (ptr(W.p_att), ptr(XA_r), 2 * R + E, None, None, ptr(self.gate_fc),
                                                                    ptr(self.cA[rd]), rows, R, qoff(XL_r, R), qoff(XA_w, R + E),
                                                                    ptr(self.cA[wr]), ptr(self.gate_slab), ptr(self.gate_counters))))
            elif self.ks_att:
                wp_att = W.p_att_ks if self.ks_pad else W.p_att
                out.append(("att_lstm", L.cvc_packed_lstm_ks_fwd, (ptr(wp_att), ptr(XA_r), 2 * R + E, None, None, ptr(self.gate_fc),
                                                                   ptr(self.cA[rd]), rows, R, qoff(XL_r, R), qoff(XA_w, R + E),
                                                                   ptr(self.cA[wr]), ptr(self.gate_slab), wp_att.stride(0))))
            else:
                out.append(("att_lstm", L.cvc_packed_lstm_fwd, (ptr(W.p_att), ptr(XA_r), 2 * R + E, None, None, ptr(self.gate_fc),
                                                                ptr(self.cA[rd]), rows, R, qoff(XL_r, R), qoff(XA_w, R + E),
                                                                ptr(self.cA[wr]))))
            out.append(("h2attn", L.cvc_packed_linear_fwd, (ptr(W.p_h), qoff(XL_r, R), R, None, rows, A, self.QSPLIT,
                                                            ptr(self.q_parts), A, None)))
            sets = (hip.AttnSet * 2)()
            sets[0] = hip.AttnSet(ptr(ppool), ptr(pool), ptr(self.mask), None, ptr(self.scores_r), None,
                                  ptr(self.att_steps[t]), None, N, self.stream_r)
            sets[1] = hip.AttnSet(ptr(pconv), ptr(conv), None, None, ptr(self.scores_f), None, ptr(self.attn_f), None, Fr,
                                  self.stream_f)
            out.append(("attn_scores", L.cvc_attn_scores_qparts, (W.kind, ptr(self.q_parts), self.QSPLIT, ptr(W.b_h), ptr(W.w_a),
                                                                  ptr(W.b_a), self.inv_temp, sets, 2, B, 1, A)))
            out.append(("attn_wsum", L.cvc_attn_wsum_quad, (sets, 2, B, 1, R, ptr(XL_r))))
            if self.ks_lang and self.gate_fused:
                out.append(("lang_lstm", L.cvc_packed_lstm_ksf_fwd, (ptr(W.p_lang), ptr(XL_r), 3 * R, ptr(W.b_ih_lang),
                                                                     ptr(W.b_hh_lang), None, ptr(self.cL[rd]), rows, R, ptr(XA_w),
                                                                     qoff(XL_w, 2 * R), ptr(self.cL[wr]), ptr(self.gate_slab),
                                                                     ptr(self.gate_counters))))
            elif self.ks_lang:
                wp_lang = W.p_lang_ks if self.ks_pad else W.p_lang
                out.append(("lang_lstm", L.cvc_packed_lstm_ks_fwd, (ptr(wp_lang), ptr(XL_r), 3 * R, ptr(W.b_ih_lang),
                                                                    ptr(W.b_hh_lang), None, ptr(self.cL[rd]), rows, R, ptr(XA_w),
                                                                    qoff(XL_w, 2 * R), ptr(self.cL[wr]), ptr(self.gate_slab),
                                                                    wp_lang.stride(0))))
            elif self.lang_ksx:
                out.append(self._lang_ksx_launch(t, XL_r, XA_w, XL_w, rd, wr))
            else:
                out.append(("lang_lstm", L.cvc_packed_lstm_fwd, (ptr(W.p_lang), ptr(XL_r), 3 * R, ptr(W.b_ih_lang), ptr(W.b_hh_lang),
                                                                 None, ptr(self.cL[rd]), rows, R, ptr(XA_w), qoff(XL_w, 2 * R),
                                                                 ptr(self.cL[wr]))))
            out.append(("logits", L.cvc_packed_linear_fwd, (ptr(W.p_o), ptr(XA_w), R, ptr(W.b_o), rows, V, 1, None, V,
                                                            ptr(self.top2_part))))
            out.append(("word_select", L.cvc_top2_final, (ptr(self.top2_part), nblk_v, rows, self.unk, ptr(self.words[t + 1]), 1,
                                                          ptr(self.logprob[t]), ptr(W.embed), E, qoff(XA_w, R), 0)))
            self._keep.append(sets)
        return out

    def _build_embgate_steps(self):
        """The T steps of the embedding-gate schedule (the launch list csrc/decode_driver.hip::run_packed_eg enqueues)."""
        L, W = hip.lib(), self.W
        B, N, Fr, R, A, V, rows = self.B, self.N, self.F, W.R, W.A, W.V, self.rows
        fc, conv, pconv, pool, ppool = self.feats
        ptr = lambda t: None if t is None else t.data_ptr()
        qoff = lambda buf, k0: buf.data_ptr() + (k0 // 4) * 64 * 4 * 4
        out = []
        nblk_v = (V + 31) // 32
        for t in range(self.T):
            rd, wr = t & 1, (t + 1) & 1
            XA_r, XA_w, XL_r, XL_w = self.XA[rd], self.XA[wr], self.XL[rd], self.XL[wr]
            # step 0 multiplies the all-zero initial state: one chunk of the attention cell's K, the language cell without its
            # h_lang columns (see run_packed_eg)
            first = t == 0 and hip.gemm_packed_split(-1) == 2
            out.append(("att_lstm", L.cvc_packed_lstm_embgate_ex_fwd, (ptr(W.p_att2), (2 * R // 4) * 128, ptr(XA_r), 32 if first else 2 * R, None, None,
                                                                       ptr(self.gate_fc), ptr(W.t_embgate), ptr(self.words[t]), ptr(self.cA[rd]),
                                                                       rows, R, qoff(XL_r, R), qoff(XA_w, R), ptr(self.cA[wr]),
                                                                       1 if self.att_w_cached else 0)))
            out.append(("h2attn", L.cvc_packed_linear_fwd, (ptr(W.p_h), qoff(XL_r, R), R, None, rows, A, self.QSPLIT,
                                                            ptr(self.q_parts), A, None)))
            sets = (hip.AttnSet * 2)()
            sets[0] = hip.AttnSet(ptr(ppool), ptr(pool), ptr(self.mask), None, ptr(self.scores_r), None,
                                  ptr(self.att_steps[t]), None, N, self.stream_r)
            sets[1] = hip.AttnSet(ptr(pconv), ptr(conv), None, None, ptr(self.scores_f), None, ptr(self.attn_f), None, Fr,
                                  self.stream_f)
            out.append(("attn_scores", L.cvc_attn_scores_qparts, (W.kind, ptr(self.q_parts), self.QSPLIT, ptr(W.b_h), ptr(W.w_a),
                                                                  ptr(W.b_a), self.inv_temp, sets, 2, B, 1, A)))
            out.append(("attn_wsum", L.cvc_attn_wsum_quad, (sets, 2, B, 1, R, ptr(XL_r))))
            if first:
                out.append(("lang_lstm", L.cvc_packed_lstm_late_fwd, (ptr(W.p_lang), (3 * R // 4) * 128, ptr(XL_r), 2 * R, ptr(W.b_ih_lang),
                                                                      ptr(W.b_hh_lang), None, ptr(self.cL[rd]), rows, R, ptr(XA_w),
                                                                      qoff(XL_w, 2 * R), ptr(self.cL[wr]), None)))
            elif self.lang_ksx:
                out.append(self._lang_ksx_launch(t, XL_r, XA_w, XL_w, rd, wr))
            else:
                out.append(("lang_lstm", L.cvc_packed_lstm_fwd, (ptr(W.p_lang), ptr(XL_r), 3 * R, ptr(W.b_ih_lang), ptr(W.b_hh_lang),
                                                                 None, ptr(self.cL[rd]), rows, R, ptr(XA_w), qoff(XL_w, 2 * R),
                                                                 ptr(self.cL[wr]))))
            # (cvc_packed_linear_select_fwd, the one-launch form whose last workgroup merges the records, measured 34.9 us against
            # 20.0 + 7.4 us for these two launches: atomics, fence and a serial merge on one CU cost more than a launch boundary)
            out.append(("logits", L.cvc_packed_linear_fwd, (ptr(W.p_o), ptr(XA_w), R, ptr(W.b_o), rows, V, 1, None, V,
                                                            ptr(self.top2_part))))
            out.append(("word_select", L.cvc_top2_final, (ptr(self.top2_part), nblk_v, rows, self.unk, ptr(self.words[t + 1]), 1,
                                                          ptr(self.logprob[t]), None, 0, None, 0)))
            self._keep.append(sets)
        return out
