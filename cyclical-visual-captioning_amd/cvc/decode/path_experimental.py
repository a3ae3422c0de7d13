"""Schedules that were measured and do not pay (include/cvc_hip_experimental.h, only in a library built with
CVC_EXPERIMENTAL=1): the grouped stream-K schedule, the K-split gate GEMM with the exchange finish.  Mixin of
cvc.decode.engine.DecodeEngine; nothing here runs unless asked for (gsk / gate_ksplit / lang_ksx)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch

from .. import hip
from .weights import *          # noqa: F401,F403  (packers, layouts, cache plan, switches)
from .weights import _segs


class ExperimentalPaths:
    def _lang_ksx_launch(self, t, XL_r, XA_w, XL_w, rd, wr):
        """The language cell of step t on cvc_packed_lstm_ksx_fwd (same operands and destinations as the full-K launch)."""
        L, W, R = hip.lib(), self.W, self.W.R
        ptr = lambda x: None if x is None else x.data_ptr()
        qoff = lambda buf, k0: buf.data_ptr() + (k0 // 4) * 64 * 4 * 4
        return ("lang_lstm", L.cvc_packed_lstm_ksx_fwd, (ptr(W.p_lang), ptr(XL_r), 3 * R, ptr(W.b_ih_lang), ptr(W.b_hh_lang), None, None, None,
                                                          ptr(self.cL[rd]), self.rows, R, ptr(XA_w), qoff(XL_w, 2 * R), ptr(self.cL[wr]),
                                                          ptr(self.ksx_slab), ptr(self.ksx_flags), t + 1))

    def _build_gsk_steps(self):
        """The T steps of the grouped stream-K schedule (the launch list csrc/decode_driver.hip::run_packed_gsk enqueues)."""
        L, W = hip.lib(), self.W
        B, N, Fr, R, A, E, V, rows = self.B, self.N, self.F, W.R, W.A, W.E, W.V, self.rows
        fc, conv, pconv, pool, ppool = self.feats
        ptr = lambda t: None if t is None else t.data_ptr()
        qoff = lambda buf, k0: buf.data_ptr() + (k0 // 4) * 64 * 4 * 4
        ws_att, ws_lang, ws_r = (2 * R + E) // 4 * 128, 3 * R // 4 * 128, R // 4 * 128
        pa, po, pl = self.plan_a, self.plan_o, self.plan_l
        segs = lambda slab, plan, g, nchunk: hip.GskSegs(ptr(slab), plan["unit0"][g], nchunk, plan["U"], plan["maxseg"][g])
        seg_att, seg_o_a = segs(self.slab_att, pa, 0, 2 * R // 32), segs(self.slab_o, pa, 1, R // 32)
        seg_o_o = segs(self.slab_o, po, 0, R // 32)
        seg_lang, seg_q = segs(self.slab_lang, pl, 0, 2 * R // 32), segs(self.slab_q, pl, 1, R // 32)
        self._keep += [seg_att, seg_o_a, seg_o_o, seg_lang, seg_q]
        byref = C.byref
        out = []
        for t in range(self.T):
            rd, wr = t & 1, (t + 1) & 1
            XA_r, XA_w, XL_r, XL_w = self.XA[rd], self.XA[wr], self.XL[rd], self.XL[wr]
            out.append(("att_late", L.cvc_packed_lstm_late_fwd, (ptr(W.p_att) + (R // 4) * 128 * 4, ws_att, qoff(XA_r, R), E, None, None,
                                                                 ptr(self.gate_fc), ptr(self.cA[rd]), rows, R, qoff(XL_r, R),
                                                                 qoff(XA_w, R + E), ptr(self.cA[wr]),
                                                                 None if t == 0 else byref(seg_att))))
            gl = (hip.GskGroup * 2)()
            gl[0] = hip.GskGroup(ptr(W.p_lang), ws_lang, ptr(XL_r), R // 8, 2 * R // 32, 0, R // 32, ptr(self.slab_lang), pl["maxseg"][0])
            gl[1] = hip.GskGroup(ptr(W.p_h), ws_r, qoff(XL_r, R), A // 32, R // 32, 0, 0, ptr(self.slab_q), pl["maxseg"][1])
            out.append(("lang_early_h2attn", L.cvc_gsk_gemm, (gl, 2, pl["U"])))
            sets = (hip.AttnSet * 2)()
            sets[0] = hip.AttnSet(ptr(ppool), ptr(pool), ptr(self.mask), None, ptr(self.scores_r), None,
                                  ptr(self.att_steps[t]), None, N, self.stream_r)
            sets[1] = hip.AttnSet(ptr(pconv), ptr(conv), None, None, ptr(self.scores_f), None, ptr(self.attn_f), None, Fr,
                                  self.stream_f)
            out.append(("attn_scores", L.cvc_attn_scores_qslab, (W.kind, byref(seg_q), ptr(W.b_h), ptr(W.w_a), ptr(W.b_a),
                                                                 self.inv_temp, sets, 2, B, 1, A)))
            out.append(("attn_wsum", L.cvc_attn_wsum_quad, (sets, 2, B, 1, R, ptr(XL_r))))
            out.append(("lang_late", L.cvc_packed_lstm_late_fwd, (ptr(W.p_lang), ws_lang, ptr(XL_r), R, ptr(W.b_ih_lang),
                                                                  ptr(W.b_hh_lang), None, ptr(self.cL[rd]), rows, R, ptr(XA_w),
                                                                  qoff(XL_w, 2 * R), ptr(self.cL[wr]), byref(seg_lang))))
            last = t + 1 == self.T
            ga = (hip.GskGroup * 2)()
            ga[0] = hip.GskGroup(ptr(W.p_att), ws_att, ptr(XA_w), R // 8, 2 * R // 32, R // 32, E // 32, ptr(self.slab_att), pa["maxseg"][0])
            ga[1] = hip.GskGroup(ptr(W.p_o), ws_r, ptr(XA_w), (V + 31) // 32, R // 32, 0, 0, ptr(self.slab_o),
                                 po["maxseg"][0] if last else pa["maxseg"][1])
            if last:        # no next step: the vocabulary projection alone
                go = (hip.GskGroup * 1)()
                go[0] = ga[1]
                out.append(("att_early_logits", L.cvc_gsk_gemm, (go, 1, po["U"])))
                self._keep.append(go)
            else:
                out.append(("att_early_logits", L.cvc_gsk_gemm, (ga, 2, pa["U"])))
            out.append(("word_select", L.cvc_top2_slab, (byref(seg_o_o if last else seg_o_a), ptr(W.b_o), V, rows, self.unk,
                                                         ptr(self.words[t + 1]), 1, ptr(self.logprob[t]), ptr(W.embed), E,
                                                         qoff(XA_w, R), 0)))
            self._keep += [sets, gl, ga]
        return out

    # ------------------------------------------------------------------ tile path (rows > 64 or beam search)

    def check_ksx(self):
        """(host sync, once per engine) After the first decode with lang_ksx: if a K slice's wait for its tile ran out -- the
        tile's workgroups were not resident together or not on one XCD -- the exchange's error word is set and that decode is
        not valid: re-bind on the full-K kernel and say so.  Returns True when the engine was re-bound."""
        if not self.lang_ksx or self._ksx_checked:
            return False
        self._ksx_checked = True
        torch.cuda.synchronize()
        if int(self.ksx_flags[-1]) == 0:
            return False
        hip.warn_once("lang-ksx", "decode: the K-split language cell's in-launch exchange reported a failed wait (workgroup placement); "
                      "falling back to the full-K gate GEMM")
        self.lang_ksx = False
        self._launches = self._build_packed()
        if self._plan is not None:
            hip.lib().cvc_decode_plan_destroy(self._plan)
            self._plan = None
            self._bind_driver()
        return True
