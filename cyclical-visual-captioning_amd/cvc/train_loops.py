"""The two recurrent loops of the cyclical training pass as ONE autograd node each, driven from C.

Loop A (teacher-forced decode, reference model/captioner.py:242-270 calling decoder_core.py:30-66) and loop C (reconstruction
from the localized regions, captioner.py:348-362 calling decoder_core.py:86-113) used to be unrolled through autograd step by
step (cvc/functional.py: one Function per cell / attention / linear, ~1 100 launches per training step, a tenth of the step in
framework add / cat / copy / fill kernels).  Here each loop is one `torch.autograd.Function` whose forward and backward are one
call into libcvc_hip.so (`cvc_train_loop_fwd` / `cvc_train_loop_bwd`, csrc/train_driver.hip): every per-step launch is
enqueued from C, the fan-in of h's three consumers is summed inside the gate-gradient kernel, nothing is packed or
concatenated between steps.  What remains on this side is dense and happens once per loop:

* before the loop -- the hoisted input products (embedded words, fc_feats, loop C's localized context) on the tile GEMM;
* after BOTH loops' back-propagation -- every weight gradient as one product over all (2 x) T x B sample rows.  The two loops
  share the LSTM cells (captioner.py:86-87), so their rows live side by side in one `LoopArena` and the last loop to finish its
  backward multiplies all of them at once: no per-loop dW, no accumulation.

Rows are t-major everywhere (row t * B + b).  Parity: tests/test_gpu_parity.py (a9 goldens), tests/test_gpu_train.py
(train mode, in-kernel dropout), tests/test_gpu_fullsize.py (config 3 / 4 sizes) run through this path by default; the per-step
path of cvc/functional.py stays selectable (`cvc.train_loops.ENABLED = False`, dictated dropout masks, more than 64 clips).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import functional as F_
from . import hip

ENABLED = os.environ.get("CVC_TRAIN_LOOPS", "1") != "0"        # False: the per-step autograd path (A/B switch)
PACKED_H2ATTN = os.environ.get("CVC_TRAIN_PACKED_H2ATTN", "1") != "0"   # False: h2attn of loop A on the row-major ring kernel (A/B)
JOINT_BWD = os.environ.get("CVC_TRAIN_JOINT_BWD", "1") != "0"   # False: the two loops' back-propagation as two passes even when 2B <= 64 (A/B)
JOINT_BWD_128 = os.environ.get("CVC_TRAIN_JOINT_BWD_128", "1") != "0"   # False: no joint pass for 64 < 2B <= 128 (A/B)

Tensor = torch.Tensor


MAX_CLIPS = 256           # per GPU; a loop takes 64 clips, a larger batch runs as several 64-clip groups (clip_groups)
CHUNKED = os.environ.get("CVC_TRAIN_LOOP_GROUPS", "1") != "0"           # False: more than 64 clips -> the per-step path, as before round 6 (A/B)


def eligible(B: int, R: int, E: int, A: int, like: Tensor, T: int = 1) -> bool:
    """shapes / placement the C-driven loops take (everything else keeps the per-step path)"""
    b_ok = 1 <= B <= 64 or (CHUNKED and B <= MAX_CLIPS and T <= 64)
    return bool(ENABLED and like.is_cuda and like.dtype == torch.float32 and b_ok and R % 32 == 0 and E % 16 == 0
                and A % 8 == 0 and F_.PACKED_LSTM_FORWARD and hip.gemm_packed_split(-1) >= 0)


def clip_groups(B: int):
    """[(b0, b1)]: the batch as groups of at most 64 clips.  A loop is 64 rows wide (the packed gate GEMM's operand, the
    backward-data product's row group); clips are independent through both loops, so a larger per-GPU batch (reference
    cfgs/cyclical.yml: 48; a 288 GB part invites 128+) runs the loops once per group -- own arena, own joint back-propagation, own
    weight-gradient products, which autograd sums -- while everything batch-wide (embedding, vocabulary head and criteria,
    grounder, localizer) stays one call over all B clips.  Full groups first (96 -> 64 + 32, 130 -> 64 + 64 + 2): a cell launch
    costs the same for 33 .. 64 rows and ~20 % less for <= 32, and a last group of <= 32 clips takes the one-operand joint backward."""
    return [(b0, min(b0 + 64, B)) for b0 in range(0, B, 64)]


class LoopArena:
    """Row-major buffers of the two loops' T * B sample rows each, side by side (slot 0 = loop A, slot 1 = loop C): the X and dY
    operands of the weight-gradient products, which the last backward to finish takes over all rows at once."""

    def __init__(self, nslots: int, T: int, B: int, R: int, E: int, device):
        self.nslots, self.T, self.B, self.R, self.E = nslots, T, B, R, E
        S = nslots * T * B
        e = lambda *shape: torch.empty(*shape, device=device, dtype=torch.float32)
        self.h_lang_prev, self.h_att_prev, self.h_att, self.ctx = e(S, R), e(S, R), e(S, R), e(S, R)
        self.emb = e(S, E)
        self.dg_att, self.dg_lang = e(S, 4 * R), e(S, 4 * R)
        self.dgsum_att, self.dgsum_lang = e(nslots, B, 4 * R), e(nslots, B, 4 * R)     # sums over the T steps, per loop (bias / fc gradients)
        self.extra = {}                 # slot 0's attention-side buffers (dq, dwa_part, ds_r, ds_f), set by its backward
        self.done = []                  # slots whose backward has run
        self.loop_a = None              # loop A's autograd context, for the joint back-propagation run from loop C's node
        self.joint_done = False         # the joint pass has run: loop A's node only has its dense input gradients left
        self.a_feat_grads = None
        # True: this arena's weight-gradient products are the parameters' ONLY producers this step, so a gradient written straight into
        # a sink (the reducer's arena view) is announced final and its bucket's exchange may leave at once.  False (the batch runs as
        # several 64-clip groups, each with its own arena): the groups' gradients are summed -- the first to arrive may still write in
        # place, but nothing is final before autograd has accumulated the others (the bucket then leaves from the parameter's hook)
        self.sole = True

    def joint_ok(self) -> bool:
        """Both loops' back-propagation through time runs as one pass (cvc_train_loops_bwd_joint), every backward-data product streaming
        the shared LSTM weights once for both: their rows as ONE 64-row operand (2 B <= 64: config 4's share) or as two 64-row operand
        groups on the 128-row form of the product (config 3: B = 64; split-product arithmetic only)."""
        fits = 2 * self.B <= 64 or (JOINT_BWD_128 and hip.gemm_packed_split(-1) != 0)
        return bool(JOINT_BWD and fits and self.nslots == 2 and self.loop_a is not None and torch.is_grad_enabled())

    def rows(self, slot: int) -> slice:
        n = self.T * self.B
        return slice(slot * n, (slot + 1) * n)


def _ptr(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


class _Cfg:
    """non-tensor arguments of a loop (one object so that Function.apply's positional list stays readable)"""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def _hoisted(x_tb: Tensor, w_cols: Tensor) -> Tensor:
    """[T * B, k] x W[:, cols]^T -> [T * B, 4R] on the tile GEMM (skinny kernel for <= 64 rows)"""
    if x_tb.shape[0] > 64:
        return hip.tile_mm(x_tb, hip.weight_operand(w_cols))
    return hip.linear_fwd([{"x": x_tb, "w": w_cols}], None, x_tb.shape[0], w_cols.shape[0])


_iota_cache = {}


def _iota(n: int, device) -> Tensor:
    key = (n, str(device))
    t = _iota_cache.get(key)
    if t is None:
        t = _iota_cache[key] = torch.arange(n, device=device, dtype=torch.int64)
    return t


_ws_cache = {}


def _bwd_ws(B: int, R: int, A: int, device) -> Tensor:
    """the backward driver's scratch (quad operands whose rows beyond B must read zero: allocated zero ONCE, the kernels only
    ever write rows < B); one per (shape, device, stream-capture state) -- a captured graph keeps its own"""
    key = (B, R, A, str(device), torch.cuda.is_current_stream_capturing())
    t = _ws_cache.get(key)
    if t is None:
        n = int(hip.lib().cvc_train_loop_bwd_ws(B, R, A))
        assert n > 0
        t = _ws_cache[key] = torch.zeros(n, device=device, dtype=torch.float32)
    return t


class _Out:
    """where one weight's gradient goes: the owner's .grad buffer itself when a gradient sink hands it out (cvc.functional.
    GRAD_SINKS: written in place, autograd gets None), else a fresh tensor returned through autograd"""

    def __init__(self, param: Optional[Tensor], final: bool = True):
        self.param = param
        self.final = final
        self.buf, self.sink = F_.claim_grad(param) if param is not None else (None, None)
        self.t = self.buf if self.buf is not None else (torch.empty_like(param) if param is not None else None)
        self.announced = False

    def done(self):
        """-> what the Function returns for this weight.  With a sink: announced as the parameter's COMPLETE gradient of this backward
        (the loops' flush is its only producer), so its bucket's exchange can leave behind the product that was just enqueued."""
        if self.sink is not None and not self.announced:
            self.sink.written(self.param, final=self.final)
            self.announced = True
        return None if self.sink is not None else self.t


def _weight_grads(arena: LoopArena, cfg, W):
    """All weight gradients of the loops in `arena.done`, one product per (weight, input segment) over all their rows.
    W: name -> parameter (w_ih_a, w_hh_a, b_ih_a, b_hh_a, w_ih_l, w_hh_l, b_ih_l, b_hh_l, w_h, b_h, w_a, b_a) + fc."""
    T, B, R, E = arena.T, arena.B, arena.R, arena.E
    slots = sorted(arena.done)
    n = T * B
    assert slots == list(range(slots[0], slots[-1] + 1))
    rows = slice(slots[0] * n, (slots[-1] + 1) * n)
    nl = len(slots)
    DGa, DGl = arena.dg_att[rows], arena.dg_lang[rows]
    Hl, Ha_prev, Ha, Cx, Em = arena.h_lang_prev[rows], arena.h_att_prev[rows], arena.h_att[rows], arena.ctx[rows], arena.emb[rows]
    O = {k: _Out(W[k], arena.sole) for k in ("w_ih_a", "w_hh_a", "b_ih_a", "b_hh_a", "w_ih_l", "w_hh_l", "b_ih_l", "b_hh_l")}
    # Largest gradient bucket first (cvc.distributed.GradReducer: every LSTM weight matrix is its own bucket, the biases ride with
    # weight_hh), each announced as soon as its products are enqueued: its exchange then runs under the products that follow.
    # ---- attention cell weight_ih = [h_lang | (fc) | emb]   (168 MB at D = 2048)
    Dp = hip.TileOperand(DGa, kmajor=True)                    # dG^T packed once for every product of the cell
    Hlp = hip.TileOperand(Hl, kmajor=True)                    # h_lang(t-1): att weight_ih[:, :R] AND lang weight_hh
    d_ih = O["w_ih_a"].t
    hip.tile_mm(Dp, Hlp, out=d_ih[:, :R])
    # [B, 4R] sum over every step of every loop (accumulated by the gate-gradient kernel): the fc columns' dY (fc is the same row
    # every step) and the biases
    DGsum = arena.dgsum_att[slots[0]:slots[-1] + 1].view(nl * B, 4 * R)      # (two loops: the rows of both, summed by the products below)
    e0 = R
    if cfg.has_fc:
        fc2 = W["fc"] if nl == 1 else W["fc"].repeat(nl, 1)
        hip.tile_mm(DGsum, fc2, a_kmajor=True, b_kmajor=True, out=d_ih[:, R:2 * R])
        e0 = 2 * R
    hip.tile_mm(Dp, Em, b_kmajor=True, out=d_ih[:, e0:])
    O["w_ih_a"].done()
    # ---- language cell weight_ih = [ctx | h_att]   (134 MB)
    Dl = hip.TileOperand(DGl, kmajor=True)
    d_il = O["w_ih_l"].t
    hip.tile_mm(Dl, Cx, b_kmajor=True, out=d_il[:, :R])
    Hap = hip.TileOperand(Ha, kmajor=True)
    hip.tile_mm(Dl, Hap, out=d_il[:, R:])
    O["w_ih_l"].done()
    # ---- the two weight_hh + biases   (67 MB each)
    hip.tile_mm(Dp, Ha_prev, b_kmajor=True, out=O["w_hh_a"].t)
    hip.col_sum(DGsum, O["b_ih_a"].t, O["b_hh_a"].t)
    for k in ("w_hh_a", "b_ih_a", "b_hh_a"):
        O[k].done()
    hip.tile_mm(Dl, Hlp, out=O["w_hh_l"].t)
    hip.col_sum(arena.dgsum_lang[slots[0]:slots[-1] + 1].view(nl * B, 4 * R), O["b_ih_l"].t, O["b_hh_l"].t)
    for k in ("w_hh_l", "b_ih_l", "b_hh_l"):
        O[k].done()
    # ---- h2attn / alpha_net: loop A's rows only
    if 0 in slots and arena.extra:
        x = arena.extra
        DQ = x["dq"].view(n, -1)
        O["w_h"], O["b_h"] = _Out(W["w_h"], arena.sole), _Out(W["b_h"], arena.sole)
        hip.tile_mm(DQ, arena.h_att[arena.rows(0)], a_kmajor=True, b_kmajor=True, out=O["w_h"].t)
        hip.col_sum(DQ, O["b_h"].t)
        if x.get("dwa_part") is not None:
            O["w_a"], O["b_a"] = _Out(W["w_a"], arena.sole), _Out(W["b_a"], arena.sole)
            hip.col_sum(x["dwa_part"].view(n, -1), O["w_a"].t.view(-1))
            O["b_a"].t.copy_((x["ds_r"].sum() + x["ds_f"].sum()).reshape(1))
    return {k: o.done() for k, o in O.items()}


class _Loop(torch.autograd.Function):
    """kind 0: loop A; kind 1: loop C.  Tensor inputs (fixed order, absent ones None):
       emb [B, T, E], fc [B, R] | None, ctx_all [B, T, R] (kind 1), pool, ppool, conv, pconv (kind 0),
       w_ih_a, w_hh_a, b_ih_a, b_hh_a, w_ih_l, w_hh_l, b_ih_l, b_hh_l, w_h, b_h, w_a, b_a (w_a / b_a additive attention only; loop C
       takes the attention weights too, unused in its forward: whichever loop finishes its backward LAST returns every weight's
       gradient, and autograd only accepts a gradient for an input that was a tensor)
       pass_out, pass_fm: loop A's two outputs handed THROUGH loop C's node (cfg.joint; returned as they are): every consumer of
       loop A then sits behind loop C's node in the graph, so that node's backward holds both loops' output gradients and runs the
       joint back-propagation.  (Nothing of loop A's forward depends on loop C -- the localizer reads the argmax words, an integer
       cut, captioner.py:313 -- so the order of the two nodes' backward is free.)
    outputs: out [T, B, R] (dropout(h_lang)), fm [T, B, N] (kind 0 with frame masks, else a 0-d placeholder), pass_out, pass_fm"""

    @staticmethod
    def forward(ctx, cfg, arena, emb, fc, ctx_all, pool, ppool, conv, pconv, w_ih_a, w_hh_a, b_ih_a, b_hh_a, w_ih_l, w_hh_l,
                b_ih_l, b_hh_l, w_h, b_h, w_a, b_a, pass_out, pass_fm):
        kind, T, B, R, E = cfg.kind, arena.T, arena.B, arena.R, arena.E
        dev = emb.device
        slot = cfg.slot
        rows = arena.rows(slot)
        e = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        # ---- hoisted input products (dense, once per loop)
        e0 = 2 * R if cfg.has_fc else R
        # (the embedding kernel writes [B, T, E] -- its dropout mask is indexed that way -- the loops want t-major rows: one
        # strided copy into the arena, which is also the X operand of the weight-gradient product)
        arena.emb[rows].view(T, B, E).copy_(emb.transpose(0, 1))
        gpre_att = _hoisted(arena.emb[rows], w_ih_a[:, e0:])
        row_bias = None
        if cfg.has_fc:       # fc_feats x W_ih[:, R:2R]^T + b_ih + b_hh, one row per clip (decoder_core.py:46)
            row_bias = hip.linear_fwd([{"x": fc.contiguous(), "w": w_ih_a[:, R:2 * R]}], b_ih_a, B, 4 * R, bias2=b_hh_a)
        gpre_lang = None
        if kind == 1:
            arena.ctx[rows].view(T, B, R).copy_(ctx_all.transpose(0, 1))
            gpre_lang = _hoisted(arena.ctx[rows], w_ih_l[:, :R])
        # ---- weight packs (rebuilt once per optimizer step, shared by both loops)
        wp_att = hip.lstm_train_pack(w_ih_a, w_hh_a, cols=[(0, R)])
        wp_lang = hip.lstm_train_pack(w_ih_l, w_hh_l) if kind == 0 else hip.lstm_train_pack(w_ih_l, w_hh_l, cols=[(R, R)])
        L = hip.TrainLoop()
        L.kind, L.B, L.T, L.R = kind, B, T, R
        keep = dict(gpre_att=gpre_att, row_bias=row_bias, gpre_lang=gpre_lang, wp_att=wp_att, wp_lang=wp_lang)
        L.wp_att, L.wp_lang = _ptr(wp_att), _ptr(wp_lang)
        if row_bias is None:
            L.b_ih_att, L.b_hh_att = _ptr(b_ih_a), _ptr(b_hh_a)
        else:
            L.row_bias, L.row_index = _ptr(row_bias), _ptr(_iota(B, dev))
        L.b_ih_lang, L.b_hh_lang = _ptr(b_ih_l), _ptr(b_hh_l)
        L.w_ih_att, L.w_hh_att, L.w_ih_lang, L.w_hh_lang = _ptr(w_ih_a), _ptr(w_hh_a), _ptr(w_ih_l), _ptr(w_hh_l)
        L.ld_ih_att, L.ld_ih_lang = w_ih_a.stride(0), w_ih_l.stride(0)
        L.gpre_att, L.gpre_lang = _ptr(gpre_att), _ptr(gpre_lang)
        if cfg.drop is not None:
            state, site0, p = cfg.drop
            L.rng_state, L.site0, L.p = hip._rng_ptr(state), int(site0), float(p)
        buf = dict(out=e(T, B, R), c_att=e(T + 1, B, R), c_lang=e(T + 1, B, R), g_att=e(T, B, 4 * R), g_lang=e(T, B, 4 * R),
                   xa0=e(2 * R // 4, 64, 4), xa1=e(2 * R // 4, 64, 4))
        kl = 3 * R if kind == 0 else 2 * R
        buf["xl0"], buf["xl1"] = e(kl // 4, 64, 4), e(kl // 4, 64, 4)
        L.out, L.c_att, L.c_lang, L.g_att, L.g_lang = (_ptr(buf[k]) for k in ("out", "c_att", "c_lang", "g_att", "g_lang"))
        L.h_att, L.h_att_prev, L.h_lang_prev = _ptr(arena.h_att[rows]), _ptr(arena.h_att_prev[rows]), _ptr(arena.h_lang_prev[rows])
        L.xa[0], L.xa[1], L.xl[0], L.xl[1] = _ptr(buf["xa0"]), _ptr(buf["xa1"]), _ptr(buf["xl0"]), _ptr(buf["xl1"])
        fm = None
        if kind == 0:
            A, N, Fr = w_h.shape[0], pool.shape[1], conv.shape[1]
            L.A, L.N, L.F, L.attn_kind, L.inv_temp = A, N, Fr, cfg.attn_kind, float(cfg.inv_temp)
            L.w_h, L.b_h, L.w_a, L.b_a = _ptr(w_h), _ptr(b_h), _ptr(None if w_a is None else w_a.reshape(-1)), _ptr(b_a)
            qs = 1
            if PACKED_H2ATTN and A % 32 == 0 and R % 128 == 0:
                # the query GEMM split-K over the whole chip on the packed kernel; its planes are summed by the score pass
                qs = 8 if R % 256 == 0 else 4
                keep["wp_h"] = hip.linear_train_pack(w_h)
                L.wp_h, L.q_split = _ptr(keep["wp_h"]), qs
            pool, ppool, conv, pconv = (t.contiguous() for t in (pool, ppool, conv, pconv))
            L.pool, L.ppool, L.conv, L.pconv = _ptr(pool), _ptr(ppool), _ptr(conv), _ptr(pconv)
            keep.update(pool=pool, ppool=ppool, conv=conv, pconv=pconv, mask=cfg.mask, fmask=cfg.frame_mask)
            L.mask, L.frame_mask = _ptr(cfg.mask), _ptr(cfg.frame_mask)
            buf.update(q=e(T, qs, B, A), attn_r=e(T, B, N), attn_f=e(T, B, Fr), scores=e(B, N + Fr))
            if cfg.frame_mask is not None:
                fm = buf["fm"] = e(T, B, N)
            L.ctx, L.q, L.attn_r, L.attn_f, L.fm, L.scores_ws = (_ptr(arena.ctx[rows]), _ptr(buf["q"]), _ptr(buf["attn_r"]),
                                                                 _ptr(buf["attn_f"]), _ptr(fm), _ptr(buf["scores"]))
        hip._check(hip.lib().cvc_train_loop_fwd(C.byref(L), hip._stream()), "cvc_train_loop_fwd")
        need_bwd = any(ctx.needs_input_grad)
        if need_bwd:
            ctx.L, ctx.buf, ctx.keep, ctx.cfg, ctx.arena = L, buf, keep, cfg, arena
            ctx.save_for_backward(fc, w_ih_a, w_hh_a, b_ih_a, b_hh_a, w_ih_l, w_hh_l, b_ih_l, b_hh_l, w_h, b_h, w_a, b_a)
            ctx.key = ("loops", id(arena))
            F_._BATCHER.note_use(ctx.key)
            if kind == 0:
                arena.loop_a = ctx
        ctx.set_materialize_grads(False)
        out = buf["out"]
        if fm is None:
            fm = out.new_empty(())
        return out, fm, pass_out, pass_fm

    @staticmethod
    def _arm(ctx, d_out, d_fm):
        """fill the backward half of the loop descriptor; -> (the gradients of the feature inputs, tensors to keep alive)"""
        L, keep, cfg, arena = ctx.L, ctx.keep, ctx.cfg, ctx.arena
        kind, T, B, R = cfg.kind, arena.T, arena.B, arena.R
        dev = arena.emb.device
        rows = arena.rows(cfg.slot)
        ni = ctx.needs_input_grad            # (cfg, arena, emb_tb, fc, ctx_tb, pool, ppool, conv, pconv, 12 weights...)
        e = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        if d_out is None:
            d_out = torch.zeros(T, B, R, device=dev)
        d_out = d_out.contiguous()
        L.d_out = _ptr(d_out)
        L.dg_att, L.dg_lang = _ptr(arena.dg_att[rows]), _ptr(arena.dg_lang[rows])
        L.dgsum_att, L.dgsum_lang = _ptr(arena.dgsum_att[cfg.slot]), _ptr(arena.dgsum_lang[cfg.slot])
        grads_feat = [None, None, None, None]
        live = [d_out]
        if kind == 0:
            A, N, Fr = L.A, L.N, L.F
            if d_fm is not None and d_fm.dim() == 3:
                d_fm = d_fm.contiguous()
                L.d_fm = _ptr(d_fm)
                live.append(d_fm)
            x = dict(dq=e(T, B, A), ds_r=e(T, B, N), ds_f=e(T, B, Fr),
                     dwa_part=e(T, B, A) if cfg.attn_kind == hip.ATTN_ADDITIVE else None)
            L.dq, L.ds_r, L.ds_f, L.dwa_part = _ptr(x["dq"]), _ptr(x["ds_r"]), _ptr(x["ds_f"]), _ptr(x["dwa_part"])
            arena.extra = x
            for i, (name, src) in enumerate((("d_pool", "pool"), ("d_ppool", "ppool"), ("d_conv", "conv"), ("d_pconv", "pconv"))):
                if ni[5 + i]:
                    grads_feat[i] = torch.zeros_like(keep[src])
                    setattr(L, name, _ptr(grads_feat[i]))
            if grads_feat[0] is not None or grads_feat[2] is not None:
                # every step's context gradient, so that the context-feature gradients are one pass after the loop (cvc_hip.h)
                x["d_ctx_all"] = e(T, 128, R)
                L.d_ctx_all = _ptr(x["d_ctx_all"])
        return grads_feat, live

    @staticmethod
    def backward(ctx, d_out, d_fm, d_pass_out, d_pass_fm):
        L, cfg, arena = ctx.L, ctx.cfg, ctx.arena
        fc, w_ih_a, w_hh_a, b_ih_a, b_hh_a, w_ih_l, w_hh_l, b_ih_l, b_hh_l, w_h, b_h, w_a, b_a = ctx.saved_tensors
        kind, T, B, R, E = cfg.kind, arena.T, arena.B, arena.R, arena.E
        dev = w_ih_a.device
        rows = arena.rows(cfg.slot)
        ni = ctx.needs_input_grad
        order = ("w_ih_a", "w_hh_a", "b_ih_a", "b_hh_a", "w_ih_l", "w_hh_l", "b_ih_l", "b_hh_l", "w_h", "b_h", "w_a", "b_a")
        owners = (w_ih_a, w_hh_a, b_ih_a, b_hh_a, w_ih_l, w_hh_l, b_ih_l, b_hh_l, w_h, b_h, w_a, b_a)
        W = dict(zip(order, owners), fc=fc)

        def flush(_items):
            g = _weight_grads(arena, cfg, W)
            arena.done = []
            return tuple(g.get(k) for k in order)

        got = None
        if kind == 0 and arena.joint_done:
            # loop C's node ran the back-propagation of both loops (and the weight gradients): only the dense input gradients are left
            grads_feat, live = arena.a_feat_grads, None
            arena.a_feat_grads, arena.loop_a = None, None
        elif getattr(cfg, "joint", False):
            a = arena.loop_a
            grads_feat_a, live_a = _Loop._arm(a, d_pass_out, d_pass_fm)
            grads_feat, live = _Loop._arm(ctx, d_out, None)
            ws = _bwd_ws(2 * B, R, a.L.A, dev)
            a.L.bwd_ws = L.bwd_ws = _ptr(ws)
            hip._check(hip.lib().cvc_train_loops_bwd_joint(C.byref(a.L), C.byref(L), hip._stream()), "cvc_train_loops_bwd_joint")
            arena.a_feat_grads, arena.joint_done = grads_feat_a, True
            arena.done += [0, 1]
            F_._BATCHER.add(ctx.key, (0,), owners, flush)
            got = F_._BATCHER.add(ctx.key, (1,), owners, flush)
            del live_a
        else:
            grads_feat, live = _Loop._arm(ctx, d_out, d_fm)
            ws = _bwd_ws(B, R, L.A if kind == 0 else 0, dev)
            L.bwd_ws = _ptr(ws)
            hip._check(hip.lib().cvc_train_loop_bwd(C.byref(L), hip._stream()), "cvc_train_loop_bwd")
            arena.done.append(cfg.slot)
            # ---- weight gradients: by whichever loop finishes its backward last, over all loops' rows at once -- BEFORE this
            # loop's input gradients, so that the gradient buckets (486 MB) are complete, and their exchange under way, as early
            # as possible.  The owner list carries every weight of both loops; a loop that does not own h2attn (loop C) still
            # returns its share of the flush through the same tuple positions when it happens to be last
            got = F_._BATCHER.add(ctx.key, (cfg.slot,), owners, flush)
        wg = [None] * 12
        if got is not None:
            for i, gi in enumerate(got):
                if gi is not None and owners[i] is not None and ni[9 + i]:
                    wg[i] = gi.view_as(owners[i]) if gi.numel() == owners[i].numel() else gi
        # ---- gradients of the hoisted inputs (dense, once per loop)
        DGa, DGl = arena.dg_att[rows], arena.dg_lang[rows]
        e0 = 2 * R if cfg.has_fc else R
        d_emb = hip.tile_mm(DGa, hip.weight_operand(w_ih_a[:, e0:], kmajor=True)).view(T, B, E).transpose(0, 1) if ni[2] else None
        d_fc = None
        if cfg.has_fc and ni[3]:
            d_fc = F_._mm_nn(arena.dgsum_att[cfg.slot], w_ih_a[:, R:2 * R])
        d_ctx = None
        if kind == 1 and ni[4]:
            d_ctx = hip.tile_mm(DGl, hip.weight_operand(w_ih_l[:, :R], kmajor=True)).view(T, B, R).transpose(0, 1)
        del live
        if kind == 0:
            arena.loop_a = None          # (the arena <-> context cycle would keep this loop's buffers alive until a gc pass)
        return (None, None, d_emb, d_fc, d_ctx, *grads_feat, *wg, None, None)


def decode_loop(arena: LoopArena, emb: Tensor, fc: Optional[Tensor], feats, mask, frame_mask, att_lstm, lang_lstm, soft_attn,
                attn_kind: int, inv_temp: float, drop):
    """Loop A.  emb [B, T, E] embedded teacher-forced words; feats = (pool, ppool, conv, pconv); mask [B, N] / frame_mask [T, B, N]
    bool or None; drop = (rng state, first site id, p) or None.  -> out [T, B, R], frame-masked pre-softmax scores [T, B, N] | None"""
    additive = attn_kind == hip.ATTN_ADDITIVE
    cfg = _Cfg(kind=0, slot=0, has_fc=fc is not None, attn_kind=attn_kind, inv_temp=inv_temp, drop=drop, mask=hip._mask(mask),
               frame_mask=hip._mask(frame_mask))
    pool, ppool, conv, pconv = feats
    out, fm, _, _ = _Loop.apply(cfg, arena, emb, fc, None, pool, ppool, conv, pconv, att_lstm.weight_ih, att_lstm.weight_hh,
                                att_lstm.bias_ih, att_lstm.bias_hh, lang_lstm.weight_ih, lang_lstm.weight_hh, lang_lstm.bias_ih,
                                lang_lstm.bias_hh, soft_attn.h2attn.weight, soft_attn.h2attn.bias,
                                soft_attn.alpha_net.weight if additive else None, soft_attn.alpha_net.bias if additive else None,
                                None, None)
    return out, (fm if frame_mask is not None else None)


def recon_loop(arena: LoopArena, emb: Tensor, fc: Optional[Tensor], ctx_all: Tensor, att_lstm, lang_lstm, soft_attn, attn_kind: int, drop,
               joint=None):
    """Loop C.  ctx_all [B, T, R]: the localized context (regions + frames) of every step; soft_attn: loop A's attention module
    (gradient conduit only, see _Loop).  joint: loop A's (out, fm) when `arena.joint_ok()` -- they come back as the tensors loop A's
    consumers must read from here on (see _Loop).  -> out [T, B, R], or (out, loop A's out, loop A's fm) with `joint`"""
    additive = attn_kind == hip.ATTN_ADDITIVE
    cfg = _Cfg(kind=1, slot=1, has_fc=fc is not None, attn_kind=attn_kind, inv_temp=1.0, drop=drop, mask=None, frame_mask=None,
               joint=joint is not None)
    p_out, p_fm = joint if joint is not None else (None, None)
    out, _, p_out, p_fm = _Loop.apply(cfg, arena, emb, fc, ctx_all, None, None, None, None, att_lstm.weight_ih, att_lstm.weight_hh,
                                      att_lstm.bias_ih, att_lstm.bias_hh, lang_lstm.weight_ih, lang_lstm.weight_hh, lang_lstm.bias_ih,
                                      lang_lstm.bias_hh, soft_attn.h2attn.weight, soft_attn.h2attn.bias,
                                      soft_attn.alpha_net.weight if additive else None, soft_attn.alpha_net.bias if additive else None,
                                      p_out, p_fm)
    return out if joint is None else (out, p_out, p_fm)
