"""`torch.library` registration of the hot path's operators as `cvc::*` (last row): the same HIP launches
the `nn.Module` mirrors reach through cvc.functional, visible to the dispatcher -- schemas, fake (meta) implementations for shape
propagation / `torch.compile` tracing, autograd formulas that are themselves `cvc::*` ops.  Importing this module registers them;
nothing in the default path depends on it (cvc.functional's autograd Functions stay the product path: they carry the deferred
weight-gradient batching and the gradient sinks, which an op-by-op registration cannot express).

    cvc::attn_fwd        AdditiveSoftAttention / SoftAttention over one feature set (model/modules.py:24-159)
    cvc::lstm_cell       nn.LSTMCell over a virtual concat of input segments (decoder_core.py:45-50, 59-61)
    cvc::embed_relu      relu(Embedding(idx)) (captioner.py:53-68, eval-mode dropout)
    cvc::vocab_nll       log_softmax + masked NLL sum + argmax from raw logits (captioner.py:266, :313; misc/utils.py:139-146)
    cvc::grounder        dot-product grounder (captioner.py:132-173)
    cvc::top2_unk        greedy word selection with UNK suppression (captioner.py:415-422)
    cvc::linear          nn.Linear on the skinny / tile GEMM

Backward ops: cvc::attn_bwd, cvc::lstm_cell_bwd, cvc::embed_relu_bwd, cvc::vocab_nll_bwd, cvc::grounder_bwd, cvc::linear_bwd.
There is no CPU implementation (the ops raise on CPU tensors, like the rest of the package)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import functional as F_
from . import hip

_lib = torch.library


# ------------------------------------------------------------------------------------------------ embed_relu
@_lib.custom_op("cvc::embed_relu", mutates_args=())
def embed_relu(table: Tensor, idx: Tensor) -> Tensor:
    return hip.embed_relu_fwd(table, idx.reshape(-1).contiguous(), None).reshape(*idx.shape, table.shape[1])


@embed_relu.register_fake
def _(table, idx):
    return table.new_empty(*idx.shape, table.shape[1])


@_lib.custom_op("cvc::embed_relu_bwd", mutates_args=())
def embed_relu_bwd(table: Tensor, idx: Tensor, d_out: Tensor) -> Tensor:
    return hip.embed_relu_bwd(table, idx.reshape(-1).contiguous(), None, d_out.reshape(-1, table.shape[1]).contiguous())


@embed_relu_bwd.register_fake
def _(table, idx, d_out):
    return torch.empty_like(table)


def _embed_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _embed_bwd(ctx, g):
    table, idx = ctx.saved_tensors
    return torch.ops.cvc.embed_relu_bwd(table, idx, g.contiguous()), None


embed_relu.register_autograd(_embed_bwd, setup_context=_embed_setup)


# ------------------------------------------------------------------------------------------------ linear
@_lib.custom_op("cvc::linear", mutates_args=())
def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    if x2.shape[0] > F_.TILE_LINEAR_ROWS:
        y = hip.tile_mm(x2, weight.contiguous(), bias=bias)
    else:
        y = hip.linear_fwd([{"x": x2, "w": weight}], bias, x2.shape[0], weight.shape[0])
    return y.reshape(*x.shape[:-1], weight.shape[0])


@linear.register_fake
def _(x, weight, bias):
    return x.new_empty(*x.shape[:-1], weight.shape[0])


@_lib.custom_op("cvc::linear_bwd", mutates_args=())
def linear_bwd(dy: Tensor, x: Tensor, weight: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    dy2, x2 = dy.reshape(-1, dy.shape[-1]).contiguous(), x.reshape(-1, x.shape[-1]).contiguous()
    d_x = F_._mm_nn(dy2, weight).reshape(x.shape)
    d_w = hip.tile_mm(dy2, x2, a_kmajor=True, b_kmajor=True)
    return d_x, d_w, dy2.sum(0)


@linear_bwd.register_fake
def _(dy, x, weight):
    return torch.empty_like(x), torch.empty_like(weight), weight.new_empty(weight.shape[0])


def _linear_setup(ctx, inputs, output):
    x, weight, bias = inputs
    ctx.save_for_backward(x, weight)
    ctx.has_bias = bias is not None


def _linear_bwd(ctx, g):
    x, weight = ctx.saved_tensors
    d_x, d_w, d_b = torch.ops.cvc.linear_bwd(g.contiguous(), x, weight)
    return d_x, d_w, (d_b if ctx.has_bias else None)


linear.register_autograd(_linear_bwd, setup_context=_linear_setup)


# ------------------------------------------------------------------------------------------------ attention (one feature set)
@_lib.custom_op("cvc::attn_fwd", mutates_args=())
def attn_fwd(kind: int, q: Tensor, w_a: Optional[Tensor], b_a: Optional[Tensor], inv_temp: float, proj: Tensor, ctx: Tensor,
             mask: Optional[Tensor], frame_mask: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
    """-> (ctx_out [rows, R], attn [rows, n], frame_masked [rows, n] pre-softmax copy, or an empty tensor without frame_mask)"""
    nclip = proj.shape[0]
    nq = q.shape[0] // nclip
    sets = [{"proj": proj.contiguous(), "ctx": ctx.contiguous(), "mask": mask, "frame_mask": frame_mask}]
    outs, _ = hip.attn_fwd(kind, q.contiguous(), None if w_a is None else w_a.reshape(-1), b_a, inv_temp, sets, nclip, nq)
    _scores, fm, attn, ctx_out = outs[0]
    return ctx_out, attn, (fm if fm is not None else q.new_empty(0))


@attn_fwd.register_fake
def _(kind, q, w_a, b_a, inv_temp, proj, ctx, mask, frame_mask):
    rows, n = q.shape[0], proj.shape[1]
    return q.new_empty(rows, ctx.shape[2]), q.new_empty(rows, n), (q.new_empty(rows, n) if frame_mask is not None else q.new_empty(0))


@_lib.custom_op("cvc::attn_bwd", mutates_args=())
def attn_bwd(kind: int, q: Tensor, w_a: Optional[Tensor], inv_temp: float, proj: Tensor, ctx: Tensor, attn: Tensor,
             d_ctx: Optional[Tensor], d_fm: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """-> (d_q, d_w_alpha [A] (zeros for dot attention), d_b_alpha [1], d_proj, d_ctx_feats)"""
    nclip = proj.shape[0]
    nq = q.shape[0] // nclip
    d_scores, d_q, d_w_part, d_proj, d_cf = hip.attn_bwd(
        kind, q.contiguous(), None if w_a is None else w_a.reshape(-1), inv_temp, proj.contiguous(), ctx.contiguous(), attn,
        None if d_ctx is None else d_ctx.contiguous(), None if d_fm is None else d_fm.contiguous(), nclip, nq, True, d_ctx is not None,
        w_a is not None)
    d_w = d_w_part.sum(0) if d_w_part is not None else q.new_zeros(q.shape[1])
    return d_q, d_w, d_scores.sum().reshape(1), d_proj, (d_cf if d_cf is not None else torch.zeros_like(ctx))


@attn_bwd.register_fake
def _(kind, q, w_a, inv_temp, proj, ctx, attn, d_ctx, d_fm):
    return torch.empty_like(q), q.new_empty(q.shape[1]), q.new_empty(1), torch.empty_like(proj), torch.empty_like(ctx)


def _attn_setup(ctx, inputs, output):
    kind, q, w_a, b_a, inv_temp, proj, cfeat, mask, frame_mask = inputs
    ctx.kind, ctx.inv_temp = kind, inv_temp
    ctx.has_w, ctx.has_b, ctx.has_fm = w_a is not None, b_a is not None, frame_mask is not None
    ctx.save_for_backward(q, w_a, proj, cfeat, output[1])
    ctx.set_materialize_grads(False)


def _attn_bwd(ctx, d_ctx, _d_attn, d_fm):
    q, w_a, proj, cfeat, attn = ctx.saved_tensors
    if d_fm is not None and not ctx.has_fm:
        d_fm = None
    d_q, d_w, d_b, d_proj, d_cf = torch.ops.cvc.attn_bwd(ctx.kind, q, w_a, ctx.inv_temp, proj, cfeat, attn, d_ctx, d_fm)
    return (None, d_q, (d_w.reshape(w_a.shape) if ctx.has_w else None), (d_b if ctx.has_b else None), None, d_proj, d_cf, None, None)


attn_fwd.register_autograd(_attn_bwd, setup_context=_attn_setup)


# ------------------------------------------------------------------------------------------------ LSTM cell
@_lib.custom_op("cvc::lstm_cell", mutates_args=())
def lstm_cell(x: Tensor, h: Tensor, c: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """nn.LSTMCell: -> (h', c', activated gates [M, 4R] kept for the backward)"""
    segs = [{"x": x.contiguous(), "w": w_ih}, {"x": h.contiguous(), "w": w_hh}]
    h2, c2, gates = hip.lstm_cell_fwd(segs, b_ih, b_hh, c.contiguous(), want_gates=True)
    return h2, c2, gates


@lstm_cell.register_fake
def _(x, h, c, w_ih, w_hh, b_ih, b_hh):
    return torch.empty_like(h), torch.empty_like(c), h.new_empty(h.shape[0], 4 * h.shape[1])


@_lib.custom_op("cvc::lstm_cell_bwd", mutates_args=())
def lstm_cell_bwd(d_h: Optional[Tensor], d_c: Optional[Tensor], gates: Tensor, c_prev: Tensor, c_new: Tensor, x: Tensor, h_prev: Tensor,
                  w_ih: Tensor, w_hh: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """-> (d_x, d_h_prev, d_c_prev, d_w_ih, d_w_hh, d_bias)"""
    d_gates, d_c_prev = hip.lstm_pointwise_bwd(None if d_h is None else d_h.contiguous(), None if d_c is None else d_c.contiguous(), gates,
                                               c_prev.contiguous(), c_new)
    d_x, d_hp = F_._mm_nn(d_gates, w_ih), F_._mm_nn(d_gates, w_hh)
    d_wi, d_wh = F_._mm_tn(d_gates, x.contiguous()), F_._mm_tn(d_gates, h_prev.contiguous())
    return d_x, d_hp, d_c_prev, d_wi, d_wh, d_gates.sum(0)


@lstm_cell_bwd.register_fake
def _(d_h, d_c, gates, c_prev, c_new, x, h_prev, w_ih, w_hh):
    return (torch.empty_like(x), torch.empty_like(h_prev), torch.empty_like(c_prev), torch.empty_like(w_ih), torch.empty_like(w_hh),
            w_ih.new_empty(w_ih.shape[0]))


def _cell_setup(ctx, inputs, output):
    x, h, c, w_ih, w_hh, b_ih, b_hh = inputs
    ctx.save_for_backward(x, h, c, w_ih, w_hh, output[1], output[2])
    ctx.set_materialize_grads(False)


def _cell_bwd(ctx, d_h, d_c, _d_gates):
    x, h, c, w_ih, w_hh, c_new, gates = ctx.saved_tensors
    d_x, d_hp, d_cp, d_wi, d_wh, d_b = torch.ops.cvc.lstm_cell_bwd(d_h, d_c, gates, c, c_new, x, h, w_ih, w_hh)
    return d_x, d_hp, d_cp, d_wi, d_wh, d_b, d_b


lstm_cell.register_autograd(_cell_bwd, setup_context=_cell_setup)


# ------------------------------------------------------------------------------------------------ vocabulary criterion
@_lib.custom_op("cvc::vocab_nll", mutates_args=())
def vocab_nll(logits: Tensor, target: Tensor, w: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """-> (sum_m w[m] * -log_softmax(logits)[m, target[m]] [1], argmax [M] int64, lse [M] kept for the backward)"""
    loss, lse, amax = hip.vocab_nll_fwd(logits.contiguous(), target.contiguous(), w.contiguous())
    return loss, amax, lse


@vocab_nll.register_fake
def _(logits, target, w):
    M = logits.shape[0]
    return logits.new_empty(1), target.new_empty(M), logits.new_empty(M)


@_lib.custom_op("cvc::vocab_nll_bwd", mutates_args=())
def vocab_nll_bwd(logits: Tensor, lse: Tensor, target: Tensor, w: Tensor, g: Tensor) -> Tensor:
    return hip.vocab_nll_bwd(logits.contiguous(), lse, target.contiguous(), w.contiguous(), g.contiguous().reshape(1))


@vocab_nll_bwd.register_fake
def _(logits, lse, target, w, g):
    return torch.empty_like(logits)


def _nll_setup(ctx, inputs, output):
    logits, target, w = inputs
    ctx.save_for_backward(logits, output[2], target, w)


def _nll_bwd(ctx, g, _a, _l):
    logits, lse, target, w = ctx.saved_tensors
    return torch.ops.cvc.vocab_nll_bwd(logits, lse, target, w, g), None, None


vocab_nll.register_autograd(_nll_bwd, setup_context=_nll_setup)


# ------------------------------------------------------------------------------------------------ grounder, word selection
@_lib.custom_op("cvc::grounder", mutates_args=())
def grounder(xt: Tensor, feats: Tensor, bias: Optional[Tensor], mask: Optional[Tensor]) -> Tensor:
    return hip.grounder_fwd(xt.contiguous(), feats.contiguous(), None if bias is None else bias.contiguous(), mask)


@grounder.register_fake
def _(xt, feats, bias, mask):
    return xt.new_empty(xt.shape[0], xt.shape[1], feats.shape[1])


@_lib.custom_op("cvc::grounder_bwd", mutates_args=())
def grounder_bwd(d: Tensor, xt: Tensor, feats: Tensor) -> Tuple[Tensor, Tensor]:
    d_xt, d_feats = hip.grounder_bwd(d.contiguous(), xt.contiguous(), feats.contiguous(), True, True)
    return d_xt, d_feats


@grounder_bwd.register_fake
def _(d, xt, feats):
    return torch.empty_like(xt), torch.empty_like(feats)


def _ground_setup(ctx, inputs, output):
    xt, feats, bias, mask = inputs
    ctx.save_for_backward(xt, feats, mask)
    ctx.has_bias = bias is not None


def _ground_bwd(ctx, d):
    xt, feats, mask = ctx.saved_tensors
    d = d.masked_fill(mask, 0) if mask is not None else d
    d_xt, d_feats = torch.ops.cvc.grounder_bwd(d, xt, feats)
    return d_xt, d_feats, (d if ctx.has_bias else None), None


grounder.register_autograd(_ground_bwd, setup_context=_ground_setup)


@_lib.custom_op("cvc::top2_unk", mutates_args=())
def top2_unk(logits: Tensor, unk_idx: int) -> Tuple[Tensor, Tensor]:
    """-> (word [M] int64, its log-prob [M]): captioner.py:415-422"""
    M = logits.shape[0]
    word = torch.empty(M, dtype=torch.int64, device=logits.device)
    logprob = torch.empty(M, dtype=torch.float32, device=logits.device)
    hip.top2_unk(logits.contiguous(), unk_idx, word, 1, logprob)
    return word, logprob


@top2_unk.register_fake
def _(logits, unk_idx):
    M = logits.shape[0]
    return logits.new_empty(M, dtype=torch.int64), logits.new_empty(M)


REGISTERED = ("embed_relu", "embed_relu_bwd", "linear", "linear_bwd", "attn_fwd", "attn_bwd", "lstm_cell", "lstm_cell_bwd", "vocab_nll",
              "vocab_nll_bwd", "grounder", "grounder_bwd", "top2_unk")
