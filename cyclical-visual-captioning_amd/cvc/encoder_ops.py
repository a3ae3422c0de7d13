"""Fused pieces of the once-per-clip encoder on the build's own kernels: class-similarity softmax, layer-norm + concat, the frame
embeddings' bias / ReLU / concat / BatchNorm / ReLU epilogue, the ReLU -> Dropout tail of the Linear blocks.
Inference forms (csrc/encoder_ops.hip): `usable(x, module)` -- GPU tensor, fp32, no autograd, module in eval mode.
Training forms (csrc/encoder_train.hip, round 4): `usable_train(x)` -- the same pieces under autograd with the reference modules'
train() semantics: nn.Dropout masks generated in the kernels (cvc/dropout.py sites enc.*), BatchNorm1d on batch statistics with the
running-statistics update, backward passes of the softmax and the layer norms.  Anything outside both (CPU tensors, other dtypes,
dictated dropout masks) keeps the torch formulation and says why once (cvc.hip.warn_once)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import hip

ENABLED = True


def usable(x: torch.Tensor, module: nn.Module) -> bool:
    return ENABLED and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and not module.training


def _cached(module: nn.Module, name: str, params: Sequence[torch.Tensor], build):
    """A derived tensor / operand cached on the module, keyed like the other packs (weights generation + versions)."""
    stamp = (hip.weights_generation(),) + tuple((p.data_ptr(), p._version) for p in params)
    ent = getattr(module, name, None)
    if ent is not None and ent[0] == stamp:
        return ent[1]
    val = build()
    setattr(module, name, (stamp, val))
    return val


def class_similarity(enc, g_pool_feats: torch.Tensor, pad: torch.Tensor):
    """backbone.py:222-235 -> (sim [B, C, N], sim_rows [B, N, C]): logits on the tile GEMM against the ReLU'd class table
    (packed once per weights generation), then ONE kernel for bias + pad fill + softmax over classes + both layouts."""
    B, N, G = g_pool_feats.shape
    w = enc.vis_embed[0].weight
    table = _cached(enc, "_cvc_class_table", [w], lambda: hip.TileOperand(torch.relu(w.detach()).contiguous()))
    logits = hip.tile_mm(g_pool_feats.reshape(B * N, G), table)                     # [B*N, C]
    Cn = logits.shape[1]
    sim = torch.empty(B, Cn, N, device=logits.device, dtype=torch.float32)
    rows = torch.empty(B, N, Cn, device=logits.device, dtype=torch.float32)
    padm = hip._mask(pad)
    hip._check(hip.lib().cvc_class_softmax_fwd(logits.data_ptr(), Cn, enc.vis_classifiers_bias.data_ptr(), padm.data_ptr(), B, N, Cn,
                                               sim.data_ptr(), rows.data_ptr(), hip._stream()), "cvc_class_softmax_fwd")
    return sim, rows


def layernorm_cat(xs: Sequence[torch.Tensor], eps: float = 1e-5) -> torch.Tensor:
    """cat([F.layer_norm(x, [x.shape[-1]]) for x in xs], -1) for up to three inputs sharing their leading shape."""
    lead = xs[0].shape[:-1]
    flat = [x.reshape(-1, x.shape[-1]) for x in xs]
    flat = [x if x.stride(1) == 1 else x.contiguous() for x in flat]
    rows = flat[0].shape[0]
    widths = [x.shape[1] for x in flat]
    out = torch.empty(rows, sum(widths), device=flat[0].device, dtype=torch.float32)
    n = len(flat)
    ptrs = (C.c_void_p * n)(*[x.data_ptr() for x in flat])
    lds = (C.c_longlong * n)(*[x.stride(0) for x in flat])
    ws = (C.c_int * n)(*widths)
    hip._check(hip.lib().cvc_layernorm_cat_fwd(ptrs, lds, ws, n, rows, float(eps), out.data_ptr(), out.stride(0), hip._stream()),
               "cvc_layernorm_cat_fwd")
    return out.view(*lead, -1)


def frame_embed(enc, rgb: torch.Tensor, motion: torch.Tensor) -> torch.Tensor:
    """backbone.py:325-333 in eval mode: the two dense products on the tile GEMM, then ONE kernel for their biases, ReLUs, the
    concat, BatchNorm1d (running statistics folded into scale / shift per channel, cached per weights generation) and the ReLU."""
    from . import dense
    l0, l1, bn = enc.att_embed[0][0], enc.att_embed[1][0], enc.att_embed_aux[0]
    lead = rgb.shape[:-1]
    y0 = hip.tile_mm(rgb.reshape(-1, rgb.shape[-1]), dense._weight_operand(l0))
    y1 = hip.tile_mm(motion.reshape(-1, motion.shape[-1]), dense._weight_operand(l1))

    def fold():
        scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
        return scale.contiguous(), (bn.bias.detach() - bn.running_mean * scale).contiguous()
    scale, shift = _cached(enc, "_cvc_bn_fold", [bn.weight, bn.bias, bn.running_mean, bn.running_var], fold)
    out = torch.empty(y0.shape[0], y0.shape[1] + y1.shape[1], device=y0.device, dtype=torch.float32)
    hip._check(hip.lib().cvc_frame_embed_fwd(y0.data_ptr(), l0.bias.data_ptr(), y0.shape[1], y1.data_ptr(), l1.bias.data_ptr(), y1.shape[1],
                                             scale.data_ptr(), shift.data_ptr(), y0.shape[0], out.data_ptr(), hip._stream()),
               "cvc_frame_embed_fwd")
    return out.view(*lead, -1)


# ====================================================================== training forms (csrc/encoder_train.hip)
def usable_train(x: torch.Tensor) -> bool:
    from . import dropout
    return ENABLED and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and not dropout.active()


class _ReluDropout(torch.autograd.Function):
    """dropout(relu(x + bias)) with the keep-mask generated in the kernel (the tail of the reference's Linear -> ReLU -> Dropout
    blocks, backbone.py:55-79); rng = (state, site, p) or None (eval: plain ReLU)"""

    @staticmethod
    def forward(ctx, x, bias, rng):
        x = x.contiguous()
        N = x.shape[-1]
        rows = x.numel() // N
        y = torch.empty_like(x)
        st, site, p = rng if rng is not None else (None, 0, 0.0)
        hip._check(hip.lib().cvc_relu_dropout_fwd(x.data_ptr(), hip._dev(bias), rows, N, None if st is None else hip._rng_ptr(st), int(site),
                                                  float(p), y.data_ptr(), hip._stream()), "cvc_relu_dropout_fwd")
        ctx.save_for_backward(y)
        ctx.rng, ctx.has_bias = rng, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        st, site, p = ctx.rng if ctx.rng is not None else (None, 0, 0.0)
        hip._check(hip.lib().cvc_relu_dropout_bwd(dy.data_ptr(), y.data_ptr(), y.numel(), None if st is None else hip._rng_ptr(st), int(site),
                                                  float(p), dx.data_ptr(), hip._stream()), "cvc_relu_dropout_bwd")
        db = dx.reshape(-1, dx.shape[-1]).sum(0) if (ctx.has_bias and ctx.needs_input_grad[1]) else None
        return dx, db, None


def relu_dropout(x: torch.Tensor, drop: Optional[nn.Dropout], site: str, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.Dropout(relu(x [+ bias])) as ONE kernel; `site` names the dropout (cvc/dropout.py) whose in-kernel mask is used"""
    from . import dropout
    rng = None
    if drop is not None and drop.training and 0 < drop.p < 1:
        rng = (dropout.rng_state(x.device), dropout.site_id(site), float(drop.p))
    return _ReluDropout.apply(x, bias, rng)


class _BatchNormRelu(torch.autograd.Function):
    """relu(BatchNorm1d(x)) over the rows of x [rows, C] on batch statistics (att_embed_aux, backbone.py:81, 332), running statistics
    updated in place as the module does"""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum):
        x = x.contiguous()
        C_ = x.shape[-1]
        rows = x.numel() // C_
        L = hip.lib()
        y = torch.empty_like(x)
        mean, invstd = torch.empty(C_, device=x.device), torch.empty(C_, device=x.device)
        ws = torch.empty(int(L.cvc_bn_workspace(rows, C_)), device=x.device)
        hip._check(L.cvc_bn_relu_train_fwd(x.data_ptr(), hip._dev(gamma), hip._dev(beta), float(eps), float(momentum), hip._dev(running_mean),
                                           hip._dev(running_var), rows, C_, y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(),
                                           hip._stream()), "cvc_bn_relu_train_fwd")
        ctx.save_for_backward(x, y, gamma, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mean, invstd = ctx.saved_tensors
        C_ = x.shape[-1]
        rows = x.numel() // C_
        L = hip.lib()
        dy = dy.contiguous()
        dx, dgamma, dbeta = torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)
        ws = torch.empty(int(L.cvc_bn_workspace(rows, C_)), device=x.device)
        hip._check(L.cvc_bn_relu_train_bwd(x.data_ptr(), dy.data_ptr(), y.data_ptr(), hip._dev(gamma), mean.data_ptr(), invstd.data_ptr(), rows,
                                           C_, dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), hip._stream()),
                   "cvc_bn_relu_train_bwd")
        return dx, dgamma, dbeta, None, None, None, None


def batchnorm_relu_train(x: torch.Tensor, bn: nn.BatchNorm1d) -> torch.Tensor:
    """att_embed_aux in train(): x [..., C] normalised per channel over all leading positions (what BatchNorm1d does on the
    transposed [B, C, F] tensor, backbone.py:332), then ReLU"""
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    # momentum None = cumulative moving average, factor 1 / num_batches_tracked (nn.BatchNorm1d; the host read is the module's own)
    if bn.momentum is None:
        momentum = 1.0 / float(bn.num_batches_tracked) if (bn.track_running_stats and bn.num_batches_tracked is not None) else 0.0
    else:
        momentum = bn.momentum
    y = _BatchNormRelu.apply(x, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                             bn.running_var if bn.track_running_stats else None, bn.eps, momentum)
    if bn.track_running_stats:
        # the kernel updated the running statistics through raw pointers: their version counters did not move, and the eval-mode
        # fold of this layer (cached on weights generation + data_ptr + _version) would keep the old statistics after a train()
        # forward that no optimizer step follows (round-4 advisor finding)
        hip.bump_weights_generation()
    return y


class _ClassSoftmax(torch.autograd.Function):
    """cvc_class_softmax_fwd under autograd: logits [B*N, C] (+ class bias, pad fill) -> (sim [B, C, N], sim_rows [B, N, C])"""

    @staticmethod
    def forward(ctx, logits, bias, pad, B, N):
        logits = logits.contiguous()
        Cn = logits.shape[1]
        sim = torch.empty(B, Cn, N, device=logits.device, dtype=torch.float32)
        rows = torch.empty(B, N, Cn, device=logits.device, dtype=torch.float32)
        padm = hip._mask(pad)
        hip._check(hip.lib().cvc_class_softmax_fwd(logits.data_ptr(), Cn, hip._dev(bias), padm.data_ptr(), B, N, Cn, sim.data_ptr(),
                                                   rows.data_ptr(), hip._stream()), "cvc_class_softmax_fwd")
        ctx.save_for_backward(rows, padm)
        ctx.dims = (B, N, Cn)
        ctx.set_materialize_grads(False)
        return sim, rows

    @staticmethod
    def backward(ctx, d_sim, d_rows):
        rows, padm = ctx.saved_tensors
        B, N, Cn = ctx.dims
        if d_sim is None and d_rows is None:
            return None, None, None, None, None
        d_logits = torch.empty(B * N, Cn, device=rows.device, dtype=torch.float32)
        hip._check(hip.lib().cvc_class_softmax_bwd(rows.data_ptr(), None if d_rows is None else d_rows.contiguous().data_ptr(),
                                                   None if d_sim is None else d_sim.contiguous().data_ptr(), padm.data_ptr(), B, N, Cn,
                                                   d_logits.data_ptr(), hip._stream()), "cvc_class_softmax_bwd")
        d_bias = d_logits.sum(0) if ctx.needs_input_grad[1] else None
        return d_logits, d_bias, None, None, None


def class_similarity_train(enc, g_pool_feats: torch.Tensor, pad: torch.Tensor):
    """backbone.py:222-235 under autograd: the class table goes through vis_embed (Embedding -> ReLU -> Dropout, in-kernel mask), the
    logits through the hot path's linear (tile GEMM forward and backward), the softmax through cvc_class_softmax_fwd / _bwd"""
    from . import dropout, functional as F_
    B, N, G = g_pool_feats.shape
    ve = enc.vis_embed
    idx = torch.arange(enc.detect_size + 1, device=g_pool_feats.device)
    if ve[2].training and 0 < ve[2].p < 1:
        table = F_.embed_relu(ve[0].weight, idx, rng=(dropout.rng_state(idx.device), dropout.site_id("enc.vis_table"), float(ve[2].p)))
    else:
        table = F_.embed_relu(ve[0].weight, idx)
    logits = F_.linear(g_pool_feats.reshape(B * N, G), table, None)
    return _ClassSoftmax.apply(logits, enc.vis_classifiers_bias, pad, B, N)


class _LayerNormCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eps, *xs):
        flat = [x.reshape(-1, x.shape[-1]) for x in xs]
        flat = [x if x.stride(1) == 1 else x.contiguous() for x in flat]
        out = layernorm_cat(flat, eps)
        ctx.save_for_backward(*flat)
        ctx.eps = eps
        ctx.shapes = [x.shape for x in xs]
        return out.view(*xs[0].shape[:-1], -1)

    @staticmethod
    def backward(ctx, d_out):
        flat = ctx.saved_tensors
        n = len(flat)
        rows = flat[0].shape[0]
        d_out = d_out.reshape(rows, -1).contiguous()
        dxs = [torch.empty_like(x) if ctx.needs_input_grad[1 + i] else None for i, x in enumerate(flat)]
        ptrs = (C.c_void_p * n)(*[x.data_ptr() for x in flat])
        lds = (C.c_longlong * n)(*[x.stride(0) for x in flat])
        ws = (C.c_int * n)(*[x.shape[1] for x in flat])
        dptrs = (C.c_void_p * n)(*[None if d is None else d.data_ptr() for d in dxs])
        dlds = (C.c_longlong * n)(*[0 if d is None else d.stride(0) for d in dxs])
        hip._check(hip.lib().cvc_layernorm_cat_bwd(ptrs, lds, ws, n, rows, float(ctx.eps), d_out.data_ptr(), d_out.stride(0), dptrs, dlds,
                                                   hip._stream()), "cvc_layernorm_cat_bwd")
        return (None, *[None if d is None else d.view(sh) for d, sh in zip(dxs, ctx.shapes)])


def layernorm_cat_train(xs: Sequence[torch.Tensor], eps: float = 1e-5) -> torch.Tensor:
    return _LayerNormCat.apply(eps, *xs)


def frame_embed_train(enc, rgb: torch.Tensor, motion: torch.Tensor) -> torch.Tensor:
    """backbone.py:325-333 in train(): the two Linear -> ReLU -> Dropout blocks (hot path's linear + the fused ReLU / dropout tail), the
    concat, BatchNorm1d on batch statistics + ReLU"""
    from . import functional as F_
    l0, l1 = enc.att_embed[0], enc.att_embed[1]
    y0 = relu_dropout(F_.linear(rgb, l0[0].weight, None), l0[2], "enc.att0", bias=l0[0].bias)
    y1 = relu_dropout(F_.linear(motion, l1[0].weight, None), l1[2], "enc.att1", bias=l1[0].bias)
    x = torch.cat((y0, y1), dim=-1)
    return batchnorm_relu_train(x, enc.att_embed_aux[0])
