"""Inference-time fused pieces of the once-per-clip encoder on the build's own kernels (csrc/encoder_ops.hip): class-similarity
softmax, layer-norm + concat, the frame embeddings' bias / ReLU / concat / BatchNorm(eval) / ReLU epilogue.  `usable(x)` says
whether a call takes them (GPU tensor, fp32, no autograd, module in eval mode); otherwise the caller keeps the torch formulation
and says why once (cvc.hip.warn_once) -- training through the encoder uses autograd on the torch ops for these few MB."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

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
