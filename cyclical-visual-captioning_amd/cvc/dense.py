"""Inference-time dense layers of the once-per-clip encoder on the tile GEMM (csrc/gemm_tile.hip): `nn.Linear`, or the
reference's Linear -> ReLU -> Dropout blocks (backbone.py:55-79), applied to thousands of rows at a time (B*F frame rows,
B*N region rows).  Split products on the bf16 MFMA with fp32 accumulation: fp32-grade results (tests/test_encoder.py) at
about twice the rate of the library's fp32 GEMM.  Autograd, CPU tensors and small row counts keep the module itself."""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn as nn

from . import hip

ENABLED = True
MIN_ROWS = 1024          # below this the launch-bound library kernel is as good


def _weight_operand(lin: nn.Linear):
    w = lin.weight
    stamp = (hip.weights_generation(), w.data_ptr(), w._version)      # generation: updates that leave _version alone (fused Adam)
    ent = getattr(lin, "_cvc_tile_operand", None)      # lives on the module
    if ent is not None and ent[0] == stamp:
        return ent[1]
    op = hip.TileOperand(w.detach().float().contiguous())
    lin._cvc_tile_operand = (stamp, op)
    return op


def usable(x: torch.Tensor) -> bool:
    """Inference path (packed weight operand cached on the module)."""
    return (ENABLED and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
            and x.numel() // max(1, x.shape[-1]) >= MIN_ROWS)


def usable_train(x: torch.Tensor) -> bool:
    """Autograd path: cvc.functional.linear (skinny / tile GEMM forward, tile GEMM dW / dX in the backward), any row count."""
    return ENABLED and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled()


def _tail(rest, y, site):
    """ReLU / Dropout modules of a block applied one by one (torch formulation); the Dropout goes through cvc.dropout.apply so that
    its mask is the site's (in-kernel generator, or dictated by a test)"""
    from . import dropout
    for m in rest:
        y = dropout.apply(m, y, site) if (isinstance(m, nn.Dropout) and site is not None) else m(y)
    return y


def apply(layer: nn.Module, x: torch.Tensor, site: str = None) -> torch.Tensor:
    """layer(x) for `nn.Linear` or `nn.Sequential(nn.Linear, nn.ReLU[, nn.Dropout])`.  site: the name of the block's dropout
    (cvc/dropout.py) -- in train() its mask comes from the in-kernel generator."""
    lin = layer if isinstance(layer, nn.Linear) else (layer[0] if isinstance(layer, nn.Sequential) and len(layer) > 0 else None)
    if not isinstance(lin, nn.Linear):
        return layer(x)
    rest = [] if layer is lin else list(layer)[1:]
    if not all(isinstance(m, (nn.ReLU, nn.Dropout)) for m in rest):
        return layer(x)
    if usable_train(x) and lin.out_features % 4 == 0:
        # under autograd: the hot path's linear (tile GEMM forward, dW / dX products in the backward) and, for the reference's
        # Linear -> ReLU -> Dropout blocks, ONE kernel for bias + ReLU + dropout with the mask generated there (csrc/encoder_train.hip)
        from . import functional as F_, encoder_ops, dropout
        w, xx = lin.weight, x
        if x.shape[-1] % 4 != 0:                            # an odd input width (loc_fc: 5): zero columns up to a multiple of 4
            pad = 4 - x.shape[-1] % 4
            xx, w = torch.nn.functional.pad(x, (0, pad)), torch.nn.functional.pad(lin.weight, (0, pad))
        has_relu = any(isinstance(m, nn.ReLU) for m in rest)
        drop = next((m for m in rest if isinstance(m, nn.Dropout)), None)
        drop_on = drop is not None and drop.training and drop.p > 0
        if has_relu and encoder_ops.usable_train(x) and (site is not None or not drop_on) and (not drop_on or (dropout.in_kernel(x) and drop.p < 1)):
            return encoder_ops.relu_dropout(F_.linear(xx, w, None), drop if drop_on else None, site or "", bias=lin.bias)
        return _tail(rest, F_.linear(xx, w, lin.bias), site)
    if not usable(x) or layer.training:
        if x.is_cuda and ENABLED and x.numel() // max(1, x.shape[-1]) >= MIN_ROWS:
            hip.warn_once("dense-library", "a dense encoder layer over >= %d rows runs on the library GEMM (%s)" % (
                MIN_ROWS, "module in train() mode without autograd" if layer.training else f"dtype {x.dtype}"))
        return _tail(rest, lin(x), site)
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    y = hip.tile_mm(x2, _weight_operand(lin), bias=lin.bias)
    if any(isinstance(m, nn.ReLU) for m in rest):
        y.relu_()
    return y.view(*lead, -1)
