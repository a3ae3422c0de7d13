"""Dropout sites of the cyclical pass and a way to dictate their masks.

The reference trains with nn.Dropout(drop_prob_lm) on the embedded word (model/captioner.py:53-68: three embeddings per pass --
teacher-forced decode, localizer queries, reconstruction) and on the language LSTM's output (model/decoder_core.py:62, 109: once
per step of the decode and reconstruction loops).  Every site has a name:

    emb_a, emb_b, emb_c              [B * T, E]   embedded words of loop A (decode), B (localize), C (reconstruct)
    out_a.<t>, out_c.<t>             [B, R]       output of step t of loop A / loop C
    vis_embed                        [B, T, G]    the grounder's class embeddings, roi_feat_extractor.vis_embed (model/backbone.py:55-57,
                                                  used at captioner.py:284)

`keep_mask(site, shape, p, device)` returns the site's keep-mask ALREADY divided by (1 - p) (what the kernels multiply by).
By default it is drawn with torch.bernoulli; under `injected(fn)` it is whatever fn(site, shape) returns -- the train-mode
parity tests hand the same masks to the CPU oracle (tests/test_gpu_train.py), so that the whole train-mode pass (losses and
every gradient) is compared with dropout ON, not only in eval mode."""
from __future__ import annotations

import contextlib
from typing import Callable, Optional

import torch

_inject: Optional[Callable] = None


@contextlib.contextmanager
def injected(fn: Callable):
    """fn(site: str, shape: tuple) -> tensor of keep / (1 - p) values (any device; moved as needed)."""
    global _inject
    prev, _inject = _inject, fn
    try:
        yield
    finally:
        _inject = prev


def active() -> bool:
    return _inject is not None


def keep_mask(site: str, shape, p: float, device) -> torch.Tensor:
    if _inject is not None:
        m = _inject(site, tuple(shape))
        return m.to(device=device, dtype=torch.float32).reshape(tuple(shape)).contiguous()
    return torch.bernoulli(torch.full(tuple(shape), 1.0 - p, device=device)).div_(1.0 - p)


def apply(module: torch.nn.Dropout, x: torch.Tensor, site: Optional[str]) -> torch.Tensor:
    """module(x), with the site's dictated mask when masks are injected."""
    if _inject is None or site is None or not module.training or module.p <= 0:
        return module(x)
    return x * keep_mask(site, x.shape, module.p, x.device)
