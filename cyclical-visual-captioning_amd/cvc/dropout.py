"""Dropout sites of the cyclical pass and a way to dictate their masks.

The reference trains with nn.Dropout(drop_prob_lm) on the embedded word (model/captioner.py:53-68: three embeddings per pass --
teacher-forced decode, localizer queries, reconstruction) and on the language LSTM's output (model/decoder_core.py:62, 109: once
per step of the decode and reconstruction loops).  Every site has a name:

    emb_a, emb_b, emb_c              [B * T, E]   embedded words of loop A (decode), B (localize), C (reconstruct)
    out_a.<t>, out_c.<t>             [B, R]       output of step t of loop A / loop C
    vis_embed                        [B, T, G]    the grounder's class embeddings, roi_feat_extractor.vis_embed (model/backbone.py:55-57,
                                                  used at captioner.py:284)

Two ways a site gets its mask:

* **in the kernels (the default on the GPU)**: the consuming kernel generates the mask from a counter-based hash of
  (seed, step, site id, flat element index) -- `csrc/dropout_rng.h`; no mask tensor, no library RNG launch.  The generator
  state is 4 int32 words of device memory per device, `rng_state(device)`; `advance(device)` adds 1 to its step word with a
  device-side add (the captioner does so once per training pass; captured into a HIP graph it runs at every replay, so replays
  draw fresh masks); `seed(n)` re-seeds (default: torch.initial_seed() at first use).  `host_mask(site, shape, p, device)`
  reproduces a site's mask on the host from `cvc/synth.py::dropout_keep` -- what the train-mode parity tests hand to the oracle.
* **dictated**: under `injected(fn)` every site's mask is whatever fn(site, shape) returns (`keep_mask`), multiplied in as a
  tensor; the first round of train-mode parity tests works this way and stays.

`keep_mask(site, shape, p, device)` returns the site's keep-mask ALREADY divided by (1 - p) (what the kernels multiply by)."""
from __future__ import annotations

import contextlib
import os
from typing import Callable, Optional

import torch

from . import hip

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


# ------------------------------------------------------------------------------- in-kernel generator
IN_KERNEL = os.environ.get("CVC_DROPOUT_KERNEL", "1") != "0"        # False: masks as tensors from torch.bernoulli (A/B)
_SITES = {"emb_a": 1, "emb_b": 2, "emb_c": 3, "vis_embed": 4,
          # the once-per-clip encoder's dropouts in train() (model/backbone.py:55-79, 103-106): Linear -> ReLU -> Dropout blocks, the class
          # table's dropout inside the class similarity, the GRU's inter-layer dropout
          "enc.loc_fc": 16, "enc.fc_embed": 17, "enc.seg_info": 18, "enc.att0": 19, "enc.att1": 20, "enc.pool_embed": 21,
          "enc.ctx2pool_grd": 22, "enc.vis_table": 23, "enc.gru.0": 24, "enc.gru.1": 25, "enc.gru.2": 26}
_states = {}
_seed: Optional[int] = None


def site_id(site: str) -> int:
    """the number the kernels hash for a named site: emb_* / vis_embed 1..4, out_a.<t> 0x100 + t, out_c.<t> 0x200 + t"""
    if site in _SITES:
        return _SITES[site]
    kind, t = site.split(".")
    t = int(t)
    assert kind in ("out_a", "out_c") and 0 <= t < 256, site
    return (0x100 if kind == "out_a" else 0x200) + t


def seed(n: int) -> None:
    """re-seed the in-kernel generator of every device (step back to 0)"""
    global _seed
    _seed = int(n) & 0xFFFFFFFFFFFFFFFF
    for st in _states.values():
        st.copy_(torch.tensor(_words(_rank_seed(_seed)), dtype=torch.int32))


def _rank_seed(s: int) -> int:
    """Data-parallel ranks are all seeded with the same opts.seed (cvc.main); the hash has no rank term, so without this every
    rank would draw bit-identical masks for its shard -- the reference's DataParallel replicas draw independent ones.  Rank 0 (and a
    single process) keeps the plain seed."""
    import torch.distributed as dist
    r = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    return (s ^ (r * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF


def _words(s: int):
    i32 = lambda v: v - (1 << 32) if v >= (1 << 31) else v
    return [i32(s & 0xFFFFFFFF), i32((s >> 32) & 0xFFFFFFFF), 0, 0]


def rng_state(device) -> torch.Tensor:
    """the device's generator state: int32 [4] = {seed_lo, seed_hi, step, 0} (bit patterns of uint32 words)"""
    global _seed
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    st = _states.get(key)
    if st is None:
        if _seed is None:
            _seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
        st = torch.tensor(_words(_rank_seed(_seed)), dtype=torch.int32).to(device)
        _states[key] = st
    return st


def advance(device) -> None:
    """next training pass: step += 1, on the device (no host round trip; capturable)"""
    rng_state(device)[2:3].add_(1)


def in_kernel(x: torch.Tensor) -> bool:
    """does this tensor's dropout run inside the kernels? (GPU tensor, switch on, no dictated masks)"""
    return IN_KERNEL and _inject is None and x.is_cuda


def host_mask(site: str, shape, p: float, device) -> torch.Tensor:
    """the mask the kernels apply at `site` in the CURRENT step, restated on the host (synchronises; tests only)"""
    from . import synth
    w = [int(v) & 0xFFFFFFFF for v in rng_state(device).cpu().tolist()]
    n = 1
    for d in shape:
        n *= int(d)
    return torch.from_numpy(synth.dropout_keep(w[0], w[1], w[2], site_id(site), n, p)).reshape(tuple(shape))


class _DropoutRng(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, state, site, p):
        ctx.args = (state, site, p)
        return hip.dropout_rng(x.contiguous(), state, site, p)

    @staticmethod
    def backward(ctx, d):
        state, site, p = ctx.args
        return hip.dropout_rng(d.contiguous(), state, site, p), None, None, None


def apply(module: torch.nn.Dropout, x: torch.Tensor, site: Optional[str]) -> torch.Tensor:
    """module(x): in training, with the site's in-kernel mask (GPU) or its dictated mask (under `injected`)."""
    if site is None or not module.training or module.p <= 0:
        return module(x)
    if _inject is not None:
        return x * keep_mask(site, x.shape, module.p, x.device)
    if in_kernel(x) and x.dtype == torch.float32 and module.p < 1:
        return _DropoutRng.apply(x, rng_state(x.device), site_id(site), float(module.p))
    return module(x)


def apply_p(x: torch.Tensor, p: float, site: str) -> torch.Tensor:
    """F.dropout(x, p, training=True) with the mask of `site`: in-kernel generator on the GPU, dictated under `injected`, else torch's"""
    if p <= 0:
        return x
    if _inject is not None:
        return x * keep_mask(site, x.shape, p, x.device)
    if in_kernel(x) and x.dtype == torch.float32 and p < 1:
        return _DropoutRng.apply(x, rng_state(x.device), site_id(site), float(p))
    return torch.nn.functional.dropout(x, p, True)
