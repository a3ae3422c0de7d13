"""Inference decode driver: the T-step greedy / beam caption loop as a flat, pre-bound list of
kernel launches (reference model/captioner.py:384-443, `_sample`).

Per step (7 launches, nothing returns to the host, no allocation):
  att-LSTM  : concat-GEMM over [h_lang(t-1) | relu(Emb[word])] + h_att(t-1) + hoisted fc gate term, fused cell update
  h2attn    : q = W_h h_att + b_h
  attention : score pass over p_pool/p_conv, softmax + weighted-sum pass over pool/conv
  lang-LSTM : concat-GEMM over [ctx_regions + ctx_frames | h_att] + h_lang(t-1), fused cell update
  logits    : W_o h_lang + b_o
  word      : top-2 with UNK suppression (greedy) or beam selection + state gather
Greedy with <= 64 rows and R % 64 == 0 runs the same arithmetic as the grouped stream-K schedule (csrc/gemm_gsk.hip,
csrc/decode_driver.hip::run_packed_gsk): the K ranges of a gate GEMM that do not depend on the step's critical path (h_lang,
h_att) are multiplied one launch early, in the same balanced launch as the small GEMM of that moment (logits / h2attn), and
the late launch (embedded word / attended context) sums their partial tiles -- still 7 launches per step.
The whole loop can be captured once into a HIP graph (torch.cuda.CUDAGraph) and replayed.
Dropout is inactive (model.eval(), trainer.py:158).  State buffers ping-pong so that no kernel
writes a tensor another workgroup of the same launch still reads.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import torch

from .. import hip


class DecodeWeights:
    """Flat views of the hot-path parameters under the reference's state_dict names."""

    def __init__(self, sd: Dict[str, torch.Tensor], softattn_type: str = "additive"):
        g = lambda k: sd[k].detach().contiguous()
        self.w_ih_att, self.w_hh_att = g("decoder_core.att_lstm.weight_ih"), g("decoder_core.att_lstm.weight_hh")
        self.b_ih_att, self.b_hh_att = g("decoder_core.att_lstm.bias_ih"), g("decoder_core.att_lstm.bias_hh")
        self.w_ih_lang, self.w_hh_lang = g("decoder_core.lang_lstm.weight_ih"), g("decoder_core.lang_lstm.weight_hh")
        self.b_ih_lang, self.b_hh_lang = g("decoder_core.lang_lstm.bias_ih"), g("decoder_core.lang_lstm.bias_hh")
        self.w_h, self.b_h = g("decoder_core.soft_attn.h2attn.weight"), g("decoder_core.soft_attn.h2attn.bias")
        self.kind = hip.ATTN_ADDITIVE if softattn_type == "additive" else hip.ATTN_DOT
        if self.kind == hip.ATTN_ADDITIVE:
            self.w_a = g("decoder_core.soft_attn.alpha_net.weight").reshape(-1)
            self.b_a = g("decoder_core.soft_attn.alpha_net.bias")
        else:
            self.w_a = self.b_a = None
        self.embed = g("embed.0.weight")
        self.w_o, self.b_o = g("logit.weight"), g("logit.bias")
        self.R = self.w_hh_att.shape[1]
        self.A = self.w_h.shape[0]
        self.E = self.embed.shape[1]
        self.V = self.w_o.shape[0]
        for t in vars(self).values():
            if isinstance(t, torch.Tensor) and (not t.is_cuda or t.dtype != torch.float32):
                raise RuntimeError("DecodeWeights: parameters must be fp32 tensors on the GPU (no CPU fallback)")


def _segs(items):
    arr = (hip.GemmSeg * len(items))()
    for i, (x, idx, w, relu) in enumerate(items):
        arr[i] = hip.GemmSeg(x.data_ptr(), None if idx is None else idx.data_ptr(), w.data_ptr(), w.shape[1], x.stride(0),
                             w.stride(0), 1 if relu else 0)
    return arr


def pack_weights(w: torch.Tensor, lstm_R: Optional[int] = None, pad_quads: int = 0) -> torch.Tensor:
    """[Nout, K] row-major -> MFMA-fragment-native [ceil(Nout/32)][K/4 (+ pad_quads)][32][4] (include/cvc_hip.h, "Packed
    path").  For an LSTM gate matrix (Nout = 4R) block b holds the 4 gates of hidden units 8b..8b+7.  pad_quads unused
    quads per block stagger the blocks in HBM (cvc_packed_lstm_ks_fwd's w_blk_stride)."""
    n, k = w.shape
    assert k % 32 == 0, k
    if lstm_R is not None:
        R = lstm_R
        assert n == 4 * R and R % 8 == 0
        i = torch.arange(32, device=w.device)
        rows = ((i >> 3) * R + (i & 7)).view(1, 32) + (torch.arange(R // 8, device=w.device) * 8).view(-1, 1)
        w = w[rows.reshape(-1)]
        nb = R // 8
    else:
        nb = (n + 31) // 32
        if nb * 32 != n:
            w = torch.cat([w, w.new_zeros(nb * 32 - n, k)], 0)
    out = w.view(nb, 32, k // 4, 4).permute(0, 2, 1, 3)
    if pad_quads:
        padded = w.new_zeros(nb, k // 4 + pad_quads, 32, 4)
        padded[:, :k // 4] = out
        return padded
    return out.contiguous()


def lstm_packed_rows(R: int, device) -> torch.Tensor:
    """checkpoint row of every packed gate row: packed row 32 b + i is row (i >> 3) * R + 8 b + (i & 7) of a [4R, K] gate matrix"""
    i = torch.arange(32, device=device)
    return (((i >> 3) * R + (i & 7)).view(1, 32) + (torch.arange(R // 8, device=device) * 8).view(-1, 1)).reshape(-1)


EMBGATE_MAX_BYTES = 1 << 30      # largest embedding-gate table the engine builds on its own (cfg2: 164 MB, cfg5: 328 MB)


def embgate_table(W: "DecodeWeights") -> torch.Tensor:
    """[V, 4R] table of cvc_packed_lstm_embgate_fwd / cvc_tile_lstm_finish_embgate: row v = relu(Emb[v]) x W_ih_att[:, emb
    columns]^T (the xt segment of decoder_core.py:45-48 with xt = embed(it), captioner.py:53-68 in eval mode), gates in checkpoint
    order.  One dense product per checkpoint binding on the tile GEMM (split products, fp32-grade) -- no library GEMM."""
    R, E = W.R, W.E
    return hip.tile_mm(torch.relu(W.embed), W.w_ih_att[:, 2 * R:2 * R + E])                          # [V, 4R]


def to_quad(x: torch.Tensor) -> torch.Tensor:
    """[M<=64, K] row-major -> activation quad layout [K/4][64][4] (rows beyond M are zero)."""
    m, k = x.shape
    out = x.new_zeros(k // 4, 64, 4)
    out[:, :m] = x.view(m, k // 4, 4).permute(1, 0, 2)
    return out


def from_quad(xq: torch.Tensor, m: int) -> torch.Tensor:
    return xq[:, :m].permute(1, 0, 2).reshape(m, -1)


# ------------------------------------------------------------------ tile path operands (csrc/gemm_tile.hip)
def split3_bf16(x: torch.Tensor):
    """fp32 -> the three bf16 terms of the split-product arithmetic as int16 bit patterns (hi, mid, lo): truncation,
    both remainders exact (csrc/gemm_split.h)."""
    def top(v):
        return (v.view(torch.int32) & -65536).view(torch.float32)
    hi = top(x)
    r1 = x - hi
    mid = top(r1)
    lo = top(r1 - mid)
    bits = lambda v: (v.view(torch.int32) >> 16).to(torch.int16)
    return bits(hi), bits(mid), bits(lo)


def to_frag(x: torch.Tensor, rows_alloc: Optional[int] = None) -> torch.Tensor:
    """[M, K] fp32 row-major -> fragments [rows_alloc/32][K/16][3 terms][2 k halves][32 rows][8 k] (int16 bit patterns of
    bf16); rows beyond M are zero.  The layout the tile GEMM reads (include/cvc_hip.h, "Tile path")."""
    m, k = x.shape
    assert k % 16 == 0, k
    ra = rows_alloc if rows_alloc is not None else (m + 31) // 32 * 32
    xp = x.new_zeros(ra, k)
    xp[:m] = x
    planes = torch.stack(split3_bf16(xp.contiguous()), 0)                       # [3, ra, k]
    return planes.view(3, ra // 32, 32, k // 16, 2, 8).permute(1, 3, 0, 4, 2, 5).contiguous()


def from_frag(xb: torch.Tensor, m: int) -> torch.Tensor:
    """inverse of to_frag (sum of the three terms)."""
    nb, ks = xb.shape[0], xb.shape[1]
    f = (xb.to(torch.int32) << 16).view(torch.float32)                           # [nb, ks, 3, 2, 32, 8]
    x = f.sum(2).permute(0, 3, 1, 2, 4).reshape(nb * 32, ks * 16)               # [nb, 32, ks, 2, 8]
    return x[:m]


def pack_weights_tile(w: torch.Tensor, lstm_R: Optional[int] = None) -> torch.Tensor:
    """[Nout, K] row-major fp32 -> tile-GEMM weight fragments [ceil(Nout/128)*4][K/16][3][2][32][8] (bf16 bit patterns),
    zero rows beyond Nout.  LSTM gate matrices use the packed row order of `pack_weights` (block b = 4 gates x hidden
    units 8b..8b+7), so that a 128-row tile holds complete hidden units."""
    n, k = w.shape
    assert k % 16 == 0, k
    if lstm_R is not None:
        R = lstm_R
        assert n == 4 * R and R % 8 == 0
        i = torch.arange(32, device=w.device)
        rows = ((i >> 3) * R + (i & 7)).view(1, 32) + (torch.arange(R // 8, device=w.device) * 8).view(-1, 1)
        w = w[rows.reshape(-1)]
    return to_frag(w, (n + 127) // 128 * 128)


GATE_KSPLIT_DEFAULT = False   # measured (profiles/README.md, r02): the K-split kernel itself is 8-9 us faster per GEMM, its finishing launch costs the same
KS_PAD_QUADS = 0          # extra quads between the 32-row weight blocks of the K-split gate GEMM (measured: no effect; 0 = share the dense pack)

# bytes re-read every step that are left cacheable in the 256 MiB Infinity Cache (CVC_CACHE_BUDGET_MB: A/B override)
CACHE_BUDGET = int(os.environ.get("CVC_CACHE_BUDGET_MB", "208")) << 20
# The language cell on the K-split gate GEMM with the exchange finish.  Off by default: standalone (operands flushed from the caches
# between calls) it is 10 us faster than the full-K kernel (61.0 -> 51.3 us), inside the decode graph it is not (329.5 / 327.3 k
# steps/s without it, 326.1 / 323.5 k with it on one box: the 16.8 MB of partial tiles pass through the L2 / Infinity Cache that
# holds the attention cell's weights, whose launch slows down by 1.5 us).  CVC_LANG_KSX=1 or lang_ksx=True switches it on.
LANG_KSX_DEFAULT = os.environ.get("CVC_LANG_KSX", "0") == "1"
CACHE_GATE_WEIGHTS = os.environ.get("CVC_ATT_W_CACHED", "1") != "0"       # False: gate weights always stream (A/B)


def cache_plan(linear_weight_bytes: int, feature_bytes: Dict[str, int], budget: int = CACHE_BUDGET,
               gate_weight_bytes: Optional[int] = None) -> Dict[str, bool]:
    """Which per-step streams stay cacheable (True) and which are read non-temporally (False).

    A decode step re-reads the same ~0.85 GB; the Infinity Cache holds 256 MiB of it.  The small linear weights (vocabulary
    head, h2attn) always stay cacheable.  gate_weight_bytes: the attention cell's gate matrix in the embedding-gate schedule
    (key "att_w" of the result) -- it goes first when it fits next to them: the gate GEMM is bound by the latency of its
    weight loads, not by bandwidth, so a cached byte buys more there than in the attention passes, which stream at the
    memory's rate either way (measured at cfg2: its launch 40.3 -> 35.4 us; decode 322 -> 328 k steps/s).  The language
    cell's matrix (201 MB at cfg2) never fits and always streams.  Of the four feature tensors the subset with the most
    bytes that still fits the remaining room stays cacheable, the rest is marked `stream` in its cvc_attn_set.
    Measured at cfg2: nothing streamed 283 k steps/s, features only 302-304 k (round 2) / 322 k (round 3 kernels)."""
    names = list(feature_bytes)
    room = budget - linear_weight_bytes
    plan = {}
    if gate_weight_bytes is not None:
        plan["att_w"] = bool(CACHE_GATE_WEIGHTS and 0 < gate_weight_bytes <= room)
        if plan["att_w"]:
            room -= gate_weight_bytes
    best, best_bytes = (), 0
    for pick in range(1 << len(names)):
        chosen = [n for i, n in enumerate(names) if pick >> i & 1]
        tot = sum(feature_bytes[n] for n in chosen)
        if best_bytes < tot <= room:
            best, best_bytes = tuple(chosen), tot
    plan.update({n: n in best for n in names})
    return plan
