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

from . import hip


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


class DecodeEngine:
    """Binds weights + one batch of clip features to preallocated state and a launch list."""
    _warm = set()

    def __init__(self, weights: DecodeWeights, feats: Dict[str, torch.Tensor], T: int, unk_idx: int, beam: int = 1,
                 inv_temp: float = 1.0, own_features: bool = False, path: str = "auto", gate_ksplit: Optional[bool] = None,
                 driver: bool = True, gsk: Optional[bool] = None, embgate: Optional[bool] = None, lang_ksx: Optional[bool] = None):
        """driver: enqueue the decode through the C-ABI drivers cvc_decode_greedy / cvc_decode_beam (one host call per decode);
        False walks the launch list in Python (one ctypes call per kernel; tests compare the two).
        embgate: packed path only -- the embedding-gate schedule (the embedded word's share of the att-LSTM gates is a row of
        a per-checkpoint table: 34 MB less to stream per step at config 2, and the gate GEMM no longer waits for the word).  None = on when the table fits EMBGATE_MAX_BYTES; tests compare on / off.
        lang_ksx: packed path, R = 2048 -- the language cell on the K-split gate GEMM with the exchange finish
        (cvc_packed_lstm_ksx_fwd: activations read once per 256 gate rows instead of once per 32; the tile's 8 K slices
        exchange their partial tiles inside the launch).  None = off (measured no faster inside the decode graph; CVC_LANG_KSX=1:
        on); the exchange's error word is checked after the first decode and the engine re-binds without it if it is set.
        gsk: packed path only -- True selects the grouped stream-K schedule (csrc/gemm_gsk.hip; measured slower than the
        embedding-gate schedule, kept selectable and tested; needs R % 64 == 0, split-product arithmetic).
        path: "auto" picks packed (greedy, <= 64 rows) / tile (> 64 rows or beams) / ring (odd widths); "ring" forces the
        row-major fallback kernels (tests compare the paths).
        own_features: keep private copies of the clip features, so that the bound launch list (and a captured HIP
        graph) can be reused for the next batch of the same shape through load_features()."""
        W = self.W = weights
        self.T, self.unk, self.beam = int(T), int(unk_idx), int(beam)
        fc, conv, pconv = feats["fc_feats"], feats["conv_feats"], feats["p_conv_feats"]
        pool, ppool = feats["pool_feats"], feats["p_pool_feats"]
        mask = feats["pnt_mask"][:, 1:] if feats["pnt_mask"].shape[1] == pool.shape[1] + 1 else feats["pnt_mask"]
        self.B, self.N, self.F = pool.shape[0], pool.shape[1], conv.shape[1]
        B, N, Fr, R, A, V = self.B, self.N, self.F, W.R, W.A, W.V
        dev = pool.device
        for name, t, shape in (("fc_feats", fc, (B, R)), ("conv_feats", conv, (B, Fr, R)), ("p_conv_feats", pconv, (B, Fr, A)),
                               ("pool_feats", pool, (B, N, R)), ("p_pool_feats", ppool, (B, N, A))):
            if tuple(t.shape) != shape:
                raise RuntimeError(f"DecodeEngine: {name} has shape {tuple(t.shape)}, expected {shape}")
            hip._dev(t, name=name)
        self.mask = hip._mask(mask)
        if own_features:
            fc, conv, pconv, pool, ppool = (t.clone() for t in (fc, conv, pconv, pool, ppool))
            self.mask = self.mask.clone()
        self.own_features = own_features
        self.feats = (fc, conv, pconv, pool, ppool)
        nb = lambda t: t.numel() * t.element_size()
        rows = self.rows = B * self.beam
        f32 = dict(device=dev, dtype=torch.float32)
        z = lambda *s: torch.zeros(*s, **f32)
        # ping-pong recurrent state: index t & 1 is read, (t+1) & 1 is written
        self.h_att, self.c_att = [z(rows, R), z(rows, R)], [z(rows, R), z(rows, R)]
        self.h_lang, self.c_lang = [z(rows, R), z(rows, R)], [z(rows, R), z(rows, R)]
        self.q = z(rows, A)
        self.scores_r, self.scores_f = z(rows, N), z(rows, Fr)
        self.attn_f = z(rows, Fr)
        self.ctx_sum = z(rows, R)
        self.logits = z(rows, V)
        self.gate_fc = z(rows, 4 * R)      # step-invariant part of the att-LSTM gates: fc x W_ih[:, R:2R] + b_ih + b_hh
        self.QSPLIT = 8                    # h2attn runs split-K over the chip; attn_scores sums the slices
        self.q_parts = z(self.QSPLIT, rows, A)
        self.emb = z(rows, W.E)            # relu(Emb[word_t]), written by the word-selection kernel of step t-1
        self.top2_part = z((V + 31) // 32, 64, 6)
        self.att_steps = z(self.T, rows, N)                       # post-softmax region attention per step
        self.words = torch.zeros(self.T + 1, rows, dtype=torch.int64, device=dev)   # words[0] = BOS = 0
        self.logprob = z(self.T, rows)
        # fc is per clip; beams of a clip read the same row through a row-gather index
        self.fc = fc
        self.clip_of_row = torch.arange(rows, device=dev, dtype=torch.int64) // self.beam
        if self.beam > 1:
            self.score = z(2, rows)
            self.done = torch.zeros(2, rows, dtype=torch.uint8, device=dev)
            self.parent = torch.zeros(self.T, rows, dtype=torch.int64, device=dev)
            self.bt_seq = torch.zeros(self.B, self.T, dtype=torch.int64, device=dev)      # rank-0 hypothesis (cvc_beam_backtrack)
            self.bt_att = z(self.B, self.T, N)
            self.gather_tmp = [z(rows, R) for _ in range(4)]
            self.beam_ws = z(17 * rows)
        self.inv_temp = float(inv_temp)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self._keep: List = []
        self.packed = self.beam == 1 and rows <= 64 and R % 32 == 0 and W.E % 32 == 0 and A % 32 == 0
        # packed path: K-split gate GEMMs (activations shared through LDS, csrc/gemm_packed_ks.hip) where the shape allows
        # (True: partial tiles + a finishing launch; "fused": the last-arriving K slice of a tile finishes it in the same launch)
        self.gate_ksplit = GATE_KSPLIT_DEFAULT if gate_ksplit is None else gate_ksplit
        self.gate_fused = self.gate_ksplit == "fused"
        self.gate_ksplit = bool(self.gate_ksplit)
        gsk_ok = self.packed and R % 64 == 0 and not self.gate_ksplit and hip.gemm_packed_split(-1) == 2
        if gsk and not gsk_ok:
            raise RuntimeError("DecodeEngine: the stream-K schedule needs the packed path, R % 64 == 0 and cvc_gemm_packed_split(2)")
        self.gsk = False if gsk is None else bool(gsk)
        # more than 64 live rows (beam search, big greedy batches): bf16-fragment tile GEMMs (csrc/gemm_tile.hip)
        self.tile = (not self.packed) and (self.beam > 1 or rows > 64) and R % 16 == 0 and W.E % 16 == 0 and path != "ring"
        eg_ok = (self.packed and not self.gsk and not self.gate_ksplit) or self.tile
        if embgate and not eg_ok:
            raise RuntimeError("DecodeEngine: the embedding-gate schedule needs the packed path (without gsk / gate_ksplit) or the tile path")
        self.embgate = (eg_ok and 4 * V * 4 * R <= EMBGATE_MAX_BYTES) if embgate is None else bool(embgate)
        # what stays in the Infinity Cache between steps: small linear weights, then (embedding-gate schedule on the packed path) the
        # attention cell's gate matrix over K = 2R if it fits, then the largest subset of the feature tensors
        keep = cache_plan(4 * (V * R + A * R), {"ppool": nb(ppool), "pconv": nb(pconv), "pool": nb(pool), "conv": nb(conv)},
                          gate_weight_bytes=4 * 4 * R * 2 * R if (self.packed and self.embgate and rows > 32) else None)
        self.att_w_cached = bool(keep.get("att_w", False))
        # cvc_attn_set.stream: bit 0 = proj read non-temporally, bit 1 = ctx
        self.stream_r = (0 if keep["ppool"] else 1) | (0 if keep["pool"] else 2)
        self.stream_f = (0 if keep["pconv"] else 1) | (0 if keep["conv"] else 2)
        self._plan = None
        self._driver = driver
        ksx_ok = (self.packed and not self.gsk and not self.gate_ksplit and R == 2048 and self.T > 1 and
                  hip.gemm_packed_split(-1) == 2 and int(hip.lib().cvc_packed_lstm_ks_slices(3 * R, R)) == 8)
        if lang_ksx and not ksx_ok:
            raise RuntimeError("DecodeEngine: lang_ksx needs the packed path at R = 2048, T > 1, split-product arithmetic")
        self.lang_ksx = ksx_ok and (LANG_KSX_DEFAULT if lang_ksx is None else bool(lang_ksx))
        self._ksx_checked = False
        if self.lang_ksx:
            self.ksx_slab = torch.empty(8 * (R // 8) * 2048, device=dev, dtype=torch.float32)
            self.ksx_flags = torch.zeros(R // 8 + 1, device=dev, dtype=torch.int32)
        if self.packed:
            self._alloc_packed()
            self._launches = self._build_packed()
        elif self.tile:
            self._alloc_tile()
            self._launches = self._build_tile()
        else:
            self._launches = self._build()
        if driver and (self.tile or (self.packed and not (self.ks_att or self.ks_lang))):
            self._bind_driver()

    # ------------------------------------------------------------------ C-ABI decode driver (csrc/decode_driver.hip)
    def _bind_driver(self):
        """Bind every buffer of this engine into a cvc_decode_desc and create the plan: run() / capture() then enqueue the
        whole decode with ONE call (cvc_decode_greedy / cvc_decode_beam) instead of walking the launch list in Python.  The
        Python launch list stays for run_timed() (per-launch HIP events) and as the reference the driver is tested against."""
        W, L = self.W, hip.lib()
        ptr = lambda t: None if t is None else t.data_ptr()
        d = hip.DecodeDesc()
        d.B, d.beam, d.T, d.N, d.F, d.R, d.A, d.E, d.V = self.B, self.beam, self.T, self.N, self.F, W.R, W.A, W.E, W.V
        d.unk_idx, d.attn_kind, d.inv_temp = self.unk, W.kind, self.inv_temp
        d.stream_r, d.stream_f = self.stream_r, self.stream_f
        for k in ("b_ih_att", "b_hh_att", "b_ih_lang", "b_hh_lang", "b_h", "w_a", "b_a", "b_o", "embed"):
            setattr(d, k, ptr(getattr(W, k)))
        fc, conv, pconv, pool, ppool = self.feats
        d.fc, d.conv, d.pconv, d.pool, d.ppool, d.mask = ptr(fc), ptr(conv), ptr(pconv), ptr(pool), ptr(ppool), ptr(self.mask)
        d.words, d.att_steps, d.logprob = ptr(self.words), ptr(self.att_steps), ptr(self.logprob)
        d.scores_r, d.scores_f, d.attn_f = ptr(self.scores_r), ptr(self.scores_f), ptr(self.attn_f)
        if self.packed:
            R = W.R
            d.path, d.qsplit = 0, self.QSPLIT
            d.w_att, d.w_lang, d.w_h, d.w_o = ptr(W.p_att), ptr(W.p_lang), ptr(W.p_h), ptr(W.p_o)
            w_fc = W.w_ih_att[:, R:2 * R]
            d.w_fc, d.ld_w_fc = w_fc.data_ptr(), w_fc.stride(0)
            d.gate_fc, d.q_parts, d.top2_part = ptr(self.gate_fc), ptr(self.q_parts), ptr(self.top2_part)
            for name, bufs in (("xa", self.XA), ("xl", self.XL), ("ca", self.cA), ("cl", self.cL)):
                arr = getattr(d, name)
                arr[0], arr[1] = ptr(bufs[0]), ptr(bufs[1])
            d.xa0_init = ptr(self.XA0_init)
            if self.embgate:
                d.w_att = ptr(W.p_att2)
                d.emb_gate, d.sel_counter = ptr(W.t_embgate), ptr(self.sel_counter)
                d.att_w_cached = int(self.att_w_cached)
            if self.lang_ksx:
                d.lang_ksx, d.ksx_slab, d.ksx_flags = 1, ptr(self.ksx_slab), ptr(self.ksx_flags)
            if self.gsk:
                d.gsk_nwg = self.gsk_nwg
                d.slab_att, d.slab_lang, d.slab_q, d.slab_o = (ptr(self.slab_att), ptr(self.slab_lang), ptr(self.slab_q),
                                                              ptr(self.slab_o))
        else:
            d.path = 1
            d.ks_gate, d.ks_q, d.ks_o, d.ks_fc = self.ks_gate, self.ks_q, self.ks_o, self.ks_fc
            d.w_att, d.w_lang, d.w_h, d.w_o, d.w_fc_frag = ptr(W.t_att2 if self.embgate else W.t_att), ptr(W.t_lang), ptr(W.t_h), ptr(W.t_o), ptr(W.t_fc)
            if self.embgate:
                d.emb_gate = ptr(W.t_embgate)
            d.gate_fc, d.q, d.q_parts, d.logits = ptr(self.gate_fc_clip), ptr(self.q), ptr(self.parts_q), ptr(self.logits)
            for name, t in (("xaf", self.XAf), ("xlf", self.XLf), ("xhf", self.XHf), ("xff", self.XFf)):
                p_, s_ = hip._frag_ptr(t)
                setattr(d, name, p_)
                setattr(d, name + "_stride", s_)
            d.parts_gate, d.parts_o, d.parts_fc = ptr(self.parts_gate), ptr(self.parts_o), ptr(self.parts_fc)
            d.h_att, d.c_att, d.h_lang, d.c_lang = ptr(self.t_h_att), ptr(self.t_c_att), ptr(self.t_h_lang), ptr(self.t_c_lang)
            d.c_att_prev, d.c_lang_prev, d.zero_state = ptr(self.t_c_att_prev), ptr(self.t_c_lang_prev), ptr(self.t_zero)
            if self.beam > 1:
                d.score, d.done, d.parent, d.beam_ws = ptr(self.score), ptr(self.done), ptr(self.parent), ptr(self.beam_ws)
        plan = C.c_void_p()
        hip._check(L.cvc_decode_plan_create(C.byref(d), C.byref(plan)), "cvc_decode_plan_create")
        self._desc, self._plan = d, plan
        self._plan_call = L.cvc_decode_beam if self.beam > 1 else L.cvc_decode_greedy

    def __del__(self):
        plan = getattr(self, "_plan", None)
        if plan is not None and plan.value:
            try:
                hip.lib().cvc_decode_plan_destroy(plan)
            except Exception:
                pass
            self._plan = None

    def _run_driver(self):
        hip._check(self._plan_call(self._plan, torch.cuda.current_stream().cuda_stream), "cvc_decode_greedy/beam")

    # ------------------------------------------------------------------ packed path (greedy, rows <= 64)
    def _alloc_packed(self):
        """Fragment-native operands: packed weight copies (once per checkpoint binding) and the
        recurrent activations as two ping-pong concat buffers in quad layout:
          XA = [h_lang(t-1) | relu(Emb[word_t]) | h_att(t-1)]   (att-LSTM input, K = 2R + E)
          XL = [ctx_regions + ctx_frames | h_att(t) | h_lang(t-1)]  (lang-LSTM input, K = 3R)"""
        W, R, E = self.W, self.W.R, self.W.E
        dev = self.fc.device
        if not hasattr(W, "p_att"):
            W.p_att = pack_weights(torch.cat([W.w_ih_att[:, 0:R], W.w_ih_att[:, 2 * R:2 * R + E], W.w_hh_att], 1), R)
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

    def _lang_ksx_launch(self, t, XL_r, XA_w, XL_w, rd, wr):
        """The language cell of step t on cvc_packed_lstm_ksx_fwd (same operands and destinations as the full-K launch)."""
        L, W, R = hip.lib(), self.W, self.W.R
        ptr = lambda x: None if x is None else x.data_ptr()
        qoff = lambda buf, k0: buf.data_ptr() + (k0 // 4) * 64 * 4 * 4
        return ("lang_lstm", L.cvc_packed_lstm_ksx_fwd, (ptr(W.p_lang), ptr(XL_r), 3 * R, ptr(W.b_ih_lang), ptr(W.b_hh_lang), None, None, None,
                                                          ptr(self.cL[rd]), self.rows, R, ptr(XA_w), qoff(XL_w, 2 * R), ptr(self.cL[wr]),
                                                          ptr(self.ksx_slab), ptr(self.ksx_flags), t + 1))

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
    @staticmethod
    def _ksplit(n_out: int, K: int) -> int:
        """K slices of a tile GEMM: enough workgroups (128-row tiles x slices) to cover the 256 CUs, at least 8 k steps per
        slice; 4 / 8 slices preferred (they map onto whole XCDs)."""
        ntile = (n_out + 127) // 128
        ks = max(1, min(256 // ntile, (K // 16) // 8))
        for p in (8, 4, 2):
            if ks >= p and ks < 2 * p and ntile * p >= 192:
                return p
        return ks

    def _alloc_tile(self):
        """Weight fragments (once per checkpoint binding) and the activation fragment buffers of the tile path:
          XA = [h_lang(t-1) | relu(Emb[word_t]) | h_att(t-1)]       att-LSTM input, K = 2R + E
          XL = [ctx_regions + ctx_frames | h_att(t) | h_lang(t-1)]   lang-LSTM input, K = 3R (h2attn reads its middle segment)
          XH = h_lang(t)                                             vocabulary head input"""
        W, R, E, A, V = self.W, self.W.R, self.W.E, self.W.A, self.W.V
        dev = self.fc.device
        if self.embgate and not hasattr(W, "t_att2"):
            W.t_att2 = pack_weights_tile(torch.cat([W.w_ih_att[:, 0:R], W.w_hh_att], 1), R)     # K = 2R: [h_lang | h_att]
            if not hasattr(W, "t_embgate"):
                W.t_embgate = embgate_table(W)
        if not self.embgate and not hasattr(W, "t_att"):
            W.t_att = pack_weights_tile(torch.cat([W.w_ih_att[:, 0:R], W.w_ih_att[:, 2 * R:2 * R + E], W.w_hh_att], 1), R)
        if not hasattr(W, "t_lang"):
            W.t_lang = pack_weights_tile(torch.cat([W.w_ih_lang, W.w_hh_lang], 1), R)
            W.t_h = pack_weights_tile(W.w_h)
            W.t_o = pack_weights_tile(W.w_o)
            W.t_fc = pack_weights_tile(W.w_ih_att[:, R:2 * R].contiguous())
        rows, B = self.rows, self.B
        ra, rb = hip.tile_rows_alloc(rows), hip.tile_rows_alloc(B)
        zf = lambda r, k: torch.zeros(r // 32, k // 16, 3, 2, 32, 8, device=dev, dtype=torch.int16)
        self.ka_tile = 2 * R if self.embgate else 2 * R + E
        self.XAf, self.XLf, self.XHf, self.XFf = zf(ra, self.ka_tile), zf(ra, 3 * R), zf(ra, R), zf(rb, R)
        f32 = dict(device=dev, dtype=torch.float32)
        self.ks_gate, self.ks_q, self.ks_o, self.ks_fc = (self._ksplit(4 * R, min(self.ka_tile, 3 * R)), self._ksplit(A, R),
                                                          self._ksplit(V, R), self._ksplit(4 * R, R))
        self.parts_gate = torch.empty(self.ks_gate, rows, 4 * R, **f32)
        self.parts_q = torch.empty(self.ks_q, rows, A, **f32)
        self.parts_o = torch.empty(self.ks_o, rows, V, **f32)
        self.parts_fc = torch.empty(self.ks_fc, B, 4 * R, **f32)
        self.gate_fc_clip = torch.empty(B, 4 * R, **f32)
        z = lambda: torch.zeros(rows, R, **f32)
        self.t_h_att, self.t_c_att, self.t_h_lang, self.t_c_lang = z(), z(), z(), z()        # state of the current step
        self.t_c_att_prev, self.t_c_lang_prev, self.t_zero = z(), z(), z()

    def _build_tile(self):
        L, W = hip.lib(), self.W
        B, N, Fr, R, A, E, V, rows, beam = self.B, self.N, self.F, W.R, W.A, W.E, W.V, self.rows, self.beam
        fc, conv, pconv, pool, ppool = self.feats
        ptr = lambda t: None if t is None else t.data_ptr()
        fp = hip._frag_ptr
        out = []
        # ---- once per decode: hoisted fc gate term (+ both biases), one row per clip; step-0 operands from the zero state
        xf_p, xf_s = fp(self.XFf)
        out.append(("gate_fc", L.cvc_tile_pack_rows, (ptr(fc), fc.stride(0), None, 0, B, R, xf_p, xf_s)))
        out.append(("gate_fc", L.cvc_tile_gemm, (ptr(W.t_fc), xf_p, xf_s, R, B, 4 * R, self.ks_fc, ptr(self.parts_fc), 4 * R,
                                                 B * 4 * R)))
        out.append(("gate_fc", L.cvc_tile_linear_finish, (ptr(self.parts_fc), self.ks_fc, B * 4 * R, 4 * R, ptr(W.b_ih_att),
                                                          ptr(W.b_hh_att), B, 4 * R, ptr(self.gate_fc_clip), 4 * R)))
        xa_p, xa_s = fp(self.XAf)
        xl_p, xl_s = fp(self.XLf)
        xl_hatt, _ = fp(self.XLf, R)
        xl_hlang, _ = fp(self.XLf, 2 * R)
        xh_p, xh_s = fp(self.XHf)
        zero = ptr(self.t_zero)
        eg = self.embgate
        E_pack = 0 if eg else E                        # embedding-gate form: no embedding segment in XA, the word enters in the finish
        out.append(("beam_reorder", L.cvc_tile_reorder_pack, (None, ptr(self.words[0]), beam, zero, zero, zero, zero, ptr(W.embed), E_pack, V,
                                                              ptr(self.t_c_att_prev), ptr(self.t_c_lang_prev), xa_p, xa_s, xl_hlang,
                                                              xl_s, rows, R)))
        for t in range(self.T):
            out.append(("att_lstm", L.cvc_tile_gemm, (ptr(W.t_att2 if eg else W.t_att), xa_p, xa_s, self.ka_tile, rows, 4 * R, self.ks_gate,
                                                      ptr(self.parts_gate), 4 * R, rows * 4 * R)))
            if eg:
                out.append(("att_finish", L.cvc_tile_lstm_finish_embgate, (ptr(self.parts_gate), self.ks_gate, rows * 4 * R, None, None,
                                                                           ptr(self.gate_fc_clip), beam, ptr(W.t_embgate), ptr(self.words[t]), V,
                                                                           ptr(self.t_c_att_prev), rows, R, ptr(self.t_c_att),
                                                                           ptr(self.t_h_att), xl_hatt, xl_s, None, 0)))
            else:
                out.append(("att_finish", L.cvc_tile_lstm_finish, (ptr(self.parts_gate), self.ks_gate, rows * 4 * R, None, None,
                                                                   ptr(self.gate_fc_clip), beam, ptr(self.t_c_att_prev), rows, R,
                                                                   ptr(self.t_c_att), ptr(self.t_h_att), xl_hatt, xl_s, None, 0)))
            out.append(("h2attn", L.cvc_tile_gemm, (ptr(W.t_h), xl_hatt, xl_s, R, rows, A, self.ks_q, ptr(self.parts_q), A, rows * A)))
            sets = (hip.AttnSet * 2)()
            sets[0] = hip.AttnSet(ptr(ppool), ptr(pool), ptr(self.mask), None, ptr(self.scores_r), None,
                                  ptr(self.att_steps[t]), None, N, self.stream_r)
            sets[1] = hip.AttnSet(ptr(pconv), ptr(conv), None, None, ptr(self.scores_f), None, ptr(self.attn_f), None, Fr,
                                  self.stream_f)
            # the query slabs are summed ONCE here: every one of a clip's ~19 score workgroups would otherwise re-sum
            # ks_q x beam x A partials on its own (measured: 106 -> us of the score pass were that)
            out.append(("h2attn_finish", L.cvc_tile_linear_finish, (ptr(self.parts_q), self.ks_q, rows * A, A, ptr(W.b_h), None, rows, A,
                                                                    ptr(self.q), A)))
            out.append(("attn_scores", L.cvc_attn_scores, (W.kind, ptr(self.q), ptr(W.w_a), ptr(W.b_a), self.inv_temp, sets, 2, B,
                                                           beam, A)))
            out.append(("attn_wsum", L.cvc_attn_wsum_frag, (sets, 2, B, beam, R, xl_p, xl_s)))
            out.append(("lang_lstm", L.cvc_tile_gemm, (ptr(W.t_lang), xl_p, xl_s, 3 * R, rows, 4 * R, self.ks_gate,
                                                       ptr(self.parts_gate), 4 * R, rows * 4 * R)))
            out.append(("lang_finish", L.cvc_tile_lstm_finish, (ptr(self.parts_gate), self.ks_gate, rows * 4 * R, ptr(W.b_ih_lang),
                                                                ptr(W.b_hh_lang), None, 1, ptr(self.t_c_lang_prev), rows, R,
                                                                ptr(self.t_c_lang), ptr(self.t_h_lang), xh_p, xh_s, None, 0)))
            out.append(("logits", L.cvc_tile_gemm, (ptr(W.t_o), xh_p, xh_s, R, rows, V, self.ks_o, ptr(self.parts_o), V, rows * V)))
            out.append(("logits_finish", L.cvc_tile_linear_finish, (ptr(self.parts_o), self.ks_o, rows * V, V, ptr(W.b_o), None, rows, V,
                                                                    ptr(self.logits), V)))
            if beam == 1:
                out.append(("word_select", L.cvc_top2_unk, (ptr(self.logits), rows, V, self.unk, ptr(self.words[t + 1]), 1,
                                                            ptr(self.logprob[t]))))
                parent = None
            else:
                srd, swr = t & 1, (t + 1) & 1
                out.append(("word_select", L.cvc_beam_select, (ptr(self.logits), ptr(self.score[srd]), ptr(self.done[srd]), B,
                                                               beam, V, self.unk, 1 if t == 0 else 0, ptr(self.parent[t]),
                                                               ptr(self.words[t + 1]), ptr(self.score[swr]),
                                                               ptr(self.done[swr]), ptr(self.beam_ws))))
                parent = ptr(self.parent[t])
            if t + 1 < self.T:
                out.append(("beam_reorder", L.cvc_tile_reorder_pack, (parent, ptr(self.words[t + 1]), beam, ptr(self.t_h_att),
                                                                      ptr(self.t_c_att), ptr(self.t_h_lang), ptr(self.t_c_lang),
                                                                      ptr(W.embed), E_pack, V, ptr(self.t_c_att_prev),
                                                                      ptr(self.t_c_lang_prev), xa_p, xa_s, xl_hlang, xl_s, rows, R)))
            self._keep.append(sets)
        return out

    # ------------------------------------------------------------------ launch list
    def _build(self):
        L, W = hip.lib(), self.W
        B, N, Fr, R, A, E, V, rows, beam = self.B, self.N, self.F, W.R, W.A, W.E, W.V, self.rows, self.beam
        fc, conv, pconv, pool, ppool = self.feats
        ptr = lambda t: None if t is None else t.data_ptr()
        out = []
        # fc_feats does not change over the T steps (decoder_core.py:46): its gate contribution and the
        # two bias vectors are computed once per decode, inside the timed/captured region
        seg_fc = _segs([(fc, self.clip_of_row if beam > 1 else None, W.w_ih_att[:, R:2 * R], False)])
        out.append(("gate_fc", L.cvc_linear_fwd, (seg_fc, 1, ptr(W.b_ih_att), ptr(W.b_hh_att), rows, 4 * R, ptr(self.gate_fc),
                                                  4 * R)))
        self._keep.append(seg_fc)
        if beam == 1 and rows <= 64:
            out.append(("embed_bos", L.cvc_embed_relu_fwd, (ptr(W.embed), ptr(self.words[0]), None, rows, E, ptr(self.emb))))
        for t in range(self.T):
            rd, wr = t & 1, (t + 1) & 1
            # att-LSTM: [h_lang(t-1) | relu(Emb[word_t])] x W_ih  +  h_att(t-1) x W_hh  + gate_fc
            fused_head = beam == 1 and rows <= 64
            seg_att = _segs([(self.h_lang[rd], None, W.w_ih_att[:, 0:R], False),
                             (self.emb, None, W.w_ih_att[:, 2 * R:2 * R + E], False) if fused_head else
                             (W.embed, self.words[t], W.w_ih_att[:, 2 * R:2 * R + E], True),
                             (self.h_att[rd], None, W.w_hh_att, False)])
            out.append(("att_lstm", L.cvc_lstm_cell_fwd, (seg_att, 3, None, None, ptr(self.gate_fc), ptr(self.c_att[rd]),
                                                          rows, R, ptr(self.h_att[wr]), ptr(self.c_att[wr]), None)))
            seg_q = _segs([(self.h_att[wr], None, W.w_h, False)])
            split_q = rows <= 64
            if split_q:
                out.append(("h2attn", L.cvc_linear_splitk_fwd, (seg_q, 1, None, rows, A, self.QSPLIT, ptr(self.q_parts))))
            else:
                out.append(("h2attn", L.cvc_linear_fwd, (seg_q, 1, ptr(W.b_h), None, rows, A, ptr(self.q), A)))
            sets = (hip.AttnSet * 2)()
            sets[0] = hip.AttnSet(ptr(ppool), ptr(pool), ptr(self.mask), None, ptr(self.scores_r), None,
                                  ptr(self.att_steps[t]), None, N, self.stream_r)
            sets[1] = hip.AttnSet(ptr(pconv), ptr(conv), None, None, ptr(self.scores_f), None, ptr(self.attn_f), None, Fr,
                                  self.stream_f)
            if split_q:
                out.append(("attn_scores", L.cvc_attn_scores_qparts, (W.kind, ptr(self.q_parts), self.QSPLIT, ptr(W.b_h),
                                                                      ptr(W.w_a), ptr(W.b_a), self.inv_temp, sets, 2, B, beam, A)))
            else:
                out.append(("attn_scores", L.cvc_attn_scores, (W.kind, ptr(self.q), ptr(W.w_a), ptr(W.b_a), self.inv_temp, sets,
                                                               2, B, beam, A)))
            out.append(("attn_wsum", L.cvc_attn_wsum, (sets, 2, B, beam, R, ptr(self.ctx_sum))))
            seg_lang = _segs([(self.ctx_sum, None, W.w_ih_lang[:, 0:R], False),
                              (self.h_att[wr], None, W.w_ih_lang[:, R:2 * R], False),
                              (self.h_lang[rd], None, W.w_hh_lang, False)])
            out.append(("lang_lstm", L.cvc_lstm_cell_fwd, (seg_lang, 3, ptr(W.b_ih_lang), ptr(W.b_hh_lang), None,
                                                           ptr(self.c_lang[rd]), rows, R, ptr(self.h_lang[wr]),
                                                           ptr(self.c_lang[wr]), None)))
            seg_o = _segs([(self.h_lang[wr], None, W.w_o, False)])
            if fused_head:
                # vocabulary projection with the top-2 / log-sum-exp partials in its epilogue ([B,V] logits are
                # never written), then merge + UNK rule + next step's embedded word
                out.append(("logits", L.cvc_linear_top2_fwd, (seg_o, 1, ptr(W.b_o), rows, V, None, ptr(self.top2_part))))
                out.append(("word_select", L.cvc_top2_final, (ptr(self.top2_part), (V + 31) // 32, rows, self.unk,
                                                              ptr(self.words[t + 1]), 1, ptr(self.logprob[t]), ptr(W.embed), E,
                                                              ptr(self.emb), E)))
            elif beam == 1:
                out.append(("logits", L.cvc_linear_fwd, (seg_o, 1, ptr(W.b_o), None, rows, V, ptr(self.logits), V)))
                out.append(("word_select", L.cvc_top2_unk, (ptr(self.logits), rows, V, self.unk, ptr(self.words[t + 1]), 1,
                                                            ptr(self.logprob[t]))))
            else:
                out.append(("logits", L.cvc_linear_fwd, (seg_o, 1, ptr(W.b_o), None, rows, V, ptr(self.logits), V)))
                srd, swr = t & 1, (t + 1) & 1
                out.append(("word_select", L.cvc_beam_select, (ptr(self.logits), ptr(self.score[srd]), ptr(self.done[srd]), B,
                                                               beam, V, self.unk, 1 if t == 0 else 0, ptr(self.parent[t]),
                                                               ptr(self.words[t + 1]), ptr(self.score[swr]),
                                                               ptr(self.done[swr]), ptr(self.beam_ws))))
                # reorder the freshly written state rows by parent (gather into tmp, copy back)
                for i, buf in enumerate((self.h_att[wr], self.c_att[wr], self.h_lang[wr], self.c_lang[wr])):
                    out.append(("beam_reorder", L.cvc_gather_rows, (ptr(buf), ptr(self.parent[t]), rows, beam, R,
                                                                    ptr(self.gather_tmp[i]))))
                    out.append(("beam_reorder", "copy", (buf, self.gather_tmp[i])))
            self._keep += [seg_att, seg_q, sets, seg_lang, seg_o]
        return out

    def _reset(self):
        if self.tile:
            self.words[0].zero_()
            if self.beam > 1:
                self.score.zero_()
                self.done.zero_()
            return
        if self.packed:
            self.XA[0].copy_(self.XA0_init)
            self.XL[0].zero_()
            self.cA[0].zero_()
            self.cL[0].zero_()
            self.words[0].zero_()
            return
        for bufs in (self.h_att, self.c_att, self.h_lang, self.c_lang):
            bufs[0].zero_()
        self.words[0].zero_()
        if self.beam > 1:
            self.score.zero_()
            self.done.zero_()

    def _run_launches(self, timers=None):
        """timers: optional dict name -> list of (start_event, end_event), filled per launch
        (HIP events on the launch stream; used by bench.py for per-kernel durations)."""
        stream = torch.cuda.current_stream().cuda_stream
        for name, fn, args in self._python_launches():
            if timers is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if fn == "copy":
                args[0].copy_(args[1])
            else:
                rc = fn(*args, stream)
                if rc != 0:
                    hip._check(rc, name)
            if timers is not None:
                e1.record()
                timers.setdefault(name, []).append((e0, e1))

    def load_features(self, feats: Dict[str, torch.Tensor]):
        """Next batch of the same shape into the engine's own feature buffers (own_features=True): ~0.2 ms of device
        copies at cfg2 instead of a new binding and a new graph capture (~6 ms)."""
        if not self.own_features:
            raise RuntimeError("DecodeEngine.load_features needs own_features=True")
        pool = feats["pool_feats"]
        mask = feats["pnt_mask"][:, 1:] if feats["pnt_mask"].shape[1] == pool.shape[1] + 1 else feats["pnt_mask"]
        for dst, src in zip(self.feats, (feats["fc_feats"], feats["conv_feats"], feats["p_conv_feats"], pool, feats["p_pool_feats"])):
            if dst.shape != src.shape:
                raise RuntimeError(f"DecodeEngine.load_features: shape {tuple(src.shape)} != bound {tuple(dst.shape)}")
            dst.copy_(src)
        self.mask.copy_(hip._mask(mask))
        return self

    def bind_features(self, feats: Dict[str, torch.Tensor]):
        """Next batch of the same shape WITHOUT copying it: the C-ABI plan (and this engine) is pointed at the caller's
        feature tensors.  Only for engines that run through the driver without a captured graph (a graph keeps the pointers it
        was captured with: use load_features there)."""
        if self._plan is None or self.graph is not None:
            raise RuntimeError("DecodeEngine.bind_features needs a driver-bound engine without a captured graph")
        pool = feats["pool_feats"]
        mask = feats["pnt_mask"][:, 1:] if feats["pnt_mask"].shape[1] == pool.shape[1] + 1 else feats["pnt_mask"]
        new = (feats["fc_feats"], feats["conv_feats"], feats["p_conv_feats"], pool, feats["p_pool_feats"])
        for dst, src, name in zip(self.feats, new, ("fc_feats", "conv_feats", "p_conv_feats", "pool_feats", "p_pool_feats")):
            if dst.shape != src.shape:
                raise RuntimeError(f"DecodeEngine.bind_features: shape {tuple(src.shape)} != bound {tuple(dst.shape)}")
            hip._dev(src, name=name)
        self.mask = hip._mask(mask)
        self.feats = new
        self.fc = new[0]
        hip._check(hip.lib().cvc_decode_plan_set_features(self._plan, *(t.data_ptr() for t in new), self.mask.data_ptr()),
                   "cvc_decode_plan_set_features")
        self._launches = None                       # the Python launch list holds the old pointers: rebuilt on demand
        return self

    def _python_launches(self):
        if self._launches is None:
            self._keep = []
            self._launches = self._build_packed() if self.packed else (self._build_tile() if self.tile else self._build())
        return self._launches

    def run_timed(self):
        """One eager decode with a HIP-event pair around every launch.  Returns name -> list of ms."""
        timers = {}
        self._reset()
        self._run_launches(timers)
        torch.cuda.synchronize()
        return {k: [a.elapsed_time(b) for a, b in v] for k, v in timers.items()}

    def _run_once(self):
        """One decode on the current stream: through the C-ABI driver when bound (it resets its state itself), else the
        Python launch list."""
        if self._plan is not None:
            self._run_driver()
        else:
            self._reset()
            self._run_launches()
        if self.beam > 1:                                  # rank-0 hypothesis: one launch (was ~60 indexing launches per decode)
            hip._check(hip.lib().cvc_beam_backtrack(self.words[1:].data_ptr(), self.parent.data_ptr(), self.att_steps.data_ptr(),
                                                    self.B, self.beam, self.T, self.N, self.bt_seq.data_ptr(),
                                                    self.bt_att.data_ptr(), hip._stream()), "cvc_beam_backtrack")

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

    def capture(self):
        """Capture the T-step loop into a HIP graph (launch-bound inner loop -> one replay)."""
        if self.lang_ksx and not self._ksx_checked:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._run_once()
            torch.cuda.current_stream().wait_stream(s)
            self.check_ksx()
        key = (self.packed, self.tile, self.beam > 1)
        if key not in DecodeEngine._warm:                 # first capture of this path in the process: run once outside capture
            s = torch.cuda.Stream()                       # (module load, lazy init); later engines skip the extra decode
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._run_once()
            torch.cuda.current_stream().wait_stream(s)
            DecodeEngine._warm.add(key)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._run_once()
        self.graph = g
        return self

    def run(self):
        """One full T-step decode.  Returns (seq [B,T] int64, att2_weights [B,T,N]) -- views of
        engine-owned buffers (clone to keep across runs)."""
        if self.graph is not None:
            self.graph.replay()
        else:
            self._run_once()
            if self.lang_ksx and not self._ksx_checked and self.check_ksx():
                self._run_once()                           # the fallback's results
        if self.beam == 1:
            return self.words[1:].t(), self.att_steps.permute(1, 0, 2)
        return self._backtrack()

    def _backtrack(self):
        """Rank-0 hypothesis of every clip (cvc_beam_backtrack, enqueued with the decode) and the final beam scores."""
        return self.bt_seq, self.bt_att, self.score[self.T & 1].view(self.B, self.beam)

    def _backtrack_host(self):
        """The same by indexing on the host side of torch (kept as the cross-check of the kernel in the tests)."""
        B, beam, T, N = self.B, self.beam, self.T, self.N
        words = self.words[1:].view(T, B, beam)
        parent = self.parent.view(T, B, beam)
        att = self.att_steps.view(T, B, beam, N)
        k = torch.zeros(B, dtype=torch.int64, device=words.device)
        ar = torch.arange(B, device=words.device)
        seq, atts = [], []
        for t in range(T - 1, -1, -1):
            seq.append(words[t, ar, k])
            k_parent = parent[t, ar, k]
            atts.append(att[t, ar, k_parent])     # attention was computed for the parent row at step t
            k = k_parent
        seq.reverse()
        atts.reverse()
        final_scores = self.score[self.T & 1].view(B, beam)
        return torch.stack(seq, 1), torch.stack(atts, 1), final_scores
