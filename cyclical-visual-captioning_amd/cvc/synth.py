"""Seeded synthetic weights / features / captions for the caption-decode hot path.

Everything here is a pure function of (seed, stream name, shape): a counter-based
generator (splitmix64 finaliser over the element index) implemented with numpy integer
arithmetic, so the bench, the tests, the oracle and the golden-vector script all see the
same numbers on any box without shipping weight files.

Shapes / names follow the reference checkpoint layout ("Hot-path
parameter inventory"); distributions follow SURVEY.md section 8(d).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from dataclasses import dataclass, asdict

import numpy as np

_MASK64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK64
        z = z ^ (z >> np.uint64(31))
    return z


def _stream_key(seed: int, stream: str) -> np.uint64:
    return np.uint64((int(seed) * 0x100000001B3 + zlib.crc32(stream.encode())) & 0xFFFFFFFFFFFFFFFF)


def _u01_range(key: np.uint64, lo: int, hi: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        ctr = (np.arange(lo, hi, dtype=np.uint64) * np.uint64(2) + np.uint64(1)) * np.uint64(0x2545F4914F6CDD1D)
        bits = _splitmix64((ctr ^ key) & _MASK64)
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


_CHUNK = 1 << 22


def _chunked(n: int, fn, dtype) -> np.ndarray:
    """fn(lo, hi) -> values of counters [lo, hi); large arrays are filled chunk by chunk on a few threads (numpy releases
    the GIL inside its loops).  The values depend on the counter only, so the result is identical for any chunking."""
    if n <= _CHUNK:
        return fn(0, n).astype(dtype, copy=False)
    import os
    from concurrent.futures import ThreadPoolExecutor
    out = np.empty(n, dtype=dtype)

    def work(lo):
        out[lo:min(lo + _CHUNK, n)] = fn(lo, min(lo + _CHUNK, n))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(work, range(0, n, _CHUNK)))
    return out


def _key(seed: int, stream: str, lane: int = 0) -> np.uint64:
    return _splitmix64(np.asarray([_stream_key(seed, stream) + np.uint64(lane)], dtype=np.uint64))[0]


def _u01(n: int, seed: int, stream: str, lane: int = 0) -> np.ndarray:
    """n doubles in (0, 1), 53-bit mantissa, from counters [0, n)."""
    key = _key(seed, stream, lane)
    return _chunked(n, lambda lo, hi: _u01_range(key, lo, hi), np.float64)


def uniform(shape, seed: int, stream: str, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    key = _key(seed, stream)
    return _chunked(n, lambda a, b: (lo + (hi - lo) * _u01_range(key, a, b)).astype(np.float32), np.float32).reshape(shape)


def normal(shape, seed: int, stream: str) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    k0, k1 = _key(seed, stream, 0), _key(seed, stream, 1)

    def box_muller(a, b):
        u1, u2 = _u01_range(k0, a, b), _u01_range(k1, a, b)
        return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32)
    return _chunked(n, box_muller, np.float32).reshape(shape)


def randint(shape, seed: int, stream: str, lo: int, hi: int) -> np.ndarray:
    """integers in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    return (lo + np.floor(_u01(n, seed, stream) * (hi - lo)).astype(np.int64)).reshape(shape)


@dataclass
class Dims:
    """Hot-path dimensions (D == R, A == E == D/2 by default)."""
    B: int = 64     # clips
    N: int = 100    # regions per clip
    F: int = 480    # frames per clip (t_attn_size)
    R: int = 2048   # rnn_size (== BASELINE "D")
    A: int = 1024   # att_hid_size
    E: int = 1024   # input_encoding_size
    V: int = 5000   # vocab_size
    T: int = 20     # seq_length
    G: int = 2048   # vis_encoding_size (grounder feature width)
    DET: int = 431  # detect_size (visually groundable classes)
    K: int = 8      # padded GT boxes per clip (label glue only)

    def as_dict(self):
        return asdict(self)


CONFIGS = {
    # BASELINE.json configs[0..4] (SURVEY.md section 8(d) "Configs -> concrete dims")
    "cfg1": Dims(B=4, N=20, F=120, R=1024, A=512, E=512, V=5000, T=10),
    "cfg2": Dims(B=64, N=100, F=480, R=2048, A=1024, E=1024, V=5000, T=20),
    "cfg3": Dims(B=64, N=100, F=480, R=2048, A=1024, E=1024, V=5000, T=20),
    "cfg4": Dims(B=32, N=100, F=480, R=2048, A=1024, E=1024, V=5000, T=20),
    "cfg5": Dims(B=64, N=300, F=480, R=4096, A=2048, E=2048, V=5000, T=30),
    # tiny case used for the full golden fixture G1
    "tiny": Dims(B=3, N=7, F=5, R=32, A=16, E=16, V=50, T=4, G=24, DET=6, K=3),
}

UNK_IDX = 1


def hot_path_state_dict(d: Dims, seed: int = 1234, logit_gain: float = 8.0, embed_gain: float = 2.0,
                        alpha_gain: float = 4.0, loc_gain: float = 0.1,
                        softattn_type: str = "additive", vocab_plus_1: bool = False) -> "OrderedDict[str, np.ndarray]":
    """Random-init weights under the reference's state_dict key names.

    nn.LSTMCell / nn.Linear default init U(+-1/sqrt(fan)), nn.Embedding N(0,1)
    (SURVEY.md section 9.13).  Gains (logit, embedding, alpha_net, localizer query) are chosen
    so that random-init greedy captions are diverse and attention is neither uniform nor one-hot;
    otherwise parity checks on sequences / attention maps would be vacuous.
    Shared LSTM cells appear under both prefixes, as in the reference checkpoint
    (captioner.py:86-87).
    """
    R, A, E, V = d.R, d.A, d.E, d.V + (1 if vocab_plus_1 else 0)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def lin(name, out_f, in_f, bias=True, gain=1.0):
        k = 1.0 / np.sqrt(in_f)
        sd[name + ".weight"] = uniform((out_f, in_f), seed, name + ".weight", -k, k) * np.float32(gain)
        if bias:
            sd[name + ".bias"] = uniform((out_f,), seed, name + ".bias", -k, k)

    def lstm(name, in_f):
        k = 1.0 / np.sqrt(R)
        sd[name + ".weight_ih"] = uniform((4 * R, in_f), seed, name + ".weight_ih", -k, k)
        sd[name + ".weight_hh"] = uniform((4 * R, R), seed, name + ".weight_hh", -k, k)
        sd[name + ".bias_ih"] = uniform((4 * R,), seed, name + ".bias_ih", -k, k)
        sd[name + ".bias_hh"] = uniform((4 * R,), seed, name + ".bias_hh", -k, k)

    sd["roi_feat_extractor.vis_classifiers_bias"] = uniform((d.DET + 1,), seed, "vis_classifiers_bias", -0.1, 0.1)
    sd["roi_feat_extractor.vis_embed.0.weight"] = normal((d.DET + 1, d.G), seed, "vis_embed") * np.float32(0.05)
    lstm("decoder_core.att_lstm", E + 2 * R)
    lin("decoder_core.i2h_2", R, 2 * R)
    lin("decoder_core.h2h_2", R, R)
    lin("decoder_core.localied_fc", A, R)
    lin("decoder_core.soft_attn.h2attn", A, R)
    if softattn_type == "additive":
        lin("decoder_core.soft_attn.alpha_net", 1, A, gain=alpha_gain)
    lstm("decoder_core.lang_lstm", 2 * R)
    sd["embed.0.weight"] = normal((V, E), seed, "embed.0.weight") * np.float32(embed_gain)
    lin("logit", V, R, gain=logit_gain)
    lin("localizer_core.soft_attn.h2attn", A, E, gain=loc_gain)
    for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
        sd["attended_roi_decoder_core.att_lstm." + k] = sd["decoder_core.att_lstm." + k]
    lin("attended_roi_decoder_core.soft_attn.h2attn", A, R)
    if softattn_type == "additive":
        lin("attended_roi_decoder_core.soft_attn.alpha_net", 1, A)
    for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
        sd["attended_roi_decoder_core.lang_lstm." + k] = sd["decoder_core.lang_lstm." + k]
    return sd


def clip_features(d: Dims, seed: int = 1234, projected: bool = True,
                  full_mask_clip: int | None = None, feat_scale: float = 0.1) -> "OrderedDict[str, np.ndarray]":
    """What the once-per-clip encoder hands to the hot path (backbone.py:350-351), synthetic.

    Features are post-ReLU (backbone.py:84-86); masked regions are zero rows
    (modules.py:174); every 4th clip has its last floor(0.1 N) regions masked; `full_mask_clip`
    (optional) is fully masked to exercise the uniform-softmax corner (modules.py:96-98).
    pnt_mask carries the leading sentinel column the trainer adds (trainer.py:78-79).
    """
    B, N, F, R, A = d.B, d.N, d.F, d.R, d.A
    f: "OrderedDict[str, np.ndarray]" = OrderedDict()
    fs = np.float32(feat_scale)
    f["fc_feats"] = np.maximum(normal((B, R), seed, "fc_feats"), 0) * fs
    f["conv_feats"] = np.maximum(normal((B, F, R), seed, "conv_feats"), 0) * fs
    f["pool_feats"] = np.maximum(normal((B, N, R), seed, "pool_feats"), 0) * fs
    mask = np.zeros((B, N), dtype=bool)
    n_masked = max(int(0.1 * N), 1)
    for b in range(B):
        if b % 4 == 0:
            mask[b, N - n_masked:] = True
    if full_mask_clip is not None:
        mask[full_mask_clip, :] = True
    f["pool_feats"][mask] = 0
    if projected:
        k = 1.0 / np.sqrt(R)
        wp = uniform((R, A), seed, "ctx2pool_fc", -k, k)
        wc = uniform((R, A), seed, "ctx2att_fc", -k, k)
        pg = np.float32(2.45 / feat_scale)   # projected features come out ~unit variance
        f["p_pool_feats"] = ((f["pool_feats"].reshape(-1, R) @ wp) * pg).reshape(B, N, A).astype(np.float32)
        f["p_conv_feats"] = ((f["conv_feats"].reshape(-1, R) @ wc) * pg).reshape(B, F, A).astype(np.float32)
    else:
        f["p_pool_feats"] = normal((B, N, A), seed, "p_pool_feats")
        f["p_conv_feats"] = normal((B, F, A), seed, "p_conv_feats")
    f["p_pool_feats"][mask] = 0
    g = np.maximum(normal((B, N, d.G), seed, "g_pool_feats"), 0) * np.float32(0.05)
    g[mask] = 0
    f["g_pool_feats"] = g
    f["pnt_mask"] = np.concatenate([np.zeros((B, 1), dtype=bool), mask], axis=1)
    return f


def captions(d: Dims, seed: int = 1234) -> np.ndarray:
    """gt captions [B, T] int64: words U{2..V-1}, lengths U{min(5,T)..T}, 0-padded (EOS/pad = 0)."""
    words = randint((d.B, d.T), seed, "gt_words", 2, d.V)
    lens = randint((d.B,), seed, "gt_lens", min(5, d.T), d.T + 1)
    words[np.arange(d.T)[None, :] >= lens[:, None]] = 0
    return words


def label_glue_batch(d: Dims, seed: int = 1234) -> "OrderedDict[str, np.ndarray]":
    """The remaining members of the 11-tensor model call that only feed
    the supervised att2/ground losses: proposals, gt boxes, box masks, frame masks, input_seq."""
    B, N, K, T = d.B, d.N, d.K, d.T
    o: "OrderedDict[str, np.ndarray]" = OrderedDict()
    xy = uniform((B, N, 2), seed, "ppl_xy", 0, 200)
    wh = uniform((B, N, 2), seed, "ppl_wh", 10, 120)
    frm = randint((B, N, 1), seed, "ppl_frm", 0, 4).astype(np.float32)
    cls = randint((B, N, 1), seed, "ppl_cls", 1, d.DET + 1).astype(np.float32)
    sc = uniform((B, N, 1), seed, "ppl_score", 0.2, 1.0)
    o["proposals"] = np.concatenate([xy, xy + wh, frm, cls, sc], axis=2).astype(np.float32)
    # GT boxes: jittered copies of the first K proposals so that IoU > 0.5 happens
    jit = uniform((B, K, 4), seed, "gt_jit", -4, 4)
    o["gt_bboxs"] = np.concatenate([o["proposals"][:, :K, :4] + jit, o["proposals"][:, :K, 4:5],
                                    o["proposals"][:, :K, 5:6]], axis=2).astype(np.float32)
    gt = captions(d, seed)
    # every 3rd caption position names a groundable class (index >= V, captioner.py:282-283)
    det_word = randint((B, T), seed, "det_word", 1, d.DET + 1)
    groundable = (np.arange(T)[None, :] % 3 == 1) & (gt > 0)
    iseq = np.zeros((B, 1, T + 1, 4), dtype=np.int64)
    iseq[:, 0, 1:, 0] = np.where(groundable, d.V + det_word, gt)
    iseq[:, 0, 1:, 1] = groundable
    iseq[:, 0, 1:, 2] = np.where(groundable, det_word, 0)
    iseq[:, 0, 1:, 3] = gt
    o["input_seq"] = iseq
    o["gt_seq"] = np.broadcast_to(gt[:, None, :], (B, 10, T)).copy()
    # box k grounds word t (mask False) when t is groundable and k == t mod K
    bm = np.ones((B, 1, K, T + 1), dtype=bool)
    for t in range(T):
        k = t % K
        bm[:, 0, k, t + 1] = ~groundable[:, t]
    o["box_mask"] = bm
    o["frm_mask"] = o["proposals"][:, :, None, 4] != o["gt_bboxs"][:, None, :, 4]
    num = np.zeros((B, 7), dtype=np.float32)
    num[:, 0], num[:, 1], num[:, 2] = 1, N, K
    o["num"] = num
    o["sample_idx"] = np.tile(np.asarray([[0, d.F]], dtype=np.int64), (B, 1))
    return o


# ------------------------------------------------------------------ once-per-clip encoder (section 8(f) rank 1)
SEG_FEAT_DIM = 3072   # 2048 rgb + 1024 motion, hard-coded in the reference (backbone.py:68,73,328)


def detectron_tables(d: Dims, seed: int = 1234, n_vg: int = 9) -> "OrderedDict[str, np.ndarray]":
    """Synthetic stand-ins for the four Detectron pickles and the two GloVe tables the encoder's
    constructor reads (backbone.py:110-146; opts.glove_clss / glove_vg_cls from the data loader)."""
    o: "OrderedDict[str, np.ndarray]" = OrderedDict()
    k = np.float32(1.0 / np.sqrt(d.G))
    o["fc7_w"] = uniform((d.G, d.G), seed, "fc7_w", -k, k)
    o["fc7_b"] = uniform((d.G,), seed, "fc7_b", -k, k)
    o["cls_score_w"] = normal((n_vg, d.G), seed, "cls_score_w") * np.float32(0.5)
    o["cls_score_b"] = normal((n_vg,), seed, "cls_score_b") * np.float32(0.5)
    o["glove_clss"] = normal((d.DET + 1, 300), seed, "glove_clss")
    o["glove_vg_cls"] = normal((n_vg, 300), seed, "glove_vg_cls")
    return o


# projections feeding the attention scores get a gain so that attention over encoder outputs is not uniform
ENCODER_GAIN = {"ctx2pool_fc.weight": 6.0, "ctx2att_fc.weight": 8.0, "fc_embed.0.weight": 0.3, "pool_embed.0.weight": 0.5}


def encoder_fill(name: str, shape, seed: int = 1234) -> np.ndarray:
    """Deterministic value for one encoder state_dict entry (both the golden generator and the
    tests call this, so weights never need to be stored in a fixture)."""
    shape = tuple(int(s) for s in shape)
    if name.endswith("num_batches_tracked"):
        return np.asarray(3, dtype=np.int64)
    if name.endswith("running_var"):
        return uniform(shape, seed, "enc." + name, 0.5, 1.5)
    if name.endswith("running_mean"):
        return normal(shape, seed, "enc." + name) * np.float32(0.1)
    fan = shape[-1] if len(shape) > 1 else shape[0]
    k = np.float32(ENCODER_GAIN.get(name, 1.0) / np.sqrt(max(fan, 1)))
    return uniform(shape, seed, "enc." + name, -k, k)


ENCODER_CTOR_KEYS = ("vis_embed.0.weight", "vis_classifiers_bias", "ctx2pool_grd.0.weight", "ctx2pool_grd.0.bias",
                     "det_fc.0.weight")


def encoder_inputs(d: Dims, seed: int = 1234) -> "OrderedDict[str, np.ndarray]":
    """Raw clip inputs of RegionalFeatureExtractorGVD.forward (backbone.py:296-298) on top of
    `label_glue_batch`: ragged proposal counts, segment info and sampled-frame windows."""
    o = label_glue_batch(d, seed)
    B, N, F = d.B, d.N, d.F
    o["segs_feat"] = np.maximum(normal((B, F, SEG_FEAT_DIM), seed, "segs_feat"), 0) * np.float32(0.5)
    o["region_feats"] = np.maximum(normal((B, N, d.G), seed, "region_feats"), 0)
    o["num"][:, 1] = N - (np.arange(B) * 2) % (N // 2 + 1)          # ragged proposal counts, clip 0 full
    o["num"][:, 3:7] = uniform((B, 4), seed, "seg_info", 0, 1)
    lo = randint((B,), seed, "sample_lo", 0, max(F // 3, 1))
    hi = F - randint((B,), seed, "sample_hi", 0, max(F // 3, 1))
    o["sample_idx"] = np.stack([lo, hi], axis=1).astype(np.int64)
    o["pnt_mask_in"] = np.arange(N + 1)[None, :] > o["num"][:, 1:2]
    return o


# ------------------------------------------------------------------------------- in-kernel dropout, restated
def dropout_hash(seed_lo: int, seed_hi: int, step: int, site: int, idx: np.ndarray) -> np.ndarray:
    """csrc/dropout_rng.h::cvc_drop_hash on the host, word for word (uint32 arithmetic, wrapping)."""
    u = np.uint32
    with np.errstate(over="ignore"):
        x = (idx.astype(np.uint32) * u(0x9E3779B1) + u((site * 0x85EBCA77) & 0xFFFFFFFF) + u((step * 0xC2B2AE3D) & 0xFFFFFFFF)
             + u(seed_lo & 0xFFFFFFFF))
        x ^= x >> u(16); x *= u(0x7FEB352D); x ^= x >> u(15); x *= u(0x846CA68B); x ^= x >> u(16)
        x += u(seed_hi & 0xFFFFFFFF)
        x ^= x >> u(15); x *= u(0x2C1B3C6D); x ^= x >> u(12); x *= u(0x297A2D39); x ^= x >> u(15)
    return x


def dropout_keep(seed_lo: int, seed_hi: int, step: int, site: int, n: int, p: float) -> np.ndarray:
    """The multipliers (0 or 1 / (1 - p), fp32) the kernels apply to elements 0 .. n-1 of dropout site `site` at generator state
    (seed_lo, seed_hi, step): element i is DROPPED when its hash is below p * 2^32 (cvc_drop_spec / cvc_drop_mult)."""
    t = float(np.float32(p)) * 4294967296.0
    thresh = np.uint32(0xFFFFFFFF if t >= 4294967295.0 else int(t))
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    out = np.empty(n, dtype=np.float32)
    for lo in range(0, n, 1 << 22):
        hi = min(n, lo + (1 << 22))
        h = dropout_hash(seed_lo, seed_hi, step, site, np.arange(lo, hi, dtype=np.uint32))
        out[lo:hi] = np.where(h >= thresh, scale, np.float32(0.0))
    return out
