"""What the CPU oracle (oracle/ref_cpu.py) returns on the seeded full-size inputs of the `-m gpu` tests, computed once and kept
as data under tests/golden/fullsize/ (tools/make_fullsize_fixtures.py writes the files by calling the functions below with
CVC_WRITE_FULLSIZE_FIXTURES=1 -- on the CPU, no GPU involved).

Why: at BASELINE's full sizes the oracle takes 15-75 s per case on the GPU box's host cores -- four minutes of a six-minute GPU
suite were the CPU recomputing the same numbers in every run.  The inputs are a pure function of (config, seed) (cvc/synth.py's
counter-based generator); a digest of them is stored with every file and checked on load, so a changed generator, config or oracle
input is a miss, and a miss runs the oracle live exactly as before.  `tests/test_oracle_golden.py` re-derives the smallest
fixture from the live oracle on the CPU (the pipeline is honest); regenerate all of them after any change to the oracle.

Test infrastructure only: nothing under cyclical-visual-captioning_amd/ imports this."""
import hashlib
import os

import numpy as np
import torch

from cvc import synth

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize")
WRITE = os.environ.get("CVC_WRITE_FULLSIZE_FIXTURES") == "1"
GRAD_SAMPLES = 4096


def inputs_digest(*dicts) -> str:
    """sha256 over the shapes, dtypes and ALL bytes of every array (round-4 review: the first / last 2 KB were not the inputs;
    sha256 runs at 1.2 GB/s here -- 1 s at config 2 / 3, 3 s at config 5)"""
    h = hashlib.sha256()
    for dct in dicts:
        for k in sorted(dct):
            a = np.ascontiguousarray(dct[k])
            h.update(f"{k}:{a.shape}:{a.dtype}".encode())
            h.update(a.view(np.uint8).reshape(-1).data)
    return h.hexdigest()[:32]


def oracle_digest() -> str:
    """sha256 of oracle/ref_cpu.py: a fixture written by another oracle is a miss (round-4 advisor finding)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "oracle", "ref_cpu.py"), "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:32]


def _cached(key, digest, compute):
    """-> dict of numpy arrays: from tests/golden/fullsize/<key>.npz when it was made from the same inputs BY THE SAME ORACLE,
    else compute() now"""
    path = os.path.join(HERE, key + ".npz")
    if os.path.exists(path) and not WRITE:
        z = np.load(path)
        if str(z["inputs_digest"]) == digest and "oracle_digest" in z.files and str(z["oracle_digest"]) == oracle_digest():
            return {k: z[k] for k in z.files if k not in ("inputs_digest", "oracle_digest")}, "fixture"
    out = compute()
    if WRITE:
        os.makedirs(HERE, exist_ok=True)
        np.savez_compressed(path, inputs_digest=np.array(digest), oracle_digest=np.array(oracle_digest()), **out)
    return out, "live"


def deciding_gaps(ref_logp, unk_idx=synth.UNK_IDX):
    """[B, T] margin by which the oracle's word wins at every step (best minus second-best word that is not UNK,
    captioner.py:415-422) -- all that the tie-aware sequence comparison needs of the [B, T, V] log-probs"""
    lp = np.array(ref_logp, dtype=np.float64, copy=True)
    lp[..., unk_idx] = -np.inf
    top = -np.partition(-lp, 1, axis=-1)[..., :2]
    return top[..., 0] - top[..., 1]


def _top2_non_unk(logp, unk_idx):
    """indices [.., 2] of the best and second-best word that is not UNK"""
    lp = np.array(logp, dtype=np.float64, copy=True)
    lp[..., unk_idx] = -np.inf
    return np.argsort(-lp, axis=-1, kind="stable")[..., :2]


def greedy(name, seed, d, sd, f_np, referee=True):
    """oracle.greedy_sample in fp32 -> dict(seq [B, T] int64, att [B, T, N] f32, gaps [B, T] f64) and, the REFEREE of near-ties
    (round-4 review item 5), the same oracle on .double() weights and features: seq64 [B, T], gaps64 [B, T] (its deciding
    margins), top2_64 / lp64_top2 [B, T, 2] (its two deciding words and their log-probs: what a GPU run's log-prob of its own
    selected word is measured against) and dev32 [B, T] -- how far the fp32 oracle's log-probs of the referee's two deciding words are from the referee's
    own, measured at the steps where both oracles have followed the same words so far (identical inputs; -1 elsewhere).  A GPU
    sequence is then held to the REFEREE's words wherever the referee's margin exceeds twice the largest such deviation: the
    tolerance of the tie rule is measured, not chosen.  -> (dict, source)"""
    def compute():
        from oracle import ref_cpu as O
        with torch.no_grad():
            seq_o, att_o, _, logp_o = O.greedy_sample(O.to_torch(sd), O.to_torch(f_np), d.T, synth.UNK_IDX, return_logprobs=True)
        out = dict(seq=seq_o.numpy(), att=att_o.numpy(), gaps=deciding_gaps(logp_o.numpy()))
        if referee:
            dbl = lambda dct: {k: (v.double() if v.dtype.is_floating_point else v) for k, v in O.to_torch(dct).items()}
            prev = torch.get_default_dtype()
            torch.set_default_dtype(torch.float64)            # (init_hidden's zeros)
            try:
                with torch.no_grad():
                    seq_r, _att_r, _, logp_r = O.greedy_sample(dbl(sd), dbl(f_np), d.T, synth.UNK_IDX, return_logprobs=True)
            finally:
                torch.set_default_dtype(prev)
            lp32, lp64 = logp_o.numpy().astype(np.float64), logp_r.numpy()
            top = _top2_non_unk(lp64, synth.UNK_IDX)
            dev = np.abs(np.take_along_axis(lp32, top, -1) - np.take_along_axis(lp64, top, -1)).max(-1)        # [B, T]
            same = (seq_o.numpy() == seq_r.numpy())
            # step t's log-probs see the words of steps < t: comparable while every earlier word agreed
            prefix = np.concatenate([np.ones((same.shape[0], 1), bool), np.cumprod(same, 1).astype(bool)[:, :-1]], 1)
            out.update(seq64=seq_r.numpy(), gaps64=deciding_gaps(lp64), dev32=np.where(prefix, dev, -1.0),
                       top2_64=top.astype(np.int64), lp64_top2=np.take_along_axis(lp64, top, -1))
        return out
    return _cached(f"{name}_seed{seed}_greedy", inputs_digest(sd, f_np), compute)


def beam(name, seed, d, sd, f_np, beam_size):
    """oracle.beam_search -> dict(seq [B, T], att [B, T, N], scores [B, beam]), source"""
    def compute():
        from oracle import ref_cpu as O
        with torch.no_grad():
            seq_o, att_o, sc_o = O.beam_search(O.to_torch(sd), O.to_torch(f_np), d.T, synth.UNK_IDX, beam_size)
        return dict(seq=seq_o.numpy(), att=att_o.numpy(), scores=sc_o.numpy())
    return _cached(f"{name}_seed{seed}_beam{beam_size}", inputs_digest(sd, f_np), compute)


def sample_index(name: str, numel: int) -> np.ndarray:
    """which elements of parameter `name`'s gradient the cyclical fixture keeps (all of a small tensor)"""
    if numel <= GRAD_SAMPLES:
        return np.arange(numel)
    g = np.random.Generator(np.random.PCG64(int.from_bytes(hashlib.sha256(name.encode()).digest()[:8], "little")))
    return np.sort(g.choice(numel, GRAD_SAMPLES, replace=False))


def cyclical_eval(name, seed, d, sd, f_np, b_np):
    """oracle.cyclical_forward + autograd of 0.5 lm + 0.5 lm_recon in eval mode -> dict(losses [5], ground_weights [B, T, N],
    per parameter: grad_norm.<n> (the WHOLE gradient's 2-norm, f64), grad_at.<n> (its elements at sample_index(n)), or
    grad_none.<n>), source.  The sampled elements are compared against the same elements on the GPU, scaled to the whole norm."""
    def compute():
        from oracle import ref_cpu as O
        P = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in O.to_torch(sd).items()}
        for k in list(P):                                           # the reconstructor shares the decoder's LSTM cells
            if k.startswith("attended_roi_decoder_core.") and "lstm" in k:
                P[k] = P[k.replace("attended_roi_decoder_core.", "decoder_core.")]
        col = {}
        ref = O.cyclical_forward(P, O.to_torch(f_np), O.to_torch(b_np), T=d.T, vocab_size=d.V, collect=col)
        O.training_loss(ref, xe_loss_weight=0.5, w_att2=0.0, w_cls=0.0, caption_consistency_loss_weight=0.5).backward()
        out = dict(losses=np.array([float(x.detach()) for x in ref], dtype=np.float64),
                   ground_weights=col["ground_weights"].detach().numpy())
        for n, p in P.items():
            if not p.dtype.is_floating_point:
                continue
            if p.grad is None:
                out["grad_none." + n] = np.array(1)
                continue
            g = p.grad.double().reshape(-1)
            out["grad_norm." + n] = np.array(float(g.norm()))
            out["grad_at." + n] = g[torch.from_numpy(sample_index(n, g.numel()))].numpy()
        return out
    return _cached(f"{name}_seed{seed}_cyclical_eval", inputs_digest(sd, f_np, b_np), compute)
