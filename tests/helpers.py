"""Shared builders for the parity tests (GPU side builds the product model; CPU side the oracle)."""
import argparse

import numpy as np
import torch

from cvc import synth


def make_opts(d, **over):
    o = argparse.Namespace(
        vocab_size=d.V, itow={str(i): "w%d" % i for i in range(d.V)}, wtoi={"UNK": synth.UNK_IDX},
        seq_length=d.T, seq_per_img=1, rnn_size=d.R, input_encoding_size=d.E, att_hid_size=d.A,
        drop_prob_lm=0.5, softattn_type="additive", softmax_temp=1.0, localizer_softmax_temp=1.0,
        global_img_in_attn_lstm=1, embedding_vocab_plus_1=False, train_decoder_only=False, beam_size=1,
        detect_size=d.DET, vis_encoding_size=d.G)
    for k, v in over.items():
        setattr(o, k, v)
    return o


def to_dev(d, dev):
    return {k: (torch.from_numpy(np.ascontiguousarray(v)).to(dev) if isinstance(v, np.ndarray) else v) for k, v in d.items()}


def build_model(d, sd_np, dev, **over):
    from cvc.model.captioner import DecodeAndGroundCaptionerGVDROI, PrecomputedRegionFeatures
    opts = make_opts(d, **over)
    model = DecodeAndGroundCaptionerGVDROI(opts, roi_extractor=PrecomputedRegionFeatures(d.DET, d.G))
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return model.to(dev).eval()


def model_call(model, feats, batch, lang_eval=False):
    B = feats["fc_feats"].shape[0]
    dummy = torch.zeros(B, 1, 1, device=feats["fc_feats"].device)
    return model(feats, batch["input_seq"], batch["gt_seq"], batch["num"], batch["proposals"], batch["gt_bboxs"],
                 batch["box_mask"], dummy, batch["frm_mask"], batch["sample_idx"], feats["pnt_mask"], lang_eval)


def deciding_gaps(ref_logp, unk_idx=synth.UNK_IDX):
    """[B, T] margin by which the oracle's word wins at every step.  The sampler takes the best word that is not UNK
    (captioner.py:415-422: #2 iff #1 == unk_idx), so the margin is best non-UNK minus second-best non-UNK log-prob."""
    lp = np.array(ref_logp, dtype=np.float64, copy=True)
    lp[..., unk_idx] = -np.inf
    top = -np.partition(-lp, 1, axis=-1)[..., :2]
    return top[..., 0] - top[..., 1]


def tie_aware_seq_equal(seq, ref_seq, ref_logp, tol=1e-4, unk_idx=synth.UNK_IDX, clear_gap=1e-3, stats=None, gaps=None):
    """Greedy sequences must match the oracle's, except where the margin that decided the oracle's own word (best against
    second-best word that is not UNK) is inside fp32 noise: then the prefix up to the tie must match and the rest of that clip is
    not comparable.  Clips whose smallest deciding margin exceeds `clear_gap` must match exactly, whole sequence."""
    seq, ref_seq = np.asarray(seq), np.asarray(ref_seq)
    B, T = ref_seq.shape
    if gaps is None:          # (gaps given: the margins of a stored oracle run, tests/fullsize_oracle.py)
        gaps = deciding_gaps(ref_logp, unk_idx)
    clear = gaps.min(axis=1) > clear_gap
    assert np.array_equal(seq[clear], ref_seq[clear]), \
        f"clips with every deciding margin > {clear_gap} differ: {np.nonzero((seq != ref_seq).any(1) & clear)[0].tolist()}"
    n_exact, flips, worst = 0, 0, 0.0
    for b in range(B):
        for t in range(T):
            if seq[b, t] == ref_seq[b, t]:
                n_exact += 1
                continue
            assert gaps[b, t] < tol, \
                f"clip {b} step {t}: got {seq[b, t]} want {ref_seq[b, t]} with a clear margin {gaps[b, t]} (tolerance {tol})"
            flips, worst = flips + 1, max(worst, float(gaps[b, t]))
            break
    if stats is not None:      # how close the comparison came to its tolerance (goes to the test log)
        stats.update(clips=B, clear_clips=int(clear.sum()), flipped_clips=flips, largest_margin_at_a_flip=worst, tol=tol)
    return n_exact
