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


def tie_aware_seq_equal(seq, ref_seq, ref_logp, tol=1e-4):
    """Greedy sequences must match the oracle's, except where the oracle's own top-2 gap is inside
    fp32 noise (then the prefix up to the tie must match and the rest is not comparable)."""
    seq, ref_seq = np.asarray(seq), np.asarray(ref_seq)
    B, T = ref_seq.shape
    n_exact = 0
    for b in range(B):
        for t in range(T):
            if seq[b, t] == ref_seq[b, t]:
                n_exact += 1
                continue
            top = np.sort(np.asarray(ref_logp[b, t]))[::-1]
            assert top[0] - top[1] < tol or (top[1] - top[2] < tol), \
                f"clip {b} step {t}: got {seq[b, t]} want {ref_seq[b, t]} with a clear margin {top[0] - top[1]}"
            break
    return n_exact
