"""Pins oracle/ref_cpu.py to vectors produced by the reference itself (tests/golden/*.npz,
tools/make_golden.py).  CPU only.  Tolerance: both sides are torch fp32 CPU running the same
ATen ops, so outputs agree to ~1e-6; the bound below is 1e-5 abs/rel.
"""
import dataclasses
import os

import numpy as np
import pytest
import torch

from cvc import synth
from oracle import ref_cpu as O

TOL = dict(rtol=1e-5, atol=1e-5)


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def close(a, b, **kw):
    kw = {**TOL, **kw}
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, **kw)


@pytest.fixture(scope="module")
def tiny(g1):
    d = synth.CONFIGS["tiny"]
    P = O.to_torch(g1.sub("sd."))
    feats = O.to_torch(g1.sub("feats."))
    batch = O.to_torch(g1.sub("batch."))
    unit = O.to_torch(g1.sub("unit."))
    return d, P, feats, batch, unit


def test_synth_regenerates_fixture_inputs(g1):
    """The fixture's stored weights/inputs are exactly what cvc.synth regenerates from the seed
    (so G2/G3 can store outputs only)."""
    d = synth.CONFIGS["tiny"]
    seed = int(g1["meta.seed"])
    for k, v in synth.hot_path_state_dict(d, seed).items():
        np.testing.assert_array_equal(v, g1["sd." + k])
    for k, v in synth.clip_features(d, seed, full_mask_clip=2).items():
        np.testing.assert_array_equal(v, g1["feats." + k])
    for k, v in synth.label_glue_batch(d, seed).items():
        np.testing.assert_array_equal(v, g1["batch." + k])


def test_a1_additive_attention(tiny, g1):
    d, P, f, _, u = tiny
    w = [P["decoder_core.soft_attn." + k] for k in ("h2attn.weight", "h2attn.bias", "alpha_net.weight", "alpha_net.bias")]
    mask = f["pnt_mask"][:, 1:]
    ctx, a, fm = O.additive_attention(u["h"], f["p_pool_feats"], f["pool_feats"], mask, u["fmask"], *w)
    close(ctx, g1["a1.regions.ctx"]); close(a, g1["a1.regions.attn"]); close(fm, g1["a1.regions.fm"])
    # the fully masked clip (clip 2) is uniform, not NaN (modules.py:96-98)
    close(a[2], np.full(d.N, 1.0 / d.N, np.float32))
    # masked positions carry exactly zero weight when the row is not fully masked
    assert float(a[0][mask[0]].abs().max()) == 0.0
    ctx, a, fm = O.additive_attention(u["h"], f["p_conv_feats"], f["conv_feats"], None, None, *w)
    close(ctx, g1["a1.frames.ctx"]); close(a, g1["a1.frames.attn"]); assert fm is None
    ctx, a, _ = O.additive_attention(u["h"], f["p_pool_feats"], None, mask, None, *w)
    close(ctx, g1["a1.noctx.ctx"]); close(a, g1["a1.noctx.attn"])


@pytest.mark.parametrize("temp", [1.0, 2.5])
def test_a2_dot_attention(tiny, g1, temp):
    d, P, f, _, u = tiny
    w = [P["localizer_core.soft_attn." + k] for k in ("h2attn.weight", "h2attn.bias")]
    ctx, a, fm = O.dot_attention(u["emb"], f["p_pool_feats"], f["pool_feats"], f["pnt_mask"][:, 1:], u["fmask"], *w, temp)
    pre = "a2.temp%g." % temp
    close(ctx, g1[pre + "ctx"]); close(a, g1[pre + "attn"]); close(fm, g1[pre + "fm"])


def test_with_sentinel_minus_inf_fill_golden(tiny):
    """with_sentinel=True (modules.py:40-41, 123-124): masked positions filled with -inf -- the oracle against the reference's own
    outputs (tests/golden/g7_sentinel.npz, tools/make_golden.py g7).  The fully masked clip is NaN there, as in the reference."""
    from conftest import Golden
    g7 = Golden("g7_sentinel.npz")
    d, P, f, _, u = tiny
    mask = f["pnt_mask"][:, 1:]
    w = [P["decoder_core.soft_attn." + k] for k in ("h2attn.weight", "h2attn.bias", "alpha_net.weight", "alpha_net.bias")]
    ctx, a, fm = O.additive_attention(u["h"], f["p_pool_feats"], f["pool_feats"], mask, u["fmask"], *w, with_sentinel=True)
    close(ctx, g7["add.ctx"]); close(a, g7["add.attn"]); close(fm, g7["add.fm"])
    assert np.isnan(g7["add.attn"][2]).all() and np.isinf(g7["add.fm"]).any()
    w = [P["localizer_core.soft_attn." + k] for k in ("h2attn.weight", "h2attn.bias")]
    ctx, a, fm = O.dot_attention(u["emb"], f["p_pool_feats"], f["pool_feats"], mask, u["fmask"], *w, 2.5, with_sentinel=True)
    close(ctx, g7["dot.ctx"]); close(a, g7["dot.attn"]); close(fm, g7["dot.fm"])


def test_a4_lstm_cell_forms_agree(tiny):
    d, P, f, _, u = tiny
    x = torch.cat([u["state_h"][1], f["fc_feats"], u["emb"]], 1)
    cell = O._cell(P, "decoder_core.att_lstm")
    h1, c1 = O.lstm_cell(x, u["state_h"][0], u["state_c"][0], *cell)
    h2, c2 = O.lstm_cell_explicit(x, u["state_h"][0], u["state_c"][0], *cell)
    close(h1, h2.numpy()); close(c1, c2.numpy())


def test_a3_decoder_step(tiny, g1):
    d, P, f, _, u = tiny
    out, (h, c), ra, fm, ctx_r = O.decoder_step(P, u["emb"], f["fc_feats"], f["conv_feats"], f["p_conv_feats"],
                                                f["pool_feats"], f["p_pool_feats"], f["pnt_mask"][:, 1:],
                                                (u["state_h"], u["state_c"]), u["fmask"])
    for k, v in dict(out=out, h=h, c=c, roi_attn=ra, fm=fm, ctx_r=ctx_r).items():
        close(v, g1["a3." + k])


def test_a5_reconstructor_step(tiny, g1):
    d, P, f, _, u = tiny
    out, (h, c) = O.reconstructor_step(P, u["emb"], f["fc_feats"], u["loc_pool"], u["loc_conv"],
                                       (u["state_h"], u["state_c"]))
    close(out, g1["a5.out"]); close(h, g1["a5.h"]); close(c, g1["a5.c"])


def test_a6_localizer_step(tiny, g1):
    d, P, f, _, u = tiny
    lp, lc, prob = O.localizer_step(P, u["emb"], f["conv_feats"], f["p_conv_feats"], f["pool_feats"],
                                    f["p_pool_feats"], f["pnt_mask"][:, 1:], u["fmask"])
    close(lp, g1["a6.loc_pool"]); close(lc, g1["a6.loc_conv"]); close(prob, g1["a6.prob"])


def test_a8_greedy_sample_tiny(tiny, g1):
    d, P, f, _, _ = tiny
    seq, att, lps, logp = O.greedy_sample(P, f, d.T, synth.UNK_IDX, return_logprobs=True)
    np.testing.assert_array_equal(seq.numpy(), g1["a8.seq"])
    close(att, g1["a8.att2_weights"]); close(logp, g1["a8.logp"])
    assert seq.shape == (d.B, d.T) and not (seq == synth.UNK_IDX).any()


def _grad_check(P, feats, got_total, golden, loss):
    loss.backward()
    close(loss, golden["total"])
    n = 0
    for k, v in golden.items():
        if not k.startswith("grad."):
            continue
        name = k[len("grad."):]
        g = feats[name[3:]].grad if name.startswith("in.") else P[name].grad
        if v is None:
            assert g is None or float(g.abs().max()) == 0.0, name
        else:
            assert g is not None, name
            close(g, v, rtol=2e-5, atol=2e-6)
            n += 1
    assert n > 10


def _shared(P):
    """Tie the reconstructor's LSTM cells to the decoder's, as the reference module does
    (captioner.py:86-87) so gradients accumulate into one tensor."""
    P = dict(P)
    for k in list(P):
        if k.startswith("attended_roi_decoder_core.") and "lstm" in k:
            P[k] = P[k.replace("attended_roi_decoder_core.", "decoder_core.")]
    return P


@pytest.mark.parametrize("variant,kw,mix", [
    ("a9.cyc.", dict(), dict(xe_loss_weight=0.5, w_att2=0.0, w_cls=0.0, caption_consistency_loss_weight=0.5)),
    ("a9.sup.", dict(), dict(xe_loss_weight=0.5, w_att2=0.05, w_cls=0.0, caption_consistency_loss_weight=0.5)),
    ("a9.dec.", dict(train_decoder_only=True), dict(xe_loss_weight=0.5, w_att2=0.0, w_cls=0.0, caption_consistency_loss_weight=0.0)),
])
def test_a9_cyclical_losses_and_grads(g1, variant, kw, mix):
    d = synth.CONFIGS["tiny"]
    P = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in O.to_torch(g1.sub("sd.")).items()}
    P = _shared(P)
    feats = O.to_torch(g1.sub("feats."))
    for k in ("fc_feats", "conv_feats", "p_conv_feats", "pool_feats", "p_pool_feats", "g_pool_feats"):
        feats[k].requires_grad_(True)
    batch = O.to_torch(g1.sub("batch."))
    col = {}
    losses = O.cyclical_forward(P, feats, batch, T=d.T, vocab_size=d.V, collect=col, **kw)
    gold = g1.sub(variant)
    assert len(losses) == (4 if kw else 5)
    for i, l in enumerate(losses):
        assert l.shape == (1,)
        close(l, gold["loss%d" % i].reshape(1))
    close(col["ground_weights"], gold["ground_weights"])      # a10
    _grad_check(P, feats, None, gold, O.training_loss(losses, **mix))
    # dead parameters never receive a gradient
    for dead in ("decoder_core.i2h_2.weight", "decoder_core.h2h_2.weight", "decoder_core.localied_fc.weight",
                 "attended_roi_decoder_core.soft_attn.h2attn.weight"):
        assert P[dead].grad is None


def test_g2_cfg1_greedy_and_losses(g2):
    d = synth.CONFIGS["cfg1"]
    seed = int(g2["meta.seed"])
    P = O.to_torch(synth.hot_path_state_dict(d, seed))
    feats = O.to_torch(synth.clip_features(d, seed))
    with torch.no_grad():
        seq, att, lps, logp = O.greedy_sample(P, feats, d.T, synth.UNK_IDX, return_logprobs=True)
    np.testing.assert_array_equal(seq.numpy(), g2["a8.seq"])
    close(att, g2["a8.att2_weights"])
    close(torch.gather(logp, 2, T(g2["a8.top8_idx"])), g2["a8.top8_logp"], rtol=1e-4, atol=1e-4)
    # greedy captions are non-degenerate with the x4 logit gain
    assert len(np.unique(seq.numpy())) > 8
    Pg = _shared({k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in P.items()})
    batch = O.to_torch(synth.label_glue_batch(d, seed))
    losses = O.cyclical_forward(Pg, feats, batch, T=d.T, vocab_size=d.V)
    gold = g2.sub("a9.cyc.")
    for i, l in enumerate(losses):
        close(l, gold["loss%d" % i].reshape(1), rtol=1e-5, atol=1e-5)
    O.training_loss(losses, xe_loss_weight=0.5, w_att2=0.0, w_cls=0.0, caption_consistency_loss_weight=0.5).backward()
    for k, v in gold.items():
        if k.endswith(".norm") and not k.startswith("grad.in."):
            name = k[len("grad."):-len(".norm")]
            g = Pg[name].grad
            np.testing.assert_allclose(float(g.double().norm()), float(v), rtol=1e-4, atol=1e-7)  # alpha_net.bias grad is ~0 (softmax shift invariance)
            np.testing.assert_allclose(g.reshape(-1)[T(gold["grad." + name + ".idx"])].numpy(),
                                       gold["grad." + name + ".val"], rtol=1e-3, atol=1e-7)


def test_g3_shard_mean_equals_allreduce_target(g3):
    """DataParallel semantics: mean over shards of per-shard grads."""
    d = dataclasses.replace(synth.CONFIGS["tiny"], B=4)
    seed = int(g3["meta.seed"])
    sd = synth.hot_path_state_dict(d, seed)
    feats_np, batch_np = synth.clip_features(d, seed), synth.label_glue_batch(d, seed)
    acc = {}
    for s, sl in enumerate((slice(0, 2), slice(2, 4))):
        P = _shared({k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in O.to_torch(sd).items()})
        feats = O.to_torch({k: v[sl] for k, v in feats_np.items()})
        batch = O.to_torch({k: v[sl] for k, v in batch_np.items()})
        losses = O.cyclical_forward(P, feats, batch, T=d.T, vocab_size=d.V)
        for i, l in enumerate(losses):
            close(l, g3["shard%d.loss%d" % (s, i)].reshape(1))
        O.training_loss(losses, xe_loss_weight=0.5, w_att2=0.0, w_cls=0.0, caption_consistency_loss_weight=0.5).backward()
        for k, p in P.items():
            if p.grad is not None and not k.startswith("attended_roi_decoder_core.att_lstm") \
                    and not k.startswith("attended_roi_decoder_core.lang_lstm"):
                acc[k] = acc.get(k, 0) + p.grad / 2
    mean = g3.sub("mean.grad.")
    n = 0
    for k, v in mean.items():
        if v is None:
            assert k not in acc
        else:
            close(acc[k], v, rtol=2e-5, atol=2e-6)
            n += 1
    assert n > 10


def test_beam1_equals_greedy_and_beam_bounds(tiny):
    d, P, f, _, _ = tiny
    with torch.no_grad():
        seq_g, att_g, lps, _ = O.greedy_sample(P, f, d.T, synth.UNK_IDX, return_logprobs=True)
        seq1, att1, sc1 = O.beam_search(P, f, d.T, synth.UNK_IDX, 1)
        # greedy never freezes after EOS, beam does: compare up to and including the first 0
        for b in range(d.B):
            s = seq_g[b].tolist()
            L = (s.index(0) + 1) if 0 in s else d.T
            assert seq1[b, :L].tolist() == s[:L]
            assert (seq1[b, L:] == 0).all()
            close(att1[b, :L], att_g[b, :L].numpy())
        seq3, _, sc3 = O.beam_search(P, f, d.T, synth.UNK_IDX, 3)
        assert (sc3[:, :-1] >= sc3[:, 1:]).all()               # non-increasing in rank
        assert (sc3[:, 0] >= sc1[:, 0] - 1e-5).all()            # at least as good as beam 1


def test_beam_matches_exhaustive_on_micro_case():
    d = dataclasses.replace(synth.CONFIGS["tiny"], V=5, T=3, B=2)
    P = O.to_torch(synth.hot_path_state_dict(d, 7))
    f = O.to_torch(synth.clip_features(d, 7))
    with torch.no_grad():
        best = O.exhaustive_best_sequence(P, f, d.T, synth.UNK_IDX)
        _, _, sc = O.beam_search(P, f, d.T, synth.UNK_IDX, d.V ** (d.T - 1))
    close(sc[:, 0], best.numpy(), rtol=1e-5, atol=1e-5)


def test_label_glue_matches_reference_side_products(g1):
    """overlaps/roi_labels/frame masks are exercised through a9 (att2/ground losses); here: shapes
    and the documented conventions."""
    d = synth.CONFIGS["tiny"]
    f = O.to_torch(g1.sub("feats."))
    b = O.to_torch(g1.sub("batch."))
    ov = O.bbox_overlaps(b["proposals"], b["gt_bboxs"], b["frm_mask"] | f["pnt_mask"][:, 1:].unsqueeze(-1))
    assert ov.shape == (d.B, d.N, d.K) and float(ov.max()) <= 1.0
    assert float(ov[b["frm_mask"]].abs().max()) == 0.0
    lab = O.bbox_target(b["box_mask"][:, :, :, 1], ov)
    assert lab.dtype == torch.bool and lab.shape == (d.B, d.N)


def test_stored_fullsize_oracle_results_are_what_the_oracle_returns():
    """tests/golden/fullsize/cfg2_seed1236_greedy.npz (read by the -m gpu full-size tests instead of re-running the oracle on the
    host, tests/fullsize_oracle.py) against the live oracle on the same seeded inputs: same words (up to fp32 near-ties, should the
    host's BLAS differ from the one the file was written on), same attention maps, same deciding margins."""
    import torch
    from helpers import tie_aware_seq_equal
    import fullsize_oracle as FO
    from oracle import ref_cpu as O
    d = synth.CONFIGS["cfg2"]
    sd, f_np = synth.hot_path_state_dict(d, 1236), synth.clip_features(d, 1236)
    ref, src = FO.greedy("cfg2", 1236, d, sd, f_np)
    assert src == "fixture", "tests/golden/fullsize/cfg2_seed1236_greedy.npz is missing or was made from other inputs: run tools/make_fullsize_fixtures.py"
    with torch.no_grad():
        seq, att, _, logp = O.greedy_sample(O.to_torch(sd), O.to_torch(f_np), d.T, synth.UNK_IDX, return_logprobs=True)
    n = tie_aware_seq_equal(seq.numpy(), ref["seq"], None, gaps=ref["gaps"])
    assert n >= 0.99 * d.B * d.T
    same = (seq.numpy() == ref["seq"]).all(1)
    np.testing.assert_allclose(att.numpy()[same], ref["att"][same], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(FO.deciding_gaps(logp.numpy())[same], ref["gaps"][same], rtol=0, atol=2e-5)


def test_fullsize_oracle_fixtures_equal_the_reference_at_benchmark_size():
    """BASELINE's full sizes pinned to the REFERENCE (round-4 review item 3).  tests/golden/g9_fullsize_ref.npz holds what the
    reference itself returns (tools/make_golden.py g9: its _sample at config 2 / 5, its _forward_3_loops + autograd at config 3);
    tests/golden/fullsize/*.npz hold what oracle/ref_cpu.py returns on the same inputs and carry the digest of the oracle that
    wrote them (checked here; the config-2 file is additionally re-derived from the live oracle by the test above).  Same ATen op
    sequence on the same host => bit-identical: words, attention maps, margins, the five losses, ground_weights, every gradient
    norm and every sampled gradient element."""
    import fullsize_oracle as FO
    from conftest import load_g9
    for cfg, seed in (("cfg2", 1236), ("cfg5", 1505)):
        z = np.load(os.path.join(FO.HERE, f"{cfg}_seed{seed}_greedy.npz"))
        g9 = load_g9(cfg + ".greedy.")
        assert str(z["oracle_digest"]) == FO.oracle_digest(), "tests/golden/fullsize was written by another oracle: run tools/make_fullsize_fixtures.py"
        assert str(z["inputs_digest"]) == str(g9["inputs_digest"])
        assert np.array_equal(z["seq"], g9["seq"])
        np.testing.assert_array_equal(z["att"], g9["att2_weights"])
        np.testing.assert_array_equal(z["gaps"], g9["gaps"])
        # the fp64 referee agrees with the reference's words on every clip at both sizes (margins >= 1e-5, fp32 CPU deviation ~1e-6)
        assert np.array_equal(z["seq64"], g9["seq"])
        assert 0 < float(z["dev32"].max()) < 1e-5 and float(z["gaps64"].min()) > 2 * float(z["dev32"].max())
    z = np.load(os.path.join(FO.HERE, "cfg3_seed1303_cyclical_eval.npz"))
    g9 = load_g9("cfg3.cyclical.")
    assert str(z["oracle_digest"]) == FO.oracle_digest() and str(z["inputs_digest"]) == str(g9["inputs_digest"])
    np.testing.assert_array_equal(z["losses"], g9["losses"])
    np.testing.assert_array_equal(z["ground_weights"], g9["ground_weights"])
    names = [k[len("grad_norm."):] for k in g9 if k.startswith("grad_norm.")]
    assert len(names) >= 15
    for n in names:
        np.testing.assert_allclose(z["grad_norm." + n], g9["grad_norm." + n], rtol=1e-12)
        np.testing.assert_array_equal(z["grad_at." + n], g9["grad_at." + n])
    assert sorted(k for k in g9 if k.startswith("grad_none.")) == sorted(k for k in z.files if k.startswith("grad_none.") and
                                                                           not k.startswith("grad_none.roi_feat_extractor"))


def test_inputs_digest_covers_every_byte():
    """the digest of the full-size fixtures hashes whole arrays (round 4 hashed the first and last 2 KB)"""
    import fullsize_oracle as FO
    a = {"x": np.zeros(1 << 16, dtype=np.float32)}
    b = {"x": a["x"].copy()}
    b["x"][1 << 15] = 1.0
    assert FO.inputs_digest(a) != FO.inputs_digest(b)
    assert FO.inputs_digest(a) == FO.inputs_digest({"x": a["x"].copy()})


def test_g10_embedding_vocab_plus_1_oracle_vs_reference():
    """opts.embedding_vocab_plus_1 = True (opts.py:197, captioner.py:53-60, 72-76): V + 1 embedding rows and head rows.  The
    oracle on the reference's golden (tools/make_golden.py g10): greedy words + per-step log-probs, the five losses, every gradient."""
    import torch
    from conftest import Golden
    from helpers import tie_aware_seq_equal
    from oracle import ref_cpu as O
    g = Golden("g10_vocab_plus_1.npz")
    d = synth.CONFIGS["tiny"]
    seed = int(g["meta.seed"])
    sd = synth.hot_path_state_dict(d, seed, vocab_plus_1=True)
    assert sd["logit.weight"].shape[0] == d.V + 1 and sd["embed.0.weight"].shape[0] == d.V + 1
    f, b = synth.clip_features(d, seed), synth.label_glue_batch(d, seed)
    with torch.no_grad():
        seq, att, _, logp = O.greedy_sample(O.to_torch(sd), O.to_torch(f), d.T, synth.UNK_IDX, return_logprobs=True)
    assert logp.shape[-1] == d.V + 1
    assert tie_aware_seq_equal(seq.numpy(), g["a8.seq"], g["a8.logp"]) == d.B * d.T
    np.testing.assert_allclose(att.numpy(), g["a8.att2_weights"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(logp.numpy(), g["a8.logp"], rtol=1e-5, atol=1e-5)
    P = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in O.to_torch(sd).items()}
    for k in list(P):
        if k.startswith("attended_roi_decoder_core.") and "lstm" in k:
            P[k] = P[k.replace("attended_roi_decoder_core.", "decoder_core.")]
    ref = O.cyclical_forward(P, O.to_torch(f), O.to_torch(b), T=d.T, vocab_size=d.V)
    for i, x in enumerate(ref):
        np.testing.assert_allclose(float(x.detach()), float(g["a9.cyc.loss%d" % i].reshape(-1)[0]), rtol=1e-5, atol=1e-6)
    O.training_loss(ref, xe_loss_weight=0.5, w_att2=0.0, w_cls=0.0, caption_consistency_loss_weight=0.5).backward()
    gold = g.sub("a9.cyc.grad.")
    n = 0
    for k, want in gold.items():
        if k.startswith("in.") or k.startswith("roi_feat_extractor") or k not in P:
            continue
        if want is None:
            assert P[k].grad is None or float(P[k].grad.abs().max()) == 0.0, k
            continue
        np.testing.assert_allclose(P[k].grad.numpy(), want, rtol=1e-4, atol=1e-6, err_msg=k)
        n += 1
    assert n >= 15
