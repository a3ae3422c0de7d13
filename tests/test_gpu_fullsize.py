"""Full-size parity (pytest -m gpu): BASELINE configs 3 and 5 at their real dimensions -- the HIP path on the GPU against
the CPU oracle's results on the same seeded inputs, plus the grounder (a10) element-wise.  The oracle's side is read from
tests/golden/fullsize/ (tests/fullsize_oracle.py: written by tools/make_fullsize_fixtures.py from the oracle itself, keyed to a
digest of the inputs) and is recomputed on the host, as in earlier rounds, whenever a file is absent or its digest does not match.

  cfg3-i   B=64, N=100, F=480, D=2048, T=20, beam=5 decode           vs oracle.beam_search
  cfg3-ii  same dims, cyclical forward + backward (eval-mode dropout, loss mix 0.5 / 0.5) vs oracle.cyclical_forward autograd
  cfg5     B=64, N=300, F=480, D=4096, T=30, greedy and beam=5         vs oracle.greedy_sample / beam_search

Reference behaviour: model/captioner.py:196-382 (training pass), :384-443 (sampler); beam search is build-defined
(SURVEY.md section 7), pinned by the oracle.  Tolerances as in test_gpu_parity.py: 1e-4 after T recurrent steps,
2e-4 (sums of T log-probs) on beam scores, gradient error <= 5e-4 of the gradient's norm.
"""
import numpy as np
import pytest
import torch

from cvc import synth

pytestmark = pytest.mark.gpu

SEQ_TOL = dict(rtol=1e-4, atol=1e-4)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("the gpu-marked tests need a visible MI355X (torch.cuda.is_available() is False)")
    from cvc import hip
    hip.lib()
    return torch.device("cuda:0")


def close(a, b, **tol):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, **tol)


_CACHE = {}
_TIE_STATS = {}         # test name -> how close the greedy comparison came to its tolerance (tests/helpers.py::tie_aware_seq_equal)
_BEAM_MATCH = {}        # (config, beam) -> (identical rank-0 sequences, clips), filled by _beam_check; reported at the end of the module


def _inputs(name, seed):
    """Synthetic weights + clip features of a BASELINE config, generated once per (config, seed): at full size the
    counter-based generator takes longer than the GPU side of the test.  One entry is kept (memory)."""
    if _CACHE.get("key") != (name, seed):
        _CACHE.clear()
        d = synth.CONFIGS[name]
        _CACHE.update(key=(name, seed), val=(d, synth.hot_path_state_dict(d, seed), synth.clip_features(d, seed)))
    return _CACHE["val"]


def _beam_check(name, seed, dev, beam=5, min_same=0.9):
    """Beam decode at full size.  Rank-0 sequences must equal the oracle's for at least `min_same` of the clips (a
    near-tie between two hypotheses may flip under fp32 reordering); on those clips the attention maps and the scores
    must agree; on every clip the beam's own best score must be within fp32 noise of, or above, what the oracle found
    minus a near-tie margin, and scores must come out sorted."""
    from helpers import to_dev
    import fullsize_oracle as FO
    from cvc.decode import DecodeEngine, DecodeWeights
    d, sd, f_np = _inputs(name, seed)
    ref, src = FO.beam(name, seed, d, sd, f_np, beam)
    seq_o, att_o, sc_o = torch.from_numpy(ref["seq"]), torch.from_numpy(ref["att"]), torch.from_numpy(ref["scores"])
    eng = DecodeEngine(DecodeWeights(to_dev(sd, dev)), to_dev(f_np, dev), d.T, synth.UNK_IDX, beam=beam)
    seq, att, sc = eng.run()
    seq, att, sc = seq.clone(), att.clone(), sc.clone()
    assert seq.shape == (d.B, d.T) and att.shape == (d.B, d.T, d.N) and sc.shape == (d.B, beam)
    assert bool((sc[:, :-1] >= sc[:, 1:] - 1e-6).all())
    same = (seq.cpu() == seq_o).all(1)
    # the observed rate goes to the test output (pytest -rP / -s, and the GPU log the driver keeps): a slide from 64/64 to 58/64
    # stays inside the 90 % bar but must be visible
    worst = float((sc_o[:, 0] - sc[:, 0].cpu()).max())
    print(f"[beam-check] {name} beam={beam} (oracle: {src}): {int(same.sum())}/{d.B} rank-0 sequences identical to the oracle's; "
          f"largest oracle-minus-engine best-score gap {worst:.2e}")
    _BEAM_MATCH[(name, beam)] = (int(same.sum()), d.B)
    assert int(same.sum()) >= min_same * d.B, f"only {int(same.sum())} of {d.B} rank-0 sequences match the oracle"
    close(att[same.to(dev)], att_o[same], **SEQ_TOL)
    close(sc[same.to(dev)][:, 0], sc_o[same][:, 0], rtol=2e-4, atol=2e-4)
    # a clip whose sequence differs must still have found a hypothesis as good as the oracle's (within a near-tie)
    assert bool((sc[:, 0].cpu() >= sc_o[:, 0] - 5e-3).all())
    # properties: attention rows sum to 1, masked regions carry exactly 0, replay is bitwise deterministic
    close(att.sum(2), torch.ones(d.B, d.T), rtol=1e-5, atol=1e-5)
    m = torch.from_numpy(f_np["pnt_mask"][:, 1:]).to(dev)
    assert float(att.permute(0, 2, 1)[m].abs().max()) == 0.0
    seq2, att2, sc2 = eng.run()
    assert torch.equal(seq, seq2) and torch.equal(att, att2) and torch.equal(sc, sc2)


def test_cfg3_beam5_decode_full_size_vs_oracle(dev):
    """BASELINE config 3 (i): B=64, N=100, F=480, D=2048, T=20, beam=5 (320 live rows per step)."""
    _beam_check("cfg3", 1303, dev)


def test_cfg3_cyclical_forward_backward_full_size_vs_oracle(dev):
    """BASELINE config 3 (ii): the cyclical pass (decode -> ground -> argmax cut -> localize -> reconstruct) at
    B=64, D=2048, T=20 in eval mode, objective 0.5 lm + 0.5 lm_recon (trainer.py:101-109 with cyclical.yml:65-66):
    five losses and every parameter gradient against the oracle's autograd (K = 8192 backward-data GEMMs inside BPTT
    over T = 20, the LDS-DMA ring kernel at R = 2048, the T-batched weight-gradient products)."""
    from helpers import build_model, to_dev, model_call
    from conftest import load_g9
    import fullsize_oracle as FO
    seed = 1303
    d, sd, f = _inputs("cfg3", seed)
    b = synth.label_glue_batch(d, seed)
    ref, _src = FO.cyclical_eval("cfg3", seed, d, sd, f, b)
    # ... and against the REFERENCE's own _forward_3_loops + autograd on the same inputs (tests/golden/g9_fullsize_ref.npz)
    g9 = load_g9("cfg3.cyclical.", FO.inputs_digest(sd, f, b))
    model = build_model(d, sd, dev)
    model.debug_collect = {}
    out = model_call(model, to_dev(f, dev), to_dev(b, dev), False)
    assert len(out) == 5
    (0.5 * out[0].mean() + 0.5 * out[4].mean()).backward()
    grads = {n: (None if p.grad is None else p.grad.detach().reshape(-1).cpu().double()) for n, p in model.named_parameters()
             if not n.startswith("roi_feat_extractor")}
    for src, want in (("oracle", ref), ("reference", g9)):
        for got, w_ in zip(out, want["losses"]):
            assert got.shape == (1,)
            assert float(got.detach()) == pytest.approx(float(w_), rel=1e-4, abs=1e-5), src
        # a10 at full size: grounder output element-wise (masked slots are exactly -1e8 on both sides)
        close(model.debug_collect["ground_weights"], want["ground_weights"], rtol=1e-4, atol=2e-4)
        checked = 0
        for n, got in grads.items():
            if not ("grad_norm." + n in want or "grad_none." + n in want):
                continue
            if "grad_none." + n in want:
                assert got is None or float(got.abs().max()) == 0.0, (src, n)
                continue
            # the gradient is kept as its 2-norm + its elements at a fixed random sample (the whole set is 500 MB): the sampled
            # error, scaled from m sampled to all numel elements, against 5e-4 of the whole norm, and the two norms against each other
            norm = float(want["grad_norm." + n])
            idx = torch.from_numpy(FO.sample_index(n, got.numel()))
            err = float((got[idx] - torch.from_numpy(want["grad_at." + n])).norm()) * (got.numel() / idx.numel()) ** 0.5
            assert err <= 5e-4 * norm + 1e-6, (src, n, err, norm)
            assert abs(float(got.norm()) - norm) <= 2e-4 * norm + 1e-6, (src, n, float(got.norm()), norm)
            checked += 1
        assert checked >= 15, (src, checked)


def test_cfg5_full_size_greedy_and_beam5_vs_oracle(dev):
    """BASELINE config 5 at its real size: B=64, N=300, F=480, D=4096, A=E=2048, T=30 -- greedy through the packed
    engine, then beam=5 (320 rows)."""
    from helpers import to_dev, referee_seq_check
    from conftest import load_g9
    import fullsize_oracle as FO
    from cvc.decode import DecodeEngine, DecodeWeights
    seed = 1505
    d, sd, f_np = _inputs("cfg5", seed)
    ref, _src = FO.greedy("cfg5", seed, d, sd, f_np)
    g9 = load_g9("cfg5.greedy.", FO.inputs_digest(sd, f_np))           # the REFERENCE's own _sample at this size
    seq_o, att_o = torch.from_numpy(g9["seq"]), torch.from_numpy(g9["att2_weights"])
    W, f = DecodeWeights(to_dev(sd, dev)), to_dev(f_np, dev)
    eng = DecodeEngine(W, f, d.T, synth.UNK_IDX).capture()
    seq, att = eng.run()
    seq, att = seq.clone(), att.clone()
    # The tie rule is settled by an fp64 referee (tests/helpers.py::referee_seq_check; round 4 had widened a hand-chosen tolerance
    # after one flake): the GPU's words are held to the fp64 oracle's wherever its margin exceeds twice the MEASURED log-prob
    # deviation (GPU vs referee, asserted <= 1e-4, and fp32 CPU oracle vs referee -- both printed), and a failure names clip, step
    # and all three margins.
    st = referee_seq_check(seq.cpu().numpy(), eng.logprob.t().cpu().numpy(), ref, "cfg5 greedy", ref_seq=g9["seq"])
    _TIE_STATS["cfg5 greedy"] = st
    assert st["reference_equals_referee"]
    same = (seq.cpu() == seq_o).all(1)
    assert int(same.sum()) >= 0.95 * d.B
    close(att[same.to(dev)], att_o[same], **SEQ_TOL)
    seq2, att2 = eng.run()
    assert torch.equal(seq, seq2) and torch.equal(att, att2)
    del eng, W, f, seq_o, att_o
    torch.cuda.empty_cache()
    _beam_check("cfg5", seed, dev)
    _CACHE.clear()


# ------------------------------------------------------------------ a10: grounder forward element-wise + backward
def test_a10_ground_weights_elementwise_golden(g1, dev):
    """captioner.py:132-173 through the product's training pass: ground_weights [B,T,N] element-wise against the
    reference's own tensor (tests/golden/g1_tiny.npz, a9.*.ground_weights)."""
    from helpers import build_model, to_dev, model_call
    d = synth.CONFIGS["tiny"]
    for variant, over in (("a9.cyc.", {}), ("a9.dec.", dict(train_decoder_only=True))):
        model = build_model(d, g1.sub("sd."), dev, **over)
        model.debug_collect = {}
        model_call(model, to_dev(g1.sub("feats."), dev), to_dev(g1.sub("batch."), dev), False)
        gw = model.debug_collect["ground_weights"]
        gold = g1[variant + "ground_weights"]
        assert tuple(gw.shape) == gold.shape == (d.B, d.T, d.N)
        close(gw, gold, rtol=2e-5, atol=2e-5)
        assert np.array_equal(gw.detach().cpu().numpy() == -1e8, gold == -1e8)


@pytest.mark.parametrize("B,T,N,G", [(3, 4, 7, 24), (64, 20, 100, 2048), (2, 1, 1, 4), (5, 3, 130, 36)])
def test_a10_grounder_backward_vs_oracle_autograd(dev, B, T, N, G):
    """cvc.functional._Grounder.backward (captioner.py:171's autograd-visible masked_fill_: no gradient at filled slots)
    against torch autograd of the oracle's grounder, for xt, the region features and the bias."""
    from oracle import ref_cpu as O
    from cvc import functional as F_
    g = torch.Generator().manual_seed(B * 1000 + N)
    xt = torch.randn(B, T, G, generator=g) * 0.3
    feats = torch.relu(torch.randn(B, N, G, generator=g)) * 0.2
    bias = torch.randn(B, T, N, generator=g)
    mask = torch.rand(B, T, N, generator=g) < 0.3
    up = torch.randn(B, T, N, generator=g)
    ref_in = [t.clone().requires_grad_(True) for t in (xt, feats, bias)]
    ref = O.grounder(ref_in[0], ref_in[1], mask, ref_in[2])
    (ref * up).sum().backward()
    got_in = [t.clone().to(dev).requires_grad_(True) for t in (xt, feats, bias)]
    got = F_.grounder(got_in[0], got_in[1], got_in[2], mask.to(dev))
    close(got, ref.detach(), rtol=2e-5, atol=2e-5)
    (got * up.to(dev)).sum().backward()
    for a, r, name in zip(got_in, ref_in, ("xt", "feats", "bias")):
        want = r.grad.double()
        err = float((a.grad.cpu().double() - want).norm())
        assert err <= 2e-5 * float(want.norm()) + 1e-7, (name, err, float(want.norm()))


def test_ground_loss_gradient_through_the_model(g1, dev):
    """A test-only loss mix that gives ground_loss (losses[2], never optimised by trainer.py:106-109) a non-zero weight,
    so that _Grounder.backward runs inside the model's graph: parameter gradients vs the oracle's autograd."""
    from helpers import build_model, to_dev, model_call
    from oracle import ref_cpu as O
    d = synth.CONFIGS["tiny"]
    sd = g1.sub("sd.")
    P = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in O.to_torch(sd).items()}
    for k in list(P):
        if k.startswith("attended_roi_decoder_core.") and "lstm" in k:
            P[k] = P[k.replace("attended_roi_decoder_core.", "decoder_core.")]
    fo = O.to_torch(g1.sub("feats."))
    fo["g_pool_feats"].requires_grad_(True)
    ref = O.cyclical_forward(P, fo, O.to_torch(g1.sub("batch.")), T=d.T, vocab_size=d.V)
    (0.5 * ref[0].mean() + 0.7 * ref[2].mean() + 0.5 * ref[4].mean()).backward()
    model = build_model(d, sd, dev)
    f = to_dev(g1.sub("feats."), dev)
    f["g_pool_feats"].requires_grad_(True)
    out = model_call(model, f, to_dev(g1.sub("batch."), dev), False)
    assert float(out[2].detach()) == pytest.approx(float(ref[2].detach()), rel=1e-4, abs=1e-5)
    (0.5 * out[0].mean() + 0.7 * out[2].mean() + 0.5 * out[4].mean()).backward()
    close(f["g_pool_feats"].grad, fo["g_pool_feats"].grad, rtol=2e-4, atol=2e-6)
    n = 0
    for name, p in model.named_parameters():
        if name not in P or P[name].grad is None:
            continue
        want = P[name].grad.double()
        err = float((p.grad.cpu().double() - want).norm())
        assert err <= 5e-4 * float(want.norm()) + 1e-6, (name, err, float(want.norm()))
        n += 1
    # the grounder's own parameters (vis_embed, vis_classifiers_bias) must have received a gradient
    pe = dict(model.named_parameters())
    assert float(pe["roi_feat_extractor.vis_embed.0.weight"].grad.abs().max()) > 0
    # (vis_classifiers_bias enters as a per-(clip, word) constant over the regions: log_softmax is invariant to it, its true gradient is 0)
    assert n >= 15


def test_nan_logits_keep_selected_indices_in_range(dev):
    """A diverged checkpoint produces NaN logits: torch.topk still returns valid indices (captioner.py:415-416); the
    kernels' sentinel index must never reach a gather (next step's embedding row, beam state reorder)."""
    from cvc import hip
    V, M = 50, 5
    logits = torch.full((M, V), float("nan"), device=dev)
    word = torch.full((M,), -7, dtype=torch.int64, device=dev)
    hip.top2_unk(logits, synth.UNK_IDX, word)
    assert bool(((word >= 0) & (word < V)).all())
    parent, w, score, done = hip.beam_select(logits, torch.zeros(M, device=dev), torch.zeros(M, dtype=torch.uint8, device=dev), 1, 5,
                                            synth.UNK_IDX, False)
    assert bool(((parent >= 0) & (parent < 5)).all()) and bool(((w >= 0) & (w < V)).all())
    lg = torch.full((3, V), float("nan"), device=dev)
    loss, lse, amax = hip.vocab_nll_fwd(lg, torch.zeros(3, dtype=torch.int64, device=dev), torch.ones(3, device=dev))
    assert bool(((amax >= 0) & (amax < V)).all())
    # the fused head: NaN hidden state -> NaN logits inside the GEMM epilogue -> word must still index the table
    d = synth.CONFIGS["tiny"]
    from helpers import to_dev
    from cvc.decode import DecodeEngine, DecodeWeights
    sd = synth.hot_path_state_dict(d, 5)
    sd["logit.bias"] = np.full_like(sd["logit.bias"], np.nan)
    seq, att = DecodeEngine(DecodeWeights(to_dev(sd, dev)), to_dev(synth.clip_features(d, 5), dev), d.T, synth.UNK_IDX).run()
    assert bool(((seq >= 0) & (seq < d.V)).all())


def test_zz_report_beam_match_rates():
    """Not a check of its own: writes the match rates observed by the full-size beam tests of this module into the pytest
    summary (record via warnings so that -q runs show them too)."""
    import warnings
    for (name, beam), (n, B) in sorted(_BEAM_MATCH.items()):
        warnings.warn(f"beam match rate {name} beam={beam}: {n}/{B} rank-0 sequences identical to the oracle's")
    for name, st in sorted(_TIE_STATS.items()):
        warnings.warn(f"greedy tie check {name}: {st}")
