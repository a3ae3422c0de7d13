"""GPU parity tests (pytest -m gpu): the HIP path, called through the C-ABI (cvc.hip ->
libcvc_hip.so), against (1) the golden vectors captured from the reference and (2) the CPU oracle
on the same seeded inputs.

Tolerances (fp32, different reduction orders: MFMA k-permuted fma chains, wave/LDS tree sums,
hardware exp/rcp in tanh/sigmoid): 2e-5 abs+rel on single-op outputs, 1e-4 on anything that went
through T recurrent steps or a log-softmax over V, 2e-4 rel on gradients (SURVEY.md section 7 "Hard parts" 1).
"""
import dataclasses

import numpy as np
import pytest
import torch

from cvc import synth

pytestmark = pytest.mark.gpu

OP_TOL = dict(rtol=2e-5, atol=2e-5)
SEQ_TOL = dict(rtol=1e-4, atol=1e-4)
GRAD_TOL = dict(rtol=2e-4, atol=2e-5)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("the gpu-marked tests need a visible MI355X (torch.cuda.is_available() is False)")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def lib(dev):
    from cvc import hip
    hip.lib()   # fails loudly if the extension is missing
    return hip


def close(a, b, **tol):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, **tol)


@pytest.fixture(scope="module")
def tiny(g1, dev, lib):
    from helpers import build_model, to_dev
    d = synth.CONFIGS["tiny"]
    sd = g1.sub("sd.")
    model = build_model(d, sd, dev)
    return d, model, to_dev(g1.sub("feats."), dev), to_dev(g1.sub("batch."), dev), to_dev(g1.sub("unit."), dev)


# ------------------------------------------------------------------ single ops vs golden (a1..a7)
def test_a1_additive_attention_golden(tiny, g1):
    d, model, f, _, u = tiny
    att = model.decoder_core.soft_attn
    mask = f["pnt_mask"][:, 1:]
    ctx, a, fm = att(u["h"], f["p_pool_feats"], context=f["pool_feats"], mask=mask, proposal_frame_mask=u["fmask"])
    close(ctx, g1["a1.regions.ctx"], **OP_TOL); close(a, g1["a1.regions.attn"], **OP_TOL); close(fm, g1["a1.regions.fm"], **OP_TOL)
    close(a[2], np.full(d.N, 1.0 / d.N, np.float32), rtol=1e-6, atol=0)            # all-masked clip: uniform
    assert float(a[0][mask[0]].abs().max()) == 0.0                                # masked -> exactly 0
    ctx, a, fm = att(u["h"], f["p_conv_feats"], context=f["conv_feats"])
    close(ctx, g1["a1.frames.ctx"], **OP_TOL); close(a, g1["a1.frames.attn"], **OP_TOL); assert fm is None
    ctx, a, _ = att(u["h"], f["p_pool_feats"], mask=mask)
    close(ctx, g1["a1.noctx.ctx"], **OP_TOL); close(a, g1["a1.noctx.attn"], **OP_TOL)


@pytest.mark.parametrize("temp", [1.0, 2.5])
def test_a2_dot_attention_golden(tiny, g1, temp):
    d, model, f, _, u = tiny
    att = model.localizer_core.soft_attn
    att.temp = temp
    ctx, a, fm = att(u["emb"], f["p_pool_feats"], context=f["pool_feats"], mask=f["pnt_mask"][:, 1:],
                     proposal_frame_mask=u["fmask"])
    att.temp = 1.0
    pre = "a2.temp%g." % temp
    close(ctx, g1[pre + "ctx"], **OP_TOL); close(a, g1[pre + "attn"], **OP_TOL); close(fm, g1[pre + "fm"], **OP_TOL)


def test_with_sentinel_minus_inf_fill_golden(tiny, dev):
    """with_sentinel=True through the module shims (modules.py:40-41, 123-124; forwarded by the cores to the region attention,
    decoder_core.py:55, localizer_core.py:37): masked positions are filled with -inf -- against the reference's own outputs
    (tests/golden/g7_sentinel.npz).  Clip 2 is fully masked: NaN there, as in the reference."""
    from conftest import Golden
    g7 = Golden("g7_sentinel.npz")
    d, model, f, _, u = tiny
    mask = f["pnt_mask"][:, 1:]
    core, loc = model.decoder_core, model.localizer_core
    ctx, a, fm = core.soft_attn(u["h"], f["p_pool_feats"], context=f["pool_feats"], mask=mask, proposal_frame_mask=u["fmask"], with_sentinel=True)
    close(ctx, g7["add.ctx"], **OP_TOL); close(a, g7["add.attn"], **OP_TOL); close(fm, g7["add.fm"], **OP_TOL)
    assert torch.isnan(a[2]).all() and torch.isinf(fm).any()
    dot = type(loc.soft_attn)(d.E, d.A, temp=2.5).to(dev)
    dot.load_state_dict(loc.soft_attn.state_dict())
    ctx, a, fm = dot(u["emb"], f["p_pool_feats"], context=f["pool_feats"], mask=mask, proposal_frame_mask=u["fmask"], with_sentinel=True)
    close(ctx, g7["dot.ctx"], **OP_TOL); close(a, g7["dot.attn"], **OP_TOL); close(fm, g7["dot.fm"], **OP_TOL)
    o, st, ra, fma, wp = core(u["emb"], f["fc_feats"], f["conv_feats"], f["p_conv_feats"], f["pool_feats"], f["p_pool_feats"], mask,
                              (u["state_h"], u["state_c"]), proposal_frame_mask=u["fmask"], with_sentinel=True)
    close(o, g7["core.out"], **OP_TOL); close(st[0], g7["core.h"], **OP_TOL); close(st[1], g7["core.c"], **OP_TOL)
    close(ra, g7["core.roi_attn"], **OP_TOL); close(fma, g7["core.fm"], **OP_TOL); close(wp, g7["core.ctx_r"], **OP_TOL)
    a_, b_, c_, _ = loc(u["emb"], f["fc_feats"], f["conv_feats"], f["p_conv_feats"], f["pool_feats"], f["p_pool_feats"], mask, None, None,
                        proposal_frame_mask=u["fmask"], with_sentinel=True)
    close(a_, g7["loc.loc_pool"], **OP_TOL); close(b_, g7["loc.loc_conv"], **OP_TOL); close(c_, g7["loc.prob"], **OP_TOL)


def test_a3_decoder_step_golden(tiny, g1):
    d, model, f, _, u = tiny
    out, (h, c), ra, fm, ctx_r = model.decoder_core(u["emb"], f["fc_feats"], f["conv_feats"], f["p_conv_feats"],
                                                    f["pool_feats"], f["p_pool_feats"], f["pnt_mask"][:, 1:],
                                                    (u["state_h"], u["state_c"]), proposal_frame_mask=u["fmask"])
    for k, v in dict(out=out, h=h, c=c, roi_attn=ra, fm=fm, ctx_r=ctx_r).items():
        close(v, g1["a3." + k], **OP_TOL)


def test_a5_reconstructor_step_golden(tiny, g1):
    d, model, f, _, u = tiny
    out, (h, c) = model.attended_roi_decoder_core(u["emb"], f["fc_feats"], u["loc_pool"], u["loc_conv"],
                                                  (u["state_h"], u["state_c"]))
    close(out, g1["a5.out"], **OP_TOL); close(h, g1["a5.h"], **OP_TOL); close(c, g1["a5.c"], **OP_TOL)


def test_a6_localizer_step_golden(tiny, g1):
    d, model, f, _, u = tiny
    lp, lc, prob, st = model.localizer_core(u["emb"], f["fc_feats"], f["conv_feats"], f["p_conv_feats"], f["pool_feats"],
                                            f["p_pool_feats"], f["pnt_mask"][:, 1:], "state", None, proposal_frame_mask=u["fmask"])
    assert st == "state"
    close(lp, g1["a6.loc_pool"], **OP_TOL); close(lc, g1["a6.loc_conv"], **OP_TOL); close(prob, g1["a6.prob"], **OP_TOL)


def test_a7_embed_logits_vs_oracle(tiny, g1, dev):
    from oracle import ref_cpu as O
    d, model, f, _, u = tiny
    P = O.to_torch(g1.sub("sd."))
    words = torch.tensor([0, 7, synth.UNK_IDX], device=dev)
    close(model._embed(words), O.embed(P, words.cpu()), rtol=0, atol=0)            # gather + relu: bit-exact
    out = u["state_h"][1]
    close(model._logprobs(out), O.logits_logsoftmax(P, out.cpu()), **OP_TOL)


# ------------------------------------------------------------------ attention fwd + recomputing bwd, ragged shapes
@pytest.mark.parametrize("kind", ["additive", "dot"])
@pytest.mark.parametrize("nclip,nq,N,A,R", [
    (3, 1, 37, 20, 28),          # nothing is a multiple of the tile sizes
    (2, 5, 130, 1024, 512),      # 5 beams share a clip's features
    (2, 20, 9, 16, 32),          # the T localizer queries of a clip in one call
    (1, 1, 1, 4, 4),             # a single region
    (4, 1, 600, 260, 2052),      # A, R just past a 256-column block
])
def test_attention_fwd_bwd_vs_oracle(dev, lib, kind, nclip, nq, N, A, R):
    from cvc import functional as F_
    from oracle import ref_cpu as O
    g = torch.Generator().manual_seed(N * 7 + A)
    rows = nclip * nq
    q = torch.randn(rows, A, generator=g)
    proj = torch.randn(nclip, N, A, generator=g)
    ctx = torch.randn(nclip, N, R, generator=g)
    w_a = torch.randn(1, A, generator=g) * 0.3
    b_a = torch.randn(1, generator=g)
    mask = torch.rand(nclip, N, generator=g) < 0.3
    if nclip > 1:
        mask[1] = True                                             # one fully masked clip
    fmask = torch.rand(rows, N, generator=g) < 0.3
    d_ctx = torch.randn(rows, R, generator=g)
    d_fm = torch.randn(rows, N, generator=g) * 0.1
    temp = 1.7

    def run(device, hip_path):
        t = lambda x: x.clone().to(device).requires_grad_(x.dtype.is_floating_point)
        q_, proj_, ctx_, w_, b_ = t(q), t(proj), t(ctx), t(w_a), t(b_a)
        m_, fm_ = mask.to(device), fmask.to(device)
        if hip_path:
            if kind == "additive":
                _, ((c, a, fm),) = F_.attention(lib.ATTN_ADDITIVE, q_, w_, b_, 1.0, [(proj_, ctx_, m_, fm_)])
            else:
                _, ((c, a, fm),) = F_.attention(lib.ATTN_DOT, q_, None, None, 1.0 / temp, [(proj_, ctx_, m_, fm_)])
        else:
            # oracle: identity "h2attn" so that q is the query itself; beams/queries expand the clip
            eye, zero = torch.eye(A), torch.zeros(A)
            pe, ce, me = (x.repeat_interleave(nq, 0) for x in (proj_, ctx_, m_))
            if kind == "additive":
                c, a, fm = O.additive_attention(q_, pe, ce, me, fm_, eye, zero, w_, b_)
            else:
                c, a, fm = O.dot_attention(q_, pe, ce, me, fm_, eye, zero, temp)
        ((c * d_ctx.to(device)).sum() + (fm * d_fm.to(device)).sum()).backward()
        grads = [q_.grad, proj_.grad, ctx_.grad] + ([w_.grad, b_.grad] if kind == "additive" else [])
        return [x.detach().cpu() for x in (c, a, fm)] + [x.detach().cpu() for x in grads]

    got, want = run(dev, True), run("cpu", False)
    names = ["ctx", "attn", "frame_masked", "d_q", "d_proj", "d_ctx_feats", "d_w_alpha", "d_b_alpha"]
    for n, a_, b_ in zip(names, got, want):
        # unscaled randn queries give dot-product scores of magnitude ~sqrt(A): exp() amplifies the fp32
        # rounding of a 1024-term dot product, so the forward bound here is 1e-4 rather than OP_TOL
        tol = dict(rtol=2e-4, atol=2e-4) if n.startswith("d_") else dict(rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(a_.numpy(), b_.numpy(), err_msg=n, **tol)
    close(got[1].sum(1), torch.ones(rows), rtol=1e-5, atol=1e-5)       # softmax rows sum to one
    if nclip > 1:
        close(got[1][nq:2 * nq], torch.full((nq, N), 1.0 / N), rtol=1e-5, atol=0)   # all-masked clip: uniform


@pytest.mark.parametrize("nq", [1, 5, 3])
@pytest.mark.parametrize("N,F,R", [(512, 16, 300), (513, 15, 256), (15, 512, 260), (16, 17, 2048), (1, 600, 64)])
def test_two_set_weighted_sum_at_the_edges_of_its_hoisted_forms(dev, lib, nq, N, F, R):
    """The weighted sums' round-6 forms -- both feature sets' softmax rows made up front (n <= 512 in every set), the one-query form's
    first round of context rows requested before them (n >= 16), 4 or 8 waves per workgroup for groups of 5 queries -- at the sizes
    where each condition flips, against torch fp64: region + frame attention of one decoder step, masks on the region set, one clip
    fully masked, R beside a 256-column block."""
    from cvc import functional as F_
    g = torch.Generator().manual_seed(N * 31 + F * 7 + R + nq)
    nclip, A = 3, 24
    rows = nclip * nq
    q = torch.randn(rows, A, generator=g).to(dev)
    w_a, b_a = (torch.randn(1, A, generator=g) * 0.3).to(dev), torch.randn(1, generator=g).to(dev)
    pr, cr = torch.randn(nclip, N, A, generator=g).to(dev), torch.randn(nclip, N, R, generator=g).to(dev)
    pf, cf = torch.randn(nclip, F, A, generator=g).to(dev), torch.randn(nclip, F, R, generator=g).to(dev)
    mask = (torch.rand(nclip, N, generator=g) < 0.3).to(dev)
    mask[1] = True
    total, ((c_r, a_r, _), (c_f, a_f, _)) = F_.attention(lib.ATTN_ADDITIVE, q, w_a, b_a, 1.0, [(pr, cr, mask, None), (pf, cf, None, None)])

    def ref(proj, ctx, m):
        pe, ce = proj.double().repeat_interleave(nq, 0), ctx.double().repeat_interleave(nq, 0)
        sc = (torch.tanh(pe + q.double()[:, None, :]) * w_a.double()).sum(-1) + b_a.double()
        if m is not None:
            sc = sc.masked_fill(m.repeat_interleave(nq, 0), -1e8)
        a = torch.softmax(sc, 1)
        return a, torch.bmm(a[:, None, :], ce).squeeze(1)
    ar, xr = ref(pr, cr, mask)
    af, xf = ref(pf, cf, None)
    close(a_r, ar.float(), rtol=2e-5, atol=2e-6)
    close(a_f, af.float(), rtol=2e-5, atol=2e-6)
    close(c_r, xr.float(), rtol=2e-5, atol=2e-5)
    close(c_f, xf.float(), rtol=2e-5, atol=2e-5)
    close(total, (xr + xf).float(), rtol=2e-5, atol=2e-5)
    close(a_r[nq:2 * nq], torch.full((nq, N), 1.0 / N, device=dev), rtol=1e-5, atol=0)        # the all-masked clip: uniform


# ------------------------------------------------------------------ concat-GEMM kernels (both code paths)
@pytest.mark.parametrize("M,Nout,ks,gather", [
    (64, 8192, (2048, 2048, 1024, 2048), True),     # att-LSTM shape of cfg2 (LDS-DMA fast path)
    (64, 5000, (2048,), False),                     # logits: Nout not a multiple of 32
    (17, 1024, (2048,), False),                     # ragged M, one 32-row tile
    (33, 520, (128, 256), False),                   # ragged M, two tiles, ragged Nout
    (5, 96, (36, 20), True),                        # generic path: k not a multiple of 128
    (64, 64, (4,), False),                          # minimum k
])
def test_concat_gemm_vs_fp64(dev, lib, M, Nout, ks, gather):
    g = torch.Generator(device="cpu").manual_seed(M * 1000 + Nout)
    K = sum(ks)
    w = (torch.randn(Nout, K, generator=g) / K ** 0.5).to(dev)
    bias = torch.randn(Nout, generator=g).to(dev)
    xs, segs, k0 = [], [], 0
    for si, k in enumerate(ks):
        if gather and si == len(ks) - 1:
            table = torch.randn(50, k, generator=g).to(dev)
            idx = torch.randint(0, 50, (M,), generator=g).to(dev)
            segs.append({"x": table, "idx": idx, "w": w[:, k0:k0 + k], "relu": True})
            xs.append(torch.relu(table[idx]))
        else:
            x = torch.randn(M, k, generator=g).to(dev)
            segs.append({"x": x, "w": w[:, k0:k0 + k]})
            xs.append(x)
        k0 += k
    ref = (torch.cat(xs, 1).double() @ w.double().t() + bias.double()).float()
    outs = []
    for force in (False, True):
        lib.gemm_force_generic(force)
        try:
            outs.append(lib.linear_fwd(segs, bias, M, Nout))
        finally:
            lib.gemm_force_generic(False)
        close(outs[-1], ref, rtol=2e-5, atol=2e-5)
    # LSTM epilogue on the same operands (Nout = 4R)
    if Nout % 32 == 0 and (Nout // 4) % 8 == 0:
        R = Nout // 4
        c_prev = torch.randn(M, R, generator=g).to(dev)
        b2 = torch.randn(Nout, generator=g).to(dev)
        gates = ref.double() + b2.double()
        i_, f_, g_, o_ = gates.chunk(4, 1)
        c_ref = torch.sigmoid(f_) * c_prev.double() + torch.sigmoid(i_) * torch.tanh(g_)
        h_ref = torch.sigmoid(o_) * torch.tanh(c_ref)
        for force in (False, True):
            lib.gemm_force_generic(force)
            try:
                h, c, act = lib.lstm_cell_fwd(segs, bias, b2, c_prev, want_gates=True)
            finally:
                lib.gemm_force_generic(False)
            close(h, h_ref.float(), rtol=2e-5, atol=2e-5); close(c, c_ref.float(), rtol=2e-5, atol=2e-5)
            close(act[:, :R], torch.sigmoid(i_).float(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("M,R,widths", [
    (64, 2048, (2048, 2048, 2048, 1024)),           # lang/att-LSTM of cfg2: h_prev + three input ranges
    (64, 2048, (2048, 1024)),                       # some inputs need no gradient (pre-extracted fc_feats)
    (17, 256, (256, 260, 36)),                      # ragged M, widths not multiples of 128
    (33, 32, (32, 32, 16)),                         # the tiny fixture's shapes: K = 128, one K slice
    (1, 8, (4,)),                                   # minimum
])
def test_backward_data_gemm_vs_fp64(dev, lib, M, R, widths):
    """cvc_lstm_pointwise_bwd (quad output) -> cvc_linear_nn_fwd: dX ranges of the LSTM cell's backward."""
    g = torch.Generator(device="cpu").manual_seed(M * 7 + R)
    K = 4 * R
    gates = torch.rand(M, K, generator=g).to(dev) * 0.8 + 0.1
    c_prev, c_new = torch.randn(M, R, generator=g).to(dev), torch.randn(M, R, generator=g).to(dev)
    d_h, d_c = torch.randn(M, R, generator=g).to(dev), torch.randn(M, R, generator=g).to(dev)
    d_gates, d_c_prev, dq = lib.lstm_pointwise_bwd(d_h, d_c, gates, c_prev, c_new, want_quad=True)
    d_gates2, d_c_prev2 = lib.lstm_pointwise_bwd(d_h, d_c, gates, c_prev, c_new)
    assert torch.equal(d_gates, d_gates2) and torch.equal(d_c_prev, d_c_prev2)
    assert torch.equal(dq.permute(1, 0, 2).reshape(64, K)[:M], d_gates)          # quad layout = [K/4][64][4]
    w_hh = (torch.randn(K, widths[0], generator=g) / K ** 0.5).to(dev)
    w_ih = (torch.randn(K, sum(widths[1:]) + 8, generator=g) / K ** 0.5).to(dev)  # +8: a leading range that is skipped
    ranges, c0 = [(w_hh, 0, widths[0])], 8
    for n in widths[1:]:
        ranges.append((w_ih, c0, n))
        c0 += n
    assert lib.linear_nn_ok(M, K, ranges)
    outs = lib.linear_nn(dq, M, K, ranges)
    again = lib.linear_nn(dq, M, K, ranges)
    for (w, c, n), o, o2 in zip(ranges, outs, again):
        ref = (d_gates.double() @ w[:, c:c + n].double()).float()
        close(o, ref, rtol=2e-5, atol=2e-5)
        assert torch.equal(o, o2)                                               # fixed-order K-slice sum


@pytest.mark.parametrize("M,V", [(1280, 5000), (7, 50), (1, 3)])
def test_fused_vocab_criterion_vs_torch(dev, lib, M, V):
    """cvc_vocab_nll_fwd/bwd == log_softmax -> masked NLL sum -> argmax, and its autograd (fp64 reference)."""
    from cvc import functional as F_
    g = torch.Generator(device="cpu").manual_seed(M + V)
    logits = (torch.randn(M, V, generator=g) * 3).to(dev)
    logits[0, min(5, V - 1)] = logits[0, 1] = logits[0].max() + 1                  # an exact tie: lowest index wins
    target = torch.randint(0, V, (M,), generator=g).to(dev)
    w = (torch.rand(M, generator=g) < 0.7).float().to(dev)
    x = logits.clone().requires_grad_(True)
    loss, amax = F_.vocab_nll(x, target, w)
    (loss * 0.37).sum().backward()
    xr = logits.double().requires_grad_(True)
    lp = torch.log_softmax(xr, 1)
    ref = -(lp.gather(1, target[:, None]).squeeze(1) * w.double()).sum()
    (ref * 0.37).backward()
    close(loss, ref.float().reshape(1), rtol=2e-6 * max(M, 8), atol=1e-5)
    assert torch.equal(amax, lp.max(1)[1]) and int(amax[0]) == 1
    close(x.grad, xr.grad.float(), rtol=2e-5, atol=2e-6)
    loss2, amax2 = F_.vocab_nll(logits, target, w)
    assert torch.equal(loss2, loss.detach()) and torch.equal(amax2, amax)           # fixed-order row sum


@pytest.mark.parametrize("B,beam,V,first", [(64, 5, 5000, False), (3, 2, 50, True), (7, 8, 97, False), (1, 1, 9, True)])
def test_beam_bookkeeping_ops_vs_bruteforce(dev, lib, B, beam, V, first):
    """cvc_beam_select / cvc_gather_rows / cvc_top2_unk against a full scan in torch fp64."""
    g = torch.Generator(device="cpu").manual_seed(B * 100 + beam)
    rows, unk = B * beam, 1
    logits = (torch.randn(rows, V, generator=g) * 2).to(dev)
    score = torch.randn(rows, generator=g).to(dev)
    done = (torch.rand(rows, generator=g) < 0.2).to(dev)
    if first:
        done[:] = False
    parent, word, new_score, new_done = lib.beam_select(logits, score, done.to(torch.uint8), B, beam, unk, first)
    lp = torch.log_softmax(logits.double(), 1)
    cand = score.double().view(B, beam, 1) + lp.view(B, beam, V)
    cand[:, :, unk] = -float("inf")
    frozen = torch.full_like(cand, -float("inf"))
    frozen[:, :, 0] = score.double().view(B, beam)
    cand = torch.where(done.view(B, beam, 1), frozen, cand)
    if first:
        cand[:, 1:] = -float("inf")
    top_v, top_i = torch.sort(cand.view(B, beam * V), dim=1, descending=True, stable=True)
    top_v, top_i = top_v[:, :beam], top_i[:, :beam]
    live = torch.isfinite(top_v)                       # fewer live candidates than `beam`: the fillers are unspecified
    assert torch.equal(parent.view(B, beam)[live], (top_i // V)[live])
    assert torch.equal(word.view(B, beam)[live], (top_i % V)[live])
    close(new_score.view(B, beam)[live], top_v[live].float(), rtol=1e-5, atol=1e-5)
    want_done = done.view(B, beam).gather(1, (top_i // V)) | ((top_i % V) == 0)
    assert torch.equal(new_done.view(B, beam).bool()[live], want_done[live])
    # reorder the recurrent state by parent
    state = torch.randn(rows, 64, generator=g).to(dev)
    moved = lib.gather_rows(state, parent, beam)
    src = (torch.arange(rows, device=dev) // beam) * beam + parent
    assert torch.equal(moved, state[src])
    # greedy top-2 with the UNK rule
    words = torch.empty(rows, dtype=torch.int64, device=dev)
    logp = torch.empty(rows, dtype=torch.float32, device=dev)
    lib.top2_unk(logits, unk, words, 1, logp)
    masked = lp.clone()
    masked[:, unk] = -float("inf")
    assert torch.equal(words, masked.max(1)[1])
    close(logp, masked.max(1)[0].float(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,beam,V,nparts", [(64, 5, 5000, 6), (5, 3, 1024, 2), (2, 8, 6144, 4), (3, 4, 8192, 8), (4, 5, 5000, 3),
                                             (3, 2, 50, 4)])
def test_beam_select_over_gemm_slabs_equals_select_over_the_finished_logits(dev, lib, B, beam, V, nparts):
    """cvc_beam_select_parts (the beam decode's selection reads the vocabulary GEMM's K-slice slabs + bias itself, round 6) against
    cvc_beam_select over the matrix cvc_tile_linear_finish makes of the same slabs: the scan adds slab 0 + slab 1 + ... + bias in
    that pass's order, so parents, words, scores and done flags are EQUAL, bit for bit -- in the float4 form of the row scan
    (V % 4 == 0, 2 / 4 / 6 / 8 slabs) and in the general scan (the last two cases)."""
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + beam * 10 + nparts)
    rows, unk = B * beam, 1
    parts = (torch.randn(nparts, rows, V, generator=g) * 1.5).to(dev)
    bias = torch.randn(V, generator=g).to(dev)
    score = torch.randn(rows, generator=g).to(dev)
    done = (torch.rand(rows, generator=g) < 0.2).to(torch.uint8).to(dev)
    logits = torch.empty(rows, V, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.lib().cvc_tile_linear_finish(parts.data_ptr(), nparts, rows * V, V, bias.data_ptr(), None, rows, V, logits.data_ptr(), V, st) == 0
    for first in (False, True):
        want = lib.beam_select(logits, score, done if not first else torch.zeros_like(done), B, beam, unk, first)
        got = lib.beam_select_parts(parts, bias, score, done if not first else torch.zeros_like(done), B, beam, unk, first)
        for a, b_, what in zip(got, want, ("parent", "word", "score", "done")):
            assert torch.equal(a, b_), (what, first)


@pytest.mark.parametrize("M,K,N", [(64, 6144, 8192), (64, 2048, 5000), (17, 256, 96)])
def test_split_product_gemm_is_fp32_grade(dev, lib, M, K, N):
    """The packed GEMM's default arithmetic (every fp32 operand split exactly into three bf16 terms, six cross
    products on the bf16 MFMA, fp32 accumulate) must be as close to the fp64 result as the plain fp32-MFMA path:
    rms and max error no worse than 1.25x, on operands with mixed magnitudes."""
    from cvc.decode import pack_weights, to_quad
    g = torch.Generator(device="cpu").manual_seed(K + N)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    x = torch.randn(M, K, generator=g).to(dev)
    x[:, ::7] *= 1e-3
    x[:, 3::11] *= 64.0
    x[:, 5::13] *= 1e18                       # wide exponent range: the split terms must not over- / underflow
    w[:, 5::13] *= 1e-18
    x[0, :4] = torch.tensor([0.0, -0.0, 1.0, -2.0 ** -100])
    b = torch.randn(N, generator=g).to(dev)
    wp, xq = pack_weights(w), to_quad(x)
    ref = x.double() @ w.double().t() + b.double()
    st = torch.cuda.current_stream().cuda_stream
    err = {}
    prev = lib.gemm_packed_split(-1)
    try:
        for mode in (0, 1, 2):
            lib.gemm_packed_split(mode)
            y = torch.empty(M, N, device=dev)
            assert lib.lib().cvc_packed_linear_fwd(wp.data_ptr(), xq.data_ptr(), K, b.data_ptr(), M, N, 1, y.data_ptr(), N, None, st) == 0
            e = (y.double() - ref).abs()
            err[mode] = (e.pow(2).mean().sqrt().item(), e.max().item())
            assert err[mode][0] <= 1e-6 * ref.pow(2).mean().sqrt().item(), (mode, err[mode])   # ~1e-7 relative rms in practice
    finally:
        lib.gemm_packed_split(prev)
    for mode in (1, 2):
        assert err[mode][0] <= 1.25 * err[0][0] and err[mode][1] <= 1.25 * err[0][1], err


@pytest.mark.parametrize("nq", [2, 5, 7])
def test_multi_query_additive_scores_large_values_take_the_direct_form(dev, lib, nq):
    """Several additive queries per clip use tanh(p + q) = 1 - 2 / (1 + 2^(Cp) 2^(Cq)) with the two factors computed separately
    (one transcendental per element-query instead of two; csrc/attn_scores.h, round 6).  A workgroup whose queries hold a value
    beyond |10.4| runs the direct form (the product of the factors could overflow / flush); inside a factored workgroup C p is
    clamped to +-62, which changes nothing: with |q| <= 10.4 such a p has tanh(p + q) = +-1 to the last bit either way.  Checked
    against fp64 with planted large entries of both signs (p = 45 with q = -44 is tanh(1), not tanh(inf))."""
    g = torch.Generator().manual_seed(nq)
    nclip, N, A, R = 3, 70, 256, 64
    P = torch.randn(nclip, N, A, generator=g)
    q = torch.randn(nclip * nq, A, generator=g)
    P[0, 3, 5], P[0, 3, 6], P[1, 40, 0] = 45.0, -60.0, 21.0             # clip 0: direct form (its q below); clip 1: factored, p = 21 unclamped
    P[1, 41, 3], P[1, 42, 4], P[1, 42, 5] = 45.0, -60.0, 1e30           # clip 1, factored form: clamped p (saturated tanh)
    q[nq * 2 + 1, 7] = -44.0                                                # clip 2: direct form for the whole workgroup
    q[0, 5] = -44.0                                                         # clip 0, query 0: p + q = 1 at [3, 5]
    w = torch.randn(A, generator=g) * 0.2
    b = torch.randn(1, generator=g)
    ctx = torch.randn(nclip, N, R, generator=g)
    outs, _ = lib.attn_fwd(lib.ATTN_ADDITIVE, q.to(dev), w.to(dev), b.to(dev), 1.0, [dict(proj=P.to(dev), ctx=ctx.to(dev))], nclip, nq)
    scores = outs[0][0].cpu()
    ref = (torch.tanh(P.double().repeat_interleave(nq, 0) + q.double().unsqueeze(1)) * w.double()).sum(2) + b.double()
    close(scores, ref.float(), rtol=2e-5, atol=2e-5)
    assert bool(torch.isfinite(scores).all())


@pytest.mark.gpu_experimental
@pytest.mark.parametrize("M,R,E", [(64, 2048, 1024), (37, 256, 96), (1, 64, 32), (64, 4096, 2048)])
def test_packed_lstm_ksplit_equals_full_k_kernel(dev, lib, M, R, E):
    """cvc_packed_lstm_ks_fwd (256 gate rows x K / S per workgroup, activations shared through LDS, slabs + finishing kernel)
    against cvc_packed_lstm_fwd on the same packed operands: same products, different K summation order -> fp32 noise; and
    against fp64; bitwise run-to-run determinism."""
    from cvc.decode import pack_weights, to_quad, from_quad
    g = torch.Generator().manual_seed(M + R)
    K = 2 * R + E
    w = (torch.randn(4 * R, K, generator=g) / K ** 0.5).to(dev)
    x = torch.randn(M, K, generator=g).to(dev)
    b1, b2 = (torch.randn(4 * R, generator=g) * 0.1).to(dev), (torch.randn(4 * R, generator=g) * 0.1).to(dev)
    gb = (torch.randn(M, 4 * R, generator=g) * 0.2).to(dev)
    c_prev = torch.randn(M, R, generator=g).to(dev)
    wp, xq, cq = pack_weights(w, R), to_quad(x), to_quad(c_prev)
    L = lib.lib()
    S = L.cvc_packed_lstm_ks_slices(K, R)
    assert S in (1, 2, 4, 8)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for mode in ("full", "ks", "ks"):
        h1, h2, c2 = torch.zeros(R // 4, 64, 4, device=dev), torch.zeros(R // 4, 64, 4, device=dev), torch.zeros(R // 4, 64, 4, device=dev)
        if mode == "full":
            rc = L.cvc_packed_lstm_fwd(wp.data_ptr(), xq.data_ptr(), K, b1.data_ptr(), b2.data_ptr(), gb.data_ptr(), cq.data_ptr(), M, R,
                                       h1.data_ptr(), h2.data_ptr(), c2.data_ptr(), st)
        else:
            slab = torch.empty(S * (R // 8) * 2048, device=dev)
            wpk = pack_weights(w, R, pad_quads=8 if len(outs) == 1 else 0)           # staggered and dense block strides
            rc = L.cvc_packed_lstm_ks_fwd(wpk.data_ptr(), xq.data_ptr(), K, b1.data_ptr(), b2.data_ptr(), gb.data_ptr(), cq.data_ptr(), M, R,
                                          h1.data_ptr(), h2.data_ptr(), c2.data_ptr(), slab.data_ptr(), wpk.stride(0), st)
        assert rc == 0
        outs.append((from_quad(h1, M), from_quad(h2, M), from_quad(c2, M)))
    (hf, _, cf), (hk, hk2, ck), (hk_b, _, ck_b) = outs
    assert torch.equal(hk, hk2) and torch.equal(hk, hk_b) and torch.equal(ck, ck_b)
    # finish fused into the GEMM launch (last-arriving K slice sums the slabs in slice order): the two-launch form's result up
    # to the contraction of the cell arithmetic (1 ulp), the same bits launch after launch whichever slice arrives last, and
    # the arrival counters back at zero
    counters = torch.zeros(R // 64, dtype=torch.int32, device=dev)
    slab = torch.empty(S * (R // 8) * 2048, device=dev)
    for _ in range(5):
        h1, h2, c2 = torch.zeros(R // 4, 64, 4, device=dev), torch.zeros(R // 4, 64, 4, device=dev), torch.zeros(R // 4, 64, 4, device=dev)
        rc = L.cvc_packed_lstm_ksf_fwd(wp.data_ptr(), xq.data_ptr(), K, b1.data_ptr(), b2.data_ptr(), gb.data_ptr(), cq.data_ptr(), M, R,
                                       h1.data_ptr(), h2.data_ptr(), c2.data_ptr(), slab.data_ptr(), counters.data_ptr(), st)
        assert rc == 0
        got = (from_quad(h1, M), from_quad(h2, M), from_quad(c2, M))
        close(got[0], hk, rtol=1e-6, atol=1e-6); close(got[2], ck, rtol=1e-6, atol=1e-6)
        assert torch.equal(got[0], got[1])
        if _ > 0:
            assert torch.equal(got[0], first[0]) and torch.equal(got[2], first[2])
        first = got
        assert int(counters.abs().sum()) == 0
    close(hk, hf, rtol=2e-5, atol=2e-5); close(ck, cf, rtol=2e-5, atol=2e-5)
    pre = x.double() @ w.double().t() + b1.double() + b2.double() + gb.double()
    i, f, gg, o = pre.chunk(4, 1)
    c_ref = torch.sigmoid(f) * c_prev.double() + torch.sigmoid(i) * torch.tanh(gg)
    h_ref = torch.sigmoid(o) * torch.tanh(c_ref)
    close(hk, h_ref.float(), rtol=2e-5, atol=2e-5); close(ck, c_ref.float(), rtol=2e-5, atol=2e-5)


@pytest.mark.gpu_experimental
@pytest.mark.parametrize("M,eg,mode", [(64, False, 3), (64, True, 1), (37, True, 2), (64, True, 0), (1, False, 3)])
def test_packed_lstm_exchange_finish_equals_full_k_kernel(dev, lib, M, eg, mode):
    """cvc_packed_lstm_ksx_fwd (K split over 8 workgroups per 256-row tile, every slice finishing one of the tile's blocks after
    the in-launch exchange of the partial tiles) at R = 2048, with and without the embedding-gate term, XCD-local (modes 1-3: the
    slab rows read back with ordinary / non-temporal / sc1 loads) and system-scope (mode 0) exchange: against the full-K kernel (fp32 noise: another K summation order) and fp64, the same bits
    launch after launch, and the error word stays clear."""
    from cvc.decode import pack_weights, to_quad, from_quad
    R, V = 2048, 97
    g = torch.Generator().manual_seed(M + 7 * eg + mode)
    K = 2 * R if eg else 3 * R
    w = (torch.randn(4 * R, K, generator=g) / K ** 0.5).to(dev)
    x = torch.randn(M, K, generator=g).to(dev)
    b1, b2 = (torch.randn(4 * R, generator=g) * 0.1).to(dev), (torch.randn(4 * R, generator=g) * 0.1).to(dev)
    gb = (torch.randn(M, 4 * R, generator=g) * 0.2).to(dev)
    c_prev = torch.randn(M, R, generator=g).to(dev)
    table = (torch.randn(V, 4 * R, generator=g) * 0.3).to(dev) if eg else None
    word = torch.randint(0, V, (M,), generator=g).to(dev) if eg else None
    wp, xq, cq = pack_weights(w, R), to_quad(x), to_quad(c_prev)
    L = lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: None if t is None else t.data_ptr()
    new3 = lambda: [torch.zeros(R // 4, 64, 4, device=dev) for _ in range(3)]
    h1, h2, c2 = new3()
    if eg:
        rc = L.cvc_packed_lstm_embgate_fwd(p(wp), p(xq), K, p(b1), p(b2), p(gb), p(table), p(word), p(cq), M, R, p(h1), p(h2), p(c2), st)
    else:
        rc = L.cvc_packed_lstm_fwd(p(wp), p(xq), K, p(b1), p(b2), p(gb), p(cq), M, R, p(h1), p(h2), p(c2), st)
    assert rc == 0
    hf, cf = from_quad(h1, M), from_quad(c2, M)
    slab = torch.empty(8 * (R // 8) * 2048, device=dev)
    flags = torch.zeros(R // 8 + 1, dtype=torch.int32, device=dev)
    prev = L.cvc_packed_lstm_ksx_local(mode)
    try:
        first = None
        for seq in range(1, 6):
            h1, h2, c2 = new3()
            rc = L.cvc_packed_lstm_ksx_fwd(p(wp), p(xq), K, p(b1), p(b2), p(gb), p(table), p(word), p(cq), M, R, p(h1), p(h2), p(c2),
                                           p(slab), p(flags), seq, st)
            assert rc == 0
            got = (from_quad(h1, M), from_quad(h2, M), from_quad(c2, M))
            assert int(flags[R // 8]) == 0, "exchange finish: a slice's wait ran out (workgroups of a tile not resident together / not on one XCD)"
            assert torch.equal(got[0], got[1])
            if first is not None:
                assert torch.equal(got[0], first[0]) and torch.equal(got[2], first[2])
            first = got
    finally:
        L.cvc_packed_lstm_ksx_local(prev)
    close(first[0], hf, rtol=2e-5, atol=2e-5); close(first[2], cf, rtol=2e-5, atol=2e-5)
    pre = x.double() @ w.double().t() + b1.double() + b2.double() + gb.double()
    if eg:
        pre = pre + table.double()[word]
    i_, f_, gg, o_ = pre.chunk(4, 1)
    c_ref = torch.sigmoid(f_) * c_prev.double() + torch.sigmoid(i_) * torch.tanh(gg)
    h_ref = torch.sigmoid(o_) * torch.tanh(c_ref)
    close(first[0], h_ref.float(), rtol=2e-5, atol=2e-5); close(first[2], c_ref.float(), rtol=2e-5, atol=2e-5)
    # a repeated seq is refused by contract only (the caller's duty); seq 0 is refused outright
    assert L.cvc_packed_lstm_ksx_fwd(p(wp), p(xq), K, p(b1), p(b2), p(gb), p(table), p(word), p(cq), M, R, p(h1), p(h2), p(c2), p(slab),
                                     p(flags), 0, st) != 0


@pytest.mark.parametrize("M,K,N,ksplit", [(320, 6144, 8192, 4), (320, 2048, 5000, 6), (150, 512, 130, 3), (65, 32, 50, 2),
                                          (700, 256, 256, 1), (1, 16, 1, 1), (512, 96, 256, 1), (768, 400, 512, 2), (2560, 2048, 1024, 1),
                                          (1024, 48, 256, 3)])
def test_tile_gemm_vs_fp64(dev, lib, M, K, N, ksplit):
    """cvc_tile_gemm (rows > 64: both operands as bf16 split-term fragments, LDS-DMA ring, K split over workgroups): the slab
    sum against fp64 with an error no worse than an fp32 GEMM's, round trip of the fragment packers, and run-to-run
    bitwise determinism.  Covers the cfg3 beam shapes (320 x 6144 x 8192, the V = 5000 head), ragged M / N, M > 320
    (row chunks) and the one-k-step corner."""
    from cvc.decode import to_frag, from_frag, pack_weights_tile
    g = torch.Generator().manual_seed(M + K + N)
    x = (torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-6, 3, (M, 1), generator=g).float())).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    ra = lib.tile_rows_alloc(M)
    assert ra % 32 == 0 and ra >= M
    xb = torch.zeros(ra // 32, K // 16, 3, 2, 32, 8, dtype=torch.int16, device=dev)
    lib.tile_pack_rows(x, xb)
    assert torch.equal(xb, to_frag(x, ra))                       # device packer == host packer, bit for bit
    assert torch.equal(from_frag(xb, M), x)                      # hi + mid + lo reproduces every fp32 exactly
    wb = pack_weights_tile(w)
    parts = lib.tile_gemm(wb, xb, 0, K, M, N, ksplit)
    y = parts.sum(0)
    ref = x.double() @ w.double().t()
    err = float((y.double() - ref).norm() / ref.norm())
    err32 = float(((x @ w.t()).double() - ref).norm() / ref.norm())
    assert err <= max(2.0 * err32, 3e-7), (err, err32)
    assert torch.equal(parts, lib.tile_gemm(wb, xb, 0, K, M, N, ksplit))
    # the forms of the kernel (every wave copies / loader waves + 8 computing waves / loader waves + 4 wide computing waves / the
    # same with register-load loader waves, round 6) take the same products in the same order: bit-identical slabs
    L = lib.lib()
    prev = L.cvc_tile_gemm_loaders(-1)
    try:
        for form in (0, 1, 2, 4):
            L.cvc_tile_gemm_loaders(form)
            assert torch.equal(parts, lib.tile_gemm(wb, xb, 0, K, M, N, ksplit)), form
    finally:
        L.cvc_tile_gemm_loaders(prev)
    # ... and the 256 x 256 form (whole tiles only; forced here for any grid size)
    if M % 256 == 0 and N % 256 == 0 and M >= 512:
        prev_big = L.cvc_tile_gemm_big(1, 1)
        try:
            assert torch.equal(parts, lib.tile_gemm(wb, xb, 0, K, M, N, ksplit))
        finally:
            L.cvc_tile_gemm_big(prev_big, 192)
    # a K segment of a wider activation buffer: point at its first k step
    if K >= 64:
        xw = torch.zeros(ra // 32, (K + 32) // 16, 3, 2, 32, 8, dtype=torch.int16, device=dev)
        lib.tile_pack_rows(x, xw, k0=32)
        assert torch.equal(lib.tile_gemm(wb, xw, 32, K, M, N, ksplit), parts)


@pytest.mark.parametrize("M,N,K,ak,bk", [(8192, 5120, 2560, True, True), (1280, 2048, 5000, False, True), (50, 16, 12, True, True),
                                          (3, 70, 37, False, False), (1280, 1024, 2048, False, True), (200, 130, 1, True, False)])
def test_tile_mm_backward_products_vs_fp64(dev, lib, M, N, K, ak, bk):
    """cvc.hip.tile_mm: C = A B^T with either operand given transposed (the training pass's dW = dY^T X over T*B rows, the
    vocabulary head's dX over V = 5000 columns, odd little shapes): operands packed by the transposing / any-size packers,
    product on the tile GEMM, error against fp64 no worse than an fp32 GEMM's."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(*((K, M) if ak else (M, K)), generator=g).to(dev)
    b = torch.randn(*((K, N) if bk else (N, K)), generator=g).to(dev)
    c = lib.tile_mm(a, b, a_kmajor=ak, b_kmajor=bk)
    A, B = (a.t() if ak else a), (b.t() if bk else b)
    ref = A.double() @ B.double().t()
    err = float((c.double() - ref).norm() / ref.norm())
    err32 = float(((A @ B.t()).double() - ref).norm() / ref.norm())
    assert c.shape == (M, N) and err <= max(2.0 * err32, 4e-7), (err, err32)
    # operands that are column slices of wider matrices (a weight's K segment) and a packed operand used twice
    if N >= 64:
        wide = torch.randn(b.shape[0], b.shape[1] + 24, generator=g).to(dev)
        sl = wide[:, 8:8 + b.shape[1]]
        close(lib.tile_mm(a, sl, a_kmajor=ak, b_kmajor=bk), (A.double() @ (sl.t() if bk else sl).double().t()).float(), rtol=1e-4, atol=1e-4 * K ** 0.5)
        Ap = lib.TileOperand(a, kmajor=ak)
        assert torch.equal(lib.tile_mm(Ap, b, b_kmajor=bk), c) and torch.equal(lib.tile_mm(Ap, b, b_kmajor=bk), c)


@pytest.mark.parametrize("M,R,beam,nparts", [(320, 2048, 5, 4), (70, 32, 1, 2), (15, 48, 3, 1)])
def test_tile_lstm_finish_and_reorder_pack(dev, lib, M, R, beam, nparts):
    """The tile path's LSTM epilogue (slab sum + biases + per-clip gate term + cell update, h' as fragments) and the
    beam-state reorder fused with the next step's operand packing, against plain torch."""
    from cvc.decode import from_frag
    import ctypes as C
    g = torch.Generator().manual_seed(M * 7 + R)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    E, V = 32, 41
    nclip = M // beam
    parts = rnd(nparts, M, 4 * R) * 0.5                          # packed feature order: blk * 32 + gate * 8 + unit
    b_ih, b_hh, gate_bias, c_prev = rnd(4 * R) * 0.1, rnd(4 * R) * 0.1, rnd(nclip, 4 * R) * 0.3, rnd(M, R)
    ra = lib.tile_rows_alloc(M)
    c_out, h_out = torch.empty(M, R, device=dev), torch.empty(M, R, device=dev)
    f1 = torch.zeros(ra // 32, (3 * R) // 16, 3, 2, 32, 8, dtype=torch.int16, device=dev)
    f2 = torch.zeros(ra // 32, R // 16, 3, 2, 32, 8, dtype=torch.int16, device=dev)
    p1, s1 = lib._frag_ptr(f1, R)
    p2, s2 = lib._frag_ptr(f2, 0)
    L = lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    rc = L.cvc_tile_lstm_finish(parts.data_ptr(), nparts, M * 4 * R, b_ih.data_ptr(), b_hh.data_ptr(), gate_bias.data_ptr(), beam,
                                c_prev.data_ptr(), M, R, c_out.data_ptr(), h_out.data_ptr(), p1, s1, p2, s2, st)
    assert rc == 0
    pre = parts.sum(0).view(M, R // 8, 4, 8).permute(0, 2, 1, 3).reshape(M, 4 * R)        # -> checkpoint order gate * R + hidden
    pre = pre + b_ih + b_hh + gate_bias.repeat_interleave(beam, 0)
    i, f, gg, o = pre.chunk(4, 1)
    c_ref = torch.sigmoid(f) * c_prev + torch.sigmoid(i) * torch.tanh(gg)
    h_ref = torch.sigmoid(o) * torch.tanh(c_ref)
    close(c_out, c_ref, **OP_TOL); close(h_out, h_ref, **OP_TOL)
    assert torch.equal(from_frag(f1, M)[:, R:2 * R], h_out) and torch.equal(from_frag(f2, M), h_out)
    assert float(from_frag(f1, M)[:, :R].abs().max()) == 0.0
    # reorder + pack
    parent = torch.randint(0, beam, (M,), generator=g).to(dev)
    word = torch.randint(0, V, (M,), generator=g).to(dev)
    h_att, c_att, h_lang, c_lang, table = rnd(M, R), rnd(M, R), rnd(M, R), rnd(M, R), rnd(V, E)
    xa = torch.zeros(ra // 32, (2 * R + E) // 16, 3, 2, 32, 8, dtype=torch.int16, device=dev)
    xl = torch.zeros(ra // 32, (3 * R) // 16, 3, 2, 32, 8, dtype=torch.int16, device=dev)
    cap, clp = torch.empty(M, R, device=dev), torch.empty(M, R, device=dev)
    pa, sa = lib._frag_ptr(xa, 0)
    pl, sl = lib._frag_ptr(xl, 2 * R)
    for par in (parent, None):
        rc = L.cvc_tile_reorder_pack(None if par is None else par.data_ptr(), word.data_ptr(), beam, h_att.data_ptr(), c_att.data_ptr(),
                                     h_lang.data_ptr(), c_lang.data_ptr(), table.data_ptr(), E, V, cap.data_ptr(), clp.data_ptr(),
                                     pa, sa, pl, sl, M, R, st)
        assert rc == 0
        src = torch.arange(M, device=dev) if par is None else (torch.arange(M, device=dev) // beam) * beam + par
        want_xa = torch.cat([h_lang[src], torch.relu(table[word]), h_att[src]], 1)
        assert torch.equal(from_frag(xa, M), want_xa)
        assert torch.equal(from_frag(xl, M)[:, 2 * R:], h_lang[src]) and torch.equal(cap, c_att[src]) and torch.equal(clp, c_lang[src])


@pytest.mark.parametrize("M,V,E,pad", [(1280, 5000, 1024, 0.35), (37, 11, 16, 0.5), (1, 3, 4, 0.0), (64, 2, 8, 0.9)])
def test_embedding_backward_long_runs(dev, lib, M, V, E, pad):
    """cvc_embed_relu_bwd (sorted runs summed as 16-row pieces) vs index_add in fp64; a third of the rows share
    word 0 (BOS / padding), as in a teacher-forced batch."""
    g = torch.Generator(device="cpu").manual_seed(M + V)
    table = torch.randn(V, E, generator=g).to(dev)
    idx = torch.randint(0, V, (M,), generator=g)
    idx[torch.rand(M, generator=g) < pad] = 0
    idx = idx.to(dev)
    drop = ((torch.rand(M, E, generator=g) > 0.5).float() * 2).to(dev)
    d_out = torch.randn(M, E, generator=g).to(dev)
    for dm in (drop, None):
        got = lib.embed_relu_bwd(table, idx, dm, d_out)
        ref = torch.zeros(V, E, dtype=torch.float64, device=dev)
        ref.index_add_(0, idx, (d_out if dm is None else d_out * dm).double())
        ref = ref * (table > 0)
        close(got, ref.float(), rtol=2e-5, atol=2e-5)
        assert torch.equal(got, lib.embed_relu_bwd(table, idx, dm, d_out))       # fixed summation order


@pytest.mark.parametrize("graph", [False, True])
def test_module_decode_reuses_engine_across_batches(dev, lib, graph):
    """captioner._sample keeps one bound engine (+ captured graph) per batch shape and copies the next batch's features
    into it: the results must be those of a fresh binding, batch after batch, and a new shape must re-bind."""
    import dataclasses
    from helpers import build_model, to_dev, model_call
    d = dataclasses.replace(synth.CONFIGS["tiny"], B=5, R=64, A=32, E=32, V=97, T=6)
    sd = synth.hot_path_state_dict(d, 3)
    model = build_model(d, sd, dev, hip_graph=graph)
    outs = []
    with torch.no_grad():
        for seed in (21, 22, 21):
            f, b = to_dev(synth.clip_features(d, seed), dev), to_dev(synth.label_glue_batch(d, seed), dev)
            seq, att, _ = model_call(model, f, b, True)
            outs.append((seq.clone(), att.clone()))
            fresh = build_model(d, sd, dev, hip_graph=False)
            seq_f, att_f, _ = model_call(fresh, f, b, True)
            assert torch.equal(seq, seq_f) and torch.equal(att, att_f)
        engine = model._engine_cache[1]
        d2 = dataclasses.replace(d, B=3)
        f, b = to_dev(synth.clip_features(d2, 5), dev), to_dev(synth.label_glue_batch(d2, 5), dev)
        seq, att, _ = model_call(model, f, b, True)
        assert seq.shape[0] == 3 and model._engine_cache[1] is not engine
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
    assert not torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("B,N,F,R,A,E,V,T", [(33, 3, 5, 160, 128, 96, 50, 1), (2, 257, 64, 128, 64, 16, 513, 6),
                                             (31, 257, 130, 64, 64, 32, 513, 6), (7, 1, 5, 256, 32, 32, 1000, 6),
                                             (100, 100, 130, 160, 16, 96, 1000, 3), (64, 3, 64, 32, 16, 32, 513, 6)])
def test_decode_other_widths_vs_oracle(dev, lib, B, N, F, R, A, E, V, T):
    """Greedy decode vs the CPU oracle on widths the fixtures do not use (R, A, E not powers of two / not multiples of
    32, vocabularies that end inside a 32-column block): whichever GEMM path the engine picks must agree."""
    import dataclasses
    from helpers import to_dev, tie_aware_seq_equal
    from oracle import ref_cpu as O
    from cvc.decode import DecodeEngine, DecodeWeights
    d = dataclasses.replace(synth.CONFIGS["tiny"], B=B, N=N, F=F, R=R, A=A, E=E, V=V, T=T)
    sd, f_np = synth.hot_path_state_dict(d, R + A), synth.clip_features(d, R + A)
    with torch.no_grad():
        seq_o, att_o, _, logp_o = O.greedy_sample(O.to_torch(sd), O.to_torch(f_np), T, synth.UNK_IDX, return_logprobs=True)
    eng = DecodeEngine(DecodeWeights(to_dev(sd, dev)), to_dev(f_np, dev), T, synth.UNK_IDX)
    seq, att = eng.run()
    n_exact = tie_aware_seq_equal(seq.cpu().numpy(), seq_o.numpy(), logp_o.numpy())
    assert n_exact >= 0.95 * B * T
    if n_exact == B * T:
        close(att, att_o, **SEQ_TOL)


# ------------------------------------------------------------------ greedy decode (a8)
def test_a8_greedy_tiny_golden(tiny, g1):
    from helpers import model_call, tie_aware_seq_equal
    d, model, f, b, _ = tiny
    seq, att, none = model_call(model, f, b, True)
    assert none is None and seq.shape == (d.B, d.T) and seq.dtype == torch.int64
    n = tie_aware_seq_equal(seq.cpu().numpy(), g1["a8.seq"], g1["a8.logp"])
    assert n == d.B * d.T
    close(att, g1["a8.att2_weights"], **SEQ_TOL)
    assert not (seq == synth.UNK_IDX).any()


@pytest.mark.parametrize("graph", [False, True])
def test_a8_greedy_cfg1_golden(g2, dev, lib, graph):
    from helpers import build_model, to_dev, model_call
    d = synth.CONFIGS["cfg1"]
    seed = int(g2["meta.seed"])
    model = build_model(d, synth.hot_path_state_dict(d, seed), dev, hip_graph=graph)
    f, b = to_dev(synth.clip_features(d, seed), dev), to_dev(synth.label_glue_batch(d, seed), dev)
    seq, att, _ = model_call(model, f, b, True)
    np.testing.assert_array_equal(seq.cpu().numpy(), g2["a8.seq"])      # margins in G2 are >> fp32 noise
    close(att, g2["a8.att2_weights"], **SEQ_TOL)
    # run-to-run bitwise determinism (ordered reductions, no float atomics)
    seq2, att2, _ = model_call(model, f, b, True)
    assert torch.equal(seq, seq2) and torch.equal(att, att2)


def test_a8_greedy_cfg2_vs_oracle_and_properties(dev, lib):
    """BASELINE config 2 (the benchmarked size): decode on the GPU, oracle on the host CPU
    (a few seconds), plus size-independent properties."""
    from helpers import to_dev, referee_seq_check
    from conftest import load_g9
    import fullsize_oracle as FO
    from cvc.decode import DecodeEngine, DecodeWeights
    d = synth.CONFIGS["cfg2"]
    sd = synth.hot_path_state_dict(d, 1236)
    f_np = synth.clip_features(d, 1236)
    ref, _src = FO.greedy("cfg2", 1236, d, sd, f_np)     # oracle.greedy_sample (fp32 + the fp64 referee) on these inputs (stored, or run now)
    g9 = load_g9("cfg2.greedy.", FO.inputs_digest(sd, f_np))     # the REFERENCE's own _sample on the same inputs (tools/make_golden.py g9)
    seq_o, att_o = torch.from_numpy(g9["seq"]), torch.from_numpy(g9["att2_weights"])
    W = DecodeWeights(to_dev(sd, dev))
    f = to_dev(f_np, dev)
    eng = DecodeEngine(W, f, d.T, synth.UNK_IDX).capture()
    seq, att = eng.run()
    seq, att = seq.clone(), att.clone()
    # words: held to the fp64 referee with the measured tie tolerance; attention maps: against the reference's tensor on every clip
    # whose words equal the reference's
    st = referee_seq_check(seq.cpu().numpy(), eng.logprob.t().cpu().numpy(), ref, "cfg2 greedy", ref_seq=g9["seq"])
    assert st["reference_equals_referee"]
    same = (seq.cpu() == seq_o).all(1)
    assert int(same.sum()) >= 0.95 * d.B
    close(att[same.to(dev)], att_o[same], **SEQ_TOL)
    # properties: rows sum to 1; masked regions carry exactly 0 weight
    close(att.sum(2), torch.ones(d.B, d.T), rtol=1e-5, atol=1e-5)
    m = f["pnt_mask"][:, 1:]
    assert float(att.permute(0, 2, 1)[m].abs().max()) == 0.0
    # replay determinism
    seq2, att2 = eng.run()
    assert torch.equal(seq, seq2) and torch.equal(att, att2)
    # permuting the clips permutes the outputs (clips are independent units)
    perm = torch.randperm(d.B, generator=torch.Generator().manual_seed(0)).to(dev)
    fp = {k: v[perm].contiguous() for k, v in f.items()}
    seq_p, att_p = DecodeEngine(W, fp, d.T, synth.UNK_IDX).run()
    assert torch.equal(seq_p, seq[perm])
    close(att_p, att[perm], rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------ cyclical training forward/backward (a9, a10, a11)
@pytest.mark.parametrize("variant,over,mix", [
    ("a9.cyc.", dict(), dict(xe=0.5, w_att2=0.0, cons=0.5)),
    ("a9.sup.", dict(), dict(xe=0.5, w_att2=0.05, cons=0.5)),
    ("a9.dec.", dict(train_decoder_only=True), dict(xe=0.5, w_att2=0.0, cons=0.0)),
])
def test_a9_cyclical_tiny_golden(g1, dev, lib, variant, over, mix):
    from helpers import build_model, to_dev, model_call
    d = synth.CONFIGS["tiny"]
    model = build_model(d, g1.sub("sd."), dev, **over)
    f, b = to_dev(g1.sub("feats."), dev), to_dev(g1.sub("batch."), dev)
    fkeys = ("fc_feats", "conv_feats", "p_conv_feats", "pool_feats", "p_pool_feats", "g_pool_feats")
    for k in fkeys:
        f[k].requires_grad_(True)
    gold = g1.sub(variant)
    losses = model_call(model, f, b, False)
    assert len(losses) == (4 if over else 5)
    for i, l in enumerate(losses):
        assert l.shape == (1,)
        close(l, gold["loss%d" % i].reshape(1), **SEQ_TOL)
    loss = mix["xe"] * losses[0].mean() + mix["w_att2"] * losses[1].mean()
    if len(losses) > 4:
        loss = loss + mix["cons"] * losses[4].mean()
    close(loss, gold["total"], **SEQ_TOL)
    loss.backward()
    params = dict(model.named_parameters())
    n = 0
    for k, v in gold.items():
        if not k.startswith("grad."):
            continue
        name = k[len("grad."):]
        g = f[name[3:]].grad if name.startswith("in.") else params[name].grad
        if v is None:
            assert g is None or float(g.abs().max()) == 0.0, name
        else:
            assert g is not None, name
            close(g, v, **GRAD_TOL)
            n += 1
    assert n > 10


def test_a9_cyclical_cfg1_golden(g2, dev, lib):
    from helpers import build_model, to_dev, model_call
    d = synth.CONFIGS["cfg1"]
    seed = int(g2["meta.seed"])
    model = build_model(d, synth.hot_path_state_dict(d, seed), dev)
    f, b = to_dev(synth.clip_features(d, seed), dev), to_dev(synth.label_glue_batch(d, seed), dev)
    losses = model_call(model, f, b, False)
    gold = g2.sub("a9.cyc.")
    for i, l in enumerate(losses):
        close(l, gold["loss%d" % i].reshape(1), **SEQ_TOL)
    (0.5 * losses[0].mean() + 0.5 * losses[4].mean()).backward()
    params = dict(model.named_parameters())
    for k, v in gold.items():
        if k.endswith(".norm") and not k.startswith("grad.in."):
            name = k[len("grad."):-len(".norm")]
            g = params[name].grad
            np.testing.assert_allclose(float(g.double().norm()), float(v), rtol=1e-3, atol=1e-7)
            idx = torch.from_numpy(gold["grad." + name + ".idx"]).to(dev)
            close(g.reshape(-1)[idx], gold["grad." + name + ".val"], rtol=2e-3, atol=1e-6)
    for dead in ("decoder_core.i2h_2.weight", "decoder_core.localied_fc.weight",
                 "attended_roi_decoder_core.soft_attn.h2attn.weight"):
        assert params[dead].grad is None


@pytest.mark.parametrize("T,B,n,R", [(20, 64, 100, 2048), (4, 3, 7, 32), (32, 5, 130, 260), (1, 2, 65, 8)])
def test_context_feature_gradient_of_all_steps_in_one_pass(dev, lib, T, B, n, R):
    """cvc_ctxfeat_bwd_steps: d_feat[b, i, :] += sum_t attn[t, b, i] d_ctx[t, b, :] with the training loop's arenas ([T][B][n]
    weights, [T][128][R] context gradients) -- the accumulated `bmm(att, context)` backward of modules.py:66-69 / 150-153 over the T
    decoder steps -- against fp64; accumulates onto what d_feat already holds (the localizer's share)."""
    import ctypes as C
    g = torch.Generator().manual_seed(T * 1000 + n)
    attn = torch.softmax(torch.randn(T, B, n, generator=g), -1).to(dev)
    d_ctx = torch.randn(T, 128, R, generator=g).to(dev)
    base = torch.randn(B, n, R, generator=g).to(dev)
    out = base.clone()
    fn = lib.lib().cvc_ctxfeat_bwd_steps
    assert fn(attn.data_ptr(), d_ctx.data_ptr(), T, B, n, R, out.data_ptr(), None) == 0
    ref = base.double() + torch.einsum("tbi,tbr->bir", attn.double(), d_ctx[:, :B].double())
    close(out, ref.float(), rtol=1e-5, atol=1e-5)
    assert fn(attn.data_ptr(), d_ctx.data_ptr(), 33, B, n, R, out.data_ptr(), None) == -1          # T beyond the staged rows: refused


@pytest.mark.parametrize("T,B,n,A,planes", [(20, 64, 100, 1024, 4), (4, 3, 7, 16, 1), (32, 5, 130, 260, 2)])
def test_projected_feature_gradient_of_all_steps_in_one_pass(dev, lib, T, B, n, A, planes):
    """cvc_dproj_bwd_steps: d_proj[b, i, :] += sum_t d_s[t, b, i] w (1 - tanh^2(P[b, i, :] + q_t[b, :])), q_t given as the K-slice
    planes of the h2attn product + bias (what the training loop keeps), against fp64 autograd of the additive score
    (modules.py:112-120) summed over the T steps."""
    g = torch.Generator().manual_seed(T * 100 + n)
    qp = (torch.randn(T, planes, B, A, generator=g) * 0.5).to(dev)
    qb = (torch.randn(A, generator=g) * 0.1).to(dev)
    w = (torch.randn(A, generator=g) * 0.3).to(dev)
    P = torch.randn(B, n, A, generator=g).to(dev)
    ds = torch.randn(T, B, n, generator=g).to(dev)
    base = torch.randn(B, n, A, generator=g).to(dev)
    out = base.clone()
    rc = lib.lib().cvc_dproj_bwd_steps(qp.data_ptr(), planes * B * A, B * A, planes, qb.data_ptr(), w.data_ptr(), P.data_ptr(), ds.data_ptr(),
                                       T, B, n, A, out.data_ptr(), None)
    assert rc == 0
    Pd = P.double().requires_grad_(True)
    q = qp.double().sum(1) + qb.double()                                   # [T, B, A]
    scores = (torch.tanh(Pd.unsqueeze(0) + q.unsqueeze(2)) * w.double()).sum(-1)        # [T, B, n]
    (scores * ds.double()).sum().backward()
    close(out, (base.double() + Pd.grad).float(), rtol=2e-5, atol=2e-5)


# ------------------------------------------------------------------ beam search (build-defined)
def test_beam_vs_oracle(tiny, g1, dev):
    from oracle import ref_cpu as O
    from cvc.decode import DecodeEngine
    d, model, f, _, _ = tiny
    P, fo = O.to_torch(g1.sub("sd.")), O.to_torch(g1.sub("feats."))
    W = model.decode_weights()
    for beam in (1, 3):
        with torch.no_grad():
            seq_o, att_o, sc_o = O.beam_search(P, fo, d.T, synth.UNK_IDX, beam)
        if beam == 1:
            seq, att = DecodeEngine(W, f, d.T, synth.UNK_IDX, beam=1).run()
            # greedy never freezes after EOS; the beam rule does: compare up to the first EOS
            for bi in range(d.B):
                s = seq_o[bi].tolist()
                L = (s.index(0) + 1) if 0 in s else d.T
                assert seq[bi, :L].tolist() == s[:L]
        else:
            seq, att, sc = DecodeEngine(W, f, d.T, synth.UNK_IDX, beam=beam).run()
            np.testing.assert_array_equal(seq.cpu().numpy(), seq_o.numpy())
            close(att, att_o, **SEQ_TOL)
            close(sc, sc_o, **SEQ_TOL)
            assert bool((sc[:, :-1] >= sc[:, 1:]).all())


@pytest.mark.parametrize("B,N,F,R,A,E,V,T,beam", [(3, 7, 5, 32, 16, 16, 50, 4, 3), (20, 30, 12, 256, 64, 48, 300, 5, 5),
                                                 (70, 9, 4, 64, 32, 32, 97, 3, 1), (130, 5, 3, 48, 20, 16, 50, 3, 2)])
def test_tile_path_equals_ring_path(dev, lib, B, N, F, R, A, E, V, T, beam):
    """The fragment / tile-GEMM path of the decode engine (rows > 64 or beam search) against the row-major ring path on the
    same inputs: sequences identical, attention maps and scores to fp32 reordering noise.  Includes more than 320 rows
    (row chunks of the tile GEMM) and greedy with B > 64."""
    from helpers import to_dev
    from cvc.decode import DecodeEngine, DecodeWeights
    d = dataclasses.replace(synth.CONFIGS["tiny"], B=B, N=N, F=F, R=R, A=A, E=E, V=V, T=T)
    sd, f_np = synth.hot_path_state_dict(d, 31 + B), synth.clip_features(d, 31 + B)
    W, f = DecodeWeights(to_dev(sd, dev)), to_dev(f_np, dev)
    e_tile, e_ring = DecodeEngine(W, f, T, synth.UNK_IDX, beam=beam), DecodeEngine(W, f, T, synth.UNK_IDX, beam=beam, path="ring")
    assert e_tile.tile and not e_ring.tile and not e_ring.packed
    a, b = [x.clone() for x in e_tile.run()], [x.clone() for x in e_ring.run()]
    same = (a[0] == b[0]).all(1)
    assert int(same.sum()) >= B - max(1, B // 20)
    close(a[1][same], b[1][same], **SEQ_TOL)
    if beam > 1:
        close(a[2][same], b[2][same], rtol=2e-4, atol=2e-4)
    if beam > 1:                                                   # the backtrack kernel against host-side indexing
        h = e_tile._backtrack_host()
        assert torch.equal(a[0], h[0]) and torch.equal(a[1], h[1]) and torch.equal(a[2], h[2])
    a2 = e_tile.capture().run()
    assert all(torch.equal(x, y) for x, y in zip(a, a2))          # graph replay == eager, bitwise


@pytest.mark.parametrize("B,beam,dims", [(4, 1, dict(N=20, F=12, R=128, A=64, E=64, V=300, T=6)),        # packed path
                                         (70, 1, dict(N=9, F=4, R=64, A=32, E=32, V=97, T=3)),          # tile path, greedy
                                         (5, 3, dict(N=7, F=5, R=32, A=16, E=16, V=50, T=4))])          # tile path, beam
def test_cabi_decode_driver_equals_python_launch_list(dev, lib, B, beam, dims):
    """cvc_decode_greedy / cvc_decode_beam (csrc/decode_driver.hip: the whole T-step decode enqueued by one host call from a
    bound cvc_decode_desc) against the same engine walking its launch list in Python: bitwise identical outputs, eager and
    from a captured HIP graph; the plan reports its launch count."""
    from helpers import to_dev
    from cvc.decode import DecodeEngine, DecodeWeights
    d = dataclasses.replace(synth.CONFIGS["tiny"], B=B, **dims)
    sd, f_np = synth.hot_path_state_dict(d, 91 + B), synth.clip_features(d, 91 + B)
    W, f = DecodeWeights(to_dev(sd, dev)), to_dev(f_np, dev)
    e_drv, e_py = DecodeEngine(W, f, d.T, synth.UNK_IDX, beam=beam), DecodeEngine(W, f, d.T, synth.UNK_IDX, beam=beam, driver=False)
    assert e_drv._plan is not None and e_py._plan is None
    a, b = [x.clone() for x in e_drv.run()], [x.clone() for x in e_py.run()]
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    n = lib.lib().cvc_decode_num_launches(e_drv._plan)
    assert n >= 7 * d.T
    a2 = [x.clone() for x in e_drv.capture().run()]
    assert all(torch.equal(x, y) for x, y in zip(a, a2))
    assert all(torch.equal(x, y) for x, y in zip(a, e_drv.run()))          # replay again: state is reset inside the driver


@pytest.mark.gpu_experimental
@pytest.mark.parametrize("form", [True, "fused"])
def test_decode_with_ksplit_gate_gemms_equals_default_engine(dev, lib, form):
    """Greedy decode with the K-split gate GEMMs (two-launch form and finish fused into the last-arriving slice,
    csrc/gemm_packed_ks.hip) against the default full-K engine: same word sequences, attention within the recurrent
    tolerance, eager == captured graph replayed twice (the fused form's arrival counters come back to zero)."""
    from helpers import to_dev
    from cvc.decode import DecodeEngine, DecodeWeights
    d = dataclasses.replace(synth.CONFIGS["tiny"], B=64, N=20, F=12, R=256, A=64, E=64, V=300, T=6)
    sd, f_np = synth.hot_path_state_dict(d, 55), synth.clip_features(d, 55)
    W, f = DecodeWeights(to_dev(sd, dev)), to_dev(f_np, dev)
    ref = [x.clone() for x in DecodeEngine(W, f, d.T, synth.UNK_IDX).run()]
    e = DecodeEngine(W, f, d.T, synth.UNK_IDX, gate_ksplit=form)
    assert e.ks_att > 0 and e.ks_lang > 0 and e.gate_fused == (form == "fused")
    a = [x.clone() for x in e.run()]
    same = (a[0] == ref[0]).all(1)
    assert int(same.sum()) >= d.B - 1
    close(a[1][same], ref[1][same], **SEQ_TOL)
    e.capture()
    for _ in range(2):
        assert all(torch.equal(x, y) for x, y in zip(a, e.run()))


@pytest.mark.gpu_experimental
def test_decode_with_two_block_gate_gemm_workgroups(dev, lib):
    """The selectable 64-row-workgroup form of the packed gate GEMM (cvc_packed_lstm_wg_blocks(2): two weight blocks share
    every activation line through the L1) against the default: same sequences, attention within the recurrent tolerance."""
    from helpers import to_dev
    from cvc.decode import DecodeEngine, DecodeWeights
    d = dataclasses.replace(synth.CONFIGS["tiny"], B=64, N=20, F=12, R=256, A=64, E=64, V=300, T=6)
    sd, f_np = synth.hot_path_state_dict(d, 56), synth.clip_features(d, 56)
    W, f = DecodeWeights(to_dev(sd, dev)), to_dev(f_np, dev)
    ref = [x.clone() for x in DecodeEngine(W, f, d.T, synth.UNK_IDX).run()]
    prev = lib.lib().cvc_packed_lstm_wg_blocks(2)
    try:
        e = DecodeEngine(W, f, d.T, synth.UNK_IDX)
        a = [x.clone() for x in e.run()]
        assert all(torch.equal(x, y) for x, y in zip(a, e.run()))
    finally:
        lib.lib().cvc_packed_lstm_wg_blocks(prev)
    same = (a[0] == ref[0]).all(1)
    assert int(same.sum()) >= d.B - 1
    close(a[1][same], ref[1][same], **SEQ_TOL)


def test_beam5_cfg1_vs_oracle(dev, lib):
    """beam=5 at config-1 size (rows = 20, V = 5000): sequences/scores vs the CPU beam oracle; where the oracle's
    own candidate margin is inside fp32 noise the comparison stops at that step (tie-aware)."""
    from helpers import to_dev
    from oracle import ref_cpu as O
    from cvc.decode import DecodeEngine, DecodeWeights
    d = synth.CONFIGS["cfg1"]
    sd, f_np = synth.hot_path_state_dict(d, 77), synth.clip_features(d, 77)
    with torch.no_grad():
        seq_o, att_o, sc_o = O.beam_search(O.to_torch(sd), O.to_torch(f_np), d.T, synth.UNK_IDX, 5)
    seq, att, sc = DecodeEngine(DecodeWeights(to_dev(sd, dev)), to_dev(f_np, dev), d.T, synth.UNK_IDX, beam=5).run()
    close(sc[:, 0], sc_o[:, 0], rtol=2e-4, atol=2e-4)
    assert bool((sc[:, :-1] >= sc[:, 1:] - 1e-6).all())
    same = (seq.cpu() == seq_o).all(1)
    assert int(same.sum()) >= d.B - 1
    close(att[same.to(dev)], att_o[same], **SEQ_TOL)


def test_cfg5_dims_beam_and_greedy_vs_oracle(dev, lib):
    """BASELINE config 5 dimensions (N=300, D=4096, A=E=2048: the widest template instantiations, 8 KB
    feature rows) at a reduced clip count / step count so that the CPU oracle finishes in seconds."""
    import dataclasses
    from helpers import to_dev, tie_aware_seq_equal
    from oracle import ref_cpu as O
    from cvc.decode import DecodeEngine, DecodeWeights
    d = dataclasses.replace(synth.CONFIGS["cfg5"], B=3, T=5, F=96)
    sd, f_np = synth.hot_path_state_dict(d, 55), synth.clip_features(d, 55)
    P, fo = O.to_torch(sd), O.to_torch(f_np)
    W, f = DecodeWeights(to_dev(sd, dev)), to_dev(f_np, dev)
    with torch.no_grad():
        seq_o, att_o, _, logp_o = O.greedy_sample(P, fo, d.T, synth.UNK_IDX, return_logprobs=True)
        seq_b, att_b, sc_b = O.beam_search(P, fo, d.T, synth.UNK_IDX, 5)
    seq, att = DecodeEngine(W, f, d.T, synth.UNK_IDX).run()
    assert tie_aware_seq_equal(seq.cpu().numpy(), seq_o.numpy(), logp_o.numpy()) == d.B * d.T
    close(att, att_o, **SEQ_TOL)
    seq5, att5, sc5 = DecodeEngine(W, f, d.T, synth.UNK_IDX, beam=5).run()
    close(sc5[:, 0], sc_b[:, 0], rtol=2e-4, atol=2e-4)
    same = (seq5.cpu() == seq_b).all(1)
    assert int(same.sum()) >= d.B - 1
    close(att5[same.to(dev)], att_b[same], **SEQ_TOL)


def test_dot_product_decoder_variant(g1, dev, lib):
    """opts.softattn_type != 'additive' (reference decoder_core.py:24-25): SoftAttention inside the decoder, with
    softmax_temp applied (modules.py:37)."""
    from helpers import to_dev
    from oracle import ref_cpu as O
    from cvc.decode import DecodeEngine, DecodeWeights
    d = synth.CONFIGS["tiny"]
    sd = {k: v for k, v in synth.hot_path_state_dict(d, 3, softattn_type="dot").items()}
    f_np = synth.clip_features(d, 3, full_mask_clip=2)
    # make the dot-product scores O(1) so that the comparison is not dominated by exp() of huge scores
    sd["decoder_core.soft_attn.h2attn.weight"] = sd["decoder_core.soft_attn.h2attn.weight"] * np.float32(0.3)
    with torch.no_grad():
        seq_o, att_o, _, logp_o = O.greedy_sample(O.to_torch(sd), O.to_torch(f_np), d.T, synth.UNK_IDX, softattn_type="dot",
                                                  temp=2.0, return_logprobs=True)
    eng = DecodeEngine(DecodeWeights(to_dev(sd, dev), softattn_type="dot"), to_dev(f_np, dev), d.T, synth.UNK_IDX,
                       inv_temp=1.0 / 2.0)
    seq, att = eng.run()
    np.testing.assert_array_equal(seq.cpu().numpy(), seq_o.numpy())
    close(att, att_o, **SEQ_TOL)


@pytest.mark.parametrize("B,N,F", [(1, 1, 1), (33, 50, 17), (63, 129, 5), (64, 1000, 64), (5, 3, 300), (65, 9, 4), (150, 20, 7)])
def test_decode_ragged_batch_and_region_counts(dev, lib, B, N, F):
    """Edge shapes of the packed decode path (M < 64 padding, one / two MFMA row tiles, a single region, N beyond
    one softmax pass per thread) vs the CPU oracle; R = 64, A = E = 32 keep the oracle fast."""
    import dataclasses
    from helpers import to_dev, tie_aware_seq_equal
    from oracle import ref_cpu as O
    from cvc.decode import DecodeEngine, DecodeWeights
    d = dataclasses.replace(synth.CONFIGS["tiny"], B=B, N=N, F=F, R=64, A=32, E=32, V=97, T=6)
    sd, f_np = synth.hot_path_state_dict(d, 11), synth.clip_features(d, 11, full_mask_clip=0 if B > 1 else None)
    with torch.no_grad():
        seq_o, att_o, _, logp_o = O.greedy_sample(O.to_torch(sd), O.to_torch(f_np), d.T, synth.UNK_IDX, return_logprobs=True)
    eng = DecodeEngine(DecodeWeights(to_dev(sd, dev)), to_dev(f_np, dev), d.T, synth.UNK_IDX)
    assert eng.packed == (B <= 64)          # more than 64 clips per GPU: the row-major path (any M), same results
    seq, att = eng.run()
    n_exact = tie_aware_seq_equal(seq.cpu().numpy(), seq_o.numpy(), logp_o.numpy())
    assert n_exact >= 0.95 * B * d.T
    same = (seq.cpu() == seq_o).all(1)
    close(att[same.to(dev)], att_o[same], **SEQ_TOL)


def test_bad_arguments_are_rejected_not_executed(dev, lib):
    """The C-ABI returns CVC_E_BADARG for violated preconditions; the binding raises."""
    x = torch.zeros(4, 6, device=dev)                                  # k = 6 is not a multiple of 4
    w = torch.zeros(8, 6, device=dev)
    with pytest.raises(RuntimeError, match="bad argument"):
        lib.linear_fwd([{"x": x, "w": w}], None, 4, 8)
    with pytest.raises(RuntimeError, match="contiguous|GPU|float32"):
        lib.log_softmax_fwd(torch.zeros(4, 8, device=dev).t())
    with pytest.raises(RuntimeError, match="bad argument"):
        lib.lstm_cell_fwd([{"x": torch.zeros(2, 8, device=dev), "w": torch.zeros(4 * 12, 8, device=dev)}], None, None,
                          torch.zeros(2, 12, device=dev))            # R = 12 is not a multiple of 8


def test_product_path_has_no_cpu_fallback(lib):
    from cvc import functional as F_
    x = torch.zeros(4, 8)
    w = torch.zeros(8, 8)
    with pytest.raises(RuntimeError, match="GPU"):
        F_.linear(x, w, None)@pytest.mark.gpu
@pytest.mark.parametrize("M,R,widths", [(64, 64, (32, 64, 32)), (20, 128, (64, 128, 96)), (3, 40, (20, 36)), (64, 1024, (512, 1024, 512))])
def test_lstm_train_form_of_packed_gemm(dev, lib, M, R, widths):
    """Training forward of nn.LSTMCell on the decode engine's packed gate GEMM (cvc_pack_lstm_weights + cvc_pack_quad_segs +
    cvc_packed_lstm_train_fwd, decoder_core.py:45-50): the weight pack equals the host-side pack bit for bit, h / c / gates
    match fp64 and the ring kernel, the autograd function gives the same gradients on either forward, and a pack is rebuilt
    after an in-place weight update."""
    import cvc.functional as F_
    from cvc.decode import pack_weights, to_quad
    g = torch.Generator().manual_seed(M * 13 + R)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    K_ih = sum(widths)
    w_ih, w_hh = rnd(4 * R, K_ih) / (K_ih + R) ** 0.5, rnd(4 * R, R) / (K_ih + R) ** 0.5
    b_ih, b_hh = rnd(4 * R) * 0.1, rnd(4 * R) * 0.1
    xs, h_prev, c_prev = [rnd(M, w) for w in widths], rnd(M, R), rnd(M, R)
    assert lib.lstm_train_ok(M, R, list(widths) + [R])
    # operands
    wp = lib.lstm_train_pack(w_ih, w_hh)
    assert torch.equal(wp, pack_weights(torch.cat([w_ih, w_hh], 1), R))
    assert torch.equal(lib.pack_quad_segs([*xs, h_prev]), to_quad(torch.cat([*xs, h_prev], 1)))
    # forward
    h, c, gates = lib.lstm_cell_train_fwd(xs, h_prev, c_prev, wp, b_ih, b_hh)
    segs, k0 = [], 0
    for x in xs:
        segs.append({"x": x, "w": w_ih[:, k0:k0 + x.shape[1]]}); k0 += x.shape[1]
    segs.append({"x": h_prev, "w": w_hh})
    h_r, c_r, g_r = lib.lstm_cell_fwd(segs, b_ih, b_hh, c_prev, want_gates=True)
    X = torch.cat([*xs, h_prev], 1).double()
    pre = X @ torch.cat([w_ih, w_hh], 1).double().t() + b_ih.double() + b_hh.double()
    i, f, gg, o = pre.chunk(4, 1)
    c64 = torch.sigmoid(f) * c_prev.double() + torch.sigmoid(i) * torch.tanh(gg)
    h64 = torch.sigmoid(o) * torch.tanh(c64)
    g64 = torch.cat([torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)], 1)
    for got, ring, ref in ((h, h_r, h64), (c, c_r, c64), (gates, g_r, g64)):
        close(got, ref.float(), **OP_TOL); close(got, ring, **OP_TOL)
    # autograd: same gradients whichever forward ran
    grads = {}
    for packed in (True, False):
        F_.PACKED_LSTM_FORWARD = packed
        try:
            leaves = [t.clone().requires_grad_(True) for t in (w_ih, w_hh, b_ih, b_hh, h_prev, c_prev, *xs)]
            h2, c2 = F_.lstm_cell(leaves[6:], leaves[4], leaves[5], *leaves[:4])
            ((h2 * h64.float()).sum() + (c2 * c64.float()).sum() * 0.5).backward()
            grads[packed] = [t.grad for t in leaves]
        finally:
            F_.PACKED_LSTM_FORWARD = True
    for a, b in zip(grads[True], grads[False]):
        close(a, b, rtol=1e-4, atol=1e-4)
    # one copy of h' per consumer (copies = 3): identical tensors out, and gradients fed through three of them give what the
    # sum fed through one does (the copies' gradients are summed inside the gate-gradient kernel, not by autograd)
    for packed in (True, False):
        F_.PACKED_LSTM_FORWARD = packed
        try:
            la = [t.clone().requires_grad_(True) for t in (w_ih, w_hh, b_ih, b_hh, h_prev, c_prev, *xs)]
            ha, hb, hc, c3 = F_.lstm_cell(la[6:], la[4], la[5], *la[:4], copies=3)
            assert torch.equal(ha, hb) and torch.equal(ha, hc)
            g1, g2, g3 = h64.float(), torch.sin(h64.float() * 3), torch.cos(h64.float() * 5)
            ((ha * g1).sum() + (hb * g2).sum() + (hc * g3).sum() + (c3 * c64.float()).sum() * 0.5).backward()
            lb = [t.clone().requires_grad_(True) for t in (w_ih, w_hh, b_ih, b_hh, h_prev, c_prev, *xs)]
            h1, c1 = F_.lstm_cell(lb[6:], lb[4], lb[5], *lb[:4])
            ((h1 * (g1 + g2 + g3)).sum() + (c1 * c64.float()).sum() * 0.5).backward()
            for a, b in zip(la, lb):
                close(a.grad, b.grad, rtol=1e-4, atol=1e-4)
        finally:
            F_.PACKED_LSTM_FORWARD = True
    # an in-place update of the weights invalidates the pack
    w_ih.mul_(0.5)
    assert lib.lstm_train_pack(w_ih, w_hh) is wp            # same buffer, rebuilt in place
    h3, _, _ = lib.lstm_cell_train_fwd(xs, h_prev, c_prev, wp, b_ih, b_hh)
    pre = X @ torch.cat([w_ih, w_hh], 1).double().t() + b_ih.double() + b_hh.double()
    i, f, gg, o = pre.chunk(4, 1)
    h64b = torch.sigmoid(o) * torch.tanh(torch.sigmoid(f) * c_prev.double() + torch.sigmoid(i) * torch.tanh(gg))
    close(h3, h64b.float(), **OP_TOL)





# ------------------------------------------------------------------ label glue + supervised attention criteria (section 8(f) rank 2)
@pytest.mark.parametrize("cfg", ["tiny", "cfg1", "cfg3"])
def test_label_glue_kernels_bit_exact_vs_oracle(dev, lib, cfg):
    """cvc_bbox_overlaps_fwd / cvc_label_glue_fwd (all T steps per launch) against the oracle's per-step bbox_overlaps /
    bbox_target / frame_mask_on_proposals: IoU values and every bool, bit for bit.  Degenerate boxes, frame mismatches and a
    masked proposal column included."""
    from oracle import ref_cpu as O
    from cvc import hip
    d = synth.CONFIGS[cfg]
    b = synth.label_glue_batch(d, 31)
    f = synth.clip_features(d, 31)
    prop, gtb = torch.from_numpy(b["proposals"]).clone(), torch.from_numpy(b["gt_bboxs"]).clone()
    prop[0, 1, 2:4] = prop[0, 1, 0:2]                       # degenerate proposal (1 x 1): -1 everywhere
    gtb[-1, 0, 2:4] = gtb[-1, 0, 0:2]                       # degenerate ground-truth box: 0
    frm_mask = torch.from_numpy(b["frm_mask"])
    pnt = torch.from_numpy(f["pnt_mask"]).clone()
    pnt[0, 3] = True
    box_mask = torch.from_numpy(b["box_mask"])
    T, N = d.T, d.N
    ov_o = O.bbox_overlaps(prop, gtb, frm_mask | pnt[:, 1:].unsqueeze(-1))
    ov = hip.bbox_overlaps(prop.to(dev), gtb.to(dev), frm_mask.to(dev), pnt[:, 1:].to(dev))
    assert torch.equal(ov.cpu().view(torch.int32), ov_o.view(torch.int32)), float((ov.cpu() - ov_o).abs().max())
    lab, fmo, steps = hip.label_glue(ov, box_mask.to(dev)[:, 0, :, 1:T + 1], frm_mask.to(dev), pnt.to(dev))
    for t in range(T):
        bm_t = box_mask[:, 0, :, t + 1]
        assert torch.equal(lab[:, t].cpu(), O.bbox_target(bm_t, ov_o)), t
        want = O.frame_mask_on_proposals(bm_t, frm_mask, pnt)
        assert torch.equal(fmo[:, t].cpu(), want), t
        assert torch.equal(steps[t].cpu(), want[:, 1:]), t
    assert lab.any() and not lab.all() and fmo[:, :, 1:].any() and not fmo.all()


@pytest.mark.parametrize("B,T,N", [(3, 4, 7), (64, 20, 100)])
def test_attn_nll_kernels_vs_fp64(dev, lib, B, T, N):
    """att2_loss / ground_loss (misc/utils.py:150-162) forward + backward from cvc_attn_nll_* against the formula in fp64; one
    input a transposed view (the decode loop hands att2_weights over as [T, B, N] storage); the no-label case gives exactly 0."""
    from cvc import functional as F_
    g = torch.Generator().manual_seed(B + N)
    x0s = torch.randn(T, B, N, generator=g).to(dev).requires_grad_(True)          # storage [T, B, N], used as [B, T, N]
    x1 = (torch.randn(B, T, N, generator=g) * 3).to(dev).requires_grad_(True)
    x1.data[0, 0, :3] = -1e8                                                       # masked slots of the grounder
    tgt = (torch.rand(B, T, N, generator=g) < 0.05).to(dev)
    l0, l1 = F_.attn_nll(x0s.transpose(0, 1), x1, tgt)
    (0.3 * l0.sum() + 0.7 * l1.sum()).backward()
    a0, a1 = x0s.detach().double().transpose(0, 1).clone().requires_grad_(True), x1.detach().double().clone().requires_grad_(True)
    cnt = tgt.double().sum().clamp(min=1.0)
    r0 = -(torch.log_softmax(a0, 2) * tgt.double()).sum() / cnt
    r1 = -(torch.log_softmax(a1, 2) * tgt.double()).sum() / cnt
    (0.3 * r0 + 0.7 * r1).backward()
    np.testing.assert_allclose(float(l0), float(r0), rtol=2e-5)
    np.testing.assert_allclose(float(l1), float(r1), rtol=2e-5)
    np.testing.assert_allclose(x0s.grad.transpose(0, 1).cpu().double().numpy(), a0.grad.cpu().numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(x1.grad.cpu().double().numpy(), a1.grad.cpu().numpy(), rtol=1e-4, atol=1e-7)
    z0, z1 = F_.attn_nll(x0s.detach().transpose(0, 1), x1.detach(), torch.zeros_like(tgt))
    assert float(z0) == 0.0 and float(z1) == 0.0


# ------------------------------------------------------------------ torch.library registration (last row)
def test_torch_library_ops_match_the_functional_path(dev, lib, tiny):
    """cvc::* ops (cvc/ops.py): same kernels as cvc.functional behind dispatcher-visible schemas -- forward values and gradients
    equal the functional path's, and torch.library.opcheck accepts schema / fake / autograd registration."""
    import cvc.ops  # noqa: F401
    from cvc import functional as F_
    d = tiny[0]
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    B, N, A, R, V, E = 3, 7, 16, 32, 50, 16
    # attention, additive, with masks
    q, w_a, b_a, proj, cf = r(B, A).requires_grad_(), r(1, A).requires_grad_(), r(1).requires_grad_(), r(B, N, A), r(B, N, R)
    mask = torch.zeros(B, N, dtype=torch.bool, device=dev); mask[0, -2:] = True
    fmk = torch.zeros(B, N, dtype=torch.bool, device=dev); fmk[1, :3] = True
    c1, a1, f1 = torch.ops.cvc.attn_fwd(0, q, w_a, b_a, 1.0, proj, cf, mask, fmk)
    (c1.sum() + 0.3 * f1.sum()).backward()
    g_op = [t.grad.clone() for t in (q, w_a, b_a)]
    for t in (q, w_a, b_a):
        t.grad = None
    _, ((c2, a2, f2),) = F_.attention(0, q, w_a, b_a, 1.0, [(proj, cf, mask, fmk)])
    (c2.sum() + 0.3 * f2.sum()).backward()
    close(c1, c2, rtol=1e-6, atol=1e-6); close(a1, a2, rtol=1e-6, atol=1e-7); close(f1, f2, rtol=1e-6, atol=1e-6)
    for a_, t in zip(g_op, (q, w_a, b_a)):
        close(a_, t.grad, rtol=1e-5, atol=1e-6)
    # LSTM cell against torch's formula
    x, h, c = r(B, 3 * R).requires_grad_(), r(B, R).requires_grad_(), r(B, R).requires_grad_()
    cell = torch.nn.LSTMCell(3 * R, R).to(dev)
    h1, c1_, _ = torch.ops.cvc.lstm_cell(x, h, c, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)
    (h1.sum() + 0.5 * c1_.sum()).backward()
    got = [t.grad.clone() for t in (x, h, c, cell.weight_ih, cell.weight_hh, cell.bias_ih)]
    for t in (x, h, c, *cell.parameters()):
        t.grad = None
    h2, c2_ = cell(x, (h, c))
    (h2.sum() + 0.5 * c2_.sum()).backward()
    close(h1, h2, rtol=2e-5, atol=2e-5); close(c1_, c2_, rtol=2e-5, atol=2e-5)
    for a_, t in zip(got, (x, h, c, cell.weight_ih, cell.weight_hh, cell.bias_ih)):
        close(a_, t.grad, rtol=2e-4, atol=2e-5)
    # embedding, criterion, word selection
    table, idx = r(V, E).requires_grad_(), torch.randint(0, V, (B, 4), generator=g).to(dev)
    e1 = torch.ops.cvc.embed_relu(table, idx)
    assert torch.equal(e1, torch.relu(table.detach()[idx]))
    e1.square().sum().backward()
    ref = table.detach().clone().requires_grad_()
    torch.relu(ref[idx]).square().sum().backward()
    close(table.grad, ref.grad, rtol=1e-6, atol=1e-6)
    logits, tgt, w = r(12, V).requires_grad_(), torch.randint(0, V, (12,), generator=g).to(dev), (torch.rand(12, generator=g) > 0.3).float().to(dev)
    loss, amax, _ = torch.ops.cvc.vocab_nll(logits, tgt, w)
    loss.sum().backward()
    ref = logits.detach().double().requires_grad_()
    (-(torch.log_softmax(ref, 1)[torch.arange(12), tgt.cpu()] * w.double())).sum().backward()
    close(logits.grad, ref.grad, rtol=1e-5, atol=1e-6)
    assert torch.equal(amax, logits.argmax(1))
    word, lp = torch.ops.cvc.top2_unk(logits.detach(), synth.UNK_IDX)
    assert not (word == synth.UNK_IDX).any()
    torch.library.opcheck(torch.ops.cvc.embed_relu, (table.detach().requires_grad_(), idx), test_utils=("test_schema", "test_faketensor"))
    torch.library.opcheck(torch.ops.cvc.vocab_nll, (logits.detach(), tgt, w), test_utils=("test_schema", "test_faketensor"))


def test_stable_order_and_col_sum_blocks_vs_torch():
    """the embedding backward's row grouping (== torch.argsort(stable=True), bit-exact, duplicates included) and the bias-gradient
    column sums (fp32 sums in a different order: 1e-6 relative)"""
    from cvc import hip
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    for n, hi in ((1, 5), (7, 3), (33, 2), (256, 50), (1280, 300), (1281, 9000), (7168, 17)):
        idx = torch.randint(0, hi, (n,), generator=g).to(dev)
        got = hip.stable_order(idx)
        assert torch.equal(got, torch.argsort(idx, stable=True)), n
    for S, n in ((1, 1), (5, 64), (64, 4096), (128, 100), (129, 70), (1280, 512), (1283, 1001), (5000, 130)):
        x = torch.randn(S, n + 3, generator=g).to(dev)[:, :n]          # row stride > n
        ref = x.double().sum(0)
        o1, o2 = torch.empty(n, device=dev), torch.empty(n, device=dev)
        hip.col_sum(x, o1, o2)
        assert torch.equal(o1, o2)
        tol = 1e-6 * float(x.abs().double().sum(0).max())
        assert float((o1.double() - ref).abs().max()) <= tol, (S, n)


def test_embedding_vocab_plus_1_golden(dev, lib):
    """opts.embedding_vocab_plus_1 = True (reference opts.py:197, captioner.py:53-60, 72-76): V + 1 rows in the embedding table and
    the vocabulary head.  The product's greedy decode (engine: embedding-gate table and head over V + 1 words), the cyclical pass's
    five losses and every parameter gradient against the REFERENCE's golden (tools/make_golden.py g10)."""
    from conftest import Golden
    from helpers import build_model, to_dev, model_call, tie_aware_seq_equal
    g = Golden("g10_vocab_plus_1.npz")
    d = synth.CONFIGS["tiny"]
    seed = int(g["meta.seed"])
    sd = synth.hot_path_state_dict(d, seed, vocab_plus_1=True)
    f, b = to_dev(synth.clip_features(d, seed), dev), to_dev(synth.label_glue_batch(d, seed), dev)
    for graph in (False, True):
        model = build_model(d, sd, dev, embedding_vocab_plus_1=True, hip_graph=graph)
        assert model.logit.weight.shape[0] == d.V + 1 and model.embed[0].weight.shape[0] == d.V + 1
        seq, att, _ = model_call(model, f, b, True)
        assert tie_aware_seq_equal(seq.cpu().numpy(), g["a8.seq"], g["a8.logp"]) == d.B * d.T
        close(att, g["a8.att2_weights"], **SEQ_TOL)
    out = model_call(model, f, b, False)
    for i, x in enumerate(out):
        assert float(x.detach().mean()) == pytest.approx(float(g["a9.cyc.loss%d" % i].reshape(-1)[0]), rel=2e-5, abs=2e-6)
    (0.5 * out[0].mean() + 0.5 * out[4].mean()).backward()
    gold = g.sub("a9.cyc.grad.")
    n = 0
    for name, p in model.named_parameters():
        if name.startswith("roi_feat_extractor") or name not in gold:
            continue
        want = gold[name]
        if want is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        want = torch.from_numpy(want).double()
        err = float((p.grad.cpu().double() - want).norm())
        assert err <= 5e-4 * float(want.norm()) + 1e-6, (name, err, float(want.norm()))
        n += 1
    assert n >= 15
