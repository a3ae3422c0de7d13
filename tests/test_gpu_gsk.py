"""GPU tests (pytest -m gpu) of the grouped stream-K decode schedule (csrc/gemm_gsk.hip, the SLAB form of the packed LSTM
kernel, cvc_attn_scores_qslab, cvc_top2_slab; include/cvc_hip.h "Grouped stream-K form"): every piece against fp64 / the
one-launch kernels it replaces, and the whole greedy decode against the one-launch-per-GEMM schedule.  The oracle-level parity
of the schedule is covered by tests/test_gpu_parity.py and test_gpu_fullsize.py, whose engines pick it by default.

Tolerances as in test_gpu_parity.py: 2e-5 on single ops (different K summation orders), SEQ_TOL after T recurrent steps."""
import ctypes as C
import dataclasses

import numpy as np
import pytest
import torch

from cvc import synth

pytestmark = pytest.mark.gpu_experimental      # the grouped stream-K schedule: include/cvc_hip_experimental.h

SEQ_TOL = dict(rtol=1e-4, atol=1e-4)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("the gpu-marked tests need a visible MI355X (torch.cuda.is_available() is False)")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def lib(dev):
    from cvc import hip
    hip.lib()
    return hip


def close(a, b, **tol):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, **tol)


def slab_sum(slab, plan, g, nchunk, nblk):
    """Sum a group's partial tiles in segment order on the host side of torch -> [nblk * 32 weight rows, 64 batch rows]."""
    U, u0, ms = plan["U"], plan["unit0"][g], plan["maxseg"][g]
    ntile = (nblk + 7) // 8
    s = slab.view(ntile, ms, 8, 64, 32)
    out = torch.zeros(ntile * 8, 64, 32, device=slab.device)
    for t in range(ntile):
        nseg = (u0 + (t + 1) * nchunk - 1) // U - (u0 + t * nchunk) // U + 1
        acc = s[t, 0].clone()
        for k in range(1, nseg):
            acc += s[t, k]
        out[t * 8:(t + 1) * 8] = acc
    return out[:nblk].permute(0, 2, 1).reshape(nblk * 32, 64)          # [packed row, m]


def lstm_rows(R):
    """packed row (blk, i) of an LSTM gate matrix -> checkpoint row (i >> 3) * R + blk * 8 + (i & 7)"""
    i = torch.arange(32)
    return (((i >> 3) * R + (i & 7)).view(1, 32) + (torch.arange(R // 8) * 8).view(-1, 1)).reshape(-1)


@pytest.mark.parametrize("M,R,E,V,nwg", [(64, 2048, 1024, 5000, 256), (37, 256, 96, 300, 256), (5, 64, 32, 50, 7), (64, 512, 256, 1000, 304)])
def test_stream_k_partial_tiles_and_late_kernel_vs_fp64_and_full_k(dev, lib, M, R, E, V, nwg):
    """One launch of {att-early over (h_lang, h_att) with the embedding segment skipped, vocabulary logits over h_lang}: the
    segment sums against fp64 products; then the late kernel (embedding K range + partial tiles + hoisted term, cell update)
    against cvc_packed_lstm_fwd over the whole K and against fp64; bitwise run-to-run determinism of the pair."""
    from cvc.decode import pack_weights, to_quad, from_quad
    hip, L = lib, lib.lib()
    g = torch.Generator().manual_seed(M + R)
    K = 2 * R + E
    w = (torch.randn(4 * R, K, generator=g) / K ** 0.5).to(dev)
    wo = (torch.randn(V, R, generator=g) / R ** 0.5).to(dev)
    x = torch.randn(M, K, generator=g).to(dev)
    gb = (torch.randn(M, 4 * R, generator=g) * 0.2).to(dev)
    c_prev = torch.randn(M, R, generator=g).to(dev)
    wp, wop, xq, cq = pack_weights(w, R), pack_weights(wo), to_quad(x), to_quad(c_prev)
    nblk_v = (V + 31) // 32
    plan = hip.gsk_plan([R // 64, (nblk_v + 7) // 8], [2 * R // 32, R // 32], nwg)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for rep in range(2):
        slab_a = torch.full((R // 64 * plan["maxseg"][0] * 16384,), float("nan"), device=dev)
        slab_o = torch.full(((nblk_v + 7) // 8 * plan["maxseg"][1] * 16384,), float("nan"), device=dev)
        grp = (hip.GskGroup * 2)()
        grp[0] = hip.GskGroup(wp.data_ptr(), K // 4 * 128, xq.data_ptr(), R // 8, 2 * R // 32, R // 32, E // 32, slab_a.data_ptr(), plan["maxseg"][0])
        grp[1] = hip.GskGroup(wop.data_ptr(), R // 4 * 128, xq.data_ptr(), nblk_v, R // 32, 0, 0, slab_o.data_ptr(), plan["maxseg"][1])
        assert L.cvc_gsk_gemm(grp, 2, plan["U"], st) == 0
        seg = hip.GskSegs(slab_a.data_ptr(), plan["unit0"][0], 2 * R // 32, plan["U"], plan["maxseg"][0])
        h1, h2, c2 = (torch.zeros(R // 4, 64, 4, device=dev) for _ in range(3))
        assert L.cvc_packed_lstm_late_fwd(wp.data_ptr() + (R // 4) * 128 * 4, K // 4 * 128, xq.data_ptr() + (R // 4) * 64 * 16, E, None, None,
                                          gb.data_ptr(), cq.data_ptr(), M, R, h1.data_ptr(), h2.data_ptr(), c2.data_ptr(), C.byref(seg), st) == 0
        outs.append((slab_sum(slab_a, plan, 0, 2 * R // 32, R // 8), slab_sum(slab_o, plan, 1, R // 32, nblk_v),
                     from_quad(h1, M), from_quad(h2, M), from_quad(c2, M)))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    pa, po, h, h_b, c = outs[0]
    assert torch.equal(h, h_b)
    # early partial products: K ranges [0, R) and [R + E, 2R + E) of the gate matrix; all of K for the logits
    xd, wd = x.double(), w.double()
    early = torch.cat([xd[:, :R], xd[:, R + E:]], 1) @ torch.cat([wd[:, :R], wd[:, R + E:]], 1).t()          # [M, 4R] checkpoint order
    close(pa[:, :M].t(), early[:, lstm_rows(R).to(dev)].float(), rtol=2e-5, atol=2e-5)
    close(po[:V, :M].t(), (xd[:, :R] @ wo.double().t()).float(), rtol=2e-5, atol=2e-5)
    assert float(pa[:, M:].abs().max() if M < 64 else 0.0) == 0.0                    # rows beyond M: products of zero rows
    # the late kernel finishes the cell
    pre = xd @ wd.t() + gb.double()
    i, f, gg, o = pre.chunk(4, 1)
    c_ref = torch.sigmoid(f) * c_prev.double() + torch.sigmoid(i) * torch.tanh(gg)
    h_ref = torch.sigmoid(o) * torch.tanh(c_ref)
    close(h, h_ref.float(), rtol=2e-5, atol=2e-5); close(c, c_ref.float(), rtol=2e-5, atol=2e-5)
    hf, hf2, cf = (torch.zeros(R // 4, 64, 4, device=dev) for _ in range(3))
    assert L.cvc_packed_lstm_fwd(wp.data_ptr(), xq.data_ptr(), K, None, None, gb.data_ptr(), cq.data_ptr(), M, R, hf.data_ptr(),
                                 hf2.data_ptr(), cf.data_ptr(), st) == 0
    close(h, from_quad(hf, M), rtol=2e-5, atol=2e-5); close(c, from_quad(cf, M), rtol=2e-5, atol=2e-5)
    # without early tiles (step 0): the late K range alone
    assert L.cvc_packed_lstm_late_fwd(wp.data_ptr() + (R // 4) * 128 * 4, K // 4 * 128, xq.data_ptr() + (R // 4) * 64 * 16, E, None, None,
                                      gb.data_ptr(), cq.data_ptr(), M, R, hf.data_ptr(), None, cf.data_ptr(), None, st) == 0
    pre0 = xd[:, R:R + E] @ wd[:, R:R + E].t() + gb.double()
    i, f, gg, o = pre0.chunk(4, 1)
    c0 = torch.sigmoid(f) * c_prev.double() + torch.sigmoid(i) * torch.tanh(gg)
    close(from_quad(cf, M), c0.float(), rtol=2e-5, atol=2e-5)
    close(from_quad(hf, M), (torch.sigmoid(o) * torch.tanh(c0)).float(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("M,R,V,E,unk_first", [(64, 2048, 5000, 1024, False), (33, 64, 50, 16, True), (1, 128, 513, 32, False),
                                               (64, 256, 8190, 64, False)])
def test_word_selection_from_partial_tiles(dev, lib, M, R, V, E, unk_first):
    """cvc_top2_slab over the partial tiles of a stream-K vocabulary GEMM against torch on the summed logits: word (UNK rule,
    captioner.py:415-422), log-prob, the next step's embedded word in the quad layout."""
    from cvc.decode import pack_weights, to_quad, from_quad
    hip, L = lib, lib.lib()
    g = torch.Generator().manual_seed(V + M)
    wo = (torch.randn(V, R, generator=g) / R ** 0.5).to(dev)
    bo = (torch.randn(V, generator=g) * 0.1).to(dev)
    x = torch.randn(M, R, generator=g).to(dev)
    table = torch.randn(V, E, generator=g).to(dev)
    unk = 1
    if unk_first:                                                       # make UNK the arg-max of every row
        bo[unk] = 50.0
    nblk_v = (V + 31) // 32
    plan = hip.gsk_plan([(nblk_v + 7) // 8], [R // 32], 256)
    slab = torch.full(((nblk_v + 7) // 8 * plan["maxseg"][0] * 16384,), float("nan"), device=dev)
    wop, xq = pack_weights(wo), to_quad(x)
    grp = (hip.GskGroup * 1)()
    grp[0] = hip.GskGroup(wop.data_ptr(), R // 4 * 128, xq.data_ptr(), nblk_v, R // 32, 0, 0, slab.data_ptr(), plan["maxseg"][0])
    st = torch.cuda.current_stream().cuda_stream
    assert L.cvc_gsk_gemm(grp, 1, plan["U"], st) == 0
    seg = hip.GskSegs(slab.data_ptr(), 0, R // 32, plan["U"], plan["maxseg"][0])
    word = torch.full((M,), -1, dtype=torch.int64, device=dev)
    lp = torch.zeros(M, device=dev)
    embq = torch.zeros(E // 4, 64, 4, device=dev)
    assert L.cvc_top2_slab(C.byref(seg), bo.data_ptr(), V, M, unk, word.data_ptr(), 1, lp.data_ptr(), table.data_ptr(), E,
                           embq.data_ptr(), 0, st) == 0
    logits = slab_sum(slab, plan, 0, R // 32, nblk_v)[:V, :M].t() + bo                 # the kernel's own sums, on the host side of torch
    close(logits, (x.double() @ wo.double().t() + bo.double()).float(), rtol=2e-5, atol=2e-5)
    logp = torch.log_softmax(logits.double(), 1)
    top = logp.topk(2, 1)
    want = torch.where(top.indices[:, 0] == unk, top.indices[:, 1], top.indices[:, 0])
    assert torch.equal(word, want)
    close(lp, logp.gather(1, want.view(-1, 1)).view(-1).float(), rtol=1e-5, atol=1e-5)
    assert torch.equal(from_quad(embq, M), torch.relu(table[want]))
    if unk_first:
        assert not (word == unk).any()


@pytest.mark.parametrize("kind", ["additive", "dot"])
def test_scores_with_the_query_from_partial_tiles(dev, lib, kind):
    """cvc_attn_scores_qslab (query = segment sums of a stream-K h2attn group + bias) against cvc_attn_scores on the summed
    query."""
    from cvc.decode import pack_weights, to_quad
    hip, L = lib, lib.lib()
    B, N, F, R, A = 37, 20, 33, 256, 128
    g = torch.Generator().manual_seed(5)
    wh = (torch.randn(A, R, generator=g) / R ** 0.5).to(dev)
    bh = (torch.randn(A, generator=g) * 0.1).to(dev)
    h = torch.randn(B, R, generator=g).to(dev)
    pr, pf = torch.randn(B, N, A, generator=g).to(dev), torch.randn(B, F, A, generator=g).to(dev)
    w_a, b_a = torch.randn(A, generator=g).to(dev), torch.randn(1, generator=g).to(dev)
    mask = (torch.rand(B, N, generator=g) < 0.2).to(torch.uint8).to(dev)
    plan = hip.gsk_plan([1], [R // 32], 256)
    slab = torch.full((plan["maxseg"][0] * 16384,), float("nan"), device=dev)
    whp, hq = pack_weights(wh), to_quad(h)
    grp = (hip.GskGroup * 1)()
    grp[0] = hip.GskGroup(whp.data_ptr(), R // 4 * 128, hq.data_ptr(), A // 32, R // 32, 0, 0, slab.data_ptr(), plan["maxseg"][0])
    st = torch.cuda.current_stream().cuda_stream
    assert L.cvc_gsk_gemm(grp, 1, plan["U"], st) == 0
    seg = hip.GskSegs(slab.data_ptr(), 0, R // 32, plan["U"], plan["maxseg"][0])
    k = hip.ATTN_ADDITIVE if kind == "additive" else hip.ATTN_DOT

    def sets():
        sr, sf = torch.zeros(B, N, device=dev), torch.zeros(B, F, device=dev)
        arr = (hip.AttnSet * 2)()
        arr[0] = hip.AttnSet(pr.data_ptr(), pr.data_ptr(), mask.data_ptr(), None, sr.data_ptr(), None, sr.data_ptr(), None, N, 0)
        arr[1] = hip.AttnSet(pf.data_ptr(), pf.data_ptr(), None, None, sf.data_ptr(), None, sf.data_ptr(), None, F, 0)
        return arr, sr, sf
    a1, sr1, sf1 = sets()
    assert L.cvc_attn_scores_qslab(k, C.byref(seg), bh.data_ptr(), w_a.data_ptr(), b_a.data_ptr(), 0.7, a1, 2, B, 1, A, st) == 0
    q = slab_sum(slab, plan, 0, R // 32, A // 32)[:, :B].t().contiguous() + bh
    close(q, (h.double() @ wh.double().t() + bh.double()).float(), rtol=2e-5, atol=2e-5)
    a2, sr2, sf2 = sets()
    assert L.cvc_attn_scores(k, q.data_ptr(), w_a.data_ptr(), b_a.data_ptr(), 0.7, a2, 2, B, 1, A, st) == 0
    assert torch.equal(sr1, sr2) and torch.equal(sf1, sf2)


@pytest.mark.parametrize("B,dims", [(64, dict(N=20, F=12, R=256, A=64, E=64, V=300, T=6)),
                                    (5, dict(N=7, F=5, R=64, A=32, E=32, V=50, T=4)),
                                    (33, dict(N=100, F=48, R=512, A=256, E=256, V=1000, T=5))])
def test_stream_k_schedule_equals_one_launch_per_gemm_schedule(dev, lib, B, dims):
    """Greedy decode, grouped stream-K schedule against the one-launch-per-GEMM schedule (same products, other K summation
    orders): same words up to near-ties, attention within the recurrent tolerance; C driver == Python launch list bit for bit;
    eager == HIP-graph replay, twice."""
    from helpers import to_dev
    from cvc.decode import DecodeEngine, DecodeWeights
    d = dataclasses.replace(synth.CONFIGS["tiny"], B=B, **dims)
    sd, f_np = synth.hot_path_state_dict(d, 77 + B), synth.clip_features(d, 77 + B)
    W, f = DecodeWeights(to_dev(sd, dev)), to_dev(f_np, dev)
    e_ref = DecodeEngine(W, f, d.T, synth.UNK_IDX, gsk=False, embgate=False)
    e = DecodeEngine(W, f, d.T, synth.UNK_IDX, gsk=True)
    e_py = DecodeEngine(W, f, d.T, synth.UNK_IDX, gsk=True, driver=False)
    assert e.gsk and e_py.gsk and not e_ref.gsk and not e_ref.embgate and e._plan is not None and e_py._plan is None
    ref = [x.clone() for x in e_ref.run()]
    a = [x.clone() for x in e.run()]
    b = [x.clone() for x in e_py.run()]
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    same = (a[0] == ref[0]).all(1)
    assert int(same.sum()) >= B - max(1, B // 32)
    close(a[1][same], ref[1][same], **SEQ_TOL)
    close(e.logprob[:, same], e_ref.logprob[:, same], **SEQ_TOL)
    assert lib.lib().cvc_decode_num_launches(e._plan) == 6 + 7 * d.T
    e.capture()
    for _ in range(2):
        assert all(torch.equal(x, y) for x, y in zip(a, e.run()))


def test_stream_k_schedule_cfg2_vs_one_launch_schedule_and_determinism(dev, lib):
    """BASELINE config 2 size: both schedules from the same weights and clips -- words equal except near-ties, attention
    within the recurrent tolerance, replays bitwise identical."""
    from helpers import to_dev
    from cvc.decode import DecodeEngine, DecodeWeights
    d = synth.CONFIGS["cfg2"]
    sd, f_np = synth.hot_path_state_dict(d, 4321), synth.clip_features(d, 4321)
    W, f = DecodeWeights(to_dev(sd, dev)), to_dev(f_np, dev)
    ref = [x.clone() for x in DecodeEngine(W, f, d.T, synth.UNK_IDX, gsk=False, embgate=False).run()]
    e = DecodeEngine(W, f, d.T, synth.UNK_IDX, gsk=True).capture()
    assert e.gsk
    a = [x.clone() for x in e.run()]
    same = (a[0] == ref[0]).all(1)
    assert int(same.sum()) >= d.B - 2
    close(a[1][same], ref[1][same], **SEQ_TOL)
    for _ in range(3):
        assert all(torch.equal(x, y) for x, y in zip(a, e.run()))


def test_stream_k_entry_points_reject_bad_arguments(dev, lib):
    hip, L = lib, lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    buf = torch.zeros(1 << 16, device=dev)
    grp = (hip.GskGroup * 1)()
    grp[0] = hip.GskGroup(buf.data_ptr(), 8 * 128, buf.data_ptr(), 1, 1, 0, 0, buf.data_ptr(), 1)
    assert L.cvc_gsk_gemm(grp, 1, 0, st) == -1                    # U < 1
    assert L.cvc_gsk_gemm(grp, 4, 1, st) == -1                    # too many groups
    grp[0].nchunk = 5
    assert L.cvc_gsk_gemm(grp, 1, 2, st) == -1                    # 3 segments for a slab of 1
    seg = hip.GskSegs(None, 0, 4, 2, 1)
    assert L.cvc_top2_slab(C.byref(seg), None, 50, 4, 1, buf.data_ptr(), 1, None, None, 0, None, 0, st) == -1
    assert L.cvc_packed_lstm_late_fwd(buf.data_ptr(), 4, buf.data_ptr(), 32, None, None, None, buf.data_ptr(), 4, 64, None, None,
                                      buf.data_ptr(), None, st) == -1                                  # block stride < K range


# ------------------------------------------------------------------ embedding-gate schedule
@pytest.mark.parametrize("M,R,E,V", [(64, 2048, 1024, 5000), (37, 256, 96, 300), (1, 64, 32, 50)])
def test_embedding_gate_table_form_of_the_att_lstm(dev, lib, M, R, E, V):
    """cvc_packed_lstm_embgate_fwd (K = 2R GEMM + a row of the per-checkpoint table W_ih[:, emb] x relu(Emb[v])) against
    cvc_packed_lstm_fwd over [h_lang | relu(Emb[word]) | h_att] and against fp64; the table itself against fp64."""
    from cvc.decode import pack_weights, to_quad, from_quad
    hip, L = lib, lib.lib()
    g = torch.Generator().manual_seed(M + R + 1)
    K = 2 * R + E
    w = (torch.randn(4 * R, K, generator=g) / K ** 0.5).to(dev)
    emb = torch.randn(V, E, generator=g).to(dev)
    word = torch.randint(0, V, (M,), generator=g).to(dev)
    hl, ha = torch.randn(M, R, generator=g).to(dev), torch.randn(M, R, generator=g).to(dev)
    gb = (torch.randn(M, 4 * R, generator=g) * 0.2).to(dev)
    c_prev = torch.randn(M, R, generator=g).to(dev)
    table = hip.tile_mm(torch.relu(emb), w[:, R:R + E])                                          # [V, 4R], checkpoint gate order
    t_ref = torch.relu(emb).double() @ w[:, R:R + E].double().t()
    close(table, t_ref.float(), rtol=2e-5, atol=2e-5)
    wp2 = pack_weights(torch.cat([w[:, :R], w[:, R + E:]], 1), R)
    xq2, cq = to_quad(torch.cat([hl, ha], 1)), to_quad(c_prev)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for _ in range(2):
        h1, h2, c2 = (torch.zeros(R // 4, 64, 4, device=dev) for _ in range(3))
        assert L.cvc_packed_lstm_embgate_fwd(wp2.data_ptr(), xq2.data_ptr(), 2 * R, None, None, gb.data_ptr(), table.data_ptr(), word.data_ptr(),
                                             cq.data_ptr(), M, R, h1.data_ptr(), h2.data_ptr(), c2.data_ptr(), st) == 0
        outs.append((from_quad(h1, M), from_quad(h2, M), from_quad(c2, M)))
    assert all(torch.equal(a, b) for a, b in zip(*outs)) and torch.equal(outs[0][0], outs[0][1])
    h, _, c = outs[0]
    x = torch.cat([hl, torch.relu(emb[word]), ha], 1)
    pre = x.double() @ w.double().t() + gb.double()
    i, f, gg, o = pre.chunk(4, 1)
    c_ref = torch.sigmoid(f) * c_prev.double() + torch.sigmoid(i) * torch.tanh(gg)
    close(c, c_ref.float(), rtol=2e-5, atol=2e-5)
    close(h, (torch.sigmoid(o) * torch.tanh(c_ref)).float(), rtol=2e-5, atol=2e-5)
    hf, hf2, cf = (torch.zeros(R // 4, 64, 4, device=dev) for _ in range(3))
    wp, xq = pack_weights(w, R), to_quad(x)
    assert L.cvc_packed_lstm_fwd(wp.data_ptr(), xq.data_ptr(), K, None, None, gb.data_ptr(), cq.data_ptr(), M, R, hf.data_ptr(),
                                 hf2.data_ptr(), cf.data_ptr(), st) == 0
    close(h, from_quad(hf, M), rtol=2e-5, atol=2e-5); close(c, from_quad(cf, M), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("M,R,V,unk_first", [(64, 2048, 5000, False), (33, 64, 50, True), (1, 128, 513, False), (64, 256, 8190, False)])
def test_vocabulary_projection_selects_the_word_itself(dev, lib, M, R, V, unk_first):
    """cvc_packed_linear_select_fwd (top-2 records merged by the last workgroup to arrive) against torch on fp64 logits and
    against cvc_packed_linear_fwd + cvc_top2_final; the arrival counter is back at zero; repeated launches agree bit for bit."""
    from cvc.decode import pack_weights, to_quad
    hip, L = lib, lib.lib()
    g = torch.Generator().manual_seed(V + M + 3)
    wo = (torch.randn(V, R, generator=g) / R ** 0.5).to(dev)
    bo = (torch.randn(V, generator=g) * 0.1).to(dev)
    x = torch.randn(M, R, generator=g).to(dev)
    unk = 1
    if unk_first:
        bo[unk] = 50.0
    wop, xq = pack_weights(wo), to_quad(x)
    nblk = (V + 31) // 32
    st = torch.cuda.current_stream().cuda_stream
    counter = torch.zeros(4, dtype=torch.int32, device=dev)
    got = []
    for _ in range(3):
        part = torch.full((nblk, 64, 6), float("nan"), device=dev)
        word = torch.full((M,), -1, dtype=torch.int64, device=dev)
        lp = torch.zeros(M, device=dev)
        assert L.cvc_packed_linear_select_fwd(wop.data_ptr(), xq.data_ptr(), R, bo.data_ptr(), M, V, part.data_ptr(), counter.data_ptr(), unk,
                                              word.data_ptr(), 1, lp.data_ptr(), st) == 0
        got.append((word, lp))
        assert int(counter[0]) == 0
    assert all(torch.equal(got[0][0], w) and torch.equal(got[0][1], l) for w, l in got[1:])
    word, lp = got[0]
    logp = torch.log_softmax(x.double() @ wo.double().t() + bo.double(), 1)
    top = logp.topk(3, 1)
    want = torch.where(top.indices[:, 0] == unk, top.indices[:, 1], top.indices[:, 0])
    clear = (top.values[:, 0] - top.values[:, 1] > 1e-5) & (top.values[:, 1] - top.values[:, 2] > 1e-5)
    assert torch.equal(word[clear], want[clear]) and int(clear.sum()) >= M - 2
    close(lp[clear], logp.gather(1, want.view(-1, 1)).view(-1)[clear].float(), rtol=1e-5, atol=2e-5)
    if unk_first:
        assert not (word == unk).any()
    # the two-launch form on the same operands
    part2 = torch.zeros(nblk, 64, 6, device=dev)
    word2, lp2 = torch.zeros(M, dtype=torch.int64, device=dev), torch.zeros(M, device=dev)
    assert L.cvc_packed_linear_fwd(wop.data_ptr(), xq.data_ptr(), R, bo.data_ptr(), M, V, 1, None, V, part2.data_ptr(), st) == 0
    assert L.cvc_top2_final(part2.data_ptr(), nblk, M, unk, word2.data_ptr(), 1, lp2.data_ptr(), None, 0, None, 0, st) == 0
    assert torch.equal(word, word2)
    close(lp, lp2, rtol=1e-6, atol=2e-6)


@pytest.mark.parametrize("B,dims", [(64, dict(N=20, F=12, R=256, A=64, E=64, V=300, T=6)),
                                    (5, dict(N=7, F=5, R=32, A=32, E=32, V=50, T=4)),
                                    (33, dict(N=100, F=48, R=512, A=256, E=256, V=1000, T=5))])
def test_embedding_gate_schedule_equals_one_launch_per_gemm_schedule(dev, lib, B, dims):
    """Greedy decode, embedding-gate schedule (default) against the 7-launch schedule: same words up to near-ties, attention and
    log-probs within the recurrent tolerance; C driver == Python launch list bit for bit; eager == HIP-graph replay, twice."""
    from helpers import to_dev
    from cvc.decode import DecodeEngine, DecodeWeights
    d = dataclasses.replace(synth.CONFIGS["tiny"], B=B, **dims)
    sd, f_np = synth.hot_path_state_dict(d, 78 + B), synth.clip_features(d, 78 + B)
    W, f = DecodeWeights(to_dev(sd, dev)), to_dev(f_np, dev)
    e_ref = DecodeEngine(W, f, d.T, synth.UNK_IDX, embgate=False)
    e = DecodeEngine(W, f, d.T, synth.UNK_IDX)
    e_py = DecodeEngine(W, f, d.T, synth.UNK_IDX, driver=False)
    assert e.embgate and e_py.embgate and not e_ref.embgate and not e.gsk and e._plan is not None and e_py._plan is None
    ref = [x.clone() for x in e_ref.run()]
    a = [x.clone() for x in e.run()]
    b = [x.clone() for x in e_py.run()]
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    same = (a[0] == ref[0]).all(1)
    assert int(same.sum()) >= B - max(1, B // 32)
    close(a[1][same], ref[1][same], **SEQ_TOL)
    close(e.logprob[:, same], e_ref.logprob[:, same], **SEQ_TOL)
    assert lib.lib().cvc_decode_num_launches(e._plan) == 6 + 7 * d.T
    e.capture()
    for _ in range(2):
        assert all(torch.equal(x, y) for x, y in zip(a, e.run()))


@pytest.mark.parametrize("B,beam,dims", [(20, 5, dict(N=30, F=12, R=256, A=64, E=48, V=300, T=5)), (70, 1, dict(N=9, F=4, R=64, A=32, E=32, V=97, T=3)),
                                         (3, 3, dict(N=7, F=5, R=32, A=16, E=16, V=50, T=4))])
def test_tile_path_embedding_gate_form_equals_full_k_form(dev, lib, B, beam, dims):
    """Beam search / more than 64 rows (tile path): the embedding-gate form (gate GEMM over K = 2R, the word's share from the
    table in cvc_tile_lstm_finish_embgate) against the form with the embedding segment in the GEMM: same hypotheses up to
    near-ties, scores / attention within the recurrent tolerance; C driver == Python launch list bit for bit."""
    from helpers import to_dev
    from cvc.decode import DecodeEngine, DecodeWeights
    d = dataclasses.replace(synth.CONFIGS["tiny"], B=B, **dims)
    sd, f_np = synth.hot_path_state_dict(d, 31 + B), synth.clip_features(d, 31 + B)
    W, f = DecodeWeights(to_dev(sd, dev)), to_dev(f_np, dev)
    e_ref = DecodeEngine(W, f, d.T, synth.UNK_IDX, beam=beam, embgate=False)
    e = DecodeEngine(W, f, d.T, synth.UNK_IDX, beam=beam)
    e_py = DecodeEngine(W, f, d.T, synth.UNK_IDX, beam=beam, driver=False)
    assert e.tile and e.embgate and e_py.embgate and not e_ref.embgate
    ref = [x.clone() for x in e_ref.run()]
    a = [x.clone() for x in e.run()]
    b = [x.clone() for x in e_py.run()]
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    same = (a[0] == ref[0]).all(1)
    assert int(same.sum()) >= B - max(1, B // 16)
    close(a[1][same], ref[1][same], **SEQ_TOL)
    if beam > 1:
        close(a[2][same], ref[2][same], rtol=2e-4, atol=2e-4)
    e.capture()
    for _ in range(2):
        assert all(torch.equal(x, y) for x, y in zip(a, e.run()))


def test_language_cell_on_the_exchange_finish_kernel_equals_default_schedule(dev, lib):
    """DecodeEngine(lang_ksx=True) at R = 2048 (the K-split gate GEMM with the in-launch exchange for the language cell) against
    the default schedule: same words up to near-ties, attention and log-probs within the recurrent tolerance (another K summation
    order); C driver == Python launch list bit for bit; eager == HIP-graph replay; the exchange's error word stays clear."""
    from helpers import to_dev
    from cvc.decode import DecodeEngine, DecodeWeights
    B = 64
    d = dataclasses.replace(synth.CONFIGS["tiny"], B=B, N=10, F=6, R=2048, A=64, E=64, V=200, T=3)
    sd, f_np = synth.hot_path_state_dict(d, 5), synth.clip_features(d, 5)
    W, f = DecodeWeights(to_dev(sd, dev)), to_dev(f_np, dev)
    e_ref = DecodeEngine(W, f, d.T, synth.UNK_IDX, lang_ksx=False)
    e = DecodeEngine(W, f, d.T, synth.UNK_IDX, lang_ksx=True)
    e_py = DecodeEngine(W, f, d.T, synth.UNK_IDX, lang_ksx=True, driver=False)
    assert e.lang_ksx and e_py.lang_ksx and not e_ref.lang_ksx and e._plan is not None and e_py._plan is None
    ref = [x.clone() for x in e_ref.run()]
    a = [x.clone() for x in e.run()]
    b = [x.clone() for x in e_py.run()]
    assert e.lang_ksx and e_py.lang_ksx, "the exchange reported a failed wait and the engine fell back"
    assert int(e.ksx_flags[-1]) == 0 and int(e_py.ksx_flags[-1]) == 0
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    same = (a[0] == ref[0]).all(1)
    assert int(same.sum()) >= B - 2
    close(a[1][same], ref[1][same], **SEQ_TOL)
    close(e.logprob[:, same], e_ref.logprob[:, same], **SEQ_TOL)
    e.capture()
    for _ in range(2):
        assert all(torch.equal(x, y) for x, y in zip(a, e.run()))
    assert int(e.ksx_flags[-1]) == 0
