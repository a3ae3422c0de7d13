"""Once-per-clip encoder mirror (cvc/model/backbone.py) against the reference's own outputs
(tests/golden/g5_encoder.npz, made by tools/make_golden.py g5 from
/root/reference/anet-video-captioning/model/backbone.py).  The encoder is library work (torch
GEMMs + GRU), so the stand-alone checks run on CPU as well; the end-to-end check drives the HIP hot
path behind it and is GPU-only."""
import pickle

import numpy as np
import pytest
import torch

from cvc import synth
from conftest import Golden
from helpers import make_opts, to_dev

D = synth.CONFIGS["tiny"]
OUT = ("fc_feats", "conv_feats", "p_conv_feats", "pool_feats", "p_pool_feats", "g_pool_feats", "pnt_mask",
       "overlaps_expanded")


@pytest.fixture(scope="module")
def g5():
    return Golden("g5_encoder.npz")


def encoder_opts(tmp_path, tables, seq_per_img=1, **over):
    wdir = tmp_path / "detectron_weights"
    wdir.mkdir(exist_ok=True)
    for k in ("fc7_w", "fc7_b", "cls_score_w", "cls_score_b"):
        with open(wdir / (k + ".pkl"), "wb") as f:
            pickle.dump(tables[k], f)
    base = dict(test_mode=False)
    base.update(over)
    return make_opts(D, seq_per_img=seq_per_img, enable_BUTD=False, att_input_mode="both",
                     num_sampled_frm=4, finetune_cnn=False, att_feat_size=D.G, fc_feat_size=synth.SEG_FEAT_DIM,
                     t_attn_size=D.F, second_drop_prob=0.3, att_model="topdown", t_attn_mode="bigru",
                     itod={i + 1: "d%d" % i for i in range(D.DET)},
                     vg_cls=["vg%d" % i for i in range(tables["glove_vg_cls"].shape[0])],
                     glove_clss=torch.from_numpy(tables["glove_clss"]),
                     glove_vg_cls=torch.from_numpy(tables["glove_vg_cls"]),
                     detectron_weights_dir=str(wdir), **base)


def build_encoder(tmp_path, seed, seq_per_img=1, **over):
    from cvc.model.backbone import RegionalFeatureExtractorGVD
    enc = RegionalFeatureExtractorGVD(encoder_opts(tmp_path, synth.detectron_tables(D, seed), seq_per_img, **over))
    ctor = {k: v.detach().clone() for k, v in enc.state_dict().items() if k in synth.ENCODER_CTOR_KEYS}
    sd = {k: (ctor[k] if k in ctor else torch.from_numpy(np.asarray(synth.encoder_fill(k, v.shape, seed))))
          for k, v in enc.state_dict().items()}
    enc.load_state_dict(sd)
    return enc.eval(), ctor


def run_encoder(enc, inp, overlaps):
    return enc(inp["segs_feat"], inp["proposals"], inp["num"], inp["box_mask"], inp["region_feats"], inp["gt_bboxs"],
               overlaps, inp["sample_idx"])


def probe_loss(outs):
    fc, conv, pconv, pool, ppool, g = outs[:6]
    return (0.01 * fc.sum() + conv.pow(2).mean() + pconv.mean() + pool.pow(2).mean() + ppool.pow(2).mean()
            + g.pow(2).mean() + outs[9].sum())


def test_constructor_matches_reference(g5, tmp_path):
    seed = int(g5["meta.seed"])
    _, ctor = build_encoder(tmp_path, seed)
    want = g5.sub("s2.ctor.")
    assert set(want) == set(ctor)
    for k, v in want.items():
        np.testing.assert_array_equal(ctor[k].numpy(), v, err_msg=k)


def test_missing_detectron_pickles_fail_loudly(tmp_path):
    from cvc.model.backbone import RegionalFeatureExtractorGVD
    o = encoder_opts(tmp_path, synth.detectron_tables(D, 1))
    o.detectron_weights_dir = str(tmp_path / "nowhere")
    with pytest.raises(FileNotFoundError):
        RegionalFeatureExtractorGVD(o)


@pytest.mark.parametrize("case,S,over", [("s2.", 2, {}), ("s1test.", 1, {"test_mode": True})])
def test_forward_matches_reference(g5, tmp_path, case, S, over):
    seed = int(g5["meta.seed"])
    enc, _ = build_encoder(tmp_path, seed, S, collect_cls_pred=True, **over)
    inp = to_dev(synth.encoder_inputs(D, seed), "cpu")
    res = run_encoder(enc, inp, torch.from_numpy(g5["overlaps"]))
    want = g5.sub(case + "out.")
    for k, got in zip(OUT, res[:8]):
        if got.dtype == torch.bool:
            np.testing.assert_array_equal(got.numpy(), want[k], err_msg=k)
        else:
            np.testing.assert_allclose(got.detach().numpy(), want[k], rtol=1e-5, atol=2e-6, err_msg=k)
    np.testing.assert_allclose(res[9].detach().numpy().reshape(-1), want["cls_loss"].reshape(-1), rtol=1e-5, atol=1e-6)
    if not over:
        np.testing.assert_array_equal(res[8].numpy(), want["cls_pred"])


def test_gradients_match_reference(g5, tmp_path):
    seed = int(g5["meta.seed"])
    enc, _ = build_encoder(tmp_path, seed, 2)
    inp = to_dev(synth.encoder_inputs(D, seed), "cpu")
    probe_loss(run_encoder(enc, inp, torch.from_numpy(g5["overlaps"]))).backward()
    want = g5.sub("s2.grad.")
    params = dict(enc.named_parameters())
    assert set(k[:-len(".norm")] if k.endswith(".norm") else k for k in want) == set(params)
    for k, v in want.items():
        if v is None:
            assert params[k].grad is None, k
        else:
            got = params[k[:-len(".norm")]].grad.double().norm().item()
            assert got == pytest.approx(float(v), rel=1e-4, abs=1e-7), k


def test_no_positive_targets_gives_zero_loss(tmp_path):
    enc, _ = build_encoder(tmp_path, 5)
    inp = to_dev(synth.encoder_inputs(D, 5), "cpu")
    res = run_encoder(enc, inp, torch.zeros(D.B, D.N, D.K))
    assert float(res[9].detach()) == 0.0
    assert torch.isfinite(res[3]).all()


@pytest.mark.gpu
def test_encoder_plus_hot_path_end_to_end(g5, tmp_path):
    """reference captioner over the reference encoder == HIP hot path over the mirrored encoder."""
    from cvc.model.captioner import DecodeAndGroundCaptionerGVDROI
    dev = torch.device("cuda:0")
    seed = int(g5["meta.seed"])
    enc, _ = build_encoder(tmp_path, seed, 1)
    model = DecodeAndGroundCaptionerGVDROI(make_opts(D), roi_extractor=enc)
    sd = {k: torch.from_numpy(v) for k, v in synth.hot_path_state_dict(D, seed).items()
          if not k.startswith("roi_feat_extractor.")}
    sd.update({"roi_feat_extractor." + k: v for k, v in enc.state_dict().items()})
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    b = to_dev(synth.encoder_inputs(D, seed), dev)

    def call(lang_eval):
        return model(b["segs_feat"], b["input_seq"], b["gt_seq"], b["num"], b["proposals"], b["gt_bboxs"], b["box_mask"],
                     b["region_feats"], b["frm_mask"], b["sample_idx"], b["pnt_mask_in"], lang_eval)
    with torch.no_grad():
        seq, att2, _ = call(True)
    np.testing.assert_array_equal(seq.cpu().numpy(), g5["e2e.seq"])
    np.testing.assert_allclose(att2.cpu().numpy(), g5["e2e.att2_weights"], rtol=1e-4, atol=2e-5)
    # MIOpen's RNN backward needs the GRU in training mode; switch its inter-layer dropout off to stay deterministic
    model.roi_feat_extractor.context_enc.train()
    model.roi_feat_extractor.context_enc.dropout = 0.0
    losses = call(False)
    for i, l in enumerate(losses):
        assert float(l.detach().mean()) == pytest.approx(float(g5["e2e.loss%d" % i][0]), rel=2e-5, abs=2e-6), i
    lm, a2, _g, cls, rec = [x.mean() for x in losses]
    (0.5 * lm + 0.05 * a2 + 0.1 * cls + 0.5 * rec).backward()
    params = dict(model.named_parameters())
    for k, v in g5.sub("e2e.grad.").items():
        if v is None:
            p = params[k]
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            got = params[k[:-len(".norm")]].grad.double().norm().item()
            assert got == pytest.approx(float(v), rel=2e-4, abs=1e-6), k


# ------------------------------------------------------------------ frame-context GRU (cvc/gru.py, cvc_gru_seq_fwd)
def _gru(inp, H, layers, bidir, seed):
    torch.manual_seed(seed)
    g = torch.nn.GRU(inp, H, layers, dropout=0.2 if layers > 1 else 0.0, bidirectional=bidir, batch_first=True).eval()
    with torch.no_grad():
        for p in g.parameters():
            p.mul_(1.5)                      # wider gate pre-activations than the default init gives
    return g


@pytest.mark.parametrize("B,F,inp,H,layers,bidir", [(3, 5, 32, 16, 2, True), (5, 9, 24, 40, 1, False), (4, 6, 64, 32, 3, True)])
def test_oracle_gru_restatement_matches_the_library_module(B, F, inp, H, layers, bidir):
    """oracle.gru_sequence (the step-by-step restatement the GPU kernel is checked against) == nn.GRU, the module the
    reference calls (backbone.py:103-106, 335-338)."""
    from oracle import ref_cpu as O
    g = _gru(inp, H, layers, bidir, 3)
    x = torch.randn(B, F, inp)
    with torch.no_grad():
        want = g(x)[0]
        got = O.gru_sequence(x, dict(g.named_parameters()), layers, bidir)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=1e-6)


def test_gru_weight_pack_layout():
    from cvc.gru import pack_gru_weights
    H = 40
    w = torch.arange(3 * H * H, dtype=torch.float32).view(3 * H, H)
    p = pack_gru_weights(w, H)                                   # [H/8][Kp/4][32][4]
    assert p.shape == (H // 8, 64 // 4, 32, 4)
    for blk, q, i, e in ((0, 0, 0, 0), (2, 3, 9, 1), (4, 9, 23, 3), (1, 5, 17, 2)):
        gate, unit, k = i >> 3, blk * 8 + (i & 7), 4 * q + e
        assert float(p[blk, q, i, e]) == float(w[gate * H + unit, k])
    assert float(p[:, :, 24:].abs().max()) == 0 and float(p[:, H // 4:].abs().max()) == 0      # zero fourth gate, zero K padding


@pytest.mark.gpu
@pytest.mark.parametrize("B,F,inp,H,layers,bidir", [(3, 5, 32, 16, 2, True), (64, 20, 256, 128, 2, True), (70, 7, 48, 40, 1, False),
                                                    (33, 11, 100, 72, 2, True), (20, 33, 64, 384, 2, True), (64, 17, 96, 256, 1, False)])
def test_gru_hip_vs_oracle(B, F, inp, H, layers, bidir):
    """cvc.gru.gru_forward (tile GEMM input projections + the recurrence in its persistent and its per-step form) against the
    CPU oracle and against each other; more than 64 clips run in chunks;
    bitwise run-to-run determinism."""
    from oracle import ref_cpu as O
    from cvc import gru as G
    g = _gru(inp, H, layers, bidir, 5)
    x = torch.randn(B, F, inp)
    with torch.no_grad():
        want = O.gru_sequence(x, dict(g.named_parameters()), layers, bidir)
        gd = g.to("cuda:0")
        assert G.supported(gd, x.cuda())
        got = G.gru_forward(gd, x.cuda())
        assert G.last_form == ("persistent" if H % 128 == 0 else "steps")
        again = G.gru_forward(gd, x.cuda())
        G.PERSISTENT = False
        try:
            steps = G.gru_forward(gd, x.cuda())
            assert G.last_form == "steps"
        finally:
            G.PERSISTENT = True
    assert torch.equal(got, again)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(steps.cpu().numpy(), want.numpy(), rtol=2e-5, atol=2e-5)
    if H % 256 == 0:                   # 4 waves per workgroup instead of 8: K split four ways instead of eight (fp32 summation order)
        from cvc import hip
        prev = hip.lib().cvc_gru_persistent_waves8(0)
        try:
            with torch.no_grad():
                w4 = G.gru_forward(gd, x.cuda())
            assert G.last_form == "persistent"
        finally:
            hip.lib().cvc_gru_persistent_waves8(prev)
        np.testing.assert_allclose(w4.cpu().numpy(), want.numpy(), rtol=2e-5, atol=2e-5)
    from cvc import hip as _hip
    if H % 128 == 0 and B > 32 and _hip.experimental_built():     # the interleaved-halves form (cvc_hip_experimental.h): same bits
        from cvc import hip
        prev = hip.lib().cvc_gru_persistent_halves(1)
        try:
            with torch.no_grad():
                halves = G.gru_forward(gd, x.cuda())
            assert G.last_form == "persistent" and torch.equal(halves, got)
        finally:
            hip.lib().cvc_gru_persistent_halves(prev)
    # (the two forms split K over 4 and 8 waves: same products, different fp32 summation order)
    np.testing.assert_allclose(got.cpu().numpy(), steps.cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_gru_hip_full_size_vs_library_cpu():
    """Config-2 encoder size (B=64 clips, F=480 frames, R=2048 -> H=1024, 2 layers, bidirectional): HIP path against nn.GRU
    on the host CPU (the module the reference calls)."""
    from cvc import gru as G
    g = _gru(2048, 1024, 2, True, 7)
    x = torch.randn(64, 480, 2048)
    with torch.no_grad():
        want = g(x)[0]
        got = G.gru_forward(g.to("cuda:0"), x.cuda()).cpu()
    err = (got - want).abs().max().item()
    assert err < 5e-5, err
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-4, atol=5e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,K,N,block", [(2000, 300, 200, False), (1500, 2780, 256, True), (4096, 1024, 512, True)])
def test_dense_layers_on_the_tile_gemm(rows, K, N, block):
    """cvc.dense.apply (inference-time nn.Linear / Linear-ReLU-Dropout blocks of the encoder on the tile GEMM) against the module
    in fp64: an error no worse than the library's fp32 result; grad mode and small inputs keep the module itself."""
    from cvc import dense
    torch.manual_seed(rows + K)
    lin = torch.nn.Linear(K, N)
    layer = (torch.nn.Sequential(lin, torch.nn.ReLU(), torch.nn.Dropout(0.5)) if block else lin).cuda().eval()
    x = torch.randn(3, rows // 3 + 1, K, device="cuda")
    with torch.no_grad():
        assert dense.usable(x)
        got = dense.apply(layer, x)
        lib32 = layer(x)
        ref = layer.double()(x.double())
        layer.float()
    e_got = float((got.double() - ref).norm() / ref.norm())
    e_lib = float((lib32.double() - ref).norm() / ref.norm())
    assert got.shape == lib32.shape and e_got <= max(2.0 * e_lib, 3e-7), (e_got, e_lib)
    assert not dense.usable(x[:, :4])                        # few rows: library kernel
    with torch.enable_grad():
        y = dense.apply(layer, x.requires_grad_(True))       # autograd: the module itself
        assert y.requires_grad


@pytest.mark.gpu
@pytest.mark.parametrize("B,F,inp,H,layers,bidir", [(5, 7, 48, 128, 2, True), (64, 12, 96, 256, 1, False), (70, 5, 64, 128, 2, True),
                                                    (37, 9, 80, 256, 2, True), (64, 6, 64, 512, 1, True),
                                                    # widths outside the persistent forms: the per-step training forward + backward
                                                    (6, 5, 24, 16, 2, True), (33, 4, 40, 200, 1, True), (16, 3, 64, 2048, 1, True)])
def test_gru_hip_autograd_vs_library_cpu(B, F, inp, H, layers, bidir):
    """cvc.gru.gru_forward_train (the recurrence keeping the gates -- persistent for H % 128 == 0, H <= 1024, else the per-step
    training form, e.g. config 5's encoder width H = 2048 -- + the backward recurrence + dense dW / dX products on the tile GEMM)
    against torch autograd of nn.GRU on the CPU: output, input gradient and every parameter gradient."""
    from cvc import gru as G
    g = _gru(inp, H, layers, bidir, 11)
    x = torch.randn(B, F, inp)
    probe = torch.randn(B, F, (2 if bidir else 1) * H)
    xc = x.clone().requires_grad_(True)
    (g(xc)[0] * probe).sum().backward()
    want = {k: p.grad.clone() for k, p in g.named_parameters()}
    want_dx = xc.grad.clone()
    for p in g.parameters():
        p.grad = None
    gd = g.to("cuda:0")
    xg = x.cuda().requires_grad_(True)
    assert G.supported_train(gd, xg)
    y = G.gru_forward_train(gd, xg)
    with torch.no_grad():
        ref_y = g.cpu()(x)[0]
        g.to("cuda:0")
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref_y.numpy(), rtol=2e-5, atol=2e-5)
    (y * probe.cuda()).sum().backward()
    def rel(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    assert G.last_bwd_form == ("persistent" if (H % 256 == 0 and H <= 1024) else "steps")
    assert G.last_train_form == ("persistent" if (H % 128 == 0 and H <= 1024) else "steps")
    assert rel(xg.grad.cpu(), want_dx) < 5e-5
    for k, p in gd.named_parameters():
        assert p.grad is not None and rel(p.grad.cpu(), want[k]) < 5e-5, (k, rel(p.grad.cpu(), want[k]))
    if H % 256 == 0 and H <= 1024:         # the per-step backward on the same inputs: the two forms agree to summation order
        first = {k: p.grad.clone() for k, p in gd.named_parameters()}
        for p in gd.parameters():
            p.grad = None
        G.BWD_PERSISTENT = False
        try:
            xg2 = x.cuda().requires_grad_(True)
            (G.gru_forward_train(gd, xg2) * probe.cuda()).sum().backward()
            assert G.last_bwd_form == "steps"
        finally:
            G.BWD_PERSISTENT = True
        for k, p in gd.named_parameters():
            assert rel(p.grad, first[k]) < 2e-5, k


@pytest.mark.gpu
def test_encoder_with_hip_gru_matches_library_gru_forward_and_backward(tmp_path):
    """The whole encoder (backbone.py:298-351) with its frame-context GRU on the HIP kernels -- inference path and autograd
    path -- against the same module with the library GRU, at a width the kernels' persistent form takes (R = 256 -> H = 128)."""
    import dataclasses
    from cvc.model import backbone
    from cvc.model.backbone import RegionalFeatureExtractorGVD
    Dw = dataclasses.replace(D, R=256, A=64, F=9, B=5)
    tables = synth.detectron_tables(Dw, 3)
    o = make_opts(Dw, seq_per_img=1, enable_BUTD=False, att_input_mode="both", num_sampled_frm=4, finetune_cnn=False,
                  att_feat_size=Dw.G, fc_feat_size=synth.SEG_FEAT_DIM, t_attn_size=Dw.F, second_drop_prob=0.3, att_model="topdown",
                  t_attn_mode="bigru", itod={i + 1: "d%d" % i for i in range(Dw.DET)},
                  vg_cls=["vg%d" % i for i in range(tables["glove_vg_cls"].shape[0])],
                  glove_clss=torch.from_numpy(tables["glove_clss"]), glove_vg_cls=torch.from_numpy(tables["glove_vg_cls"]),
                  detectron_tables=tables, test_mode=False)
    torch.manual_seed(0)
    enc = RegionalFeatureExtractorGVD(o).to("cuda:0").eval()
    inp = to_dev(synth.encoder_inputs(Dw, 3), torch.device("cuda:0"))
    from cvc.misc import utils
    overlaps = utils.bbox_overlaps(inp["proposals"], inp["gt_bboxs"], inp["frm_mask"] | inp["pnt_mask_in"][:, 1:].unsqueeze(-1))
    from cvc import dense, encoder_ops
    res = {}
    for hip_gru in (True, False):
        backbone.HIP_GRU = hip_gru
        dense.ENABLED = encoder_ops.ENABLED = hip_gru   # ... and its dense layers / fused pieces on the own kernels vs the library
        min_rows, dense.MIN_ROWS = dense.MIN_ROWS, 8      # (this test's 45 rows would otherwise stay on the library kernels)
        try:
            with torch.no_grad():
                out_inf = run_encoder(enc, inp, overlaps)
            enc.zero_grad(set_to_none=True)
            enc.context_enc.train(); enc.context_enc.dropout = 0.0          # MIOpen's backward needs train mode; no dropout: deterministic
            out = run_encoder(enc, inp, overlaps)
            probe_loss(out).backward()
            enc.context_enc.eval()
            res[hip_gru] = (out_inf, out, {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None})
        finally:
            backbone.HIP_GRU = True
            dense.ENABLED = encoder_ops.ENABLED = True
            dense.MIN_ROWS = min_rows
    for a, b in ((res[True][0], res[False][0]), (res[True][1], res[False][1])):
        for name, x, y in zip(OUT, a, b):
            if torch.is_tensor(x) and x.dtype.is_floating_point:
                np.testing.assert_allclose(x.detach().cpu().numpy(), y.detach().cpu().numpy(), rtol=2e-4, atol=2e-5, err_msg=name)
    assert set(res[True][2]) == set(res[False][2])
    for k, g in res[True][2].items():
        ref = res[False][2][k]
        assert float((g - ref).norm() / (ref.norm() + 1e-20)) < 2e-4, k


@pytest.mark.gpu
def test_gru_hip_autograd_full_size_vs_library_cpu():
    """Config-2 encoder width and batch (B=64, R=2048 -> H=1024, 2 layers, both directions) over F=240 frames under autograd: output,
    input gradient and every parameter gradient of cvc.gru.gru_forward_train against torch autograd of nn.GRU on the host CPU.
    (Half of config 2's 480 frames: the host's forward + backward is what this test waits for -- 67 s at F=480; the 480-step
    recurrence itself is covered by test_gru_hip_full_size_vs_library_cpu.)"""
    from cvc import gru as G
    g = _gru(2048, 1024, 2, True, 13)
    x = torch.randn(64, 240, 2048)
    probe = torch.randn(64, 240, 2048) / 240 ** 0.5
    xc = x.clone().requires_grad_(True)
    ref_y = g(xc)[0]
    (ref_y * probe).sum().backward()
    want = {k: p.grad.clone() for k, p in g.named_parameters()}
    want_dx, ref_y = xc.grad.clone(), ref_y.detach()
    for p in g.parameters():
        p.grad = None
    gd = g.to("cuda:0")
    xg = x.cuda().requires_grad_(True)
    y = G.gru_forward_train(gd, xg)
    assert float((y.detach().cpu() - ref_y).abs().max()) < 5e-5
    (y * probe.cuda()).sum().backward()
    def rel(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    assert rel(xg.grad.cpu(), want_dx) < 1e-4
    for k, p in gd.named_parameters():
        assert rel(p.grad.cpu(), want[k]) < 1e-4, (k, rel(p.grad.cpu(), want[k]))


@pytest.mark.gpu
def test_encoder_packs_follow_weight_updates_that_bypass_version_counters(tmp_path):
    """Eval -> optimizer step -> eval: a fused Adam step (and a HIP-graph replay of the training step) changes parameters without
    touching their _version, which is all the encoder's packed GRU / dense operands used to watch -- every evaluation after the
    first then ran on the weights of the first.  Trainer._weights_changed / invalidate_decode_cache bump hip.weights_generation(),
    which is part of the packs' stamps: the second evaluation must match the library modules on the UPDATED weights."""
    import dataclasses
    from cvc import dense, hip
    from cvc.model import backbone
    from cvc.model.backbone import RegionalFeatureExtractorGVD
    Dw = dataclasses.replace(D, R=256, A=64, F=9, B=5)
    tables = synth.detectron_tables(Dw, 4)
    o = make_opts(Dw, seq_per_img=1, enable_BUTD=False, att_input_mode="both", num_sampled_frm=4, finetune_cnn=False,
                  att_feat_size=Dw.G, fc_feat_size=synth.SEG_FEAT_DIM, t_attn_size=Dw.F, second_drop_prob=0.3, att_model="topdown",
                  t_attn_mode="bigru", itod={i + 1: "d%d" % i for i in range(Dw.DET)},
                  vg_cls=["vg%d" % i for i in range(tables["glove_vg_cls"].shape[0])],
                  glove_clss=torch.from_numpy(tables["glove_clss"]), glove_vg_cls=torch.from_numpy(tables["glove_vg_cls"]),
                  detectron_tables=tables, test_mode=True)
    torch.manual_seed(1)
    enc = RegionalFeatureExtractorGVD(o).to("cuda:0").eval()
    inp = to_dev(synth.encoder_inputs(Dw, 4), torch.device("cuda:0"))
    from cvc.misc import utils
    overlaps = utils.bbox_overlaps(inp["proposals"], inp["gt_bboxs"], inp["frm_mask"] | inp["pnt_mask_in"][:, 1:].unsqueeze(-1))
    min_rows, dense.MIN_ROWS = dense.MIN_ROWS, 8
    try:
        from cvc import encoder_ops

        def evaluate(hip_path):
            backbone.HIP_GRU, dense.ENABLED, encoder_ops.ENABLED = hip_path, hip_path, hip_path
            with torch.no_grad():
                return [x.clone() for x in run_encoder(enc, inp, overlaps)[:6]]
        first = evaluate(True)
        # a parameter update of the kind the trainer makes: in place, _version untouched (torch._foreach / fused kernels do this;
        # here: a raw write through .data, which bypasses the version counter the same way)
        params = [p for p in enc.parameters() if p.requires_grad]
        versions = [p._version for p in params]
        g = torch.Generator(device="cuda:0").manual_seed(5)
        for p in params:
            p.data.add_(torch.randn(p.shape, device=p.device, generator=g) * 0.05 * p.data.abs().mean())
        assert [p._version for p in params] == versions
        stale, lib_out = evaluate(True), evaluate(False)
        # nobody announced the update: the HIP path still multiplies with the packs of the OLD weights (this is the bug the
        # generation stamp fixes -- without the bump below the comparison with the library modules fails)
        assert float((stale[1] - lib_out[1]).abs().max()) > 1e-3 * float(lib_out[1].abs().max())
        hip.bump_weights_generation()                         # what Trainer._weights_changed() does after every optimizer step
        fresh = evaluate(True)
        assert not torch.equal(first[1], fresh[1])
        for name, x, y in zip(OUT, fresh, lib_out):
            np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=2e-4, atol=2e-5, err_msg=name)
    finally:
        backbone.HIP_GRU, dense.ENABLED, dense.MIN_ROWS = True, True, min_rows
        encoder_ops.ENABLED = True


# ------------------------------------------------------------------ fused inference pieces (csrc/encoder_ops.hip)
@pytest.mark.gpu
@pytest.mark.parametrize("B,N,C,G", [(64, 100, 432, 2048), (3, 7, 7, 24), (5, 33, 100, 36)])
def test_class_similarity_softmax_kernel(B, N, C, G):
    """logits (tile GEMM) + bias + pad fill + softmax over classes, in both layouts, against the torch formulation of
    backbone.py:222-235 (fp64); padded regions come out uniform (-1e8 in every class), as in the reference."""
    from cvc import hip
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B + C)
    feats = torch.randn(B, N, G, generator=g).to(dev)
    table = torch.randn(C, G, generator=g).to(dev) * 0.3
    bias = torch.randn(C, generator=g).to(dev)
    pad = (torch.rand(B, N, generator=g) < 0.3).to(dev)
    logits = hip.tile_mm(feats.reshape(B * N, G), hip.TileOperand(table))
    sim, rows = torch.empty(B, C, N, device=dev), torch.empty(B, N, C, device=dev)
    hip._check(hip.lib().cvc_class_softmax_fwd(logits.data_ptr(), C, bias.data_ptr(), hip._mask(pad).data_ptr(), B, N, C, sim.data_ptr(),
                                               rows.data_ptr(), hip._stream()), "cvc_class_softmax_fwd")
    ref = torch.matmul(table.double(), feats.double().transpose(1, 2)) + bias.double().view(1, -1, 1)
    ref = torch.softmax(ref.masked_fill(pad.unsqueeze(1), -1e8), dim=1)
    # (a probability's relative error is the ABSOLUTE error of its logit: |logit| reaches ~40 here, fp32 products over K = 2048)
    np.testing.assert_allclose(sim.cpu().numpy(), ref.float().cpu().numpy(), rtol=2e-4, atol=1e-7)
    assert torch.equal(rows, sim.transpose(1, 2).contiguous())
    np.testing.assert_allclose(sim[pad.unsqueeze(1).expand(B, C, N)].cpu().numpy(), 1.0 / C, rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,widths", [(6400, (2048, 300, 432)), (64, (3072, 4)), (5, (7,)), (33, (24, 300, 7))])
def test_layernorm_concat_kernel(rows, widths):
    from cvc import encoder_ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows)
    xs = [(torch.randn(rows, w, generator=g) * (1 + i) + i).to(dev) for i, w in enumerate(widths)]
    got = encoder_ops.layernorm_cat(xs)
    ref = torch.cat([torch.nn.functional.layer_norm(x.double(), [x.shape[-1]]) for x in xs], -1)
    np.testing.assert_allclose(got.cpu().numpy(), ref.float().cpu().numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.gpu
def test_encoder_forward_wide_golden_on_the_hip_forms(tmp_path):
    """The reference encoder's own eval outputs at rnn_size 256 (tests/golden/g8_encoder_wide.npz, tools/make_golden.py g8) against
    the mirror on the GPU with every HIP form active: persistent GRU recurrence (H = 128), frame / region rows on the tile GEMM,
    fused class-similarity softmax, layer norms + concat, frame-embedding epilogue with the folded BatchNorm."""
    import dataclasses
    from cvc import dense, gru as gru_hip
    from cvc.model.backbone import RegionalFeatureExtractorGVD
    g8 = Golden("g8_encoder_wide.npz")
    Dw = dataclasses.replace(D, B=5, N=20, F=9, R=256, A=64, E=32, K=3)
    seed = int(g8["meta.seed"])
    tables = synth.detectron_tables(Dw, seed)
    dev = torch.device("cuda:0")
    inp = to_dev(synth.encoder_inputs(Dw, seed), dev)
    from cvc.misc import utils
    overlaps = utils.bbox_overlaps(inp["proposals"], inp["gt_bboxs"], inp["frm_mask"] | inp["pnt_mask_in"][:, 1:].unsqueeze(-1))
    min_rows, dense.MIN_ROWS = dense.MIN_ROWS, 8
    try:
        for name, over in (("train.", {}), ("test.", dict(test_mode=True))):
            o = make_opts(Dw, seq_per_img=1, enable_BUTD=False, att_input_mode="both", num_sampled_frm=4, finetune_cnn=False,
                          att_feat_size=Dw.G, fc_feat_size=synth.SEG_FEAT_DIM, t_attn_size=Dw.F, second_drop_prob=0.3, att_model="topdown",
                          t_attn_mode="bigru", itod={i + 1: "d%d" % i for i in range(Dw.DET)},
                          vg_cls=["vg%d" % i for i in range(tables["glove_vg_cls"].shape[0])],
                          glove_clss=torch.from_numpy(tables["glove_clss"]), glove_vg_cls=torch.from_numpy(tables["glove_vg_cls"]),
                          detectron_tables=tables, **dict(dict(test_mode=False), **over))
            enc = RegionalFeatureExtractorGVD(o)
            ctor = {k: v.detach().clone() for k, v in enc.state_dict().items() if k in synth.ENCODER_CTOR_KEYS}
            enc.load_state_dict({k: (ctor[k] if k in ctor else torch.from_numpy(np.asarray(synth.encoder_fill(k, v.shape, seed))))
                                 for k, v in enc.state_dict().items()})
            enc = enc.to(dev).eval()
            gru_hip.last_form = None
            with torch.no_grad():
                res = run_encoder(enc, inp, overlaps)
            assert gru_hip.last_form == "persistent"                       # the HIP recurrence ran (not the library module)
            assert getattr(enc, "_cvc_class_table", None) is not None and getattr(enc, "_cvc_bn_fold", None) is not None   # fused pieces ran
            for k, x in zip(OUT, res[:8]):
                want = g8[name + "out." + k]
                if x.dtype.is_floating_point:
                    np.testing.assert_allclose(x.cpu().numpy(), want, rtol=2e-4, atol=2e-5, err_msg=name + k)
                else:
                    np.testing.assert_array_equal(x.cpu().numpy(), want, err_msg=name + k)
            np.testing.assert_allclose(res[9].cpu().numpy(), g8[name + "out.cls_loss"], rtol=2e-4, atol=1e-6)
    finally:
        dense.MIN_ROWS = min_rows


@pytest.mark.gpu
def test_encoder_train_mode_runs_on_own_kernels_and_matches_the_torch_formulation(tmp_path):
    """train(): every Linear -> ReLU -> Dropout block (mask generated in the kernel), BatchNorm1d on batch statistics + ReLU, the
    class-similarity softmax, the layer norms and the GRU's inter-layer dropout on the build's kernels (csrc/encoder_train.hip) --
    outputs, every parameter gradient and the BatchNorm running statistics against the torch formulation of the same module fed the
    SAME masks (the generator's host restatement, dictated through cvc.dropout.injected); no library fallback announced."""
    import dataclasses
    from cvc import dropout, hip
    from cvc.model.backbone import RegionalFeatureExtractorGVD
    from cvc.misc import utils
    dev = torch.device("cuda:0")
    Dw = dataclasses.replace(D, R=256, A=64, F=9, B=5)
    tables = synth.detectron_tables(Dw, 3)
    o = make_opts(Dw, seq_per_img=1, enable_BUTD=False, att_input_mode="both", num_sampled_frm=4, finetune_cnn=False,
                  att_feat_size=Dw.G, fc_feat_size=synth.SEG_FEAT_DIM, t_attn_size=Dw.F, second_drop_prob=0.3, att_model="topdown",
                  t_attn_mode="bigru", itod={i + 1: "d%d" % i for i in range(Dw.DET)},
                  vg_cls=["vg%d" % i for i in range(tables["glove_vg_cls"].shape[0])],
                  glove_clss=torch.from_numpy(tables["glove_clss"]), glove_vg_cls=torch.from_numpy(tables["glove_vg_cls"]),
                  detectron_tables=tables, test_mode=False)
    torch.manual_seed(0)
    enc = RegionalFeatureExtractorGVD(o).to(dev).train()
    inp = to_dev(synth.encoder_inputs(Dw, 3), dev)
    overlaps = utils.bbox_overlaps(inp["proposals"], inp["gt_bboxs"], inp["frm_mask"] | inp["pnt_mask_in"][:, 1:].unsqueeze(-1))
    bn = enc.att_embed_aux[0]
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    dropout.seed(424242)
    dropout.advance(dev)
    warned = set(hip._warned)
    enc.zero_grad(set_to_none=True)
    out = run_encoder(enc, inp, overlaps)
    probe_loss(out).backward()
    assert set(hip._warned) == warned, set(hip._warned) - warned             # no library fallback announced in train()
    got = ([t.detach().clone() if isinstance(t, torch.Tensor) else t for t in out],
           {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None}, bn.running_mean.clone(), bn.running_var.clone())
    # ---- the torch formulation with the same masks
    ps = {"enc.pool_embed": 0.3, "enc.gru.0": 0.2}
    used = []

    def masks(site, shape):
        used.append(site)
        return dropout.host_mask(site, shape, ps.get(site, 0.5), dev)
    bn.running_mean.copy_(rm0); bn.running_var.copy_(rv0)
    enc.zero_grad(set_to_none=True)
    with dropout.injected(masks):
        ref = run_encoder(enc, inp, overlaps)
        probe_loss(ref).backward()
    assert {"enc.loc_fc", "enc.fc_embed", "enc.seg_info", "enc.att0", "enc.att1", "enc.pool_embed", "enc.ctx2pool_grd", "enc.vis_table",
            "enc.gru.0"} <= set(used), sorted(set(used))
    for a, b in zip(got[0], ref):
        if isinstance(a, torch.Tensor) and a.dtype.is_floating_point:
            np.testing.assert_allclose(a.cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-4, atol=2e-5)
        elif isinstance(a, torch.Tensor):
            assert torch.equal(a, b)
    np.testing.assert_allclose(got[2].cpu().numpy(), bn.running_mean.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got[3].cpu().numpy(), bn.running_var.cpu().numpy(), rtol=1e-4, atol=1e-6)
    checked = 0
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    for k, p in enc.named_parameters():
        if p.grad is None:
            assert k not in got[1], k
            continue
        assert k in got[1], k
        assert rel(got[1][k], p.grad) < 1e-3, (k, rel(got[1][k], p.grad))
        checked += 1
    assert checked >= 20


@pytest.mark.gpu
@pytest.mark.parametrize("momentum", [0.1, None])
def test_batchnorm_train_kernel_updates_running_stats_like_the_module_and_invalidates_the_eval_fold(momentum):
    """cvc.encoder_ops.batchnorm_relu_train (att_embed_aux in train(), backbone.py:81, 332): the kernel writes running_mean /
    running_var through raw pointers -- their version counters do not move -- so the call itself bumps the weights generation the
    eval-mode fold is cached on (round-4 advisor finding: train() forward, then eval() without an optimizer step, reused the old
    statistics); momentum=None is nn.BatchNorm1d's cumulative average (factor 1 / num_batches_tracked), not 0.1."""
    import torch.nn as nn
    from cvc import encoder_ops, hip
    dev = torch.device("cuda:0")
    C_, rows = 48, 96
    g = torch.Generator().manual_seed(3)
    ref = nn.BatchNorm1d(C_, momentum=momentum)
    with torch.no_grad():
        ref.weight.copy_(torch.rand(C_, generator=g) + 0.5); ref.bias.copy_(torch.randn(C_, generator=g) * 0.1)
    mine = nn.BatchNorm1d(C_, momentum=momentum).to(dev)
    mine.load_state_dict(ref.state_dict())
    ref.train(); mine.train()
    for step in range(3):
        x = torch.randn(rows, C_, generator=g) * (1 + step) + step
        gen0 = hip.weights_generation()
        y = encoder_ops.batchnorm_relu_train(x.to(dev).requires_grad_(True), mine)
        assert hip.weights_generation() > gen0                  # whatever is cached on the old statistics is stale now
        want = torch.relu(ref(x))
        np.testing.assert_allclose(y.detach().cpu().numpy(), want.detach().numpy(), rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(mine.running_mean.cpu().numpy(), ref.running_mean.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(mine.running_var.cpu().numpy(), ref.running_var.numpy(), rtol=1e-5, atol=1e-6)
        assert int(mine.num_batches_tracked) == int(ref.num_batches_tracked) == step + 1
