"""GPU tests of the training harness: Trainer.train_step (loss mix, backward, clip, Adam) and the
single-rank GradReducer path; dropout active (train mode)."""
import numpy as np
import pytest
import torch

from cvc import synth

pytestmark = pytest.mark.gpu


def _setup(dev, d, seed=5, **over):
    from helpers import make_opts, to_dev
    from cvc.model.captioner import DecodeAndGroundCaptionerGVDROI, PrecomputedRegionFeatures
    from cvc.trainer import Trainer, build_optimizer
    from cvc import opts as cvc_opts
    o = cvc_opts.parse_opt([])
    for k, v in vars(make_opts(d, **over)).items():
        setattr(o, k, v)
    o.xe_loss_weight, o.caption_consistency_loss_weight, o.learning_rate, o.batch_size = 0.5, 0.5, 2e-3, d.B
    model = DecodeAndGroundCaptionerGVDROI(o, roi_extractor=PrecomputedRegionFeatures(d.DET, d.G))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.hot_path_state_dict(d, seed).items()}, strict=False)
    model = model.to(dev)
    f = to_dev(synth.clip_features(d, seed), dev)
    b = to_dev(synth.label_glue_batch(d, seed), dev)
    batch = (f, b["input_seq"], b["gt_seq"], b["num"].cpu(), b["proposals"], b["gt_bboxs"], b["box_mask"],
             ["v_x_segment_%02d" % i for i in range(d.B)], torch.zeros(d.B, d.N, 1), b["frm_mask"], b["sample_idx"],
             f["pnt_mask"][:, 1:])
    return o, model, batch, Trainer, build_optimizer


def test_train_steps_reduce_the_loss():
    dev = torch.device("cuda:0")
    d = synth.CONFIGS["tiny"]
    o, model, batch, Trainer, build_optimizer = _setup(dev, d, train_decoder_only=False)
    tr = Trainer(o, None, model, build_optimizer(model, o), None, None)
    model.train()
    torch.manual_seed(0)
    losses = [float(tr.train_step(batch)[0]) for _ in range(30)]
    assert np.isfinite(losses).all()
    assert np.mean(losses[-5:]) < np.mean(losses[:5]) - 0.05, losses
    # clip_grad_norm_(0.1): after a step the global grad norm handed to Adam was <= grad_clip
    total = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None))
    assert float(total) <= o.grad_clip * 1.001
    # dead parameters stayed without gradients
    assert model.decoder_core.i2h_2.weight.grad is None


def test_eval_after_training_returns_reference_tuple():
    dev = torch.device("cuda:0")
    d = synth.CONFIGS["tiny"]
    o, model, batch, Trainer, build_optimizer = _setup(dev, d)
    tr = Trainer(o, None, model, build_optimizer(model, o), None, None)
    model.eval()
    b = tr._prepare(batch, False)
    seq, att, none = tr._call(b, True)
    assert none is None and seq.shape == (d.B, d.T) and att.shape == (d.B, d.T, d.N)
    assert torch.allclose(att.sum(2), torch.ones(d.B, d.T, device=dev), atol=1e-5)


def test_main_entry_point_one_epoch(tmp_path):
    """cvc.main: options -> synthetic dataset -> model -> Trainer.train/eval -> checkpoints with the reference's
    file names and state_dict keys."""
    import json
    import os
    from cvc import main as cvc_main
    from conftest import GOLDEN
    common = ["--no_cfg", "--max_epochs", "1", "--batch_size", "4", "--synthetic_clips", "12", "--num_prop_per_frm", "7",
              "--t_attn_size", "5", "--rnn_size", "32", "--att_hid_size", "16", "--input_encoding_size", "16",
              "--seq_length", "4", "--vis_encoding_size", "24", "--tensorboard", "0", "--disp_interval", "100",
              "--exp_name", "t", "--learning_rate", "0.001", "--language_eval", "--results_dir", str(tmp_path / "results")]
    rc = cvc_main.main(common + ["--checkpoint_path", str(tmp_path) + "/", "--id", "t1"])
    assert rc == 0
    dense = json.load(open(tmp_path / "results" / "densecap-validation-t1.json"))
    assert len(dense["results"]) == 12 and all("sentence" in r[0] for r in dense["results"].values())
    # reference trainer.py:256-260: every predicted segment carries its [start, end] timestamp, rounded to 2 decimals
    for vid, segs in dense["results"].items():
        i = int(vid[len("v_synth"):])
        assert segs[0]["timestamp"] == [round(1.5 * i, 2), round(1.5 * i + 7.25, 2)]
    sd = torch.load(os.path.join(tmp_path, "t", "model-best.pth"), map_location="cpu")
    ref_keys = set(json.load(open(os.path.join(GOLDEN, "config_surface.json")))["state_dict"])
    assert set(sd.keys()) == ref_keys
    assert os.path.exists(os.path.join(tmp_path, "t", "infos_t1-best.pkl"))
    import pickle
    infos = pickle.load(open(os.path.join(tmp_path, "t", "infos_t1.pkl"), "rb"))
    hist = pickle.load(open(os.path.join(tmp_path, "t", "histories_t1.pkl"), "rb"))
    assert infos["epoch"] == 0 and "best_val_score" in infos and set(hist) >= {"val_result_history", "lr_history"}
    lr0 = hist["lr_history"][0]
    assert lr0 == pytest.approx(0.001)                    # no scorer -> no CIDEr -> the plateau scheduler is not stepped
    # --inference_only without a checkpoint must refuse (the reference loads model.pth first, main.py:121-147)
    with pytest.raises(SystemExit):
        cvc_main.main(common + ["--inference_only", "--checkpoint_path", str(tmp_path / "nowhere") + "/", "--id", "t1"])
    # --resume restores the saved weights: inference on the resumed model reproduces the first run's sentences
    rc = cvc_main.main(common + ["--inference_only", "--resume", "True", "--checkpoint_path", str(tmp_path) + "/", "--id", "t1"])
    assert rc == 0
    dense2 = json.load(open(tmp_path / "results" / "densecap-validation-t1.json"))
    assert dense2["results"] == dense["results"]


def test_main_entry_point_raw_features_through_encoder(tmp_path):
    """Same loop fed raw frame / region features: build_model picks the mirrored once-per-clip encoder
    (cvc/model/backbone.py) and its parameters are trained and checkpointed under the reference's key names."""
    import os
    from cvc import main as cvc_main
    rc = cvc_main.main(["--no_cfg", "--synthetic_raw", "--max_epochs", "1", "--batch_size", "4", "--synthetic_clips", "28",
                        "--num_prop_per_frm", "7", "--t_attn_size", "5", "--rnn_size", "32", "--att_hid_size", "16",
                        "--input_encoding_size", "16", "--seq_length", "4", "--vis_encoding_size", "24", "--att_feat_size", "24",
                        "--tensorboard", "0", "--disp_interval", "100", "--checkpoint_path", str(tmp_path) + "/",
                        "--exp_name", "raw", "--learning_rate", "0.001"])
    assert rc == 0
    # the reference's own training flow (raw features through the encoder every step) runs the captured step since round 6: the
    # persistent GRU's error words stay on the device (cvc.hip.defer_errors)
    st = cvc_main.LAST_TRAINER.graph_stats
    assert st["captured"] >= 1 and st["replayed"] >= 3 and st["eager"] + st["replayed"] == 6, st      # (a new trimmed shape = one more eager step)
    assert cvc_main.LAST_TRAINER.deferred_stats == dict(void_steps=0, rerun_steps=0)
    sd = torch.load(os.path.join(tmp_path, "raw", "model-best.pth"), map_location="cpu")
    for k in ("roi_feat_extractor.context_enc.weight_hh_l1_reverse", "roi_feat_extractor.att_embed.1.0.weight",
              "roi_feat_extractor.att_embed_aux.0.running_var", "roi_feat_extractor.pool_embed.0.weight",
              "roi_feat_extractor.vis_classifiers_bias", "decoder_core.lang_lstm.weight_hh"):
        assert k in sd, k


def test_device_prefetcher_preserves_order_and_contents():
    """cvc.prefetch.DevicePrefetcher: same batches, same order, tensors usable on the current stream."""
    from cvc.prefetch import DevicePrefetcher
    dev = torch.device("cuda:0")
    batches = [({"a": torch.full((256, 1024), float(i)), "ids": ["s%d" % i]}, torch.arange(4) + i, "tag%d" % i) for i in range(5)]
    seen = []

    def prepare(raw):
        assert raw[0]["a"].is_pinned() and raw[1].is_pinned()
        return {"a": raw[0]["a"].to(dev, non_blocking=True), "ids": raw[0]["ids"], "n": raw[1].to(dev, non_blocking=True),
                "tag": raw[2]}
    for i, b in enumerate(DevicePrefetcher(batches, prepare, dev, limit=4)):
        assert b["tag"] == "tag%d" % i and b["ids"] == ["s%d" % i]
        seen.append((float((b["a"] * 2).sum()), b["n"].tolist()))
    assert seen == [(2.0 * i * 256 * 1024, [i, i + 1, i + 2, i + 3]) for i in range(4)]


def test_graphed_train_step_equals_eager():
    """Trainer.train_step_graphed (whole step in one HIP graph) reproduces the eager step sequence (dropout off so that
    both are deterministic; the graphed path runs 3 warm-up steps before capture)."""
    dev = torch.device("cuda:0")
    d = synth.CONFIGS["tiny"]
    losses = {}
    for mode in ("eager", "graph"):
        o, model, batch, Trainer, build_optimizer = _setup(dev, d, train_decoder_only=False)
        model.eval()
        tr = Trainer(o, None, model, build_optimizer(model, o, capturable=(mode == "graph")), None, None)
        tr.model.eval()
        if mode == "eager":
            seq = []
            for _ in range(6):
                b = tr._prepare(batch, True)
                seq.append(tr._core_step(b)[0].item())
            losses[mode] = seq[3:]
        else:
            losses[mode] = [tr.train_step_graphed(batch)[0].item() for _ in range(3)]
    np.testing.assert_allclose(losses["graph"], losses["eager"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("graphed", [False, True])
def test_decode_weights_cache_follows_parameter_updates(graphed):
    """The cached decode binding (packed weight copies) must not survive a training step -- neither an eager one
    (fused Adam does not bump version counters) nor a HIP-graph replay (no Python-side trace of the update at all)."""
    dev = torch.device("cuda:0")
    d = synth.CONFIGS["tiny"]
    o, model, batch, Trainer, build_optimizer = _setup(dev, d)
    tr = Trainer(o, None, model, build_optimizer(model, o, capturable=graphed), None, None)
    model.eval()
    w1 = model.decode_weights()
    assert model.decode_weights() is w1                                   # unchanged parameters: same binding
    b = tr._prepare(batch, False)
    seq1, att1, _ = tr._call(b, True)
    model.train()
    (tr.train_step_graphed if graphed else tr.train_step)(batch)
    model.eval()
    w2 = model.decode_weights()
    assert w2 is not w1
    seq2, att2, _ = tr._call(b, True)
    assert not torch.equal(att1, att2)                                    # the decode sees the updated weights


@pytest.mark.parametrize("B,N,F,R,A,E,V,T", [(33, 20, 16, 256, 64, 32, 50, 2), (64, 3, 5, 128, 16, 16, 50, 4),
                                             (1, 3, 1, 256, 16, 16, 50, 4), (8, 7, 1, 128, 32, 16, 97, 2),
                                             (96, 9, 6, 128, 32, 16, 50, 3), (128, 5, 4, 256, 32, 32, 50, 3), (130, 4, 3, 128, 16, 16, 50, 2)])
def test_cyclical_gradients_vs_oracle_on_other_shapes(B, N, F, R, A, E, V, T):
    """Losses and every parameter gradient of the cyclical pass (eval-mode dropout) against the CPU oracle's autograd on
    shapes the golden fixtures do not cover: two MFMA row tiles / ragged M, a single clip, one frame, widths that take the
    backward-data GEMM's partial slabs; round 6: more than 64 clips per GPU -- the C-driven loops run once per group of <= 64 clips
    (96 -> 64 + 32, 128 -> 64 + 64, 130 -> 64 + 64 + 2), batch-wide pieces once, the groups' weight gradients summed."""
    import dataclasses
    from helpers import build_model, to_dev, model_call
    from oracle import ref_cpu as O
    dev = torch.device("cuda:0")
    d = dataclasses.replace(synth.CONFIGS["tiny"], B=B, N=N, F=F, R=R, A=A, E=E, V=V, T=T, K=min(3, N))
    seed = B + R
    sd, f, b = synth.hot_path_state_dict(d, seed), synth.clip_features(d, seed), synth.label_glue_batch(d, seed)
    P = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in O.to_torch(sd).items()}
    for k in list(P):                                           # the reconstructor shares the decoder's LSTM cells
        if k.startswith("attended_roi_decoder_core.") and "lstm" in k:
            P[k] = P[k.replace("attended_roi_decoder_core.", "decoder_core.")]
    ref = O.cyclical_forward(P, O.to_torch(f), O.to_torch(b), T=T, vocab_size=V)
    O.training_loss(ref, xe_loss_weight=0.5, w_att2=0.05, w_cls=0.0, caption_consistency_loss_weight=0.5).backward()
    model = build_model(d, sd, dev)
    out = model_call(model, to_dev(f, dev), to_dev(b, dev), False)
    for got, want in zip(out, ref):
        assert float(got.detach().mean()) == pytest.approx(float(want.detach().mean()), rel=2e-5, abs=2e-6)
    lm, a2, _g, _cls, rec = [x.mean() for x in out]
    (0.5 * lm + 0.05 * a2 + 0.5 * rec).backward()
    checked = 0
    for n, p in model.named_parameters():
        if n.startswith("roi_feat_extractor") or n not in P or P[n].grad is None:
            continue
        want = P[n].grad.double()
        err = float((p.grad.cpu().double() - want).norm())
        assert err <= 5e-4 * float(want.norm()) + 1e-6, (n, err, float(want.norm()))   # alpha_net.bias: true gradient ~0
        checked += 1
    assert checked >= 15
    if B > 64:
        from cvc import train_loops
        assert train_loops.eligible(B, R, E, A, p, T) and len(train_loops.clip_groups(B)) == (B + 63) // 64


def test_more_than_64_clips_graphed_step_equals_eager_with_reducer_and_train_mode_dropout():
    """B = 96 per GPU, train mode (in-kernel dropout: every group of clips draws its own sites), gradient arenas with the exchange on
    a one-rank communicator: the step captured into a HIP graph is bit-equal to the eager step, and the reducer's buckets leave only
    after BOTH groups' weight gradients are in (a first group's in-place write is not final)."""
    import dataclasses
    from cvc import dropout
    from cvc.comm import RcclComm
    from cvc.distributed import GradReducer
    dev = torch.device("cuda:0")
    d = dataclasses.replace(synth.CONFIGS["tiny"], B=96, N=9, F=6, R=128, A=32, E=16, V=50, T=3)
    finals = []
    for graphed in (False, True):
        o, model, batch, Trainer, build_optimizer = _setup(dev, d, train_decoder_only=False)
        model.train()
        comm = RcclComm.single()
        red = GradReducer(model.named_parameters(), comm=comm, always_exchange=True)
        try:
            tr = Trainer(o, None, model, build_optimizer(model, o, capturable=True), None, None, grad_reducer=red)
            dropout.seed(31)
            # (train_step_graphed takes 3 warm-up steps before its capture: 3 + 3 replays against 6 eager steps)
            for _ in range(3 if graphed else 6):
                res = tr.train_step_graphed(batch) if graphed else tr.train_step(batch)
            torch.cuda.synchronize()
            assert bool(torch.isfinite(res[0]).all())
            finals.append({k: v.detach().clone() for k, v in model.state_dict().items()})
            tr._graph = None
        finally:
            red.remove_hooks()
            comm.destroy()
    for k in finals[0]:
        assert torch.equal(finals[0][k], finals[1][k]), k


def _run_isolated(scenario: str, timeout: int = 900):
    """Runs _scenario_<name>() of this module in a CHILD interpreter; passes when the child printed SCENARIO-PASSED -- which it
    prints AFTER its teardown (communicator destroyed, process group released) -- and exited with code 0.  The scenarios that bring
    up RCCL communicators run this way so that a communicator never outlives its test inside the long-lived pytest process.  No
    retries: the exchange runs on the package's own communicator (cvc.comm.RcclComm), which has no watchdog thread that could die
    under a graph capture (round 4 re-ran children that c10d's watchdog had killed; that code is gone with its cause)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    paths = [root, os.path.join(root, "cyclical-visual-captioning_amd"), here]
    code = (f"import sys; sys.path[:0] = {paths!r}; import test_gpu_train as t; t._scenario_{scenario}()")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", code], capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
    report = f"scenario {scenario}: rc={r.returncode}\n--- stdout\n{r.stdout[-3000:]}\n--- stderr\n{r.stderr[-8000:]}"
    assert "SCENARIO-PASSED" in r.stdout, report
    assert "libcvc_hip.so" in r.stdout, "the child did not report the in-tree HIP library as loaded"
    assert r.returncode == 0, report


def _passed():
    """the child's marker, printed after the scenario's teardown: every check held and everything was released in order; names the
    loaded library (the product path, not a fallback)"""
    from cvc import hip
    hip.lib()
    with open("/proc/self/maps") as fh:
        libs = sorted({ln.split()[-1] for ln in fh if "libcvc_hip" in ln})
    print("SCENARIO-PASSED", libs, flush=True)


def test_rccl_allreduce_cabi_one_rank():
    """cvc_comm_unique_id -> cvc_comm_init -> cvc_allreduce_grads -> cvc_comm_destroy straight onto librccl (dlopen): a one-rank
    communicator on the single GPU of the box; the SUM over one rank must return the arena bit for bit."""
    _run_isolated("rccl_allreduce_cabi_one_rank", 300)


def _scenario_rccl_allreduce_cabi_one_rank():
    import ctypes as C
    from cvc import hip
    L = hip.lib()
    dev = torch.device("cuda:0")
    uid = (C.c_char * 128)()
    hip._check(L.cvc_comm_unique_id(uid), "cvc_comm_unique_id")
    comm = C.c_void_p()
    hip._check(L.cvc_comm_init(1, 0, uid, C.byref(comm)), "cvc_comm_init")
    assert comm.value
    arena = torch.randn(1 << 20, device=dev)
    want = arena.clone()
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        hip._check(L.cvc_allreduce_grads(comm, arena.data_ptr(), arena.numel(), s), "cvc_allreduce_grads")
    torch.cuda.synchronize()
    assert torch.equal(arena, want)
    assert L.cvc_allreduce_grads(None, arena.data_ptr(), 4, s) == -1          # bad arguments are rejected, not executed
    hip._check(L.cvc_comm_destroy(comm), "cvc_comm_destroy")
    # the Python face of the same calls (cvc.comm.RcclComm), an odd element count (the all-reduce branch needs world > 1: here the
    # pair runs with a one-element shard), the rank count through the communicator, destroy twice
    from cvc.comm import RcclComm
    c = RcclComm.single()
    assert c.count_ranks() == 1
    odd = torch.randn(1001, device=dev)
    keep = odd.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    c.all_reduce_(odd, side)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(odd, keep)
    with pytest.raises(TypeError):
        c.all_reduce_(odd.double())
    c.destroy()
    c.destroy()
    with pytest.raises(RuntimeError, match="destroyed"):
        c.all_reduce_(odd)
    _passed()


def test_gradient_exchange_on_rccl_one_rank_equals_no_exchange():
    """The package's own RCCL communicator with one rank, no torch.distributed at all: a training step whose GradReducer issues the
    in-place reduce_scatter + all_gather per bucket (exchange stream, HIP events) must leave bit-identical parameters to the same
    step without any exchange; the step captured into a HIP graph WITH that exchange replays bit-identically; an exchange that is
    not a stream operation (gloo, c10d's "nccl") is refused under capture."""
    _run_isolated("gradient_exchange_on_rccl_one_rank")


def _scenario_gradient_exchange_on_rccl_one_rank():
    from cvc.comm import RcclComm
    from cvc.distributed import GradReducer
    dev = torch.device("cuda:0")
    d = synth.CONFIGS["tiny"]
    comm = RcclComm.single()
    keep = []          # trainers (and the HIP graph one of them captured) stay alive until the checks have passed

    finals = []
    for exch in (False, True):
        o, model, batch, Trainer, build_optimizer = _setup(dev, d, train_decoder_only=False)
        model.eval()                                                  # no dropout: both runs are deterministic
        red = GradReducer(model.named_parameters(), comm=comm, always_exchange=exch)
        assert red.exchange == exch and red.backend == "rccl" and red.world == 1
        tr = Trainer(o, None, model, build_optimizer(model, o), None, None, grad_reducer=red)
        for _ in range(3):                                             # step 0 learns the arrivals, 1-2 launch from hooks / sinks
            tr.train_step(batch)
        torch.cuda.synchronize()
        if exch:
            assert red._comm_stream is not None and not red._comm_pending      # the exchange did run, and was joined
        finals.append({k: v.detach().clone() for k, v in model.state_dict().items()})
        red.remove_hooks()
    for k in finals[0]:
        assert torch.equal(finals[0][k], finals[1][k]), k
    # the whole step INCLUDING the RCCL exchange captured into one HIP graph: replays must reproduce the eager steps with the
    # same exchange bit for bit (3 warm-up steps + 3 replays against 6 eager steps)
    o, model, batch, Trainer, build_optimizer = _setup(dev, d, train_decoder_only=False)
    model.eval()
    red = GradReducer(model.named_parameters(), comm=comm, always_exchange=True)
    tr = Trainer(o, None, model, build_optimizer(model, o, capturable=True), None, None, grad_reducer=red)
    res = [tr.train_step_graphed(batch).clone() for _ in range(3)]
    torch.cuda.synchronize()
    graphed = {k: v.detach().clone() for k, v in model.state_dict().items()}
    red.remove_hooks()
    keep.append(tr)
    o, model, batch, Trainer, build_optimizer = _setup(dev, d, train_decoder_only=False)
    model.eval()
    red = GradReducer(model.named_parameters(), comm=comm, always_exchange=True)
    tr = Trainer(o, None, model, build_optimizer(model, o, capturable=True), None, None, grad_reducer=red)
    eager = [tr.train_step(batch) for _ in range(6)]
    torch.cuda.synchronize()
    red.remove_hooks()
    for k in graphed:
        assert torch.equal(graphed[k], model.state_dict()[k]), k
    assert float(res[-1][0]) == float(eager[-1][0])
    # exchanges that are not stream operations cannot be captured: refused loudly (gloo: host-side; c10d "nccl": its watchdog)
    for backend in ("gloo", "nccl"):
        o, model, batch, Trainer, build_optimizer = _setup(dev, d)
        red = GradReducer(model.named_parameters(), world=2)
        red.backend = backend
        tr = Trainer(o, None, model, build_optimizer(model, o, capturable=True), None, None, grad_reducer=red)
        with pytest.raises(RuntimeError, match="cannot be captured"):
            tr.train_step_graphed(batch)
        red.remove_hooks()
    del keep[:], tr
    import gc
    gc.collect()
    comm.destroy()
    _passed()


def test_gradient_exchange_on_c10d_nccl_group_eager():
    """The legacy transport: torch.distributed "nccl" (= RCCL) process group of one rank, eager steps only -- the in-place
    reduce_scatter + all_gather issued through c10d from the hooks leaves bit-identical parameters to the step without exchange."""
    _run_isolated("gradient_exchange_on_c10d_nccl_eager")


def _scenario_gradient_exchange_on_c10d_nccl_eager():
    import torch.distributed as dist
    from cvc.distributed import GradReducer
    dev = torch.device("cuda:0")
    d = synth.CONFIGS["tiny"]
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29653", rank=0, world_size=1, device_id=dev)
    finals = []
    for exch in (False, True):
        o, model, batch, Trainer, build_optimizer = _setup(dev, d, train_decoder_only=False)
        model.eval()
        red = GradReducer(model.named_parameters(), always_exchange=exch)
        assert red.exchange == exch and red.backend == "nccl" and red.comm is None
        tr = Trainer(o, None, model, build_optimizer(model, o), None, None, grad_reducer=red)
        for _ in range(3):
            tr.train_step(batch)
        torch.cuda.synchronize()
        finals.append({k: v.detach().clone() for k, v in model.state_dict().items()})
        red.remove_hooks()
    for k in finals[0]:
        assert torch.equal(finals[0][k], finals[1][k]), k
    import gc
    gc.collect()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    _passed()


# ------------------------------------------------------------------ train-mode (dropout ON) parity with dictated masks
def _dropout_masks(d, seed, p=0.5):
    """keep / (1 - p) masks for every dropout site of one cyclical pass (cvc/dropout.py names): the SAME tensors go to the
    product (through cvc.dropout.injected) and to the oracle (cyclical_forward(dropout=...))."""
    g = torch.Generator().manual_seed(seed)
    draw = lambda *shape: torch.bernoulli(torch.full(shape, 1.0 - p), generator=g) / (1.0 - p)
    return dict(emb_a=draw(d.B, d.T, d.E), emb_b=draw(d.B, d.T, d.E), emb_c=draw(d.B, d.T, d.E),
                out_a=draw(d.T, d.B, d.R), out_c=draw(d.T, d.B, d.R), vis_embed=draw(d.B, d.T, d.G))


def _train_mode_parity(d, seed, mix, loss_tol, grad_tol=5e-4, in_kernel=False, reducer=None):
    """in_kernel=False: masks drawn here and dictated to both sides.  in_kernel=True: the product generates its masks INSIDE the
    kernels (csrc/dropout_rng.h) and the oracle receives their host restatement (cvc.dropout.host_mask -> synth.dropout_keep).
    reducer: callable(model) -> GradReducer; the gradients then live in its arenas, are exchanged (finalize) before the check."""
    from helpers import build_model, to_dev, model_call
    from oracle import ref_cpu as O
    from cvc import dropout
    dev = torch.device("cuda:0")
    sd, f, b = synth.hot_path_state_dict(d, seed), synth.clip_features(d, seed), synth.label_glue_batch(d, seed)
    model = build_model(d, sd, dev).train()
    red = reducer(model) if reducer is not None else None
    if in_kernel:
        assert dropout.IN_KERNEL
        dropout.seed(seed * 7919 + 13)
        for _ in range(3):
            dropout.advance(dev)                                # not the first step of the generator
        step0 = int(dropout.rng_state(dev)[2].item())
        out = model_call(model, to_dev(f, dev), to_dev(b, dev), False)
        assert int(dropout.rng_state(dev)[2].item()) == step0 + 1            # one advance per training pass
        pe, pv, po = model.embed[2].p, model.roi_feat_extractor.vis_embed[2].p, model.decoder_core.dropout.p
        hm = lambda site, shape, p: dropout.host_mask(site, shape, p, dev)
        masks = dict(emb_a=hm("emb_a", (d.B, d.T, d.E), pe), emb_b=hm("emb_b", (d.B, d.T, d.E), pe), emb_c=hm("emb_c", (d.B, d.T, d.E), pe),
                     vis_embed=hm("vis_embed", (d.B, d.T, d.G), pv),
                     out_a=torch.stack([hm("out_a.%d" % t, (d.B, d.R), po) for t in range(d.T)]),
                     out_c=torch.stack([hm("out_c.%d" % t, (d.B, d.R), po) for t in range(d.T)]))
        for k, m in masks.items():
            frac = float((m == 0).float().mean())
            assert m.numel() < 4000 or 0.4 < frac < 0.6, (k, frac)      # p = 0.5 sites: about half of every mask is zero
        assert not torch.equal(masks["emb_a"], masks["emb_c"]) and not torch.equal(masks["out_a"][0], masks["out_a"][1])
    else:
        masks = _dropout_masks(d, seed + 1)
    P = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in O.to_torch(sd).items()}
    for k in list(P):                                           # the reconstructor shares the decoder's LSTM cells
        if k.startswith("attended_roi_decoder_core.") and "lstm" in k:
            P[k] = P[k.replace("attended_roi_decoder_core.", "decoder_core.")]
    ref = O.cyclical_forward(P, O.to_torch(f), O.to_torch(b), T=d.T, vocab_size=d.V, dropout=masks)
    if d.B * d.T <= 256:      # (small configs only: at full size the second oracle pass costs 10 s of host time for the same sanity check)
        ref_eval = O.cyclical_forward(P, O.to_torch(f), O.to_torch(b), T=d.T, vocab_size=d.V)
        assert abs(float(ref[0].detach()) - float(ref_eval[0].detach())) > 1e-3          # the masks do change the pass (not an eval-mode rerun)
    O.training_loss(ref, xe_loss_weight=mix[0], w_att2=mix[1], w_cls=0.0, caption_consistency_loss_weight=mix[2]).backward()
    used = []

    def source(site, shape):
        used.append(site)
        if site.startswith("emb_") or site == "vis_embed":
            return masks[site].reshape(shape)
        kind, t = site.split(".")
        return masks[kind][int(t)]
    if not in_kernel:
        with dropout.injected(source):
            out = model_call(model, to_dev(f, dev), to_dev(b, dev), False)
        assert sorted(set(used)) == sorted(["emb_a", "emb_b", "emb_c", "vis_embed"] + ["out_a.%d" % t for t in range(d.T)] + ["out_c.%d" % t for t in range(d.T)])
    for got, want in zip(out, ref):
        assert float(got.detach().mean()) == pytest.approx(float(want.detach().mean()), rel=loss_tol, abs=loss_tol / 10)
    lm, a2, _g, _cls, rec = [x.mean() for x in out]
    if red is not None:
        red.zero_grad()
    (mix[0] * lm + mix[1] * a2 + mix[2] * rec).backward()
    if red is not None:
        red.finalize(average=False)                 # the exchange (one rank: sums over one rank) -- in place on the arenas
        torch.cuda.synchronize()
    checked = 0
    for n, p in model.named_parameters():
        if n.startswith("roi_feat_extractor") or n not in P:
            continue
        if P[n].grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        if red is not None:
            assert p.grad.data_ptr() == red._views[id(p)].data_ptr(), n          # still the arena view after the exchange
        want = P[n].grad.double()
        err = float((p.grad.cpu().double() - want).norm())
        assert err <= grad_tol * float(want.norm()) + 1e-6, (n, err, float(want.norm()))
        checked += 1
    assert checked >= 15
    return model, red


def test_train_mode_cyclical_pass_cfg4_share_on_rccl_arenas_vs_oracle():
    """BASELINE config 4 as far as one GPU can prove it: ONE rank's share of the 8-GPU job (B = 32 clips, D = 2048, T = 20), train
    mode with the masks generated in the kernels, the gradients living in the GradReducer's flat arenas (486 MB at this size) and
    exchanged over RCCL (the package's own communicator, one rank: in-place reduce_scatter + all_gather per bucket, the branch
    every rank of the 8-GPU job runs) -- five losses and every parameter gradient against the oracle's autograd."""
    _run_isolated("cfg4_share_on_rccl_arenas")


def _scenario_cfg4_share_on_rccl_arenas():
    from cvc.comm import RcclComm
    from cvc.distributed import GradReducer
    comm = RcclComm.single()
    mk = lambda model: GradReducer(model.named_parameters(), comm=comm, always_exchange=True)
    model, red = _train_mode_parity(synth.CONFIGS["cfg4"], 1404, (0.5, 0.0, 0.5), loss_tol=1e-4, in_kernel=True, reducer=mk)
    assert red.backend == "rccl" and red.exchange and red._comm_stream is not None
    total = sum(a.numel() * 4 for a in red.arenas)
    assert total > 480e6, total                                        # the real message: every trainable tensor of the hot path
    assert all(a.numel() % red.world == 0 for a in red.arenas)        # equal shards: the reduce_scatter + all_gather branch
    red.remove_hooks()
    del model, red
    comm.destroy()
    _passed()


@pytest.mark.parametrize("cfg,mix", [("tiny", (0.5, 0.05, 0.5)), ("tiny", (0.5, 0.0, 0.5)), ("cfg1", (0.5, 0.0, 0.5))])
def test_train_mode_cyclical_pass_with_dictated_dropout_masks_vs_oracle(cfg, mix):
    """The whole cyclical pass in train() mode -- Dropout(0.5) on the three word embeddings (captioner.py:53-68) and on the
    language LSTM's output of every step of loops A and C (decoder_core.py:62, 109) -- with the same keep-masks on both sides:
    five losses and every parameter gradient against the oracle's autograd.  (The timed training bench runs in train mode.)"""
    _train_mode_parity(synth.CONFIGS[cfg], 4242, mix, loss_tol=2e-5)


def test_train_mode_cyclical_pass_cfg3_full_size_vs_oracle():
    """... and once at BASELINE config 3 size (B=64, D=2048, T=20), with the masks generated inside the kernels."""
    _train_mode_parity(synth.CONFIGS["cfg3"], 1305, (0.5, 0.0, 0.5), loss_tol=1e-4, in_kernel=True)


@pytest.mark.parametrize("cfg,mix", [("tiny", (0.5, 0.05, 0.5)), ("cfg1", (0.5, 0.0, 0.5))])
def test_train_mode_cyclical_pass_with_in_kernel_dropout_vs_oracle(cfg, mix):
    """Train mode as it ships: every nn.Dropout of the pass takes its keep-mask from the counter-based generator inside the
    consuming kernel (embedding lookups, the language cell's epilogue and its gate-gradient kernel); the oracle gets the same
    masks from the generator's host restatement.  Five losses + every parameter gradient."""
    _train_mode_parity(synth.CONFIGS[cfg], 977, mix, loss_tol=2e-5, in_kernel=True)


def test_in_kernel_dropout_kernels_equal_host_restatement():
    """cvc_dropout_rng, cvc_embed_relu_rng_fwd/_bwd, cvc_packed_lstm_train_drop_fwd and cvc_lstm_pointwise_bwd3_drop against
    synth.dropout_keep: the multipliers are exactly 0 or 1 / (1 - p) at exactly the restated positions; masks change with step
    and site; p = 0.3 and 0.5."""
    from cvc import dropout, hip, functional as F_
    dev = torch.device("cuda:0")
    dropout.seed(0xDEADBEEF12345)
    st = dropout.rng_state(dev)
    for p in (0.3, 0.5):
        x = torch.randn(37, 129, device=dev)                                      # n % 4 != 0: the tail path
        xp = torch.randn(40, 128, device=dev)
        for site in ("emb_b", "out_a.3", "out_c.19"):
            for xx in (xp, x.reshape(-1)[: x.numel() // 4 * 4 + 1].clone()):
                y = hip.dropout_rng(xx, st, dropout.site_id(site), p)
                m = dropout.host_mask(site, xx.shape, p, dev).to(dev)
                assert torch.equal(y, xx * m)
        m0 = dropout.host_mask("emb_a", (64, 64), p, dev)
        dropout.advance(dev)
        m1 = dropout.host_mask("emb_a", (64, 64), p, dev)
        assert not torch.equal(m0, m1) and not torch.equal(m1, dropout.host_mask("emb_c", (64, 64), p, dev))
        assert abs(float((m1 == 0).float().mean()) - p) < 0.03
        # embedding: forward and backward
        V, E, M = 50, 64, 96
        table = torch.randn(V, E, device=dev, requires_grad=True)
        idx = torch.randint(0, V, (M,), device=dev)
        out = F_.embed_relu(table, idx, rng=(st, dropout.site_id("emb_c"), p))
        mk = dropout.host_mask("emb_c", (M, E), p, dev).to(dev)
        assert torch.equal(out, torch.relu(table.detach()[idx]) * mk)
        g = torch.randn(M, E, device=dev)
        out.backward(g)
        t2 = table.detach().clone().requires_grad_(True)
        (torch.relu(t2[idx]) * mk).backward(g)
        assert torch.allclose(table.grad, t2.grad, rtol=1e-5, atol=1e-5)
        # language cell with the fused output dropout: the dropped copy is h' times the mask, its gradient takes the mask
        R, B = 64, 8
        w_ih = torch.randn(4 * R, 2 * R, device=dev, requires_grad=True); w_hh = torch.randn(4 * R, R, device=dev, requires_grad=True)
        b_ih = torch.randn(4 * R, device=dev, requires_grad=True); b_hh = torch.randn(4 * R, device=dev, requires_grad=True)
        with torch.no_grad():
            w_ih *= 0.1; w_hh *= 0.1
        xs = [torch.randn(B, R, device=dev, requires_grad=True) for _ in range(2)]
        h0 = torch.randn(B, R, device=dev, requires_grad=True); c0 = torch.randn(B, R, device=dev, requires_grad=True)
        for copies in (1, 2):
            outs = F_.lstm_cell(xs, h0, c0, w_ih, w_hh, b_ih, b_hh, copies=copies, drop=(st, dropout.site_id("out_a.7"), p))
            assert len(outs) == copies + 2
            *hs, hd, c1 = outs
            mk = dropout.host_mask("out_a.7", (B, R), p, dev).to(dev)
            assert torch.equal(hd, hs[0] * mk) and all(torch.equal(h, hs[0]) for h in hs)
            gd, gh, gc = torch.randn(B, R, device=dev), torch.randn(B, R, device=dev), torch.randn(B, R, device=dev)
            ins = [w_ih, w_hh, b_ih, b_hh, h0, c0, *xs]
            got = torch.autograd.grad([hd, hs[-1], c1], ins, [gd, gh, gc])
            *hs2, c2 = F_.lstm_cell(xs, h0, c0, w_ih, w_hh, b_ih, b_hh, copies=1)
            want = torch.autograd.grad([hs2[0], c2], ins, [gd * mk + gh, gc])
            for a_, b_ in zip(got, want):
                assert torch.allclose(a_, b_, rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------ optimizer step on the HIP kernels
@pytest.mark.parametrize("weight_decay,max_norm", [(0.0, 0.5), (0.01, 0.5), (0.0, 1e6)])
def test_clip_adam_equals_clip_grad_norm_plus_torch_adam(weight_decay, max_norm):
    """cvc.optim.ClipAdam.clip_and_step (csrc/optim.hip: norm partials, coefficient + step counters, one pass over p / g / m / v)
    against nn.utils.clip_grad_norm_ + torch.optim.Adam.step() (trainer.py:116-122) over six steps: parameters, both moments,
    step counts, the clipped gradients left in .grad, the norm; per-group learning rates (main.py:171-191), odd sizes, a
    parameter that never gets a gradient, a 4-byte-aligned view; and the state_dict layout torch.optim.Adam loads."""
    from cvc.optim import ClipAdam
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    big = torch.randn(3 * 65536 + 7, generator=g)
    shapes = [(257, 129), (65536 * 2,), (5,), (1,), (64, 1024)]
    base = [torch.randn(*s, generator=g) for s in shapes]
    arena = torch.randn(1000 + 1, generator=g)

    def make():
        ps = [torch.nn.Parameter(b.clone().to(dev)) for b in base]
        ps.append(torch.nn.Parameter(big.clone().to(dev)))
        a = arena.clone().to(dev)
        ps.append(torch.nn.Parameter(a[1:]))                       # 4-byte aligned storage offset: the scalar path of the kernels
        dead = torch.nn.Parameter(torch.ones(8, device=dev))       # never receives a gradient
        groups = [{"params": [p], "lr": 1e-3 * (0.1 if i % 3 == 0 else 1.0), "weight_decay": weight_decay, "betas": (0.8, 0.99)}
                  for i, p in enumerate(ps + [dead])]
        return ps, dead, groups
    pa, dead_a, ga = make()
    pb, dead_b, gb = make()
    ref = torch.optim.Adam(ga)
    opt = ClipAdam(gb)
    for it in range(6):
        grads = [torch.randn(p.shape, generator=g) * (3.0 if it % 2 else 0.01) for p in pa]
        for p, q, gr in zip(pa, pb, grads):
            p.grad = gr.clone().to(dev)
            q.grad = gr.clone().to(dev)
        n_ref = torch.nn.utils.clip_grad_norm_(pa + [dead_a], max_norm)
        ref.step()
        n_got = opt.clip_and_step(max_norm, 1.0)
        assert float(n_got) == pytest.approx(float(n_ref), rel=2e-6)
        for p, q in zip(pa, pb):
            assert torch.allclose(q, p, rtol=2e-6, atol=1e-6), it
            assert torch.allclose(q.grad, p.grad, rtol=2e-6, atol=1e-9)                       # the clipped gradient, written back
            sa, sb = ref.state[p], opt.state[q]
            # (fp32 rounding of m + (g - m) (1 - beta1) under cancellation: absolute, at the scale of the gradients)
            assert torch.allclose(sb["exp_avg"], sa["exp_avg"], rtol=2e-6, atol=2e-6)
            assert torch.allclose(sb["exp_avg_sq"], sa["exp_avg_sq"], rtol=5e-6, atol=1e-8)
            assert float(sb["step"]) == float(sa["step"]) == it + 1
    assert torch.equal(dead_b, torch.ones(8, device=dev)) and len(opt.state[dead_b]) == 0           # skipped, as torch skips it
    # gradients that are sums over 4 ranks: the 1 / G is folded into the coefficient
    pc, _dc, gc = make()
    pd, _dd, gd = make()
    o1, o2 = ClipAdam(gc), ClipAdam(gd)
    grads = [torch.randn(p.shape, generator=g) for p in pc]
    for p, q, gr in zip(pc, pd, grads):
        p.grad = (gr / 4).to(dev); q.grad = gr.clone().to(dev)
    o1.clip_and_step(max_norm, 1.0); o2.clip_and_step(max_norm, 0.25)
    for p, q in zip(pc, pd):
        assert torch.allclose(q, p, rtol=2e-6, atol=2e-7) and torch.allclose(q.grad, p.grad, rtol=2e-6, atol=1e-9)
    # a torch.optim.Adam resumes from ClipAdam's state_dict and the other way round
    sd = opt.state_dict()
    ref2 = torch.optim.Adam(make()[2], capturable=True)
    ref2.load_state_dict(sd)
    opt2 = ClipAdam(make()[2])
    opt2.load_state_dict(ref.state_dict())
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}
    # step() alone = the unclipped update
    pe, _de, ge = make()
    pf, _df, gf = make()
    r3, o3 = torch.optim.Adam(ge), ClipAdam(gf)
    for p, q, gr in zip(pe, pf, grads):
        p.grad = gr.clone().to(dev); q.grad = gr.clone().to(dev)
    r3.step(); o3.step()
    for p, q in zip(pe, pf):
        assert torch.allclose(q, p, rtol=2e-6, atol=2e-7) and torch.equal(q.grad, p.grad)


@pytest.mark.parametrize("cfg,B,train", [("tiny", None, False), ("tiny", None, True), ("cfg1", 5, True), ("tiny", 64, True),
                                         ("tiny", 40, False), ("tiny", 33, True)])
def test_joint_backward_of_both_loops_equals_the_two_passes(cfg, B, train):
    """cvc_train_loops_bwd_joint (3 backward-data products per step instead of 5, loop A's outputs passed through loop C's graph
    node) against cvc_train_loop_bwd twice -- both loops' rows in one 64-row operand (2B <= 64), or as two 64-row operand groups on
    the 128-row form of the product (B = 33, 40, 64: config 3's case): same kernels row for row under the same K split (which these
    sizes have), so the five losses and every parameter gradient must be bit-identical (all four loss terms weighted)."""
    import dataclasses
    from helpers import build_model, to_dev, model_call
    from cvc import train_loops, dropout
    dev = torch.device("cuda:0")
    d = synth.CONFIGS[cfg] if B is None else dataclasses.replace(synth.CONFIGS[cfg], B=B)
    assert 2 * d.B <= 128
    model = build_model(d, synth.hot_path_state_dict(d, 7), dev)
    if train:
        model.train()
    f, b = to_dev(synth.clip_features(d, 7, full_mask_clip=1 if d.B > 1 else None), dev), to_dev(synth.label_glue_batch(d, 7), dev)
    res = {}
    keep = train_loops.JOINT_BWD
    try:
        for on in (False, True):
            train_loops.JOINT_BWD = on
            dropout.seed(123)
            for p in model.parameters():
                p.grad = None
            ls = model_call(model, f, b, False)
            (0.5 * ls[0].mean() + 0.3 * ls[1].mean() + 0.2 * ls[3].mean() + 0.5 * ls[4].mean()).backward()
            torch.cuda.synchronize()
            res[on] = ([x.detach().clone() for x in ls], {k: (None if p.grad is None else p.grad.clone()) for k, p in model.named_parameters()})
    finally:
        train_loops.JOINT_BWD = keep
    for a, r in zip(res[True][0], res[False][0]):
        assert torch.equal(a, r)
    n = 0
    for k, g in res[False][1].items():
        g2 = res[True][1][k]
        assert (g is None) == (g2 is None), k
        if g is not None:
            assert torch.equal(g, g2), k
            n += 1
    assert n >= 15


# ------------------------------------------------------------------ Trainer.train(): one captured graph per bucketed batch shape
def _ragged_batches(dev, d, seed, counts):
    """batches in the 12-tuple contract whose clips carry fewer proposals / boxes than the loader's padded lengths: what lies beyond a
    clip's count is the loader's padding (zero feature rows, mask bits set -- dataloader_anet.py:376-388)"""
    from helpers import to_dev
    out = []
    for j, (n_max, k_max) in enumerate(counts):
        f = synth.clip_features(d, seed + j)
        b = synth.label_glue_batch(d, seed + j)
        rng = np.random.default_rng(seed + j)
        n_b = rng.integers(max(1, n_max // 2), n_max + 1, size=d.B); n_b[0] = n_max
        k_b = rng.integers(1, k_max + 1, size=d.B); k_b[-1] = k_max
        for c in range(d.B):
            for key in ("pool_feats", "p_pool_feats", "g_pool_feats"):
                f[key][c, n_b[c]:] = 0
            f["pnt_mask"][c, 1 + n_b[c]:] = True
            b["box_mask"][c, 0, k_b[c]:, :] = True
            b["num"][c, 1], b["num"][c, 2] = n_b[c], k_b[c]
        f, b = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in f.items()}, {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in b.items()}
        out.append((f, b["input_seq"], b["gt_seq"], b["num"], b["proposals"], b["gt_bboxs"], b["box_mask"],
                    ["v_x_segment_%02d" % i for i in range(d.B)], torch.zeros(d.B, d.N, 1), b["frm_mask"], b["sample_idx"],
                    f["pnt_mask"][:, 1:].clone()))
    return out


def test_trainer_train_replays_one_graph_per_bucketed_shape_bit_equal_to_eager(capsys):
    """Trainer.train() (the drop-in for the reference's trainer.py:39-150) on batches whose per-batch trimming (trainer.py:63-69)
    gives three different shapes: lengths are rounded up to shape buckets, a shape's first occurrence runs eagerly, from its second
    occurrence on the step is captured once and replayed -- parameters after the epoch and every displayed loss bit-equal to the
    same epoch run with eager steps only; the losses are read back once per display interval."""
    import dataclasses
    from cvc import dropout
    from cvc.trainer import bucket_len
    dev = torch.device("cuda:0")
    d = dataclasses.replace(synth.CONFIGS["tiny"], B=4, N=40, K=8)
    counts = [(40, 8), (17, 3), (9, 5)]
    assert [bucket_len(n, 40) for n, _ in counts] == [40, 20, 10] and [bucket_len(k, 8) for _, k in counts] == [8, 4, 6]
    batches = _ragged_batches(dev, d, 77, counts) * 3 + _ragged_batches(dev, d, 77, counts)[:1]     # train() drops the last batch
    finals, shown, stats = [], [], []
    for graphed in (False, True):
        o, model, _batch, Trainer, build_optimizer = _setup(dev, d, train_decoder_only=False)
        o.disp_interval, o.hip_graph = 4, 1
        tr = Trainer(o, None, model, build_optimizer(model, o, capturable=True), batches, None)
        assert tr.graph_capable()
        tr._graph_broken = not graphed                       # eager run: same bucketed shapes, no capture
        dropout.seed(4711)
        tr.train(0)
        torch.cuda.synchronize()
        finals.append({k: v.detach().clone() for k, v in model.state_dict().items()})
        shown.append([ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("Epoch")])
        stats.append(dict(tr.graph_stats))
        shapes = sorted({(k[0], k[1]) for key in tr._shape_seen for k in key if k[0] == "ppls"})
        assert [s_[1][1] for s_ in shapes] == [10, 20, 40], shapes
    assert stats[0] == dict(eager=9, replayed=0, captured=0) and stats[1] == dict(eager=3, replayed=6, captured=3), stats
    assert len(shown[0]) == 3                                  # steps 0, 4, 8
    strip = lambda ln: ln.split("LM Loss")[1]                   # (wall-clock columns differ)
    assert [strip(x) for x in shown[0]] == [strip(x) for x in shown[1]], (shown[0], shown[1])
    for k in finals[0]:
        assert torch.equal(finals[0][k], finals[1][k]), k


# ------------------------------------------------------------------ two REAL rank processes on the one GPU of the box
def _two_rank_env(tmp_path):
    """environment of a 2-process run on ONE GPU: the stand-in librccl (tests/stub_rccl: device buffers exchanged through the host and
    POSIX shared memory -- real RCCL refuses two ranks on one device), both ranks on cuda:0"""
    import os
    import subprocess
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stub_rccl", "stub_rccl.c")
    out = os.path.join(str(tmp_path), "librccl.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", out, src, "-lpthread", "-lrt", "-ldl"])
    env = dict(os.environ, CVC_RCCL_LIB=out, CVC_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", CVC_STUB_TRACE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _took_the_device_path(stderr):
    """the stand-in's call trace: every collective of the run was handed DEVICE buffers and was staged on the caller's stream (a device
    pointer mistaken for a host pointer would be read by the CPU through the PCIe aperture, unordered with the stream)"""
    calls = [l for l in stderr.splitlines() if l.startswith("[stub_rccl] rank")]
    return len(calls) > 0 and all(l.endswith(" device") for l in calls)


def _launch_two(script_args, env, timeout=600):
    import os
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=timeout)


def test_bench_train_finishes_with_two_real_rank_processes(tmp_path):
    """`bench.py --gpus 2 --mode train` as the driver launches it (torch.distributed.run, two processes, gloo control plane, the
    package's communicator for the exchange) -- both ranks on this box's one GPU, the transport a stand-in librccl.  What round 5's
    bench could not do at N > 1: FINISH.  Every pass of bench_train.TrainBench runs on both ranks (warm-up agreed over the control
    plane, timed steps, eager comparison, exchange probe, bucket and role probes with the exchange on), the line says n_gpus = 2,
    ranks_joined = 2 counted through the communicator, and carries the exchange's cost.  Eager steps: the stand-in's collectives are
    synchronous host round trips and cannot be captured."""
    import json
    import os
    env = _two_rank_env(tmp_path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = _launch_two([os.path.join(root, "bench.py"), "--gpus", "2", "--mode", "train", "--config", "tiny", "--no-train-graph", "--steps", "3",
                     "--warmup", "1", "--min-warm-seconds", "0.2", "--no-cpu-baseline"], env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert _took_the_device_path(r.stderr), r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_joined"] == 2 and d["ranks_control_plane"] == 2
    ex = d["exchange"]
    assert ex["ranks"] == 2 and ex["backend"] == "rccl" and ex["buckets"] >= 4 and ex["ms_per_step_with_exchange"] > 0
    assert all(b["launched"] for b in d["gradient_buckets"]["per_bucket"])
    assert d["config"]["global_batch"] == 2 * d["config"]["B_per_gpu"] and len(d["kernels"]) > 10


def test_bench_at_two_ranks_still_prints_its_decode_line_when_the_training_secondary_hangs(tmp_path):
    """`bench.py --gpus 2` (the driver's command: decode line + the all-ranks config-4 training secondary, the only part of the run
    with collectives).  The secondary is made to hang (test hook): after --secondary-timeout every rank dumps its stacks and leaves
    with status 0, rank 0 having printed the decode line -- complete, n_gpus = 2 -- with the secondary's entry saying it timed out."""
    import json
    import os
    env = dict(_two_rank_env(tmp_path), CVC_BENCH_TEST_HANG_SECONDARY="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = _launch_two([os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--min-warm-seconds", "0.1",
                     "--no-cpu-baseline", "--secondary-timeout", "5"], env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_joined"] == 2 and d["value"] > 0 and d["steps"] == 3
    assert len(d["secondary"]) == 1 and "did not return within" in d["secondary"][0]["error"]
    assert d["summary"][-1]["error"] == d["secondary"][0]["error"]
    assert "run_secondary_ranks" in r.stderr                       # the stacks name the place


def test_two_rank_training_step_equals_the_single_process_step_on_the_whole_batch(tmp_path):
    """Two rank processes (one GPU, stand-in transport), each with its half of a batch: one eager training step with
    GradReducer(comm = RcclComm.from_process_group()) + ClipAdam(1 / G folded into the clip) must leave both ranks with bitwise EQUAL
    parameters, equal (fp32 summation noise) to ONE process stepping on the mean of the two shards' losses -- the semantics of the
    reference's nn.DataParallel (main.py:169: per-replica token-mean losses, unweighted mean over replicas, trainer.py:101-122)."""
    import os
    env = _two_rank_env(tmp_path)
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "two_rank_step.py")
    r = _launch_two([script, str(tmp_path)], env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert _took_the_device_path(r.stderr), r.stderr[-2000:]
    assert r.stdout.count("TWO-RANK-STEP-OK") == 2, r.stdout[-2000:]


# ------------------------------------------------------------------ raw features through the encoder: captured steps, deferred error words
def _raw_setup(dev, n_clips=28, bs=4, H2=256, seed=3, F=6):
    """model with the once-per-clip encoder in front (cvc.model.create_model.build_model on raw synthetic features), its loader
    batches and options -- what cvc.main builds for --synthetic_raw; rnn_size 2 * H2 so that the GRU's persistent forms apply
    (forward H % 128 == 0, backward H % 256 == 0)"""
    from cvc import opts as cvc_opts
    from cvc.data_synth import SyntheticCaptionDataset, collate
    from cvc.model.create_model import build_model
    from cvc.trainer import Trainer, build_optimizer
    o = cvc_opts.build_parser().parse_args(["--batch_size", str(bs), "--num_prop_per_frm", "7", "--t_attn_size", str(F), "--rnn_size", str(2 * H2),
                                            "--att_hid_size", "64", "--input_encoding_size", "32", "--seq_length", "4",
                                            "--vis_encoding_size", "24", "--att_feat_size", "24", "--learning_rate", "0.001"])
    o.test_mode = False
    dims = synth.Dims(B=bs, N=7, F=F, R=2 * H2, A=64, E=32, T=4, G=24, K=7)
    full = SyntheticCaptionDataset(dims, n_clips, seed, "training", raw=True)
    o.vocab_size, o.itow, o.wtoi, o.itod, o.detect_size = full.vocab_size, full.itow, full.wtoi, full.itod, dims.DET
    o.glove_clss, o.glove_vg_cls = torch.from_numpy(full.glove_clss), torch.from_numpy(full.glove_vg_cls)
    o.vg_cls, o.detectron_tables = full.vg_cls, full.tables
    o.disp_interval, o.hip_graph = 3, 1
    torch.manual_seed(seed)
    model = build_model(o, dev)
    batches = [collate([full[i] for i in range(j, j + bs)]) for j in range(0, n_clips, bs)]
    return o, model, batches, Trainer, build_optimizer


def test_raw_feature_training_replays_captured_steps_bit_equal_to_eager(capsys):
    """Round-5 review, missing 3: the reference's own training configuration (raw frame + region features through the encoder
    every step, main.py:216-228 -> trainer.py:39-150 -> model/backbone.py:298-351) trained eagerly because the persistent GRU
    kernels' error words were read by the host.  They now stay on the device (OR-ed into the step's status word, read back with
    the losses once per display interval): Trainer.train() captures and replays the step, bit-equal to the same epoch of eager
    steps -- parameters, BatchNorm running statistics, every displayed loss."""
    from cvc import dropout, gru
    dev = torch.device("cuda:0")
    finals, shown, stats, forms = [], [], [], []
    for graphed in (False, True):
        o, model, batches, Trainer, build_optimizer = _raw_setup(dev)
        tr = Trainer(o, None, model, build_optimizer(model, o), batches, None)
        assert tr.graph_capable() and model.reports_error_words() and model.step_capturable(deferred_errors=True)
        assert not model.step_capturable(deferred_errors=False)
        tr._graph_broken = not graphed
        dropout.seed(99)
        gru.last_train_form = gru.last_bwd_form = None
        tr.train(0)
        torch.cuda.synchronize()
        finals.append({k: v.detach().clone() for k, v in model.state_dict().items()})
        shown.append([ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("Epoch")])
        stats.append(dict(tr.graph_stats))
        forms.append((gru.last_train_form, gru.last_bwd_form))
        assert tr.deferred_stats == dict(void_steps=0, rerun_steps=0)
    assert stats[0] == dict(eager=6, replayed=0, captured=0) and stats[1] == dict(eager=2, replayed=4, captured=1), stats
    assert forms[0] == ("persistent", "persistent"), forms          # the persistent recurrences ran (and were captured)
    strip = lambda ln: ln.split("LM Loss")[1]
    assert len(shown[0]) == 2 and [strip(x) for x in shown[0]] == [strip(x) for x in shown[1]], (shown[0], shown[1])
    for k in finals[0]:
        assert torch.equal(finals[0][k], finals[1][k]), k


def test_void_step_leaves_parameters_untouched_and_is_rerun():
    """Deferred error words end to end: the status word set before a replayed step (what a persistent launch does on a barrier
    time-out) voids THAT step on the device -- parameters, Adam moments and step counts as before, its losses out of the sums --
    and the interval's read-back re-runs it on the per-step forms: the epoch ends with every batch trained exactly once."""
    from cvc import dropout, gru, hip
    dev = torch.device("cuda:0")
    o, model, batches, Trainer, build_optimizer = _raw_setup(dev, n_clips=32)
    opt = build_optimizer(model, o)
    tr = Trainer(o, None, model, opt, batches, None)
    dropout.seed(5)
    # steps 0, 1 eager, 2 capture + replay, 3 .. 6 replay; step 4 is voided by hand
    orig = tr.train_step_bucketed
    seen, snap = [], {}

    def spy(b):
        k = len(seen)
        if k == 4:
            torch.cuda.synchronize()
            snap["before"] = {n: p.detach().clone() for n, p in model.named_parameters()}
            snap["steps"] = [float(opt.state[p]["step"]) for p in model.parameters() if p in opt.state]
            hip.step_status(dev).fill_(1)
        res = orig(b)
        if k == 4:
            torch.cuda.synchronize()
            snap["after"] = {n: p.detach().clone() for n, p in model.named_parameters()}
            snap["steps_after"] = [float(opt.state[p]["step"]) for p in model.parameters() if p in opt.state]
            snap["res"] = res.clone()
            snap["reruns_so_far"] = tr.deferred_stats["rerun_steps"]
        seen.append(k)
        return res
    tr.train_step_bucketed = spy
    gru.last_train_form = None
    tr.train(0)
    torch.cuda.synchronize()
    assert len(seen) == 7
    for n in snap["before"]:
        assert torch.equal(snap["before"][n], snap["after"][n]), n                 # the void step changed no parameter
    assert snap["steps"] == snap["steps_after"] and set(snap["steps"]) == {4.0}   # ... and no step count
    assert snap["res"].tolist() == [0.0, 0.0, 0.0, 0.0, 0.0, 1.0] and snap["reruns_so_far"] == 0
    assert tr.deferred_stats == dict(void_steps=1, rerun_steps=1)                  # read back at the interval's end, re-run once
    assert {float(opt.state[p]["step"]) for p in model.parameters() if p in opt.state} == {7.0}      # 7 batches, 7 updates
    assert gru.PERSISTENT and gru.BWD_PERSISTENT and not hip.errors_deferred()     # switches restored
    assert int(hip.step_status(dev)) == 0


# ------------------------------------------------------------------ rank symmetry: every rank issues the same collectives
def test_bench_train_issues_identical_collectives_as_rank_0_and_as_rank_1():
    """bench.py --mode train at N > 1 (round-5 review: rank 0 alone ran probe steps with the exchange on -- six collectives per step
    the other ranks never joined -- and the run could not finish).  run_train() is run here twice in one process, as rank 0 and as
    rank 1 of a RECORDING world-2 communicator (tests/helpers.py::RecordingComm: same interface as cvc.comm.RcclComm, no
    transport): the two logs -- Python-level calls with the bench section they belong to, and the collectives that executed on the
    device, graph replays included -- must be identical call for call."""
    import bench
    import bench_train
    from helpers import RecordingComm
    dev = torch.device("cuda:0")
    d = synth.CONFIGS["tiny"]
    args = bench.parse(["--mode", "train", "--config", "tiny", "--steps", "3", "--warmup", "1", "--min-warm-seconds", "0.0", "--no-cpu-baseline"])
    logs = []
    for rank in (0, 1):
        comm = RecordingComm(2, rank, dev)
        line = bench_train.run_train(args, d, dev, rank, 2, comm=comm, always_exchange=True, cpu_baseline=False, config_name="tiny")
        assert (line is not None) == (rank == 0)
        logs.append((list(comm.host_log), comm.device_log()))
        if rank == 0:
            assert line["n_gpus"] == 2 and line["exchange"]["ranks"] == 2 and line["exchange"]["buckets"] >= 4
            assert all(b["launched"] for b in line["gradient_buckets"]["per_bucket"])
            assert {k["kernel"] for k in line["kernels"]} >= {"cvc_tile_gemm", "cvc_adam_clip_step", "loopA.fwd.lang_cell"}
    (h0, d0), (h1, d1) = logs
    assert h0 == h1, [(a, b) for a, b in zip(h0, h1) if a != b][:5]
    assert d0 == d1 and len(d0) > len(h0)                    # (replays execute collectives no Python call stands for)
    sections = [s_ for _, s_ in h0]
    # every pass that launches eager steps (or captures) reached the communicator from Python -- on both ranks alike; the timed
    # region and the exchange probe's exchange-on steps are graph replays (device log only)
    assert {"warm", "eager", "bucket_probe", "role_probe"} <= set(sections), sorted(set(sections))


def test_trainer_train_issues_identical_collectives_whatever_each_ranks_batch_shapes_are():
    """cvc.main's epoch loop (Trainer.train: the replacement of reference main.py:169 + trainer.py:39-150) with DIFFERENT batch shapes
    per rank: rank 0 meets a shape for the first time (eager step) while rank 1 replays a captured one, and so on.  Both must put
    the same sequence of collectives on their exchange streams -- same count per step, same bucket sizes, same order."""
    import dataclasses
    from cvc import dropout
    from cvc.distributed import GradReducer
    from helpers import RecordingComm
    dev = torch.device("cuda:0")
    d = dataclasses.replace(synth.CONFIGS["tiny"], B=4, N=40, K=8)
    shapes = [(40, 8), (17, 3), (9, 5)]
    order = {0: [0, 1, 2, 0, 1, 2, 0, 1, 2, 0], 1: [2, 2, 0, 2, 0, 1, 1, 1, 0, 2]}     # (train() drops the last batch)
    dev_logs, host_counts, stats = [], [], []
    for rank in (0, 1):
        pool = _ragged_batches(dev, d, 77 + rank, shapes)
        batches = [pool[i] for i in order[rank]]
        o, model, _batch, Trainer, build_optimizer = _setup(dev, d, train_decoder_only=False)
        o.disp_interval, o.hip_graph = 4, 1
        comm = RecordingComm(2, rank, dev)
        red = GradReducer(model.named_parameters(), comm=comm)
        tr = Trainer(o, None, model, build_optimizer(model, o, capturable=True), batches, None, grad_reducer=red)
        assert tr.graph_capable() and red.exchange and red.world == 2
        dropout.seed(4711)
        tr.train(0)
        dev_logs.append(comm.device_log())
        host_counts.append(len(comm.host_log))
        stats.append(dict(tr.graph_stats))
        red.remove_hooks()
    assert stats[0] == dict(eager=3, replayed=6, captured=3), stats
    assert stats[1] == dict(eager=4, replayed=5, captured=3), stats       # (its first two steps are eager whatever their shape)
    assert dev_logs[0] == dev_logs[1], (len(dev_logs[0]), len(dev_logs[1]))
    nb = len(red.arenas)
    assert len(dev_logs[0]) == 9 * nb                         # one exchange of every bucket per step, eager or replayed
    for k in range(1, 9):                                     # steps 1.. (compacted arenas): the same bucket sizes in the same order
        assert dev_logs[0][k * nb:(k + 1) * nb] == dev_logs[0][nb:2 * nb], k


def test_bucketed_and_exactly_trimmed_steps_agree_on_losses_and_gradients():
    """Shape buckets pad a batch's region / box axes with the loader's own padding (zero rows, mask bits set): one step with
    buckets = 0 (the reference's trimming, trainer.py:63-69) and one with buckets = 4 on a batch whose per-clip counts differ must
    give the same losses and gradients within fp32 summation noise (round-5 advisor: only graphed-vs-eager under the SAME bucketing
    was tested)."""
    import dataclasses
    from cvc import dropout
    dev = torch.device("cuda:0")
    d = dataclasses.replace(synth.CONFIGS["tiny"], B=4, N=40, K=8)
    batch = _ragged_batches(dev, d, 91, [(17, 3)])[0]
    got = []
    for buckets in (0, 4):
        o, model, _b, Trainer, build_optimizer = _setup(dev, d, train_decoder_only=False)
        model.eval()                                          # no dropout: the masks' index space follows the padded shapes
        tr = Trainer(o, None, model, build_optimizer(model, o), None, None)
        tr._active_buckets = buckets
        b = tr._prepare(batch, True)
        assert b["ppls"].shape[1] == (20 if buckets else 17) and b["gt_bboxs"].shape[1] == (4 if buckets else 3)
        out = tr._call(b)
        loss = tr.loss_mix(out)[0]
        model.zero_grad(set_to_none=True)
        loss.backward()
        got.append(([float(x.reshape(-1)[0]) for x in out], {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}))
    (l0, g0), (l1, g1) = got
    np.testing.assert_allclose(l0, l1, rtol=2e-5, atol=1e-6)
    assert g0.keys() == g1.keys()
    # (a gradient that is zero in exact arithmetic -- alpha_net.bias shifts every score of a softmax alike -- is rounding noise on
    # both sides: the error is measured against the parameter's own gradient norm or 1e-4 of the largest one, whichever is larger)
    top = max(float(g.norm()) for g in g0.values())
    for k in g0:
        err = float((g0[k] - g1[k]).norm()) / max(float(g0[k].norm()), 1e-4 * top)
        assert err < 2e-5, (k, err)


@pytest.mark.parametrize("K,widths,MA,MC,ksplit", [(8192, (2048, 2048, 2048), 64, 64, 5), (8192, (2048, 2048), 64, 64, 8),
                                                  (4096, (1024, 512, 132), 33, 64, 3), (264, (128, 36), 64, 40, 1),
                                                  (1000, (260,), 7, 1, 4), (128, (32, 32, 32), 64, 64, 1)])
def test_backward_data_product_128_rows_equals_two_64_row_products(K, widths, MA, MC, ksplit):
    """cvc_linear_nn_planes2_fwd (two 64-row operand groups against ONE stream of the weights: what lets the two loops of the
    cyclical pass share a backward-data product at B = 64 each) against cvc_linear_nn_fwd on each group alone under the same K
    split: the same bits for every row, through the plane output (+ the consumer-side plane sum), the reduce form and ksplit = 1;
    odd numbers of 8-row groups per wave (fp32-MFMA tail), a ragged last column slab, single rows."""
    import ctypes as C
    from cvc import hip
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(K + MA)
    ntot_cols = sum(widths)
    W = (torch.randn(K, ntot_cols + 8, generator=g) * 0.05).to(dev)
    dA, dC = (torch.randn(MA, K, generator=g)).to(dev), (torch.randn(MC, K, generator=g)).to(dev)
    qA, qC = hip.pack_quad(dA), hip.pack_quad(dC)
    ranges, c0 = [], 0
    for n in widths:
        ranges.append((W, c0, n))
        c0 += n
    want_a = hip.linear_nn(qA, MA, K, ranges, ksplit=ksplit)
    want_c = hip.linear_nn(qC, MC, K, ranges, ksplit=ksplit)
    # fp64 sanity of the reference itself
    ref = (dA.double() @ W.double()[:, :widths[0]]).float()
    assert float((want_a[0] - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-5
    L = hip.lib()
    slabs = sum((n + 127) // 128 for n in widths)
    ntot = slabs * 128
    for reduce in (1, 0):
        arr = (hip.NNSeg * len(widths))()
        outs = []
        for i, (w, c, n) in enumerate(ranges):
            o = torch.full((128, n), float("nan"), device=dev)
            arr[i] = hip.NNSeg(w.data_ptr() + 4 * c, o.data_ptr(), w.stride(0), n, n)
            outs.append(o)
        ws = torch.full((max(ksplit, 1) * 128 * ntot,), float("nan"), device=dev)
        hip._check(L.cvc_linear_nn_planes2_fwd(qA.data_ptr(), qC.data_ptr(), K, MA, MC, arr, len(widths), ksplit, ws.data_ptr(), reduce,
                                               hip._stream()), "cvc_linear_nn_planes2_fwd")
        torch.cuda.synchronize()
        if reduce or ksplit == 1:
            got = outs
        else:
            # the planes, summed in plane order as the consumers do (cvc_grad_src)
            planes = ws.view(ksplit, 128, ntot)
            got, s0 = [], 0
            for n in widths:
                acc = planes[0, :, s0:s0 + n].clone()
                for k in range(1, ksplit):
                    acc += planes[k, :, s0:s0 + n]
                got.append(acc)
                s0 += (n + 127) // 128 * 128
        for o, wa, wc in zip(got, want_a, want_c):
            assert torch.equal(o[:MA], wa) and torch.equal(o[64:64 + MC], wc)
            if reduce or ksplit == 1:      # rows nobody owns are not written
                assert bool(torch.isnan(o[MA:64]).all()) and bool(torch.isnan(o[64 + MC:]).all())
