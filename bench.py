#!/usr/bin/env python3
"""Headline benchmark: caption decode-steps/sec (B x T per full greedy decode) on MI355X.

  python bench.py                                                     # BASELINE config 2, 300 timed decodes (>= 1 s)
  python bench.py --gpus 1 --steps 20 --warmup 5                      # what the driver runs
  python bench.py --gpus 8 [--mode train --config cfg4]               # starts the 8 rank processes itself (torch.distributed.run
                                                                      # child, before anything here touches a GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...      # or under a launcher
--gpus N is a promise: the line carries n_gpus = N only if N ranks joined the process group (checked with an all-reduce);
anything else exits non-zero without a line.

A "step" is one pass of the hot path over one batch: the full T-step greedy decode of B clips
(embed, att-LSTM, both attentions, lang-LSTM, vocab projection, word selection per decode step),
features and weights resident in HBM, replayed from one HIP graph.  value = clips x T x ranks x
steps / max-over-ranks wall time.  Clips are independent: ranks decode disjoint batches, no
collective on the data path (weak scaling).

Extra objects in the JSON line:
  roofline      -- the dominant kernel (largest share of the step's GPU time), achieved
                   algorithmic bytes (or flops) per launch / its average duration measured with HIP
                   events in this process; `kernels` lists every kernel of the step the same way.
  cpu_baseline  -- the CPU oracle (torch fp32, the reference's own ATen op sequence) timed on this
                   box's host cores on the same workload (rank 0, N=1 only).
  secondary     -- default run only (N=1, headline config): short measurements (>= 0.3 s each) of the other BASELINE configs --
                   config 3 beam 5, config 3 cyclical train step, config 4 per-GPU train step, config 5 greedy + beam 5, the
                   once-per-clip encoder -- each with its own ms_per_step and roofline.  The headline fields are not affected.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cyclical-visual-captioning_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench_common import HBM_PEAK_GBS, MFMA_F32_PEAK_TFLOPS, MFMA_BF16_PEAK_TFLOPS, usable_cores, pmc_traffic  # noqa: E402
from bench_train import run_train  # noqa: E402


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=300, help="timed steps (default: >= 1 s of decodes at cfg2)")
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--min-warm-seconds", type=float, default=0.5,
                   help="untimed: keep replaying after the W warm-up steps until this much wall time has passed, so that the "
                        "timed K steps start at settled clocks (a 20-step region is 80 ms)")
    p.add_argument("--encoder-forward-only", action="store_true",
                   help="--mode encoder: the HIP forward and its roofline only, no per-piece / library-module comparison passes (PMC "
                        "collection: MIOpen's GRU does not run under rocprofv3 --pmc on this pool)")
    p.add_argument("--config", default="cfg2", help="cvc.synth.CONFIGS key (cfg2 = B=64,N=100,D=2048,T=20 greedy)")
    for k in ("B", "N", "F", "R", "A", "E", "V", "T"):
        p.add_argument("--" + k, type=int, default=None)
    p.add_argument("--beam", type=int, default=1)
    p.add_argument("--mode", default="decode", choices=["decode", "train", "encoder", "e2e-train", "e2e-eval"],
                   help="decode = headline metric (default); train = cyclical fwd+bwd+all-reduce+Adam step (configs 3-ii / 4); "
                        "encoder = the once-per-clip region / frame encoder (SURVEY 8(f) rank 1, library ops) per piece; "
                        "e2e-train / e2e-eval = the reference's real flow, raw features through the encoder into the hot path every "
                        "step (bench_e2e.py; --config also takes `refdefault`, the reference's default shape)")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--gate-ksplit", type=int, default=None, choices=[0, 1, 2],
                   help="packed path: 1 = K-split gate GEMMs (activations shared through LDS + finishing kernel), 0 = full-K "
                        "kernel; default = cvc.decode.GATE_KSPLIT_DEFAULT")
    p.add_argument("--tile-loaders", type=int, default=None, choices=[0, 1, 2, 3, 4],
                   help="tile GEMM form (A/B): 0 = every wave copies, 1 = loader waves + 8 computing waves, 2 = loader waves + 4 wide "
                        "computing waves, 3 = 2 for long K loops, 1 otherwise (default of the library), 4 = register-load loader "
                        "waves (in the beam step: lang / att gate GEMMs 148 -> 153 / 103 -> 107 us, slower)")
    p.add_argument("--lstm-blocks", type=int, default=None, choices=[1, 2],
                   help="packed decode LSTM gate GEMM (A/B): 32-row weight blocks per workgroup (library default 1)")
    p.add_argument("--gsk", type=int, default=None, choices=[0, 1],
                   help="packed path (A/B): 1 = grouped stream-K schedule (measured slower; off by default)")
    p.add_argument("--embgate", type=int, default=None, choices=[0, 1],
                   help="packed path (A/B): 0 = the 7-launch schedule without the embedding-gate table; default = on")
    p.add_argument("--train-graph", action="store_true", help="(default since round 4; kept for old command lines)")
    p.add_argument("--no-train-graph", action="store_true",
                   help="--mode train: time eager steps instead of replays of the step captured into one HIP graph (forward, backward, "
                        "RCCL exchange, clip + Adam)")
    p.add_argument("--always-exchange", action="store_true",
                   help="--mode train at N = 1: run the gradient exchange anyway -- the per-bucket in-place reduce-scatter + all-gather on a "
                        "ONE-rank RCCL communicator, inside the captured step -- and report what it costs (exchange.in_graph_ms)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--spawn", action="store_true",
                   help="start the rank processes through the torch.distributed.run child also for --gpus 1 (the path every N > 1 run takes)")
    p.add_argument("--no-secondary", action="store_true", help="default decode run: skip the short runs of the other configs")
    p.add_argument("--secondary-seconds", type=float, default=0.35, help="timed seconds per secondary measurement")
    p.add_argument("--secondary-timeout", type=float, default=600.0,
                   help="bench.py --gpus N under a launcher: if the all-ranks training secondary (config 4, the exchange captured in the "
                        "step) has not returned after this many seconds, rank 0 prints the decode line with the secondary's entry saying "
                        "so and every rank exits 0 -- a hang in the N-rank exchange must not cost the decode measurement")
    p.add_argument("--watchdog-seconds", type=float, default=2400.0,
                   help="a run still going after this much wall time writes every thread's Python stack to stderr and exits 3 (a hung "
                        "bench must not hold a GPU box until somebody's outer limit; 0 = off).  SIGTERM writes the stacks as well.")
    p.add_argument("--cpu-repeats", type=int, default=3)
    p.add_argument("--seed", type=int, default=1234)
    return p.parse_args(argv)


def algorithmic_work(d, beam):
    """Per-launch algorithmic bytes / flops of every kernel of one decode step (DESIGN.md section 4;
    SURVEY.md section 8(d) per-unit figures x the units one launch processes = B clips)."""
    B, N, F, R, A, E, V = d.B, d.N, d.F, d.R, d.A, d.E, d.V
    rows = B * beam
    w = {}
    w["attn_scores"] = dict(bound="hbm", bytes=4 * B * (N + F) * A + B * N + 4 * rows * (A + N + F))
    w["attn_wsum"] = dict(bound="hbm", bytes=4 * B * (N + F) * R + 4 * rows * (N + F + R))
    w["gate_fc"] = dict(bound="hbm", bytes=4 * (4 * R * R + 8 * R) + 4 * rows * (R + 4 * R), flops=2 * rows * 4 * R * R)
    k_att = E + 2 * R           # the fc segment is hoisted out of the step loop (gate_fc, once per decode)
    w["att_lstm"] = dict(bound="hbm", bytes=4 * (4 * R * k_att + 8 * R) + 4 * rows * (k_att + 3 * R), flops=2 * rows * 4 * R * k_att)
    w["lang_lstm"] = dict(bound="hbm", bytes=4 * (4 * R * 3 * R + 8 * R) + 4 * rows * (3 * R + 3 * R), flops=2 * rows * 4 * R * 3 * R)
    w["h2attn"] = dict(bound="hbm", bytes=4 * (A * R + A) + 4 * rows * (R + A), flops=2 * rows * A * R)
    w["logits"] = dict(bound="hbm", bytes=4 * (V * R + V) + 4 * rows * (R + V), flops=2 * rows * V * R)
    w["word_select"] = dict(bound="hbm", bytes=4 * rows * 6 * ((V + 31) // 32) + 4 * rows * E)
    # grouped stream-K schedule (csrc/gemm_gsk.hip): the same weight bytes regrouped -- every launch streams its groups' weights
    # once; the partial tiles (fp32, 256 x 64 per segment) are written by the stream-K launch and read by its consumer
    tile_b = 4 * 256 * 64
    n_lstm_tiles, segs = R // 64, lambda k: k / 32 / 21.0 + 1       # ~ segments per tile at ~21 chunks per workgroup
    w["att_early_logits"] = dict(bound="hbm", bytes=4 * (4 * R * 2 * R + V * R) + 4 * rows * 2 * R +
                                 int(tile_b * (n_lstm_tiles * segs(2 * R) + (V + 255) // 256 * segs(R))),
                                 flops=2 * rows * (4 * R * 2 * R + V * R))
    w["lang_early_h2attn"] = dict(bound="hbm", bytes=4 * (4 * R * 2 * R + A * R) + 4 * rows * 2 * R +
                                  int(tile_b * (n_lstm_tiles * segs(2 * R) + (A + 255) // 256 * segs(R))),
                                  flops=2 * rows * (4 * R * 2 * R + A * R))
    w["att_late"] = dict(bound="hbm", bytes=4 * (4 * R * E) + 4 * rows * (E + 4 * R + 3 * R) + int(tile_b * n_lstm_tiles * segs(2 * R)),
                         flops=2 * rows * 4 * R * E)
    w["lang_late"] = dict(bound="hbm", bytes=4 * (4 * R * R + 8 * R) + 4 * rows * (R + 3 * R) + int(tile_b * n_lstm_tiles * segs(2 * R)),
                          flops=2 * rows * 4 * R * R)
    return w


def run_encoder(args, d, dev, brief=False, steps=None, warmup=None):
    """Once-per-clip encoder (cvc/model/backbone.py, mirror of the reference's RegionalFeatureExtractorGVD, backbone.py:189-351)
    at the hot path's dimensions: raw frame features [B, F, 3072] and region features [B, N, G] -> the tensors the decoder reads.
    The frame-context GRU and the dense layers run on the HIP kernels (cvc/gru.py, cvc/dense.py); this mode times the forward, its
    pieces and (not brief) forward + backward against the library modules.  clips/s, not decode-steps/s: the encoder runs once per
    clip, the decoder T times.  Returns the bench line."""
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    import argparse as ap
    from cvc import synth
    from cvc.model import backbone
    from cvc.model.backbone import RegionalFeatureExtractorGVD
    from cvc import gru as gru_hip
    tables = synth.detectron_tables(d, args.seed)
    o = ap.Namespace(
        vocab_size=d.V, seq_length=d.T, seq_per_img=1, rnn_size=d.R, input_encoding_size=d.E, att_hid_size=d.A, drop_prob_lm=0.5,
        detect_size=d.DET, vis_encoding_size=d.G, enable_BUTD=False, att_input_mode="both", num_sampled_frm=10, finetune_cnn=False,
        att_feat_size=d.G, fc_feat_size=synth.SEG_FEAT_DIM, t_attn_size=d.F, second_drop_prob=0.3, att_model="topdown",
        t_attn_mode="bigru", test_mode=False, itod={i + 1: "d%d" % i for i in range(d.DET)},
        vg_cls=["vg%d" % i for i in range(tables["glove_vg_cls"].shape[0])], glove_clss=torch.from_numpy(tables["glove_clss"]),
        glove_vg_cls=torch.from_numpy(tables["glove_vg_cls"]), detectron_tables=tables)
    enc = RegionalFeatureExtractorGVD(o).to(dev).eval()
    inp = {k: (torch.from_numpy(np.ascontiguousarray(v)).to(dev) if isinstance(v, np.ndarray) else v)
           for k, v in synth.encoder_inputs(d, args.seed).items()}
    from cvc.misc import utils
    overlaps = utils.bbox_overlaps(inp["proposals"], inp["gt_bboxs"], inp["frm_mask"] | inp["pnt_mask_in"][:, 1:].unsqueeze(-1))

    def fwd():
        return enc(inp["segs_feat"], inp["proposals"], inp["num"], inp["box_mask"], inp["region_feats"], inp["gt_bboxs"], overlaps,
                   inp["sample_idx"])

    def timed(fn, n):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    from cvc import hip
    with torch.no_grad():
        for _ in range(warmup):
            fwd()
        ms = timed(fwd, max(3, min(steps, 30)))
        # ---- per-entry-point GPU time of one forward (HIP events around every C-ABI launch) and the dominant kernel's roofline
        timers = hip.enable_timers()
        fwd(); torch.cuda.synchronize(); timers.clear()
        for _ in range(2):
            fwd()
        torch.cuda.synchronize()
        hip.disable_timers()
    tot = {k: sum(a.elapsed_time(b) for a, b in v) / 2 for k, v in timers.items()}
    cnt = {k: len(v) // 2 for k, v in timers.items()}
    B_, F_, N_, R_, A_, G_ = d.B, d.F, d.N, d.R, d.A, d.G
    H_ = R_ // 2
    pool_in = enc.pool_feat_size
    work = {
        # recurrence of one GRU layer, both directions: W_hh once, the input projections of every step read, every h_t written
        # (2 layers) x (W_hh of both directions once + the input projections of every step read + every h_t written);
        # flops: h W_hh^T of every step, both directions, both layers
        "cvc_gru_seq_persistent_fwd": dict(bytes=2 * (4 * 2 * 3 * H_ * H_ + 4 * B_ * F_ * 2 * 3 * H_ + 4 * B_ * F_ * 2 * H_), flops=2 * 2 * B_ * F_ * 2 * 3 * H_ * H_),
        "cvc_gru_seq_fwd": dict(bytes=2 * (4 * F_ * 2 * 3 * H_ * H_ + 4 * B_ * F_ * 2 * 3 * H_ + 4 * B_ * F_ * 2 * H_), flops=2 * 2 * B_ * F_ * 2 * 3 * H_ * H_),
        # every dense product of the forward on the tile GEMM: GRU input projections (2 layers), frame embeddings, ctx2att_fc, region side
        "cvc_tile_gemm": dict(bytes=0, flops=2 * B_ * F_ * (2 * R_ * 3 * R_ + 2048 * H_ + 1024 * H_ + R_ * A_) +
                              2 * B_ * N_ * (G_ * G_ + pool_in * R_ + R_ * A_)),
    }
    kernels = []
    for name in sorted(tot, key=lambda k: -tot[k]):
        ent = dict(kernel=name, launches=cnt[name], ms=round(tot[name], 3), share=round(tot[name] / ms, 4))
        wk = work.get(name)
        if wk:
            secs = tot[name] * 1e-3
            ent.update(achieved_GBs=round(wk["bytes"] / secs / 1e9, 1), achieved_TFLOPs=round(wk["flops"] / secs / 1e12, 2),
                       algorithmic_bytes=wk["bytes"], algorithmic_flops=wk["flops"])
        kernels.append(ent)
    roof = None
    dom = next((e for e in kernels if "achieved_GBs" in e), None)
    if dom is not None:
        # PMC traffic of the dominant entry point: per-launch average of its kernel x its launches per forward (the work
        # figures above are per forward too)
        traffic, tnote = pmc_traffic(dom["kernel"], args, None, mode="encoder", beam=1, config_name=args.config)
        if traffic is not None:
            traffic, tnote = traffic * dom["launches"], tnote + f" x {dom['launches']} launches per forward"
        if dom["kernel"] == "cvc_tile_gemm":
            peak = MFMA_BF16_PEAK_TFLOPS / 6
            roof = dict(kernel=dom["kernel"], bound="mfma", achieved=dom["achieved_TFLOPs"], peak=round(peak, 1), unit="TFLOP/s",
                        frac=round(dom["achieved_TFLOPs"] / peak, 4), traffic=traffic, traffic_source=tnote, ms=dom["ms"],
                        peak_note="fp32-equivalent flops; split products issue 6 bf16 MFMAs each: dense bf16 peak / 6")
        else:
            roof = dict(kernel=dom["kernel"], bound="hbm", achieved=dom["achieved_GBs"], peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(dom["achieved_GBs"] / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=tnote, ms=dom["ms"],
                        algorithmic_bytes=dom["algorithmic_bytes"],
                        note="the recurrence is a chain of F dependent steps (10.6 us each: arrival poll, state from L2, MFMAs, "
                             "write-through store): latency-bound far below either roof, DESIGN.md section 7")
    line = {
        "metric": "once-per-clip encoder clips/sec (not the headline metric)", "value": round(d.B / (ms * 1e-3), 1),
        "unit": "clips/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: RegionalFeatureExtractorGVD forward (eval), raw frame features [B,{d.F},3072] + region "
                               f"features [B,{d.N},{d.G}]", "B_per_gpu": d.B, "N": d.N, "F": d.F, "D": d.R, "G": d.G},
        "roofline": roof, "cpu_baseline": None, "kernels": kernels}
    if brief or args.encoder_forward_only:
        return line
    with torch.no_grad():
        # pieces (same tensors, eval mode)
        B, N, F = d.B, d.N, d.F
        regions = inp["region_feats"]
        seg = inp["segs_feat"].float()
        x_rnn = torch.randn(B, F, d.R, device=dev)
        pieces = {
            "ctx2pool_grd  [B*N, G] x [G, G] + ReLU (backbone.py:206-210)": lambda: enc.ctx2pool_grd(regions),
            "class similarity softmax [B, DET+1, N] (backbone.py:216-235)": lambda: enc.class_similarity(regions, inp["pnt_mask_in"][:, 1:]),
            "frame embeddings 2 x Linear + BatchNorm (backbone.py:325-333)": lambda: enc.att_embed_aux(
                torch.cat((enc.att_embed[0](seg[..., :2048]), enc.att_embed[1](seg[..., 2048:3072])), 2).transpose(1, 2)),
            "2-layer BiGRU over F frames, library module (MIOpen) (backbone.py:335-338)": lambda: enc.context_enc(x_rnn),
            "2-layer BiGRU over F frames, HIP path (cvc/gru.py: tile GEMM + cvc_gru_seq_fwd)": lambda: gru_hip.gru_forward(enc.context_enc, x_rnn),
            "ctx2att_fc [B*F, R] x [R, A] (backbone.py:343)": lambda: enc.ctx2att_fc(x_rnn),
        }
        piece_ms = {k: round(timed(fn, 10), 3) for k, fn in pieces.items()}
    # the GRU under autograd (what end-to-end training through the encoder pays): forward + backward, both implementations
    gru_mod = enc.context_enc
    gru_mod.train()
    xg = x_rnn.clone().requires_grad_(True)
    probe = torch.randn(d.B, d.F, d.R, device=dev)

    def fb_hip():
        gru_mod.zero_grad(set_to_none=True); xg.grad = None
        (gru_hip.gru_forward_train(gru_mod, xg) * probe).sum().backward()

    def fb_lib():
        gru_mod.zero_grad(set_to_none=True); xg.grad = None
        gru_mod.flatten_parameters()
        (gru_mod(xg)[0] * probe).sum().backward()

    piece_ms["2-layer BiGRU forward + backward, HIP path (cvc.gru.gru_forward_train: cvc_gru_seq_bwd + tile GEMM)"] = round(timed(fb_hip, 3), 3)
    piece_ms["2-layer BiGRU forward + backward, library module (MIOpen)"] = round(timed(fb_lib, 3), 3)
    # the whole encoder under autograd (forward + backward of a probe loss over its outputs), HIP kernels vs library modules
    from cvc import dense as dense_mod

    def enc_fb():
        enc.zero_grad(set_to_none=True)
        outs = fwd()
        sum(o.float().pow(2).mean() for o in outs[:6] if torch.is_tensor(o) and o.dtype.is_floating_point).backward()

    enc.train()
    piece_ms["whole encoder forward + backward, HIP GRU + tile-GEMM dense layers"] = round(timed(enc_fb, 3), 3)
    backbone.HIP_GRU, dense_mod.ENABLED = False, False
    piece_ms["whole encoder forward + backward, library GRU + library GEMMs"] = round(timed(enc_fb, 3), 3)
    backbone.HIP_GRU, dense_mod.ENABLED = True, True
    enc.eval()
    gru_mod.eval()
    with torch.no_grad():
        backbone.HIP_GRU = False
        ms_library = timed(fwd, max(3, min(steps, 10)))
        backbone.HIP_GRU = True
    line.update(ms_per_step_with_library_gru=round(ms_library, 3), pieces_ms=piece_ms)
    return line


def spawn_ranks(args) -> int:
    """--gpus N (N > 1) outside a launcher: start the N rank processes as a torch.distributed.run child and hand its exit code
    back.  Nothing here has touched a GPU (torch.cuda.device_count() does not initialise one), so no process that holds GPU
    state is ever replaced or forked.  Fails loudly -- non-zero, no bench line -- when the machine has fewer than N devices or
    any rank fails to come up (the launcher then returns non-zero itself)."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but this machine shows {n_dev} GPU(s); refusing to run fewer ranks than asked for "
              f"(there is no CPU fallback)", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this pool
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def run_decode(args, d, dev, rank, world, dist_on, beam, steps, warmup, min_warm, cpu_baseline, over, config_name, sd_np=None):
    """The headline measurement: K timed full decodes of B clips (graph replay), per-kernel HIP-event timing, roofline, CPU oracle.
    Returns the bench line on rank 0, None elsewhere."""
    from cvc import synth, hip
    from cvc.decode import DecodeEngine, DecodeWeights
    seed = args.seed + rank                       # every rank decodes its own clips
    if sd_np is None:
        sd_np = synth.hot_path_state_dict(d, args.seed)
    feats_np = synth.clip_features(d, seed)
    W = DecodeWeights({k: torch.from_numpy(v).to(dev) for k, v in sd_np.items()})
    feats = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in feats_np.items()}
    eng = DecodeEngine(W, feats, d.T, synth.UNK_IDX, beam=beam,
                       gate_ksplit=None if args.gate_ksplit is None else (bool(args.gate_ksplit) if args.gate_ksplit < 2 else "fused"),
                       gsk=None if args.gsk is None else bool(args.gsk), embgate=None if args.embgate is None else bool(args.embgate))
    gemm_mode = hip.gemm_packed_split(-1)
    if not args.no_graph:
        eng.capture()

    def sync_all():
        torch.cuda.synchronize()
        if dist_on:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    w0 = time.perf_counter()
    for _ in range(warmup):
        eng.run()
    torch.cuda.synchronize()
    while time.perf_counter() - w0 < min_warm:      # untimed: DVFS / cache warm-up beyond the W steps
        eng.run()
        torch.cuda.synchronize()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.run()
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    sync_all()
    elapsed = t_local
    if dist_on:
        from cvc.distributed import control_all_reduce
        elapsed = control_all_reduce([t_local], "max")[0]
    units = d.B * d.T * world * steps
    value = units / elapsed
    if rank != 0:
        return None

    # ---- per-kernel durations with HIP events on the launch stream (eager pass, same buffers)
    kernels, roof = [], None
    work = algorithmic_work(d, beam)
    split_mode_now = gemm_mode
    if getattr(eng, "embgate", False):
        # embedding-gate schedule: the att-LSTM streams only the recurrent K range (2R) and gathers one 4R-row of the per-checkpoint
        # table per clip
        R_, rows_ = d.R, d.B * beam          # every live hypothesis row is multiplied (beam > 1: the tile path's finishing launch adds the table row)
        work["att_lstm"] = dict(bound="hbm", bytes=4 * (4 * R_ * 2 * R_) + 4 * rows_ * (2 * R_ + 3 * R_ + 4 * R_ + 4 * R_),
                                flops=2 * rows_ * 4 * R_ * 2 * R_)
        work["word_select"] = dict(bound="hbm", bytes=4 * rows_ * 6 * ((d.V + 31) // 32))
        if eng.packed and split_mode_now == 2 and d.T > 1:
            # step 0 multiplies the all-zero initial state and contracts over less K (one chunk for the attention cell, 2R of 3R for
            # the language cell): the per-launch figures are the averages over the decode's T launches, as the durations are
            T_ = d.T
            wa, wl = work["att_lstm"], work["lang_lstm"]
            a0 = 4 * (4 * R_ * 32) + 4 * rows_ * (32 + 3 * R_ + 4 * R_ + 4 * R_)
            l0 = 4 * (4 * R_ * 2 * R_ + 8 * R_) + 4 * rows_ * (2 * R_ + 3 * R_)
            work["att_lstm"] = dict(bound="hbm", bytes=((T_ - 1) * wa["bytes"] + a0) // T_,
                                    flops=((T_ - 1) * wa["flops"] + 2 * rows_ * 4 * R_ * 32) // T_)
            work["lang_lstm"] = dict(bound="hbm", bytes=((T_ - 1) * wl["bytes"] + l0) // T_,
                                     flops=((T_ - 1) * wl["flops"] + 2 * rows_ * 4 * R_ * 2 * R_) // T_)
    if getattr(eng, "gsk", False):
        work["word_select"] = dict(bound="hbm", bytes=int(4 * 256 * 64 * ((d.V + 255) // 256) * (d.R / 32 / 21.0 + 1)) + 4 * d.B * d.E)
    split_mode = gemm_mode
    acc = {}
    eng.run_timed()                            # warm
    for _ in range(3):
        for k, v in eng.run_timed().items():
            acc.setdefault(k, []).extend(v)
    decode_ms = sum(float(np.sum(v)) / 3 for v in acc.values())
    for name, ms in acc.items():
        avg = float(np.mean(ms))
        wk = work.get(name)
        ent = dict(kernel=name, avg_us=round(avg * 1e3, 2), launches_per_decode=len(ms) // 3,
                   share=round(float(np.sum(ms)) / 3 / decode_ms, 4) if decode_ms > 0 else None)
        if wk:
            gbs = wk["bytes"] / (avg * 1e-3) / 1e9
            ent.update(algorithmic_bytes=wk["bytes"], achieved_GBs=round(gbs, 1), frac_hbm=round(gbs / HBM_PEAK_GBS, 4))
            bound = "hbm"
            if "flops" in wk:
                # matrix work as EXECUTED: the packed path takes every fp32 product as six bf16 cross terms
                # (cvc_gemm_packed_split), the row-major path issues fp32 MFMAs
                split = (eng.packed and split_mode > 0) or getattr(eng, "tile", False)
                mult, peak, what = (6, MFMA_BF16_PEAK_TFLOPS, "bf16 32x32x16, 6 per fp32 product") if split else \
                                   (1, MFMA_F32_PEAK_TFLOPS, "f32 32x32x2")
                tf = mult * wk["flops"] / (avg * 1e-3) / 1e12
                ent.update(algorithmic_flops=wk["flops"], mfma=what, executed_mfma_flops=mult * wk["flops"],
                           achieved_TFLOPs=round(tf, 2), mfma_peak_TFLOPs=peak, frac_mfma=round(tf / peak, 4))
                # the binding roof is the one with the larger floor time
                if mult * wk["flops"] / (peak * 1e12) > wk["bytes"] / (HBM_PEAK_GBS * 1e9):
                    bound = "mfma"
            ent["bound"] = bound
            ent["traffic"] = pmc_traffic(name, args, over, beam=beam, config_name=config_name)[0]      # PMC HBM bytes per launch
        kernels.append(ent)
    kernels.sort(key=lambda e: -(e["share"] or 0))
    # dominant KERNEL, not launch: the two LSTM gate GEMMs of a step are launches of one kernel symbol; when their combined
    # share is the largest the roofline is quoted on the longer of the two launches (its own bytes over its own duration)
    dom = next(e for e in kernels if "achieved_GBs" in e)
    gates = [e for e in kernels if e["kernel"] in ("att_lstm", "lang_lstm") and "achieved_GBs" in e]
    sk = [e for e in kernels if e["kernel"] in ("att_early_logits", "lang_early_h2attn") and "achieved_GBs" in e]
    if sk and sum(e["share"] or 0 for e in sk) >= (dom["share"] or 0):         # the two stream-K launches are one kernel symbol
        gates = sk
    gate_share = sum(e["share"] or 0 for e in gates)
    if gates and gate_share >= (dom["share"] or 0):
        dom = max(gates, key=lambda e: e["avg_us"])
        dom["share_of_kernel_symbol"] = round(gate_share, 4)
    traffic, traffic_note = pmc_traffic(dom["kernel"], args, over, beam=beam, config_name=config_name)
    if dom["bound"] == "mfma":
        roof = dict(kernel=dom["kernel"], bound="mfma", achieved=dom["achieved_TFLOPs"], peak=dom["mfma_peak_TFLOPs"],
                    unit="TFLOP/s", frac=dom["frac_mfma"], traffic=traffic, traffic_source=traffic_note,
                    avg_us=dom["avg_us"], hbm_frac=dom["frac_hbm"])
    else:
        roof = dict(kernel=dom["kernel"], bound="hbm", achieved=dom["achieved_GBs"], peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=dom["frac_hbm"], traffic=traffic, traffic_source=traffic_note, avg_us=dom["avg_us"],
                    algorithmic_bytes=dom["algorithmic_bytes"])

    # ---- CPU baseline: the oracle on this box's host cores, same workload (rank 0, N=1 only)
    cpu = None
    if world == 1 and cpu_baseline and not args.no_cpu_baseline:
        from oracle import ref_cpu as O
        ncores = usable_cores()
        torch.set_num_threads(ncores)
        P_cpu, f_cpu = O.to_torch(sd_np), O.to_torch(feats_np)
        best = None
        with torch.no_grad():
            fn = (lambda: O.greedy_sample(P_cpu, f_cpu, d.T, synth.UNK_IDX)) if beam == 1 else \
                (lambda: O.beam_search(P_cpu, f_cpu, d.T, synth.UNK_IDX, beam))
            c0 = time.perf_counter()
            fn()
            t_warm = time.perf_counter() - c0
            # a bounded sample of about 10 s of CPU work: the greedy decode (1.7 s each) repeats more often than the beam search (7 s)
            reps = max(args.cpu_repeats, min(8, int(10.0 / max(t_warm, 1e-3)) - 1))
            t_all = t_warm
            for _ in range(reps):
                c0 = time.perf_counter()
                fn()
                dt = time.perf_counter() - c0
                t_all += dt
                best = dt if best is None else min(best, dt)
        cpu = dict(value=round(d.B * d.T / best, 1), unit="decode-steps/s", cores=torch.get_num_threads(), kind="port",
                   sample=f"one full decode of the same workload (B={d.B}, T={d.T}), warm-up 1, best of {reps} ({t_all:.0f} s of CPU work "
                          f"in all); torch {torch.__version__} CPU, {ncores} host cores", seconds=round(best, 3))

    sched = ("grouped stream-K (early K ranges of the gate GEMMs ride with logits / h2attn)" if getattr(eng, "gsk", False) else
             ("embedding-gate table (att-LSTM GEMM over K = 2R + one table row per word)" if getattr(eng, "embgate", False) else
              "one GEMM over the full K per cell"))
    line = {
        "metric": f"caption decode-steps/sec (BxT) at N={d.N},D={d.R}", "value": round(value, 1), "unit": "decode-steps/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{config_name}: greedy caption decode" if beam == 1 else f"{config_name}: beam={beam} caption decode",
                   "B_per_gpu": d.B, "N": d.N, "F": d.F, "D": d.R, "A": d.A, "E": d.E, "V": d.V, "T": d.T, "beam": beam,
                   "hip_graph": not args.no_graph, "parallelism": f"clips sharded over {world} rank(s), no collective",
                   "gemm_arithmetic": ("f32 in / f32 accumulate; products = exact 3-way bf16 split of both operands, 6 leading "
                                       "cross terms on the bf16 MFMA (error vs f64 <= the f32-MFMA path's, tests/test_gpu_parity.py)")
                   if ((eng.packed and gemm_mode > 0) or getattr(eng, "tile", False)) else "f32 MFMA",
                   "engine_path": "packed" if eng.packed else ("tile" if getattr(eng, "tile", False) else "ring"),
                   "schedule": sched if eng.packed else None},
        "roofline": roof, "cpu_baseline": cpu, "kernels": kernels,
    }
    if cpu:
        line["gpu_over_cpu"] = round(value / cpu["value"], 1)
    del eng
    return line


def brief(line):
    """One secondary entry: what was measured, its time and its roofline (the per-kernel table stays with the headline)."""
    keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "exchange", "gradient_buckets", "role_split")
    out = {k: line[k] for k in keep if k in line and line[k] is not None}
    ks = [k for k in line.get("kernels", []) if "kernel" in k][:4]
    out["top_kernels"] = [{kk: k[kk] for kk in ("kernel", "avg_us", "ms", "ms_per_step", "share", "frac_hbm", "frac_mfma") if kk in k} for k in ks]
    return out


def run_secondary(args, dev):
    """Short runs of the other BASELINE configs (>= args.secondary_seconds of timed steps each), same process, rank 0 of an N=1 run.
    A failing entry records its error instead of taking the headline line down with it."""
    import dataclasses
    from cvc import synth
    secs = args.secondary_seconds
    out = []

    def attempt(name, fn):
        t0 = time.perf_counter()
        try:
            e = fn()
        except Exception as ex:       # noqa: BLE001 -- recorded, not hidden: the entry says what failed
            e = {"error": f"{type(ex).__name__}: {ex}"}
        e["name"] = name
        e["wall_s"] = round(time.perf_counter() - t0, 2)
        out.append(e)
        torch.cuda.empty_cache()

    def decode(cfg, beam, est_ms):
        d = synth.CONFIGS[cfg]
        steps = max(3, int(secs * 1e3 / est_ms) + 1)
        return brief(run_decode(args, d, dev, 0, 1, False, beam, steps, 2, 0.15, False, {}, cfg))

    def train(cfg, est_ms, always_exchange=False, B=None):
        d = synth.CONFIGS[cfg] if B is None else dataclasses.replace(synth.CONFIGS[cfg], B=B)
        steps = max(3, int(secs * 1e3 / est_ms) + 1)
        return brief(run_train(args, d, dev, 0, 1, steps=steps, warmup=2, min_warm=0.2, cpu_baseline=False, config_name=cfg, regions=3,
                               always_exchange=always_exchange))

    attempt("cfg3 beam=5 decode", lambda: decode("cfg3", 5, 11.0))
    attempt("cfg3 cyclical train step (B=64)", lambda: train("cfg3", 25.0))
    attempt("cfg4 cyclical train step, one GPU's share (B=32 per GPU) WITH the per-bucket RCCL exchange captured in the step, on a "
            "one-rank communicator (what 8 ranks add is the time on the xGMI links)", lambda: train("cfg4", 15.0, always_exchange=True))
    attempt("cfg3 cyclical train step at B=128 per GPU (two 64-clip groups of the C-driven loops, batch-wide head / criteria / localizer)",
            lambda: train("cfg3", 40.0, B=128))
    attempt("cfg5 greedy decode", lambda: decode("cfg5", 1, 19.0))
    attempt("cfg5 beam=5 decode", lambda: decode("cfg5", 5, 48.0))
    attempt("once-per-clip encoder (cfg2 size)", lambda: brief(run_encoder(args, synth.CONFIGS["cfg2"], dev, brief=True, steps=8, warmup=2)))
    # the reference's real flow: raw features through the encoder into the hot path every step (bench_e2e.py; CPU baselines of these
    # are in the `--mode e2e-*` lines under profiles/: a host pass of this size takes 10 - 30 s)
    from bench_e2e import dims_of, run_e2e

    def e2e(cfg, what, steps):
        return brief(run_e2e(args, dims_of(cfg), dev, what, steps=steps, warmup=2, config_name=cfg, cpu_baseline=False))
    attempt("end-to-end train step, raw features through the encoder (cfg2 size), HIP-graph replay", lambda: e2e("cfg2", "train", 5))
    attempt("end-to-end eval, raw features: encoder + greedy decode (cfg2 size)", lambda: e2e("cfg2", "eval", 8))
    attempt("end-to-end train step at the reference's default shape (B=48, N=1000, F=480, R=1024)", lambda: e2e("refdefault", "train", 5))
    return out


def run_secondary_ranks(args, dev, rank, world, comm):
    """ALL RANKS (bench.py --gpus N under a launcher, decode mode, default config).  BASELINE config 4: the cyclical training step at
    B = 32 clips per rank with the per-bucket RCCL exchange inside the captured step, `exchange.exposed_ms` and the efficiency against
    the same process's exchange-off step.  No rank-0-only probe passes (bench_train.py's symmetry rule); whether the entry can run
    at all (a communicator exists) is the same on every rank by construction.  Returns the list for `secondary` (rank 0) or []."""
    from cvc import synth
    from cvc.distributed import control_all_reduce
    name = (f"cfg4 cyclical train step, {world} rank(s): B=32 per GPU (global {32 * world}), the per-bucket RCCL exchange captured in the "
            f"step on the {world}-rank communicator")
    t0 = time.perf_counter()
    if os.environ.get("CVC_BENCH_TEST_HANG_SECONDARY") == "1":      # test hook (tests/test_gpu_train.py): a secondary that never returns
        while True:
            time.sleep(1.0)
    if comm is None:
        ent = {"error": "no RCCL communicator in this run (see `rccl`): the training step needs its exchange"}
    else:
        d = synth.CONFIGS["cfg4"]
        steps = max(3, int(args.secondary_seconds * 1e3 / 12.0) + 1)
        err = None
        try:
            ln = run_train(args, d, dev, rank, world, steps=steps, warmup=2, min_warm=0.2, cpu_baseline=False, config_name="cfg4",
                           regions=3, comm=comm, always_exchange=True, probe=True)
        except Exception as ex:      # noqa: BLE001 -- recorded in the entry; the decode line still goes out
            ln, err = None, f"{type(ex).__name__}: {ex}"[:400]
        # a failure on ANY rank is every rank's failure (the others may have finished their steps or not: nothing further runs)
        bad = control_all_reduce([1.0 if err else 0.0], "max")[0] > 0
        ent = {"error": err or "the step failed on another rank"} if bad else (brief(ln) if rank == 0 else {})
    if rank != 0:
        return []
    ent["name"] = name
    ent["wall_s"] = round(time.perf_counter() - t0, 2)
    return [ent]


def _secondary_guard(args, rank, world, line, finish):
    """A timer over the all-ranks training secondary of `bench.py --gpus N`: that step holds the only collectives of the run (the
    decode path has none), captured into a HIP graph on an N-rank communicator no 1-GPU box can rehearse.  If it does not return within
    --secondary-timeout seconds, every rank writes its Python stacks to stderr, rank 0 prints the decode line -- already measured, with
    `secondary[0].error` saying what happened -- and the process leaves with status 0 without waiting for the GPU (os._exit: a
    collective that never completes cannot be synchronised with)."""
    import faulthandler
    import threading

    def fire():
        try:
            faulthandler.dump_traceback(all_threads=True)
            if rank == 0:
                line["secondary"] = [{"name": f"RCCL communicator + cfg4 cyclical train step, {world} rank(s)",
                                      "error": f"did not return within --secondary-timeout = {args.secondary_timeout:.0f} s (stacks on stderr); "
                                               "the decode line above it is complete"}]
                finish(line)
        finally:
            os._exit(0)
    t = threading.Timer(args.secondary_timeout, fire) if args.secondary_timeout > 0 else threading.Timer(1e9, lambda: None)
    t.daemon = True
    t.start()
    return t


def main():
    args = parse()
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    import faulthandler
    import signal
    faulthandler.register(signal.SIGTERM, all_threads=True, chain=True)      # a launcher that gives up on us learns where we were
    if args.watchdog_seconds > 0:
        faulthandler.dump_traceback_later(args.watchdog_seconds, exit=True)
    if (args.gpus > 1 or args.spawn) and not under_launcher:
        # N ranks were asked for and nobody started them: do it here, BEFORE anything below touches a GPU
        raise SystemExit(spawn_ranks(args))
    # stdout carries ONE thing: the bench line.  Libraries that write to file descriptor 1 from native code (RCCL prints a
    # version banner there when a communicator comes up) are sent to stderr; the line itself goes to the saved descriptor.
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        os.write(line_fd, (json.dumps(line) + "\n").encode())
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: n_gpus must be the number of ranks that run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    if os.environ.get("CVC_BENCH_DEVICE") is not None:
        # test knob: every rank on THIS device (two rank processes time-slicing the one GPU of a test box, exchanging through the
        # stand-in librccl of tests/stub_rccl -- real RCCL refuses two ranks on one device); never set by a measurement
        local_rank = int(os.environ["CVC_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.tile_loaders is not None:
        from cvc import hip as _hip
        _hip.lib().cvc_tile_gemm_loaders(int(args.tile_loaders))
    if args.lstm_blocks is not None:
        from cvc import hip as _hip
        _hip.lib().cvc_packed_lstm_wg_blocks(int(args.lstm_blocks))
    # under a launcher (RANK set) the process group is always initialised, also for a single rank, so that the
    # barrier / max-over-ranks path is the same code at every N
    dist_on = under_launcher
    ranks_joined = None                # counted THROUGH RCCL (one 1.0 per rank summed by the communicator itself), or null
    ranks_control_plane = 1            # counted on the gloo control plane (says nothing about RCCL)
    comm = None
    rccl_error = None
    if dist_on:
        # torch.distributed is the CONTROL plane (rendezvous, the communicator's unique id, host barriers, max over ranks) on gloo;
        # the data plane is the package's own RCCL communicator over xGMI (cvc.comm.RcclComm): no c10d RCCL group, hence no c10d
        # watchdog thread next to the graph captures below
        import torch.distributed as dist
        from cvc.comm import RcclComm
        from cvc.distributed import control_all_reduce
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("gloo")
        ranks_control_plane = int(round(control_all_reduce([1.0], "sum")[0]))
        if ranks_control_plane != args.gpus or dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: {ranks_control_plane} rank(s) joined the process group, --gpus asked for {args.gpus}")

    def make_comm():
        """The data plane: the package's own RCCL communicator, and the ranks counted through it.  Every rank ends up on the same side
        of the fallback: RcclComm.from_process_group() agrees on the outcome over the control plane before it returns (a rank-local
        ncclCommInitRank failure reaches every rank), so either all ranks hold a communicator or all of them take the except branch."""
        nonlocal comm, ranks_joined, rccl_error
        try:
            comm = RcclComm.from_process_group()
            ranks_joined = comm.count_ranks()
        except Exception as e:
            # the decode path shards by clips and has no data-path collective: its measurement does not
            # depend on the communicator, so a box on which RCCL cannot be brought up still gets its decode line -- with the
            # failure in it, `ranks_joined` null and the ranks counted on the control plane only.  The training step exchanges
            # gradients: fatal there.
            if args.mode == "train":
                raise
            comm = None
            rccl_error = f"{type(e).__name__}: {e}"[:300]
            print(f"bench.py: rank {rank}: RCCL communicator unavailable ({rccl_error}); decode has no collective, continuing",
                  file=sys.stderr, flush=True)
        if ranks_joined is not None and ranks_joined != args.gpus:
            raise SystemExit(f"bench.py: {ranks_joined} rank(s) joined the RCCL communicator, --gpus asked for {args.gpus}")

    # --mode train needs the communicator for what it measures; a decode run measures FIRST (no collective in that path) and brings
    # the communicator up afterwards, inside the timed guard below: neither a communicator that cannot be created nor one whose
    # creation never returns can cost the decode line
    if dist_on and args.mode == "train":
        make_comm()

    import dataclasses
    from cvc import synth
    from cvc import hip
    hip.lib()

    from bench_e2e import dims_of, run_e2e
    d = dims_of(args.config)
    over = {k: getattr(args, k) for k in ("B", "N", "F", "R", "A", "E", "V", "T") if getattr(args, k) is not None}
    if over:
        d = dataclasses.replace(d, **over)

    def finish(line):
        """rank 0: the run-level keys + the summary, then the ONE line"""
        import build_hip
        line["ranks_joined"] = ranks_joined
        line["ranks_control_plane"] = ranks_control_plane
        if rccl_error is not None:
            line["rccl"] = "unavailable (ranks_joined is null; ranks_control_plane counts the gloo group): " + rccl_error
        # which library ran: a CVC_LIB variant (A/B builds) can never be taken for the in-tree product build
        line["library"] = dict(path=os.path.relpath(hip.LIB_PATH, ROOT) if hip.LIB_PATH.startswith(ROOT) else hip.LIB_PATH,
                               in_tree_default=not os.environ.get("CVC_LIB"), version=hip.version(),
                               source_hash=build_hip.source_hash(), note="source_hash = sha256 of the kernel sources in this tree")
        # LAST key of the line (a log that keeps only the tail of stdout still shows it): every measurement's headline numbers
        summ = [dict(name="headline: " + line["config"]["workload"], value=line["value"], unit=line["unit"], ms_per_step=line["ms_per_step"],
                     roofline_frac=(line.get("roofline") or {}).get("frac"))]
        for e in line.get("secondary", []):
            summ.append(dict(name=e.get("name"), value=e.get("value"), unit=e.get("unit"), ms_per_step=e.get("ms_per_step"),
                             roofline_kernel=(e.get("roofline") or {}).get("kernel"), roofline_frac=(e.get("roofline") or {}).get("frac"),
                             **({"exchange_in_graph_ms": e["exchange"].get("in_graph_ms"),
                                 "exchange_efficiency": e["exchange"].get("efficiency_vs_exchange_off")} if e.get("exchange") else {}),
                             **({"error": e["error"]} if "error" in e else {})))
        line["summary"] = summ
        emit(line)

    line = None
    if args.mode == "encoder":
        if rank == 0:
            line = run_encoder(args, d, dev)
    elif args.mode in ("e2e-train", "e2e-eval"):
        if rank == 0:
            line = run_e2e(args, d, dev, "eval" if args.mode == "e2e-eval" else "train")
    elif args.mode == "train":
        line = run_train(args, d, dev, rank, world, comm=comm, always_exchange=args.always_exchange or dist_on)
    else:
        line = run_decode(args, d, dev, rank, world, dist_on, args.beam, args.steps, args.warmup, args.min_warm_seconds, True, over, args.config)
        default_cfg = (args.config == "cfg2" and not over and args.beam == 1 and not args.no_secondary and
                       args.gsk is None and args.embgate is None and args.gate_ksplit is None and not args.no_graph)
        if default_cfg and world == 1 and not dist_on:
            t0 = time.perf_counter()
            line["secondary"] = run_secondary(args, dev)
            line["secondary_wall_s"] = round(time.perf_counter() - t0, 1)
        elif dist_on:
            # N ranks (the driver's `bench.py --gpus N`, or --spawn at N = 1): the decode line is measured; now the communicator
            # (`ranks_joined` is counted through it) and, for the default configuration, BASELINE config 4's training step run by
            # EVERY rank -- B = 32 clips per rank, the six-bucket RCCL exchange captured inside the step's graph -- so that one
            # command on an 8-GPU node measures config 4 (global B = 256) with its exposed exchange time.  All of it under a timer.
            t0 = time.perf_counter()
            guard = _secondary_guard(args, rank, world, line, lambda ln: finish(ln))
            make_comm()
            sec = run_secondary_ranks(args, dev, rank, world, comm) if default_cfg else None
            guard.cancel()
            if rank == 0 and sec is not None:
                line["secondary"] = sec
                line["secondary_wall_s"] = round(time.perf_counter() - t0, 1)
    if rank == 0:
        finish(line)
    if dist_on:
        # everything that holds work on the communicator (the captured training step, reducers' events) is released before the
        # group is torn down
        import gc
        import torch.distributed as dist
        gc.collect()
        torch.cuda.synchronize()
        dist.barrier()
        if comm is not None:
            comm.destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
