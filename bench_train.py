"""`bench.py --mode train` (and the training secondaries of the default run): the cyclical training step of BASELINE configs 3-ii / 4.

RANK SYMMETRY is the design rule of this file.  A training step of an N-rank run contains the gradient exchange -- six in-place
reduce-scatter + all-gather pairs on the package's RCCL communicator -- so EVERY function below that runs a step is called by EVERY
rank, the same number of times, in the same order; nothing that can reach a collective sits under `if rank == 0`.  Only the
formatting of what was measured (and the CPU baseline, which runs no GPU step) is rank 0's alone.  Round 5's single function ran
its probe passes on rank 0 only with the exchange still on: ranks 1 .. N-1 never joined those collectives and the run hung at N > 1.
tests/test_gpu_train.py runs run_train() once as rank 0 and once as rank 1 of a recording world-2 communicator and compares the
two collective logs call for call.
"""
from __future__ import annotations

import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "cyclical-visual-captioning_amd"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench_common import HBM_PEAK_GBS, MFMA_F32_PEAK_TFLOPS, MFMA_BF16_PEAK_TFLOPS, usable_cores, pmc_traffic  # noqa: E402


def train_work(d):
    """Algorithmic bytes / flops PER LAUNCH of the roles inside the two C-driven training loops (cyclical pass; SURVEY.md section
    8(d) training formulas).  Weights stream once per launch; M = B rows.  `mfma`: which matrix instruction executes the products.
    Names = the launch roles (loop.direction.role, csrc/train_driver.hip)."""
    B, N, F, R, A = d.B, d.N, d.F, d.R, d.A
    gemm = lambda k, nout=4 * R: dict(bytes=4 * nout * k + 4 * B * (k + nout), flops=2 * B * nout * k, mfma="split")
    w = {}
    # forward cells: the recurrent columns only (fc / word / localized-context terms are hoisted into dense products)
    for lp in ("loopA", "loopC"):
        w[f"{lp}.fwd.att_cell"] = gemm(2 * R)
        w[f"{lp}.bwd.nn_att"] = gemm(2 * R)
    w["loopA.fwd.lang_cell"], w["loopC.fwd.lang_cell"] = gemm(3 * R), gemm(2 * R)
    w["loopA.bwd.nn_lang"], w["loopC.bwd.nn_lang"] = gemm(3 * R), gemm(2 * R)
    w["loopA.fwd.h2attn"] = gemm(R, A)
    w["loopA.bwd.nn_h2attn"] = gemm(R, A)
    # joint backward (both loops' rows against one stream of the weights, cvc_train_loops_bwd_joint): 2B rows per product
    gemm2 = lambda k: dict(bytes=4 * 4 * R * k + 4 * 2 * B * (k + 4 * R), flops=2 * 2 * B * 4 * R * k, mfma="split")
    w["loops.bwd.nn_lang"], w["loops.bwd.nn_att"] = gemm2(3 * R), gemm2(2 * R)
    # attention of one step: projected rows once (scores), context rows once (weighted sum); backward: context rows once (d_attn),
    # projected rows once (tanh recomputed)
    w["loopA.fwd.attn_scores"] = dict(bytes=4 * B * (N + F) * A, flops=B * (N + F) * 4 * A, mfma="none")
    w["loopA.fwd.attn_wsum"] = dict(bytes=4 * B * (N + F) * R, flops=2 * B * (N + F) * R, mfma="none")
    w["loopA.bwd.attn_bwd"] = dict(bytes=4 * B * (N + F) * (A + R), flops=B * (N + F) * (8 * A + 2 * R), mfma="none")
    # gate-gradient kernels (both loops of a step in one launch): per row and hidden unit, the four activated gates + c + c' + the
    # upstream dh planes in, dG (row-major and as the next product's operand) out
    gg = lambda rows: dict(bytes=4 * rows * R * (4 + 2 + 3 + 2 * 4), flops=0, mfma="none")
    w["loops.bwd.gate_grad_lang"], w["loops.bwd.gate_grad_att"] = gg(2 * B), gg(2 * B)
    for lp in ("loopA", "loopC"):
        w[f"{lp}.bwd.gate_grad_lang"], w[f"{lp}.bwd.gate_grad_att"] = gg(B), gg(B)
    return w


def shape_work(shapes_by_entry, nprobe, optim_elems, optim_write_grad=True):
    """Work per LAUNCH (averaged over a step's launches of that entry point) of the entry points called from Python, from the
    sizes cvc.hip.enable_timers() recorded (cvc.hip.TIMED_SHAPES): dense products, operand packs, the localizer's attention
    backward, the optimizer pass."""
    w = {}
    sh = shapes_by_entry.get("cvc_tile_gemm", [])
    if sh:
        # fp32-equivalent flops 2 M N K; operand fragments read once (6 B per element: three bf16 terms) + the K-slice slabs written
        per = len(sh)
        w["cvc_tile_gemm"] = dict(bytes=sum(6.0 * K * (M + N) + 4.0 * M * N * ks for K, M, N, ks in sh) / per,
                                  flops=sum(2.0 * K * M * N for K, M, N, _ in sh) / per, mfma="split")
    for name in ("cvc_tile_pack_rows_any", "cvc_tile_pack_cols"):
        sh = shapes_by_entry.get(name, [])
        if sh:      # fp32 in (4 B), three bf16 terms out (6 B) per element
            w[name] = dict(bytes=sum(10.0 * a * b for a, b in sh) / len(sh), flops=0, mfma="none")
    sh = shapes_by_entry.get("cvc_pack_lstm_segs", [])
    if sh:          # [4R, K] fp32 in, the same elements out in fragment order
        w["cvc_pack_lstm_segs"] = dict(bytes=sum(8.0 * 4 * R * K for K, R in sh) / len(sh), flops=0, mfma="none")
    sh = shapes_by_entry.get("cvc_attn_bwd", [])
    if sh:
        # (nclip, nq, n, A, R, has d_ctx, wants d_proj, wants d_ctxfeat): projected rows once (score backward), context rows once
        # when the context path carries a gradient, the feature gradients written when asked for
        tot = 0.0
        for nclip, nq, n, A, R, has_dctx, d_proj, d_ctxfeat in sh:
            tot += 4.0 * nclip * n * (A + (R if has_dctx else 0)) + 4.0 * nclip * nq * (3 * n + 2 * A + (R if has_dctx else 0))
            tot += 4.0 * nclip * n * ((A if d_proj else 0) + (R if d_ctxfeat else 0))
        w["cvc_attn_bwd"] = dict(bytes=tot / len(sh), flops=0, mfma="none")
    if optim_elems:
        # sum-of-squares pass reads g; the update pass reads p, g, m, v and writes p, m, v (+ g: the clipped value or the zero that
        # spares the next step's fill)
        w["cvc_adam_clip_step"] = dict(bytes=4.0 * optim_elems * (1 + 4 + 3 + (1 if optim_write_grad else 0)), flops=0, mfma="none")
    return w


def _mark(reducer, section: str):
    """tell a recording communicator (tests) which part of the bench the following collectives belong to"""
    mark = getattr(getattr(reducer, "comm", None), "mark", None)
    if mark is not None:
        mark(section)


class TrainBench:
    """One model + optimizer + reducer + static batch; the measurement passes as methods.  Each method's docstring states its rank
    symmetry: "all ranks" = must be called by every rank of the run (it runs training steps, which may contain collectives)."""

    def __init__(self, args, d, dev, rank, world, o, model, optim, reducer, use_graph, config_name):
        from cvc import synth
        from cvc.trainer import Trainer
        self.args, self.d, self.dev, self.rank, self.world = args, d, dev, rank, world
        self.o, self.model, self.optim, self.reducer, self.use_graph, self.config_name = o, model, optim, reducer, use_graph, config_name
        self.tr = Trainer(o, None, model, optim, None, None, grad_reducer=reducer)
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        feats = {k: t(v) for k, v in synth.clip_features(d, args.seed + rank).items()}
        b = {k: t(v) for k, v in synth.label_glue_batch(d, args.seed + rank).items()}
        self.batch = (feats, b["input_seq"], b["gt_seq"], b["num"].cpu(), b["proposals"], b["gt_bboxs"], b["box_mask"],
                      ["v_x_segment_%02d" % i for i in range(d.B)], torch.zeros(d.B, d.N, 1), b["frm_mask"], b["sample_idx"],
                      feats["pnt_mask"][:, 1:])
        import torch.distributed as dist
        self.dist_on = dist.is_available() and dist.is_initialized()
        self.step = self.tr.train_step_graphed if use_graph else self.tr.train_step

    # ------------------------------------------------------------------ passes that run steps: ALL RANKS
    def warm(self, warmup: int, min_warm: float):
        """ALL RANKS.  W untimed steps, then more until `min_warm` seconds have passed ON THE SLOWEST RANK'S CLOCK: whether another
        step runs is agreed over the control plane every time, so that every rank runs the same number of steps (a per-rank clock
        would leave one rank a step -- six collectives -- ahead of the others, waiting for partners that sit in the next barrier)."""
        from cvc.distributed import control_all_reduce
        _mark(self.reducer, "warm")
        w0 = time.perf_counter()
        for _ in range(warmup):
            self.step(self.batch)
        torch.cuda.synchronize()
        while control_all_reduce([time.perf_counter() - w0], "min")[0] < min_warm:
            self.step(self.batch)
            torch.cuda.synchronize()

    def timed(self, n: int, step=None):
        """ALL RANKS.  Exactly n steps between barrier + synchronize on both sides; returns (max-over-ranks seconds, last loss)."""
        from cvc.distributed import control_all_reduce
        import torch.distributed as dist
        step = step or self.step
        if self.dist_on:
            dist.barrier()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = None
        for _ in range(n):
            loss = step(self.batch)[0]
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if self.dist_on:
            dist.barrier()
            el = control_all_reduce([el], "max")[0]
        return float(el), loss

    def headline(self, steps: int, regions: int):
        """ALL RANKS.  The K timed steps (regions > 1, the short secondaries of the default run only: that many regions of K steps,
        the MEDIAN reported -- one host hiccup inside a 0.3 s region moved an entry by 10 % in one collection)."""
        _mark(self.reducer, "timed")
        el, loss = self.timed(steps)
        region_ms = [round(el / steps * 1e3, 3)]
        for _ in range(regions - 1):
            el_r, loss = self.timed(steps)
            region_ms.append(round(el_r / steps * 1e3, 3))
        if regions > 1:
            el = sorted(region_ms)[len(region_ms) // 2] * steps / 1e3
        return el, loss, region_ms

    def eager_compare(self, steps: int):
        """ALL RANKS.  The same step launched eagerly (host-side launch path in the loop) beside the graph replay."""
        if not self.use_graph:
            return None
        _mark(self.reducer, "eager")
        n_e = max(3, min(steps, 10))
        self.tr.train_step(self.batch)
        el_e, _ = self.timed(n_e, self.tr.train_step)
        return round(el_e / n_e * 1e3, 3)

    def exchange_probe(self, steps: int):
        """ALL RANKS.  What the exchange costs inside the step: the same kind of step (graph replay when the headline is one) with
        the exchange switched off on every rank alike (every rank then trains on its own shard: a measurement, not a training
        mode) -- a graph holds the exchange it was captured with, so a second graph is captured for it."""
        red, tr = self.reducer, self.tr
        if not red.exchange:
            return None
        _mark(red, "exchange_probe")
        n_x = max(3, min(steps, 10))
        el1, _ = self.timed(n_x)
        red.exchange, keep_overlap = False, red.overlap
        red.overlap = False
        keep_graph, tr._graph = tr._graph, None
        try:
            self.step(self.batch)
            torch.cuda.synchronize()
            el0, _ = self.timed(n_x)
        finally:
            tr._graph = keep_graph
            red.exchange, red.overlap = True, keep_overlap
        self.step(self.batch)
        ms0, ms1 = el0 / n_x * 1e3, el1 / n_x * 1e3
        grad_bytes = sum(a.numel() * 4 for a in red.arenas)
        return dict(ms_per_step_without_exchange=round(ms0, 3), ms_per_step_with_exchange=round(ms1, 3),
                    in_graph_ms=round(ms1 - ms0, 3) if self.use_graph else None, exposed_ms=round(ms1 - ms0, 3),
                    efficiency_vs_exchange_off=round(ms0 / ms1, 4) if ms1 > 0 else None,
                    measured_on="HIP-graph replays (exchange captured inside the step's graph)" if self.use_graph else "eager steps",
                    ranks=red.world, gradient_bytes=grad_bytes,
                    algorithm="per bucket: in-place reduce_scatter + all_gather on the package's own RCCL communicator (cvc_allreduce_grads), "
                              "on an exchange stream forked from / joined to the step's stream by HIP events, launched the moment the "
                              "bucket's last gradient product is enqueued",
                    backend=red.backend, buckets=len(red.arenas))

    def bucket_probe(self):
        """ALL RANKS (one eager step, exchange as configured).  When does each gradient bucket become complete, relative to the end
        of the step (events on the launch stream)?  At N > 1 a bucket's exchange is enqueued at that moment, behind the product
        that completed it.  Every rank measures; rank 0's numbers are the ones printed."""
        red = self.reducer
        _mark(red, "bucket_probe")
        red.track_ready = True
        try:
            self.tr.train_step(self.batch)
            end_ev = torch.cuda.Event(enable_timing=True)
            end_ev.record()
            torch.cuda.synchronize()
        finally:
            red.track_ready = False
        per = []
        total_b = sum(a.numel() * 4 for a in red.arenas)
        for i, a in enumerate(red.arenas):
            ev = red.ready_events.get(i)
            names = [n_ for n_, _ in red.buckets[i]]
            per.append(dict(bucket=i, first=names[0], tensors=len(names), bytes=a.numel() * 4,
                            ready_ms_before_step_end=None if ev is None else round(ev.elapsed_time(end_ev), 3),
                            complete_by=(red.last_done_how[i] if i < len(getattr(red, "last_done_how", [])) else None) or "finalize",
                            launched=bool(red.exchange)))
        early = sum(b_["bytes"] for b_ in per if (b_["ready_ms_before_step_end"] or 0) >= 1.5)
        return dict(total_bytes=total_b, bytes_ready_1p5ms_before_end=early, fraction=round(early / total_b, 4), per_bucket=per)

    def role_probe(self, ms_step: float, nprobe: int = 2):
        """ALL RANKS (1 + nprobe eager steps, exchange as configured).  GPU time of one step by launch role: HIP events around every
        C-ABI launch (launch stream); the launches INSIDE the two C-driven loops come from the drivers' own per-launch event pairs
        (cvc_train_loop_profile).  Returns (kernel rows, roofline of the dominant role, timing note)."""
        import ctypes as C
        from cvc import hip
        d, tr = self.d, self.tr
        _mark(self.reducer, "role_probe")
        L = hip.lib()
        tr.train_step(self.batch)
        torch.cuda.synchronize()
        cap = 4 * 16 * d.T * (nprobe + 1)
        hip._check(L.cvc_train_loop_profile(cap), "cvc_train_loop_profile")
        timers = hip.enable_timers()
        try:
            for _ in range(nprobe):
                tr.train_step(self.batch)
            torch.cuda.synchronize()
        finally:
            hip.disable_timers()
        kind, loop, ms = (C.c_int * cap)(), (C.c_int * cap)(), (C.c_float * cap)()
        nrec = L.cvc_train_loop_profile_read(kind, loop, ms, cap)
        L.cvc_train_loop_profile(0)
        KN = ["zero_fill", "att_cell", "h2attn", "attn_scores", "attn_wsum", "lang_cell", "gate_grad_lang", "nn_lang", "attn_bwd",
              "nn_h2attn", "gate_grad_att", "nn_att"]
        LN = ["loopA.fwd", "loopC.fwd", "loopA.bwd", "loopC.bwd", "loops.bwd"]
        tot, cnt = {}, {}
        for i in range(nrec):
            name = f"{LN[loop[i]]}.{KN[kind[i]]}"
            tot[name] = tot.get(name, 0.0) + ms[i] / nprobe
            cnt[name] = cnt.get(name, 0) + 1
        cnt = {k: v // nprobe for k, v in cnt.items()}
        for k, v in timers.items():                      # entry points called from Python (dense products, criteria, optimizer)
            if k in ("cvc_train_loop_fwd", "cvc_train_loop_bwd", "cvc_train_loops_bwd_joint"):
                continue
            tot[k] = sum(a.elapsed_time(b) for a, b in v) / nprobe
            cnt[k] = len(v) // nprobe
        work = train_work(d)
        opt_elems = sum(p.numel() for g in self.optim.param_groups for p in g["params"] if p.grad is not None)
        work.update(shape_work(hip.TIMED_SHAPES, nprobe, opt_elems, optim_write_grad=self.reducer is not None))
        ours = sum(tot.values())
        kernels = []
        for name in sorted(tot, key=lambda k: -tot[k]):
            ent = dict(kernel=name, launches_per_step=cnt[name], ms_per_step=round(tot[name], 3),
                       avg_us=round(tot[name] / max(1, cnt[name]) * 1e3, 2), share=round(tot[name] / ms_step, 4))
            wk = work.get(name)
            if wk and cnt[name]:      # work per LAUNCH of this role
                avg_s = tot[name] / cnt[name] * 1e-3
                gbs, tf = wk["bytes"] / avg_s / 1e9, wk.get("flops", 0) / avg_s / 1e12
                peak_tf = MFMA_BF16_PEAK_TFLOPS / 6 if wk.get("mfma") == "split" else MFMA_F32_PEAK_TFLOPS
                bound = "mfma" if wk.get("flops", 0) / (peak_tf * 1e12) > wk["bytes"] / (HBM_PEAK_GBS * 1e9) else "hbm"
                ent.update(algorithmic_bytes=int(wk["bytes"]), algorithmic_flops=int(wk.get("flops", 0)), achieved_GBs=round(gbs, 1),
                           frac_hbm=round(gbs / HBM_PEAK_GBS, 4), achieved_TFLOPs=round(tf, 2), mfma_peak_TFLOPs=round(peak_tf, 1),
                           frac_mfma=round(tf / peak_tf, 4), bound=bound, mfma=wk.get("mfma", "none"),
                           traffic=pmc_traffic(name, self.args, None, mode="train", beam=1, config_name=self.config_name)[0])
            kernels.append(ent)
        # (these are event-timed EAGER launches; the step itself is timed as a graph replay, so the rows need not add up to it: the
        # difference -- launch gaps of the eager pass against library kernels and gaps of the replay -- is stated, not booked as a row)
        note = dict(sum_of_rows_ms=round(ours, 3), step_ms=round(ms_step, 3),
                    note="rows: HIP events around eager launches (probe pass); step: HIP-graph replay" if self.use_graph else "rows and step: eager")
        roof = None
        dom = next((e for e in kernels if "bound" in e), None)
        if dom is not None:
            traffic, tnote = pmc_traffic(dom["kernel"], self.args, None, mode="train", beam=1, config_name=self.config_name)
            if dom["bound"] == "mfma":
                roof = dict(kernel=dom["kernel"], bound="mfma", achieved=dom["achieved_TFLOPs"], peak=dom["mfma_peak_TFLOPs"],
                            unit="TFLOP/s", frac=dom["frac_mfma"], traffic=traffic, traffic_source=tnote, avg_us=dom["avg_us"],
                            share=dom["share"], launches_per_step=dom["launches_per_step"], algorithmic_flops=dom["algorithmic_flops"],
                            algorithmic_bytes=dom["algorithmic_bytes"],
                            peak_note="fp32-equivalent flops; split-product kernels issue 6 bf16 MFMAs per fp32 product, so their "
                                      "roof is the dense bf16 peak / 6" if dom["mfma"] == "split" else "f32 MFMA 32x32x2")
            else:
                roof = dict(kernel=dom["kernel"], bound="hbm", achieved=dom["achieved_GBs"], peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=dom["frac_hbm"], traffic=traffic, traffic_source=tnote, avg_us=dom["avg_us"], share=dom["share"],
                            algorithmic_bytes=dom["algorithmic_bytes"])
        return kernels, roof, note

    # ------------------------------------------------------------------ rank 0 only: runs NO GPU step
    def cpu_baseline(self):
        """RANK 0 of a ONE-rank run only; touches no GPU stream and no communicator.  The oracle's cyclical forward + autograd
        backward on this box's host cores, one step of the same workload (eval-mode dropout: the reference's train-mode backward
        does not run on torch 2.x)."""
        from cvc import synth
        from oracle import ref_cpu as O
        args, d = self.args, self.d
        ncores = usable_cores()
        torch.set_num_threads(ncores)
        f_np, b_np = synth.clip_features(d, args.seed), synth.label_glue_batch(d, args.seed)
        sd_np = synth.hot_path_state_dict(d, args.seed)
        best = None
        for rep in range(1 + max(1, args.cpu_repeats - 1)):
            P = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in O.to_torch(sd_np).items()}
            for k in list(P):
                if k.startswith("attended_roi_decoder_core.") and "lstm" in k:
                    P[k] = P[k.replace("attended_roi_decoder_core.", "decoder_core.")]
            c0 = time.perf_counter()
            ls = O.cyclical_forward(P, O.to_torch(f_np), O.to_torch(b_np), T=d.T, vocab_size=d.V)
            O.training_loss(ls, xe_loss_weight=0.5, w_att2=0.0, w_cls=0.0, caption_consistency_loss_weight=0.5).backward()
            dt = time.perf_counter() - c0
            if rep > 0 or args.cpu_repeats == 1:
                best = dt if best is None else min(best, dt)
            del P, ls
        return dict(value=round(d.B * d.T / best, 1), unit="decode-steps/s", cores=torch.get_num_threads(), kind="port",
                    sample=f"one cyclical forward + backward of the same workload (B={d.B}, T={d.T}; no optimizer step), warm-up 1, "
                           f"best of {max(1, args.cpu_repeats - 1)}; torch {torch.__version__} CPU autograd, {ncores} host cores",
                    seconds=round(best, 3))


def run_train(args, d, dev, rank, world, steps=None, warmup=None, min_warm=None, cpu_baseline=True, config_name=None, probe=True,
              regions=1, comm=None, always_exchange=False):
    """Cyclical training step (BASELINE configs 3-ii / 4): decode -> localize -> reconstruct forward, backward, one RCCL gradient
    exchange (world > 1, or always_exchange on a one-rank communicator), clip_grad_norm_(0.1), Adam.  Train-mode dropout.
    ALL RANKS call it with the same arguments (`probe` included); returns the bench line on rank 0, None elsewhere."""
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    min_warm = args.min_warm_seconds if min_warm is None else min_warm
    config_name = config_name or args.config
    from cvc import synth, opts as cvc_opts
    from cvc.model.captioner import DecodeAndGroundCaptionerGVDROI, PrecomputedRegionFeatures
    from cvc.trainer import build_optimizer
    from cvc.distributed import GradReducer
    o = cvc_opts.parse_opt([])
    o.vocab_size, o.itow, o.wtoi = d.V, {str(i): "w%d" % i for i in range(d.V)}, {"UNK": synth.UNK_IDX}
    o.seq_length, o.rnn_size, o.input_encoding_size, o.att_hid_size = d.T, d.R, d.E, d.A
    o.detect_size, o.vis_encoding_size, o.train_decoder_only = d.DET, d.G, False
    o.xe_loss_weight, o.caption_consistency_loss_weight, o.learning_rate, o.batch_size = 0.5, 0.5, 1e-4, d.B
    model = DecodeAndGroundCaptionerGVDROI(o, roi_extractor=PrecomputedRegionFeatures(d.DET, d.G))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.hot_path_state_dict(d, args.seed).items()}, strict=False)
    model = model.to(dev).train()
    use_graph = not args.no_train_graph
    optim = build_optimizer(model, o, capturable=use_graph)
    # flat gradient arenas (also for one rank: one fill / one clip multiply); the exchange runs on the package's own RCCL
    # communicator (cvc.comm.RcclComm) -- with always_exchange also at N = 1, on a one-rank communicator
    own_comm = None
    if comm is None and always_exchange:
        from cvc.comm import RcclComm
        comm = own_comm = RcclComm.single()
    reducer = GradReducer(model.named_parameters(), comm=comm, always_exchange=always_exchange)
    try:
        tb = TrainBench(args, d, dev, rank, world, o, model, optim, reducer, use_graph, config_name)
        # ---- every rank, same order, same counts
        tb.warm(warmup, min_warm)
        el, loss, region_ms = tb.headline(steps, regions)
        ms_step = el / steps * 1e3
        eager_ms = tb.eager_compare(steps)
        exchange = tb.exchange_probe(steps)
        buckets, kernels, roof, timed_note = None, [], None, None
        if probe:
            buckets = tb.bucket_probe()
            kernels, roof, timed_note = tb.role_probe(ms_step)
        _mark(reducer, "done")
        # ---- rank 0 alone from here: no GPU step, no collective
        if rank != 0:
            return None
        cpu = tb.cpu_baseline() if (probe and world == 1 and cpu_baseline and not args.no_cpu_baseline) else None
        line = {
            "metric": "cyclical train decode-steps/sec (BxT per fwd+bwd+update)", "value": round(d.B * d.T * world * steps / el, 1),
            "unit": "decode-steps/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "samples_per_s": round(d.B * world * steps / el, 2), "loss": float(loss),
            "config": {"workload": f"{config_name}: cyclical train step (decode+localize+reconstruct fwd, bwd, clip, Adam), train-mode dropout",
                       "B_per_gpu": d.B, "global_batch": d.B * world, "N": d.N, "F": d.F, "D": d.R, "T": d.T, "hip_graph": bool(use_graph),
                       "eager_ms_per_step": eager_ms, **({"timed_regions": regions, "region_ms_per_step": region_ms} if regions > 1 else {}),
                       "parallelism": f"dp{world}: clips sharded, " + (f"one RCCL gradient exchange per step ({reducer.world}-rank communicator, "
                                                                          f"inside the step)" if reducer.exchange else
                                                                          "NO gradient exchange in this run (one rank; --always-exchange runs it)")},
            "roofline": roof, "cpu_baseline": cpu, "exchange": exchange, "gradient_buckets": buckets, "kernel_timing": timed_note,
            "kernels": kernels}
        if cpu:
            line["gpu_over_cpu"] = round(line["value"] / cpu["value"], 1)
        return line
    finally:
        # the reducer registers itself in process-global lists (cvc.functional.GRAD_SINKS / LATE_GRAD_LISTENERS): without this every
        # run_train of the default run would leave its parameters and ~0.5 GB of arenas alive and its sinks in every later claim
        reducer.remove_hooks()
        if own_comm is not None:
            own_comm.destroy()
