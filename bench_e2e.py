"""`bench.py --mode e2e-train | e2e-eval`: the reference's REAL flow, end to end -- raw frame + region features through the
once-per-clip encoder (model/backbone.py:298-351) into the hot path, every step (main.py:216-228 -> trainer.py:39-150 for
training; trainer.py:152-250 for evaluation).  Not the headline metric (BASELINE.json's decode-steps/s is quoted on pre-extracted
features); these lines say what a user of the reference's own configuration gets.

  e2e-train : one optimisation step = encoder forward, cyclical forward, backward through both, clip, Adam -- replayed from the
              HIP graph Trainer.train() captures (the persistent GRU kernels' error words stay on the device: cvc.hip.defer_errors)
  e2e-eval  : encoder forward + T-step greedy decode of B clips (model(..., lang_eval=True))
Workloads: any cvc.synth config (cfg2 = B=64, N=100, F=480, R=2048) and `refdefault`, the reference's default shape
(cfgs/cyclical.yml:43, 69-71 + opts.py:55: B=48, N = 10 sampled frames x 100 proposals = 1000 regions, F=480, R=1024, A=E=512).
CPU baseline: a BOUNDED sample (a few clips of the same shapes): the encoder as the library modules the reference calls (nn.Linear,
nn.BatchNorm1d, nn.GRU ... on the host, through the mirror's CPU path) feeding the oracle's hot path (oracle/ref_cpu.py).
"""
from __future__ import annotations

import dataclasses
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "cyclical-visual-captioning_amd"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench_common import HBM_PEAK_GBS, MFMA_BF16_PEAK_TFLOPS, usable_cores  # noqa: E402


def dims_of(config: str):
    from cvc import synth
    if config == "refdefault":
        return synth.Dims(B=48, N=1000, F=480, R=1024, A=512, E=512, V=5000, T=20, G=2048)
    return synth.CONFIGS[config]


def build_raw(d, device, seed: int, n_clips=None):
    """(opts, model with RegionalFeatureExtractorGVD in front, one collated raw batch of n_clips clips) -- what cvc.main builds for
    --synthetic_raw, at the dims of `d`"""
    from cvc import opts as cvc_opts, synth
    from cvc.data_synth import SyntheticCaptionDataset, collate
    from cvc.model.create_model import build_model
    n_clips = d.B if n_clips is None else n_clips
    o = cvc_opts.build_parser().parse_args([])
    o.test_mode = False
    o.batch_size, o.rnn_size, o.att_hid_size, o.input_encoding_size, o.seq_length = n_clips, d.R, d.A, d.E, d.T
    o.t_attn_size, o.vis_encoding_size, o.att_feat_size = d.F, d.G, d.G
    o.num_prop_per_frm, o.num_sampled_frm = (d.N // 10, 10) if d.N % 10 == 0 else (d.N, 1)
    o.train_decoder_only, o.xe_loss_weight, o.caption_consistency_loss_weight, o.learning_rate = False, 0.5, 0.5, 1e-4
    o.disp_interval, o.hip_graph = 1 << 30, 1
    full = SyntheticCaptionDataset(dataclasses.replace(d, B=n_clips), n_clips, seed, "training", raw=True)
    o.vocab_size, o.itow, o.wtoi, o.itod, o.detect_size = full.vocab_size, full.itow, full.wtoi, full.itod, d.DET
    o.glove_clss, o.glove_vg_cls = torch.from_numpy(full.glove_clss), torch.from_numpy(full.glove_vg_cls)
    o.vg_cls, o.detectron_tables = full.vg_cls, full.tables
    torch.manual_seed(seed)
    model = build_model(o, device)
    batch = collate([full[i] for i in range(n_clips)])
    return o, model, batch


GROUPS = (("encoder recurrence (GRU persistent forward / backward)", ("cvc_gru_",)),
          ("dense products on the tile GEMM (encoder layers, GRU input projections, hoisted products, every dW, vocabulary head)", ("cvc_tile_gemm",)),
          ("operand packs", ("cvc_tile_pack", "cvc_pack_")),
          ("hot-path loops A / C (cells, attention, back-propagation through time)", ("loopA.", "loopC.", "loops.", "cvc_train_loop")),
          ("whole-decode driver (T greedy steps)", ("cvc_decode_",)),
          ("optimizer (clip + Adam)", ("cvc_adam_clip_step",)),
          ("encoder fused kernels (layer norms, class softmax, BatchNorm, dropout epilogues)",
           ("cvc_layernorm", "cvc_class_softmax", "cvc_bn_", "cvc_relu_dropout", "cvc_frame_embed", "cvc_dropout")))


def role_split(rows, ms_step):
    out, rest = [], 0.0
    used = set()
    for title, prefixes in GROUPS:
        ms = 0.0
        for r in rows:
            if r["kernel"] not in used and any(r["kernel"].startswith(p) for p in prefixes):
                ms += r["ms_per_step"]
                used.add(r["kernel"])
        if ms > 0:
            out.append(dict(role=title, ms_per_step=round(ms, 3), share=round(ms / ms_step, 4)))
    rest = sum(r["ms_per_step"] for r in rows if r["kernel"] not in used)
    out.append(dict(role="everything else on own kernels (criteria, label glue, embedding, grounder, small linear layers)",
                    ms_per_step=round(rest, 3), share=round(rest / ms_step, 4)))
    return out


def _cpu_sample(args, d, what: str, n_clips: int = 4):
    """rank 0, no GPU work.  Encoder: the mirror's CPU path = the library modules the reference's backbone.py calls; hot path: the
    oracle.  `n_clips` clips of the workload's shapes (bounded: the full batch takes minutes on a host)."""
    from cvc import synth
    from oracle import ref_cpu as O
    ncores = usable_cores()
    torch.set_num_threads(ncores)
    o, model, batch = build_raw(d, torch.device("cpu"), args.seed, n_clips=n_clips)
    seg, iseq, gts, num, props, bboxs, box_mask, _ids, region, frm_mask, sample_idx, ppl_mask = batch
    pnt_mask = torch.cat((ppl_mask.new_zeros(ppl_mask.size(0), 1), ppl_mask), 1)
    enc = model.roi_feat_extractor
    P = {k: v.detach() for k, v in model.state_dict().items() if not k.startswith("roi_feat_extractor.")}
    P["roi_feat_extractor.vis_embed.0.weight"] = enc.vis_embed[0].weight.detach()
    P["roi_feat_extractor.vis_classifiers_bias"] = enc.vis_classifiers_bias.detach()
    glue = dict(input_seq=iseq, gt_seq=gts, num=num, proposals=props, gt_bboxs=bboxs, box_mask=box_mask, frm_mask=frm_mask,
                sample_idx=sample_idx)

    def encode():
        overlaps = O.bbox_overlaps(props, bboxs, frm_mask | pnt_mask[:, 1:].unsqueeze(-1))
        fc, conv, p_conv, pool, p_pool, g_pool, pm, _ov, _cp, cls_loss = enc(seg.float(), props, num, box_mask, region, bboxs, overlaps, sample_idx)
        return dict(fc_feats=fc, conv_feats=conv, p_conv_feats=p_conv, pool_feats=pool, p_pool_feats=p_pool, g_pool_feats=g_pool,
                    pnt_mask=pm), cls_loss

    best = None
    reps = 1                       # ONE timed pass, no warm-up pass: a pass is 10 - 30 s at config-2 size
    for rep in range(reps):
        c0 = time.perf_counter()
        if what == "eval":
            model.eval()
            with torch.no_grad():
                feats, _ = encode()
                O.greedy_sample(P, feats, d.T, synth.UNK_IDX)
        else:
            model.eval()          # (eval-mode dropout: the reference's train-mode backward does not run on torch 2.x)
            Pg = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in P.items()}
            for k in list(Pg):
                if k.startswith("attended_roi_decoder_core.") and "lstm" in k:
                    Pg[k] = Pg[k.replace("attended_roi_decoder_core.", "decoder_core.")]
            model.zero_grad(set_to_none=True)
            feats, cls_loss = encode()
            ls = O.cyclical_forward(Pg, feats, glue, T=d.T, vocab_size=d.V)
            O.training_loss(ls, xe_loss_weight=0.5, w_att2=0.0, w_cls=0.0, caption_consistency_loss_weight=0.5).backward()
        dt = time.perf_counter() - c0
        best = dt if best is None else min(best, dt)
    unit = "clips/s"
    return dict(value=round(n_clips / best, 2), unit=unit, cores=torch.get_num_threads(), kind="port",
                sample=f"{n_clips} clips of the same shapes (the bench batch is {d.B}): encoder = the library modules the reference's "
                       f"backbone.py calls, on the host; hot path = the oracle ({'greedy decode' if what == 'eval' else 'cyclical forward + autograd backward, no optimizer step'}); "
                       f"one pass, no warm-up; torch {torch.__version__}, {ncores} host cores", seconds=round(best, 3))


def run_e2e(args, d, dev, what: str, steps=None, warmup=None, config_name=None, cpu_baseline=True, probe=True):
    """N = 1 only (rank 0).  Returns the bench line."""
    import ctypes as C
    from cvc import hip
    from cvc.distributed import GradReducer
    from cvc.trainer import Trainer, build_optimizer
    from bench_train import TrainBench, shape_work, train_work
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    config_name = config_name or args.config
    o, model, batch = build_raw(d, dev, args.seed)
    shape = dict(B_per_gpu=d.B, N=d.N, F=d.F, D=d.R, A=d.A, E=d.E, V=d.V, T=d.T, G=d.G)
    if what == "eval":
        model.eval()
        tr = Trainer(o, None, model, None, None, None)
        b = tr._prepare(batch, False)
        with torch.no_grad():
            for _ in range(max(1, warmup)):
                seq, _att, _ = tr._call(b, True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                seq, _att, _ = tr._call(b, True)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            ms_step = el / steps * 1e3
            rows = []
            if probe:
                timers = hip.enable_timers()
                try:
                    for _ in range(2):
                        tr._call(b, True)
                    torch.cuda.synchronize()
                finally:
                    hip.disable_timers()
                for k, v in timers.items():
                    rows.append(dict(kernel=k, launches_per_step=len(v) // 2, ms_per_step=round(sum(a.elapsed_time(b_) for a, b_ in v) / 2, 3)))
                rows.sort(key=lambda r: -r["ms_per_step"])
                for r in rows:
                    r["share"] = round(r["ms_per_step"] / ms_step, 4)
        line = {"metric": "end-to-end evaluation clips/sec: encoder + greedy decode (not the headline metric)", "value": round(d.B * steps / el, 1),
                "unit": "clips/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "decode_steps_per_s": round(d.B * d.T * steps / el, 1),
                "config": {"workload": f"{config_name}: raw frame features [B,{d.F},3072] + region features [B,{d.N},{d.G}] -> encoder -> "
                                       f"T={d.T} greedy decode (model(..., lang_eval=True), reference trainer.py:208-211)", **shape},
                "roofline": None, "cpu_baseline": None, "role_split": role_split(rows, ms_step) if rows else None, "kernels": rows[:12]}
    else:
        model.train()
        optim = build_optimizer(model, o)
        reducer = GradReducer(model.named_parameters())
        try:
            tr = Trainer(o, None, model, optim, None, None, grad_reducer=reducer)
            capable = tr.graph_capable() and not args.no_train_graph
            with tr.deferred_errors() as deferred:
                b = tr._prepare(batch, True)
                step = tr.train_step_bucketed if capable else (lambda bb: torch.stack([x.reshape(()) for x in tr.train_step_prepared(bb)]))
                for _ in range(max(3, warmup)):                   # two eager steps, then the capture
                    res = step(b)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    res = step(b)
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                ms_step = el / steps * 1e3
                void = float(res[5]) if (deferred and res.numel() > 5) else 0.0
                # the same step, eager
                n_e = max(2, min(steps, 5))
                tr._core_step(b)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n_e):
                    tr._core_step(b)
                torch.cuda.synchronize()
                eager_ms = (time.perf_counter() - t0) / n_e * 1e3
                rows, roof = [], None
                if probe:
                    L = hip.lib()
                    nprobe = 2
                    cap = 4 * 16 * d.T * (nprobe + 1)
                    hip._check(L.cvc_train_loop_profile(cap), "cvc_train_loop_profile")
                    timers = hip.enable_timers()
                    try:
                        for _ in range(nprobe):
                            tr._core_step(b)
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
                    for k, v in timers.items():
                        if k in ("cvc_train_loop_fwd", "cvc_train_loop_bwd", "cvc_train_loops_bwd_joint"):
                            continue
                        tot[k] = sum(a.elapsed_time(b_) for a, b_ in v) / nprobe
                        cnt[k] = len(v) // nprobe
                    opt_elems = sum(p.numel() for g in optim.param_groups for p in g["params"] if p.grad is not None)
                    work = train_work(d)
                    work.update(shape_work(hip.TIMED_SHAPES, nprobe, opt_elems))
                    for name in sorted(tot, key=lambda k: -tot[k]):
                        ent = dict(kernel=name, launches_per_step=cnt[name], ms_per_step=round(tot[name], 3), share=round(tot[name] / ms_step, 4))
                        wk = work.get(name)
                        if wk and cnt[name]:
                            avg_s = tot[name] / cnt[name] * 1e-3
                            ent.update(achieved_GBs=round(wk["bytes"] / avg_s / 1e9, 1), frac_hbm=round(wk["bytes"] / avg_s / 1e9 / HBM_PEAK_GBS, 4))
                            if wk.get("flops"):
                                peak = MFMA_BF16_PEAK_TFLOPS / 6
                                ent.update(achieved_TFLOPs=round(wk["flops"] / avg_s / 1e12, 2), frac_mfma=round(wk["flops"] / avg_s / 1e12 / peak, 4))
                        rows.append(ent)
                    dom = next((r for r in rows if r["kernel"] == "cvc_tile_gemm"), None)
                    if dom is not None and "achieved_TFLOPs" in dom:
                        roof = dict(kernel="cvc_tile_gemm", bound="mfma", achieved=dom["achieved_TFLOPs"], peak=round(MFMA_BF16_PEAK_TFLOPS / 6, 1),
                                    unit="TFLOP/s", frac=dom["frac_mfma"], traffic=None, share=dom["share"],
                                    peak_note="fp32-equivalent flops; split products issue 6 bf16 MFMAs each: dense bf16 peak / 6")
        finally:
            reducer.remove_hooks()
        line = {"metric": "end-to-end training clips/sec: encoder + cyclical forward, backward, clip, Adam (not the headline metric)",
                "value": round(d.B * steps / el, 1), "unit": "clips/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
                "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                "data": "synthetic", "decode_steps_per_s": round(d.B * d.T * steps / el, 1),
                "config": {"workload": f"{config_name}: raw frame features [B,{d.F},3072] + region features [B,{d.N},{d.G}] -> encoder -> cyclical "
                                       f"train step (reference main.py:216-228 -> trainer.py:39-150), train-mode dropout", **shape,
                           "hip_graph": bool(capable), "deferred_error_words": bool(deferred), "eager_ms_per_step": round(eager_ms, 3),
                           "graph_stats": dict(tr.graph_stats), "last_step_void": bool(void)},
                "roofline": roof, "cpu_baseline": None, "role_split": role_split(rows, ms_step) if rows else None, "kernels": rows[:16]}
    del model
    torch.cuda.empty_cache()
    if cpu_baseline and not args.no_cpu_baseline:
        cpu = _cpu_sample(args, d, "eval" if what == "eval" else "train")
        line["cpu_baseline"] = cpu
        line["gpu_over_cpu"] = round(line["value"] / cpu["value"], 1)
    return line
