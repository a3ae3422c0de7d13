#!/usr/bin/env python3
"""profiles/traffic/<config>_beam<b>_<mode>.json from two rocprofv3 PMC passes over bench.py (separate --pmc FETCH_SIZE and
--pmc WRITE_SIZE runs, `--kernel-trace` only), as /opt/skills/guides/MI355X_MICROARCH.md prescribes:

    hbm_bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024

FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B), both counters are in KiB.

One kernel symbol plays several roles in a step (the packed LSTM kernel runs the attention cell and the language cell; the tile
GEMM runs four products per beam step; the backward-data kernel runs three products per training step and loop).  The launch
order is fixed, so a role is (symbol regex, period, which positions of the period): the dispatches of the symbol are numbered in
order and position = number mod period.

The file records the kernel symbol behind every entry and the sha256 of the kernel sources it was collected on
(build_hip.source_hash()); bench.py refuses the numbers when that hash differs from the build it runs.

usage: collect_traffic.py FETCH.db WRITE.db --config cfg2 --beam 1 [--mode decode|train|encoder] [--outdir profiles/traffic] [--md file.md]
"""
import argparse
import json
import os
import re
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cyclical-visual-captioning_amd"))


def roles_for(mode, beam, T, B=64):
    """role name (bench.py's launch name) -> (regex on the kernel symbol, period, set of positions inside the period)"""
    if mode == "encoder":
        # per-launch averages; bench.py multiplies by the launches per forward
        return {
            "cvc_gru_seq_persistent_fwd": (r"gru_persistent_kernel", 1, {0}),
            "cvc_tile_gemm": (r"tile_gemm_ld2?_kernel", 1, {0}),
            "cvc_tile_pack_rows_any": (r"tile_pack_rows_(any|blk)_kernel", 1, {0}),
        }
    if mode == "train" and 64 < 2 * B <= 128:
        # joint backward of both loops on the 128-row form of the backward-data product (two 64-row operand groups): per step the
        # language product and the attention product (missing at t = 0) on skinny_gemm_nn_split128_kernel, h2attn on the 64-row kernel
        lstm = r"skinny_gemm_packed_kernelILi\dELb1E"
        nn2 = r"skinny_gemm_nn_split128_kernel"
        pa = 2 * T - 1
        return {
            "loopA.fwd.att_cell": (lstm, 4 * T, {j for j in range(2 * T) if j % 2 == 0}),
            "loopA.fwd.lang_cell": (lstm, 4 * T, {j for j in range(2 * T) if j % 2 == 1}),
            "loopC.fwd.att_cell": (lstm, 4 * T, {2 * T + j for j in range(2 * T) if j % 2 == 0}),
            "loopC.fwd.lang_cell": (lstm, 4 * T, {2 * T + j for j in range(2 * T) if j % 2 == 1}),
            "loops.bwd.nn_lang": (nn2, pa, {j for j in range(pa) if j % 2 == 0}),
            "loops.bwd.nn_att": (nn2, pa, {j for j in range(pa) if j % 2 == 1}),
            "loopA.bwd.nn_h2attn": (r"skinny_gemm_nn_split_kernel", 1, {0}),
            "loopA.fwd.attn_scores": (r"attn_scores_kernelILi0ELi\d+ELi1E", 1, {0}),
            "loopA.fwd.attn_wsum": (r"attn_wsum_kernelI", 1, {0}),
            "loopA.bwd.attn_bwd": [(r"attn_scores_kernelILi1ELi\d+ELi1E", 1, {0}), (r"softmax_bwd2_kernel", 1, {0}), (r"attn_score_bwd2", 1, {0})],
            "cvc_tile_gemm": (r"tile_gemm_ld2?_kernel", 1, {0}),
            # entry points called from Python (round 6: every timed row of the bench line carries its PMC traffic): per-launch averages
            "cvc_adam_clip_step": [(r"optim_sumsq_kernel", 1, {0}), (r"optim_finalize_kernel", 1, {0}), (r"optim_adam_kernel", 1, {0})],
            "cvc_tile_pack_cols": (r"tile_pack_cols_kernel", 1, {0}),
            "cvc_tile_pack_rows_any": (r"tile_pack_rows_(any|blk)_kernel", 1, {0}),
            "cvc_pack_lstm_segs": (r"pack_lstm_w_kernel", 1, {0}),
            # the localizer's T-query attention backward (two launches per step: regions, frames): d_attn on the several-queries dot
            # score kernel, softmax backward, d_q as weighted rows
            "cvc_attn_bwd": [(r"attn_scores_kernelILi1ELi\d+ELi[45]E", 1, {0}), (r"softmax_bwd_kernel", 1, {0}), (r"attn_wsum_mq_kernel", 1, {0})],
        }
    if mode == "train" and 2 * B <= 64:
        # joint backward of both loops (cvc_train_loops_bwd_joint): T x (language product, h2attn product, attention product), the
        # last one missing at t = 0
        lstm = r"skinny_gemm_packed_kernelILi\dELb1E"
        nn = r"skinny_gemm_nn_split_kernel"
        pa = 3 * T - 1
        return {
            "loopA.fwd.att_cell": (lstm, 4 * T, {j for j in range(2 * T) if j % 2 == 0}),
            "loopA.fwd.lang_cell": (lstm, 4 * T, {j for j in range(2 * T) if j % 2 == 1}),
            "loopC.fwd.att_cell": (lstm, 4 * T, {2 * T + j for j in range(2 * T) if j % 2 == 0}),
            "loopC.fwd.lang_cell": (lstm, 4 * T, {2 * T + j for j in range(2 * T) if j % 2 == 1}),
            "loops.bwd.nn_lang": (nn, pa, {j for j in range(pa) if j % 3 == 0}),
            "loopA.bwd.nn_h2attn": (nn, pa, {j for j in range(pa) if j % 3 == 1}),
            "loops.bwd.nn_att": (nn, pa, {j for j in range(pa) if j % 3 == 2}),
            "loopA.fwd.attn_scores": (r"attn_scores_kernelILi0ELi\d+ELi1E", 1, {0}),
            "loopA.fwd.attn_wsum": (r"attn_wsum_kernelI", 1, {0}),
            # one role = three kernels per step: d_attn = C . d_ctx on the dot-score kernel, softmax backward, score backward
            "loopA.bwd.attn_bwd": [(r"attn_scores_kernelILi1ELi\d+ELi1E", 1, {0}), (r"softmax_bwd2_kernel", 1, {0}), (r"attn_score_bwd2", 1, {0})],
            "cvc_tile_gemm": (r"tile_gemm_ld2?_kernel", 1, {0}),
            # entry points called from Python (round 6: every timed row of the bench line carries its PMC traffic): per-launch averages
            "cvc_adam_clip_step": [(r"optim_sumsq_kernel", 1, {0}), (r"optim_finalize_kernel", 1, {0}), (r"optim_adam_kernel", 1, {0})],
            "cvc_tile_pack_cols": (r"tile_pack_cols_kernel", 1, {0}),
            "cvc_tile_pack_rows_any": (r"tile_pack_rows_(any|blk)_kernel", 1, {0}),
            "cvc_pack_lstm_segs": (r"pack_lstm_w_kernel", 1, {0}),
            # the localizer's T-query attention backward (two launches per step: regions, frames): d_attn on the several-queries dot
            # score kernel, softmax backward, d_q as weighted rows
            "cvc_attn_bwd": [(r"attn_scores_kernelILi1ELi\d+ELi[45]E", 1, {0}), (r"softmax_bwd_kernel", 1, {0}), (r"attn_wsum_mq_kernel", 1, {0})],
        }
    if mode == "train":
        # forward of a step: loop A = T x (attention cell, language cell), then loop C = T x (attention cell, language cell);
        # backward: loop C first (T x (language product, attention product), none for the attention cell at t = 0), then loop A
        # (T x (language product, h2attn product, attention product), the last one missing at t = 0)
        lstm = r"skinny_gemm_packed_kernelILi\dELb1E"
        nn = r"skinny_gemm_nn_split_kernel"
        pc = 2 * T - 1
        pa = 3 * T - 1
        return {
            "loopA.fwd.att_cell": (lstm, 4 * T, {j for j in range(2 * T) if j % 2 == 0}),
            "loopA.fwd.lang_cell": (lstm, 4 * T, {j for j in range(2 * T) if j % 2 == 1}),
            "loopC.fwd.att_cell": (lstm, 4 * T, {2 * T + j for j in range(2 * T) if j % 2 == 0}),
            "loopC.fwd.lang_cell": (lstm, 4 * T, {2 * T + j for j in range(2 * T) if j % 2 == 1}),
            "loopC.bwd.nn_lang": (nn, pc + pa, {j for j in range(pc) if j % 2 == 0}),
            "loopC.bwd.nn_att": (nn, pc + pa, {j for j in range(pc) if j % 2 == 1}),
            "loopA.bwd.nn_lang": (nn, pc + pa, {pc + j for j in range(pa) if j % 3 == 0}),
            "loopA.bwd.nn_h2attn": (nn, pc + pa, {pc + j for j in range(pa) if j % 3 == 1}),
            "loopA.bwd.nn_att": (nn, pc + pa, {pc + j for j in range(pa) if j % 3 == 2}),
            "loopA.fwd.attn_scores": (r"attn_scores_kernelILi0ELi\d+ELi1E", 1, {0}),
            "loopA.fwd.attn_wsum": (r"attn_wsum_kernelI", 1, {0}),
            # one role = three kernels per step: d_attn = C . d_ctx on the dot-score kernel, softmax backward, score backward
            "loopA.bwd.attn_bwd": [(r"attn_scores_kernelILi1ELi\d+ELi1E", 1, {0}), (r"softmax_bwd2_kernel", 1, {0}), (r"attn_score_bwd2", 1, {0})],
            "cvc_tile_gemm": (r"tile_gemm_ld2?_kernel", 1, {0}),
            # entry points called from Python (round 6: every timed row of the bench line carries its PMC traffic): per-launch averages
            "cvc_adam_clip_step": [(r"optim_sumsq_kernel", 1, {0}), (r"optim_finalize_kernel", 1, {0}), (r"optim_adam_kernel", 1, {0})],
            "cvc_tile_pack_cols": (r"tile_pack_cols_kernel", 1, {0}),
            "cvc_tile_pack_rows_any": (r"tile_pack_rows_(any|blk)_kernel", 1, {0}),
            "cvc_pack_lstm_segs": (r"pack_lstm_w_kernel", 1, {0}),
            # the localizer's T-query attention backward (two launches per step: regions, frames): d_attn on the several-queries dot
            # score kernel, softmax backward, d_q as weighted rows
            "cvc_attn_bwd": [(r"attn_scores_kernelILi1ELi\d+ELi[45]E", 1, {0}), (r"softmax_bwd_kernel", 1, {0}), (r"attn_wsum_mq_kernel", 1, {0})],
        }
    if beam > 1:
        # tile path: per decode one hoisted fc product, then per step the four products in this order
        tg = r"tile_gemm_ld2?_kernel"
        per = 1 + 4 * T
        pos = lambda k: {1 + 4 * t + k for t in range(T)}
        return {
            "att_lstm": (tg, per, pos(0)), "h2attn": (tg, per, pos(1)), "lang_lstm": (tg, per, pos(2)), "logits": (tg, per, pos(3)),
            "attn_scores": (r"attn_scores_kernel", 1, {0}),
            "attn_wsum": (r"attn_wsum(_mq)?_kernel", 1, {0}),
        }
    return {
        "att_lstm": (r"skinny_gemm_packed_kernelILi\dELb1E", 2, {0}),
        "lang_lstm": (r"skinny_gemm_packed_kernelILi\dELb1E", 2, {1}),
        "h2attn": (r"skinny_gemm_packed_kernelILi\dELb0E", 2, {0}),
        "logits": (r"skinny_gemm_packed_kernelILi\dELb0E", 2, {1}),
        "attn_scores": (r"attn_scores_kernel", 1, {0}),
        "attn_wsum": (r"attn_wsum_kernel", 1, {0}),
        "word_select": (r"top2_final_kernel", 1, {0}),
    }


def per_dispatch(path, counter):
    """[(kernel_name, value)] in dispatch order for one PMC counter."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    T = lambda p: next(t for t in tabs if t.startswith(p))
    kd, ks, pe, pi = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
    q = (f"select s.kernel_name, sum(e.value), d.start from {pe} e join {pi} p on e.pmc_id = p.id "
         f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id where p.name = ? "
         f"group by d.event_id order by d.start, d.event_id")
    return [(n, v) for n, v, _ in cur.execute(q, (counter,))]


def part_bytes(role, fetch, write, pat, period, positions, skip):
    """-> (symbol, dispatches, mean FETCH_SIZE KiB, mean WRITE_SIZE KiB) of the launches of /pat/ at `positions` of every period, or None"""
    rx = re.compile(pat)
    f = [(n, v) for n, v in fetch if rx.search(n)]
    w = [(n, v) for n, v in write if rx.search(n)]
    if not f:
        return None

    # one-off launches of the same kernel before the first period (the embedding-gate table is one dense product at bind time) are
    # dropped from the front; the two passes may hold different numbers of periods (bench.py's untimed settling replays are
    # time-bounded), so each is trimmed on its own
    def trim(xs, what):
        lead = len(xs) % period
        if lead > 2:
            print(f"[collect_traffic] {role}: {len(xs)} {what} dispatches of /{pat}/ are not a multiple of the period {period}: skipped", file=sys.stderr)
            return None
        if lead:
            print(f"[collect_traffic] {role}: {len(xs)} {what} dispatches of /{pat}/ = {lead} one-off launch(es) + {len(xs) // period} periods of {period}", file=sys.stderr)
        return xs[lead:]
    f, w = trim(f, "FETCH"), trim(w, "WRITE")
    if f is None or w is None:
        return None
    sel = lambda xs: [v for i, (n, v) in enumerate(xs) if i % period in positions and i // period >= skip]
    fsel, wsel = sel(f), sel(w)
    if not fsel:
        return None
    sym = next(n for i, (n, v) in enumerate(f) if i % period in positions)
    return sym, len(fsel), sum(fsel) / len(fsel), (sum(wsel) / len(wsel) if wsel else 0.0)


def workload_key(config, beam, mode):
    return f"{config}_beam{beam}_{mode}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_db")
    ap.add_argument("write_db")
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--beam", type=int, default=1)
    ap.add_argument("--mode", default="decode")
    ap.add_argument("--outdir", default=os.path.join(ROOT, "profiles", "traffic"))
    ap.add_argument("--md", default=None)
    ap.add_argument("--skip", type=int, default=0, help="ignore the first N periods of every role (warm-up)")
    a = ap.parse_args()
    import build_hip
    from cvc import synth
    T, B = synth.CONFIGS[a.config].T, synth.CONFIGS[a.config].B
    fetch, write = per_dispatch(a.fetch_db, "FETCH_SIZE"), per_dispatch(a.write_db, "WRITE_SIZE")
    out = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python3 bench.py ...; "
                   "hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch (MI355X_MICROARCH.md, HBM section: FETCH_SIZE "
                   "counts the 128-B requests of wide coalesced reads as 64 B on gfx950); tools/collect_traffic.py",
           "source_hash": build_hip.source_hash(),
           "workload": {"config": a.config, "beam": a.beam, "mode": a.mode},
           "kernels": {}}
    try:
        out["git_head"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        out["git_head"] = None
    rows = []
    for role, spec in roles_for(a.mode, a.beam, T, B).items():
        # a role is one kernel launch or, as a list, several launches whose bytes add up (an entry point made of several kernels)
        parts = [part_bytes(role, fetch, write, pat, period, positions, a.skip) for pat, period, positions in (spec if isinstance(spec, list) else [spec])]
        if any(p is None for p in parts):
            continue
        fk, wk = sum(p[2] for p in parts), sum(p[3] for p in parts)
        sym = " + ".join(p[0] for p in parts)
        n = min(p[1] for p in parts)
        out["kernels"][role] = {"symbol": sym, "dispatches": n, "FETCH_SIZE_KiB": round(fk, 1),
                                "WRITE_SIZE_KiB": round(wk, 1), "hbm_bytes": int(round((2 * fk + wk) * 1024))}
        rows.append((role, sym, n, fk, wk, (2 * fk + wk) * 1024))
    os.makedirs(a.outdir, exist_ok=True)
    path = os.path.join(a.outdir, workload_key(a.config, a.beam, a.mode) + ".json")
    json.dump(out, open(path, "w"), indent=1)
    md = [f"### {workload_key(a.config, a.beam, a.mode)}", "",
          "| launch | kernel symbol | dispatches | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB | HBM bytes per launch = (2F + W) x 1024 |",
          "|---|---|---:|---:|---:|---:|"]
    for role, sym, n, fk, wk, b in rows:
        md.append(f"| {role} | `{sym[:90]}`{' ...' if len(sym) > 90 else ''} | {n} | {fk:.1f} | {wk:.1f} | {b / 1e6:.1f} MB |")
    md.append("")
    md.append(f"source_hash {out['source_hash']}, git {out['git_head']}, workload {out['workload']}")
    text = "\n".join(md)
    print(text)
    if a.md:
        with open(a.md, "a") as fh:
            fh.write(text + "\n\n")


if __name__ == "__main__":
    main()
