#!/usr/bin/env python3
"""profiles/traffic.json from two rocprofv3 PMC passes over bench.py (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs,
`--kernel-trace` only, eager launches: `bench.py --no-graph`), as /opt/skills/guides/MI355X_MICROARCH.md prescribes:

    hbm_bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024

FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B), both counters are in KiB.

Launches of one kernel symbol that play different roles in a decode step (the packed LSTM kernel runs the attention LSTM and
the language LSTM; the packed linear kernel runs h2attn and the vocabulary projection) are separated by their position in the
dispatch sequence, which is fixed: --roles "skinny_gemm_packed_kernel<…true…>=att_lstm,lang_lstm".

The file records the kernel symbol behind every entry and the sha256 of the kernel sources it was collected on
(build_hip.source_hash()); bench.py refuses the numbers when that hash differs from the build it runs.

usage: collect_traffic.py FETCH.db WRITE.db --config cfg2 --beam 1 [--mode decode] [--out profiles/traffic.json] [--md profiles/rNN_pmc.md]
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

# launch name in bench.py -> (regex on the demangled-ish kernel symbol, position among that symbol's dispatches per cycle, cycle length)
DECODE_ROLES = {
    "att_lstm": (r"skinny_gemm_packed_kernelILi\dELb1E", 0, 2),
    "lang_lstm": (r"skinny_gemm_packed_kernelILi\dELb1E", 1, 2),
    "h2attn": (r"skinny_gemm_packed_kernelILi\dELb0E", 0, 2),
    "logits": (r"skinny_gemm_packed_kernelILi\dELb0E", 1, 2),
    "attn_scores": (r"attn_scores_kernel", 0, 1),
    "attn_wsum": (r"attn_wsum_kernel", 0, 1),
    "word_select": (r"top2_final_kernel", 0, 1),
}


# --mode train: C-ABI entry point of bench.py's training table -> kernel symbol; every dispatch of the symbol is averaged (the
# entry points run the same kernel on a few shapes per step: both cells, both loops), as bench.py averages their durations
TRAIN_ROLES = {
    "cvc_linear_nn_fwd": (r"skinny_gemm_nn_split_kernel", 0, 1),
    "cvc_packed_lstm_train_fwd": (r"skinny_gemm_packed_kernelILi\dELb1E", 0, 1),
    "cvc_tile_gemm": (r"tile_gemm_ld2?_kernel", 0, 1),
    "cvc_attn_score_bwd": (r"attn_score_bwd_kernel", 0, 1),
    "cvc_attn_wsum": (r"attn_wsum_kernel", 0, 1),
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_db")
    ap.add_argument("write_db")
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--beam", type=int, default=1)
    ap.add_argument("--mode", default="decode")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "traffic.json"))
    ap.add_argument("--md", default=None)
    ap.add_argument("--skip-decodes", type=int, default=0, help="ignore the dispatches of the first N role cycles (warm-up)")
    a = ap.parse_args()
    import build_hip
    fetch, write = per_dispatch(a.fetch_db, "FETCH_SIZE"), per_dispatch(a.write_db, "WRITE_SIZE")
    out = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python3 bench.py --no-graph ...; "
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
    for role, (pat, pos, cyc) in (TRAIN_ROLES if a.mode == "train" else DECODE_ROLES).items():
        rx = re.compile(pat)
        f = [(n, v) for n, v in fetch if rx.search(n)]
        w = [(n, v) for n, v in write if rx.search(n)]
        if not f:
            continue
        fsel = [v for i, (n, v) in enumerate(f) if i % cyc == pos][a.skip_decodes:]
        wsel = [v for i, (n, v) in enumerate(w) if i % cyc == pos][a.skip_decodes:]
        if not fsel:
            continue
        fk, wk = sum(fsel) / len(fsel), (sum(wsel) / len(wsel) if wsel else 0.0)
        sym = f[pos][0]
        out["kernels"][role] = {"symbol": sym, "dispatches": len(fsel), "FETCH_SIZE_KiB": round(fk, 1),
                                "WRITE_SIZE_KiB": round(wk, 1), "hbm_bytes": int(round((2 * fk + wk) * 1024))}
        rows.append((role, sym, len(fsel), fk, wk, (2 * fk + wk) * 1024))
    json.dump(out, open(a.out, "w"), indent=1)
    md = ["| launch | kernel symbol | dispatches | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB | HBM bytes per launch = (2F + W) x 1024 |",
          "|---|---|---:|---:|---:|---:|"]
    for role, sym, n, fk, wk, b in rows:
        md.append(f"| {role} | `{sym[:90]}` | {n} | {fk:.1f} | {wk:.1f} | {b / 1e6:.1f} MB |")
    md.append("")
    md.append(f"source_hash {out['source_hash']}, git {out['git_head']}, workload {out['workload']}")
    text = "\n".join(md)
    print(text)
    if a.md:
        open(a.md, "w").write(text + "\n")


if __name__ == "__main__":
    main()
