#!/usr/bin/env python3
"""Matrix-pipe busy per LAUNCH ROLE from a rocprofv3 PMC pass (`--kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
GRBM_GUI_ACTIVE -- python3 bench.py ...`, counters only -- no trace domains beside it):

    MFMA busy = sum over the chip's SIMDs of SQ_VALU_MFMA_BUSY_CYCLES  /  (GRBM_GUI_ACTIVE per XCD x 4 SIMDs x CUs)

(rocprof's MfmaUtil; SQ_VALU_MFMA_BUSY_CYCLES counts cycles -- 32 per v_mfma_f32_32x32x16_bf16 -- MI355X_MICROARCH.md).  A role
is (kernel symbol, position in the fixed launch order), as in tools/collect_traffic.py: the two gate GEMMs of a decode step, the four
tile GEMMs of a beam step, the products of a training step are told apart.  `MFMA peak share` = busy x (bf16 dense peak): what the
busy cycles amount to against the 2.5 PFLOP/s the pipe delivers when it never idles.

usage: rocpd_mfma_busy.py PMC.db --config cfg2 --beam 1 --mode decode [--cus 256]"""
import argparse
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "cyclical-visual-captioning_amd")]
from collect_traffic import roles_for          # noqa: E402


def per_dispatch(path, counter):
    """[(kernel_name, sum over instances, instances)] in dispatch order"""
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    T = lambda p: next(t for t in tabs if t.startswith(p))
    kd, ks, pe, pi = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
    q = (f"select s.kernel_name, sum(e.value), count(*), d.start from {pe} e join {pi} p on e.pmc_id = p.id "
         f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id where p.name = ? "
         f"group by d.event_id order by d.start, d.event_id")
    return [(n, v, c) for n, v, c, _ in cur.execute(q, (counter,))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--beam", type=int, default=1)
    ap.add_argument("--mode", default="decode")
    ap.add_argument("--cus", type=int, default=256)
    a = ap.parse_args()
    from cvc import synth
    d = synth.CONFIGS[a.config]
    mfma, sq, gui = (per_dispatch(a.db, c) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))
    print(f"### {a.config} beam={a.beam} {a.mode}: matrix-pipe busy per launch role\n")
    print("| launch role | kernel symbol | dispatches | SQ_VALU_MFMA_BUSY_CYCLES (sum over SIMDs) | GRBM_GUI_ACTIVE (per XCD) | "
          "MFMA busy = busy / (active x 4 SIMDs x CUs) |")
    print("|---|---|---:|---:|---:|---:|")
    for role, spec in roles_for(a.mode, a.beam, d.T, d.B).items():
        for pat, period, positions in (spec if isinstance(spec, list) else [spec]):
            rx = re.compile(pat)
            m = [(n, v, c) for n, v, c in mfma if rx.search(n)]
            g = [(n, v, c) for n, v, c in gui if rx.search(n)]
            if not m or not g or len(m) != len(g):
                continue
            lead = len(m) % period
            m, g = m[lead:], g[lead:]
            idx = [i for i in range(len(m)) if i % period in positions]
            if not idx:
                continue
            busy = sum(m[i][1] for i in idx) / len(idx)
            act = sum(g[i][1] / g[i][2] for i in idx) / len(idx)
            frac = busy / (act * 4 * a.cus) if act > 0 else 0.0
            sym = m[idx[0]][0]
            sym = re.sub(r"^_ZN?\d*(_GLOBAL__N_1)?\d*", "", sym)[:70]
            print(f"| {role} | `{sym}` | {len(idx)} | {busy:,.0f} | {act:,.0f} | **{frac:.3f}** |")
    print(f"\n(CUs = {a.cus}; bf16 dense peak 2.5 PFLOP/s = every SIMD's pipe busy every cycle; split-product kernels spend six MFMAs per "
          "fp32 product, so their fp32-equivalent rate is busy x 417 TFLOP/s)")


if __name__ == "__main__":
    main()
