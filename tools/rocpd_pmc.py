#!/usr/bin/env python3
"""Per-kernel average of every PMC counter in a rocprofv3 rocpd sqlite DB
(`rocprofv3 --pmc FETCH_SIZE --kernel-trace -d DIR -o NAME -- cmd`).  Prints a markdown table
and, with --json OUT, a {kernel_short_name: {counter: avg}} file.
Usage: python tools/rocpd_pmc.py x_results.db [--json profiles/pmc.json]"""
import json
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"(attn_scores_kernel|attn_wsum_kernel|skinny_gemm_ring_kernelILi\d+ELb[01]|skinny_gemm_kernelILi\d+ELi\d+ELb[01]|"
                  r"top2_final_kernel|top2_unk_kernel|embed_relu_fwd_kernel|beam_select_kernel|gather_rows_kernel)", name)
    return m.group(1) if m else name[:60]


def main(path, out_json=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    T = lambda p: next(t for t in tabs if t.startswith(p))
    kd, ks, pe, pi = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
    q = (f"select s.kernel_name, p.name, count(*), avg(e.value), sum(e.value) from {pe} e "
         f"join {pi} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id "
         f"group by s.kernel_name, p.name order by 5 desc")
    res = {}
    print("| kernel | counter | dispatches | avg per dispatch |")
    print("|---|---|---:|---:|")
    for name, ctr, n, avg, tot in cur.execute(q):
        print(f"| `{short(name)}` | {ctr} | {n} | {avg:.1f} |")
        res.setdefault(short(name), {})[ctr] = avg
    if out_json:
        json.dump(res, open(out_json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--json" else None)
