#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max ns, share) from a rocprofv3 rocpd sqlite DB
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes NAME_results.db on ROCm 7.2).
Usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/r01_kernel_stats.md"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    kcols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in kcols else ("display_name" if "display_name" in kcols else kcols[-1])
    q = (f"select s.{name_col}, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
         f"max(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc")
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | share |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, n, tot, avg, mn, mx in rows:
        short = name if len(name) < 110 else name[:107] + "..."
        print(f"| `{short}` | {n} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.1f}% |")


if __name__ == "__main__":
    main(sys.argv[1])
