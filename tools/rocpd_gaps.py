#!/usr/bin/env python3
"""Where does the GPU idle?  From a rocprofv3 rocpd DB: over the last `window_ms` of the trace, the busy time (union of kernel
intervals), the idle time, and the idle time booked on the kernel that FOLLOWS each gap (top entries) -- the launch the GPU waited for.
Usage: python tools/rocpd_gaps.py x_results.db [window_ms]"""
import sqlite3
import sys
from collections import defaultdict


def main(path, window_ms=400.0):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    kcols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in kcols else kcols[-1]
    rows = list(cur.execute(f"select d.start, d.end, s.{name_col} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    t_end = max(r[1] for r in rows)
    t0 = t_end - window_ms * 1e6
    rows = [r for r in rows if r[0] >= t0]
    busy, idle, cover = 0, 0, rows[0][0]
    after = defaultdict(lambda: [0, 0])
    for s, e, name in rows:
        if s > cover:
            idle += s - cover
            a = after[name[:70]]
            a[0] += s - cover; a[1] += 1
            cover = s
        if e > cover:
            busy += e - cover
            cover = e
    span = cover - rows[0][0]
    print(f"window {span / 1e6:.1f} ms, {len(rows)} launches: busy {busy / 1e6:.1f} ms ({100 * busy / span:.1f} %), idle {idle / 1e6:.1f} ms; "
          f"sum of kernel durations {sum(e - s for s, e, _ in rows) / 1e6:.1f} ms (overlap = concurrent kernels)")
    print("| idle before this kernel | total ms | gaps | avg us |")
    print("|---|---:|---:|---:|")
    for name, (tot, n) in sorted(after.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"| `{name}` | {tot / 1e6:.2f} | {n} | {tot / n / 1e3:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 400.0)
