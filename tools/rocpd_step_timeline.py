#!/usr/bin/env python3
"""Decode-step timeline from a rocprofv3 kernel trace (rocpd sqlite DB): for the most frequent repeating window of `period`
dispatches (the greedy step: 7 launches), the average duration of every position in the step and the average idle gap in front
of it (previous kernel's end -> this kernel's start).  Separates the two LSTM launches of the shared kernel symbol.
Usage: python tools/rocpd_step_timeline.py DB [period]"""
import sqlite3
import sys
from collections import Counter, defaultdict


def main(path, period=7):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    kcols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in kcols else ("display_name" if "display_name" in kcols else kcols[-1])
    rows = list(cur.execute(f"select s.{name_col}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    names = [r[0] for r in rows]
    # most frequent window of `period` consecutive kernel names
    win = Counter(tuple(names[i:i + period]) for i in range(len(names) - period))
    # rotate to the canonical start: prefer the window that repeats back to back most often
    best, _ = max(win.items(), key=lambda kv: kv[1])
    dur, gap, n = defaultdict(float), defaultdict(float), 0
    i = 0
    while i + period <= len(names):
        if tuple(names[i:i + period]) == best and i > 0:
            for p in range(period):
                dur[p] += rows[i + p][2] - rows[i + p][1]
                gap[p] += rows[i + p][1] - rows[i + p - 1][2]
            n += 1
            i += period
        else:
            i += 1
    print(f"windows matched: {n}")
    print("| pos | kernel | avg us | gap before us |")
    print("|---:|---|---:|---:|")
    tot = 0.0
    for p in range(period):
        nm = best[p] if len(best[p]) < 90 else best[p][:87] + "..."
        print(f"| {p} | `{nm}` | {dur[p] / n / 1e3:.2f} | {gap[p] / n / 1e3:.2f} |")
        tot += (dur[p] + gap[p]) / n / 1e3
    print(f"step = {tot:.2f} us")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 7)
