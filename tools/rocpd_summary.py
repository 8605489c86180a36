#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace into the per-kernel stats table we commit under
profiles/:   python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.md"""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:110]


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    ncol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {ncol}, start, end from kernels").fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print(f"| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{k}` | {a[0]} | {a[1] / 1e3:.2f} | {a[1] / a[0]:.1f} | {a[2]:.1f} | {a[3]:.1f} | {100 * a[1] / total:.1f} |")
    print(f"\ntotal kernel time {total / 1e3:.2f} ms over {sum(a[0] for a in agg.values())} dispatches")


if __name__ == "__main__":
    main(sys.argv[1])
