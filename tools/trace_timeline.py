#!/usr/bin/env python
"""Concurrency / idle-time analysis of a rocprofv3 kernel trace (ROCm 7.2 rocpd sqlite) of bench.py:

    python tools/trace_timeline.py gpurun_out/prof/x_results.db [n_steps] > profiles/rNN_timeline.md

The per-kernel stats table (tools/rocpd_summary.py) sums kernel durations; in the two-stream step that sum exceeds the step
time, and it cannot say whether the GPU ever sits idle (dependent launches of short kernels) or how much of the step runs with
only one kernel resident.  Here the dispatch intervals [start, end) of the steady-state steps are swept in time order:
  * busy-0 time   = no kernel executing (launch gaps, host stalls, stream joins),
  * busy-1 time   = exactly one kernel executing (no overlap partner: its idle CUs are wasted unless it fills the chip),
  * busy-2+ time  = two or more kernels executing,
per step, plus the largest idle gaps with the kernels before / after them and the busy-1 time by kernel name.
Steps are delimited by `step_advance_kernel` (one launch per optimisation step)."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def main(path, n_steps=3):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    ncol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    qcol = next((c for c in ("queue_id", "stream_id", "queue", "stream") if c in cols), None)
    gcol = "grid_x" if "grid_x" in cols and "workgroup_x" in cols else None
    sel = (f"select {ncol}, start, end" + (f", {qcol}" if qcol else ", 0") + (", grid_x, workgroup_x" if gcol else ", 0, 1") +
           " from kernels order by start")
    rows = []
    for r in cur.execute(sel):
        n = short(r[0])
        if gcol and n.startswith(("gemm_", "attn_")):   # workgroup count tells the shapes of one kernel apart
            n += f" [{int(r[4]) // max(1, int(r[5]))} wgs]"
        rows.append((n, int(r[1]), int(r[2]), r[3]))
    marks = [i for i, r in enumerate(rows) if r[0].startswith("step_advance_kernel")]
    dur = defaultdict(lambda: [0, 0.0])
    if len(marks) < 2:
        print(f"columns: {cols}\nonly {len(marks)} step_advance_kernel launches: need >= 2 steps")
        return
    n_steps = min(n_steps, len(marks) - 1)
    lo, hi = marks[-1 - n_steps], marks[-1]          # the last n_steps full steps (from the end of a step to the end of a later one)
    t_lo, t_hi = rows[lo][2], rows[hi][2]
    win = [r for r in rows if r[2] > t_lo and r[1] < t_hi]
    span = (t_hi - t_lo) / 1e6
    # sweep
    ev = []
    for i, (n, s, e, q) in enumerate(win):
        ev.append((max(s, t_lo), 1, i))
        ev.append((min(e, t_hi), -1, i))
    ev.sort(key=lambda x: (x[0], x[1]))
    active = set()
    t_prev = t_lo
    busy = defaultdict(float)
    solo = defaultdict(float)
    gaps = []
    last_ended = None
    for t, d, i in ev:
        dt = t - t_prev
        if dt > 0:
            k = min(len(active), 2)
            busy[k] += dt
            if len(active) == 0:
                gaps.append((dt, t_prev, last_ended))
            elif len(active) == 1:
                solo[win[next(iter(active))][0]] += dt
        if d == 1:
            if not active and gaps and gaps[-1][1] + gaps[-1][0] == t:
                gaps[-1] = gaps[-1] + (win[i][0],)
            active.add(i)
        else:
            active.discard(i)
            last_ended = win[i][0]
        t_prev = t
    per_q = defaultdict(float)
    for n, s, e, q in win:
        per_q[q] += (min(e, t_hi) - max(s, t_lo))
        dur[n][0] += 1
        dur[n][1] += e - s
    print(f"# Kernel-timeline concurrency over the last {n_steps} steps of the trace ({span / n_steps:.2f} ms per step under the profiler)\n")
    print(f"columns of the kernels view: {', '.join(cols)}\n")
    print("| state | ms per step | share |")
    print("|---|---|---|")
    for k, lab in ((0, "no kernel executing"), (1, "exactly one kernel executing"), (2, "two or more kernels executing")):
        print(f"| {lab} | {busy[k] / 1e6 / n_steps:.2f} | {100 * busy[k] / (t_hi - t_lo):.1f} % |")
    print(f"\nkernel launches per step: {len(win) / n_steps:.0f}; summed kernel time per step: {sum(per_q.values()) / 1e6 / n_steps:.2f} ms")
    if qcol:
        print(f"\nbusy time by {qcol} (ms per step): " + ", ".join(f"{q}: {v / 1e6 / n_steps:.2f}" for q, v in sorted(per_q.items(), key=lambda kv: -kv[1])))
    print("\n## Idle gaps (no kernel executing)\n")
    hist = defaultdict(lambda: [0, 0.0])
    for g in gaps:
        us = g[0] / 1e3
        b = "< 2 us" if us < 2 else "2-5 us" if us < 5 else "5-20 us" if us < 20 else "20-100 us" if us < 100 else ">= 100 us"
        hist[b][0] += 1
        hist[b][1] += us
    print("| gap length | count per step | ms per step |")
    print("|---|---|---|")
    for b in ("< 2 us", "2-5 us", "5-20 us", "20-100 us", ">= 100 us"):
        print(f"| {b} | {hist[b][0] / n_steps:.1f} | {hist[b][1] / 1e3 / n_steps:.3f} |")
    print("\nLargest gaps (us, kernel that ended before -> kernel that started after):\n")
    for g in sorted(gaps, key=lambda x: -x[0])[:15]:
        after = g[3] if len(g) > 3 else "?"
        print(f"* {g[0] / 1e3:8.1f} us   `{g[2]}` -> `{after}`")
    print("\n## Time with exactly ONE kernel executing, by kernel (ms per step)\n")
    print("| kernel [workgroups] | ms per step alone | launches per step | mean duration us | summed duration ms per step |")
    print("|---|---|---|---|---|")
    for n, v in sorted(solo.items(), key=lambda kv: -kv[1])[:28]:
        c, d = dur[n]
        print(f"| `{n}` | {v / 1e6 / n_steps:.2f} | {c / n_steps:.1f} | {d / c / 1e3:.1f} | {d / 1e6 / n_steps:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
