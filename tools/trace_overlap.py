#!/usr/bin/env python3
"""Timeline figures from a rocprofv3 kernel trace (CSV with Start_Timestamp / End_Timestamp per dispatch): for the densest window of
the run (the timed, pipelined steps), how much of the time at least one kernel was executing, how much two or more were, and the
per-kernel share.  usage: trace_overlap.py <kernel_trace.csv> [window_ms]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    rows = []
    with open(path) as f:
        for d in csv.DictReader(f):
            rows.append((int(d["Start_Timestamp"]), int(d["End_Timestamp"]), d["Kernel_Name"].split("(")[0]))
    rows.sort()
    win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 60e6
    # the densest window: the one with the largest sum of kernel durations among windows starting at a dispatch
    t0 = rows[0][0]
    nb = int((max(r[1] for r in rows) - t0) // 1e6) + 2
    bins = [0.0] * nb   # kernel-nanoseconds per millisecond
    for a, b, k in rows:
        i = int((a - t0) // 1e6)
        bins[i] += b - a
    w = int(win // 1e6)
    best, acc = 0, sum(bins[:w])
    best_acc = acc
    for i in range(1, max(1, nb - w)):
        acc += bins[i + w - 1] - bins[i - 1]
        if acc > best_acc:
            best, best_acc = i, acc
    lo = t0 + best * 1e6
    rows = [r for r in rows if r[0] < lo + win]
    sel = [(max(a, lo), min(b, lo + win), k) for a, b, k in rows if b > lo]
    ev = []
    for a, b, k in sel:
        ev.append((a, 1))
        ev.append((b, -1))
    ev.sort()
    depth, last, busy1, busy2 = 0, lo, 0, 0
    for t, d in ev:
        if depth >= 1:
            busy1 += t - last
        if depth >= 2:
            busy2 += t - last
        depth += d
        last = t
    per = defaultdict(float)
    for a, b, k in sel:
        per[k] += b - a
    print(f"densest {win / 1e6:.1f} ms window of the run: some kernel executing {100 * busy1 / win:.1f} %, two or more {100 * busy2 / win:.1f} %, "
          f"sum of kernel durations {sum(per.values()) / win:.3f} x the window")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:12]:
        print(f"  {k[:60]:60s} {100 * v / win:6.1f} % of the window")


if __name__ == "__main__":
    main()
