#!/usr/bin/env python3
"""Concurrency statistics of a rocprofv3 kernel trace: how much of the wall-clock span has 0, 1, 2, ... kernels in flight,
and how much of it is covered by "wide" kernels (>= 1024 workgroups).  usage: tools/overlap.py <kernel_trace.csv> [t_skip_frac]"""
import csv
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
assert skip < 0.5
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) // max(int(r["Workgroup_Size_X"]), 1)
    ev.append((s, e, wg, r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", "")))
ev.sort()
# the steady-state window: between the kernels at the `skip` and 1-`skip` quantiles of launch order
lo, t1 = ev[int(skip * len(ev))][0], ev[int((1 - skip) * len(ev)) - 1][0]
ev = [e for e in ev if e[1] > lo and e[0] < t1]
pts = []
for s, e, wg, _, _ in ev:
    if e <= lo:
        continue
    s, e = max(s, lo), min(e, t1)
    pts.append((s, 1, wg))
    pts.append((e, -1, wg))
pts.sort()
depth, wide, last = 0, 0, lo
hist, wide_t = Counter(), 0
for t, d, wg in pts:
    hist[depth] += t - last
    if wide:
        wide_t += t - last
    last = t
    depth += d
    if wg >= 1024:
        wide += d
span = t1 - lo
print(f"span {span/1e6:.1f} ms, kernels {len(ev)}, queues {len(set(e[4] for e in ev))}")
for k in sorted(hist):
    print(f"  {k} kernels in flight: {100*hist[k]/span:5.1f} %")
print(f"  >=1 wide kernel (>=1024 workgroups) in flight: {100*wide_t/span:5.1f} %")
tot = sum(min(e[1], t1) - max(e[0], lo) for e in ev)
print(f"  sum of kernel durations / span = {tot/span:.2f}")
