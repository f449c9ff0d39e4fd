#!/usr/bin/env python3
"""Per-proof kernel time from a rocprofv3 kernel trace of `bench.py --concurrency 1`: usage tools/kernel_breakdown.py <trace.csv> <n_proofs>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
agg = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in rows:
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    if "_rate_kernel" in k or "twiddle" in k:  # the multiplier-peak probes and the one-time table builds are not part of a proof
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[k][0] += 1
    agg[k][1] += d
    tot += d
print(f"kernel time per proof: {tot / n:.1f} us in {sum(v[0] for v in agg.values()) / n:.1f} dispatches")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[:52]:54s}{v[0] / n:7.1f} x {v[1] / v[0]:8.1f} us = {v[1] / n:8.1f} us/proof")
