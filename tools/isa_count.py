#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc -S listing (development aid for DESIGN.md's ISA counts).
usage: isa_count.py file.s kernel-name-substring [top]"""
import collections
import re
import sys

src, pat = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
cur, hist, label = None, collections.Counter(), None
for line in open(src):
    m = re.match(r"^(\w+):", line)
    if m:
        cur = m.group(1) if pat in m.group(1) else None
        if cur:
            label = cur
        continue
    if cur is None:
        continue
    if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
        cur = None
        continue
    t = line.strip()
    if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
        continue
    hist[t.split()[0]] += 1
total = sum(hist.values())
print(label, "total", total)
for k, v in hist.most_common(top):
    print(f"  {k:28s} {v}")
