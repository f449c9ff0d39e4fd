#!/usr/bin/env python3
"""the Fiat-Shamir round trip, launch + synchronise (pk_prove today) vs a persistent kernel's pinned mailbox.
GPU box.  Writes gpurun_out/r04_roundtrip.json."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import provekit_amd  # noqa: E402
from provekit_amd._lib import lib  # noqa: E402
from tools.pk_probes import lib as probes

ctx = provekit_amd.Context(0)
rows = []
for work in (0, 5):
    for rep in range(3):
        a, b = C.c_double(), C.c_double()
        ctx._check(probes.pk_probe_roundtrip(ctx.handle, 2000, work, C.byref(a), C.byref(b)))
        rows.append({"host_work_permutes": work, "us_per_round_launch_sync": a.value, "us_per_round_mailbox": b.value})
chain = []
for threads in (64, 64 * 1024):
    v = C.c_double()
    ctx._check(probes.pk_probe_launch_chain(ctx.handle, 5000, threads, C.byref(v)))
    chain.append({"threads_per_launch": threads, "us_per_dependent_launch_no_host": v.value})
best = lambda k, w: min(r[k] for r in rows if r["host_work_permutes"] == w)
launch, mailbox = best("us_per_round_launch_sync", 5), best("us_per_round_mailbox", 5)
res = {"probe": "pk_probe_roundtrip: 2000 dependent round trips, one workgroup, idle chip", "runs": rows,
       "us_per_round_launch_sync": launch, "us_per_round_mailbox": mailbox, "saving_us_per_round": launch - mailbox,
       "back_to_back_dependent_launches": chain}
print(json.dumps(res, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r04_roundtrip.json"), "w"), indent=1)
