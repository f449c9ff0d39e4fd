#!/usr/bin/env python3
"""VERDICT r04 item 3, as a direct measurement: would ONE leaf-hash launch over the same-stage trees of k provers (k x 2^18 leaves)
beat what throughput mode already does -- k provers' launches of 2^18 leaves each, in flight together on k streams?
For k = 1, 2, 4, 8, 16: (a) one launch over k * 2^18 leaves of width 32 on one stream; (b) k launches of 2^18 leaves on k contexts
(k host threads, k streams), enqueued together.  Rates in M leaves/s over the wall time of `reps` repetitions, buffers resident.
A batched launch pays off only if (a) is clearly above (b) at the k the bench runs (16).  usage: tools/batch_ab.py [log2_leaves=18]"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch  # noqa: E402

torch.cuda.is_available()
import provekit_amd  # noqa: E402
from provekit_amd._lib import lib  # noqa: E402
from provekit_amd.field import random_field  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n, width, reps = 1 << lg, 32, 20
ks = [1, 2, 4, 8, 16]
kmax = max(ks)
base = random_field(n * width, 3)
ctxs = [provekit_amd.Context(0) for _ in range(kmax)]
big_in = ctxs[0].alloc_fe(kmax * n * width)
for j in range(kmax):  # the same 2^18 x 32 matrix k times: the hash does not care
    ctxs[0].upload_into(big_in.ptr + 32 * j * n * width, base)
big_out = ctxs[0].alloc_fe(kmax * n)
ins = [ctxs[0].upload(base)] + [c.upload(base) for c in ctxs[1:]]
outs = [c.alloc_fe(n) for c in ctxs]
rows = []
for k in ks:
    c0 = ctxs[0]
    # (a) one launch over k * n leaves (column-major: column j of the k*n-leaf matrix is contiguous -- here the layout of the input does not
    # matter for the ALU-bound kernel; leaf-major keeps the k copies valid as ONE matrix of k*n leaves)
    for _ in range(2):
        c0._check(lib.pk_leaf_hash(c0.handle, big_in.ptr, k * n, width, 0, big_out.ptr))
    c0.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        c0._check(lib.pk_leaf_hash(c0.handle, big_in.ptr, k * n, width, 0, big_out.ptr))
    c0.sync()
    ta = time.perf_counter() - t0
    # (b) k streams, one launch of n leaves each, repeated
    gate = threading.Barrier(k + 1)

    def work(j):
        c = ctxs[j]
        for _ in range(2):
            c._check(lib.pk_leaf_hash(c.handle, ins[j].ptr, n, width, 0, outs[j].ptr))
        c.sync()
        gate.wait()
        for _ in range(reps):
            c._check(lib.pk_leaf_hash(c.handle, ins[j].ptr, n, width, 0, outs[j].ptr))
        c.sync()
        gate.wait()

    ths = [threading.Thread(target=work, args=(j,)) for j in range(k)]
    for t in ths:
        t.start()
    gate.wait()
    t0 = time.perf_counter()
    gate.wait()
    tb = time.perf_counter() - t0
    for t in ths:
        t.join()
    rows.append({"k": k, "leaves_per_launch_a": k * n, "one_launch_Mleaves_per_s": round(reps * k * n / ta / 1e6, 2),
                 "k_streams_Mleaves_per_s": round(reps * k * n / tb / 1e6, 2), "one_launch_over_k_streams": round(tb / ta, 3)})
    print(json.dumps(rows[-1]), flush=True)
print(json.dumps({"log2_leaves_per_tree": lg, "width": width, "layout": "leaf-major, Montgomery in (pk_leaf_hash)", "rows": rows}))
