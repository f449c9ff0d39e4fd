"""Development aid: isolated timings of the Reed-Solomon encode (pk_rs_encode: de-interleave + the column NTTs) and of the plain
NTT for the library selected by PK_LIB_PATH (A/B against another build), with a digest of every output so two builds can be
compared bit for bit.  One JSON line.  usage: ntt_ab.py [n_vars ...]   (default 21 23 25 26; batch 2, rate 1/2, fold 16)"""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import provekit_amd
from provekit_amd.field import random_field
from provekit_amd.rs import rs_encode_device

ctx = provekit_amd.Context(0)
sizes = [int(a) for a in sys.argv[1:]] or [21, 23, 25, 26]
out = {"lib": os.path.basename(provekit_amd.LIB_PATH)}
for n in sizes:
    batch, fold, rate = 2, 4, 1
    rows, w = 1 << (n + rate - fold), batch << fold
    polys = [ctx.upload(random_field(1 << n, 100 + b)) for b in range(batch)]
    leaves, scratch = ctx.alloc_fe(rows * w), ctx.alloc_fe(2 * rows * w)
    run = lambda: rs_encode_device(ctx, [p.ptr for p in polys], n, rate, fold, leaves.ptr, scratch.ptr)
    run()
    ctx.sync()
    ts = []
    for _ in range(7 if n < 25 else 4):
        ctx.timer_start()
        run()
        ts.append(ctx.timer_stop())
    # digest: the whole matrix up to 2^21 leaves-elements, a strided sample of rows of every column above that
    if rows * w <= 1 << 24:
        h = hashlib.sha256(ctx.download_fe(leaves, rows * w).tobytes()).hexdigest()[:16]
    else:
        hh = hashlib.sha256()
        for c in range(0, w, 5):
            hh.update(ctx.download_fe(leaves.ptr + 32 * (c * rows + (c * 7919) % (rows - 4096)), 4096).tobytes())
        h = hh.hexdigest()[:16]
    out[f"rs_encode_2^{n}"] = {"ms": round(min(ts), 4), "median_ms": round(sorted(ts)[len(ts) // 2], 4), "digest": h}
    del polys, leaves, scratch
print(json.dumps(out))
