"""Quick device-side timing of compress_many / merkle commit (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import provekit_amd
from provekit_amd._lib import lib, PK_COL_MAJOR
from provekit_amd.field import random_field

ctx = provekit_amd.Context(0)
n = 1 << 22
msgs = ctx.upload(np.random.default_rng(0).integers(0, 2**63, size=(2 * n, 4), dtype=np.uint64))
out = ctx.alloc_fe(n)
for _ in range(2):
    ctx._check(lib.pk_compress_many(ctx.handle, msgs.ptr, out.ptr, n))
ctx.sync()
ts = []
for _ in range(5):
    ctx.timer_start(); ctx._check(lib.pk_compress_many(ctx.handle, msgs.ptr, out.ptr, n)); ts.append(ctx.timer_stop())
t = min(ts)
print(f"compress_many n=2^22: {t:.3f} ms  {n/t/1e6:.2f} Gcompress/s  {96*n/t/1e6:.1f} GB/s")
for logn, w in [(18, 32), (17, 16), (20, 32)]:
    nl = 1 << logn
    cols = ctx.upload(random_field(nl * w, 1))
    nodes = ctx.alloc_fe(2 * nl)
    ctx._check(lib.pk_merkle_commit(ctx.handle, cols.ptr, nl, w, PK_COL_MAJOR, nodes.ptr)); ctx.sync()
    ts = []
    for _ in range(5):
        ctx.timer_start(); ctx._check(lib.pk_merkle_commit(ctx.handle, cols.ptr, nl, w, PK_COL_MAJOR, nodes.ptr)); ts.append(ctx.timer_stop())
    t = min(ts)
    nc = nl * (w - 1) + nl - 1
    print(f"merkle_commit 2^{logn} x {w}: {t:.3f} ms  {nc/t/1e6:.2f} Gcompress/s")

from provekit_amd.rs import rs_encode_device
for n in (21, 17):
    rows = 1 << (n + 1 - 4)
    polys = [ctx.upload(random_field(1 << n, 3 + b)) for b in range(2)]
    leaves = ctx.alloc_fe(rows * 32); scratch = ctx.alloc_fe(2 * rows * 32)
    rs_encode_device(ctx, [p.ptr for p in polys], n, 1, 4, leaves.ptr, scratch.ptr); ctx.sync()
    ts = []
    for _ in range(5):
        ctx.timer_start(); rs_encode_device(ctx, [p.ptr for p in polys], n, 1, 4, leaves.ptr, scratch.ptr); ts.append(ctx.timer_stop())
    t = min(ts)
    print(f"rs_encode batch2 n={n}: {t:.3f} ms  ({32*rows*32*2/t/1e6:.1f} GB/s algorithmic r+w of leaves)")
