"""Development aid: isolated timings of the hashing entry points (leaf hash, inner tree, compress_many, PoW) for the
library selected by PK_LIB_PATH (A/B against another build)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import provekit_amd
from provekit_amd._lib import PK_COL_MAJOR, lib
from provekit_amd.field import random_field

ctx = provekit_amd.Context(0)


def best(fn, reps=7):
    fn()
    ctx.sync()
    ts = []
    for _ in range(reps):
        ctx.timer_start()
        fn()
        ts.append(ctx.timer_stop())
    return min(ts)


print("lib:", provekit_amd.LIB_PATH)
for logn, w in [(18, 32), (17, 16), (14, 16), (20, 32)]:
    nl = 1 << logn
    cols = ctx.upload(random_field(nl * w, 1))
    dig = ctx.alloc_fe(nl)
    t = best(lambda: ctx._check(lib.pk_leaf_hash(ctx.handle, cols.ptr, nl, w, PK_COL_MAJOR, dig.ptr)))
    print(f"leaf_hash 2^{logn} x {w}: {t:.4f} ms  {nl * (w - 1) / t / 1e6:.2f} Gcompress/s")
    nodes = ctx.alloc_fe(2 * nl)
    t = best(lambda: ctx._check(lib.pk_merkle_inner(ctx.handle, nodes.ptr, nl)))
    print(f"merkle_inner 2^{logn}: {t:.4f} ms")
n = 1 << 22
msgs = ctx.upload(np.random.default_rng(0).integers(0, 2**63, size=(2 * n, 4), dtype=np.uint64))
out = ctx.alloc_fe(n)
t = best(lambda: ctx._check(lib.pk_compress_many(ctx.handle, msgs.ptr, out.ptr, n)))
print(f"compress_many 2^22: {t:.4f} ms  {n / t / 1e6:.2f} Gcompress/s")
for bits in (11.0, 16.0, 19.0):
    rng = np.random.default_rng(int(bits))
    tot, k = 0.0, 24
    for i in range(k):
        ch = rng.integers(0, 2**62, size=4, dtype=np.uint64)
        nonce = C.c_uint64()
        t0 = time.perf_counter()
        ctx._check(lib.pk_pow_solve(ctx.handle, ch.ctypes.data, bits, C.byref(nonce)))
        tot += time.perf_counter() - t0
    print(f"pow_solve {bits} bits: {1e3 * tot / k:.3f} ms mean wall over {k} challenges")
