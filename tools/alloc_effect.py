#!/usr/bin/env python3
"""Why does a 2^26 commit run 25 % slower after other large buffers were allocated and released in the same process?  (bench.py runs its
size-class probes in fresh processes because of it.)  Times the commit (a) in a clean process, (b) right after allocating and freeing
`GB` gigabytes through the library, (c) again after a pause, (d) with its own buffers re-allocated.  GPU box."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import provekit_amd
from provekit_amd._lib import lib

GB = int(sys.argv[1]) if len(sys.argv) > 1 else 150
ctx = provekit_amd.Context(0)
n_vars = 26
n = 1 << n_vars

def make():
    polys = [torch.randint(0, 2**62, (n, 4), dtype=torch.int64, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(17 + b)) for b in range(2)]
    for t in polys:
        t[:, 3] &= (1 << 60) - 1
    torch.cuda.synchronize()
    szs = [C.c_size_t() for _ in range(3)]
    ctx._check(lib.pk_commit_sizes(ctx.handle, 2, n_vars, 1, 4, *[C.byref(x) for x in szs]))
    bufs = [ctx.alloc_fe(x.value) for x in szs]
    return polys, bufs

def commit(polys, bufs, reps=4):
    ptrs = (C.c_void_p * 2)(*[int(t.data_ptr()) for t in polys])
    root = (C.c_uint8 * 32)()
    ms = []
    for _ in range(reps):
        ctx.timer_start()
        ctx._check(lib.pk_commit_into(ctx.handle, ptrs, 2, n_vars, 1, 4, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, root, None))
        ms.append(round(ctx.timer_stop(), 1))
    return ms

polys, bufs = make()
print("(a) clean process:", commit(polys, bufs))
big = [ctx.alloc_fe((GB << 30) // 32 // 8) for _ in range(8)]
for b in big:
    ctx.zero(b, 1 << 20)
for b in big:
    b.free()
print(f"(b) same buffers, right after allocating and freeing {GB} GB:", commit(polys, bufs))
time.sleep(3)
print("(c) after a 3 s pause:", commit(polys, bufs))
for b in bufs:
    b.free()
del polys
torch.cuda.empty_cache()
polys, bufs = make()
print("(d) its own buffers re-allocated:", commit(polys, bufs))
big = [ctx.alloc_fe((GB << 30) // 32 // 8) for _ in range(8)]
print("(e) while the other buffers are held:", commit(polys, bufs))
for b in bufs:
    b.free()
del polys
torch.cuda.empty_cache()
for b in big:
    b.free()
polys, bufs = make()
print("(f) everything released, then its buffers allocated afresh:", commit(polys, bufs))
