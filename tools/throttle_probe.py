#!/usr/bin/env python3
"""Does sustained load lower the clocks?  The register-resident squaring rate (pk_probe_modmul_rate) and rocm-smi's sclk / power before
and after N seconds of chip-filling work (2^26 commits).  GPU box."""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import provekit_amd
from provekit_amd._lib import lib
from tools.pk_probes import lib as probes

ctx = provekit_amd.Context(0)

def rate():
    v = C.c_double()
    ctx._check(probes.pk_probe_modmul_rate(ctx.handle, 8, 2, 3000, C.byref(v)))
    return round(v.value / 1e12, 4)

def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "Power", "junction", "mclk"))]
        return keep[:6]
    except Exception as e:
        return [str(e)]

n_vars = 26
n = 1 << n_vars
polys = [torch.randint(0, 2**62, (n, 4), dtype=torch.int64, device="cuda:0") for _ in range(2)]
for t in polys:
    t[:, 3] &= (1 << 60) - 1
torch.cuda.synchronize()
ptrs = (C.c_void_p * 2)(*[int(t.data_ptr()) for t in polys])
szs = [C.c_size_t() for _ in range(3)]
ctx._check(lib.pk_commit_sizes(ctx.handle, 2, n_vars, 1, 4, *[C.byref(x) for x in szs]))
leaves, nodes, scratch = (ctx.alloc_fe(x.value) for x in szs)
root = (C.c_uint8 * 32)()
def commit():
    ctx.timer_start()
    ctx._check(lib.pk_commit_into(ctx.handle, ptrs, 2, n_vars, 1, 4, leaves.ptr, nodes.ptr, scratch.ptr, root, None))
    return ctx.timer_stop()
commit()
print("idle: rate", rate(), smi())
t0 = time.time()
ms = []
while time.time() - t0 < 25:
    ms.append(commit())
    if len(ms) % 40 == 0:
        print(f"t={time.time()-t0:5.1f}s commit {ms[-1]:.1f} ms rate {rate()}", smi()[:3])
print("first 5", [round(x, 1) for x in ms[:5]], "last 5", [round(x, 1) for x in ms[-5:]])
print("after load: rate", rate(), smi())
time.sleep(8)
print("after 8 s idle: rate", rate(), "commit", round(commit(), 1), smi())
