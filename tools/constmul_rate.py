import ctypes as C, sys, json
sys.path.insert(0, "/root/repo")
import provekit_amd
from provekit_amd._lib import lib
from tools.pk_probes import lib as probes
ctx = provekit_amd.Context(0)
res = {}
for shoup in (0, 1):
    best = 0
    for w in (2, 4, 8):
        for ilp in (1, 2):
            v = C.c_double()
            ctx._check(probes.pk_probe_constmul_rate(ctx.handle, w, ilp, 1500, shoup, C.byref(v)))
            best = max(best, v.value)
            print(shoup, w, ilp, v.value / 1e12)
    res["shoup" if shoup else "mont261"] = best / 1e12
res["ratio"] = res["shoup"] / res["mont261"]
print(json.dumps(res))
