"""the ceiling of a hand-scheduled / hand-allocated Skyscraper square round, measured by ablation
(tools/probes/probes.hip sq_round_ablated): the product's round against the same round without the instructions assembly could fold
away.  Alternating, several repetitions, at the occupancies the leaf hash runs at.  One JSON object."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import provekit_amd
from tools.pk_probes import lib as probes

ctx = provekit_amd.Context(0)
names = {0: "product", 1: "no_rc_adds", 2: "no_rc_no_r_no_zero_ext", 3: "also_no_doubling"}
out = {"what": "Skyscraper square rounds per second (T/s), best of 5 alternating repetitions; variants 1-3 compute wrong values on purpose", "rates": {}}
for waves, ilp in ((4, 1), (6, 1), (8, 1), (4, 2)):
    best = {v: 0.0 for v in names}
    for rep in range(5):
        for v in names:
            r = C.c_double()
            ctx._check(probes.pk_probe_sq_round_rate(ctx.handle, v, waves, ilp, 3000, C.byref(r)))
            best[v] = max(best[v], r.value)
    key = f"waves{waves}_ilp{ilp}"
    out["rates"][key] = {names[v]: round(best[v] / 1e12, 4) for v in names}
    out["rates"][key]["ceiling_speedup_vs_product"] = {names[v]: round(best[v] / best[0], 4) for v in (1, 2, 3)}
print(json.dumps(out))
