#!/usr/bin/env python3
"""Rate of register-resident Montgomery squarings on this GPU: the 9 x 29-bit integer multiplier of the product path
(csrc/fe29.hpp) next to the f64-FMA 5 x 52-bit prototype (csrc/fe52.hpp), same probe, same box.  Output: one JSON object.
usage: python tools/modmul_rates.py > gpurun_out/modmul_rates.json"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import provekit_amd
from provekit_amd._lib import lib
from tools.pk_probes import lib as probes

ctx = provekit_amd.Context(0)
res = {"unit": "T squarings/s", "iters": 4096, "int29": {}, "fp52": {}}
for name, fn in (("int29", probes.pk_probe_modmul_rate), ("fp52", probes.pk_probe_modmul_rate_fp52)):
    for waves in (1, 2, 4, 6, 8):
        for ilp in (1, 2, 4):
            best = 0.0
            for _ in range(3):
                r = C.c_double(0)
                ctx._check(fn(ctx.handle, waves, ilp, 4096, C.byref(r)))
                best = max(best, r.value)
            res[name][f"waves{waves}_ilp{ilp}"] = round(best * 1e-12, 4)
res["best_int29"] = max(res["int29"].values())
res["best_fp52"] = max(res["fp52"].values())
res["fp52_over_int29"] = round(res["best_fp52"] / res["best_int29"], 3)
print(json.dumps(res, indent=1))
