#!/usr/bin/env python3
"""X4 development aid: wall time of pk_witness_solve for a two-level list -- 64 ACIR inputs, then N builders of ONE kind reading
them -- per kind, so the per-level latency of each WitnessBuilder variant can be read off.  usage: python tools/witness_ops.py [N=8192]"""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import oracle_lib as oracle  # noqa: E402
import provekit_amd  # noqa: E402
from provekit_amd._lib import lib  # noqa: E402
from provekit_amd.witness import WitnessBuilder as WB, WitnessProgram  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
P = oracle.P
rnd = random.Random(3)
fe = lambda: rnd.randrange(1, P)
ctx = provekit_amd.Context(0)
n_in = 64
acir = [fe() for _ in range(n_in)]
pick = lambda: rnd.randrange(n_in)
kinds = {
    "none": lambda i: None,
    "product": lambda i: WB.Product(i, pick(), pick()),
    "sum5": lambda i: WB.Sum(i, [(fe(), pick()) for _ in range(5)]),
    "inverse": lambda i: WB.Inverse(i, pick()),
    "prod_linear": lambda i: WB.ProductLinearOperation(i, pick(), fe(), fe(), pick(), fe(), fe()),
    "spice_factor": lambda i: WB.SpiceMultisetFactor(i, pick(), pick(), fe(), pick(), pick(), fe(), pick()),
    "binop_denom": lambda i: WB.BinOpLookupDenominator(i, pick(), pick(), pick(), ("w", pick()), ("c", fe()), ("w", pick())),
    "logup": lambda i: WB.LogUpDenominator(i, pick(), fe(), pick()),
    "const": lambda i: WB.Constant(i, fe()),
}
LONG = 1 << 17  # one Sum of 2^17 terms (a LogUp grand sum): a single builder, not N of them
mixes = {}
for k in (1, 8, 64, 512, 4096):
    def mk(i, k=k, chosen={}):
        if k not in chosen:
            chosen[k] = set(rnd.sample(range(n_in, n_in + N), k))
        return WB.Inverse(i, pick()) if i in chosen[k] else rnd.choice([kinds["product"], kinds["sum5"], kinds["logup"], kinds["prod_linear"]])(i)
    mixes[f"mixed, {k} inverses"] = mk
kinds.update(mixes)
kinds["one sum of 2^17 terms"] = None
for name, mk in kinds.items():
    b = [WB.Acir(i, i) for i in range(n_in)]
    if mk is None:
        b.append(WB.Sum(n_in, [(fe(), pick()) for _ in range(LONG)]))
        b += [WB.Constant(n_in + 1 + i, 1) for i in range(N - 1)]
    elif name != "none":
        b += [mk(n_in + i) for i in range(N)]
    nw = n_in + N
    prog = WitnessProgram(ctx, b)
    d_ac = ctx.upload(oracle.to_mont(oracle.ints_to_limbs(acir)))
    d_w, d_set = ctx.alloc_fe(nw), ctx.alloc(nw)
    best = 1e9
    for _ in range(7):
        t0 = time.perf_counter()
        ctx._check(lib.pk_witness_solve(ctx.handle, prog.handle, d_ac.ptr, n_in, None, 0, d_w.ptr, nw, d_set.ptr))
        best = min(best, time.perf_counter() - t0)
    print(json.dumps({"kind": name, "items": 0 if name == "none" else N, "solve_us": round(1e6 * best, 1)}), flush=True)
    prog.close()
