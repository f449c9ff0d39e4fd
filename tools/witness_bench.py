#!/usr/bin/env python3
"""X4 measurement: pk_witness_solve on synthetic builder lists (tests/witness_gen.py) of 2^log2 builders -- a wide list (inputs
drawn from everything solved so far: few, large levels) and a chain (every builder consumes its predecessor: one level each, the
worst case for a levelled solver) -- against the sequential restatement (oracle/witness_ref.py, pure Python: a reported
baseline only).  One JSON object per list.  usage: python tools/witness_bench.py [log2=18]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import oracle_lib as oracle  # noqa: E402
import provekit_amd  # noqa: E402
import witness_ref as R  # noqa: E402
from provekit_amd._lib import lib  # noqa: E402
from provekit_amd.witness import WitnessProgram, encode_witness_builders, inspect_witness_builders  # noqa: E402
from witness_gen import random_program  # noqa: E402

log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 18
ctx = provekit_amd.Context(0)
for name, n, chain in (("wide", 1 << log2, False), ("chain", 1 << min(log2, 14), True)):
    builders, acir, ch, nw = random_program(7, n, chain=chain)
    data = encode_witness_builders(builders)
    info = inspect_witness_builders(data)
    t0 = time.perf_counter()
    prog = WitnessProgram(ctx, data)
    t_create = time.perf_counter() - t0
    mont = lambda xs: oracle.to_mont(oracle.ints_to_limbs([int(x) for x in xs]))
    d_ac, chm = ctx.upload(mont(acir)), mont(ch)
    d_w, d_set = ctx.alloc_fe(nw), ctx.alloc(nw)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        ctx._check(lib.pk_witness_solve(ctx.handle, prog.handle, d_ac.ptr, len(acir), chm.ctypes.data, len(ch), d_w.ptr, nw, d_set.ptr))
        best = min(best, time.perf_counter() - t0)
    t0 = time.perf_counter()
    want = R.solve_witness_vec(builders, acir, ch, nw)
    t_py = time.perf_counter() - t0
    got = oracle.limbs_to_ints(oracle.from_mont(ctx.download_fe(d_w, nw)))
    ok = all(got[i] == (x or 0) for i, x in enumerate(want))
    print(json.dumps({"list": name, "builders": len(builders), "witnesses": nw, "levels": info["n_levels"], "work_items": info["n_items"],
                      "postcard_bytes": len(data), "create_ms": round(1e3 * t_create, 2), "solve_ms": round(1e3 * best, 3),
                      "builders_per_s": round(len(builders) / best), "python_restatement_ms": round(1e3 * t_py, 1), "bit_identical": ok}), flush=True)
    prog.close()
