#!/usr/bin/env python3
"""The wavefront-cooperative Skyscraper square round (csrc/selftest.hip coop_sq_round, a prototype) against the lone lane's round:
values must agree mod p, and the cycle counts of the two loops say what a cooperative compression would buy.  One JSON object.
usage: python tools/coop_round.py [iters=2000]"""
import ctypes as C
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import provekit_amd
from provekit_amd._lib import lib
from tools.pk_probes import lib as probes

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000


def limbs(v):
    return [(v >> (29 * k)) & ((1 << 29) - 1) if k < 8 else v >> 232 for k in range(9)]


def value(ls):
    return sum(int(x) << (29 * k) for k, x in enumerate(ls))


ctx = provekit_amd.Context(0)
rnd = random.Random(5)
res = {"iters": iters, "cases": 0}
best = None
for case in range(8):
    l, r = rnd.randrange(P), rnd.randrange(P)
    L, R = (C.c_uint32 * 9)(*limbs(l)), (C.c_uint32 * 9)(*limbs(r))
    out, cyc = (C.c_uint32 * 36)(), (C.c_uint64 * 4)()
    # a compression runs at most 6 square rounds between two reductions (the state grows by ~2p per round and the limbs hold it);
    # parity is checked over up to 17 rounds, the timing loop runs longer and its (overflowed) values are not compared
    for n in (1, 2, 3, 6, 12, 17):
        ctx._check(probes.pk_probe_coop_round(ctx.handle, L, R, n, out, cyc))
        o = list(out)
        cl, cr, sl, sr = value(o[0:9]), value(o[9:18]), value(o[18:27]), value(o[27:36])
        assert cl % P == sl % P and cr % P == sr % P, (case, n)
        assert max(o[0:8]) <= (1 << 29) and o[8] < (1 << 30), o[:9]
    ctx._check(probes.pk_probe_coop_round(ctx.handle, L, R, iters, out, cyc))
    res["cases"] += 1
    if best is None or cyc[2] < best[2]:
        best = (cyc[0], cyc[1], cyc[2], cyc[3])
res["coop_cycles_per_round"] = round(best[0] / iters, 1)
res["lone_lane_cycles_per_round"] = round(best[1] / iters, 1)
res["coop_ns_per_round"] = round(best[2] / iters, 1)
res["lone_lane_ns_per_round"] = round(best[3] / iters, 1)
res["square_round_speedup"] = round(best[3] / best[2], 3)
# a compression = 14 square rounds + 4 bar rounds; the bars stay on one lane (exact division, byte S-box, exact reduction) and pay
# two transpositions (9 readlanes in, 9 writes out); bar round / square round on the lone lane = 312 / 243 instructions
bar = res["lone_lane_ns_per_round"] * 312 / 243
res["compression_speedup_estimate"] = round((14 * best[3] / iters + 4 * bar) / (14 * best[2] / iters + 4 * (bar + 50)), 3)
print(json.dumps(res))
