#!/usr/bin/env python3
"""Does a wavefront run slower when only some of its lanes are active?  The lone-lane Skyscraper square round of
pk_probe_coop_round (100 k rounds, one wavefront, hipEvent-timed) under different lane masks (PK_COOP_ACTIVE).  Measured on an
idle MI355X: 409-437 ns per round whatever the mask -- no.  (Asked because a witness-builder level with a few Inverse lanes per
wavefront ran 2.5x longer than one with 32; grouping a level's items by variant removed that)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import provekit_amd
from provekit_amd._lib import lib
from tools.pk_probes import lib as probes

ctx = provekit_amd.Context(0)
L, R = (C.c_uint32 * 9)(*range(1, 10)), (C.c_uint32 * 9)(*range(11, 20))
out, cyc = (C.c_uint32 * 36)(), (C.c_uint64 * 4)()
for mask in ["ffffffffffffffff", "00000000ffffffff", "000000000000ffff", "f", "1", "1000100010001", "8000000000000000", "1111111111111111"]:
    os.environ["PK_COOP_ACTIVE"] = mask
    best = min(cyc[3] for _ in range(3) if not ctx._check(probes.pk_probe_coop_round(ctx.handle, L, R, 100000, out, cyc)))
    print(mask, "ns/round", best / 100000)
