#!/usr/bin/env python3
"""Device time of the field inverse (csrc/feinv.hpp) through pk_probe_arith_device: op 22 = constant instruction sequence,
op 23 = variable-time steps (what the witness builders run), op 0 = one Montgomery product for scale; one wavefront (latency), 8192
and 2^20 elements (throughput).  Measured: 41 / 41 / 6 us for one wavefront, 600 / 370 / 18 us for 2^20 -- the variable-time form
does fewer instructions but no shorter a dependent chain.  usage: python tools/inverse_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, provekit_amd
from provekit_amd._lib import lib
from tools.pk_probes import lib as probes
from provekit_amd.field import random_field
ctx=provekit_amd.Context(0)
for n in (64, 8192, 1<<20):
    a=ctx.upload(random_field(n,1)); o=ctx.alloc_fe(n)
    for op in (22,23,0):
        for _ in range(2): ctx._check(probes.pk_probe_arith_device(ctx.handle,op,a.ptr,a.ptr,o.ptr,n))
        ctx.sync(); ts=[]
        for _ in range(5):
            ctx.timer_start(); ctx._check(probes.pk_probe_arith_device(ctx.handle,op,a.ptr,a.ptr,o.ptr,n)); ts.append(ctx.timer_stop())
        print(n, "op", op, round(min(ts)*1e3,1), "us")
