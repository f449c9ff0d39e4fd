#!/usr/bin/env python3
"""Many proofs in latency mode against the plain prover, same seeds: any race in the gates or the side stream shows up as a different
transcript.  usage: latency_soak.py [m=17] [proofs=300]   (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import provekit_amd
from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for

m = int(sys.argv[1]) if len(sys.argv) > 1 else 17
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
m_0 = m - 1
n_wit = (1 << (m - 1)) - 5
ctx = provekit_amd.Context(0)
r1cs, mats, interner, nc, n_in = bench.synth_r1cs(ctx, m_0, n_wit, seed=77)
d_z, _ = bench.satisfying_witness(ctx, r1cs, n_wit, nc, n_in, 5)
s = WhirR1CSScheme(ctx, r1cs, m, m_0, WhirConfig.derive(m), blinding_config_for(m_0))
t0 = time.time()
bad = 0
for i in range(n):
    ctx.set_latency_mode(False)
    a = s.prove(d_z, seed=1000 + i)
    ctx.set_latency_mode(True)
    b = s.prove(d_z, seed=1000 + i)
    c = s.prove(d_z, seed=1000 + i)
    if not (a == b == c):
        bad += 1
        print("MISMATCH at seed", 1000 + i, len(a), len(b), len(c))
print(f"m={m}: {n} seeds x (plain, latency, latency): {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
