#!/usr/bin/env python3
"""A/B of pk_ctx_set_latency_mode on the bench statement, one proof at a time, modes alternated (GPU box)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import provekit_amd
from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for

m = int(sys.argv[1]) if len(sys.argv) > 1 else 21
m_0 = m - 1
n_wit = (1 << (m - 1)) - 5
ctx = provekit_amd.Context(0)
r1cs, mats, interner, nc, n_in = bench.synth_r1cs(ctx, m_0, n_wit, seed=1234)
d_z, _ = bench.satisfying_witness(ctx, r1cs, n_wit, nc, n_in, 99)
s = WhirR1CSScheme(ctx, r1cs, m, m_0, WhirConfig.derive(m), blinding_config_for(m_0))
for i in range(5):
    s.prove_nocopy(d_z, seed=i)
res = {"plain": [], "latency": [], "latency_no_overlap": []}
for rep in range(6):
    for mode in ("plain", "latency", "latency_no_overlap"):
        ctx.set_latency_mode(mode != "plain")
        os.environ.pop("PK_NO_BLINDING_OVERLAP", None)
        if mode == "latency_no_overlap":
            os.environ["PK_NO_BLINDING_OVERLAP"] = "1"
        ts = []
        for i in range(12):
            t = time.perf_counter()
            s.prove_nocopy(d_z, seed=100 + i)
            ts.append(time.perf_counter() - t)
        res[mode].append(round(1e3 * sorted(ts)[6], 3))
os.environ.pop("PK_NO_BLINDING_OVERLAP", None)
ctx.set_latency_mode(False)
print(json.dumps({"m": m, "median_ms_per_proof": res}))
if os.environ.get("PK_PROVE_TIMING_AB"):
    os.environ["PK_PROVE_TIMING"] = "1"
    for mode in ("plain", "latency"):
        ctx.set_latency_mode(mode == "latency")
        print("----", mode, file=sys.stderr)
        s.prove_nocopy(d_z, seed=7)
