"""Soak: 240 proofs over six scheme shapes (incl. pow_bits 0 and fractional difficulties), every one checked by the independent
verifier, every tenth with the R1CS matrix evaluation.  Development aid (about a minute on one MI355X): python tools/soak.py"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(R, sub))
import numpy as np, torch
torch.cuda.is_available()
import oracle_lib as oracle, provekit_amd, verifier as V
from test_gpu_prove import satisfiable_r1cs, to_sparse
from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
from provekit_amd.sparse_matrix import R1CS
ctx = provekit_amd.Context(0)
def vcfg(c):
    return V.WhirConfig(c.n_vars, c.batch_size, c.folding_factor, c.starting_log_inv_rate, c.num_queries, c.ood_samples, c.pow_bits, c.final_queries, c.final_pow_bits, c.commitment_ood_samples, c.final_folding_pow_bits)
t0 = time.time(); n_ok = 0
for case, (m, m_0, nc, n_in, pb) in enumerate([(9, 7, 100, 60, 5.0), (10, 8, 200, 100, 7.3), (12, 9, 500, 700, 3.0), (13, 11, 2000, 1000, 9.9), (8, 6, 60, 40, 0.0), (11, 10, 1000, 20, 12.0)]):
    nw, z, coeffs, trips = satisfiable_r1cs(nc, n_in, 100 + case)
    r1cs = R1CS(ctx, *(to_sparse(nc, nw, t) for t in trips), oracle.to_mont(oracle.ints_to_limbs(coeffs)))
    cfg_w, cfg_b = WhirConfig.for_size(m, pb), blinding_config_for(m_0, pb)
    scheme = WhirR1CSScheme(ctx, r1cs, m, m_0, cfg_w, cfg_b)
    d_z = ctx.upload(oracle.to_mont(oracle.ints_to_limbs(z)))
    mats = [(t[0], t[1], [coeffs[v] for v in t[2]]) for t in trips]
    for seed in range(40):
        ctx.set_latency_mode(seed % 2 == 1)  # every other proof in latency mode (gated rounds, side stream): the same verifier must accept it
        proof = scheme.prove(d_z, seed=seed * 7919 + case)
        if seed % 8 == 1:
            ctx.set_latency_mode(False)
            assert scheme.prove(d_z, seed=seed * 7919 + case) == proof, (case, seed, "latency mode wrote another transcript")
        assert V.verify(proof, scheme.domain_separator, m, m_0, vcfg(cfg_w), vcfg(cfg_b), r1cs=(nc, nw, mats) if seed % 10 == 0 else None), (case, seed)
        n_ok += 1
    ctx.set_latency_mode(False)
    scheme.close(); r1cs.close()
    print("case", case, "ok", n_ok, round(time.time() - t0, 1), "s", flush=True)
print("soak passed:", n_ok, "proofs")
