#!/usr/bin/env python3
"""Everything after ACVM on the device (pk_noir_prove: witness transcript -> witness builders -> fill_witness -> prove) next to
pk_prove on a witness that is already there, at the bench size (m = 21) under the reference's own WHIR schedule.  The statement is
built together with its builder list (tests/test_gpu_witness._noir_instance at scale: every constraint is one builder's defining
equation -- products, inverses, sums -- over inputs, two transcript challenges and earlier outputs), so the proof is of a
satisfiable instance and the verifier accepts it.  One JSON object.
usage: python tools/noir_bench.py [n_builders=780000] [concurrency=16] [proofs_per_prover=6] [m=21]"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np  # noqa: E402
import oracle_lib as oracle  # noqa: E402
import provekit_amd  # noqa: E402
from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for  # noqa: E402
from provekit_amd.sparse_matrix import R1CS  # noqa: E402
from provekit_amd.witness import WitnessProgram, encode_witness_builders, inspect_witness_builders  # noqa: E402
from test_gpu_prove import to_sparse  # noqa: E402
from test_gpu_witness import _mont, _noir_instance  # noqa: E402

n_builders = int(sys.argv[1]) if len(sys.argv) > 1 else 780000
conc = int(sys.argv[2]) if len(sys.argv) > 2 else 16
per = int(sys.argv[3]) if len(sys.argv) > 3 else 6
m = int(sys.argv[4]) if len(sys.argv) > 4 else 21
m_0 = m - 1
t0 = time.perf_counter()
builders, acir, pub_idx, nw, coeffs, trips = _noir_instance(oracle, 5, n_in=1000, n_prod=n_builders)
nc = trips[0][0][-1] + 1
assert nw <= 1 << (m - 1) and nc <= 1 << m_0
data = encode_witness_builders(builders)
info = inspect_witness_builders(data)
t_build = time.perf_counter() - t0
interner = oracle.to_mont(oracle.ints_to_limbs(coeffs))
acir_m = _mont(oracle, acir)
cfg_w, cfg_b = WhirConfig.derive(m), blinding_config_for(m_0)
mats = [to_sparse(nc, nw, t) for t in trips]

workers = []
for w in range(conc):
    c = provekit_amd.Context(0)
    r1cs = R1CS(c, *mats, interner)
    s = WhirR1CSScheme(c, r1cs, m, m_0, cfg_w, cfg_b)
    prog = WitnessProgram(c, data)
    d_acir = c.upload(acir_m)
    workers.append((c, r1cs, s, prog, d_acir))

# the witness pk_noir_prove computes, once, for the pk_prove leg and the satisfaction check
c, r1cs, s, prog, d_acir = workers[0]
proof = s.noir_prove(prog, d_acir, len(acir), pub_idx, seed=1)
from provekit_amd._lib import lib  # noqa: E402
from provekit_amd.witness import fill_witness, witness_challenges  # noqa: E402

pub = [acir[i] for i in pub_idx]
ch = witness_challenges(nc, nw, _mont(oracle, pub), info["n_challenges"])
d_ws = []
for (c, r1cs, s, prog, d_acir) in workers:
    d_w, d_set = c.alloc_fe(nw), c.alloc(nw)
    c._check(lib.pk_witness_solve(c.handle, prog.handle, d_acir.ptr, len(acir), ch.ctypes.data, len(ch), d_w.ptr, nw, d_set.ptr))
    fill_witness(c, d_w, d_set, nw, seed=1)
    d_ws.append(d_w)
assert workers[0][1].test_witness_satisfaction(d_ws[0]) is None
assert workers[0][2].prove(d_ws[0], seed=1) == proof, "pk_noir_prove != its four steps"


def run(kind):
    def work(w):
        c, r1cs, s, prog, d_acir = workers[w]
        for i in range(per):
            if kind == "noir":
                s.noir_prove_nocopy(prog, d_acir, len(acir), pub_idx, seed=100 + i)
            else:
                s.prove_nocopy(d_ws[w], seed=100 + i)

    ths = [threading.Thread(target=work, args=(w,)) for w in range(conc)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return conc * per / (time.perf_counter() - t0)


def single(kind, reps=5):
    c, r1cs, s, prog, d_acir = workers[0]
    best = 1e9
    for i in range(reps):
        t0 = time.perf_counter()
        if kind == "noir":
            s.noir_prove_nocopy(prog, d_acir, len(acir), pub_idx, seed=200 + i)
        else:
            s.prove_nocopy(d_ws[0], seed=200 + i)
        best = min(best, time.perf_counter() - t0)
    return 1e3 * best


run("prove"), run("noir")  # warm-up
res = {"m": m, "builders": len(builders), "witnesses": nw, "constraints": nc, "levels": info["n_levels"], "postcard_bytes": len(data),
       "host_build_s": round(t_build, 1), "provers": conc,
       "prove_proofs_per_s": round(run("prove"), 1), "noir_prove_proofs_per_s": round(run("noir"), 1),
       "prove_single_ms": round(single("prove"), 2), "noir_prove_single_ms": round(single("noir"), 2)}
# every prover thread must produce the same bytes for the same seed (shared R1CS-independent state: twiddle tables, nothing else)
agree = [None] * conc


def check(w):
    c, r1cs, s, prog, d_acir = workers[w]
    agree[w] = [s.noir_prove(prog, d_acir, len(acir), pub_idx, seed=300 + i) for i in range(3)]


ths = [threading.Thread(target=check, args=(w,)) for w in range(conc)]
for t in ths:
    t.start()
for t in ths:
    t.join()
res["threads_agree"] = all(a == agree[0] for a in agree)
import verifier as V  # noqa: E402


def vcfg(c):
    return V.WhirConfig(c.n_vars, c.batch_size, c.folding_factor, c.starting_log_inv_rate, c.num_queries, c.ood_samples, c.pow_bits,
                        c.final_queries, c.final_pow_bits, c.commitment_ood_samples, c.final_folding_pow_bits)


res["verifier_accepts"] = bool(V.verify(proof, workers[0][2].domain_separator, m, m_0, vcfg(cfg_w), vcfg(cfg_b)))
print(json.dumps(res), flush=True)
