"""Stress (development aid, ~2 minutes on one MI355X): 16 prover threads on one GPU, bench-size statement (m = 21), a
satisfiable instance; every thread proves the same seeds, all transcripts of a seed must be identical across threads
(no cross-context interference under full load) and a sample is checked by the independent verifier."""
import os
import sys
import threading
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(R, sub))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np
import torch

torch.cuda.is_available()
import oracle_lib as oracle
import provekit_amd
import verifier as V
from provekit_amd.field import random_field
from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
from provekit_amd.sparse_matrix import R1CS, SparseMatrix

m, m_0 = 21, 20
nc, n_in = 1 << 19, (1 << 19) - 8
nw = 1 + n_in + nc
rng = np.random.default_rng(21)
coeffs = [1, 2, 3, 5, oracle.P - 1, 7, oracle.P - 2, 11]
interner = oracle.to_mont(oracle.ints_to_limbs(coeffs))
z = np.concatenate([oracle.to_mont(oracle.ints_to_limbs([1])), random_field(n_in, 5), np.zeros((nc, 4), dtype=np.uint64)])
mats = []
for _ in range(2):
    cols = np.sort(rng.integers(0, 1 + n_in - 2, size=(nc, 3), dtype=np.int64), axis=1) + np.arange(3)
    mats.append((np.arange(nc, dtype=np.uint32) * 3, cols.reshape(-1).astype(np.uint32), rng.integers(0, len(coeffs), size=3 * nc).astype(np.uint32)))
az, bz = (oracle.spmv(nc, nw, nri, ci, v, interner, z) for nri, ci, v in mats)
z[1 + n_in :] = oracle.hadamard(az, bz)
mats.append((np.arange(nc, dtype=np.uint32), (1 + n_in + np.arange(nc)).astype(np.uint32), np.zeros(nc, dtype=np.uint32)))
cfg_w, cfg_b = WhirConfig.poseidon_witness(), blinding_config_for(m_0)
T, SEEDS = 16, list(range(100, 100 + int(os.environ.get("PK_STRESS_SEEDS", "12"))))
workers = []
for _ in range(T):
    c = provekit_amd.Context(0)
    r = R1CS(c, *(SparseMatrix(nc, nw, *t) for t in mats), interner)
    # PK_STRESS_LATENCY=1: every odd prover in latency mode (gated kernels + side streams of 8 provers among 8 plain ones): not a configuration
    # to run for speed -- a check that the gates cannot deadlock or corrupt anything when the chip is shared
    if os.environ.get("PK_STRESS_LATENCY") == "1" and len(workers) % 2 == 1:
        c.set_latency_mode(True)
    workers.append((c, r, WhirR1CSScheme(c, r, m, m_0, cfg_w, cfg_b), c.upload(z)))
out = [None] * T


def run(i):
    _, _, s, d = workers[i]
    order = SEEDS[i % len(SEEDS):] + SEEDS[: i % len(SEEDS)]  # different phase per thread: different kernels overlap
    res = {}
    for sd in order:
        res[sd] = s.prove(d, seed=sd)
    out[i] = res


t0 = time.time()
ths = [threading.Thread(target=run, args=(i,)) for i in range(T)]
for t in ths:
    t.start()
for t in ths:
    t.join()
dt = time.time() - t0
print(f"{T * len(SEEDS)} proofs in {dt:.1f} s ({T * len(SEEDS) / dt:.1f} proofs/s incl. Python copies)")
for sd in SEEDS:
    ref = out[0][sd]
    assert all(out[i][sd] == ref for i in range(T)), f"seed {sd}: transcripts differ between prover threads"
print("all", T, "threads agree on all", len(SEEDS), "seeds")


def vcfg(c):
    return V.WhirConfig(c.n_vars, c.batch_size, c.folding_factor, c.starting_log_inv_rate, c.num_queries, c.ood_samples, c.pow_bits, c.final_queries, c.final_pow_bits, c.commitment_ood_samples, c.final_folding_pow_bits)


ds = workers[0][2].domain_separator
for sd in SEEDS[:4]:
    assert V.verify(out[3][sd], ds, m, m_0, vcfg(cfg_w), vcfg(cfg_b))
print("sampled proofs accepted by the independent verifier; stress passed")
