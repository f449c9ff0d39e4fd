"""GPU end-to-end: pk_prove (C++ driver + HIP kernels + Skyscraper-sponge transcript) on satisfiable synthetic R1CS
instances; the proof must be accepted by the independent pure-Python verifier (oracle/verifier.py, which restates
provekit/verifier/src/whir_r1cs.rs and the Go WHIR verifier equations) and rejected after tampering."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def satisfiable_r1cs(nc, n_in, seed):
    """rows i: (sum a z)(sum b z) = z[out_i]; out_i is a fresh witness so any inputs extend to a satisfying z"""
    import pyref as pr

    rng = np.random.default_rng(seed)
    nw = 1 + n_in + nc
    coeffs = [1, 2, 3, 5, pr.P - 1, 7, pr.P - 2, 11]
    z = [1] + [int(rng.integers(0, 2**62)) * 1234567891011 % pr.P for _ in range(n_in)] + [0] * nc
    A, B, Cm = ([], [], []), ([], [], []), ([], [], [])
    for i in range(nc):
        lim = 1 + n_in + i
        sa = sb = 0
        for M, which in ((A, 0), (B, 1)):
            cols = sorted(set(int(c) for c in rng.integers(0, lim, size=int(rng.integers(1, 4)))))
            for c in cols:
                v = int(rng.integers(0, len(coeffs)))
                M[0].append(i); M[1].append(c); M[2].append(v)
                if which == 0:
                    sa += coeffs[v] * z[c]
                else:
                    sb += coeffs[v] * z[c]
        z[lim] = sa * sb % pr.P
        Cm[0].append(i); Cm[1].append(lim); Cm[2].append(0)
    return nw, z, coeffs, (A, B, Cm)


def to_sparse(nc, nw, trip):
    from provekit_amd.sparse_matrix import SparseMatrix

    rows, cols, vals = (np.array(x, dtype=np.int64) for x in trip)
    nri = np.searchsorted(rows, np.arange(nc)).astype(np.uint32)
    return SparseMatrix(nc, nw, nri, cols.astype(np.uint32), vals.astype(np.uint32))


def run_case(ctx, oracle, m, m_0, nc, n_in, seed, pow_bits):
    import pyref as pr
    import verifier as V
    from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
    from provekit_amd.sparse_matrix import R1CS

    nw, z, coeffs, trips = satisfiable_r1cs(nc, n_in, seed)
    assert nw <= 1 << (m - 1) and nc <= 1 << m_0
    interner = oracle.to_mont(oracle.ints_to_limbs(coeffs))
    r1cs = R1CS(ctx, *(to_sparse(nc, nw, t) for t in trips), interner)
    cfg_w = WhirConfig.for_size(m, pow_bits)
    cfg_w.num_queries = [20, 12, 9, 8][: cfg_w.n_rounds]
    cfg_b = blinding_config_for(m_0, pow_bits)
    scheme = WhirR1CSScheme(ctx, r1cs, m, m_0, cfg_w, cfg_b)
    d_z = ctx.upload(oracle.to_mont(oracle.ints_to_limbs(z)))
    proof = scheme.prove(d_z, seed=seed)
    proof2 = scheme.prove(d_z, seed=seed)
    assert proof == proof2, "same witness + same RNG seed must give the same transcript"
    assert scheme.prove(d_z, seed=seed + 1) != proof

    def vcfg(c):
        return V.WhirConfig(c.n_vars, c.batch_size, c.folding_factor, c.starting_log_inv_rate, c.num_queries, c.ood_samples, c.pow_bits,
                            c.final_queries, c.final_pow_bits, c.commitment_ood_samples, c.final_folding_pow_bits)

    mats = [(t[0], t[1], [coeffs[v] for v in t[2]]) for t in trips]
    args = (scheme.domain_separator, m, m_0, vcfg(cfg_w), vcfg(cfg_b))
    assert V.verify(proof, *args, r1cs=(nc, nw, mats))
    # tampering anywhere must be rejected
    rng = np.random.default_rng(seed)
    for pos in [0, 40, len(proof) // 3, len(proof) // 2, len(proof) - 40] + [int(x) for x in rng.integers(0, len(proof), size=4)]:
        bad = bytearray(proof)
        bad[pos] ^= 1
        with pytest.raises((V.VerifyError, Exception)):
            V.verify(bytes(bad), *args, r1cs=(nc, nw, mats))
    # a witness that does not satisfy the R1CS must not verify
    z_bad = list(z)
    z_bad[-1] = (z_bad[-1] + 1) % pr.P
    bad_proof = scheme.prove(ctx.upload(oracle.to_mont(oracle.ints_to_limbs(z_bad))), seed=seed)
    with pytest.raises(V.VerifyError):
        V.verify(bad_proof, *args, r1cs=(nc, nw, mats))
    scheme.close()
    r1cs.close()
    return len(proof)


def test_prove_verify_small(ctx, oracle):
    # m = 9 (one WHIR round + final on 1 variable), m_0 = 7 -> blinding WHIR on 6 variables (no main round)
    n = run_case(ctx, oracle, m=9, m_0=7, nc=100, n_in=60, seed=3, pow_bits=6.0)
    assert n > 1000


def test_prove_verify_two_rounds(ctx, oracle):
    # m = 12 -> 2 WHIR rounds, final on 0 variables; m_0 = 9
    run_case(ctx, oracle, m=12, m_0=9, nc=500, n_in=700, seed=5, pow_bits=4.0)


def test_prove_rejects_wrong_witness_length(ctx, oracle):
    from provekit_amd import ProveKitHipError
    from provekit_amd._lib import lib
    from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
    from provekit_amd.sparse_matrix import R1CS
    import ctypes as C

    nw, z, coeffs, trips = satisfiable_r1cs(20, 10, 1)
    r1cs = R1CS(ctx, *(to_sparse(20, nw, t) for t in trips), oracle.to_mont(oracle.ints_to_limbs(coeffs)))
    scheme = WhirR1CSScheme(ctx, r1cs, 9, 5, WhirConfig.for_size(9, 0.0), blinding_config_for(5, 0.0))
    d = ctx.upload(oracle.to_mont(oracle.ints_to_limbs(z)))
    n = C.c_size_t()
    with pytest.raises(ProveKitHipError):  # "Unexpected witness length for R1CS instance" (whir_r1cs.rs:43-46)
        ctx._check(lib.pk_prove(ctx.handle, scheme.handle, d.ptr, nw - 1, None, scheme._buf, len(scheme._buf), C.byref(n)))
    with pytest.raises(ProveKitHipError):  # scheme capacity (whir_r1cs.rs:47-54)
        WhirR1CSScheme(ctx, r1cs, 5, 5, WhirConfig.for_size(5, 0.0), blinding_config_for(5, 0.0))


def test_witness_satisfaction(ctx, oracle):
    """pk_r1cs_test_witness_satisfaction == R1CSSolver::test_witness_satisfaction (provekit/prover/src/r1cs.rs:41-60):
    Ok for a satisfying witness; "Constraint {row} failed" with the FIRST failing row otherwise; length check."""
    import pyref as pr
    from provekit_amd import ProveKitHipError
    from provekit_amd.sparse_matrix import R1CS

    nc = 3000
    nw, z, coeffs, trips = satisfiable_r1cs(nc, 500, 11)
    r1cs = R1CS(ctx, *(to_sparse(nc, nw, t) for t in trips), oracle.to_mont(oracle.ints_to_limbs(coeffs)))
    up = lambda v: ctx.upload(oracle.to_mont(oracle.ints_to_limbs(v)))
    r1cs.test_witness_satisfaction(up(z))
    # break two outputs: rows 1700 and 2900 (output witness of row i is z[1 + n_in + i]); later rows that READ those
    # outputs may also fail, so the first failing row is exactly 1700 only if no earlier row reads z[1+500+1700] -- true,
    # rows only read earlier outputs
    zb = list(z)
    zb[1 + 500 + 2900] = (zb[1 + 500 + 2900] + 1) % pr.P
    zb[1 + 500 + 1700] = (zb[1 + 500 + 1700] + 5) % pr.P
    with pytest.raises(ProveKitHipError, match="Constraint 1700 failed") as ei:
        r1cs.test_witness_satisfaction(up(zb))
    assert ei.value.row == 1700 and ei.value.code == -6
    # oracle agrees row by row: first row with (Az)(Bz) != Cz
    mats = [(np.array(t[0]), np.array(t[1]), [coeffs[v] for v in t[2]]) for t in trips]
    def mv(M):
        out = [0] * nc
        for r, c, v in zip(*M):
            out[int(r)] = (out[int(r)] + v * zb[int(c)]) % pr.P
        return out
    a, b, c = mv(mats[0]), mv(mats[1]), mv(mats[2])
    assert next(i for i in range(nc) if a[i] * b[i] % pr.P != c[i]) == 1700
    with pytest.raises(ProveKitHipError, match="Witness size does not match"):
        r1cs.test_witness_satisfaction(up(z), n_witness=nw - 1)
    r1cs.close()


def test_prove_verify_midsize(ctx, oracle):
    """m = 17, m_0 = 16: three WHIR rounds on trees of 2^14..2^12 leaves -- the register-NTT, fused-Merkle-level and
    pinned-mailbox paths of the bench-size prover, under the independent verifier (incl. the R1CS matrix check)."""
    run_case(ctx, oracle, m=17, m_0=16, nc=60000, n_in=5000, seed=17, pow_bits=10.0)


def size_class_instance(oracle, m):
    """satisfiable synthetic R1CS of the size class (m, m_0 = m - 1), built vectorised: 2^(m-2) constraints with 3 entries per
    row in A and B over the inputs, C selecting a fresh output per row -> (nc, nw, [(nri, cols, vals)]*3, interner, z)"""
    from provekit_amd.field import random_field

    nc, n_in = 1 << (m - 2), (1 << (m - 2)) - 8
    nw = 1 + n_in + nc
    rng = np.random.default_rng(m)
    coeffs = [1, 2, 3, 5, oracle.P - 1, 7, oracle.P - 2, 11]
    interner = oracle.to_mont(oracle.ints_to_limbs(coeffs))
    one = oracle.to_mont(oracle.ints_to_limbs([1]))
    z = np.concatenate([one, random_field(n_in, 5), np.zeros((nc, 4), dtype=np.uint64)])
    mats = []
    for _ in range(2):  # A, B read only the inputs, so the outputs c_i = (A z)_i (B z)_i come from two products
        cols = np.sort(rng.integers(0, 1 + n_in - 2, size=(nc, 3), dtype=np.int64), axis=1) + np.arange(3)
        mats.append((np.arange(nc, dtype=np.uint32) * 3, cols.reshape(-1).astype(np.uint32), rng.integers(0, len(coeffs), size=3 * nc).astype(np.uint32)))
    az, bz = (oracle.spmv(nc, nw, nri, ci, v, interner, z) for nri, ci, v in mats)
    z[1 + n_in :] = oracle.hadamard(az, bz)
    mats.append((np.arange(nc, dtype=np.uint32), (1 + n_in + np.arange(nc)).astype(np.uint32), np.zeros(nc, dtype=np.uint32)))
    return nc, nw, mats, interner, z


def prove_verify_size_class(ctx, oracle, m, check_layout=False):
    """A SATISFIABLE instance of the size class (m, m_0 = m - 1) under the reference's own WHIR schedule
    (WhirConfig.derive: queries, OOD samples and grinding difficulties of new_whir_config_for_size): the proof must pass every
    check of the independent verifier (transcript, Merkle openings of the 2^(m-3)-leaf tree, both sumchecks, folds, PoW) and
    the matrix evaluation of the deferred weights -- eq(alpha)^T {A, B, C} eq(point), recursive-verifier's matrix_evaluation.go,
    the relation provekit/verifier/src/whir_r1cs.rs:78-86 leaves to it -- evaluated with the C oracle's SpMV."""
    import verifier as V
    from provekit_amd.field import random_field
    from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
    from provekit_amd.sparse_matrix import R1CS, SparseMatrix

    m_0 = m - 1
    nc, nw, mats, interner, z = size_class_instance(oracle, m)
    r1cs = R1CS(ctx, *(SparseMatrix(nc, nw, *t) for t in mats), interner)
    d_z = ctx.upload(z)
    r1cs.test_witness_satisfaction(d_z)
    cfg_w, cfg_b = WhirConfig.derive(m), blinding_config_for(m_0)
    scheme = WhirR1CSScheme(ctx, r1cs, m, m_0, cfg_w, cfg_b)
    proof = scheme.prove(d_z, seed=1)
    assert scheme.prove(d_z, seed=1) == proof
    # latency mode (gated sumcheck rounds, blinding commitment and external rows on a side stream) writes the same bytes at this size too
    ctx.set_latency_mode(True)
    try:
        assert scheme.prove(d_z, seed=1) == proof
    finally:
        ctx.set_latency_mode(False)

    def vcfg(c):
        return V.WhirConfig(c.n_vars, c.batch_size, c.folding_factor, c.starting_log_inv_rate, c.num_queries, c.ood_samples, c.pow_bits,
                            c.final_queries, c.final_pow_bits, c.commitment_ood_samples, c.final_folding_pow_bits)

    args = (scheme.domain_separator, m, m_0, vcfg(cfg_w), vcfg(cfg_b))
    matrices = oracle.matrix_evaluator(nc, nw, mats, interner)
    assert V.verify(proof, *args, r1cs=matrices)
    fresh = scheme.prove(d_z)  # production randomness (OS CSPRNG): another transcript, equally valid
    assert fresh != proof and V.verify(fresh, *args, r1cs=matrices)
    # the matrix check bites: the same proof against a statement with one coefficient changed is refused
    other = [tuple(np.copy(a) for a in t) for t in mats]
    other[1][2][len(other[1][2]) // 2] ^= 1
    with pytest.raises(V.VerifyError, match="does not match the R1CS matrix"):
        V.verify(proof, *args, r1cs=oracle.matrix_evaluator(nc, nw, other, interner))
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 1
    with pytest.raises((V.VerifyError, Exception)):
        V.verify(bytes(bad), *args)
    # an unsatisfying witness of the right length must not verify
    z_bad = z.copy()
    z_bad[nw - 1, 0] ^= np.uint64(1)
    d_bad = ctx.upload(z_bad)
    with pytest.raises((V.VerifyError, Exception)):
        assert V.verify(scheme.prove(d_bad, seed=1), *args)
    scheme.close()
    r1cs.close()
    return proof


def test_prove_verify_bench_size(ctx, oracle):
    """BASELINE configs[1]: the bench's own statement size and WHIR schedule (m = 21, m_0 = 20, queries 109/28/16/11, final 9,
    pow_bits 19/16/16/18, final 11)."""
    proof = prove_verify_size_class(ctx, oracle, 21)
    # same wire layout as the reference's proof of this size class (decoded from poseidon-1000.np):
    # 3304 bytes of scalars -- root, 2 OOD answers, blinding root, 2 OOD answers, sum G, 20 x 4 sumcheck coefficients,
    # 2 polynomial sums, 4 x 3 sumcheck evaluations, round root, OOD answer, 8-byte nonce -- then the first hint,
    # `stir_answers`: u32 length | u64 count | count x (u64 width = 32 | 32 canonical field elements)
    import struct

    ln = struct.unpack_from("<I", proof, 3304)[0]
    k = struct.unpack_from("<Q", proof, 3308)[0]
    assert 1 <= k <= 32 and ln == 8 + k * (8 + 32 * 32)
    assert all(struct.unpack_from("<Q", proof, 3316 + q * (8 + 1024))[0] == 32 for q in range(k))
    assert 260_000 < len(proof) < 277_000  # the reference's proof of this shape is 268,756 bytes (query de-duplication varies)


def test_prove_verify_sha256_size_class(ctx, oracle):
    """BASELINE configs[2] (noir-native-sha256 is a size class): m = 23, m_0 = 22, 2^20-leaf initial tree, 3-pass NTTs"""
    prove_verify_size_class(ctx, oracle, 23)


def test_prove_verify_p256_size_class(ctx, oracle):
    """BASELINE configs[3]'s size class on one GPU: m = 25, m_0 = 24 (2^22-leaf initial tree, 2^26-point codewords)"""
    prove_verify_size_class(ctx, oracle, 25)


def test_concurrent_provers_are_deterministic(oracle):
    """Four prover threads, each with its own context / stream / arena (the bench's throughput mode), proving the same
    statement with the same seeds at the same time: every transcript must equal the one a lone prover produced -- no state
    is shared between contexts."""
    import threading

    import provekit_amd
    from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
    from provekit_amd.sparse_matrix import R1CS

    m, m_0, nc, n_in = 12, 9, 500, 700
    nw, z, coeffs, trips = satisfiable_r1cs(nc, n_in, 5)
    zm = oracle.to_mont(oracle.ints_to_limbs(z))
    interner = oracle.to_mont(oracle.ints_to_limbs(coeffs))

    def make():
        c = provekit_amd.Context(0)
        r = R1CS(c, *(to_sparse(nc, nw, t) for t in trips), interner)
        s = WhirR1CSScheme(c, r, m, m_0, WhirConfig.for_size(m, 6.0), blinding_config_for(m_0, 6.0))
        return c, r, s, c.upload(zm)

    seeds = [11, 12, 13, 14, 15]
    c0, r0, s0, d0 = make()
    want = [s0.prove(d0, seed=sd) for sd in seeds]
    workers = [make() for _ in range(4)]
    got = [None] * 4

    def run(i):
        _, _, s, d = workers[i]
        got[i] = [s.prove(d, seed=sd) for sd in seeds]

    ths = [threading.Thread(target=run, args=(i,)) for i in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for i in range(4):
        assert got[i] == want, f"prover thread {i} diverged"
    for c, r, s, _ in workers + [(c0, r0, s0, d0)]:
        s.close()
        r.close()
        c.close()


def test_one_r1cs_upload_serves_every_context_of_the_device(oracle):
    """a pk_r1cs is immutable after creation (the heavy-line sums live in the calling context's workspace): four contexts prove
    concurrently over ONE uploaded R1CS -- with rows and a column above the heavy threshold -- and agree with the lone prover"""
    import threading

    import provekit_amd
    from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
    from provekit_amd.sparse_matrix import R1CS
    from test_gpu_witness import _mont, _noir_instance
    from provekit_amd.witness import WitnessProgram

    builders, acir, pub_idx, nw, coeffs, trips = _noir_instance(oracle, 3, n_in=6, n_prod=3000)  # constant column in half the rows
    nc = trips[0][0][-1] + 1
    m, m_0 = 13, 12
    interner = oracle.to_mont(oracle.ints_to_limbs(coeffs))
    c0 = provekit_amd.Context(0)
    shared = R1CS(c0, *(to_sparse(nc, nw, t) for t in trips), interner)
    cfgs = (WhirConfig.for_size(m, 6.0), blinding_config_for(m_0, 6.0))

    def make(c):
        return c, WhirR1CSScheme(c, shared, m, m_0, *cfgs), WitnessProgram(c, builders), c.upload(_mont(oracle, acir))

    lone = make(c0)
    seeds = [21, 22, 23]
    want = [lone[1].noir_prove(lone[2], lone[3], len(acir), pub_idx, seed=sd) for sd in seeds]
    workers = [make(provekit_amd.Context(0)) for _ in range(4)]
    got = [None] * 4

    def run(i):
        _, s, prog, d_acir = workers[i]
        got[i] = [s.noir_prove(prog, d_acir, len(acir), pub_idx, seed=sd) for sd in seeds]

    ths = [threading.Thread(target=run, args=(i,)) for i in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert all(g == want for g in got)
    for c, s, prog, _ in workers + [lone]:
        prog.close()
        s.close()
    shared.close()
    for c, *_ in workers + [lone]:
        c.close()


@pytest.mark.parametrize("m,m_0,nc,nw", [(10, 12, 4000, 500), (14, 6, 60, 8000), (12, 12, 4096, 2048), (9, 9, 300, 256)])
def test_scheme_shapes_fit_their_arena(ctx, oracle, m, m_0, nc, nw):
    """pk_scheme_create sizes the proof arena from (m, m_0, rate, batch, fold, num_witnesses): shapes far from m_0 = m - 1
    (more constraints than witnesses and the reverse) must prove without exhausting it, deterministically.  The instances are
    random (not satisfiable): this exercises allocation and control flow, the verifier-checked cases are above."""
    from test_gpu_r1cs_pow_commit import synth_r1cs

    from provekit_amd.field import random_field
    from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
    from provekit_amd.sparse_matrix import R1CS

    a, b, c = synth_r1cs(nc, nw, m * 31 + m_0)
    interner = random_field(17, 3)
    r1cs = R1CS(ctx, a, b, c, interner)
    scheme = WhirR1CSScheme(ctx, r1cs, m, m_0, WhirConfig.for_size(m, 5.0), blinding_config_for(m_0, 5.0))
    d_z = ctx.upload(random_field(nw, 9))
    p1, p2 = scheme.prove(d_z, seed=4), scheme.prove(d_z, seed=4)
    assert p1 == p2 and len(p1) > 1000
    scheme.close()
    r1cs.close()


def test_the_bench_statement_verifies(ctx, oracle):
    """bench.py's own workload -- its satisfiable synthetic R1CS, the satisfying witness the library computes for it, the
    reference's schedule for m = 21 -- gives a proof the independent verifier accepts: the number bench.py reports is the
    rate of proofs like this one."""
    import verifier as V

    sys.path.insert(0, ROOT)
    import bench

    from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for

    m, m_0 = 21, 20
    n_wit = (1 << (m - 1)) - 5
    r1cs, mats, interner, nc, n_in = bench.synth_r1cs(ctx, m_0, n_wit, seed=1234)
    d_z, _ = bench.satisfying_witness(ctx, r1cs, n_wit, nc, n_in, 99)
    cfg_w, cfg_b = WhirConfig.derive(m), blinding_config_for(m_0)
    scheme = WhirR1CSScheme(ctx, r1cs, m, m_0, cfg_w, cfg_b)
    proof = scheme.prove(d_z)  # production randomness

    def vcfg(c):
        return V.WhirConfig(c.n_vars, c.batch_size, c.folding_factor, c.starting_log_inv_rate, c.num_queries, c.ood_samples, c.pow_bits,
                            c.final_queries, c.final_pow_bits, c.commitment_ood_samples, c.final_folding_pow_bits)

    # ... including the R1CS relation itself: the deferred weight evaluations against bench's own matrices (C oracle SpMV)
    csr = [(M.new_row_indices, M.col_indices, M.values) for M in mats]
    assert V.verify(proof, scheme.domain_separator, m, m_0, vcfg(cfg_w), vcfg(cfg_b), r1cs=oracle.matrix_evaluator(nc, n_wit, csr, interner))
    scheme.close()
    r1cs.close()


def test_prove_under_a_caller_supplied_io_pattern(ctx, oracle):
    """pk_scheme_set_io_pattern: the drop-in caller's `create_io_pattern().as_bytes()` (provekit/common/src/whir_r1cs.rs:28-39)
    replaces the library's restatement -- other labels, same operations.  The proof then starts from THAT pattern's IV (the
    verifier built on the same bytes accepts it, the one built on the library's pattern does not), and a pattern that does not
    declare pk_prove's operations is refused up front."""
    import verifier as V
    from provekit_amd import ProveKitHipError
    from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for, create_io_pattern
    from provekit_amd.sparse_matrix import R1CS

    m, m_0, nc, n_in = 10, 8, 200, 150
    nw, z, coeffs, trips = satisfiable_r1cs(nc, n_in, 21)
    r1cs = R1CS(ctx, *(to_sparse(nc, nw, t) for t in trips), oracle.to_mont(oracle.ints_to_limbs(coeffs)))
    cfg_w, cfg_b = WhirConfig.for_size(m, 5.0), blinding_config_for(m_0, 5.0)
    scheme = WhirR1CSScheme(ctx, r1cs, m, m_0, cfg_w, cfg_b)
    ours = scheme.domain_separator
    assert ours == create_io_pattern(m_0, cfg_w, cfg_b)
    d_z = ctx.upload(oracle.to_mont(oracle.ints_to_limbs(z)))
    p_ours = scheme.prove(d_z, seed=4)

    def vcfg(c):
        return V.WhirConfig(c.n_vars, c.batch_size, c.folding_factor, c.starting_log_inv_rate, c.num_queries, c.ood_samples, c.pow_bits,
                            c.final_queries, c.final_pow_bits, c.commitment_ood_samples, c.final_folding_pow_bits)

    mats = [(t[0], t[1], [coeffs[v] for v in t[2]]) for t in trips]
    shape = (m, m_0, vcfg(cfg_w), vcfg(cfg_b))
    theirs = ours.replace(b"merkle_digest", b"root").replace(b"\0A2ood_ans", b"\0A1ood_ans\0A1ood_ans").replace(b"stir_queries", b"stir_challenge_indexes")
    scheme.set_io_pattern(theirs)
    assert scheme.domain_separator == theirs
    p_theirs = scheme.prove(d_z, seed=4)
    assert p_theirs != p_ours and p_theirs[:32] == p_ours[:32]  # same commitment, different challenges from the first squeeze on
    assert V.verify(p_theirs, theirs, *shape, r1cs=(nc, nw, mats))
    with pytest.raises(V.VerifyError):
        V.verify(p_theirs, ours, *shape)
    with pytest.raises(ProveKitHipError, match="IO pattern"):
        scheme.set_io_pattern(ours.replace(b"\0Hclaimed_evaluations", b"", 1))
    with pytest.raises(ProveKitHipError, match="IO pattern"):
        scheme.set_io_pattern(create_io_pattern(m_0, WhirConfig.for_size(m, 0.0), cfg_b))  # declares no grinding
    assert scheme.domain_separator == theirs  # a refused pattern changes nothing
    scheme.set_io_pattern(None)
    assert scheme.domain_separator == ours and scheme.prove(d_z, seed=4) == p_ours
    scheme.close()
    r1cs.close()


@pytest.mark.parametrize("m,m_0,nc,n_in,pow_bits", [(9, 7, 100, 60, 5.0), (12, 9, 500, 700, 4.0), (17, 16, 60000, 5000, 8.0)])
def test_latency_mode_gives_the_same_transcript(oracle, m, m_0, nc, n_in, pow_bits):
    """pk_ctx_set_latency_mode: every sumcheck round (20 cubic ones at the bench size, 4 per WHIR round) is enqueued one ahead,
    gated on a word of the pinned page the host writes once it has squeezed the challenge.  Same kernels, same challenges:
    the transcript must be byte for byte the one the plain mode writes -- seeded and, with both modes on one key, for fresh
    randomness too -- and the verifier accepts it; switching the mode off again restores the plain path."""
    import provekit_amd
    import verifier as V
    from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
    from provekit_amd.sparse_matrix import R1CS

    ctx = provekit_amd.Context(0)
    nw, z, coeffs, trips = satisfiable_r1cs(nc, n_in, 31)
    r1cs = R1CS(ctx, *(to_sparse(nc, nw, t) for t in trips), oracle.to_mont(oracle.ints_to_limbs(coeffs)))
    cfg_w, cfg_b = WhirConfig.for_size(m, pow_bits), blinding_config_for(m_0, pow_bits)
    scheme = WhirR1CSScheme(ctx, r1cs, m, m_0, cfg_w, cfg_b)
    d_z = ctx.upload(oracle.to_mont(oracle.ints_to_limbs(z)))
    plain = [scheme.prove(d_z, seed=s) for s in (1, 2, 3)]
    ctx.set_latency_mode(True)
    fast = [scheme.prove(d_z, seed=s) for s in (1, 2, 3)]
    assert fast == plain
    for _ in range(20):  # many proofs back to back: the gate's sequence numbers keep advancing, nothing is left pending
        assert scheme.prove(d_z, seed=2) == plain[1]
    ctx.set_latency_mode(False)
    assert scheme.prove(d_z, seed=3) == plain[2]

    def vcfg(c):
        return V.WhirConfig(c.n_vars, c.batch_size, c.folding_factor, c.starting_log_inv_rate, c.num_queries, c.ood_samples, c.pow_bits,
                            c.final_queries, c.final_pow_bits, c.commitment_ood_samples, c.final_folding_pow_bits)

    mats = [(t[0], t[1], [coeffs[v] for v in t[2]]) for t in trips]
    assert V.verify(fast[0], scheme.domain_separator, m, m_0, vcfg(cfg_w), vcfg(cfg_b), r1cs=(nc, nw, mats) if m <= 12 else None)
    scheme.close()
    r1cs.close()
    ctx.close()


def test_a_gate_that_gives_up_fails_the_proof_instead_of_proving_with_zero():
    """in latency mode a gated sumcheck kernel whose host stalled past the device-side bound used to run on with a
    ZERO challenge and pk_prove returned PK_OK with a transcript that does not verify.  Now the giving-up workgroup leaves a word in
    the pinned page and the host abandons the proof with PK_ERR_HIP.  A fresh process: the two test hooks (a short device bound, a host
    that sleeps before publishing each challenge) are process-wide (pk_selftest_set_hook)."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    code = r'''
import sys
sys.path[:0] = [%r, %r]
import torch; torch.cuda.is_available()
import oracle_lib as oracle
import provekit_amd
from provekit_amd._lib import ProveKitHipError
from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
from provekit_amd.sparse_matrix import R1CS
from test_gpu_prove import satisfiable_r1cs, to_sparse
from provekit_amd._lib import lib
assert lib.pk_selftest_set_hook(0, 64) == 0 and lib.pk_selftest_set_hook(1, 30000) == 0   # gate spin bound, host stall in us
m, m_0, nc, n_in = 12, 9, 500, 700
ctx = provekit_amd.Context(0)
nw, z, coeffs, trips = satisfiable_r1cs(nc, n_in, 31)
r1cs = R1CS(ctx, *(to_sparse(nc, nw, t) for t in trips), oracle.to_mont(oracle.ints_to_limbs(coeffs)))
scheme = WhirR1CSScheme(ctx, r1cs, m, m_0, WhirConfig.for_size(m, 4.0), blinding_config_for(m_0, 4.0))
d_z = ctx.upload(oracle.to_mont(oracle.ints_to_limbs(z)))
plain = scheme.prove(d_z, seed=1)          # no gates in the plain mode: unaffected by the hooks
ctx.set_latency_mode(True)
for attempt in range(2):
    try:
        scheme.prove(d_z, seed=1)
        print("SILENT")                     # the old behaviour: PK_OK
    except ProveKitHipError as e:
        print("REFUSED", e.code, "gave up waiting" in str(e))
ctx.set_latency_mode(False)
print("PLAIN_AGAIN", scheme.prove(d_z, seed=1) == plain)
''' % (os.path.dirname(here), here)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.split(" ")[0] in ("SILENT", "REFUSED", "PLAIN_AGAIN")]
    assert lines == ["REFUSED -3 True", "REFUSED -3 True", "PLAIN_AGAIN True"], (lines, out.stderr[-1500:])


def _vcfg(c):
    import verifier as V

    return V.WhirConfig(c.n_vars, c.batch_size, c.folding_factor, c.starting_log_inv_rate, list(c.num_queries), list(c.ood_samples), list(c.pow_bits),
                        c.final_queries, c.final_pow_bits, c.commitment_ood_samples, c.final_folding_pow_bits)


@pytest.mark.parametrize("m,m_0,nc,n_in,pow_bits", [(9, 7, 100, 60, 5.0), (12, 9, 500, 700, 4.0), (17, 16, 60000, 5000, 8.0)])
def test_transcript_equals_the_oracle_provers(ctx, oracle, m, m_0, nc, n_in, pow_bits):
    """WHOLE-PROOF parity: for the same statement, witness and 32-byte key, pk_prove (HIP kernels + the C++ host driver) and the
    oracle's prover (oracle/prover_ref.py: the reference's prove restated on the C oracle's kernels, transcript and scalar algebra in
    Python) must write the same bytes -- every commitment root, OOD answer, sumcheck message, proof-of-work nonce, opened leaf and
    authentication path, in order.  Also in latency mode."""
    import prover_ref as PR
    from test_prover_ref import small_instance

    from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
    from provekit_amd.sparse_matrix import R1CS

    nw, z, coeffs, trips, mats = small_instance(nc, n_in, 31)
    interner = oracle.to_mont(oracle.ints_to_limbs(coeffs))
    zm = oracle.to_mont(oracle.ints_to_limbs(z))
    cfg_w, cfg_b = WhirConfig.for_size(m, pow_bits), blinding_config_for(m_0, pow_bits)
    r1cs = R1CS(ctx, *(to_sparse(nc, nw, t) for t in trips), interner)
    scheme = WhirR1CSScheme(ctx, r1cs, m, m_0, cfg_w, cfg_b)
    d_z = ctx.upload(zm)
    for seed in (3, 4):
        want = PR.prove(scheme.domain_separator, m, m_0, _vcfg(cfg_w), _vcfg(cfg_b), (nc, nw, mats, interner), zm, seed.to_bytes(32, "little"))
        got = scheme.prove(d_z, seed=seed)
        assert len(got) == len(want)
        if got != want:
            first = next(i for i in range(len(got)) if got[i] != want[i])
            raise AssertionError(f"pk_prove's transcript differs from the oracle prover's from byte {first} of {len(got)}")
    ctx.set_latency_mode(True)
    try:
        assert scheme.prove(d_z, seed=4) == want
    finally:
        ctx.set_latency_mode(False)
    scheme.close()
    r1cs.close()


@pytest.mark.parametrize("m", [21, 23, 25])
def test_transcript_equals_the_oracle_provers_at_the_bench_size(ctx, oracle, m):
    """the same equality at BASELINE configs[1]'s size (m = 21, m_0 = 20) under the reference's own derived schedule (109 / 28 / 16 / 11
    queries, grinding up to 19 bits): 260 KB of proof, byte for byte -- at configs[2]'s size class (m = 23: 2^20-leaf trees, three-pass
    NTTs, five WHIR rounds) -- and at configs[3]'s (m = 25: 2^22-leaf trees, six WHIR rounds, 23-bit grinding; ~20 GB of host memory
    and half a minute of all cores for the oracle's side)"""
    import prover_ref as PR

    from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
    from provekit_amd.sparse_matrix import R1CS, SparseMatrix

    m_0 = m - 1
    nc, nw, mats, interner, z = size_class_instance(oracle, m)
    cfg_w, cfg_b = WhirConfig.derive(m), blinding_config_for(m_0)
    r1cs = R1CS(ctx, *(SparseMatrix(nc, nw, *t) for t in mats), interner)
    scheme = WhirR1CSScheme(ctx, r1cs, m, m_0, cfg_w, cfg_b)
    stage = {}
    want = PR.prove(scheme.domain_separator, m, m_0, _vcfg(cfg_w), _vcfg(cfg_b), (nc, nw, mats, interner), z, (11).to_bytes(32, "little"), stage)
    print("oracle prover stage_s", stage)
    got = scheme.prove(ctx.upload(z), seed=11)
    assert len(got) == len(want) and got == want
    # ... and the oracle prover's proof is one the oracle verifier accepts, matrix evaluation of the deferred weights included
    import verifier as V

    assert V.verify(want, scheme.domain_separator, m, m_0, _vcfg(cfg_w), _vcfg(cfg_b), r1cs=oracle.matrix_evaluator(nc, nw, mats, interner))
    assert abs(sum(v for k, v in stage.items() if k != "total") - stage["total"]) < 0.02 * stage["total"]
    scheme.close()
    r1cs.close()


@pytest.mark.parametrize("mode", ["block", "poll"])
def test_blocking_host_wait_gives_the_same_proofs(ctx, oracle, mode):
    """pk_device_set_host_wait: PK_WAIT_BLOCK -- prover threads sleep on the completion interrupt instead of spinning -- and PK_WAIT_POLL -- the
    library's own query-and-sleep loop -- chosen before any context exists (a fresh process): four provers in flight write the transcripts
    this process's spinning prover writes"""
    import hashlib
    import subprocess

    from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
    from provekit_amd.sparse_matrix import R1CS

    m, m_0, nc, n_in = 12, 9, 500, 700
    nw, z, coeffs, trips = satisfiable_r1cs(nc, n_in, 31)
    r1cs = R1CS(ctx, *(to_sparse(nc, nw, t) for t in trips), oracle.to_mont(oracle.ints_to_limbs(coeffs)))
    scheme = WhirR1CSScheme(ctx, r1cs, m, m_0, WhirConfig.for_size(m, 4.0), blinding_config_for(m_0, 4.0))
    d_z = ctx.upload(oracle.to_mont(oracle.ints_to_limbs(z)))
    want = [hashlib.sha256(scheme.prove(d_z, seed=s)).hexdigest() for s in range(1, 9)]
    scheme.close()
    r1cs.close()
    here = os.path.dirname(os.path.abspath(__file__))
    code = r'''
import sys, threading, hashlib
sys.path[:0] = [%r, %r]
import torch; torch.cuda.is_available()
import oracle_lib as oracle
import provekit_amd
provekit_amd.Context.set_host_wait(0, %r)   # before the first context of this process
from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
from provekit_amd.sparse_matrix import R1CS
from test_gpu_prove import satisfiable_r1cs, to_sparse
m, m_0, nc, n_in = 12, 9, 500, 700
nw, z, coeffs, trips = satisfiable_r1cs(nc, n_in, 31)
out = {}
def work(w):
    c = provekit_amd.Context(0)
    r1cs = R1CS(c, *(to_sparse(nc, nw, t) for t in trips), oracle.to_mont(oracle.ints_to_limbs(coeffs)))
    s = WhirR1CSScheme(c, r1cs, m, m_0, WhirConfig.for_size(m, 4.0), blinding_config_for(m_0, 4.0))
    d = c.upload(oracle.to_mont(oracle.ints_to_limbs(z)))
    for seed in (2 * w + 1, 2 * w + 2):
        out[seed] = hashlib.sha256(s.prove(d, seed=seed)).hexdigest()
ths = [threading.Thread(target=work, args=(w,)) for w in range(4)]
[t.start() for t in ths]; [t.join() for t in ths]
print("HASHES", " ".join(out[s] for s in range(1, 9)))
''' % (os.path.dirname(here), here, mode)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    got = [l for l in res.stdout.splitlines() if l.startswith("HASHES ")][-1].split()[1:]
    assert got == want
