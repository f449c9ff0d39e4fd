"""GPU: the R1CS witness builders on the device (csrc/witness.hip through the C ABI) against the restatement of
the reference's sequential solver (oracle/witness_ref.py): every witness bit for bit, the None pattern, and the reference's panics."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))


def _mont(oracle, ints):
    return oracle.to_mont(oracle.ints_to_limbs([int(x) for x in ints])) if len(ints) else np.zeros((0, 4), np.uint64)


def _check(ctx, oracle, builders, acir, ch, n):
    import witness_ref as R

    from provekit_amd.witness import WitnessProgram

    want = R.solve_witness_vec(builders, acir, ch, n)
    prog = WitnessProgram(ctx, builders)
    assert prog.n_challenges == len(ch) and prog.n_acir == len(acir)
    w, is_set = prog.solve_witness_vec(_mont(oracle, acir), _mont(oracle, ch), n)
    prog.close()
    assert [bool(x) for x in is_set] == [x is not None for x in want]
    got = oracle.limbs_to_ints(oracle.from_mont(w))
    for i, x in enumerate(want):
        assert got[i] == (x if x is not None else 0), f"witness {i}"
    return want


@pytest.mark.parametrize("seed,n,chain", [(11, 60, False), (12, 700, False), (13, 700, True), (14, 6000, False), (15, 3000, True)])
def test_random_programs_match_the_sequential_solver(ctx, oracle, seed, n, chain):
    from witness_gen import random_program

    builders, acir, ch, nw = random_program(seed, n, chain=chain)
    want = _check(ctx, oracle, builders, acir, ch, nw)
    assert sum(x is None for x in want) >= 2


def test_reference_digit_case_on_the_device(ctx, oracle):
    """digits.rs:88-99 through the whole path: 3 + 2*256 + 256*256 in bases [8, 8, 4] -> 3, 2, 1"""
    from provekit_amd.witness import WitnessBuilder as WB

    builders = [WB.Acir(0, 0), WB.DigitalDecomposition([8, 8, 4], [0], 1)]
    want = _check(ctx, oracle, builders, [3 + 2 * 256 + 256 * 256], [], 4)
    assert want == [3 + 2 * 256 + 256 * 256, 3, 2, 1]


def test_spice_block_replays_like_the_reference(ctx, oracle):
    """loads and stores that hit the same cell repeatedly, a cell never touched, a load of a never-written value (None stays None)"""
    from provekit_amd.witness import WitnessBuilder as WB

    M = 4
    b = [WB.Acir(i, i) for i in range(8)]  # 0..3: addresses 0,1,2,1 ; 4..7: initial values
    acir = [0, 1, 2, 1, 100, 101, 102, 103]
    b += [WB.Constant(8, 777), WB.Constant(9, 888)]
    # witness 10 is never written: loading "it" leaves None in memory
    ops = [("store", 1, 20, 8, 21), ("load", 3, 8, 22), ("store", 3, 23, 9, 24), ("load", 0, 10, 25), ("store", 0, 26, 8, 27), ("load", 2, 9, 28)]
    b.append(WB.SpiceWitnesses(M, 4, ops, 30, 34))
    want = _check(ctx, oracle, b, acir, [], 40)
    assert want[20] == 101 and want[21] == 0 and want[22] == 1 and want[23] == 777 and want[24] == 2
    assert want[26] is None and want[27] == 4  # the old value of cell 0 is the None a load left there
    assert want[30:34] == [777, 888, 888, 103] and want[34:38] == [5, 3, 6, 0]


@pytest.mark.parametrize("case,msg", [("inverse", "inverse of zero"), ("digits", "Higher order bits are not zero"),
                                       ("range", "multiplicity table"), ("spice", "memory address")])
def test_the_reference_panics_are_errors_naming_the_builder(ctx, oracle, case, msg):
    import witness_ref as R

    from provekit_amd import ProveKitHipError
    from provekit_amd.witness import WitnessBuilder as WB
    from provekit_amd.witness import WitnessProgram

    b = [WB.Constant(0, 1), WB.Acir(1, 0)]
    acir = {"inverse": [0], "digits": [1 << 20], "range": [300], "spice": [9]}[case]
    if case == "inverse":
        b.append(WB.Inverse(2, 1))
    elif case == "digits":
        b.append(WB.DigitalDecomposition([8, 8, 4], [1], 2))
    elif case == "range":
        b.append(WB.MultiplicitiesForRange(2, 256, [1]))
    else:
        b += [WB.Acir(2, 0), WB.SpiceWitnesses(1, 2, [("load", 1, 0, 5)], 6, 7)]
    with pytest.raises(R.SolverPanic):
        R.solve_witness_vec(b, acir, [], 300)
    prog = WitnessProgram(ctx, b)
    with pytest.raises(ProveKitHipError, match=msg) as e:
        prog.solve_witness_vec(_mont(oracle, acir), np.zeros((0, 4), np.uint64), 300)
    assert f"witness builder {len(b) - 1}" in str(e.value)
    prog.close()


def test_a_poseidon_sized_list(ctx, oracle):
    """2^17 builders with wide levels (the shape of a real circuit's list): solved witnesses equal the sequential solver's"""
    from witness_gen import random_program

    builders, acir, ch, nw = random_program(99, 1 << 17)
    _check(ctx, oracle, builders, acir, ch, nw)


def _noir_instance(oracle, seed, n_in=6, n_prod=40):
    """a builder list that derives a whole R1CS witness and the R1CS it satisfies: z0 = 1, inputs from ACIR, two challenges drawn
    from the witness transcript, then products / sums / inverses / a challenge-dependent linear operation, each with its constraint;
    the last three witnesses are written by nobody (fill_witness) and constrained by nothing"""
    import witness_ref as R

    from provekit_amd.witness import WitnessBuilder as WB

    rng = np.random.default_rng(seed)
    acir = [int.from_bytes(rng.bytes(32), "little") % R.P for _ in range(n_in + 3)]  # ACIR indices 3.. hold the inputs
    b = [WB.Constant(0, 1)] + [WB.Acir(1 + i, 3 + i) for i in range(n_in)] + [WB.Challenge(1 + n_in), WB.Challenge(2 + n_in)]
    nxt = 3 + n_in
    coeffs = [1, R.P - 1, 5]
    A, B, Cm = ([], [], []), ([], [], []), ([], [], [])

    def row(a_terms, b_terms, c_terms):
        i = (A[0][-1] + 1) if A[0] else 0
        for M, terms in ((A, a_terms), (B, b_terms), (Cm, c_terms)):
            for col, v in sorted(terms):
                M[0].append(i); M[1].append(col); M[2].append(v)

    for k in range(n_prod):
        x, y = (int(v) for v in rng.integers(1, nxt, size=2))
        kind = k % 4
        if (kind == 2 and x == y) or (kind == 3 and x == 1 + n_in):
            kind = 0
        if kind == 0:
            b.append(WB.Product(nxt, x, y)); row([(x, 0)], [(y, 0)], [(nxt, 0)])
        elif kind == 1:
            b.append(WB.Inverse(nxt, x)); row([(x, 0)], [(nxt, 0)], [(0, 0)])
        elif kind == 2:
            b.append(WB.Sum(nxt, [(None, x), (5, y)])); row([(x, 0), (y, 2)], [(0, 0)], [(nxt, 0)])
        else:  # x + the first challenge
            b.append(WB.Sum(nxt, [(None, x), (None, 1 + n_in)])); row([(x, 0), (1 + n_in, 0)], [(0, 0)], [(nxt, 0)])
        nxt += 1
    nw = nxt + 3
    return b, acir, [0, 4], nw, coeffs, (A, B, Cm)


@pytest.mark.parametrize("seed", [31, 32])
def test_noir_prove_is_challenges_then_builders_then_fill_then_prove(ctx, oracle, seed):
    """pk_noir_prove (NoirProofSchemeProver::prove after ACVM, noir_proof_scheme.rs:63-92) = its four steps called one by one, byte
    for byte; the witness is the sequential solver's on the transcript's challenges; the proof is accepted by the verifier"""
    import verifier as V
    import witness_ref as R
    from test_gpu_prove import to_sparse

    from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
    from provekit_amd.sparse_matrix import R1CS
    from provekit_amd.witness import WitnessProgram, fill_witness, witness_challenges
    from provekit_amd._lib import lib

    builders, acir, pub_idx, nw, coeffs, trips = _noir_instance(oracle, seed)
    nc = trips[0][0][-1] + 1
    m, m_0 = 10, 6
    assert nw <= 1 << (m - 1) and nc <= 1 << m_0
    r1cs = R1CS(ctx, *(to_sparse(nc, nw, t) for t in trips), oracle.to_mont(oracle.ints_to_limbs(coeffs)))
    cfg_w, cfg_b = WhirConfig.for_size(m, 4.0), blinding_config_for(m_0, 4.0)
    scheme = WhirR1CSScheme(ctx, r1cs, m, m_0, cfg_w, cfg_b)
    prog = WitnessProgram(ctx, builders)
    d_acir = ctx.upload(_mont(oracle, acir))
    proof = scheme.noir_prove(prog, d_acir, len(acir), pub_idx, seed=seed)
    assert proof == scheme.noir_prove(prog, d_acir, len(acir), pub_idx, seed=seed)
    ctx.set_latency_mode(True)  # the same call in latency mode (pk_ctx_set_latency_mode): the same bytes
    try:
        assert proof == scheme.noir_prove(prog, d_acir, len(acir), pub_idx, seed=seed)
    finally:
        ctx.set_latency_mode(False)

    # step by step, with the oracle beside every step
    pub = [acir[i] for i in pub_idx]
    ch = R.witness_challenges(nc, nw, pub, 2)
    got_ch = witness_challenges(nc, nw, _mont(oracle, pub), 2)
    assert oracle.limbs_to_ints(oracle.from_mont(got_ch)) == ch
    want = R.solve_witness_vec(builders, acir, ch, nw)
    assert sum(x is None for x in want) == 3
    d_w, d_set = ctx.alloc_fe(nw), ctx.alloc(nw)
    ctx._check(lib.pk_witness_solve(ctx.handle, prog.handle, d_acir.ptr, len(acir), got_ch.ctypes.data, 2, d_w.ptr, nw, d_set.ptr))
    assert fill_witness(ctx, d_w, d_set, nw, seed=seed) == 3
    z = oracle.limbs_to_ints(oracle.from_mont(ctx.download_fe(d_w, nw)))
    for i, x in enumerate(want):
        assert (z[i] == x) if x is not None else (0 < z[i] < 1 << 128), f"witness {i}"
    assert len({z[i] for i, x in enumerate(want) if x is None}) == 3
    assert r1cs.test_witness_satisfaction(d_w) is None
    assert scheme.prove(d_w, seed=seed) == proof

    def vcfg(c):
        return V.WhirConfig(c.n_vars, c.batch_size, c.folding_factor, c.starting_log_inv_rate, c.num_queries, c.ood_samples, c.pow_bits,
                            c.final_queries, c.final_pow_bits, c.commitment_ood_samples, c.final_folding_pow_bits)

    mats = [(t[0], t[1], [coeffs[v] for v in t[2]]) for t in trips]
    assert V.verify(proof, scheme.domain_separator, m, m_0, vcfg(cfg_w), vcfg(cfg_b), r1cs=(nc, nw, mats))
    # fresh OS randomness: another transcript, same statement, still accepted
    fresh = scheme.noir_prove(prog, d_acir, len(acir), pub_idx)
    assert fresh != proof and V.verify(fresh, scheme.domain_separator, m, m_0, vcfg(cfg_w), vcfg(cfg_b), r1cs=(nc, nw, mats))
    prog.close()
    scheme.close()


def test_fill_witness_draws_one_u128_per_unset_entry(ctx, oracle):
    """fill_witness (mod.rs:15-30): set entries untouched, unset ones = word (i mod 4) of ChaCha12 block i / 4, stream 6 (the block
    function is RFC-pinned in test_gpu_rng), as a field element; the count is the reference's log line"""
    import ctypes as C

    from provekit_amd._lib import lib
    from provekit_amd.witness import fill_witness

    n = 1000
    rng = np.random.default_rng(5)
    is_set = rng.integers(0, 2, size=n).astype(np.uint8)
    vals = [int(x) for x in rng.integers(1, 2**62, size=n)]
    d_w, d_set = ctx.upload(_mont(oracle, vals)), ctx.alloc(n)
    ctx._check(lib.pk_memcpy_h2d(ctx.handle, d_set.ptr, is_set.ctypes.data, n))
    seed = bytes(range(32))
    assert fill_witness(ctx, d_w, d_set, n, seed=seed) == int((is_set == 0).sum())
    got = oracle.limbs_to_ints(oracle.from_mont(ctx.download_fe(d_w, n)))
    blk = (C.c_uint8 * 64)()
    for i in range(n):
        if is_set[i]:
            assert got[i] == vals[i]
        else:
            assert lib.pk_selftest_chacha((C.c_uint8 * 32).from_buffer_copy(seed), i >> 2, 6, 0, 12, blk) == 0
            assert got[i] == int.from_bytes(bytes(blk)[16 * (i & 3): 16 * (i & 3) + 16], "little"), i


def test_long_sums_are_summed_by_workgroups(ctx, oracle):
    """Sum builders with 128 / 129 / 1024 / 1025 / 5000 / 70000 terms (a LogUp grand sum has a term per lookup): above 128 terms
    the sum leaves the per-lane item list (csrc/witness.hip "long sums"); values, levels (a long sum of long sums, consumers one
    level up) and the None pattern still equal the sequential solver's"""
    import random

    import witness_ref as R

    from provekit_amd.witness import WitnessBuilder as WB, inspect_witness_builders, encode_witness_builders

    rnd = random.Random(21)
    n_in = 3000
    acir = [rnd.randrange(R.P) for _ in range(n_in)]
    b = [WB.Acir(i, i) for i in range(n_in)]
    nxt = n_in
    sums = []
    for n_terms in (128, 129, 1024, 1025, 5000, 70000):
        terms = [(None if rnd.random() < 0.3 else rnd.randrange(R.P), rnd.randrange(n_in)) for _ in range(n_terms)]
        b.append(WB.Sum(nxt, terms))
        sums.append(nxt)
        nxt += 1
    # products of the sums (consumers one level up), then a long sum over those products and the first sums again
    prods = []
    for s_ in sums:
        b.append(WB.Product(nxt, s_, rnd.randrange(n_in)))
        prods.append(nxt)
        nxt += 1
    terms = [(rnd.randrange(R.P), rnd.choice(sums + prods + list(range(n_in)))) for _ in range(300)] + [(3, prods[-1]), (None, sums[1])]
    b.append(WB.Sum(nxt, terms))
    top = nxt
    nxt += 1
    b.append(WB.Inverse(nxt, top))
    nxt += 2  # one witness nobody writes
    info = inspect_witness_builders(encode_witness_builders(b))
    assert info["n_levels"] == 5 and info["n_items"] == len(b) - 6  # the six sums above 128 terms are not items
    want = _check(ctx, oracle, b, acir, [], nxt)
    assert want[-1] is None and want[top] == sum((1 if c is None else c) * want[i] for c, i in terms) % R.P


def test_a_large_spice_block(ctx, oracle):
    """60 k loads / stores over 4096 cells (the radix sort spans many workgroups; cells hit dozens of times, some never): read
    timestamps, old values, final values and final timestamps equal the sequential replay's"""
    import random

    from provekit_amd.witness import WitnessBuilder as WB

    rnd = random.Random(8)
    M, K = 4096, 60000
    acir = [rnd.randrange(M) if rnd.random() < 0.97 else rnd.randrange(16) for _ in range(K)] + [rnd.randrange(1 << 200) for _ in range(K + M)]
    b = [WB.Acir(i, i) for i in range(2 * K + M)]  # 0..K-1 addresses, K..2K-1 values to store / loaded values, 2K..2K+M-1 initial memory
    nxt = 2 * K + M
    ops = []
    for k in range(K):
        if rnd.random() < 0.5:
            ops.append(("load", k, K + k, nxt))
            nxt += 1
        else:
            ops.append(("store", k, nxt, K + k, nxt + 1))
            nxt += 2
    rv, rt = nxt, nxt + M
    b.append(WB.SpiceWitnesses(M, 2 * K, ops, rv, rt))
    want = _check(ctx, oracle, b, acir, [], rt + M)
    assert want[rt + M - 1] is not None and max(want[rt : rt + M]) > 50000


def test_noir_prove_at_scale(ctx, oracle):
    """100 k builders / constraints (m = 18), the constant-one column in half of the rows (heavy lines), a LogUp-style closing: one
    Sum over 20 k inverses with its one-row constraint (a heavy row and a long sum): the device witness equals the sequential
    solver's everywhere, satisfies the R1CS, and the one-call proof is accepted by the verifier with the matrices"""
    import verifier as V
    import witness_ref as R
    from test_gpu_prove import to_sparse

    from provekit_amd._lib import lib
    from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
    from provekit_amd.sparse_matrix import R1CS
    from provekit_amd.witness import WitnessBuilder as WB, WitnessProgram, fill_witness, witness_challenges

    builders, acir, pub_idx, nw, coeffs, trips = _noir_instance(oracle, 41, n_in=50, n_prod=100000)
    A, B, Cm = trips
    # the closing sum: every inverse the list produced, one Sum builder, one constraint  (sum a_i z_i) * 1 = z_out
    inv = [b[1] for b in builders if b[0] == 7][:20000]
    out = nw - 3  # the first of the three witnesses nobody wrote
    builders.append(WB.Sum(out, [(None, i) for i in inv]))
    row = A[0][-1] + 1
    for col in sorted(inv):
        A[0].append(row); A[1].append(col); A[2].append(0)
    B[0].append(row); B[1].append(0); B[2].append(0)
    Cm[0].append(row); Cm[1].append(out); Cm[2].append(0)
    nc = row + 1
    m, m_0 = 18, 17
    assert nw <= 1 << (m - 1) and nc <= 1 << m_0
    r1cs = R1CS(ctx, *(to_sparse(nc, nw, t) for t in trips), oracle.to_mont(oracle.ints_to_limbs(coeffs)))
    cfg_w, cfg_b = WhirConfig.for_size(m, 8.0), blinding_config_for(m_0, 8.0)
    scheme = WhirR1CSScheme(ctx, r1cs, m, m_0, cfg_w, cfg_b)
    prog = WitnessProgram(ctx, builders)
    d_acir = ctx.upload(_mont(oracle, acir))
    proof = scheme.noir_prove(prog, d_acir, len(acir), pub_idx, seed=3)
    # the witness, step by step, against the oracle
    ch = R.witness_challenges(nc, nw, [acir[i] for i in pub_idx], 2)
    want = R.solve_witness_vec(builders, acir, ch, nw)
    d_w, d_set = ctx.alloc_fe(nw), ctx.alloc(nw)
    chm = _mont(oracle, ch)
    ctx._check(lib.pk_witness_solve(ctx.handle, prog.handle, d_acir.ptr, len(acir), chm.ctypes.data, 2, d_w.ptr, nw, d_set.ptr))
    assert fill_witness(ctx, d_w, d_set, nw, seed=3) == 2
    z = oracle.limbs_to_ints(oracle.from_mont(ctx.download_fe(d_w, nw)))
    assert [zi for zi, x in zip(z, want) if x is not None] == [x for x in want if x is not None]
    assert r1cs.test_witness_satisfaction(d_w) is None
    assert scheme.prove(d_w, seed=3) == proof

    def vcfg(c):
        return V.WhirConfig(c.n_vars, c.batch_size, c.folding_factor, c.starting_log_inv_rate, c.num_queries, c.ood_samples, c.pow_bits,
                            c.final_queries, c.final_pow_bits, c.commitment_ood_samples, c.final_folding_pow_bits)

    mats = [(t[0], t[1], [coeffs[v] for v in t[2]]) for t in trips]
    assert V.verify(proof, scheme.domain_separator, m, m_0, vcfg(cfg_w), vcfg(cfg_b), r1cs=(nc, nw, mats))
    prog.close()
    scheme.close()
    r1cs.close()


def test_placement_advice_and_acir_reads(ctx, oracle):
    """pk_witness_program_placement: a chain (every builder consumes its predecessor: one level per builder, ~2.6 us each) is sent
    to the host solver, a wide list of the same length stays on the device; pk_witness_program_acir_reads lists exactly the ACIR
    indices the Acir builders read (the Rust side checks its WitnessMap against it where the reference would unwrap)."""
    from witness_gen import random_program

    from provekit_amd.witness import WitnessBuilder as WB, WitnessProgram

    n = 20000
    chain = [WB.Acir(0, 7)] + [WB.Product(i, i - 1, i - 1) for i in range(1, n)]
    wide = [WB.Acir(0, 7), WB.Acir(1, 3)] + [WB.Product(i, 0, 1) for i in range(2, n)]
    pc, pw = WitnessProgram(ctx, chain), WitnessProgram(ctx, wide)
    a, b = pc.placement(), pw.placement()
    assert a["n_levels"] >= n - 1 and a["prefer_host"] and a["est_device_us"] > 10 * a["est_host_us"]
    assert b["n_levels"] <= 3 and not b["prefer_host"] and b["est_device_us"] < b["est_host_us"]
    assert pc.acir_reads() == [7] and pw.acir_reads() == [3, 7]
    builders, acir, ch, nw = random_program(5, 1 << 15)
    big = WitnessProgram(ctx, builders)
    assert not big.placement()["prefer_host"]  # a list shaped like a constraint system's: depth far below length
    reads = big.acir_reads()
    assert reads == sorted(set(reads)) and all(r < big.n_acir for r in reads) and (not reads or reads[-1] == big.n_acir - 1)
    for p in (pc, pw, big):
        p.close()


def test_overwritten_witnesses_keep_their_sequential_meaning(ctx, oracle):
    """the reference's solve_witness_vec runs the list in order, so a builder may overwrite a witness an earlier one
    solved (the last writer wins) and readers in between see the value of their place in the list; a Spice block may copy a None a
    later builder solves.  The levelled solver must agree with the sequential restatement on such lists: a writer is ordered after
    the previous writer and after every reader of the version it replaces."""
    import witness_ref as R

    from provekit_amd.witness import WitnessBuilder as WB

    P = oracle.P
    builders = [WB.Acir(0, 0), WB.Acir(1, 1), WB.Constant(2, 7), WB.Product(3, 2, 0),  # w3 = 7 a0
                WB.Constant(2, 11),                                                     # w2 overwritten
                WB.Product(4, 2, 1),                                                    # w4 = 11 a1 (the new version)
                WB.Sum(2, [(None, 2), (5, 3)]),                                         # w2 = w2 + 5 w3: reads and rewrites itself
                WB.Product(5, 2, 2), WB.Inverse(6, 5), WB.Constant(3, 1), WB.Product(7, 3, 6)]
    for k in range(40):  # a longer ping-pong on two cells, wide enough to cross the narrow-run path
        builders += [WB.Product(8, 7 if k == 0 else 9, 2), WB.Sum(9, [(3, 8), (None, 0)]), WB.Constant(8, k + 2)]
    acir = [123456789, 987654321]
    nw = 10
    want = R.solve_witness_vec(builders, acir, [], nw)
    got = _check(ctx, oracle, builders, acir, [], nw)
    assert got == want and want[2] == (11 + 5 * 7 * acir[0]) % P and want[8] == 41
