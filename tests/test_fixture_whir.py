"""Known answers DERIVED FROM THE REFERENCE'S OWN PROOF (tests/golden/fixture_whir.json, minted by gen_fixture_whir.py from
tooling/provekit-bench/benches/poseidon-1000.np): commitments, folds, OOD evaluations and sumcheck relations the real
prover produced.  CPU half: the C oracle reproduces them.  GPU half (marked gpu): the HIP library reproduces them through
the C ABI.  Hash version 1 throughout (the fixture predates the Skyscraper v2 switch)."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))  # pyref, verifier

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FX = json.load(open(os.path.join(G, "fixture_whir.json")))
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
ints = lambda hs: [int(h, 16) for h in hs]


def quad(evals, x):
    """value at x of the quadratic through (0,e0),(1,e1),(2,e2) -- utilities.go:148-154"""
    e0, e1, e2 = evals
    a = (e2 - 2 * e1 + e0) * pow(2, -1, P) % P
    return (a * x * x + (e1 - e0 - a) * x + e0) % P


def mont(o, xs):
    return o.to_mont(o.ints_to_limbs(xs))


def canon_ints(o, a):
    return o.limbs_to_ints(o.from_mont(a))


# ------------------------------------------------------------------------------------------------ CPU: oracle
def test_oracle_blinding_commitment_is_the_references(oracle):
    b = FX["blinding"]
    f0, f1 = ints(b["f0"]), ints(b["f1"])
    leaves = oracle.rs_encode(mont(oracle, f0 + f1), 2, 8, 1, 4)
    assert [canon_ints(oracle, l) for l in leaves] == [ints(l) for l in b["leaves_T0"]]
    assert oracle.limbs_to_ints(oracle.merkle_commit(leaves, version=1)[1:2])[0] == int(b["root_T0"], 16)
    # evaluation layout: [4*m_0 blinding coefficients | zero padding | mask]  (whir_r1cs.rs:228-250, 186-199)
    ev = canon_ints(oracle, oracle.to_evals(mont(oracle, f0), 8))
    assert not any(ev[80:128]) and all(ev[:80]) and all(ev[128:])
    assert canon_ints(oracle, oracle.to_coeffs(oracle.to_evals(mont(oracle, f0), 8), 8)) == f0
    # sum_over_hypercube (whir_r1cs.rs:172-180) of the 20 cubics == the scalar absorbed by the real prover
    cub = lambda c, x: (c[0] + c[1] * x + c[2] * x * x + c[3] * x ** 3) % P
    assert pow(2, 19, P) * sum(cub(ev[4 * i : 4 * i + 4], 0) + cub(ev[4 * i : 4 * i + 4], 1) for i in range(20)) % P == int(b["sum_g"], 16)
    # OOD answers are univariate evaluations at one common point
    z = mont(oracle, [int(b["ood_point"], 16)])[0]
    for f, ans in zip((f0, f1), b["ood_answers"]):
        assert canon_ints(oracle, oracle.eval_univariate(mont(oracle, f), z)[None])[0] == int(ans, 16)


def test_oracle_fold_batching_and_round_tree(oracle):
    b = FX["blinding"]
    f0, f1, beta, r = mont(oracle, ints(b["f0"])), mont(oracle, ints(b["f1"])), mont(oracle, [int(b["batching_randomness"], 16)])[0], mont(oracle, ints(b["folding_randomness"]))
    folded = oracle.fold_coeffs(oracle.vec_axpy(f0, beta, f1), 8, r)
    assert canon_ints(oracle, folded) == ints(b["f_folded"])
    # sumcheck binding (whir_utilities.go:102-125): h_k(r_k) == h_{k+1}(0) + h_{k+1}(1)
    H, rr = [ints(h) for h in b["sumcheck_evals"]], ints(b["folding_randomness"])
    for k in range(3):
        assert quad(H[k], rr[k]) == (H[k + 1][0] + H[k + 1][1]) % P
    # the round commitment of the folded polynomial: 16 leaves at rate 2^-4, every leaf = the 16 coefficients
    leaves = oracle.rs_encode(folded, 1, 4, 4, 4)
    assert leaves.shape[0] == 16 and all(canon_ints(oracle, l) == ints(b["f_folded"]) for l in leaves)
    assert oracle.limbs_to_ints(oracle.merkle_commit(leaves, version=1)[1:2])[0] == int(b["root_T1"], 16)


def test_oracle_witness_tail(oracle):
    w = FX["witness_tail"]
    f4 = mont(oracle, ints(w["T4"]["f4"]))
    leaves = oracle.rs_encode(f4, 1, w["T4"]["n_vars"], w["T4"]["log_inv_rate"], 4)
    assert leaves.shape[0] == 1 << w["T4"]["height"]
    for i, l in zip(w["T4"]["opened"], w["T4"]["leaves"]):
        assert canon_ints(oracle, leaves[i]) == ints(l)
    assert oracle.limbs_to_ints(oracle.merkle_commit(leaves, version=1)[1:2])[0] == int(w["T4"]["root"], 16)
    r4 = mont(oracle, ints(w["final"]["folding_randomness"]))
    assert canon_ints(oracle, oracle.fold_coeffs(f4, 5, r4)) == ints(w["final"]["final_coefficients"])
    H, rr = [ints(h) for h in w["final"]["sumcheck_evals"]], ints(w["final"]["folding_randomness"])
    for k in range(3):
        assert quad(H[k], rr[k]) == (H[k + 1][0] + H[k + 1][1]) % P
    # STIR consistency one level up (whir_utilities.go:180-186, whir.go:99,141): fold(leaf_i, r) == f4(w_{2^15}^i)
    r3 = mont(oracle, ints(w["T3"]["folding_randomness"]))
    w15 = oracle.root_of_unity(15)
    for i, l in zip(w["T3"]["opened"], w["T3"]["leaves"]):
        x = np.empty(4, dtype=np.uint64)
        oracle.L.pko_fe_pow(oracle._p(w15), int(i), oracle._p(x))
        lhs = oracle.fold_coeffs(mont(oracle, ints(l)), 4, r3)[0]
        assert np.array_equal(lhs, oracle.eval_univariate(f4, x))



def test_oracle_zk_sumcheck_messages_are_the_references(oracle):
    """Round 4: the reference's 20 zk-sumcheck challenges and rho, RECOVERED from its proof (gen_fixture_whir.py): alpha_i is a root of
    hhat_i(X) = hhat_{i+1}(0) + hhat_{i+1}(1) -- the relation provekit/verifier/src/whir_r1cs.rs:131-144 checks, with the four scalars read as
    monomial coefficients, low degree first -- and exactly ONE choice among the 3^10 candidate vectors also gives both "Polynomial sums" the
    prover absorbed: <expand_powers(alpha) zero-extended, f_b> over the EVALUATION forms of the two committed blinding polynomials
    (whir_r1cs.rs:347-366, 371-380), the last challenge being the common root of two cubics.  That pins, against the reference's own bytes: the
    cubic message convention (row S3's wire form), the weight layout w[4 i + k] = alpha_i^k (S6) and the weighted sums over evaluation tables
    (S5) -- rows the earlier rounds could only pin to definitions."""
    z, b = FX["zk_sumcheck"], FX["blinding"]
    hh, alpha, rho = [ints(h) for h in z["coefficients"]], ints(z["alpha"]), int(z["rho"], 16)
    cub = lambda c, x: (c[0] + x * (c[1] + x * (c[2] + x * c[3]))) % P
    assert len(alpha) == 20 and z["candidates_per_round"].count(3) == 10
    assert rho * int(b["sum_g"], 16) % P == (cub(hh[0], 0) + cub(hh[0], 1)) % P  # whir_r1cs.rs:126-129: saved = rho * sum_g
    for i in range(19):
        assert cub(hh[i], alpha[i]) == (cub(hh[i + 1], 0) + cub(hh[i + 1], 1)) % P, i
    # the statement over the blinding commitment, through the C oracle: weights, to_evals, dot
    w = [0] * 256
    for i, a in enumerate(alpha):
        w[4 * i : 4 * i + 4] = [1, a, a * a % P, a * a * a % P]
    wm = mont(oracle, w)
    for f, want in zip((b["f0"], b["f1"]), z["polynomial_sums"]):
        ev = oracle.to_evals(mont(oracle, ints(f)), 8)
        assert canon_ints(oracle, oracle.dot(wm, ev)[None])[0] == int(want, 16)
    # the same sums from this repository's restatement of the prover's host algebra: f0's first 80 evaluations ARE the blinding cubics
    ev0 = canon_ints(oracle, oracle.to_evals(mont(oracle, ints(b["f0"])), 8))
    assert sum(cub(ev0[4 * i : 4 * i + 4], alpha[i]) for i in range(20)) % P == int(z["polynomial_sums"][0], 16)


# ------------------------------------------------------------------------------------------------ GPU: HIP path
@pytest.fixture()
def ctx_v1(ctx):
    ctx.set_hash_version(1)
    yield ctx
    ctx.set_hash_version(2)


def test_the_references_own_proof_verifies_with_its_recovered_challenges():
    """Round 4: every challenge of the reference proof's first 47,228 bytes -- blinding commitment, the 20 zk-sumcheck rounds, the whole
    blinding WHIR proof -- was recovered from the proof by algebra (gen_fixture_whir.py; each is the UNIQUE solution of the verifier's own
    equations).  Replaying those bytes through oracle/verifier.py with the recovered values in place of the sponge (the sponge's IV is the
    one thing that cannot be recovered) runs every scalar relation of WhirR1CSVerifier::verify / RunZKWhir on the reference's real data:
    sumcheck chains (cubic and quadratic), OOD and STIR point conventions, the batched-leaf combination, the coefficient fold against the
    final polynomial, the deferred weight evaluation (MLE of expand_powers(alpha) at the reversed folding point), computeWPoly and the final
    WHIR check -- under Skyscraper v1 for the 45 Merkle openings.  Only the two checks that consume challenge BYTES (proof of work, STIR
    indices) are skipped.  A verifier restated from the Go circuit that accepts the reference's own proof, and rejects it when any
    recovered challenge or any proof byte is changed, is what pins rows E1, W1-W3, S3, S5, S6 and the verifier equations themselves."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import verifier as V
    from provekit_amd.scheme import WhirConfig, blinding_config_for

    b, z, w = FX["blinding"], FX["zk_sumcheck"], FX["blinding_whir"]
    prefix = bytes.fromhex(w["transcript_prefix_hex"])
    assert len(prefix) == 47228
    m, m_0 = 21, 20

    def vcfg(c):
        return V.WhirConfig(c.n_vars, c.batch_size, c.folding_factor, c.starting_log_inv_rate, c.num_queries, c.ood_samples, c.pow_bits,
                            c.final_queries, c.final_pow_bits, c.commitment_ood_samples, c.final_folding_pow_bits)

    cfg_w, cfg_b = vcfg(WhirConfig.for_size(m)), vcfg(blinding_config_for(m_0))
    h = lambda x: int(x, 16)
    challenges = ([None, None] + [None] * m_0  # the witness commitment's OOD point and batching randomness, r: not recoverable, not used here
                  + [h(b["ood_point"]), h(b["batching_randomness"]), h(z["rho"])] + ints(z["alpha"])
                  + [h(w["initial_combination_randomness"])] + ints(b["folding_randomness"])
                  + [h(w["round0_ood_point"]), h(w["round0_combination_randomness"])] + ints(w["round0_folding_randomness"]))
    alpha, rev = V.verify_solved_prefix(prefix, challenges, m, m_0, cfg_w, cfg_b, hash_version=1)
    assert alpha == ints(z["alpha"]) and rev == (ints(b["folding_randomness"]) + ints(w["round0_folding_randomness"]))[::-1]
    # it is a real check: one changed challenge, or one changed byte of a scalar or a hint, and the same walk fails
    for pos in (len([None, None] + [None] * m_0) + 2, len(challenges) - 1, len(challenges) - 6, len(challenges) - 5):
        bad = list(challenges)
        bad[pos] = (bad[pos] + 1) % P
        with pytest.raises(V.VerifyError):
            V.verify_solved_prefix(prefix, bad, m, m_0, cfg_w, cfg_b, hash_version=1)
    for off in (200, 2784, 2900, 3264, 3400, 40000, 47200):
        bad = bytearray(prefix)
        bad[off] ^= 1
        with pytest.raises(V.VerifyError):
            V.verify_solved_prefix(bytes(bad), challenges, m, m_0, cfg_w, cfg_b, hash_version=1)
    with pytest.raises(V.VerifyError):  # ... and under the v2 hash the openings do not reach their roots
        V.verify_solved_prefix(prefix, challenges, m, m_0, cfg_w, cfg_b, hash_version=2)


@pytest.mark.gpu
def test_hip_blinding_commitment_is_the_references(ctx_v1, oracle):
    from provekit_amd import sumcheck as sc
    from provekit_amd.whir import commit_batch

    ctx, b = ctx_v1, FX["blinding"]
    f0, f1 = ints(b["f0"]), ints(b["f1"])
    d0, d1 = ctx.upload(mont(oracle, f0)), ctx.upload(mont(oracle, f1))
    c = commit_batch(ctx, [d0, d1], 8, 1, 4)
    assert int.from_bytes(c.root, "little") == int(b["root_T0"], 16)
    lv, _, _ = c.open(np.arange(32, dtype=np.uint64), canonical_leaves=True)
    assert [oracle.limbs_to_ints(l) for l in lv] == [ints(l) for l in b["leaves_T0"]]
    c.close()
    z = mont(oracle, [int(b["ood_point"], 16)])[0]
    for d, ans in zip((d0, d1), b["ood_answers"]):
        assert canon_ints(oracle, sc.eval_univariate(ctx, d, 256, z)[None])[0] == int(ans, 16)
    # batching + fold + round commitment
    beta, r = mont(oracle, [int(b["batching_randomness"], 16)])[0], mont(oracle, ints(b["folding_randomness"]))
    sc.axpy(ctx, d0, beta, d1, 256)  # d0 <- f0 + beta f1
    d_f = sc.fold_coeffs(ctx, d0, 8, r)
    assert canon_ints(oracle, ctx.download_fe(d_f, 16)) == ints(b["f_folded"])
    c1 = commit_batch(ctx, [d_f], 4, 4, 4)
    assert int.from_bytes(c1.root, "little") == int(b["root_T1"], 16)
    c1.close()
    # to_evals layout on the device
    d_e = ctx.upload(mont(oracle, f0))
    sc.to_evals(ctx, d_e, 8)
    ev = canon_ints(oracle, ctx.download_fe(d_e, 256))
    assert not any(ev[80:128]) and all(ev[:80]) and all(ev[128:])


@pytest.mark.gpu
def test_hip_witness_tail(ctx_v1, oracle):
    from provekit_amd import sumcheck as sc
    from provekit_amd.whir import commit_batch

    ctx, w = ctx_v1, FX["witness_tail"]
    d4 = ctx.upload(mont(oracle, ints(w["T4"]["f4"])))
    c = commit_batch(ctx, [d4], w["T4"]["n_vars"], w["T4"]["log_inv_rate"], 4)
    assert c.n_leaves == 1 << w["T4"]["height"]
    assert int.from_bytes(c.root, "little") == int(w["T4"]["root"], 16)
    lv, _, _ = c.open(np.array(w["T4"]["opened"], dtype=np.uint64), canonical_leaves=True)
    assert [oracle.limbs_to_ints(l) for l in lv] == [ints(l) for l in w["T4"]["leaves"]]
    c.close()
    d5 = sc.fold_coeffs(ctx, d4, 5, mont(oracle, ints(w["final"]["folding_randomness"])))
    assert canon_ints(oracle, ctx.download_fe(d5, 2)) == ints(w["final"]["final_coefficients"])
    r3 = mont(oracle, ints(w["T3"]["folding_randomness"]))
    w15 = oracle.root_of_unity(15)
    # the 11 opened leaves, zero-padded to 16, are folded by ONE coefficient fold of a 2^8 vector (leaf k = block k)
    n3 = len(w["T3"]["leaves"])
    flat = [v for l in w["T3"]["leaves"] for v in ints(l)] + [0] * (16 * (16 - n3))
    folded = ctx.download_fe(sc.fold_coeffs(ctx, ctx.upload(mont(oracle, flat)), 8, r3), 16)
    for k, i in enumerate(w["T3"]["opened"]):
        x = np.empty(4, dtype=np.uint64)
        oracle.L.pko_fe_pow(oracle._p(w15), int(i), oracle._p(x))
        assert np.array_equal(folded[k], sc.eval_univariate(ctx, d4, 32, x))


@pytest.mark.gpu
def test_hip_blinding_statement_sums_are_the_references(ctx_v1, oracle):
    """the two "Polynomial sums" of the reference's proof through the C ABI: to_evals of the two committed blinding polynomials (pk_to_evals)
    and their weighted sums against expand_powers(alpha) (pk_dot), alpha = the reference's own challenges as recovered from its proof"""
    from provekit_amd import sumcheck as sc

    ctx, z, b = ctx_v1, FX["zk_sumcheck"], FX["blinding"]
    alpha = ints(z["alpha"])
    w = [0] * 256
    for i, a in enumerate(alpha):
        w[4 * i : 4 * i + 4] = [1, a, a * a % P, a * a * a % P]
    d_w = ctx.upload(mont(oracle, w))
    for f, want in zip((b["f0"], b["f1"]), z["polynomial_sums"]):
        d = ctx.upload(mont(oracle, ints(f)))
        sc.to_evals(ctx, d, 8)
        assert canon_ints(oracle, sc.weighted_sum(ctx, d_w, d, 256)[None])[0] == int(want, 16)
        assert canon_ints(oracle, sc.weighted_sum(ctx, d_w, d, 80)[None])[0] == int(want, 16)  # the weight is zero beyond 4 m_0 entries


# ------------------------------------------------------------------------------------------------ the blinding WHIR proof, REPLAYED
def _blinding_whir_inputs():
    b, z, w = FX["blinding"], FX["zk_sumcheck"], FX["blinding_whir"]
    hx = lambda x: int(x, 16)
    alpha = ints(z["alpha"])
    table = [0] * 256
    for i, a in enumerate(alpha):
        table[4 * i : 4 * i + 4] = [1, a, a * a % P, a * a * a % P]
    return dict(f0=ints(b["f0"]), f1=ints(b["f1"]), beta=hx(b["batching_randomness"]), z0=hx(b["ood_point"]), r03=ints(b["folding_randomness"]),
                gamma0=hx(w["initial_combination_randomness"]), z1=hx(w["round0_ood_point"]), gamma1=hx(w["round0_combination_randomness"]),
                r47=ints(w["round0_folding_randomness"]), table=table, alpha=alpha, prefix=bytes.fromhex(w["transcript_prefix_hex"]))


def _frame_hint(payload: bytes) -> bytes:
    return len(payload).to_bytes(4, "little") + payload


def _stir_answers_payload(leaves_canon) -> bytes:  # Vec<Vec<F>>, ark-serialize uncompressed (common.go:36-61)
    out = len(leaves_canon).to_bytes(8, "little")
    for leaf in leaves_canon:
        out += len(leaf).to_bytes(8, "little") + b"".join(int(x).to_bytes(32, "little") for x in leaf)
    return out


def _expected_sections(prefix):
    """the reference's bytes of the blinding WHIR proof, split where the two 8-byte nonces sit (a nonce is found by search on a challenge this
    replay does not have): [2848, 3296) | nonce | [3304, n1) | nonce | [n1 + 8, 47228)"""
    import struct

    off = 3304
    for _ in range(2):
        off += 4 + struct.unpack_from("<I", prefix, off)[0]
    n1 = off + 384 + 32  # the round's sumcheck, the final coefficient
    return prefix[2848:3296], prefix[3304:n1], prefix[n1 + 8 : 47228]


def test_oracle_replays_the_references_blinding_whir_bytes(oracle):
    """With the polynomials recovered from the blinding commitment and every challenge recovered from the proof, the WHIR PROVER's side can
    be replayed: the C oracle regenerates the reference proof's bytes 2848..47228 -- 24 sumcheck evaluations, the round commitment's root, the
    OOD answer, the final coefficient, both `stir_answers` hints and the deferred weight evaluation -- byte for byte (the two `merkle_proof`
    hints are compared in tests/test_host_only.py and the GPU half; the two nonces are found by search on a challenge nobody has)."""
    import pyref as pr

    I = _blinding_whir_inputs()
    m = lambda xs: mont(oracle, xs)
    one = m([1])[0]
    sc32 = lambda x: int(x).to_bytes(32, "little")
    c = oracle.vec_axpy(m(I["f0"]), m([I["beta"]])[0], m(I["f1"]))  # f0 + beta f1 (mtUtilities.go:98-114)
    p = oracle.to_evals(c, 8)
    w = oracle.eq_accumulate_point(np.zeros((256, 4), np.uint64), 8, m(pr.expand_from_univariate(I["z0"], 8)), one)
    w = oracle.vec_axpy(w, m([I["gamma0"]])[0], m(I["table"]))

    def group(p, w, rs):
        out_bytes, fold = b"", None
        for r in rs:
            ev, p, w = oracle.sumcheck_quadratic_round(p, w, fold)
            if fold is not None:
                p, w = p[: len(p) // 2], w[: len(w) // 2]
            out_bytes += b"".join(sc32(x) for x in canon_ints(oracle, ev))
            fold = m([r])[0]
        p, w = p.copy(), w.copy()
        oracle.L.pko_fold_pairs(oracle._p(p), len(p), oracle._p(fold))
        oracle.L.pko_fold_pairs(oracle._p(w), len(w), oracle._p(fold))
        return out_bytes, p[: len(p) // 2], w[: len(w) // 2]

    sec_a, p, w = group(p, w, I["r03"])
    fprime = oracle.fold_coeffs(c, 8, m(I["r03"]))
    leaves1 = oracle.rs_encode(fprime, 1, 4, 4, 4)
    root1 = oracle.limbs_to_ints(oracle.merkle_commit(leaves1, version=1)[1:2])[0]
    ans1 = canon_ints(oracle, oracle.eval_univariate(fprime, m([I["z1"]])[0])[None])[0]
    sec_a += sc32(root1) + sc32(ans1)
    leaves0 = oracle.rs_encode(m(I["f0"] + I["f1"]), 2, 8, 1, 4)
    sec_b = _frame_hint(_stir_answers_payload([canon_ints(oracle, l) for l in leaves0]))
    exp_gen = pow(pr.root_of_unity(9), 16, P)
    g = 1
    for pt in [I["z1"]] + [pow(exp_gen, i, P) for i in range(32)]:
        w = oracle.eq_accumulate_point(w, 4, m(pr.expand_from_univariate(pt, 4)), m([g])[0])
        g = g * I["gamma1"] % P
    sc_bytes, p, w = group(p, w, I["r47"])
    final = canon_ints(oracle, oracle.fold_coeffs(fprime, 4, m(I["r47"])))
    assert len(final) == 1 and canon_ints(oracle, p) == final  # the sumcheck's polynomial IS the final polynomial
    rev = (I["r03"] + I["r47"])[::-1]
    deferred = canon_ints(oracle, oracle.dot(m(I["table"]), oracle.eq_table(m(rev)))[None])[0]
    want_a, want_b, want_c = _expected_sections(I["prefix"])
    assert sec_a == want_a
    assert want_b.startswith(sec_b) and want_b.endswith(sc_bytes + sc32(final[0]))
    idx1 = [0, 1, 2, 3, 6, 7, 8, 9, 10, 11, 12, 13, 14]  # the 13 leaves of the round tree the reference opened
    sec_c = _frame_hint(_stir_answers_payload([canon_ints(oracle, leaves1[i]) for i in idx1]))
    assert want_c.startswith(sec_c) and want_c.endswith(_frame_hint((1).to_bytes(8, "little") + sc32(deferred)))


@pytest.mark.gpu
def test_hip_replays_the_references_blinding_whir_bytes(ctx_v1, oracle):
    """the same replay through the C ABI, Merkle hints included: pk_to_evals, pk_eq_accumulate, pk_fe_axpy, pk_sumcheck_quadratic_round,
    pk_fold_pairs, pk_fold_coeffs, pk_commit, pk_tree_open, pk_multipath_serialize, pk_eval_univariate, pk_eq_table and pk_dot regenerate
    bytes 2848..47228 of the reference prover's own proof -- everything but the two nonces -- bit for bit (Skyscraper v1)."""
    import pyref as pr
    from provekit_amd import sumcheck as sc
    from provekit_amd.whir import commit_batch, multipath_serialize

    ctx, I = ctx_v1, _blinding_whir_inputs()
    m = lambda xs: mont(oracle, xs)
    sc32 = lambda x: int(x).to_bytes(32, "little")
    d_f0, d_f1 = ctx.upload(m(I["f0"])), ctx.upload(m(I["f1"]))
    com0 = commit_batch(ctx, [d_f0, d_f1], 8, 1, 4)
    d_c = ctx.upload(m(I["f0"]))
    sc.axpy(ctx, d_c, m([I["beta"]])[0], d_f1, 256)
    d_p = ctx.upload(ctx.download_fe(d_c, 256))
    sc.to_evals(ctx, d_p, 8)
    d_w = ctx.alloc_fe(256)
    sc.eq_accumulate(ctx, d_w, 8, m(pr.expand_from_univariate(I["z0"], 8)), m([1]), overwrite=True)
    sc.axpy(ctx, d_w, m([I["gamma0"]])[0], ctx.upload(m(I["table"])), 256)

    def group(d_p, d_w, length, rs):
        out, fold = b"", None
        for r in rs:
            if fold is None:
                ev = sc.sumcheck_quadratic_round(ctx, d_p, d_w, length)
            else:
                d_p2, d_w2 = ctx.alloc_fe(length // 2), ctx.alloc_fe(length // 2)
                ev = sc.sumcheck_quadratic_round(ctx, d_p, d_w, length, fold, d_p2, d_w2)
                d_p, d_w, length = d_p2, d_w2, length // 2
            out += b"".join(sc32(x) for x in canon_ints(oracle, ev))
            fold = m([r])[0]
        d_p2, d_w2 = ctx.alloc_fe(max(length // 2, 1)), ctx.alloc_fe(max(length // 2, 1))
        sc.fold_pairs(ctx, d_p, length, fold, d_p2)
        sc.fold_pairs(ctx, d_w, length, fold, d_w2)
        return out, d_p2, d_w2, length // 2

    got, d_p, d_w, length = group(d_p, d_w, 256, I["r03"])
    d_fp = sc.fold_coeffs(ctx, d_c, 8, m(I["r03"]))
    com1 = commit_batch(ctx, [d_fp], 4, 4, 4)
    got += com1.root + sc32(canon_ints(oracle, sc.eval_univariate(ctx, d_fp, 16, m([I["z1"]])[0])[None])[0])
    want_a, want_b, want_c = _expected_sections(I["prefix"])
    assert got == want_a
    lv, sib, paths = com0.open(np.arange(32, dtype=np.uint64), canonical_leaves=True)
    got_b = _frame_hint(_stir_answers_payload([oracle.limbs_to_ints(l) for l in lv])) + _frame_hint(multipath_serialize(list(range(32)), sib, paths))
    exp_gen = pow(pr.root_of_unity(9), 16, P)
    pts = [I["z1"]] + [pow(exp_gen, i, P) for i in range(32)]
    scales = [pow(I["gamma1"], j, P) for j in range(33)]
    sc.eq_accumulate(ctx, d_w, 4, np.concatenate([m(pr.expand_from_univariate(pt, 4)) for pt in pts]), m(scales), overwrite=False)
    evs, d_p, d_w, length = group(d_p, d_w, 16, I["r47"])
    d_fin = sc.fold_coeffs(ctx, d_fp, 4, m(I["r47"]))
    final = canon_ints(oracle, ctx.download_fe(d_fin, 1))
    assert canon_ints(oracle, ctx.download_fe(d_p, 1)) == final
    got_b += evs + sc32(final[0])
    assert got_b == want_b
    idx1 = [0, 1, 2, 3, 6, 7, 8, 9, 10, 11, 12, 13, 14]
    lv, sib, paths = com1.open(np.array(idx1, dtype=np.uint64), canonical_leaves=True)
    rev = (I["r03"] + I["r47"])[::-1]
    d_eq = sc.calculate_evaluations_over_boolean_hypercube_for_eq(ctx, m(rev))
    deferred = canon_ints(oracle, sc.weighted_sum(ctx, ctx.upload(m(I["table"])), d_eq, 256)[None])[0]
    got_c = (_frame_hint(_stir_answers_payload([oracle.limbs_to_ints(l) for l in lv])) + _frame_hint(multipath_serialize(idx1, sib, paths))
             + _frame_hint((1).to_bytes(8, "little") + sc32(deferred)))
    assert got_c == want_c
    com0.close()
    com1.close()
