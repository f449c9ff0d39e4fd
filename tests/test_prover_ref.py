"""CPU: the oracle's own prover (oracle/prover_ref.py: WhirR1CSProver::prove restated on the C oracle's kernels) against the oracle's
verifier (oracle/verifier.py: WhirR1CSVerifier::verify + the Go WHIR verifier's equations) -- two restatements written against
different halves of the reference must agree -- and the C restatement of the proof's random draws against the Python one.
The GPU suite then asks pk_prove for the same bytes (tests/test_gpu_prove.py::test_transcript_equals_the_oracle_provers)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]


def small_instance(nc, n_in, seed):
    """the satisfiable synthetic R1CS of tests/test_gpu_prove.py, as CSR arrays for the oracle"""
    from test_gpu_prove import satisfiable_r1cs

    nw, z, coeffs, trips = satisfiable_r1cs(nc, n_in, seed)
    mats = []
    for rows, cols, vals in trips:
        rows = np.array(rows, dtype=np.int64)
        mats.append((np.searchsorted(rows, np.arange(nc)).astype(np.uint32), np.array(cols, dtype=np.uint32), np.array(vals, dtype=np.uint32)))
    return nw, z, coeffs, trips, mats


def configs(m, m_0, pow_bits, queries=None):
    """the reference's derived schedule through the library's host-only restatement (no GPU involved), flat test difficulty"""
    import verifier as V
    from provekit_amd.scheme import WhirConfig, blinding_config_for, create_io_pattern

    cw, cb = WhirConfig.for_size(m, pow_bits), blinding_config_for(m_0, pow_bits)
    if queries:
        cw.num_queries = queries[: cw.n_rounds]

    def vcfg(c):
        return V.WhirConfig(c.n_vars, c.batch_size, c.folding_factor, c.starting_log_inv_rate, list(c.num_queries), list(c.ood_samples), list(c.pow_bits),
                            c.final_queries, c.final_pow_bits, c.commitment_ood_samples, c.final_folding_pow_bits)

    return vcfg(cw), vcfg(cb), create_io_pattern(m_0, cw, cb)


@pytest.mark.parametrize("m,m_0,nc,n_in,pow_bits", [(9, 7, 100, 60, 5.0), (12, 9, 500, 700, 4.0), (16, 14, 12000, 3000, 6.0)])
def test_oracle_prover_is_accepted_by_the_oracle_verifier(oracle, m, m_0, nc, n_in, pow_bits):
    import prover_ref as PR
    import verifier as V

    nw, z, coeffs, trips, mats = small_instance(nc, n_in, 31)
    cfg_w, cfg_b, ds = configs(m, m_0, pow_bits, [20, 12, 9, 8])
    interner = oracle.to_mont(oracle.ints_to_limbs(coeffs))
    zm = oracle.to_mont(oracle.ints_to_limbs(z))
    stage = {}
    proof = PR.prove(ds, m, m_0, cfg_w, cfg_b, (nc, nw, mats, interner), zm, (7).to_bytes(32, "little"), stage)
    assert abs(sum(v for k, v in stage.items() if k != "total") - stage["total"]) < 0.05 * stage["total"] + 0.01
    vm = [(t[0], t[1], [coeffs[v] for v in t[2]]) for t in trips]
    assert V.verify(proof, ds, m, m_0, cfg_w, cfg_b, r1cs=(nc, nw, vm))
    assert PR.prove(ds, m, m_0, cfg_w, cfg_b, (nc, nw, mats, interner), zm, (7).to_bytes(32, "little")) == proof
    assert PR.prove(ds, m, m_0, cfg_w, cfg_b, (nc, nw, mats, interner), zm, (8).to_bytes(32, "little")) != proof  # another key, other masks
    # a witness that does not satisfy the system gives a proof the verifier refuses
    bad = list(z)
    bad[1 + n_in] = (bad[1 + n_in] + 1) % oracle.P  # the first constraint's output: C z no longer equals (A z) o (B z)
    p_bad = PR.prove(ds, m, m_0, cfg_w, cfg_b, (nc, nw, mats, interner), oracle.to_mont(oracle.ints_to_limbs(bad)), (7).to_bytes(32, "little"))
    with pytest.raises(V.VerifyError):
        V.verify(p_bad, ds, m, m_0, cfg_w, cfg_b)
    # tampering is refused
    t = bytearray(proof)
    t[len(t) // 2] ^= 1
    with pytest.raises(V.VerifyError):
        V.verify(bytes(t), ds, m, m_0, cfg_w, cfg_b)


def test_c_random_draw_equals_the_python_restatement(oracle):
    import prover_ref as PR
    from test_host_only import random_fe_py

    seed = bytes((7 * i + 3) & 0xFF for i in range(32))
    for stream, n in ((1, 301), (2, 64), (77, 1000)):
        got = oracle.limbs_to_ints(PR.random_fe(seed, stream, n))
        assert all(v < oracle.P for v in got)
        for i in list(range(60)) + [n - 1, n // 2]:
            assert got[i] == random_fe_py(seed, stream, i), (stream, i)


def test_parallel_horner_equals_the_definition(oracle):
    """pko_eval_univariate evaluates long polynomials in blocks: same value as the serial rule"""
    import prover_ref as PR
    from provekit_amd.field import random_field

    n = (1 << 16) + 12345
    c = random_field(n, 5)
    z = 0x1234567890ABCDEF1234567890ABCDEF % oracle.P
    ints = oracle.limbs_to_ints(oracle.from_mont(c))
    acc = 0
    for v in reversed(ints):
        acc = (acc * z + v) % oracle.P
    assert PR.eval_univariate(c, z) == acc
