"""GPU parity: NTT / interleaved Reed-Solomon encode vs the oracle and the
definition-derived golden vectors."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VEC = json.load(open(os.path.join(G, "vectors.json")))


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 9, 10, 11, 13, 18, 19, 20])
def test_ntt_vs_oracle(ctx, oracle, log_n):
    from provekit_amd.field import random_field
    from provekit_amd.rs import ntt

    ncols = 3 if log_n < 16 else 1
    x = random_field(ncols << log_n, 100 + log_n).reshape(ncols, 1 << log_n, 4)
    got = ntt(x, ctx=ctx)
    for c in range(ncols):
        assert np.array_equal(got[c], oracle.ntt(x[c], log_n)), (log_n, c)


def test_ntt_column_counts(ctx, oracle):
    from provekit_amd.field import random_field
    from provekit_amd.rs import ntt

    for ncols in (1, 2, 4, 5, 8, 9):
        for log_n in (4, 12):
            x = random_field(ncols << log_n, 7 * ncols + log_n).reshape(ncols, 1 << log_n, 4)
            got = ntt(x, ctx=ctx)
            for c in range(ncols):
                assert np.array_equal(got[c], oracle.ntt(x[c], log_n))


@pytest.mark.parametrize("name", ["rs_b2_n6_r1_f4", "rs_b1_n5_r2_f2", "rs_b1_n4_r3_f4"])
def test_rs_encode_golden_by_definition(ctx, oracle, name):
    """vectors computed by naive evaluation of the definition (tests/golden/gen_golden.py)"""
    from provekit_amd.rs import rs_encode

    v = VEC[name]
    coeffs = np.stack([oracle.to_mont(oracle.hex_to_limbs(p)) for p in v["coeffs"]])
    got = rs_encode(coeffs, v["n_vars"], v["log_inv_rate"], v["fold"], ctx=ctx)
    exp = np.stack([oracle.to_mont(oracle.hex_to_limbs(l)) for l in v["leaves"]])
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("batch,n_vars,rho,fold", [(2, 8, 1, 4), (2, 12, 1, 4), (1, 13, 4, 4), (2, 14, 1, 4), (1, 9, 7, 4), (1, 4, 1, 4), (2, 16, 1, 4), (1, 20, 1, 4)])
def test_rs_encode_vs_oracle(ctx, oracle, batch, n_vars, rho, fold):
    from provekit_amd.field import random_field
    from provekit_amd.rs import rs_encode

    coeffs = random_field(batch << n_vars, n_vars * 31 + rho).reshape(batch, 1 << n_vars, 4)
    got = rs_encode(coeffs, n_vars, rho, fold, ctx=ctx)
    exp = oracle.rs_encode(coeffs.reshape(-1, 4), batch, n_vars, rho, fold)
    assert np.array_equal(got, exp)


def test_rs_encode_full_size_properties(ctx, oracle):
    """BASELINE config-2 size (batch 2, n = 21, rate 1/2, fold 16): linearity of the code and
    spot-checks of single codeword symbols against direct evaluation f_j(w^i) by the oracle."""
    from provekit_amd._lib import lib
    from provekit_amd.field import random_field
    from provekit_amd.rs import rs_encode_device

    n, rho, fold = 21, 1, 4
    rows, w = 1 << (n + rho - fold), 32
    f = random_field(1 << n, 1)
    g = random_field(1 << n, 2)
    s = oracle.binop("pko_fe_add", f[:4096], g[:4096])  # only used for the linearity spot rows below
    df, dg = ctx.upload(f), ctx.upload(g)
    dsum = ctx.alloc_fe(1 << n)
    ctx._check(lib.pk_fe_add(ctx.handle, df.ptr, dg.ptr, dsum.ptr, 1 << n))
    leaves = ctx.alloc_fe(rows * w)
    leaves2 = ctx.alloc_fe(rows * 16)
    scratch = ctx.alloc_fe(2 * rows * w)
    rs_encode_device(ctx, [df.ptr, dg.ptr], n, rho, fold, leaves.ptr, scratch.ptr)
    rs_encode_device(ctx, [dsum.ptr], n, rho, fold, leaves2.ptr, scratch.ptr)
    M = ctx.download(leaves, (w, rows, 4))
    S = ctx.download(leaves2, (16, rows, 4))
    # linearity: encode(f+g) == encode(f) + encode(g) on a sample of rows, all 16 columns
    pick = np.random.default_rng(0).integers(0, rows, size=64)
    for j in range(16):
        assert np.array_equal(S[j, pick], oracle.binop("pko_fe_add", M[j, pick], M[16 + j, pick]))
    # direct evaluation: leaf_i[j] = f_j(w^i) with f_j(X) = sum_t f[16t+j] X^t  (Horner in the oracle)
    wroot = oracle.root_of_unity(n + rho - fold)
    for i, j in [(0, 0), (1, 3), (12345, 7), (rows - 1, 15)]:
        x = np.frombuffer(b"", dtype=np.uint64)
        pw = np.empty(4, dtype=np.uint64)
        oracle.L.pko_fe_pow(oracle._p(wroot), i, oracle._p(pw))
        assert np.array_equal(M[j, i], oracle.eval_univariate(np.ascontiguousarray(f[j::16]), pw))
        assert np.array_equal(M[16 + j, i], oracle.eval_univariate(np.ascontiguousarray(g[j::16]), pw))
