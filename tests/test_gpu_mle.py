"""GPU parity: MLE / sumcheck kernels vs the oracle + golden vectors."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VEC = json.load(open(os.path.join(G, "vectors.json")))


def M(oracle, hexes):
    return oracle.to_mont(oracle.hex_to_limbs(hexes))


@pytest.mark.parametrize("n_vars", [0, 1, 3, 10, 11, 12, 15, 20, 21])
def test_to_coeffs_and_back(ctx, oracle, n_vars):
    from provekit_amd import sumcheck as sc
    from provekit_amd.field import random_field

    ev = random_field(1 << n_vars, 5 + n_vars)
    d = ctx.upload(ev)
    sc.to_coeffs(ctx, d, n_vars)
    got = ctx.download_fe(d, 1 << n_vars)
    if n_vars <= 15:
        assert np.array_equal(got, oracle.to_coeffs(ev, n_vars))
    else:  # spot-check through the definition c[S] = sum_{T subset S} (-1)^{|S|-|T|} f[T] on sparse masks
        for idx in (0, 1, 1 << (n_vars - 1), (1 << (n_vars - 1)) | 5, 0b1011 << 7):
            bits = [b for b in range(n_vars) if idx >> b & 1]
            acc = 0
            for m in range(1 << len(bits)):
                t = sum(1 << bits[j] for j in range(len(bits)) if m >> j & 1)
                sign = -1 if (len(bits) - bin(m).count("1")) & 1 else 1
                acc += sign * oracle.limbs_to_ints(ev[t])[0]
            assert oracle.limbs_to_ints(got[idx])[0] == acc % oracle.P
    sc.to_evals(ctx, d, n_vars)
    assert np.array_equal(ctx.download_fe(d, 1 << n_vars), ev)  # round trip


def test_to_coeffs_golden(ctx, oracle):
    from provekit_amd import sumcheck as sc

    v = VEC["to_coeffs_n3"]
    d = ctx.upload(M(oracle, v["evals"]))
    sc.to_coeffs(ctx, d, 3)
    assert np.array_equal(ctx.download_fe(d, 8), M(oracle, v["coeffs"]))


@pytest.mark.parametrize("m", [0, 1, 2, 3, 7, 12, 17, 20])
def test_eq_table(ctx, oracle, m):
    from provekit_amd import sumcheck as sc
    from provekit_amd.field import random_field

    r = random_field(max(m, 1), 90 + m)[:m]
    d = sc.calculate_evaluations_over_boolean_hypercube_for_eq(ctx, r)
    got = ctx.download_fe(d, 1 << m)
    if m <= 17:
        assert np.array_equal(got, oracle.eq_table(r))
    else:
        full = oracle.eq_table(r)
        assert np.array_equal(got, full)


def test_eq_table_golden(ctx, oracle):
    from provekit_amd import sumcheck as sc

    v = VEC["eq_table_m3"]
    d = sc.calculate_evaluations_over_boolean_hypercube_for_eq(ctx, M(oracle, v["r"]))
    assert np.array_equal(ctx.download_fe(d, 8), M(oracle, v["table"]))


def test_eq_accumulate_many_points(ctx, oracle):
    from provekit_amd import sumcheck as sc
    from provekit_amd.field import random_field

    n, q = 9, 13
    w0 = random_field(1 << n, 3)
    zs = random_field(q, 4)
    scales = random_field(q, 5)
    # points = ExpandFromUnivariate(z, n) (utilities.go:182-190), as whir's STIR/OOD constraints are
    pts = np.empty((q, n, 4), dtype=np.uint64)
    exp = w0.copy()
    for t in range(q):
        acc = zs[t].copy()
        for i in range(n):
            pts[t, n - 1 - i] = acc
            acc = oracle.binop("pko_fe_mul", acc, acc)[0]
        exp = oracle.eq_accumulate_univariate(exp, n, zs[t], scales[t])
    d = ctx.upload(w0)
    sc.eq_accumulate(ctx, d, n, pts, scales)
    assert np.array_equal(ctx.download_fe(d, 1 << n), exp)
    # golden: eq table of an expanded univariate point
    g = VEC["eq_univariate_n4"]
    z = M(oracle, [g["z"]])[0]
    p4 = np.empty((1, 4, 4), dtype=np.uint64)
    acc = z.copy()
    for i in range(4):
        p4[0, 3 - i] = acc
        acc = oracle.binop("pko_fe_mul", acc, acc)[0]
    d4 = ctx.alloc_fe(16)
    sc.eq_accumulate(ctx, d4, 4, p4, oracle.to_mont(oracle.ints_to_limbs([1])), overwrite=True)
    assert np.array_equal(ctx.download_fe(d4, 16), M(oracle, g["table"]))


@pytest.mark.parametrize("log_len", [1, 2, 3, 8, 13, 16, 18])
def test_sumcheck_cubic_rounds_vs_oracle(ctx, oracle, log_len):
    """run the whole m_0-round loop of run_zk_sumcheck_prover's hot part (whir_r1cs.rs:280-345)"""
    from provekit_amd import sumcheck as sc
    from provekit_amd.field import random_field

    n = 1 << log_len
    arrs = [random_field(n, 20 + k + log_len) for k in range(4)]
    d = [ctx.upload(a) for a in arrs]
    alphas = random_field(log_len, 99)
    cur, length, fold = [a.copy() for a in arrs], n, None
    for rnd in range(log_len):
        got = sc.sumcheck_fold_map_reduce(ctx, *d, length, fold)
        exp, *cur = oracle.sumcheck_cubic_round(*[c[:length] for c in cur], fold)
        assert np.array_equal(got, exp), (log_len, rnd)
        if fold is not None:
            length //= 2
            for k in range(4):  # folded prefix must match the reference's in-place result
                assert np.array_equal(ctx.download_fe(d[k], length), cur[k][:length])
        fold = alphas[rnd]


def test_sumcheck_cubic_golden_and_errors(ctx, oracle):
    from provekit_amd import ProveKitHipError
    from provekit_amd import sumcheck as sc

    v = VEC["sumcheck_cubic"]
    d = [ctx.upload(M(oracle, v[k])) for k in ("a", "b", "c", "eq")]
    assert np.array_equal(sc.sumcheck_fold_map_reduce(ctx, *d, 16), M(oracle, v["round0"]))
    alpha = M(oracle, [v["alpha"]])[0]
    assert np.array_equal(sc.sumcheck_fold_map_reduce(ctx, *d, 16, alpha), M(oracle, v["round1"]))
    assert np.array_equal(ctx.download_fe(d[0], 8), M(oracle, v["a_folded"]))
    with pytest.raises(ProveKitHipError):  # sumcheck.rs:22-23 asserts
        sc.sumcheck_fold_map_reduce(ctx, *d, 12)
    with pytest.raises(ProveKitHipError):
        sc.sumcheck_fold_map_reduce(ctx, *d, 1)
    with pytest.raises(ProveKitHipError):  # sumcheck.rs:27
        sc.sumcheck_fold_map_reduce(ctx, *d, 2, alpha)


@pytest.mark.parametrize("log_len", [1, 2, 5, 12, 17])
def test_sumcheck_quadratic_rounds_vs_oracle(ctx, oracle, log_len):
    from provekit_amd import sumcheck as sc
    from provekit_amd.field import random_field

    n = 1 << log_len
    f, w = random_field(n, 31 + log_len), random_field(n, 32 + log_len)
    bufs = [[ctx.upload(f), ctx.alloc_fe(n)], [ctx.upload(w), ctx.alloc_fe(n)]]
    rs = random_field(log_len, 77)
    cf, cw, length, fold, cur = f, w, n, None, 0
    for rnd in range(log_len):
        if fold is None:
            got = sc.sumcheck_quadratic_round(ctx, bufs[0][cur], bufs[1][cur], length)
        else:
            got = sc.sumcheck_quadratic_round(ctx, bufs[0][cur], bufs[1][cur], length, fold, bufs[0][1 - cur], bufs[1][1 - cur])
            cur, length = 1 - cur, length // 2
        exp, cf, cw = oracle.sumcheck_quadratic_round(cf[: (length * 2 if fold is not None else length)], cw[: (length * 2 if fold is not None else length)], fold)
        assert np.array_equal(got, exp), (log_len, rnd)
        if fold is not None:
            assert np.array_equal(ctx.download_fe(bufs[0][cur], length), cf[:length])
        fold = rs[rnd]
    # last fold down to a single element
    out = ctx.alloc_fe(1)
    sc.fold_pairs(ctx, bufs[0][cur], 2, fold, out)
    a = cf[:2]
    e = oracle.binop("pko_fe_add", a[0], oracle.binop("pko_fe_mul", fold, oracle.binop("pko_fe_sub", a[1], a[0])[0])[0])[0]
    assert np.array_equal(ctx.download_fe(out, 1)[0], e)


def test_sumcheck_quadratic_golden(ctx, oracle):
    from provekit_amd import sumcheck as sc

    v = VEC["sumcheck_quadratic"]
    df, dw = ctx.upload(M(oracle, v["f"])), ctx.upload(M(oracle, v["w"]))
    assert np.array_equal(sc.sumcheck_quadratic_round(ctx, df, dw, 16), M(oracle, v["round0"]))
    fo, wo = ctx.alloc_fe(8), ctx.alloc_fe(8)
    r = M(oracle, [v["r"]])[0]
    assert np.array_equal(sc.sumcheck_quadratic_round(ctx, df, dw, 16, r, fo, wo), M(oracle, v["round1"]))
    assert np.array_equal(ctx.download_fe(fo, 8), M(oracle, v["f_folded"]))


@pytest.mark.parametrize("n", [0, 1, 5, 1000, 70000, 1 << 18])
def test_dot_and_univariate(ctx, oracle, n):
    from provekit_amd import sumcheck as sc
    from provekit_amd.field import random_field

    w, f = random_field(max(n, 1), 1 + n)[:n], random_field(max(n, 1), 2 + n)[:n]
    dw, df = ctx.upload(w if n else np.zeros((1, 4), np.uint64)), ctx.upload(f if n else np.zeros((1, 4), np.uint64))
    assert np.array_equal(sc.weighted_sum(ctx, dw, df, n), oracle.dot(w, f) if n else np.zeros(4, np.uint64))
    z = random_field(1, 1234)[0]
    exp = oracle.eval_univariate(f, z) if n else np.zeros(4, np.uint64)
    assert np.array_equal(sc.eval_univariate(ctx, df, n, z), exp)


def test_fold_and_univariate_golden(ctx, oracle):
    from provekit_amd import sumcheck as sc

    v = VEC["fold_coeffs"]
    c = M(oracle, v["coeffs"])
    d = ctx.upload(c)
    out = sc.fold_coeffs(ctx, d, 6, M(oracle, v["r"]))
    assert np.array_equal(ctx.download_fe(out, 4), M(oracle, v["out"]))
    e = VEC["eval_univariate"]
    assert np.array_equal(sc.eval_univariate(ctx, d, 64, M(oracle, [e["z"]])[0]), M(oracle, [e["out"]])[0])


@pytest.mark.parametrize("n_vars,k", [(4, 4), (5, 4), (10, 4), (17, 4), (19, 4), (9, 1), (6, 0), (8, 8)])
def test_fold_coeffs_vs_oracle(ctx, oracle, n_vars, k):
    from provekit_amd import sumcheck as sc
    from provekit_amd.field import random_field

    c = random_field(1 << n_vars, n_vars + k)
    r = random_field(max(k, 1), 3)[:k]
    out = sc.fold_coeffs(ctx, ctx.upload(c), n_vars, r)
    assert np.array_equal(ctx.download_fe(out, 1 << (n_vars - k)), oracle.fold_coeffs(c, n_vars, r))


def test_fold_commutes_with_evaluation(ctx, oracle):
    """size-independent property at BASELINE size (2^21 coefficients): folding by (r0..r3) then evaluating the
    folded univariate at y equals sum_j prod r^bits(j) * f_j(y) -- the identity the verifier's computeFold uses."""
    from provekit_amd import sumcheck as sc
    from provekit_amd.field import random_field

    n = 21
    c = random_field(1 << n, 11)
    r = random_field(4, 12)
    y = random_field(1, 13)[0]
    d = ctx.upload(c)
    folded = sc.fold_coeffs(ctx, d, n, r)
    lhs = sc.eval_univariate(ctx, folded, 1 << (n - 4), y)
    # rhs via the axpy/dot-free route: evaluate each of the 16 strided sub-polynomials on the host oracle (2^17 each)
    acc = np.zeros(4, dtype=np.uint64)
    for j in range(16):
        wj = oracle.to_mont(oracle.ints_to_limbs([1]))[0]
        for b in range(4):
            if j >> b & 1:
                wj = oracle.binop("pko_fe_mul", wj, r[b])[0]
        fj = oracle.eval_univariate(np.ascontiguousarray(c[j::16]), y)
        acc = oracle.binop("pko_fe_add", acc, oracle.binop("pko_fe_mul", wj, fj)[0])[0]
    assert np.array_equal(lhs, acc)


def test_axpy(ctx, oracle):
    from provekit_amd import sumcheck as sc
    from provekit_amd.field import random_field

    n = 5001
    y, x, beta = random_field(n, 1), random_field(n, 2), random_field(1, 3)[0]
    dy = ctx.upload(y)
    sc.axpy(ctx, dy, beta, ctx.upload(x), n)
    exp = oracle.binop("pko_fe_add", y, oracle.binop("pko_fe_mul", x, np.tile(beta, (n, 1))))
    assert np.array_equal(ctx.download_fe(dy, n), exp)


@pytest.mark.parametrize("n_vars", [0, 5, 12, 21])
def test_to_coeffs_out_of_place(ctx, oracle, n_vars):
    from provekit_amd._lib import lib
    from provekit_amd.field import random_field

    ev = random_field(1 << n_vars, 333 + n_vars)
    src, dst, ref = ctx.upload(ev), ctx.alloc_fe(1 << n_vars), ctx.upload(ev)
    ctx._check(lib.pk_to_coeffs_into(ctx.handle, src.ptr, dst.ptr, n_vars))
    ctx._check(lib.pk_to_coeffs(ctx.handle, ref.ptr, n_vars))
    assert np.array_equal(ctx.download_fe(dst, 1 << n_vars), ctx.download_fe(ref, 1 << n_vars))
    assert np.array_equal(ctx.download_fe(src, 1 << n_vars), ev)  # source untouched
    back = ctx.alloc_fe(1 << n_vars)
    ctx._check(lib.pk_to_evals_into(ctx.handle, dst.ptr, back.ptr, n_vars))
    assert np.array_equal(ctx.download_fe(back, 1 << n_vars), ev)


@pytest.mark.parametrize("n_vars,q", [(1, 1), (1, 5), (2, 3), (5, 1), (6, 4), (7, 5), (9, 9), (10, 13)])
def test_eq_accumulate_group_boundaries(ctx, oracle, n_vars, q):
    """eq_accumulate sums its q products per element with one Montgomery reduction per group of four (fe29.hpp dot29): every
    residue of q mod 4, odd and even splits of the variables, accumulate and overwrite, against the oracle point by point"""
    import ctypes as C

    from provekit_amd._lib import lib
    from provekit_amd.field import random_field

    pts = random_field(q * n_vars, 40 + n_vars + q).reshape(q, n_vars, 4)
    scales = random_field(q, 50 + q)
    w0 = random_field(1 << n_vars, 60 + n_vars)
    for overwrite in (1, 0):
        d_w = ctx.upload(w0)
        ctx._check(lib.pk_eq_accumulate(ctx.handle, d_w.ptr, n_vars, pts.ctypes.data, scales.ctypes.data, q, overwrite))
        want = np.zeros_like(w0) if overwrite else w0.copy()
        for t in range(q):
            want = oracle.eq_accumulate_point(want, n_vars, pts[t], scales[t])
        assert np.array_equal(ctx.download_fe(d_w, 1 << n_vars), want), (n_vars, q, overwrite)
