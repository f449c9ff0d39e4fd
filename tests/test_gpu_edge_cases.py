"""GPU: the corners -- empty and minimal inputs, every tree height through the fused-level / top-kernel boundary, openings
of tiny trees, repeated proof-of-work launches (ticket re-arm), the mailbox under many points.  Bit-exact vs the oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_n", [0, 1, 2, 9, 10, 11, 12, 14, 15, 16])
def test_tree_every_height_and_openings(ctx, oracle, log_n):
    """heights 1..: n=1 (leaf digest is the root), top kernel only (<= 2^10 leaves), one or two fused-level launches"""
    from provekit_amd._lib import PK_LEAF_MAJOR
    from provekit_amd.field import random_field
    from provekit_amd.whir import tree_from_leaves

    n, width = 1 << log_n, 3
    leaves = random_field(n * width, 500 + log_n).reshape(n, width, 4)
    d = ctx.upload(leaves)
    t = tree_from_leaves(ctx, d, n, width, PK_LEAF_MAJOR)
    nodes = oracle.merkle_commit(leaves)
    assert t.root == nodes[1].tobytes()
    idx = np.unique(np.array([0, n - 1, n // 2, n // 3], dtype=np.uint64))
    lv, sib, paths = t.open(idx, canonical_leaves=False)
    assert np.array_equal(lv, leaves[idx.astype(np.int64)])
    for q, i in enumerate(idx):
        i = int(i)
        if log_n >= 1:
            assert np.array_equal(sib[q], nodes[(n + i) ^ 1])
        for dpt in range(1, log_n):
            assert np.array_equal(paths[q, dpt - 1], nodes[((n + i) >> (log_n - dpt)) ^ 1])
    t.close()


def test_empty_and_minimal_inputs(ctx, oracle):
    from provekit_amd import ProveKitHipError
    from provekit_amd import sumcheck as sc
    from provekit_amd._lib import lib
    from provekit_amd.field import random_field

    one = oracle.to_mont(oracle.ints_to_limbs([1]))[0]
    z = random_field(1, 1)[0]
    d = ctx.alloc_fe(4)
    out = np.zeros(8, dtype=np.uint64)
    # zero-length reductions are zero, not errors
    ctx._check(lib.pk_dot(ctx.handle, d.ptr, d.ptr, 0, out.ctypes.data))
    assert not out[:4].any()
    ctx._check(lib.pk_eval_univariate(ctx.handle, d.ptr, 0, z.ctypes.data, out.ctypes.data))
    assert not out[:4].any()
    ctx._check(lib.pk_compress_many(ctx.handle, d.ptr, d.ptr, 0))
    # a constant polynomial: eval = c, to_coeffs / to_evals on 0 variables are the identity, eq table of 0 variables is [1]
    c = random_field(1, 2)
    dc = ctx.upload(c)
    assert np.array_equal(sc.eval_univariate(ctx, dc, 1, z), c[0])
    sc.to_coeffs(ctx, dc, 0)
    sc.to_evals(ctx, dc, 0)
    assert np.array_equal(ctx.download_fe(dc, 1), c)
    t = sc.calculate_evaluations_over_boolean_hypercube_for_eq(ctx, np.zeros((0, 4), dtype=np.uint64))
    assert np.array_equal(ctx.download_fe(t, 1)[0], one)
    # fold every variable away (n_vars == k) == multivariate evaluation of the coefficient list
    cs, r = random_field(8, 3), random_field(3, 4)
    got = ctx.download_fe(sc.fold_coeffs(ctx, ctx.upload(cs), 3, r), 1)
    assert np.array_equal(got, oracle.fold_coeffs(cs, 3, r))
    # the smallest sumcheck rounds: one pair
    f, w = random_field(2, 5), random_field(2, 6)
    h = sc.sumcheck_quadratic_round(ctx, ctx.upload(f), ctx.upload(w), 2)
    exp, _, _ = oracle.sumcheck_quadratic_round(f, w)
    assert np.array_equal(h, exp)
    a, b, cc, e = (random_field(2, 7 + i) for i in range(4))
    h3 = sc.sumcheck_fold_map_reduce(ctx, *(ctx.upload(x) for x in (a, b, cc, e)), 2)
    exp3 = oracle.sumcheck_cubic_round(a, b, cc, e)[0]
    assert np.array_equal(h3, exp3)
    # sizes that are not powers of two are the reference's assertion failures (sumcheck.rs:22-23)
    with pytest.raises(ProveKitHipError):
        sc.sumcheck_quadratic_round(ctx, ctx.upload(random_field(6, 1)), ctx.upload(random_field(6, 2)), 6)


def test_smallest_encodes(ctx, oracle):
    """n_vars == fold (one coefficient per column: the NTT input is a constant) and rows in {1, 2, 4}"""
    from provekit_amd.field import random_field
    from provekit_amd.whir import commit_batch

    for n_vars, rate, fold, batch in [(2, 0, 2, 1), (2, 1, 2, 2), (4, 1, 4, 1), (4, 2, 4, 2), (5, 1, 4, 1), (3, 3, 1, 1)]:
        polys = [random_field(1 << n_vars, 900 + 10 * n_vars + b) for b in range(batch)]
        c = commit_batch(ctx, [ctx.upload(p) for p in polys], n_vars, rate, fold)
        leaves = oracle.rs_encode(np.concatenate(polys), batch, n_vars, rate, fold)
        assert c.n_leaves == leaves.shape[0] == 1 << (n_vars + rate - fold)
        assert c.root == oracle.merkle_commit(leaves)[1].tobytes()
        lv, _, _ = c.open(np.arange(c.n_leaves, dtype=np.uint64), canonical_leaves=False)
        assert np.array_equal(lv, leaves)
        c.close()


def test_pow_many_launches_rearm(ctx, oracle):
    """the device-side best/ticket words are re-armed by the kernel itself: 40 searches in a row, alternating difficulty,
    each must return the SMALLEST valid nonce (checked by exhaustive verification below it at low difficulty)"""
    from provekit_amd._lib import lib

    rng = np.random.default_rng(7)
    for it in range(40):
        ch = rng.integers(0, 256, size=32, dtype=np.uint8)
        ch[31] &= 0x0F
        bits = [3.0, 9.5, 13.0, 0.0][it % 4]
        nonce = C.c_uint64()
        ctx._check(lib.pk_pow_solve(ctx.handle, ch.ctypes.data, bits, C.byref(nonce)))
        ok = C.c_int()
        ctx._check(lib.pk_pow_check(ctx.handle, ch.ctypes.data, bits, nonce.value, C.byref(ok)))
        assert ok.value == 1
        if 0 < bits <= 9.5:  # the prover searches with the +0.01 bias (pow.rs:6,37): smallest nonce under THAT threshold
            chw = np.frombuffer(ch.tobytes(), dtype=np.uint64)
            assert all(not oracle.pow_verify(chw, bits + 0.01, k) for k in range(nonce.value))
            assert oracle.pow_verify(chw, bits + 0.01, nonce.value)


def test_eq_accumulate_mailbox_many_points(ctx, oracle):
    """300 points x 13 variables (125 KiB through the pinned mailbox), twice in a row without a synchronisation between,
    then small shapes: n_vars 1 and 2"""
    from provekit_amd import sumcheck as sc
    from provekit_amd.field import random_field

    n, q = 13, 300
    w0 = random_field(1 << n, 21)
    d = ctx.upload(w0)
    exp = w0
    for rep in range(2):
        zs, scales = random_field(q, 30 + rep), random_field(q, 40 + rep)
        pts = np.empty((q, n, 4), dtype=np.uint64)
        for t in range(q):
            acc = zs[t].copy()
            for i in range(n):
                pts[t, n - 1 - i] = acc
                acc = oracle.binop("pko_fe_mul", acc, acc)[0]
            exp = oracle.eq_accumulate_univariate(exp, n, zs[t], scales[t])
        sc.eq_accumulate(ctx, d, n, pts, scales)
        pts[:] = 0  # the call has copied the host arrays: clobbering them must not matter
    assert np.array_equal(ctx.download_fe(d, 1 << n), exp)
    for nv in (1, 2):
        pt, s = random_field(nv, 50 + nv), random_field(1, 60 + nv)
        dd = ctx.alloc_fe(1 << nv)
        sc.eq_accumulate(ctx, dd, nv, pt[None], s, overwrite=True)
        e = oracle.eq_accumulate_point(np.zeros((1 << nv, 4), dtype=np.uint64), nv, pt, s[0])
        assert np.array_equal(ctx.download_fe(dd, 1 << nv), e)


def test_pow_reference_solve_verify_case(ctx, oracle):
    """skyscraper/core/src/pow.rs:105-111 `test_solve_verify`: challenge = [u64::MAX; 4] (above p: the reduce_partial
    path), difficulty 0 and pi"""
    import math

    from provekit_amd._lib import lib

    ch = np.full(32, 0xFF, dtype=np.uint8)
    chw = np.frombuffer(ch.tobytes(), dtype=np.uint64)
    for bits in (0.0, math.pi):
        nonce = C.c_uint64(123)
        ctx._check(lib.pk_pow_solve(ctx.handle, ch.ctypes.data, bits, C.byref(nonce)))
        ok = C.c_int()
        ctx._check(lib.pk_pow_check(ctx.handle, ch.ctypes.data, bits, nonce.value, C.byref(ok)))
        assert ok.value == 1 and oracle.pow_verify(chw, bits, nonce.value)
        if bits == 0.0:
            assert nonce.value == 0  # pow.rs:34-36
        else:
            assert nonce.value == oracle.pow_solve(chw, bits)  # both return the smallest nonce under the biased threshold


def test_host_wait_environment_is_applied_before_the_stream_and_unknown_values_are_refused():
    """PK_HOST_WAIT = spin | block | poll makes pk_ctx_create choose the wait mode BEFORE it creates the context's stream; anything else
    is PK_ERR_BAD_ARG, not silently 'spin' (ADVICE r05).  Fresh processes: the mode is process-wide."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\nimport torch; torch.cuda.is_available()\nimport provekit_amd\n"
            "from provekit_amd._lib import ProveKitHipError\n"
            "try:\n    c = provekit_amd.Context(0); c.sync(); print('CREATED')\nexcept ProveKitHipError as e:\n    print('REFUSED', e.code)\n") % root
    for value, want in (("poll", "CREATED"), ("block", "CREATED"), ("spin", "CREATED"), ("b", "REFUSED -1"), ("sleepy", "REFUSED -1")):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PK_HOST_WAIT=value), capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert out.stdout.strip().splitlines()[-1] == want, (value, out.stdout, out.stderr[-500:])
