"""GPU: the proof RNG (ChaCha12-keyed rejection sampling, provekit_amd/csrc/prover.hip) against its Python restatement, and
the production path of pk_prove (fresh OS randomness per proof)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_device_draw_matches_restatement(ctx, oracle):
    from test_host_only import random_fe_py

    from provekit_amd._lib import lib

    seed = bytes((7 * i + 3) & 0xFF for i in range(32))
    n = 5000
    d = ctx.alloc_fe(n)
    for stream in (1, 2, 77):
        ctx._check(lib.pk_selftest_random_fe(ctx.handle, seed, stream, d.ptr, n))
        got = oracle.limbs_to_ints(ctx.download_fe(d, n))
        assert all(v < oracle.P for v in got)
        for i in list(range(200)) + [n - 1, 1234, 4097]:
            assert got[i] == random_fe_py(seed, stream, i), (stream, i)
    a = oracle.limbs_to_ints(ctx.download_fe(d, n))
    ctx._check(lib.pk_selftest_random_fe(ctx.handle, bytes(32), 77, d.ptr, n))
    assert oracle.limbs_to_ints(ctx.download_fe(d, n)) != a  # another key, another draw


def test_draw_is_uniform_enough(ctx, oracle):
    """sanity statistics on 2^20 elements: top-limb mean near p/2 and no duplicate values"""
    from provekit_amd._lib import lib

    n = 1 << 20
    d = ctx.alloc_fe(n)
    ctx._check(lib.pk_selftest_random_fe(ctx.handle, bytes(range(32)), 5, d.ptr, n))
    x = ctx.download_fe(d, n)
    top = x[:, 3].astype(np.float64)
    p_top = float(oracle.P >> 192)
    assert abs(top.mean() / p_top - 0.5) < 0.005
    assert len(np.unique(x[:, 0])) > n - 8
