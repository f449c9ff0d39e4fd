"""GPU: the library's __host__ __device__ arithmetic must give the same bits when compiled for the device as when
compiled for the host (which tests/test_fe29_host.py ties to the oracle).  This is the guard against device-codegen
surprises (it caught an AMDGPU mul24 miscompile of the 24-bit Montgomery step)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("op", list(range(15)) + [21, 22, 23])  # 21-23: the safegcd inverse (feinv.hpp), both step forms
def test_device_equals_host(ctx, oracle, op):
    from provekit_amd._lib import lib
    from tools.pk_probes import lib as probes
    from provekit_amd.field import random_field

    n = 4096
    a, b = random_field(n, 100 + op), random_field(n, 200 + op)
    a[:6] = oracle.ints_to_limbs([0, 1, oracle.P - 1, oracle.P - 2, (1 << 253) - 1, 1 << 232])
    b[:6] = oracle.ints_to_limbs([0, oracle.P - 1, oracle.P - 1, 1, 12345, (1 << 200) + 7])
    if op in (1, 2, 4, 5, 9):  # these accept any 256-bit input
        rng = np.random.default_rng(op)
        a[6:600] = rng.integers(0, 2**64, size=(594, 4), dtype=np.uint64)
        b[6:600] = rng.integers(0, 2**64, size=(594, 4), dtype=np.uint64)
    host = np.empty_like(a)
    assert lib.pk_selftest_arith(op, a.ctypes.data, b.ctypes.data, host.ctypes.data, n) == 0
    da, db, do = ctx.upload(a), ctx.upload(b), ctx.alloc_fe(n)
    ctx._check(probes.pk_probe_arith_device(ctx.handle, op, da.ptr, db.ptr, do.ptr, n))
    assert np.array_equal(ctx.download_fe(do, n), host)


def test_fp52_prototype_device_equals_host_and_definition(ctx):
    """csrc/fe52.hpp on the device (MODE.FP_ROUND = RTZ set by the kernel) against the host execution under
    fesetround(FE_TOWARDZERO) and against x^2 * 2^-260 mod p."""
    import ctypes as C

    from provekit_amd._lib import lib
    from tools.pk_probes import lib as probes
    from test_fp52_host import check_fp52, fp52_inputs, limbs4

    vals = fp52_inputs(100_000, 53)
    a = limbs4(vals)
    n = len(vals)
    host = np.zeros((n, 5), dtype=np.uint64)
    assert probes.pk_probe_fp52_sqr(a.ctypes.data, host.ctypes.data, n) == 0
    da, do = ctx.upload(a), ctx.alloc_fe(2 * n)
    ctx._check(probes.pk_probe_fp52_sqr_device(ctx.handle, da.ptr, do.ptr, n))
    dev = np.zeros((n, 5), dtype=np.uint64)
    ctx._check(lib.pk_memcpy_d2h(ctx.handle, dev.ctypes.data, do.ptr, dev.nbytes))
    assert np.array_equal(dev, host)
    check_fp52(vals[:5000], dev)


def test_cooperative_square_round_prototype_matches_the_lane(ctx):
    """north_star's "one-wavefront-per-node Skyscraper rounds" as a prototype (csrc/selftest.hip coop_sq_round: limbs in lanes,
    v_readlane broadcasts, DPP window shift) against the product's lone-lane round: equal mod p over the round counts a compression
    runs between reductions, limbs within the bounds the next round needs.  (Its speed is measured by tools/coop_round.py:
    profiles/r03_coop_round.json, DESIGN.md 4 -- slower than the lane.)"""
    import ctypes as C
    import random

    from provekit_amd._lib import lib
    from tools.pk_probes import lib as probes

    P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    limbs = lambda v: [(v >> (29 * k)) & ((1 << 29) - 1) if k < 8 else v >> 232 for k in range(9)]
    value = lambda ls: sum(int(x) << (29 * k) for k, x in enumerate(ls))
    rnd = random.Random(9)
    for v in [(0, 0), (1, 0), (P - 1, P - 1), ((1 << 254) - 1, 5)] + [(rnd.randrange(P), rnd.randrange(P)) for _ in range(12)]:
        L, R = (C.c_uint32 * 9)(*limbs(v[0])), (C.c_uint32 * 9)(*limbs(v[1]))
        out, cyc = (C.c_uint32 * 36)(), (C.c_uint64 * 4)()
        for n in (1, 2, 3, 6):
            ctx._check(probes.pk_probe_coop_round(ctx.handle, L, R, n, out, cyc))
            o = list(out)
            assert value(o[0:9]) % P == value(o[18:27]) % P and value(o[9:18]) % P == value(o[27:36]) % P, (v, n)
            assert max(o[0:8]) <= 1 << 29


def test_matrix_core_reduction_prototype_is_exact(ctx):
    """VERDICT r03 item 5 (measured, not adopted): x^2 * 2^-256 mod p as a constant-matrix product on the matrix core --
    v_mfma_i32_16x16x64_i8 over the 8/8/8/5-bit digits of the square's 29-bit limbs, the Montgomery factor inside the constants.
    The 36 column sums of every value must add up (at bit positions 29 (i / 4) + 8 (i % 4)) to a non-negative integer congruent to
    the definition, for random and edge inputs (tools/mfma_reduce.py; rates in profiles/r04_modmul_rates.json)."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import mfma_reduce

    info = mfma_reduce.check_parity(ctx, n=1000, seed=11)
    assert info["max_bits_of_the_unfolded_sum"] <= 271
