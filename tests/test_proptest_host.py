"""CPU, property-based (hypothesis; the reference uses proptest the same way -- skyscraper/core/src/simple.rs:20-27,
block-multiplier/src/scalar.rs:146-153): the library's host-compiled kernel arithmetic against the independent pure-Python
restatement (oracle/pyref.py), with hypothesis steering towards boundary values."""
import os
import sys

import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import pyref as pr  # noqa: E402

P = pr.P
u256 = st.integers(min_value=0, max_value=(1 << 256) - 1)
near = st.builds(lambda k, d: max(0, min((1 << 256) - 1, k * P + d)), st.integers(0, 5), st.integers(-(1 << 64), 1 << 64))
any256 = st.one_of(u256, near)
felt = st.one_of(st.integers(0, P - 1), st.builds(lambda d: (P - 1 - d) % P, st.integers(0, 1 << 32)))


def limbs(xs):
    return np.array([[(x >> (64 * i)) & ((1 << 64) - 1) for i in range(4)] for x in xs], dtype=np.uint64)


def run(op, a, b=None):
    from provekit_amd._lib import lib

    a = limbs(a)
    out = np.empty_like(a)
    bb = limbs(b) if b is not None else None
    assert lib.pk_selftest_arith(op, a.ctypes.data, bb.ctypes.data if bb is not None else None, out.ctypes.data, a.shape[0]) == 0
    return [int(sum(int(out[i, j]) << (64 * j) for j in range(4))) for i in range(out.shape[0])]


@settings(max_examples=300, deadline=None)
@given(any256, any256)
def test_compress_v2_equals_the_restatement(l, r):
    assert run(1, [l], [r]) == [pr.compress(l, r)]


@settings(max_examples=150, deadline=None)
@given(any256, any256)
def test_compress_v1_equals_the_restatement(l, r):
    assert run(2, [l], [r]) == [pr.compress_v1(l, r)]


@settings(max_examples=300, deadline=None)
@given(felt, felt)
def test_montgomery_product(a, b):
    assert run(0, [a], [b]) == [a * b * pr.R_INV % P]


@settings(max_examples=300, deadline=None)
@given(any256, any256)
def test_lazy_product_any_input(a, b):
    assert run(4, [a], [b]) == [a * b * pr.R_INV % P]
    assert run(5, [a]) == [a * a * pr.R_INV % P]
