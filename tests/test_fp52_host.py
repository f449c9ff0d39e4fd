"""The f64-FMA Montgomery square prototype (csrc/fe52.hpp; the reference's block-multiplier portable_simd.rs:17-196 under
round-toward-zero) executed on the host through pk_probe_fp52_sqr, against its definition x^2 * 2^-260 mod p.
Inputs follow the reference's own multiplier tests: the 100k seeded loop of scalar.rs:163-206 (here 100k seeded values) and the
proptest regression inputs."""
import random

import numpy as np

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
# skyscraper/block-multiplier/proptest-regressions/scalar.txt: the shrunk failing inputs the reference keeps (l, r as 4 x u64)
def _u256(w):
    return sum(x << (64 * i) for i, x in enumerate(w))


REGRESSIONS = [
    _u256([0, 0, 0, 1]),
    _u256([0, 887, 0, 15778841185528309819]),
    _u256([458854615557053794, 8784556235901218364, 1751211468174275388, 16873806747226852460]),
]


def limbs4(vals):
    a = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        for k in range(4):
            a[i, k] = (v >> (64 * k)) & (2**64 - 1)
    return a


def fp52_inputs(n_random, seed):
    rnd = random.Random(seed)
    edge = [0, 1, 2, P - 1, P - 2, P, P + 1, 2 * P, 2**256 - 1, 2**255, 2**256 - 2 * P, 2**52 - 1, 2**52, 2**104 - 1, (2**52 - 1) * (1 + 2**52 + 2**104 + 2**156 + 2**208) % 2**256]
    return edge + REGRESSIONS + [rnd.randrange(P) for _ in range(n_random)] + [rnd.randrange(2**256) for _ in range(n_random // 4)]


def check_fp52(vals, out):
    inv = pow(2, -260, P)
    for i, v in enumerate(vals):
        limbs = [int(out[i, k]) for k in range(5)]
        assert all(l < 2**52 for l in limbs), (i, hex(v))
        r = sum(l << (52 * k) for k, l in enumerate(limbs))
        assert r < 2**257, (i, hex(v))
        assert r % P == v * v * inv % P, (i, hex(v))


def test_fp52_square_host_matches_definition():
    from provekit_amd._lib import lib
    from tools.pk_probes import lib as probes

    vals = fp52_inputs(100_000, 52)
    a = limbs4(vals)
    out = np.zeros((len(vals), 5), dtype=np.uint64)
    assert probes.pk_probe_fp52_sqr(a.ctypes.data, out.ctypes.data, len(vals)) == 0
    check_fp52(vals, out)


def test_fp52_square_chain_stays_in_its_lazy_domain():
    """the bound the rate probe relies on: outputs (< 2^257, limbs < 2^52) are valid inputs again"""
    from provekit_amd._lib import lib
    from tools.pk_probes import lib as probes

    rnd = random.Random(7)
    vals = [rnd.randrange(2**256) for _ in range(2000)] + [2**256 - 1]
    a = limbs4(vals)
    out = np.zeros((len(vals), 5), dtype=np.uint64)
    assert probes.pk_probe_fp52_sqr(a.ctypes.data, out.ctypes.data, len(vals)) == 0
    worst = max(sum(int(out[i, k]) << (52 * k) for k in range(5)) for i in range(len(vals)))
    assert worst < 2**256  # x < 2^256 -> x^2/2^260 + (4 * 2^52 + 1) p / 2^52 ... < 2^256
