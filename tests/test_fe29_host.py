"""CPU: the library's 29-bit-limb arithmetic and Skyscraper (the exact __host__ __device__ source the kernels
compile) executed on the host through pk_selftest_arith, against the oracle and the golden vectors."""
import json
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VEC = json.load(open(os.path.join(G, "vectors.json")))
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def run(op, a, b=None):
    from provekit_amd._lib import lib

    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    out = np.empty_like(a)
    bb = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4) if b is not None else None
    rc = lib.pk_selftest_arith(op, a.ctypes.data, bb.ctypes.data if bb is not None else None, out.ctypes.data, a.shape[0])
    assert rc == 0
    return out


def rand_fe(n, seed, bound=P):
    rng = np.random.default_rng(seed)
    return [int.from_bytes(rng.bytes(32), "little") % bound for _ in range(n)]


def test_mul_edge_and_random(oracle):
    edge = [0, 1, 2, P - 1, P - 2, (1 << 253), (1 << 29) - 1, (1 << 58) - 1, P // 2, P // 3]
    xs = [a for a in edge for _ in edge] + rand_fe(3000, 1)
    ys = [b for _ in edge for b in edge] + rand_fe(3000, 2)
    a, b = oracle.ints_to_limbs(xs), oracle.ints_to_limbs(ys)
    assert np.array_equal(run(0, a, b), oracle.binop("pko_fe_mul", a, b))


def test_lazy_product_and_square_wide_inputs(oracle):
    """inputs anywhere below 2^256 (the lazy contract), exact result mod p"""
    xs = rand_fe(2000, 3, 1 << 256) + [(1 << 256) - 1, P, 2 * P, 5 * P]
    ys = rand_fe(2000, 4, 1 << 256) + [(1 << 256) - 1, P, 2 * P + 1, 5 * P]
    rinv = pow(1 << 256, -1, P)
    a, b = oracle.ints_to_limbs(xs), oracle.ints_to_limbs(ys)
    assert oracle.limbs_to_ints(run(4, a, b)) == [x * y * rinv % P for x, y in zip(xs, ys)]
    assert oracle.limbs_to_ints(run(5, a)) == [x * x * rinv % P for x in xs]


def test_from_mont(oracle):
    xs = rand_fe(2000, 5) + [0, 1, P - 1]
    a = oracle.ints_to_limbs(xs)
    assert np.array_equal(run(3, a), oracle.from_mont(a))


@pytest.mark.parametrize("version", [2, 1])
def test_compress_golden_and_random(oracle, version):
    vec = VEC["compress_v2" if version == 2 else "compress_v1"]
    a = oracle.ints_to_limbs(int(x, 16) for x, _, _ in vec)
    b = oracle.ints_to_limbs(int(y, 16) for _, y, _ in vec)
    assert oracle.limbs_to_ints(run(1 if version == 2 else 2, a, b)) == [int(e, 16) for _, _, e in vec]
    n = 20000
    msgs = np.random.default_rng(9).integers(0, 256, size=64 * n, dtype=np.uint8)
    m = msgs.view(np.uint64).reshape(n, 8)
    got = run(1 if version == 2 else 2, np.ascontiguousarray(m[:, :4]), np.ascontiguousarray(m[:, 4:]))
    assert got.tobytes() == oracle.compress_many(msgs.tobytes(), version)


def test_compress_near_modulus_boundaries(oracle):
    """values that exercise the 'almost reduced' paths: multiples of p plus/minus small offsets"""
    vals = []
    for k in range(6):
        for d in (-3, -1, 0, 1, 2, 1 << 20, 1 << 232):
            v = k * P + d
            if 0 <= v < 1 << 256:
                vals.append(v)
    xs = [a for a in vals for _ in vals]
    ys = [b for _ in vals for b in vals]
    a, b = oracle.ints_to_limbs(xs), oracle.ints_to_limbs(ys)
    msgs = np.concatenate([a, b], axis=1).astype("<u8").tobytes()
    assert run(1, a, b).tobytes() == oracle.compress_many(msgs)


def test_host_sponge_permutation_and_tag(oracle):
    """the transcript's host pieces: 64-bit Skyscraper permutation == the Python restatement (which passes the
    reference KATs), Keccak duplex tag == hashlib's SHA3 on a padded single block"""
    import ctypes as C
    import hashlib
    import random
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(G), "..", "oracle"))
    import pyref as pr
    from provekit_amd._lib import lib

    random.seed(7)
    cases = [(0, 0), (P - 1, P - 1), (P, (1 << 256) - 1)] + [(random.getrandbits(256), random.getrandbits(256)) for _ in range(300)]
    for a, b in cases:
        l = np.array(pr.int_to_limbs(a), dtype=np.uint64)
        r = np.array(pr.int_to_limbs(b), dtype=np.uint64)
        assert lib.pk_selftest_permute(l.ctypes.data, r.ctypes.data) == 0
        assert (pr.limbs_to_int(l), pr.limbs_to_int(r)) == pr.permute(a, b)
    for m in (b"", b"abc", b"x" * 100):
        data = m + b"\x06" + bytes(136 - len(m) - 2) + b"\x80"
        out = (C.c_uint8 * 32)()
        assert lib.pk_selftest_keccak_tag(data, len(data), out) == 0
        assert bytes(out) == hashlib.sha3_256(m).digest()


def test_wide_reduce(oracle):
    """reduce.hpp wide_reduce: limb sums of up to 1024 field elements -> the sum mod p (op 14 forms 700 x + 324 y)"""
    edge = [0, 1, P - 1, P - 2, P // 2, (1 << 253), P - (1 << 224), (1 << 224) - 1]
    xs = [a for a in edge for _ in edge] + rand_fe(3000, 21)
    ys = [b for _ in edge for b in edge] + rand_fe(3000, 22)
    got = oracle.limbs_to_ints(run(14, oracle.ints_to_limbs(xs), oracle.ints_to_limbs(ys)))
    assert got == [(700 * x + 324 * y) % P for x, y in zip(xs, ys)]


def test_lazy_reductions_at_their_limits(oracle):
    """mirror of skyscraper/core/src/reduce.rs:76-126 (reduce / reduce_partial incl. the *_max cases): the 'almost reduced'
    form of ANY 256-bit input is congruent to it and below p (1 + 2^-10); the exact conditional subtraction is exact"""
    edge = [0, 1, P - 1, P, P + 1, 2 * P - 1, 2 * P, 3 * P + 7, 4 * P - 1, 5 * P, 5 * P + 123, (1 << 256) - 1, (1 << 256) - P, (1 << 255), (1 << 232) - 1]
    xs = edge + rand_fe(4000, 31, 1 << 256)
    out = oracle.limbs_to_ints(run(9, oracle.ints_to_limbs(xs)))  # pack29(unpack_reduce29(x))
    bound = P + (P >> 10)
    for x, r in zip(xs, out):
        assert r % P == x % P and r < bound, hex(x)
    ys = [v for v in edge if v < 2 * P] + rand_fe(2000, 32, 2 * P)
    assert oracle.limbs_to_ints(run(12, oracle.ints_to_limbs(ys))) == [y % P for y in ys]  # cond_sub_p29


def test_bar_every_sbox_input(oracle):
    """bar (skyscraper/core/src/bar.rs:15-31, sbox :40-42) on canonical inputs whose bytes run through all 256 values (the
    reference's proptest `test_sbox_ref`, exhaustively): the result is congruent to the restatement's bar and below 2.3 p"""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref as pr

    xs = []
    for base in range(0, 256, 31):  # 31 free bytes per input; the top byte stays below 0x30 so the value is < p
        b = bytes((base + i) % 256 for i in range(31)) + bytes([0x2F])
        xs.append(int.from_bytes(b, "little"))
    for top in (0x00, 0x30):  # p's own top byte with a smaller next byte is still < p
        xs.append(int.from_bytes(bytes(range(200, 231)) + bytes([top]), "little") % P)
    xs += rand_fe(3000, 33)
    seen = set()
    for x in xs:
        seen.update(x.to_bytes(32, "little"))
    assert len(seen) == 256
    got = oracle.limbs_to_ints(run(13, oracle.ints_to_limbs(xs)))
    for x, g in zip(xs, got):
        assert g % P == pr.bar(x) and g < 2 * P + (3 * P) // 10, hex(x)


@pytest.mark.parametrize("version", [2, 1])
def test_scaled_compress_equals_reference_compress(oracle, version):
    """skyscraper29s.hpp (state scaled by 32, fused rounds) against the plain path, the golden vectors and the oracle"""
    vec = VEC["compress_v2" if version == 2 else "compress_v1"]
    a = oracle.ints_to_limbs(int(x, 16) for x, _, _ in vec)
    b = oracle.ints_to_limbs(int(y, 16) for _, y, _ in vec)
    assert oracle.limbs_to_ints(run(15 if version == 2 else 16, a, b)) == [int(e, 16) for _, _, e in vec]
    edge = [0, 1, P - 1, P, P + 1, 2 * P, (1 << 256) - 1, (1 << 255), 5 * P, (1 << 253) - 1]
    xs = [x for x in edge for _ in edge] + rand_fe(20000, 31 + version, 1 << 256)
    ys = [y for _ in edge for y in edge] + rand_fe(20000, 41 + version, 1 << 256)
    a, b = oracle.ints_to_limbs(xs), oracle.ints_to_limbs(ys)
    assert np.array_equal(run(15 if version == 2 else 16, a, b), run(1 if version == 2 else 2, a, b))
    msgs = np.concatenate([a, b], axis=1).astype("<u8").tobytes()
    assert run(15 if version == 2 else 16, a, b).astype("<u8").tobytes() == oracle.compress_many(msgs, version)


def test_scaled_conversions(oracle):
    xs = rand_fe(5000, 51, 1 << 256) + [0, 1, P - 1, P, P + 1, (1 << 256) - 1, 31 * P // 32, P // 32 + 1]
    a = oracle.ints_to_limbs(xs)
    assert oracle.limbs_to_ints(run(17, a)) == [x % P for x in xs]  # x -> 32x mod p -> /32: the identity mod p
    ms = rand_fe(5000, 52) + [0, 1, P - 1]
    m = oracle.ints_to_limbs(ms)
    assert np.array_equal(run(18, m), oracle.from_mont(m))  # Montgomery image -> scaled -> canonical == into_bigint()


def test_scaled_fold_stays_in_domain(oracle):
    """three chained compressions without leaving the scaled domain (what the leaf fold does)"""
    xs, ys = rand_fe(3000, 61, 1 << 256), rand_fe(3000, 62, 1 << 256)
    a, b = oracle.ints_to_limbs(xs), oracle.ints_to_limbs(ys)
    c = lambda u, v: np.frombuffer(oracle.compress_many(np.concatenate([u, v], axis=1).astype("<u8").tobytes()), dtype="<u8").reshape(-1, 4)
    assert np.array_equal(run(19, a, b), c(c(c(a, b), a), b))


def test_grouped_dot_product(oracle):
    """fe29.hpp dot29 (one Montgomery reduction per group of products): 3xy + x^2 + y^2 as Montgomery products"""
    xs, ys = rand_fe(4000, 71) + [0, P - 1, 1, P - 1], rand_fe(4000, 72) + [0, P - 1, P - 1, 1]
    rinv = pow(1 << 256, -1, P)
    a, b = oracle.ints_to_limbs(xs), oracle.ints_to_limbs(ys)
    assert oracle.limbs_to_ints(run(20, a, b)) == [(3 * x * y + x * x + y * y) * rinv % P for x, y in zip(xs, ys)]


def test_safegcd_inverse_is_the_field_inverse(oracle):
    """feinv.hpp (the witness builders' Inverse, witness_builder.rs:66-69) against pow(x, -1, p): plain integers (op 22) and
    Montgomery in / Montgomery out (op 21); both step forms (ops 22, 23); edge values, every bit length, 20 k random elements; 0 -> 0"""
    import random

    rnd = random.Random(1)
    vals = [0, 1, 2, 3, P - 1, P - 2, (P + 1) // 2, (P - 1) // 2, 1 << 253, 1 << 128, (1 << 30) - 1, 1 << 30]
    vals += [rnd.randrange(1 << k) for k in range(1, 254)] + [P - 1 - rnd.randrange(1 << k) for k in range(1, 250, 7)] + rand_fe(20000, 77)
    want = [pow(v, -1, P) if v else 0 for v in vals]
    a = oracle.ints_to_limbs(vals)
    assert oracle.limbs_to_ints(run(22, a)) == want
    assert oracle.limbs_to_ints(run(23, a)) == want  # the variable-time steps (what the witness builders run)
    assert oracle.limbs_to_ints(oracle.from_mont(run(21, oracle.to_mont(a)))) == want


def test_shoup_product_by_a_constant(oracle):
    """shoup261_29 (fe29.hpp): a * w mod p with the precomputed quotient floor(w 2^261 / p) -- 143 multiply-adds against the
    Montgomery product's 171 -- on the host build of the device source: any 256-bit multiplicand (up to 5.3 p), every multiplier
    below p incl. the edges, and the lazy-limb form the NTT butterflies feed it (a sum, limbs above 2^29)"""
    edge = [0, 1, 2, P - 1, P - 2, (1 << 253), (1 << 29) - 1, (1 << 58) - 1, P // 2, P // 3, (1 << 254) - 1]
    xs = [a for a in edge + [(1 << 256) - 1, 5 * P, 2 * P + 1] for _ in edge] + rand_fe(4000, 31, 1 << 256)
    ys = [b for _ in edge + [0, 0, 0] for b in edge] + rand_fe(4000, 32)
    a, b = oracle.ints_to_limbs(xs), oracle.ints_to_limbs(ys)
    assert oracle.limbs_to_ints(run(24, a, b)) == [x * (y % P) % P for x, y in zip(xs, ys)]
    assert oracle.limbs_to_ints(run(25, a, b)) == [((x % P) + (y % P)) * (y % P) % P for x, y in zip(xs, ys)]


ROOT28 = 19103219067921713944291392827692070036145651957329286315305642004821462161904  # ark-bn254 Fr::TWO_ADIC_ROOT_OF_UNITY
W8 = pow(ROOT28, 1 << 25, P)


def test_ntt_w8_constants_and_their_shoup_quotients(oracle):
    """ntt_regs.hpp: the three in-register multipliers w_8, w_8^2, w_8^3 of the NTT's butterfly network are compile-time constants;
    recompute them from the two-adic root and check each stored quotient floor(w 2^261 / p) against the long division"""
    assert pow(W8, 8, P) == 1 and pow(W8, 4, P) == P - 1
    got = oracle.limbs_to_ints(run(26, oracle.ints_to_limbs([1, 2, 3])))
    assert got == [pow(W8, e, P) for e in (1, 2, 3)]  # bit 255 clear: the quotients are the exact ones


def dft_regs(le, d, xs, tw=None):
    from provekit_amd._lib import lib

    a = np.ascontiguousarray(xs, dtype=np.uint64).reshape(-1, 4)
    assert a.shape[0] % (1 << le) == 0
    out = np.empty_like(a)
    t = np.ascontiguousarray(tw, dtype=np.uint64).reshape(-1, 4) if tw is not None else None
    assert lib.pk_selftest_dft(a.ctypes.data, t.ctypes.data if t is not None else None, out.ctypes.data, le, d, a.shape[0] >> le) == 0
    return out


def dft_by_definition(le, d, xs, tw=None):
    """register (bitrev_d(a) << e) | batch holds X[a][batch] = sum_n x[(n << e) | batch] w^(a n), w = w_(2^d), e = le - d"""
    e, nx, out = le - d, 1 << le, []
    w = pow(W8, 8 >> d, P)
    for g in range(0, len(xs), nx):
        y = [0] * nx
        for a in range(1 << d):
            ra = int(format(a, f"0{d}b")[::-1], 2)
            for b in range(1 << e):
                y[(ra << e) | b] = sum(xs[g + ((n << e) | b)] * pow(w, a * n, P) for n in range(1 << d)) % P
        if tw is not None:
            y = [v * t % P for v, t in zip(y, tw[g:g + nx])]
        out += y
    return out


@pytest.mark.parametrize("le,d", [(3, 1), (3, 2), (2, 1), (2, 2)])
def test_ntt_butterfly_network_on_the_host(oracle, le, d):
    """ntt_regs.hpp dft_regs<le, d> -- the device source of the NTT pass's register rounds, Shoup products and lazy sums included --
    against the definition of the DFT: random inputs, and the extremes of the network's input contract (normalised limbs, value
    below 1.2p: what a pass loads or a round leaves), with and without the twiddle product the pass kernel applies to the network's
    unreduced outputs"""
    HI = 12 * P // 10
    ALL_ONES = (((HI >> 232) - 1) << 232) | ((1 << 232) - 1)  # every limb below the top one at its maximum
    edge = [0, 1, P - 1, P, P + 1, HI, ALL_ONES, (1 << 253), HI - (1 << 29)]
    xs = rand_fe(8 * 300, 41 + d) + rand_fe(8 * 100, 51 + d, HI)
    xs += [v for v in edge for _ in range(8)]  # all equal: sums reach 2^le x the value before the reductions
    xs += [HI if (i >> k) & 1 else 0 for k in range(3) for i in range(8)]  # extremes alternating at every stride
    xs += [0 if (i >> k) & 1 else HI for k in range(3) for i in range(8)]
    tws = rand_fe(len(xs) - 16, 61 + d) + [P - 1] * 8 + [1] * 8
    assert oracle.limbs_to_ints(dft_regs(le, d, oracle.ints_to_limbs(xs))) == dft_by_definition(le, d, xs)
    assert oracle.limbs_to_ints(dft_regs(le, d, oracle.ints_to_limbs(xs), oracle.ints_to_limbs(tws))) == dft_by_definition(le, d, xs, tws)
