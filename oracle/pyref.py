"""Pure-Python big-int restatement of the ProveKit WHIR hot path.

TEST INFRASTRUCTURE ONLY.  Second, independent restatement used to (a) pin the C
oracle (oracle/pk_oracle.c) on small cases and (b) mint the golden vectors under
tests/golden/ (tests/golden/gen_golden.py).  Pure-Python loops: small sizes only.
Nothing in provekit_amd/ imports this module.

All values here are canonical Python ints in [0, p) unless a name says `mont`.
Citations are paths inside worldfnd/provekit @ 2025-08-29.
"""
from __future__ import annotations

import math
import struct

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
R = (1 << 256) % P
R_INV = pow(R, -1, P)
SIGMA_INV = 9915499612839321149637521777990102151350674507940716049588462388200839649614  # reference.rs:22-26
assert SIGMA_INV == R_INV
ROOT28 = pow(5, (P - 1) >> 28, P)  # ark-bn254 Fr two-adic root of unity

# skyscraper/core/src/constants.rs:30-49 (little-endian u64 limbs)
_RC_LIMBS = [
    [0x0000000000000000, 0x0000000000000000, 0x0000000000000000, 0x0000000000000000],
    [0x903C4324270BD744, 0x873125F708A7D269, 0x081DD27906C83855, 0x276B1823EA6D7667],
    [0x7AC8EDBB4B378D71, 0xE29D79F3D99E2CB7, 0x751417914C1A5A18, 0x0CF02BD758A484A6],
    [0xFA7ADC6769E5BC36, 0x1C3F8E297CCA387D, 0x0EB7730D63481DB0, 0x25B0E03F18EDE544],
    [0x57847E652F03CFB7, 0x33440B9668873404, 0x955A32E849AF80BC, 0x002882FCBE14AE70],
    [0x979231396257D4D7, 0x29989C3E1B37D3C1, 0x12EF02B47F1277BA, 0x039AD8571E2B7A9C],
    [0xB5B48465ABBB7887, 0xA72A6BC5E6BA2D2B, 0x4CD48043712F7B29, 0x1142D5410FC1FC1A],
    [0x7AB2C156059075D3, 0x17CB3594047999B2, 0x44F2C93598F289F7, 0x1D78439F69BC0BEC],
    [0x05D7A965138B8EDB, 0x36EF35A3D55C48B1, 0x8DDFB8A1AC6F1628, 0x258588A508F4FF82],
    [0x1596FB9AFCCB49E9, 0x9A7367D69A09A95B, 0x9BC43F6984E4C157, 0x13087879D2F514FE],
    [0x295CCD233B4109FA, 0xE1D72F89ED868012, 0x2E9E1EEA4BC88A8E, 0x17DADEE898C45232],
    [0x9A8590B4AA1F486F, 0xB75834B430E9130E, 0xB8E90B1034D5DE31, 0x295C6D1546E7F4A6],
    [0x850ADCB74C6EB892, 0x07699EF305B92FC3, 0x4EF96A2BA1720F2D, 0x1288CA0E1D3ED446],
    [0x01960F9349D1B5EE, 0x8CCAD30769371C69, 0xE5C81E8991C98662, 0x17563B4D1AE023F3],
    [0x6BA01E9476B32917, 0xA1CB0A3ADD977BC9, 0x86815A945815F030, 0x2869043BE91A1EEA],
    [0x81776C885511D976, 0x7475D34F47F414E7, 0x5D090056095D96CF, 0x14941F0AFF59E79A],
    [0xBC40B4FD8FC8C034, 0xBB7142C3CCE4FD48, 0x318356758A39005A, 0x1CE337A190F4379F],
    [0x0000000000000000, 0x0000000000000000, 0x0000000000000000, 0x0000000000000000],
]


def limbs_to_int(l):
    return sum(int(x) << (64 * i) for i, x in enumerate(l))


def int_to_limbs(x):
    return [(x >> (64 * i)) & (2**64 - 1) for i in range(4)]


RC = [limbs_to_int(l) for l in _RC_LIMBS]


def to_mont(x):
    return x * R % P


def from_mont(x):
    return x * R_INV % P


# ---- H1 -------------------------------------------------------------------
def sbox(v):  # reference.rs:96-98
    rot = lambda x, k: ((x << k) | (x >> (8 - k))) & 0xFF
    return rot(v ^ (rot(~v & 0xFF, 1) & rot(v, 2) & rot(v, 3)), 1)


def bar(x):  # reference.rs:80-94
    b = x.to_bytes(32, "little")
    b = b[16:] + b[:16]
    b = bytes(sbox(v) for v in b)
    return int.from_bytes(b, "little") % P


def sq(x):
    return x * x * SIGMA_INV % P


def ss(rnd, l, r):  # reference.rs:63-69
    r = (r + sq(l) + RC[rnd]) % P
    l, r = r, l
    r = (r + sq(l) + RC[rnd + 1]) % P
    l, r = r, l
    return l, r


def bb(rnd, l, r):  # reference.rs:72-78
    r = (r + bar(l) + RC[rnd]) % P
    l, r = r, l
    r = (r + bar(l) + RC[rnd + 1]) % P
    l, r = r, l
    return l, r


def permute(l, r):  # reference.rs:49-60
    l %= P
    r %= P
    l, r = ss(0, l, r)
    l, r = ss(2, l, r)
    l, r = ss(4, l, r)
    l, r = bb(6, l, r)
    l, r = ss(8, l, r)
    l, r = bb(10, l, r)
    l, r = ss(12, l, r)
    l, r = ss(14, l, r)
    l, r = ss(16, l, r)
    return l, r


def compress(l, r):  # reference.rs:41-46
    return (permute(l, r)[0] + l) % P


def compress_v1(l, r):  # v1.rs:19-32
    l %= P
    r %= P
    t = l
    fs = [sq, sq, bar, bar, sq, sq, bar, bar, sq, sq]
    rcs = [0, RC[1], RC[2], RC[3], RC[4], RC[5], RC[6], RC[7], RC[8], 0]
    for f, rc in zip(fs, rcs):
        l, r = (r + f(l) + rc) % P, l
    return (l + t) % P


def leaf_hash(leaf, version=2):  # provekit/common/src/skyscraper/whir.rs:30-48
    c = compress if version == 2 else compress_v1
    h = leaf[0]
    for x in leaf[1:]:
        h = c(h, x)
    return h


def merkle_nodes(leaf_digests, version=2):
    """heap layout: nodes[1] root, nodes[n+i] leaf digest i"""
    c = compress if version == 2 else compress_v1
    n = len(leaf_digests)
    nodes = [0] * n + list(leaf_digests)
    for i in range(n - 1, 0, -1):
        nodes[i] = c(nodes[2 * i], nodes[2 * i + 1])
    return nodes


# ---- A1 -------------------------------------------------------------------
def mont_mul(a, b):
    return a * b * R_INV % P


# ---- T1 / N1 / E1 / W1 ----------------------------------------------------
def to_coeffs(evals):
    v = list(evals)
    n = len(v)
    h = 1
    while h < n:
        for i in range(n):
            if i & h:
                v[i] = (v[i] - v[i ^ h]) % P
        h <<= 1
    return v


def root_of_unity(log_n):
    return pow(ROOT28, 1 << (28 - log_n), P)


def rs_encode_naive(coeff_polys, n_vars, log_inv_rate, fold):
    """SURVEY 8a row N1, by the definition: leaf_i[b*2^fold + j] = f_{b,j}(w_rows^i),
    f_{b,j}(X) = sum_t c_b[2^fold t + j] X^t."""
    fw = 1 << fold
    rows = 1 << (n_vars + log_inv_rate - fold)
    w = root_of_unity(n_vars + log_inv_rate - fold)
    leaves = []
    for i in range(rows):
        x = pow(w, i, P)
        leaf = []
        for c in coeff_polys:
            for j in range(fw):
                acc = 0
                for t in reversed(range(len(c) // fw)):
                    acc = (acc * x + c[fw * t + j]) % P
                leaf.append(acc)
        leaves.append(leaf)
    return leaves


def eval_univariate(c, z):
    acc = 0
    for x in reversed(c):
        acc = (acc * z + x) % P
    return acc


def multivar_poly(coefs, vs):  # utilities.go:15-22
    if not vs:
        return coefs[0]
    h = len(coefs) // 2
    return (multivar_poly(coefs[:h], vs[:-1]) + vs[-1] * multivar_poly(coefs[h:], vs[:-1])) % P


def fold_coeffs(c, r):
    k = 1 << len(r)
    return [multivar_poly(c[i : i + k], r) for i in range(0, len(c), k)]


# ---- S2 / S3 / W2 / W3 ----------------------------------------------------
def eq_table(r):  # sumcheck.rs:146-171, first variable = MSB
    m = len(r)
    out = []
    for i in range(1 << m):
        acc = 1
        for j in range(m):
            bit = (i >> (m - 1 - j)) & 1
            acc = acc * (r[j] if bit else (1 - r[j])) % P
        out.append(acc)
    return out


def expand_from_univariate(z, n):  # utilities.go:182-190
    res = [0] * n
    acc = z
    for i in range(n):
        res[n - 1 - i] = acc
        acc = acc * acc % P
    return res


def sumcheck_cubic_round(a, b, c, eq, fold=None):
    """returns ((f0, f_em1, f_inf), new_a, new_b, new_c, new_eq)"""
    arrs = [list(a), list(b), list(c), list(eq)]
    n = len(a)
    if fold is not None:
        q = n // 4
        for m in arrs:
            for i in range(q):
                m[i] = (m[i] + fold * (m[2 * q + i] - m[i])) % P
                m[q + i] = (m[q + i] + fold * (m[3 * q + i] - m[q + i])) % P
        arrs = [m[: n // 2] for m in arrs]
        n //= 2
    h = n // 2
    s = [0, 0, 0]
    A, B, C, E = arrs
    for i in range(h):
        a0, a1, b0, b1, c0, c1, e0, e1 = A[i], A[i + h], B[i], B[i + h], C[i], C[i + h], E[i], E[i + h]
        s[0] += e0 * (a0 * b0 - c0)
        s[1] += (2 * e0 - e1) * ((2 * a0 - a1) * (2 * b0 - b1) - (2 * c0 - c1))
        s[2] += (e1 - e0) * (a1 - a0) * (b1 - b0)
    return tuple(x % P for x in s), A, B, C, E


def sumcheck_quadratic_round(f, w, fold=None):
    f, w = list(f), list(w)
    if fold is not None:
        f = [(f[2 * i] + fold * (f[2 * i + 1] - f[2 * i])) % P for i in range(len(f) // 2)]
        w = [(w[2 * i] + fold * (w[2 * i + 1] - w[2 * i])) % P for i in range(len(w) // 2)]
    h0 = h1 = h2 = 0
    for i in range(len(f) // 2):
        f0, f1, w0, w1 = f[2 * i], f[2 * i + 1], w[2 * i], w[2 * i + 1]
        h0 += f0 * w0
        h1 += f1 * w1
        h2 += (2 * f1 - f0) * (2 * w1 - w0)
    return (h0 % P, h1 % P, h2 % P), f, w


# ---- P1 -------------------------------------------------------------------
def f64_to_u256(f):  # pow.rs:61-82
    bits = struct.unpack("<Q", struct.pack("<d", f))[0]
    sign = bits >> 63
    exp_bits = (bits >> 52) & 0x7FF
    frac = bits & ((1 << 52) - 1)
    if exp_bits == 0:
        exp, sig = -1022, frac
    else:
        exp, sig = exp_bits - 1023, frac + (1 << 52)
    if sign:
        return 0
    if exp > 256:
        return (1 << 256) - 1
    shift = exp - 52
    if shift < 0:
        r = math.floor(abs(f) + 0.5)  # f64::round = half away from zero
        return min(r, 2**64 - 1)
    limb, sh = shift // 64, shift % 64
    out = [0, 0, 0, 0]
    out[limb] = (sig << sh) & (2**64 - 1)
    if sh != 0 and limb < 3:
        out[limb + 1] = sig >> (64 - sh)
    return limbs_to_int(out)


def pow_threshold(difficulty):  # pow.rs:14-22
    modulus = float(_P3) * 2.0**192
    return f64_to_u256(2.0 ** (-difficulty) * modulus)


_P3 = (P >> 192) & (2**64 - 1)
