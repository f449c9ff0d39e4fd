"""Host-side BN254-Fr helpers: limb packing and seeded sampling (no arithmetic hot path here)."""
from __future__ import annotations

import numpy as np

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
R = (1 << 256) % P
R_INV = pow(R, -1, P)
_P_LIMBS = np.array([(P >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)


def ints_to_limbs(xs) -> np.ndarray:
    """list of ints < 2^256 -> (n,4) uint64 little-endian limbs"""
    xs = list(xs)
    buf = b"".join(int(x).to_bytes(32, "little") for x in xs)
    return np.frombuffer(buf, dtype="<u8").reshape(len(xs), 4).copy()


def limbs_to_ints(a: np.ndarray) -> list[int]:
    a = np.ascontiguousarray(a, dtype="<u8").reshape(-1, 4)
    b = a.tobytes()
    return [int.from_bytes(b[32 * i : 32 * i + 32], "little") for i in range(a.shape[0])]


def random_canonical(n: int, seed: int) -> np.ndarray:
    """n uniform canonical field elements (< p) as (n,4) uint64; seeded, vectorised rejection sampling."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, 4), dtype=np.uint64)
    filled = 0
    while filled < n:
        m = max(16, int((n - filled) * 1.4) + 8)
        cand = rng.integers(0, 2**64, size=(m, 4), dtype=np.uint64)
        cand[:, 3] &= np.uint64((1 << 62) - 1)  # < 2^254
        ok = _lt_p(cand)
        cand = cand[ok]
        k = min(len(cand), n - filled)
        out[filled : filled + k] = cand[:k]
        filled += k
    return out


def random_field(n: int, seed: int) -> np.ndarray:
    """n seeded field elements to be *interpreted as Montgomery-form* memory images.
    Any value < p is a valid Montgomery representative, so uniform canonical
    sampling is also uniform over the field."""
    return random_canonical(n, seed)


def _lt_p(a: np.ndarray) -> np.ndarray:
    lt = np.zeros(a.shape[0], dtype=bool)
    eq = np.ones(a.shape[0], dtype=bool)
    for i in (3, 2, 1, 0):
        lt |= eq & (a[:, i] < _P_LIMBS[i])
        eq &= a[:, i] == _P_LIMBS[i]
    return lt


def to_mont_int(x: int) -> int:
    return x * R % P


def from_mont_int(x: int) -> int:
    return x * R_INV % P
