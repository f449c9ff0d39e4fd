"""SkyscraperPoW: the spongefish_pow::PowStrategy plug-in (provekit/common/src/skyscraper/pow.rs:14-30) on the GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import lib
from .runtime import Context, default_context


class SkyscraperPoW:
    def __init__(self, challenge: bytes, bits: float, ctx: Context | None = None):
        if not (0.0 <= bits < 60.0):
            raise ValueError("bits must be smaller than 60")  # pow.rs:16
        if len(challenge) != 32:
            raise ValueError("challenge must be 32 bytes")
        self.challenge = (C.c_uint8 * 32)(*challenge)
        self.bits = float(bits)
        self.ctx = ctx or default_context()

    def check(self, nonce: int) -> bool:
        ok = C.c_int(0)
        self.ctx._check(lib.pk_pow_check(self.ctx.handle, self.challenge, self.bits, nonce, C.byref(ok)))
        return bool(ok.value)

    def solve(self) -> int:
        n = C.c_uint64(0)
        self.ctx._check(lib.pk_pow_solve(self.ctx.handle, self.challenge, self.bits, C.byref(n)))
        return int(n.value)


def threshold(difficulty: float) -> np.ndarray:
    out = np.empty(4, dtype=np.uint64)
    rc = lib.pk_pow_threshold(difficulty, out.ctypes.data)
    if rc:
        raise ValueError("Difficulty must be in the range [0, 80)")
    return out
