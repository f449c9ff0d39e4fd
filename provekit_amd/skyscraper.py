"""Skyscraper hashing entry points, mirroring the reference's plug-in interfaces.

  compress_many(messages, hashes)      <-> skyscraper::CompressManyFn (skyscraper/core/src/lib.rs:26)
  SkyscraperCRH.evaluate(leaf)         <-> provekit/common/src/skyscraper/whir.rs:30-48
  SkyscraperTwoToOne.{evaluate,compress} <-> whir.rs:53-74
Every call runs on the GPU through libprovekit_hip; there is no host fallback.
"""
from __future__ import annotations

import numpy as np

from ._lib import PK_COL_MAJOR, PK_LEAF_MAJOR, lib
from .runtime import Context, default_context


def compress_many(messages, hashes=None, ctx: Context | None = None) -> bytes:
    """n two-to-one compressions: `messages` is 64*n bytes, result (and `hashes`, if a
    writable buffer is given) is 32*n bytes.  Raises ValueError where the reference
    panics (length not a multiple of 64 / 32, or mismatched lengths; generic.rs:18-25)."""
    ctx = ctx or default_context()
    m = np.frombuffer(bytes(messages) if not isinstance(messages, np.ndarray) else messages.tobytes(), dtype=np.uint8)
    if m.size % 64:
        raise ValueError("Message length not a multiple of 64")
    if hashes is not None:
        hlen = len(hashes)
        if hlen % 32:
            raise ValueError("Hashes length not a multiple of 32")
        if m.size != 2 * hlen:
            raise ValueError("Messages and hashes length mismatch")
    out = np.empty(m.size // 2, dtype=np.uint8)
    ctx._check(lib.pk_compress_many_host(ctx.handle, m.ctypes.data, m.size, out.ctypes.data, out.size))
    res = out.tobytes()
    if hashes is not None:
        hashes[:] = res
    return res


def compress_many_device(ctx: Context, d_messages: int, d_hashes: int, n: int):
    ctx._check(lib.pk_compress_many(ctx.handle, d_messages, d_hashes, n))


class SkyscraperCRH:
    """Leaf hash: left fold of compress over the leaf's field elements (Montgomery in,
    canonical digest out)."""

    @staticmethod
    def evaluate(leaf_mont: np.ndarray, ctx: Context | None = None) -> np.ndarray:
        leaf = np.ascontiguousarray(leaf_mont, dtype=np.uint64).reshape(-1, 4)
        if leaf.shape[0] == 0:
            raise ValueError("IncorrectInputLength(0)")  # whir.rs:47
        return leaf_hash(leaf[None, :, :], ctx=ctx)[0]


def leaf_hash(leaves_mont: np.ndarray, ctx: Context | None = None) -> np.ndarray:
    """(n_leaves, width, 4) uint64 Montgomery leaves (leaf-major) -> (n_leaves, 4) canonical digests."""
    ctx = ctx or default_context()
    leaves = np.ascontiguousarray(leaves_mont, dtype=np.uint64)
    n, w = leaves.shape[0], leaves.shape[1]
    if w == 0:
        raise ValueError("IncorrectInputLength(0)")
    d_l = ctx.upload(leaves)
    d_d = ctx.alloc_fe(max(n, 1))
    ctx._check(lib.pk_leaf_hash(ctx.handle, d_l.ptr, n, w, PK_LEAF_MAJOR, d_d.ptr))
    return ctx.download_fe(d_d, n)


class SkyscraperTwoToOne:
    @staticmethod
    def evaluate(l_canon: np.ndarray, r_canon: np.ndarray, ctx: Context | None = None) -> np.ndarray:
        msg = np.concatenate([np.asarray(l_canon, dtype=np.uint64).reshape(4), np.asarray(r_canon, dtype=np.uint64).reshape(4)])
        return np.frombuffer(compress_many(msg.tobytes(), ctx=ctx), dtype=np.uint64).copy()

    compress = evaluate
