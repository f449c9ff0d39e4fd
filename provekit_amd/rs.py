"""Reed-Solomon encode / NTT host wrappers (rows N1, N2)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import lib
from .runtime import Context, default_context


def ntt(data_mont: np.ndarray, ctx: Context | None = None) -> np.ndarray:
    """(ncols, N, 4) or (N, 4) Montgomery FEs -> same shape, natural-order NTT of each vector."""
    ctx = ctx or default_context()
    a = np.ascontiguousarray(data_mont, dtype=np.uint64)
    single = a.ndim == 2
    if single:
        a = a[None]
    ncols, n = a.shape[0], a.shape[1]
    log_n = n.bit_length() - 1
    if n != 1 << log_n:
        raise ValueError("NTT size must be a power of two")
    d_in = ctx.upload(a)
    d_out = ctx.alloc_fe(ncols * n)
    ctx._check(lib.pk_ntt(ctx.handle, d_in.ptr, d_out.ptr, log_n, ncols))
    out = ctx.download(d_out, (ncols, n, 4))
    return out[0] if single else out


def rs_encode_device(ctx: Context, d_polys: list[int], n_vars: int, log_inv_rate: int, fold: int, d_leaves: int, d_scratch: int):
    arr = (C.c_void_p * len(d_polys))(*d_polys)
    ctx._check(lib.pk_rs_encode(ctx.handle, arr, len(d_polys), n_vars, log_inv_rate, fold, d_leaves, d_scratch))


def rs_encode(coeffs_mont: np.ndarray, n_vars: int, log_inv_rate: int, fold: int, ctx: Context | None = None) -> np.ndarray:
    """(batch, 2^n_vars, 4) coefficient vectors -> leaf-major (rows, batch*2^fold, 4) codeword matrix
    (the GPU keeps it column-major; the transpose here is host-side convenience for tests)."""
    ctx = ctx or default_context()
    c = np.ascontiguousarray(coeffs_mont, dtype=np.uint64)
    batch = c.shape[0]
    rows = 1 << (n_vars + log_inv_rate - fold)
    w = batch << fold
    bufs = [ctx.upload(c[b]) for b in range(batch)]
    d_leaves = ctx.alloc_fe(rows * w)
    d_scratch = ctx.alloc_fe(2 * rows * w)
    rs_encode_device(ctx, [b.ptr for b in bufs], n_vars, log_inv_rate, fold, d_leaves.ptr, d_scratch.ptr)
    cols = ctx.download(d_leaves, (w, rows, 4))
    return np.ascontiguousarray(cols.transpose(1, 0, 2))
