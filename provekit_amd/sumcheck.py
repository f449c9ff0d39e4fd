"""MLE / sumcheck kernels, named after provekit/common/src/utils/sumcheck.rs and the whir calls they replace.
All arrays are device-resident (DeviceBuffer / raw device pointers); scalars cross as (4,) uint64 Montgomery."""
from __future__ import annotations

import numpy as np

from ._lib import lib
from .runtime import Context, DeviceBuffer


def _ptr(x):
    return x.ptr if isinstance(x, DeviceBuffer) else x


def _fe(x):
    a = np.ascontiguousarray(x, dtype=np.uint64)
    return a


def to_coeffs(ctx: Context, d_evals, n_vars: int):
    """EvaluationsList::to_coeffs, in place"""
    ctx._check(lib.pk_to_coeffs(ctx.handle, _ptr(d_evals), n_vars))


def to_evals(ctx: Context, d_coeffs, n_vars: int):
    ctx._check(lib.pk_to_evals(ctx.handle, _ptr(d_coeffs), n_vars))


def calculate_evaluations_over_boolean_hypercube_for_eq(ctx: Context, r: np.ndarray, d_out=None) -> DeviceBuffer:
    """sumcheck.rs:146-153"""
    r = _fe(r).reshape(-1, 4)
    m = r.shape[0]
    d_out = d_out or ctx.alloc_fe(1 << m)
    ctx._check(lib.pk_eq_table(ctx.handle, r.ctypes.data, m, _ptr(d_out)))
    return d_out


def eq_accumulate(ctx: Context, d_w, n_vars: int, points: np.ndarray, scales: np.ndarray, overwrite=False):
    points = _fe(points).reshape(-1, max(n_vars, 1), 4) if n_vars else _fe(points)
    scales = _fe(scales).reshape(-1, 4)
    q = scales.shape[0]
    ctx._check(lib.pk_eq_accumulate(ctx.handle, _ptr(d_w), n_vars, points.ctypes.data, scales.ctypes.data, q, int(overwrite)))


def sumcheck_fold_map_reduce(ctx: Context, d_a, d_b, d_c, d_eq, length: int, fold=None) -> np.ndarray:
    """sumcheck_fold_map_reduce([a,b,c,eq], fold, cubic map) -> (3,4) [f(0), f(-1), f_inf]; folds in place."""
    out = np.empty((3, 4), dtype=np.uint64)
    f = _fe(fold) if fold is not None else None
    ctx._check(lib.pk_sumcheck_cubic_round(ctx.handle, _ptr(d_a), _ptr(d_b), _ptr(d_c), _ptr(d_eq), length,
                                           f.ctypes.data if f is not None else None, out.ctypes.data))
    return out


def sumcheck_quadratic_round(ctx: Context, d_f, d_w, length: int, fold=None, d_f_out=None, d_w_out=None) -> np.ndarray:
    out = np.empty((3, 4), dtype=np.uint64)
    f = _fe(fold) if fold is not None else None
    ctx._check(lib.pk_sumcheck_quadratic_round(ctx.handle, _ptr(d_f), _ptr(d_w), length, f.ctypes.data if f is not None else None,
                                               _ptr(d_f_out) if d_f_out is not None else None,
                                               _ptr(d_w_out) if d_w_out is not None else None, out.ctypes.data))
    return out


def fold_pairs(ctx: Context, d_v, length: int, r, d_out):
    r = _fe(r)
    ctx._check(lib.pk_fold_pairs(ctx.handle, _ptr(d_v), length, r.ctypes.data, _ptr(d_out)))


def weighted_sum(ctx: Context, d_w, d_f, n: int) -> np.ndarray:
    """Weights::linear(w).weighted_sum(f)"""
    out = np.empty(4, dtype=np.uint64)
    ctx._check(lib.pk_dot(ctx.handle, _ptr(d_w), _ptr(d_f), n, out.ctypes.data))
    return out


def eval_univariate(ctx: Context, d_coeffs, n: int, z) -> np.ndarray:
    out = np.empty(4, dtype=np.uint64)
    z = _fe(z)
    ctx._check(lib.pk_eval_univariate(ctx.handle, _ptr(d_coeffs), n, z.ctypes.data, out.ctypes.data))
    return out


def fold_coeffs(ctx: Context, d_coeffs, n_vars: int, r: np.ndarray, d_out=None) -> DeviceBuffer:
    r = _fe(r).reshape(-1, 4)
    k = r.shape[0]
    d_out = d_out or ctx.alloc_fe(1 << (n_vars - k))
    ctx._check(lib.pk_fold_coeffs(ctx.handle, _ptr(d_coeffs), n_vars, r.ctypes.data, k, _ptr(d_out)))
    return d_out


def axpy(ctx: Context, d_y, beta, d_x, n: int):
    beta = _fe(beta)
    ctx._check(lib.pk_fe_axpy(ctx.handle, _ptr(d_y), beta.ctypes.data, _ptr(d_x), n))
