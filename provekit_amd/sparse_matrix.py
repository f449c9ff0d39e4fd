"""R1CS sparse matrices on the device, mirroring provekit_common::{SparseMatrix, R1CS}
(provekit/common/src/sparse_matrix.rs, r1cs.rs) and the two products of utils/sumcheck.rs."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from ._lib import SparseMatrixStruct, lib
from .runtime import Context, DeviceBuffer


@dataclass
class SparseMatrix:
    """Same fields as the reference struct (sparse_matrix.rs:12-27); values index the interner."""
    num_rows: int
    num_cols: int
    new_row_indices: np.ndarray
    col_indices: np.ndarray
    values: np.ndarray

    def __post_init__(self):
        # The C ABI takes bare pointers and reads new_row_indices[0..num_rows) and col_indices/values[0..nnz): the array
        # lengths (which C cannot see) and the u32 ranges are checked here, the contents again in pk_r1cs_create.
        for name in ("new_row_indices", "col_indices", "values"):
            a = np.asarray(getattr(self, name))
            if a.ndim != 1 or (a.size and (a.min() < 0 or a.max() >= 1 << 32)):
                raise ValueError(f"SparseMatrix.{name} must be a 1-D array of u32 values")
            setattr(self, name, np.ascontiguousarray(a, dtype=np.uint32))
        if self.new_row_indices.shape[0] != self.num_rows:
            raise ValueError(f"new_row_indices has {self.new_row_indices.shape[0]} entries for {self.num_rows} rows")
        if self.values.shape[0] != self.col_indices.shape[0]:
            raise ValueError("values and col_indices differ in length")
        nnz = self.col_indices.shape[0]
        if self.num_rows and (np.any(np.diff(self.new_row_indices.astype(np.int64)) < 0) or int(self.new_row_indices[-1]) > nnz):
            raise ValueError("new_row_indices must be non-decreasing offsets into the entry arrays")
        if nnz and int(self.col_indices.max()) >= self.num_cols:
            raise ValueError("column index out of bounds")

    @property
    def nnz(self) -> int:
        return int(self.col_indices.shape[0])


class R1CS:
    """Device-resident R1CS {A, B, C} + interner (uploaded once per proof scheme)."""

    def __init__(self, ctx: Context, a: SparseMatrix, b: SparseMatrix, c: SparseMatrix, interner_mont: np.ndarray):
        self.ctx = ctx
        self.num_constraints, self.num_witnesses = a.num_rows, a.num_cols
        mats = (SparseMatrixStruct * 3)()
        self._keep = []
        for k, m in enumerate((a, b, c)):
            if (m.num_rows, m.num_cols) != (self.num_constraints, self.num_witnesses):
                raise ValueError("matrix shape mismatch")
            nri = np.ascontiguousarray(m.new_row_indices, dtype=np.uint32)
            ci = np.ascontiguousarray(m.col_indices, dtype=np.uint32)
            vv = np.ascontiguousarray(m.values, dtype=np.uint32)
            self._keep += [nri, ci, vv]
            mats[k] = SparseMatrixStruct(nri.ctypes.data, ci.ctypes.data, vv.ctypes.data, ci.shape[0])
        it = np.ascontiguousarray(interner_mont, dtype=np.uint64).reshape(-1, 4)
        h = C.c_void_p()
        ctx._check(lib.pk_r1cs_create(ctx.handle, self.num_constraints, self.num_witnesses, C.cast(mats, C.c_void_p), it.ctypes.data,
                                      it.shape[0], C.byref(h)))
        self.handle = h.value

    @classmethod
    def from_postcard(cls, ctx: Context, data: bytes) -> "R1CS":
        """upload from postcard(R1CS), the bytes a Rust caller gets from `postcard::to_allocvec(&scheme.r1cs)`
        (pk_r1cs_from_postcard; layout in provekit_amd/file.py)"""
        self = cls.__new__(cls)
        self.ctx, self._keep = ctx, []
        h = C.c_void_p()
        nc, nw, npub, used = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        ctx._check(lib.pk_r1cs_from_postcard(ctx.handle, buf, len(data), C.byref(h), C.byref(nc), C.byref(nw), C.byref(npub), C.byref(used)))
        self.handle = h.value
        self.num_constraints, self.num_witnesses, self.num_public_inputs, self.bytes_consumed = nc.value, nw.value, npub.value, used.value
        return self

    def close(self):
        if self.handle is not None and self.ctx.handle is not None:
            lib.pk_r1cs_destroy(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def calculate_witness_bounds(self, d_z, m0: int):
        """sumcheck.rs:181-193 -> (a, b, c) device buffers of 2^m0 FEs"""
        n = 1 << m0
        a, b, c = (self.ctx.alloc_fe(n) for _ in range(3))
        self.ctx._check(lib.pk_r1cs_witness_bounds(self.ctx.handle, self.handle, d_z.ptr if isinstance(d_z, DeviceBuffer) else d_z, m0, a.ptr, b.ptr, c.ptr))
        return a, b, c

    def calculate_external_row_of_r1cs_matrices(self, d_eq_alpha) -> DeviceBuffer:
        """sumcheck.rs:207-218 -> device buffer of 3*num_witnesses FEs: [eq^T A | eq^T B | eq^T C]"""
        out = self.ctx.alloc_fe(3 * max(self.num_witnesses, 1))
        self.ctx._check(lib.pk_r1cs_external_row(self.ctx.handle, self.handle, d_eq_alpha.ptr if isinstance(d_eq_alpha, DeviceBuffer) else d_eq_alpha, out.ptr))
        return out

    def matvec(self, matrix: int, d_x, transpose=False) -> DeviceBuffer:
        n_out = self.num_witnesses if transpose else self.num_constraints
        out = self.ctx.alloc_fe(max(n_out, 1))
        self.ctx._check(lib.pk_r1cs_matvec(self.ctx.handle, self.handle, matrix, int(transpose), d_x.ptr if isinstance(d_x, DeviceBuffer) else d_x, out.ptr))
        return out

    def test_witness_satisfaction(self, d_witness, n_witness: int | None = None):
        """R1CSSolver::test_witness_satisfaction (provekit/prover/src/r1cs.rs:41-60): raises ProveKitHipError
        ("Constraint {row} failed", .row = first failing row) unless (A z) o (B z) == C z."""
        from ._lib import ProveKitHipError

        row = C.c_int64(-1)
        n = self.num_witnesses if n_witness is None else n_witness
        try:
            self.ctx._check(lib.pk_r1cs_test_witness_satisfaction(self.ctx.handle, self.handle, d_witness.ptr if isinstance(d_witness, DeviceBuffer) else d_witness, n, C.byref(row)))
        except ProveKitHipError as e:
            e.row = row.value
            raise
