"""Commitment handle and openings: the data-parallel body of whir's CommitmentWriter::commit_batch and
MerkleTree::generate_multi_proof (call sites provekit/prover/src/whir_r1cs.rs:200-206)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import lib
from .runtime import Context, DeviceBuffer


class Commitment:
    """pk_tree handle: codeword matrix (column-major) + Merkle heap resident in HBM."""

    def __init__(self, ctx: Context, handle: int, root: bytes):
        self.ctx, self.handle, self.root = ctx, handle, root
        n, w, dl, dn = C.c_size_t(), C.c_size_t(), C.c_void_p(), C.c_void_p()
        lib.pk_tree_info(handle, C.byref(n), C.byref(w), C.byref(dl), C.byref(dn))
        self.n_leaves, self.width, self.d_leaves, self.d_nodes = n.value, w.value, dl.value, dn.value

    def close(self):
        if self.handle is not None and self.ctx.handle is not None:
            lib.pk_tree_destroy(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def open(self, indices, canonical_leaves=True):
        """-> (leaves (k,width,4), sibling_digests (k,4), auth_paths (k, log2(n)-1, 4) root->leaf)"""
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
        k = idx.shape[0]
        plen = max(self.n_leaves.bit_length() - 2, 0)
        leaves = np.empty((k, self.width, 4), dtype=np.uint64)
        sib = np.zeros((k, 4), dtype=np.uint64)
        paths = np.empty((k, plen, 4), dtype=np.uint64)
        self.ctx._check(lib.pk_tree_open(self.ctx.handle, self.handle, idx.ctypes.data, k, int(canonical_leaves), leaves.ctypes.data,
                                         sib.ctypes.data, paths.ctypes.data))
        return leaves, sib, paths


def commit_batch(ctx: Context, d_polys, n_vars: int, log_inv_rate: int = 1, fold: int = 4) -> Commitment:
    ptrs = [p.ptr if isinstance(p, DeviceBuffer) else p for p in d_polys]
    arr = (C.c_void_p * len(ptrs))(*ptrs)
    root = (C.c_uint8 * 32)()
    h = C.c_void_p()
    ctx._check(lib.pk_commit(ctx.handle, arr, len(ptrs), n_vars, log_inv_rate, fold, root, C.byref(h)))
    return Commitment(ctx, h.value, bytes(root))


def tree_from_leaves(ctx: Context, d_leaves, n_leaves: int, width: int, layout: int) -> Commitment:
    root = (C.c_uint8 * 32)()
    h = C.c_void_p()
    ctx._check(lib.pk_tree_from_leaves(ctx.handle, d_leaves.ptr if isinstance(d_leaves, DeviceBuffer) else d_leaves, n_leaves, width,
                                       layout, root, C.byref(h)))
    c = Commitment(ctx, h.value, bytes(root))
    c._keep = d_leaves
    return c


def multipath_serialize(indices, sibling_digests: np.ndarray, auth_paths: np.ndarray) -> bytes:
    idx = np.ascontiguousarray(indices, dtype=np.uint64)
    sib = np.ascontiguousarray(sibling_digests, dtype=np.uint64).reshape(-1, 4)
    k = idx.shape[0]
    paths = np.ascontiguousarray(auth_paths, dtype=np.uint64).reshape(k, -1, 4) if k else np.zeros((0, 0, 4), np.uint64)
    plen = paths.shape[1]
    n = C.c_size_t(0)
    rc = lib.pk_multipath_serialize(idx.ctypes.data, k, plen, sib.ctypes.data, paths.ctypes.data, None, 0, C.byref(n))
    if rc:
        raise ValueError("pk_multipath_serialize failed")
    buf = (C.c_uint8 * max(n.value, 1))()
    rc = lib.pk_multipath_serialize(idx.ctypes.data, k, plen, sib.ctypes.data, paths.ctypes.data, buf, n.value, C.byref(n))
    if rc:
        raise ValueError("pk_multipath_serialize failed")
    return bytes(buf)[: n.value]


def smoke(ctx: Context, oracle):
    """tiny commit checked against the oracle (used by __graft_entry__.smoke)"""
    from .field import random_field

    n = 10
    polys = [random_field(1 << n, 40 + b) for b in range(2)]
    c = commit_batch(ctx, [ctx.upload(p) for p in polys], n)
    leaves = oracle.rs_encode(np.concatenate(polys), 2, n, 1, 4)
    assert oracle.limbs_to_ints(oracle.merkle_commit(leaves)[1])[0] == int.from_bytes(c.root, "little"), "commit root mismatch"
    c.close()
