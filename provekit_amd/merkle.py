"""Merkle tree over Skyscraper, mirroring ark_crypto_primitives::merkle_tree::MerkleTree
as configured by SkyscraperMerkleConfig (provekit/common/src/skyscraper/whir.rs:79-86)."""
from __future__ import annotations

import numpy as np

from ._lib import PK_COL_MAJOR, PK_LEAF_MAJOR, lib
from .runtime import Context, default_context


class MerkleTree:
    """MerkleTree::new(leaf_hash_params, two_to_one_params, leaves): all node digests stay
    resident in HBM as a heap (nodes[1] = root; leaf digest i at nodes[n + i]), canonical."""

    def __init__(self, leaves_mont: np.ndarray | None = None, ctx: Context | None = None, *, d_leaves: int | None = None,
                 n_leaves: int | None = None, width: int | None = None, layout: int = PK_LEAF_MAJOR):
        self.ctx = ctx or default_context()
        if leaves_mont is not None:
            leaves = np.ascontiguousarray(leaves_mont, dtype=np.uint64)
            n_leaves, width = leaves.shape[0], leaves.shape[1]
            self._leaf_buf = self.ctx.upload(leaves)
            d_leaves = self._leaf_buf.ptr
        if n_leaves is None or n_leaves < 1 or n_leaves & (n_leaves - 1):
            raise ValueError("number of leaves must be a power of two")  # ark MerkleTree::new
        self.n_leaves, self.width, self.layout = n_leaves, width, layout
        self.d_leaves = d_leaves
        self.nodes = self.ctx.alloc_fe(2 * n_leaves)
        self.ctx._check(lib.pk_merkle_commit(self.ctx.handle, d_leaves, n_leaves, width, layout, self.nodes.ptr))

    def root(self) -> np.ndarray:
        return self.ctx.download_fe(self.nodes.view_fe(1), 1)[0]

    def all_nodes(self) -> np.ndarray:
        return self.ctx.download_fe(self.nodes, 2 * self.n_leaves)

    def height(self) -> int:
        return self.n_leaves.bit_length()
