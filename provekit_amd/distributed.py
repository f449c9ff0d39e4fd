"""One WHIR commit sharded over the GPUs of a node.

Rank g of G owns the codeword rows (Merkle leaves) i = g + G*t: it runs the encode shard and hashes its own leaves
with no communication, then ONE collective -- an all-gather of the 32-byte leaf digests (RCCL over xGMI when the
process group is "nccl"; 32*rows bytes in total: 8 MiB at the poseidon size, 256 MiB at 2^26) -- after which every
rank builds the (cheap) inner tree redundantly and holds the root.  Openings of leaf i are served by rank i mod G.
The collective and the digest interleave are backend-agnostic torch code, so the N>1 path is exercised on CPU with
gloo (tests/test_distributed_cpu.py); the compute backend below is the HIP library and has no CPU fallback.

Process note: PyTorch wheels bundle their own HIP runtime.  In a process that uses both, import torch (and touch
torch.cuda) BEFORE creating a provekit_amd.Context so that one runtime serves both; bench.py does.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from ._lib import PK_COL_MAJOR, lib
from .runtime import Context, DeviceBuffer


class HipShardBackend:
    """Per-rank compute over libprovekit_hip.  Work is enqueued on torch's current stream so that the collective that
    follows is ordered after the kernels without a host synchronisation."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        if not torch.cuda.is_available():
            raise RuntimeError("torch sees no GPU: initialise torch.cuda before creating the provekit_amd Context (see module docstring)")
        self.device = torch.device("cuda", ctx.device)
        # one explicit (non-default) stream shared by the library's kernels and torch's ops / the collective
        self.stream = torch.cuda.Stream(device=self.device)
        ctx.set_stream(self.stream.cuda_stream)
        self._scratch = None  # grow-only encode workspace
        self._pool = {}       # released leaf-shard buffers by size: a steady stream of commits never calls hipMalloc

    def release(self, leaves: DeviceBuffer):
        """hand a leaf shard returned by commit() back for reuse (callers that keep it for openings simply do not)"""
        self._pool.setdefault(leaves.nbytes, []).append(leaves)

    def encode_and_hash_shard(self, d_polys, n_vars, log_inv_rate, fold, shard, n_shards):
        ctx = self.ctx
        batch = len(d_polys)
        rows = 1 << (n_vars + log_inv_rate - fold)
        width = batch << fold
        local_rows = rows // n_shards
        free = self._pool.get(32 * width * local_rows)
        leaves = free.pop() if free else ctx.alloc_fe(width * local_rows)
        need = 32 * width * (rows + 2 * local_rows)
        if self._scratch is None or self._scratch.nbytes < need:
            self._scratch = None
            self._scratch = ctx.alloc(need)
        scratch = self._scratch
        ptrs = (C.c_void_p * batch)(*[p.ptr if isinstance(p, DeviceBuffer) else p for p in d_polys])
        ctx._check(lib.pk_rs_encode_shard(ctx.handle, ptrs, batch, n_vars, log_inv_rate, fold, shard, n_shards, leaves.ptr, scratch.ptr))
        with torch.cuda.stream(self.stream):
            digests = torch.empty((local_rows, 4), dtype=torch.int64, device=self.device)
        ctx._check(lib.pk_leaf_hash(ctx.handle, leaves.ptr, local_rows, width, PK_COL_MAJOR, digests.data_ptr()))
        return leaves, digests  # the caller owns the shard of the codeword matrix (resident for openings)

    def gather_local_leaves(self, leaves, local_rows_total, width, local_rows):
        """rows `local_rows` of this rank's shard (column-major on the device) -> (len, width, 4) int64 tensor on the device"""
        idx = np.ascontiguousarray(local_rows, dtype=np.uint64)
        out = np.zeros((len(idx), width, 4), dtype=np.uint64)
        if len(idx):
            self.ctx._check(lib.pk_gather_leaves(self.ctx.handle, leaves.ptr, local_rows_total, width, PK_COL_MAJOR, idx.ctypes.data, len(idx), 0,
                                                 out.ctypes.data))
        with torch.cuda.stream(self.stream):
            return torch.from_numpy(out.view(np.int64)).to(self.device)

    def new_nodes(self, rows):
        with torch.cuda.stream(self.stream):
            return torch.zeros((2 * rows, 4), dtype=torch.int64, device=self.device)

    def merkle_inner(self, nodes, rows):
        self.ctx._check(lib.pk_merkle_inner(self.ctx.handle, nodes.data_ptr(), rows))

    def stream_ctx(self):
        return torch.cuda.stream(self.stream)


class ShardedCommitter:
    def __init__(self, backend, rank: int | None = None, world: int | None = None, group=None):
        self.backend, self.group = backend, group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        if self.world & (self.world - 1):
            raise ValueError("the number of shards must be a power of two")

    def commit(self, d_polys, n_vars: int, log_inv_rate: int = 1, fold: int = 4):
        """-> (root (4,) uint64 canonical, nodes tensor (2*rows, 4), this rank's leaf shard)"""
        import contextlib

        rows = 1 << (n_vars + log_inv_rate - fold)
        leaves_local, dig_local = self.backend.encode_and_hash_shard(d_polys, n_vars, log_inv_rate, fold, self.rank, self.world)
        scope = self.backend.stream_ctx() if hasattr(self.backend, "stream_ctx") else contextlib.nullcontext()
        with scope:  # torch ops and the collective run on the backend's stream, after its kernels
            if self.world > 1:
                gathered = [torch.empty_like(dig_local) for _ in range(self.world)]
                dist.all_gather(gathered, dig_local, group=self.group)
            else:
                gathered = [dig_local]
            nodes = self.backend.new_nodes(rows)
            # leaf i = g + G*t is gathered[g][t]: stack on a new axis 1 and flatten -> row i
            nodes[rows:] = torch.stack(gathered, dim=1).reshape(rows, 4)
            self.backend.merkle_inner(nodes, rows)
            root = nodes[1].cpu().numpy().view(np.uint64).copy()
        return root, nodes, leaves_local

    def owner_of_leaf(self, i: int) -> tuple[int, int]:
        """(rank, local row) that holds leaf i"""
        return i % self.world, i // self.world

    def open(self, indices, leaves_local, nodes, width: int):
        """STIR openings of a sharded commitment (SURVEY 8e "Openings"): leaf i is served by rank i mod G -- every rank
        gathers the rows it owns and ONE all-reduce (each row is non-zero on exactly one rank; k*width*32 bytes, ~100 KiB)
        gives everybody the opened leaves; sibling digests and auth paths come from the replicated inner tree.
        -> (leaves (k, width, 4) uint64 Montgomery, sibling digests (k, 4), auth paths root->leaf (k, log2(rows)-1, 4)),
        the same triple Commitment.open(canonical_leaves=False) returns for an unsharded tree."""
        import contextlib

        idx = [int(i) for i in indices]
        rows = nodes.shape[0] // 2
        logn = rows.bit_length() - 1
        mine = [(q, i // self.world) for q, i in enumerate(idx) if i % self.world == self.rank]
        got = self.backend.gather_local_leaves(leaves_local, rows // self.world, width, [r for _, r in mine])
        scope = self.backend.stream_ctx() if hasattr(self.backend, "stream_ctx") else contextlib.nullcontext()
        with scope:
            full = torch.zeros((len(idx), width, 4), dtype=torch.int64, device=nodes.device)
            if mine:
                full[torch.tensor([q for q, _ in mine], device=nodes.device)] = got.to(nodes.device)
            if self.world > 1:
                dist.all_reduce(full, op=dist.ReduceOp.SUM, group=self.group)
            pos = torch.tensor(idx, dtype=torch.int64, device=nodes.device) + rows
            sib = nodes[pos ^ 1]
            paths = torch.stack([nodes[(pos >> (logn - d)) ^ 1] for d in range(1, logn)], dim=1) if logn > 1 else torch.zeros(
                (len(idx), 0, 4), dtype=torch.int64, device=nodes.device)
            to_np = lambda t: t.cpu().numpy().view(np.uint64)
            return to_np(full), to_np(sib), to_np(paths)
