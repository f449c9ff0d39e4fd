"""Joining the ranks of a launcher (torch.distributed.run: one process per GPU) into the library's device set.

The sharded commit / opening / proof itself lives behind the C ABI (csrc/comm.hip, tree.hip, prover.hip; include/provekit_hip.h
"device sets"): a context that carries a communicator shards by leaf index.  What is left above the ABI is the rendezvous:

  * transport "rccl": rank 0 makes the 128-byte RCCL unique id, the launcher's process group broadcasts it, every rank calls
    pk_comm_init_rank.  Collectives then run on the context's stream over xGMI.  One rank per GPU.
  * transport "host": the library's bring-your-own-transport hook (pk_comm_init_host) with torch.distributed's all_gather on
    CPU tensors as the callback -- gloo, or any backend that moves host bytes.  Slower (two PCIe hops per collective) but it
    works wherever a host collective does, in particular with several ranks on ONE GPU, where RCCL refuses to form a
    communicator: that is how the multi-process launch is exercised on a single-GPU box (PK_BENCH_ONE_GPU=1) and how the CPU
    suite drives the exchange (tests/test_distributed_cpu.py).

The leaf-index shard map (who owns leaf i, where an all-gather's blocks land in the node heap) is the library's own
(csrc/shard_map.hpp, compiled for device and host); `owner_of_leaf` / `interleave_digests` below are its host entry points.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import lib

_HOST_ALL_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


class HostTransport:
    """pk_host_all_gather_fn over a torch.distributed process group whose backend moves CPU tensors (gloo).  Keep the object
    alive as long as the context's communicator: the library holds a pointer to its callback."""

    def __init__(self, dist, group=None):
        import torch

        self.dist, self.group, self.torch = dist, group, torch
        self.world = dist.get_world_size(group)
        self.calls = 0
        self.error = None
        self.callback = _HOST_ALL_GATHER(self._all_gather)

    def all_gather_bytes(self, send: np.ndarray) -> np.ndarray:
        """(n,) uint8 on every rank -> (world, n) uint8 on every rank, block r = rank r's send"""
        t = self.torch.from_numpy(np.ascontiguousarray(send, dtype=np.uint8))
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        self.calls += 1
        return np.stack([o.numpy() for o in out])

    def _all_gather(self, _user, send, recv, nbytes):
        try:
            src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,)) if nbytes else np.zeros(0, np.uint8)
            got = self.all_gather_bytes(src)
            if nbytes:
                np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(self.world * nbytes,))[:] = got.reshape(-1)
            return 0
        except BaseException as e:  # noqa: BLE001 -- nothing may propagate into the C caller
            self.error = e
            return 1


def join_device_set(ctx, rank: int, world: int, dist, transport: str = "rccl"):
    """make `ctx` rank `rank` of the run's `world` ranks; returns the transport object to keep alive (None for RCCL)"""
    if world <= 1:
        return None
    if transport == "rccl":
        import provekit_amd

        box = [provekit_amd.Context.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ctx.comm_init_rank(box[0], world, rank)
        return None
    if transport == "host":
        ht = HostTransport(dist)
        ctx._check(lib.pk_comm_init_host(ctx.handle, world, rank, ht.callback, None))
        return ht
    raise ValueError(f"unknown transport {transport!r}")


def max_over_ranks(seconds: float, dist, device=None) -> float:
    """the launcher contract's clock: the slowest rank's time"""
    if dist is None:
        return seconds
    import torch

    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def owner_of_leaf(i: int, n_shards: int) -> tuple[int, int]:
    """(rank, local row) holding leaf i of a commit sharded over n_shards ranks (pk_shard_of_leaf)"""
    r, row = C.c_uint(), C.c_uint64()
    rc = lib.pk_shard_of_leaf(int(i), int(n_shards), C.byref(r), C.byref(row))
    if rc:
        raise ValueError("n_shards must be a power of two")
    return r.value, row.value


def interleave_digests(gathered: np.ndarray, n_shards: int) -> np.ndarray:
    """an all-gather's output ((n_shards, rows / n_shards, 4) uint64: block r = rank r's local digests) -> the node heap
    (2 * rows, 4) with its leaf layer filled (pk_shard_interleave_digests); the inner levels are left zero"""
    g = np.ascontiguousarray(gathered, dtype=np.uint64).reshape(-1, 4)
    rows = g.shape[0]
    nodes = np.zeros((2 * rows, 4), dtype=np.uint64)
    rc = lib.pk_shard_interleave_digests(g.ctypes.data, rows, int(n_shards), nodes.ctypes.data)
    if rc:
        raise ValueError("rows and n_shards must be powers of two")
    return nodes
