"""CPU, world_size 2, gloo: the N>1 exchange of the sharded commit and of its openings.

The sharded commit lives behind the C ABI and needs a GPU for its compute; what can run here is everything around the
kernels, and it is the PRODUCT's code, not a stand-in: the host transport the library calls back into
(provekit_amd.device_set.HostTransport over a gloo group -- the transport `pk_comm_init_host` takes) and the library's own
leaf-index shard map (csrc/shard_map.hpp through pk_shard_of_leaf / pk_shard_interleave_digests, the same functions the device
kernels are compiled from).  Per-rank compute (encode + leaf hash of the rows a rank owns) is done by the CPU oracle.  The
GPU suite runs the same exchange through pk_commit / pk_tree_open / pk_prove (tests/test_gpu_sharded.py, and across two
processes in tests/test_gpu_distributed.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPEN_IDX = [0, 1, 2, 7, 20, 33, 62, 63]  # owners alternate between the two ranks


def _worker(rank, world, port, n_vars, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as o
    from provekit_amd.device_set import HostTransport, interleave_digests, max_over_ranks, owner_of_leaf
    from provekit_amd.field import random_field

    polys = [random_field(1 << n_vars, 70 + b) for b in range(2)]  # every rank holds the full coefficient vectors
    full = o.rs_encode(np.concatenate(polys), 2, n_vars, 1, 4)  # (rows, width, 4)
    rows, width = full.shape[:2]
    # what rank g computes with no communication: the rows it owns and their digests
    mine = [i for i in range(rows) if owner_of_leaf(i, world)[0] == rank]
    assert [owner_of_leaf(i, world)[1] for i in mine] == list(range(rows // world))  # local rows are dense and ordered
    local = np.ascontiguousarray(full[mine])
    dig_local = o.leaf_hash(local)
    ht = HostTransport(dist)
    # the callback the library invokes (pk_host_all_gather_fn), called the way the library calls it: raw pointers
    send = np.ascontiguousarray(dig_local).view(np.uint8).reshape(-1)
    recv = np.zeros(world * send.size, dtype=np.uint8)
    assert ht.callback(None, send.ctypes.data, recv.ctypes.data, send.size) == 0 and ht.error is None
    gathered = recv.view(np.uint64).reshape(world, rows // world, 4)
    nodes = interleave_digests(gathered, world)  # the library's shard map places block r's digests in the leaf layer
    nodes = o.merkle_inner(nodes[rows:])
    # openings: every rank contributes the opened rows it owns into a zeroed buffer; the sum over ranks is the gather
    opened = np.zeros((len(OPEN_IDX), width, 4), dtype=np.uint64)
    for k, i in enumerate(OPEN_IDX):
        r, row = owner_of_leaf(i, world)
        if r == rank:
            opened[k] = local[row]
    parts = ht.all_gather_bytes(opened.view(np.uint8).reshape(-1)).view(np.uint64).reshape(world, *opened.shape)
    opened = parts.sum(axis=0, dtype=np.uint64)
    slowest = max_over_ranks(1.0 + rank, dist)
    q.put((rank, nodes[1].tolist(), nodes[1:5].tolist(), len(mine), owner_of_leaf(5, world), opened.tolist(), slowest, ht.calls))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2])
def test_sharded_commit_exchange_gloo(oracle, world):
    n_vars = 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_vars, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from provekit_amd.field import random_field

    polys = [random_field(1 << n_vars, 70 + b) for b in range(2)]
    leaves = oracle.rs_encode(np.concatenate(polys), 2, n_vars, 1, 4)
    exp = oracle.merkle_commit(leaves)
    n = leaves.shape[0]
    for rank, root, top, n_local, owner, opened, slowest, calls in res:
        assert root == exp[1].tolist(), f"rank {rank} root mismatch"
        assert top == exp[1:5].tolist()
        assert n_local == n // world
        assert owner == (5 % world, 5 // world)
        assert np.array_equal(np.array(opened, dtype=np.uint64), leaves[OPEN_IDX]), f"rank {rank}: opened leaves"
        assert slowest == float(world) and calls == 2


def test_shard_map_host_entry_points():
    """pk_shard_of_leaf / pk_shard_interleave_digests (csrc/shard_map.hpp compiled for the host)"""
    from provekit_amd.device_set import interleave_digests, owner_of_leaf

    for G in (1, 2, 4, 8, 16):
        rows = 64
        assert [owner_of_leaf(i, G) for i in range(rows)] == [(i % G, i // G) for i in range(rows)]
        blocks = np.zeros((G, rows // G, 4), dtype=np.uint64)
        for g in range(G):
            for t in range(rows // G):
                blocks[g, t] = [g + G * t, 1, 2, 3]  # the digest of leaf g + G t, tagged with its leaf index
        nodes = interleave_digests(blocks, G)
        assert np.array_equal(nodes[rows:, 0], np.arange(rows, dtype=np.uint64)) and not nodes[:rows].any()
    with pytest.raises(ValueError):
        owner_of_leaf(3, 3)
