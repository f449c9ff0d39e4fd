"""CPU, world_size 2, gloo: the N>1 path of the sharded commit (provekit_amd/distributed.py).  The collective and the
digest interleave are the product's code; the per-rank compute backend is replaced by the CPU oracle here (there is no
GPU in this container) -- the HIP backend's shard kernel is parity-tested on the GPU in tests/test_gpu_distributed.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleShardBackend:
    """test stand-in: same interface as HipShardBackend, computed by oracle/pk_oracle.c on the CPU"""

    def __init__(self, oracle):
        self.o = oracle

    def encode_and_hash_shard(self, polys, n_vars, log_inv_rate, fold, shard, n_shards):
        full = self.o.rs_encode(np.concatenate(polys), len(polys), n_vars, log_inv_rate, fold)  # (rows, width, 4)
        local = np.ascontiguousarray(full[shard::n_shards])
        dig = self.o.leaf_hash(local)
        return local, torch.from_numpy(dig.view(np.int64).copy())

    def gather_local_leaves(self, leaves, local_rows_total, width, local_rows):
        assert leaves.shape[:2] == (local_rows_total, width)
        return torch.from_numpy(np.ascontiguousarray(leaves[list(local_rows)]).view(np.int64).reshape(len(local_rows), width, 4))

    def new_nodes(self, rows):
        return torch.zeros((2 * rows, 4), dtype=torch.int64)

    def merkle_inner(self, nodes, rows):
        arr = nodes.numpy().view(np.uint64)
        out = self.o.merkle_inner(arr[rows:])
        nodes.copy_(torch.from_numpy(out.view(np.int64)))


def _worker(rank, world, port, n_vars, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as o
    from provekit_amd.distributed import ShardedCommitter
    from provekit_amd.field import random_field

    polys = [random_field(1 << n_vars, 70 + b) for b in range(2)]  # every rank holds the full coefficient vectors
    sc = ShardedCommitter(OracleShardBackend(o))
    root, nodes, local = sc.commit(polys, n_vars)
    opened = sc.open(OPEN_IDX, local, nodes, 32)
    q.put((rank, root.tolist(), nodes[1:5].numpy().view(np.uint64).tolist(), local.shape[0], sc.owner_of_leaf(5), [a.tolist() for a in opened]))
    dist.barrier()
    dist.destroy_process_group()


OPEN_IDX = [0, 1, 2, 7, 20, 33, 62, 63]  # owners alternate between the two ranks


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2])
def test_sharded_commit_gloo(oracle, world):
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
    logn = n.bit_length() - 1
    for rank, root, top, n_local, owner, opened in res:
        assert root == exp[1].tolist(), f"rank {rank} root mismatch"
        assert top == exp[1:5].tolist()
        assert n_local == n // world
        assert owner == (5 % world, 5 // world)
        lv, sib, paths = (np.array(a, dtype=np.uint64) for a in opened)
        assert np.array_equal(lv, leaves[OPEN_IDX]), f"rank {rank}: opened leaves"
        for q, i in enumerate(OPEN_IDX):
            assert np.array_equal(sib[q], exp[(n + i) ^ 1])
            for d in range(1, logn):
                assert np.array_equal(paths[q, d - 1], exp[((n + i) >> (logn - d)) ^ 1])
