"""GPU: the shard kernel of the multi-GPU commit (pk_rs_encode_shard): the G shards, interleaved, must equal the unsharded
encode bit for bit, and the ShardedCommitter (world of 1 on this box, and G simulated ranks run one after another on the
same GPU) must reproduce the unsharded Merkle root."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("batch,n_vars,rho,G", [(2, 10, 1, 2), (2, 10, 1, 4), (2, 13, 1, 8), (1, 12, 4, 2), (2, 17, 1, 8), (1, 9, 7, 4), (2, 8, 1, 16)])
def test_shards_interleave_to_unsharded_encode(ctx, oracle, batch, n_vars, rho, G):
    from provekit_amd._lib import lib
    from provekit_amd.field import random_field

    fold = 4
    rows, w = 1 << (n_vars + rho - fold), batch << fold
    polys = [ctx.upload(random_field(1 << n_vars, 5 * b + n_vars)) for b in range(batch)]
    ptrs = (C.c_void_p * batch)(*[p.ptr for p in polys])
    full = ctx.alloc_fe(rows * w)
    scratch = ctx.alloc_fe(2 * rows * w)
    ctx._check(lib.pk_rs_encode(ctx.handle, ptrs, batch, n_vars, rho, fold, full.ptr, scratch.ptr))
    ref = ctx.download(full, (w, rows, 4))
    loc_rows = rows // G
    local = ctx.alloc_fe(w * loc_rows)
    sc2 = ctx.alloc_fe(w * (rows + 2 * loc_rows))
    for g in range(G):
        ctx._check(lib.pk_rs_encode_shard(ctx.handle, ptrs, batch, n_vars, rho, fold, g, G, local.ptr, sc2.ptr))
        got = ctx.download(local, (w, loc_rows, 4))
        assert np.array_equal(got, ref[:, g::G]), (g, G)


def test_sharded_committer_simulated_ranks(ctx, oracle):
    import torch

    from provekit_amd.distributed import HipShardBackend, ShardedCommitter
    from provekit_amd.field import random_field
    from provekit_amd.whir import commit_batch

    n_vars, G = 12, 4
    host = [random_field(1 << n_vars, 90 + b) for b in range(2)]
    polys = [ctx.upload(p) for p in host]
    expect = commit_batch(ctx, polys, n_vars).root
    be = HipShardBackend(ctx)
    try:
        # world of 1: the real code path end to end
        root, nodes, _ = ShardedCommitter(be, rank=0, world=1).commit(polys, n_vars)
        assert root.tobytes() == expect
        # G ranks simulated sequentially: gather by hand what all_gather would deliver
        rows = 1 << (n_vars + 1 - 4)
        digs = []
        for g in range(G):
            _, d = be.encode_and_hash_shard(polys, n_vars, 1, 4, g, G)
            with be.stream_ctx():
                digs.append(d.clone())
        with be.stream_ctx():
            nodes = be.new_nodes(rows)
            nodes[rows:] = torch.stack(digs, dim=1).reshape(rows, 4)
            be.merkle_inner(nodes, rows)
            assert nodes[1].cpu().numpy().view(np.uint64).tobytes() == expect
    finally:
        ctx.set_stream(None)


def test_sharded_open_matches_unsharded(ctx, oracle):
    """ShardedCommitter.open (world 1 on this box): the same triple Commitment.open returns; and pk_gather_leaves on a
    real shard (rank g of 4) returns rows g, g+4, ... of the unsharded codeword."""
    from provekit_amd._lib import PK_COL_MAJOR, lib
    from provekit_amd.distributed import HipShardBackend, ShardedCommitter
    from provekit_amd.field import random_field
    from provekit_amd.whir import commit_batch

    n_vars, G = 11, 4
    polys = [ctx.upload(random_field(1 << n_vars, 190 + b)) for b in range(2)]
    ref = commit_batch(ctx, polys, n_vars)
    rows = ref.n_leaves
    idx = np.array([0, 3, 4, 77, rows // 2 + 1, rows - 1], dtype=np.uint64)
    lv_ref, sib_ref, paths_ref = ref.open(idx, canonical_leaves=False)
    be = HipShardBackend(ctx)
    try:
        sc = ShardedCommitter(be, rank=0, world=1)
        root, nodes, local = sc.commit(polys, n_vars)
        assert root.tobytes() == ref.root
        lv, sib, paths = sc.open(idx, local, nodes, 32)
        assert np.array_equal(lv, lv_ref) and np.array_equal(sib, sib_ref) and np.array_equal(paths, paths_ref)
        all_rows = np.arange(rows, dtype=np.uint64)
        full, _, _ = ref.open(all_rows, canonical_leaves=False)
        for g in range(G):
            shard, _ = be.encode_and_hash_shard(polys, n_vars, 1, 4, g, G)
            want = np.arange(g, rows, G)
            local_rows = np.array([1, 0, rows // G - 1, 5], dtype=np.uint64)
            out = np.zeros((len(local_rows), 32, 4), dtype=np.uint64)
            ctx._check(lib.pk_gather_leaves(ctx.handle, shard.ptr, rows // G, 32, PK_COL_MAJOR, local_rows.ctypes.data, len(local_rows), 0, out.ctypes.data))
            assert np.array_equal(out, full[want[local_rows.astype(np.int64)]])
            be.release(shard)
    finally:
        ctx.set_stream(None)
    ref.close()


def test_bench_multirank_control_flow_on_one_gpu(ctx, oracle):
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one process per rank), with
    PK_BENCH_ONE_GPU=1 putting both ranks on GPU 0 over gloo: the sharded commit's root must be the unsharded root, and the
    prove workload must aggregate over the ranks and print exactly one JSON line."""
    import json
    import os
    import subprocess
    import sys

    import torch

    from provekit_amd.whir import commit_batch

    import socket

    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PK_BENCH_ONE_GPU="1", MASTER_ADDR="127.0.0.1")

    def free_port():
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            return sk.getsockname()[1]

    def launch(extra, port):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
               str(port), os.path.join(root_dir, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"] + extra
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, lines  # one JSON line on stdout, whatever the libraries print
        return json.loads(lines[0])

    m = 15
    d = launch(["--workload", "commit", "--log2-size", str(m)], free_port())
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    polys = []
    for b in range(2):  # the seeded coefficients bench.py's commit workload generates
        t = torch.randint(0, 2**62, (1 << m, 4), dtype=torch.int64, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(17 + b))
        t[:, 3] &= (1 << 60) - 1
        polys.append(t)
    torch.cuda.synchronize()
    ref = commit_batch(ctx, [int(t.data_ptr()) for t in polys], m)
    assert d["config"]["root"] == ref.root.hex()
    ref.close()
    p = launch(["--workload", "prove", "--log2-size", "13", "--concurrency", "2", "--no-cpu-baseline"], free_port())
    assert p["n_gpus"] == 2 and p["steps"] == 2 and p["scaling"] == "weak" and p["value"] > 0
    # whole-job aggregate: one step = one wave of `concurrency` proofs per GPU -> ranks x concurrency x steps proofs / time
    assert p["config"]["proofs_per_step"] == 2 * 2
    assert abs(p["value"] - 2 * 2 * 2 / (p["ms_per_step"] * 2 * 1e-3)) / p["value"] < 1e-6
