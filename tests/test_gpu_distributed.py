"""GPU: the shard kernel of the multi-GPU commit (pk_rs_encode_shard): the G shards, interleaved, must equal the unsharded
encode bit for bit; and bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one process per
rank) -- on this one-GPU box with both ranks on GPU 0 over the library's host transport (gloo), on a box with >= 2 GPUs
(skipped here) with one rank per GPU over RCCL."""
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


def _launch_bench(extra, one_gpu, nproc=2, timeout=900):
    import json
    import os
    import socket
    import subprocess
    import sys

    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("PK_BENCH_ONE_GPU", None)
    if one_gpu:
        env["PK_BENCH_ONE_GPU"] = "1"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port",
           str(port), os.path.join(root_dir, "bench.py"), "--gpus", str(nproc), "--steps", "2", "--warmup", "1"] + extra
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # one JSON line on stdout, whatever the libraries print
    return json.loads(lines[0])


def _check_multirank_bench(ctx, one_gpu):
    """the launcher contract at two ranks: sharded commit root == unsharded root; independent provers aggregate over the ranks;
    ONE proof sharded over the two ranks (--sharded) runs through join_device_set and prints one line"""
    import torch

    from provekit_amd.whir import commit_batch

    def launch(extra):
        return _launch_bench(extra, one_gpu)

    m = 15
    d = launch(["--workload", "commit", "--log2-size", str(m)])
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
    p = launch(["--workload", "prove", "--log2-size", "13", "--concurrency", "2", "--no-cpu-baseline"])
    assert p["n_gpus"] == 2 and p["steps"] == 2 and p["scaling"] == "weak" and p["value"] > 0
    # whole-job aggregate: one step = one wave of `concurrency` proofs per GPU -> ranks x concurrency x steps proofs / time
    assert p["config"]["proofs_per_step"] == 2 * 2
    assert abs(p["value"] - 2 * 2 * 2 / (p["ms_per_step"] * 2 * 1e-3)) / p["value"] < 1e-6
    # the DEFAULT line's secondary figures under world > 1 (what the driver's `--gpus N` run yields besides the weak-scaling
    # proofs/s): the commit probe runs SHARDED over the ranks (north_star's strong-scaling curve; here 2^17 instead of 2^26 to
    # keep the test short) and gives the unsharded root; the PCIe-inclusive rate is a second timed pass
    m2 = 17
    q = launch(["--workload", "prove", "--concurrency", "2", "--no-cpu-baseline", "--commit-log2-size", str(m2), "--size-classes", "", "--sharded-proof-log2-size", "16"])
    assert q["n_gpus"] == 2 and q["scaling"] == "weak" and q["h2d_inclusive_proofs_per_s"] > 0
    cp = q[f"commit_2p{m2}"]
    assert cp["n_gpus"] == 2 and cp["scaling"] == "strong" and cp["ms_per_commit"] > 0 and "sharded by leaf index over 2 ranks" in cp["workload"]
    polys = []
    for b in range(2):
        t = torch.randint(0, 2**62, (1 << m2, 4), dtype=torch.int64, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(17 + b))
        t[:, 3] &= (1 << 60) - 1
        polys.append(t)
    torch.cuda.synchronize()
    ref = commit_batch(ctx, [int(t.data_ptr()) for t in polys], m2)
    assert cp["root"] == ref.root.hex()
    ref.close()
    # ... and the same invocation carries the whole curve (G = 1 and 2 of the 2 ranks) and the library's own communicator's evidence
    assert sorted(cp["curve"]) == ["1", "2"] and cp["roots_agree"] and cp["curve"]["1"]["root"] == cp["root"]
    assert cp["curve"]["1"]["transport"] == "none" and cp["curve"]["2"]["transport"] == ("host" if one_gpu else "rccl")
    assert q["rccl"]["world"] == 2 and q["rccl"]["ranks_seen"] == [0, 1] and q["rccl"]["transport"] == ("host" if one_gpu else "rccl")
    assert q["rccl"]["allgather_32MiB_us"] > 0
    # ... and configs[3]'s shape: one proof sharded over the two ranks, the same bytes as the lone prover's
    sp = q["sharded_proof_m16"]
    assert sp["n_gpus"] == 2 and sp["scaling"] == "strong" and sp["ms_per_proof"] > 0 and sp["equals_the_lone_provers_transcript"] is True
    # latency mode: one proof at a time sharded over the two ranks (commits of >= 64 rows per rank split by leaf index)
    sh = launch(["--workload", "prove", "--sharded", "--log2-size", "15", "--no-cpu-baseline"])
    assert sh["n_gpus"] == 2 and sh["scaling"] == "strong" and sh["config"]["proofs_per_step"] == 1 and sh["value"] > 0


def test_bench_multirank_on_one_gpu_host_transport(ctx, oracle):
    """PK_BENCH_ONE_GPU=1: both ranks on GPU 0 (RCCL refuses that), joined by the library's host transport over the launcher's
    gloo group -- the same pk_commit_into / pk_prove sharding code as under RCCL, two real processes"""
    _check_multirank_bench(ctx, one_gpu=True)


def test_bench_gpus_2_without_a_launcher():
    """plain `python bench.py --gpus 2 --steps 2` -- no torchrun around it, WORLD_SIZE unset -- starts its own
    two ranks (here both on GPU 0: PK_BENCH_ONE_GPU=1, host transport), reads --gpus, and the one line says so: n_gpus 2, the
    library's communicator saw 2 ranks, and the 2^26 commit ran sharded over them next to the one-rank figure"""
    import json
    import os
    import subprocess
    import sys

    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TORCHELASTIC_RUN_ID")}
    env["PK_BENCH_ONE_GPU"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root_dir, "bench.py"), "--gpus", "2", "--steps", "2", "--size-classes", "", "--no-cpu-baseline", "--sharded-proof-log2-size", "19"],
                         env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["proofs_per_step"] == 2 * 20 and "started its own ranks" in d["config"]["launcher"]
    assert d["rccl"]["world"] == 2 and d["rccl"]["transport"] == "host" and d["rccl"]["ranks_seen"] == [0, 1]
    c = d["commit_2p26"]
    assert c["n_gpus"] == 2 and c["scaling"] == "strong" and sorted(c["curve"]) == ["1", "2"] and c["roots_agree"]
    assert c["root"].startswith("592836a1")  # the 2^26 root every round has reported
    assert d["sharded_proof_m19"]["n_gpus"] == 2 and d["sharded_proof_m19"]["equals_the_lone_provers_transcript"] is True


@pytest.mark.parametrize("where,limit_env", [("probe", "PK_BENCH_RCCL_LIMIT_S"), ("commit", "PK_BENCH_COMMIT_LIMIT_S")])
def test_a_stuck_library_collective_costs_the_line_only_its_own_keys(where, limit_env):
    """the watchdogs of the multi-rank line: a communicator that never forms (probe) or a sharded step that never returns (commit) --
    simulated by a thread that blocks for ever -- must leave the judged figure intact: the weak-scaling proofs/s is measured and printed,
    the keys that needed the library's communicator carry an error string, every rank leaves with exit code 0"""
    import json
    import os
    import subprocess
    import sys

    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TORCHELASTIC_RUN_ID")}
    env.update(PK_BENCH_ONE_GPU="1", PK_BENCH_TEST_HANG=where)
    env[limit_env] = "3"
    out = subprocess.run([sys.executable, os.path.join(root_dir, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--concurrency", "2", "--size-classes", "",
                          "--no-cpu-baseline", "--commit-log2-size", "17", "--sharded-proof-log2-size", "15"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["proofs_per_step"] == 4
    if where == "probe":
        assert "error" in d["rccl"] and "error" in d["commit_2p17"] and "sharded_proof_m15" not in d
    else:
        assert d["rccl"]["ranks_seen"] == [0, 1] and "error" in d["commit_2p17"] and "sharded_proof_m15" not in d


def test_bench_gpus_more_than_present_is_clamped():
    """--gpus 8 on a box with fewer GPUs runs on what is there (and says so) instead of failing or claiming 8"""
    import ctypes as C
    import json
    import os
    import subprocess
    import sys

    from provekit_amd._lib import lib

    n = C.c_int(0)
    lib.pk_device_count(C.byref(n))
    if n.value != 1:
        pytest.skip("written for the one-GPU box")
    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TORCHELASTIC_RUN_ID", "PK_BENCH_ONE_GPU")}
    out = subprocess.run([sys.executable, os.path.join(root_dir, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--log2-size", "13", "--concurrency", "2",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
    assert d["n_gpus"] == 1 and d["rccl"]["world"] == 1


def test_bench_multirank_over_rccl_needs_two_gpus(ctx, oracle):
    """one rank per GPU over RCCL / xGMI, exactly as the driver's SCALE run launches it.  Needs >= 2 GPUs: skipped on this box."""
    import ctypes as C

    from provekit_amd._lib import lib

    n = C.c_int(0)
    lib.pk_device_count(C.byref(n))
    if n.value < 2:
        pytest.skip(f"needs >= 2 GPUs for RCCL between ranks (this box has {n.value})")
    _check_multirank_bench(ctx, one_gpu=False)


def test_bench_line_contract(ctx):
    """the ONE JSON line `python bench.py` prints at N = 1 carries every field the driver and the judge read (here at a small
    size, CPU baseline included): metric / value / unit / n_gpus / steps / warmup / ms_per_step / scaling / dtype / config.workload,
    `roofline` {bound, achieved, peak, unit, frac, traffic} with frac = achieved / peak, and `cpu_baseline` {value, unit, cores,
    kind, sample}"""
    import json
    import os
    import subprocess
    import sys

    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root_dir, "bench.py"), "--log2-size", "13", "--steps", "2", "--warmup", "1", "--concurrency", "4"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "proofs/s" and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["proofs_per_step"] * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1 and (r["traffic"] is None or r["traffic"] > 0)
    assert r["launches_per_step"] == r["launches_per_proof"] * d["config"]["proofs_per_step"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "proofs/s" and "proof" in c["sample"]
