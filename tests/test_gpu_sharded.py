"""GPU: the multi-GPU path behind the C ABI (include/provekit_hip.h "device sets") on ONE GPU.

RCCL refuses two ranks on one device, so the G ranks of these tests are G contexts of this process on GPU 0 joined by the
library's in-process transport (pk_ctx_create_set with a repeated device), one host thread per rank -- the same
commit_into / pk_tree_open / pk_prove code the RCCL transport drives on a real node; only the two collectives differ.
Checks: collectives; sharded pk_commit root == unsharded root and sharded openings == unsharded openings; sharded pk_prove
transcripts byte-identical to the lone prover's for the same seed (m = 17 and the bench size m = 21), accepted by the
verifier; RCCL itself is exercised at world size 1 (library loads, communicator forms, both collectives run)."""
import ctypes as C
import threading

import numpy as np
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))


def run_ranks(ctxs, fn):
    """fn(rank, ctx) on one thread per rank; re-raises the first failure"""
    out, err = [None] * len(ctxs), []

    def go(r):
        try:
            out[r] = fn(r, ctxs[r])
        except BaseException as e:  # noqa: BLE001
            err.append(e)

    ths = [threading.Thread(target=go, args=(r,)) for r in range(len(ctxs))]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=600)
    if err:
        raise err[0]
    return out


@pytest.fixture()
def rank_sets():
    import torch

    torch.cuda.is_available()
    import provekit_amd

    made = []

    def make(G):
        cs = provekit_amd.Context.create_set([0] * G)
        made.append(cs)
        return cs

    yield make
    for cs in made:
        for c in cs:
            c.close()


@pytest.mark.parametrize("G", [2, 4])
def test_local_transport_collectives(rank_sets, G):
    from provekit_amd._lib import PK_OK, lib

    ctxs = rank_sets(G)
    assert [c.comm_info() for c in ctxs] == [(r, G, 1) for r in range(G)]  # kind 1 = PK_COMM_LOCAL
    n = 1000

    def fn(r, c):
        send = np.full((n,), r + 1, np.uint64) * np.arange(1, n + 1, dtype=np.uint64)
        d_send, d_recv = c.upload(send), c.alloc(8 * n * G)
        c._check(lib.pk_comm_all_gather(c.handle, d_send.ptr, d_recv.ptr, 8 * n))
        got = c.download(d_recv, (G, n))
        red = np.zeros(n, np.uint64)
        red[r::G] = 7 + r  # each element non-zero on exactly one rank
        d_red = c.upload(red)
        c._check(lib.pk_comm_all_reduce_sum_u64(c.handle, d_red.ptr, n))
        return got, c.download(d_red, (n,))

    for got, red in run_ranks(ctxs, fn):
        for p in range(G):
            assert np.array_equal(got[p], np.full((n,), p + 1, np.uint64) * np.arange(1, n + 1, dtype=np.uint64))
        assert np.array_equal(red, np.array([7 + (i % G) for i in range(n)], np.uint64))


def test_rccl_communicator_of_one_rank(ctx):
    """the RCCL transport on the one GPU there is: librccl resolves, ncclCommInitRank succeeds, both collectives run"""
    import provekit_amd
    from provekit_amd._lib import lib

    c = provekit_amd.Context(0)
    c.comm_init_rank(provekit_amd.Context.comm_unique_id(), 1, 0)
    assert c.comm_info() == (0, 1, 2)  # PK_COMM_RCCL
    x = np.arange(4096, dtype=np.uint64)
    d, e = c.upload(x), c.alloc(x.nbytes)
    c._check(lib.pk_comm_all_gather(c.handle, d.ptr, e.ptr, x.nbytes))
    c._check(lib.pk_comm_all_reduce_sum_u64(c.handle, e.ptr, x.size))
    c.sync()
    assert np.array_equal(c.download(e, x.shape), x)
    c.comm_destroy()
    c.close()


@pytest.mark.parametrize("G,batch,n_vars", [(2, 2, 12), (4, 2, 16), (8, 1, 17), (4, 2, 21)])
def test_sharded_commit_and_openings_match_unsharded(ctx, oracle, rank_sets, G, batch, n_vars):
    from provekit_amd.field import random_field
    from provekit_amd.whir import commit_batch

    polys = [random_field(1 << n_vars, 70 + b + n_vars) for b in range(batch)]
    ref = commit_batch(ctx, [ctx.upload(p) for p in polys], n_vars)
    rows = ref.n_leaves
    rng = np.random.default_rng(G + n_vars)
    idx = np.unique(np.concatenate([rng.integers(0, rows, size=60), [0, rows - 1]])).astype(np.uint64)
    want = [ref.open(idx, canonical_leaves=cl) for cl in (True, False)]
    ctxs = rank_sets(G)

    def fn(r, c):
        com = commit_batch(c, [c.upload(p) for p in polys], n_vars)
        res = (com.root, [com.open(idx, canonical_leaves=cl) for cl in (True, False)])
        com.close()
        return res

    for root, opened in run_ranks(ctxs, fn):
        assert root == ref.root
        for got, exp in zip(opened, want):
            for a, b in zip(got, exp):
                assert np.array_equal(a, b)
    ref.close()


def _sharded_prove_case(ctx, oracle, rank_sets, G, m, m_0, nc, n_in, seed, test_pow, verify_r1cs):
    import verifier as V
    from test_gpu_prove import satisfiable_r1cs, to_sparse

    from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
    from provekit_amd.sparse_matrix import R1CS

    nw, z, coeffs, trips = satisfiable_r1cs(nc, n_in, seed)
    zm = oracle.to_mont(oracle.ints_to_limbs(z))
    interner = oracle.to_mont(oracle.ints_to_limbs(coeffs))
    cfg_w, cfg_b = WhirConfig.for_size(m, test_pow), blinding_config_for(m_0, test_pow)

    def prove_on(c):
        r1cs = R1CS(c, *(to_sparse(nc, nw, t) for t in trips), interner)
        s = WhirR1CSScheme(c, r1cs, m, m_0, cfg_w, cfg_b)
        proofs = [s.prove(c.upload(zm), seed=sd) for sd in (seed, seed + 1)]
        ds = s.domain_separator
        s.close()
        r1cs.close()
        return proofs, ds

    want, ds = prove_on(ctx)
    ranks = rank_sets(G)
    for proofs, _ in run_ranks(ranks, lambda r, c: prove_on(c)):
        assert proofs == want, "a rank of the sharded prover diverged from the lone prover's transcript"

    # production randomness (no injected seed): rank 0's OS key is shared, so the ranks still agree with each other
    def prove_fresh(c):
        r1cs = R1CS(c, *(to_sparse(nc, nw, t) for t in trips), interner)
        s = WhirR1CSScheme(c, r1cs, m, m_0, cfg_w, cfg_b)
        proof = s.prove(c.upload(zm))
        s.close()
        r1cs.close()
        return proof

    fresh = run_ranks(ranks, lambda r, c: prove_fresh(c))
    assert all(p == fresh[0] for p in fresh) and fresh[0] not in want

    def vcfg(c):
        return V.WhirConfig(c.n_vars, c.batch_size, c.folding_factor, c.starting_log_inv_rate, c.num_queries, c.ood_samples, c.pow_bits,
                            c.final_queries, c.final_pow_bits, c.commitment_ood_samples, c.final_folding_pow_bits)

    mats = [(t[0], t[1], [coeffs[v] for v in t[2]]) for t in trips]
    assert V.verify(want[0], ds, m, m_0, vcfg(cfg_w), vcfg(cfg_b), r1cs=(nc, nw, mats) if verify_r1cs else None)
    assert V.verify(fresh[0], ds, m, m_0, vcfg(cfg_w), vcfg(cfg_b))


@pytest.mark.parametrize("G", [2, 4])
def test_sharded_prove_m17_transcript_identical(ctx, oracle, rank_sets, G):
    _sharded_prove_case(ctx, oracle, rank_sets, G, m=17, m_0=16, nc=60000, n_in=5000, seed=17, test_pow=8.0, verify_r1cs=False)


def test_sharded_prove_bench_size_transcript_identical(ctx, oracle, rank_sets):
    """m = 21 under the reference's own schedule on a satisfiable instance, two ranks: every commit of the witness WHIR down to
    2^7 leaves is sharded; all transcripts equal the lone prover's and the verifier accepts"""
    import verifier as V
    from test_gpu_prove import size_class_instance

    from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
    from provekit_amd.sparse_matrix import R1CS, SparseMatrix

    m, m_0 = 21, 20
    nc, nw, mats, interner, z = size_class_instance(oracle, m)
    cfg_w, cfg_b = WhirConfig.derive(m), blinding_config_for(m_0)

    def prove_on(c):
        r1cs = R1CS(c, *(SparseMatrix(nc, nw, *t) for t in mats), interner)
        s = WhirR1CSScheme(c, r1cs, m, m_0, cfg_w, cfg_b)
        proof = s.prove(c.upload(z), seed=5)
        ds = s.domain_separator
        s.close()
        r1cs.close()
        return proof, ds

    want, ds = prove_on(ctx)
    for proof, _ in run_ranks(rank_sets(2), lambda r, c: prove_on(c)):
        assert proof == want

    def vcfg(c):
        return V.WhirConfig(c.n_vars, c.batch_size, c.folding_factor, c.starting_log_inv_rate, c.num_queries, c.ood_samples, c.pow_bits,
                            c.final_queries, c.final_pow_bits, c.commitment_ood_samples, c.final_folding_pow_bits)

    assert V.verify(want, ds, m, m_0, vcfg(cfg_w), vcfg(cfg_b))


def sharded_profile(ctx, oracle, m, G, seed=25):
    """One proof of the size class (m, m_0 = m - 1) by the lone prover and by G ranks of the in-process transport on GPU 0 under
    the TURNSTILE (comm.hip: the ranks' segments run one after another, so pk_profile_* times each rank's kernels as if it had
    the chip to itself).  Returns (lone transcript, rank transcripts, domain separator, report): report carries the per-kernel
    milliseconds of the lone prover and of rank 0, each rank's kernel total, and the REPLICATED share derived from them:
    T_rank = R + (T_lone - R) / G  =>  R = (T_rank - T_lone / G) / (1 - 1 / G)."""
    import provekit_amd
    from test_gpu_prove import size_class_instance

    from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
    from provekit_amd.sparse_matrix import R1CS, SparseMatrix

    m_0 = m - 1
    nc, nw, mats, interner, z = size_class_instance(oracle, m)
    cfg_w, cfg_b = WhirConfig.derive(m), blinding_config_for(m_0)
    gate = threading.Barrier(G)

    def prove_on(c, wait=None):
        r1cs = R1CS(c, *(SparseMatrix(nc, nw, *t) for t in mats), interner)
        s = WhirR1CSScheme(c, r1cs, m, m_0, cfg_w, cfg_b)
        d_z = c.upload(z)
        s.prove(d_z, seed=seed + 1)  # warm-up: twiddle tables, workspaces
        c.sync()
        if wait is not None:
            wait.wait()
        c.profile(True)
        c.profile_reset()
        proof = s.prove(d_z, seed=seed)
        prof = c.profile_read()
        c.profile(False)
        ds = s.domain_separator
        s.close()
        r1cs.close()
        d_z.free()
        return proof, ds, prof

    want, ds, lone = prove_on(ctx)
    os.environ["PK_LOCAL_TURNSTILE"] = "1"
    try:
        ranks = provekit_amd.Context.create_set([0] * G)
    finally:
        del os.environ["PK_LOCAL_TURNSTILE"]
    try:
        res = run_ranks(ranks, lambda r, c: prove_on(c, gate))
    finally:
        for c in ranks:
            c.close()
    kernel_ms = lambda prof: sum(v[1] for k, v in prof.items() if not k.startswith("comm_"))
    t_lone = kernel_ms(lone)
    t_rank = [kernel_ms(p) for _, _, p in res]
    worst = max(t_rank)
    report = {
        "m": m, "ranks": G, "transport": "in-process, one GPU, turnstile (ranks take turns between collectives)",
        "lone_kernel_ms": round(t_lone, 3), "rank_kernel_ms": [round(t, 3) for t in t_rank],
        "rank_over_lone": round(worst / t_lone, 4), "ideal": round(1.0 / G, 4),
        "replicated_share_of_lone": round((worst - t_lone / G) / (1.0 - 1.0 / G) / t_lone, 4),
        "amdahl_speedup_bound": round(t_lone / worst, 2),
        "lone_ms_by_kernel": {k: round(v[1], 3) for k, v in sorted(lone.items(), key=lambda kv: -kv[1][1])},
        "rank0_ms_by_kernel": {k: round(v[1], 3) for k, v in sorted(res[0][2].items(), key=lambda kv: -kv[1][1])},
    }
    return want, [p for p, _, _ in res], ds, report


def test_sharded_prove_p256_size_class_eight_ranks(ctx, oracle):
    """BASELINE configs[3] as far as one GPU can take it: the m = 25 size class proven by EIGHT ranks (in-process transport, all on
    GPU 0) under the derived schedule -- every rank's transcript is byte-identical to the lone prover's, the verifier accepts it,
    and the work a rank still does in full (replicated) is under a quarter of the lone prover's kernel time: commits, both
    sumchecks, the equality weights, the statement weights, OOD evaluations and proof-of-work are all split over the ranks."""
    import json

    import verifier as V

    from provekit_amd.scheme import WhirConfig, blinding_config_for

    m, G = 25, 8
    want, got, ds, report = sharded_profile(ctx, oracle, m, G)
    print("sharded_profile", json.dumps(report))
    for proof in got:
        assert proof == want, "a rank of the 8-way sharded prover diverged from the lone prover's transcript"
    assert report["replicated_share_of_lone"] < 0.25, report
    cfg_w, cfg_b = WhirConfig.derive(m), blinding_config_for(m - 1)

    def vcfg(c):
        return V.WhirConfig(c.n_vars, c.batch_size, c.folding_factor, c.starting_log_inv_rate, c.num_queries, c.ood_samples, c.pow_bits,
                            c.final_queries, c.final_pow_bits, c.commitment_ood_samples, c.final_folding_pow_bits)

    assert V.verify(want, ds, m, m - 1, vcfg(cfg_w), vcfg(cfg_b))


def test_a_failing_rank_wakes_its_peers_instead_of_hanging_them(rank_sets):
    """a rank that fails before reaching a collective used to leave the others blocked on the barrier.  Here rank 1
    asks for a commit it cannot do (a null polynomial pointer fails inside commit_into, after the layout was decided and
    before the all-gather): it aborts the group, rank 0 -- already waiting in the all-gather -- returns PK_ERR_RCCL, and every
    later collective on the set fails at once on both ranks."""
    from provekit_amd._lib import ProveKitHipError, lib
    from provekit_amd.field import random_field

    ctxs = rank_sets(2)
    n_vars = 14
    poly = random_field(1 << n_vars, 3)

    def fn(r, c):
        d = c.upload(poly)
        ptrs = (C.c_void_p * 1)(d.ptr if r == 0 else None)
        szs = [C.c_size_t() for _ in range(3)]
        c._check(lib.pk_commit_sizes(c.handle, 1, n_vars, 1, 4, *[C.byref(x) for x in szs]))
        leaves, nodes, scratch = (c.alloc_fe(x.value) for x in szs)
        root = (C.c_uint8 * 32)()
        rc = lib.pk_commit_into(c.handle, ptrs, 1, n_vars, 1, 4, leaves.ptr, nodes.ptr, scratch.ptr, root, None)
        x = c.upload(np.arange(8, dtype=np.uint64))
        y = c.alloc(128)
        rc2 = lib.pk_comm_all_gather(c.handle, x.ptr, y.ptr, 64)
        return rc, rc2

    res = run_ranks(ctxs, fn)
    assert res[1][0] == -1  # PK_ERR_BAD_ARG on the rank that failed
    assert res[0][0] == -4  # PK_ERR_RCCL on the rank that was waiting for it
    assert res[0][1] == -4 and res[1][1] == -4  # the communicator stays unusable: nobody waits for anybody

    # ... until every rank, back from the failed call, resets it (a recoverable caller error must not cost the device set)
    def again(r, c):
        c._check(lib.pk_comm_reset(c.handle))
        return r

    run_ranks(ctxs, again)

    def gather(r, c):
        x = c.upload(np.arange(8, dtype=np.uint64) + 10 * r)
        y = c.alloc(128)
        c._check(lib.pk_comm_all_gather(c.handle, x.ptr, y.ptr, 64))
        return c.download(y, (2, 8))

    for got in run_ranks(ctxs, gather):
        assert np.array_equal(got, np.stack([np.arange(8, dtype=np.uint64), np.arange(8, dtype=np.uint64) + 10]))


def test_a_rank_that_fails_inside_a_sharded_proof_wakes_its_peers(oracle, rank_sets):
    """only the commit path aborted the group; every other early return inside pk_prove left the peers at a barrier
    for ever.  Rank 1 is refused at the door (wrong witness length -- any exit that is not PK_OK takes the same path: arena
    exhausted, a HIP error, an unsatisfied witness in pk_noir_prove): it gets its own error, rank 0 -- already inside the proof's
    first collective -- returns PK_ERR_RCCL instead of hanging, and the communicator stays unusable."""
    from test_gpu_prove import satisfiable_r1cs, to_sparse

    from provekit_amd._lib import lib
    from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
    from provekit_amd.sparse_matrix import R1CS

    m, m_0, nc, n_in = 13, 11, 1500, 2000
    nw, z, coeffs, trips = satisfiable_r1cs(nc, n_in, 6)
    zm = oracle.to_mont(oracle.ints_to_limbs(z))
    interner = oracle.to_mont(oracle.ints_to_limbs(coeffs))
    cfg_w, cfg_b = WhirConfig.for_size(m, 4.0), blinding_config_for(m_0, 4.0)

    def fn(r, c):
        r1cs = R1CS(c, *(to_sparse(nc, nw, t) for t in trips), interner)
        s = WhirR1CSScheme(c, r1cs, m, m_0, cfg_w, cfg_b)
        d = c.upload(zm)
        n = C.c_size_t()
        rc = lib.pk_prove(c.handle, s.handle, d.ptr, nw - (1 if r == 1 else 0), None, s._buf, len(s._buf), C.byref(n))
        rc2 = lib.pk_prove(c.handle, s.handle, d.ptr, nw, None, s._buf, len(s._buf), C.byref(n))
        s.close()
        r1cs.close()
        return rc, rc2

    res = run_ranks(rank_sets(2), fn)
    assert res[1][0] == -1 and res[0][0] == -4, res  # PK_ERR_BAD_ARG where it happened, PK_ERR_RCCL on the rank that was waiting
    assert res[0][1] == -4 and res[1][1] == -4, res  # nobody waits for anybody afterwards


def test_host_transport_in_process_and_its_failure_path(ctx):
    """pk_comm_init_host with a Python callback standing in for MPI / gloo: two contexts of this process exchange through a
    rendezvous written here; then a callback that reports failure must surface as PK_ERR_RCCL and poison the communicator."""
    import provekit_amd
    from provekit_amd._lib import lib
    from provekit_amd.field import random_field
    from provekit_amd.whir import commit_batch

    G = 2
    CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
    gate = threading.Barrier(G)
    slots = [None] * G
    fail = {"on": False}

    def make_cb(rank):
        def cb(_user, send, recv, nbytes):
            if fail["on"]:
                return 7
            slots[rank] = C.string_at(send, nbytes)
            gate.wait()
            C.memmove(recv, b"".join(slots), G * nbytes)
            gate.wait()
            return 0

        return CB(cb)

    cbs = [make_cb(r) for r in range(G)]
    ctxs = [provekit_amd.Context(0) for _ in range(G)]
    try:
        for r, c in enumerate(ctxs):
            c._check(lib.pk_comm_init_host(c.handle, G, r, cbs[r], None))
            assert c.comm_info() == (r, G, 3)  # PK_COMM_HOST
        n_vars = 14
        polys = [random_field(1 << n_vars, 40 + b) for b in range(2)]
        ref = commit_batch(ctx, [ctx.upload(p) for p in polys], n_vars)
        idx = np.array([0, 5, 77, ref.n_leaves - 1], dtype=np.uint64)
        want = ref.open(idx, canonical_leaves=True)

        def fn(r, c):
            com = commit_batch(c, [c.upload(p) for p in polys], n_vars)
            out = (com.root, com.open(idx, canonical_leaves=True))
            com.close()
            return out

        for root, opened in run_ranks(ctxs, fn):
            assert root == ref.root
            for a, b in zip(opened, want):
                assert np.array_equal(a, b)
        ref.close()
        fail["on"] = True
        x = ctxs[0].upload(np.arange(8, dtype=np.uint64))
        y = ctxs[0].alloc(128)
        assert lib.pk_comm_all_gather(ctxs[0].handle, x.ptr, y.ptr, 64) == -4
        fail["on"] = False
        assert lib.pk_comm_all_gather(ctxs[0].handle, x.ptr, y.ptr, 64) == -4  # sticky
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("G,m", [(2, 13), (4, 15), (16, 17), (2, 16)])
def test_sharded_prove_around_the_sharding_thresholds(ctx, oracle, rank_sets, G, m):
    """sizes at which some arrays of a proof are sharded and others are not (commits from 64 rows per rank, sumcheck tables from
    2^13 entries per rank), and the widest set the library takes (16 ranks): transcripts still equal the lone prover's"""
    _sharded_prove_case(ctx, oracle, rank_sets, G, m=m, m_0=m - 1, nc=(1 << (m - 2)) - 37, n_in=(1 << (m - 3)) - 5, seed=m + G, test_pow=6.0,
                        verify_r1cs=False)


# ---- >= 2 GPUs: RCCL between distinct devices.  Skipped on a one-GPU box; the first multi-GPU box runs them unchanged. ----------
def _gpu_count():
    from provekit_amd._lib import lib

    n = C.c_int(0)
    lib.pk_device_count(C.byref(n))
    return n.value


@pytest.fixture()
def rccl_set():
    """pk_ctx_create_set over the first 2 (4, 8 when present) distinct GPUs: ncclCommInitAll, one host thread per rank"""
    n = _gpu_count()
    if n < 2:
        pytest.skip(f"needs >= 2 GPUs for RCCL between ranks (this box has {n})")
    import provekit_amd

    made = []

    def make(G):
        if G > n:
            pytest.skip(f"needs {G} GPUs (this box has {n})")
        cs = provekit_amd.Context.create_set(list(range(G)))
        assert [c.comm_info() for c in cs] == [(r, G, 2) for r in range(G)]  # kind 2 = PK_COMM_RCCL
        made.append(cs)
        return cs

    yield make
    for cs in made:
        for c in cs:
            c.close()


@pytest.mark.parametrize("G", [2, 4, 8])
def test_rccl_collectives_between_gpus(rccl_set, G):
    from provekit_amd._lib import lib

    ctxs = rccl_set(G)
    n = 100_000

    def fn(r, c):
        send = np.full((n,), r + 1, np.uint64) * np.arange(1, n + 1, dtype=np.uint64)
        d_send, d_recv = c.upload(send), c.alloc(8 * n * G)
        c._check(lib.pk_comm_all_gather(c.handle, d_send.ptr, d_recv.ptr, 8 * n))
        red = np.zeros(n, np.uint64)
        red[r::G] = 7 + r
        d_red = c.upload(red)
        c._check(lib.pk_comm_all_reduce_sum_u64(c.handle, d_red.ptr, n))
        c.sync()
        return c.download(d_recv, (G, n)), c.download(d_red, (n,))

    for got, red in run_ranks(ctxs, fn):
        for p in range(G):
            assert np.array_equal(got[p], np.full((n,), p + 1, np.uint64) * np.arange(1, n + 1, dtype=np.uint64))
        assert np.array_equal(red, np.array([7 + (i % G) for i in range(n)], np.uint64))


@pytest.mark.parametrize("G,batch,n_vars", [(2, 2, 16), (2, 2, 21), (4, 2, 21), (8, 2, 23)])
def test_rccl_sharded_commit_and_openings_match_unsharded(ctx, oracle, rccl_set, G, batch, n_vars):
    from provekit_amd.field import random_field
    from provekit_amd.whir import commit_batch

    ctxs = rccl_set(G)
    polys = [random_field(1 << n_vars, 70 + b + n_vars) for b in range(batch)]
    ref = commit_batch(ctx, [ctx.upload(p) for p in polys], n_vars)
    rows = ref.n_leaves
    rng = np.random.default_rng(G + n_vars)
    idx = np.unique(np.concatenate([rng.integers(0, rows, size=60), [0, rows - 1]])).astype(np.uint64)
    want = [ref.open(idx, canonical_leaves=cl) for cl in (True, False)]

    def fn(r, c):
        com = commit_batch(c, [c.upload(p) for p in polys], n_vars)
        res = (com.root, [com.open(idx, canonical_leaves=cl) for cl in (True, False)])
        com.close()
        return res

    for root, opened in run_ranks(ctxs, fn):
        assert root == ref.root
        for got, exp in zip(opened, want):
            for a, b in zip(got, exp):
                assert np.array_equal(a, b)
    ref.close()


@pytest.mark.parametrize("G,m", [(2, 17), (2, 21), (8, 21), (8, 25)])
def test_rccl_sharded_prove_transcript_identical(ctx, oracle, rccl_set, G, m):
    """one proof sharded over G GPUs (BASELINE configs[3] at m = 25, G = 8): byte-identical to the lone prover's transcript"""
    import verifier as V
    from test_gpu_prove import size_class_instance

    from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
    from provekit_amd.sparse_matrix import R1CS, SparseMatrix

    ctxs = rccl_set(G)
    m_0 = m - 1
    nc, nw, mats, interner, z = size_class_instance(oracle, m)
    cfg_w, cfg_b = WhirConfig.derive(m), blinding_config_for(m_0)

    def prove_on(c):
        r1cs = R1CS(c, *(SparseMatrix(nc, nw, *t) for t in mats), interner)
        s = WhirR1CSScheme(c, r1cs, m, m_0, cfg_w, cfg_b)
        proofs = [s.prove(c.upload(z), seed=5), s.prove(c.upload(z))]  # injected seed, then production randomness (rank 0's key)
        ds = s.domain_separator
        s.close()
        r1cs.close()
        return proofs, ds

    want, ds = prove_on(ctx)
    got = run_ranks(ctxs, lambda r, c: prove_on(c))
    for proofs, _ in got:
        assert proofs[0] == want[0]
        assert proofs[1] == got[0][0][1] and proofs[1] != want[1]

    def vcfg(c):
        return V.WhirConfig(c.n_vars, c.batch_size, c.folding_factor, c.starting_log_inv_rate, c.num_queries, c.ood_samples, c.pow_bits,
                            c.final_queries, c.final_pow_bits, c.commitment_ood_samples, c.final_folding_pow_bits)

    assert V.verify(want[0], ds, m, m_0, vcfg(cfg_w), vcfg(cfg_b))
    assert V.verify(got[0][0][1], ds, m, m_0, vcfg(cfg_w), vcfg(cfg_b))


def test_noir_prove_on_a_device_set(ctx, oracle, rank_sets):
    """pk_noir_prove with every rank of a set calling it (witness builders replicated, the proof sharded): seeded transcripts equal
    the lone prover's; with OS randomness rank 0's key reaches every rank BEFORE fill_witness, so the ranks fill the three unset
    witnesses alike and still agree byte for byte; the verifier accepts both"""
    import verifier as V
    from test_gpu_prove import to_sparse
    from test_gpu_witness import _mont, _noir_instance

    from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
    from provekit_amd.sparse_matrix import R1CS
    from provekit_amd.witness import WitnessProgram

    builders, acir, pub_idx, nw, coeffs, trips = _noir_instance(oracle, 77, n_in=6, n_prod=20000)
    nc = trips[0][0][-1] + 1
    m, m_0 = 16, 15
    assert nw <= 1 << (m - 1) and nc <= 1 << m_0
    interner = oracle.to_mont(oracle.ints_to_limbs(coeffs))
    cfg_w, cfg_b = WhirConfig.for_size(m, 6.0), blinding_config_for(m_0, 6.0)
    acir_m = _mont(oracle, acir)

    def prove_on(c):
        r1cs = R1CS(c, *(to_sparse(nc, nw, t) for t in trips), interner)
        s = WhirR1CSScheme(c, r1cs, m, m_0, cfg_w, cfg_b)
        prog = WitnessProgram(c, builders)
        d_acir = c.upload(acir_m)
        out = (s.noir_prove(prog, d_acir, len(acir), pub_idx, seed=9), s.noir_prove(prog, d_acir, len(acir), pub_idx), s.domain_separator)
        prog.close()
        s.close()
        r1cs.close()
        return out

    seeded, lone_fresh, ds = prove_on(ctx)
    got = run_ranks(rank_sets(2), lambda r, c: prove_on(c))
    assert all(g[0] == seeded for g in got), "a rank diverged from the lone prover's seeded transcript"
    assert got[0][1] == got[1][1] and got[0][1] not in (seeded, lone_fresh)

    def vcfg(c):
        return V.WhirConfig(c.n_vars, c.batch_size, c.folding_factor, c.starting_log_inv_rate, c.num_queries, c.ood_samples, c.pow_bits,
                            c.final_queries, c.final_pow_bits, c.commitment_ood_samples, c.final_folding_pow_bits)

    mats = [(t[0], t[1], [coeffs[v] for v in t[2]]) for t in trips]
    for proof in (seeded, got[0][1]):
        assert V.verify(proof, ds, m, m_0, vcfg(cfg_w), vcfg(cfg_b), r1cs=(nc, nw, mats))
