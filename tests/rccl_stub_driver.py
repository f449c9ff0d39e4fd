"""Runs INSIDE a fresh process started by tests/test_gpu_rccl_stub.py with PK_RCCL_LIB = tests/stub_rccl/libpk_stub_rccl.so and
pk_selftest_set_hook(2, 1): the library's RCCL transport (csrc/comm.hip, kind PK_COMM_RCCL) driven at G = 2, 4, 8 on the one GPU of
the box, the "RCCL" being the in-process stand-in.  A fresh process because the library resolves its RCCL once (dlopen at first
use).  Prints one JSON object: what was checked and what the stand-in was asked to do."""
import ctypes as C
import json
import os
import sys
import time
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, HERE, os.path.join(ROOT, "oracle")]


def run_ranks(ctxs, fn):
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


def main():
    assert os.environ.get("PK_RCCL_LIB", "").endswith("libpk_stub_rccl.so")
    import torch

    torch.cuda.is_available()  # torch's HIP runtime first (see conftest.py)
    import oracle_lib as oracle
    import provekit_amd
    from provekit_amd._lib import lib
    from provekit_amd.field import random_field
    from provekit_amd.whir import commit_batch

    assert lib.pk_selftest_set_hook(2, 1) == 0  # the RCCL branch also for a repeated device (the stand-in allows what real RCCL refuses)

    stub = C.CDLL(os.environ["PK_RCCL_LIB"])
    stub.ncclStubCalls.argtypes = [C.POINTER(C.c_ulonglong)]

    def calls():
        a = (C.c_ulonglong * 8)()
        stub.ncclStubCalls(a)
        return dict(zip(["all_gather", "all_reduce", "init_rank", "init_all", "abort", "destroy", "bytes_gathered", "in_place_reduce"], list(a)))

    report = {"version": provekit_amd.Context.rccl_version(), "cases": []}
    assert report["version"][0] == 9990000 and report["version"][1] == os.environ["PK_RCCL_LIB"]
    ctx = provekit_amd.Context(0)
    sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2, 4, 8]

    for G in sizes:
        before = calls()
        ctxs = provekit_amd.Context.create_set([0] * G)  # ncclCommInitAll through the RCCL branch (test hook 2)
        assert [c.comm_info() for c in ctxs] == [(r, G, 2) for r in range(G)]  # kind 2 = PK_COMM_RCCL
        n = 1000

        def coll(r, c):
            send = np.full((n,), r + 1, np.uint64) * np.arange(1, n + 1, dtype=np.uint64)
            d_send, d_recv = c.upload(send), c.alloc(8 * n * G)
            c._check(lib.pk_comm_all_gather(c.handle, d_send.ptr, d_recv.ptr, 8 * n))
            got = c.download(d_recv, (G, n))
            red = np.zeros(n, np.uint64)
            red[r::G] = 7 + r
            d_red = c.upload(red)
            c._check(lib.pk_comm_all_reduce_sum_u64(c.handle, d_red.ptr, n))
            return got, c.download(d_red, (n,))

        for got, red in run_ranks(ctxs, coll):
            for p in range(G):
                assert np.array_equal(got[p], np.full((n,), p + 1, np.uint64) * np.arange(1, n + 1, dtype=np.uint64))
            assert np.array_equal(red, np.array([7 + (i % G) for i in range(n)], np.uint64))
        mid = calls()
        # counts in BYTES as ncclUint8 elements: G ranks x one all-gather of 8n bytes from each of G ranks (+ the all-reduce's none: RCCL does it)
        assert mid["all_gather"] - before["all_gather"] == G and mid["bytes_gathered"] - before["bytes_gathered"] == G * G * 8 * n
        assert mid["all_reduce"] - before["all_reduce"] == G and mid["in_place_reduce"] - before["in_place_reduce"] == G

        # sharded commit + openings == unsharded
        n_vars, batch = (16, 2) if G <= 4 else (17, 1)
        polys = [random_field(1 << n_vars, 70 + b + n_vars) for b in range(batch)]
        ref = commit_batch(ctx, [ctx.upload(p) for p in polys], n_vars)
        rows = ref.n_leaves
        rng = np.random.default_rng(G + n_vars)
        idx = np.unique(np.concatenate([rng.integers(0, rows, size=60), [0, rows - 1]])).astype(np.uint64)
        want = [ref.open(idx, canonical_leaves=cl) for cl in (True, False)]

        def com_fn(r, c):
            com = commit_batch(c, [c.upload(p) for p in polys], n_vars)
            res = (com.root, [com.open(idx, canonical_leaves=cl) for cl in (True, False)])
            com.close()
            return res

        for root, opened in run_ranks(ctxs, com_fn):
            assert root == ref.root
            for got, exp in zip(opened, want):
                for a, b in zip(got, exp):
                    assert np.array_equal(a, b)
        ref.close()

        # sharded proof: every rank's transcript == the lone prover's, accepted by the verifier
        import verifier as V
        from test_gpu_prove import satisfiable_r1cs, to_sparse

        from provekit_amd.scheme import WhirConfig, WhirR1CSScheme, blinding_config_for
        from provekit_amd.sparse_matrix import R1CS

        m = 17 if G <= 4 else 18
        m_0, nc, n_in, seed = m - 1, (1 << (m - 2)) - 37, (1 << (m - 3)) - 5, m + G
        nw, z, coeffs, trips = satisfiable_r1cs(nc, n_in, seed)
        zm = oracle.to_mont(oracle.ints_to_limbs(z))
        interner = oracle.to_mont(oracle.ints_to_limbs(coeffs))
        cfg_w, cfg_b = WhirConfig.for_size(m, 6.0), blinding_config_for(m_0, 6.0)

        def prove_on(c):
            r1cs = R1CS(c, *(to_sparse(nc, nw, t) for t in trips), interner)
            s = WhirR1CSScheme(c, r1cs, m, m_0, cfg_w, cfg_b)
            proof = s.prove(c.upload(zm), seed=seed)
            ds = s.domain_separator
            s.close()
            r1cs.close()
            return proof, ds

        lone, ds = prove_on(ctx)
        for proof, _ in run_ranks(ctxs, lambda r, c: prove_on(c)):
            assert proof == lone, "a rank driven through the RCCL branch diverged from the lone prover's transcript"

        def vcfg(c):
            return V.WhirConfig(c.n_vars, c.batch_size, c.folding_factor, c.starting_log_inv_rate, c.num_queries, c.ood_samples, c.pow_bits,
                                c.final_queries, c.final_pow_bits, c.commitment_ood_samples, c.final_folding_pow_bits)

        assert V.verify(lone, ds, m, m_0, vcfg(cfg_w), vcfg(cfg_b))
        after = calls()
        for c in ctxs:
            c.close()
        report["cases"].append({"G": G, "commit_n_vars": n_vars, "prove_m": m, "proof_bytes": len(lone),
                                "all_gathers": after["all_gather"] - before["all_gather"], "all_reduces": after["all_reduce"] - before["all_reduce"],
                                "bytes_gathered": after["bytes_gathered"] - before["bytes_gathered"]})

    # the multi-process shape of the same branch: one unique id, every rank joins with pk_comm_init_rank (here: threads)
    G = 2
    before = calls()
    uid = provekit_amd.Context.comm_unique_id()
    ctxs = [provekit_amd.Context(0) for _ in range(G)]

    def join(r, c):
        c.comm_init_rank(uid, G, r)
        x = c.upload(np.arange(64, dtype=np.uint64) + 100 * r)
        y = c.alloc(8 * 64 * G)
        c._check(lib.pk_comm_all_gather(c.handle, x.ptr, y.ptr, 8 * 64))
        return c.download(y, (G, 64))

    for got in run_ranks(ctxs, join):
        assert np.array_equal(got, np.stack([np.arange(64, dtype=np.uint64) + 100 * r for r in range(G)]))
    assert calls()["init_rank"] - before["init_rank"] == G

    # a rank that fails before its collective aborts ITS OWN communicator (ncclCommAbort is local); its peer, already waiting in the
    # all-gather, is not woken by that: its collective fails when the wait for the missing rank ends (the stand-in's timeout, shortened by
    # the test; with the real library the deadline of comm.hip comm_wait) and it returns PK_ERR_RCCL.  Nobody hangs.
    n_vars = 14
    poly = random_field(1 << n_vars, 3)

    def failing(r, c):
        d = c.upload(poly)
        ptrs = (C.c_void_p * 1)(d.ptr if r == 0 else None)
        szs = [C.c_size_t() for _ in range(3)]
        c._check(lib.pk_commit_sizes(c.handle, 1, n_vars, 1, 4, *[C.byref(x) for x in szs]))
        leaves, nodes, scratch = (c.alloc_fe(x.value) for x in szs)
        root = (C.c_uint8 * 32)()
        rc = lib.pk_commit_into(c.handle, ptrs, 1, n_vars, 1, 4, leaves.ptr, nodes.ptr, scratch.ptr, root, None)
        x = c.upload(np.arange(8, dtype=np.uint64))
        y = c.alloc(128)
        return rc, lib.pk_comm_all_gather(c.handle, x.ptr, y.ptr, 64)

    res = run_ranks(ctxs, failing)
    assert res[1][0] == -1 and res[0][0] == -4, res  # PK_ERR_BAD_ARG where it happened, PK_ERR_RCCL on the rank that was waiting
    assert res[0][1] == -4 and res[1][1] == -4, res
    assert calls()["abort"] - before["abort"] >= 1
    for c in ctxs:
        c.close()

    # a collective that HANGS on the stream (its peer never joins -- what a dead rank looks like to the real library): the wait behind it has a
    # deadline (PK_COMM_TIMEOUT_S, 2 s here): the rank aborts its OWN communicator, which ends its stuck kernel, and reports PK_ERR_RCCL; honest
    # work queued behind a collective that DID complete is not mistaken for a hang
    uid = provekit_amd.Context.comm_unique_id()
    pair = [provekit_amd.Context(0) for _ in range(2)]
    run_ranks(pair, lambda r, c: c.comm_init_rank(uid, 2, r))
    c0 = pair[0]
    x, y = c0.upload(np.arange(8, dtype=np.uint64)), c0.alloc(128)
    stub.ncclStubHangNext(1)
    before = calls()
    assert lib.pk_comm_all_gather(c0.handle, x.ptr, y.ptr, 64) == 0  # enqueued: the call itself cannot know
    t0 = time.time()
    rc = lib.pk_ctx_sync(c0.handle)
    waited = time.time() - t0
    assert rc == -4 and "did not complete within" in c0.last_error(), (rc, c0.last_error())
    assert 1.5 < waited < 30.0, waited
    assert calls()["abort"] - before["abort"] == 1
    assert lib.pk_comm_all_gather(c0.handle, x.ptr, y.ptr, 64) == -4  # the communicator is gone
    assert lib.pk_ctx_sync(c0.handle) == 0  # the stream is usable: the aborted "kernel" ended
    report["hang_deadline_s"] = round(waited, 2)
    for c in pair:
        c.close()
    ctx.close()
    report["stub_calls"] = calls()
    print("RCCL_STUB_REPORT " + json.dumps(report))


if __name__ == "__main__":
    main()
