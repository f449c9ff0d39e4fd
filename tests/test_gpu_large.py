"""GPU parity at BASELINE.json's full sizes (configs[4]: the 2^26-coefficient batch-2 WHIR commit, and the 3-pass NTT shapes it
and the m >= 23 proofs use), through the C ABI:
  * pk_ntt at 2^21 .. 2^23 against the oracle's transform (the 3-pass tilings; test_gpu_ntt.py stops at 2^20);
  * the 2^26 commit: the root equals the root the ORACLE builds from the same coefficients (its own encode of the whole codeword, its own
    268 M compressions); opened leaves equal the DEFINITION leaf_i[b*16+j] = f_{b,j}(w^i) evaluated by the oracle on the downloaded
    coefficients, their auth paths chain to the root under the oracle's Skyscraper, and the root equals the one the G = 8
    sharded encode (pk_rs_encode_shard, the multi-GPU path, all shards on this GPU) interleaves to."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_n", [21, 22, 23])
def test_ntt_three_pass_shapes_vs_oracle(ctx, oracle, log_n):
    from provekit_amd.field import random_field
    from provekit_amd.rs import ntt

    x = random_field(1 << log_n, 900 + log_n).reshape(1, 1 << log_n, 4)
    got = ntt(x, ctx=ctx)
    assert np.array_equal(got[0], oracle.ntt(x[0], log_n))


def test_commit_2p26_batch2_against_definition_and_sharded_root(ctx, oracle):
    from provekit_amd._lib import PK_COL_MAJOR, lib

    n_vars, batch, fold, rate = 26, 2, 4, 1
    n = 1 << n_vars
    log_rows = n_vars + rate - fold
    rows, width, fw = 1 << log_rows, batch << fold, 1 << fold
    # seeded uniform coefficients generated on the device by the library's own RNG kernel
    polys = [ctx.alloc_fe(n) for _ in range(batch)]
    seed = bytes(range(32))
    for b, p in enumerate(polys):
        ctx._check(lib.pk_selftest_random_fe(ctx.handle, seed, 100 + b, p.ptr, n))
    ptrs = (C.c_void_p * batch)(*[p.ptr for p in polys])
    root = (C.c_uint8 * 32)()
    tree = C.c_void_p()
    ctx._check(lib.pk_commit(ctx.handle, ptrs, batch, n_vars, rate, fold, root, C.byref(tree)))
    root = np.frombuffer(bytes(root), dtype=np.uint64)
    try:
        idx = np.array([0, 5, 12345678 % rows, rows - 1], dtype=np.uint64)
        k, plen = len(idx), log_rows - 1
        leaves = np.zeros((k, width, 4), np.uint64)
        sib = np.zeros((k, 4), np.uint64)
        paths = np.zeros((k, plen, 4), np.uint64)
        ctx._check(lib.pk_tree_open(ctx.handle, tree, idx.ctypes.data, k, 0, leaves.ctypes.data, sib.ctypes.data, paths.ctypes.data))
        # (1) the definition: leaf_i[b*16 + j] = sum_t c_b[16 t + j] w^(i t), w of order `rows`  
        w = oracle.from_mont(oracle.root_of_unity(log_rows).reshape(1, 4))
        w_int = oracle.limbs_to_ints(w)[0]
        host = [ctx.download_fe(p, n) for p in polys]
        for q in (1, 2):  # two leaves x 32 columns x 2^22-term Horner evaluations in the oracle
            pt = oracle.to_mont(oracle.ints_to_limbs([pow(w_int, int(idx[q]), oracle.P)]))[0]
            for b in range(batch):
                for j in range(fw):
                    want = oracle.eval_univariate(np.ascontiguousarray(host[b][j::fw]), pt)
                    assert np.array_equal(leaves[q, b * fw + j], want), (q, b, j)
        # leaf 0 is the evaluation at w^0 = 1: the plain sum of each sub-sequence
        one = oracle.to_mont(oracle.ints_to_limbs([1]))[0]
        for b in range(batch):
            assert np.array_equal(leaves[0, b * fw + 3], oracle.eval_univariate(np.ascontiguousarray(host[b][3::fw]), one))
        # (1b) the root itself against the oracle's own commit of the same coefficients: the whole 2^23 x 32 codeword re-encoded and all
        # 268 M compressions redone on the host (~12 GB of host memory, about a minute of all cores)
        want_leaves = oracle.rs_encode(np.stack(host), batch, n_vars, rate, fold)
        del host
        assert np.array_equal(want_leaves[idx.astype(np.int64)], leaves)
        want_root = oracle.merkle_commit(want_leaves)[1]
        del want_leaves
        assert np.array_equal(want_root, root), "pk_commit's 2^26 root differs from the oracle-built root"
        # (2) every opened leaf chains to the root under the oracle's Skyscraper (leaf fold, then the auth path)
        dig = oracle.leaf_hash(leaves)
        for q in range(k):
            h, i = dig[q], int(idx[q])
            sibs = [sib[q]] + [paths[q, d] for d in range(plen - 1, -1, -1)]  # leaf level upward (paths are root -> leaf)
            for s in sibs:
                pair = np.concatenate([s, h]) if i & 1 else np.concatenate([h, s])
                h = np.frombuffer(oracle.compress_many(pair.tobytes()), dtype=np.uint64)
                i >>= 1
            assert np.array_equal(h, root), q
        # (3) the multi-GPU path: 8 leaf-index shards encoded and hashed independently, digests interleaved, inner tree on top
        G = 8
        loc = rows // G
        d_loc, d_dig = ctx.alloc_fe(width * loc), ctx.alloc_fe(loc)
        d_scr = ctx.alloc_fe(width * (rows + 2 * loc))
        all_dig = np.zeros((rows, 4), np.uint64)
        for g in range(G):
            ctx._check(lib.pk_rs_encode_shard(ctx.handle, ptrs, batch, n_vars, rate, fold, g, G, d_loc.ptr, d_scr.ptr))
            ctx._check(lib.pk_leaf_hash(ctx.handle, d_loc.ptr, loc, width, PK_COL_MAJOR, d_dig.ptr))
            all_dig[g::G] = ctx.download_fe(d_dig, loc)
            # shard row t is codeword row g + G t: compare whole rows with openings of the unsharded tree
            ts = np.array([0, 77, loc - 1], dtype=np.uint64)
            got = np.zeros((len(ts), width, 4), np.uint64)
            ctx._check(lib.pk_gather_leaves(ctx.handle, d_loc.ptr, loc, width, PK_COL_MAJOR, ts.ctypes.data, len(ts), 0, got.ctypes.data))
            gi = (ts * np.uint64(G) + np.uint64(g)).astype(np.uint64)
            lv, sb, pt = np.zeros((len(ts), width, 4), np.uint64), np.zeros((len(ts), 4), np.uint64), np.zeros((len(ts), plen, 4), np.uint64)
            ctx._check(lib.pk_tree_open(ctx.handle, tree, gi.ctypes.data, len(ts), 0, lv.ctypes.data, sb.ctypes.data, pt.ctypes.data))
            assert np.array_equal(got, lv)
        d_nodes = ctx.alloc_fe(2 * rows)
        ctx.upload_into(d_nodes.view_fe(rows), all_dig)
        ctx._check(lib.pk_merkle_inner(ctx.handle, d_nodes.ptr, rows))
        assert np.array_equal(ctx.download_fe(d_nodes.view_fe(1), 1)[0], root)
    finally:
        lib.pk_tree_destroy(ctx.handle, tree)


@pytest.mark.parametrize("log_len", [20, 22])
def test_sumcheck_cubic_rounds_full_size_vs_oracle(ctx, oracle, log_len):
    """the Spartan sumcheck's hot loop (sumcheck.rs:16-104 with the map of whir_r1cs.rs:284-291) at the sizes the bench and the
    sha256 size class run it, every round's three evaluations against the oracle and the folded arrays compared at the end
    (tests/test_gpu_mle.py stops at 2^18)"""
    from provekit_amd import sumcheck as sc
    from provekit_amd.field import random_field

    n = 1 << log_len
    arrs = [random_field(n, 520 + k + log_len) for k in range(4)]
    d = [ctx.upload(a) for a in arrs]
    alphas = random_field(log_len, 599)
    cur, length, fold = [a.copy() for a in arrs], n, None
    for rnd in range(log_len):
        got = sc.sumcheck_fold_map_reduce(ctx, *d, length, fold)
        exp, *cur = oracle.sumcheck_cubic_round(*[c[:length] for c in cur], fold)
        assert np.array_equal(got, exp), (log_len, rnd)
        if fold is not None:
            length //= 2
            if rnd in (1, 4, 9) or length <= 64:
                for k in range(4):
                    assert np.array_equal(ctx.download_fe(d[k], length), cur[k][:length]), (rnd, k)
        fold = alphas[rnd]


@pytest.mark.parametrize("log_len", [21, 23])
def test_sumcheck_quadratic_rounds_full_size_vs_oracle(ctx, oracle, log_len):
    """the WHIR sumcheck (whir_utilities.go:102-125) at the witness polynomial's size of the bench (2^21) and of the sha256 size
    class (2^23): h(0), h(1), h(2) of every round and the folded tables against the oracle"""
    from provekit_amd import sumcheck as sc
    from provekit_amd.field import random_field

    n = 1 << log_len
    f, w = random_field(n, 631 + log_len), random_field(n, 632 + log_len)
    bufs = [[ctx.upload(f), ctx.alloc_fe(n)], [ctx.upload(w), ctx.alloc_fe(n)]]
    rs = random_field(log_len, 677)
    cf, cw, length, fold, cur = f, w, n, None, 0
    for rnd in range(log_len):
        if fold is None:
            got = sc.sumcheck_quadratic_round(ctx, bufs[0][cur], bufs[1][cur], length)
        else:
            got = sc.sumcheck_quadratic_round(ctx, bufs[0][cur], bufs[1][cur], length, fold, bufs[0][1 - cur], bufs[1][1 - cur])
            cur, length = 1 - cur, length // 2
        full = length * 2 if fold is not None else length
        exp, cf, cw = oracle.sumcheck_quadratic_round(cf[:full], cw[:full], fold)
        assert np.array_equal(got, exp), (log_len, rnd)
        if fold is not None and (rnd in (1, 5) or length <= 64):
            assert np.array_equal(ctx.download_fe(bufs[0][cur], length), cf[:length])
            assert np.array_equal(ctx.download_fe(bufs[1][cur], length), cw[:length])
        fold = rs[rnd]
