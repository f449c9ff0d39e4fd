"""GPU parity: sparse R1CS products (S1, S4), proof of work (P1), commitment handle + openings (N1+N2+M1+M2, Q1)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def synth_r1cs(num_constraints, num_witnesses, seed, nnz_per_row=3, n_interned=17):
    """R1CS-shaped sparse matrices: ~3 entries per row, sorted unique columns, interned small coefficients,
    some empty rows."""
    from provekit_amd.sparse_matrix import SparseMatrix

    rng = np.random.default_rng(seed)
    mats = []
    for _ in range(3):
        nri, ci, vv = [], [], []
        for i in range(num_constraints):
            nri.append(len(ci))
            k = 0 if rng.random() < 0.05 else int(rng.integers(1, 2 * nnz_per_row))
            cols = np.sort(rng.choice(num_witnesses, size=min(k, num_witnesses), replace=False))
            ci += cols.tolist()
            vv += rng.integers(0, n_interned, size=len(cols)).tolist()
        mats.append(SparseMatrix(num_constraints, num_witnesses, np.array(nri, np.uint32), np.array(ci, np.uint32), np.array(vv, np.uint32)))
    return mats


@pytest.mark.parametrize("nc,nw", [(1, 1), (37, 50), (1000, 777), (5000, 4096)])
def test_r1cs_products(ctx, oracle, nc, nw):
    from provekit_amd.field import random_field
    from provekit_amd.sparse_matrix import R1CS

    a, b, c = synth_r1cs(nc, nw, nc + nw)
    interner = random_field(17, 5)
    interner[0] = oracle.to_mont(oracle.ints_to_limbs([1]))[0]
    r = R1CS(ctx, a, b, c, interner)
    z = random_field(nw, 9)
    m0 = max((nc - 1).bit_length(), 1)
    da, db, dc = r.calculate_witness_bounds(ctx.upload(z), m0)
    ea = oracle.spmv(nc, nw, a.new_row_indices, a.col_indices, a.values, interner, z)
    eb = oracle.spmv(nc, nw, b.new_row_indices, b.col_indices, b.values, interner, z)
    pad = (1 << m0) - nc
    zpad = np.zeros((pad, 4), np.uint64)
    assert np.array_equal(ctx.download_fe(da, 1 << m0), np.concatenate([ea, zpad]))
    assert np.array_equal(ctx.download_fe(db, 1 << m0), np.concatenate([eb, zpad]))
    assert np.array_equal(ctx.download_fe(dc, 1 << m0), np.concatenate([oracle.hadamard(ea, eb), zpad]))
    eq = random_field(1 << m0, 10)
    out = ctx.download_fe(r.calculate_external_row_of_r1cs_matrices(ctx.upload(eq)), 3 * nw).reshape(3, nw, 4)
    for k, m in enumerate((a, b, c)):
        exp = oracle.spmv(nc, nw, m.new_row_indices, m.col_indices, m.values, interner, eq[:nc], transpose=True)
        assert np.array_equal(out[k], exp)
    r.close()


def heavy_r1cs(nc, nw, seed):
    """matrices with the long lines real constraint systems have: a column nearly every row touches (the constant-one witness),
    rows and columns whose lengths sit on the heavy threshold (64 / 65) and on the chunk size (2048 / 2049 / 4097), a row with a
    term per witness (a grand sum), besides ordinary short rows"""
    from provekit_amd.sparse_matrix import SparseMatrix

    rng = np.random.default_rng(seed)
    mats = []
    for which in range(3):
        rows = []
        special = {0: 64, 1: 65, 2: 2048, 3: 2049, 4: 4097, 5: nw, 6: 0}
        for i in range(nc):
            k = special.get(i, int(rng.integers(1, 5)))
            cols = set(int(c) for c in rng.choice(nw, size=min(k, nw), replace=False))
            if i > 6 and rng.random() < 0.9:
                cols.add(0)  # the heavy column
            if i > 6 and i % 3 == which:
                cols.add(1 + which)  # three more columns of ~nc/3 entries each
            if 100 <= i < 164:
                cols.add(7)  # exactly 64 entries: not heavy
            if 200 <= i < 265:
                cols.add(8)  # 65: heavy
            rows.append(sorted(cols))
        nri, ci = [], []
        for r_ in rows:
            nri.append(len(ci))
            ci += r_
        vv = rng.integers(0, 17, size=len(ci))
        mats.append(SparseMatrix(nc, nw, np.array(nri, np.uint32), np.array(ci, np.uint32), vv.astype(np.uint32)))
    return mats


@pytest.mark.parametrize("nc,nw", [(3000, 5000), (9000, 4100)])
def test_r1cs_products_with_heavy_rows_and_columns(ctx, oracle, nc, nw):
    """lines longer than 64 entries are summed by workgroups before the gather (csrc/r1cs.hip "heavy lines"): every product, the
    satisfaction check and the single-matrix products still equal the oracle's row-by-row sums bit for bit"""
    import ctypes as C

    from provekit_amd import ProveKitHipError
    from provekit_amd._lib import lib
    from provekit_amd.field import random_field
    from provekit_amd.sparse_matrix import R1CS

    a, b, c = heavy_r1cs(nc, nw, nc)
    interner = random_field(17, 5)
    r = R1CS(ctx, a, b, c, interner)
    z = random_field(nw, 9)
    d_z = ctx.upload(z)
    m0 = (nc - 1).bit_length()
    da, db, dc = r.calculate_witness_bounds(d_z, m0)
    ea, eb, ec = (oracle.spmv(nc, nw, m.new_row_indices, m.col_indices, m.values, interner, z) for m in (a, b, c))
    zpad = np.zeros(((1 << m0) - nc, 4), np.uint64)
    assert np.array_equal(ctx.download_fe(da, 1 << m0), np.concatenate([ea, zpad]))
    assert np.array_equal(ctx.download_fe(db, 1 << m0), np.concatenate([eb, zpad]))
    assert np.array_equal(ctx.download_fe(dc, 1 << m0), np.concatenate([oracle.hadamard(ea, eb), zpad]))
    eq = random_field(1 << m0, 10)
    d_eq = ctx.upload(eq)
    out = ctx.download_fe(r.calculate_external_row_of_r1cs_matrices(d_eq), 3 * nw).reshape(3, nw, 4)
    for k, m in enumerate((a, b, c)):
        exp = oracle.spmv(nc, nw, m.new_row_indices, m.col_indices, m.values, interner, eq[:nc], transpose=True)
        assert np.array_equal(out[k], exp), k
        # the single-matrix entry points (pk_r1cs_matvec), both directions
        d_y = ctx.alloc_fe(max(nc, nw))
        ctx._check(lib.pk_r1cs_matvec(ctx.handle, r.handle, k, 0, d_z.ptr, d_y.ptr))
        assert np.array_equal(ctx.download_fe(d_y, nc), (ea, eb, ec)[k])
        ctx._check(lib.pk_r1cs_matvec(ctx.handle, r.handle, k, 1, d_eq.ptr, d_y.ptr))
        assert np.array_equal(ctx.download_fe(d_y, nw), exp)
    # satisfaction: random z does not satisfy; the first failing row is the oracle's
    want_bad = next(i for i in range(nc) if not np.array_equal(oracle.hadamard(ea[i : i + 1], eb[i : i + 1])[0], ec[i]))
    with pytest.raises(ProveKitHipError) as e:
        r.test_witness_satisfaction(d_z)
    assert e.value.row == want_bad
    r.close()


def test_r1cs_rejects_bad_input(ctx):
    from provekit_amd import ProveKitHipError
    from provekit_amd.field import random_field
    from provekit_amd.sparse_matrix import R1CS, SparseMatrix

    good = SparseMatrix(2, 2, np.array([0, 1], np.uint32), np.array([0, 1], np.uint32), np.array([0, 0], np.uint32))
    with pytest.raises(ValueError):  # caught by the Python mirror before the C ABI is reached
        SparseMatrix(2, 2, np.array([0, 1], np.uint32), np.array([0, 5], np.uint32), np.array([0, 0], np.uint32))
    bad_col = SparseMatrix(2, 2, np.array([0, 1], np.uint32), np.array([0, 1], np.uint32), np.array([0, 0], np.uint32))
    bad_col.col_indices = np.array([0, 5], np.uint32)  # past the Python check: pk_r1cs_create validates the contents itself
    bad_val = SparseMatrix(2, 2, np.array([0, 1], np.uint32), np.array([0, 1], np.uint32), np.array([0, 9], np.uint32))
    it = random_field(2, 1)
    with pytest.raises(ProveKitHipError):
        R1CS(ctx, good, bad_col, good, it)
    with pytest.raises(ProveKitHipError):  # "Value not in interner."
        R1CS(ctx, good, good, bad_val, it)


@pytest.mark.parametrize("bits", [0.0, 3.141592653589793, 10.0, 16.0, 20.0])
def test_pow_solve_check(ctx, oracle, bits):
    """pow.rs:105-112 round trip; the GPU returns the smallest valid nonce == the oracle's sequential search"""
    from provekit_amd.pow import SkyscraperPoW

    for challenge in (b"\xff" * 32, bytes(range(32))):
        p = SkyscraperPoW(challenge, bits, ctx=ctx)
        nonce = p.solve()
        assert p.check(nonce)
        ch = np.frombuffer(challenge, dtype=np.uint64)
        assert oracle.pow_verify(ch, bits, nonce)
        if bits <= 16.0:
            assert nonce == oracle.pow_solve(ch, bits)
        if bits > 0:
            assert not p.check(nonce + 1) or oracle.pow_verify(ch, bits, nonce + 1)
    with pytest.raises(ValueError):
        SkyscraperPoW(b"\0" * 32, 60.0, ctx=ctx)


@pytest.mark.parametrize("batch,n_vars", [(2, 8), (2, 13), (1, 12), (2, 16)])
def test_commit_root_and_openings(ctx, oracle, batch, n_vars):
    from provekit_amd.field import random_field
    from provekit_amd.whir import commit_batch, multipath_serialize

    polys = [random_field(1 << n_vars, 50 + b + n_vars) for b in range(batch)]
    c = commit_batch(ctx, [ctx.upload(p) for p in polys], n_vars, 1, 4)
    leaves = oracle.rs_encode(np.concatenate(polys), batch, n_vars, 1, 4)
    nodes = oracle.merkle_commit(leaves)
    n = leaves.shape[0]
    assert int.from_bytes(c.root, "little") == oracle.limbs_to_ints(nodes[1])[0]
    rng = np.random.default_rng(n_vars)
    idx = np.unique(rng.integers(0, n, size=min(40, n)))
    lv, sib, paths = c.open(idx, canonical_leaves=True)
    assert np.array_equal(lv, np.stack([oracle.from_mont(leaves[i]) for i in idx]))
    lv_m, _, _ = c.open(idx, canonical_leaves=False)
    assert np.array_equal(lv_m, leaves[idx])
    logn = n.bit_length() - 1
    for q, i in enumerate(idx):
        assert np.array_equal(sib[q], nodes[(n + i) ^ 1])
        for d in range(1, logn):
            anc = (n + i) >> (logn - d)
            assert np.array_equal(paths[q, d - 1], nodes[anc ^ 1])
    # every opening verifies against the root with the verifier's own recurrence (whir_utilities.go:13-46)
    for q, i in enumerate(idx[:5]):
        h = oracle.leaf_hash(leaves[i][None])[0]
        chain = [sib[q]] + [paths[q, d] for d in range(logn - 2, -1, -1)]
        j = int(i)
        for s in chain:
            m = np.concatenate([s, h] if j & 1 else [h, s]).astype("<u8").tobytes()
            h = np.frombuffer(oracle.compress_many(m), dtype=np.uint64)
            j >>= 1
        assert h.tobytes() == c.root
    blob = multipath_serialize(idx, sib, paths)
    assert len(blob) > 32 * len(idx)
    c.close()


def test_commit_full_size(ctx, oracle):
    """BASELINE config 2 commit (batch 2, n=21, rate 1/2, fold 16 -> 2^18 leaves x 32): root is reproducible, the
    opened leaves are codeword symbols (direct evaluation in the oracle), and the paths chain to the root."""
    from provekit_amd.field import random_field
    from provekit_amd.whir import commit_batch

    n_vars = 21
    polys = [random_field(1 << n_vars, 60 + b) for b in range(2)]
    bufs = [ctx.upload(p) for p in polys]
    c = commit_batch(ctx, bufs, n_vars)
    c2 = commit_batch(ctx, bufs, n_vars)
    assert c.root == c2.root
    c2.close()
    n = c.n_leaves
    assert n == 1 << 18 and c.width == 32
    idx = np.array([0, 1, 77777, n - 1], dtype=np.uint64)
    lv, sib, paths = c.open(idx, canonical_leaves=False)
    wroot = oracle.root_of_unity(18)
    for q, i in enumerate(idx):
        pw = np.empty(4, dtype=np.uint64)
        oracle.L.pko_fe_pow(oracle._p(wroot), int(i), oracle._p(pw))
        for b in range(2):
            for j in (0, 9, 15):
                assert np.array_equal(lv[q, 16 * b + j], oracle.eval_univariate(np.ascontiguousarray(polys[b][j::16]), pw))
        h = oracle.leaf_hash(lv[q][None])[0]
        chain = [sib[q]] + [paths[q, d] for d in range(16, -1, -1)]
        k = int(i)
        for s in chain:
            m = np.concatenate([s, h] if k & 1 else [h, s]).astype("<u8").tobytes()
            h = np.frombuffer(oracle.compress_many(m), dtype=np.uint64)
            k >>= 1
        assert h.tobytes() == c.root
    c.close()


def test_r1cs_from_postcard_matches_direct_upload(ctx, oracle):
    """pk_r1cs_from_postcard (the Rust caller's route: postcard::to_allocvec(&scheme.r1cs)) builds the same device R1CS as
    pk_r1cs_create: identical products; malformed bytes are rejected, never read out of bounds"""
    from provekit_amd import ProveKitHipError
    from provekit_amd import file as F
    from provekit_amd.field import random_field
    from provekit_amd.sparse_matrix import R1CS

    nc, nw = 700, 500
    a, b, c = synth_r1cs(nc, nw, 77)
    interner_canon = [int(x) for x in np.random.default_rng(1).integers(1, 2**62, size=17)]
    interner = oracle.to_mont(oracle.ints_to_limbs(interner_canon))
    direct = R1CS(ctx, a, b, c, interner)
    blob = F.encode_r1cs_postcard(2, interner_canon, [(nc, nw, m.new_row_indices, m.col_indices, m.values) for m in (a, b, c)])
    viapc = R1CS.from_postcard(ctx, blob)
    assert (viapc.num_constraints, viapc.num_witnesses, viapc.num_public_inputs, viapc.bytes_consumed) == (nc, nw, 2, len(blob))
    z = ctx.upload(random_field(nw, 3))
    m0 = 10
    for x, y in zip(direct.calculate_witness_bounds(z, m0), viapc.calculate_witness_bounds(z, m0)):
        assert np.array_equal(ctx.download_fe(x, 1 << m0), ctx.download_fe(y, 1 << m0))
    eq = ctx.upload(random_field(1 << m0, 4))
    assert np.array_equal(ctx.download_fe(direct.calculate_external_row_of_r1cs_matrices(eq), 3 * nw),
                          ctx.download_fe(viapc.calculate_external_row_of_r1cs_matrices(eq), 3 * nw))
    for bad in (blob[: len(blob) // 2], blob[:1] + b"\xff\xff\xff\xff\xff\xff\xff\xff\xff\x7f" + blob[2:], b""):
        with pytest.raises(ProveKitHipError):
            R1CS.from_postcard(ctx, bad if bad else b"\x00")
    # mutated bytes: an error or a well-formed R1CS, quickly and without giant allocations
    import random
    import time

    rnd = random.Random(4)
    t0, seen = time.time(), {"ok": 0, "err": 0}
    for trial in range(400):
        m = bytearray(blob)
        if trial % 3 == 0:
            m[rnd.randrange(len(m))] ^= 1 << rnd.randrange(8)
        elif trial % 3 == 1:
            pos = rnd.randrange(len(m))
            m[pos:pos] = bytes([0xff] * rnd.randrange(1, 10))
        else:
            pos = rnd.randrange(len(m))
            del m[pos : pos + rnd.randrange(1, 30)]
        try:
            R1CS.from_postcard(ctx, bytes(m)).close()
            seen["ok"] += 1
        except ProveKitHipError:
            seen["err"] += 1
    assert seen["err"] > 100 and time.time() - t0 < 60, seen
    direct.close()
    viapc.close()


@pytest.mark.parametrize("batch,n_vars", [(2, 13), (2, 14), (1, 14), (1, 15), (2, 17)])
def test_commit_across_the_hash_ready_boundary(ctx, oracle, batch, n_vars):
    """pk_commit holds its codeword hash-ready (32 * value) from 2^11 rows up and as Montgomery images below (tree.hip
    codeword_scaled): roots and BOTH opening forms must equal the oracle's on either side of the switch, and pk_commit_into
    (caller-owned buffers) must give the same root as pk_commit."""
    import ctypes as C

    from provekit_amd._lib import lib
    from provekit_amd.field import random_field
    from provekit_amd.whir import commit_batch

    polys = [random_field(1 << n_vars, 900 + b + n_vars) for b in range(batch)]
    bufs = [ctx.upload(p) for p in polys]
    c = commit_batch(ctx, bufs, n_vars, 1, 4)
    leaves = oracle.rs_encode(np.concatenate(polys), batch, n_vars, 1, 4)
    nodes = oracle.merkle_commit(leaves)
    n = leaves.shape[0]
    assert int.from_bytes(c.root, "little") == oracle.limbs_to_ints(nodes[1])[0]
    idx = np.unique(np.random.default_rng(n_vars).integers(0, n, size=25)).astype(np.uint64)
    lv_c, sib, paths = c.open(idx, canonical_leaves=True)
    lv_m, _, _ = c.open(idx, canonical_leaves=False)
    assert np.array_equal(lv_m, leaves[idx])
    assert np.array_equal(lv_c, np.stack([oracle.from_mont(leaves[i]) for i in idx]))
    assert np.array_equal(sib, nodes[(n + idx.astype(np.int64)) ^ 1])
    szs = [C.c_size_t() for _ in range(3)]
    ctx._check(lib.pk_commit_sizes(ctx.handle, batch, n_vars, 1, 4, *[C.byref(x) for x in szs]))
    assert szs[0].value == n * 16 * batch and szs[1].value == 2 * n
    d_l, d_n, d_s = (ctx.alloc_fe(x.value) for x in szs)
    ptrs = (C.c_void_p * batch)(*[b.ptr for b in bufs])
    root = (C.c_uint8 * 32)()
    from provekit_amd._lib import LEAVES_MONTGOMERY, LEAVES_SCALED32, CommitLayout

    lay = CommitLayout()
    ctx._check(lib.pk_commit_into(ctx.handle, ptrs, batch, n_vars, 1, 4, d_l.ptr, d_n.ptr, d_s.ptr, root, C.byref(lay)))
    assert bytes(root) == c.root
    assert np.array_equal(ctx.download_fe(d_n.view_fe(1), 2 * n - 1), nodes[1:])  # every node of the heap
    # the layout the commit reports: hash-ready from 2^11 rows, Montgomery below; one shard outside a device set
    assert (lay.n_shards, lay.shard) == (1, 0)
    assert lay.encoding == (LEAVES_SCALED32 if n >= 2048 else LEAVES_MONTGOMERY)
    tl = CommitLayout()
    assert lib.pk_tree_layout(c.handle, C.byref(tl)) == 0 and (tl.n_shards, tl.shard, tl.encoding) == (1, 0, lay.encoding)
    # openings of the raw buffers under that layout == the handle's openings == the oracle's
    k, w, plen = len(idx), 16 * batch, int(np.log2(n)) - 1
    for canon, want in ((1, lv_c), (0, lv_m)):
        lo = np.zeros((k, w, 4), dtype=np.uint64)
        so = np.zeros((k, 4), dtype=np.uint64)
        po = np.zeros((k, plen, 4), dtype=np.uint64)
        ctx._check(lib.pk_commit_open(ctx.handle, d_l.ptr, d_n.ptr, n, w, C.byref(lay), idx.ctypes.data, k, canon, lo.ctypes.data, so.ctypes.data, po.ctypes.data))
        assert np.array_equal(lo, want) and np.array_equal(so, sib) and np.array_equal(po, paths)
        ro = np.zeros((k, w, 4), dtype=np.uint64)
        ctx._check(lib.pk_gather_leaves_enc(ctx.handle, d_l.ptr, n, w, 1, lay.encoding, idx.ctypes.data, k, canon, ro.ctypes.data))
        assert np.array_equal(ro, want)
    c.close()
