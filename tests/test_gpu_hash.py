"""GPU parity: Skyscraper compress / leaf hash / Merkle tree
through the C ABI vs the CPU oracle, bit-exact."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def H(h):
    return int(h, 16)


def test_library_is_the_hip_build(ctx):
    from provekit_amd import _lib

    assert os.path.exists(_lib.LIB_PATH) and _lib.lib.pk_abi_version() == 2


def test_field_ops_vs_oracle(ctx, oracle):
    from provekit_amd._lib import lib
    from provekit_amd.field import random_field

    n = 5000
    a, b = random_field(n, 1), random_field(n, 2)
    # edge values
    a[:4] = oracle.ints_to_limbs([0, 1, oracle.P - 1, oracle.P - 2])
    b[:4] = oracle.ints_to_limbs([0, oracle.P - 1, oracle.P - 1, 1])
    da, db, do = ctx.upload(a), ctx.upload(b), ctx.alloc_fe(n)
    for name in ("add", "sub", "mul"):
        ctx._check(getattr(lib, f"pk_fe_{name}")(ctx.handle, da.ptr, db.ptr, do.ptr, n))
        got = ctx.download_fe(do, n)
        exp = oracle.binop(f"pko_fe_{name}", a, b)
        assert np.array_equal(got, exp), name
    ctx._check(lib.pk_fe_from_mont(ctx.handle, da.ptr, do.ptr, n))
    assert np.array_equal(ctx.download_fe(do, n), oracle.from_mont(a))
    ctx._check(lib.pk_fe_to_mont(ctx.handle, da.ptr, do.ptr, n))
    assert np.array_equal(ctx.download_fe(do, n), oracle.to_mont(a))


@pytest.mark.parametrize("version", [2, 1])
def test_compress_golden_vectors(ctx, version):
    from provekit_amd.skyscraper import compress_many

    vec = json.load(open(os.path.join(G, "vectors.json")))["compress_v2" if version == 2 else "compress_v1"]
    msgs = b"".join(H(a).to_bytes(32, "little") + H(b).to_bytes(32, "little") for a, b, _ in vec)
    ctx.set_hash_version(version)
    try:
        out = compress_many(msgs, ctx=ctx)
    finally:
        ctx.set_hash_version(2)
    assert out == b"".join(H(e).to_bytes(32, "little") for _, _, e in vec)


def test_compress_kats(ctx):
    """permute(0,0) / permute(random) KATs (reference.rs:155-187) through compress = permute.0 + l"""
    from provekit_amd.skyscraper import compress_many

    k = json.load(open(os.path.join(G, "skyscraper_kats.json")))["permute"]
    P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    msgs = b"".join((int(x["l"]) % 2**256).to_bytes(32, "little") + int(x["r"]).to_bytes(32, "little") for x in k)
    out = compress_many(msgs, ctx=ctx)
    for i, x in enumerate(k):
        assert int.from_bytes(out[32 * i : 32 * i + 32], "little") == (int(x["el"]) + int(x["l"])) % P


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 1000, 70001])
def test_compress_many_random_vs_oracle(ctx, oracle, n):
    """skyscraper/core/src/block4.rs:40-54 pattern: fast implementation == reference on random bytes"""
    from provekit_amd.skyscraper import compress_many

    msgs = np.random.default_rng(n).integers(0, 256, size=64 * n, dtype=np.uint8).tobytes()
    assert compress_many(msgs, ctx=ctx) == oracle.compress_many(msgs)


def test_compress_many_length_errors(ctx):
    from provekit_amd.skyscraper import compress_many

    with pytest.raises(ValueError):
        compress_many(b"\0" * 65, ctx=ctx)
    with pytest.raises(ValueError):
        compress_many(b"\0" * 64, hashes=bytearray(31), ctx=ctx)
    with pytest.raises(ValueError):
        compress_many(b"\0" * 64, hashes=bytearray(64), ctx=ctx)


@pytest.mark.parametrize("width", [1, 2, 16, 32])
def test_leaf_hash_both_layouts(ctx, oracle, width):
    from provekit_amd._lib import PK_COL_MAJOR, PK_LEAF_MAJOR, lib
    from provekit_amd.field import random_field

    n = 777
    leaves = random_field(n * width, 10 + width).reshape(n, width, 4)
    exp = oracle.leaf_hash(leaves)
    d_out = ctx.alloc_fe(n)
    d_l = ctx.upload(leaves)
    ctx._check(lib.pk_leaf_hash(ctx.handle, d_l.ptr, n, width, PK_LEAF_MAJOR, d_out.ptr))
    assert np.array_equal(ctx.download_fe(d_out, n), exp)
    d_c = ctx.upload(np.ascontiguousarray(leaves.transpose(1, 0, 2)))
    ctx._check(lib.pk_leaf_hash(ctx.handle, d_c.ptr, n, width, PK_COL_MAJOR, d_out.ptr))
    assert np.array_equal(ctx.download_fe(d_out, n), exp)


def test_leaf_hash_empty_leaf_is_an_error(ctx):
    from provekit_amd import ProveKitHipError
    from provekit_amd._lib import lib

    d = ctx.alloc_fe(1)
    with pytest.raises(ProveKitHipError):
        ctx._check(lib.pk_leaf_hash(ctx.handle, d.ptr, 1, 0, 0, d.ptr))


@pytest.mark.parametrize("log_n,width", [(0, 3), (1, 16), (5, 32), (10, 16), (13, 32)])
def test_merkle_tree_vs_oracle(ctx, oracle, log_n, width):
    from provekit_amd.field import random_field
    from provekit_amd.merkle import MerkleTree

    n = 1 << log_n
    leaves = random_field(n * width, 77 + log_n).reshape(n, width, 4)
    t = MerkleTree(leaves, ctx=ctx)
    exp = oracle.merkle_commit(leaves)
    assert np.array_equal(t.all_nodes()[1:], exp[1:])
    assert np.array_equal(t.root(), exp[1])


def test_merkle_rejects_non_power_of_two(ctx):
    from provekit_amd.field import random_field
    from provekit_amd.merkle import MerkleTree

    with pytest.raises(ValueError):
        MerkleTree(random_field(3 * 2, 1).reshape(3, 2, 4), ctx=ctx)


def test_fixture_blinding_tree_v1_root(ctx, oracle):
    """Rebuild the fully-opened 32-leaf tree of the reference's proof fixture on the GPU (Skyscraper v1)."""
    from provekit_amd.merkle import MerkleTree

    fix = json.load(open(os.path.join(G, "fixture_merkle.json")))
    t = fix["trees"][0]
    leaves = np.stack([oracle.to_mont(oracle.hex_to_limbs(l)) for l in t["leaves"]])
    leaves = leaves[np.argsort(t["multipath"]["leaf_indexes"])]
    ctx.set_hash_version(1)
    try:
        root = MerkleTree(leaves, ctx=ctx).root()
    finally:
        ctx.set_hash_version(2)
    assert oracle.limbs_to_ints(root)[0] == H(t["root"])


def test_fixture_openings_v1_leaf_digests(ctx, oracle):
    """Leaf digests of opened leaves from the big witness tree (2^18 leaves x 32) chain to the fixture root."""
    from provekit_amd.skyscraper import compress_many, leaf_hash

    fix = json.load(open(os.path.join(G, "fixture_merkle.json")))
    ctx.set_hash_version(1)
    try:
        for t in fix["trees"][2:]:
            mp = t["multipath"]
            leaves = np.stack([oracle.to_mont(oracle.hex_to_limbs(l)) for l in t["leaves"]])
            digs = leaf_hash(leaves, ctx=ctx)
            for h, idx, sib, path in zip(digs, mp["leaf_indexes"], mp["leaf_sibling_hashes"], mp["auth_paths_root_to_leaf"]):
                chain = [H(sib)] + [H(p) for p in reversed(path)]
                cur = oracle.limbs_to_ints(h)[0]
                i = idx
                for s in chain:
                    l, r = (s, cur) if i & 1 else (cur, s)
                    out = compress_many(l.to_bytes(32, "little") + r.to_bytes(32, "little"), ctx=ctx)
                    cur = int.from_bytes(out, "little")
                    i >>= 1
                assert cur == H(t["root"])
    finally:
        ctx.set_hash_version(2)


def test_merkle_full_size_property(ctx, oracle):
    """BASELINE config-2 size (2^18 leaves x 32 FE): size-independent properties -- the root depends on
    every leaf (flip one element -> different root), is reproducible, and spot-checked subtrees match
    the oracle."""
    from provekit_amd._lib import PK_COL_MAJOR, lib
    from provekit_amd.field import random_field

    n, w = 1 << 18, 32
    cols = random_field(n * w, 1234).reshape(w, n, 4)  # column-major, as pk_commit keeps it
    d_c = ctx.upload(cols)
    nodes = ctx.alloc_fe(2 * n)
    ctx._check(lib.pk_merkle_commit(ctx.handle, d_c.ptr, n, w, PK_COL_MAJOR, nodes.ptr))
    all1 = ctx.download_fe(nodes, 2 * n)
    ctx._check(lib.pk_merkle_commit(ctx.handle, d_c.ptr, n, w, PK_COL_MAJOR, nodes.ptr))
    assert np.array_equal(all1, ctx.download_fe(nodes, 2 * n))
    # spot-check: 64 leaf digests and the inner-node recurrence on random nodes against the oracle
    rng = np.random.default_rng(5)
    pick = rng.integers(0, n, size=64)
    exp = oracle.leaf_hash(np.ascontiguousarray(cols[:, pick].transpose(1, 0, 2)))
    assert np.array_equal(all1[n + pick], exp)
    inner = rng.integers(1, n, size=256)
    msgs = np.concatenate([all1[2 * inner], all1[2 * inner + 1]], axis=1).astype("<u8").tobytes()
    assert np.frombuffer(oracle.compress_many(msgs), dtype=np.uint64).reshape(-1, 4).tolist() == all1[inner].tolist()
    # sensitivity
    cols2 = cols.copy()
    cols2[17, 12345, 0] ^= np.uint64(1)
    d_c2 = ctx.upload(cols2)
    ctx._check(lib.pk_merkle_commit(ctx.handle, d_c2.ptr, n, w, PK_COL_MAJOR, nodes.ptr))
    assert not np.array_equal(ctx.download_fe(nodes.view_fe(1), 1)[0], all1[1])


def test_compress_structured_inputs(ctx, oracle):
    """The scaled-by-32 hash kernels have their own boundary conditions (the exact division by 32, the conditional subtraction
    of 32p, quotient estimates up to 169): structured inputs aimed at them -- multiples of p/32 and of p, powers of two, all byte
    patterns the S-box treats specially, values just below 2^256 -- each with small offsets, in every (l, r) combination of a
    sample, on the device against the oracle; plus 2^20 random messages."""
    from provekit_amd.skyscraper import compress_many

    P = oracle.P
    base = [0, 1, P - 1, P, P + 1, 2 * P, 5 * P, (1 << 256) - 1, (1 << 255), (1 << 254), (1 << 253), (1 << 232), (1 << 29) - 1]
    base += [k * P // 32 for k in (1, 2, 3, 15, 16, 17, 31, 32, 33, 100, 169)] + [k * P for k in (3, 4)]
    base += [int.from_bytes(bytes([b]) * 32, "little") for b in (0x00, 0xFF, 0x80, 0x7F, 0x01, 0xFE, 0x55, 0xAA, 0x0F, 0xF0)]
    base += [(1 << 261) // 32 // 5, ((1 << 256) - 1) // 3]
    vals = sorted({(v + d) % (1 << 256) for v in base for d in (-2, -1, 0, 1, 2)})
    rng = np.random.default_rng(8)
    pick = [vals[i] for i in rng.permutation(len(vals))[:60]]
    pairs = [(a, b) for a in pick for b in pick] + [(v, v) for v in vals] + [(v, 0) for v in vals] + [(0, v) for v in vals]
    msgs = b"".join(a.to_bytes(32, "little") + b.to_bytes(32, "little") for a, b in pairs)
    assert compress_many(msgs, ctx=ctx) == oracle.compress_many(msgs)
    big = rng.integers(0, 256, size=64 << 20, dtype=np.uint8).tobytes()
    assert compress_many(big, ctx=ctx) == oracle.compress_many(big)
