// hash.hip -- Skyscraper batch compression and Merkle-tree kernels.
//
// One lane = one compression chain.  The work is integer-ALU bound (14 Montgomery
// squarings + 4 bars per v2 compression: ~1.9k v_mad_u64_u32 and ~3.6k other VALU
// instructions per lane), so the kernels are organised for occupancy and
// coalesced 32-byte-per-lane traffic, not for LDS reuse:
//   compress_many : lane i reads message i (64 B contiguous), writes hash i (32 B).
//   leaf_hash     : lane i owns leaf i; in PK_COL_MAJOR (the layout pk_commit keeps
//                   in HBM) column j of all leaves is contiguous, so every step of
//                   the 31-deep left fold is one coalesced 2 KiB wave read.
//   merkle_levels : lane i owns inner node i of the widest of up to 5 fused levels;
//                   children 2i,2i+1 are adjacent in the heap, so a wave reads 4 KiB
//                   contiguous; the workgroup then walks up its own subtree.
#include "ctx.hpp"
#include "skyscraper29s.hpp"

using namespace pk;

template <int VERSION>
__global__ __launch_bounds__(256) void compress_many_kernel(const fe* __restrict__ msgs, fe* __restrict__ out, size_t n) {
    PK_LATENCY_PRIO();
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        fe29 l = to_scaled29(fe_load(msgs + 2 * i));  // any 256-bit value (generic.rs:81-82: inputs >= p are legal)
        fe29 r = to_scaled29(fe_load(msgs + 2 * i + 1));
        fe_store(out + i, from_scaled_canon(compress29s<VERSION>(l, r)));
    }
}

// SkyscraperCRH::evaluate (provekit/common/src/skyscraper/whir.rs:30-48): h = x0; h = C(h, x_j)
// SCALED_IN: the leaves are in the hash-ready encoding the commit's own NTT emits (32 * value as a plain integer < p,
// ntt.hip ntt_scaled_available): they enter the fold as they are.  Otherwise they are Montgomery images (the C ABI form).
template <int VERSION, int LAYOUT, bool SCALED_IN>
__global__ __launch_bounds__(256) void leaf_hash_kernel(const fe* __restrict__ leaves, size_t n_leaves, unsigned width,
                                                        fe* __restrict__ digests) {
    PK_LATENCY_PRIO();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_leaves) return;
    size_t step = LAYOUT == PK_COL_MAJOR ? n_leaves : 1;
    const fe* p = leaves + (LAYOUT == PK_COL_MAJOR ? i : i * (size_t)width);
    // into_bigint() (whir.rs:21) and the hash's internal scaling in one Montgomery reduction: x*2^256 -> 32x
    fe29 h = SCALED_IN ? unpack29<0>(fe_load(p)) : mont_to_scaled29(fe_load(p));
    fe nxt = width > 1 ? fe_load(p + step) : fe_zero();
    for (unsigned j = 1; j < width; j++) {
        fe29 x = SCALED_IN ? unpack29<0>(nxt) : mont_to_scaled29(nxt);
        if (j + 1 < width) nxt = fe_load(p + (size_t)(j + 1) * step);  // prefetch next column
        h = compress29s<VERSION>(h, x);  // the fold never leaves the scaled domain
    }
    fe_store(digests + i, from_scaled_canon(h));  // exact canonical form once per leaf (also for width 1)
}

// `levels` consecutive levels of ark MerkleTree::new in one launch: nodes[i] = C(nodes[2i], nodes[2i+1]).  A workgroup
// owns 256 adjacent nodes of the widest level (node count `count`, heap offset = count) and the 128, 64, ... nodes above
// them; whole wavefronts retire as the subtree narrows, so nothing is wasted, and a level costs one compression latency
// instead of a launch.
template <int VERSION>
__global__ __launch_bounds__(256) void merkle_levels_kernel(fe* __restrict__ nodes, size_t count, unsigned levels) {
    if (gridDim.x <= 128u) __builtin_amdgcn_s_setprio(3);  // a chain of dependent levels: latency-bound at any width
    else __builtin_amdgcn_s_setprio(2);
    size_t base = (size_t)blockIdx.x * 256;  // first owned node of the widest level, relative to that level
    unsigned width = 256;
    for (unsigned l = 0; l < levels; l++) {
        if (threadIdx.x < width && base + threadIdx.x < count) {
            size_t i = count + base + threadIdx.x;
            fe29 a = to_scaled29(fe_load(nodes + 2 * i)), b = to_scaled29(fe_load(nodes + 2 * i + 1));  // digests are canonical
            fe_store(nodes + i, from_scaled_canon(compress29s<VERSION>(a, b)));
        }
        __threadfence_block();
        __syncthreads();
        count >>= 1;
        base >>= 1;
        width >>= 1;
    }
}

// the top of the tree (<= 512 nodes per level) in one workgroup; also clears the unused heap slot 0 and mirrors the root
// into pinned host memory (the host reads it after the stream synchronisation, no copy operation)
template <int VERSION>
__global__ __launch_bounds__(512) void merkle_top_kernel(fe* __restrict__ nodes, size_t top_leaves, fe* __restrict__ host_root) {
    PK_LATENCY_PRIO();
    for (size_t lvl = top_leaves / 2; lvl >= 1; lvl >>= 1) {
        if (threadIdx.x < lvl) {
            size_t i = lvl + threadIdx.x;
            fe29 l = to_scaled29(fe_load(nodes + 2 * i)), r = to_scaled29(fe_load(nodes + 2 * i + 1));
            fe x = from_scaled_canon(compress29s<VERSION>(l, r));
            fe_store(nodes + i, x);
            if (i == 1 && host_root) fe_store(host_root, x);
        }
        __threadfence_block();
        __syncthreads();
    }
    if (threadIdx.x == 0) fe_store(nodes, fe_zero());
}

static int leaf_hash_launch(pk_ctx* ctx, const uint64_t* d_leaves, size_t n_leaves, size_t width, int layout, uint64_t* d_digests,
                            bool scaled_in) {
    ProfScope prof(ctx, "leaf_hash");
    unsigned block = 256;
    unsigned grid = (unsigned)((n_leaves + block - 1) / block);
    const fe* L = (const fe*)d_leaves;
    fe* D = (fe*)d_digests;
    unsigned w = (unsigned)width;
#define PK_LH(V, LAY, SC) leaf_hash_kernel<V, LAY, SC><<<grid, block, 0, ctx->stream>>>(L, n_leaves, w, D)
    if (ctx->hash_version == 2) {
        if (layout == PK_COL_MAJOR) {
            if (scaled_in) PK_LH(2, PK_COL_MAJOR, true);
            else PK_LH(2, PK_COL_MAJOR, false);
        } else {
            PK_LH(2, PK_LEAF_MAJOR, false);
        }
    } else {
        if (layout == PK_COL_MAJOR) {
            if (scaled_in) PK_LH(1, PK_COL_MAJOR, true);
            else PK_LH(1, PK_COL_MAJOR, false);
        } else {
            PK_LH(1, PK_LEAF_MAJOR, false);
        }
    }
#undef PK_LH
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}

namespace pk {
// leaf hash of a column-major codeword in the commit's internal (hash-ready) encoding, or in Montgomery form
int leaf_hash_x(pk_ctx* ctx, const uint64_t* d_leaves, size_t n_leaves, size_t width, uint64_t* d_digests, bool scaled_in) {
    if (!n_leaves) return PK_OK;
    return leaf_hash_launch(ctx, d_leaves, n_leaves, width, PK_COL_MAJOR, d_digests, scaled_in);
}
// root of the tree pk_merkle_inner / pk_merkle_commit just built on this context's stream
int read_root(pk_ctx* ctx, const uint64_t* d_nodes, size_t n_leaves, uint64_t root[4]) {
    if (n_leaves < 2) return pk_memcpy_d2h(ctx, root, d_nodes + 4, 32);
    int rc = sync_stream(ctx);
    if (rc) return rc;
    memcpy(root, (char*)ctx->h_pinned + PK_PIN_ROOT, 32);
    return PK_OK;
}
}  // namespace pk

extern "C" {

int pk_compress_many(pk_ctx* ctx, const uint8_t* d_messages, uint8_t* d_hashes, size_t n) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, n == 0 || (d_messages && d_hashes), "null pointer");
    if (!n) return PK_OK;
    ProfScope prof(ctx, "compress_many");
    unsigned grid = grid_for(ctx, n, 256, 16);
    if (ctx->hash_version == 2)
        compress_many_kernel<2><<<grid, 256, 0, ctx->stream>>>((const fe*)d_messages, (fe*)d_hashes, n);
    else
        compress_many_kernel<1><<<grid, 256, 0, ctx->stream>>>((const fe*)d_messages, (fe*)d_hashes, n);
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}

int pk_compress_many_host(pk_ctx* ctx, const uint8_t* messages, size_t messages_len, uint8_t* hashes, size_t hashes_len) {
    PK_ENTER(ctx);
    // generic.rs:18-25: "Message length not a multiple of 64" / "Hashes length not a multiple of 32" / mismatch
    PK_REQUIRE(ctx, messages_len % 64 == 0, "message length not a multiple of 64");
    PK_REQUIRE(ctx, hashes_len % 32 == 0, "hashes length not a multiple of 32");
    PK_REQUIRE(ctx, messages_len == 2 * hashes_len, "messages and hashes length mismatch");
    size_t n = hashes_len / 32;
    if (!n) return PK_OK;
    PK_REQUIRE(ctx, messages && hashes, "null pointer");
    void *dm = nullptr, *dh = nullptr;
    int rc = pk_malloc(ctx, messages_len, &dm);
    if (rc) return rc;
    rc = pk_malloc(ctx, hashes_len, &dh);
    if (rc) {
        pk_free(ctx, dm);
        return rc;
    }
    rc = pk_memcpy_h2d(ctx, dm, messages, messages_len);
    if (!rc) rc = pk_compress_many(ctx, (const uint8_t*)dm, (uint8_t*)dh, n);
    if (!rc) rc = pk_memcpy_d2h(ctx, hashes, dh, hashes_len);
    pk_free(ctx, dm);
    pk_free(ctx, dh);
    return rc;
}

int pk_leaf_hash(pk_ctx* ctx, const uint64_t* d_leaves, size_t n_leaves, size_t width, int layout, uint64_t* d_digests) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, width >= 1, "leaf width must be >= 1 (IncorrectInputLength(0))");  // whir.rs:47
    PK_REQUIRE(ctx, width < (1u << 20), "leaf width too large");
    PK_REQUIRE(ctx, layout == PK_LEAF_MAJOR || layout == PK_COL_MAJOR, "unknown layout");
    PK_REQUIRE(ctx, n_leaves == 0 || (d_leaves && d_digests), "null pointer");
    if (!n_leaves) return PK_OK;
    return leaf_hash_launch(ctx, d_leaves, n_leaves, width, layout, d_digests, false);
}

int pk_merkle_inner(pk_ctx* ctx, uint64_t* d_nodes, size_t n_leaves) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, is_pow2(n_leaves), "n_leaves must be a power of two");  // ark MerkleTree::new asserts this
    PK_REQUIRE(ctx, d_nodes, "null pointer");
    fe* N = (fe*)d_nodes;
    int rc = ensure_pinned(ctx);
    if (rc) return rc;
    fe* host_root = (fe*)((char*)ctx->h_pinned + PK_PIN_ROOT);
    ProfScope prof(ctx, "merkle_inner");
    if (n_leaves == 1) {  // the leaf digest is the root (heap slot 1); nothing to hash
        PK_HIP(ctx, hipMemsetAsync(d_nodes, 0, 32, ctx->stream));
        return PK_OK;
    }
    size_t lvl = n_leaves / 2;
    while (lvl > 512) {
        unsigned levels = 0;
        for (size_t c = lvl; c > 512 && levels < 5; c >>= 1) levels++;
        unsigned grid = (unsigned)((lvl + 255) / 256);
        if (ctx->hash_version == 2)
            merkle_levels_kernel<2><<<grid, 256, 0, ctx->stream>>>(N, lvl, levels);
        else
            merkle_levels_kernel<1><<<grid, 256, 0, ctx->stream>>>(N, lvl, levels);
        lvl >>= levels;
    }
    if (ctx->hash_version == 2)
        merkle_top_kernel<2><<<1, 512, 0, ctx->stream>>>(N, lvl * 2, host_root);
    else
        merkle_top_kernel<1><<<1, 512, 0, ctx->stream>>>(N, lvl * 2, host_root);
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}

}  // extern "C"

namespace pk {
// the levels above heap slots [top_leaves, 2 top_leaves) (top_leaves <= 1024, a power of two), the root also to the pinned page:
// the top of a subtree-sharded tree, whose G subtree roots arrive through an all-gather (tree.hip)
int merkle_top_x(pk_ctx* ctx, uint64_t* d_nodes, size_t top_leaves) {
    int rc = ensure_pinned(ctx);
    if (rc) return rc;
    fe* host_root = (fe*)((char*)ctx->h_pinned + PK_PIN_ROOT);
    ProfScope prof(ctx, "merkle_inner");
    if (ctx->hash_version == 2)
        merkle_top_kernel<2><<<1, 512, 0, ctx->stream>>>((fe*)d_nodes, top_leaves, host_root);
    else
        merkle_top_kernel<1><<<1, 512, 0, ctx->stream>>>((fe*)d_nodes, top_leaves, host_root);
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}
}  // namespace pk

extern "C" {

int pk_merkle_commit(pk_ctx* ctx, const uint64_t* d_leaves, size_t n_leaves, size_t width, int layout, uint64_t* d_nodes) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, is_pow2(n_leaves), "n_leaves must be a power of two");
    PK_REQUIRE(ctx, d_nodes, "null pointer");
    int rc = pk_leaf_hash(ctx, d_leaves, n_leaves, width, layout, d_nodes + 4 * n_leaves);
    if (rc) return rc;
    return pk_merkle_inner(ctx, d_nodes, n_leaves);
}

}  // extern "C"
