// tree.hip -- commitment handle: RS-encode + Merkle commit, and STIR openings (SURVEY 8a rows N1+N2+M1+M2, Q1).
//
// pk_commit replaces the data-parallel body of whir's CommitmentWriter::commit_batch
// (call site provekit/prover/src/whir_r1cs.rs:200-206): encode the batch, hash the leaves, build the
// tree, hand back the root.  The codeword matrix (column-major) and every tree level stay resident
// in HBM behind the pk_tree handle because STIR queries open them later (rows Q1); at P size that is
// 256 MiB + 16 MiB, at 2^26 it is 8.5 GiB -- far inside 288 GB, so nothing is recomputed or spilled.
//
// pk_tree_open replaces MerkleTree::generate_multi_proof + the leaf gather (ark-crypto-primitives
// 0.5, external); pk_multipath_serialize writes ark's MultiPath wire form (mirrored at
// recursive-verifier/app/circuit/types.go:17-22, decoded by utilities.go:71-82).
#include <algorithm>
#include <vector>

// memory- / latency-bound kernels: their wavefronts issue ahead of the ALU-bound hash / NTT / grinder kernels they share SIMDs with
#define PK_BASE_PRIO 2
#include "ctx.hpp"
#include "shard_map.hpp"
#include "skyscraper29s.hpp"

using namespace pk;

struct pk_tree {
    fe* d_leaves = nullptr;  // column-major [width][n_leaves / n_shards], Montgomery; owned unless borrowed
    fe* d_nodes = nullptr;   // heap of 2*n_leaves canonical digests (replicated on every rank of a sharded commit)
    size_t n_leaves = 0, width = 0;
    bool owns_leaves = true;
    int layout = PK_COL_MAJOR;
    // sharded commit: this rank holds the codeword rows i = shard (mod n_shards), local row t = i / n_shards
    unsigned shard = 0, n_shards = 1;
    // the codeword is held in the hash-ready encoding the commit's NTT emits (32 * value, plain integer < p) instead of
    // Montgomery images: the leaf hash consumes it without conversion, openings convert the ~100 opened rows
    bool scaled = false;
};

namespace pk {
int rs_encode_x(pk_ctx* ctx, const uint64_t* const* d_coeffs, unsigned batch, unsigned n_vars, unsigned log_inv_rate, unsigned fold,
                uint64_t* d_leaves, uint64_t* d_scratch, bool scaled);
int rs_encode_shard_x(pk_ctx* ctx, const uint64_t* const* d_coeffs, unsigned batch, unsigned n_vars, unsigned log_inv_rate, unsigned fold,
                      unsigned shard, unsigned n_shards, uint64_t* d_leaves_local, uint64_t* d_scratch, bool scaled);
int leaf_hash_x(pk_ctx* ctx, const uint64_t* d_leaves, size_t n_leaves, size_t width, uint64_t* d_digests, bool scaled_in);
int merkle_top_x(pk_ctx* ctx, uint64_t* d_nodes, size_t top_leaves);  // hash.hip: the levels above heap slots [top_leaves, 2 top_leaves)
bool ntt_scaled_available(unsigned log_n);
}

namespace {

// an element of the stored codeword -> what an opening returns: canonical (the ark-serialize form) or Montgomery
__device__ __forceinline__ fe opened_element(const fe& x, bool scaled, int canonical) {
    if (!scaled) return canonical ? fe_from_montx(x) : x;
    if (canonical) return from_scaled_canon(unpack29<0>(x));  // 32v -> v
    fe k;  // 2^507 mod p: mont(32v, 2^507) = 32v * 2^507 * 2^-256 = v * 2^256
    k.v[0] = 0xc0f10b6eu; k.v[1] = 0x95e64f0du; k.v[2] = 0x2e33c2c0u; k.v[3] = 0xf2087f4eu;
    k.v[4] = 0x7fcae90cu; k.v[5] = 0xc4610290u; k.v[6] = 0x4be93745u; k.v[7] = 0x25df13cfu;
    return fe_mulx(x, k);
}

// One launch serves an opening: the first k*width lanes gather the opened leaves to leaf-major order (optionally
// converted to canonical, the ark-serialize form), the next k*(plen+1) lanes the sibling digests -- out_sib[q] =
// nodes[(n+i)^1]; out_path[q][d-1] = sibling of the depth-d ancestor, d = 1..logn-1 (root -> leaf).  idx and the three
// outputs live in the context's pinned mailbox: the kernel reads the indices from and writes the openings to host memory
// directly, so an opening costs no copy operations.
__global__ __launch_bounds__(256) void gather_opening_kernel(const fe* __restrict__ leaves, const fe* __restrict__ nodes, size_t n_leaves,
                                                             unsigned width, int layout, unsigned logn,
                                                             const unsigned long long* __restrict__ idx, size_t k, int canonical,
                                                             fe* __restrict__ out_leaves, fe* __restrict__ out_sib, fe* __restrict__ out_path,
                                                             bool scaled) {
    PK_LATENCY_PRIO();
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n1 = k * width;
    if (t < n1) {
        size_t q = t / width, j = t % width;
        size_t i = idx[q];
        fe x = fe_load(leaves + (layout == PK_COL_MAJOR ? j * n_leaves + i : i * (size_t)width + j));
        fe_store(out_leaves + t, opened_element(x, scaled, canonical));
        return;
    }
    t -= n1;
    const unsigned plen = logn ? logn - 1 : 0;
    if (t >= k * (size_t)(plen + 1)) return;
    size_t q = t / (plen + 1), d = t % (plen + 1);
    size_t node = n_leaves + idx[q];
    if (d == plen) {
        if (logn) fe_store(out_sib + q, fe_load(nodes + (node ^ 1)));
    } else {
        size_t anc = node >> (logn - (d + 1));
        fe_store(out_path + q * plen + d, fe_load(nodes + (anc ^ 1)));
    }
}

// the all-gather delivers rank g's digests as one block: leaf i = g + G t is gathered[g*loc + t]; the heap wants it at rows + i
__global__ __launch_bounds__(256) void interleave_digests_kernel(const fe* __restrict__ gathered, fe* __restrict__ nodes, size_t rows, unsigned G) {
    PK_LATENCY_PRIO();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    fe_store(nodes + rows + i, fe_load(gathered + shard_gathered_slot(i, rows, G)));
}

// a rank's subtree, built as a compact heap H over loc = rows / G leaves (H[1] its root), copied into its place in the tree's
// heap: local node y of the level with c' nodes (H slots [c', 2c')) is global node G c' + g c' + (y - c')
__global__ __launch_bounds__(256) void scatter_subtree_kernel(const fe* __restrict__ H, fe* __restrict__ nodes, size_t loc, unsigned G, unsigned g) {
    PK_LATENCY_PRIO();
    const size_t y = (size_t)blockIdx.x * blockDim.x + threadIdx.x + 1;  // 1 .. loc - 1: the inner nodes (the leaf layer is in place)
    if (y >= loc) return;
    size_t c = 1;
    while (2 * c <= y) c <<= 1;
    fe_store(nodes + (size_t)G * c + (size_t)g * c + (y - c), fe_load(H + y));
}
__global__ void place_subtree_roots_kernel(const fe* __restrict__ roots, fe* __restrict__ nodes, unsigned G) {
    if (threadIdx.x < G) fe_store(nodes + G + threadIdx.x, fe_load(roots + threadIdx.x));
}

// sibling digests and authentication paths of a subtree-sharded tree: the nodes this rank holds, into a ZEROED device buffer
// laid out [k siblings | k x plen path nodes] like gather_opening_kernel's outputs; the all-reduce that follows completes them
__global__ __launch_bounds__(256) void gather_owned_nodes_kernel(const fe* __restrict__ nodes, size_t n_leaves, unsigned logn, unsigned G, unsigned shard,
                                                                 const unsigned long long* __restrict__ idx, size_t k, fe* __restrict__ out_sib,
                                                                 fe* __restrict__ out_path) {
    PK_LATENCY_PRIO();
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned plen = logn ? logn - 1 : 0;
    if (t >= k * (size_t)(plen + 1)) return;
    const size_t q = t / (plen + 1), d = t % (plen + 1);
    const size_t node = n_leaves + idx[q];
    const size_t x = d == plen ? (node ^ 1) : ((node >> (logn - (d + 1))) ^ 1);
    // the leaf layer is complete on every rank (it came through the all-gather): rank 0 answers for it, as for the replicated top
    const unsigned owner = x >= n_leaves ? 0u : subtree_owner_of_node(x, G);
    if (owner != shard) return;
    if (d == plen) fe_store(out_sib + q, fe_load(nodes + x));
    else fe_store(out_path + q * plen + d, fe_load(nodes + x));
}

// the opened rows this rank owns, gathered leaf-major into a ZEROED device buffer (the rest stays zero: the all-reduce that
// follows is then a gather); canonical = the ark-serialize form
__global__ __launch_bounds__(256) void gather_owned_rows_kernel(const fe* __restrict__ leaves_local, size_t loc, unsigned width, unsigned shard,
                                                                unsigned G, const unsigned long long* __restrict__ idx, size_t k, int canonical,
                                                                fe* __restrict__ out, bool scaled) {
    PK_LATENCY_PRIO();
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k * width) return;
    const size_t q = t / width, j = t % width;
    const size_t i = idx[q];
    if (shard_rank_of_leaf(i, G) != shard) return;
    fe x = fe_load(leaves_local + j * loc + shard_local_row(i, G));
    fe_store(out + t, opened_element(x, scaled, canonical));
}

}  // namespace

namespace pk {
// 1 = this commit is computed whole on every rank; G = it is sharded by leaf index over the context's G ranks.  A pure
// function of (communicator size, rows): every rank and every later opening of the tree take the same decision.
unsigned shard_factor(const pk_ctx* ctx, size_t rows) {
    const unsigned G = (unsigned)comm_world(ctx);
    return (G > 1 && rows >= (size_t)64 * G) ? G : 1;
}
// is the codeword commit_into writes for `rows` leaves held in the hash-ready encoding (see pk_tree::scaled)?  Also a pure
// function of (communicator size, rows).
bool codeword_scaled(const pk_ctx* ctx, size_t rows) { return ntt_scaled_available(ilog2(rows / shard_factor(ctx, rows))); }
// device scratch (in FEs) commit_into needs for a codeword of `rows` x `width`
size_t commit_scratch_fes(const pk_ctx* ctx, size_t rows, size_t width) {
    const unsigned G = shard_factor(ctx, rows);
    if (G == 1) return 2 * width * rows;
    const size_t loc = rows / G;
    return width * (rows + 2 * loc) + loc + rows;  // encode-shard scratch + local digests + gathered digests
}
// RS-encode + Merkle commit into caller-owned device buffers (no allocation): leaves = width*rows/shard_factor FEs
// (column-major; the local shard when sharded), nodes = 2*rows FEs, scratch = commit_scratch_fes FEs.
pk_commit_layout commit_layout(const pk_ctx* ctx, size_t rows) {
    pk_commit_layout l;
    l.n_shards = shard_factor(ctx, rows);
    l.shard = l.n_shards > 1 ? (unsigned)comm_rank(ctx) : 0;
    l.encoding = codeword_scaled(ctx, rows) ? PK_LEAVES_SCALED32 : PK_LEAVES_MONTGOMERY;
    return l;
}
int commit_into(pk_ctx* ctx, const uint64_t* const* d_coeffs, unsigned batch, unsigned n_vars, unsigned log_inv_rate, unsigned fold,
                uint64_t* d_leaves, uint64_t* d_nodes, uint64_t* d_scratch, pk_commit_layout* layout_out) {
    const size_t rows = (size_t)1 << (n_vars + log_inv_rate - fold);
    const size_t width = (size_t)batch << fold;
    const pk_commit_layout lay = commit_layout(ctx, rows);
    if (layout_out) *layout_out = lay;  // what an opening of these buffers must know: kept WITH the commitment by the caller
    const unsigned G = lay.n_shards;
    const bool scaled = lay.encoding == PK_LEAVES_SCALED32;
    if (G == 1) {
        int rc = rs_encode_x(ctx, d_coeffs, batch, n_vars, log_inv_rate, fold, d_leaves, d_scratch, scaled);
        if (!rc) rc = leaf_hash_x(ctx, d_leaves, rows, width, d_nodes + 4 * rows, scaled);
        if (!rc) rc = pk_merkle_inner(ctx, d_nodes, rows);
        return rc;
    }
    // rank g encodes and hashes the rows i = g (mod G) with no communication; ONE all-gather of the 32-byte leaf
    // digests (32*rows bytes in total: 8 MiB at the poseidon size, 256 MiB at 2^26); the inner tree is built on every rank
    const size_t loc = rows / G;
    fe* dig_local = (fe*)d_scratch + width * (rows + 2 * loc);
    fe* gathered = dig_local + loc;
    int rc = rs_encode_shard_x(ctx, d_coeffs, batch, n_vars, log_inv_rate, fold, (unsigned)comm_rank(ctx), G, d_leaves, d_scratch, scaled);
    if (!rc) rc = leaf_hash_x(ctx, d_leaves, loc, width, (uint64_t*)dig_local, scaled);
    if (rc) {
        comm_abort(ctx);  // the other ranks are (or will be) waiting in the all-gather this rank never reaches
        return rc;
    }
    rc = comm_all_gather(ctx, dig_local, gathered, loc * 32);
    if (rc) return rc;
    interleave_digests_kernel<<<(unsigned)((rows + 255) / 256), 256, 0, ctx->stream>>>(gathered, (fe*)d_nodes, rows, G);
    PK_LAUNCH_CHECK(ctx);
    if (!subtree_sharded(G, rows)) return pk_merkle_inner(ctx, d_nodes, rows);  // small tree: hash all of it on every rank
    // SURVEY 8e, first option: this rank hashes only the subtree over its contiguous range of leaves (1/G of the inner nodes),
    // as a compact heap in the -- now free -- gather buffer; the G subtree roots are all-gathered (32 bytes per rank) and the
    // top log2 G levels hashed everywhere.  Inner nodes of the other ranks' subtrees are never formed here: openings collect
    // them from their owners (pk_tree_open).
    const unsigned g = (unsigned)comm_rank(ctx);
    // heap slots [2G, rows) are the inner nodes below the subtree roots: this rank fills only its own 1/G of them -- the rest is
    // zeroed, so that a caller who walks d_nodes himself reads zeros there, not whatever the buffer held (include/provekit_hip.h)
    if (hipMemsetAsync((fe*)d_nodes + 2 * (size_t)G, 0, 32 * (rows - 2 * (size_t)G), ctx->stream) != hipSuccess) {
        comm_abort(ctx);  // the peers are on their way to the all-gather of the subtree roots
        return set_err(ctx, PK_ERR_HIP, "clearing the inner-node slots of the other ranks' subtrees failed");
    }
    fe* H = gathered;  // 2 * loc <= rows entries
    PK_HIP(ctx, hipMemcpyAsync(H + loc, (fe*)d_nodes + rows + (size_t)g * loc, 32 * loc, hipMemcpyDeviceToDevice, ctx->stream));
    rc = pk_merkle_inner(ctx, (uint64_t*)H, loc);
    if (rc) {
        comm_abort(ctx);
        return rc;
    }
    scatter_subtree_kernel<<<(unsigned)((loc + 255) / 256), 256, 0, ctx->stream>>>(H, (fe*)d_nodes, loc, G, g);
    PK_LAUNCH_CHECK(ctx);
    fe* roots = dig_local;  // G <= loc entries, free since the all-gather of the digests
    rc = comm_all_gather(ctx, H + 1, roots, 32);
    if (rc) return rc;
    place_subtree_roots_kernel<<<1, 64, 0, ctx->stream>>>(roots, (fe*)d_nodes, G);
    PK_LAUNCH_CHECK(ctx);
    return merkle_top_x(ctx, d_nodes, G);
}
// open k leaves of a tree described by raw buffers (same outputs as pk_tree_open)
int open_raw(pk_ctx* ctx, const uint64_t* d_leaves, const uint64_t* d_nodes, size_t n_leaves, size_t width, const pk_commit_layout& lay,
             const uint64_t* indices, size_t k, int canonical_leaves, uint64_t* leaves_out, uint64_t* sibling_digests, uint64_t* auth_paths) {
    pk_tree t;
    t.d_leaves = (fe*)d_leaves;
    t.d_nodes = (fe*)d_nodes;
    t.n_leaves = n_leaves;
    t.width = width;
    t.owns_leaves = false;
    t.layout = PK_COL_MAJOR;
    t.n_shards = lay.n_shards;  // the decisions commit_into took for this tree, as recorded at commit time
    t.shard = lay.shard;
    t.scaled = lay.encoding == PK_LEAVES_SCALED32;
    return pk_tree_open(ctx, &t, indices, k, canonical_leaves, leaves_out, sibling_digests, auth_paths);
}
}  // namespace pk

extern "C" {

int pk_tree_destroy(pk_ctx* ctx, pk_tree* t) {
    PK_ENTER(ctx);
    if (!t) return PK_OK;
    (void)wait_ctx(ctx);
    if (t->owns_leaves) (void)hipFree(t->d_leaves);
    (void)hipFree(t->d_nodes);
    delete t;
    return PK_OK;
}

int pk_commit(pk_ctx* ctx, const uint64_t* const* d_coeffs, unsigned batch, unsigned n_vars, unsigned log_inv_rate, unsigned fold,
              uint8_t root_out[32], pk_tree** out) {
    if (!ctx || !out) return PK_ERR_BAD_ARG;
    PK_ENTER(ctx);
    *out = nullptr;
    PK_REQUIRE(ctx, d_coeffs, "null pointer");
    PK_REQUIRE(ctx, batch >= 1 && batch <= 16, "batch out of range");
    PK_REQUIRE(ctx, fold <= n_vars && fold <= 8, "fold out of range");
    PK_REQUIRE(ctx, n_vars + log_inv_rate - fold <= 27, "domain too large (two-adicity 28)");
    const size_t rows = (size_t)1 << (n_vars + log_inv_rate - fold);
    const size_t width = (size_t)batch << fold;
    pk_tree* t = new (std::nothrow) pk_tree();
    if (!t) return PK_ERR_OOM;
    t->n_leaves = rows;
    t->width = width;
    const pk_commit_layout lay = commit_layout(ctx, rows);
    t->n_shards = lay.n_shards;  // > 1: this context is one rank of a device set and keeps only its rows
    t->shard = lay.shard;
    t->scaled = lay.encoding == PK_LEAVES_SCALED32;
    int rc = PK_OK;
    if (hipMalloc((void**)&t->d_leaves, rows / t->n_shards * width * 32) != hipSuccess || hipMalloc((void**)&t->d_nodes, 2 * rows * 32) != hipSuccess) {
        pk_tree_destroy(ctx, t);
        return set_err(ctx, PK_ERR_OOM, "hipMalloc of the codeword matrix failed");
    }
    rc = ensure_ws(ctx, commit_scratch_fes(ctx, rows, width) * 32);
    if (!rc) rc = commit_into(ctx, d_coeffs, batch, n_vars, log_inv_rate, fold, (uint64_t*)t->d_leaves, (uint64_t*)t->d_nodes, (uint64_t*)ctx->d_ws, nullptr);
    if (!rc && root_out) rc = read_root(ctx, (const uint64_t*)t->d_nodes, rows, (uint64_t*)root_out);
    if (rc) {
        pk_tree_destroy(ctx, t);
        return rc;
    }
    *out = t;
    return PK_OK;
}

// the same commit into caller-owned buffers (nothing is allocated: the steady-state form; pk_prove's own path)
int pk_commit_sizes(const pk_ctx* ctx, unsigned batch, unsigned n_vars, unsigned log_inv_rate, unsigned fold, size_t* leaves_fes,
                    size_t* nodes_fes, size_t* scratch_fes) {
    if (!ctx || fold > n_vars || n_vars + log_inv_rate - fold > 27) return PK_ERR_BAD_ARG;
    const size_t rows = (size_t)1 << (n_vars + log_inv_rate - fold), width = (size_t)batch << fold;
    if (leaves_fes) *leaves_fes = rows * width / shard_factor(ctx, rows);
    if (nodes_fes) *nodes_fes = 2 * rows;
    if (scratch_fes) *scratch_fes = commit_scratch_fes(ctx, rows, width);
    return PK_OK;
}
int pk_commit_into(pk_ctx* ctx, const uint64_t* const* d_coeffs, unsigned batch, unsigned n_vars, unsigned log_inv_rate, unsigned fold,
                   uint64_t* d_leaves, uint64_t* d_nodes, uint64_t* d_scratch, uint8_t root_out[32], pk_commit_layout* layout_out) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, d_coeffs && d_leaves && d_nodes && d_scratch, "null pointer");
    PK_REQUIRE(ctx, batch >= 1 && batch <= 16, "batch out of range");
    PK_REQUIRE(ctx, fold <= n_vars && fold <= 8, "fold out of range");
    PK_REQUIRE(ctx, n_vars + log_inv_rate - fold <= 27, "domain too large (two-adicity 28)");
    int rc = commit_into(ctx, d_coeffs, batch, n_vars, log_inv_rate, fold, d_leaves, d_nodes, d_scratch, layout_out);
    if (!rc && root_out) rc = read_root(ctx, d_nodes, (size_t)1 << (n_vars + log_inv_rate - fold), (uint64_t*)root_out);
    return rc;
}
// the opening that goes with pk_commit_into: the same outputs as pk_tree_open, from the raw buffers and the layout recorded
// at commit time (so it stays right if the context's communicator changes or goes away between commit and opening -- except
// that a sharded layout still needs the communicator it was committed under for the exchange of the opened rows)
int pk_commit_open(pk_ctx* ctx, const uint64_t* d_leaves, const uint64_t* d_nodes, size_t n_leaves, size_t width, const pk_commit_layout* layout,
                   const uint64_t* indices, size_t k, int canonical_leaves, uint64_t* leaves_out, uint64_t* sibling_digests, uint64_t* auth_paths) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, d_leaves && d_nodes && layout, "null pointer");
    PK_REQUIRE(ctx, is_pow2(n_leaves) && width >= 1, "n_leaves must be a power of two");
    PK_REQUIRE(ctx, layout->n_shards >= 1 && is_pow2(layout->n_shards) && layout->shard < layout->n_shards &&
                        (layout->encoding == PK_LEAVES_MONTGOMERY || layout->encoding == PK_LEAVES_SCALED32), "bad commit layout");
    PK_REQUIRE(ctx, layout->n_shards == 1 || (layout->n_shards == (unsigned)comm_world(ctx) && layout->shard == (unsigned)comm_rank(ctx)),
               "a sharded commitment must be opened on the device set (and rank) it was committed on");
    return open_raw(ctx, d_leaves, d_nodes, n_leaves, width, *layout, indices, k, canonical_leaves, leaves_out, sibling_digests, auth_paths);
}

// MerkleTree::new over leaves the caller already holds on the device (borrowed, not copied)
int pk_tree_from_leaves(pk_ctx* ctx, const uint64_t* d_leaves, size_t n_leaves, size_t width, int layout, uint8_t root_out[32],
                        pk_tree** out) {
    if (!ctx || !out) return PK_ERR_BAD_ARG;
    PK_ENTER(ctx);
    *out = nullptr;
    PK_REQUIRE(ctx, d_leaves, "null pointer");
    PK_REQUIRE(ctx, is_pow2(n_leaves), "n_leaves must be a power of two");
    pk_tree* t = new (std::nothrow) pk_tree();
    if (!t) return PK_ERR_OOM;
    t->n_leaves = n_leaves;
    t->width = width;
    t->owns_leaves = false;
    t->layout = layout;
    t->d_leaves = (fe*)d_leaves;
    if (hipMalloc((void**)&t->d_nodes, 2 * n_leaves * 32) != hipSuccess) {
        delete t;
        return set_err(ctx, PK_ERR_OOM, "hipMalloc of the tree failed");
    }
    int rc = pk_merkle_commit(ctx, d_leaves, n_leaves, width, layout, (uint64_t*)t->d_nodes);
    if (!rc && root_out) rc = read_root(ctx, (const uint64_t*)t->d_nodes, n_leaves, (uint64_t*)root_out);
    if (rc) {
        pk_tree_destroy(ctx, t);
        return rc;
    }
    *out = t;
    return PK_OK;
}

int pk_tree_info(const pk_tree* t, size_t* n_leaves, size_t* width, const uint64_t** d_leaves, const uint64_t** d_nodes) {
    if (!t) return PK_ERR_BAD_ARG;
    if (n_leaves) *n_leaves = t->n_leaves;
    if (width) *width = t->width;
    if (d_leaves) *d_leaves = (const uint64_t*)t->d_leaves;
    if (d_nodes) *d_nodes = (const uint64_t*)t->d_nodes;
    return PK_OK;
}

int pk_shard_of_leaf(uint64_t leaf, unsigned n_shards, unsigned* rank, uint64_t* local_row) {
    if (!n_shards || !is_pow2(n_shards)) return PK_ERR_BAD_ARG;
    if (rank) *rank = shard_rank_of_leaf((size_t)leaf, n_shards);
    if (local_row) *local_row = shard_local_row((size_t)leaf, n_shards);
    return PK_OK;
}
int pk_shard_interleave_digests(const uint64_t* gathered, size_t rows, unsigned n_shards, uint64_t* nodes) {
    if (!gathered || !nodes || !is_pow2(rows) || !n_shards || !is_pow2(n_shards) || rows % n_shards) return PK_ERR_BAD_ARG;
    for (size_t i = 0; i < rows; i++) memcpy(nodes + 4 * (rows + i), gathered + 4 * shard_gathered_slot(i, rows, n_shards), 32);
    return PK_OK;
}

int pk_tree_layout(const pk_tree* t, pk_commit_layout* layout) {
    if (!t || !layout) return PK_ERR_BAD_ARG;
    layout->n_shards = t->n_shards;
    layout->shard = t->shard;
    layout->encoding = t->scaled ? PK_LEAVES_SCALED32 : PK_LEAVES_MONTGOMERY;
    return PK_OK;
}

int pk_tree_root(pk_ctx* ctx, const pk_tree* t, uint8_t root[32]) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, t && root, "null pointer");
    return pk_memcpy_d2h(ctx, root, t->d_nodes + 1, 32);
}

int pk_tree_open(pk_ctx* ctx, const pk_tree* t, const uint64_t* indices, size_t k, int canonical_leaves, uint64_t* leaves_out,
                 uint64_t* sibling_digests, uint64_t* auth_paths) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, t && (k == 0 || (indices && leaves_out && sibling_digests)), "null pointer");
    if (!k) return PK_OK;
    const unsigned logn = ilog2(t->n_leaves);
    const unsigned plen = logn ? logn - 1 : 0;
    PK_REQUIRE(ctx, plen == 0 || auth_paths, "null pointer");
    for (size_t q = 0; q < k; q++) PK_REQUIRE(ctx, indices[q] < t->n_leaves, "leaf index out of range");
    const size_t idx_bytes = ((k * 8 + 63) / 64) * 64;
    const size_t n1 = k * t->width, n2 = k * (size_t)(plen + 1);
    char* mail = nullptr;
    int rc = mail_alloc(ctx, idx_bytes + 32 * (n1 + k + k * plen), (void**)&mail);
    if (rc) return rc;
    unsigned long long* m_idx = (unsigned long long*)mail;
    fe* m_leaves = (fe*)(mail + idx_bytes);
    fe* m_sib = m_leaves + n1;
    fe* m_path = m_sib + k;
    memcpy(m_idx, indices, k * 8);
    if (t->n_shards > 1) {
        // SURVEY 8e "Openings": leaf i is served by rank i mod G.  Every rank gathers the rows it owns into a zeroed buffer;
        // one all-reduce (k*width*32 bytes, ~100 KiB; each element is non-zero on exactly one rank) hands all of them to
        // everybody; sibling digests and auth paths come from the replicated inner tree.
        // Trees big enough for subtree sharding (shard_map.hpp) hold only their own subtree's inner nodes: the digests are collected
        // the same way, in the same all-reduce.
        PK_REQUIRE(ctx, t->layout == PK_COL_MAJOR, "sharded trees are column-major");
        const bool sub = subtree_sharded(t->n_shards, t->n_leaves);
        const size_t n_red = sub ? n1 + n2 : n1;  // [rows | siblings | paths]: the order of the mailbox block
        rc = ensure_scratch(ctx, ((size_t)1 << 20) + 32 * n_red);
        if (rc) return rc;
        fe* d_rows = (fe*)((char*)ctx->d_scratch + ((size_t)1 << 19));  // clear of the reduction area (head) and the PoW words (tail)
        PK_HIP(ctx, hipMemsetAsync(d_rows, 0, 32 * n_red, ctx->stream));
        gather_owned_rows_kernel<<<(unsigned)((n1 + 255) / 256), 256, 0, ctx->stream>>>(t->d_leaves, t->n_leaves / t->n_shards, (unsigned)t->width, t->shard,
                                                                                     t->n_shards, m_idx, k, canonical_leaves, d_rows, t->scaled);
        PK_LAUNCH_CHECK(ctx);
        if (sub && n2) {
            gather_owned_nodes_kernel<<<(unsigned)((n2 + 255) / 256), 256, 0, ctx->stream>>>(t->d_nodes, t->n_leaves, logn, t->n_shards, t->shard, m_idx, k,
                                                                                          d_rows + n1, d_rows + n1 + k);
            PK_LAUNCH_CHECK(ctx);
        }
        rc = comm_all_reduce_sum_u64(ctx, (uint64_t*)d_rows, 4 * n_red);
        if (rc) return rc;
        PK_HIP(ctx, hipMemcpyAsync(m_leaves, d_rows, 32 * n_red, hipMemcpyDeviceToHost, ctx->stream));
        if (!sub && n2) gather_opening_kernel<<<(unsigned)((n2 + 255) / 256), 256, 0, ctx->stream>>>(nullptr, t->d_nodes, t->n_leaves, 0, t->layout, logn, m_idx, k, 0,
                                                                                                  m_leaves, m_sib, m_path, false);
    } else {
        gather_opening_kernel<<<(unsigned)((n1 + n2 + 255) / 256), 256, 0, ctx->stream>>>(t->d_leaves, t->d_nodes, t->n_leaves, (unsigned)t->width, t->layout,
                                                                                          logn, m_idx, k, canonical_leaves, m_leaves, m_sib, m_path, t->scaled);
    }
    PK_LAUNCH_CHECK(ctx);
    PK_WAIT(ctx);  // not sync_stream: the mailbox is read below
    memcpy(leaves_out, m_leaves, 32 * n1);
    if (logn) memcpy(sibling_digests, m_sib, 32 * k);
    if (plen) memcpy(auth_paths, m_path, 32 * k * plen);
    ctx->mail_off = 0;
    return PK_OK;
}

// The leaf half of an opening on its own: gather k rows of a codeword matrix the caller holds (e.g. one rank's shard of
// a multi-GPU commit, SURVEY 8e "Openings": leaf i is served by GPU i mod G) to leaf-major host memory.
int pk_gather_leaves(pk_ctx* ctx, const uint64_t* d_leaves, size_t n_leaves, size_t width, int layout, const uint64_t* indices, size_t k,
                     int canonical_leaves, uint64_t* leaves_out) {
    return pk_gather_leaves_enc(ctx, d_leaves, n_leaves, width, layout, PK_LEAVES_MONTGOMERY, indices, k, canonical_leaves, leaves_out);
}
int pk_gather_leaves_enc(pk_ctx* ctx, const uint64_t* d_leaves, size_t n_leaves, size_t width, int layout, int encoding, const uint64_t* indices,
                         size_t k, int canonical_leaves, uint64_t* leaves_out) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, encoding == PK_LEAVES_MONTGOMERY || encoding == PK_LEAVES_SCALED32, "unknown leaf encoding");
    PK_REQUIRE(ctx, k == 0 || (d_leaves && indices && leaves_out), "null pointer");
    PK_REQUIRE(ctx, layout == PK_COL_MAJOR || layout == PK_LEAF_MAJOR, "unknown layout");
    if (!k || !width) return PK_OK;
    for (size_t q = 0; q < k; q++) PK_REQUIRE(ctx, indices[q] < n_leaves, "leaf index out of range");
    const size_t idx_bytes = ((k * 8 + 63) / 64) * 64, n1 = k * width;
    char* mail = nullptr;
    int rc = mail_alloc(ctx, idx_bytes + 32 * n1, (void**)&mail);
    if (rc) return rc;
    unsigned long long* m_idx = (unsigned long long*)mail;
    fe* m_leaves = (fe*)(mail + idx_bytes);
    memcpy(m_idx, indices, k * 8);
    gather_opening_kernel<<<(unsigned)((n1 + 255) / 256), 256, 0, ctx->stream>>>((const fe*)d_leaves, nullptr, n_leaves, (unsigned)width, layout, 0, m_idx, k,
                                                                                canonical_leaves, m_leaves, nullptr, nullptr, encoding == PK_LEAVES_SCALED32);
    PK_LAUNCH_CHECK(ctx);
    PK_WAIT(ctx);
    memcpy(leaves_out, m_leaves, 32 * n1);
    ctx->mail_off = 0;
    return PK_OK;
}

// ark MultiPath, uncompressed ark-serialize: Vec<T> = u64 length + items; digests = 32 B canonical LE.
// Fields in order: leaf_siblings_hashes, auth_paths_prefix_lenghts, auth_paths_suffixes, leaf_indexes.
// `indices` must be sorted ascending and unique (whir sorts+dedups STIR queries); paths are root->leaf.
int pk_multipath_serialize(const uint64_t* indices, size_t k, size_t path_len, const uint64_t* sibling_digests,
                           const uint64_t* auth_paths, uint8_t* out, size_t out_cap, size_t* out_len) {
    if (!out_len || (k && (!indices || !sibling_digests || (path_len && !auth_paths)))) return PK_ERR_BAD_ARG;
    std::vector<uint8_t> buf;
    auto put_u64 = [&](uint64_t v) {
        for (int i = 0; i < 8; i++) buf.push_back((uint8_t)(v >> (8 * i)));
    };
    auto put_fe = [&](const uint64_t* p) {
        const uint8_t* b = (const uint8_t*)p;
        buf.insert(buf.end(), b, b + 32);
    };
    put_u64(k);
    for (size_t q = 0; q < k; q++) put_fe(sibling_digests + 4 * q);
    std::vector<size_t> prefix(k, 0);
    for (size_t q = 1; q < k; q++) {
        size_t c = 0;
        while (c < path_len && memcmp(auth_paths + 4 * (q * path_len + c), auth_paths + 4 * ((q - 1) * path_len + c), 32) == 0) c++;
        prefix[q] = c;
    }
    put_u64(k);
    for (size_t q = 0; q < k; q++) put_u64(prefix[q]);
    put_u64(k);
    for (size_t q = 0; q < k; q++) {
        put_u64(path_len - prefix[q]);
        for (size_t c = prefix[q]; c < path_len; c++) put_fe(auth_paths + 4 * (q * path_len + c));
    }
    put_u64(k);
    for (size_t q = 0; q < k; q++) put_u64(indices[q]);
    *out_len = buf.size();
    if (!out || out_cap < buf.size()) return out ? PK_ERR_BAD_ARG : PK_OK;  // out == NULL: size query
    memcpy(out, buf.data(), buf.size());
    return PK_OK;
}

}  // extern "C"
