/*
 * provekit_hip.h -- C ABI of libprovekit_hip.so, the MI355X (gfx950) backend for
 * ProveKit's WHIR prover hot path.
 *
 * This is the drop-in boundary.  The reference has no C ABI: its
 * hot path is Rust calling Rust.  Each entry point below cites the reference
 * interface it replaces (paths relative to worldfnd/provekit @ 2025-08-29); the
 * Rust-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - FE = BN254-Fr element, 32 bytes, 4 x uint64 little-endian limbs, MONTGOMERY
 *     form (x * 2^256 mod p), i.e. the in-memory layout of ark-ff
 *     Fp256<MontBackend<_,4>> (provekit/common/src/lib.rs:19), unless a parameter
 *     says "canonical" (plain little-endian integer < p, what into_bigint() and
 *     the transcript use).
 *   - Pointers named d_* are DEVICE pointers (hipMalloc'ed by pk_malloc or by the
 *     caller, e.g. a torch tensor's data_ptr); all others are host pointers.
 *   - Every call returns PK_OK (0) or a negative PK_ERR_*; pk_last_error(ctx)
 *     returns a message.  Nothing throws or aborts across the boundary (the
 *     reference panics via expect() at prover/src/whir_r1cs.rs:206,434; the Rust
 *     shim maps a status to anyhow::Error instead).
 *   - A pk_ctx is bound to one device and one stream and is single-caller (not
 *     thread-safe); distinct contexts may be used from distinct threads.
 *     Work is enqueued on the context's stream; calls that return values to the
 *     host synchronise that stream, all others are asynchronous.
 *   - Ownership: the caller owns every host pointer for the duration of the call
 *     and every device buffer it allocated; the library owns only what is behind
 *     its opaque handles (pk_ctx, pk_tree, pk_r1cs) until the matching destroy.
 */
#ifndef PROVEKIT_HIP_H
#define PROVEKIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PK_OK 0
#define PK_ERR_BAD_ARG (-1)
#define PK_ERR_OOM (-2)
#define PK_ERR_HIP (-3)
#define PK_ERR_RCCL (-4)
#define PK_ERR_NO_DEVICE (-5)

/* leaf-matrix layouts: element (leaf i, column j) lives at */
#define PK_LEAF_MAJOR 0 /* i*width + j : ark / whir order (a leaf is contiguous)        */
#define PK_COL_MAJOR 1  /* j*n_leaves + i : the HBM-resident order used by pk_commit    */

typedef struct pk_ctx pk_ctx;
typedef struct pk_tree pk_tree;
typedef struct pk_r1cs pk_r1cs;

/* ------------------------------------------------------------------ context */
int pk_abi_version(void);
int pk_device_count(int *n);
int pk_ctx_create(int device, pk_ctx **out);
int pk_ctx_destroy(pk_ctx *ctx);
const char *pk_last_error(const pk_ctx *ctx);
/* run on a caller-owned hipStream_t (e.g. torch's current stream); NULL = the ctx's own stream */
int pk_ctx_set_stream(pk_ctx *ctx, void *hip_stream);
int pk_ctx_sync(pk_ctx *ctx);
/* Skyscraper version used by every hashing entry point: 2 (default; HEAD of the
 * reference, provekit/common/src/skyscraper/whir.rs:23) or 1 (skyscraper/core/src/v1.rs;
 * only to replay the reference's stale proof fixture). */
int pk_ctx_set_hash_version(pk_ctx *ctx, int version);

/* ------------------------------------------------------------------ device memory + timing */
int pk_malloc(pk_ctx *ctx, size_t bytes, void **d_ptr);
int pk_free(pk_ctx *ctx, void *d_ptr);
int pk_memcpy_h2d(pk_ctx *ctx, void *d_dst, const void *src, size_t bytes);
int pk_memcpy_d2h(pk_ctx *ctx, void *dst, const void *d_src, size_t bytes);
int pk_memcpy_d2d(pk_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
int pk_memset_zero(pk_ctx *ctx, void *d_dst, size_t bytes);
/* hipEvent pair on the ctx stream: elapsed ms between start and stop (stop synchronises) */
int pk_timer_start(pk_ctx *ctx);
int pk_timer_stop(pk_ctx *ctx, float *ms);

/* ------------------------------------------------------------------ A1/A2: field arithmetic
 * ark-ff Fp256 (+,-,*) and block_multiplier::scalar_mul
 * (skyscraper/block-multiplier/src/scalar.rs:73-132): elementwise over n FEs. */
int pk_fe_add(pk_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t n);
int pk_fe_sub(pk_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t n);
int pk_fe_mul(pk_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t n);
int pk_fe_to_mont(pk_ctx *ctx, const uint64_t *d_canon, uint64_t *d_out, size_t n);   /* FieldElement::new(BigInt) */
int pk_fe_from_mont(pk_ctx *ctx, const uint64_t *d_mont, uint64_t *d_out, size_t n); /* into_bigint()            */

/* ------------------------------------------------------------------ H1: Skyscraper compress
 * skyscraper::CompressManyFn = fn(&[u8] /64n/, &mut [u8] /32n/) (skyscraper/core/src/lib.rs:26):
 * n two-to-one compressions of canonical little-endian 256-bit integers (any value < 2^256).
 * The _host form has exactly that shape (host slices; returns PK_ERR_BAD_ARG where
 * generic.rs:18-25 panics on a length mismatch); the device form is what the
 * library's own kernels and bench use. */
int pk_compress_many(pk_ctx *ctx, const uint8_t *d_messages, uint8_t *d_hashes, size_t n);
int pk_compress_many_host(pk_ctx *ctx, const uint8_t *messages, size_t messages_len, uint8_t *hashes,
                          size_t hashes_len);

/* ------------------------------------------------------------------ M1/M2: Merkle hashing
 * SkyscraperCRH::evaluate  (provekit/common/src/skyscraper/whir.rs:30-48): leaf digest =
 *   left fold of compress over the leaf's `width` FEs (Montgomery in, converted as
 *   whir.rs:20-25 does); digests are written CANONICAL (what add_digest puts on the
 *   transcript, whir.rs:96-102).
 * SkyscraperTwoToOne / ark MerkleTree::new (whir.rs:53-86): d_nodes is a heap of
 *   2*n_leaves canonical digests: d_nodes[1] = root, children of i are 2i and 2i+1,
 *   leaf digest i sits at d_nodes[n_leaves + i]; d_nodes[0] is zero. */
int pk_leaf_hash(pk_ctx *ctx, const uint64_t *d_leaves, size_t n_leaves, size_t width, int layout,
                 uint64_t *d_digests);
int pk_merkle_inner(pk_ctx *ctx, uint64_t *d_nodes, size_t n_leaves);
int pk_merkle_commit(pk_ctx *ctx, const uint64_t *d_leaves, size_t n_leaves, size_t width, int layout,
                     uint64_t *d_nodes);

/* ------------------------------------------------------------------ N1/N2: Reed-Solomon encode (NTT)
 * Replaces the RS-encode inside whir's CommitmentWriter::commit_batch / round re-commit
 * (call site provekit/prover/src/whir_r1cs.rs:200-206).  For each of `batch` coefficient
 * vectors c_b (2^n_vars FEs, univariate order): rows = 2^(n_vars+log_inv_rate-fold),
 *   leaf_i[b*2^fold + j] = sum_t c_b[2^fold t + j] * w^(i t),  w = generator of the order-rows subgroup
 * (recursive-verifier/app/circuit/whir_utilities.go:180-186, whir.go:99,141).
 * d_coeffs: HOST array of `batch` DEVICE pointers.  d_leaves: COLUMN-major matrix
 * [batch*2^fold][rows].  d_scratch: 2 * batch * 2^fold * rows FEs of device scratch. */
int pk_rs_encode(pk_ctx *ctx, const uint64_t *const *d_coeffs, unsigned batch, unsigned n_vars,
                 unsigned log_inv_rate, unsigned fold, uint64_t *d_leaves, uint64_t *d_scratch);
/* plain NTT of `ncols` contiguous vectors of 2^log_n FEs, natural order in and out:
 * out[c][k] = sum_i in[c][i] * w_N^(i k)   (helper of pk_rs_encode; exposed for tests/bench) */
int pk_ntt(pk_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, unsigned log_n, unsigned ncols);

#ifdef __cplusplus
}
#endif
#endif /* PROVEKIT_HIP_H */
