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
 *     thread-safe); distinct contexts may be used from distinct threads (every
 *     entry point selects the context's device on the calling thread).
 *     Work is enqueued on the context's stream; calls that return values to the
 *     host synchronise that stream, all others are asynchronous.  Small results
 *     (sums, roots, nonces, openings) reach the host through pinned memory written
 *     by the producing kernel, not through copy operations.
 *   - Above this C ABI: include/provekit_hip.hpp (C++17, the reference's names) and
 *     provekit_amd/ (Python/ctypes).
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
#define PK_ERR_UNSATISFIED (-6) /* pk_r1cs_test_witness_satisfaction: a constraint failed */
#define PK_ERR_IO_PATTERN (-7) /* an IO pattern does not declare the operations the prover performs (spongefish: InvalidIOPattern) */

/* leaf-matrix layouts: element (leaf i, column j) lives at */
#define PK_LEAF_MAJOR 0 /* i*width + j : ark / whir order (a leaf is contiguous)        */
#define PK_COL_MAJOR 1  /* j*n_leaves + i : the HBM-resident order used by pk_commit    */

typedef struct pk_ctx pk_ctx;
typedef struct pk_tree pk_tree;
typedef struct pk_r1cs pk_r1cs;
typedef struct pk_scheme pk_scheme;

/* ------------------------------------------------------------------ context */
int pk_abi_version(void);
int pk_device_count(int *n);
/* How host threads wait for `device` (every blocking call of this library ends in a stream synchronisation; a proof makes ~65 of
 * them).  PK_WAIT_SPIN: the waiting thread polls, lowest latency, one busy core per waiting thread: right for one proof at a time.
 * (HIP's own default is hipDeviceScheduleAuto -- spin, then yield; on a device whose flag this library has never changed
 * PK_WAIT_SPIN leaves it at that and only ends PK_WAIT_POLL, which is safe at any time.)  PK_WAIT_BLOCK: the thread sleeps until the
 * completion interrupt -- right for many provers per GPU: with 16 provers in flight spinning burns 16 cores for nothing, and on a
 * host that grants fewer (a container CPU quota) the throttling stalls every prover (measured: 24 provers under a 16-CPU quota, 186
 * proofs/s spinning, 257 blocking; docs/HISTORY_r01-r05.md 5).  Process-wide for the device (hipSetDeviceFlags).  Choose PK_WAIT_BLOCK BEFORE
 * creating contexts on the device and leave it: the runtime builds its completion signals for the mode in force, and a wait that
 * blocks on a signal made for polling never wakes (measured: switching to blocking while provers were running hung one of them in
 * its next synchronisation).  The environment variable PK_HOST_WAIT = spin | block | poll makes pk_ctx_create apply the mode before it
 * creates the context's stream (any other value: PK_ERR_BAD_ARG).  While it sleeps in PK_WAIT_POLL the calling thread's timer slack
 * is lowered to 2 us and restored when the wait returns. */
#define PK_WAIT_SPIN 0
#define PK_WAIT_BLOCK 1
#define PK_WAIT_POLL 2 /* the library's own wait: hipStreamQuery with sleeps of 20 .. 100 us in between -- the cheapest for the host (no core per
                        * waiting thread, none of the runtime's spinning before it blocks), completion noticed up to one interval late.  Unlike the
                        * two runtime modes it may be chosen and left at any time. */
int pk_device_set_host_wait(int device, int mode);
int pk_ctx_create(int device, pk_ctx **out);
int pk_ctx_destroy(pk_ctx *ctx);
const char *pk_last_error(const pk_ctx *ctx);
/* run on a caller-owned hipStream_t (e.g. torch's current stream); NULL = the ctx's own stream */
int pk_ctx_set_stream(pk_ctx *ctx, void *hip_stream);
/* Latency mode (default off).  For ONE proof at a time on an otherwise idle GPU (BASELINE configs[3]'s latency case): pk_prove
 * enqueues each sumcheck round one ahead -- the next round's kernel, or the closing fold, is already in the queue, gated on a word of the
 * pinned host page, while the host absorbs the current round's three evaluations and squeezes the challenge -- so those Fiat-Shamir round
 * trips cost the host link's latency instead of a kernel launch plus a stream synchronisation (4 us against 15, profiles/r04_roundtrip.json);
 * and the blinding commitment (which depends on nothing but the proof's key) and the statement's external rows and sums (on alpha only) run
 * on a second stream of the same GPU underneath the witness commitment and the blinding WHIR proof.  9.5 -> 8.6 ms per proof at the poseidon size; the transcript is byte-identical either way.  Leave it off when several provers share the GPU: a gated kernel holds its workgroup
 * slots while it waits for the host. */
int pk_ctx_set_latency_mode(pk_ctx *ctx, int on);
int pk_ctx_sync(pk_ctx *ctx);
/* Skyscraper version used by every hashing entry point: 2 (default; HEAD of the
 * reference, provekit/common/src/skyscraper/whir.rs:23) or 1 (skyscraper/core/src/v1.rs;
 * only to replay the reference's stale proof fixture). */
int pk_ctx_set_hash_version(pk_ctx *ctx, int version);

/* ------------------------------------------------------------------ device sets: one commit sharded over the GPUs of a node
 * SURVEY 8b/8e.  The reference is one process on one host, so there is no interface to mirror; the path needs one collective
 * per commit (all-gather of the 32-byte leaf digests) and one per opening (the opened rows, each owned by one rank).  A context
 * that carries a communicator makes pk_commit / pk_tree_open / pk_prove shard every large commit by leaf index: rank g of G
 * encodes and hashes the codeword rows i = g (mod G), the digests are all-gathered (RCCL over xGMI) and the inner tree is
 * built on every rank; every rank holds the full polynomials and produces the same transcript.
 *   pk_ctx_create_set   one process, n GPUs: n contexts joined by ncclCommInitAll (drive them from n host threads: a
 *                       collective blocks until every rank has entered it).  A device listed more than once gets the
 *                       in-process transport instead of RCCL (which refuses two ranks per device).
 *   pk_comm_unique_id / pk_comm_init_rank   one process per GPU (torch.distributed.run, MPI, ...): rank 0 makes the id,
 *                       the launcher broadcasts its 128 bytes, every rank joins.
 *   pk_comm_init_local  the in-process transport on existing contexts (any devices, also all on one device): copies between
 *                       the ranks' buffers and a host barrier; what a single-GPU box can run.
 *   pk_comm_init_host   bring-your-own transport: the caller supplies the all-gather over HOST buffers (MPI, gloo, a test
 *                       harness); the library stages the send block through pinned memory, calls it, and uploads the result.
 *                       fn(user, send, recv, bytes_per_rank) must fill recv[r*bytes .. (r+1)*bytes) with rank r's send block on
 *                       every rank and return 0; it is called on the thread that entered the library, in the same order on all
 *                       ranks.  Slower than RCCL (two PCIe hops) but works wherever a host collective does -- several processes
 *                       on one GPU included, which is how the multi-process launch is tested on a single-GPU box.
 * PK_ERR_RCCL: librccl could not be loaded (it is resolved with dlopen at first use), one of its calls failed, a host
 * transport's callback returned non-zero, or another rank of the set failed (the communicator is then unusable).
 * A rank whose sharded call fails before it reaches a collective aborts its communicator so that its peers do not wait for it
 * for ever.  In-process group: the waiting ranks wake at once with PK_ERR_RCCL.  RCCL: ncclCommAbort is local -- the failing rank
 * tears down its OWN communicator, nothing reaches its peers; every collective therefore has a deadline (environment
 * PK_COMM_TIMEOUT_S, default 120 s, measured on an event recorded right behind the collective -- work queued after it does not count;
 * ncclCommGetAsyncError is polled meanwhile): the first wait that finds its collective overdue aborts this rank's own communicator
 * too (which ends its stuck kernel) and returns PK_ERR_RCCL.  Host transport: the peers are the caller's to time out.
 * After any of these: pk_comm_destroy on every rank and join again.  (A refusal every rank makes identically before the call has
 * enqueued a collective -- PK_ERR_BAD_ARG, PK_ERR_UNSATISFIED, PK_ERR_IO_PATTERN -- leaves an RCCL communicator usable.) */
#define PK_MAX_RANKS 16
#define PK_COMM_ID_BYTES 128
#define PK_COMM_NONE 0
#define PK_COMM_LOCAL 1
#define PK_COMM_RCCL 2
#define PK_COMM_HOST 3
typedef int (*pk_host_all_gather_fn)(void *user, const void *send, void *recv, size_t bytes_per_rank);
int pk_comm_init_host(pk_ctx *ctx, int world, int rank, pk_host_all_gather_fn fn, void *user);
/* The leaf-index shard map, host only (the same functions the device code is compiled from, csrc/shard_map.hpp): leaf i of a
 * commit sharded over n_shards ranks lives on rank i mod n_shards at local row i / n_shards; pk_shard_interleave_digests
 * places an all-gather's output (rank r's local digests as block r) into the leaf layer of the node heap, nodes[rows + i]. */
int pk_shard_of_leaf(uint64_t leaf, unsigned n_shards, unsigned *rank, uint64_t *local_row);
int pk_shard_interleave_digests(const uint64_t *gathered, size_t rows, unsigned n_shards, uint64_t *nodes /* 2*rows FEs */);
int pk_ctx_create_set(const int *devices, int n, pk_ctx **out /* n entries */);
int pk_comm_unique_id(uint8_t id[PK_COMM_ID_BYTES]);
int pk_comm_init_rank(pk_ctx *ctx, const uint8_t id[PK_COMM_ID_BYTES], int world, int rank);
int pk_comm_init_local(pk_ctx *const *ctxs, int n);
int pk_comm_info(const pk_ctx *ctx, int *rank, int *world, int *kind);
/* which RCCL the library resolved (dlopen at first use; the environment variable PK_RCCL_LIB names one outright, otherwise
 * the copy already in the process, then librccl.so.1): ncclGetVersion's code (e.g. 22703; 0 if the library lacks the call)
 * and the name it was opened by.  PK_ERR_RCCL: none could be loaded. */
int pk_comm_rccl_version(int *version, char *path, size_t cap);
int pk_comm_destroy(pk_ctx *ctx);
/* After a sharded call failed on some rank the communicator is poisoned on purpose (nobody may wait for a rank that left).
 * pk_comm_reset makes it usable again once EVERY rank is back from the failed call: the in-process and host transports clear
 * their sticky flag (call it on every rank; harmless on a healthy communicator); an RCCL communicator that was aborted is gone --
 * PK_ERR_RCCL: pk_comm_destroy + pk_comm_init_rank again. */
int pk_comm_reset(pk_ctx *ctx);
/* the two collectives, on the context's stream (exposed for tests and for callers that shard their own steps):
 * d_recv[r*bytes_per_rank ...] = rank r's d_send;  d_buf[i] = sum over ranks of d_buf[i] (wrapping u64) */
int pk_comm_all_gather(pk_ctx *ctx, const void *d_send, void *d_recv, size_t bytes_per_rank);
int pk_comm_all_reduce_sum_u64(pk_ctx *ctx, uint64_t *d_buf, size_t count);

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
/* Per-kernel timing (the CLI's SpanStats layer of the reference, tooling/cli/src/span_stats.rs, scaled to the
 * device side): when enabled every major kernel launch is bracketed by a hipEvent pair on the work stream.
 * pk_profile_read sums launches / milliseconds for one kernel name ("leaf_hash", "ntt_pass", "ntt_pass_last",
 * "merkle_inner", "sumcheck_cubic", ...); pk_profile_names lists the names seen (comma-separated). */
int pk_profile_enable(pk_ctx *ctx, int on);
int pk_profile_reset(pk_ctx *ctx);
int pk_profile_read(pk_ctx *ctx, const char *name, uint64_t *launches, double *total_ms);
int pk_profile_names(pk_ctx *ctx, char *buf, size_t cap);

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
/* One shard of the same encode for a commit spread over n_shards GPUs: shard g keeps the codeword rows
 * (leaves) i = g + n_shards*t, t < rows/n_shards.  Every rank holds the full coefficient vectors; the shard is a
 * log2(n_shards)-stage decimation-in-frequency pre-step restricted to residue g followed by NTTs of size rows/n_shards,
 * so ranks run independently until the digest all-gather.  d_leaves_local: column-major [batch*2^fold][rows/n_shards];
 * d_scratch: batch*2^fold*(rows + 2*rows/n_shards) FEs. */
int pk_rs_encode_shard(pk_ctx *ctx, const uint64_t *const *d_coeffs, unsigned batch, unsigned n_vars, unsigned log_inv_rate,
                       unsigned fold, unsigned shard, unsigned n_shards, uint64_t *d_leaves_local, uint64_t *d_scratch);
/* plain NTT of `ncols` contiguous vectors of 2^log_n FEs, natural order in and out:
 * out[c][k] = sum_i in[c][i] * w_N^(i k)   (helper of pk_rs_encode; exposed for tests/bench) */
int pk_ntt(pk_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, unsigned log_n, unsigned ncols);

/* ------------------------------------------------------------------ T1: multilinear evals <-> coefficients
 * EvaluationsList::to_coeffs (whir; call sites provekit/prover/src/whir_r1cs.rs:195,198) and its
 * inverse, in place over 2^n_vars FEs: for every index bit h, v[i|h] -= v[i] (resp. +=). */
int pk_to_coeffs(pk_ctx *ctx, uint64_t *d_evals, unsigned n_vars);
int pk_to_evals(pk_ctx *ctx, uint64_t *d_coeffs, unsigned n_vars);
/* out-of-place forms (d_src untouched; saves the copy when both forms are needed, whir_r1cs.rs:193-198) */
int pk_to_coeffs_into(pk_ctx *ctx, const uint64_t *d_src, uint64_t *d_dst, unsigned n_vars);
int pk_to_evals_into(pk_ctx *ctx, const uint64_t *d_src, uint64_t *d_dst, unsigned n_vars);

/* ------------------------------------------------------------------ S2 / W2: equality-polynomial tables
 * calculate_evaluations_over_boolean_hypercube_for_eq / eval_eq
 * (provekit/common/src/utils/sumcheck.rs:146-171): d_out[i] = prod_j (bit_j(i) ? r_j : 1-r_j), variable 0
 * <-> most significant index bit; r = m host FEs; d_out = 2^m device FEs. */
int pk_eq_table(pk_ctx *ctx, const uint64_t *r, unsigned m, uint64_t *d_out);
/* d_w[i] (+)= sum_{t<q} scales[t] * eq(points[t], i): whir's add-equality-weights for the OOD and STIR
 * points of a round.  points = q*n_vars host FEs, scales = q host FEs;
 * overwrite != 0 starts from zero instead of accumulating. */
int pk_eq_accumulate(pk_ctx *ctx, uint64_t *d_w, unsigned n_vars, const uint64_t *points, const uint64_t *scales,
                     unsigned q, int overwrite);

/* ------------------------------------------------------------------ S3: Spartan cubic sumcheck round
 * sumcheck_fold_map_reduce::<4,3> (provekit/common/src/utils/sumcheck.rs:16-104) with the cubic map of
 * provekit/prover/src/whir_r1cs.rs:284-291.  len = current length of each of the four arrays.
 * fold_or_null == NULL: pairs (i, i+len/2).  Otherwise the leading variable is first folded in place
 * (p0 += fold*(p2-p0), p1 += fold*(p3-p1) on quarters, sumcheck.rs:92-97), pairs are (i, i+len/4), and
 * the caller continues with len/2 (the reference truncates, whir_r1cs.rs:292-297).
 * out = [f(0), f(-1), f_inf] as 3 host FEs.  Blocking (synchronises the stream). */
int pk_sumcheck_cubic_round(pk_ctx *ctx, uint64_t *d_a, uint64_t *d_b, uint64_t *d_c, uint64_t *d_eq, size_t len,
                            const uint64_t *fold_or_null, uint64_t out[12]);

/* ------------------------------------------------------------------ W3: WHIR quadratic sumcheck sub-round
 * h(X) = sum_i f(i,X) w(i,X) over adjacent pairs (2i, 2i+1); out = [h(0), h(1), h(2)]
 * (recursive-verifier/app/circuit/whir_utilities.go:102-125; utilities.go:148-154).
 * With fold != NULL, f and w (length len) are first folded by it, v'[i] = v[2i] + fold*(v[2i+1]-v[2i]),
 * into d_f_out / d_w_out (length len/2, must not alias the inputs) and the sums run over the folded
 * arrays.  Blocking. */
int pk_sumcheck_quadratic_round(pk_ctx *ctx, const uint64_t *d_f, const uint64_t *d_w, size_t len,
                                const uint64_t *fold_or_null, uint64_t *d_f_out, uint64_t *d_w_out, uint64_t out[12]);
/* v'[i] = v[2i] + r*(v[2i+1]-v[2i]); out-of-place, length len -> len/2 */
int pk_fold_pairs(pk_ctx *ctx, const uint64_t *d_v, size_t len, const uint64_t *r, uint64_t *d_out);

/* ------------------------------------------------------------------ S5 / E1 / W1 / batching
 * pk_dot: Weights::linear(w).weighted_sum(f) = sum w[i] f[i] (provekit/prover/src/whir_r1cs.rs:401-405).
 * pk_eval_univariate: sum c[i] z^i = the multilinear coefficient form evaluated at
 *   (z^(2^(n-1)), ..., z^2, z), i.e. an OOD answer (utilities.go:182-190).
 * pk_fold_coeffs: out[t] = sum_j c[2^k t + j] prod_b r_b^bit_b(j), r[0] <-> bit 0
 *   (MultivarPoly, utilities.go:15-22; whir_utilities.go:180-186); 2^n_vars -> 2^(n_vars-k) FEs.
 * pk_fe_axpy: y += beta*x (batching f + beta*g, mtUtilities.go:98-114). */
int pk_dot(pk_ctx *ctx, const uint64_t *d_w, const uint64_t *d_f, size_t n, uint64_t out[4]);
/* <w,f> and <w,g> in one pass over w: out = [ <w,f>, <w,g> ] */
int pk_dot2(pk_ctx *ctx, const uint64_t *d_w, const uint64_t *d_f, const uint64_t *d_g, size_t n, uint64_t out[8]);
int pk_eval_univariate(pk_ctx *ctx, const uint64_t *d_coeffs, size_t n, const uint64_t z[4], uint64_t out[4]);
int pk_fold_coeffs(pk_ctx *ctx, const uint64_t *d_coeffs, unsigned n_vars, const uint64_t *r, unsigned k,
                   uint64_t *d_out);
int pk_fe_axpy(pk_ctx *ctx, uint64_t *d_y, const uint64_t *beta, const uint64_t *d_x, size_t n);

/* ------------------------------------------------------------------ S1 / S4: R1CS sparse products
 * pk_sparse_matrix mirrors provekit_common::SparseMatrix (provekit/common/src/sparse_matrix.rs:12-27):
 * new_row_indices[num_rows] = offset of each row's first entry, col_indices[nnz], values[nnz] =
 * indices into the Interner's table of distinct field elements (interner.rs).  mats = {A, B, C}.
 * A pk_r1cs is immutable once created: every context of its device may use it (one upload serves all the prover threads of a
 * GPU); destroy it after the schemes that were bound to it.  Rows and columns longer than 64 entries -- the constant-one
 * witness' column, a grand sum's row -- are summed by workgroups instead of one lane; nothing for the caller to do. */
typedef struct pk_sparse_matrix {
    const uint32_t *new_row_indices;
    const uint32_t *col_indices;
    const uint32_t *values;
    size_t nnz;
} pk_sparse_matrix;
int pk_r1cs_create(pk_ctx *ctx, size_t num_constraints, size_t num_witnesses, const pk_sparse_matrix mats[3],
                   const uint64_t *interner, size_t n_interned, pk_r1cs **out);
/* The same upload from the postcard bytes of the reference's `R1CS` (provekit/common/src/r1cs.rs:8-14 with the serde impls of
 * sparse_matrix.rs:12-27 and interner.rs:6-13; postcard is the encoding of the reference's own .nps files,
 * file/bin.rs:22-71).  This is how a Rust caller passes an R1CS: SparseMatrix keeps its three arrays private, but
 * `postcard::to_allocvec(&scheme.r1cs)` is available to any crate.  Interned values arrive canonical (ark-serialize) and
 * are converted to Montgomery form here.  *consumed = bytes read (an R1CS embedded in a longer stream). */
int pk_r1cs_from_postcard(pk_ctx *ctx, const uint8_t *bytes, size_t len, pk_r1cs **out, size_t *num_constraints,
                          size_t *num_witnesses, size_t *num_public_inputs, size_t *consumed);
int pk_r1cs_destroy(pk_ctx *ctx, pk_r1cs *r1cs);
/* calculate_witness_bounds (sumcheck.rs:181-193): a = A z, b = B z, c = a o b, each zero-padded to 2^m0 */
int pk_r1cs_witness_bounds(pk_ctx *ctx, const pk_r1cs *r1cs, const uint64_t *d_z, unsigned m0, uint64_t *d_a,
                           uint64_t *d_b, uint64_t *d_c);
/* HydratedSparseMatrix * v (transpose == 0, sparse_matrix.rs:150-165) or v * HydratedSparseMatrix
 * (transpose != 0, sparse_matrix.rs:169-184); matrix = 0 (A), 1 (B), 2 (C) */
int pk_r1cs_matvec(pk_ctx *ctx, const pk_r1cs *r1cs, int matrix, int transpose, const uint64_t *d_x, uint64_t *d_y);
/* calculate_external_row_of_r1cs_matrices (sumcheck.rs:207-218): d_out = [eq^T A | eq^T B | eq^T C],
 * 3 * num_witnesses FEs; d_eq_alpha holds at least num_constraints FEs */
int pk_r1cs_external_row(pk_ctx *ctx, const pk_r1cs *r1cs, const uint64_t *d_eq_alpha, uint64_t *d_out);
/* R1CSSolver::test_witness_satisfaction (provekit/prover/src/r1cs.rs:41-60): PK_OK when (A z) o (B z) == C z;
 * otherwise PK_ERR_UNSATISFIED with *first_failed_row = the lowest failing row ("Constraint {row} failed") -- -1 on
 * success.  n_witness != num_witnesses is PK_ERR_BAD_ARG ("Witness size does not match").  Synchronises the stream. */
int pk_r1cs_test_witness_satisfaction(pk_ctx *ctx, const pk_r1cs *r1cs, const uint64_t *d_witness, size_t n_witness,
                                      int64_t *first_failed_row);

/* ------------------------------------------------------------------ P1: proof of work
 * spongefish_pow::PowStrategy for Skyscraper (provekit/common/src/skyscraper/pow.rs:14-30):
 * pk_pow_solve = solve() -> skyscraper::pow::solve (skyscraper/core/src/pow.rs:33-41; adds the 0.01
 * prover bias); returns the SMALLEST nonce with compress(challenge, [nonce,0,0,0]) < threshold (the
 * reference returns whichever valid nonce its threads find first, generic.rs:42-71).
 * pk_pow_check = check() -> pow::verify (pow.rs:24-26).  pk_pow_threshold = pow.rs:14-22 (host only). */
int pk_pow_threshold(double difficulty, uint64_t out[4]);
int pk_pow_solve(pk_ctx *ctx, const uint8_t challenge[32], double bits, uint64_t *nonce);
int pk_pow_check(pk_ctx *ctx, const uint8_t challenge[32], double bits, uint64_t nonce, int *ok);

/* The leaf half of an opening by itself: k rows of a codeword matrix the caller holds on the device (one rank's shard of
 * a multi-GPU commit -- SURVEY 8e: leaf i is served by GPU i mod G) gathered to leaf-major host memory, k*width FEs. */
int pk_gather_leaves_enc(pk_ctx *ctx, const uint64_t *d_leaves, size_t n_leaves, size_t width, int layout, int encoding,
                         const uint64_t *indices, size_t k, int canonical_leaves, uint64_t *leaves_out);
/* the same for Montgomery leaves (encoding = PK_LEAVES_MONTGOMERY) */
int pk_gather_leaves(pk_ctx *ctx, const uint64_t *d_leaves, size_t n_leaves, size_t width, int layout,
                     const uint64_t *indices, size_t k, int canonical_leaves, uint64_t *leaves_out);
/* ------------------------------------------------------------------ commitment handle + openings (N1+N2+M1+M2, Q1)
 * pk_commit: the data-parallel body of whir's CommitmentWriter::commit_batch
 * (provekit/prover/src/whir_r1cs.rs:200-206): RS-encode `batch` coefficient vectors, hash the leaves
 * (width = batch*2^fold), build the tree; root_out = canonical 32-byte root (what add_digest sends,
 * provekit/common/src/skyscraper/whir.rs:96-102).  The codeword matrix and all tree levels stay in
 * HBM behind *out until pk_tree_destroy.
 * pk_tree_from_leaves: MerkleTree::new over leaves already on the device (borrowed, not copied).
 * pk_tree_open: MerkleTree::generate_multi_proof + leaf gather for k leaf indices: leaves_out =
 * k*width FEs leaf-major (Montgomery, or canonical as ark-serialize writes them if canonical_leaves),
 * sibling_digests = k canonical digests, auth_paths = k*(log2(n_leaves)-1) canonical digests in
 * root->leaf order.  pk_multipath_serialize (host only): ark MultiPath wire form, prefix-compressed
 * (types.go:17-22; utilities.go:71-82); out == NULL queries the length. */
int pk_commit(pk_ctx *ctx, const uint64_t *const *d_coeffs, unsigned batch, unsigned n_vars, unsigned log_inv_rate,
              unsigned fold, uint8_t root_out[32], pk_tree **out);
/* pk_commit into caller-owned device buffers (no allocation; sizes in FEs from pk_commit_sizes, which account for the
 * context's device set: a rank of G keeps 1/G of the codeword rows).
 * WHICH SLOTS OF d_nodes ARE VALID.  d_nodes has room for the full heap (2 * rows digests, root at [1], leaf digests at
 * [rows, 2 rows)) on every rank.  Outside a device set, and in a device set with fewer than 2^13 rows per rank, every rank
 * holds the whole heap.  From 2^13 rows per rank up the inner tree is sharded by contiguous subtree: rank g of G holds the
 * leaf layer [rows, 2 rows) and the top levels [1, 2G) like everyone else, and of every level c in [2G, rows) only its own
 * slots [c + g c/G, c + (g+1) c/G) -- the other ranks' inner nodes are ZERO here, never computed.  (pk_tree_info's d_nodes
 * is the same buffer under the same rule.)  Do not walk d_nodes yourself in a device set: pk_commit_open / pk_tree_open
 * collect authentication paths from the ranks that own them; pk_shard_* below state the map.
 * ENCODING OF d_leaves.  The codeword a commit leaves behind is the library's own working form, described by the
 * pk_commit_layout it reports (layout_out, or pk_tree_layout for a pk_tree):
 *   n_shards / shard   rows i = shard (mod n_shards) are present, local row t = i / n_shards (1 / 0 outside a device set)
 *   encoding           PK_LEAVES_MONTGOMERY: elements are Montgomery images (what pk_rs_encode / pk_leaf_hash take and give);
 *                      PK_LEAVES_SCALED32: elements are the plain integers 32*v mod p, lazily reduced (< 1.6 p) -- the form the
 *                      Skyscraper kernels compute in (csrc/skyscraper29s.hpp), emitted by the commit's NTT from 2^11 local rows
 *                      up, so the leaf hash converts nothing.
 * Which encoding a commit uses is a function of (device set, rows) only.  Raw buffers in PK_LEAVES_SCALED32 must NOT be handed
 * to pk_tree_from_leaves / pk_leaf_hash / pk_gather_leaves (they assume Montgomery): open them with pk_commit_open, or gather
 * rows with pk_gather_leaves_enc; both convert the opened rows back to canonical / Montgomery. */
#define PK_LEAVES_MONTGOMERY 0
#define PK_LEAVES_SCALED32 1
typedef struct pk_commit_layout {
    unsigned n_shards, shard;
    int encoding;
} pk_commit_layout;
int pk_commit_sizes(const pk_ctx *ctx, unsigned batch, unsigned n_vars, unsigned log_inv_rate, unsigned fold,
                    size_t *leaves_fes, size_t *nodes_fes, size_t *scratch_fes);
int pk_commit_into(pk_ctx *ctx, const uint64_t *const *d_coeffs, unsigned batch, unsigned n_vars, unsigned log_inv_rate,
                   unsigned fold, uint64_t *d_leaves, uint64_t *d_nodes, uint64_t *d_scratch, uint8_t root_out[32],
                   pk_commit_layout *layout_out);
/* pk_tree_open for buffers written by pk_commit_into, under the layout it reported */
int pk_commit_open(pk_ctx *ctx, const uint64_t *d_leaves, const uint64_t *d_nodes, size_t n_leaves, size_t width,
                   const pk_commit_layout *layout, const uint64_t *indices, size_t k, int canonical_leaves,
                   uint64_t *leaves_out, uint64_t *sibling_digests, uint64_t *auth_paths);
int pk_tree_layout(const pk_tree *tree, pk_commit_layout *layout);
int pk_tree_from_leaves(pk_ctx *ctx, const uint64_t *d_leaves, size_t n_leaves, size_t width, int layout,
                        uint8_t root_out[32], pk_tree **out);
int pk_tree_info(const pk_tree *tree, size_t *n_leaves, size_t *width, const uint64_t **d_leaves,
                 const uint64_t **d_nodes);
int pk_tree_root(pk_ctx *ctx, const pk_tree *tree, uint8_t root[32]);
int pk_tree_open(pk_ctx *ctx, const pk_tree *tree, const uint64_t *indices, size_t k, int canonical_leaves,
                 uint64_t *leaves_out, uint64_t *sibling_digests, uint64_t *auth_paths);
int pk_tree_destroy(pk_ctx *ctx, pk_tree *tree);
int pk_multipath_serialize(const uint64_t *indices, size_t k, size_t path_len, const uint64_t *sibling_digests,
                           const uint64_t *auth_paths, uint8_t *out, size_t out_cap, size_t *out_len);

/* ------------------------------------------------------------------ the seam: WhirR1CSProver::prove
 * pk_whir_config carries the WhirConfig fields the prover consumes (the ones
 * tooling/provekit-gnark/src/gnark_config.rs:32-57 exports); WhirConfig::new itself lives in the external
 * `whir` crate, so the caller (Rust side) fills these from its WhirConfig.
 * pk_scheme = WhirR1CSScheme {m, m_0, whir_witness, whir_for_hiding_spartan} (provekit/common/src/whir_r1cs.rs:17-24)
 * bound to an uploaded R1CS; it owns a device arena sized for one proof, so pk_prove allocates nothing.
 * pk_prove = WhirR1CSProver::prove(&self, &R1CS, Vec<FieldElement>) -> WhirR1CSProof{transcript}
 * (provekit/prover/src/whir_r1cs.rs:36-100): d_witness = n_witness Montgomery FEs on the device; the proof string is
 * written to transcript_out (capacity cap; *len receives its length).
 * Randomness: the reference takes the ZK mask, the random polynomial and the blinding univariates from thread_rng
 * (provekit/common/src/utils/zk_utils.rs:13-22, provekit/prover/src/whir_r1cs.rs:197,212-221).  With
 * rng_seed32 == NULL (production) pk_prove draws a fresh 256-bit key from the OS CSPRNG (getrandom) for every proof
 * and expands it on the device with ChaCha12 (the cipher of rand's ThreadRng) + rejection sampling; a non-NULL rng_seed32 (32 bytes) injects the key
 * instead -- a TEST HOOK for reproducible transcripts, never to be used with a fixed value in deployment. */
#define PK_MAX_WHIR_ROUNDS 16
typedef struct pk_whir_config {
    unsigned n_vars;
    unsigned batch_size;
    unsigned folding_factor;
    unsigned starting_log_inv_rate;
    unsigned n_rounds;
    unsigned num_queries[PK_MAX_WHIR_ROUNDS];
    unsigned ood_samples[PK_MAX_WHIR_ROUNDS];
    double pow_bits[PK_MAX_WHIR_ROUNDS];
    unsigned final_queries;
    double final_pow_bits;
    unsigned commitment_ood_samples;
    double final_folding_pow_bits; /* PoW after the final sumcheck when > 0 (recursive-verifier/app/circuit/whir.go:196-201;
                                      exported at tooling/provekit-gnark/src/gnark_config.rs:52,91) */
} pk_whir_config;
/* WhirConfig::new as provekit calls it (provekit/r1cs-compiler/src/whir_r1cs.rs:38-53): ConjectureList soundness,
 * security_level bits, the given folding factor / starting rate / batch size, and pow_bits < 0 meaning
 * default_max_pow(n_vars, starting_log_inv_rate).  WhirConfig::new itself lives in the external `whir` crate (rev 3e7f8c2,
 * not in the reference tree): this is its published derivation restated -- round count (n/k - 1 rounds, n mod k final
 * variables), queries = ceil((security - pow)/log_inv_rate) against the OLD rate, 1 OOD sample while
 * 2*list_size + n < field_bits, pow_bits[r] = max(0, security - min(queries*log_inv_rate, combination error)) -- and it is
 * pinned by the reference's proof fixture: for n = 21 it yields queries 109/28/16/11, final 9, one OOD sample per round,
 * and a nonce in every round, exactly the shape SURVEY Appendix A decodes (tests/test_host_only.py).  Host only.
 * Smallest size: n_vars >= folding_factor (PK_ERR_BAD_ARG below it): pk_prove always folds folding_factor variables before the
 * first re-commit, so the blinding scheme of m_0 = 1 (3 variables at fold 4) does not exist here; m_0 >= 2. */
int pk_whir_config_derive(unsigned n_vars, unsigned batch_size, unsigned folding_factor, unsigned starting_log_inv_rate,
                          unsigned security_level, int pow_bits, pk_whir_config *out);
int pk_scheme_create(pk_ctx *ctx, const pk_r1cs *r1cs, size_t num_constraints, size_t num_witnesses, unsigned m, unsigned m_0,
                     const pk_whir_config *whir_witness, const pk_whir_config *whir_for_hiding_spartan, pk_scheme **out);
int pk_scheme_destroy(pk_ctx *ctx, pk_scheme *scheme);
int pk_prove(pk_ctx *ctx, pk_scheme *scheme, const uint64_t *d_witness, size_t n_witness, const uint8_t *rng_seed32,
             uint8_t *transcript_out, size_t cap, size_t *len);
/* The IO pattern (spongefish DomainSeparator) of a proof: WhirR1CSScheme::create_io_pattern(), provekit/common/src/whir_r1cs.rs:28-39.
 * Its bytes fix the sponge IV (Keccak tag, spongefish HashStateWithInstructions::new) and declare every absorb / squeeze / hint
 * of the proof, which spongefish checks operation by operation on both sides (prover/src/whir_r1cs.rs:57-58,
 * verifier/src/whir_r1cs.rs:40-41).
 *   pk_scheme_set_io_pattern    the drop-in caller hands over `scheme.create_io_pattern().as_bytes()`: the library parses it the way
 *                               DomainSeparator::finalize does ("\0"-separated <A|S><count><label> / H<label> / R, neighbours merged),
 *                               REFUSES it (PK_ERR_IO_PATTERN, the first differing operation in pk_last_error) unless it declares
 *                               exactly the operation sequence pk_prove performs for this scheme, and from then on derives the IV
 *                               from those bytes -- so the proof is the one the reference's verifier state expects.  NULL / 0
 *                               restores the library's own restatement.
 *   pk_whir_r1cs_io_pattern     that restatement (host only, no context): provekit's labels are in the tree, the labels inside
 *                               whir's commit_statement / add_whir_proof are whir's as published and UNPINNED here except the four
 *                               hint labels and "pow-nonce", which the in-tree Go verifier's pattern walker names
 *                               (recursive-verifier/app/circuit/common.go:41-100).  Size query with buf = NULL.
 *   pk_io_pattern_check         host only: does `pattern` declare pk_prove's operations for (m_0, configs)?  PK_OK, or
 *                               PK_ERR_IO_PATTERN with the reason in `why`.
 *   pk_scheme_domain_separator  the bytes currently in force.
 * pk_prove itself enforces the pattern in force while it writes the proof (PK_ERR_IO_PATTERN if an operation strays: a library
 * bug, never a caller error). */
int pk_scheme_set_io_pattern(pk_ctx *ctx, pk_scheme *scheme, const uint8_t *pattern, size_t n);
int pk_whir_r1cs_io_pattern(unsigned m_0, const pk_whir_config *whir_witness, const pk_whir_config *whir_for_hiding_spartan,
                            uint8_t *buf, size_t cap, size_t *len);
int pk_io_pattern_check(const uint8_t *pattern, size_t n, unsigned m_0, const pk_whir_config *whir_witness,
                        const pk_whir_config *whir_for_hiding_spartan, char *why, size_t why_cap);
int pk_scheme_domain_separator(const pk_scheme *scheme, char *buf, size_t cap, size_t *len);
/* host only, for capacity planning: the device memory pk_scheme_create will allocate for a prover of this shape (its arena: every
 * buffer one proof needs, ~20.6 x 32 bytes x 2^m at rate 1/2, fold 16, batch 2) -- how many provers fit next to each other */
int pk_scheme_arena_bytes(unsigned m, unsigned m_0, size_t num_witnesses, const pk_whir_config *whir_witness, size_t *bytes);

/* ------------------------------------------------------------------ X4: the R1CS witness builders
 * R1CSSolver::solve_witness_vec (provekit/prover/src/r1cs.rs:29-40): the loop over &[WitnessBuilder] calling
 * WitnessBuilderSolver::solve (provekit/prover/src/witness/witness_builder.rs:27-193; digits.rs:12-59; ram.rs:13-47).  ACVM
 * execution stays on the host; its result (the ACIR witness map, as a dense array indexed by ACIR witness index, Montgomery
 * elements -- noir_to_native is the identity on the limbs) and the challenges the host transcript draws for the
 * WitnessBuilder::Challenge entries (in list order; Challenge reads the transcript and nothing else) are the inputs.
 *   pk_witness_builders_from_postcard   `bytes` = postcard(&Vec<WitnessBuilder>) -- the enum and everything inside it are defined
 *                                        in the reference tree (provekit/common/src/witness/witness_builder.rs:33-117), and this is
 *                                        how the list sits inside a .nps.  Decodes, checks (an input no EARLIER builder produced is a
 *                                        None the reference would unwrap: PK_ERR_BAD_ARG names it; a witness written by several builders keeps the
 *                                        sequential meaning -- the last writer wins, readers see the version of their place in the list),
 *                                        levels the list by data dependence and uploads it.
 *   pk_witness_builders_inspect         the same decode + levelling on the host only (no device): shape of the program.
 *   pk_witness_solve                    d_witness[n_witness] (zeroed, then every solved entry written), d_is_set[n_witness] = 1 where
 *                                        the reference's Vec<Option<F>> is Some.  The points where the reference panics -- inverse of
 *                                        zero, "Higher order bits are not zero", a multiplicity / memory index out of range -- return
 *                                        PK_ERR_UNSATISFIED naming the first builder (in list order) that hits one.
 * fill_witness (random values for the None entries, prover/src/witness/mod.rs:15-30): pk_witness_fill below, or the caller.
 * A pk_witness_program belongs to the context that created it and keeps per-solve state on the device (histograms, the error
 * word): one program per prover thread, like the scheme. */
typedef struct pk_witness_program pk_witness_program;
int pk_witness_builders_from_postcard(pk_ctx *ctx, const uint8_t *bytes, size_t len, pk_witness_program **out,
                                      size_t *n_witnesses, size_t *n_challenges, size_t *n_acir);
int pk_witness_builders_inspect(const uint8_t *bytes, size_t len, size_t *n_builders, size_t *n_witnesses,
                                size_t *n_challenges, size_t *n_acir, size_t *n_levels, size_t *n_items, size_t *consumed,
                                char *err, size_t err_cap);
int pk_witness_solve(pk_ctx *ctx, pk_witness_program *prog, const uint64_t *d_acir, size_t n_acir, const uint64_t *challenges,
                     size_t n_challenges, uint64_t *d_witness, size_t n_witness, uint8_t *d_is_set);
/* The distinct ACIR witness indices the list's WitnessBuilder::Acir entries read, ascending (host only; size query with idx = NULL).
 * The reference's solver does `acir_witness_idx_to_value_map.get_index(..).unwrap()` (witness_builder.rs:36-41): a value missing
 * from the WitnessMap is a panic there.  The dense d_acir array of pk_witness_solve / pk_noir_prove cannot express "missing", so
 * the caller checks its map against this list first (rust/provekit-prover-hip: HipNoirProver::prove names the missing index). */
int pk_witness_program_acir_reads(const pk_witness_program *prog, uint32_t *idx, size_t cap, size_t *n);
/* Placement advice (host only): a list whose dependence depth approaches its length is latency-bound on the device (one level
 * ~ 2.6 us whatever its width; 16 k chained builders = 43 ms) and belongs with the reference's sequential solver on a host core
 * (~60 ns per builder); *prefer_host = 1 then, and the Rust side keeps the stock `solve_witness_vec` for that scheme and hands
 * pk_prove the finished witness (HipNoirProver::new -> Placement::Host).  Estimates in microseconds, from the measurements in
 * DESIGN.md 9; pk_witness_solve itself never refuses a list on these grounds. */
int pk_witness_program_placement(const pk_witness_program *prog, size_t *n_levels, size_t *n_items, double *est_device_us,
                                 double *est_host_us, int *prefer_host);
int pk_witness_program_destroy(pk_ctx *ctx, pk_witness_program *prog);

/* NoirProofSchemeProver::prove after ACVM execution (provekit/prover/src/noir_proof_scheme.rs:63-92):
 *   pk_witness_challenges  HOST ONLY.  The witness transcript: create_witness_io_pattern (noir_proof_scheme.rs:94-109;
 *                          witness_io_pattern.rs:18-41 -- "📜", add_scalars(2, "shape"), add_scalars(n, "pub_inputs") when n > 0,
 *                          challenge_scalars(k, "wb:challenges") when k > 0) over the Skyscraper sponge, seed_witness_merlin
 *                          (noir_proof_scheme.rs:111-133: absorb num_constraints, num_witnesses, then the public input values in
 *                          index order), then one challenge scalar per WitnessBuilder::Challenge in list order
 *                          (witness_builder.rs:94-98).  public_inputs / challenges: Montgomery elements.
 *   pk_witness_fill        fill_witness (prover/src/witness/mod.rs:15-30): every entry with d_is_set == 0 takes
 *                          FieldElement::from(u128 drawn from the proof RNG) -- ChaCha12 under rng_seed32, or under a fresh
 *                          OS-CSPRNG key when NULL (the reference's rng(); on a device set rank 0's key reaches every rank through one
 *                          32-byte all-gather, so all ranks must make the call); *n_filled (may be NULL) = how many.
 *   pk_noir_prove          the three steps and pk_prove in one call, the witness never leaving the device: public values =
 *                          d_acir[public_acir_idx[i]] (Circuit::public_inputs().indices(), ascending), challenges, builders
 *                          (PK_ERR_UNSATISFIED where the reference panics), fill, WhirR1CSProver::prove.  `builders` must not write
 *                          past the scheme's num_witnesses.  rng_seed32 as pk_prove's (NULL in production). */
int pk_witness_challenges(size_t num_constraints, size_t num_witnesses, const uint64_t *public_inputs, size_t n_public,
                          uint64_t *challenges, size_t n_challenges);
int pk_witness_fill(pk_ctx *ctx, uint64_t *d_witness, const uint8_t *d_is_set, size_t n, const uint8_t *rng_seed32,
                    size_t *n_filled);
int pk_noir_prove(pk_ctx *ctx, pk_scheme *scheme, pk_witness_program *builders, const uint64_t *d_acir, size_t n_acir,
                  const uint32_t *public_acir_idx, size_t n_public, const uint8_t *rng_seed32, uint8_t *transcript_out,
                  size_t cap, size_t *len);

/* The library also exports a handful of pk_selftest_* entry points (the host build of the device arithmetic, the proof RNG): test
 * infrastructure for this repository's CPU suite, declared in tools/probes/pk_selftest.h -- not part of the binder's API.
 * Measurement probes (multiplier peak rates, round-trip costs), the device-side run of the self-test op table and the prototypes that
 * were measured and rejected are NOT part of this library: tools/probes builds them into libpk_probes.so (tools/probes/pk_probes.h). */

#ifdef __cplusplus
}
#endif
#endif /* PROVEKIT_HIP_H */
