/*
 * pk_oracle.h -- CPU restatement of the ProveKit WHIR hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for libprovekit_hip. It is NOT part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link
 * or call it. The product path (provekit_amd/, libprovekit_hip.so) never does.
 *
 * Every function cites the reference file:line (paths relative to
 * worldfnd/provekit @ 2025-08-29) whose behaviour it restates.
 *
 * Pinning status (see DESIGN.md "Oracle pinning"):
 *   - Skyscraper v2/v1, sbox, bar, permute : PINNED by the in-tree KATs
 *       (skyscraper/core/src/reference.rs:104-188) and by the 218 Merkle
 *       openings of tooling/provekit-bench/benches/poseidon-1000.np (v1).
 *   - Montgomery multiplier (A1)           : PINNED by the proptest regression
 *       inputs + the algebraic definition a*b*2^-256 mod p checked in Python.
 *   - PoW threshold conversion             : PINNED by skyscraper/core/src/pow.rs:88-103.
 *   - Merkle leaf layout / tree orientation: PINNED by the proof fixture.
 *   - RS-encode, batch stacking, to_coeffs layout, coefficient fold, OOD
 *       evaluation, quadratic-sumcheck binding, blinding sum: PINNED by values
 *       derived from the reference's own proof (tests/golden/fixture_whir.json:
 *       recovered coefficient vectors -> the reference's leaves and Merkle roots,
 *       fold -> the next committed polynomial, OOD answers, final coefficients).
 *   - eq table, cubic sumcheck round, sparse products, eq_accumulate: the
 *       reference holds no golden vectors and the fixture exposes none of their
 *       inputs ("parity unpinned" at value level); pinned to the mathematical
 *       definition by an independent Python big-int restatement
 *       (tests/golden/gen_golden.py) and by the verifier equations in the Go
 *       files under recursive-verifier/app/circuit/.
 *
 * Conventions: a field element (FE) is 4 x uint64 little-endian limbs. Unless a
 * function says "canonical", FEs are in Montgomery form (x*2^256 mod p), which
 * is ark-ff's in-memory representation (provekit/common/src/lib.rs:19).
 */
#ifndef PK_ORACLE_H
#define PK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- A2: ark-ff Fp256<MontBackend> arithmetic (fully reduced) ---- */
void pko_fe_add(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void pko_fe_sub(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void pko_fe_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]); /* a*b*2^-256 mod p */
void pko_fe_to_mont(const uint64_t canon[4], uint64_t out[4]);
void pko_fe_from_mont(const uint64_t mont[4], uint64_t out[4]);
void pko_fe_to_mont_many(const uint64_t *canon, uint64_t *out, size_t n);
void pko_fe_from_mont_many(const uint64_t *mont, uint64_t *out, size_t n);
void pko_fe_pow(const uint64_t base_mont[4], uint64_t exp, uint64_t out[4]);
/* domain generator of the size-2^log_n subgroup, Montgomery form */
void pko_root_of_unity(unsigned log_n, uint64_t out[4]);

/* ---- A1: block_multiplier::scalar_{mul,sqr} (skyscraper/block-multiplier/src/scalar.rs:12-132) ---- */
void pko_scalar_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void pko_scalar_sqr(const uint64_t a[4], uint64_t out[4]);

/* ---- H1: Skyscraper (canonical integers in and out) ---- */
uint8_t pko_sbox(uint8_t v);                                                   /* reference.rs:96-98 */
void pko_bar(const uint64_t x[4], uint64_t out[4]);                            /* reference.rs:80-94 */
void pko_permute(const uint64_t l[4], const uint64_t r[4], uint64_t ol[4], uint64_t orr[4]); /* reference.rs:49-60 */
void pko_compress(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]);  /* reference.rs:41-46, generic.rs:77-102 */
void pko_compress_v1(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]); /* v1.rs:19-32 */
/* CompressManyFn shape: skyscraper/core/src/lib.rs:26; returns -1 on length mismatch (generic.rs:18-25 panics) */
int pko_compress_many(const uint8_t *messages, size_t messages_len, uint8_t *hashes, size_t hashes_len);
int pko_compress_many_v1(const uint8_t *messages, size_t messages_len, uint8_t *hashes, size_t hashes_len);

/* ---- H2/M1/M2: provekit/common/src/skyscraper/whir.rs:20-74 ---- */
/* leaf hash: left fold of compress over w Montgomery FEs; digest returned CANONICAL */
int pko_leaf_hash(const uint64_t *leaf_mont, size_t w, uint64_t digest_canon[4], int version);
/* leaves: n_leaves x w Montgomery FEs, leaf-major. nodes: heap layout, 2*n_leaves FEs,
 * canonical; nodes[1] = root, children of i are 2i, 2i+1, leaf digests at [n_leaves + i]. */
int pko_merkle_commit(const uint64_t *leaves_mont, size_t n_leaves, size_t w, uint64_t *nodes, int version);
/* inner levels only, from canonical leaf digests already placed at nodes[n..2n) */
int pko_merkle_inner(uint64_t *nodes, size_t n_leaves, int version);

/* ---- T1: EvaluationsList::to_coeffs (call sites prover/src/whir_r1cs.rs:195,198) ---- */
void pko_to_coeffs(uint64_t *evals_mont, unsigned n_vars);

/* ---- N1/N2: interleaved RS encode (whir_utilities.go:180-186, whir.go:99,141) ----
 * coeffs: `batch` polynomials of 2^n_vars coefficients each (poly-major).
 * out: rows x (batch*2^fold) leaf-major, rows = 2^(n_vars+log_inv_rate-fold). */
int pko_rs_encode(const uint64_t *coeffs_mont, unsigned batch, unsigned n_vars, unsigned log_inv_rate,
                  unsigned fold, uint64_t *leaves_mont);
/* plain NTT, natural in -> natural out, in place, size 2^log_n (helper used by rs_encode) */
void pko_ntt(uint64_t *data_mont, unsigned log_n);

/* ---- E1: OOD evaluation = univariate Horner (utilities.go:182-190) ---- */
void pko_eval_univariate(const uint64_t *coeffs_mont, size_t n, const uint64_t z_mont[4], uint64_t out[4]);

/* ---- S2: eval_eq (provekit/common/src/utils/sumcheck.rs:146-171) ---- */
void pko_eq_table(const uint64_t *r_mont, unsigned m, uint64_t *out_mont);

/* ---- S3: sumcheck_fold_map_reduce::<4,3> + cubic map (sumcheck.rs:16-104; prover/src/whir_r1cs.rs:284-291)
 * len = current length of each array. If fold != NULL the arrays are folded in place first
 * (the live prefix becomes len/2; the tail is left as the reference leaves it). out = 3 FEs. */
int pko_sumcheck_cubic_round(uint64_t *a, uint64_t *b, uint64_t *c, uint64_t *eq, size_t len,
                             const uint64_t *fold_or_null, uint64_t out[12]);

/* ---- S1/S4: sparse mat-vec (provekit/common/src/sparse_matrix.rs:150-184) ----
 * CSR as the reference stores it: new_row_indices[num_rows], col_indices[nnz],
 * values[nnz] = indices into the interner table (Montgomery FEs). */
int pko_spmv(size_t num_rows, size_t num_cols, const uint32_t *new_row_indices, const uint32_t *col_indices,
             const uint32_t *values, size_t nnz, const uint64_t *interner_mont, const uint64_t *x_mont,
             uint64_t *y_mont);
int pko_spmv_t(size_t num_rows, size_t num_cols, const uint32_t *new_row_indices, const uint32_t *col_indices,
               const uint32_t *values, size_t nnz, const uint64_t *interner_mont, const uint64_t *x_mont,
               uint64_t *y_mont);
/* c = a o b */
void pko_hadamard(const uint64_t *a, const uint64_t *b, uint64_t *c, size_t n);

/* ---- S5: Weights::linear(..).weighted_sum (prover/src/whir_r1cs.rs:401-405) ---- */
void pko_vec_add(const uint64_t *a, const uint64_t *b, uint64_t *c, size_t n);
void pko_vec_axpy(const uint64_t *a, const uint64_t s[4], const uint64_t *b, uint64_t *c, size_t n);
void pko_dot(const uint64_t *w, const uint64_t *f, size_t n, uint64_t out[4]);

/* ---- W1: coefficient fold by k challenges (whir_utilities.go:180-186; utilities.go:15-22)
 * r[0] pairs with index bit 0. in: 2^n coeffs -> out: 2^(n-k). */
void pko_fold_coeffs(const uint64_t *coeffs, unsigned n_vars, const uint64_t *r, unsigned k, uint64_t *out);
/* ---- W2: w[i] += scale * eq(point, i), point = ExpandFromUnivariate(z, n) (utilities.go:182-190) or explicit */
void pko_eq_accumulate_univariate(uint64_t *w, unsigned n_vars, const uint64_t z[4], const uint64_t scale[4]);
void pko_eq_accumulate_point(uint64_t *w, unsigned n_vars, const uint64_t *point, const uint64_t scale[4]);
/* ---- W3: WHIR quadratic sumcheck sub-round (whir_utilities.go:102-125; utilities.go:148-154)
 * h(X) = sum_i f(i,X) w(i,X) over adjacent pairs (2i,2i+1); out = h(0),h(1),h(2).
 * If fold != NULL, f and w are first folded: v'[i] = v[2i] + r (v[2i+1]-v[2i]) (len halves). */
int pko_sumcheck_quadratic_round(uint64_t *f, uint64_t *w, size_t len, const uint64_t *fold_or_null,
                                 uint64_t out[12]);
void pko_fold_pairs(uint64_t *v, size_t len, const uint64_t r[4]);
/* multilinear evals -> coeffs inverse (coeffs -> evals over hypercube) */
void pko_to_evals(uint64_t *coeffs_mont, unsigned n_vars);

/* ---- P1: PoW (skyscraper/core/src/pow.rs:14-41, generic.rs:42-71) ---- */
int pko_pow_threshold(double difficulty, uint64_t out[4]);                 /* pow.rs:14-22,61-82 */
int pko_pow_verify(const uint64_t challenge[4], double difficulty, uint64_t nonce); /* pow.rs:24-26 */
uint64_t pko_pow_solve(const uint64_t challenge[4], double difficulty);   /* smallest valid nonce, bias 0.01 */

/* ---- the proof's random draws (restatement of the HIP library's keyed expansion, csrc/prover.hip random_fe_kernel; the reference
 * draws from thread_rng, zk_utils.rs:13-22) and the opened rows of a codeword, canonical ---- */
void pko_chacha_block(const uint8_t key[32], uint64_t counter, uint32_t n0, uint32_t n1, int rounds, uint32_t out[16]);
void pko_random_fe(const uint8_t key[32], uint32_t stream, uint64_t *out, size_t n);
void pko_gather_rows_canonical(const uint64_t *leaves_mont, size_t width, const uint64_t *idx, size_t k, uint64_t *out_canon);

/* misc */
int pko_num_threads(void);
void pko_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
