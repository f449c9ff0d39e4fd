// prover.hip -- host driver: WhirR1CSProver::prove (provekit/prover/src/whir_r1cs.rs:42-100) over the device kernels.
//
// This is the compiled host side of the drop-in (the reference's is Rust): transcript, challenge bookkeeping and the
// O(m_0^2) blinding algebra run here on the CPU; every data-parallel step is a call into this library's kernels on
// buffers that never leave HBM.  One proof allocates nothing: all device memory comes from a per-scheme arena.
//
//   pk_prove
//    +- batch_commit (whir_r1cs.rs:182-209): mask / random polynomial on device, to_coeffs x2, commit_batch
//    +- run_zk_sumcheck (whir_r1cs.rs:228-369): witness bounds, eq table, blinding commitment, m_0 cubic rounds,
//    |                                          small WHIR proof of the blinding polynomial
//    +- external rows, weighted sums, claimed_evaluations hint (whir_r1cs.rs:81-91)
//    +- whir_prove (whir::Prover::prove; structure pinned by recursive-verifier/app/circuit/whir.go:51-220)
#include <sys/random.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <chrono>
#include <cstdlib>
#include <string>
#include <vector>

#include "ctx.hpp"
#include "shard_map.hpp"
#include "transcript.hpp"

using namespace pk;

namespace pk {
int commit_into(pk_ctx* ctx, const uint64_t* const* d_coeffs, unsigned batch, unsigned n_vars, unsigned log_inv_rate, unsigned fold,
                uint64_t* d_leaves, uint64_t* d_nodes, uint64_t* d_scratch, pk_commit_layout* layout_out);
int open_raw(pk_ctx* ctx, const uint64_t* d_leaves, const uint64_t* d_nodes, size_t n_leaves, size_t width, const pk_commit_layout& lay,
             const uint64_t* indices, size_t k, int canonical_leaves, uint64_t* leaves_out, uint64_t* sibling_digests, uint64_t* auth_paths);
unsigned shard_factor(const pk_ctx* ctx, size_t rows);
size_t commit_scratch_fes(const pk_ctx* ctx, size_t rows, size_t width);
int dot_rows_x(pk_ctx* ctx, const uint64_t* d_w, size_t row_stride, unsigned nrows, const uint64_t* d_f, const uint64_t* d_g, size_t n, uint64_t* out,
               bool defer);
int lincomb2(pk_ctx* ctx, uint64_t* d_out, const uint64_t* d_a, const uint64_t* beta, const uint64_t* d_b, size_t n);
int fold_pairs2(pk_ctx* ctx, const uint64_t* d_v0, uint64_t* d_out0, const uint64_t* d_v1, uint64_t* d_out1, size_t len, const uint64_t* r);
// latency mode (ctx.hpp): launch-only rounds, gated on a challenge the host publishes later (mle.hip, reduce.hpp)
int fold_pairs2_gated(pk_ctx* ctx, const uint64_t* d_v0, uint64_t* d_out0, const uint64_t* d_v1, uint64_t* d_out1, size_t len, const uint64_t* r,
                      unsigned gate_seq);
int sumcheck_cubic_launch(pk_ctx* ctx, uint64_t* d_a, uint64_t* d_b, uint64_t* d_c, uint64_t* d_eq, size_t len, const uint64_t* fold_or_null,
                          unsigned gate_seq, unsigned* red_seq_out);
int sumcheck_quadratic_launch(pk_ctx* ctx, const uint64_t* d_f, const uint64_t* d_w, size_t len, const uint64_t* fold_or_null, unsigned gate_seq,
                              uint64_t* d_f_out, uint64_t* d_w_out, unsigned* red_seq_out);
int sumcheck_collect_spin(pk_ctx* ctx, unsigned red_seq, uint64_t out[12]);
unsigned sumcheck_gate_next(pk_ctx* ctx);
int sumcheck_gate_check(pk_ctx* ctx);
void sumcheck_gate_clear(pk_ctx* ctx);
void sumcheck_gate_publish(pk_ctx* ctx, unsigned gate_seq, const uint64_t challenge[4]);
int witness_bounds_strided(pk_ctx* ctx, const pk_r1cs* r, const uint64_t* d_z, unsigned m0, unsigned stride, unsigned offset, uint64_t* d_a,
                           uint64_t* d_b, uint64_t* d_c);
int external_row_range(pk_ctx* ctx, const pk_r1cs* r, const uint64_t* d_eq_alpha, size_t first, size_t last, uint64_t* d_out);
void witness_program_shape(const pk_witness_program* p, size_t* n_witnesses, size_t* n_challenges, size_t* n_acir);  // witness.hip
}

struct pk_scheme {
    const pk_r1cs* r1cs = nullptr;
    size_t num_constraints = 0, num_witnesses = 0;
    unsigned m = 0, m_0 = 0;
    pk_whir_config whir_witness{}, whir_hiding{};
    char* arena = nullptr;
    size_t arena_bytes = 0;
    std::string domain_separator;
    char* noir_witness = nullptr;  // pk_noir_prove: num_witnesses elements + num_witnesses is-set bytes, allocated on first use
    pk_ctx* side = nullptr;        // latency mode: a second context (stream, workspace) of the same device for the blinding commitment
};

namespace {

// ------------------------------------------------------------------ device CSPRNG
// The reference draws the ZK mask, the random polynomial g and the Spartan blinding univariates from thread_rng
// (provekit/common/src/utils/zk_utils.rs:13-22, provekit/prover/src/whir_r1cs.rs:197,212-221): rand's ThreadRng, i.e. ChaCha12
// seeded from the OS.  Here: one 256-bit key per proof (getrandom(2) inside pk_prove unless the caller injects a seed -- a test
// hook), expanded on the device with the same cipher -- the ChaCha block function (RFC 8439 quarter rounds and state layout),
// 12 rounds.  Elements 2j and 2j+1 of draw `stream` share the blocks (counter = j, nonce = {stream, attempt}): a block holds
// two 254-bit candidates, the first for element 2j, the second for 2j+1, each accepted iff < p (what ark-ff's Fp::rand does,
// so every element is uniform on [0, p)); an element whose candidate was rejected takes its candidate of the next attempt.
struct RngKey {
    u32 k[8];
};
constexpr int PK_RNG_ROUNDS = 12;
#define PK_QR(a, b, c, d)                    \
    a += b; d ^= a; d = (d << 16) | (d >> 16); \
    c += d; b ^= c; b = (b << 12) | (b >> 20); \
    a += b; d ^= a; d = (d << 8) | (d >> 24);  \
    c += d; b ^= c; b = (b << 7) | (b >> 25)
__host__ __device__ __forceinline__ void chacha_block(const RngKey& key, u64 counter, u32 n0, u32 n1, int rounds, u32 (&out)[16]) {
    u32 s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key.k[0], key.k[1], key.k[2], key.k[3],
                 key.k[4],    key.k[5],    key.k[6],    key.k[7],    (u32)counter, (u32)(counter >> 32), n0, n1};
    u32 x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = s[i];
#pragma unroll 1
    for (int r = 0; r < rounds / 2; r++) {
        PK_QR(x[0], x[4], x[8], x[12]);
        PK_QR(x[1], x[5], x[9], x[13]);
        PK_QR(x[2], x[6], x[10], x[14]);
        PK_QR(x[3], x[7], x[11], x[15]);
        PK_QR(x[0], x[5], x[10], x[15]);
        PK_QR(x[1], x[6], x[11], x[12]);
        PK_QR(x[2], x[7], x[8], x[13]);
        PK_QR(x[3], x[4], x[9], x[14]);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}
#undef PK_QR
// A lane walks its pairs j = g, g + stride, ... as a state machine (pair, attempt): one ChaCha block per loop iteration, whichever pair and
// attempt the lane is at.  (Until round 5 the retry loop sat INSIDE the loop over pairs, so a wavefront repeated a pair's block until its
// unluckiest lane -- 128 candidates, each rejected with probability 0.244 -- was done: ~4 blocks per pair instead of the 1.43 a lane needs.
// Now the waiting averages out over a lane's pairs: launched with ~8 pairs per lane, a wavefront runs ~17 blocks for 8 pairs.)
__global__ __launch_bounds__(256) void random_fe_kernel(fe* __restrict__ out, size_t n, RngKey key, u32 stream) {
    PK_LATENCY_PRIO();
    const size_t stride = (size_t)gridDim.x * blockDim.x, pairs = (n + 1) / 2;
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32 attempt = 0;
    bool done0 = false, done1 = 2 * j + 1 >= n;
    while (j < pairs) {
        u32 blk[16];
        chacha_block(key, (u64)j, stream, attempt, PK_RNG_ROUNDS, blk);
#pragma unroll
        for (int half = 0; half < 2; half++) {
            if (half == 0 ? done0 : done1) continue;
            fe x;
#pragma unroll
            for (int w = 0; w < 8; w++) x.v[w] = blk[8 * half + w];
            x.v[7] &= 0x3fffffffu;  // < 2^254
            u32 borrow = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) (void)__builtin_subc(x.v[k], kPlimb(k), borrow, &borrow);
            if (borrow) {  // x < p: accepted
                fe_store(out + 2 * j + half, x);
                if (half == 0) done0 = true;
                else done1 = true;
            }
        }
        if (done0 && done1) {
            j += stride;
            attempt = 0;
            done0 = false;
            done1 = 2 * j + 1 >= n;
        } else {
            attempt++;
        }
    }
}
// ~8 pairs per lane (see the kernel), at least one wavefront per SIMD's worth of workgroups when the draw is long enough
inline unsigned random_fe_grid(const pk_ctx* ctx, size_t n) {
    const size_t pairs = (n + 1) / 2;
    size_t blocks = (pairs + 256 * 8 - 1) / (256 * 8);
    const size_t cap = (size_t)ctx->num_cus * 8;
    if (blocks < 1) blocks = 1;
    return (unsigned)(blocks < cap ? blocks : cap);
}
// draws of one proof (the `stream` word of the nonce)
enum { RNG_MASK = 1, RNG_G = 2, RNG_BLIND = 3, RNG_MASK_B = 4, RNG_G_B = 5, RNG_FILL = 6 };

// latency mode: a gated kernel is in the queue waiting for a challenge; whatever path leaves the scope, it must be released
// (with a zero challenge on an error path: the proof is abandoned anyway) so that the stream can drain
struct PendingGate {
    pk_ctx* c;
    unsigned seq = 0;
    explicit PendingGate(pk_ctx* ctx) : c(ctx) {}
    void arm(unsigned s) { seq = s; }
    void publish(const fe& challenge) {
        if (!seq) return;
        const long stall_us = test_hook(PK_HOOK_GATE_STALL_US);  // the test-suite's stand-in for a host thread that was stopped
        if (stall_us > 0) usleep((useconds_t)stall_us);
        uint64_t w[4];
        h_store(w, challenge);
        sumcheck_gate_publish(c, seq, w);
        seq = 0;
    }
    ~PendingGate() {
        if (seq) publish(fe_zero());
    }
};

struct Arena {
    char* base;
    size_t cap, off = 0;
    fe* alloc(size_t n_fe) {
        size_t bytes = ((n_fe * 32 + 255) / 256) * 256;
        if (off + bytes > cap) return nullptr;
        fe* p = (fe*)(base + off);
        off += bytes;
        return p;
    }
};

#define CK(expr)              \
    do {                      \
        int _rc = (expr);     \
        if (_rc) return _rc;  \
    } while (0)
#define ALLOC(var, n)                                                                   \
    fe* var = A.alloc(n);                                                               \
    if (!var) return set_err(ctx, PK_ERR_OOM, "prover arena exhausted (%s)", #var)

inline uint64_t* U(fe* p) { return (uint64_t*)p; }
inline const uint64_t* U(const fe* p) { return (const uint64_t*)p; }

// ExpandFromUnivariate (recursive-verifier/app/utilities/utilities.go:182-190): point[n-1-i] = z^(2^i)
void expand_from_univariate(const fe& z, unsigned n, fe* out) {
    fe acc = z;
    for (unsigned i = 0; i < n; i++) {
        out[n - 1 - i] = acc;
        acc = h_mul(acc, acc);
    }
}

// ------------------------------------------------------------------ one proof over a device set
// Besides the commits (tree.hip), the linear-size arrays of a sharded proof are split over the G ranks:
//   * the Spartan sumcheck's a, b, c, eq by the LOW index bits (rank g holds i = g mod G at local index i / G): the leading
//     variable is folded first and pairs i with i + len/2 (sumcheck.rs:28-33), both on one rank, so the existing kernels run
//     unchanged on the local arrays;
//   * the WHIR sumcheck's polynomial and weight tables, the equality weights, the statement weights (external rows) and the
//     OOD evaluations by contiguous BLOCKS (the high index bits): WHIR folds the lowest variable first (pairs 2i, 2i+1).
// Per round every rank reduces its share and the 96 bytes are summed over the ranks (ctx->red_across: reduce.hpp,
// comm_collect_fe); once the local length falls to SHARD_MIN_LOCAL the arrays are all-gathered and the tail of the sumcheck
// runs replicated.  Every sum is an exact field sum, so the transcript is byte-identical to the lone prover's.
constexpr size_t SHARD_MIN_LOCAL = (size_t)1 << 12;

struct Across {  // scope in which this context's reductions are partial sums to be added over the ranks
    pk_ctx* c;
    bool on;
    int rc = PK_OK;
    Across(pk_ctx* ctx, bool enable, const fe* scales = nullptr) : c(ctx), on(enable) {
        if (on) {
            rc = red_across_begin(c);
            c->red_scales = scales;
        }
    }
    ~Across() {
        if (on) {
            c->red_across = false;
            c->red_scales = nullptr;
        }
    }
};
// eq((x_0 .. x_{lg-1}), bits of g), x_0 <-> the most significant of the lg bits (eval_eq's order, sumcheck.rs:146-171)
fe eq_bits(const fe* x, unsigned lg, unsigned g) {
    fe acc = fe_one();
    for (unsigned t = 0; t < lg; t++) acc = h_mul(acc, ((g >> (lg - 1 - t)) & 1u) ? x[t] : h_sub(fe_one(), x[t]));
    return acc;
}
// all-gather of the ranks' local arrays: gathered[r * len + j]; STRIDED additionally re-interleaves to full[j * G + r]
__global__ __launch_bounds__(256) void interleave_fe_kernel(const fe* __restrict__ gathered, fe* __restrict__ full, size_t total, unsigned G) {
    PK_LATENCY_PRIO();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) fe_store(full + i, fe_load(gathered + shard_gathered_slot(i, total, G)));
}
int gather_blocks(pk_ctx* ctx, const fe* local, size_t len, fe* full) { return comm_all_gather(ctx, local, full, 32 * len); }
int gather_strided(pk_ctx* ctx, const fe* local, size_t len, fe* tmp, fe* full) {
    const unsigned G = (unsigned)comm_world(ctx);
    int rc = comm_all_gather(ctx, local, tmp, 32 * len);
    if (rc) return rc;
    const size_t total = len * G;
    interleave_fe_kernel<<<(unsigned)((total + 255) / 256), 256, 0, ctx->stream>>>(tmp, full, total, G);
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}
// sum_i c[i] z^i, split into the ranks' blocks when the polynomial is long: rank r evaluates its block in z and the ranks'
// values are combined as sum_r z^(r*B) * partial_r
int eval_univariate_multi_x(pk_ctx* ctx, const fe* const* d_polys, unsigned np, size_t n, const fe& z, fe* out) {
    const unsigned G = (unsigned)comm_world(ctx);
    uint64_t zz[4], o[8];
    h_store(zz, z);
    const uint64_t* ptrs[2] = {nullptr, nullptr};
    if (G > 1 && n / G >= 4 * SHARD_MIN_LOCAL) {
        const size_t B = n / G;
        fe scales[PK_MAX_RANKS];
        const fe zB = h_pow(z, B);
        scales[0] = fe_one();
        for (unsigned r = 1; r < G; r++) scales[r] = h_mul(scales[r - 1], zB);
        for (unsigned q = 0; q < np; q++) ptrs[q] = U(d_polys[q] + (size_t)comm_rank(ctx) * B);
        Across ac(ctx, true, scales);
        CK(ac.rc);
        CK(eval_univariate_multi(ctx, ptrs, np, B, zz, o));
    } else {
        for (unsigned q = 0; q < np; q++) ptrs[q] = U(d_polys[q]);
        CK(eval_univariate_multi(ctx, ptrs, np, n, zz, o));
    }
    for (unsigned q = 0; q < np; q++) out[q] = h_load(o + 4 * q);
    return PK_OK;
}
int eval_univariate_x(pk_ctx* ctx, const fe* d_poly, size_t n, const fe& z, fe& out) { return eval_univariate_multi_x(ctx, &d_poly, 1, n, z, &out); }

// ------------------------------------------------------------------ S6: blinding algebra (host, O(m_0^2))
fe eval_cubic(const fe c[4], const fe& x) {  // provekit/common/src/utils/sumcheck.rs:174-176
    return h_add(c[0], h_mul(x, h_add(c[1], h_mul(x, h_add(c[2], h_mul(x, c[3]))))));
}
// compute_blinding_coefficients_for_round (provekit/prover/src/whir_r1cs.rs:103-170)
void blinding_coefficients_for_round(const std::vector<fe>& g /*4 per variable*/, size_t compute_for, const fe* alphas, fe out[4]) {
    const size_t n = g.size() / 4;
    bool all_fixed = false;
    if (compute_for == n) {
        all_fixed = true;
        compute_for = n - 1;
    }
    fe prefix_sum = fe_zero();
    for (size_t i = 0; i < compute_for; i++) prefix_sum = h_add(prefix_sum, eval_cubic(&g[4 * i], alphas[i]));
    fe suffix_sum = fe_zero();
    const fe zero = fe_zero(), one = fe_one();
    for (size_t i = compute_for + 1; i < n; i++)
        suffix_sum = h_add(suffix_sum, h_add(eval_cubic(&g[4 * i], zero), eval_cubic(&g[4 * i], one)));
    fe prefix_multiplier = fe_one();
    for (size_t i = 0; i < n - 1 - compute_for; i++) prefix_multiplier = h_add(prefix_multiplier, prefix_multiplier);
    fe suffix_multiplier = h_mul(prefix_multiplier, h_half());
    fe constant = h_add(h_mul(prefix_multiplier, prefix_sum), h_mul(suffix_multiplier, suffix_sum));
    const fe* cur = &g[4 * compute_for];
    fe c[4] = {h_add(h_mul(prefix_multiplier, cur[0]), constant), h_mul(prefix_multiplier, cur[1]), h_mul(prefix_multiplier, cur[2]),
               h_mul(prefix_multiplier, cur[3])};
    if (all_fixed) {
        out[0] = eval_cubic(c, alphas[compute_for]);
        out[1] = out[2] = out[3] = fe_zero();
        return;
    }
    for (int i = 0; i < 4; i++) out[i] = c[i];
}

// ------------------------------------------------------------------ STIR query indices
// recursive-verifier/app/circuit/whir_utilities.go:48-77: per query ceil(log2(folded)/8) bytes, big-endian, low bits kept;
// then sorted + deduplicated as whir does.
std::vector<uint64_t> stir_queries(Transcript& T, size_t domain_size, unsigned fold, unsigned num_queries) {
    const size_t folded = domain_size >> fold;
    unsigned bits = ilog2(folded);
    const size_t nbytes = (bits + 7) / 8;
    std::vector<uint8_t> raw(nbytes * num_queries);
    if (!raw.empty()) T.challenge_bytes(raw.data(), raw.size());
    std::vector<uint64_t> idx(num_queries);
    for (unsigned q = 0; q < num_queries; q++) {
        uint64_t v = 0;
        for (size_t j = 0; j < nbytes; j++) v = (v << 8) | raw[q * nbytes + j];
        idx[q] = v & (folded - 1);
    }
    std::sort(idx.begin(), idx.end());
    idx.erase(std::unique(idx.begin(), idx.end()), idx.end());
    return idx;
}

void pow_round(pk_ctx* ctx, Transcript& T, double bits, int* rc) {
    if (bits <= 0.0) return;
    uint8_t challenge[32];
    T.challenge_bytes(challenge, 32);
    uint64_t nonce = 0;
    *rc = pow_solve_x(ctx, challenge, bits, &nonce, comm_world(ctx) > 1);  // nonce ranges striped over the ranks of a device set
    uint8_t be[8];
    for (int i = 0; i < 8; i++) be[i] = (uint8_t)(nonce >> (56 - 8 * i));  // utilities.go:89-95
    T.add_bytes(be, 8);
}

// hints: stir_answers = Vec<Vec<F>> and merkle_proof = ark MultiPath, ark-serialize uncompressed (common.go:36-61)
int emit_opening_hints(pk_ctx* ctx, Transcript& T, const fe* d_leaves, const fe* d_nodes, size_t n_leaves, size_t width,
                       const pk_commit_layout& lay, const std::vector<uint64_t>& idx) {
    const size_t k = idx.size();
    const unsigned logn = ilog2(n_leaves);
    const size_t plen = logn ? logn - 1 : 0;
    std::vector<uint64_t> leaves(4 * k * width), sib(4 * (k ? k : 1)), paths(4 * (k * plen ? k * plen : 1));
    CK(open_raw(ctx, U(d_leaves), U(d_nodes), n_leaves, width, lay, idx.data(), k, /*canonical=*/1, leaves.data(), sib.data(), paths.data()));
    const auto ser0 = std::chrono::steady_clock::now();
    std::vector<uint8_t> buf;
    auto put_u64 = [&](uint64_t v) {
        for (int i = 0; i < 8; i++) buf.push_back((uint8_t)(v >> (8 * i)));
    };
    put_u64(k);
    for (size_t q = 0; q < k; q++) {
        put_u64(width);
        const uint8_t* b = (const uint8_t*)(leaves.data() + 4 * q * width);
        buf.insert(buf.end(), b, b + 32 * width);
    }
    T.hint(buf.data(), buf.size());
    size_t len = 0;
    pk_multipath_serialize(idx.data(), k, plen, sib.data(), paths.data(), nullptr, 0, &len);
    std::vector<uint8_t> mp(len ? len : 1);
    CK(pk_multipath_serialize(idx.data(), k, plen, sib.data(), paths.data(), mp.data(), len, &len));
    T.hint(mp.data(), len);
    T.hint_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - ser0).count();
    return PK_OK;
}

// ------------------------------------------------------------------ WHIR
struct Commitment {  // whir::committer::Witness
    unsigned n_vars = 0, batch = 0;
    fe* polys[4] = {};  // coefficient form
    fe* evals[4] = {};  // the same polynomials as evaluation tables, when the committer still holds them (else null)
    fe* leaves = nullptr;
    fe* nodes = nullptr;
    size_t rows = 0, width = 0;
    pk_commit_layout layout{};    // how commit_into laid the codeword out (shards, leaf encoding): recorded, not recomputed
    std::vector<fe> ood_points;   // Montgomery
    std::vector<fe> ood_answers;  // [poly][point], Montgomery
    fe beta;                      // batching randomness
};

// CommitmentWriter::commit_batch (call site provekit/prover/src/whir_r1cs.rs:200-206; transcript order mtUtilities.go:51-76)
// the commitment's device work (RS-encode, leaf hashes, tree) on `ctx`'s stream, nothing read back: the half of whir_commit that needs no
// transcript -- in latency mode the blinding commitment's runs on a second stream while the witness commitment fills the chip
int whir_commit_compute(pk_ctx* ctx, Arena& A, const pk_whir_config& cfg, fe* const* polys, unsigned batch, Commitment& C) {
    C.n_vars = cfg.n_vars;
    C.batch = batch;
    const unsigned k = cfg.folding_factor;
    C.rows = (size_t)1 << (cfg.n_vars + cfg.starting_log_inv_rate - k);
    C.width = (size_t)batch << k;
    for (unsigned b = 0; b < batch; b++) C.polys[b] = polys[b];
    ALLOC(leaves, C.rows * C.width / shard_factor(ctx, C.rows));  // a rank of a device set keeps only its rows (tree.hip)
    ALLOC(nodes, 2 * C.rows);
    C.leaves = leaves;
    C.nodes = nodes;
    CK(ensure_ws(ctx, commit_scratch_fes(ctx, C.rows, C.width) * 32));
    const uint64_t* ptrs[4];
    for (unsigned b = 0; b < batch; b++) ptrs[b] = U(polys[b]);
    return commit_into(ctx, ptrs, batch, cfg.n_vars, cfg.starting_log_inv_rate, k, U(leaves), U(nodes), (uint64_t*)ctx->d_ws, &C.layout);
}
int whir_commit_transcript(pk_ctx* ctx, const pk_whir_config& cfg, const fe& root, Transcript& T, Commitment& C);
int whir_commit(pk_ctx* ctx, Arena& A, const pk_whir_config& cfg, fe* const* polys, unsigned batch, Transcript& T, Commitment& C) {
    CK(whir_commit_compute(ctx, A, cfg, polys, batch, C));
    fe root;
    CK(read_root(ctx, U(C.nodes), C.rows, (uint64_t*)root.v));
    return whir_commit_transcript(ctx, cfg, root, T, C);
}
// ... and the half that talks: root, OOD points and answers, batching randomness (mtUtilities.go:51-76)
int whir_commit_transcript(pk_ctx* ctx, const pk_whir_config& cfg, const fe& root, Transcript& T, Commitment& C) {
    const unsigned batch = C.batch;
    fe* const* polys = C.polys;
    T.add_canon(root);
    C.ood_points.resize(cfg.commitment_ood_samples);
    T.challenge_scalars(C.ood_points.data(), C.ood_points.size());
    C.ood_answers.resize((size_t)batch * C.ood_points.size());
    for (size_t j = 0; j < C.ood_points.size(); j++) {
        // the polynomials of a batch are evaluated at the same point: two per launch (and per host round trip)
        for (unsigned b = 0; b < batch; b += 2) {
            const unsigned np = batch - b >= 2 ? 2 : 1;
            fe ans[2];
            CK(eval_univariate_multi_x(ctx, polys + b, np, (size_t)1 << cfg.n_vars, C.ood_points[j], ans));
            for (unsigned q = 0; q < np; q++) C.ood_answers[(b + q) * C.ood_points.size() + j] = ans[q];
        }
    }
    for (unsigned b = 0; b < batch; b++) T.add_scalars(&C.ood_answers[b * C.ood_points.size()], C.ood_points.size());
    C.beta = batch > 1 ? T.challenge_scalar() : fe_one();  // whir draws the batching randomness only for a real batch
    return PK_OK;
}

// whir::Prover::prove with `n_weights` linear statement weights: evaluation tables over the 2^n hypercube of which only the
// first weight_len[i] entries are stored -- the rest is zero by construction (create_combined_statement_over_two_polynomials
// zero-extends each row, whir_r1cs.rs:382-412), so nothing is spent on the zero half.
// On a device set (and a polynomial long enough) the sumcheck tables, the weights and the OOD evaluations are split into the
// ranks' blocks; of a sharded weight only the entries inside this rank's block need to be present in d_weights[i].
bool whir_sharded(const pk_ctx* ctx, unsigned n_vars) {
    const size_t G = (size_t)comm_world(ctx);
    return G > 1 && (((size_t)1 << n_vars) / G) >= 2 * SHARD_MIN_LOCAL;
}
int whir_prove(pk_ctx* ctx, Arena& A, const pk_whir_config& cfg, const Commitment& C, fe* const* d_weights, const size_t* weight_len,
               unsigned n_weights, Transcript& T) {
    const unsigned n = cfg.n_vars, k = cfg.folding_factor;
    const size_t N = (size_t)1 << n;
    const unsigned G = (unsigned)comm_world(ctx), lgG = ilog2(G), rank = (unsigned)comm_rank(ctx);
    bool sharded = whir_sharded(ctx, n);              // the sumcheck tables are this rank's block [off, off + len)
    const size_t B0 = sharded ? N / G : N, off0 = sharded ? (size_t)rank * B0 : 0;
    // working polynomial c = sum_b beta^b poly_b (mtUtilities.go:98-114), whole on every rank: it is folded and re-committed
    ALLOC(d_c, N);
    // sum_b beta^b x_b over `cnt` entries from `from` of coefficient tables or evaluation tables: the first two in one pass
    auto batch_combine = [&](fe* dst, fe* const* x, size_t from, size_t cnt) -> int {
        if (C.batch == 1) return pk_memcpy_d2d(ctx, dst, x[0] + from, 32 * cnt);
        uint64_t s[4];
        h_store(s, C.beta);
        int rc = lincomb2(ctx, U(dst), U(x[0] + from), s, U(x[1] + from), cnt);
        fe bp = h_mul(C.beta, C.beta);
        for (unsigned b = 2; b < C.batch && !rc; b++) {
            h_store(s, bp);
            rc = pk_fe_axpy(ctx, U(dst), s, U(x[b] + from), cnt);
            bp = h_mul(bp, C.beta);
        }
        return rc;
    };
    CK(batch_combine(d_c, C.polys, 0, N));
    // sumcheck operands: p = evaluations of c over the hypercube, w = combined weights; ping-pong halves
    fe* bp_[2];
    fe* bw_[2];
    ALLOC(p0, B0);
    ALLOC(p1, B0 / 2 ? B0 / 2 : 1);
    ALLOC(w0, B0);
    ALLOC(w1, B0 / 2 ? B0 / 2 : 1);
    bp_[0] = p0; bp_[1] = p1; bw_[0] = w0; bw_[1] = w1;
    bool have_evals = true;
    for (unsigned b = 0; b < C.batch; b++) have_evals = have_evals && C.evals[b] != nullptr;
    if (have_evals) {
        CK(batch_combine(p0, C.evals, off0, B0));  // to_evals is linear: combine the tables the committer kept instead of transforming d_c
    } else if (!sharded) {
        CK(pk_to_evals_into(ctx, U(d_c), U(p0), n));
    } else {
        ALLOC(d_ev, N);
        CK(pk_to_evals_into(ctx, U(d_c), U(d_ev), n));
        CK(pk_memcpy_d2d(ctx, p0, d_ev + off0, 32 * B0));
    }
    // equality weights of `q` points (each nv coordinates, variable 0 <-> the top index bit) scaled by scales[j], accumulated
    // into a weight table: the whole table, or -- sharded -- this rank's block, whose top lgG index bits are the rank: that
    // factor of eq goes into the scale and the table is built over the remaining variables
    auto eq_weights = [&](fe* dst, unsigned nv, std::vector<fe>& pts, std::vector<fe>& scales, size_t q, int overwrite) -> int {
        if (!sharded) return pk_eq_accumulate(ctx, U(dst), nv, (const uint64_t*)pts.data(), (const uint64_t*)scales.data(), (unsigned)q, overwrite);
        const unsigned nl = nv - lgG;
        std::vector<fe> lp(q * (nl ? nl : 1)), ls(q ? q : 1);
        for (size_t j = 0; j < q; j++) {
            ls[j] = h_mul(scales[j], eq_bits(&pts[j * nv], lgG, rank));
            for (unsigned t = 0; t < nl; t++) lp[j * nl + t] = pts[j * nv + lgG + t];
        }
        return pk_eq_accumulate(ctx, U(dst), nl, (const uint64_t*)lp.data(), (const uint64_t*)ls.data(), (unsigned)q, overwrite);
    };
    // initial combination randomness; weights = sum gamma^i w_i over [OOD constraints..., statement weights...]
    fe gamma = T.challenge_scalar();
    fe g = fe_one();
    {
        const size_t q = C.ood_points.size();
        std::vector<fe> pts(q * (n ? n : 1)), scales(q ? q : 1);
        for (size_t j = 0; j < q; j++) {
            expand_from_univariate(C.ood_points[j], n, &pts[j * n]);
            scales[j] = g;
            g = h_mul(g, gamma);
        }
        CK(eq_weights(w0, n, pts, scales, q, /*overwrite=*/1));
        for (unsigned i = 0; i < n_weights; i++) {
            uint64_t s[4];
            h_store(s, g);
            const size_t hi = weight_len[i] < off0 + B0 ? weight_len[i] : off0 + B0;  // the part of the weight inside this block
            if (hi > off0) CK(pk_fe_axpy(ctx, U(w0), s, U(d_weights[i] + off0), hi - off0));
            g = h_mul(g, gamma);
        }
    }
    int cur = 0;
    size_t len = B0;         // local length of p and w
    std::vector<fe> all_r;  // every folding challenge, in squeeze order
    // latency mode (one GPU): round t+1's kernel -- and after the last round the fold -- is enqueued BEFORE round t's result is read,
    // gated on the challenge the host publishes once it has squeezed it; the round trip then costs the link, not a launch + sync
    auto sumcheck_rounds_pipelined = [&](unsigned rounds, std::vector<fe>& rs) -> int {
        rs.clear();
        if (!rounds) return PK_OK;
        unsigned red_cur = 0, red_next = 0;
        CK(sumcheck_quadratic_launch(ctx, U(bp_[cur]), U(bw_[cur]), len, nullptr, 0, nullptr, nullptr, &red_cur));
        for (unsigned t = 0; t < rounds; t++) {
            PendingGate gate(ctx);
            // what consumes this round's challenge: the next round (folding first), or the closing fold
            const bool more = t + 1 < rounds;
            if (more || len >= 2) {
                gate.arm(sumcheck_gate_next(ctx));
                if (more) CK(sumcheck_quadratic_launch(ctx, U(bp_[cur]), U(bw_[cur]), len, nullptr, gate.seq, U(bp_[1 - cur]), U(bw_[1 - cur]), &red_next));
                else CK(fold_pairs2_gated(ctx, U(bp_[cur]), U(bp_[1 - cur]), U(bw_[cur]), U(bw_[1 - cur]), len, nullptr, gate.seq));
                cur = 1 - cur;
                len /= 2;
            }
            uint64_t out[12];
            CK(sumcheck_collect_spin(ctx, red_cur, out));
            fe h[3] = {h_load(out), h_load(out + 4), h_load(out + 8)};
            T.add_scalars(h, 3);
            const fe fold = T.challenge_scalar();
            gate.publish(fold);
            rs.push_back(fold);
            all_r.push_back(fold);
            red_cur = red_next;
        }
        return PK_OK;
    };
    auto sumcheck_rounds = [&](unsigned rounds, std::vector<fe>& rs) -> int {
        if (ctx->latency_mode && G == 1 && rounds && len >= ((size_t)1 << rounds)) return sumcheck_rounds_pipelined(rounds, rs);
        rs.clear();
        bool have_fold = false;
        fe fold = fe_zero();
        Across ac(ctx, sharded);  // sharded: h(0), h(1), h(2) are sums over the ranks' blocks
        CK(ac.rc);
        for (unsigned t = 0; t < rounds; t++) {
            uint64_t out[12], f[4];
            if (!have_fold) {
                CK(pk_sumcheck_quadratic_round(ctx, U(bp_[cur]), U(bw_[cur]), len, nullptr, nullptr, nullptr, out));
            } else {
                h_store(f, fold);
                CK(pk_sumcheck_quadratic_round(ctx, U(bp_[cur]), U(bw_[cur]), len, f, U(bp_[1 - cur]), U(bw_[1 - cur]), out));
                cur = 1 - cur;
                len /= 2;
            }
            fe h[3] = {h_load(out), h_load(out + 4), h_load(out + 8)};
            T.add_scalars(h, 3);
            fold = T.challenge_scalar();
            have_fold = true;
            rs.push_back(fold);
            all_r.push_back(fold);
        }
        if (have_fold && len >= 2) {  // apply the last challenge: p, w now describe the folded polynomial
            uint64_t f[4];
            h_store(f, fold);
            CK(fold_pairs2(ctx, U(bp_[cur]), U(bp_[1 - cur]), U(bw_[cur]), U(bw_[1 - cur]), len, f));
            cur = 1 - cur;
            len /= 2;
        }
        return PK_OK;
    };
    // once a rank's block is short the blocks are all-gathered (block r of the gather IS index range r) and the rest of the
    // sumcheck runs replicated; a sumcheck_rounds call shrinks the block 2^k-fold, so blocks never run out inside one
    fe *gp[2] = {nullptr, nullptr}, *gw[2] = {nullptr, nullptr};
    if (sharded) {
        const size_t cap = G * SHARD_MIN_LOCAL;
        ALLOC(gp0, cap);
        ALLOC(gp1, cap / 2);
        ALLOC(gw0, cap);
        ALLOC(gw1, cap / 2);
        gp[0] = gp0; gp[1] = gp1; gw[0] = gw0; gw[1] = gw1;
    }
    auto maybe_gather = [&]() -> int {
        if (!sharded || len > SHARD_MIN_LOCAL) return PK_OK;
        CK(gather_blocks(ctx, bp_[cur], len, gp[0]));
        CK(gather_blocks(ctx, bw_[cur], len, gw[0]));
        bp_[0] = gp[0]; bp_[1] = gp[1]; bw_[0] = gw[0]; bw_[1] = gw[1];
        cur = 0;
        len *= G;
        sharded = false;
        return PK_OK;
    };
    std::vector<fe> rs;
    CK(sumcheck_rounds(k, rs));
    CK(maybe_gather());

    const fe* prev_leaves = C.leaves;
    const fe* prev_nodes = C.nodes;
    size_t prev_rows = C.rows, prev_width = C.width;
    pk_commit_layout prev_layout = C.layout;
    unsigned nv = n, log_inv_rate = cfg.starting_log_inv_rate;
    size_t domain_size = (size_t)1 << (n + log_inv_rate);
    // generator of the starting domain and its 2^k-th power (whir.go:99)
    fe exp_gen;
    {
        fe root28;
        const uint64_t l[4] = {0x9bd61b6e725b19f0ULL, 0x402d111e41112ed4ULL, 0x00e0a7eb8ef62abcULL, 0x2a3c09f0a58a7e85ULL};
        memcpy(root28.v, l, 32);
        fe gen = h_from_canon(root28);
        for (unsigned i = n + log_inv_rate; i < 28; i++) gen = h_mul(gen, gen);
        exp_gen = gen;
        for (unsigned i = 0; i < k; i++) exp_gen = h_mul(exp_gen, exp_gen);
    }
    for (unsigned r = 0; r < cfg.n_rounds; r++) {
        // W1: fold the coefficient form by this round's randomness
        const unsigned nv2 = nv - k;
        ALLOC(d_c2, (size_t)1 << nv2);
        CK(pk_fold_coeffs(ctx, U(d_c), nv, (const uint64_t*)rs.data(), k, U(d_c2)));
        d_c = d_c2;
        nv = nv2;
        log_inv_rate += k - 1;  // the domain halves while the polynomial shrinks 2^k-fold
        // N1+N2+M1+M2: re-commit
        const size_t rows = (size_t)1 << (nv + log_inv_rate - k), width = (size_t)1 << k;
        ALLOC(leaves, rows * width / shard_factor(ctx, rows));
        ALLOC(nodes, 2 * rows);
        CK(ensure_ws(ctx, commit_scratch_fes(ctx, rows, width) * 32));
        const uint64_t* ptr = U(d_c);
        pk_commit_layout layout;
        CK(commit_into(ctx, &ptr, 1, nv, log_inv_rate, k, U(leaves), U(nodes), (uint64_t*)ctx->d_ws, &layout));
        fe root;
        CK(read_root(ctx, U(nodes), rows, (uint64_t*)root.v));
        T.add_canon(root);
        // E1: OOD
        std::vector<fe> ood(cfg.ood_samples[r]);
        T.challenge_scalars(ood.data(), ood.size());
        std::vector<fe> ood_ans(ood.size());
        for (size_t j = 0; j < ood.size(); j++) CK(eval_univariate_x(ctx, d_c, (size_t)1 << nv, ood[j], ood_ans[j]));
        T.add_scalars(ood_ans.data(), ood_ans.size());
        // P1
        int prc = PK_OK;
        pow_round(ctx, T, cfg.pow_bits[r], &prc);
        CK(prc);
        // Q1: STIR queries into the previous tree
        std::vector<uint64_t> idx = stir_queries(T, domain_size, k, cfg.num_queries[r]);
        CK(emit_opening_hints(ctx, T, prev_leaves, prev_nodes, prev_rows, prev_width, prev_layout, idx));
        // W2: equality weights of the OOD and STIR points, scaled by powers of the combination randomness
        gamma = T.challenge_scalar();
        g = fe_one();
        const size_t q = ood.size() + idx.size();
        std::vector<fe> pts(q * (nv ? nv : 1)), scales(q ? q : 1);
        size_t j = 0;
        for (size_t t = 0; t < ood.size(); t++, j++) {
            expand_from_univariate(ood[t], nv, &pts[j * nv]);
            scales[j] = g;
            g = h_mul(g, gamma);
        }
        for (size_t t = 0; t < idx.size(); t++, j++) {
            expand_from_univariate(h_pow(exp_gen, idx[t]), nv, &pts[j * nv]);
            scales[j] = g;
            g = h_mul(g, gamma);
        }
        CK(eq_weights(bw_[cur], nv, pts, scales, q, 0));
        // W3
        CK(sumcheck_rounds(k, rs));
        CK(maybe_gather());
        prev_leaves = leaves;
        prev_nodes = nodes;
        prev_rows = rows;
        prev_width = width;
        prev_layout = layout;
        domain_size /= 2;
        exp_gen = h_mul(exp_gen, exp_gen);
    }
    // final round: the folded polynomial in the clear, PoW, final STIR openings, final sumcheck
    {
        if (sharded) {  // a schedule that ends before the blocks got short: finish replicated
            const size_t cap = len * G;
            ALLOC(fp0, cap);
            ALLOC(fp1, cap / 2 ? cap / 2 : 1);
            ALLOC(fw0, cap);
            ALLOC(fw1, cap / 2 ? cap / 2 : 1);
            CK(gather_blocks(ctx, bp_[cur], len, fp0));
            CK(gather_blocks(ctx, bw_[cur], len, fw0));
            bp_[0] = fp0; bp_[1] = fp1; bw_[0] = fw0; bw_[1] = fw1;
            cur = 0;
            len = cap;
            sharded = false;
        }
        const unsigned nv2 = nv - k;
        ALLOC(d_final, (size_t)1 << nv2);
        CK(pk_fold_coeffs(ctx, U(d_c), nv, (const uint64_t*)rs.data(), k, U(d_final)));
        nv = nv2;
        std::vector<fe> fin((size_t)1 << nv);
        CK(pk_memcpy_d2h(ctx, fin.data(), d_final, 32 * fin.size()));
        T.add_scalars(fin.data(), fin.size());
        int prc = PK_OK;
        pow_round(ctx, T, cfg.final_pow_bits, &prc);
        CK(prc);
        std::vector<uint64_t> idx = stir_queries(T, domain_size, k, cfg.final_queries);
        CK(emit_opening_hints(ctx, T, prev_leaves, prev_nodes, prev_rows, prev_width, prev_layout, idx));
        CK(sumcheck_rounds(nv, rs));
        prc = PK_OK;
        pow_round(ctx, T, cfg.final_folding_pow_bits, &prc);  // whir.go:196-201
        CK(prc);
    }
    // deferred_weight_evaluations hint (common.go:63-73): each linear weight's MLE at the full folding point.
    // Round t folds index bit t (LSB first), so the point in eval_eq's MSB-first order is reverse(all_r).
    if (n_weights) {
        std::vector<fe> point(all_r.rbegin(), all_r.rend());
        const bool sh = whir_sharded(ctx, n);  // the eq table and the dot products by blocks, like the weights themselves
        ALLOC(d_eq, B0);
        if (sh) {
            const fe sc = eq_bits(point.data(), lgG, rank);
            CK(pk_eq_accumulate(ctx, U(d_eq), n - lgG, (const uint64_t*)(point.data() + lgG), (const uint64_t*)&sc, 1, 1));
        } else {
            CK(pk_eq_table(ctx, (const uint64_t*)point.data(), n, U(d_eq)));
        }
        std::vector<uint8_t> buf;
        uint64_t cnt = n_weights;
        for (int i = 0; i < 8; i++) buf.push_back((uint8_t)(cnt >> (8 * i)));
        Across ac(ctx, sh);
        CK(ac.rc);
        uint64_t outs[4 * 8] = {};
        const bool rows3 = n_weights == 3 && weight_len[0] == weight_len[1] && weight_len[1] == weight_len[2] && d_weights[1] > d_weights[0] &&
                           d_weights[1] - d_weights[0] == d_weights[2] - d_weights[1];
        if (rows3) {  // the three external rows against the eq table in one pass
            const size_t hi = weight_len[0] < off0 + B0 ? weight_len[0] : off0 + B0;
            if (sh || weight_len[0])
                CK(dot_rows(ctx, U(d_weights[0] + off0), (size_t)(d_weights[1] - d_weights[0]), 3, U(d_eq), nullptr, hi > off0 ? hi - off0 : 0, outs));
        }
        for (unsigned i = 0; i < n_weights; i++) {
            uint64_t* out = outs + 4 * (i < 8 ? i : 7);
            const size_t hi = weight_len[i] < off0 + B0 ? weight_len[i] : off0 + B0;
            if (!rows3) {
                if (sh || weight_len[i]) CK(pk_dot(ctx, U(d_weights[i] + off0), U(d_eq), hi > off0 ? hi - off0 : 0, out));
                else memset(out, 0, 32);
            }
            fe c = h_to_canon(h_load(out));
            const uint8_t* b = (const uint8_t*)c.v;
            buf.insert(buf.end(), b, b + 32);
        }
        T.hint(buf.data(), buf.size());
    }
    return PK_OK;
}

// batch_commit_to_polynomial (provekit/prover/src/whir_r1cs.rs:182-209)
struct BatchCommit {
    Commitment com;
    fe* f_evals = nullptr;  // masked polynomial, evaluation form (2^m)
    fe* g_evals = nullptr;  // random polynomial, evaluation form (2^m)
};
// masks, coefficient forms and the commitment's device work on `ctx`'s stream; nothing is read back and the transcript is not touched
int batch_commit_compute(pk_ctx* ctx, Arena& A, unsigned m, const pk_whir_config& cfg, const fe* d_evals, size_t n_evals, const RngKey& key,
                         u32 stream_mask, u32 stream_g, BatchCommit& out) {
    const size_t half = (size_t)1 << (m - 1), N = 2 * half;
    ALLOC(f, N);
    ALLOC(g, N);
    ALLOC(fe_, N);
    ALLOC(ge_, N);
    // f = [witness (zero padded) || mask]   (zk_utils.rs:3-22)
    CK(pk_memset_zero(ctx, f, 32 * half));
    CK(pk_memcpy_d2d(ctx, f, d_evals, 32 * n_evals));
    {
        ProfScope prof(ctx, "random_fe");
        random_fe_kernel<<<random_fe_grid(ctx, half), 256, 0, ctx->stream>>>(f + half, half, key, stream_mask);
        random_fe_kernel<<<random_fe_grid(ctx, N), 256, 0, ctx->stream>>>(g, N, key, stream_g);
    }
    PK_LAUNCH_CHECK(ctx);
    // f, g hold the evaluation forms (kept for the weighted sums); the coefficient forms go to fc, gc
    CK(pk_to_coeffs_into(ctx, U(f), U(fe_), m));
    CK(pk_to_coeffs_into(ctx, U(g), U(ge_), m));
    out.f_evals = f;
    out.g_evals = g;
    fe* polys[2] = {fe_, ge_};
    int rc = whir_commit_compute(ctx, A, cfg, polys, 2, out.com);
    out.com.evals[0] = f;
    out.com.evals[1] = g;
    return rc;
}
int batch_commit(pk_ctx* ctx, Arena& A, unsigned m, const pk_whir_config& cfg, const fe* d_evals, size_t n_evals, const RngKey& key,
                 u32 stream_mask, u32 stream_g, Transcript& T, BatchCommit& out) {
    CK(batch_commit_compute(ctx, A, m, cfg, d_evals, n_evals, key, stream_mask, stream_g, out));
    fe root;
    CK(read_root(ctx, U(out.com.nodes), out.com.rows, (uint64_t*)root.v));
    return whir_commit_transcript(ctx, cfg, root, T, out.com);
}

// 256-bit key of one proof's random draws: fresh from the OS CSPRNG (the reference's thread_rng) unless injected.  One proof
// sharded over a device set: every rank must mask with the SAME polynomials -- rank 0's key goes to everybody.
int proof_key(pk_ctx* ctx, const uint8_t* rng_seed32, RngKey& key) {
    if (rng_seed32) {
        memcpy(key.k, rng_seed32, 32);
        return PK_OK;
    }
    size_t got = 0;
    while (got < 32) {
        ssize_t r = getrandom((char*)key.k + got, 32 - got, 0);
        if (r < 0) {
            if (errno == EINTR) continue;
            return set_err(ctx, PK_ERR_HIP, "getrandom failed: %s", strerror(errno));
        }
        got += (size_t)r;
    }
    if (comm_world(ctx) > 1) {
        int rc = ensure_scratch(ctx, ((size_t)1 << 19) + 32 * (size_t)(PK_MAX_RANKS + 1));
        if (rc) return rc;
        char* d_key = (char*)ctx->d_scratch + ((size_t)1 << 19);  // clear of the reduction area (head) and the PoW words (tail)
        PK_HIP(ctx, hipMemcpyAsync(d_key, key.k, 32, hipMemcpyHostToDevice, ctx->stream));
        rc = comm_all_gather(ctx, d_key, d_key + 32, 32);
        if (rc) return rc;
        PK_HIP(ctx, hipMemcpyAsync(key.k, d_key + 32, 32, hipMemcpyDeviceToHost, ctx->stream));
        PK_WAIT(ctx);
    }
    return PK_OK;
}

// fill_witness (provekit/prover/src/witness/mod.rs:15-30): every None entry takes FieldElement::from(rng.random::<u128>()).
// Entry i reads the (i mod 4)-th 128-bit word of ChaCha block i / 4 of stream RNG_FILL, so the result does not depend on the
// launch shape.  *n_filled counts them (the reference logs the count).
__global__ __launch_bounds__(256) void fill_witness_kernel(fe* __restrict__ w, const uint8_t* __restrict__ is_set, size_t n, RngKey key, u32 stream,
                                                           unsigned long long* n_filled) {
    PK_LATENCY_PRIO();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned mine = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (is_set[i]) continue;
        u32 blk[16];
        chacha_block(key, (u64)(i >> 2), stream, 0, PK_RNG_ROUNDS, blk);
        fe x = fe_zero();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u32 v = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) v = (i & 3) == (size_t)q ? blk[4 * q + k] : v;
            x.v[k] = v;
        }
        fe_store(w + i, fe_to_montx(x));
        mine++;
    }
    if (mine) atomicAdd(n_filled, (unsigned long long)mine);  // n_filled is DEVICE memory: agent-scope atomics are exact there
}

// public inputs of the witness transcript: the ACIR witness values at the circuit's public indices
__global__ void gather_fe_kernel(const fe* __restrict__ src, const uint32_t* __restrict__ idx, size_t n, fe* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fe_store(dst + i, fe_load(src + idx[i]));
}

// create_witness_io_pattern (provekit/prover/src/noir_proof_scheme.rs:94-109) with witness_io_pattern.rs:18-41: the spongefish
// op list "<domain>\0A2shape[\0A<n>pub_inputs][\0S<n>wb:challenges]"
std::string witness_io_pattern(size_t n_public, size_t n_challenges) {
    std::string d = "\xF0\x9F\x93\x9C";  // "📜"
    d.push_back('\0');
    d += "A2shape";
    if (n_public) {
        d.push_back('\0');
        d += "A" + std::to_string(n_public) + "pub_inputs";
    }
    if (n_challenges) {
        d.push_back('\0');
        d += "S" + std::to_string(n_challenges) + "wb:challenges";
    }
    return d;
}

// WhirR1CSScheme::create_io_pattern (provekit/common/src/whir_r1cs.rs:28-39) restated:
//   IOPattern::new("🌪️").commit_statement(w).add_rand(m_0).commit_statement(b).add_zk_sumcheck_polynomials(m_0)
//            .add_whir_proof(b).hint("claimed_evaluations").add_whir_proof(w)
// provekit's own labels (utils/sumcheck.rs:119-142) are in the tree.  commit_statement / add_whir_proof live in whir @3e7f8c2
// (absent): their OPERATIONS are pinned by the in-tree Go verifier's read order (mtUtilities.go:51-76, whir.go:51-220) and the
// labels "stir_answers", "merkle_proof", "deferred_weight_evaluations", "pow-nonce" by its pattern walker (common.go:41-100);
// the remaining labels are whir's / spongefish-pow's as published (merkle_digest, ood_query, ood_ans, sumcheck_poly,
// folding_randomness, combination_randomness, pow_queries, stir_queries, final_coeffs, final_queries) -- UNPINNED here, which is
// why a caller that holds the reference's bytes overrides this string (pk_scheme_set_io_pattern).  Zero-count operations are
// omitted exactly where whir guards them (no OOD samples, no grinding).
std::string whir_r1cs_io_pattern(unsigned m_0, const pk_whir_config& w, const pk_whir_config& h) {
    std::string d = "\xF0\x9F\x8C\xAA\xEF\xB8\x8F";  // "🌪️"
    auto op = [&](char kind, size_t count, const char* label) {
        d.push_back('\0');
        d.push_back(kind);
        if (kind == 'A' || kind == 'S') d += std::to_string(count);
        d += label;
    };
    auto A = [&](size_t n, const char* l) { if (n) op('A', n, l); };
    auto S = [&](size_t n, const char* l) { if (n) op('S', n, l); };
    auto challenge_bytes = [&](size_t n, const char* l) { S((n + 14) / 15, l); };  // 15 uniform bytes per squeezed element
    auto pow = [&](double bits) {  // spongefish-pow challenge_pow: 32 challenge bytes, 8-byte nonce
        if (bits > 0.0) {
            challenge_bytes(32, "pow_queries");
            A(8, "pow-nonce");
        }
    };
    auto add_ood = [&](size_t samples, size_t batch) {
        S(samples, "ood_query");
        A(samples * batch, "ood_ans");
    };
    auto add_sumcheck = [&](unsigned rounds) {
        for (unsigned i = 0; i < rounds; i++) {
            A(3, "sumcheck_poly");
            S(1, "folding_randomness");
        }
    };
    auto commit_statement = [&](const pk_whir_config& c) {
        A(1, "merkle_digest");
        add_ood(c.commitment_ood_samples, c.batch_size);
        if (c.batch_size > 1) S(1, "batching_randomness");  // drawn right after the commitment (mtUtilities.go:71-75)
    };
    auto query_bytes = [](size_t domain, unsigned fold) {
        const size_t folded = domain >> fold;
        return (size_t)((ilog2(folded) + 7) / 8);
    };
    auto add_whir_proof = [&](const pk_whir_config& c) {
        const unsigned k = c.folding_factor;
        S(1, "initial_combination_randomness");
        add_sumcheck(k);
        size_t domain = (size_t)1 << (c.n_vars + c.starting_log_inv_rate);
        for (unsigned r = 0; r < c.n_rounds; r++) {
            A(1, "merkle_digest");
            add_ood(c.ood_samples[r], 1);
            pow(c.pow_bits[r]);
            challenge_bytes((size_t)c.num_queries[r] * query_bytes(domain, k), "stir_queries");
            op('H', 0, "stir_answers");
            op('H', 0, "merkle_proof");
            S(1, "combination_randomness");
            add_sumcheck(k);
            domain >>= 1;
        }
        const unsigned final_vars = c.n_vars - k * (c.n_rounds + 1);
        A((size_t)1 << final_vars, "final_coeffs");
        pow(c.final_pow_bits);
        challenge_bytes((size_t)c.final_queries * query_bytes(domain, k), "final_queries");
        op('H', 0, "stir_answers");
        op('H', 0, "merkle_proof");
        add_sumcheck(final_vars);
        pow(c.final_folding_pow_bits);  // once, after the last round (whir.go:196-201)
        op('H', 0, "deferred_weight_evaluations");
    };
    commit_statement(w);
    S(m_0, "rand");
    commit_statement(h);
    A(1, "Sum of G over boolean hypercube");
    S(1, "Rho");
    for (unsigned i = 0; i < m_0; i++) {
        A(4, "Sumcheck Polynomials");
        S(1, "Sumcheck Random");
    }
    A(2, "Polynomial sums");
    add_whir_proof(h);
    op('H', 0, "claimed_evaluations");
    add_whir_proof(w);
    return d;
}

// do the caller's IO-pattern bytes declare the operations pk_prove performs for (m_0, w, h)?  "" = yes, else the first difference
std::string io_pattern_mismatch(const std::string& theirs, unsigned m_0, const pk_whir_config& w, const pk_whir_config& h) {
    std::vector<IoOp> a, b;
    std::string err;
    if (!io_pattern_parse(theirs, a, err)) return err;
    if (!io_pattern_parse(whir_r1cs_io_pattern(m_0, w, h), b, err)) return "internal: " + err;
    auto name = [](const IoOp& o) { return std::string(1, o.kind) + (o.kind == 'A' || o.kind == 'S' ? std::to_string(o.count) : std::string()); };
    for (size_t i = 0; i < a.size() && i < b.size(); i++)
        if (a[i].kind != b[i].kind || a[i].count != b[i].count)
            return "IO pattern operation #" + std::to_string(i + 1) + " (after merging) is " + name(a[i]) + " but this scheme's prover performs " + name(b[i]);
    if (a.size() != b.size())
        return "IO pattern declares " + std::to_string(a.size()) + " operations (after merging), this scheme's prover performs " + std::to_string(b.size());
    return "";
}

// One proof over a device set: a rank that leaves early (arena exhausted, a HIP error, an unsatisfied witness on this rank only)
// never reaches the collectives its peers are -- or will be -- waiting in.  Whatever the exit path, a failing rank aborts the
// group (in-process transport: the waiting ranks wake with PK_ERR_RCCL; host transport: this rank's communicator is marked
// failed; RCCL: this rank's OWN communicator is aborted -- its peers are rescued by their collective deadline, comm.hip comm_wait).
// One exception, RCCL only, where an abort is irreversible: a refusal that every rank of the set makes identically (a bad argument,
// an unsatisfied witness, an IO-pattern mismatch) BEFORE this call has enqueued any collective leaves the ranks in step, so the
// communicator stays usable for the next call.
struct AbortOnFailure {
    pk_ctx* c;
    bool ok = false;
    unsigned long long issued0;
    explicit AbortOnFailure(pk_ctx* ctx) : c(ctx), issued0(comm_collectives_issued(ctx)) {}
    ~AbortOnFailure() {
        if (ok) return;
        struct Drain {  // whatever happens to the communicator, an abandoned proof leaves nothing behind on the stream: a gated kernel
            pk_ctx* c;  // released with a zero challenge must have finished -- and its give-up word be cleared -- before the next proof starts
            ~Drain() {
                (void)wait_ctx(c);
                sumcheck_gate_clear(c);
            }
        } drain{c};
        if (comm_world(c) <= 1) return;
        const bool same_everywhere = c->err_code == PK_ERR_BAD_ARG || c->err_code == PK_ERR_UNSATISFIED || c->err_code == PK_ERR_IO_PATTERN;
        if (comm_rccl(c) && same_everywhere && comm_collectives_issued(c) == issued0) return;
        comm_abort(c);
    }
};

}  // namespace

extern "C" {

int pk_scheme_destroy(pk_ctx* ctx, pk_scheme* s) {
    PK_ENTER(ctx);
    if (!s) return PK_OK;
    (void)wait_ctx(ctx);
    (void)hipFree(s->arena);
    (void)hipFree(s->noir_witness);
    if (s->side) (void)pk_ctx_destroy(s->side);
    delete s;
    return PK_OK;
}

// arena = the sum of pk_prove's allocations (nothing is freed inside a proof).  With N = 2^m, R = 2^starting_log_inv_rate,
// F = 2^folding_factor: f, g in both forms 4N; initial codeword batch*R*N and its tree 2R/F N; working polynomial and the
// sumcheck ping-pong 4N; round codewords (domain halves each round) < R N, their trees < 2R/F N, folded polynomials
// < 2/F N; deferred eq table N; a, b, c, eq and the second eq table 5*2^m_0; external rows 3*num_witnesses.  The small
// blinding scheme (2^(nb+1) <= 2^9 elements) and alignment are covered by the constant.
static size_t scheme_arena_bytes(unsigned m, unsigned m_0, size_t num_witnesses, const pk_whir_config& w) {
    const double N = (double)((size_t)1 << m), R = (double)((size_t)1 << w.starting_log_inv_rate), F = (double)((size_t)1 << w.folding_factor);
    const double units = 4.0 + w.batch_size * R + 2.0 * R / F + 4.0 + R + 2.0 * R / F + 2.0 / F + 1.0;
    const double fes = units * N + 5.0 * (double)((size_t)1 << m_0) + 3.0 * (double)num_witnesses;
    return (size_t)(1.05 * 32.0 * fes) + ((size_t)64 << 20);
}
int pk_scheme_arena_bytes(unsigned m, unsigned m_0, size_t num_witnesses, const pk_whir_config* whir_witness, size_t* bytes) {
    if (!whir_witness || !bytes || m > 28 || m_0 > m) return PK_ERR_BAD_ARG;
    *bytes = scheme_arena_bytes(m, m_0, num_witnesses, *whir_witness);
    return PK_OK;
}

int pk_scheme_create(pk_ctx* ctx, const pk_r1cs* r1cs, size_t num_constraints, size_t num_witnesses, unsigned m, unsigned m_0,
                     const pk_whir_config* whir_witness, const pk_whir_config* whir_for_hiding_spartan, pk_scheme** out) {
    if (!ctx || !out) return PK_ERR_BAD_ARG;
    PK_ENTER(ctx);
    *out = nullptr;
    PK_REQUIRE(ctx, r1cs && whir_witness && whir_for_hiding_spartan, "null pointer");
    PK_REQUIRE(ctx, m >= 1 && m <= 27 && m_0 >= 1 && m_0 <= 27, "scheme size out of range (1 <= m, m_0 <= 27)");
    // ensure!(...) of provekit/prover/src/whir_r1cs.rs:43-54
    PK_REQUIRE(ctx, num_witnesses <= ((size_t)1 << (m - 1)), "R1CS witness length exceeds scheme capacity");
    PK_REQUIRE(ctx, num_constraints <= ((size_t)1 << m_0), "R1CS constraints exceed scheme capacity");
    PK_REQUIRE(ctx, whir_witness->n_vars == m && whir_witness->batch_size == 2, "whir_witness config does not match m / batch 2");
    // pk_prove commits the blinding polynomial next to its mask exactly as the witness (whir_r1cs.rs:212-226): batch 2, nothing else
    PK_REQUIRE(ctx, whir_for_hiding_spartan->batch_size == 2, "whir_for_hiding_spartan must have batch_size 2");
    for (const pk_whir_config* c : {whir_witness, whir_for_hiding_spartan}) {
        PK_REQUIRE(ctx, c->folding_factor >= 1 && c->folding_factor <= 8 && c->n_rounds <= PK_MAX_WHIR_ROUNDS, "bad WHIR config");
        PK_REQUIRE(ctx, c->n_vars >= c->folding_factor * (c->n_rounds + 1), "WHIR rounds exceed the number of variables");
        PK_REQUIRE(ctx, c->commitment_ood_samples <= 4, "too many OOD samples");
        PK_REQUIRE(ctx, c->batch_size >= 1 && c->batch_size <= 4, "batch size out of range");
        // the evaluation domain must exist in BN254-Fr (two-adicity 28) and the codeword must fit pk_rs_encode's bound
        PK_REQUIRE(ctx, c->starting_log_inv_rate >= 1 && c->n_vars + c->starting_log_inv_rate <= 28, "n_vars + starting_log_inv_rate exceeds 28");
        PK_REQUIRE(ctx, c->n_vars + c->starting_log_inv_rate - c->folding_factor <= 27, "codeword has more than 2^27 rows");
        for (unsigned r = 0; r < c->n_rounds; r++) PK_REQUIRE(ctx, c->ood_samples[r] <= 4, "too many OOD samples");
    }
    unsigned nb = 0;
    while (((size_t)1 << nb) < 4 * (size_t)m_0) nb++;
    PK_REQUIRE(ctx, whir_for_hiding_spartan->n_vars == nb + 1, "whir_for_hiding_spartan must have next_power_of_two(4*m_0)+1 variables");
    pk_scheme* s = new (std::nothrow) pk_scheme();
    if (!s) return PK_ERR_OOM;
    s->r1cs = r1cs;
    s->num_constraints = num_constraints;
    s->num_witnesses = num_witnesses;
    s->m = m;
    s->m_0 = m_0;
    s->whir_witness = *whir_witness;
    s->whir_hiding = *whir_for_hiding_spartan;
    s->domain_separator = whir_r1cs_io_pattern(s->m_0, s->whir_witness, s->whir_hiding);
    s->arena_bytes = scheme_arena_bytes(m, m_0, num_witnesses, s->whir_witness);
    if (hipMalloc((void**)&s->arena, s->arena_bytes) != hipSuccess) {
        delete s;
        return set_err(ctx, PK_ERR_OOM, "hipMalloc of the %zu MiB prover arena failed", s->arena_bytes >> 20);
    }
    *out = s;
    return PK_OK;
}

int pk_prove(pk_ctx* ctx, pk_scheme* s, const uint64_t* d_witness, size_t n_witness, const uint8_t* rng_seed32, uint8_t* transcript_out,
             size_t cap, size_t* len) {
    PK_ENTER(ctx);
    AbortOnFailure guard(ctx);  // before the argument checks: a rank refused here never joins its peers' collectives either
    PK_REQUIRE(ctx, s && d_witness && len, "null pointer");
    PK_REQUIRE(ctx, n_witness == s->num_witnesses, "Unexpected witness length for R1CS instance");  // whir_r1cs.rs:43-46
    RngKey key;
    {
        int rc = proof_key(ctx, rng_seed32, key);
        if (rc) return rc;
    }
    struct Turn {  // see comm.hip: a no-op outside the test-suite's one-GPU timing mode
        pk_ctx* c;
        explicit Turn(pk_ctx* ctx) : c(ctx) { comm_turn_begin(c); }
        ~Turn() { comm_turn_end(c); }
    } turn(ctx);
    Arena A{s->arena, s->arena_bytes};
    Transcript T(s->domain_separator);
    (void)sumcheck_gate_check(ctx);  // a word left by an earlier, abandoned proof is not this proof's
    const bool timing = getenv("PK_PROVE_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_start = now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        (void)wait_ctx(ctx);
        auto t = now();
        fprintf(stderr, "[pk_prove] %-28s %8.3f ms (sponge: %u permutes, %.3f ms; hint serialisation so far %.3f ms)\n", what,
                1e3 * std::chrono::duration<double>(t - t_start).count(), T.permutes, 1e3 * T.permute_seconds, 1e3 * T.hint_seconds);
        t_start = t;
    };
    const unsigned m = s->m, m_0 = s->m_0;

    // blinding univariates: 4 random coefficients per variable [RNG], committed with the small WHIR (whir_r1cs.rs:212-226)
    unsigned nb = 0;
    while (((size_t)1 << nb) < 4 * (size_t)m_0) nb++;
    const size_t NB = (size_t)1 << nb;
    BatchCommit B;
    fe* d_blind = nullptr;
    fe* side_univ = nullptr;  // pinned: the 4 m_0 blinding coefficients as the side stream copied them out
    // Latency mode, one GPU: the blinding commitment depends on nothing but the proof's key -- 0.6 ms of launches that keep 32 lanes
    // busy (its two leaf hashes are chains of 31 compressions).  Its device work goes to a second stream NOW and runs underneath the
    // witness commitment, which fills the chip for 2.6 ms; only the transcript half (root, OOD, batching) waits for its turn.
    const bool overlap_blinding = ctx->latency_mode && comm_world(ctx) == 1 && !getenv("PK_NO_BLINDING_OVERLAP");  // (the env switch: A/B only)
    struct SideDrain {  // whatever path leaves pk_prove, the side stream must not still be working in this proof's arena
        pk_scheme* s;
        bool used = false;
        ~SideDrain() {
            if (used && s->side) {
                (void)wait_stream(s->side->device, s->side->stream);
                s->side->mail_off = 0;
            }
        }
    } side_drain{s};
    // --- commit to the masked witness polynomial (whir_r1cs.rs:57-69)
    BatchCommit W;
    if (overlap_blinding) {
        side_drain.used = true;
        // the witness commitment's launches first (the chip starts on them at once), then the side stream's, then the witness root
        CK(batch_commit_compute(ctx, A, m, s->whir_witness, (const fe*)d_witness, n_witness, key, RNG_MASK, RNG_G, W));
        if (!s->side) {
            int dev = 0;
            PK_HIP(ctx, hipGetDevice(&dev));
            CK(pk_ctx_create(dev, &s->side));
        }
        pk_ctx* sc = s->side;
        sc->hash_version = ctx->hash_version;
        d_blind = A.alloc(NB);
        if (!d_blind) return set_err(ctx, PK_ERR_OOM, "prover arena exhausted (d_blind)");
        int rc = pk_memset_zero(sc, d_blind, 32 * NB);
        if (!rc) {
            random_fe_kernel<<<1, 256, 0, sc->stream>>>(d_blind, 4 * (size_t)m_0, key, RNG_BLIND);
            rc = mail_alloc(sc, 32 * 4 * (size_t)m_0, (void**)&side_univ);
        }
        if (!rc && hipMemcpyAsync(side_univ, d_blind, 32 * 4 * (size_t)m_0, hipMemcpyDeviceToHost, sc->stream) != hipSuccess) rc = PK_ERR_HIP;
        if (!rc) rc = batch_commit_compute(sc, A, nb + 1, s->whir_hiding, d_blind, NB, key, RNG_MASK_B, RNG_G_B, B);
        if (rc) return set_err(ctx, rc, "blinding commitment on the side stream: %s", pk_last_error(sc));
        fe root_w;
        CK(read_root(ctx, U(W.com.nodes), W.com.rows, (uint64_t*)root_w.v));
        CK(whir_commit_transcript(ctx, s->whir_witness, root_w, T, W.com));
    } else {
        CK(batch_commit(ctx, A, m, s->whir_witness, (const fe*)d_witness, n_witness, key, RNG_MASK, RNG_G, T, W));
    }

    lap("witness commit");
    // --- run_zk_sumcheck_prover (whir_r1cs.rs:228-369)
    std::vector<fe> r(m_0);
    T.challenge_scalars(r.data(), m_0);
    const size_t M0 = (size_t)1 << m_0;
    // On a device set the four sumcheck arrays are split by the LOW index bits: rank g holds the entries i = g (mod G) at local
    // index i / G (see "one proof over a device set" above).  eq(r, i) factors into eq over the high variables (the local
    // table) times eq(last lgG variables, bits of g), a scalar that goes in as the table's scale.
    const unsigned G = (unsigned)comm_world(ctx), lgG = ilog2(G), rank = (unsigned)comm_rank(ctx);
    bool zk_sharded = G > 1 && M0 / G >= 2 * SHARD_MIN_LOCAL;
    const size_t Lz = zk_sharded ? M0 / G : M0;
    ALLOC(d_a, Lz);
    ALLOC(d_b, Lz);
    ALLOC(d_cc, Lz);
    ALLOC(d_eq, Lz);
    if (zk_sharded) {
        CK(witness_bounds_strided(ctx, s->r1cs, d_witness, m_0, G, rank, U(d_a), U(d_b), U(d_cc)));  // S1, this rank's rows
        const fe sc = eq_bits(&r[m_0 - lgG], lgG, rank);
        CK(pk_eq_accumulate(ctx, U(d_eq), m_0 - lgG, (const uint64_t*)r.data(), (const uint64_t*)&sc, 1, 1));  // S2
    } else {
        CK(pk_r1cs_witness_bounds(ctx, s->r1cs, d_witness, m_0, U(d_a), U(d_b), U(d_cc)));  // S1
        CK(pk_eq_table(ctx, (const uint64_t*)r.data(), m_0, U(d_eq)));                       // S2
    }
    std::vector<fe> g_univ(4 * (size_t)m_0);
    if (overlap_blinding) {  // the side stream finished long ago: take its root and the coefficients, then the transcript half here
        pk_ctx* sc = s->side;
        if (wait_stream(sc->device, sc->stream) != hipSuccess) return set_err(ctx, PK_ERR_HIP, "side stream synchronisation failed");
        memcpy(g_univ.data(), side_univ, 32 * g_univ.size());
        sc->mail_off = 0;
        fe root_b;
        if (B.com.rows < 2) CK(pk_memcpy_d2h(ctx, root_b.v, B.com.nodes + 1, 32));
        else memcpy(root_b.v, (char*)sc->h_pinned + PK_PIN_ROOT, 32);
        CK(whir_commit_transcript(ctx, s->whir_hiding, root_b, T, B.com));
    } else {
        ALLOC(d_blind_, NB);
        d_blind = d_blind_;
        CK(pk_memset_zero(ctx, d_blind, 32 * NB));
        random_fe_kernel<<<1, 256, 0, ctx->stream>>>(d_blind, 4 * (size_t)m_0, key, RNG_BLIND);
        PK_LAUNCH_CHECK(ctx);
        CK(pk_memcpy_d2h(ctx, g_univ.data(), d_blind, 32 * g_univ.size()));
        CK(batch_commit(ctx, A, nb + 1, s->whir_hiding, d_blind, NB, key, RNG_MASK_B, RNG_G_B, T, B));
    }
    lap("bounds+eq+blinding commit");
    // sum_over_hypercube (whir_r1cs.rs:172-180)
    fe sum_g;
    {
        fe c[4];
        blinding_coefficients_for_round(g_univ, 0, nullptr, c);
        sum_g = h_add(eval_cubic(c, fe_zero()), eval_cubic(c, fe_one()));
    }
    T.add_scalar(sum_g);
    const fe rho = T.challenge_scalar();
    fe saved = h_mul(rho, sum_g);
    std::vector<fe> alpha;
    alpha.reserve(m_0);
    {
        size_t length = Lz;  // local length while sharded
        const fe half = h_half();
        fe *za = d_a, *zb = d_b, *zc = d_cc, *ze = d_eq;
        fe* zfull[4] = {nullptr, nullptr, nullptr, nullptr};
        fe* ztmp = nullptr;
        if (zk_sharded) {
            const size_t cap = G * SHARD_MIN_LOCAL;
            for (int q = 0; q < 4; q++) {
                ALLOC(zf, cap);
                zfull[q] = zf;
            }
            ALLOC(zt, cap);
            ztmp = zt;
        }
        const bool pipelined = ctx->latency_mode && G == 1 && m_0 >= 2;
        unsigned red_cur = 0, red_next = 0;
        if (pipelined) CK(sumcheck_cubic_launch(ctx, U(za), U(zb), U(zc), U(ze), length, nullptr, 0, &red_cur));
        double t_launch = 0, t_wait = 0, t_host = 0;  // PK_PROVE_TIMING: where a pipelined round's wall time goes
        for (unsigned idx = 0; pipelined && idx < m_0; idx++) {  // latency mode: round idx+1 is in the queue, gated, while round idx is absorbed
            PendingGate gate(ctx);
            auto q0 = now();
            if (idx + 1 < m_0) {
                gate.arm(sumcheck_gate_next(ctx));
                CK(sumcheck_cubic_launch(ctx, U(za), U(zb), U(zc), U(ze), length, nullptr, gate.seq, &red_next));
                length /= 2;
            }
            // the round's blinding coefficients depend only on the earlier challenges: computed while the kernel runs
            fe gp[4];
            blinding_coefficients_for_round(g_univ, idx, alpha.data(), gp);
            const fe g_m1 = h_sub(h_add(h_sub(gp[0], gp[1]), gp[2]), gp[3]);
            const fe rg0 = h_mul(rho, gp[0]), rgm1 = h_mul(rho, g_m1), rg3 = h_mul(rho, gp[3]);
            auto q1 = now();
            uint64_t out[12];
            CK(sumcheck_collect_spin(ctx, red_cur, out));
            auto q2 = now();
            t_launch += std::chrono::duration<double>(q1 - q0).count();
            t_wait += std::chrono::duration<double>(q2 - q1).count();
            const fe h0 = h_load(out), hm1 = h_load(out + 4), hinf = h_load(out + 8);
            fe c[4];
            c[0] = h_add(h0, rg0);
            const fe at_m1 = h_add(hm1, rgm1);
            c[2] = h_mul(half, h_sub(h_sub(h_sub(h_add(saved, at_m1), c[0]), c[0]), c[0]));
            c[3] = h_add(hinf, rg3);
            c[1] = h_sub(h_sub(h_sub(h_sub(saved, c[0]), c[0]), c[3]), c[2]);
            T.add_scalars(c, 4);
            const fe a_i = T.challenge_scalar();
            gate.publish(a_i);
            alpha.push_back(a_i);
            saved = eval_cubic(c, a_i);
            red_cur = red_next;
            t_host += std::chrono::duration<double>(now() - q2).count();
        }
        if (timing && pipelined)
            fprintf(stderr, "[pk_prove]   pipelined cubic rounds: launch %.3f ms, wait %.3f ms, host %.3f ms\n", 1e3 * t_launch, 1e3 * t_wait, 1e3 * t_host);
        for (unsigned idx = 0; !pipelined && idx < m_0; idx++) {  // the hot loop, whir_r1cs.rs:280-345
            uint64_t out[12], f[4];
            if (zk_sharded && length <= SHARD_MIN_LOCAL) {  // short shares: gather, re-interleave, finish replicated
                fe* loc[4] = {za, zb, zc, ze};
                for (int q = 0; q < 4; q++) CK(gather_strided(ctx, loc[q], length, ztmp, zfull[q]));
                za = zfull[0]; zb = zfull[1]; zc = zfull[2]; ze = zfull[3];
                length *= G;
                zk_sharded = false;
            }
            Across ac(ctx, zk_sharded);  // sharded: the three evaluations are sums over the ranks' shares
            CK(ac.rc);
            if (idx == 0) {
                CK(pk_sumcheck_cubic_round(ctx, U(za), U(zb), U(zc), U(ze), length, nullptr, out));
            } else {
                h_store(f, alpha.back());
                CK(pk_sumcheck_cubic_round(ctx, U(za), U(zb), U(zc), U(ze), length, f, out));
                length /= 2;
            }
            const fe h0 = h_load(out), hm1 = h_load(out + 4), hinf = h_load(out + 8);
            fe gp[4];
            blinding_coefficients_for_round(g_univ, idx, alpha.data(), gp);
            fe c[4];
            c[0] = h_add(h0, h_mul(rho, gp[0]));
            const fe g_m1 = h_sub(h_add(h_sub(gp[0], gp[1]), gp[2]), gp[3]);
            const fe at_m1 = h_add(hm1, h_mul(rho, g_m1));
            c[2] = h_mul(half, h_sub(h_sub(h_sub(h_add(saved, at_m1), c[0]), c[0]), c[0]));
            c[3] = h_add(hinf, h_mul(rho, gp[3]));
            c[1] = h_sub(h_sub(h_sub(h_sub(saved, c[0]), c[0]), c[3]), c[2]);
            T.add_scalars(c, 4);
            const fe a_i = T.challenge_scalar();
            alpha.push_back(a_i);
            saved = eval_cubic(c, a_i);
        }
    }
    lap("zk sumcheck rounds");
    // latency mode: the witness statement's external rows and its six weighted sums depend on alpha and nothing else, and the transcript
    // wants them only AFTER the blinding WHIR proof -- 0.4 ms of kernels that go to the side stream now and run underneath that proof
    fe* d_eq_alpha_side = nullptr;
    fe* d_rows_side = nullptr;
    const bool overlap_rows = overlap_blinding && n_witness > 0;
    if (overlap_rows) {
        pk_ctx* sc = s->side;
        d_eq_alpha_side = A.alloc((size_t)1 << m_0);
        d_rows_side = A.alloc(3 * n_witness);
        if (!d_eq_alpha_side || !d_rows_side) return set_err(ctx, PK_ERR_OOM, "prover arena exhausted (external rows)");
        int rc = pk_eq_table(sc, (const uint64_t*)alpha.data(), m_0, U(d_eq_alpha_side));
        if (!rc) rc = pk_r1cs_external_row(sc, s->r1cs, U(d_eq_alpha_side), U(d_rows_side));
        if (!rc) rc = dot_rows_x(sc, U(d_rows_side), n_witness, 3, U(W.f_evals), U(W.g_evals), n_witness, nullptr, /*defer=*/true);
        if (rc) return set_err(ctx, rc, "external rows on the side stream: %s", pk_last_error(sc));
    }
    // statement over the blinding commitment: weight = expand_powers(alpha) zero-extended (whir_r1cs.rs:347-366,371-380)
    {
        const size_t NB2 = 2 * NB;
        std::vector<fe> wv(NB2, fe_zero());
        for (unsigned i = 0; i < m_0; i++) {
            wv[4 * i] = fe_one();
            wv[4 * i + 1] = alpha[i];
            wv[4 * i + 2] = h_mul(alpha[i], alpha[i]);
            wv[4 * i + 3] = h_mul(wv[4 * i + 2], alpha[i]);
        }
        const size_t nbw = 4 * (size_t)m_0;  // the weight is zero beyond the 4 m_0 blinding coefficients
        ALLOC(d_bw, nbw);
        CK(pk_memcpy_h2d(ctx, d_bw, wv.data(), 32 * nbw));
        uint64_t fg[8];
        CK(pk_dot2(ctx, U(d_bw), U(B.f_evals), U(B.g_evals), nbw, fg));
        fe sums[2] = {h_load(fg), h_load(fg + 4)};
        T.add_scalars(sums, 2);
        fe* wts[1] = {d_bw};
        CK(whir_prove(ctx, A, s->whir_hiding, B.com, wts, &nbw, 1, T));
    }
    lap("blinding WHIR proof");
    // --- external rows and the statement over the witness commitment (whir_r1cs.rs:81-91, 382-412)
    // sharded witness WHIR: a rank needs (and computes) only the columns of the rows inside its block of the hypercube
    const bool st_sharded = whir_sharded(ctx, m);
    const size_t blk = st_sharded ? ((size_t)1 << m) / G : (size_t)1 << m, blk_lo = st_sharded ? (size_t)rank * blk : 0;
    const size_t col_hi = n_witness < blk_lo + blk ? n_witness : blk_lo + blk, col_n = col_hi > blk_lo ? col_hi - blk_lo : 0;
    fe* d_rows = d_rows_side;
    if (!overlap_rows) {
        ALLOC(d_eq_alpha, M0);
        CK(pk_eq_table(ctx, (const uint64_t*)alpha.data(), m_0, U(d_eq_alpha)));
        ALLOC(d_rows_, 3 * (n_witness ? n_witness : 1));
        d_rows = d_rows_;
        if (st_sharded) CK(external_row_range(ctx, s->r1cs, U(d_eq_alpha), blk_lo, col_hi, U(d_rows)));  // S4, this rank's columns
        else CK(pk_r1cs_external_row(ctx, s->r1cs, U(d_eq_alpha), U(d_rows)));                            // S4
    }
    fe* wts[3];
    const size_t wlen[3] = {n_witness, n_witness, n_witness};
    std::vector<uint8_t> claimed;
    {
        std::vector<fe> fsum(3), gsum(3);
        Across ac(ctx, st_sharded);
        CK(ac.rc);
        // the statement weights are the rows zero-extended to 2^m (whir_r1cs.rs:391-400): only their support is stored and summed;
        // the three rows share f and g, so all six sums come from one pass (S5)
        for (int k = 0; k < 3; k++) wts[k] = d_rows + (size_t)k * n_witness;
        uint64_t o[24] = {};
        if (overlap_rows) {  // launched before the blinding WHIR proof: finished long ago
            CK(sync_stream(s->side));
            memcpy(o, s->side->h_pinned, 32 * 6);
        } else if (st_sharded || n_witness) {
            CK(dot_rows(ctx, U(d_rows + blk_lo), n_witness, 3, U(W.f_evals + blk_lo), U(W.g_evals + blk_lo), col_n, o));
        }
        for (int k = 0; k < 3; k++) {
            fsum[k] = h_load(o + 8 * k);
            gsum[k] = h_load(o + 8 * k + 4);
        }
        // hint::<(Vec<F>, Vec<F>)>: two ark-serialize vectors (u64 length + canonical elements)
        for (const std::vector<fe>* v : {&fsum, &gsum}) {
            uint64_t cnt = 3;
            for (int i = 0; i < 8; i++) claimed.push_back((uint8_t)(cnt >> (8 * i)));
            for (const fe& x : *v) {
                fe c = h_to_canon(x);
                const uint8_t* b = (const uint8_t*)c.v;
                claimed.insert(claimed.end(), b, b + 32);
            }
        }
    }
    T.hint(claimed.data(), claimed.size());
    lap("external rows + sums");
    // --- WHIR weighted batch opening (whir_r1cs.rs:94-95)
    CK(whir_prove(ctx, A, s->whir_witness, W.com, wts, wlen, 3, T));
    CK(pk_ctx_sync(ctx));
    lap("witness WHIR proof");
    // latency mode: every gated kernel of this proof has completed by now; one that gave up on its challenge computed with zero
    CK(sumcheck_gate_check(ctx));
    // the proof performed exactly the operations its IO pattern declares (what spongefish enforces on the reference's side)
    if (!T.finished())
        return set_err(ctx, PK_ERR_IO_PATTERN, "%s", T.violation().empty() ? "the proof ended before its IO pattern did" : T.violation().c_str());

    guard.ok = true;  // every collective of this proof has been passed
    *len = T.narg.size();
    if (!transcript_out) return PK_OK;  // size query
    PK_REQUIRE(ctx, cap >= T.narg.size(), "transcript buffer too small");
    memcpy(transcript_out, T.narg.data(), T.narg.size());
    return PK_OK;
}

// WhirConfig::new for provekit's parameters (provekit/r1cs-compiler/src/whir_r1cs.rs:38-53); see include/provekit_hip.h
int pk_whir_config_derive(unsigned n_vars, unsigned batch_size, unsigned folding_factor, unsigned starting_log_inv_rate,
                          unsigned security_level, int pow_bits, pk_whir_config* out) {
    // n_vars < folding_factor: whir would run no folding round at all; this prover always folds folding_factor variables before
    // the first re-commit (pk_scheme_create requires n_vars >= folding_factor * (n_rounds + 1)), so the smallest scheme is
    // n_vars = folding_factor -- m_0 >= 2 for the blinding scheme at fold 4 (include/provekit_hip.h)
    if (!out || folding_factor < 1 || folding_factor > 8 || n_vars < folding_factor || starting_log_inv_rate < 1 || batch_size < 1) return PK_ERR_BAD_ARG;
    const unsigned k = folding_factor;
    const double field_bits = 254.0, sec = (double)security_level;
    // default_max_pow(num_variables, log_inv_rate) = num_variables + log_inv_rate - 3 (whir::parameters)
    const double pow_param = pow_bits >= 0 ? (double)pow_bits : (double)(n_vars + starting_log_inv_rate) - 3.0;
    const double protocol_sec = sec > pow_param ? sec - pow_param : 0.0;
    // ConjectureList: log_eta = -(log_inv_rate + 1); list_size_bits = (nv + log_inv_rate) - log_eta
    auto list_size_bits = [](unsigned nv, unsigned rate) { return (double)(nv + rate) + (double)(rate + 1); };
    auto ood_for = [&](unsigned nv, unsigned rate) -> unsigned {
        for (unsigned s = 1; s < 64; s++) {
            double err = 2.0 * list_size_bits(nv, rate) + (double)nv * s;
            if ((double)s * field_bits + 1.0 - err >= sec) return s;
        }
        return 64;
    };
    auto queries_for = [&](unsigned rate) { return (unsigned)ceil(protocol_sec / (double)rate); };
    pk_whir_config c;
    memset(&c, 0, sizeof c);
    c.n_vars = n_vars;
    c.batch_size = batch_size;
    c.folding_factor = k;
    c.starting_log_inv_rate = starting_log_inv_rate;
    const unsigned final_vars = n_vars % k;
    c.n_rounds = (n_vars - final_vars) / k - 1;
    if (c.n_rounds > PK_MAX_WHIR_ROUNDS) return PK_ERR_BAD_ARG;
    c.commitment_ood_samples = ood_for(n_vars, starting_log_inv_rate);
    unsigned nv = n_vars - k, rate = starting_log_inv_rate;
    for (unsigned r = 0; r < c.n_rounds; r++) {
        const unsigned next_rate = rate + (k - 1);
        c.num_queries[r] = queries_for(rate);  // queries against the OLD rate, the rest against the new one
        c.ood_samples[r] = ood_for(nv, next_rate);
        const double query_error = (double)c.num_queries[r] * rate;
        const double combination_error = field_bits - (log2((double)(c.ood_samples[r] + c.num_queries[r])) + list_size_bits(nv, next_rate) + 1.0);
        const double e = query_error < combination_error ? query_error : combination_error;
        c.pow_bits[r] = sec > e ? sec - e : 0.0;
        nv -= k;
        rate = next_rate;
    }
    c.final_queries = queries_for(rate);
    const double fq = (double)c.final_queries * rate;
    c.final_pow_bits = sec > fq ? sec - fq : 0.0;
    c.final_folding_pow_bits = sec > field_bits - 1.0 ? sec - (field_bits - 1.0) : 0.0;
    *out = c;
    return PK_OK;
}

// host-only: one block of the proof RNG's cipher (RFC 8439 layout: words 12,13 = counter, 14,15 = nonce); rounds = 20 for
// the RFC's vectors, 12 (PK_RNG_ROUNDS) for what random_fe_kernel runs
int pk_selftest_chacha(const uint8_t key[32], uint64_t counter, uint32_t n0, uint32_t n1, int rounds, uint8_t out[64]) {
    if (!key || !out || rounds < 2 || (rounds & 1)) return PK_ERR_BAD_ARG;
    RngKey k;
    memcpy(k.k, key, 32);
    u32 blk[16];
    chacha_block(k, counter, n0, n1, rounds, blk);
    memcpy(out, blk, 64);
    return PK_OK;
}
// the device draw itself: n uniform field elements of stream `stream` under `seed32` (what pk_prove fills the mask with)
int pk_selftest_random_fe(pk_ctx* ctx, const uint8_t seed32[32], uint32_t stream, uint64_t* d_out, size_t n) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, seed32 && (n == 0 || d_out), "null pointer");
    if (!n) return PK_OK;
    RngKey k;
    memcpy(k.k, seed32, 32);
    random_fe_kernel<<<random_fe_grid(ctx, n), 256, 0, ctx->stream>>>((fe*)d_out, n, k, stream);
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}

/* ---- NoirProofSchemeProver::prove after ACVM execution (provekit/prover/src/noir_proof_scheme.rs:63-92) ---- */

// the witness transcript (host only): IOPattern + seed_witness_merlin (noir_proof_scheme.rs:111-133), then one
// fill_challenge_scalars per WitnessBuilder::Challenge (witness_builder.rs:94-98)
int pk_witness_challenges(size_t num_constraints, size_t num_witnesses, const uint64_t* public_inputs, size_t n_public, uint64_t* challenges,
                          size_t n_challenges) {
    if ((n_public && !public_inputs) || (n_challenges && !challenges)) return PK_ERR_BAD_ARG;
    Transcript T(witness_io_pattern(n_public, n_challenges));
    T.add_scalar(h_from_u64(num_constraints));
    T.add_scalar(h_from_u64(num_witnesses));
    for (size_t i = 0; i < n_public; i++) T.add_scalar(h_load(public_inputs + 4 * i));
    for (size_t i = 0; i < n_challenges; i++) h_store(challenges + 4 * i, T.challenge_scalar());
    return PK_OK;
}

int pk_witness_fill(pk_ctx* ctx, uint64_t* d_witness, const uint8_t* d_is_set, size_t n, const uint8_t* rng_seed32, size_t* n_filled) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, n == 0 || (d_witness && d_is_set), "null pointer");
    if (n_filled) *n_filled = 0;
    if (!n) return PK_OK;
    RngKey key;
    int rc = proof_key(ctx, rng_seed32, key);
    if (rc) return rc;
    // the count lives in device memory (the first word of the transient workspace; the stream is in order): atomics on the pinned
    // host mailbox would need PCIe atomics, which not every host link provides
    rc = ensure_ws(ctx, 256);
    if (rc) return rc;
    unsigned long long* d_count = (unsigned long long*)ctx->d_ws;
    PK_HIP(ctx, hipMemsetAsync(d_count, 0, 8, ctx->stream));
    fill_witness_kernel<<<grid_for(ctx, n, 256), 256, 0, ctx->stream>>>((fe*)d_witness, d_is_set, n, key, RNG_FILL, d_count);
    PK_LAUNCH_CHECK(ctx);
    if (n_filled) {
        unsigned long long h = 0;
        PK_HIP(ctx, hipMemcpyAsync(&h, d_count, 8, hipMemcpyDeviceToHost, ctx->stream));
        rc = sync_stream(ctx);
        if (rc) return rc;
        *n_filled = (size_t)h;
    }
    return PK_OK;
}

int pk_noir_prove(pk_ctx* ctx, pk_scheme* s, pk_witness_program* builders, const uint64_t* d_acir, size_t n_acir, const uint32_t* public_acir_idx,
                  size_t n_public, const uint8_t* rng_seed32, uint8_t* transcript_out, size_t cap, size_t* len) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, s && builders && len && (d_acir || n_acir == 0) && (public_acir_idx || n_public == 0), "null pointer");
    size_t touched = 0, n_chal = 0, acir_read = 0;
    witness_program_shape(builders, &touched, &n_chal, &acir_read);
    PK_REQUIRE(ctx, touched <= s->num_witnesses, "the witness builders write past the R1CS's witness vector");
    for (size_t i = 0; i < n_public; i++) PK_REQUIRE(ctx, public_acir_idx[i] < n_acir, "public input index outside the ACIR witness vector");
    const size_t nw = s->num_witnesses;
    if (!s->noir_witness) PK_HIP(ctx, hipMalloc((void**)&s->noir_witness, 33 * nw));
    uint64_t* d_w = (uint64_t*)s->noir_witness;
    uint8_t* d_set = (uint8_t*)s->noir_witness + 32 * nw;
    // public values -> host
    std::vector<uint64_t> pub(4 * n_public), chal(4 * n_chal);
    if (n_public) {
        uint32_t* m_idx = nullptr;
        int rc = mail_alloc(ctx, 4 * n_public, (void**)&m_idx);
        if (rc) return rc;
        memcpy(m_idx, public_acir_idx, 4 * n_public);
        gather_fe_kernel<<<(unsigned)((n_public + 255) / 256), 256, 0, ctx->stream>>>((const fe*)d_acir, m_idx, n_public, (fe*)d_w);
        PK_LAUNCH_CHECK(ctx);
        PK_HIP(ctx, hipMemcpyAsync(pub.data(), d_w, 32 * n_public, hipMemcpyDeviceToHost, ctx->stream));
        rc = sync_stream(ctx);
        if (rc) return rc;
    }
    AbortOnFailure guard(ctx);  // a witness that fails to solve on this rank only must not leave the others inside pk_prove's collectives
    int rc = pk_witness_challenges(s->num_constraints, nw, pub.data(), n_public, chal.data(), n_chal);
    if (rc) return set_err(ctx, rc, "witness transcript");
    rc = pk_witness_solve(ctx, builders, d_acir, n_acir, chal.data(), n_chal, d_w, nw, d_set);
    if (rc) return rc;
    RngKey key;  // one key for the fill and the proof's masks (distinct streams)
    rc = proof_key(ctx, rng_seed32, key);
    if (rc) return rc;
    rc = pk_witness_fill(ctx, d_w, d_set, nw, (const uint8_t*)key.k, nullptr);
    if (rc) return rc;
    guard.ok = true;  // pk_prove carries its own guard
    return pk_prove(ctx, s, d_w, nw, (const uint8_t*)key.k, transcript_out, cap, len);
}

int pk_scheme_domain_separator(const pk_scheme* s, char* buf, size_t cap, size_t* len) {
    if (!s || !len) return PK_ERR_BAD_ARG;
    *len = s->domain_separator.size();
    if (buf && cap >= *len) memcpy(buf, s->domain_separator.data(), *len);
    return PK_OK;
}

static bool whir_config_sane(const pk_whir_config* c) {
    return c && c->folding_factor >= 1 && c->folding_factor <= 8 && c->n_rounds <= PK_MAX_WHIR_ROUNDS && c->batch_size >= 1 && c->batch_size <= 4 &&
           c->n_vars >= c->folding_factor * (c->n_rounds + 1) && c->n_vars + c->starting_log_inv_rate <= 28 && c->commitment_ood_samples <= 4;
}

int pk_whir_r1cs_io_pattern(unsigned m_0, const pk_whir_config* whir_witness, const pk_whir_config* whir_for_hiding_spartan, uint8_t* buf,
                            size_t cap, size_t* len) {
    if (!len || m_0 < 1 || m_0 > 27 || !whir_config_sane(whir_witness) || !whir_config_sane(whir_for_hiding_spartan)) return PK_ERR_BAD_ARG;
    const std::string p = whir_r1cs_io_pattern(m_0, *whir_witness, *whir_for_hiding_spartan);
    *len = p.size();
    if (buf && cap >= p.size()) memcpy(buf, p.data(), p.size());
    return PK_OK;
}

int pk_io_pattern_check(const uint8_t* pattern, size_t n, unsigned m_0, const pk_whir_config* whir_witness,
                        const pk_whir_config* whir_for_hiding_spartan, char* why, size_t why_cap) {
    if (why && why_cap) why[0] = 0;
    if (!pattern || m_0 < 1 || m_0 > 27 || !whir_config_sane(whir_witness) || !whir_config_sane(whir_for_hiding_spartan)) return PK_ERR_BAD_ARG;
    const std::string bad = io_pattern_mismatch(std::string((const char*)pattern, n), m_0, *whir_witness, *whir_for_hiding_spartan);
    if (bad.empty()) return PK_OK;
    if (why && why_cap) snprintf(why, why_cap, "%s", bad.c_str());
    return PK_ERR_IO_PATTERN;
}

int pk_scheme_set_io_pattern(pk_ctx* ctx, pk_scheme* s, const uint8_t* pattern, size_t n) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, s, "null pointer");
    if (!pattern || !n) {  // back to the library's restatement
        s->domain_separator = whir_r1cs_io_pattern(s->m_0, s->whir_witness, s->whir_hiding);
        return PK_OK;
    }
    const std::string theirs((const char*)pattern, n);
    const std::string bad = io_pattern_mismatch(theirs, s->m_0, s->whir_witness, s->whir_hiding);
    if (!bad.empty()) return set_err(ctx, PK_ERR_IO_PATTERN, "%s", bad.c_str());
    s->domain_separator = theirs;
    return PK_OK;
}

}  // extern "C"
