// pow.hip -- Skyscraper proof-of-work grinding.
//
// Replaces skyscraper::pow::{threshold, verify, solve} (skyscraper/core/src/pow.rs:14-41) and the
// rayon::broadcast search of generic::solve (skyscraper/core/src/generic.rs:42-71) behind the
// spongefish_pow::PowStrategy plug-in (provekit/common/src/skyscraper/pow.rs:14-30).
// The reference returns *a* valid nonce that depends on thread timing; this grinder returns the
// SMALLEST valid nonce (any valid nonce verifies; the smallest is reproducible): nonces are
// searched in ascending windows, one compression per lane, device-wide atomicMin.
#include <cmath>

// pure ALU like the hash, but a prover is waiting for the nonce: one step above the hash kernels
#define PK_BASE_PRIO 1
#include "ctx.hpp"
#include "skyscraper29s.hpp"

using namespace pk;

namespace {

struct fe_arg {
    u32 v[8];
};
__device__ __forceinline__ fe from_arg(const fe_arg& a) {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = a.v[i];
    return r;
}

// best / ticket: two device words, best == ~0 and ticket == 0 between launches.  Lanes walk the window in ascending
// grid-stride order and stop as soon as a smaller valid nonce is known (a lane's later nonces are all larger), so the work
// done is the expected 2^bits hashes plus about one stride, whatever the window; the result is still the SMALLEST valid
// nonce (every nonce below the final `best` has been tried).  The left input of every hash is the challenge: its round-0
// square is computed once per lane, 13 squarings per hash instead of 14.  The workgroup that draws the last ticket
// publishes the result (or ~0) to pinned host memory and re-arms both words: a window costs one launch and one stream
// synchronisation -- no copy operations.
// (world, rank): nonce ranges striped over the ranks of a device set: this launch tries base + t for t = rank (mod
// world) only; (1, 0) is the whole window.
__global__ __launch_bounds__(256) void pow_search_kernel(fe_arg challenge_arg, fe_arg threshold_arg, unsigned long long base,
                                                         unsigned long long count, unsigned long long* best, unsigned* ticket,
                                                         unsigned long long* host_best, unsigned world, unsigned rank) {
    PK_LATENCY_PRIO();
    const fe29 challenge = to_scaled29(from_arg(challenge_arg));  // generic.rs:81 reduce_partial on arbitrary input
    fe29 s0 = challenge, zero;
#pragma unroll
    for (int k = 0; k < 9; k++) zero.v[k] = 0;
    sky_sq_round_s<0>(s0, zero);  // round 0 for r = 0, once per lane: s0 = 32 sq(challenge)
    const fe threshold = from_arg(threshold_arg);
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x * world;
    for (unsigned long long t = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * world + rank; t < count; t += stride) {
        const unsigned long long nonce = base + t;
        if (nonce > __hip_atomic_load(best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        fe r = fe_zero();
        r.v[0] = (u32)nonce;
        r.v[1] = (u32)(nonce >> 32);
        fe h = from_scaled_canon(compress29s_v2_fixed_left(challenge, s0, unpack29<5>(r)));  // 32*nonce < 2^69: three limbs
        if (fe_lt(h, threshold)) atomicMin(best, nonce);
    }
    __syncthreads();  // every lane of the workgroup has issued its atomicMin
    if (threadIdx.x == 0) {
        __threadfence();  // device scope: orders the atomics above before the ticket (L2 is the coherence point; no write-back)
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
            unsigned long long b = atomicExch(best, ~0ull);
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(host_best, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// skyscraper/core/src/pow.rs:44-82
void f64_to_u256(double f, uint64_t out[4]) {
    uint64_t bits;
    memcpy(&bits, &f, 8);
    bool sign = bits >> 63;
    int exp_bits = (int)((bits >> 52) & 0x7ff);
    uint64_t frac = bits & ((1ull << 52) - 1);
    int exp = exp_bits == 0 ? -1022 : exp_bits - 1023;
    uint64_t significand = exp_bits == 0 ? frac : frac + (1ull << 52);
    memset(out, 0, 32);
    if (sign) return;
    if (exp > 256) {
        memset(out, 0xff, 32);
        return;
    }
    int shift = exp - 52;
    if (shift < 0) {
        double r = std::round(f);
        out[0] = r >= 18446744073709551616.0 ? UINT64_MAX : (r > 0 ? (uint64_t)r : 0);
    } else {
        unsigned limb = (unsigned)shift / 64, sh = (unsigned)shift % 64;
        if (limb > 3) return;
        out[limb] = significand << sh;
        if (sh != 0 && limb < 3) out[limb + 1] = significand >> (64 - sh);
    }
}

// device words + the pinned result slot; first use (or a re-allocated scratch buffer) initialises best = ~0, ticket = 0
int pow_words(pk_ctx* ctx, unsigned long long** best, unsigned** ticket, unsigned long long** host_best) {
    int rc = ensure_scratch(ctx, 64);
    if (!rc) rc = ensure_pinned(ctx);
    if (rc) return rc;
    char* base = (char*)ctx->d_scratch + ctx->scratch_bytes - 64;  // the tail of the scratch buffer: clear of the reduction area
    *best = (unsigned long long*)base;
    *ticket = (unsigned*)(base + 8);
    *host_best = (unsigned long long*)((char*)ctx->h_pinned + PK_PIN_POW);
    if (!ctx->pow_armed) {
        PK_HIP(ctx, hipMemsetAsync(base, 0xff, 8, ctx->stream));
        PK_HIP(ctx, hipMemsetAsync(base + 8, 0, 8, ctx->stream));
        ctx->pow_armed = true;
    }
    return PK_OK;
}

}  // namespace

extern "C" {

// pow.rs:14-22
int pk_pow_threshold(double difficulty, uint64_t out[4]) {
    if (!out || !(difficulty >= 0.0 && difficulty < 80.0)) return PK_ERR_BAD_ARG;  // "Difficulty must be in the range [0, 80)"
    const double modulus = (double)0x30644e72e131a029ull * std::ldexp(1.0, 192);
    const double prob = std::exp2(-difficulty);
    f64_to_u256(prob * modulus, out);
    return PK_OK;
}

// PowStrategy::solve (provekit/common/src/skyscraper/pow.rs:27-29) -> pow::solve (pow.rs:33-41)
int pk_pow_solve(pk_ctx* ctx, const uint8_t challenge[32], double bits, uint64_t* nonce) {
    PK_ENTER(ctx);
    return pk::pow_solve_x(ctx, challenge, bits, nonce, false);
}
}  // extern "C"
namespace pk {
// striped = every rank of the context's device set is inside this call with the same challenge: the nonce space of each window
// is striped over the ranks (rank g tries base + t, t = g mod G), then ONE all-gather of the ranks' 8-byte minima; the result is
// the same smallest valid nonce the lone search returns, on every rank.
int pow_solve_x(pk_ctx* ctx, const uint8_t challenge[32], double bits, uint64_t* nonce, bool striped) {
    PK_REQUIRE(ctx, challenge && nonce, "null pointer");
    PK_REQUIRE(ctx, bits >= 0.0 && bits < 60.0, "bits must be smaller than 60");  // skyscraper/pow.rs:16
    if (bits == 0.0) {  // pow.rs:34-36
        *nonce = 0;
        return PK_OK;
    }
    uint64_t thr[4];
    int rc = pk_pow_threshold(bits + 0.01, thr);  // PROVER_BIAS, pow.rs:6,37
    if (rc) return set_err(ctx, rc, "threshold");
    unsigned long long *d_best, *h_best;
    unsigned* d_ticket;
    rc = ensure_scratch(ctx, ((size_t)1 << 19) + 8 * (size_t)(PK_MAX_RANKS + 1));  // before pow_words: a re-allocation moves them
    if (!rc) rc = pow_words(ctx, &d_best, &d_ticket, &h_best);
    if (rc) return rc;
    fe_arg ch, th;
    memcpy(ch.v, challenge, 32);
    memcpy(th.v, thr, 32);
    const unsigned world = striped ? (unsigned)comm_world(ctx) : 1u, rank = striped ? (unsigned)comm_rank(ctx) : 0u;
    unsigned long long best = ~0ull, base = 0;
    // One launch almost always: the window is 2^(bits+5) nonces (miss probability e^-32) but lanes stop once a smaller valid
    // nonce is known, so the hashes actually computed are ~2^bits plus one stride.  The stride (lanes in flight) is a quarter
    // of the expected work: enough lanes to keep the search short, few enough that little is wasted after the first hit.
    unsigned wbits = (unsigned)bits + 5;
    if (wbits < 12) wbits = 12;
    if (wbits > 62) wbits = 62;  // bits in [58, 60) would shift by >= 63; the loop below moves the window on a miss
    // striped: ranks cannot see each other's hits while they search, so the window is what bounds a rank's work: 2^(bits+1)
    // nonces (miss probability e^-2) = 2^(bits+1) / G hashes per rank at most, then the exchange; a miss moves the window on
    if (world > 1 && wbits > (unsigned)bits + 1) wbits = (unsigned)bits + 1 < 12 ? 12 : (unsigned)bits + 1;
    unsigned long long window = 1ull << wbits;
    unsigned lbits = (unsigned)bits > 2 ? (unsigned)bits - 2 : 0;
    if (lbits < 12) lbits = 12;
    if (lbits > 18) lbits = 18;
    unsigned grid = 1u << (lbits - 8);
    const unsigned grid_cap = (unsigned)ctx->num_cus * 4;
    if (grid > grid_cap) grid = grid_cap;
    for (;;) {
        {
            ProfScope prof(ctx, "pow_search");  // the kernel only: the exchange below waits for the other ranks
            pow_search_kernel<<<grid, 256, 0, ctx->stream>>>(ch, th, base, window, d_best, d_ticket, h_best, world, rank);
        }
        PK_LAUNCH_CHECK(ctx);
        rc = sync_stream(ctx);
        if (rc) return rc;
        best = *(volatile unsigned long long*)h_best;
        if (world > 1) {  // the smallest over the ranks' stripes: 8 bytes per rank
            unsigned long long* d_x = (unsigned long long*)((char*)ctx->d_scratch + ((size_t)1 << 19));
            unsigned long long all[PK_MAX_RANKS];
            PK_HIP(ctx, hipMemcpyAsync(d_x, &best, 8, hipMemcpyHostToDevice, ctx->stream));
            rc = comm_all_gather(ctx, d_x, d_x + 1, 8);
            if (rc) return rc;
            PK_HIP(ctx, hipMemcpyAsync(all, d_x + 1, 8 * (size_t)world, hipMemcpyDeviceToHost, ctx->stream));
            PK_WAIT(ctx);
            for (unsigned r = 0; r < world; r++) best = all[r] < best ? all[r] : best;
        }
        if (best != ~0ull) break;
        base += window;
        if (base > (1ull << 62)) return set_err(ctx, PK_ERR_BAD_ARG, "proof of work search exhausted");
    }
    *nonce = best;
    return PK_OK;
}
}  // namespace pk
extern "C" {

// PowStrategy::check (skyscraper/pow.rs:23-25) -> pow::verify (pow.rs:24-26): NO prover bias
int pk_pow_check(pk_ctx* ctx, const uint8_t challenge[32], double bits, uint64_t nonce, int* ok) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, challenge && ok, "null pointer");
    PK_REQUIRE(ctx, bits >= 0.0 && bits < 60.0, "bits must be smaller than 60");
    if (bits == 0.0) {
        *ok = 1;
        return PK_OK;
    }
    uint64_t thr[4];
    int rc = pk_pow_threshold(bits, thr);
    if (rc) return set_err(ctx, rc, "threshold");
    unsigned long long *d_best, *h_best;
    unsigned* d_ticket;
    rc = pow_words(ctx, &d_best, &d_ticket, &h_best);
    if (rc) return rc;
    fe_arg ch, th;
    memcpy(ch.v, challenge, 32);
    memcpy(th.v, thr, 32);
    pow_search_kernel<<<1, 64, 0, ctx->stream>>>(ch, th, nonce, 1, d_best, d_ticket, h_best, 1, 0);
    PK_LAUNCH_CHECK(ctx);
    rc = sync_stream(ctx);
    if (rc) return rc;
    const unsigned long long best = *(volatile unsigned long long*)h_best;
    *ok = best == nonce;
    return PK_OK;
}

}  // extern "C"
