// selftest.hip -- host-side execution of the exact __host__ __device__ arithmetic the kernels use
// (fe29.hpp, skyscraper29.hpp), so the CPU test suite can check it against the oracle without a GPU.
#include "ctx.hpp"
#include "reduce.hpp"
#include "fe52.hpp"
#include "feinv.hpp"
#include "skyscraper29s.hpp"
#include "transcript.hpp"

using namespace pk;

static fe load_host(const uint64_t* p) {
    fe r;
    memcpy(r.v, p, 32);
    return r;
}
static void store_host(uint64_t* p, const fe& x) { memcpy(p, x.v, 32); }

PK_HD fe selftest_op(int op, const fe& x, const fe& y) {
    fe r = x;
    switch (op) {
        case 0: r = fe_mul29(x, y); break;
        case 1: r = pack29(compress29<2>(unpack_reduce29(x), unpack_reduce29(y))); break;
        case 2: r = pack29(compress29<1>(unpack_reduce29(x), unpack_reduce29(y))); break;
        case 3: r = pack29(from_mont29(x)); break;
        case 4: r = pack29(cond_sub_p29(mont256_29(unpack_reduce29(x), unpack_reduce29(y)))); break;
        case 5: r = pack29(cond_sub_p29(sqr256_29(unpack_reduce29(x)))); break;
        case 6: r = fe_from_montx(x); break;
        case 7: r = fe_to_montx(x); break;
        case 8: r = fe_sqrx(x); break;
        case 9: r = pack29(unpack_reduce29(x)); break;
        case 10: {  // raw reduce256 of the columns of x*y, packed without the final conditional subtraction
            fe29 t = mont256_29(unpack29<0>(x), unpack29<0>(y));
            r = pack29(t);
            break;
        }
        case 11: r = pack29(mont261_29(unpack29<0>(x), unpack29<0>(y))); break;
        case 12: r = pack29(cond_sub_p29(unpack29<0>(x))); break;
        case 13: r = pack29(bar29(unpack29<0>(x))); break;
        case 14: {  // wide_reduce (reduce.hpp): 700*x + 324*y as limb sums, the 1024-term worst case of a grid reduction
            wide w;
            for (int i = 0; i < 8; i++) w.l[i] = 700ull * x.v[i] + 324ull * y.v[i];
            r = wide_reduce(w);
            break;
        }
        // the scaled-by-32 fast path (skyscraper29s.hpp) through its own conversions: must equal ops 1 / 2 / x mod p / op 3
        case 15: r = from_scaled_canon(compress29s<2>(to_scaled29(x), to_scaled29(y))); break;
        case 16: r = from_scaled_canon(compress29s<1>(to_scaled29(x), to_scaled29(y))); break;
        case 17: r = from_scaled_canon(to_scaled29(x)); break;
        case 18: r = from_scaled_canon(mont_to_scaled29(x)); break;
        case 19: {  // a fold of three compressions without leaving the scaled domain: C(C(C(x, y), x), y)
            fe29 a = to_scaled29(x), b = to_scaled29(y);
            fe29 h = compress29s<2>(a, b);
            h = compress29s<2>(h, a);
            h = compress29s<2>(h, b);
            r = from_scaled_canon(h);
            break;
        }
        case 20: {  // dot29: five products (one more than a reduction group): 3 x*y + x*x + y*y, Montgomery products, x, y < p
            dot29 d;
            dot29_init(d);
            dot29_add(d, unpack29<0>(x), unpack29<5>(y));
            dot29_add(d, unpack29<0>(y), unpack29<5>(x));
            dot29_add(d, unpack29<0>(x), unpack29<5>(x));
            dot29_add(d, unpack29<0>(y), unpack29<5>(y));
            dot29_add(d, unpack29<0>(x), unpack29<5>(y));
            r = dot29_result(d);
            break;
        }
        case 21: r = fe_inverse_mont(x); break;   // feinv.hpp: Montgomery in, Montgomery out (0 -> 0)
        case 22: r = fe_inverse_plain(x); break;  // plain integers mod p, constant sequence
        case 23: r = fe_inverse_plain_var(x); break;  // the same, variable-time steps
        default: break;
    }
    return r;
}

__global__ void selftest_kernel(int op, const fe* a, const fe* b, fe* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe x = fe_load(a + i), y = b ? fe_load(b + i) : x;
    fe_store(out + i, selftest_op(op, x, y));
}

// peak-rate probe for SURVEY 8d's second roofline ("achieved modmul/s over measured peak modmul/s"): ILP independent
// register-resident chains of the 9x29-bit Montgomery squaring the hash and NTT kernels use, nothing else.
template <int ILP>
__global__ __launch_bounds__(256) void modmul_rate_kernel(const fe* __restrict__ in, fe* __restrict__ out, unsigned iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fe29 x[ILP];
#pragma unroll
    for (int k = 0; k < ILP; k++) {
        x[k] = unpack_reduce29(fe_load(in + (i % 64)));
        x[k].v[0] += (u32)k;
    }
    for (unsigned it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < ILP; k++) x[k] = sqr256_29(x[k]);
    }
    fe29 acc = x[0];
#pragma unroll
    for (int k = 1; k < ILP; k++) acc = add29(acc, x[k]);
    normalize29(acc);
    if (acc.v[8] == 0xffffffffu) fe_store(out + i, pack29(acc));  // never true (limbs stay < 2^30); keeps the chain live
}

// ---- the f64-FMA multiplier prototype (fe52.hpp) ----------------------------------------------------------------------------
__global__ void fp52_sqr_kernel(const fe* a, u64* out, size_t n) {
    f52_enter_rtz();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe52 x = unpack52(fe_load(a + i));
#pragma unroll
    for (int k = 0; k < 5; k++) asm volatile("" : "+v"(x.v[k]));  // the limbs exist only after the mode switch
    const fe52 r = sqr260_52(x);
#pragma unroll
    for (int k = 0; k < 5; k++) out[5 * i + k] = r.v[k];
}
template <int ILP>
__global__ __launch_bounds__(256) void modmul_rate_fp52_kernel(const fe* __restrict__ in, u64* __restrict__ out, unsigned iters) {
    f52_enter_rtz();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fe52 x[ILP];
#pragma unroll
    for (int k = 0; k < ILP; k++) {
        x[k] = unpack52(fe_load(in + (i % 64)));
        x[k].v[0] ^= (u64)k;
#pragma unroll
        for (int j = 0; j < 5; j++) asm volatile("" : "+v"(x[k].v[j]));
    }
    for (unsigned it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < ILP; k++) x[k] = sqr260_52(x[k]);
    }
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < ILP; k++)
#pragma unroll
        for (int j = 0; j < 5; j++) acc += x[k].v[j];
    if (acc == 0xffffffffffffffffull) out[i] = acc;  // never true (limbs < 2^52); keeps the chains live
}

// ---- wavefront-cooperative square round, PROTOTYPE (DESIGN.md 4 "Work mapping of the hash") ----------------
// One node per wavefront: lane j < 9 holds limb j of the scaled state (l, r).  Operand scanning with the accumulator window
// shifted one lane per Montgomery step: after step i lane j holds column i + 1 + j.  Broadcasts of l_i and m_i go through an SGPR
// (v_readlane), the window shift is a DPP row_shl.  Same function as sky_sq_round_s<0> (skyscraper29s.hpp): the results agree as
// integers mod p and in their limb bounds; limbs are carried in three parallel passes instead of a ripple, so individual limbs
// may differ by a carry.  Timed with s_memtime against the lone lane running the same number of rounds.
__device__ __forceinline__ u32 coop_shl1(u32 x) {  // lane j <- lane j + 1 within the row of 16; the row's last lane gets 0
    return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x101, 0xf, 0xf, true);
}
__device__ __forceinline__ u32 coop_shr1(u32 x) {  // lane j <- lane j - 1; lane 0 gets 0
    return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
}
__device__ __forceinline__ void coop_sq_round(u32& L, u32& R, u32 Pj, u32 RCj, u32 lane0_mask, bool is_top) {
    u64 A = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const u32 s = (u32)__builtin_amdgcn_readlane((int)L, i);
        A += (u64)L * s;                                   // l_j * l_i -> column i + j, held by lane j
        const u32 mloc = ((u32)A * NP29) & M29;
        const u32 m = (u32)__builtin_amdgcn_readlane((int)mloc, 0);
        A += (u64)m * Pj;                                  // column i is now a multiple of 2^29
        const u64 c = A >> 29;
        const u32 clo = (u32)__builtin_amdgcn_readlane((int)(u32)c, 0), chi = (u32)__builtin_amdgcn_readlane((int)(u32)(c >> 32), 0);
        A = (u64)coop_shl1((u32)A) | ((u64)coop_shl1((u32)(A >> 32)) << 32);  // window moves up one column
        A += (u64)(clo & lane0_mask) | ((u64)(chi & lane0_mask) << 32);        // carry of the finished column into the new lane 0
    }
    // lanes 0..7: columns 9..16; add r + 32 rc in place (lane 8's share, the top limb, is added after the carries)
    const u32 q = R + RCj;
    A += is_top ? 0u : q;
    // carries, three parallel passes (a carry is < 2^35, then < 2^7, then <= 1)
    u32 lo = (u32)A & M29;
    u64 c1 = A >> 29;
    u64 t = (u64)lo + ((u64)coop_shr1((u32)c1) | ((u64)coop_shr1((u32)(c1 >> 32)) << 32));
    u32 lo2 = (u32)t & M29, c2 = (u32)(t >> 29);
    u32 t2 = lo2 + coop_shr1(c2);
    u32 lo3 = t2 & M29, c3 = t2 >> 29;
    const u32 cin3 = coop_shr1(c3);
    u32 sres = lo3 + cin3;
    // lane 8: every carry out of column 16 (one per pass, unmasked: the top limb holds the rest) plus the top limb of r + 32 rc
    const u32 top8 = (u32)t + coop_shr1(c2) + cin3 + q;
    sres = is_top ? top8 : sres;
    R = L;
    L = sres;
}
__global__ void coop_round_kernel(const u32* __restrict__ in_l, const u32* __restrict__ in_r, unsigned iters, u32* __restrict__ out,
                                  unsigned long long* __restrict__ cycles, int mode, unsigned long long active) {
    const unsigned lane = threadIdx.x;
    u32 L = lane < 9 ? in_l[lane] : 0u, R = lane < 9 ? in_r[lane] : 0u;
    const u32 Pj = lane < 9 ? p29((int)(lane < 9 ? lane : 0)) : 0u;
    u32 RCj = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) RCj = lane == (unsigned)k ? rcs29<0>(k) : RCj;
    const u32 lane0_mask = lane == 0 ? 0xffffffffu : 0u;
    const bool is_top = lane == 8;
    unsigned long long t0 = __builtin_readcyclecounter();
    if (mode != 2)
        for (unsigned it = 0; it < iters; it++) coop_sq_round(L, R, Pj, RCj, lane0_mask, is_top);
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane < 9) {
        out[lane] = L;
        out[9 + lane] = R;
    }
    // the lone lane: the product path's own round, same count
    // (indexed through the lane id so that the compiler cannot prove the values wave-uniform and move the whole round to the
    // scalar ALU: the product's lanes hold different nodes)
    fe29 l, r;
    const unsigned off = (lane >> 6) * 32;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        l.v[k] = in_l[k + off];
        r.v[k] = in_r[k + off];
    }
    unsigned long long t2 = __builtin_readcyclecounter();
    if (mode != 1 && ((active >> lane) & 1ull))  // PK_COOP_ACTIVE: which lanes of the wavefront run the lone-lane loop
        for (unsigned it = 0; it < iters; it++) sky_sq_round_s<0>(l, r);
    unsigned long long t3 = __builtin_readcyclecounter();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 9; k++) {
            out[18 + k] = l.v[k];
            out[27 + k] = r.v[k];
        }
        cycles[0] = t1 - t0;
        cycles[1] = t3 - t2;
    }
}

extern "C" {

// PROTOTYPE probe: `iters` square rounds (round constant 0) of the scaled Skyscraper state (l, r: 9 limbs of 29 bits each) by the
// wavefront-cooperative round and by the lone lane, one wavefront on an idle GPU.  out: 36 words = coop l, coop r, lane l, lane r;
// cycles: s_memtime ticks of the two loops.
int pk_selftest_coop_round(pk_ctx* ctx, const uint32_t l[9], const uint32_t r[9], unsigned iters, uint32_t out[36], uint64_t cycles[4]) {
    if (!ctx || !l || !r || !out || !cycles) return PK_ERR_BAD_ARG;
    PK_ENTER(ctx);
    int rc = ensure_scratch(ctx, 4096);
    if (rc) return rc;
    u32* d = (u32*)ctx->d_scratch;
    PK_HIP(ctx, hipMemcpyAsync(d, l, 36, hipMemcpyHostToDevice, ctx->stream));
    PK_HIP(ctx, hipMemcpyAsync(d + 16, r, 36, hipMemcpyHostToDevice, ctx->stream));
    const char* ae = getenv("PK_COOP_ACTIVE");  // hex lane mask of the lone-lane loop (default: all 64)
    const unsigned long long active = ae ? strtoull(ae, nullptr, 16) : ~0ull;
    coop_round_kernel<<<1, 64, 0, ctx->stream>>>(d, d + 16, iters, d + 64, (unsigned long long*)(d + 128), 0, ~0ull);
    PK_LAUNCH_CHECK(ctx);
    PK_HIP(ctx, hipMemcpyAsync(out, d + 64, 144, hipMemcpyDeviceToHost, ctx->stream));
    PK_HIP(ctx, hipMemcpyAsync(cycles, d + 128, 16, hipMemcpyDeviceToHost, ctx->stream));
    rc = sync_stream(ctx);
    if (rc) return rc;
    // the same two loops timed from outside (hipEvents), each in its own launch: nanoseconds per round in cycles[2], cycles[3]
    // (PK_COOP_GRID: the same single-wavefront workgroup replicated over the chip -- every copy writes the same words -- to see
    // what the clock does when the GPU is not idle around the measured wavefront)
    const char* ge = getenv("PK_COOP_GRID");
    const unsigned grid = ge ? (unsigned)atoi(ge) : 1u;
    for (int mode = 1; mode <= 2; mode++) {
        float ms = 0;
        if ((rc = pk_timer_start(ctx))) return rc;
        coop_round_kernel<<<grid ? grid : 1u, 64, 0, ctx->stream>>>(d, d + 16, iters, d + 256, (unsigned long long*)(d + 384), mode, active);
        PK_LAUNCH_CHECK(ctx);
        if ((rc = pk_timer_stop(ctx, &ms))) return rc;
        cycles[1 + mode] = (uint64_t)(1e6 * (double)ms);  // ns for `iters` rounds (plus one launch)
    }
    return PK_OK;
}

// x (n field elements, 4 x u64, any value < 2^256) -> the five 52-bit limbs of sqr260_52(x) = x^2 * 2^-260 mod p, lazily reduced
// (value < 2^257).  Host execution of the shared code under fesetround(FE_TOWARDZERO).
int pk_selftest_fp52_sqr(const uint64_t* a, uint64_t* out5, size_t n) {
    if (!a || !out5) return PK_ERR_BAD_ARG;
    const int old = fegetround();
    if (fesetround(FE_TOWARDZERO)) return PK_ERR_BAD_ARG;
    for (size_t i = 0; i < n; i++) {
        const fe52 r = sqr260_52(unpack52(load_host(a + 4 * i)));
        for (int k = 0; k < 5; k++) out5[5 * i + k] = r.v[k];
    }
    fesetround(old);
    return PK_OK;
}
int pk_selftest_fp52_sqr_device(pk_ctx* ctx, const uint64_t* d_a, uint64_t* d_out5, size_t n) {
    if (!ctx || !d_a || !d_out5) return PK_ERR_BAD_ARG;
    PK_ENTER(ctx);
    if (!n) return PK_OK;
    fp52_sqr_kernel<<<(unsigned)((n + 63) / 64), 64, 0, ctx->stream>>>((const fe*)d_a, (u64*)d_out5, n);
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}
// the counterpart of pk_selftest_modmul_rate for the f64-FMA square: register-resident chains, nothing else
int pk_selftest_modmul_rate_fp52(pk_ctx* ctx, unsigned waves_per_simd, unsigned ilp, unsigned iters, double* modmul_per_s) {
    if (!ctx || !modmul_per_s) return PK_ERR_BAD_ARG;
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, waves_per_simd >= 1 && waves_per_simd <= 8 && (ilp == 1 || ilp == 2 || ilp == 4) && iters >= 1, "waves 1..8, ilp 1|2|4");
    int rc = ensure_scratch(ctx, (size_t)ctx->num_cus * 8 * 256 * 8);
    if (rc) return rc;
    PK_HIP(ctx, hipMemsetAsync(ctx->d_scratch, 0x11, 64 * 32, ctx->stream));
    const unsigned blocks = (unsigned)ctx->num_cus * waves_per_simd;
    auto launch = [&](unsigned n) {
        const fe* in = (const fe*)ctx->d_scratch;
        u64* out = (u64*)ctx->d_scratch;
        if (ilp == 1) modmul_rate_fp52_kernel<1><<<blocks, 256, 0, ctx->stream>>>(in, out, n);
        else if (ilp == 2) modmul_rate_fp52_kernel<2><<<blocks, 256, 0, ctx->stream>>>(in, out, n);
        else modmul_rate_fp52_kernel<4><<<blocks, 256, 0, ctx->stream>>>(in, out, n);
    };
    launch(16);
    PK_LAUNCH_CHECK(ctx);
    float ms = 0;
    if ((rc = pk_timer_start(ctx))) return rc;
    launch(iters);
    PK_LAUNCH_CHECK(ctx);
    if ((rc = pk_timer_stop(ctx, &ms))) return rc;
    *modmul_per_s = (double)blocks * 256.0 * ilp * iters / (ms * 1e-3);
    return PK_OK;
}

int pk_selftest_modmul_rate(pk_ctx* ctx, unsigned waves_per_simd, unsigned ilp, unsigned iters, double* modmul_per_s) {
    if (!ctx || !modmul_per_s) return PK_ERR_BAD_ARG;
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, waves_per_simd >= 1 && waves_per_simd <= 8 && (ilp == 1 || ilp == 2 || ilp == 4) && iters >= 1, "waves 1..8, ilp 1|2|4");
    int rc = ensure_scratch(ctx, 64 * 32);
    if (rc) return rc;
    PK_HIP(ctx, hipMemsetAsync(ctx->d_scratch, 0x11, 64 * 32, ctx->stream));
    const unsigned blocks = (unsigned)ctx->num_cus * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
    auto launch = [&](unsigned n) {
        const fe* in = (const fe*)ctx->d_scratch;
        fe* out = (fe*)ctx->d_scratch;
        if (ilp == 1) modmul_rate_kernel<1><<<blocks, 256, 0, ctx->stream>>>(in, out, n);
        else if (ilp == 2) modmul_rate_kernel<2><<<blocks, 256, 0, ctx->stream>>>(in, out, n);
        else modmul_rate_kernel<4><<<blocks, 256, 0, ctx->stream>>>(in, out, n);
    };
    launch(16);
    PK_LAUNCH_CHECK(ctx);
    float ms = 0;
    if ((rc = pk_timer_start(ctx))) return rc;
    launch(iters);
    PK_LAUNCH_CHECK(ctx);
    if ((rc = pk_timer_stop(ctx, &ms))) return rc;
    *modmul_per_s = (double)blocks * 256.0 * ilp * iters / (ms * 1e-3);
    return PK_OK;
}

// the same ops executed by a kernel (device pointers): lets the GPU suite diff device vs host codegen
int pk_selftest_arith_device(pk_ctx* ctx, int op, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, size_t n) {
    if (!ctx || !d_a || !d_out) return PK_ERR_BAD_ARG;
    PK_ENTER(ctx);
    if (!n) return PK_OK;
    selftest_kernel<<<(unsigned)((n + 63) / 64), 64, 0, ctx->stream>>>(op, (const fe*)d_a, (const fe*)d_b, (fe*)d_out, n);
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}

// domain-separator tag (Keccak duplex, overwrite mode) and one Skyscraper sponge permutation, host only
int pk_selftest_keccak_tag(const uint8_t* data, size_t len, uint8_t tag[32]) {
    if (!tag || (len && !data)) return PK_ERR_BAD_ARG;
    keccak_tag(std::string((const char*)data, len), tag);
    return PK_OK;
}
int pk_selftest_permute(uint64_t l[4], uint64_t r[4]) {
    if (!l || !r) return PK_ERR_BAD_ARG;
    fe a = load_host(l), b = load_host(r);
    sky_permute_host(a, b);
    store_host(l, a);
    store_host(r, b);
    return PK_OK;
}

// op: 0 fe_mul29(a,b)  1 compress v2  2 compress v1  3 from_mont  4 a*b*2^-256 via mont256_29  5 a^2*2^-256 via sqr256_29
// a, b, out: n field elements (4 x u64 each).  Host only; no device needed.
int pk_selftest_arith(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    if (!a || !out || (!b && (op == 0 || op == 1 || op == 2 || op == 4 || op == 15 || op == 16 || op == 19 || op == 20))) return PK_ERR_BAD_ARG;
    for (size_t i = 0; i < n; i++) {
        fe x = load_host(a + 4 * i), y = b ? load_host(b + 4 * i) : x;
        if (op < 0 || op > 23) return PK_ERR_BAD_ARG;
        fe r = selftest_op(op, x, y);
        store_host(out + 4 * i, r);
    }
    return PK_OK;
}

}  // extern "C"
