// probes.hip -- libpk_probes.so: the LAB, built next to the product and never part of it.
// Measurement probes (multiplier peak rates, Fiat-Shamir round-trip costs) and the prototypes that were measured and rejected
// (f64-FMA multiplier, wavefront-cooperative round, reduction on the matrix core, Shoup constants), plus the device-side run of
// the arithmetic self-test table.  Compiled from the product's own headers (-I provekit_amd/csrc) and linked against
// libprovekit_hip.so for the context plumbing; declared in pk_probes.h; Python binding tools/pk_probes.py.
#include "selftest_ops.hpp"
#include "fe52.hpp"
#include "transcript.hpp"
#include "pk_probes.h"

using namespace pk;

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

// ---- what hand register allocation of the square round could buy at most -----------------------------------------
// The square round of skyscraper29s.hpp (sky_sq_round_s) is 126 multiply-adds + ~99 other instructions; ~23 of the others exist only
// because the compiler cannot place r + rc into the accumulator columns for free: 8 additions r + rc, 7 moves (the zero high halves of
// the 64-bit column initial values) and 8 shifts (the doubled operand of the cross products).  Assembly could fold the first two groups
// (consecutive-register pairs with a shared zero high half; the constant through the reduction's low columns as the first
// multiply-add's 64-bit scalar addend) and nothing else.  ABLATION: the same round with those instructions simply left out -- the
// results are wrong, the multiply-add count and the dependency structure are the product's -- is an upper bound on what any
// hand-written version of the round can reach:  variant 0 = the product's round; 1 = no round constant (drops the 8 additions);
// 2 = neither r nor rc enter the columns (drops additions AND zero-extensions: the columns start from 0 like columns 0..8);
// 3 = variant 2 and the cross products use l instead of 2l (drops the 8 doublings as well).
template <int VARIANT>
__device__ __forceinline__ void sq_round_ablated(fe29& l, fe29& r) {
    if (VARIANT == 0) {
        sky_sq_round_s<3>(l, r);
        return;
    }
    u64 acc[17];
#pragma unroll
    for (int k = 0; k < 9; k++) acc[k] = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) acc[9 + k] = VARIANT == 1 ? (u64)r.v[k] : 0;
    const u32 top = VARIANT == 1 ? r.v[8] : 0u;
    u32 a2[9];
#pragma unroll
    for (int j = 0; j < 9; j++) a2[j] = VARIANT == 3 ? l.v[j] : l.v[j] << 1;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        acc[2 * i] += (u64)l.v[i] * l.v[i];
#pragma unroll
        for (int j = i + 1; j < 9; j++) acc[i + j] += (u64)l.v[i] * a2[j];
    }
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const u32 m = ((u32)acc[i] * NP29) & M29;
#pragma unroll
        for (int j = 0; j < 9; j++)
            if (i + j < 17) acc[i + j] += (u64)m * p29(j);
        acc[i + 1] += acc[i] >> 29;
    }
    fe29 s;
#pragma unroll
    for (int k = 9; k < 16; k++) {
        acc[k + 1] += acc[k] >> 29;
        s.v[k - 9] = (u32)acc[k] & M29;
    }
    s.v[7] = (u32)acc[16] & M29;
    s.v[8] = ((u32)(acc[16] >> 29) + top) & M29;  // (keeps the wrong values inside the column bound)
    r = l;
    l = s;
}
template <int VARIANT, int ILP>
__global__ __launch_bounds__(256) void sq_round_rate_kernel(const fe* __restrict__ in, fe* __restrict__ out, unsigned iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fe29 l[ILP], r[ILP];
#pragma unroll
    for (int k = 0; k < ILP; k++) {
        l[k] = unpack_reduce29(fe_load(in + (i % 64)));
        r[k] = unpack_reduce29(fe_load(in + ((i + 7) % 64)));
        l[k].v[0] += (u32)k;
    }
    for (unsigned it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < ILP; k++) sq_round_ablated<VARIANT>(l[k], r[k]);
    }
    fe29 acc = l[0];
#pragma unroll
    for (int k = 1; k < ILP; k++) acc = add29(acc, l[k]);
    normalize29(acc);
    if (acc.v[8] == 0xffffffffu) fe_store(out + i, pack29(acc));  // never true; keeps the chains live
}

// ---- the f64-FMA multiplier prototype (fe52.hpp) ----------------------------------------------------------------------------
// general products by a constant: Montgomery (mont261_29) against Shoup (shoup261_29), register-resident chains
template <int ILP, bool SHOUP>
__global__ __launch_bounds__(256) void constmul_rate_kernel(const fe* __restrict__ in, fe* __restrict__ out, unsigned iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fe29 x[ILP];
    const fe29 w = unpack_reduce29(fe_load(in + ((i + 1) % 64)));
    const fe29 wq = unpack_reduce29(fe_load(in + ((i + 2) % 64)));
#pragma unroll
    for (int k = 0; k < ILP; k++) {
        x[k] = unpack_reduce29(fe_load(in + (i % 64)));
        x[k].v[0] += (u32)k;
    }
    for (unsigned it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < ILP; k++) x[k] = SHOUP ? shoup261_29(x[k], w, wq) : mont261_29(x[k], w);
    }
    fe29 acc = x[0];
#pragma unroll
    for (int k = 1; k < ILP; k++) acc = add29(acc, x[k]);
    normalize29(acc);
    if (acc.v[8] == 0xffffffffu) fe_store(out + i, pack29(acc));
}

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
int pk_probe_coop_round(pk_ctx* ctx, const uint32_t l[9], const uint32_t r[9], unsigned iters, uint32_t out[36], uint64_t cycles[4]) {
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
int pk_probe_fp52_sqr(const uint64_t* a, uint64_t* out5, size_t n) {
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
int pk_probe_fp52_sqr_device(pk_ctx* ctx, const uint64_t* d_a, uint64_t* d_out5, size_t n) {
    if (!ctx || !d_a || !d_out5) return PK_ERR_BAD_ARG;
    PK_ENTER(ctx);
    if (!n) return PK_OK;
    fp52_sqr_kernel<<<(unsigned)((n + 63) / 64), 64, 0, ctx->stream>>>((const fe*)d_a, (u64*)d_out5, n);
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}
// the counterpart of pk_probe_modmul_rate for the f64-FMA square: register-resident chains, nothing else
int pk_probe_modmul_rate_fp52(pk_ctx* ctx, unsigned waves_per_simd, unsigned ilp, unsigned iters, double* modmul_per_s) {
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

// products by a constant per second: shoup = 0 the Montgomery product the NTT uses today, 1 the Shoup form
int pk_probe_constmul_rate(pk_ctx* ctx, unsigned waves_per_simd, unsigned ilp, unsigned iters, int shoup, double* modmul_per_s) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, modmul_per_s && waves_per_simd >= 1 && waves_per_simd <= 8 && (ilp == 1 || ilp == 2) && iters >= 1, "bad argument");
    int rc = ensure_scratch(ctx, 1 << 20);
    if (rc) return rc;
    const unsigned blocks = (unsigned)ctx->num_cus * waves_per_simd;
    fe* in = (fe*)ctx->d_scratch;
    fe* out = in + 64;
    PK_HIP(ctx, hipMemsetAsync(in, 0x11, 64 * 32, ctx->stream));
    auto launch = [&](unsigned n) {
        if (shoup) {
            if (ilp == 1) constmul_rate_kernel<1, true><<<blocks, 256, 0, ctx->stream>>>(in, out, n);
            else constmul_rate_kernel<2, true><<<blocks, 256, 0, ctx->stream>>>(in, out, n);
        } else {
            if (ilp == 1) constmul_rate_kernel<1, false><<<blocks, 256, 0, ctx->stream>>>(in, out, n);
            else constmul_rate_kernel<2, false><<<blocks, 256, 0, ctx->stream>>>(in, out, n);
        }
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

// square rounds per second of the hash's round and of its ablated variants (see sq_round_ablated): ilp = independent (l, r) chains per lane
int pk_probe_sq_round_rate(pk_ctx* ctx, int variant, unsigned waves_per_simd, unsigned ilp, unsigned iters, double* rounds_per_s) {
    if (!ctx || !rounds_per_s) return PK_ERR_BAD_ARG;
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, variant >= 0 && variant <= 3 && waves_per_simd >= 1 && waves_per_simd <= 8 && (ilp == 1 || ilp == 2) && iters >= 1, "bad argument");
    int rc = ensure_scratch(ctx, 1 << 20);
    if (rc) return rc;
    const unsigned blocks = (unsigned)ctx->num_cus * waves_per_simd;
    fe* in = (fe*)ctx->d_scratch;
    fe* out = in + 64;
    PK_HIP(ctx, hipMemsetAsync(in, 0x11, 64 * 32, ctx->stream));
    auto launch = [&](unsigned n) {
#define PK_SQR(V)                                                                              \
    if (ilp == 1) sq_round_rate_kernel<V, 1><<<blocks, 256, 0, ctx->stream>>>(in, out, n);     \
    else sq_round_rate_kernel<V, 2><<<blocks, 256, 0, ctx->stream>>>(in, out, n)
        if (variant == 0) { PK_SQR(0); }
        else if (variant == 1) { PK_SQR(1); }
        else if (variant == 2) { PK_SQR(2); }
        else { PK_SQR(3); }
#undef PK_SQR
    };
    launch(16);
    PK_LAUNCH_CHECK(ctx);
    float ms = 0;
    if ((rc = pk_timer_start(ctx))) return rc;
    launch(iters);
    PK_LAUNCH_CHECK(ctx);
    if ((rc = pk_timer_stop(ctx, &ms))) return rc;
    *rounds_per_s = (double)blocks * 256.0 * ilp * iters / (ms * 1e-3);
    return PK_OK;
}

int pk_probe_modmul_rate(pk_ctx* ctx, unsigned waves_per_simd, unsigned ilp, unsigned iters, double* modmul_per_s) {
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
int pk_probe_arith_device(pk_ctx* ctx, int op, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, size_t n) {
    if (!ctx || !d_a || !d_out) return PK_ERR_BAD_ARG;
    PK_ENTER(ctx);
    if (!n) return PK_OK;
    selftest_kernel<<<(unsigned)((n + 63) / 64), 64, 0, ctx->stream>>>(op, (const fe*)d_a, (const fe*)d_b, (fe*)d_out, n);
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}
}  // extern "C"

// ============================================================================================================================
// VERDICT r03 item 5 -- the one execution unit never tried: modular REDUCTION as a constant-matrix product on the matrix core.
//
// Formulation (the best one found; DESIGN.md 4 "Reduction on the matrix core").  The 522-bit square t = a^2 sits in 18 limbs of
// 29 bits.  A normalised limb is already four digits of 8, 8, 8 and 5 bits in the byte lanes of its register, so t is 72 digits
// d_k at bit positions e_k = 29 (k / 4) + 8 (k % 4) with NO conversion.  Reduction is linear in the digits:
//     t * 2^-256  ==  sum_k d_k * c_k  (mod p),       c_k = 2^(e_k - 256) mod p   (constants, < p)
// -- the Montgomery factor costs nothing, it is inside the constants -- and writing each c_k in the same mixed-radix digits makes
// the sum a (36 x 72) by (72 x batch) integer matrix product: v_mfma_i32_16x16x64_i8, rows = output digits, columns = values.
// i8 is signed, so digits are recoded to [-128, 127] (per limb: add 0x00808080, xor 0x00808080) and an offset multiple of p keeps
// the total positive.  The 36 column sums (|.| < 2^21) are assembled into 9 limbs by shifts and two 64-bit multiply-adds per limb,
// and one 9-multiply-add fold of the bits above 2^253 brings the result under 2^254 + small.
//
// Three probes (tools/mfma_reduce.py -> profiles/r04_modmul_rates.json):
//   pk_probe_mfma_reduce        the matrix product itself on real inputs, operands loaded straight in fragment layout: EXACT
//                                  (tests/test_gpu_selftest.py checks sum C_i 2^(f_i) == t 2^-256 mod p and the bound)
//   pk_probe_mfma_reduce_rate   the matrix pipe alone: the 24 MFMAs one wavefront (64 values) needs per squaring, register resident
//   pk_probe_mfma_valu_rate     the vector work that REMAINS per squaring (the 45-product square, the carry sweep to digits, the
//                                  signed recoding, the limb assembly, the top fold) with the matrix products and all cross-lane
//                                  movement (about 40 v_permlane swaps each way) taken as free
// The achievable rate is below min(pipe, remainder) -- both are reported next to the 29-bit integer squaring they would replace.
// ============================================================================================================================
typedef int v4i __attribute__((ext_vector_type(4)));

// host: signed mixed-radix digits (positions 29 q + 8 b, b < 4; byte 3 of a limb carries 5 bits + what is left) of 0 <= x < 2^261
static void mixed_digits_signed(const unsigned __int128 lohi[3], int out[36]) {
    // x as 9 limbs of 29 bits
    u32 limb[9];
    unsigned __int128 w0 = lohi[0], w1 = lohi[1], w2 = lohi[2];  // 128 + 128 + 5 bits
    auto bit = [&](int i) -> u32 { return i < 128 ? (u32)(w0 >> i) & 1u : i < 256 ? (u32)(w1 >> (i - 128)) & 1u : (u32)(w2 >> (i - 256)) & 1u; };
    for (int q = 0; q < 9; q++) {
        u32 v = 0;
        for (int b = 0; b < 29; b++) v |= bit(29 * q + b) << b;
        limb[q] = v;
    }
    for (int q = 0; q < 9; q++) {
        u32 y = (limb[q] + 0x00808080u) ^ 0x00808080u;
        out[4 * q + 0] = (int8_t)(y & 255);
        out[4 * q + 1] = (int8_t)((y >> 8) & 255);
        out[4 * q + 2] = (int8_t)((y >> 16) & 255);
        out[4 * q + 3] = (int)(y >> 24);  // 0 .. 32
    }
}

// fragment tables: A[T][kb][lane] = 16 signed bytes, row i = 16 T + lane % 16, k = 64 kb + 16 (lane / 16) + [0, 16)
struct MfmaReduceTables {
    v4i A[3][2][64];
    v4i C0[3][64];  // accumulator start: the digits of the offset OFFS * p, rows 16 T + 4 (lane / 16) + v
};

// c = 2^e * 2^-256 mod p by repeated doubling / halving on host 64-bit limbs
static void pow2_mod_p(int e, uint64_t out[4]) {
    using namespace pk::host64;
    uint64_t x[4] = {1, 0, 0, 0};
    if (e >= 0) {
        for (int i = 0; i < e; i++) {
            uint64_t y[4] = {x[0], x[1], x[2], x[3]};
            add_mod(x, y);
        }
    } else {
        for (int i = 0; i < -e; i++) {  // halve: (x + (x odd ? p : 0)) / 2
            unsigned __int128 c = 0;
            uint64_t t[5];
            const bool odd = x[0] & 1;
            for (int k = 0; k < 4; k++) {
                c += (unsigned __int128)x[k] + (odd ? P64[k] : 0);
                t[k] = (uint64_t)c;
                c >>= 64;
            }
            t[4] = (uint64_t)c;
            for (int k = 0; k < 4; k++) x[k] = (t[k] >> 1) | (t[k + 1] << 63);
        }
    }
    memcpy(out, x, 32);
}

constexpr int MFMA_OFFS_LOG2 = 15;  // offset 2^15 * p: above 72 * 128 * p, the most negative the signed digits can make the sum

static void build_mfma_tables(MfmaReduceTables& T) {
    static int A[48][128];
    memset(A, 0, sizeof A);
    for (int k = 0; k < 72; k++) {
        uint64_t c[4];
        pow2_mod_p(29 * (k / 4) + 8 * (k % 4) - 256, c);
        unsigned __int128 w[3] = {((unsigned __int128)c[1] << 64) | c[0], ((unsigned __int128)c[3] << 64) | c[2], 0};
        int dg[36];
        mixed_digits_signed(w, dg);
        for (int i = 0; i < 36; i++) A[i][k] = dg[i];
    }
    for (int t = 0; t < 3; t++)
        for (int kb = 0; kb < 2; kb++)
            for (int lane = 0; lane < 64; lane++) {
                int8_t b[16];
                for (int j = 0; j < 16; j++) b[j] = (int8_t)A[16 * t + lane % 16][64 * kb + 16 * (lane / 16) + j];
                memcpy(&T.A[t][kb][lane], b, 16);
            }
    // offset 2^15 p (< 2^269): plain (unsigned, unrecoded) mixed-radix digits; the top digit takes everything above bit 253
    {
        using namespace pk::host64;
        unsigned __int128 w[3] = {0, 0, 0};
        // 2^15 * p as a 269-bit integer in three 128-bit words (the third holds bits 256..)
        unsigned __int128 carry = 0;
        uint64_t o[5];
        for (int k = 0; k < 4; k++) {
            unsigned __int128 v = ((unsigned __int128)P64[k] << MFMA_OFFS_LOG2) + carry;
            o[k] = (uint64_t)v;
            carry = v >> 64;
        }
        o[4] = (uint64_t)carry;
        w[0] = ((unsigned __int128)o[1] << 64) | o[0];
        w[1] = ((unsigned __int128)o[3] << 64) | o[2];
        w[2] = o[4];
        int dg[48] = {};
        auto bit = [&](int i) -> u32 { return i < 128 ? (u32)(w[0] >> i) & 1u : i < 256 ? (u32)(w[1] >> (i - 128)) & 1u : (u32)(w[2] >> (i - 256)) & 1u; };
        for (int q = 0; q < 9; q++)
            for (int b = 0; b < 4; b++) {
                const int pos = 29 * q + 8 * b, width = (q == 8 && b == 3) ? 40 : (b == 3 ? 5 : 8);
                int v = 0;
                for (int x = 0; x < width && x < 30; x++) v |= (int)bit(pos + x) << x;
                dg[4 * q + b] = v;
            }
        for (int t = 0; t < 3; t++)
            for (int lane = 0; lane < 64; lane++) {
                int c4[4];
                for (int v = 0; v < 4; v++) c4[v] = dg[16 * t + 4 * (lane / 16) + v];
                memcpy(&T.C0[t][lane], c4, 16);
            }
    }
}

// one group of 16 values: lane l supplies value 16 g + l % 16, limbs 16 kb + 4 (l / 16) + [0, 4)
__device__ __forceinline__ v4i mfma_b_fragment(const u32* __restrict__ t_limbs, size_t value, int kb, int lane) {
    v4i b;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int q = 16 * kb + 4 * (lane / 16) + j;
        const u32 limb = q < 18 ? t_limbs[value * 18 + q] : 0u;
        b[j] = q < 18 ? (int)((limb + 0x00808080u) ^ 0x00808080u) : 0;
    }
    return b;
}

// out[value][36] = the column sums.  One wavefront per 16 values.
__global__ __launch_bounds__(64) void mfma_reduce_kernel(const u32* __restrict__ t_limbs, const MfmaReduceTables* __restrict__ tab, int* __restrict__ out,
                                                         size_t n_values) {
    const int lane = threadIdx.x;
    const size_t g = blockIdx.x;
    const size_t value = 16 * g + lane % 16;
    const bool live = value < n_values;
    const v4i b0 = live ? mfma_b_fragment(t_limbs, value, 0, lane) : v4i{0, 0, 0, 0};
    const v4i b1 = live ? mfma_b_fragment(t_limbs, value, 1, lane) : v4i{0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 3; t++) {
        v4i c = tab->C0[t][lane];
        c = __builtin_amdgcn_mfma_i32_16x16x64_i8(tab->A[t][0][lane], b0, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_i32_16x16x64_i8(tab->A[t][1][lane], b1, c, 0, 0, 0);
        if (live)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int i = 16 * t + 4 * (lane / 16) + v;
                if (i < 36) out[value * 36 + i] = c[v];
            }
    }
}

// the matrix pipe alone: per iteration the 24 MFMAs of one wavefront-squaring (4 groups x 3 row tiles x 2 K blocks)
__global__ __launch_bounds__(256) void mfma_reduce_rate_kernel(const MfmaReduceTables* __restrict__ tab, int* __restrict__ out, unsigned iters) {
    const int lane = threadIdx.x & 63;
    v4i a[3][2];
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int kb = 0; kb < 2; kb++) a[t][kb] = tab->A[t][kb][lane];
    v4i b[4][2];
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
        for (int kb = 0; kb < 2; kb++) b[g][kb] = v4i{(int)threadIdx.x + g, (int)blockIdx.x + kb, 0x01020304 * (g + 1), 0x11 * (kb + 1)};
    v4i acc = {0, 0, 0, 0};
    for (unsigned it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
            for (int t = 0; t < 3; t++) {
                v4i c = acc;
                c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t][0], b[g][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t][1], b[g][1], c, 0, 0, 0);
                b[g][0][t & 3] ^= c[0] & 0x0f0f0f0f;  // the next squaring's digits depend on this one's sums
                acc[t & 3] = c[1] & 1;
            }
    }
    if (acc[0] == 0x7fffffff) out[blockIdx.x * 256 + threadIdx.x] = acc[1] + acc[2] + acc[3] + b[0][0][0];
}

// the vector work that remains per squaring, one lane per value
template <int ILP>
__global__ __launch_bounds__(256) void mfma_valu_rate_kernel(const fe* __restrict__ in, u32* __restrict__ out, unsigned iters) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fe29 x[ILP];
#pragma unroll
    for (int k = 0; k < ILP; k++) {
        x[k] = unpack_reduce29(fe_load(in + (i % 64)));
        x[k].v[0] += (u32)k;
    }
    for (unsigned it = 0; it < iters; it++) {
#pragma unroll
        for (int s = 0; s < ILP; s++) {
            const fe29 l = x[s];
            // (1) the square: 45 multiply-adds
            u64 acc[18];
#pragma unroll
            for (int k = 0; k < 18; k++) acc[k] = 0;
            u32 a2[9];
#pragma unroll
            for (int j = 0; j < 9; j++) a2[j] = l.v[j] << 1;
#pragma unroll
            for (int a = 0; a < 9; a++) {
                acc[2 * a] += (u64)l.v[a] * l.v[a];
#pragma unroll
                for (int j = a + 1; j < 9; j++) acc[a + j] += (u64)l.v[a] * a2[j];
            }
            // (2) carry sweep to 18 normalised limbs = 72 digits, and the signed recoding of each limb
            u32 rec[18];
#pragma unroll
            for (int k = 0; k < 17; k++) {
                acc[k + 1] += acc[k] >> 29;
                rec[k] = (((u32)acc[k] & M29) + 0x00808080u) ^ 0x00808080u;
            }
            rec[17] = ((u32)acc[17] + 0x00808080u) ^ 0x00808080u;
            // (3) [matrix core: 36 column sums per value.  FREE here: stand-ins of the right width taken from live registers]
            int cs[36];
#pragma unroll
            for (int c = 0; c < 36; c++) cs[c] = (int)(rec[(c * 7) % 18] ^ rec[(c * 5 + 3) % 18]) >> 10;
            // (4) limb assembly: digits 4q .. 4q+3 at bit offsets 0, 8, 16, 24 of limb q -- one shift-add and two 64-bit multiply-adds
            long long L[9];
#pragma unroll
            for (int q = 0; q < 9; q++) {
                long long v = (long long)(cs[4 * q] + (cs[4 * q + 1] << 8));
                v += (long long)cs[4 * q + 2] * (1 << 16);
                v += (long long)cs[4 * q + 3] * (1 << 24);
                L[q] = v;
            }
            // (5) signed carry sweep, then the fold of the bits above 2^253 (limb 8 above bit 21): + hi * (2^253 mod p)
            fe29 r;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                L[q + 1] += L[q] >> 29;
                r.v[q] = (u32)L[q] & M29;
            }
            const u32 hi = (u32)(L[8] >> 21);
            r.v[8] = (u32)L[8] & ((1u << 21) - 1);
            u64 c = 0;
#pragma unroll
            for (int q = 0; q < 9; q++) {
                c += (u64)hi * kp29(5, q) + r.v[q];  // stand-in constant of the right shape (a 254-bit multiple of p's limbs)
                r.v[q] = q < 8 ? ((u32)c & M29) : (u32)c;
                c >>= 29;
            }
#pragma unroll
            for (int q = 0; q < 9; q++) r.v[q] &= M29;
            x[s] = r;
        }
    }
    u32 sum = 0;
#pragma unroll
    for (int k = 0; k < ILP; k++)
#pragma unroll
        for (int q = 0; q < 9; q++) sum += x[k].v[q];
    if (sum == 0xfffffff1u) out[i] = sum;  // keeps the chain live
}

static int mfma_tables_device(pk_ctx* ctx, MfmaReduceTables** d_tab) {
    static MfmaReduceTables host;
    static bool built = false;
    if (!built) {
        build_mfma_tables(host);
        built = true;
    }
    int rc = ensure_scratch(ctx, sizeof(MfmaReduceTables) + (1 << 20));
    if (rc) return rc;
    *d_tab = (MfmaReduceTables*)ctx->d_scratch;
    PK_HIP(ctx, hipMemcpyAsync(*d_tab, &host, sizeof host, hipMemcpyHostToDevice, ctx->stream));
    PK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PK_OK;
}

extern "C" {

// d_t_limbs: n x 18 u32 (29-bit limbs of the 522-bit squares), d_out: n x 36 i32 column sums
int pk_probe_mfma_reduce(pk_ctx* ctx, const uint32_t* d_t_limbs, int32_t* d_out, size_t n) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, d_t_limbs && d_out, "null pointer");
    if (!n) return PK_OK;
    MfmaReduceTables* tab = nullptr;
    int rc = mfma_tables_device(ctx, &tab);
    if (rc) return rc;
    mfma_reduce_kernel<<<(unsigned)((n + 15) / 16), 64, 0, ctx->stream>>>(d_t_limbs, tab, d_out, n);
    PK_LAUNCH_CHECK(ctx);
    return sync_stream(ctx);
}

// squarings/s the matrix pipe sustains when fed for free: every wavefront-iteration is 64 squarings' worth of MFMAs
int pk_probe_mfma_reduce_rate(pk_ctx* ctx, unsigned waves_per_simd, unsigned iters, double* squarings_per_s) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, squarings_per_s && waves_per_simd >= 1 && waves_per_simd <= 8 && iters >= 1, "bad argument");
    MfmaReduceTables* tab = nullptr;
    int rc = mfma_tables_device(ctx, &tab);
    if (rc) return rc;
    const unsigned blocks = (unsigned)ctx->num_cus * waves_per_simd;  // 256 threads = 4 wavefronts = one per SIMD
    int* out = (int*)((char*)ctx->d_scratch + sizeof(MfmaReduceTables));
    mfma_reduce_rate_kernel<<<blocks, 256, 0, ctx->stream>>>(tab, out, 8);
    PK_LAUNCH_CHECK(ctx);
    float ms = 0;
    if ((rc = pk_timer_start(ctx))) return rc;
    mfma_reduce_rate_kernel<<<blocks, 256, 0, ctx->stream>>>(tab, out, iters);
    PK_LAUNCH_CHECK(ctx);
    if ((rc = pk_timer_stop(ctx, &ms))) return rc;
    *squarings_per_s = (double)blocks * 256.0 * iters / (ms * 1e-3);
    return PK_OK;
}

int pk_probe_mfma_valu_rate(pk_ctx* ctx, unsigned waves_per_simd, unsigned ilp, unsigned iters, double* squarings_per_s) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, squarings_per_s && waves_per_simd >= 1 && waves_per_simd <= 8 && (ilp == 1 || ilp == 2) && iters >= 1, "bad argument");
    int rc = ensure_scratch(ctx, 1 << 20);
    if (rc) return rc;
    const unsigned blocks = (unsigned)ctx->num_cus * waves_per_simd;
    fe* in = (fe*)ctx->d_scratch;
    u32* out = (u32*)((char*)ctx->d_scratch + 4096);
    PK_HIP(ctx, hipMemsetAsync(in, 0x5a, 64 * 32, ctx->stream));
    auto launch = [&](unsigned n) {
        if (ilp == 1) mfma_valu_rate_kernel<1><<<blocks, 256, 0, ctx->stream>>>(in, out, n);
        else mfma_valu_rate_kernel<2><<<blocks, 256, 0, ctx->stream>>>(in, out, n);
    };
    launch(8);
    PK_LAUNCH_CHECK(ctx);
    float ms = 0;
    if ((rc = pk_timer_start(ctx))) return rc;
    launch(iters);
    PK_LAUNCH_CHECK(ctx);
    if ((rc = pk_timer_stop(ctx, &ms))) return rc;
    *squarings_per_s = (double)blocks * 256.0 * ilp * iters / (ms * 1e-3);
    return PK_OK;
}

}  // extern "C"

// ============================================================================================================================
// VERDICT r03 item 7 -- what a persistent sumcheck kernel could save: the Fiat-Shamir round trip, measured both ways.
//   launch form    (what pk_prove does per round): a one-workgroup kernel that publishes a word to the pinned page, then
//                  hipStreamSynchronize, then the host's sponge work, then the next launch
//   mailbox form   (what a persistent kernel would do): ONE kernel; per round it publishes a word to the pinned page and spins on a
//                  word the host writes back after the same sponge work (system-scope loads over the host link, bounded spin)
// Both run `rounds` dependent round trips with `host_work_permutes` Skyscraper permutations of host work in between (a cubic round
// absorbs four elements and squeezes one: five).  tools/roundtrip.py -> profiles/r04_roundtrip.json.
// ============================================================================================================================
__global__ void roundtrip_launch_kernel(unsigned* host_word, unsigned seq, const unsigned* challenge) {
    if (threadIdx.x == 0) {
        const unsigned c = __hip_atomic_load(challenge, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_word, seq + (c & 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void roundtrip_mailbox_kernel(unsigned* out_word, const unsigned* in_word, unsigned rounds, unsigned max_spin, unsigned* status) {
    if (threadIdx.x != 0) return;
    for (unsigned r = 1; r <= rounds; r++) {
        __hip_atomic_store(out_word, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        unsigned spins = 0;
        while (__hip_atomic_load(in_word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != r) {
            if (++spins > max_spin) {  // the host went away: leave instead of hanging the queue
                __hip_atomic_store(status, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __hip_atomic_store(status, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int pk_probe_roundtrip(pk_ctx* ctx, unsigned rounds, unsigned host_work_permutes, double* us_per_round_launch,
                                     double* us_per_round_mailbox) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, rounds >= 1 && rounds <= 100000 && us_per_round_launch && us_per_round_mailbox, "bad argument");
    int rc = ensure_pinned(ctx);
    if (rc) return rc;
    volatile unsigned* page = (volatile unsigned*)((char*)ctx->h_pinned + 3072);  // a quiet corner of the 4 KiB page
    unsigned* out_word = (unsigned*)page;
    unsigned* in_word = (unsigned*)page + 16;
    unsigned* status = (unsigned*)page + 32;
    fe l = fe_zero(), r = fe_one();
    auto host_work = [&] {
        for (unsigned k = 0; k < host_work_permutes; k++) sky_permute_host(l, r);
    };
    auto now = [] { return std::chrono::steady_clock::now(); };
    // launch form
    page[0] = page[16] = page[32] = 0;
    PK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    auto t0 = now();
    for (unsigned i = 1; i <= rounds; i++) {
        page[16] = i + (l.v[0] & 0u);
        roundtrip_launch_kernel<<<1, 64, 0, ctx->stream>>>(out_word, i, in_word);
        PK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (page[0] != i) return set_err(ctx, PK_ERR_HIP, "round-trip probe: the kernel's word did not arrive");
        host_work();
    }
    *us_per_round_launch = 1e6 * std::chrono::duration<double>(now() - t0).count() / rounds;
    // mailbox form
    page[0] = page[16] = page[32] = 0;
    t0 = now();
    roundtrip_mailbox_kernel<<<1, 64, 0, ctx->stream>>>(out_word, in_word, rounds, 1u << 22, status);
    PK_LAUNCH_CHECK(ctx);
    bool lost = false;
    for (unsigned i = 1; i <= rounds && !lost; i++) {
        auto t1 = now();
        while (__atomic_load_n(page + 0, __ATOMIC_ACQUIRE) != i) {
            if (std::chrono::duration<double>(now() - t1).count() > 2.0) {
                lost = true;
                break;
            }
        }
        host_work();
        __atomic_store_n(page + 16, i + (l.v[0] & 0u), __ATOMIC_RELEASE);
    }
    PK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *us_per_round_mailbox = 1e6 * std::chrono::duration<double>(now() - t0).count() / rounds;
    if (lost || page[32] != 0xffffffffu) return set_err(ctx, PK_ERR_HIP, "round-trip probe: the mailbox kernel gave up at round %u", (unsigned)page[32]);
    return PK_OK;
}

// back-to-back dependent launches on one stream, one synchronisation at the end: microseconds per launch (the in-queue cost of a
// kernel boundary -- dispatch, end-of-kernel release, start-of-kernel acquire -- with no host in the loop)
extern "C" int pk_probe_launch_chain(pk_ctx* ctx, unsigned launches, unsigned threads_per_launch, double* us_per_launch) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, launches >= 1 && launches <= 1000000 && us_per_launch, "bad argument");
    int rc = ensure_pinned(ctx);
    if (rc) return rc;
    unsigned* word = (unsigned*)((char*)ctx->h_pinned + 3072);
    unsigned* chal = word + 16;
    *chal = 0;
    const unsigned blocks = threads_per_launch ? (threads_per_launch + 63) / 64 : 1;
    PK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    auto t0 = std::chrono::steady_clock::now();
    for (unsigned i = 1; i <= launches; i++) roundtrip_launch_kernel<<<blocks, 64, 0, ctx->stream>>>(word, i, chal);
    PK_LAUNCH_CHECK(ctx);
    PK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *us_per_launch = 1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / launches;
    return PK_OK;
}
