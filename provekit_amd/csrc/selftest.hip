// selftest.hip -- host-side execution of the exact __host__ __device__ arithmetic the kernels use
// (fe29.hpp, skyscraper29.hpp), so the CPU test suite can check it against the oracle without a GPU.
#include "ctx.hpp"
#include "reduce.hpp"
#include "fe52.hpp"
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

extern "C" {

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
        if (op < 0 || op > 20) return PK_ERR_BAD_ARG;
        fe r = selftest_op(op, x, y);
        store_host(out + 4 * i, r);
    }
    return PK_OK;
}

}  // extern "C"
