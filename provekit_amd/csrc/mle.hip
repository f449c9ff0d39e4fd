// mle.hip -- multilinear-extension kernels of the Spartan sumcheck and the WHIR rounds
// (SURVEY 8a rows T1, S2, S3, S5, E1, W1, W2, W3).
//
// All of these stream 32-byte field elements with a handful of modular multiplies per
// element, so each is laid out for coalesced 32 B/lane access, keeps its arrays resident
// in HBM across rounds, and returns only the 3-4 field elements the Fiat-Shamir
// transcript needs per round (grid reduction in reduce.hpp).
// memory- / latency-bound kernels: their wavefronts issue ahead of the ALU-bound hash / NTT / grinder kernels they share SIMDs with
#define PK_BASE_PRIO 2
#include "ctx.hpp"
#include "fe29.hpp"
#include "reduce.hpp"

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
inline fe_arg to_arg(const uint64_t* h) {
    fe_arg a;
    memcpy(a.v, h, 32);
    return a;
}

__device__ __forceinline__ fe lds_get(const uint4* lo, const uint4* hi, int i) {
    uint4 l = lo[i], h = hi[i];
    fe x;
    x.v[0] = l.x; x.v[1] = l.y; x.v[2] = l.z; x.v[3] = l.w;
    x.v[4] = h.x; x.v[5] = h.y; x.v[6] = h.z; x.v[7] = h.w;
    return x;
}
__device__ __forceinline__ void lds_put(uint4* lo, uint4* hi, int i, const fe& x) {
    lo[i] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    hi[i] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}

// ---------------------------------------------------------------- T1: evals <-> coeffs
// EvaluationsList::to_coeffs (call sites provekit/prover/src/whir_r1cs.rs:195,198): for every
// variable (index bit) h: v[i | h] -= v[i].  SUB=false gives the inverse (to_evals).
// low kernel: bits [0, LOGT) on a contiguous tile of 2^LOGT elements held in LDS.
template <bool SUB>
__global__ __launch_bounds__(256) void wavelet_low_kernel(const fe* src, fe* data, unsigned logt) {
    PK_LATENCY_PRIO();
    extern __shared__ uint4 lds[];
    const int T = 1 << logt;
    uint4* lo = lds;
    uint4* hi = lds + T;
    fe* base = data + (size_t)blockIdx.x * T;
    const fe* sbase = src + (size_t)blockIdx.x * T;  // src == data for the in-place form
    for (int e = threadIdx.x; e < T; e += 256) {
        const uint4* q = reinterpret_cast<const uint4*>(sbase + e);
        lo[e] = q[0];
        hi[e] = q[1];
    }
    __syncthreads();
    for (unsigned s = 0; s < logt; s++) {
        const int h = 1 << s;
        for (int t = threadIdx.x; t < T / 2; t += 256) {
            int pos = t & (h - 1);
            int i0 = ((t - pos) << 1) + pos, i1 = i0 + h;
            fe a = lds_get(lo, hi, i0), b = lds_get(lo, hi, i1);
            lds_put(lo, hi, i1, SUB ? fe_sub(b, a) : fe_add(b, a));
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < T; e += 256) {
        uint4* q = reinterpret_cast<uint4*>(base + e);
        q[0] = lo[e];
        q[1] = hi[e];
    }
}
// high kernel: bits [s, s+logr): element index = u*2^(s+logr) + r*2^s + v; tile = [2^logr][4 adjacent v]
template <bool SUB, int BT>
__global__ __launch_bounds__(256) void wavelet_high_kernel(fe* __restrict__ data, unsigned s, unsigned logr) {
    PK_LATENCY_PRIO();
    extern __shared__ uint4 lds[];
    const int R = 1 << logr;
    const int TILE = R * BT;
    uint4* lo = lds;
    uint4* hi = lds + TILE;
    const size_t vblocks = ((size_t)1 << s) / BT;
    const size_t u = blockIdx.x / vblocks, vb = blockIdx.x % vblocks;
    fe* base = data + (u << (s + logr)) + vb * BT;
    for (int e = threadIdx.x; e < TILE; e += 256) {
        int b = e % BT, r = e / BT;
        const uint4* q = reinterpret_cast<const uint4*>(base + ((size_t)r << s) + b);
        lo[e] = q[0];
        hi[e] = q[1];
    }
    __syncthreads();
    for (unsigned st = 0; st < logr; st++) {
        const int h = 1 << st;
        for (int t = threadIdx.x; t < (R / 2) * BT; t += 256) {
            int b = t % BT, j = t / BT;
            int pos = j & (h - 1);
            int i0 = ((j - pos) << 1) + pos, i1 = i0 + h;
            fe a = lds_get(lo, hi, i0 * BT + b), c = lds_get(lo, hi, i1 * BT + b);
            lds_put(lo, hi, i1 * BT + b, SUB ? fe_sub(c, a) : fe_add(c, a));
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < TILE; e += 256) {
        int b = e % BT, r = e / BT;
        uint4* q = reinterpret_cast<uint4*>(base + ((size_t)r << s) + b);
        q[0] = lo[e];
        q[1] = hi[e];
    }
}

// d_src == nullptr: in place; otherwise the first sweep reads d_src and writes d_data (saves a 32 B/element copy)
template <bool SUB>
int wavelet(pk_ctx* ctx, const uint64_t* d_src, uint64_t* d_data, unsigned n_vars) {
    PK_REQUIRE(ctx, d_data, "null pointer");
    PK_REQUIRE(ctx, n_vars <= 30, "too many variables");
    fe* D = (fe*)d_data;
    const fe* S = d_src ? (const fe*)d_src : D;
    if (n_vars == 0) {
        if (d_src && d_src != d_data) PK_HIP(ctx, hipMemcpyAsync(D, S, 32, hipMemcpyDeviceToDevice, ctx->stream));
        return PK_OK;
    }
    ProfScope prof(ctx, SUB ? "to_coeffs" : "to_evals");
    unsigned logt = n_vars < 11 ? n_vars : 11;
    size_t lds = ((size_t)2 << logt) * 16;
    PK_HIP(ctx, hipFuncSetAttribute((const void*)wavelet_low_kernel<SUB>, hipFuncAttributeMaxDynamicSharedMemorySize, 1 << 16));
    wavelet_low_kernel<SUB><<<(unsigned)((size_t)1 << (n_vars - logt)), 256, lds, ctx->stream>>>(S, D, logt);
    unsigned s = logt;
    PK_HIP(ctx, hipFuncSetAttribute((const void*)wavelet_high_kernel<SUB, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 1 << 16));
    PK_HIP(ctx, hipFuncSetAttribute((const void*)wavelet_high_kernel<SUB, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 1 << 16));
    while (s < n_vars) {
        // up to 10 index bits per sweep (2^10 rows x 2 adjacent elements = 64 KiB of LDS), 9 with 4-element rows otherwise
        unsigned left = n_vars - s;
        if (left == 10 && s >= 1) {
            size_t tiles = ((size_t)1 << (n_vars - 10)) / 2;
            wavelet_high_kernel<SUB, 2><<<(unsigned)tiles, 256, ((size_t)2 << 10) * 2 * 16, ctx->stream>>>(D, s, 10);
            s += 10;
        } else {
            unsigned logr = left < 9 ? left : 9;
            size_t tiles = ((size_t)1 << (n_vars - logr)) / 4;
            wavelet_high_kernel<SUB, 4><<<(unsigned)tiles, 256, ((size_t)2 << logr) * 4 * 16, ctx->stream>>>(D, s, logr);
            s += logr;
        }
    }
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}

// ---------------------------------------------------------------- S2 / W2: eq tables
// eval_eq (provekit/common/src/utils/sumcheck.rs:146-171): out[i] += scalar * prod_j (bit_j(i) ? x_j : 1-x_j),
// variable 0 <-> most significant index bit.  eq(x, i) = eq_hi(x[0..nhi), i >> nlo) * eq_lo(x[nhi..n), i & mask):
// one block builds each half table (level by level, in global memory), then one streaming kernel
// accumulates all q points into w -- one multiply per (point, element).
// tables layout: [pt][ 2^nhi hi entries | 2^nlo lo entries ]
__global__ __launch_bounds__(256) void eq_half_tables_kernel(const fe* __restrict__ points, const fe* __restrict__ scales,
                                                             unsigned n_vars, unsigned nhi, unsigned nlo, fe* __restrict__ tables) {
    PK_LATENCY_PRIO();
    // points / scales sit in pinned host memory (the mailbox): fetch this half's <= 15 coordinates in one go
    __shared__ uint4 xs[2 * 16];
    const unsigned pt = blockIdx.x, half = blockIdx.y;
    const size_t per_pt = ((size_t)1 << nhi) + ((size_t)1 << nlo);
    fe* T = tables + pt * per_pt + (half ? ((size_t)1 << nhi) : 0);
    const fe* x = points + (size_t)pt * n_vars + (half ? nhi : 0);
    const unsigned nv = half ? nlo : nhi;
    if (threadIdx.x < nv) lds_put(xs, xs + 16, threadIdx.x, fe_load(x + threadIdx.x));
    if (threadIdx.x == 32) fe_store(T, half ? fe_one() : fe_load(scales + pt));
    __threadfence_block();
    __syncthreads();
    // Level l appends variable nv-1-l as index bit l, so the last variable is the LSB and variable 0
    // ends up as the MSB, as eval_eq's recursion orders it: T[h+i] = x*T[i] (s1), T[i] -= that (s0).
    for (unsigned l = 0; l < nv; l++) {
        const size_t h = (size_t)1 << l;
        fe xv = lds_get(xs, xs + 16, nv - 1 - l);
        for (size_t i = threadIdx.x; i < h; i += 256) {
            fe t = fe_load(T + i);
            fe up = fe_mulx(t, xv);  // s1 = s * x
            fe_store(T + h + i, up);
            fe_store(T + i, fe_sub(t, up));  // s0 = s - s1
        }
        __threadfence_block();
        __syncthreads();
    }
}
// Two adjacent outputs per lane (they share the high half-table entry) and the q products of each output summed with one
// Montgomery reduction per DOT29_GROUP terms (fe29.hpp dot29): ~100 multiply-adds per (point, element) instead of 162+.
__global__ __launch_bounds__(256) void eq_accumulate_kernel(fe* __restrict__ w, size_t n, unsigned nhi, unsigned nlo, unsigned q,
                                                            const fe* __restrict__ tables, int overwrite) {
    PK_LATENCY_PRIO();
    const size_t per_pt = ((size_t)1 << nhi) + ((size_t)1 << nlo);
    const size_t mask = ((size_t)1 << nlo) - 1;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if (nlo == 0) {  // a single low entry per point: one output per lane
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            dot29 d;
            dot29_init(d);
            const fe* T = tables;
            for (unsigned pt = 0; pt < q; pt++, T += per_pt)
                dot29_add(d, unpack29<0>(fe_load(T + (i >> nlo))), unpack29<5>(fe_load(T + ((size_t)1 << nhi) + (i & mask))));
            fe r = dot29_result(d);
            fe_store(w + i, overwrite ? r : fe_add(fe_load(w + i), r));
        }
        return;
    }
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n / 2; j += stride) {
        const size_t i = 2 * j;  // i and i + 1 differ in the lowest index bit only: same high entry
        dot29 d0, d1;
        dot29_init(d0);
        dot29_init(d1);
        const fe* T = tables;
        for (unsigned pt = 0; pt < q; pt++, T += per_pt) {
            const fe29 h = unpack29<0>(fe_load(T + (i >> nlo)));
            const fe* lo = T + ((size_t)1 << nhi) + (i & mask);
            dot29_add(d0, h, unpack29<5>(fe_load(lo)));
            dot29_add(d1, h, unpack29<5>(fe_load(lo + 1)));
        }
        fe r0 = dot29_result(d0), r1 = dot29_result(d1);
        fe_store(w + i, overwrite ? r0 : fe_add(fe_load(w + i), r0));
        fe_store(w + i + 1, overwrite ? r1 : fe_add(fe_load(w + i + 1), r1));
    }
}

// ---------------------------------------------------------------- S3: cubic sumcheck round
// sumcheck_fold_map_reduce::<4,3> (provekit/common/src/utils/sumcheck.rs:16-104) with the map of
// provekit/prover/src/whir_r1cs.rs:284-291.
template <bool FOLD>
__global__ __launch_bounds__(RED_THREADS) void sumcheck_cubic_kernel(fe* __restrict__ a, fe* __restrict__ b, fe* __restrict__ c,
                                                                     fe* __restrict__ eq, size_t len, fe_arg fold_arg, gate_args gate,
                                                                     fe* __restrict__ partials, unsigned* __restrict__ ticket,
                                                                     fe* __restrict__ result, unsigned seq) {
    PK_LATENCY_PRIO();
    __shared__ uint4 smem[3 * 16];
    const fe alpha = (FOLD && gate.host) ? gate_wait(gate) : from_arg(fold_arg);
    const size_t npairs = FOLD ? len / 4 : len / 2;
    const size_t off = npairs;          // partner of i is i + off (quarter 1 after folding, or the upper half)
    const size_t foff = len / 2;        // fold partner: p2 = p0 + len/2
    fe acc[3] = {fe_zero(), fe_zero(), fe_zero()};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npairs; i += stride) {
        fe v[4][2];
        fe* arr[4] = {a, b, c, eq};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            fe x0 = fe_load(arr[k] + i), x1 = fe_load(arr[k] + i + off);
            if (FOLD) {  // sumcheck.rs:95-96: p0 += fold*(p2-p0); p1 += fold*(p3-p1)
                fe x2 = fe_load(arr[k] + i + foff), x3 = fe_load(arr[k] + i + off + foff);
                x0 = fe_add(x0, fe_mulx(alpha, fe_sub(x2, x0)));
                x1 = fe_add(x1, fe_mulx(alpha, fe_sub(x3, x1)));
                fe_store(arr[k] + i, x0);
                fe_store(arr[k] + i + off, x1);
            }
            v[k][0] = x0;
            v[k][1] = x1;
        }
        const fe &a0 = v[0][0], &a1 = v[0][1], &b0 = v[1][0], &b1 = v[1][1], &c0 = v[2][0], &c1 = v[2][1], &e0 = v[3][0], &e1 = v[3][1];
        // f0 = eq0 * (a0*b0 - c0)
        acc[0] = fe_add(acc[0], fe_mulx(e0, fe_sub(fe_mulx(a0, b0), c0)));
        // f(-1) = (2eq0-eq1) * ((2a0-a1)(2b0-b1) - (2c0-c1))
        fe ta = fe_sub(fe_dbl(a0), a1), tb = fe_sub(fe_dbl(b0), b1), tc = fe_sub(fe_dbl(c0), c1), te = fe_sub(fe_dbl(e0), e1);
        acc[1] = fe_add(acc[1], fe_mulx(te, fe_sub(fe_mulx(ta, tb), tc)));
        // f_inf = (eq1-eq0)(a1-a0)(b1-b0)
        acc[2] = fe_add(acc[2], fe_mulx(fe_mulx(fe_sub(e1, e0), fe_sub(a1, a0)), fe_sub(b1, b0)));
    }
    grid_finish_fe<3>(acc, smem, partials, ticket, result, seq);
}

// ---------------------------------------------------------------- W3: quadratic sumcheck round
// adjacent pairs (2i, 2i+1): h(0)=sum f0 w0, h(1)=sum f1 w1, h(2)=sum (2f1-f0)(2w1-w0)
// (recursive-verifier/app/circuit/whir_utilities.go:102-125; utilities.go:148-154).
// FOLD: first v'[i] = v[2i] + r (v[2i+1]-v[2i]) written to the *_out arrays (out-of-place).
template <bool FOLD>
__global__ __launch_bounds__(RED_THREADS) void sumcheck_quadratic_kernel(const fe* __restrict__ f, const fe* __restrict__ w,
                                                                         size_t out_len, fe_arg fold_arg, gate_args gate, fe* __restrict__ f_out,
                                                                         fe* __restrict__ w_out, fe* __restrict__ partials,
                                                                         unsigned* __restrict__ ticket, fe* __restrict__ result, unsigned seq) {
    PK_LATENCY_PRIO();
    __shared__ uint4 smem[3 * 16];
    const fe r = (FOLD && gate.host) ? gate_wait(gate) : from_arg(fold_arg);
    fe acc[3] = {fe_zero(), fe_zero(), fe_zero()};
    const size_t npairs = out_len / 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npairs; i += stride) {
        fe f0, f1, w0, w1;
        if (FOLD) {
            fe x0 = fe_load(f + 4 * i), x1 = fe_load(f + 4 * i + 1), x2 = fe_load(f + 4 * i + 2), x3 = fe_load(f + 4 * i + 3);
            f0 = fe_add(x0, fe_mulx(r, fe_sub(x1, x0)));
            f1 = fe_add(x2, fe_mulx(r, fe_sub(x3, x2)));
            fe y0 = fe_load(w + 4 * i), y1 = fe_load(w + 4 * i + 1), y2 = fe_load(w + 4 * i + 2), y3 = fe_load(w + 4 * i + 3);
            w0 = fe_add(y0, fe_mulx(r, fe_sub(y1, y0)));
            w1 = fe_add(y2, fe_mulx(r, fe_sub(y3, y2)));
            fe_store(f_out + 2 * i, f0);
            fe_store(f_out + 2 * i + 1, f1);
            fe_store(w_out + 2 * i, w0);
            fe_store(w_out + 2 * i + 1, w1);
        } else {
            f0 = fe_load(f + 2 * i);
            f1 = fe_load(f + 2 * i + 1);
            w0 = fe_load(w + 2 * i);
            w1 = fe_load(w + 2 * i + 1);
        }
        acc[0] = fe_add(acc[0], fe_mulx(f0, w0));
        acc[1] = fe_add(acc[1], fe_mulx(f1, w1));
        acc[2] = fe_add(acc[2], fe_mulx(fe_sub(fe_dbl(f1), f0), fe_sub(fe_dbl(w1), w0)));
    }
    grid_finish_fe<3>(acc, smem, partials, ticket, result, seq);
}
// ---- small rounds: the work of ONE pair spread over several lanes ------------------------------------------------------
// Late sumcheck rounds have a handful of pairs; with a lane per pair the round is a chain of 16 (cubic, folding) or 7
// (quadratic, folding) dependent modular multiplications on one lane of one wavefront -- ~1 us each -- and the kernel holds a
// hardware queue for that long.  Here the independent products of a pair go to neighbouring lanes (8 resp. 4 per pair),
// the folded values are exchanged through LDS, and three lanes form the three evaluations: 3 resp. 2 multiplications deep.
// Same arithmetic, same order of the (exact) field operations per value: results are bit-identical to the wide kernels.
constexpr size_t SMALL_ROUND_PAIRS = 16384;

__global__ __launch_bounds__(RED_THREADS) void sumcheck_cubic_small_kernel(fe* __restrict__ a, fe* __restrict__ b, fe* __restrict__ c,
                                                                           fe* __restrict__ eq, size_t len, fe_arg fold_arg, gate_args gate,
                                                                           fe* __restrict__ partials, unsigned* __restrict__ ticket,
                                                                           fe* __restrict__ result, unsigned seq) {
    PK_LATENCY_PRIO();
    __shared__ uint4 smem[3 * 16];
    __shared__ uint4 xs[2 * RED_THREADS];  // folded value of lane t at [t] (lo) and [RED_THREADS + t] (hi)
    const fe alpha = gate.host ? gate_wait(gate) : from_arg(fold_arg);
    const size_t npairs = len / 4, off = npairs, foff = len / 2;
    const unsigned tid = threadIdx.x, g = tid >> 3, role = tid & 7, k = role >> 1, which = role & 1;
    const size_t i = (size_t)blockIdx.x * (RED_THREADS / 8) + g;
    fe* arr = k == 0 ? a : k == 1 ? b : k == 2 ? c : eq;
    if (i < npairs) {  // sumcheck.rs:95-96: p0 += fold*(p2-p0); p1 += fold*(p3-p1)
        fe* p = arr + i + (which ? off : 0);
        fe x = fe_load(p), x2 = fe_load(p + foff);
        x = fe_add(x, fe_mulx(alpha, fe_sub(x2, x)));
        fe_store(p, x);
        lds_put(xs, xs + RED_THREADS, tid, x);
    }
    __syncthreads();
    fe acc[3] = {fe_zero(), fe_zero(), fe_zero()};
    if (i < npairs && role < 3) {
        const unsigned base = tid & ~7u;
        auto v = [&](unsigned kk, unsigned w) { return lds_get(xs, xs + RED_THREADS, base + 2 * kk + w); };
        const fe a0 = v(0, 0), a1 = v(0, 1), b0 = v(1, 0), b1 = v(1, 1), c0 = v(2, 0), c1 = v(2, 1), e0 = v(3, 0), e1 = v(3, 1);
        // the three evaluations share one shape, X * (Y*Z - S), so the lanes of a pair run the same two multiplications on
        // operands selected by role (no divergent branches):
        //   role 0: f0    = eq0 * (a0*b0 - c0)
        //   role 1: f(-1) = (2eq0-eq1) * ((2a0-a1)(2b0-b1) - (2c0-c1))
        //   role 2: f_inf = (b1-b0) * ((eq1-eq0)(a1-a0) - 0)
        auto pick = [&](const fe& x0, const fe& x1) {
            fe m1 = fe_sub(fe_dbl(x0), x1), d = fe_sub(x1, x0), r;
#pragma unroll
            for (int w = 0; w < 8; w++) r.v[w] = role == 0 ? x0.v[w] : role == 1 ? m1.v[w] : d.v[w];
            return r;
        };
        const fe pa = pick(a0, a1), pb = pick(b0, b1), pc = pick(c0, c1), pe = pick(e0, e1);
        fe X, Y, S;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            X.v[w] = role == 2 ? pb.v[w] : pe.v[w];
            Y.v[w] = role == 2 ? pe.v[w] : pb.v[w];
            S.v[w] = role == 2 ? 0u : pc.v[w];
        }
        const fe t = fe_mulx(X, fe_sub(fe_mulx(Y, pa), S));
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int w = 0; w < 8; w++) acc[r].v[w] = role == (unsigned)r ? t.v[w] : 0u;
    }
    grid_finish_fe<3>(acc, smem, partials, ticket, result, seq);
}

template <bool FOLD>
__global__ __launch_bounds__(RED_THREADS) void sumcheck_quadratic_small_kernel(const fe* __restrict__ f, const fe* __restrict__ w, size_t out_len,
                                                                               fe_arg fold_arg, gate_args gate, fe* __restrict__ f_out, fe* __restrict__ w_out,
                                                                               fe* __restrict__ partials, unsigned* __restrict__ ticket,
                                                                               fe* __restrict__ result, unsigned seq) {
    PK_LATENCY_PRIO();
    __shared__ uint4 smem[3 * 16];
    __shared__ uint4 xs[2 * RED_THREADS];
    const fe r = (FOLD && gate.host) ? gate_wait(gate) : from_arg(fold_arg);
    const size_t npairs = out_len / 2;
    const unsigned tid = threadIdx.x, g = tid >> 2, role = tid & 3;  // role: 0 f0, 1 f1, 2 w0, 3 w1
    const size_t i = (size_t)blockIdx.x * (RED_THREADS / 4) + g;
    if (i < npairs) {
        const fe* src = role < 2 ? f : w;
        const unsigned j = role & 1;
        fe x;
        if (FOLD) {
            fe x0 = fe_load(src + 4 * i + 2 * j), x1 = fe_load(src + 4 * i + 2 * j + 1);
            x = fe_add(x0, fe_mulx(r, fe_sub(x1, x0)));
            fe_store((role < 2 ? f_out : w_out) + 2 * i + j, x);
        } else {
            x = fe_load(src + 2 * i + j);
        }
        lds_put(xs, xs + RED_THREADS, tid, x);
    }
    __syncthreads();
    fe acc[3] = {fe_zero(), fe_zero(), fe_zero()};
    if (i < npairs && role < 3) {
        const unsigned base = tid & ~3u;
        const fe f0 = lds_get(xs, xs + RED_THREADS, base), f1 = lds_get(xs, xs + RED_THREADS, base + 1);
        const fe w0 = lds_get(xs, xs + RED_THREADS, base + 2), w1 = lds_get(xs, xs + RED_THREADS, base + 3);
        // one multiplication on operands selected by role: h(0) = f0 w0, h(1) = f1 w1, h(2) = (2f1-f0)(2w1-w0)
        const fe tf = fe_sub(fe_dbl(f1), f0), tw = fe_sub(fe_dbl(w1), w0);
        fe X, Y;
#pragma unroll
        for (int v = 0; v < 8; v++) {
            X.v[v] = role == 0 ? f0.v[v] : role == 1 ? f1.v[v] : tf.v[v];
            Y.v[v] = role == 0 ? w0.v[v] : role == 1 ? w1.v[v] : tw.v[v];
        }
        const fe t = fe_mulx(X, Y);
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int v = 0; v < 8; v++) acc[q].v[v] = role == (unsigned)q ? t.v[v] : 0u;
    }
    grid_finish_fe<3>(acc, smem, partials, ticket, result, seq);
}

// the single-element tail of the fold (out_len == 1): v'[0] = v[0] + r (v[1]-v[0]); no pair to sum.  blockIdx.y selects
// one of up to two arrays folded by the same challenge (the sumcheck's polynomial and its weights) in one launch.
__global__ void fold_pairs_kernel(const fe* __restrict__ v0, fe* __restrict__ out0, const fe* __restrict__ v1, fe* __restrict__ out1,
                                  size_t out_len, fe_arg r_arg, gate_args gate) {
    PK_LATENCY_PRIO();
    const fe r = gate.host ? gate_wait(gate) : from_arg(r_arg);
    const fe* v = blockIdx.y ? v1 : v0;
    fe* out = blockIdx.y ? out1 : out0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < out_len; i += stride) {
        fe x0 = fe_load(v + 2 * i), x1 = fe_load(v + 2 * i + 1);
        fe_store(out + i, fe_add(x0, fe_mulx(r, fe_sub(x1, x0))));
    }
}

// ---------------------------------------------------------------- S5: dot product(s)
// <w,f> and optionally <w,g> in one pass over w (the statement needs both sums, whir_r1cs.rs:401-405)
template <int NV>
__global__ __launch_bounds__(RED_THREADS) void dot_kernel(const fe* __restrict__ w, const fe* __restrict__ f, const fe* __restrict__ g,
                                                          size_t n, fe* __restrict__ partials, unsigned* __restrict__ ticket,
                                                          fe* __restrict__ result, unsigned seq) {
    PK_LATENCY_PRIO();
    __shared__ uint4 smem[NV * 16];
    fe acc[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) acc[k] = fe_zero();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        fe wi = fe_load(w + i);
        acc[0] = fe_add(acc[0], fe_mulx(wi, fe_load(f + i)));
        if (NV == 2) acc[NV - 1] = fe_add(acc[NV - 1], fe_mulx(wi, fe_load(g + i)));
    }
    grid_finish_fe<NV>(acc, smem, partials, ticket, result, seq);
}

// NR weight rows (row k at w + k * row_stride) against f (and g): <w_k, f>, <w_k, g> for every k in ONE pass over f and g -- the
// three statement weights of whir_r1cs.rs:382-412 share their polynomials, so this is one launch and one round trip instead of three
template <int NR, int NV>
__global__ __launch_bounds__(RED_THREADS) void dot_rows_kernel(const fe* __restrict__ w, size_t row_stride, const fe* __restrict__ f,
                                                               const fe* __restrict__ g, size_t n, fe* __restrict__ partials,
                                                               unsigned* __restrict__ ticket, fe* __restrict__ result, unsigned seq) {
    PK_LATENCY_PRIO();
    __shared__ uint4 smem[NR * NV * 16];
    fe acc[NR * NV];
#pragma unroll
    for (int k = 0; k < NR * NV; k++) acc[k] = fe_zero();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const fe fi = fe_load(f + i);
        fe gi = fi;
        if (NV == 2) gi = fe_load(g + i);
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const fe wi = fe_load(w + (size_t)k * row_stride + i);
            acc[NV * k] = fe_add(acc[NV * k], fe_mulx(wi, fi));
            if (NV == 2) acc[NV * k + 1] = fe_add(acc[NV * k + 1], fe_mulx(wi, gi));
        }
    }
    grid_finish_fe<NR * NV>(acc, smem, partials, ticket, result, seq);
}

// ---------------------------------------------------------------- E1: univariate evaluation
// sum_i c[i] z^i.  Workgroup b owns the contiguous segment [b S, (b + 1) S), S = 256 cnt; lane t Horner-evaluates its stride-256
// subsequence c[b S + t], c[b S + t + 256], ... in z^256 (every load coalesced across the wave), scales by z^t -- one product with an
// entry of a 256-entry table a tiny kernel builds once per point -- and by z^(b S), which lane 0 computes for the whole workgroup while the
// others run their loops; the grid reduces.  cnt + 2 products per lane.  (Until round 5 the subsequences were strided over the whole grid
// and every lane raised z to its own global index: 8 select-multiplications for the lane bits + up to 10 for the block bits on top of the
// cnt = 8 of its loop -- most of the kernel's arithmetic was exponentiation.)
struct pow2_args {
    fe_arg p[28];  // z^(2^i)
};
__global__ __launch_bounds__(RED_THREADS) void zpow_table_kernel(pow2_args zp, fe* __restrict__ ztab) {  // ztab[t] = z^t, t < 256
    PK_LATENCY_PRIO();
    fe h = fe_one();
#pragma unroll 1
    for (int i = 0; i < 8; i++)
        if ((threadIdx.x >> i) & 1u) h = fe_mulx(h, from_arg(zp.p[i]));
    fe_store(ztab + threadIdx.x, h);
}
// NP polynomials of the same length at the same point in one launch (a batch commitment's OOD answers, mtUtilities.go:51-76)
template <int NP>
__global__ __launch_bounds__(RED_THREADS) void horner_kernel(const fe* __restrict__ c, const fe* __restrict__ c_second, size_t n, size_t cnt, pow2_args zp,
                                                             const fe* __restrict__ ztab, fe* __restrict__ partials, unsigned* __restrict__ ticket,
                                                             fe* __restrict__ result, unsigned seq) {
    PK_LATENCY_PRIO();
    __shared__ uint4 smem[NP * 16];
    __shared__ unsigned s_zb[8];
    static_assert(RED_THREADS == 256, "the lane table has 256 entries");
    const size_t base = (size_t)blockIdx.x * cnt * RED_THREADS;
    if (threadIdx.x == 0) {  // z^(b S) for the workgroup: the set bits of b S select entries of the host's z^(2^i) table
        fe zb = fe_one();
        bool any = false;
        for (int i = 8; i < 28; i++)
            if ((base >> i) & 1u) {
                zb = any ? fe_mulx(zb, from_arg(zp.p[i])) : from_arg(zp.p[i]);
                any = true;
            }
#pragma unroll
        for (int k = 0; k < 8; k++) s_zb[k] = zb.v[k];
    }
    const fe z256 = from_arg(zp.p[8]);
    const size_t first = base + threadIdx.x;
    fe acc[NP];
#pragma unroll
    for (int q = 0; q < NP; q++) acc[q] = fe_zero();
    size_t mine = 0;  // how many of this lane's cnt elements exist
    if (first < n) {
        mine = (n - first + RED_THREADS - 1) / RED_THREADS;
        if (mine > cnt) mine = cnt;
    }
    if (mine) {
        const fe zt = fe_load(ztab + threadIdx.x);
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const fe* cq = q == 0 ? c : c_second;
            fe h = fe_load(cq + first + (mine - 1) * RED_THREADS);
            for (size_t j = mine - 1; j-- > 0;) h = fe_add(fe_mulx(h, z256), fe_load(cq + first + j * RED_THREADS));
            acc[q] = fe_mulx(h, zt);
        }
    }
    __syncthreads();
    if (mine && base) {
        fe zb;
#pragma unroll
        for (int k = 0; k < 8; k++) zb.v[k] = s_zb[k];
#pragma unroll
        for (int q = 0; q < NP; q++) acc[q] = fe_mulx(acc[q], zb);
    }
    grid_finish_fe<NP>(acc, smem, partials, ticket, result, seq);
}

// ---------------------------------------------------------------- W1: coefficient fold
// out[t] = sum_j c[2^k t + j] * prod_b r_b^bit_b(j)   (MultivarPoly, utilities.go:15-22: r[0] <-> bit 0)
struct fold_args {
    fe_arg r[8];
};
__global__ __launch_bounds__(256) void fold_coeffs_kernel(const fe* __restrict__ c, size_t n_out, unsigned k, fold_args ra,
                                                          fe* __restrict__ out) {
    PK_LATENCY_PRIO();
    __shared__ uint4 wts[256 * 2];
    // weight j = prod_b r_b^{bit_b(j)}: thread j builds its own with at most k multiplications (no serial doubling pass)
    if (threadIdx.x < (1u << k)) {
        fe w = fe_one();
        for (unsigned b = 0; b < k; b++)
            if ((threadIdx.x >> b) & 1u) w = fe_mulx(w, from_arg(ra.r[b]));
        lds_put(wts, wts + 256, threadIdx.x, w);
    }
    __syncthreads();
    const int fw = 1 << k;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_out; t += stride) {
        fe acc = fe_load(c + t * fw);
        for (int j = 1; j < fw; j++) acc = fe_add(acc, fe_mulx(fe_load(c + t * fw + j), lds_get(wts, wts + 256, j)));
        fe_store(out + t, acc);
    }
}

// the WHIR fold (k = 4) with the 16 weights formed on the host (11 products) and passed by value: no per-workgroup weight
// phase.  LANES = 1: one output per lane.  LANES = 16: late rounds with few outputs -- the 16 products of an output go to
// 16 neighbouring lanes and are summed as limb sums (reduce.hpp), one multiplication deep instead of fifteen.
struct fold16_args {
    fe_arg w[16];
};
template <int LANES>
__global__ __launch_bounds__(256) void fold16_kernel(const fe* __restrict__ c, size_t n_out, fold16_args wa, fe* __restrict__ out) {
    PK_LATENCY_PRIO();
    __shared__ uint4 wts[16 * 2];
    if (threadIdx.x < 16) lds_put(wts, wts + 16, threadIdx.x, from_arg(wa.w[threadIdx.x]));
    __syncthreads();
    if (LANES == 1) {
        const size_t stride = (size_t)gridDim.x * blockDim.x;
        for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_out; t += stride) {
            fe acc = fe_load(c + t * 16);
#pragma unroll 1
            for (int j = 1; j < 16; j++) acc = fe_add(acc, fe_mulx(fe_load(c + t * 16 + j), lds_get(wts, wts + 16, j)));
            fe_store(out + t, acc);
        }
    } else {
        const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x, t = g >> 4;
        const unsigned j = threadIdx.x & 15;
        wide w = wide_zero();
        if (t < n_out) wide_add_fe(w, fe_mulx(fe_load(c + g), lds_get(wts, wts + 16, j)));  // c[16 t + j] * w_j (w_0 = 1)
#pragma unroll
        for (unsigned off = 8; off >= 1; off >>= 1)
#pragma unroll
            for (int i = 0; i < 8; i++) w.l[i] += shfl_down_u64(w.l[i], off);
        if (j == 0 && t < n_out) fe_store(out + t, wide_reduce(w));
    }
}

// out = a + beta * b (out may alias neither): the batching combination of two committed polynomials in one pass
__global__ __launch_bounds__(256) void lincomb_kernel(fe* __restrict__ out, const fe* __restrict__ a, const fe* __restrict__ b, size_t n,
                                                      fe_arg beta_arg) {
    PK_LATENCY_PRIO();
    const fe beta = from_arg(beta_arg);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        fe_store(out + i, fe_add(fe_load(a + i), fe_mulx(beta, fe_load(b + i))));
}
// y += beta * x ; y = a o b
__global__ __launch_bounds__(256) void axpy_kernel(fe* __restrict__ y, const fe* __restrict__ x, size_t n, fe_arg beta_arg) {
    PK_LATENCY_PRIO();
    const fe beta = from_arg(beta_arg);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        fe_store(y + i, fe_add(fe_load(y + i), fe_mulx(beta, fe_load(x + i))));
}

}  // namespace

extern "C" {

int pk_to_coeffs(pk_ctx* ctx, uint64_t* d_evals, unsigned n_vars) {
    PK_ENTER(ctx);
    return wavelet<true>(ctx, nullptr, d_evals, n_vars);
}
int pk_to_evals(pk_ctx* ctx, uint64_t* d_coeffs, unsigned n_vars) {
    PK_ENTER(ctx);
    return wavelet<false>(ctx, nullptr, d_coeffs, n_vars);
}
// out-of-place forms: d_dst receives the transform of d_src (which is left untouched)
int pk_to_coeffs_into(pk_ctx* ctx, const uint64_t* d_src, uint64_t* d_dst, unsigned n_vars) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, d_src && d_src != d_dst, "source must differ from destination");
    return wavelet<true>(ctx, d_src, d_dst, n_vars);
}
int pk_to_evals_into(pk_ctx* ctx, const uint64_t* d_src, uint64_t* d_dst, unsigned n_vars) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, d_src && d_src != d_dst, "source must differ from destination");
    return wavelet<false>(ctx, d_src, d_dst, n_vars);
}

int pk_eq_accumulate(pk_ctx* ctx, uint64_t* d_w, unsigned n_vars, const uint64_t* points, const uint64_t* scales, unsigned q,
                     int overwrite) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, d_w && (q == 0 || (points && scales)), "null pointer");
    PK_REQUIRE(ctx, n_vars <= 30, "too many variables");
    const size_t n = (size_t)1 << n_vars;
    if (q == 0) {
        if (overwrite) PK_HIP(ctx, hipMemsetAsync(d_w, 0, 32 * n, ctx->stream));
        return PK_OK;
    }
    const unsigned nlo = (n_vars + 1) / 2, nhi = n_vars - nlo;
    const size_t per_pt = ((size_t)1 << nhi) + ((size_t)1 << nlo);
    // tables use the context workspace; stream order keeps them clear of earlier kernels (NTT scratch).  The points and
    // scales go through the pinned mailbox, which the table kernel reads directly (no copy operation).
    int rc = ensure_ws(ctx, 32 * (size_t)q * per_pt + 64);
    if (rc) return rc;
    PK_REQUIRE(ctx, nlo <= 16, "too many variables");
    char* mail = nullptr;
    rc = mail_alloc(ctx, 32 * ((size_t)q * n_vars + q), (void**)&mail);
    if (rc) return rc;
    fe* d_points = (fe*)mail;
    fe* d_scales = d_points + (size_t)q * n_vars;
    fe* d_tables = (fe*)ctx->d_ws;
    if (n_vars) memcpy(d_points, points, 32 * (size_t)q * n_vars);
    memcpy(d_scales, scales, 32 * (size_t)q);
    {
        ProfScope prof(ctx, "eq_accumulate");
        eq_half_tables_kernel<<<dim3(q, 2), 256, 0, ctx->stream>>>(d_points, d_scales, n_vars, nhi, nlo, d_tables);
        eq_accumulate_kernel<<<grid_for(ctx, nlo ? (n + 1) / 2 : n, 256), 256, 0, ctx->stream>>>((fe*)d_w, n, nhi, nlo, q, d_tables, overwrite);
    }
    PK_LAUNCH_CHECK(ctx);
    // the caller's host arrays were copied into the mailbox above: they may be reused on return
    return PK_OK;
}

int pk_eq_table(pk_ctx* ctx, const uint64_t* r, unsigned m, uint64_t* d_out) {
    PK_ENTER(ctx);
    static const uint64_t one[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
    return pk_eq_accumulate(ctx, d_out, m, r, one, 1, 1);
}

}  // extern "C"
namespace pk {
// launch only: round results go to the pinned page under sequence number *red_seq_out.  gate_seq != 0 (latency mode, folding rounds
// only): the folding challenge is not known yet -- the kernel waits for gate_publish(ctx, gate_seq, challenge) (reduce.hpp)
int sumcheck_cubic_launch(pk_ctx* ctx, uint64_t* d_a, uint64_t* d_b, uint64_t* d_c, uint64_t* d_eq, size_t len, const uint64_t* fold_or_null,
                          unsigned gate_seq, unsigned* red_seq_out) {
    PK_REQUIRE(ctx, d_a && d_b && d_c && d_eq && red_seq_out, "null pointer");
    PK_REQUIRE(ctx, is_pow2(len) && len >= 2, "size must be a power of two >= 2");  // sumcheck.rs:22-23
    const bool fold = fold_or_null || gate_seq;
    PK_REQUIRE(ctx, !fold || len >= 4, "size must be >= 4 when folding");    // sumcheck.rs:27
    int rc = reduction_scratch(ctx);
    if (rc) return rc;
    size_t npairs = fold ? len / 4 : len / 2;
    unsigned blocks = reduction_blocks(ctx, npairs);
    const gate_args gate = gate_seq ? gate_for(ctx, gate_seq) : gate_none();
    const fe_arg farg = fold_or_null ? to_arg(fold_or_null) : fe_arg{};
    const unsigned seq = next_seq(ctx);
    *red_seq_out = seq;
    {
        ProfScope prof(ctx, "sumcheck_cubic");
        if (fold && npairs <= SMALL_ROUND_PAIRS)
            sumcheck_cubic_small_kernel<<<(unsigned)((npairs + RED_THREADS / 8 - 1) / (RED_THREADS / 8)), RED_THREADS, 0, ctx->stream>>>(
                (fe*)d_a, (fe*)d_b, (fe*)d_c, (fe*)d_eq, len, farg, gate, red_partials(ctx), red_ticket(ctx), red_result(ctx), seq);
        else if (fold)
            sumcheck_cubic_kernel<true><<<blocks, RED_THREADS, 0, ctx->stream>>>((fe*)d_a, (fe*)d_b, (fe*)d_c, (fe*)d_eq, len, farg, gate,
                                                                                  red_partials(ctx), red_ticket(ctx), red_result(ctx), seq);
        else
            sumcheck_cubic_kernel<false><<<blocks, RED_THREADS, 0, ctx->stream>>>((fe*)d_a, (fe*)d_b, (fe*)d_c, (fe*)d_eq, len, fe_arg{}, gate_none(),
                                                                                   red_partials(ctx), red_ticket(ctx), red_result(ctx), seq);
    }
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}
int sumcheck_quadratic_launch(pk_ctx* ctx, const uint64_t* d_f, const uint64_t* d_w, size_t len, const uint64_t* fold_or_null, unsigned gate_seq,
                              uint64_t* d_f_out, uint64_t* d_w_out, unsigned* red_seq_out) {
    PK_REQUIRE(ctx, d_f && d_w && red_seq_out, "null pointer");
    PK_REQUIRE(ctx, is_pow2(len), "size must be a power of two");
    const bool fold = fold_or_null || gate_seq;
    size_t out_len = fold ? len / 2 : len;
    PK_REQUIRE(ctx, out_len >= 2, "at least one pair is needed after folding");
    PK_REQUIRE(ctx, !fold || (d_f_out && d_w_out && d_f_out != d_f && d_w_out != d_w), "folding is out-of-place");
    int rc = reduction_scratch(ctx);
    if (rc) return rc;
    unsigned blocks = reduction_blocks(ctx, out_len / 2);
    const gate_args gate = gate_seq ? gate_for(ctx, gate_seq) : gate_none();
    const fe_arg farg = fold_or_null ? to_arg(fold_or_null) : fe_arg{};
    const unsigned seq = next_seq(ctx);
    *red_seq_out = seq;
    {
        ProfScope prof(ctx, "sumcheck_quadratic");
        const size_t npairs = out_len / 2;
        const unsigned sblocks = (unsigned)((npairs + RED_THREADS / 4 - 1) / (RED_THREADS / 4));
        if (npairs <= SMALL_ROUND_PAIRS && fold)
            sumcheck_quadratic_small_kernel<true><<<sblocks, RED_THREADS, 0, ctx->stream>>>((const fe*)d_f, (const fe*)d_w, out_len, farg, gate, (fe*)d_f_out,
                                                                                            (fe*)d_w_out, red_partials(ctx), red_ticket(ctx),
                                                                                            red_result(ctx), seq);
        else if (npairs <= SMALL_ROUND_PAIRS)
            sumcheck_quadratic_small_kernel<false><<<sblocks, RED_THREADS, 0, ctx->stream>>>((const fe*)d_f, (const fe*)d_w, out_len, fe_arg{}, gate_none(),
                                                                                             nullptr, nullptr, red_partials(ctx), red_ticket(ctx),
                                                                                             red_result(ctx), seq);
        else if (fold)
            sumcheck_quadratic_kernel<true><<<blocks, RED_THREADS, 0, ctx->stream>>>((const fe*)d_f, (const fe*)d_w, out_len, farg, gate, (fe*)d_f_out,
                                                                                      (fe*)d_w_out, red_partials(ctx), red_ticket(ctx), red_result(ctx), seq);
        else
            sumcheck_quadratic_kernel<false><<<blocks, RED_THREADS, 0, ctx->stream>>>((const fe*)d_f, (const fe*)d_w, out_len, fe_arg{}, gate_none(), nullptr,
                                                                                       nullptr, red_partials(ctx), red_ticket(ctx), red_result(ctx), seq);
    }
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}
// the host's side of a gated launch and the wait for a launch's three results without draining the stream (prover.hip, latency mode)
int sumcheck_collect_spin(pk_ctx* ctx, unsigned red_seq, uint64_t out[12]) { return collect_reduction_spin<3>(ctx, red_seq, out); }
unsigned sumcheck_gate_next(pk_ctx* ctx) { return gate_next(ctx); }
// 0 = no gated kernel of this context gave up on its challenge since the last call, else PK_ERR_HIP with the message set (reduce.hpp)
void sumcheck_gate_clear(pk_ctx* ctx) { (void)gate_timed_out(ctx); }  // forget a give-up word without touching the error message
int sumcheck_gate_check(pk_ctx* ctx) { return gate_timed_out(ctx) ? set_err(ctx, PK_ERR_HIP, "%s", PK_GATE_TIMEOUT_MSG) : PK_OK; }
void sumcheck_gate_publish(pk_ctx* ctx, unsigned gate_seq, const uint64_t challenge[4]) {
    fe c;
    memcpy(c.v, challenge, 32);
    gate_publish(ctx, gate_seq, c);
}
}  // namespace pk
extern "C" {

int pk_sumcheck_cubic_round(pk_ctx* ctx, uint64_t* d_a, uint64_t* d_b, uint64_t* d_c, uint64_t* d_eq, size_t len,
                            const uint64_t* fold_or_null, uint64_t out[12]) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, out, "null pointer");
    unsigned seq = 0;
    int rc = sumcheck_cubic_launch(ctx, d_a, d_b, d_c, d_eq, len, fold_or_null, 0, &seq);
    if (rc) return rc;
    return collect_reduction<3>(ctx, out);
}

int pk_sumcheck_quadratic_round(pk_ctx* ctx, const uint64_t* d_f, const uint64_t* d_w, size_t len, const uint64_t* fold_or_null,
                                uint64_t* d_f_out, uint64_t* d_w_out, uint64_t out[12]) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, out, "null pointer");
    unsigned seq = 0;
    int rc = sumcheck_quadratic_launch(ctx, d_f, d_w, len, fold_or_null, 0, d_f_out, d_w_out, &seq);
    if (rc) return rc;
    return collect_reduction<3>(ctx, out);
}

int pk_fold_pairs(pk_ctx* ctx, const uint64_t* d_v, size_t len, const uint64_t* r, uint64_t* d_out) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, d_v && d_out && r && d_v != d_out, "null or aliased pointer");
    PK_REQUIRE(ctx, is_pow2(len) && len >= 2, "size must be a power of two >= 2");
    fold_pairs_kernel<<<grid_for(ctx, len / 2, 256), 256, 0, ctx->stream>>>((const fe*)d_v, (fe*)d_out, nullptr, nullptr, len / 2, to_arg(r), gate_none());
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}
}  // extern "C"
namespace pk {
int lincomb2(pk_ctx* ctx, uint64_t* d_out, const uint64_t* d_a, const uint64_t* beta, const uint64_t* d_b, size_t n) {
    PK_REQUIRE(ctx, beta && (n == 0 || (d_out && d_a && d_b)), "null pointer");
    if (!n) return PK_OK;
    ProfScope prof(ctx, "lincomb");
    lincomb_kernel<<<grid_for(ctx, n, 256), 256, 0, ctx->stream>>>((fe*)d_out, (const fe*)d_a, (const fe*)d_b, n, to_arg(beta));
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}
// two arrays of the same length folded by the same challenge in one launch (the sumcheck's p and w)
// r = NULL with gate_seq != 0: the challenge arrives through the gate (latency mode)
int fold_pairs2_gated(pk_ctx* ctx, const uint64_t* d_v0, uint64_t* d_out0, const uint64_t* d_v1, uint64_t* d_out1, size_t len, const uint64_t* r,
                      unsigned gate_seq) {
    PK_REQUIRE(ctx, d_v0 && d_out0 && d_v1 && d_out1 && (r || gate_seq), "null pointer");
    PK_REQUIRE(ctx, is_pow2(len) && len >= 2, "size must be a power of two >= 2");
    if (gate_seq) {
        int rc = reduction_scratch(ctx);
        if (rc) return rc;
    }
    ProfScope prof(ctx, "fold_pairs");
    fold_pairs_kernel<<<dim3(grid_for(ctx, len / 2, 256), 2), 256, 0, ctx->stream>>>((const fe*)d_v0, (fe*)d_out0, (const fe*)d_v1, (fe*)d_out1,
                                                                                     len / 2, r ? to_arg(r) : fe_arg{},
                                                                                     gate_seq ? gate_for(ctx, gate_seq) : gate_none());
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}
int fold_pairs2(pk_ctx* ctx, const uint64_t* d_v0, uint64_t* d_out0, const uint64_t* d_v1, uint64_t* d_out1, size_t len, const uint64_t* r) {
    PK_REQUIRE(ctx, r, "null pointer");
    return fold_pairs2_gated(ctx, d_v0, d_out0, d_v1, d_out1, len, r, 0);
}
}  // namespace pk
extern "C" {

int pk_dot(pk_ctx* ctx, const uint64_t* d_w, const uint64_t* d_f, size_t n, uint64_t out[4]) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, out && (n == 0 || (d_w && d_f)), "null pointer");
    if (n == 0 && ctx->red_across) {  // an empty share still takes part in the exchange of the ranks' partial sums
        PK_HIP(ctx, hipMemsetAsync(ctx->d_xred, 0, 32, ctx->stream));
        return collect_reduction<1>(ctx, out);
    }
    if (n == 0) {
        memset(out, 0, 32);
        return PK_OK;
    }
    int rc = reduction_scratch(ctx);
    if (rc) return rc;
    unsigned blocks = reduction_blocks(ctx, n);
    {
        ProfScope prof(ctx, "dot");
        dot_kernel<1><<<blocks, RED_THREADS, 0, ctx->stream>>>((const fe*)d_w, (const fe*)d_f, nullptr, n, red_partials(ctx), red_ticket(ctx),
                                                               red_result(ctx), next_seq(ctx));
    }
    PK_LAUNCH_CHECK(ctx);
    return collect_reduction<1>(ctx, out);
}

int pk_dot2(pk_ctx* ctx, const uint64_t* d_w, const uint64_t* d_f, const uint64_t* d_g, size_t n, uint64_t out[8]) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, out && (n == 0 || (d_w && d_f && d_g)), "null pointer");
    if (n == 0 && ctx->red_across) {
        PK_HIP(ctx, hipMemsetAsync(ctx->d_xred, 0, 64, ctx->stream));
        return collect_reduction<2>(ctx, out);
    }
    if (n == 0) {
        memset(out, 0, 64);
        return PK_OK;
    }
    int rc = reduction_scratch(ctx);
    if (rc) return rc;
    unsigned blocks = reduction_blocks(ctx, n);
    {
        ProfScope prof(ctx, "dot");
        dot_kernel<2><<<blocks, RED_THREADS, 0, ctx->stream>>>((const fe*)d_w, (const fe*)d_f, (const fe*)d_g, n, red_partials(ctx), red_ticket(ctx),
                                                               red_result(ctx), next_seq(ctx));
    }
    PK_LAUNCH_CHECK(ctx);
    return collect_reduction<2>(ctx, out);
}

int pk_eval_univariate(pk_ctx* ctx, const uint64_t* d_coeffs, size_t n, const uint64_t z[4], uint64_t out[4]) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, out && z && (n == 0 || d_coeffs), "null pointer");
    const uint64_t* polys[1] = {d_coeffs};
    return pk::eval_univariate_multi(ctx, polys, 1, n, z, out);
}
}  // extern "C"
namespace pk {
// np (1 or 2) polynomials of n coefficients each at the same z, one launch: out[4 * q] = poly_q(z)
int eval_univariate_multi(pk_ctx* ctx, const uint64_t* const* d_polys, unsigned np, size_t n, const uint64_t z[4], uint64_t* out) {
    PK_REQUIRE(ctx, np == 1 || np == 2, "one or two polynomials");
    if (n == 0) {
        if (ctx->red_across) {
            PK_HIP(ctx, hipMemsetAsync(ctx->d_xred, 0, 32 * np, ctx->stream));
            return np == 1 ? collect_reduction<1>(ctx, out) : collect_reduction<2>(ctx, out);
        }
        memset(out, 0, 32 * np);
        return PK_OK;
    }
    int rc = reduction_scratch(ctx);
    if (rc) return rc;
    // ~8 coefficients per lane, at most RED_MAX_BLOCKS (1024) workgroups, each over a contiguous segment of 256 * cnt coefficients
    size_t want = (n / 8 + RED_THREADS - 1) / RED_THREADS;
    const size_t cap = want < 1 ? 1 : (want > RED_MAX_BLOCKS ? RED_MAX_BLOCKS : want);
    const size_t cnt = (n + cap * RED_THREADS - 1) / (cap * RED_THREADS);
    const unsigned blocks = (unsigned)((n + cnt * RED_THREADS - 1) / (cnt * RED_THREADS));
    PK_REQUIRE(ctx, (n >> 28) == 0, "polynomial too long for the evaluation kernel (2^28 coefficients)");
    if (!ctx->d_ztab) PK_HIP(ctx, hipMalloc(&ctx->d_ztab, 256 * 32));
    // z^(2^i) on the host (27 squarings)
    pow2_args zp;
    {
        fe b;
        memcpy(b.v, z, 32);
        for (int i = 0; i < 28; i++) {
            memcpy(zp.p[i].v, b.v, 32);
            b = fe_mulx(b, b);
        }
    }
    {
        ProfScope prof(ctx, "eval_univariate");
        zpow_table_kernel<<<1, RED_THREADS, 0, ctx->stream>>>(zp, (fe*)ctx->d_ztab);
        if (np == 1)
            horner_kernel<1><<<blocks, RED_THREADS, 0, ctx->stream>>>((const fe*)d_polys[0], nullptr, n, cnt, zp, (const fe*)ctx->d_ztab, red_partials(ctx),
                                                                     red_ticket(ctx), red_result(ctx), next_seq(ctx));
        else
            horner_kernel<2><<<blocks, RED_THREADS, 0, ctx->stream>>>((const fe*)d_polys[0], (const fe*)d_polys[1], n, cnt, zp, (const fe*)ctx->d_ztab,
                                                                     red_partials(ctx), red_ticket(ctx), red_result(ctx), next_seq(ctx));
    }
    PK_LAUNCH_CHECK(ctx);
    return np == 1 ? collect_reduction<1>(ctx, out) : collect_reduction<2>(ctx, out);
}
// nrows (1..3) weight rows against f and (nv == 2) g in one pass: out[4 * (nv * k + v)]
// defer: launch only -- the caller synchronises `ctx`'s stream later and takes the 3 * nv results from its pinned page (latency mode: the
// statement's sums run on a side stream underneath the blinding WHIR proof)
int dot_rows_x(pk_ctx* ctx, const uint64_t* d_w, size_t row_stride, unsigned nrows, const uint64_t* d_f, const uint64_t* d_g, size_t n, uint64_t* out,
               bool defer);
int dot_rows(pk_ctx* ctx, const uint64_t* d_w, size_t row_stride, unsigned nrows, const uint64_t* d_f, const uint64_t* d_g, size_t n, uint64_t* out) {
    return dot_rows_x(ctx, d_w, row_stride, nrows, d_f, d_g, n, out, false);
}
int dot_rows_x(pk_ctx* ctx, const uint64_t* d_w, size_t row_stride, unsigned nrows, const uint64_t* d_f, const uint64_t* d_g, size_t n, uint64_t* out,
               bool defer) {
    const int nv = d_g ? 2 : 1;
    PK_REQUIRE(ctx, nrows == 3 && (out || defer) && (n == 0 || (d_w && d_f)), "three weight rows");
    PK_REQUIRE(ctx, !defer || (n != 0 && !ctx->red_across), "a deferred dot product needs work and a lone context");
    if (n == 0) {
        if (ctx->red_across) {
            PK_HIP(ctx, hipMemsetAsync(ctx->d_xred, 0, 32 * 3 * nv, ctx->stream));
            return nv == 2 ? collect_reduction<6>(ctx, out) : collect_reduction<3>(ctx, out);
        }
        memset(out, 0, 32 * 3 * (size_t)nv);
        return PK_OK;
    }
    int rc = reduction_scratch(ctx);
    if (rc) return rc;
    unsigned blocks = reduction_blocks(ctx, n);
    {
        ProfScope prof(ctx, "dot");
        if (nv == 2)
            dot_rows_kernel<3, 2><<<blocks, RED_THREADS, 0, ctx->stream>>>((const fe*)d_w, row_stride, (const fe*)d_f, (const fe*)d_g, n, red_partials(ctx),
                                                                          red_ticket(ctx), red_result(ctx), next_seq(ctx));
        else
            dot_rows_kernel<3, 1><<<blocks, RED_THREADS, 0, ctx->stream>>>((const fe*)d_w, row_stride, (const fe*)d_f, nullptr, n, red_partials(ctx),
                                                                          red_ticket(ctx), red_result(ctx), next_seq(ctx));
    }
    PK_LAUNCH_CHECK(ctx);
    if (defer) return PK_OK;
    return nv == 2 ? collect_reduction<6>(ctx, out) : collect_reduction<3>(ctx, out);
}
}  // namespace pk
extern "C" {

int pk_fold_coeffs(pk_ctx* ctx, const uint64_t* d_coeffs, unsigned n_vars, const uint64_t* r, unsigned k, uint64_t* d_out) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, d_coeffs && d_out && (k == 0 || r), "null pointer");
    PK_REQUIRE(ctx, k <= 8 && k <= n_vars, "fold factor out of range");
    fold_args ra{};
    for (unsigned b = 0; b < k; b++) ra.r[b] = to_arg(r + 4 * b);
    size_t n_out = (size_t)1 << (n_vars - k);
    ProfScope prof(ctx, "fold_coeffs");
    if (k == 4) {  // weight j = prod_b r_b^{bit_b(j)} by doubling, on the host
        fe wt[16];
        wt[0] = fe_one();
        for (unsigned b = 0; b < 4; b++) {
            fe rb;
            memcpy(rb.v, r + 4 * b, 32);
            for (unsigned j = 0; j < (1u << b); j++) wt[(1u << b) + j] = fe_mulx(wt[j], rb);
        }
        fold16_args wa;
        for (int j = 0; j < 16; j++) memcpy(wa.w[j].v, wt[j].v, 32);
        if (n_out <= 8192)
            fold16_kernel<16><<<(unsigned)((n_out * 16 + 255) / 256), 256, 0, ctx->stream>>>((const fe*)d_coeffs, n_out, wa, (fe*)d_out);
        else
            fold16_kernel<1><<<grid_for(ctx, n_out, 256), 256, 0, ctx->stream>>>((const fe*)d_coeffs, n_out, wa, (fe*)d_out);
        PK_LAUNCH_CHECK(ctx);
        return PK_OK;
    }
    fold_coeffs_kernel<<<grid_for(ctx, n_out, 256), 256, 0, ctx->stream>>>((const fe*)d_coeffs, n_out, k, ra, (fe*)d_out);
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}

int pk_fe_axpy(pk_ctx* ctx, uint64_t* d_y, const uint64_t* beta, const uint64_t* d_x, size_t n) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, beta && (n == 0 || (d_y && d_x)), "null pointer");
    if (!n) return PK_OK;
    ProfScope prof(ctx, "lincomb");
    axpy_kernel<<<grid_for(ctx, n, 256), 256, 0, ctx->stream>>>((fe*)d_y, (const fe*)d_x, n, to_arg(beta));
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}

}  // extern "C"
