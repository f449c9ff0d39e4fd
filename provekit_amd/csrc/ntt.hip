// ntt.hip -- Reed-Solomon low-degree extension over BN254-Fr.
//
// What it replaces: the NTT engine of the external `whir` crate as used by
// CommitmentWriter::commit_batch and the per-round re-commit (call site
// provekit/prover/src/whir_r1cs.rs:200-206).  Semantics (pinned by the Go verifier,
// recursive-verifier/app/circuit/whir_utilities.go:180-186 and whir.go:99,141):
//   coeffs c[0..2^n), fold 2^k, domain D = 2^(n+rho), rows = D / 2^k
//   leaf_i[j] = sum_t c[2^k t + j] * w^(i t),  w = generator of the order-`rows` subgroup
// i.e. 2^k independent size-`rows` NTTs of the (zero-padded) stride-2^k sub-sequences.
//
// MI355X design.  The codeword matrix lives in HBM COLUMN-major: M[col][row]
// (col = b*2^k + j).  Then (1) every NTT is a contiguous vector, (2) the leaf hash
// reads one column per step fully coalesced, (3) nothing is ever transposed back --
// only the ~100 opened leaves are gathered to leaf-major at the boundary.
//
// A size-N transform is 1..3 passes (N = R1*R2*R3, R <= 512).  Each pass is one
// kernel: a workgroup pulls a [R x 4] tile (4 adjacent 32-byte elements = one 128 B
// line per row) into LDS, runs log2(R) radix-2 DIF stages there (twiddles w_R^j staged
// in LDS), and writes the tile back multiplied by the inter-pass twiddle w_N^(k*m) read
// from a per-size table.  The index maps are Cooley-Tukey so that natural-order input
// gives natural-order output with no bit-reversal pass over HBM:
//   n = n1*R2*R3 + n2*R3 + n3,   k = k1 + R1*k2 + R1*R2*k3.
#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include <atomic>

// partly memory-bound (VALUBusy ~75 %): between the pure-ALU hash kernels (0) and the memory-bound kernels (2)
#define PK_BASE_PRIO 1
#include "ctx.hpp"
#include "ntt_regs.hpp"

using namespace pk;

namespace {

constexpr int BT = 4;        // batch of adjacent elements per tile row (128 B)
constexpr int NTHREADS = 256;

// 2^28-th root of unity 5^((p-1)/2^28) (ark-bn254 Fr::TWO_ADIC_ROOT_OF_UNITY), canonical
// 19103219067921713944291392827692070036145651957329286315305642004821462161904 -> Montgomery
__device__ __forceinline__ fe root28_mont() {
    fe r;
    r.v[0] = 0x725b19f0u; r.v[1] = 0x9bd61b6eu; r.v[2] = 0x41112ed4u; r.v[3] = 0x402d111eu;
    r.v[4] = 0x8ef62abcu; r.v[5] = 0x00e0a7ebu; r.v[6] = 0xa58a7e85u; r.v[7] = 0x2a3c09f0u;
    return fe_to_montx(r);
}

// W[e] = w_N^e for e in [0, N): W[0] = 1, then doubling: W[h + j] = W[j] * w^h
__global__ void twiddle_init_kernel(fe* W, unsigned log_n) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        fe_store(W, fe_one());
        fe w = root28_mont();
        for (unsigned i = log_n; i < 28; i++) w = fe_sqrx(w);
        if (log_n > 0) fe_store(W + 1, w);
    }
}
__global__ __launch_bounds__(256) void twiddle_double_kernel(fe* W, size_t h) {
    // W[h + j] = W[j] * W[h]  for j in [1, h)   (W[h] itself is written by the caller chain)
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= h) return;
    fe wh = fe_load(W + h);
    if (j == 0) return;
    fe_store(W + h + j, fe_mulx(fe_load(W + j), wh));
}
__global__ void twiddle_seed_kernel(fe* W, size_t h) {
    // W[h] = W[h/2]^2
    if (threadIdx.x == 0 && blockIdx.x == 0) fe_store(W + h, fe_sqrx(fe_load(W + h / 2)));
}

// Ws[e] = 32 * w_N^e as a plain integer (W holds Montgomery images: mont(W[e], 32) = w * 32).  Multiplying a Montgomery
// image x*2^256 by it in the Montgomery product drops the 2^256 and scales by 32: the form the hash kernels consume
// (skyscraper29s.hpp), so the leaf hash needs no conversion of its inputs.
__global__ __launch_bounds__(256) void twiddle_scale_kernel(const fe* __restrict__ W, fe* __restrict__ Ws, size_t n) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    fe c = fe_zero();
    c.v[0] = 32u;
    fe_store(Ws + j, fe_mulx(fe_load(W + j), c));
}

// The register-radix kernel's multiplier table: entry j = the constant c_j that multiplying by X[j] in the Montgomery product amounts to
// (c = X[j] * 2^-256 mod p: the plain w for the table of Montgomery images W, 32 w 2^-256 for the hash-ready table Ws), canonical, as
// 29-bit limbs, followed by its Shoup quotient floor(c * 2^261 / p) (exact: 261 steps of long division, once per table).
__global__ __launch_bounds__(256) void twiddle_shoup_kernel(const fe* __restrict__ X, u32* __restrict__ T, size_t n) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const fe29 c = unpack29<0>(fe_from_montx(fe_load(X + j)));
    const fe29 cq = shoup_quotient29(c);
#pragma unroll
    for (int l = 0; l < 9; l++) T[TW29S_WORDS * j + l] = c.v[l];
#pragma unroll
    for (int l = 0; l < 9; l++) T[TW29S_WORDS * j + 9 + l] = cq.v[l];
}

// T[k * row + v] = src[(mul * k * v) & mask]: an inter-pass twiddle table re-ordered into the order the pass reads it
__global__ __launch_bounds__(256) void twiddle_pass_table_kernel(const u32* __restrict__ src, u32* __restrict__ T, size_t rows_k, size_t row, size_t mul,
                                                                 size_t mask) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows_k * row) return;
    const size_t k = i / row, v = i % row;
    const u32* q = src + TW29S_WORDS * ((mul * k * v) & mask);
#pragma unroll
    for (int l = 0; l < TW29S_WORDS; l++) T[TW29S_WORDS * i + l] = q[l];
}

struct PassParams {
    const fe* in;
    fe* out;
    size_t in_col_stride, out_col_stride;  // elements between consecutive columns
    size_t in_stride_r, in_stride_v, in_stride_u;
    size_t out_stride_r, out_stride_v, out_stride_u;
    unsigned log_v;     // extent of the v axis = 2^log_v (tiles take BT of them)
    size_t nonzero;     // pass 1 only: natural input indices >= nonzero read as zero
    size_t in_nat_r, in_nat_v, in_nat_u;  // natural-index weights for the zero test
    const fe* W;        // w_N^e table, N entries
    const fe* Wtw;      // table the inter-pass twiddle is read from: W, or its hash-ready variant 32*w_N^e (see ntt_columns)
    const u32* W29;     // W again as Shoup multipliers (ntt_regs.hpp tw29s: 9 limbs of the value + 9 of its quotient): register-radix kernel
    const u32* Wtw29;   // Wtw likewise
    const u32* Tpass29; // optional: the inter-pass twiddles of THIS pass in access order, T[k * tp_row + (v0 + b)] = w_N^(tw_mul k (v0+b))
                        // in operand form (or its hash-ready variant) -- adjacent lanes read adjacent 72-byte entries instead of
                        // gathering from the size-N table at stride k; null = gather from Wtw29
    size_t tp_row;      // entries per k of Tpass29 (the extent of the v axis)
    size_t tiles;       // register-radix kernel: tiles per column, columns, and whether the XCD-aware block order applies
    unsigned ncols;
    int xcd_tiles;
    int lazy_store;     // outputs may be stored almost reduced (< 1.6p) instead of canonical: intermediate passes, and the
                        // hash-ready final pass (its consumers reduce anyway)
    size_t n_mask;      // N - 1
    size_t tw_mul;      // twiddle exponent = tw_mul * k * v  (0 = no twiddle)
    size_t wr_step;     // w_R^j = W[j * wr_step]
};

template <int LOG_R, bool IN_R_CONTIG>
__global__ __launch_bounds__(NTHREADS) void ntt_pass_kernel(PassParams p) {
    PK_LATENCY_PRIO();
    constexpr int R = 1 << LOG_R;
    constexpr int TILE = R * BT;
    extern __shared__ uint4 lds[];
    uint4* lo = lds;                 // TILE entries: low 16 B of each element
    uint4* hi = lds + TILE;          // TILE entries: high 16 B
    uint4* wlo = lds + 2 * TILE;     // R/2 twiddles
    uint4* whi = wlo + (R / 2 > 0 ? R / 2 : 1);

    const unsigned tid = threadIdx.x;
    const size_t tile = blockIdx.x;
    const size_t vblocks = ((size_t)1 << p.log_v) / BT;
    const size_t u = tile / vblocks, vb = tile % vblocks;
    const size_t v0 = vb * BT;
    const size_t col = blockIdx.y;
    const fe* in = p.in + col * p.in_col_stride + u * p.in_stride_u + v0 * p.in_stride_v;
    fe* out = p.out + col * p.out_col_stride + u * p.out_stride_u + v0 * p.out_stride_v;
    const size_t nat0 = u * p.in_nat_u + v0 * p.in_nat_v;

    // stage twiddles w_R^j, j < R/2
    for (int j = tid; j < R / 2; j += NTHREADS) {
        const uint4* q = reinterpret_cast<const uint4*>(p.W + (size_t)j * p.wr_step);
        wlo[j] = q[0];
        whi[j] = q[1];
    }
    // load tile: LDS index = r*BT + b
    for (int e = tid; e < TILE; e += NTHREADS) {
        int r, b;
        if (IN_R_CONTIG) {
            r = e % R;
            b = e / R;
        } else {
            b = e % BT;
            r = e / BT;
        }
        size_t nat = nat0 + (size_t)r * p.in_nat_r + (size_t)b * p.in_nat_v;
        uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
        if (nat < p.nonzero) {
            const uint4* q = reinterpret_cast<const uint4*>(in + (size_t)r * p.in_stride_r + (size_t)b * p.in_stride_v);
            a0 = q[0];
            a1 = q[1];
        }
        lo[r * BT + b] = a0;
        hi[r * BT + b] = a1;
    }
    __syncthreads();

    // radix-2 DIF stages over r (natural in -> bit-reversed out)
#pragma unroll 1
    for (int s = 0; s < LOG_R; s++) {
        const int half = R >> (s + 1);
        for (int t = tid; t < (R / 2) * BT; t += NTHREADS) {
            int b = t % BT, j = t / BT;
            int pos = j & (half - 1);
            int i0 = ((j - pos) << 1) + pos;
            int i1 = i0 + half;
            int a0i = i0 * BT + b, a1i = i1 * BT + b;
            fe x0, x1;
            {
                uint4 l0 = lo[a0i], h0 = hi[a0i], l1 = lo[a1i], h1 = hi[a1i];
                x0.v[0] = l0.x; x0.v[1] = l0.y; x0.v[2] = l0.z; x0.v[3] = l0.w;
                x0.v[4] = h0.x; x0.v[5] = h0.y; x0.v[6] = h0.z; x0.v[7] = h0.w;
                x1.v[0] = l1.x; x1.v[1] = l1.y; x1.v[2] = l1.z; x1.v[3] = l1.w;
                x1.v[4] = h1.x; x1.v[5] = h1.y; x1.v[6] = h1.z; x1.v[7] = h1.w;
            }
            fe sum = fe_add(x0, x1);
            fe dif = fe_sub(x0, x1);
            int widx = pos << s;
            if (widx != 0) {
                uint4 wl = wlo[widx], wh = whi[widx];
                fe w;
                w.v[0] = wl.x; w.v[1] = wl.y; w.v[2] = wl.z; w.v[3] = wl.w;
                w.v[4] = wh.x; w.v[5] = wh.y; w.v[6] = wh.z; w.v[7] = wh.w;
                dif = fe_mulx(dif, w);
            }
            lo[a0i] = make_uint4(sum.v[0], sum.v[1], sum.v[2], sum.v[3]);
            hi[a0i] = make_uint4(sum.v[4], sum.v[5], sum.v[6], sum.v[7]);
            lo[a1i] = make_uint4(dif.v[0], dif.v[1], dif.v[2], dif.v[3]);
            hi[a1i] = make_uint4(dif.v[4], dif.v[5], dif.v[6], dif.v[7]);
        }
        __syncthreads();
    }

    // store: output index k sits at LDS row bitrev(k); multiply by w_N^(tw_mul*k*v)
    for (int e = tid; e < TILE; e += NTHREADS) {
        int b = e % BT, k = e / BT;
        int r = LOG_R == 0 ? 0 : (int)(__brev((unsigned)k) >> (LOG_R ? 32 - LOG_R : 1));
        uint4 l0 = lo[r * BT + b], h0 = hi[r * BT + b];
        fe x;
        x.v[0] = l0.x; x.v[1] = l0.y; x.v[2] = l0.z; x.v[3] = l0.w;
        x.v[4] = h0.x; x.v[5] = h0.y; x.v[6] = h0.z; x.v[7] = h0.w;
        if (p.tw_mul != 0) {
            size_t ex = (p.tw_mul * (size_t)k * (v0 + b)) & p.n_mask;
            if (ex != 0) x = fe_mulx(x, fe_load(p.W + ex));
        }
        fe_store(out + (size_t)k * p.out_stride_r + (size_t)b * p.out_stride_v, x);
    }
}

// ------------------------------------------------------------------------------------------------------
// Fast path: the same pass (same PassParams, same index maps) with the butterflies kept in REGISTERS.
// A 512-lane workgroup owns a 2048-element tile = [R = 2^LOG_R rows] x [BT = 2048/R adjacent elements]; each lane holds 2^LE = 4
// elements in the 9x29-bit lazy form (fe29.hpp) and runs a radix-2^d DIF butterfly network on them (ntt_regs.hpp; d = 2 except
// possibly the first round), so LDS is touched only BETWEEN rounds: ceil(LOG_R/2)-1 exchanges instead of LOG_R read-modify-write
// sweeps.  Every multiplication is by a constant known before the launch and takes the Shoup form (value + precomputed quotient, 72
// bytes per table entry; the powers of w_8 are compile-time constants).  118 registers per lane: four waves per SIMD -- the
// multiply-add pipe needs two ready waves to run at its rate, and a wave of this kernel waits (loads, barriers) a third of the time.
//
// Tile index I (11 bits) = r * BT + b.  Round j transforms digit D_j of r (digits are taken from the top of r: DIF),
// leaving the frequency digit a_j in the same bit positions; the output frequency is the digit reversal
// k = a_0 + 2^d0 a_1 + 2^(d0+d1) a_2 + ...  In round j a lane's registers are indexed by (digit bits | low LE-d_j bits of b).
__host__ __device__ __forceinline__ constexpr int bitrev_c(int v, int bits) {
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}

struct Ntt8Ctx {
    const fe* in;
    fe* out;
    size_t nat0, v0;
    unsigned tid;
};

// LE = log2 of the elements a lane holds (NTT_LE = 2: four registers, radix-4 rounds, 512 lanes per tile).
// one round (digit J) of the register NTT; everything about the digit layout is a compile-time constant
template <int LE, int LOG_R, int J>
__device__ __forceinline__ void ntt8_round(fe29 (&x)[1 << LE], const PassParams& p, const Ntt8Ctx& c, u32* planes) {
    constexpr int NX = 1 << LE;
    constexpr int LOGB = 11 - LOG_R, BTT = 1 << LOGB;
    constexpr int NR = (LOG_R + LE - 1) / LE;
    constexpr int D0 = LOG_R - LE * (NR - 1);
    constexpr int D = J == 0 ? D0 : LE;
    constexpr int DONE = J == 0 ? 0 : D0 + LE * (J - 1);  // r bits consumed before this round
    constexpr int POS = LOGB + LOG_R - DONE - D;           // bit offset of digit J inside the tile index
    constexpr int E = LE - D;
    constexpr int REST = POS - LOGB;                       // r bits below this digit
    constexpr int LO = POS - E;
    const unsigned tid = c.tid;
    // tile index of (tid, reg): [tid hi][digit][tid lo][extra]
    const int base_idx = (int)((tid & ((1u << LO) - 1)) << E) | (int)((tid >> LO) << (POS + D));
#define PK_TILE_INDEX(reg) (base_idx | ((reg) & ((1 << E) - 1)) | (((reg) >> E) << POS))
    if (J == 0) {
#pragma unroll
        for (int reg = 0; reg < NX; reg++) {
            const int I = PK_TILE_INDEX(reg);
            const int r = I >> LOGB, b = I & (BTT - 1);
            size_t nat = c.nat0 + (size_t)r * p.in_nat_r + (size_t)b * p.in_nat_v;
            fe t = fe_zero();
            if (nat < p.nonzero) t = fe_load(c.in + (size_t)r * p.in_stride_r + (size_t)b * p.in_stride_v);
            x[reg] = unpack29<0>(t);
        }
    } else {
        __syncthreads();
#pragma unroll
        for (int reg = 0; reg < NX; reg++) {
            const int I = PK_TILE_INDEX(reg);
#pragma unroll
            for (int l = 0; l < 9; l++) x[reg].v[l] = planes[l * 2048 + I];
        }
        __syncthreads();
    }
    dft_regs<LE, D>(x);
    const int m = (base_idx >> LOGB) & ((1 << REST) - 1);  // r bits below the digit: index inside the sub-transform
    if (J + 1 < NR) {
        const size_t unit = p.wr_step << (LOG_R - D - REST);  // w_{2^(D+REST)} in units of w_N
#pragma unroll
        for (int reg = 0; reg < NX; reg++) {
            const int a = bitrev_c(reg >> E, D);
            fe29 y = a == 0 ? red29(x[reg]) : mul_tw(x[reg], tw29s_load(p.W29, (size_t)(a * m) * unit));  // a == 0: the sum of sums
            const int I = (PK_TILE_INDEX(reg) & ~(((1 << D) - 1) << POS)) | (a << POS);  // keep the true frequency digit
#pragma unroll
            for (int l = 0; l < 9; l++) planes[l * 2048 + I] = y.v[l];
        }
    } else {
        // frequency digits of the earlier rounds sit above this digit in the index (already true frequencies): digit j, d_j bits
        // wide, sits d_0 + ... + d_j below the top of r and weighs 2^(d_0 + ... + d_(j-1)) in k
        const int rr = base_idx >> LOGB;
        int k_hi = 0;
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int dj = j == 0 ? D0 : LE, done_j = j == 0 ? 0 : D0 + LE * (j - 1);
            k_hi |= ((rr >> (LOG_R - done_j - dj)) & ((1 << dj) - 1)) << done_j;
        }
        constexpr int KLO = DONE;
#pragma unroll
        for (int reg = 0; reg < NX; reg++) {
            const int a = bitrev_c(reg >> E, D);
            const int b = PK_TILE_INDEX(reg) & (BTT - 1);
            const int k = k_hi | (a << KLO);
            fe29 y;
            if (p.tw_mul && p.Tpass29) {
                y = mul_tw(x[reg], tw29s_load(p.Tpass29, (size_t)k * p.tp_row + (c.v0 + b)));
            } else if (p.tw_mul) {
                size_t ex = (p.tw_mul * (size_t)k * (c.v0 + b)) & p.n_mask;
                y = mul_tw(x[reg], tw29s_load(p.Wtw29, ex));
            } else {
                y = a == 0 ? red29(x[reg]) : red29w(x[reg]);
            }
            fe_store(c.out + (size_t)k * p.out_stride_r + (size_t)b * p.out_stride_v, p.lazy_store ? pack29(y) : pack_canon29(y));
        }
    }
#undef PK_TILE_INDEX
}

template <int LE, int LOG_R, int J>
__device__ __forceinline__ void ntt8_rounds_from(fe29 (&x)[1 << LE], const PassParams& p, const Ntt8Ctx& c, u32* planes) {
    constexpr int NR = (LOG_R + LE - 1) / LE;
    if constexpr (J < NR) {
        ntt8_round<LE, LOG_R, J>(x, p, c, planes);
        ntt8_rounds_from<LE, LOG_R, J + 1>(x, p, c, planes);
    }
}

template <int LE, int LOG_R, bool IN_R_CONTIG>
__global__ __launch_bounds__(2048 >> LE) __attribute__((amdgpu_waves_per_eu(4, 4))) void ntt8_pass_kernel(PassParams p) {
    PK_LATENCY_PRIO();
    constexpr int LOGB = 11 - LOG_R;
    extern __shared__ u32 planes[];  // [9][2048]
    const size_t vblocks = ((size_t)1 << p.log_v) >> LOGB;
    // block -> (tile, column).  Workgroup b runs on XCD b % 8 (observed placement, used for speed only): an XCD takes the tiles
    // t = xcd (mod 8) and, for each, all columns back to back, so the tile's inter-pass twiddles (the same rows of the
    // pass-ordered table for every column) are fetched from HBM once and served from that XCD's L2 for the other columns.
    size_t tile, col;
    if (p.xcd_tiles) {
        const size_t j = blockIdx.x >> 3;
        col = j % p.ncols;
        tile = (j / p.ncols) * 8 + (blockIdx.x & 7);
    } else {
        tile = blockIdx.x % p.tiles;
        col = blockIdx.x / p.tiles;
    }
    const size_t u = tile / vblocks, vb = tile % vblocks;
    Ntt8Ctx c;
    c.v0 = vb << LOGB;
    c.in = p.in + col * p.in_col_stride + u * p.in_stride_u + c.v0 * p.in_stride_v;
    c.out = p.out + col * p.out_col_stride + u * p.out_stride_u + c.v0 * p.out_stride_v;
    c.nat0 = u * p.in_nat_u + c.v0 * p.in_nat_v;
    c.tid = threadIdx.x;
    fe29 x[1 << LE];
    ntt8_rounds_from<LE, LOG_R, 0>(x, p, c, planes);
}

// c[2^k t + j] -> S[j][t]  (t < L): stride-2^k gather done through LDS so both sides coalesce
template <int FW_MAX>
__global__ __launch_bounds__(256) void deinterleave_kernel(const fe* __restrict__ c, fe* __restrict__ S, size_t L, unsigned fw,
                                                           size_t col_stride) {
    PK_LATENCY_PRIO();
    // tile: TT consecutive t x fw columns
    extern __shared__ uint4 tl[];
    const unsigned TT = 1024 / fw;  // 1024 elements per tile = 32 KiB
    size_t t0 = (size_t)blockIdx.x * TT;
    const fe* src = c + t0 * fw;
    for (unsigned e = threadIdx.x; e < 1024; e += 256) {
        size_t t = t0 + e / fw;
        uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
        if (t < L) {
            const uint4* q = reinterpret_cast<const uint4*>(src + e);
            a0 = q[0];
            a1 = q[1];
        }
        tl[2 * e] = a0;
        tl[2 * e + 1] = a1;
    }
    __syncthreads();
    for (unsigned e = threadIdx.x; e < 1024; e += 256) {
        unsigned j = e / TT, tt = e % TT;
        size_t t = t0 + tt;
        if (t < L) {
            unsigned src_e = tt * fw + j;
            uint4* q = reinterpret_cast<uint4*>(S + (size_t)j * col_stride + t);
            q[0] = tl[2 * src_e];
            q[1] = tl[2 * src_e + 1];
        }
    }
}

// Sharded encode: rank g of G keeps the codeword rows i = g + G t.  For one column x (L nonzero
// coefficients of N):  X[g + G t] = sum_{n'} y[n'] (w_N^G)^(n' t),   y[n'] = w_N^(n' g) * sum_s x[n' + s N/G] * w_G^(s g),
// i.e. a log2(G)-stage decimation-in-frequency pre-step restricted to residue g, then one NTT of size N/G.
__global__ __launch_bounds__(256) void ntt_shard_prestep_kernel(const fe* __restrict__ S, size_t s_col_stride, size_t L, fe* __restrict__ Y,
                                                                size_t y_col_stride, size_t Ng /* N/G */, unsigned g, unsigned G,
                                                                const fe* __restrict__ W /* w_N^e */, size_t n_mask) {
    PK_LATENCY_PRIO();
    const size_t col = blockIdx.y;
    const fe* x = S + col * s_col_stride;
    fe* y = Y + col * y_col_stride;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x; n < Ng; n += stride) {
        fe acc = fe_zero();
        for (unsigned s = 0; s < G; s++) {
            size_t idx = n + (size_t)s * Ng;
            if (idx >= L) break;
            fe v = fe_load(x + idx);
            size_t e = ((size_t)((s * g) % G) * Ng) & n_mask;  // w_G^(s g) = w_N^((s g mod G) N/G)
            if (e) v = fe_mulx(v, fe_load(W + e));
            acc = fe_add(acc, v);
        }
        size_t e = (n * (size_t)g) & n_mask;
        if (e) acc = fe_mulx(acc, fe_load(W + e));
        fe_store(y + n, acc);
    }
}

// ---- twiddle tables shared by the contexts of a device --------------------------------------------------------------------------
// Sixteen provers on one GPU used to hold sixteen copies of every table (71 MB each at the poseidon size): 1.1 GB of identical
// twiddles competing for the 256 MiB Infinity Cache.  The tables are pure functions of their key, so one copy per device serves
// every context: built once under the store's mutex (the building context drains its stream before the table is published),
// kept until the last context of the device is destroyed.  A context remembers the pointers it has looked up (its own maps), so
// the hot path takes no lock.
struct TableStore {
    std::mutex mu;
    std::map<unsigned long long, void*> tables;
    unsigned contexts = 0;
};
TableStore& store_of(int device) {
    static std::mutex mu;
    static std::map<int, TableStore> stores;  // nodes of a std::map stay where they are
    std::lock_guard<std::mutex> lock(mu);
    return stores[device];
}
enum TableKind : unsigned long long { TK_W = 1, TK_WS = 2, TK_W29 = 3, TK_WS29 = 4, TK_PASS = 5 };
// looks `key` up in the device's store; on a miss `build` allocates and fills the table on ctx->stream
template <class Build>
int shared_table(pk_ctx* ctx, unsigned long long key, void** out, Build build) {
    TableStore& S = store_of(ctx->device);
    std::lock_guard<std::mutex> lock(S.mu);
    auto it = S.tables.find(key);
    if (it != S.tables.end()) {
        *out = it->second;
        return PK_OK;
    }
    void* T = nullptr;
    int rc = build(&T);
    if (rc) return rc;
    PK_WAIT(ctx);  // complete before another context's stream may read it
    S.tables[key] = T;
    *out = T;
    return PK_OK;
}

int get_twiddles(pk_ctx* ctx, unsigned log_n, const fe** out) {
    // the context's own map of pointers already looked up (a pk_ctx is single-caller: no locking on this path)
    auto it = ctx->twiddles.find(log_n);
    if (it != ctx->twiddles.end()) {
        *out = (const fe*)it->second;
        return PK_OK;
    }
    void* T = nullptr;
    int rc = shared_table(ctx, (TK_W << 56) | log_n, &T, [&](void** made) {
        size_t n = (size_t)1 << log_n;
        fe* W = nullptr;
        PK_HIP(ctx, hipMalloc((void**)&W, 32 * (n < 2 ? 2 : n)));
        twiddle_init_kernel<<<1, 64, 0, ctx->stream>>>(W, log_n);
        for (size_t h = 2; h < n; h <<= 1) {
            twiddle_seed_kernel<<<1, 64, 0, ctx->stream>>>(W, h);
            unsigned grid = (unsigned)((h + 255) / 256);
            twiddle_double_kernel<<<grid, 256, 0, ctx->stream>>>(W, h);
        }
        PK_LAUNCH_CHECK(ctx);
        *made = W;
        return (int)PK_OK;
    });
    if (rc) return rc;
    ctx->twiddles[log_n] = T;
    *out = (const fe*)T;
    return PK_OK;
}

int get_twiddles_scaled(pk_ctx* ctx, unsigned log_n, const fe* W, const fe** out) {
    auto it = ctx->twiddles_scaled.find(log_n);
    if (it != ctx->twiddles_scaled.end()) {
        *out = (const fe*)it->second;
        return PK_OK;
    }
    void* T = nullptr;
    int rc = shared_table(ctx, (TK_WS << 56) | log_n, &T, [&](void** made) {
        const size_t n = (size_t)1 << log_n;
        fe* Ws = nullptr;
        PK_HIP(ctx, hipMalloc((void**)&Ws, 32 * (n < 2 ? 2 : n)));
        twiddle_scale_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(W, Ws, n);
        PK_LAUNCH_CHECK(ctx);
        *made = Ws;
        return (int)PK_OK;
    });
    if (rc) return rc;
    ctx->twiddles_scaled[log_n] = T;
    *out = (const fe*)T;
    return PK_OK;
}

int get_twiddles29(pk_ctx* ctx, unsigned log_n, int which, const fe* W, const u32** out) {
    auto& cache = ctx->twiddles29[which];
    auto it = cache.find(log_n);
    if (it != cache.end()) {
        *out = (const u32*)it->second;
        return PK_OK;
    }
    void* T = nullptr;
    int rc = shared_table(ctx, ((which ? TK_WS29 : TK_W29) << 56) | log_n, &T, [&](void** made) {
        const size_t n = (size_t)1 << log_n;
        u32* U = nullptr;
        PK_HIP(ctx, hipMalloc((void**)&U, 4 * TW29S_WORDS * (n < 2 ? 2 : n)));
        twiddle_shoup_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(W, U, n);
        PK_LAUNCH_CHECK(ctx);
        *made = U;
        return (int)PK_OK;
    });
    if (rc) return rc;
    cache[log_n] = T;
    *out = (const u32*)T;
    return PK_OK;
}

// Pass-ordered tables pay where the size-N multiplier table no longer fits the caches (72 B x 2^19 = 36 MiB > the 32 MiB of L2): there
// the strided gathers cost 2.6x the data traffic of a pass.  Below that both tables are cache-resident and the gathers are free.
// Measured on the 2^22 / 2^21-coefficient encodes (rows 2^19 / 2^18), same box, alternating: threshold 20 -> 1.705 / 0.783 ms,
// 19 -> 1.634 / 0.793, 18 -> 1.641 / 0.814.
constexpr unsigned PASS_TABLE_MIN_LOG_N = 19;
// the pass-ordered table of one (size, pass, variant): key = log_n | pass << 8 | scaled << 12
int get_pass_table(pk_ctx* ctx, unsigned log_n, unsigned pass, bool scaled, const u32* src29, size_t rows_k, size_t row, size_t mul, const u32** out) {
    const unsigned key = log_n | (pass << 8) | ((scaled ? 1u : 0u) << 12);
    auto it = ctx->twiddles_pass.find(key);
    if (it != ctx->twiddles_pass.end()) {
        *out = (const u32*)it->second;
        return PK_OK;
    }
    void* T = nullptr;
    int rc = shared_table(ctx, (TK_PASS << 56) | key, &T, [&](void** made) {
        u32* U = nullptr;
        const size_t n = rows_k * row;
        PK_HIP(ctx, hipMalloc((void**)&U, 4 * TW29S_WORDS * n));
        twiddle_pass_table_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(src29, U, rows_k, row, mul, ((size_t)1 << log_n) - 1);
        PK_LAUNCH_CHECK(ctx);
        *made = U;
        return (int)PK_OK;
    });
    if (rc) return rc;
    ctx->twiddles_pass[key] = T;
    *out = (const u32*)T;
    return PK_OK;
}

// does a pass of radix 2^log_r over a v axis of 2^log_v take the register-radix kernel?
inline bool pass_is_fast(unsigned log_r, unsigned log_v, size_t N) { return log_r >= 3 && log_v < 62 && log_v >= 11 - log_r && N >= 8; }

// elements per lane of the register-radix kernel (log2): four per lane, 512 lanes per 2048-element tile, 118 registers -> four waves per
// SIMD.  (Eight per lane -- radix-8 rounds, one LDS exchange fewer per three stages, 190 registers, two waves per SIMD -- measured 74 %
// VALUBusy against 89 % and 6-9 % slower from 2^19 rows up: profiles/r06_ntt_le_ab.jsonl.)
constexpr int NTT_LE = 2;

template <int LE, int LOG_R, bool CONTIG>
int launch_fast(pk_ctx* ctx, const PassParams& pp, unsigned grid) {
    const size_t lds_bytes = 9 * 2048 * 4;
    // the 72 KiB dynamic-LDS opt-in is a per-function, per-device attribute: set it once per device, not per launch
    static std::atomic<unsigned long long> lds_set{0};
    const unsigned long long dev_bit = 1ull << (ctx->device & 63);
    if (!(lds_set.load(std::memory_order_acquire) & dev_bit)) {
        PK_HIP(ctx, hipFuncSetAttribute((const void*)ntt8_pass_kernel<LE, LOG_R, CONTIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        lds_set.fetch_or(dev_bit, std::memory_order_release);
    }
    ntt8_pass_kernel<LE, LOG_R, CONTIG><<<dim3(grid, 1), 2048 >> LE, lds_bytes, ctx->stream>>>(pp);
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}

template <int LOG_R>
int launch_pass_r(pk_ctx* ctx, const PassParams& p, bool in_r_contig, size_t tiles, unsigned ncols) {
    constexpr int R = 1 << LOG_R;
    // register-radix fast path: needs at least BT8 = 2048/R elements along v per tile and a real twiddle table
    if (pass_is_fast(LOG_R, p.log_v, p.n_mask + 1)) {
        ProfScope prof(ctx, in_r_contig ? "ntt_pass_last" : "ntt_pass");
        const size_t tiles8 = (tiles * BT) >> (11 - LOG_R);
        PassParams pp = p;
        pp.tiles = tiles8;
        pp.ncols = ncols;
        pp.xcd_tiles = (p.Tpass29 != nullptr && tiles8 % 8 == 0) ? 1 : 0;  // the XCD order exists to share the pass table's rows
        PK_REQUIRE(ctx, tiles8 * ncols < ((size_t)1 << 31), "NTT launch too large");
        const unsigned grid = (unsigned)(tiles8 * ncols);
        constexpr int LR = LOG_R >= 3 ? LOG_R : 3;
        return in_r_contig ? launch_fast<NTT_LE, LR, true>(ctx, pp, grid) : launch_fast<NTT_LE, LR, false>(ctx, pp, grid);
    }
    ProfScope prof(ctx, in_r_contig ? "ntt_pass_last" : "ntt_pass");
    size_t lds_bytes = (size_t)(2 * R * BT + 2 * (R / 2 > 0 ? R / 2 : 1)) * 16;
    dim3 grid((unsigned)tiles, ncols);
    if (in_r_contig) {
        PK_HIP(ctx, hipFuncSetAttribute((const void*)ntt_pass_kernel<LOG_R, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        ntt_pass_kernel<LOG_R, true><<<grid, NTHREADS, lds_bytes, ctx->stream>>>(p);
    } else {
        PK_HIP(ctx, hipFuncSetAttribute((const void*)ntt_pass_kernel<LOG_R, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        ntt_pass_kernel<LOG_R, false><<<grid, NTHREADS, lds_bytes, ctx->stream>>>(p);
    }
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}

int launch_pass(pk_ctx* ctx, unsigned log_r, const PassParams& p, bool in_r_contig, size_t tiles, unsigned ncols) {
    switch (log_r) {
        case 0: return launch_pass_r<0>(ctx, p, in_r_contig, tiles, ncols);
        case 1: return launch_pass_r<1>(ctx, p, in_r_contig, tiles, ncols);
        case 2: return launch_pass_r<2>(ctx, p, in_r_contig, tiles, ncols);
        case 3: return launch_pass_r<3>(ctx, p, in_r_contig, tiles, ncols);
        case 4: return launch_pass_r<4>(ctx, p, in_r_contig, tiles, ncols);
        case 5: return launch_pass_r<5>(ctx, p, in_r_contig, tiles, ncols);
        case 6: return launch_pass_r<6>(ctx, p, in_r_contig, tiles, ncols);
        case 7: return launch_pass_r<7>(ctx, p, in_r_contig, tiles, ncols);
        case 8: return launch_pass_r<8>(ctx, p, in_r_contig, tiles, ncols);
        case 9: return launch_pass_r<9>(ctx, p, in_r_contig, tiles, ncols);
    }
    return set_err(ctx, PK_ERR_BAD_ARG, "unsupported radix 2^%u", log_r);
}

}  // namespace

namespace pk {

void ntt_retain_ctx(pk_ctx* ctx) {
    TableStore& S = store_of(ctx->device);
    std::lock_guard<std::mutex> lock(S.mu);
    S.contexts++;
}
// the context forgets its pointers; the device's tables go with its last context
void ntt_release_ctx(pk_ctx* ctx) {
    ctx->twiddles.clear();
    ctx->twiddles_scaled.clear();
    for (auto& c : ctx->twiddles29) c.clear();
    ctx->twiddles_pass.clear();
    TableStore& S = store_of(ctx->device);
    std::lock_guard<std::mutex> lock(S.mu);
    if (S.contexts && --S.contexts == 0) {
        for (auto& kv : S.tables) (void)hipFree(kv.second);
        S.tables.clear();
    }
}

// Can a transform of this size deliver the hash-ready output (every output = 32 * value as a plain integer < p instead of
// the Montgomery image)?  It rides on the LAST inter-pass twiddle multiplication, so the size needs two or more passes and
// that pass must be the register-radix kernel.  A pure function of the size: commit and openings agree on the encoding.
bool ntt_scaled_available(unsigned log_n) {
    if (log_n <= 9 || log_n > 27) return false;
    if (log_n <= 18) {
        unsigned l1 = (log_n + 1) / 2, l2 = log_n - l1;
        return pass_is_fast(l1, l2, (size_t)1 << log_n);
    }
    unsigned l1 = (log_n + 2) / 3, l2 = (log_n - l1 + 1) / 2, l3 = log_n - l1 - l2;
    return pass_is_fast(l2, l3, (size_t)1 << log_n);
}

// Column-batched NTT, natural -> natural.  `ncols` vectors of length N = 2^log_n:
// input column c at in + c*in_col_stride holds `nonzero` leading coefficients (rest is zero
// and is never read); output column c at out + c*out_col_stride.  `scratch` must hold
// ncols columns of N elements (column stride N) when log_n > 9, and may alias nothing.
// scaled_out: deliver the hash-ready encoding (only legal when ntt_scaled_available(log_n)).
int ntt_columns(pk_ctx* ctx, const fe* in, size_t in_col_stride, size_t nonzero, fe* out, size_t out_col_stride, fe* scratch,
                unsigned log_n, unsigned ncols, bool scaled_out) {
    PK_REQUIRE(ctx, log_n <= 27, "NTT size above 2^27");
    PK_REQUIRE(ctx, !scaled_out || ntt_scaled_available(log_n), "hash-ready NTT output is not available at this size");
    const size_t N = (size_t)1 << log_n;
    const fe* W = nullptr;
    int rc = get_twiddles(ctx, log_n, &W);
    if (rc) return rc;
    const fe* Ws = W;
    if (scaled_out && (rc = get_twiddles_scaled(ctx, log_n, W, &Ws))) return rc;
    const u32 *W29 = nullptr, *Ws29 = nullptr;
    if (log_n >= 3) {  // the register-radix kernel's operand tables
        if ((rc = get_twiddles29(ctx, log_n, 0, W, &W29))) return rc;
        Ws29 = W29;
        if (scaled_out && (rc = get_twiddles29(ctx, log_n, 1, Ws, &Ws29))) return rc;
    }
    PassParams p{};
    p.W = W;
    p.Wtw = W;
    p.W29 = W29;
    p.Wtw29 = W29;
    p.lazy_store = 0;
    p.n_mask = N - 1;
    if (log_n <= 9) {
        // single pass: R = N and the tile's batch axis v runs over BT adjacent *columns*
        p.in = in;
        p.out = out;
        p.in_col_stride = 0;
        p.out_col_stride = 0;
        p.in_stride_r = 1;
        p.in_stride_v = in_col_stride;
        p.in_stride_u = 0;
        p.out_stride_r = 1;
        p.out_stride_v = out_col_stride;
        p.out_stride_u = 0;
        p.in_nat_r = 1;
        p.in_nat_v = 0;
        p.in_nat_u = 0;
        p.nonzero = nonzero;
        p.tw_mul = 0;
        p.wr_step = 1;
        unsigned full = ncols / BT, rem = ncols % BT;
        if (full) {
            PassParams q = p;
            q.log_v = 62;  // vblocks = 2^62/BT: u = 0 and vb = blockIdx.x = group of BT columns
            rc = launch_pass(ctx, log_n, q, true, full, 1);
            if (rc) return rc;
        }
        for (unsigned c = ncols - rem; c < ncols; c++) {
            // remainder columns: a tile whose 4 lanes-of-batch all read column c; only b == 0 is stored
            // (handled by pointing the v stride at 0 and letting the 4 copies write the same values)
            PassParams q = p;
            q.in = in + (size_t)c * in_col_stride;
            q.out = out + (size_t)c * out_col_stride;
            q.in_stride_v = 0;
            q.out_stride_v = 0;
            q.log_v = 62;
            rc = launch_pass(ctx, log_n, q, true, 1, 1);
            if (rc) return rc;
        }
        return PK_OK;
    }
    PK_REQUIRE(ctx, scratch != nullptr, "scratch required for N > 512");
    if (log_n <= 18) {
        unsigned l1 = (log_n + 1) / 2, l2 = log_n - l1;
        size_t R1 = (size_t)1 << l1, R2 = (size_t)1 << l2;
        // pass 1: DFT over n1 (stride R2); v = n2; in -> scratch (same layout); twiddle w_N^(k1*n2)
        p.in = in;
        p.out = scratch;
        p.in_col_stride = in_col_stride;
        p.out_col_stride = N;
        p.in_stride_r = R2; p.in_stride_v = 1; p.in_stride_u = 0;
        p.out_stride_r = R2; p.out_stride_v = 1; p.out_stride_u = 0;
        p.log_v = l2;
        p.in_nat_r = R2; p.in_nat_v = 1; p.in_nat_u = 0;
        p.nonzero = nonzero;
        p.tw_mul = 1;
        p.Wtw = Ws;  // the only inter-pass twiddle of a two-pass transform carries the output scaling
        p.Wtw29 = Ws29;
        p.lazy_store = pass_is_fast(l2, l1, N);  // the register-radix kernel takes almost reduced inputs
        p.wr_step = N >> l1;
        if (log_n >= PASS_TABLE_MIN_LOG_N && pass_is_fast(l1, l2, N) && (rc = get_pass_table(ctx, log_n, 1, scaled_out, Ws29, R1, R2, 1, &p.Tpass29))) return rc;
        p.tp_row = R2;
        rc = launch_pass(ctx, l1, p, false, R2 / BT, ncols);
        if (rc) return rc;
        p.Tpass29 = nullptr;
        p.Wtw = W;
        p.Wtw29 = W29;
        p.lazy_store = scaled_out ? 1 : 0;
        // pass 2: DFT over n2 (contiguous); v = k1 (in stride R2, out stride 1); out k2 stride R1
        p.in = scratch;
        p.out = out;
        p.in_col_stride = N;
        p.out_col_stride = out_col_stride;
        p.in_stride_r = 1; p.in_stride_v = R2; p.in_stride_u = 0;
        p.out_stride_r = R1; p.out_stride_v = 1; p.out_stride_u = 0;
        p.log_v = l1;
        p.in_nat_r = 0; p.in_nat_v = 0; p.in_nat_u = 0;
        p.nonzero = 1;  // nat is always 0 < 1: everything is read
        p.tw_mul = 0;
        p.wr_step = N >> l2;
        return launch_pass(ctx, l2, p, true, R1 / BT, ncols);
    }
    // three passes
    unsigned l1 = (log_n + 2) / 3, l2 = (log_n - l1 + 1) / 2, l3 = log_n - l1 - l2;
    size_t R1 = (size_t)1 << l1, R2 = (size_t)1 << l2, R3 = (size_t)1 << l3;
    // pass 1: DFT over n1 (stride R2*R3); v = m = n2*R3+n3; twiddle w_N^(k1*m); in -> scratch
    p.in = in;
    p.out = scratch;
    p.in_col_stride = in_col_stride;
    p.out_col_stride = N;
    p.in_stride_r = R2 * R3; p.in_stride_v = 1; p.in_stride_u = 0;
    p.out_stride_r = R2 * R3; p.out_stride_v = 1; p.out_stride_u = 0;
    p.log_v = l2 + l3;
    p.in_nat_r = R2 * R3; p.in_nat_v = 1; p.in_nat_u = 0;
    p.nonzero = nonzero;
    p.tw_mul = 1;
    p.lazy_store = pass_is_fast(l2, l3, N);
    p.wr_step = N >> l1;
    if (log_n >= PASS_TABLE_MIN_LOG_N && pass_is_fast(l1, l2 + l3, N) && (rc = get_pass_table(ctx, log_n, 1, false, W29, R1, R2 * R3, 1, &p.Tpass29))) return rc;
    p.tp_row = R2 * R3;
    rc = launch_pass(ctx, l1, p, false, (R2 * R3) / BT, ncols);
    if (rc) return rc;
    p.Tpass29 = nullptr;
    // pass 2: DFT over n2 (stride R3); v = n3; u = k1 (stride R2*R3); in place; twiddle w_N^(R1*k2*n3)
    p.in = scratch;
    p.out = scratch;
    p.in_col_stride = N;
    p.out_col_stride = N;
    p.in_stride_r = R3; p.in_stride_v = 1; p.in_stride_u = R2 * R3;
    p.out_stride_r = R3; p.out_stride_v = 1; p.out_stride_u = R2 * R3;
    p.log_v = l3;
    p.in_nat_r = 0; p.in_nat_v = 0; p.in_nat_u = 0;
    p.nonzero = 1;
    p.tw_mul = R1;
    p.Wtw = Ws;  // the last inter-pass twiddle carries the output scaling
    p.Wtw29 = Ws29;
    p.lazy_store = pass_is_fast(l3, l1, N);
    p.wr_step = N >> l2;
    if (log_n >= PASS_TABLE_MIN_LOG_N && pass_is_fast(l2, l3, N) && (rc = get_pass_table(ctx, log_n, 2, scaled_out, Ws29, R2, R3, R1, &p.Tpass29))) return rc;
    p.tp_row = R3;
    rc = launch_pass(ctx, l2, p, false, R1 * (R3 / BT), ncols);
    if (rc) return rc;
    p.Tpass29 = nullptr;
    p.Wtw = W;
    p.Wtw29 = W29;
    p.lazy_store = scaled_out ? 1 : 0;
    // pass 3: DFT over n3 (contiguous); v = k1 (in stride R2*R3, out stride 1); u = k2 (in stride R3, out stride R1)
    p.in = scratch;
    p.out = out;
    p.in_col_stride = N;
    p.out_col_stride = out_col_stride;
    p.in_stride_r = 1; p.in_stride_v = R2 * R3; p.in_stride_u = R3;
    p.out_stride_r = R1 * R2; p.out_stride_v = 1; p.out_stride_u = R1;
    p.log_v = l1;
    p.tw_mul = 0;
    p.wr_step = N >> l3;
    return launch_pass(ctx, l3, p, true, R2 * (R1 / BT), ncols);
}

int deinterleave(pk_ctx* ctx, const fe* coeffs, size_t n_coeffs, unsigned fold, fe* S, size_t col_stride) {
    size_t fw = (size_t)1 << fold;
    size_t L = n_coeffs / fw;
    PK_REQUIRE(ctx, fw <= 1024, "fold too large");
    size_t TT = 1024 / fw;
    ProfScope prof(ctx, "deinterleave");
    unsigned grid = (unsigned)((L + TT - 1) / TT);
    deinterleave_kernel<1024><<<grid, 256, 1024 * 32, ctx->stream>>>(coeffs, S, L, (unsigned)fw, col_stride);
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}

}  // namespace pk

extern "C" {

int pk_ntt(pk_ctx* ctx, const uint64_t* d_in, uint64_t* d_out, unsigned log_n, unsigned ncols) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, d_in && d_out, "null pointer");
    PK_REQUIRE(ctx, ncols >= 1, "ncols must be >= 1");
    size_t N = (size_t)1 << log_n;
    fe* scratch = nullptr;
    if (log_n > 9) PK_HIP(ctx, hipMalloc((void**)&scratch, 32 * N * ncols));
    int rc = pk::ntt_columns(ctx, (const fe*)d_in, N, N, (fe*)d_out, N, scratch, log_n, ncols, false);
    if (scratch) {
        hipError_t e = wait_ctx(ctx);
        (void)hipFree(scratch);
        if (!rc && e != hipSuccess) rc = set_err(ctx, PK_ERR_HIP, "ntt failed: %s", hipGetErrorString(e));
    }
    return rc;
}

}  // extern "C"

namespace pk {
// the two encodes with a choice of output encoding (scaled = hash-ready, see ntt_scaled_available); the C entry points
// below always return Montgomery images
int rs_encode_x(pk_ctx* ctx, const uint64_t* const* d_coeffs, unsigned batch, unsigned n_vars, unsigned log_inv_rate, unsigned fold,
                uint64_t* d_leaves, uint64_t* d_scratch, bool scaled) {
    PK_REQUIRE(ctx, d_coeffs && d_leaves && d_scratch, "null pointer");
    PK_REQUIRE(ctx, batch >= 1 && batch <= 16, "batch out of range");
    PK_REQUIRE(ctx, fold <= n_vars && fold <= 8, "fold out of range");
    PK_REQUIRE(ctx, n_vars + log_inv_rate >= fold && n_vars + log_inv_rate - fold <= 27, "domain too large (two-adicity 28)");
    unsigned log_rows = n_vars + log_inv_rate - fold;
    size_t rows = (size_t)1 << log_rows, fw = (size_t)1 << fold;
    size_t L = ((size_t)1 << n_vars) / fw;
    fe* S = (fe*)d_scratch;                 // first half: de-interleaved coefficients, [batch*fw][rows]
    fe* S2 = S + (size_t)batch * fw * rows;  // second half: inter-pass buffer
    for (unsigned b = 0; b < batch; b++) {
        PK_REQUIRE(ctx, d_coeffs[b], "null polynomial pointer");
        int rc = pk::deinterleave(ctx, (const fe*)d_coeffs[b], (size_t)1 << n_vars, fold, S + (size_t)b * fw * rows, rows);
        if (rc) return rc;
    }
    return pk::ntt_columns(ctx, S, rows, L, (fe*)d_leaves, rows, S2, log_rows, (unsigned)(batch * fw), scaled);
}

// rows of shard g only: d_leaves_local = column-major [batch*2^fold][rows/G]; local row t is global leaf g + G t.
// d_scratch: (batch*2^fold) * (rows + 2*rows/G) FEs.
int rs_encode_shard_x(pk_ctx* ctx, const uint64_t* const* d_coeffs, unsigned batch, unsigned n_vars, unsigned log_inv_rate, unsigned fold,
                      unsigned shard, unsigned n_shards, uint64_t* d_leaves_local, uint64_t* d_scratch, bool scaled) {
    PK_REQUIRE(ctx, d_coeffs && d_leaves_local && d_scratch, "null pointer");
    PK_REQUIRE(ctx, batch >= 1 && batch <= 16, "batch out of range");
    PK_REQUIRE(ctx, fold <= n_vars && fold <= 8, "fold out of range");
    PK_REQUIRE(ctx, n_vars + log_inv_rate >= fold && n_vars + log_inv_rate - fold <= 27, "domain too large (two-adicity 28)");
    PK_REQUIRE(ctx, is_pow2(n_shards) && shard < n_shards, "n_shards must be a power of two and shard < n_shards");
    const unsigned log_rows = n_vars + log_inv_rate - fold, log_g = ilog2(n_shards);
    PK_REQUIRE(ctx, log_g <= log_rows, "more shards than leaves");
    if (n_shards == 1) return rs_encode_x(ctx, d_coeffs, batch, n_vars, log_inv_rate, fold, d_leaves_local, d_scratch, scaled);
    const size_t rows = (size_t)1 << log_rows, fw = (size_t)1 << fold, Ng = rows >> log_g;
    const size_t L = ((size_t)1 << n_vars) / fw;
    const unsigned ncols = (unsigned)(batch * fw);
    fe* S = (fe*)d_scratch;               // [ncols][rows]: de-interleaved coefficients (first L of each column)
    fe* Y = S + (size_t)ncols * rows;     // [ncols][Ng]: pre-stepped input of the local NTT
    fe* S2 = Y + (size_t)ncols * Ng;      // [ncols][Ng]: inter-pass buffer
    for (unsigned b = 0; b < batch; b++) {
        PK_REQUIRE(ctx, d_coeffs[b], "null polynomial pointer");
        int rc = pk::deinterleave(ctx, (const fe*)d_coeffs[b], (size_t)1 << n_vars, fold, S + (size_t)b * fw * rows, rows);
        if (rc) return rc;
    }
    const fe* W = nullptr;
    int rc = get_twiddles(ctx, log_rows, &W);
    if (rc) return rc;
    {
        ProfScope prof(ctx, "ntt_shard_prestep");
        dim3 grid((unsigned)std::min<size_t>((Ng + 255) / 256, 1024), ncols);
        ntt_shard_prestep_kernel<<<grid, 256, 0, ctx->stream>>>(S, rows, L, Y, Ng, Ng, shard, n_shards, W, rows - 1);
    }
    PK_LAUNCH_CHECK(ctx);
    return pk::ntt_columns(ctx, Y, Ng, Ng, (fe*)d_leaves_local, Ng, S2, log_rows - log_g, ncols, scaled);
}
}  // namespace pk

extern "C" {

int pk_rs_encode(pk_ctx* ctx, const uint64_t* const* d_coeffs, unsigned batch, unsigned n_vars, unsigned log_inv_rate,
                 unsigned fold, uint64_t* d_leaves, uint64_t* d_scratch) {
    PK_ENTER(ctx);
    return pk::rs_encode_x(ctx, d_coeffs, batch, n_vars, log_inv_rate, fold, d_leaves, d_scratch, false);
}

int pk_rs_encode_shard(pk_ctx* ctx, const uint64_t* const* d_coeffs, unsigned batch, unsigned n_vars, unsigned log_inv_rate,
                       unsigned fold, unsigned shard, unsigned n_shards, uint64_t* d_leaves_local, uint64_t* d_scratch) {
    PK_ENTER(ctx);
    return pk::rs_encode_shard_x(ctx, d_coeffs, batch, n_vars, log_inv_rate, fold, shard, n_shards, d_leaves_local, d_scratch, false);
}

}  // extern "C"
