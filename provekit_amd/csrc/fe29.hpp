// fe29.hpp -- carry-free BN254-Fr arithmetic for gfx950: 9 limbs x 29 bits.
//
// Why: on MI355X v_mad_u64_u32 issues at ~4.6 cycles/wave, but every carry instruction
// (v_add_co/v_addc_co/v_lshl_add_u64) costs ~4 cycles too (profiles/r01_ubench_valu_rates.txt).  A
// 32-bit-limb Montgomery product spends 3x more issue slots on carries and register-pair moves
// than on multiplies.  With 29-bit limbs a column sum of up to 18 partial products (< 2^58 each)
// fits the 64-bit destination of v_mad_u64_u32, so the inner loops are nothing but
// `acc[k] = a*b + acc[k]` in place, and carries are swept once per product.
//
// Same role as block_multiplier::scalar_{mul,sqr} (skyscraper/block-multiplier/src/scalar.rs:12-132)
// and ark-ff's Fp256 multiplier; same contract style: lazily reduced inputs/outputs.
//
// Representation: value = sum v[k] * 2^(29k).  "normalized": every limb < 2^29.
// All functions are __host__ __device__ so the exact device code is unit-tested on the CPU
// (tests/test_fe29_host.py through pk_selftest_*).
#pragma once
#include "fe.hpp"

#define PK_HD __host__ __device__ __forceinline__

// Hide a value's known bits from the optimizer (device only; no instruction is emitted).
// Needed for the 24-bit Montgomery step: ROCm 7.2's AMDGPU mul24 combine drops the `& 0xffffff`
// on the multiplier and then selects a full 32-bit v_mad_u64_u32 on the UNMASKED register for the
// limbs of p that fit 24 bits (seen in the ISA; caught by tests/test_gpu_selftest.py).
#if defined(__HIP_DEVICE_COMPILE__)
#define PK_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define PK_OPAQUE(x) (void)(x)
#endif

namespace pk {

struct fe29 {
    u32 v[9];
};

constexpr u32 M29 = (1u << 29) - 1;
constexpr u32 NP29 = PK_NP0 & M29;  // -p^-1 mod 2^29

// limb k (29-bit) of K*p, K*p < 2^261
PK_HD constexpr u32 kp29(int K, int k) {
    constexpr u32 P[8] = {PK_P0, PK_P1, PK_P2, PK_P3, PK_P4, PK_P5, PK_P6, PK_P7};
    // compute K*p as 9 x 32-bit words then extract bits [29k, 29k+29)
    u32 w[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    u64 c = 0;
    for (int i = 0; i < 8; i++) {
        c += (u64)P[i] * (u32)K;
        w[i] = (u32)c;
        c >>= 32;
    }
    w[8] = (u32)c;
    int bit = 29 * k, wi = bit >> 5, sh = bit & 31;
    u64 window = (u64)w[wi] | ((u64)w[wi + 1] << 32);
    return (u32)(window >> sh) & M29;
}
PK_HD constexpr u32 p29(int k) { return kp29(1, k); }

// ---- packing ---------------------------------------------------------------------------------
// 8 x u32 (256-bit little-endian) -> 9 x 29-bit limbs, optionally pre-shifted left by SHL bits
// (SHL = 5 turns x into 32x for free: mont261(a, 32b) = a*b*2^-256).
template <int SHL>
PK_HD fe29 unpack29(const fe& x) {
    fe29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int bit = 29 * k - SHL;  // bit position in x of limb k's lsb (may be negative for k = 0)
        u32 limb;
        if (bit < 0) {
            limb = (x.v[0] << (-bit)) & M29;
        } else {
            const int wi = bit >> 5, sh = bit & 31;
            u32 lo = wi < 8 ? x.v[wi] : 0u;
            u32 hi = wi + 1 < 8 ? x.v[wi + 1] : 0u;
            limb = sh == 0 ? lo : ((lo >> sh) | (sh > 3 ? (hi << (32 - sh)) : 0u));
            limb &= M29;
        }
        r.v[k] = limb;
    }
    if (SHL > 0) {  // the top limb keeps the bits shifted out beyond 29*9: value < 2^(256+SHL) <= 2^261 fits
    }
    return r;
}
// normalized limbs, value < 2^256  ->  8 x u32
PK_HD fe pack29(const fe29& a) {
    fe r;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const int bit = 32 * w, k0 = bit / 29, o = bit - 29 * k0;
        u32 word = a.v[k0] >> o;
        word |= a.v[k0 + 1] << (29 - o);
        if (58 - o < 32 && k0 + 2 < 9) word |= a.v[k0 + 2] << (58 - o);
        r.v[w] = word;
    }
    return r;
}

// ---- carries ---------------------------------------------------------------------------------
// signed carry sweep: limbs may be "negative" (two's complement) on entry; on exit limbs 0..7 are in
// [0, 2^29) and limb 8 holds the rest (must be non-negative for a non-negative value).
PK_HD void normalize29(fe29& a) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
        int c = (int)a.v[k] >> 29;
        a.v[k] &= M29;
        a.v[k + 1] += (u32)c;
    }
}
PK_HD fe29 add29(const fe29& a, const fe29& b) {  // lazy: no carries
    fe29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) r.v[k] = a.v[k] + b.v[k];
    return r;
}
// a - q*p, limb-wise (signed limbs; follow with normalize29)
PK_HD void sub_qp29(fe29& a, u32 q) {
#pragma unroll
    for (int k = 0; k < 9; k++) a.v[k] -= q * p29(k);
}
// floor-ish estimate of value / p from the (possibly lazy) top limb; never above the true quotient.
// p >> 232 = 0x30644e; magic = floor(2^32 / (0x30644e + 1))
PK_HD u32 quot_estimate29(u32 top_limb) { return (u32)(((u64)top_limb * 1354u) >> 32); }

// value (< ~6p, limbs lazy non-negative) -> normalized and "almost reduced": < p*(1 + 2^-10)
// (the estimate q = top*1354 >> 32 falls short of value/p by at most 2.06e-4 of it, and inputs are < ~4.3p)
PK_HD void reduce_almost29(fe29& a) {
    u32 q = quot_estimate29(a.v[8]);
    sub_qp29(a, q);
    normalize29(a);
}
// exact: normalized value < 2p  ->  canonical [0, p)
PK_HD fe29 cond_sub_p29(const fe29& a) {
    fe29 t;
    int borrow = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        int d = (int)a.v[k] - (int)p29(k) + borrow;
        borrow = d >> 29;
        t.v[k] = (u32)d & M29;
    }
    // after the sweep `borrow` is -1 iff a < p (limb 8 of both is < 2^29, so the sign is exact)
    fe29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) r.v[k] = borrow ? a.v[k] : t.v[k];
    return r;
}

// the same exact reduction delivered as 8 x u32: pack first, then ONE conditional subtraction of p as a borrow chain over the eight
// 32-bit words (8 subtract-with-borrow + 8 selects, against three instructions a limb for the signed sweep above and its select)
PK_HD fe pack_canon29(const fe29& a) {  // normalized value < 2p  ->  canonical [0, p) as 8 x u32
    const fe w = pack29(a);
    fe d;
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d.v[i] = __builtin_subc(w.v[i], kPlimb(i), borrow, &borrow);
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = borrow ? w.v[i] : d.v[i];
    return r;
}

// ---- products ---------------------------------------------------------------------------------
// column sums of a*b.  Requires 9*max(a)*max(b) + 9*2^58 + 2^36 < 2^64 (e.g. limbs < 2^30 each).
PK_HD void mul_cols29(const fe29& a, const fe29& b, u64 (&acc)[17]) {
#pragma unroll
    for (int k = 0; k < 17; k++) acc[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) acc[i + j] += (u64)a.v[i] * b.v[j];
}
PK_HD void sqr_cols29(const fe29& a, u64 (&acc)[17]) {
    u32 a2[9];
#pragma unroll
    for (int j = 0; j < 9; j++) a2[j] = a.v[j] << 1;
#pragma unroll
    for (int k = 0; k < 17; k++) acc[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        acc[2 * i] += (u64)a.v[i] * a.v[i];
#pragma unroll
        for (int j = i + 1; j < 9; j++) acc[i + j] += (u64)a.v[i] * a2[j];
    }
}
// Montgomery reduction by exactly 2^256 = 2^(8*29 + 24): returns (T + M p) / 2^256, normalized,
// value < T / 2^256 + p.   8 steps of 29 bits, one of 24, then the 24-bit realignment.
PK_HD fe29 reduce256_29(u64 (&acc)[17]) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 m = ((u32)acc[i] * NP29) & M29;
#pragma unroll
        for (int j = 0; j < 9; j++) acc[i + j] += (u64)m * p29(j);
        acc[i + 1] += acc[i] >> 29;
    }
    {
        u32 m = ((u32)acc[8] * NP29) & 0xffffffu;
        PK_OPAQUE(m);
#pragma unroll
        for (int j = 0; j < 9; j++) acc[8 + j] += (u64)m * p29(j);
    }
    u32 d[9];
#pragma unroll
    for (int k = 8; k < 16; k++) {
        acc[k + 1] += acc[k] >> 29;
        d[k - 8] = (u32)acc[k] & M29;
    }
    fe29 r;
    const u64 top = acc[16];
    d[8] = (u32)top & M29;
#pragma unroll
    for (int j = 0; j < 8; j++) r.v[j] = (d[j] >> 24) | ((d[j + 1] << 5) & M29);
    r.v[8] = (u32)(top >> 24);
    return r;
}
// Montgomery reduction by 2^261 (9 uniform steps, no realignment): (T + M p) / 2^261
PK_HD fe29 reduce261_29(u64 (&acc)[17]) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
        u32 m = ((u32)acc[i] * NP29) & M29;
#pragma unroll
        for (int j = 0; j < 9; j++)
            if (i + j < 17) acc[i + j] += (u64)m * p29(j);
        if (i + 1 < 17) acc[i + 1] += acc[i] >> 29;
    }
    fe29 r;
#pragma unroll
    for (int k = 9; k < 16; k++) {
        acc[k + 1] += acc[k] >> 29;
        r.v[k - 9] = (u32)acc[k] & M29;
    }
    r.v[7] = (u32)acc[16] & M29;
    r.v[8] = (u32)(acc[16] >> 29);
    return r;
}

// a*b*2^-256 (lazy): inputs with limbs < 2^30, result normalized, < a*b/2^256 + p
PK_HD fe29 mont256_29(const fe29& a, const fe29& b) {
    u64 acc[17];
    mul_cols29(a, b, acc);
    return reduce256_29(acc);
}
PK_HD fe29 sqr256_29(const fe29& a) {
    u64 acc[17];
    sqr_cols29(a, acc);
    return reduce256_29(acc);
}
// a*b*2^-261 (lazy); with b = 32*y (unpack29<5>) this is a*y*2^-256
PK_HD fe29 mont261_29(const fe29& a, const fe29& b) {
    u64 acc[17];
    mul_cols29(a, b, acc);
    return reduce261_29(acc);
}

// ---- multiplication by a CONSTANT with a precomputed quotient (Shoup / Barrett on 29-bit limbs) -------------------------------
// For a multiplier w known in advance (a twiddle, a round's folding challenge) keep wq = floor(w * 2^261 / p) next to it.  Then
//     q = floor(a * wq / 2^261)   is floor(a * w / p) or one or two less (a < 8p; wq may itself be up to 2 short), and
//     r = a * w - q * p           is a * w mod p plus at most two p's: r < 2.2 p,
// and r needs only the LOW nine limbs of a*w and of q*p (it is smaller than 2^261), q only the HIGH columns of a*wq: columns 0..6
// of that product change it by less than 2^-22.  53 + 45 + 45 = 143 multiply-adds and no per-step quotient digit, against the
// 162 + 9 of a Montgomery product (mont261_29) -- and the value stays in whatever domain `a` is in (a Montgomery image times a plain
// w is the Montgomery image of the product).  q*p is subtracted as q*(2^261 - p) added, the 2^261 multiple falling off limb 8.
// a: limbs < 2^30.7 (lazy), value < 8p.  w, wq: normalised (limbs < 2^29).  Result normalised, < 2.2p.
PK_HD constexpr u32 pcomp29(int k) {  // limb k of 2^261 - p
    long long borrow = 0;
    u32 out = 0;
    for (int i = 0; i <= k; i++) {
        long long d = -(long long)p29(i) + borrow;
        borrow = d < 0 ? -1 : 0;
        out = (u32)(d & (long long)M29);
    }
    return out;
}
PK_HD fe29 shoup261_29(const fe29& a, const fe29& w, const fe29& wq) {
    u64 c[10];  // columns 7 .. 16 of a * wq
#pragma unroll
    for (int k = 0; k < 10; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++)
            if (i + j >= 7) c[i + j - 7] += (u64)a.v[i] * wq.v[j];
    c[1] += c[0] >> 29;
    c[2] += c[1] >> 29;
    u32 q[9];
#pragma unroll
    for (int k = 2; k < 9; k++) {
        c[k + 1] += c[k] >> 29;
        q[k - 2] = (u32)c[k] & M29;
    }
    q[7] = (u32)c[9] & M29;
    q[8] = (u32)(c[9] >> 29);
    u64 r[9];
#pragma unroll
    for (int k = 0; k < 9; k++) r[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++)
            if (i + j <= 8) {
                r[i + j] += (u64)a.v[i] * w.v[j];
                r[i + j] += (u64)q[i] * pcomp29(j);
            }
    fe29 out;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        r[k + 1] += r[k] >> 29;
        out.v[k] = (u32)r[k] & M29;
    }
    out.v[8] = (u32)r[8] & M29;
    return out;
}
// wq = floor(w * 2^261 / p) for a normalised w < p: 261 steps of binary long division (tests; tables use the Barrett form)
PK_HD fe29 shoup_quotient29(const fe29& w) {
    u32 r[9], q[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        r[k] = w.v[k];
        q[k] = 0;
    }
    for (int bit = 260; bit >= 0; bit--) {
        u32 carry = 0;  // r = 2r (r < p < 2^254: no overflow of limb 8)
        for (int k = 0; k < 9; k++) {
            const u32 v = (r[k] << 1) | carry;
            carry = v >> 29;
            r[k] = v & M29;
        }
        int borrow = 0;  // t = r - p
        u32 t[9];
        for (int k = 0; k < 9; k++) {
            int d = (int)r[k] - (int)p29(k) + borrow;
            borrow = d >> 29;
            t[k] = (u32)d & M29;
        }
        if (!borrow) {
            for (int k = 0; k < 9; k++) r[k] = t[k];
            q[bit / 29] |= 1u << (bit % 29);
        }
    }
    fe29 out;
#pragma unroll
    for (int k = 0; k < 9; k++) out.v[k] = q[k];
    return out;
}

// ---- sums of products with one reduction per group ------------------------------------------------
// sum_t a_t * b_t (mod p) the cheap way: the 17 column accumulators take the partial products of up to DOT29_GROUP terms
// before ONE Montgomery reduction (81 multiply-adds per term instead of 162).  a_t: any 256-bit value (unpack29<0>),
// b_t = unpack29<5>(y_t) with y_t < p, i.e. 32*y_t: a term adds < 2^61.2 to a column and a*32y/2^261 < 0.19p to the value.
// Column bound: (DOT29_GROUP + 1) * 2^61.2 < 2^64.
constexpr int DOT29_GROUP = 4;
struct dot29 {
    u64 acc[17];
    fe29 run;   // running sum of the reduced groups, almost reduced
    int pending;
};
PK_HD void dot29_init(dot29& d) {
#pragma unroll
    for (int k = 0; k < 17; k++) d.acc[k] = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) d.run.v[k] = 0;
    d.pending = 0;
}
PK_HD void dot29_flush(dot29& d) {
    fe29 r = reduce261_29(d.acc);  // < DOT29_GROUP * 0.19p + p, normalized
    d.run = add29(d.run, r);
    reduce_almost29(d.run);  // < 1.001p + 1.76p before, almost reduced after
#pragma unroll
    for (int k = 0; k < 17; k++) d.acc[k] = 0;
    d.pending = 0;
}
PK_HD void dot29_add(dot29& d, const fe29& a, const fe29& b) {
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) d.acc[i + j] += (u64)a.v[i] * b.v[j];
    if (++d.pending == DOT29_GROUP) dot29_flush(d);
}
// the sum as a fully reduced field element: sum_t a_t * y_t * 2^-256 (the Montgomery product's scaling, as fe_mul29)
PK_HD fe dot29_result(dot29& d) {
    if (d.pending) dot29_flush(d);
    return pack_canon29(d.run);
}

// ---- drop-in for the 8x32 API of fe.hpp ---------------------------------------------------------
// a, b < p (ark-ff semantics) -> a*b*2^-256 mod p, fully reduced
PK_HD fe fe_mul29(const fe& a, const fe& b) {
    fe29 r = mont261_29(unpack29<0>(a), unpack29<5>(b));  // < 32 p^2 / 2^261 + p < 1.2 p
    return pack_canon29(r);
}

PK_HD fe fe_mulx(const fe& a, const fe& b) { return fe_mul29(a, b); }
// a < p -> a^2 * 2^-256 mod p
PK_HD fe fe_sqrx(const fe& a) {
    u64 acc[17];
    sqr_cols29(unpack29<0>(a), acc);
    return pack_canon29(reduce256_29(acc));
}
// Montgomery -> canonical (x < p)
PK_HD fe fe_from_montx(const fe& a) {
    fe29 x = unpack29<0>(a);
    u64 acc[17];
#pragma unroll
    for (int k = 0; k < 9; k++) acc[k] = x.v[k];
#pragma unroll
    for (int k = 9; k < 17; k++) acc[k] = 0;
    return pack_canon29(reduce256_29(acc));
}
// canonical (< p) -> Montgomery: x * R^2 * 2^-256
PK_HD fe fe_to_montx(const fe& a) {
    fe r2;
    r2.v[0] = 0xae216da7u; r2.v[1] = 0x1bb8e645u; r2.v[2] = 0xe35c59e3u; r2.v[3] = 0x53fe3ab1u;
    r2.v[4] = 0x53bb8085u; r2.v[5] = 0x8c49833du; r2.v[6] = 0x7f4e44a5u; r2.v[7] = 0x0216d0b1u;
    return fe_mul29(a, r2);
}

}  // namespace pk
