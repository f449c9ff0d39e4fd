// skyscraper.hpp -- Skyscraper two-to-one compression on gfx950.
//
// Semantics: skyscraper/core/src/reference.rs:41-98 (v2) and v1.rs:19-32 (v1; only
// needed to replay the reference's stale proof fixture).  Structure follows
// skyscraper/core/src/generic.rs:77-102: one lane computes one compression; the
// 18 Feistel rounds are fully unrolled so every round constant is an immediate.
//
// Values are canonical integers (NOT Montgomery): the hash squares with a
// Montgomery product, which is exactly sq(x) = x^2 * 2^-256 mod p
// (reference.rs:22-26: SIGMA_INV == 2^-256).
#pragma once
#include "fe.hpp"

namespace pk {

// skyscraper/core/src/constants.rs:30-49, as 8 x u32 little-endian limbs
__host__ __device__ __forceinline__ constexpr u32 rc_limb(int rc, int i) {
    constexpr u64 RC[18][4] = {
        {0x0000000000000000ULL, 0x0000000000000000ULL, 0x0000000000000000ULL, 0x0000000000000000ULL},
        {0x903c4324270bd744ULL, 0x873125f708a7d269ULL, 0x081dd27906c83855ULL, 0x276b1823ea6d7667ULL},
        {0x7ac8edbb4b378d71ULL, 0xe29d79f3d99e2cb7ULL, 0x751417914c1a5a18ULL, 0x0cf02bd758a484a6ULL},
        {0xfa7adc6769e5bc36ULL, 0x1c3f8e297cca387dULL, 0x0eb7730d63481db0ULL, 0x25b0e03f18ede544ULL},
        {0x57847e652f03cfb7ULL, 0x33440b9668873404ULL, 0x955a32e849af80bcULL, 0x002882fcbe14ae70ULL},
        {0x979231396257d4d7ULL, 0x29989c3e1b37d3c1ULL, 0x12ef02b47f1277baULL, 0x039ad8571e2b7a9cULL},
        {0xb5b48465abbb7887ULL, 0xa72a6bc5e6ba2d2bULL, 0x4cd48043712f7b29ULL, 0x1142d5410fc1fc1aULL},
        {0x7ab2c156059075d3ULL, 0x17cb3594047999b2ULL, 0x44f2c93598f289f7ULL, 0x1d78439f69bc0becULL},
        {0x05d7a965138b8edbULL, 0x36ef35a3d55c48b1ULL, 0x8ddfb8a1ac6f1628ULL, 0x258588a508f4ff82ULL},
        {0x1596fb9afccb49e9ULL, 0x9a7367d69a09a95bULL, 0x9bc43f6984e4c157ULL, 0x13087879d2f514feULL},
        {0x295ccd233b4109faULL, 0xe1d72f89ed868012ULL, 0x2e9e1eea4bc88a8eULL, 0x17dadee898c45232ULL},
        {0x9a8590b4aa1f486fULL, 0xb75834b430e9130eULL, 0xb8e90b1034d5de31ULL, 0x295c6d1546e7f4a6ULL},
        {0x850adcb74c6eb892ULL, 0x07699ef305b92fc3ULL, 0x4ef96a2ba1720f2dULL, 0x1288ca0e1d3ed446ULL},
        {0x01960f9349d1b5eeULL, 0x8ccad30769371c69ULL, 0xe5c81e8991c98662ULL, 0x17563b4d1ae023f3ULL},
        {0x6ba01e9476b32917ULL, 0xa1cb0a3add977bc9ULL, 0x86815a945815f030ULL, 0x2869043be91a1eeaULL},
        {0x81776c885511d976ULL, 0x7475d34f47f414e7ULL, 0x5d090056095d96cfULL, 0x14941f0aff59e79aULL},
        {0xbc40b4fd8fc8c034ULL, 0xbb7142c3cce4fd48ULL, 0x318356758a39005aULL, 0x1ce337a190f4379fULL},
        {0x0000000000000000ULL, 0x0000000000000000ULL, 0x0000000000000000ULL, 0x0000000000000000ULL},
    };
    return (u32)(RC[rc][i >> 1] >> ((i & 1) * 32));
}

// byte-wise S-box on 4 packed bytes (skyscraper/core/src/bar.rs:40-42,63-67)
__host__ __device__ __forceinline__ u32 sbox4(u32 v) {
    u32 t1 = ((v & 0x80808080u) >> 7) | ((v & 0x7f7f7f7fu) << 1);
    u32 t2 = ((v & 0xc0c0c0c0u) >> 6) | ((v & 0x3f3f3f3fu) << 2);
    u32 t3 = ((v & 0xe0e0e0e0u) >> 5) | ((v & 0x1f1f1f1fu) << 3);
    u32 x = (~t1 & t2 & t3) ^ v;
    return ((x & 0x80808080u) >> 7) | ((x & 0x7f7f7f7fu) << 1);
}

// bar: canonical x in [0,p) -> canonical [0,p)   (reference.rs:80-94, bar.rs:15-31)
__device__ __forceinline__ fe bar(const fe& x) {
    fe y;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        y.v[i] = sbox4(x.v[i + 4]);
        y.v[i + 4] = sbox4(x.v[i]);
    }
    return fe_reduce_any(y);
}

// (l, r) <- (r + F(l) + RC, l), all values kept canonical in [0,p)
template <int RCI, bool BAR>
__device__ __forceinline__ void sky_round(fe& l, fe& r) {
    fe f = BAR ? bar(l) : fe_sqr(l);
    // s = r + f + rc < 3p < 2^256
    fe s;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 t = (u64)r.v[i] + f.v[i] + c;
        s.v[i] = (u32)t;
        c = (u32)(t >> 32);
    }
    if (RCI != 0 && RCI != 17) {
        c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            u64 t = (u64)s.v[i] + rc_limb(RCI, i) + c;
            s.v[i] = (u32)t;
            c = (u32)(t >> 32);
        }
        s = cond_sub_kp<2>(s);
    }
    r = l;
    l = cond_sub_kp<1>(s);
}

// Skyscraper v2 compress on canonical inputs < p (generic.rs:77-102)
__device__ __forceinline__ fe compress_reduced(const fe& l_in, const fe& r_in) {
    fe l = l_in, r = r_in;
    sky_round<0, false>(l, r);
    sky_round<1, false>(l, r);
    sky_round<2, false>(l, r);
    sky_round<3, false>(l, r);
    sky_round<4, false>(l, r);
    sky_round<5, false>(l, r);
    sky_round<6, true>(l, r);
    sky_round<7, true>(l, r);
    sky_round<8, false>(l, r);
    sky_round<9, false>(l, r);
    sky_round<10, true>(l, r);
    sky_round<11, true>(l, r);
    sky_round<12, false>(l, r);
    sky_round<13, false>(l, r);
    sky_round<14, false>(l, r);
    sky_round<15, false>(l, r);
    sky_round<16, false>(l, r);
    sky_round<17, false>(l, r);
    return fe_add(l, l_in);
}
// any 256-bit inputs (the reference accepts them: generic.rs:81-82 reduce_partial)
__device__ __forceinline__ fe compress_any(const fe& l, const fe& r) {
    return compress_reduced(fe_reduce_any(l), fe_reduce_any(r));
}

// Skyscraper v1 (v1.rs:19-32): 10 rounds [sq,sq,bar,bar,sq,sq,bar,bar,sq,sq], RC[1..8]
__device__ __forceinline__ fe compress_v1_reduced(const fe& l_in, const fe& r_in) {
    fe l = l_in, r = r_in;
    sky_round<0, false>(l, r);
    sky_round<1, false>(l, r);
    sky_round<2, true>(l, r);
    sky_round<3, true>(l, r);
    sky_round<4, false>(l, r);
    sky_round<5, false>(l, r);
    sky_round<6, true>(l, r);
    sky_round<7, true>(l, r);
    sky_round<8, false>(l, r);
    sky_round<0, false>(l, r);
    return fe_add(l, l_in);
}

template <int VERSION>
__device__ __forceinline__ fe compress_v(const fe& l, const fe& r) {
    if (VERSION == 1) return compress_v1_reduced(l, r);
    return compress_reduced(l, r);
}

}  // namespace pk
