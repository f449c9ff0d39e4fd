// fe52.hpp -- PROTOTYPE of the reference's second multiplier on gfx950: the f64-FMA Montgomery square on 5 limbs x 52 bits.
//
// What it restates (algorithm only, written for one lane of a wavefront instead of a NEON pair):
//   skyscraper/block-multiplier/src/portable_simd.rs:17-196   simd_sqr: 25 (here 15) products as hi/lo FMA pairs, the RHO fold
//   skyscraper/block-multiplier/src/utils.rs:66-147           make_initial (bias bookkeeping), smult_noinit_simd, reduce_ct_simd
//   skyscraper/block-multiplier/src/constants.rs:100-133      RHO_1..4 = 2^(-52k) mod p, C1 = 2^104, C2 = 2^104 + 2^52
//   skyscraper/fp-rounding/src/lib.rs:57-78                   the computation must run with the FPU rounding toward zero
//
// The trick: for integers x, y < 2^52 held exactly in doubles,
//     hi = fma(x, y, 2^104)            under RTZ is 2^104 + floor(xy / 2^52) * 2^52: its mantissa field IS the high half,
//     lo = fma(x, y, 2^104 + 2^52 - hi) is exactly 2^52 + (xy mod 2^52):         its mantissa field IS the low half,
// so both halves are accumulated as 64-bit integers straight from the doubles' bit patterns; the exponent fields add up to a
// constant per column that is cancelled by that column's initial value.
//
// Why this file exists: VERDICT r02 item 1 asks for the reference's "lever" to be measured against the integer multiplier of
// fe29.hpp on this chip.  The answer is a loss: v_fma_f64 / v_add_f64 issue at 4.3-4.4 cycles per wave
// and v_lshl_add_u64 at 5.1, v_mad_u64_u32 at 5.5 -- a 52x52-bit product costs 2 FMA + 1 subtract + 2 wide adds = 23 cycles
// where the 29-bit limbs get the same 2704 bit^2 from 3.2 multiply-adds = 18 cycles with the accumulation included.  On NEON
// the FMA pipes are 2-4x wider than the 64-bit integer multiplier; on CDNA4 both are quarter-rate VALU operations.
// Nothing in the product path uses this header; it is reachable only through pk_probe_fp52_* (tools/probes/probes.hip).
#pragma once
#include "fe29.hpp"

#include <cfenv>

namespace pk {

constexpr u64 M52 = (1ull << 52) - 1;
constexpr u64 F52_EXP_HI = 0x467ull << 52;  // exponent field of a double in [2^104, 2^105)
constexpr u64 F52_EXP_LO = 0x433ull << 52;  // exponent field of a double in [2^52, 2^53)

// p, -p^-1 mod 2^52 and 2^(-52k) mod p in 52-bit limbs (block-multiplier/src/constants.rs:100-133 hold the same numbers)
PK_HD constexpr u64 p52(int k) {
    constexpr u64 P[5] = {0x1f593f0000001ull, 0x4879b9709143eull, 0x181585d2833e8ull, 0xa029b85045b68ull, 0x30644e72e131ull};
    return P[k];
}
constexpr u64 NP52 = 0x1f593efffffffull;
PK_HD constexpr u64 rho52(int k, int j) {  // limb j of 2^(-52k) mod p, k = 1..4
    constexpr u64 R[4][5] = {
        {0x82e644ee4c3d2ull, 0xf93893c98b1deull, 0xd46fe04d0a4c7ull, 0x8f0aad55e2a1full, 0x005ed0447de83ull},
        {0x74eccce9a797aull, 0x16ddcc30bd8a4ull, 0x49ecd3539499eull, 0xb23a6fcc592b8ull, 0x00e3bd49f6ee5ull},
        {0x0e8c656567d77ull, 0x430d05713ae61ull, 0xea3ba6b167128ull, 0xa7dae55c5a296ull, 0x01b4afd513572ull},
        {0x22e2400e2f27dull, 0x323b46ea19686ull, 0xe6c43f0df672dull, 0x7824014c39e8bull, 0x00c6b48afe1b8ull}};
    return R[k - 1][j];
}

// ---- the three floating-point operations ---------------------------------------------------------------------------------
// Device: plain builtins; the kernel sets MODE.FP_ROUND (double) to round-toward-zero before any of them (f52_enter_rtz).
// Host (tests): the same expressions under fesetround(FE_TOWARDZERO), kept where they are by FENV_ACCESS.
#if defined(__HIP_DEVICE_COMPILE__)
PK_HD double f52_fma(double x, double y, double z) { return __builtin_fma(x, y, z); }
PK_HD double f52_sub(double x, double y) { return x - y; }
#else
#pragma STDC FENV_ACCESS ON
static inline double f52_fma(double x, double y, double z) { return __builtin_fma(x, y, z); }
static inline double f52_sub(double x, double y) { return x - y; }
#pragma STDC FENV_ACCESS OFF
#endif
PK_HD u64 f52_bits(double x) { return __builtin_bit_cast(u64, x); }
// exact for x < 2^52 in every rounding mode
PK_HD double f52_from_u52(u64 x) { return f52_sub(__builtin_bit_cast(double, x | F52_EXP_LO), 0x1p52); }

// hwreg(HW_REG_MODE, offset 2, size 2) = the double/half rounding field; 3 = toward zero.  Per-wave state, so a kernel that
// only ever wants RTZ sets it once and never restores it.  Inline asm on purpose: given __builtin_amdgcn_s_setreg the backend's
// SIModeRegister pass sees an unknown mode change and re-establishes round-to-nearest (s_setreg ... 0) in front of the first
// f64 instruction (ROCm 7.2; seen in the ISA).
__device__ __forceinline__ void f52_enter_rtz() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3" ::: "memory");
#endif
}

struct fe52 {
    u64 v[5];  // value = sum v[k] 2^(52k); "normalized": every limb < 2^52
};

// 8 x u32 (value < 2^256) <-> 5 x 52
PK_HD fe52 unpack52(const fe& x) {
    u64 w[4];
#pragma unroll
    for (int i = 0; i < 4; i++) w[i] = (u64)x.v[2 * i] | ((u64)x.v[2 * i + 1] << 32);
    fe52 r;
    r.v[0] = w[0] & M52;
    r.v[1] = ((w[0] >> 52) | (w[1] << 12)) & M52;
    r.v[2] = ((w[1] >> 40) | (w[2] << 24)) & M52;
    r.v[3] = ((w[2] >> 28) | (w[3] << 36)) & M52;
    r.v[4] = w[3] >> 16;
    return r;
}

// bias bookkeeping (utils.rs make_initial): every lo pattern added to a column carries F52_EXP_LO, every hi pattern F52_EXP_HI,
// doubled ones twice.  Returns minus the total for column k of the whole squaring (products + RHO fold + m*p), mod 2^64.
PK_HD constexpr u64 f52_sqr_column_init(int k) {
    u64 n_lo = 0, n_hi = 0;
    for (int i = 0; i < 5; i++)
        for (int j = i; j < 5; j++) {
            const u64 w = i == j ? 1 : 2;
            if (i + j == k) n_lo += w;
            if (i + j + 1 == k) n_hi += w;
        }
    if (k >= 4) {  // four RHO products rows and the m*p row land on columns 4..9: lo of limb (k-4), hi of limb (k-5)
        if (k - 4 <= 4) n_lo += 5;
        if (k - 5 >= 0 && k - 5 <= 4) n_hi += 5;
    }
    return 0ull - (n_lo * F52_EXP_LO + n_hi * F52_EXP_HI);
}

// x (limbs < 2^52, value < 2^257)  ->  x^2 * 2^-260 mod p, lazily reduced: normalized limbs, value < 2^257
// (x^2 < 2^514, the fold s < 2^306 + 4 * 2^52 * p, (s + m p) / 2^52 < 2^256.4 + p).
// Operation count per square: 40 products x (2 FMA + 1 subtract + 2 wide integer adds) + 10 limb conversions + carries.
PK_HD fe52 sqr260_52(const fe52& x) {
    double a[5];
#pragma unroll
    for (int i = 0; i < 5; i++) a[i] = f52_from_u52(x.v[i]);
    u64 t[10];
#pragma unroll
    for (int k = 0; k < 10; k++) t[k] = f52_sqr_column_init(k);
    // 15 products; the off-diagonal ones enter twice (the shift rides on v_lshl_add_u64)
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = i; j < 5; j++) {
            const double hi = f52_fma(a[i], a[j], 0x1p104);
            const double lo = f52_fma(a[i], a[j], f52_sub(0x1p104 + 0x1p52, hi));
            const int sh = i == j ? 0 : 1;
            t[i + j + 1] += f52_bits(hi) << sh;
            t[i + j] += f52_bits(lo) << sh;
        }
    // low four columns to 52 bits each
#pragma unroll
    for (int k = 0; k < 4; k++) t[k + 1] += t[k] >> 52;
    // fold them down: t_k * 2^(52k) = t_k * rho_(4-k) * 2^208 (mod p)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double s = f52_from_u52(t[k] & M52);
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const double c = (double)rho52(4 - k, j);
            const double hi = f52_fma(s, c, 0x1p104);
            const double lo = f52_fma(s, c, f52_sub(0x1p104 + 0x1p52, hi));
            t[4 + j + 1] += f52_bits(hi);
            t[4 + j] += f52_bits(lo);
        }
    }
    // one word-sized Montgomery step: (s + m p) / 2^52
    const u64 m = (t[4] * NP52) & M52;
    const double mf = f52_from_u52(m);
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const double c = (double)p52(j);
        const double hi = f52_fma(mf, c, 0x1p104);
        const double lo = f52_fma(mf, c, f52_sub(0x1p104 + 0x1p52, hi));
        t[4 + j + 1] += f52_bits(hi);
        t[4 + j] += f52_bits(lo);
    }
    fe52 r;
    u64 carry = t[4] >> 52;  // t[4] is now a multiple of 2^52
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u64 u = t[5 + k] + carry;
        r.v[k] = u & M52;
        carry = u >> 52;
    }
    r.v[4] = t[9] + carry;
    return r;
}

}  // namespace pk
