// skyscraper29s.hpp -- Skyscraper compression with the state held SCALED BY 32: the fast path
// of every hashing kernel.  Same function as skyscraper29.hpp / the reference (skyscraper/core/src/reference.rs:41-98,
// generic.rs:77-102, v1.rs:19-32), bit for bit; only the internal representation differs.
//
// Why.  A Feistel round is l' = r + F(l) + rc with F(l) = l^2 * 2^-256 (a Montgomery square) in 14 of 18 rounds.  On 9 limbs of
// 29 bits the natural Montgomery reduction is by 2^261 = 2^(9*29): nine identical steps and no 24-bit realignment.  Holding
// L = 32*l makes that reduction the right one (L^2 * 2^-261 = 32 * l^2 * 2^-256), and it buys two more things:
//   * contraction: a square maps a value of b*p to b^2/169 * p, so the state may grow by ~2p per round for a whole
//     compression without any reduction between rounds (it stays below 31p; the limbs hold 169p);
//   * fusion: r + 32*rc enter the product's accumulator columns 9..17 as initial values, so the round's sum comes out of the
//     reduction's own carry sweep already normalised -- no separate add, quotient estimate, q*p subtraction and second sweep.
// A square round is then 126 v_mad_u64_u32 + ~85 other VALU instructions (skyscraper29.hpp: 126 + ~150).
// The bar rounds need the true value's bytes: bar_s divides by 32 exactly (one 5-bit Montgomery step: add m*p with
// m = -L/p mod 32, shift), applies the byte S-box, and multiplies by 32 again while reducing (subtracting q*p as adding
// q*(2^261 - p) in unsigned 64-bit columns, the 2^261 multiple falling off the top limb).
// Values: "scaled" = 32*x mod p, limbs 0..7 < 2^29, limb 8 holds the rest; between compressions < 1.05p, inside < 31p.
#pragma once
#include "skyscraper29.hpp"

namespace pk {

// ---- compile-time constants ----------------------------------------------------------------------------------------------
struct limbs9 {
    u32 v[9];
};
// 8 x u32 value (< 2^256) -> 9 x 29-bit limbs
PK_HD constexpr limbs9 split29(const u32 (&w)[9]) {
    limbs9 r{};
    for (int k = 0; k < 9; k++) {
        int bit = 29 * k, wi = bit >> 5, sh = bit & 31;
        u64 lo = w[wi < 9 ? wi : 8];
        if (wi >= 9) lo = 0;
        u64 hi = wi + 1 < 9 ? w[wi + 1] : 0;
        r.v[k] = (u32)((lo | (hi << 32)) >> sh) & M29;
    }
    return r;
}
// (32 * RC[i]) mod p
PK_HD constexpr limbs9 make_rcs(int i) {
    u32 x[9] = {};
    for (int k = 0; k < 8; k++) x[k] = rc_limb(i, k);
    for (int d = 0; d < 5; d++) {
        u32 c = 0;
        for (int k = 0; k < 9; k++) {  // x = 2x
            u32 nc = x[k] >> 31;
            x[k] = (x[k] << 1) | c;
            c = nc;
        }
        u32 t[9] = {};
        long long borrow = 0;
        for (int k = 0; k < 9; k++) {  // t = x - p
            long long dlt = (long long)x[k] - (k < 8 ? (long long)kPlimb(k) : 0) + borrow;
            t[k] = (u32)(dlt & 0xffffffffll);
            borrow = dlt < 0 ? -1 : 0;
        }
        if (borrow == 0)
            for (int k = 0; k < 9; k++) x[k] = t[k];
    }
    return split29(x);
}
// 2^261 - p
PK_HD constexpr limbs9 make_pc() {
    limbs9 r{};
    long long borrow = 0;
    for (int k = 0; k < 9; k++) {
        long long d = -(long long)p29(k) + borrow;
        borrow = d < 0 ? -1 : 0;
        r.v[k] = (u32)(d & (long long)M29);
    }
    return r;
}
template <int RCI>
PK_HD constexpr u32 rcs29(int k) {
    constexpr limbs9 c = make_rcs(RCI);
    return c.v[k];
}
PK_HD constexpr u32 pc29(int k) {
    constexpr limbs9 c = make_pc();
    return c.v[k];
}

// ---- conversions -------------------------------------------------------------------------------------------------------
// u - q*p for u < 2^261 with u - q*p in [0, 2^261): computed as u + q*(2^261 - p) in unsigned columns; the q*2^261 falls off
// limb 8's mask.  Input limbs may be lazy (< 2^32), output is normalised.
PK_HD fe29 sub_qp_wide29(const fe29& u, u32 q) {
    fe29 r;
    u64 c = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        c += (u64)q * pc29(k) + u.v[k];
        r.v[k] = (u32)c & M29;
        c >>= 29;
    }
    return r;
}
// any 256-bit value x -> 32*x mod p, scaled form, < 1.04p
PK_HD fe29 to_scaled29(const fe& x) {
    fe29 u = unpack29<5>(x);
    return sub_qp_wide29(u, quot_estimate29(u.v[8]));
}
// Montgomery image (x*2^256 mod p, < p) -> scaled form of the canonical value: x*2^256 * 2^-251 = 32x; < p + 8
PK_HD fe29 mont_to_scaled29(const fe& x) {
    fe29 a = unpack29<0>(x);
    u64 acc[17];
#pragma unroll
    for (int k = 0; k < 9; k++) acc[k] = (u64)a.v[k] << 10;
#pragma unroll
    for (int k = 9; k < 17; k++) acc[k] = 0;
    return reduce261_29(acc);
}
// scaled (normalised, < 33p) -> the canonical value L/32 mod p as 8 x u32: one 5-bit Montgomery step (t = L + m*p is
// divisible by 32, t/32 < 2p), the 5-bit shift folded into the packing, then ONE exact conditional subtraction of p on the eight
// 32-bit words (a borrow chain of 8 subtract-with-borrow instructions and 8 selects: cheaper than the same on nine 29-bit limbs,
// whose signed sweep costs three instructions a limb)
PK_HD fe from_scaled_canon(const fe29& L) {
    const u32 m = (L.v[0] * NP29) & 31u;
    u32 t[9];
    u64 c = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        c += (u64)m * p29(k) + L.v[k];
        t[k] = k < 8 ? ((u32)c & M29) : (u32)c;
        c >>= 29;
    }
    fe w;  // t / 32 < 2p < 2^255
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int bit = 32 * i + 5, k0 = bit / 29, o = bit - 29 * k0;
        u32 word = t[k0] >> o;
        if (k0 + 1 < 9) word |= t[k0 + 1] << (29 - o);
        if (58 - o < 32 && k0 + 2 < 9) word |= t[k0 + 2] << (58 - o);
        w.v[i] = word;
    }
    fe d;
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        d.v[i] = __builtin_subc(w.v[i], kPlimb(i), borrow, &borrow);
    }
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = borrow ? w.v[i] : d.v[i];
    return r;
}

// ---- rounds ------------------------------------------------------------------------------------------------------------
// (l, r) <- (r + l^2 * 2^-261 + 32 rc, l)
template <int RCI>
PK_HD void sky_sq_round_s(fe29& l, fe29& r) {
    u64 acc[17];
#pragma unroll
    for (int k = 0; k < 9; k++) acc[k] = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) acc[9 + k] = (u64)(r.v[k] + rcs29<RCI>(k));
    const u32 top = r.v[8] + rcs29<RCI>(8);
    u32 a2[9];
#pragma unroll
    for (int j = 0; j < 9; j++) a2[j] = l.v[j] << 1;
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
    s.v[8] = (u32)(acc[16] >> 29) + top;
    r = l;
    l = s;
}

// 32 * bar(L / 32): scaled in (normalised, < 33p), scaled lazy sum out: returns r + 32*bar(l) + 32*rc normalised (< r + 2.1p)
template <int RCI>
PK_HD void sky_bar_round_s(fe29& l, fe29& r) {
    fe x = from_scaled_canon(l);
    fe y;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        y.v[i] = sbox4(x.v[i + 4]);
        y.v[i + 4] = sbox4(x.v[i]);
    }
    fe29 u = unpack29<5>(y);  // 32*y < 2^261
    const u32 q = quot_estimate29(u.v[8]);
#pragma unroll
    for (int k = 0; k < 9; k++) u.v[k] += r.v[k] + rcs29<RCI>(k);
    fe29 s = sub_qp_wide29(u, q);
    r = l;
    l = s;
}

// l, r: scaled, normalised, < 1.05p.  Returns the scaled digest, normalised, < 1.04p.
template <int VERSION>
PK_HD fe29 compress29s(const fe29& l_in, const fe29& r_in) {
    fe29 l = l_in, r = r_in;
    if (VERSION == 2) {
        sky_sq_round_s<0>(l, r);
        sky_sq_round_s<1>(l, r);
        sky_sq_round_s<2>(l, r);
        sky_sq_round_s<3>(l, r);
        sky_sq_round_s<4>(l, r);
        sky_sq_round_s<5>(l, r);
        sky_bar_round_s<6>(l, r);
        sky_bar_round_s<7>(l, r);
        sky_sq_round_s<8>(l, r);
        sky_sq_round_s<9>(l, r);
        sky_bar_round_s<10>(l, r);
        sky_bar_round_s<11>(l, r);
        sky_sq_round_s<12>(l, r);
        sky_sq_round_s<13>(l, r);
        sky_sq_round_s<14>(l, r);
        sky_sq_round_s<15>(l, r);
        sky_sq_round_s<16>(l, r);
        sky_sq_round_s<17>(l, r);
    } else {
        sky_sq_round_s<0>(l, r);
        sky_sq_round_s<1>(l, r);
        sky_bar_round_s<2>(l, r);
        sky_bar_round_s<3>(l, r);
        sky_sq_round_s<4>(l, r);
        sky_sq_round_s<5>(l, r);
        sky_bar_round_s<6>(l, r);
        sky_bar_round_s<7>(l, r);
        sky_sq_round_s<8>(l, r);
        sky_sq_round_s<0>(l, r);
    }
    fe29 s = add29(l, l_in);  // < 32p, limbs < 2^30
    return sub_qp_wide29(s, quot_estimate29(s.v[8]));
}

// v2 with round 0 hoisted for a caller that hashes many messages sharing the left input (the proof-of-work grinder):
// s0 = round 0's output for r = 0 (sky_sq_round_s<0> on (l_in, 0)), r_in scaled.  Returns the scaled digest like compress29s.
PK_HD fe29 compress29s_v2_fixed_left(const fe29& l_in, const fe29& s0, const fe29& r_in) {
    fe29 l = add29(s0, r_in), r = l_in;  // limbs < 2^30: within the squaring's column bound
    sky_sq_round_s<1>(l, r);
    sky_sq_round_s<2>(l, r);
    sky_sq_round_s<3>(l, r);
    sky_sq_round_s<4>(l, r);
    sky_sq_round_s<5>(l, r);
    sky_bar_round_s<6>(l, r);
    sky_bar_round_s<7>(l, r);
    sky_sq_round_s<8>(l, r);
    sky_sq_round_s<9>(l, r);
    sky_bar_round_s<10>(l, r);
    sky_bar_round_s<11>(l, r);
    sky_sq_round_s<12>(l, r);
    sky_sq_round_s<13>(l, r);
    sky_sq_round_s<14>(l, r);
    sky_sq_round_s<15>(l, r);
    sky_sq_round_s<16>(l, r);
    sky_sq_round_s<17>(l, r);
    fe29 s = add29(l, l_in);
    return sub_qp_wide29(s, quot_estimate29(s.v[8]));
}

}  // namespace pk
