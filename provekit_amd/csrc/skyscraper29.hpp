// skyscraper29.hpp -- Skyscraper compression on the 9x29-bit representation.
//
// Semantics as skyscraper.hpp (skyscraper/core/src/reference.rs:41-98, generic.rs:77-102, v1.rs:19-32);
// this is the plain path: the reference's structure on the 29-bit limbs, state in the true domain, one lazy reduction per
// round.  The kernels run skyscraper29s.hpp (state scaled by 32, fused rounds); this file stays as the host/device cross-check
// of that faster path (pk_selftest_arith ops 1, 2, 13) and supplies its shared pieces (rc limbs, quotient estimate, bar).  State values are canonical-domain integers kept
// "almost reduced" (< p(1 + 2^-10), limbs normalized) between rounds, the lazy-reduction idea of
// skyscraper/core/src/reduce.rs:35-55 (table of multiples indexed by the top limb) restated for 29-bit
// limbs: subtract floor(top/(p_top+1)) * p, one signed carry sweep.
#pragma once
#include "fe29.hpp"
#include "skyscraper.hpp"

namespace pk {

// limb k (29-bit) of round constant RCI (skyscraper/core/src/constants.rs:30-49)
PK_HD constexpr u32 rc29(int rci, int k) {
    int bit = 29 * k, wi = bit >> 5, sh = bit & 31;
    u64 lo = rc_limb(rci, wi < 8 ? wi : 7);
    if (wi >= 8) lo = 0;
    u64 hi = wi + 1 < 8 ? rc_limb(rci, wi + 1) : 0;
    return (u32)(((lo | (hi << 32)) >> sh)) & M29;
}

// bar on an almost-reduced value: canonicalise, swap halves, byte S-box; the result is normalized and < 2.3p (one clamped
// quotient step) -- enough for the round's own reduce_almost29, which sees r + bar + rc < 4.3p (quotient <= 4)
PK_HD fe29 bar29(const fe29& l) {
    fe x = pack29(cond_sub_p29(l));
    fe y;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        y.v[i] = sbox4(x.v[i + 4]);
        y.v[i + 4] = sbox4(x.v[i]);
    }
    fe29 r = unpack29<0>(y);  // < 2^256 < 5.3p
    u32 q = quot_estimate29(r.v[8]);
    u32 q1 = q > 3u ? 3u : q;  // keep q*p_k inside the signed 32-bit limb range; leaves < 2.3p
    sub_qp29(r, q1);
    normalize29(r);
    return r;
}

template <int RCI, bool BAR>
PK_HD void sky_round29(fe29& l, fe29& r) {
    fe29 f = BAR ? bar29(l) : sqr256_29(l);  // sqr: < l^2/2^256 + p < 1.2p;  bar: < 2.3p
    fe29 s;
#pragma unroll
    for (int k = 0; k < 9; k++) s.v[k] = r.v[k] + f.v[k] + ((RCI != 0 && RCI != 17) ? rc29(RCI, k) : 0u);
    reduce_almost29(s);  // s < 4.3p  ->  quotient estimate <= 4 (4 p_k < 2^31: the signed limb sweep still holds)
    r = l;
    l = s;
}

// l, r: normalized, almost reduced.  CANONICAL = true returns the canonical digest (< p, normalized); false returns it
// almost reduced only -- enough when the digest feeds the next compression of a leaf's left fold, and cheaper by three exact
// conditional subtractions.
template <int VERSION, bool CANONICAL = true>
PK_HD fe29 compress29(const fe29& l_in, const fe29& r_in) {
    fe29 l = l_in, r = r_in;
    if (VERSION == 2) {  // generic.rs:77-102
        sky_round29<0, false>(l, r);
        sky_round29<1, false>(l, r);
        sky_round29<2, false>(l, r);
        sky_round29<3, false>(l, r);
        sky_round29<4, false>(l, r);
        sky_round29<5, false>(l, r);
        sky_round29<6, true>(l, r);
        sky_round29<7, true>(l, r);
        sky_round29<8, false>(l, r);
        sky_round29<9, false>(l, r);
        sky_round29<10, true>(l, r);
        sky_round29<11, true>(l, r);
        sky_round29<12, false>(l, r);
        sky_round29<13, false>(l, r);
        sky_round29<14, false>(l, r);
        sky_round29<15, false>(l, r);
        sky_round29<16, false>(l, r);
        sky_round29<17, false>(l, r);
    } else {  // v1.rs:19-32
        sky_round29<0, false>(l, r);
        sky_round29<1, false>(l, r);
        sky_round29<2, true>(l, r);
        sky_round29<3, true>(l, r);
        sky_round29<4, false>(l, r);
        sky_round29<5, false>(l, r);
        sky_round29<6, true>(l, r);
        sky_round29<7, true>(l, r);
        sky_round29<8, false>(l, r);
        sky_round29<0, false>(l, r);
    }
    if (!CANONICAL) {  // l + l_in < 2.01 p: one lazy step
        fe29 s = add29(l, l_in);
        reduce_almost29(s);
        return s;
    }
    // out = l + l_in mod p, exactly
    fe29 a = cond_sub_p29(l), b = cond_sub_p29(l_in);
    fe29 s = add29(a, b);
    normalize29(s);
    return cond_sub_p29(s);
}

// any 256-bit value -> almost reduced fe29
PK_HD fe29 unpack_reduce29(const fe& x) {
    fe29 r = unpack29<0>(x);
    u32 q = quot_estimate29(r.v[8]);
    u32 q1 = q > 3u ? 3u : q;
    sub_qp29(r, q1);
    normalize29(r);
    reduce_almost29(r);
    return r;
}
// Montgomery -> canonical (into_bigint()): x * 2^-256 mod p for x < p; result < p, normalized
PK_HD fe29 from_mont29(const fe& x) {
    fe29 a = unpack29<0>(x);
    u64 acc[17];
#pragma unroll
    for (int k = 0; k < 9; k++) acc[k] = a.v[k];
#pragma unroll
    for (int k = 9; k < 17; k++) acc[k] = 0;
    return reduce256_29(acc);
}

}  // namespace pk
