// feinv.hpp -- modular inverse in BN254-Fr for one lane: Bernstein-Yang "safegcd" division steps (eprint 2019/266) in the
// half-delta form, 30 steps per batch on 32-bit words, the transition matrix applied to (f, g) and, modulo p, to (d, e) on nine
// signed 30-bit limbs.  Two forms.  fe_inverse_plain: 20 batches = 600 steps (590 suffice for any 256-bit input), every lane the
// same instruction sequence whatever its operand -- ~17 k instructions against ~114 k for x^(p-2) (253 squarings + 125 products
// of ~300 instructions each).  fe_inverse_plain_var: the steps of a batch in variable time (zero runs shifted at once, up to six
// bits cancelled per odd step: ~8 iterations instead of 30) until g = 0, ~8 k instructions; the lanes of a wavefront differ by a
// few iterations.  ark-ff's own inverse is variable-time too.  The latency of an inversion is what a witness-builder level that
// holds one Inverse costs (witness_builder.rs:66-69: `operand.inverse().unwrap()`).  The inverse of a field element is unique, so
// any correct algorithm is bit-identical to ark-ff's.
#pragma once
#include "fe29.hpp"

namespace pk {

typedef int32_t i32;
typedef int64_t i64;

struct s30 {  // value = sum v[i] 2^(30 i), limbs signed, |v[i]| < 2^30 except transiently the top one
    i32 v[9];
};

namespace inv30 {
constexpr i32 M30 = (i32)((1u << 30) - 1);
// limb i of p on 30-bit limbs
PK_HD constexpr i32 p_limb(int i) {
    const int bit = 30 * i, w = bit >> 5, s = bit & 31;
    u64 lo = kPlimb(w);
    if (w + 1 < 8) lo |= (u64)kPlimb(w + 1) << 32;
    return (i32)((lo >> s) & (u64)M30);
}
// p^-1 mod 2^30 (Newton: x <- x (2 - p x), precision doubles)
PK_HD constexpr u32 p_inv30() {
    u32 p0 = kPlimb(0), x = 1;
    for (int i = 0; i < 6; i++) x *= 2u - p0 * x;
    return x & (u32)M30;
}

struct trans {  // the 2x2 matrix of one batch, scaled by 2^30
    i32 u, v, q, r;
};

// 30 division steps on the low words.  zeta = -(delta + 1/2).
PK_HD i32 divsteps30(i32 zeta, u32 f0, u32 g0, trans& t) {
    u32 u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll
    for (int i = 0; i < 30; i++) {
        u32 c1 = (u32)(zeta >> 31);  // all ones when zeta < 0
        const u32 c2 = 0u - (g & 1u);   // all ones when g is odd
        const u32 x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;  // (f, u, v) negated when zeta < 0
        g += x & c2;
        q += y & c2;
        r += z & c2;
        c1 &= c2;  // swap (and negate) when zeta < 0 and g is odd
        zeta = (zeta ^ (i32)c1) - 1;
        f += g & c1;
        u += q & c1;
        v += r & c1;
        g >>= 1;
        u <<= 1;
        v <<= 1;
    }
    t.u = (i32)u;
    t.v = (i32)v;
    t.q = (i32)q;
    t.r = (i32)r;
    return zeta;
}

// The same 30 steps in variable time (classic form, eta = -delta): a run of zero bits of g is one shift, and an odd g loses up to six
// bits at once to a multiple w f of f, w = -g f^-1 mod 2^k with f^-1 = f (2 - f^2) mod 64 (one Newton step from f f = 1 mod 8),
// k <= eta + 1 so that no swap falls inside the run.  ~8 iterations instead of 30; lanes of a wavefront differ by a few.
PK_HD i32 divsteps30_var(i32 eta, u32 f0, u32 g0, trans& t) {
    u32 u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
    u32 finv = f * (2u - f * f);  // f^-1 mod 64; f changes only in a swap
    int i = 30;
    for (;;) {
        const int zeros = __builtin_ctz(g | (0xffffffffu << i));  // at most i
        g >>= zeros;
        u <<= zeros;
        v <<= zeros;
        eta -= zeros;
        i -= zeros;
        if (i == 0) break;
        if (eta < 0) {  // g is odd: swap when delta > 0
            eta = -eta;
            u32 x = f;
            f = g;
            g = 0u - x;
            x = u;
            u = q;
            q = 0u - x;
            x = v;
            v = r;
            r = 0u - x;
            finv = f * (2u - f * f);
        }
        int limit = eta + 1 > i ? i : eta + 1;
        if (limit > 6) limit = 6;
        const u32 w = (0u - g * finv) & ((1u << limit) - 1u);
        g += f * w;
        q += u * w;
        r += v * w;
    }
    t.u = (i32)u;
    t.v = (i32)v;
    t.q = (i32)q;
    t.r = (i32)r;
    return eta;
}

// (d, e) <- t (d, e) / 2^30 mod p; d, e stay in (-2p, p)
PK_HD void update_de(s30& d, s30& e, const trans& t) {
    const i32 u = t.u, v = t.v, q = t.q, r = t.r;
    const i32 sd = d.v[8] >> 31, se = e.v[8] >> 31;
    i32 md = (u & sd) + (v & se), me = (q & sd) + (r & se);  // start from p added once for each negative input
    i32 di = d.v[0], ei = e.v[0];
    i64 cd = (i64)u * di + (i64)v * ei, ce = (i64)q * di + (i64)r * ei;
    // the multiple of p that clears the low 30 bits
    md -= (i32)((p_inv30() * (u32)cd + (u32)md) & (u32)M30);
    me -= (i32)((p_inv30() * (u32)ce + (u32)me) & (u32)M30);
    cd += (i64)p_limb(0) * md;
    ce += (i64)p_limb(0) * me;
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        di = d.v[i];
        ei = e.v[i];
        cd += (i64)u * di + (i64)v * ei;
        ce += (i64)q * di + (i64)r * ei;
        cd += (i64)p_limb(i) * md;
        ce += (i64)p_limb(i) * me;
        d.v[i - 1] = (i32)cd & M30;
        cd >>= 30;
        e.v[i - 1] = (i32)ce & M30;
        ce >>= 30;
    }
    d.v[8] = (i32)cd;
    e.v[8] = (i32)ce;
}

// (f, g) <- t (f, g) / 2^30 (exact)
PK_HD void update_fg(s30& f, s30& g, const trans& t) {
    const i32 u = t.u, v = t.v, q = t.q, r = t.r;
    i32 fi = f.v[0], gi = g.v[0];
    i64 cf = (i64)u * fi + (i64)v * gi, cg = (i64)q * fi + (i64)r * gi;
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        fi = f.v[i];
        gi = g.v[i];
        cf += (i64)u * fi + (i64)v * gi;
        cg += (i64)q * fi + (i64)r * gi;
        f.v[i - 1] = (i32)cf & M30;
        cf >>= 30;
        g.v[i - 1] = (i32)cg & M30;
        cg >>= 30;
    }
    f.v[8] = (i32)cf;
    g.v[8] = (i32)cg;
}

// d in (-2p, p), negated when sign < 0, brought to [0, p)
PK_HD void normalize(s30& r, i32 sign) {
    i32 add = r.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] += p_limb(i) & add;
    const i32 neg = sign >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = (r.v[i] ^ neg) - neg;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.v[i + 1] += r.v[i] >> 30;
        r.v[i] &= M30;
    }
    add = r.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] += p_limb(i) & add;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.v[i + 1] += r.v[i] >> 30;
        r.v[i] &= M30;
    }
}

PK_HD s30 to_s30(const fe& a) {
    s30 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 30 * i, w = bit >> 5, s = bit & 31;
        u64 lo = a.v[w];
        if (w + 1 < 8) lo |= (u64)a.v[w + 1] << 32;
        r.v[i] = (i32)((lo >> s) & (u64)M30);
    }
    return r;
}
PK_HD fe from_s30(const s30& a) {  // limbs in [0, 2^30), value < 2^256
    fe r;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const int bit = 32 * w, i = bit / 30, s = bit % 30;  // word w starts inside limb i at bit s
        u64 x = (u64)(u32)a.v[i] >> s;
        x |= (u64)(u32)a.v[i + 1] << (30 - s);
        if (60 - s < 32 && i + 2 < 9) x |= (u64)(u32)a.v[i + 2] << (60 - s);
        r.v[w] = (u32)x;
    }
    return r;
}
}  // namespace inv30

// a^-1 mod p for 0 <= a < p as plain integers (0 -> 0)
PK_HD fe fe_inverse_plain(const fe& a) {
    using namespace inv30;
    s30 d, e, f, g = to_s30(a);
#pragma unroll
    for (int i = 0; i < 9; i++) {
        d.v[i] = 0;
        e.v[i] = i == 0;
        f.v[i] = p_limb(i);
    }
    i32 zeta = -1;
#pragma unroll 1
    for (int it = 0; it < 20; it++) {
        trans t;
        zeta = divsteps30(zeta, (u32)f.v[0], (u32)g.v[0], t);
        update_de(d, e, t);
        update_fg(f, g, t);
    }
    normalize(d, f.v[8]);  // g = 0, f = +-1: the inverse is sign(f) d
    return from_s30(d);
}

// the same inverse with the variable-time steps: batches until g = 0 (18-19 for this p; 25 bound the classic form at 256 bits)
PK_HD fe fe_inverse_plain_var(const fe& a) {
    using namespace inv30;
    s30 d, e, f, g = to_s30(a);
#pragma unroll
    for (int i = 0; i < 9; i++) {
        d.v[i] = 0;
        e.v[i] = i == 0;
        f.v[i] = p_limb(i);
    }
    i32 eta = -1;
#pragma unroll 1
    for (int it = 0; it < 30; it++) {
        i32 nz = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) nz |= g.v[i];
        if (nz == 0) break;
        trans t;
        eta = divsteps30_var(eta, (u32)f.v[0], (u32)g.v[0], t);
        update_de(d, e, t);
        update_fg(f, g, t);
    }
    normalize(d, f.v[8]);
    return from_s30(d);
}

// Montgomery in, Montgomery out: (x R)^-1 = x^-1 R^-1, times R^3 through one Montgomery product (a b R^-1) = x^-1 R
PK_HD fe fe_inverse_mont(const fe& xr) {
    fe r3;  // R^3 mod p
    r3.v[0] = 0xb4bf0040u; r3.v[1] = 0x5e94d8e1u; r3.v[2] = 0x1cfbb6b8u; r3.v[3] = 0x2a489cbeu;
    r3.v[4] = 0xa19fcfedu; r3.v[5] = 0x893cc664u; r3.v[6] = 0x7fcc657cu; r3.v[7] = 0x0cf8594bu;
    return fe_mulx(fe_inverse_plain_var(xr), r3);
}

}  // namespace pk
