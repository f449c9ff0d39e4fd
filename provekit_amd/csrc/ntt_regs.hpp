// ntt_regs.hpp -- the register-resident radix-8 butterfly network of the NTT pass (ntt.hip) and its constant multipliers.
// __host__ __device__, so the CPU suite runs the exact device source against the definition of the DFT (tests/test_fe29_host.py).
//
// Every multiplication of the network is by a value known before the launch (a power of the 8th root of unity, an inter-round or
// inter-pass twiddle), so all of them take the Shoup form (fe29.hpp shoup261_29: 143 multiply-adds and no sequential quotient chain,
// against the Montgomery product's 171).  A Shoup product a * c keeps `a`'s domain: Montgomery images stay Montgomery images.
//
// Lazy bounds (test_ntt_butterfly_network_on_the_host drives the extremes on the host build of this source):
//   inputs of a round        normalised limbs, value < 1.2p      (loads: canonical or almost reduced; exchange: products or red29)
//   sums / differences       never reduced inside a round: see dft_regs (outputs < 6.4p, limbs < 2^31.4; the sum of sums < 4.8p)
//   a - b + 2p (sub2p29)     b normalised and < 1.9p: 2p is lent limb by limb (the top limb less one), so no limb goes negative
//   red29(v)                 limbs < 2^31, v < 5p -> normalised, < 1.001p   (the quotient estimate falls short by < 1.8e-4 v/p; a
//                            quotient above 4 would overflow the signed limb differences: red29w takes up to 8p)
//   shoup261_29(v)           limbs < 2^31.4, v < 10.5p -> normalised, < 1.07p with an exact quotient (the in-register constants), < 1.2p
//                            with a table quotient up to two short
// so every value that re-enters a butterfly or is stored is below 1.2p (the almost-reduced stores promise < 1.6p).
#pragma once
#include "fe29.hpp"

namespace pk {

PK_HD fe29 red29(fe29 x) {  // lazy non-negative limbs below 2^31, value < 5p -> normalized, < 1.001p
    reduce_almost29(x);
    return x;
}
// a - b + 2p with borrow-proof limbs (b normalized, value < 1.9p): result limbs < 2^30.6, value < a + 2p
PK_HD constexpr u32 c2p29(int k) {
    return k == 0 ? kp29(2, 0) + (1u << 29) : (k < 8 ? kp29(2, k) + (1u << 29) - 1u : kp29(2, 8) - 1u);
}
PK_HD fe29 sub2p29(const fe29& a, const fe29& b) {
    fe29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) r.v[k] = a.v[k] + c2p29(k) - b.v[k];
    return r;
}

// w_8^e (e = 1, 2, 3) for w_8 = g^((p-1)/8) = TWO_ADIC_ROOT_OF_UNITY^(2^25) (ark-bn254 Fr; the root every twiddle table of ntt.hip is
// built from), canonical, as 29-bit limbs, and floor(w_8^e * 2^261 / p).  The same three values for every transform size, so they are
// compile-time constants: no table, no loads, the limbs sit in scalar registers.  tests/test_fe29_host.py recomputes all 54 limbs.
PK_HD constexpr u32 w8_29(int e, int k) {
    constexpr u32 W[3][9] = {
        {0x01bd5e80u, 0x046d6a56u, 0x05c282a5u, 0x04e6cdf0u, 0x1ef36526u, 0x0f17cb57u, 0x1c8bb26eu, 0x1c391829u, 0x002b337du},
        {0x0f703636u, 0x18902384u, 0x1cdafb08u, 0x1449edfau, 0x041045ceu, 0x170c9fecu, 0x00a4122du, 0x0e5c2634u, 0x0030644eu},
        {0x0846a566u, 0x0680df47u, 0x14944eaeu, 0x05b33670u, 0x13a6f07cu, 0x1d642844u, 0x0732f455u, 0x0c29372bu, 0x001d5937u}};
    return W[e - 1][k];
}
PK_HD constexpr u32 w8q_29(int e, int k) {
    constexpr u32 Q[3][9] = {
        {0x0a507837u, 0x0afcc55au, 0x00e98eecu, 0x128b7ff6u, 0x0e1eda7cu, 0x1c62ed03u, 0x053a5c1du, 0x0a760276u, 0x1c914bddu},
        {0x04620ebcu, 0x0276cbb1u, 0x1cc4f176u, 0x0850dde1u, 0x15092ec2u, 0x0bfdae8fu, 0x1e247f6bu, 0x1fffffffu, 0x1fffffffu},
        {0x1453d353u, 0x18ed2b8du, 0x0e86cb7eu, 0x1ea0d7fbu, 0x1cc9cb48u, 0x0ba33448u, 0x0a964880u, 0x06c188b3u, 0x13684156u}};
    return Q[e - 1][k];
}
template <int E>
PK_HD fe29 mul_w8(const fe29& a) {  // a * w_8^E, a: limbs < 2^31.4, value < 10.5p
    fe29 w, wq;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        w.v[k] = w8_29(E, k);
        wq.v[k] = w8q_29(E, k);
    }
    return shoup261_29(a, w, wq);
}

// a multiplier of the twiddle tables: the value (canonical) and its Shoup quotient, 18 words = 72 bytes per entry
struct tw29s {
    fe29 w, wq;
};
constexpr int TW29S_WORDS = 18;
PK_HD tw29s tw29s_load(const u32* __restrict__ T, size_t idx) {
    const u32* q = T + (size_t)TW29S_WORDS * idx;
    tw29s t;
#pragma unroll
    for (int l = 0; l < 9; l++) t.w.v[l] = q[l];
#pragma unroll
    for (int l = 0; l < 9; l++) t.wq.v[l] = q[9 + l];
    return t;
}
PK_HD fe29 mul_tw(const fe29& a, const tw29s& t) { return shoup261_29(a, t.w, t.wq); }

// a - b + 4p for a LAZY b (limbs < 2^30, value < 3.9p: a sum of two inputs): 4p is lent with 2^30 per limb.  Result limbs < 2^31.4.
PK_HD constexpr u32 c4p29(int k) {
    return k == 0 ? kp29(4, 0) + (1u << 30) : (k < 8 ? kp29(4, k) + (1u << 30) - 2u : kp29(4, 8) - 2u);
}
PK_HD fe29 sub4p29(const fe29& a, const fe29& b) {
    fe29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) r.v[k] = a.v[k] + c4p29(k) - b.v[k];
    return r;
}
// red29 for limbs up to 2^31.4 and values up to 8p (the signed sweep of normalize29 would read such limbs as negative, and a quotient
// above 4 times a limb of p overflows its signed difference): one unsigned sweep, then the quotient in two halves
PK_HD fe29 red29w(fe29 x) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
        x.v[k + 1] += x.v[k] >> 29;
        x.v[k] &= M29;
    }
    sub_qp29(x, quot_estimate29(x.v[8]) >> 1);
    normalize29(x);
    reduce_almost29(x);
    return x;
}

// radix-2^D DIF over the top D bits of the register index of a lane's 2^LE values (D = 1 | 2); the low LE-D bits are independent
// batches.  No reduction inside the network: a radix-4 butterfly is four additions, four subtractions and ONE product (by w_4),
//   s0 = x0 + x2   d0 = x0 - x2 + 2p   s1 = x1 + x3   d1 = (x1 - x3 + 2p) w_4
//   X0 = s0 + s1   X2 = s0 - s1 + 4p   X1 = d0 + d1   X3 = d0 - d1 + 2p          (register order X0, X2, X1, X3: bit-reversed)
// with inputs below 1.2p: X0 < 4.8p (limbs < 2^31: red29), the others < 6.4p, limbs < 2^31.4 -- what shoup261_29 (a < 10.5p: quotient
// at most one short; columns 9 * 2^31.4 * 2^29 + 9 * 2^58 < 2^64) and red29w accept.  (The subtractions themselves hold up to 1.9p.)
template <int LE, int D>
PK_HD void dft_regs(fe29 (&x)[1 << LE]) {
    static_assert(D >= 1 && D <= 2 && D <= LE && LE <= 3, "radix 2 or 4");
    constexpr int E = LE - D;
#pragma unroll
    for (int b = 0; b < (1 << E); b++) {
        if constexpr (D == 1) {
            const fe29 a = x[b], c = x[b | (1 << E)];
            x[b] = add29(a, c);
            x[b | (1 << E)] = sub2p29(a, c);
        } else {
            const fe29 x0 = x[b], x1 = x[b | (1 << E)], x2 = x[b | (2 << E)], x3 = x[b | (3 << E)];
            const fe29 s0 = add29(x0, x2), d0 = sub2p29(x0, x2), s1 = add29(x1, x3);
            const fe29 d1 = mul_w8<2>(sub2p29(x1, x3));
            x[b] = add29(s0, s1);
            x[b | (1 << E)] = sub4p29(s0, s1);
            x[b | (2 << E)] = add29(d0, d1);
            x[b | (3 << E)] = sub2p29(d0, d1);
        }
    }
}

}  // namespace pk
