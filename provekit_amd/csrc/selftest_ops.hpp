// selftest_ops.hpp -- the table of arithmetic self-test operations: each op runs one piece of the library's __host__ __device__
// arithmetic (fe29.hpp, skyscraper29.hpp, skyscraper29s.hpp, feinv.hpp, reduce.hpp) exactly as the kernels compile it.  Two users:
// pk_selftest_arith (selftest.hip, in the product: the CPU suite checks the host build against the oracle without a GPU) and
// pk_probe_arith_device (tools/probes, not shipped: the same ops run by a kernel, for a device-vs-host codegen diff).
#pragma once
#include "ctx.hpp"
#include "reduce.hpp"
#include "feinv.hpp"
#include "skyscraper29s.hpp"
#include "ntt_regs.hpp"

namespace pk {

inline fe load_host(const uint64_t* p) {
    fe r;
    memcpy(r.v, p, 32);
    return r;
}
inline void store_host(uint64_t* p, const fe& x) { memcpy(p, x.v, 32); }

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
        case 24: {  // shoup261_29: x * y mod p for y < p, x any value below 8p (here < 2^256 ~ 5.3p); result almost reduced then exact
            fe29 w = cond_sub_p29(unpack_reduce29(y));
            fe29 t = shoup261_29(unpack29<0>(x), w, shoup_quotient29(w));
            reduce_almost29(t);
            r = pack29(cond_sub_p29(t));
            break;
        }
        case 25: {  // the same with lazy limbs on the multiplicand: (x + y + 2p) * y, limbs < 2^30.6 as the butterflies make them
            fe29 w = cond_sub_p29(unpack_reduce29(y));
            fe29 a = add29(unpack_reduce29(x), w);
            fe29 t = shoup261_29(a, w, shoup_quotient29(w));
            reduce_almost29(t);
            r = pack29(cond_sub_p29(t));
            break;
        }
        case 26: {  // the in-register constants of the NTT (ntt_regs.hpp): w_8^e for e = x mod 4 (1..3), word 7 bit 31 set iff the stored
                    // quotient differs from floor(w 2^261 / p) recomputed by long division
            const int e = (int)(x.v[0] & 3u);
            fe29 w, wq;
            for (int k = 0; k < 9; k++) {
                w.v[k] = e == 1 ? w8_29(1, k) : (e == 2 ? w8_29(2, k) : w8_29(3, k));
                wq.v[k] = e == 1 ? w8q_29(1, k) : (e == 2 ? w8q_29(2, k) : w8q_29(3, k));
            }
            const fe29 q = shoup_quotient29(w);
            bool same = true;
            for (int k = 0; k < 9; k++) same = same && q.v[k] == wq.v[k];
            r = pack29(w);
            if (!same) r.v[7] |= 0x80000000u;
            break;
        }
        case 21: r = fe_inverse_mont(x); break;   // feinv.hpp: Montgomery in, Montgomery out (0 -> 0)
        case 22: r = fe_inverse_plain(x); break;  // plain integers mod p, constant sequence
        case 23: r = fe_inverse_plain_var(x); break;  // the same, variable-time steps
        default: break;
    }
    return r;
}

}  // namespace pk
