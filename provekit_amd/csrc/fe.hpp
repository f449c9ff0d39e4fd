// fe.hpp -- BN254-Fr arithmetic for gfx950 device code.
//
// The 256-bit element container, exact add/sub and conditional subtraction (ark-ff Fp256 + and -).
// The multipliers (rows A1/A2: ark-ff mul, block_multiplier::scalar_{mul,sqr}) are in fe29.hpp.
//
// In-memory format is the reference's: 4 x u64 little-endian limbs, Montgomery
// form, 32 B per element (== 8 x u32 little-endian).  Values held in registers
// are always fully reduced (< p) unless a function says otherwise.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pk {

typedef uint32_t u32;
typedef uint64_t u64;

struct fe {
    u32 v[8];
};

// p, little-endian u32 limbs (skyscraper/block-multiplier/src/constants.rs:3-8)
#define PK_P0 0xf0000001u
#define PK_P1 0x43e1f593u
#define PK_P2 0x79b97091u
#define PK_P3 0x2833e848u
#define PK_P4 0x8181585du
#define PK_P5 0xb85045b6u
#define PK_P6 0xe131a029u
#define PK_P7 0x30644e72u
#define PK_NP0 0xefffffffu  // -p^-1 mod 2^32 (low half of U64_NP0, constants.rs:1)

__host__ __device__ __forceinline__ constexpr u32 kPlimb(int i) {
    constexpr u32 P[8] = {PK_P0, PK_P1, PK_P2, PK_P3, PK_P4, PK_P5, PK_P6, PK_P7};
    return P[i];
}

template <int K>
__host__ __device__ __forceinline__ constexpr u32 p_mul_limb(int i) {
    // limb i of K*p, K in 1..4 (4p < 2^256)
    constexpr u32 P[8] = {PK_P0, PK_P1, PK_P2, PK_P3, PK_P4, PK_P5, PK_P6, PK_P7};
    u64 c = 0;
    u32 out = 0;
    for (int j = 0; j <= i; j++) {
        c += (u64)P[j] * K;
        out = (u32)c;
        c >>= 32;
    }
    return out;
}

__host__ __device__ __forceinline__ fe fe_zero() {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}
// R mod p = Montgomery one (constants.rs:17-22)
__host__ __device__ __forceinline__ fe fe_one() {
    fe r;
    r.v[0] = 0x4ffffffbu; r.v[1] = 0xac96341cu; r.v[2] = 0x9f60cd29u; r.v[3] = 0x36fc7695u;
    r.v[4] = 0x7879462eu; r.v[5] = 0x666ea36fu; r.v[6] = 0x9a07df2fu; r.v[7] = 0x0e0a77c1u;
    return r;
}
// R^2 mod p (constants.rs:25-30)
__host__ __device__ __forceinline__ fe fe_r2() {
    fe r;
    r.v[0] = 0xae216da7u; r.v[1] = 0x1bb8e645u; r.v[2] = 0xe35c59e3u; r.v[3] = 0x53fe3ab1u;
    r.v[4] = 0x53bb8085u; r.v[5] = 0x8c49833du; r.v[6] = 0x7f4e44a5u; r.v[7] = 0x0216d0b1u;
    return r;
}

__host__ __device__ __forceinline__ fe fe_load(const void* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    fe r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__host__ __device__ __forceinline__ void fe_store(void* p, const fe& x) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    q[1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}

// r = a - K*p if a >= K*p else a      (a any 256-bit value)
template <int K>
__host__ __device__ __forceinline__ fe cond_sub_kp(const fe& a) {
    fe d;
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d.v[i] = __builtin_subc(a.v[i], p_mul_limb<K>(i), borrow, &borrow);  // one v_subb_co_u32 per word
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = borrow ? a.v[i] : d.v[i];
    return r;
}
// any 256-bit value -> [0, p)   (2^256 < 6p)
__host__ __device__ __forceinline__ fe fe_reduce_any(const fe& a) {
    return cond_sub_kp<1>(cond_sub_kp<2>(cond_sub_kp<4>(a)));
}

__host__ __device__ __forceinline__ fe fe_add(const fe& a, const fe& b) {  // a,b < p -> < p
    // carry chains spelled with the add-/subtract-with-carry builtins: one v_addc_co_u32 / v_subb_co_u32 per word.  (The same chains
    // written in 64-bit arithmetic compile to 64-bit adds plus a move per word: 122 instructions for this function instead of ~30.)
    fe s;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s.v[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
    return cond_sub_kp<1>(s);  // a+b < 2p < 2^255: no carry out
}
__host__ __device__ __forceinline__ fe fe_sub(const fe& a, const fe& b) {  // a,b < p -> < p
    fe d;
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d.v[i] = __builtin_subc(a.v[i], b.v[i], borrow, &borrow);
    const u32 mask = 0u - borrow;
    u32 c = 0;
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = __builtin_addc(d.v[i], kPlimb(i) & mask, c, &c);
    return r;
}
__host__ __device__ __forceinline__ fe fe_dbl(const fe& a) { return fe_add(a, a); }
__host__ __device__ __forceinline__ fe fe_neg(const fe& a) { return fe_sub(fe_zero(), a); }

__host__ __device__ __forceinline__ bool fe_lt(const fe& a, const fe& b) {  // a < b as 256-bit integers
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) (void)__builtin_subc(a.v[i], b.v[i], borrow, &borrow);
    return borrow != 0;
}
__host__ __device__ __forceinline__ bool fe_eq(const fe& a, const fe& b) {
    u32 d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d |= a.v[i] ^ b.v[i];
    return d == 0;
}

// Multiplication lives in fe29.hpp (9 x 29-bit carry-free limbs); this header keeps the 256-bit container,
// loads/stores and the carry-chain add/sub used where a value must be an exact canonical 8 x u32 integer.

}  // namespace pk
