/*
 * pk_oracle.c -- CPU restatement of the ProveKit WHIR hot path.
 * TEST INFRASTRUCTURE ONLY (see pk_oracle.h header comment for the rules and
 * the pinning status). Plain C11 + unsigned __int128.  OpenMP marks the loops the
 * reference parallelises with rayon (par_iter / rayon::join: sumcheck.rs:53,83,163,185;
 * ark MerkleTree's `parallel` feature) so the cpu_baseline leg is a fair all-cores run;
 * the sparse mat-vecs stay serial as in sparse_matrix.rs:148,167 ("OPT: Paralelize").
 */
#include "pk_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t u64;
typedef unsigned __int128 u128;

/* skyscraper/block-multiplier/src/constants.rs:1-40 */
static const u64 P[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const u64 P2[4] = {0x87c3eb27e0000002ULL, 0x5067d090f372e122ULL, 0x70a08b6d0302b0baULL, 0x60c89ce5c2634053ULL};
static const u64 NP0 = 0xc2e1f593efffffffULL; /* -p^-1 mod 2^64 (U64_NP0 == U64_MU0) */
static const u64 R1[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}; /* R mod p */
static const u64 R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}; /* R^2 mod p */
/* constants.rs:77-96 */
static const u64 I1[4] = {0x2d3e8053e396ee4dULL, 0xca478dbeab3c92cdULL, 0xb2d8f06f77f52a93ULL, 0x24d6ba07f7aa8f04ULL};
static const u64 I2[4] = {0x18ee753c76f9dc6fULL, 0x54ad7e14a329e70fULL, 0x2b16366f4f7684dfULL, 0x133100d71fdf3579ULL};
static const u64 I3[4] = {0x9bacb016127cbe4eULL, 0x0b2051fa31944124ULL, 0xb064eea46091c76cULL, 0x2b062aaa49f80c7dULL};

/* skyscraper/core/src/constants.rs:30-49 */
static const u64 RC[18][4] = {
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

/* 2^28-th root of unity 5^((p-1)/2^28), canonical (ark-bn254 Fr TWO_ADIC_ROOT_OF_UNITY; SURVEY 8 conventions) */
static const u64 ROOT28_CANON[4] = {0x9bd61b6e725b19f0ULL, 0x402d111e41112ed4ULL, 0x00e0a7eb8ef62abcULL, 0x2a3c09f0a58a7e85ULL};

/* ------------------------------------------------------------------ */
/* 256-bit helpers                                                     */
/* ------------------------------------------------------------------ */
static inline int geq(const u64 a[4], const u64 b[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static inline int lt(const u64 a[4], const u64 b[4]) { return !geq(a, b); }

static inline u64 add4(const u64 a[4], const u64 b[4], u64 r[4]) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a[i] + b[i];
        r[i] = (u64)c;
        c >>= 64;
    }
    return (u64)c;
}
static inline u64 sub4(const u64 a[4], const u64 b[4], u64 r[4]) {
    u64 borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
    return borrow;
}
static inline void mod_p_any(const u64 x[4], u64 r[4]) { /* any 256-bit value -> [0,p) */
    u64 t[4];
    memcpy(t, x, 32);
    while (geq(t, P)) sub4(t, P, t);
    memcpy(r, t, 32);
}

/* ------------------------------------------------------------------ */
/* A2: field arithmetic (ark-ff Fp256 semantics, fully reduced)        */
/* ------------------------------------------------------------------ */
void pko_fe_add(const u64 a[4], const u64 b[4], u64 out[4]) {
    u64 t[4];
    u64 c = add4(a, b, t);
    if (c || geq(t, P)) sub4(t, P, t);
    memcpy(out, t, 32);
}
void pko_fe_sub(const u64 a[4], const u64 b[4], u64 out[4]) {
    u64 t[4];
    if (sub4(a, b, t)) add4(t, P, t);
    memcpy(out, t, 32);
}
/* CIOS Montgomery product a*b*2^-256 mod p; valid for any a < 2^256, b < p */
void pko_fe_mul(const u64 a[4], const u64 b[4], u64 out[4]) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a[j] * b[i] + t[j];
            t[j] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (u64)c;
        t[5] = (u64)(c >> 64);
        u64 m = t[0] * NP0;
        c = (u128)m * P[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * P[j] + t[j];
            t[j - 1] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (u64)c;
        t[4] = t[5] + (u64)(c >> 64);
        t[5] = 0;
    }
    u64 r[4] = {t[0], t[1], t[2], t[3]};
    while (t[4] || geq(r, P)) {
        u64 bw = sub4(r, P, r);
        t[4] -= bw;
    }
    memcpy(out, r, 32);
}
void pko_fe_to_mont(const u64 canon[4], u64 out[4]) { pko_fe_mul(canon, R2, out); }
void pko_fe_from_mont(const u64 mont[4], u64 out[4]) {
    static const u64 one[4] = {1, 0, 0, 0};
    pko_fe_mul(mont, one, out);
}
void pko_fe_to_mont_many(const u64 *canon, u64 *out, size_t n) {
#pragma omp parallel for schedule(static) if (n >= 4096)
    for (long i = 0; i < (long)n; i++) pko_fe_to_mont(canon + 4 * i, out + 4 * i);
}
void pko_fe_from_mont_many(const u64 *mont, u64 *out, size_t n) {
#pragma omp parallel for schedule(static) if (n >= 4096)
    for (long i = 0; i < (long)n; i++) pko_fe_from_mont(mont + 4 * i, out + 4 * i);
}
void pko_fe_pow(const u64 base[4], u64 e, u64 out[4]) {
    u64 acc[4], b[4];
    memcpy(acc, R1, 32);
    memcpy(b, base, 32);
    while (e) {
        if (e & 1) pko_fe_mul(acc, b, acc);
        pko_fe_mul(b, b, b);
        e >>= 1;
    }
    memcpy(out, acc, 32);
}
void pko_root_of_unity(unsigned log_n, u64 out[4]) {
    u64 w[4];
    pko_fe_to_mont(ROOT28_CANON, w);
    for (unsigned i = log_n; i < 28; i++) pko_fe_mul(w, w, w);
    memcpy(out, w, 32);
}

/* ------------------------------------------------------------------ */
/* A1: scalar_mul / scalar_sqr, literal restatement                    */
/* skyscraper/block-multiplier/src/scalar.rs:12-132, utils.rs:52-62,177-181 */
/* ------------------------------------------------------------------ */
static inline void cma(u64 a, u64 b, u64 add, u64 carry, u64 *lo, u64 *hi) { /* utils.rs carrying_mul_add */
    u128 c = (u128)a * b + carry + add;
    *lo = (u64)c;
    *hi = (u64)(c >> 64);
}
static void addv5(u64 a[5], const u64 b[5]) { /* utils.rs:52-62 */
    u64 carry = 0;
    for (int i = 0; i < 5; i++) {
        u64 s1 = a[i] + b[i];
        u64 o1 = s1 < a[i];
        u64 s2 = s1 + carry;
        u64 o2 = s2 < s1;
        a[i] = s2;
        carry = o1 + o2;
    }
}
void pko_scalar_mul(const u64 a[4], const u64 b[4], u64 out[4]) {
    u64 t[8] = {0};
    for (int i = 0; i < 4; i++) {
        u64 carry = 0;
        for (int j = 0; j < 4; j++) cma(a[i], b[j], t[i + j], carry, &t[i + j], &carry);
        t[i + 4] = carry;
    }
    u64 s[5] = {t[3], t[4], t[5], t[6], t[7]};
    const u64 *inv[3] = {I3, I2, I1};
    for (int k = 0; k < 3; k++) { /* s_r1 = t0*I3, s_r2 = t1*I2, s_r3 = t2*I1 */
        u64 sr[5] = {0};
        cma(t[k], inv[k][0], 0, 0, &sr[0], &sr[1]);
        cma(t[k], inv[k][1], sr[1], 0, &sr[1], &sr[2]);
        cma(t[k], inv[k][2], sr[2], 0, &sr[2], &sr[3]);
        cma(t[k], inv[k][3], sr[3], 0, &sr[3], &sr[4]);
        addv5(s, sr);
    }
    u64 m = NP0 * s[0];
    u64 mp[5] = {0};
    cma(m, P[0], 0, 0, &mp[0], &mp[1]);
    cma(m, P[1], mp[1], 0, &mp[1], &mp[2]);
    cma(m, P[2], mp[2], 0, &mp[2], &mp[3]);
    cma(m, P[3], mp[3], 0, &mp[3], &mp[4]);
    addv5(s, mp);
    u64 r[4] = {s[1], s[2], s[3], s[4]};
    if (r[3] >> 63) sub4(r, P2, r); /* reduce_ct */
    memcpy(out, r, 32);
}
void pko_scalar_sqr(const u64 a[4], u64 out[4]) { pko_scalar_mul(a, a, out); }

/* ------------------------------------------------------------------ */
/* H1: Skyscraper                                                      */
/* ------------------------------------------------------------------ */
static inline uint8_t rotl8(uint8_t v, int k) { return (uint8_t)((v << k) | (v >> (8 - k))); }
uint8_t pko_sbox(uint8_t v) { /* reference.rs:96-98 */
    return rotl8((uint8_t)(v ^ (rotl8((uint8_t)~v, 1) & rotl8(v, 2) & rotl8(v, 3))), 1);
}
/* canonical x in [0,p) -> canonical; reference.rs:80-94 */
void pko_bar(const u64 x[4], u64 out[4]) {
    uint8_t bytes[32], sw[32];
    memcpy(bytes, x, 32); /* little-endian host */
    memcpy(sw, bytes + 16, 16);
    memcpy(sw + 16, bytes, 16);
    for (int i = 0; i < 32; i++) sw[i] = pko_sbox(sw[i]);
    u64 y[4];
    memcpy(y, sw, 32);
    mod_p_any(y, out);
}
static inline void sq_sigma(const u64 x[4], u64 out[4]) { pko_fe_mul(x, x, out); } /* x^2 * 2^-256 */

/* one Feistel round: (l, r) <- (r + F(l) + rc, l) on canonical values */
static inline void feistel(u64 l[4], u64 r[4], int use_bar, const u64 rc[4]) {
    u64 f[4], nl[4];
    if (use_bar)
        pko_bar(l, f);
    else
        sq_sigma(l, f);
    pko_fe_add(r, f, nl);
    pko_fe_add(nl, rc, nl);
    memcpy(r, l, 32);
    memcpy(l, nl, 32);
}
void pko_permute(const u64 l_in[4], const u64 r_in[4], u64 ol[4], u64 orr[4]) { /* reference.rs:49-78 */
    u64 l[4], r[4];
    mod_p_any(l_in, l);
    mod_p_any(r_in, r);
    for (int i = 0; i < 18; i++) {
        int use_bar = (i == 6 || i == 7 || i == 10 || i == 11);
        feistel(l, r, use_bar, RC[i]);
    }
    memcpy(ol, l, 32);
    memcpy(orr, r, 32);
}
void pko_compress(const u64 l[4], const u64 r[4], u64 out[4]) { /* reference.rs:41-46 */
    u64 t[4], pl[4], pr[4];
    mod_p_any(l, t);
    pko_permute(l, r, pl, pr);
    pko_fe_add(pl, t, out);
}
void pko_compress_v1(const u64 l_in[4], const u64 r_in[4], u64 out[4]) { /* v1.rs:19-32 */
    static const int bar_round[10] = {0, 0, 1, 1, 0, 0, 1, 1, 0, 0};
    static const int rc_idx[10] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 0}; /* RC[0] == 0: first/last add nothing */
    u64 l[4], r[4], t[4];
    mod_p_any(l_in, l);
    mod_p_any(r_in, r);
    memcpy(t, l, 32);
    for (int i = 0; i < 10; i++) feistel(l, r, bar_round[i], RC[rc_idx[i]]);
    pko_fe_add(l, t, out);
}
static int compress_many_impl(const uint8_t *m, size_t mlen, uint8_t *h, size_t hlen, int version) {
    if (mlen % 64 || hlen % 32 || mlen != hlen * 2) return -1; /* generic.rs:18-25 */
    size_t n = hlen / 32;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        u64 l[4], r[4], o[4];
        memcpy(l, m + 64 * i, 32);
        memcpy(r, m + 64 * i + 32, 32);
        if (version == 1)
            pko_compress_v1(l, r, o);
        else
            pko_compress(l, r, o);
        memcpy(h + 32 * i, o, 32);
    }
    return 0;
}
int pko_compress_many(const uint8_t *m, size_t mlen, uint8_t *h, size_t hlen) {
    return compress_many_impl(m, mlen, h, hlen, 2);
}
int pko_compress_many_v1(const uint8_t *m, size_t mlen, uint8_t *h, size_t hlen) {
    return compress_many_impl(m, mlen, h, hlen, 1);
}

/* ------------------------------------------------------------------ */
/* H2 / M1 / M2: provekit/common/src/skyscraper/whir.rs:20-74          */
/* ------------------------------------------------------------------ */
static inline void compress_ver(const u64 l[4], const u64 r[4], u64 out[4], int version) {
    if (version == 1)
        pko_compress_v1(l, r, out);
    else
        pko_compress(l, r, out);
}
int pko_leaf_hash(const u64 *leaf, size_t w, u64 digest[4], int version) {
    if (w == 0) return -1; /* Error::IncorrectInputLength(0), whir.rs:47 */
    u64 h[4], x[4];
    pko_fe_from_mont(leaf, h); /* into_bigint(), whir.rs:21 */
    for (size_t j = 1; j < w; j++) {
        pko_fe_from_mont(leaf + 4 * j, x);
        compress_ver(h, x, h, version);
    }
    memcpy(digest, h, 32);
    return 0;
}
int pko_merkle_inner(u64 *nodes, size_t n, int version) {
    if (n == 0 || (n & (n - 1))) return -1;
    for (size_t lvl = n / 2; lvl >= 1; lvl /= 2) {
#pragma omp parallel for schedule(static) if (lvl >= 256)
        for (size_t i = lvl; i < 2 * lvl; i++) compress_ver(nodes + 8 * i, nodes + 8 * i + 4, nodes + 4 * i, version);
    }
    return 0;
}
int pko_merkle_commit(const u64 *leaves, size_t n, size_t w, u64 *nodes, int version) {
    if (n == 0 || (n & (n - 1)) || w == 0) return -1;
    memset(nodes, 0, 32);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) pko_leaf_hash(leaves + 4 * w * i, w, nodes + 4 * (n + i), version);
    return pko_merkle_inner(nodes, n, version);
}

/* ------------------------------------------------------------------ */
/* T1: multilinear evals <-> coefficients                               */
/* ------------------------------------------------------------------ */
void pko_to_coeffs(u64 *v, unsigned n_vars) {
    size_t n = (size_t)1 << n_vars;
    for (size_t h = 1; h < n; h <<= 1) {
#pragma omp parallel for schedule(static) if (n >= 4096)
        for (size_t blk = 0; blk < n; blk += 2 * h)
            for (size_t i = blk; i < blk + h; i++) pko_fe_sub(v + 4 * (i + h), v + 4 * i, v + 4 * (i + h));
    }
}
void pko_to_evals(u64 *v, unsigned n_vars) {
    size_t n = (size_t)1 << n_vars;
    for (size_t h = 1; h < n; h <<= 1) {
#pragma omp parallel for schedule(static) if (n >= 4096)
        for (size_t blk = 0; blk < n; blk += 2 * h)
            for (size_t i = blk; i < blk + h; i++) pko_fe_add(v + 4 * (i + h), v + 4 * i, v + 4 * (i + h));
    }
}

/* ------------------------------------------------------------------ */
/* N1/N2: NTT and interleaved RS encode                                 */
/* ------------------------------------------------------------------ */
static void ntt_with_table(u64 *a, unsigned log_n, const u64 *tw /* n/2 powers of w */) {
    size_t n = (size_t)1 << log_n;
    /* bit reversal */
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            u64 t[4];
            memcpy(t, a + 4 * i, 32);
            memcpy(a + 4 * i, a + 4 * j, 32);
            memcpy(a + 4 * j, t, 32);
        }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        size_t half = len / 2, step = n / len;
        for (size_t blk = 0; blk < n; blk += len)
            for (size_t k = 0; k < half; k++) {
                u64 u[4], v[4];
                memcpy(u, a + 4 * (blk + k), 32);
                pko_fe_mul(a + 4 * (blk + k + half), tw + 4 * (k * step), v);
                pko_fe_add(u, v, a + 4 * (blk + k));
                pko_fe_sub(u, v, a + 4 * (blk + k + half));
            }
    }
}
static u64 *make_twiddles(unsigned log_n) {
    size_t half = log_n ? ((size_t)1 << (log_n - 1)) : 1;
    u64 *tw = (u64 *)malloc(32 * half);
    u64 w[4];
    pko_root_of_unity(log_n, w);
    memcpy(tw, R1, 32);
    for (size_t i = 1; i < half; i++) pko_fe_mul(tw + 4 * (i - 1), w, tw + 4 * i);
    return tw;
}
void pko_ntt(u64 *a, unsigned log_n) {
    if (log_n == 0) return;
    u64 *tw = make_twiddles(log_n);
    ntt_with_table(a, log_n, tw);
    free(tw);
}
int pko_rs_encode(const u64 *coeffs, unsigned batch, unsigned n_vars, unsigned log_inv_rate, unsigned fold,
                  u64 *leaves) {
    if (n_vars < fold || n_vars + log_inv_rate - fold > 28) return -1;
    unsigned log_rows = n_vars + log_inv_rate - fold;
    size_t rows = (size_t)1 << log_rows, fw = (size_t)1 << fold, w = fw * batch;
    size_t sub = ((size_t)1 << n_vars) / fw; /* coefficients per column */
    u64 *tw = make_twiddles(log_rows);
    long ncols = (long)(batch * fw);
#pragma omp parallel
    {
        u64 *col = (u64 *)malloc(32 * rows);
#pragma omp for schedule(dynamic, 1)
        for (long cj = 0; cj < ncols; cj++) {
            size_t b = (size_t)cj / fw, j = (size_t)cj % fw;
            const u64 *c = coeffs + 4 * (((size_t)b) << n_vars);
            memset(col, 0, 32 * rows);
            for (size_t t = 0; t < sub; t++) memcpy(col + 4 * t, c + 4 * (fw * t + j), 32);
            if (log_rows) ntt_with_table(col, log_rows, tw);
            for (size_t i = 0; i < rows; i++) memcpy(leaves + 4 * (i * w + b * fw + j), col + 4 * i, 32);
        }
        free(col);
    }
    free(tw);
    return 0;
}

/* ------------------------------------------------------------------ */
/* E1: univariate Horner                                                */
/* ------------------------------------------------------------------ */
static void horner(const u64 *c, size_t n, const u64 z[4], u64 out[4]) {
    u64 acc[4] = {0, 0, 0, 0};
    for (size_t i = n; i-- > 0;) {
        pko_fe_mul(acc, z, acc);
        pko_fe_add(acc, c + 4 * i, acc);
    }
    memcpy(out, acc, 32);
}
/* Horner's rule; long polynomials in blocks of 2^14 coefficients evaluated independently (all cores, as every other linear-size
 * step of the path) and combined by Horner in z^(2^14): the same field value, exact arithmetic */
void pko_eval_univariate(const u64 *c, size_t n, const u64 z[4], u64 out[4]) {
    const size_t B = (size_t)1 << 14;
    if (n <= 2 * B) {
        horner(c, n, z, out);
        return;
    }
    const size_t nb = (n + B - 1) / B;
    u64 *part = (u64 *)malloc(32 * nb);
#pragma omp parallel for schedule(static)
    for (long b = 0; b < (long)nb; b++) {
        size_t lo = (size_t)b * B, len = n - lo < B ? n - lo : B;
        horner(c + 4 * lo, len, z, part + 4 * b);
    }
    u64 zB[4];
    pko_fe_pow(z, (u64)B, zB);
    horner(part, nb, zB, out);
    free(part);
}

/* ------------------------------------------------------------------ */
/* S2: eq table (sumcheck.rs:146-171). First variable = MSB of index.  */
/* ------------------------------------------------------------------ */
static void eq_accumulate(const u64 *point, unsigned m, const u64 scalar[4], u64 *out) {
    /* recursion of eval_eq unrolled level by level into a scratch table */
    size_t n = (size_t)1 << m;
    u64 *t = (u64 *)malloc(32);
    memcpy(t, scalar, 32);
    for (unsigned j = 0; j < m; j++) {
        size_t cur = (size_t)1 << j;
        /* the recursion's halves are independent (rayon::join in the reference): write level j+1 into a second buffer */
        u64 *t2 = (u64 *)malloc(64 * cur);
#pragma omp parallel for schedule(static) if (cur >= 1024)
        for (long ii = 0; ii < (long)cur; ii++) {
            size_t i = (size_t)ii;
            u64 s1[4], s0[4];
            pko_fe_mul(t + 4 * i, point + 4 * j, s1); /* s1 = scalar * x   */
            pko_fe_sub(t + 4 * i, s1, s0);             /* s0 = scalar - s1  */
            memcpy(t2 + 4 * (2 * i), s0, 32);
            memcpy(t2 + 4 * (2 * i + 1), s1, 32);
        }
        free(t);
        t = t2;
    }
#pragma omp parallel for schedule(static) if (n >= 1024)
    for (long ii = 0; ii < (long)n; ii++) pko_fe_add(out + 4 * ii, t + 4 * ii, out + 4 * ii); /* out[0] += scalar */
    free(t);
}
void pko_eq_table(const u64 *r, unsigned m, u64 *out) {
    memset(out, 0, 32 * ((size_t)1 << m));
    eq_accumulate(r, m, R1, out);
}
void pko_eq_accumulate_point(u64 *w, unsigned n_vars, const u64 *point, const u64 scale[4]) {
    eq_accumulate(point, n_vars, scale, w);
}
void pko_eq_accumulate_univariate(u64 *w, unsigned n_vars, const u64 z[4], const u64 scale[4]) {
    u64 *pt = (u64 *)malloc(32 * (n_vars ? n_vars : 1));
    u64 acc[4];
    memcpy(acc, z, 32);
    for (unsigned i = 0; i < n_vars; i++) { /* utilities.go:182-190 ExpandFromUnivariate */
        memcpy(pt + 4 * (n_vars - 1 - i), acc, 32);
        pko_fe_mul(acc, acc, acc);
    }
    eq_accumulate(pt, n_vars, scale, w);
    free(pt);
}

/* ------------------------------------------------------------------ */
/* S3: cubic sumcheck round                                             */
/* ------------------------------------------------------------------ */
static inline void dbl_sub(const u64 x0[4], const u64 x1[4], u64 out[4]) { /* 2*x0 - x1 */
    u64 t[4];
    pko_fe_add(x0, x0, t);
    pko_fe_sub(t, x1, out);
}
int pko_sumcheck_cubic_round(u64 *a, u64 *b, u64 *c, u64 *eq, size_t len, const u64 *fold, u64 out[12]) {
    if (len < 2 || (len & (len - 1))) return -1;
    if (fold && len < 4) return -1;
    u64 *m[4] = {a, b, c, eq};
    size_t npairs, off;
    if (fold) { /* sumcheck.rs:92-97 */
        size_t q = len / 4;
        for (int k = 0; k < 4; k++)
#pragma omp parallel for schedule(static) if (q >= 1024)
            for (long ii = 0; ii < (long)q; ii++) {
                size_t i = (size_t)ii;
                u64 d[4];
                pko_fe_sub(m[k] + 4 * (2 * q + i), m[k] + 4 * i, d);
                pko_fe_mul(fold, d, d);
                pko_fe_add(m[k] + 4 * i, d, m[k] + 4 * i);
                pko_fe_sub(m[k] + 4 * (3 * q + i), m[k] + 4 * (q + i), d);
                pko_fe_mul(fold, d, d);
                pko_fe_add(m[k] + 4 * (q + i), d, m[k] + 4 * (q + i));
            }
        npairs = q;
        off = q;
    } else {
        npairs = len / 2;
        off = len / 2;
    }
    u64 acc[3][4];
    memset(acc, 0, sizeof acc);
#pragma omp parallel
    {
        u64 loc[3][4];
        memset(loc, 0, sizeof loc);
#pragma omp for schedule(static) nowait
        for (long ii = 0; ii < (long)npairs; ii++) { /* prover/src/whir_r1cs.rs:284-291 */
            size_t i = (size_t)ii;
            const u64 *a0 = a + 4 * i, *a1 = a + 4 * (i + off);
            const u64 *b0 = b + 4 * i, *b1 = b + 4 * (i + off);
            const u64 *c0 = c + 4 * i, *c1 = c + 4 * (i + off);
            const u64 *e0 = eq + 4 * i, *e1 = eq + 4 * (i + off);
            u64 t[4], u[4], v[4], w[4];
            pko_fe_mul(a0, b0, t);
            pko_fe_sub(t, c0, t);
            pko_fe_mul(e0, t, t);
            pko_fe_add(loc[0], t, loc[0]);
            dbl_sub(a0, a1, u);
            dbl_sub(b0, b1, v);
            pko_fe_mul(u, v, t);
            dbl_sub(c0, c1, w);
            pko_fe_sub(t, w, t);
            dbl_sub(e0, e1, u);
            pko_fe_mul(u, t, t);
            pko_fe_add(loc[1], t, loc[1]);
            pko_fe_sub(e1, e0, u);
            pko_fe_sub(a1, a0, v);
            pko_fe_mul(u, v, t);
            pko_fe_sub(b1, b0, v);
            pko_fe_mul(t, v, t);
            pko_fe_add(loc[2], t, loc[2]);
        }
#pragma omp critical
        for (int k = 0; k < 3; k++) pko_fe_add(acc[k], loc[k], acc[k]);
    }
    memcpy(out, acc, 96);
    return 0;
}

/* ------------------------------------------------------------------ */
/* S1/S4: sparse matrix x vector (sparse_matrix.rs:150-184)             */
/* ------------------------------------------------------------------ */
static int csr_check(size_t num_rows, const uint32_t *nri, size_t nnz) {
    for (size_t i = 0; i < num_rows; i++) {
        size_t s = nri[i], e = (i + 1 < num_rows) ? nri[i + 1] : nnz;
        if (s > e || e > nnz) return -1;
    }
    return 0;
}
int pko_spmv(size_t num_rows, size_t num_cols, const uint32_t *nri, const uint32_t *ci, const uint32_t *vals,
             size_t nnz, const u64 *interner, const u64 *x, u64 *y) {
    if (csr_check(num_rows, nri, nnz)) return -1;
    memset(y, 0, 32 * num_rows);
    for (size_t i = 0; i < num_rows; i++) {
        size_t s = nri[i], e = (i + 1 < num_rows) ? nri[i + 1] : nnz;
        for (size_t k = s; k < e; k++) {
            if (ci[k] >= num_cols) return -1;
            u64 t[4];
            pko_fe_mul(interner + 4 * (size_t)vals[k], x + 4 * (size_t)ci[k], t);
            pko_fe_add(y + 4 * i, t, y + 4 * i);
        }
    }
    return 0;
}
int pko_spmv_t(size_t num_rows, size_t num_cols, const uint32_t *nri, const uint32_t *ci, const uint32_t *vals,
               size_t nnz, const u64 *interner, const u64 *x, u64 *y) {
    if (csr_check(num_rows, nri, nnz)) return -1;
    memset(y, 0, 32 * num_cols);
    for (size_t i = 0; i < num_rows; i++) {
        size_t s = nri[i], e = (i + 1 < num_rows) ? nri[i + 1] : nnz;
        for (size_t k = s; k < e; k++) {
            if (ci[k] >= num_cols) return -1;
            u64 t[4];
            pko_fe_mul(interner + 4 * (size_t)vals[k], x + 4 * i, t);
            pko_fe_add(y + 4 * (size_t)ci[k], t, y + 4 * (size_t)ci[k]);
        }
    }
    return 0;
}
void pko_hadamard(const u64 *a, const u64 *b, u64 *c, size_t n) {
#pragma omp parallel for schedule(static) if (n >= 1024)
    for (long i = 0; i < (long)n; i++) pko_fe_mul(a + 4 * i, b + 4 * i, c + 4 * i);
}
void pko_vec_add(const u64 *a, const u64 *b, u64 *c, size_t n) {
#pragma omp parallel for schedule(static) if (n >= 1024)
    for (long i = 0; i < (long)n; i++) pko_fe_add(a + 4 * i, b + 4 * i, c + 4 * i);
}
/* c = a + s*b: the batching combination whir_r1cs.rs / whir batching use on whole vectors */
void pko_vec_axpy(const u64 *a, const u64 s[4], const u64 *b, u64 *c, size_t n) {
#pragma omp parallel for schedule(static) if (n >= 1024)
    for (long i = 0; i < (long)n; i++) {
        u64 t[4];
        pko_fe_mul(s, b + 4 * i, t);
        pko_fe_add(a + 4 * i, t, c + 4 * i);
    }
}
void pko_dot(const u64 *w, const u64 *f, size_t n, u64 out[4]) {
    u64 acc[4] = {0, 0, 0, 0};
#pragma omp parallel
    {
        u64 loc[4] = {0, 0, 0, 0};
#pragma omp for schedule(static) nowait
        for (long i = 0; i < (long)n; i++) {
            u64 t[4];
            pko_fe_mul(w + 4 * i, f + 4 * i, t);
            pko_fe_add(loc, t, loc);
        }
#pragma omp critical
        pko_fe_add(acc, loc, acc);
    }
    memcpy(out, acc, 32);
}

/* ------------------------------------------------------------------ */
/* W1..W3                                                               */
/* ------------------------------------------------------------------ */
void pko_fold_coeffs(const u64 *coeffs, unsigned n_vars, const u64 *r, unsigned k, u64 *out) {
    size_t n = (size_t)1 << n_vars;
    u64 *tmp = (u64 *)malloc(32 * n);
    memcpy(tmp, coeffs, 32 * n);
    for (unsigned b = 0; b < k; b++) { /* MultivarPoly: vars[0] <-> index bit 0 (utilities.go:15-22) */
        n >>= 1;
        u64 *nx = (u64 *)malloc(32 * (n ? n : 1));
#pragma omp parallel for schedule(static) if (n >= 1024)
        for (long ii = 0; ii < (long)n; ii++) {
            size_t i = (size_t)ii;
            u64 t[4];
            pko_fe_mul(tmp + 4 * (2 * i + 1), r + 4 * b, t);
            pko_fe_add(tmp + 4 * (2 * i), t, nx + 4 * i);
        }
        free(tmp);
        tmp = nx;
    }
    memcpy(out, tmp, 32 * n);
    free(tmp);
}
void pko_fold_pairs(u64 *v, size_t len, const u64 r[4]) {
    size_t h = len / 2;
    u64 *tmp = (u64 *)malloc(32 * (h ? h : 1));
#pragma omp parallel for schedule(static) if (h >= 1024)
    for (long ii = 0; ii < (long)h; ii++) {
        size_t i = (size_t)ii;
        u64 d[4];
        pko_fe_sub(v + 4 * (2 * i + 1), v + 4 * (2 * i), d);
        pko_fe_mul(d, r, d);
        pko_fe_add(v + 4 * (2 * i), d, tmp + 4 * i);
    }
    memcpy(v, tmp, 32 * h);
    free(tmp);
}
int pko_sumcheck_quadratic_round(u64 *f, u64 *w, size_t len, const u64 *fold, u64 out[12]) {
    if (len < 1 || (len & (len - 1))) return -1;
    if (fold) {
        if (len < 2) return -1;
        pko_fold_pairs(f, len, fold);
        pko_fold_pairs(w, len, fold);
        len /= 2;
    }
    if (len < 2) return -1;
    u64 acc[3][4];
    memset(acc, 0, sizeof acc);
#pragma omp parallel
    {
        u64 loc[3][4];
        memset(loc, 0, sizeof loc);
#pragma omp for schedule(static) nowait
        for (long ii = 0; ii < (long)(len / 2); ii++) {
            size_t i = (size_t)ii;
            const u64 *f0 = f + 8 * i, *f1 = f + 8 * i + 4, *w0 = w + 8 * i, *w1 = w + 8 * i + 4;
            u64 t[4], u[4], v[4];
            pko_fe_mul(f0, w0, t);
            pko_fe_add(loc[0], t, loc[0]);
            pko_fe_mul(f1, w1, t);
            pko_fe_add(loc[1], t, loc[1]);
            dbl_sub(f1, f0, u); /* f(2) = 2 f1 - f0 */
            dbl_sub(w1, w0, v);
            pko_fe_mul(u, v, t);
            pko_fe_add(loc[2], t, loc[2]);
        }
#pragma omp critical
        for (int k = 0; k < 3; k++) pko_fe_add(acc[k], loc[k], acc[k]);
    }
    memcpy(out, acc, 96);
    return 0;
}

/* ------------------------------------------------------------------ */
/* P1: proof of work (skyscraper/core/src/pow.rs)                       */
/* ------------------------------------------------------------------ */
static void f64_to_u256(double f, u64 out[4]) { /* pow.rs:44-82 */
    u64 bits;
    memcpy(&bits, &f, 8);
    int sign = (int)(bits >> 63);
    int exp_bits = (int)((bits >> 52) & 0x7ff);
    u64 frac = bits & ((1ULL << 52) - 1);
    int exp;
    u64 significand;
    if (exp_bits == 0) {
        exp = -1022;
        significand = frac;
    } else {
        exp = exp_bits - 1023;
        significand = frac + (1ULL << 52);
    }
    memset(out, 0, 32);
    if (sign) return;
    if (exp > 256) {
        memset(out, 0xff, 32);
        return;
    }
    int shift = exp - 52;
    if (shift < 0) {
        double r = round(f);
        out[0] = (r >= 18446744073709551616.0) ? UINT64_MAX : (r > 0 ? (u64)r : 0);
    } else {
        unsigned limb = (unsigned)shift / 64, sh = (unsigned)shift % 64;
        if (limb > 3) return; /* unreachable for exp <= 256 except exp==256+: rust would panic on index */
        out[limb] = significand << sh;
        if (sh != 0 && limb < 3) out[limb + 1] = significand >> (64 - sh);
    }
}
int pko_pow_threshold(double difficulty, u64 out[4]) { /* pow.rs:14-22 */
    if (!(difficulty >= 0.0 && difficulty < 80.0)) return -1;
    double modulus = (double)P[3] * ldexp(1.0, 192);
    double prob = exp2(-difficulty);
    f64_to_u256(prob * modulus, out);
    return 0;
}
int pko_pow_verify(const u64 challenge[4], double difficulty, u64 nonce) { /* pow.rs:24-26 */
    if (difficulty == 0.0) return 1;
    u64 thr[4], h[4], n[4] = {nonce, 0, 0, 0};
    if (pko_pow_threshold(difficulty, thr)) return 0;
    pko_compress(challenge, n, h);
    return lt(h, thr);
}
u64 pko_pow_solve(const u64 challenge[4], double difficulty) { /* pow.rs:33-41; smallest valid nonce */
    if (difficulty == 0.0) return 0;
    u64 thr[4];
    pko_pow_threshold(difficulty + 0.01, thr);
    /* the reference grinds on all cores (generic.rs:42-71: rayon::broadcast + fetch_min); here: windows of nonces searched in
     * parallel, the smallest hit of the first window that has one -- i.e. still the globally smallest valid nonce */
    const u64 window = 1u << 14;
    for (u64 base = 0;; base += window) {
        u64 best = ~(u64)0;
#pragma omp parallel for schedule(static) reduction(min : best)
        for (long long t = 0; t < (long long)window; t++) {
            u64 nonce = base + (u64)t;
            u64 h[4], n[4] = {nonce, 0, 0, 0};
            pko_compress(challenge, n, h);
            if (lt(h, thr) && nonce < best) best = nonce;
        }
        if (best != ~(u64)0) return best;
    }
}

/* ------------------------------------------------------------------ */
/* The proof's random draws.  The reference takes the ZK mask, the random polynomial g and the blinding univariates from
 * rand's thread_rng (provekit/common/src/utils/zk_utils.rs:13-22, provekit/prover/src/whir_r1cs.rs:197,212-221): ChaCha12 under
 * an OS-seeded key -- nothing to be bit-exact with.  The HIP library expands ONE 256-bit key per proof with the same cipher
 * (csrc/prover.hip random_fe_kernel); with an injected key its draws are reproducible, and this is their restatement: elements
 * 2j and 2j+1 of draw `stream` come from the ChaCha12 blocks (counter = j, nonce = {stream, attempt}), attempt = 0, 1, ...:
 * words 0..7 / 8..15 masked to 254 bits, accepted iff < p (ark-ff Fp::rand's rejection), stored as they are (a uniform
 * Montgomery image is a uniform element). */
/* ------------------------------------------------------------------ */
#define PKO_QR(a, b, c, d)                          \
    a += b; d ^= a; d = (d << 16) | (d >> 16);      \
    c += d; b ^= c; b = (b << 12) | (b >> 20);      \
    a += b; d ^= a; d = (d << 8) | (d >> 24);       \
    c += d; b ^= c; b = (b << 7) | (b >> 25)
void pko_chacha_block(const uint8_t key[32], u64 counter, uint32_t n0, uint32_t n1, int rounds, uint32_t out[16]) { /* RFC 8439 2.3 */
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    memcpy(s + 4, key, 32);
    s[12] = (uint32_t)counter;
    s[13] = (uint32_t)(counter >> 32);
    s[14] = n0;
    s[15] = n1;
    uint32_t x[16];
    memcpy(x, s, 64);
    for (int r = 0; r < rounds / 2; r++) {
        PKO_QR(x[0], x[4], x[8], x[12]);
        PKO_QR(x[1], x[5], x[9], x[13]);
        PKO_QR(x[2], x[6], x[10], x[14]);
        PKO_QR(x[3], x[7], x[11], x[15]);
        PKO_QR(x[0], x[5], x[10], x[15]);
        PKO_QR(x[1], x[6], x[11], x[12]);
        PKO_QR(x[2], x[7], x[8], x[13]);
        PKO_QR(x[3], x[4], x[9], x[14]);
    }
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}
void pko_random_fe(const uint8_t key[32], uint32_t stream, u64 *out, size_t n) {
    const size_t pairs = (n + 1) / 2;
#pragma omp parallel for schedule(static) if (pairs >= 1024)
    for (long jj = 0; jj < (long)pairs; jj++) {
        const size_t j = (size_t)jj;
        int done[2] = {0, 2 * j + 1 >= n};
        for (uint32_t attempt = 0; !(done[0] && done[1]); attempt++) {
            uint32_t blk[16];
            pko_chacha_block(key, (u64)j, stream, attempt, 12, blk);
            for (int half = 0; half < 2; half++) {
                if (done[half]) continue;
                u64 x[4];
                for (int w = 0; w < 4; w++) x[w] = (u64)blk[8 * half + 2 * w] | ((u64)blk[8 * half + 2 * w + 1] << 32);
                x[3] &= 0x3fffffffffffffffULL; /* < 2^254 */
                if (lt(x, P)) {
                    memcpy(out + 4 * (2 * j + half), x, 32);
                    done[half] = 1;
                }
            }
        }
    }
}

/* opened leaves of a leaf-major codeword as ark-serialize writes them: rows idx[0..k) of leaves (width FEs each), canonical */
void pko_gather_rows_canonical(const u64 *leaves_mont, size_t width, const u64 *idx, size_t k, u64 *out_canon) {
    for (size_t q = 0; q < k; q++) pko_fe_from_mont_many(leaves_mont + 4 * width * idx[q], out_canon + 4 * width * q, width);
}

int pko_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void pko_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
