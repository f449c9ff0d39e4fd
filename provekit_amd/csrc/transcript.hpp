// transcript.hpp -- host-side Fiat-Shamir transcript: spongefish's DuplexSponge over the Skyscraper
// permutation.
//
// Mirrors provekit/common/src/skyscraper/sponge.rs:42-60 (state = 2 field elements, rate 1, IV in the
// capacity element, permutation = skyscraper::reference::permute) and the spongefish duplex discipline
// (overwrite mode: absorbing replaces the rate element; a squeeze after an absorb permutes first).
// spongefish is an un-pinned, un-vendored git dependency of the reference (Cargo.toml:130-131), so the byte
// framing below follows what the in-tree Go verifier consumes (recursive-verifier/app/circuit/common.go:30-105,
// utilities/utilities.go:84-101): scalars = 32-byte canonical little-endian, absorbed as field elements; hints =
// u32-LE length + payload, not absorbed; PoW nonce = 8 bytes big-endian, absorbed byte-wise; challenge bytes are
// taken 15 at a time from squeezed elements (spongefish's bytes_uniform_modp for a 254-bit modulus).
// The sponge IV is the Keccak tag of the IO pattern's bytes (DomainSeparator::as_bytes()); a caller that holds the
// reference's `create_io_pattern()` hands those bytes over (pk_scheme_set_io_pattern) and the transcript then both starts
// from the reference's IV and ENFORCES the pattern op by op the way spongefish's HashStateWithInstructions does
// (absorb / squeeze / hint against the declared stack), so a proof that would trip the reference verifier's pattern check
// fails here first.  Without one the library's own restatement of the pattern is used (prover.hip `whir_r1cs_io_pattern`:
// provekit's labels from the tree, whir's recalled -- DESIGN.md 6 lists which are pinned by the Go verifier's parser).
#pragma once
#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "skyscraper29.hpp"

namespace pk {

// ---- host field helpers (fe holds 8 x u32; Montgomery unless named canon) -----------------------
inline fe h_load(const uint64_t* p) {
    fe r;
    memcpy(r.v, p, 32);
    return r;
}
inline void h_store(uint64_t* p, const fe& x) { memcpy(p, x.v, 32); }
inline fe h_mul(const fe& a, const fe& b) { return fe_mulx(a, b); }
inline fe h_add(const fe& a, const fe& b) { return fe_add(a, b); }
inline fe h_sub(const fe& a, const fe& b) { return fe_sub(a, b); }
inline fe h_to_canon(const fe& mont) { return fe_from_montx(mont); }
inline fe h_from_canon(const fe& canon) { return fe_to_montx(fe_reduce_any(canon)); }
inline fe h_from_u64(uint64_t v) {
    fe c = fe_zero();
    c.v[0] = (u32)v;
    c.v[1] = (u32)(v >> 32);
    return h_from_canon(c);
}
inline fe h_pow(fe base, uint64_t e) {
    fe acc = fe_one();
    while (e) {
        if (e & 1) acc = h_mul(acc, base);
        base = h_mul(base, base);
        e >>= 1;
    }
    return acc;
}
// 1/2 (provekit/common/src/utils/mod.rs:24-26)
inline fe h_half() {
    // (p+1)/2 canonical
    fe c;
    const uint64_t l[4] = {0xa1f0fac9f8000001ULL, 0x9419f4243cdcb848ULL, 0xdc2822db40c0ac2eULL, 0x183227397098d014ULL};
    memcpy(c.v, l, 32);
    return h_from_canon(c);
}

// ---- Keccak-f[1600], only to derive the sponge IV from the domain separator (spongefish tag) ----
inline void keccak_f1600(uint64_t s[25]) {
    static const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
                                    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
                                    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                                    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
                                    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                                    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    for (int round = 0; round < 24; round++) {
        uint64_t bc[5];
        for (int i = 0; i < 5; i++) bc[i] = s[i] ^ s[i + 5] ^ s[i + 10] ^ s[i + 15] ^ s[i + 20];
        for (int i = 0; i < 5; i++) {
            uint64_t t = bc[(i + 4) % 5] ^ ((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63));
            for (int j = 0; j < 25; j += 5) s[j + i] ^= t;
        }
        uint64_t t = s[1];
        for (int i = 0; i < 24; i++) {
            int j = PIL[i];
            uint64_t b = s[j];
            s[j] = (t << ROT[i]) | (t >> (64 - ROT[i]));
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = s[j + i];
            for (int i = 0; i < 5; i++) s[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        s[0] ^= RC[round];
    }
}
// duplex sponge over bytes (rate 136), overwrite mode, zero IV: absorb `data`, squeeze 32 bytes
inline void keccak_tag(const std::string& data, uint8_t tag[32]) {
    uint64_t st[25];
    memset(st, 0, sizeof st);
    uint8_t* b = reinterpret_cast<uint8_t*>(st);
    const size_t R = 136;
    size_t pos = 0, i = 0;
    while (i < data.size()) {
        if (pos == R) {
            keccak_f1600(st);
            pos = 0;
        } else {
            size_t chunk = std::min(data.size() - i, R - pos);
            memcpy(b + pos, data.data() + i, chunk);
            pos += chunk;
            i += chunk;
        }
    }
    keccak_f1600(st);
    memcpy(tag, b, 32);
}

// Skyscraper v2 permutation on canonical values (skyscraper/core/src/reference.rs:49-60), host flavour:
// 4 x 64-bit limbs with unsigned __int128 (the CPU has a 64x64->128 multiplier; the 29-bit layout above is
// shaped for the GPU's 32x32+64 mad).  Checked against the Python restatement by tests/test_fe29_host.py.
namespace host64 {
typedef unsigned __int128 u128;
static const uint64_t P64[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t NP64 = 0xc2e1f593efffffffULL;  // -p^-1 mod 2^64
inline bool geq_p(const uint64_t a[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > P64[i]) return true;
        if (a[i] < P64[i]) return false;
    }
    return true;
}
inline void sub_p(uint64_t a[4]) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - P64[i] - borrow;
        a[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
}
// a*b*2^-256 mod p, inputs < p
inline void mont_mul(const uint64_t a[4], const uint64_t b[4], uint64_t r[4]) {
    uint64_t t[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a[j] * b[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        uint64_t t4 = t[4] + (uint64_t)c;  // p < 2^254 keeps the running value below 2^320
        uint64_t m = t[0] * NP64;
        c = (u128)m * P64[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * P64[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t4;
        t[3] = (uint64_t)c;
        t[4] = (uint64_t)(c >> 64);
    }
    for (int i = 0; i < 4; i++) r[i] = t[i];
    if (t[4] || geq_p(r)) sub_p(r);
}
inline void add_mod(uint64_t a[4], const uint64_t b[4]) {  // a = a + b mod p, both < p
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a[i] + b[i];
        a[i] = (uint64_t)c;
        c >>= 64;
    }
    if (geq_p(a)) sub_p(a);
}
inline uint64_t sbox8(uint64_t v) {  // bar.rs:63-67
    uint64_t t1 = ((v & 0x8080808080808080ULL) >> 7) | ((v & 0x7f7f7f7f7f7f7f7fULL) << 1);
    uint64_t t2 = ((v & 0xc0c0c0c0c0c0c0c0ULL) >> 6) | ((v & 0x3f3f3f3f3f3f3f3fULL) << 2);
    uint64_t t3 = ((v & 0xe0e0e0e0e0e0e0e0ULL) >> 5) | ((v & 0x1f1f1f1f1f1f1f1fULL) << 3);
    uint64_t x = (~t1 & t2 & t3) ^ v;
    return ((x & 0x8080808080808080ULL) >> 7) | ((x & 0x7f7f7f7f7f7f7f7fULL) << 1);
}
inline void bar(const uint64_t x[4], uint64_t out[4]) {  // reference.rs:80-94
    out[0] = sbox8(x[2]);
    out[1] = sbox8(x[3]);
    out[2] = sbox8(x[0]);
    out[3] = sbox8(x[1]);
    while (geq_p(out)) sub_p(out);
}
inline void permute(uint64_t l[4], uint64_t r[4]) {
    while (geq_p(l)) sub_p(l);
    while (geq_p(r)) sub_p(r);
    for (int i = 0; i < 18; i++) {
        uint64_t f[4];
        if (i == 6 || i == 7 || i == 10 || i == 11)
            bar(l, f);
        else
            mont_mul(l, l, f);
        add_mod(f, r);
        uint64_t rc[4];
        for (int k = 0; k < 4; k++) rc[k] = (uint64_t)rc_limb(i, 2 * k) | ((uint64_t)rc_limb(i, 2 * k + 1) << 32);
        add_mod(f, rc);
        for (int k = 0; k < 4; k++) {
            r[k] = l[k];
            l[k] = f[k];
        }
    }
}
}  // namespace host64

inline void sky_permute_host(fe& l_c, fe& r_c) {
    uint64_t l[4], r[4];
    memcpy(l, l_c.v, 32);
    memcpy(r, r_c.v, 32);
    host64::permute(l, r);
    memcpy(l_c.v, l, 32);
    memcpy(r_c.v, r, 32);
}

// ---- spongefish IO pattern: "<protocol id>" then "\0<A|S><count><label>", "\0H<label>", "\0R" ------------------------
// DomainSeparator::finalize: parse, reject zero counts, merge neighbouring absorbs / squeezes.
struct IoOp {
    char kind;     // 'A' absorb, 'S' squeeze, 'H' hint, 'R' ratchet
    size_t count;  // units (field elements); 1 for H / R
};
inline bool io_pattern_parse(const std::string& bytes, std::vector<IoOp>& ops, std::string& err) {
    ops.clear();
    size_t pos = bytes.find('\0');
    if (pos == std::string::npos) return true;  // a bare protocol id: no operations
    size_t index = 0;
    while (pos != std::string::npos) {
        const size_t next = bytes.find('\0', pos + 1);
        const std::string part = bytes.substr(pos + 1, next == std::string::npos ? std::string::npos : next - pos - 1);
        pos = next;
        index++;
        if (part.empty()) {
            err = "IO pattern: empty operation #" + std::to_string(index);
            return false;
        }
        const char kind = part[0];
        if (kind == 'H' || kind == 'R') {
            ops.push_back({kind, 1});
            continue;
        }
        if (kind != 'A' && kind != 'S') {
            err = "IO pattern: operation #" + std::to_string(index) + " has unknown kind";
            return false;
        }
        size_t i = 1, count = 0;
        while (i < part.size() && part[i] >= '0' && part[i] <= '9' && count < ((size_t)1 << 40)) count = count * 10 + (size_t)(part[i++] - '0');
        if (count == 0) {
            err = "IO pattern: operation #" + std::to_string(index) + " has a zero or missing count";
            return false;
        }
        if (!ops.empty() && ops.back().kind == kind)
            ops.back().count += count;
        else
            ops.push_back({kind, count});
    }
    return true;
}

class Transcript {
  public:
    std::vector<uint8_t> narg;  // the proof string (WhirR1CSProof::transcript)
    double permute_seconds = 0.0;  // host time spent in the sponge permutation (PK_PROVE_TIMING)
    double hint_seconds = 0.0;     // host time spent serialising opening hints (PK_PROVE_TIMING)
    unsigned permutes = 0;

    explicit Transcript(const std::string& io_pattern) {
        uint8_t iv[32];
        keccak_tag(io_pattern, iv);  // HashStateWithInstructions::generate_tag
        st_[0] = fe_zero();
        fe c;
        memcpy(c.v, iv, 32);
        st_[1] = fe_reduce_any(c);  // FieldElement::new(bigint_from_bytes_le(iv)), sponge.rs:46-49
        if (!io_pattern_parse(io_pattern, ops_, violation_)) ops_.clear();
    }
    // "" while every operation so far matched the declared pattern; otherwise the first mismatch (spongefish: InvalidIOPattern)
    const std::string& violation() const { return violation_; }
    // ... and nothing declared was left undone (spongefish checks this when the state is dropped)
    bool finished() const { return violation_.empty() && op_ == ops_.size(); }
    // prover -> verifier: field elements (Montgomery in memory), written canonical LE and absorbed
    void add_scalars(const fe* mont, size_t n) {
        for (size_t i = 0; i < n; i++) add_canon(h_to_canon(mont[i]));
    }
    void add_scalar(const fe& mont) { add_scalars(&mont, 1); }
    // a digest is already a canonical value (provekit/common/src/skyscraper/whir.rs:96-102)
    void add_canon(const fe& canon) {
        expect('A', 1);
        append(canon.v, 32);
        absorb(canon);
    }
    // verifier -> prover
    fe challenge_scalar() {
        expect('S', 1);
        return h_from_canon(squeeze());
    }
    void challenge_scalars(fe* out, size_t n) {
        for (size_t i = 0; i < n; i++) out[i] = challenge_scalar();
    }
    void challenge_bytes(uint8_t* out, size_t n) {
        expect('S', (n + 14) / 15);
        while (n) {
            fe c = squeeze();
            size_t take = n < 15 ? n : 15;
            memcpy(out, c.v, take);
            out += take;
            n -= take;
        }
    }
    void add_bytes(const uint8_t* b, size_t n) {
        expect('A', n);
        append(b, n);
        for (size_t i = 0; i < n; i++) {
            fe c = fe_zero();
            c.v[0] = b[i];
            absorb(c);
        }
    }
    void hint(const void* payload, size_t len) {
        expect('H', 1);
        uint32_t l = (uint32_t)len;
        append(&l, 4);
        append(payload, len);
    }

  private:
    void permute() {
        auto t0 = std::chrono::steady_clock::now();
        sky_permute_host(st_[0], st_[1]);
        permute_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        permutes++;
    }
    fe st_[2];
    int absorb_pos_ = 0, squeeze_pos_ = 1;  // R = 1
    std::vector<IoOp> ops_;
    size_t op_ = 0, used_ = 0;  // position in ops_, units of ops_[op_] already consumed
    std::string violation_;
    void expect(char kind, size_t n) {
        if (!n || !violation_.empty()) return;
        if (op_ >= ops_.size() || ops_[op_].kind != kind || ops_[op_].count - used_ < n) {
            violation_ = std::string("transcript operation ") + kind + std::to_string(n) + " does not follow the IO pattern: operation #" +
                         std::to_string(op_ + 1) + " is " +
                         (op_ < ops_.size() ? std::string(1, ops_[op_].kind) + std::to_string(ops_[op_].count - used_) + " (remaining)" : std::string("past the end"));
            return;
        }
        used_ += n;
        if (used_ == ops_[op_].count) {
            op_++;
            used_ = 0;
        }
    }
    void append(const void* p, size_t n) {
        const uint8_t* b = static_cast<const uint8_t*>(p);
        narg.insert(narg.end(), b, b + n);
    }
    void absorb(const fe& canon) {
        if (absorb_pos_ == 1) {
            permute();
            absorb_pos_ = 0;
        }
        st_[0] = canon;
        absorb_pos_ = 1;
        squeeze_pos_ = 1;
    }
    fe squeeze() {
        if (squeeze_pos_ == 1) {
            squeeze_pos_ = 0;
            absorb_pos_ = 0;
            permute();
        }
        squeeze_pos_ = 1;
        return st_[0];
    }
};

}  // namespace pk
