// selftest.hip -- host-side execution of the exact __host__ __device__ arithmetic the kernels use
// (fe29.hpp, skyscraper29.hpp), so the CPU test suite can check it against the oracle without a GPU; and the two host-only
// pieces of the transcript the parity tests pin (domain-separator tag, sponge permutation).  Measurement probes and rejected
// prototypes live in tools/probes (libpk_probes.so), not in the product.
#include "selftest_ops.hpp"
#include "transcript.hpp"

using namespace pk;

extern "C" {

// domain-separator tag (Keccak duplex, overwrite mode) and one Skyscraper sponge permutation, host only
int pk_selftest_keccak_tag(const uint8_t* data, size_t len, uint8_t tag[32]) {
    if (!tag || (len && !data)) return PK_ERR_BAD_ARG;
    keccak_tag(std::string((const char*)data, len), tag);
    return PK_OK;
}
int pk_selftest_permute(uint64_t l[4], uint64_t r[4]) {
    if (!l || !r) return PK_ERR_BAD_ARG;
    fe a = load_host(l), b = load_host(r);
    sky_permute_host(a, b);
    store_host(l, a);
    store_host(r, b);
    return PK_OK;
}

// op: 0 fe_mul29(a,b)  1 compress v2  2 compress v1  3 from_mont  4 a*b*2^-256 via mont256_29  5 a^2*2^-256 via sqr256_29
// a, b, out: n field elements (4 x u64 each).  Host only; no device needed.
int pk_selftest_arith(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    if (!a || !out || (!b && (op == 0 || op == 1 || op == 2 || op == 4 || op == 15 || op == 16 || op == 19 || op == 20 || op == 24 || op == 25))) return PK_ERR_BAD_ARG;
    for (size_t i = 0; i < n; i++) {
        fe x = load_host(a + 4 * i), y = b ? load_host(b + 4 * i) : x;
        if (op < 0 || op > 25) return PK_ERR_BAD_ARG;
        fe r = selftest_op(op, x, y);
        store_host(out + 4 * i, r);
    }
    return PK_OK;
}

}  // extern "C"
