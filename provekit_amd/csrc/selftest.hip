// selftest.hip -- host-side execution of the exact __host__ __device__ arithmetic the kernels use
// (fe29.hpp, skyscraper29.hpp), so the CPU test suite can check it against the oracle without a GPU; and the two host-only
// pieces of the transcript the parity tests pin (domain-separator tag, sponge permutation).  Measurement probes and rejected
// prototypes live in tools/probes (libpk_probes.so), not in the product.
#include "selftest_ops.hpp"
#include "ntt_regs.hpp"
#include "transcript.hpp"

using namespace pk;

// the NTT's register butterfly network on the host (ntt_regs.hpp dft_regs<le, d>): n_groups x 2^le values below 1.2p in, the network's
// outputs out.  tw (may be NULL): one multiplier below p per value, applied to the network's UNREDUCED outputs the way the pass kernel
// applies its inter-round / inter-pass twiddles (Shoup product with the table's quotient); without it the outputs are only reduced.
template <int LE, int D>
static int selftest_dft(const uint64_t* in, const uint64_t* tw, uint64_t* out, size_t n_groups) {
    constexpr int NX = 1 << LE;
    for (size_t g = 0; g < n_groups; g++) {
        fe29 x[NX];
        for (int i = 0; i < NX; i++) x[i] = unpack29<0>(load_host(in + 4 * (NX * g + i)));
        dft_regs<LE, D>(x);
        for (int i = 0; i < NX; i++) {
            fe29 y;
            if (tw) {
                tw29s t;
                t.w = unpack29<0>(load_host(tw + 4 * (NX * g + i)));
                t.wq = shoup_quotient29(t.w);
                y = mul_tw(x[i], t);
                for (int k = 0; k < 9; k++)
                    if (y.v[k] >> 29) return PK_ERR_BAD_ARG;  // a product leaves normalised limbs
            } else {
                y = (i >> (LE - D)) == 0 ? red29(x[i]) : red29w(x[i]);  // as the pass kernel: frequency digit 0 is the sum of sums
            }
            store_host(out + 4 * (NX * g + i), pack29(cond_sub_p29(y)));
        }
    }
    return PK_OK;
}
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
        if (op < 0 || op > 26) return PK_ERR_BAD_ARG;
        fe r = selftest_op(op, x, y);
        store_host(out + 4 * i, r);
    }
    return PK_OK;
}

int pk_selftest_dft(const uint64_t* in, const uint64_t* tw, uint64_t* out, int le, int d, size_t n_groups) {
    if (!in || !out) return PK_ERR_BAD_ARG;
    if (le == 3 && d == 1) return selftest_dft<3, 1>(in, tw, out, n_groups);
    if (le == 3 && d == 2) return selftest_dft<3, 2>(in, tw, out, n_groups);
    if (le == 2 && d == 1) return selftest_dft<2, 1>(in, tw, out, n_groups);
    if (le == 2 && d == 2) return selftest_dft<2, 2>(in, tw, out, n_groups);
    return PK_ERR_BAD_ARG;
}

}  // extern "C"
