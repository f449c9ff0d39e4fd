// selftest.hip -- host-side execution of the exact __host__ __device__ arithmetic the kernels use
// (fe29.hpp, skyscraper29.hpp), so the CPU test suite can check it against the oracle without a GPU.
#include "ctx.hpp"
#include "skyscraper29.hpp"
#include "transcript.hpp"

using namespace pk;

static fe load_host(const uint64_t* p) {
    fe r;
    memcpy(r.v, p, 32);
    return r;
}
static void store_host(uint64_t* p, const fe& x) { memcpy(p, x.v, 32); }

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
        default: break;
    }
    return r;
}

__global__ void selftest_kernel(int op, const fe* a, const fe* b, fe* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe x = fe_load(a + i), y = b ? fe_load(b + i) : x;
    fe_store(out + i, selftest_op(op, x, y));
}

extern "C" {

// the same ops executed by a kernel (device pointers): lets the GPU suite diff device vs host codegen
int pk_selftest_arith_device(pk_ctx* ctx, int op, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, size_t n) {
    if (!ctx || !d_a || !d_out) return PK_ERR_BAD_ARG;
    if (!n) return PK_OK;
    selftest_kernel<<<(unsigned)((n + 63) / 64), 64, 0, ctx->stream>>>(op, (const fe*)d_a, (const fe*)d_b, (fe*)d_out, n);
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}

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
    if (!a || !out || (!b && (op == 0 || op == 1 || op == 2 || op == 4))) return PK_ERR_BAD_ARG;
    for (size_t i = 0; i < n; i++) {
        fe x = load_host(a + 4 * i), y = b ? load_host(b + 4 * i) : x;
        if (op < 0 || op > 13) return PK_ERR_BAD_ARG;
        fe r = selftest_op(op, x, y);
        store_host(out + 4 * i, r);
    }
    return PK_OK;
}

}  // extern "C"
