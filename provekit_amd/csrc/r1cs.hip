// r1cs.hip -- sparse R1CS matrix x vector products.
//
// Replaces HydratedSparseMatrix * &[FieldElement] and &[FieldElement] * HydratedSparseMatrix
// (provekit/common/src/sparse_matrix.rs:150-184), which the reference runs serially per matrix
// ("OPT: Paralelize").  The matrices are fixed per proof scheme, so the handle uploads them once
// -- in the reference's own CSR form (new_row_indices / col_indices / interned value indices,
// sparse_matrix.rs:12-27) plus a CSC copy built here -- and both products become atomic-free
// gathers: one lane per output row (A*z) or per output column (eq^T * A).
//
// Heavy lines.  Real constraint systems have a few very long lines -- the column of the constant-one witness is touched by every
// constraint with a constant term, a LogUp grand sum is one row with a term per lookup -- and a lane that walks 10^5 entries holds
// the kernel for 10^5 dependent products.  Lines longer than HEAVY_DEGREE are therefore found once per R1CS, cut into chunks of
// HEAVY_CHUNK entries, and summed by a workgroup per chunk before the gather runs; the gather's lane then reads the finished sum.
// Field addition is exact, so the order of summation does not show in the result.
#include <algorithm>
#include <vector>

// memory- / latency-bound kernels: their wavefronts issue ahead of the ALU-bound hash / NTT / grinder kernels they share SIMDs with
#define PK_BASE_PRIO 2
#include "ctx.hpp"
#include "fe29.hpp"
#include "reduce.hpp"

using namespace pk;

namespace pk {
int witness_bounds_strided(pk_ctx* ctx, const pk_r1cs* r, const uint64_t* d_z, unsigned m0, unsigned stride, unsigned offset, uint64_t* d_a,
                           uint64_t* d_b, uint64_t* d_c);
int external_row_range(pk_ctx* ctx, const pk_r1cs* r, const uint64_t* d_eq_alpha, size_t first, size_t last, uint64_t* d_out);
}

constexpr uint32_t HEAVY_DEGREE = 64;    // longer lines are summed by workgroups
constexpr uint32_t HEAVY_CHUNK = 2048;   // entries per workgroup
struct heavy_chunk {
    uint32_t slot, begin, end;  // entries [begin, end) of the line set's arrays belong to heavy line number `slot` (R1CS-wide numbering)
};
struct pk_r1cs {
    size_t num_constraints = 0, num_witnesses = 0, n_interned = 0;
    fe* d_interner = nullptr;
    // per matrix: CSR (ptr has num_constraints+1 entries) and CSC (ptr has num_witnesses+1 entries)
    uint32_t *csr_ptr[3] = {}, *csr_idx[3] = {}, *csr_val[3] = {};
    uint32_t *csc_ptr[3] = {}, *csc_idx[3] = {}, *csc_val[3] = {};
    size_t nnz[3] = {};
    // heavy lines of the six line sets (0..2: rows of A, B, C; 3..5: their columns): sorted line indices per set, slots numbered
    // through all sets, the chunks of set s at d_chunks[chunk0[s] .. chunk0[s] + n_chunks[s])
    uint32_t* d_heavy_lines[6] = {};
    uint32_t n_heavy[6] = {}, slot0[6] = {}, chunk0[6] = {}, n_chunks[6] = {};
    heavy_chunk* d_chunks = nullptr;
    uint32_t *d_slot_chunk0 = nullptr, *d_slot_nchunks = nullptr;  // per slot: its chunks (absolute indices into d_chunks)
    size_t n_slots = 0, n_chunks_total = 0;
};

namespace {

struct line_set {  // one of the six (ptr, idx, val) triples with its heavy lines
    const uint32_t *ptr, *idx, *val;
    const uint32_t* heavy_lines;  // sorted
    uint32_t n_heavy, slot0;
};
__device__ __forceinline__ fe sparse_row_dot(const line_set& L, const fe* __restrict__ interner, const fe* __restrict__ x,
                                             const fe* __restrict__ heavy_vals, size_t i) {
    uint32_t k = L.ptr[i];
    const uint32_t e = L.ptr[i + 1];
    if (e - k > HEAVY_DEGREE) {  // summed beforehand (heavy_dot_kernel / heavy_sum_kernel): find the line's slot
        uint32_t lo = 0, hi = L.n_heavy;
        while (lo + 1 < hi) {
            const uint32_t mid = (lo + hi) / 2;
            if (L.heavy_lines[mid] <= (uint32_t)i) lo = mid;
            else hi = mid;
        }
        return fe_load(heavy_vals + L.slot0 + lo);
    }
    // the row's products share Montgomery reductions (fe29.hpp dot29): one per DOT29_GROUP entries
    dot29 d;
    dot29_init(d);
    for (; k < e; k++) dot29_add(d, unpack29<0>(fe_load(interner + L.val[k])), unpack29<5>(fe_load(x + L.idx[k])));
    return dot29_result(d);
}
// one workgroup per chunk of a heavy line: partial[chunk] = sum over its entries
__global__ __launch_bounds__(RED_THREADS) void heavy_dot_kernel(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ val,
                                                                const heavy_chunk* __restrict__ chunks, const fe* __restrict__ interner,
                                                                const fe* __restrict__ x, fe* __restrict__ partials) {
    PK_LATENCY_PRIO();
    __shared__ uint4 smem[16];
    const heavy_chunk c = chunks[blockIdx.x];
    dot29 d;
    dot29_init(d);
    for (uint32_t k = c.begin + threadIdx.x; k < c.end; k += RED_THREADS) dot29_add(d, unpack29<0>(fe_load(interner + val[k])), unpack29<5>(fe_load(x + idx[k])));
    wide w[1] = {wide_zero()};
    wide_add_fe(w[0], dot29_result(d));
    const fe sum = block_reduce_wide<1>(w, smem);
    if (threadIdx.x == 0) fe_store(partials + blockIdx.x, sum);
}
// one workgroup per heavy line: the sum of its chunks' partials
__global__ __launch_bounds__(RED_THREADS) void heavy_sum_kernel(const uint32_t* __restrict__ slot_chunk0, const uint32_t* __restrict__ slot_nchunks,
                                                                uint32_t first_slot, uint32_t first_chunk, const fe* __restrict__ partials,
                                                                fe* __restrict__ heavy_vals) {
    PK_LATENCY_PRIO();
    __shared__ uint4 smem[16];
    const uint32_t slot = first_slot + blockIdx.x, c0 = slot_chunk0[slot] - first_chunk, n = slot_nchunks[slot];
    fe acc = fe_zero();
    for (uint32_t j = threadIdx.x; j < n; j += RED_THREADS) acc = fe_add(acc, fe_load(partials + c0 + j));
    wide w[1] = {wide_zero()};
    wide_add_fe(w[0], acc);
    const fe sum = block_reduce_wide<1>(w, smem);
    if (threadIdx.x == 0) fe_store(heavy_vals + slot, sum);
}

// calculate_witness_bounds (provekit/common/src/utils/sumcheck.rs:181-193): a = A z, b = B z, c = a o b, zero-padded
__global__ __launch_bounds__(256) void witness_bounds_kernel(line_set A, line_set B, const fe* __restrict__ interner, const fe* __restrict__ heavy_vals,
                                                             const fe* __restrict__ z, size_t num_rows, size_t padded, fe* __restrict__ a,
                                                             fe* __restrict__ b, fe* __restrict__ c, size_t stride, size_t offset) {
    PK_LATENCY_PRIO();
    // output j holds row i = j * stride + offset: (1, 0) for the whole table, (G, g) for rank g's share of a sumcheck
    // sharded by the low index bits (the leading variable's fold pairs i with i + len/2, both on one rank)
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= padded) return;
    const size_t i = j * stride + offset;
    fe ra = fe_zero(), rb = fe_zero();
    if (i < num_rows) {
        ra = sparse_row_dot(A, interner, z, heavy_vals, i);
        rb = sparse_row_dot(B, interner, z, heavy_vals, i);
    }
    fe_store(a + j, ra);
    fe_store(b + j, rb);
    fe_store(c + j, fe_mulx(ra, rb));
}

__global__ __launch_bounds__(256) void sparse_gather_kernel(line_set L, const fe* __restrict__ interner, const fe* __restrict__ heavy_vals,
                                                            const fe* __restrict__ x, size_t n_out, fe* __restrict__ y) {
    PK_LATENCY_PRIO();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    fe_store(y + i, sparse_row_dot(L, interner, x, heavy_vals, i));
}

// R1CS::test_witness_satisfaction (provekit/prover/src/r1cs.rs:41-60): (A z)_i (B z)_i == (C z)_i for every row; the first
// failing row (the one the reference's "Constraint {row} failed" names) is reduced with an atomic min.
__global__ __launch_bounds__(256) void satisfaction_kernel(line_set A, line_set B, line_set Cm, const fe* __restrict__ interner,
                                                           const fe* __restrict__ heavy_vals, const fe* __restrict__ z, size_t num_rows,
                                                           unsigned long long* __restrict__ first_bad) {
    PK_LATENCY_PRIO();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_rows) return;
    fe ra = sparse_row_dot(A, interner, z, heavy_vals, i);
    fe rb = sparse_row_dot(B, interner, z, heavy_vals, i);
    fe rc = sparse_row_dot(Cm, interner, z, heavy_vals, i);
    if (!fe_eq(fe_mulx(ra, rb), rc)) atomicMin(first_bad, (unsigned long long)i);
}

// eq^T A, eq^T B, eq^T C in one launch: lane j walks column j of all three matrices.  Column degrees vary (lanes of a
// wavefront wait for the longest column), and the sum of three degrees varies relatively less than each one alone.
struct csc3 {
    line_set m[3];
};
__global__ __launch_bounds__(256) void sparse_gather3_kernel(csc3 m, const fe* __restrict__ interner, const fe* __restrict__ heavy_vals,
                                                             const fe* __restrict__ x, size_t n_out, fe* __restrict__ y, size_t first, size_t last) {
    PK_LATENCY_PRIO();
    size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // columns [first, last) only: a rank's block of the rows
    if (i >= last) return;
#pragma unroll 1
    for (int k = 0; k < 3; k++) fe_store(y + (size_t)k * n_out + i, sparse_row_dot(m.m[k], interner, x, heavy_vals, i));
}

int upload_u32(pk_ctx* ctx, const std::vector<uint32_t>& v, uint32_t** out) {
    PK_HIP(ctx, hipMalloc((void**)out, (v.size() ? v.size() : 1) * 4));
    if (!v.empty()) PK_HIP(ctx, hipMemcpy(*out, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    return PK_OK;
}

line_set set_of(const pk_r1cs* r, int s) {  // s: 0..2 rows of A, B, C; 3..5 their columns
    const int m = s % 3;
    const bool t = s >= 3;
    return line_set{t ? r->csc_ptr[m] : r->csr_ptr[m], t ? r->csc_idx[m] : r->csr_idx[m], t ? r->csc_val[m] : r->csr_val[m],
                    r->d_heavy_lines[s], r->n_heavy[s], r->slot0[s]};
}
// Sums the heavy lines of the line sets in `mask` against x and returns where the gather kernels find them (slot-indexed).  The
// partials and the sums live in the context's workspace: they are consumed by the very next launch on the same stream.
int heavy_prepare(pk_ctx* ctx, const pk_r1cs* r, unsigned mask, const fe* x, const fe** heavy_vals) {
    *heavy_vals = nullptr;
    bool any = false;
    for (int s = 0; s < 6; s++) any |= ((mask >> s) & 1u) && r->n_heavy[s];
    if (!any) return PK_OK;
    int rc = ensure_ws(ctx, 32 * (r->n_slots + r->n_chunks_total));
    if (rc) return rc;
    fe* vals = (fe*)ctx->d_ws;
    fe* partials = vals + r->n_slots;
    for (int s = 0; s < 6; s++) {
        if (!((mask >> s) & 1u) || !r->n_heavy[s]) continue;
        const line_set L = set_of(r, s);
        heavy_dot_kernel<<<r->n_chunks[s], RED_THREADS, 0, ctx->stream>>>(L.idx, L.val, r->d_chunks + r->chunk0[s], r->d_interner, x, partials + r->chunk0[s]);
        heavy_sum_kernel<<<r->n_heavy[s], RED_THREADS, 0, ctx->stream>>>(r->d_slot_chunk0, r->d_slot_nchunks, r->slot0[s], 0, partials, vals);
    }
    PK_LAUNCH_CHECK(ctx);
    *heavy_vals = vals;
    return PK_OK;
}

}  // namespace

extern "C" {

int pk_r1cs_destroy(pk_ctx* ctx, pk_r1cs* r) {
    PK_ENTER(ctx);
    if (!r) return PK_OK;
    (void)wait_ctx(ctx);
    (void)hipFree(r->d_interner);
    for (int m = 0; m < 3; m++) {
        (void)hipFree(r->csr_ptr[m]); (void)hipFree(r->csr_idx[m]); (void)hipFree(r->csr_val[m]);
        (void)hipFree(r->csc_ptr[m]); (void)hipFree(r->csc_idx[m]); (void)hipFree(r->csc_val[m]);
    }
    for (int s = 0; s < 6; s++) (void)hipFree(r->d_heavy_lines[s]);
    (void)hipFree(r->d_chunks);
    (void)hipFree(r->d_slot_chunk0);
    (void)hipFree(r->d_slot_nchunks);
    delete r;
    return PK_OK;
}

int pk_r1cs_create(pk_ctx* ctx, size_t num_constraints, size_t num_witnesses, const pk_sparse_matrix mats[3],
                   const uint64_t* interner, size_t n_interned, pk_r1cs** out) {
    if (!ctx || !out) return PK_ERR_BAD_ARG;
    PK_ENTER(ctx);
    *out = nullptr;
    PK_REQUIRE(ctx, mats && (interner || n_interned == 0), "null pointer");
    PK_REQUIRE(ctx, num_constraints < (1ull << 32) && num_witnesses < (1ull << 32), "dimension above u32");
    pk_r1cs* r = new (std::nothrow) pk_r1cs();
    if (!r) return PK_ERR_OOM;
    r->num_constraints = num_constraints;
    r->num_witnesses = num_witnesses;
    r->n_interned = n_interned;
    int rc = PK_OK;
    auto fail = [&](int code) {
        pk_r1cs_destroy(ctx, r);
        return code;
    };
    if (hipMalloc((void**)&r->d_interner, (n_interned ? n_interned : 1) * 32) != hipSuccess) return fail(set_err(ctx, PK_ERR_OOM, "hipMalloc interner"));
    if (n_interned && hipMemcpy(r->d_interner, interner, n_interned * 32, hipMemcpyHostToDevice) != hipSuccess)
        return fail(set_err(ctx, PK_ERR_HIP, "interner upload"));
    try {  // the host-side index arrays are sized by caller-supplied dimensions: no exception may cross the C ABI
    std::vector<uint32_t> heavy_lines[6], host_ptr[6];
    for (int m = 0; m < 3; m++) {
        const pk_sparse_matrix& M = mats[m];
        const size_t nnz = M.nnz;
        if (nnz >= (1ull << 32)) return fail(set_err(ctx, PK_ERR_BAD_ARG, "nnz above u32"));
        if (nnz && (!M.col_indices || !M.values)) return fail(set_err(ctx, PK_ERR_BAD_ARG, "null matrix arrays"));
        if (num_constraints && !M.new_row_indices) return fail(set_err(ctx, PK_ERR_BAD_ARG, "null new_row_indices"));
        r->nnz[m] = nnz;
        // CSR pointer array with the closing entry (sparse_matrix.rs:113-123 row_range)
        std::vector<uint32_t> rp(num_constraints + 1);
        for (size_t i = 0; i < num_constraints; i++) rp[i] = M.new_row_indices[i];
        rp[num_constraints] = (uint32_t)nnz;
        for (size_t i = 0; i < num_constraints; i++)
            if (rp[i] > rp[i + 1] || rp[i + 1] > nnz) return fail(set_err(ctx, PK_ERR_BAD_ARG, "new_row_indices not monotone"));
        std::vector<uint32_t> ci(M.col_indices, M.col_indices + nnz), vv(M.values, M.values + nnz);
        for (size_t k = 0; k < nnz; k++) {
            if (ci[k] >= num_witnesses) return fail(set_err(ctx, PK_ERR_BAD_ARG, "column index out of bounds"));
            if (vv[k] >= n_interned) return fail(set_err(ctx, PK_ERR_BAD_ARG, "Value not in interner."));  // sparse_matrix.rs:133
        }
        // CSC by counting sort (stable: rows ascending within a column)
        std::vector<uint32_t> cp(num_witnesses + 1, 0), ri(nnz), cv(nnz);
        for (size_t k = 0; k < nnz; k++) cp[ci[k] + 1]++;
        for (size_t j = 0; j < num_witnesses; j++) cp[j + 1] += cp[j];
        std::vector<uint32_t> cur(cp.begin(), cp.end() - 1);
        for (size_t i = 0; i < num_constraints; i++)
            for (uint32_t k = rp[i]; k < rp[i + 1]; k++) {
                uint32_t pos = cur[ci[k]]++;
                ri[pos] = (uint32_t)i;
                cv[pos] = vv[k];
            }
        for (size_t i = 0; i < num_constraints; i++)
            if (rp[i + 1] - rp[i] > HEAVY_DEGREE) heavy_lines[m].push_back((uint32_t)i);
        for (size_t j = 0; j < num_witnesses; j++)
            if (cp[j + 1] - cp[j] > HEAVY_DEGREE) heavy_lines[3 + m].push_back((uint32_t)j);
        host_ptr[m] = rp;
        host_ptr[3 + m] = cp;
        if ((rc = upload_u32(ctx, rp, &r->csr_ptr[m])) || (rc = upload_u32(ctx, ci, &r->csr_idx[m])) || (rc = upload_u32(ctx, vv, &r->csr_val[m])) ||
            (rc = upload_u32(ctx, cp, &r->csc_ptr[m])) || (rc = upload_u32(ctx, ri, &r->csc_idx[m])) || (rc = upload_u32(ctx, cv, &r->csc_val[m])))
            return fail(rc);
    }
    // heavy lines -> chunks (sets in order, lines ascending within a set, a line's chunks consecutive)
    std::vector<heavy_chunk> chunks;
    std::vector<uint32_t> slot_chunk0, slot_nchunks;
    for (int s = 0; s < 6; s++) {
        r->n_heavy[s] = (uint32_t)heavy_lines[s].size();
        r->slot0[s] = (uint32_t)slot_chunk0.size();
        r->chunk0[s] = (uint32_t)chunks.size();
        for (uint32_t line : heavy_lines[s]) {
            const uint32_t b = host_ptr[s][line], e = host_ptr[s][line + 1], slot = (uint32_t)slot_chunk0.size();
            slot_chunk0.push_back((uint32_t)chunks.size());
            for (uint32_t k = b; k < e; k += HEAVY_CHUNK) chunks.push_back(heavy_chunk{slot, k, std::min(e, k + HEAVY_CHUNK)});
            slot_nchunks.push_back((uint32_t)chunks.size() - slot_chunk0.back());
        }
        r->n_chunks[s] = (uint32_t)chunks.size() - r->chunk0[s];
        if (r->n_heavy[s] && (rc = upload_u32(ctx, heavy_lines[s], &r->d_heavy_lines[s]))) return fail(rc);
    }
    r->n_slots = slot_chunk0.size();
    r->n_chunks_total = chunks.size();
    if (r->n_slots) {
        if ((rc = upload_u32(ctx, slot_chunk0, &r->d_slot_chunk0)) || (rc = upload_u32(ctx, slot_nchunks, &r->d_slot_nchunks))) return fail(rc);
        if (hipMalloc((void**)&r->d_chunks, chunks.size() * sizeof(heavy_chunk)) != hipSuccess) return fail(set_err(ctx, PK_ERR_OOM, "hipMalloc heavy chunks"));
        if (hipMemcpy(r->d_chunks, chunks.data(), chunks.size() * sizeof(heavy_chunk), hipMemcpyHostToDevice) != hipSuccess)
            return fail(set_err(ctx, PK_ERR_HIP, "heavy chunk upload"));
    }
    } catch (const std::bad_alloc&) {
        return fail(set_err(ctx, PK_ERR_OOM, "host memory exhausted while indexing the R1CS (%zu x %zu)", num_constraints, num_witnesses));
    }
    *out = r;
    return PK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// R1CS from its postcard bytes.  The reference's SparseMatrix keeps new_row_indices / col_indices / values private
// (provekit/common/src/sparse_matrix.rs:12-27), so a Rust caller cannot hand their pointers to pk_r1cs_create; what it can
// do is serialise `&R1CS` with the serde impls the reference derives -- postcard, the encoding of its own .nps files
// (provekit/common/src/file/bin.rs:22-71) -- and pass the bytes.  Layout (postcard: usize / u32 = LEB128 varint, Vec<T> =
// varint length + items, bytes = varint length + raw):
//   R1CS { num_public_inputs: usize, interner: Interner, a, b, c: SparseMatrix }                    (r1cs.rs:8-14)
//   Interner { values: serde_ark(Vec<FieldElement>) } = bytes( u64-LE count | count x 32-byte canonical LE )   (interner.rs:6-10,
//                                                                               utils/serde_ark.rs:11-31, ark-serialize compressed)
//   SparseMatrix { num_rows, num_cols: usize, new_row_indices: Vec<u32>, col_indices: Vec<u32>, values: Vec<usize> }
namespace {
struct PcReader {
    const uint8_t* p;
    size_t n, off = 0;
    bool ok = true;
    uint64_t varint() {
        uint64_t v = 0;
        for (unsigned shift = 0; shift < 70; shift += 7) {
            if (off >= n) return ok = false, 0;
            const uint8_t b = p[off++];
            if (shift == 63 && b > 1) return ok = false, 0;
            v |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80)) return v;
        }
        return ok = false, 0;
    }
    bool vec_u32(std::vector<uint32_t>& out) {
        const uint64_t len = varint();
        if (!ok || len > n - off) return ok = false;  // every item takes at least one byte
        out.resize(len);
        for (uint64_t i = 0; i < len; i++) {
            const uint64_t v = varint();
            if (!ok || v > 0xffffffffull) return ok = false;
            out[i] = (uint32_t)v;
        }
        return true;
    }
};
}  // namespace

static int r1cs_from_postcard_impl(pk_ctx* ctx, const uint8_t* bytes, size_t len, pk_r1cs** out, size_t* num_constraints,
                                   size_t* num_witnesses, size_t* num_public_inputs, size_t* consumed);
int pk_r1cs_from_postcard(pk_ctx* ctx, const uint8_t* bytes, size_t len, pk_r1cs** out, size_t* num_constraints, size_t* num_witnesses,
                          size_t* num_public_inputs, size_t* consumed) {
    try {  // untrusted bytes size host vectors: an allocation failure is PK_ERR_OOM, never an exception through extern "C"
        return r1cs_from_postcard_impl(ctx, bytes, len, out, num_constraints, num_witnesses, num_public_inputs, consumed);
    } catch (const std::bad_alloc&) {
        if (out) *out = nullptr;
        return set_err(ctx, PK_ERR_OOM, "host memory exhausted while decoding the postcard R1CS");
    }
}
static int r1cs_from_postcard_impl(pk_ctx* ctx, const uint8_t* bytes, size_t len, pk_r1cs** out, size_t* num_constraints,
                                   size_t* num_witnesses, size_t* num_public_inputs, size_t* consumed) {
    if (!ctx || !out) return PK_ERR_BAD_ARG;
    PK_ENTER(ctx);
    *out = nullptr;
    PK_REQUIRE(ctx, bytes, "null pointer");
    PcReader rd{bytes, len};
    const uint64_t n_pub = rd.varint();
    // interner: bytes( u64 count | count * 32 )
    const uint64_t blob = rd.varint();
    PK_REQUIRE(ctx, rd.ok && blob >= 8 && blob <= len - rd.off, "postcard R1CS: truncated interner");
    uint64_t count = 0;
    memcpy(&count, bytes + rd.off, 8);
    PK_REQUIRE(ctx, count <= (blob - 8) / 32 && blob == 8 + 32 * count, "postcard R1CS: interner length mismatch (\"trailing bytes\")");
    std::vector<uint64_t> interner(4 * (size_t)count);
    {  // canonical -> Montgomery on the host (Fp::deserialize_compressed rejects values >= p)
        const uint8_t* q = bytes + rd.off + 8;
        for (size_t i = 0; i < count; i++) {
            fe c;
            memcpy(c.v, q + 32 * i, 32);
            fe red = fe_reduce_any(c);
            PK_REQUIRE(ctx, memcmp(red.v, c.v, 32) == 0, "postcard R1CS: interned value is not a canonical field element");
            fe m = fe_to_montx(c);
            memcpy(&interner[4 * i], m.v, 32);
        }
    }
    rd.off += blob;
    std::vector<uint32_t> nri[3], ci[3], vv[3];
    uint64_t rows[3], cols[3];
    for (int m = 0; m < 3; m++) {
        rows[m] = rd.varint();
        cols[m] = rd.varint();
        PK_REQUIRE(ctx, rd.ok && rd.vec_u32(nri[m]) && rd.vec_u32(ci[m]) && rd.vec_u32(vv[m]), "postcard R1CS: truncated or malformed matrix");
        PK_REQUIRE(ctx, nri[m].size() == rows[m], "postcard R1CS: new_row_indices does not have one entry per row");
        PK_REQUIRE(ctx, vv[m].size() == ci[m].size(), "postcard R1CS: values and col_indices differ in length");
        PK_REQUIRE(ctx, rows[m] == rows[0] && cols[m] == cols[0], "postcard R1CS: A, B, C differ in shape");
        // num_cols comes from untrusted bytes and nothing else in the input bounds it: cap it at the scheme capacity
        // (pk_scheme_create: m <= 27, witnesses <= 2^(m-1)) before pk_r1cs_create sizes its column arrays from it
        PK_REQUIRE(ctx, cols[m] <= ((uint64_t)1 << 27), "postcard R1CS: more than 2^27 columns");
    }
    pk_sparse_matrix mats[3];
    for (int m = 0; m < 3; m++) mats[m] = pk_sparse_matrix{nri[m].data(), ci[m].data(), vv[m].data(), ci[m].size()};
    int rc = pk_r1cs_create(ctx, (size_t)rows[0], (size_t)cols[0], mats, interner.data(), (size_t)count, out);
    if (rc) return rc;
    if (num_constraints) *num_constraints = (size_t)rows[0];
    if (num_witnesses) *num_witnesses = (size_t)cols[0];
    if (num_public_inputs) *num_public_inputs = (size_t)n_pub;
    if (consumed) *consumed = rd.off;
    return PK_OK;
}

int pk_r1cs_witness_bounds(pk_ctx* ctx, const pk_r1cs* r, const uint64_t* d_z, unsigned m0, uint64_t* d_a, uint64_t* d_b,
                           uint64_t* d_c) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, r && d_z && d_a && d_b && d_c, "null pointer");
    PK_REQUIRE(ctx, m0 <= 30 && r->num_constraints <= ((size_t)1 << m0), "R1CS constraints exceed scheme capacity");  // whir_r1cs.rs:52-54
    return witness_bounds_strided(ctx, r, d_z, m0, 1, 0, d_a, d_b, d_c);
}

int pk_r1cs_matvec(pk_ctx* ctx, const pk_r1cs* r, int matrix, int transpose, const uint64_t* d_x, uint64_t* d_y) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, r && d_x && d_y, "null pointer");
    PK_REQUIRE(ctx, matrix >= 0 && matrix < 3, "matrix must be 0 (A), 1 (B) or 2 (C)");
    size_t n_out = transpose ? r->num_witnesses : r->num_constraints;
    if (!n_out) return PK_OK;
    const int s = (transpose ? 3 : 0) + matrix;
    ProfScope prof(ctx, "sparse_matvec");
    const fe* hv = nullptr;
    int rc = heavy_prepare(ctx, r, 1u << s, (const fe*)d_x, &hv);
    if (rc) return rc;
    sparse_gather_kernel<<<(unsigned)((n_out + 255) / 256), 256, 0, ctx->stream>>>(set_of(r, s), r->d_interner, hv, (const fe*)d_x, n_out, (fe*)d_y);
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}

int pk_r1cs_test_witness_satisfaction(pk_ctx* ctx, const pk_r1cs* r, const uint64_t* d_witness, size_t n_witness, int64_t* first_failed_row) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, r && first_failed_row, "null pointer");
    PK_REQUIRE(ctx, n_witness == r->num_witnesses, "Witness size does not match");  // r1cs.rs:42-45
    *first_failed_row = -1;
    if (!r->num_constraints) return PK_OK;
    PK_REQUIRE(ctx, d_witness, "null pointer");
    int rc = ensure_scratch(ctx, 8);
    if (rc) return rc;
    unsigned long long* d_bad = (unsigned long long*)ctx->d_scratch;
    PK_HIP(ctx, hipMemsetAsync(d_bad, 0xff, 8, ctx->stream));
    {
        ProfScope prof(ctx, "r1cs_satisfaction");
        const fe* hv = nullptr;
        if ((rc = heavy_prepare(ctx, r, 7u, (const fe*)d_witness, &hv))) return rc;
        satisfaction_kernel<<<(unsigned)((r->num_constraints + 255) / 256), 256, 0, ctx->stream>>>(set_of(r, 0), set_of(r, 1), set_of(r, 2), r->d_interner, hv,
                                                                                                 (const fe*)d_witness, r->num_constraints, d_bad);
        PK_LAUNCH_CHECK(ctx);
    }
    unsigned long long bad = 0;
    PK_HIP(ctx, hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    PK_WAIT(ctx);
    if (bad != ~0ull) {
        *first_failed_row = (int64_t)bad;
        return set_err(ctx, PK_ERR_UNSATISFIED, "Constraint %llu failed", bad);  // r1cs.rs:57
    }
    return PK_OK;
}

// calculate_external_row_of_r1cs_matrices (sumcheck.rs:207-218): [eq^T A, eq^T B, eq^T C], each num_witnesses long
int pk_r1cs_external_row(pk_ctx* ctx, const pk_r1cs* r, const uint64_t* d_eq_alpha, uint64_t* d_out) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, r && d_eq_alpha && d_out, "null pointer");
    return external_row_range(ctx, r, d_eq_alpha, 0, r->num_witnesses, d_out);
}

}  // extern "C"

namespace pk {
// rank `offset` of `stride` ranks: a, b, c for the rows i = j * stride + offset, j < 2^m0 / stride (prover.hip, sharded sumcheck)
int witness_bounds_strided(pk_ctx* ctx, const pk_r1cs* r, const uint64_t* d_z, unsigned m0, unsigned stride, unsigned offset, uint64_t* d_a,
                           uint64_t* d_b, uint64_t* d_c) {
    PK_REQUIRE(ctx, r && d_z && d_a && d_b && d_c, "null pointer");
    PK_REQUIRE(ctx, m0 <= 30 && r->num_constraints <= ((size_t)1 << m0) && stride && offset < stride, "bad shard of the witness bounds");
    const size_t padded = ((size_t)1 << m0) / stride;
    ProfScope prof(ctx, "witness_bounds");
    const fe* hv = nullptr;  // heavy rows are summed whole on every rank of a sharded sumcheck (they are few)
    int rc = heavy_prepare(ctx, r, 3u, (const fe*)d_z, &hv);
    if (rc) return rc;
    witness_bounds_kernel<<<(unsigned)((padded + 255) / 256), 256, 0, ctx->stream>>>(set_of(r, 0), set_of(r, 1), r->d_interner, hv, (const fe*)d_z,
                                                                                     r->num_constraints, padded, (fe*)d_a, (fe*)d_b, (fe*)d_c, stride, offset);
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}
// columns [first, last) of the three external rows, written at their absolute positions of d_out (3 x num_witnesses)
int external_row_range(pk_ctx* ctx, const pk_r1cs* r, const uint64_t* d_eq_alpha, size_t first, size_t last, uint64_t* d_out) {
    PK_REQUIRE(ctx, r && d_eq_alpha && d_out, "null pointer");
    if (last > r->num_witnesses) last = r->num_witnesses;
    if (first >= last) return PK_OK;
    csc3 m;
    for (int k = 0; k < 3; k++) m.m[k] = set_of(r, 3 + k);
    ProfScope prof(ctx, "sparse_matvec");
    const fe* hv = nullptr;
    int rc = heavy_prepare(ctx, r, 7u << 3, (const fe*)d_eq_alpha, &hv);
    if (rc) return rc;
    sparse_gather3_kernel<<<(unsigned)((last - first + 255) / 256), 256, 0, ctx->stream>>>(m, r->d_interner, hv, (const fe*)d_eq_alpha, r->num_witnesses,
                                                                                         (fe*)d_out, first, last);
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}
}  // namespace pk
