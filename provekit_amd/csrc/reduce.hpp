// reduce.hpp -- block / grid reduction of field-element sums (used by sumcheck, dot, Horner).
#pragma once
#include <atomic>

#include "ctx.hpp"
#include "fe.hpp"

namespace pk {

constexpr int RED_THREADS = 256;
constexpr int PK_FLAG_WORD = 256;  // u32 index of the completion flag inside the pinned result page (byte 1024)
constexpr int RED_MAX_BLOCKS = 1024;

// Sum K field elements per thread across the block; result valid in thread 0.
// smem must hold K * RED_THREADS fe (as 2 uint4 each).
template <int K>
__device__ __forceinline__ void block_reduce_fe(fe (&acc)[K], uint4* smem) {
    const unsigned tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < K; k++) {
        smem[(k * RED_THREADS + tid) * 2] = make_uint4(acc[k].v[0], acc[k].v[1], acc[k].v[2], acc[k].v[3]);
        smem[(k * RED_THREADS + tid) * 2 + 1] = make_uint4(acc[k].v[4], acc[k].v[5], acc[k].v[6], acc[k].v[7]);
    }
    __syncthreads();
    for (unsigned s = RED_THREADS / 2; s >= 1; s >>= 1) {
        if (tid < s) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                uint4 l = smem[(k * RED_THREADS + tid + s) * 2], h = smem[(k * RED_THREADS + tid + s) * 2 + 1];
                fe o;
                o.v[0] = l.x; o.v[1] = l.y; o.v[2] = l.z; o.v[3] = l.w;
                o.v[4] = h.x; o.v[5] = h.y; o.v[6] = h.z; o.v[7] = h.w;
                acc[k] = fe_add(acc[k], o);
                smem[(k * RED_THREADS + tid) * 2] = make_uint4(acc[k].v[0], acc[k].v[1], acc[k].v[2], acc[k].v[3]);
                smem[(k * RED_THREADS + tid) * 2 + 1] = make_uint4(acc[k].v[4], acc[k].v[5], acc[k].v[6], acc[k].v[7]);
            }
        }
        __syncthreads();
    }
}

// Single-launch grid reduction: every block stores its K partial sums, takes a ticket, and the block that
// draws the last ticket sums all partials and writes the K results (agent-scope release/acquire around the
// ticket, MI355X_MICROARCH.md "Workgroup dispatch ... inter-workgroup visibility").  `result` may point to
// device-visible pinned host memory, so the host needs no copy after the stream sync.
// write-through (sc1) store / L1-bypassing load of one field element: the cross-workgroup hand-off uses these plus
// `s_waitcnt vmcnt(0)` before the ticket, i.e. the microarch guide's "sc1 payload -> drained -> flag" form.  No
// agent/system release fence: those write back the whole L2, which holds megabytes of freshly folded sumcheck data.
__device__ __forceinline__ void fe_store_sc1(fe* p, const fe& x) {
    unsigned* q = reinterpret_cast<unsigned*>(p);
#pragma unroll
    for (int i = 0; i < 8; i++) __hip_atomic_store(q + i, x.v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ fe fe_load_sc1(const fe* p) {
    fe v;
    const unsigned* q = reinterpret_cast<const unsigned*>(p);
#pragma unroll
    for (int i = 0; i < 8; i++) v.v[i] = __hip_atomic_load(q + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}
template <int K>
__device__ __forceinline__ void grid_finish_fe(fe (&acc)[K], uint4* smem, fe* __restrict__ partials, unsigned* __restrict__ ticket,
                                               fe* __restrict__ result, unsigned seq = 0) {
    __shared__ unsigned s_last;
    block_reduce_fe<K>(acc, smem);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) fe_store_sc1(partials + (size_t)blockIdx.x * K + k, acc[k]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // partials are out before the ticket is drawn
        unsigned t = atomicAdd(ticket, 1u);
        s_last = (t == gridDim.x - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
#pragma unroll
    for (int k = 0; k < K; k++) acc[k] = fe_zero();
    for (unsigned b = threadIdx.x; b < gridDim.x; b += RED_THREADS) {
#pragma unroll
        for (int k = 0; k < K; k++) acc[k] = fe_add(acc[k], fe_load_sc1(partials + (size_t)b * K + k));
    }
    __syncthreads();
    block_reduce_fe<K>(acc, smem);
    if (threadIdx.x == 0) {
        unsigned* out = reinterpret_cast<unsigned*>(result);  // pinned, fine-grained host memory
#pragma unroll
        for (int k = 0; k < K; k++)
#pragma unroll
            for (int i = 0; i < 8; i++) __hip_atomic_store(out + 8 * k + i, acc[k].v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm for the next launch
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // publish: the host may spin on this word instead of paying a stream synchronisation
        __hip_atomic_store(out + PK_FLAG_WORD, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// second stage: sum `nblocks` partial K-vectors (layout partials[block*K + k]) into out[k]
template <int K>
__global__ __launch_bounds__(RED_THREADS) void reduce_partials_kernel(const fe* __restrict__ partials, unsigned nblocks,
                                                                      fe* __restrict__ out) {
    __shared__ uint4 smem[K * RED_THREADS * 2];
    fe acc[K];
#pragma unroll
    for (int k = 0; k < K; k++) acc[k] = fe_zero();
    for (unsigned b = threadIdx.x; b < nblocks; b += RED_THREADS) {
#pragma unroll
        for (int k = 0; k < K; k++) acc[k] = fe_add(acc[k], fe_load(partials + (size_t)b * K + k));
    }
    block_reduce_fe<K>(acc, smem);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) fe_store(out + k, acc[k]);
    }
}

// host helper: finish a K-vector reduction whose per-block partials sit at ctx->d_scratch[0 .. nblocks*K)
// and copy the K results to `host_out` (synchronises the stream).
template <int K>
inline int finish_reduction(pk_ctx* ctx, unsigned nblocks, uint64_t* host_out) {
    fe* partials = (fe*)ctx->d_scratch;
    fe* result = partials + (size_t)RED_MAX_BLOCKS * 8;
    reduce_partials_kernel<K><<<1, RED_THREADS, 0, ctx->stream>>>(partials, nblocks, result);
    PK_LAUNCH_CHECK(ctx);
    PK_HIP(ctx, hipMemcpyAsync(host_out, result, 32 * K, hipMemcpyDeviceToHost, ctx->stream));
    PK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PK_OK;
}

inline int reduction_scratch(pk_ctx* ctx) {
    int rc = ensure_scratch(ctx, (size_t)RED_MAX_BLOCKS * 8 * 32 + 8 * 32 + 4096);
    if (rc) return rc;
    if (!ctx->h_pinned) {  // device-visible host memory for the few field elements each round returns
        PK_HIP(ctx, hipHostMalloc(&ctx->h_pinned, 4096, hipHostMallocMapped));
        ctx->pinned_bytes = 4096;
        PK_HIP(ctx, hipMemsetAsync((char*)ctx->d_scratch + (size_t)RED_MAX_BLOCKS * 8 * 32 + 8 * 32, 0, 64, ctx->stream));
    }
    return PK_OK;
}
inline fe* red_partials(pk_ctx* ctx) { return (fe*)ctx->d_scratch; }
inline unsigned* red_ticket(pk_ctx* ctx) { return (unsigned*)((char*)ctx->d_scratch + (size_t)RED_MAX_BLOCKS * 8 * 32 + 8 * 32); }
inline fe* red_result(pk_ctx* ctx) { return (fe*)ctx->h_pinned; }
// next sequence number for a reduction launch on this context (never 0)
inline unsigned next_seq(pk_ctx* ctx) {
    ctx->red_seq++;
    if (ctx->red_seq == 0) ctx->red_seq = 1;
    return ctx->red_seq;
}
// Wait for the kernel and hand the K results -- already in pinned host memory -- to the caller.  (Spinning on the
// completion word the finishing block publishes was measured: no gain over hipStreamSynchronize single-stream and a
// loss with several provers per GPU, so the plain synchronisation is kept; the word stays for diagnostics.)
template <int K>
inline int collect_reduction(pk_ctx* ctx, uint64_t* host_out) {
    PK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(host_out, ctx->h_pinned, 32 * K);
    return PK_OK;
}

inline unsigned reduction_blocks(const pk_ctx* ctx, size_t work_items) {
    size_t need = (work_items + RED_THREADS - 1) / RED_THREADS;
    size_t cap = (size_t)ctx->num_cus * 4;
    if (cap > RED_MAX_BLOCKS) cap = RED_MAX_BLOCKS;
    if (need < 1) need = 1;
    return (unsigned)(need < cap ? need : cap);
}

}  // namespace pk
