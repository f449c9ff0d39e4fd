// reduce.hpp -- block / grid reduction of field-element sums (used by sumcheck, dot, Horner).
#pragma once
#include <chrono>
#include <atomic>

#include "ctx.hpp"
#include "fe.hpp"

namespace pk {

constexpr int RED_THREADS = 256;
constexpr int PK_FLAG_WORD = 256;  // u32 index of the completion flag inside the pinned result page (byte 1024)
constexpr int RED_MAX_BLOCKS = 1024;

// ---- wide sums ----------------------------------------------------------------------------------------------------
// A sum of up to 1024 field elements is carried as 8 independent u64 limb sums (each < 2^42): adding two of them is 8
// plain 64-bit adds, no carry chain and no conditional subtraction, so the 6 shuffle steps of a wavefront reduction cost
// a quarter of what fe_add-based steps did.  One modular reduction at the very end.
struct wide {
    u64 l[8];
};
__device__ __forceinline__ wide wide_zero() {
    wide w;
#pragma unroll
    for (int i = 0; i < 8; i++) w.l[i] = 0;
    return w;
}
__device__ __forceinline__ void wide_add_fe(wide& w, const fe& x) {
#pragma unroll
    for (int i = 0; i < 8; i++) w.l[i] += x.v[i];
}
// limb sums of at most 1024 values < p  ->  the sum mod p, fully reduced
__host__ __device__ __forceinline__ fe wide_reduce(const wide& w) {
    u32 a[9];
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += w.l[i];
        a[i] = (u32)c;
        c >>= 32;
    }
    a[8] = (u32)c;  // total < 2^10 p < 2^264
    // quotient estimate from the top 40 bits: 2^256 / p = 5.2901... > 1354 / 256, so q never overshoots and is at most 2 short
    const u64 hi = ((u64)a[8] << 32) | a[7];
    const u32 q = (u32)((hi * 1354u) >> 40);
    u64 mc = 0;
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        mc += (u64)q * (i < 8 ? kPlimb(i) : 0u);
        a[i] = __builtin_subc(a[i], (u32)mc, borrow, &borrow);
        mc >>= 32;
    }
#pragma unroll
    for (int rep = 0; rep < 3; rep++) {  // remainder < 3p + p
        u32 d[9];
        u32 b = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) d[i] = __builtin_subc(a[i], i < 8 ? kPlimb(i) : 0u, b, &b);
#pragma unroll
        for (int i = 0; i < 9; i++) a[i] = b ? a[i] : d[i];
    }
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = a[i];
    return r;
}

__device__ __forceinline__ u64 shfl_down_u64(u64 x, unsigned off) {
    u32 lo = __shfl_down((u32)x, off, 64), hi = __shfl_down((u32)(x >> 32), off, 64);
    return ((u64)hi << 32) | lo;
}

// Sum K wide values per thread across the 256-thread block.  Returns the reduced sum number k in thread k (k < K);
// other threads return garbage.  smem: at least 4*K*8 u64 (the callers' K*256 fe buffer is far larger).
template <int K>
__device__ __forceinline__ fe block_reduce_wide(wide (&w)[K], uint4* smem) {
#pragma unroll
    for (unsigned off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int k = 0; k < K; k++)
#pragma unroll
            for (int i = 0; i < 8; i++) w[k].l[i] += shfl_down_u64(w[k].l[i], off);
    }
    u64* sm = reinterpret_cast<u64*>(smem);
    const unsigned tid = threadIdx.x, wave = tid >> 6;
    if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < K; k++)
#pragma unroll
            for (int i = 0; i < 8; i++) sm[(k * 4 + wave) * 8 + i] = w[k].l[i];
    }
    __syncthreads();
    wide t = wide_zero();
    if (tid < K) {
#pragma unroll
        for (int v = 0; v < RED_THREADS / 64; v++)
#pragma unroll
            for (int i = 0; i < 8; i++) t.l[i] += sm[(tid * 4 + v) * 8 + i];
    }
    return wide_reduce(t);
}

// Single-launch grid reduction: every block stores its K partial sums (write-through, sc1), drains them
// (`s_waitcnt vmcnt(0)`), takes a ticket, and the block that draws the last ticket sums all partials and writes the K
// results to `result` -- device-visible pinned host memory, so the host needs no copy after the stream sync.  This is the
// microarch guide's "sc1 payload -> drained -> flag" hand-off; no agent/system release fence, which would write back the
// whole L2 holding megabytes of freshly folded sumcheck data.
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
// Sum K field elements per thread across the 256-thread block: thread k (k < K) returns sum number k, fully reduced; the others
// return garbage.  Column sums through LDS instead of wavefront shuffles: every thread parks its 8K words in a [word][thread]
// table (rows padded by one: conflict-free both ways), then 8K x PARTS lanes of the FIRST wavefront each add up one word over
// 256/PARTS threads (a sum of 256 words fits 40 bits) and thread k collects its 8 limb sums.  The block's other three wavefronts
// are done after one store per word: ~600 wave-instructions per block for K = 3 where the shuffle tree of 64-bit limb sums
// (block_reduce_wide, still used by the one-sum kernels) took ~4,800.
template <int K>
__device__ __forceinline__ fe block_reduce_fe(const fe (&acc)[K]) {
    constexpr int W = 8 * K;
    constexpr int PARTS = W <= 8 ? 8 : (W <= 16 ? 4 : (W <= 32 ? 2 : 1));
    constexpr int LEN = RED_THREADS / PARTS;
    static_assert(W * PARTS <= 64, "one wavefront sums the columns");
    __shared__ u32 cols[W][RED_THREADS + 1];
    __shared__ u64 part_sums[PARTS][W];
    const unsigned tid = threadIdx.x;
    __syncthreads();  // the tables of an earlier call in this kernel are no longer being read
#pragma unroll
    for (int k = 0; k < K; k++)
#pragma unroll
        for (int i = 0; i < 8; i++) cols[8 * k + i][tid] = acc[k].v[i];
    __syncthreads();
    if (tid < W * PARTS) {
        const unsigned w = tid % W, part = tid / W;
        const u32* col = &cols[w][part * LEN];
        u64 s0 = 0, s1 = 0, s2 = 0, s3 = 0;  // four independent chains: the LDS reads of one hide behind the adds of the others
#pragma unroll 4
        for (int t = 0; t < LEN; t += 4) {
            s0 += col[t];
            s1 += col[t + 1];
            s2 += col[t + 2];
            s3 += col[t + 3];
        }
        part_sums[part][w] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    wide t = wide_zero();
    if (tid < K) {
#pragma unroll
        for (int p = 0; p < PARTS; p++)
#pragma unroll
            for (int i = 0; i < 8; i++) t.l[i] += part_sums[p][8 * tid + i];
    }
    return wide_reduce(t);  // limb sums of 256 values < p
}

template <int K>
__device__ __forceinline__ void grid_finish_fe(fe (&acc)[K], uint4* smem, fe* __restrict__ partials, unsigned* __restrict__ ticket,
                                               fe* __restrict__ result, unsigned seq = 0) {
    static_assert(K <= 64, "the K results live in the first wavefront");
    __shared__ unsigned s_last;
    (void)smem;
    fe mine = block_reduce_fe<K>(acc);  // thread k holds sum k
    const unsigned tid = threadIdx.x;
    if (gridDim.x == 1) {  // a single workgroup (the late, tiny rounds): its sums are the results -- no ticket, no second pass
        if (tid < 64) {
            unsigned* out = reinterpret_cast<unsigned*>(result);
            if (tid < K) {
#pragma unroll
                for (int i = 0; i < 8; i++) __hip_atomic_store(out + 8 * tid + i, mine.v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0) __hip_atomic_store(out + PK_FLAG_WORD, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    if (tid < 64) {                           // the first wavefront: lanes < K store, then lane 0 draws the ticket
        if (tid < K) fe_store_sc1(partials + (size_t)blockIdx.x * K + tid, mine);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's partials are out before the ticket is drawn
        if (tid == 0) {
            unsigned t = atomicAdd(ticket, 1u);
            s_last = (t == gridDim.x - 1) ? 1u : 0u;
        }
    }
    __syncthreads();
    if (!s_last) return;
    fe tot[K];
#pragma unroll
    for (int k = 0; k < K; k++) tot[k] = fe_zero();
    for (unsigned b = tid; b < gridDim.x; b += RED_THREADS) {  // gridDim.x <= RED_MAX_BLOCKS: at most 4 per thread
#pragma unroll
        for (int k = 0; k < K; k++) tot[k] = fe_add(tot[k], fe_load_sc1(partials + (size_t)b * K + k));
    }
    mine = block_reduce_fe<K>(tot);
    if (tid < 64) {
        unsigned* out = reinterpret_cast<unsigned*>(result);  // pinned, fine-grained host memory
        if (tid < K) {
#pragma unroll
            for (int i = 0; i < 8; i++) __hip_atomic_store(out + 8 * tid + i, mine.v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm for the next launch
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // publish: the host may spin on this word instead of paying a stream synchronisation
        if (tid == 0) __hip_atomic_store(out + PK_FLAG_WORD, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

inline int reduction_scratch(pk_ctx* ctx) {
    int rc = ensure_scratch(ctx, (size_t)RED_MAX_BLOCKS * 8 * 32 + 8 * 32 + 4096);
    if (rc) return rc;
    if (!ctx->h_pinned) {  // device-visible host memory for the few field elements each round returns
        rc = ensure_pinned(ctx);
        if (rc) return rc;
    }
    if (!ctx->red_armed) {  // [ticket | ... | 64: the gate's device mirror (seq, challenge)]
        PK_HIP(ctx, hipMemsetAsync((char*)ctx->d_scratch + (size_t)RED_MAX_BLOCKS * 8 * 32 + 8 * 32, 0, 128, ctx->stream));
        ctx->red_armed = true;
    }
    return PK_OK;
}
inline fe* red_partials(pk_ctx* ctx) { return (fe*)ctx->d_scratch; }
inline unsigned* red_ticket(pk_ctx* ctx) { return (unsigned*)((char*)ctx->d_scratch + (size_t)RED_MAX_BLOCKS * 8 * 32 + 8 * 32); }
inline fe* red_result(pk_ctx* ctx) { return ctx->red_across ? (fe*)ctx->d_xred : (fe*)ctx->h_pinned; }
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
    if (ctx->red_across) return comm_collect_fe(ctx, K, host_out);  // partial sums of a sharded operand: sum over the ranks first
    int rc = sync_stream(ctx);
    if (rc) return rc;
    memcpy(host_out, ctx->h_pinned, 32 * K);
    return PK_OK;
}

// ---- latency mode: a kernel that takes its challenge from a GATE instead of its arguments ------------------------------------
// The host enqueues round k+1's kernel before it has round k's result.  Workgroup (0,0) of that kernel polls a word of the pinned
// page (system scope, over the host link) until the host has published the challenge with the expected sequence number, copies
// the 32 bytes into a device-side mirror and releases the other workgroups, which poll the mirror (agent scope, L2).  The spin is
// bounded: a host that went away costs seconds, never a hung queue.  (Workgroups are dispatched in index order, so (0,0) is
// resident whenever any workgroup of the grid is.)
struct gate_args {
    const unsigned* host;  // pinned page + PK_PIN_GATE, or nullptr: no gate, the challenge is in the kernel arguments
    unsigned* dev;         // device mirror, same layout
    unsigned seq;
    unsigned leader_spins;  // polls of the pinned page before workgroup (0,0) gives up (PK_GATE_SPINS_LEADER; the test-suite shortens it)
};
// A gate that gives up (the host did not publish within the device-side bound: stopped in a debugger, SIGSTOP, a core shared with
// too many provers) lets its kernel run on with a ZERO challenge so that the queue drains -- and says so: workgroup (0,0) stores the
// gate's sequence number into the word PK_PIN_GATE_TIMEOUT of the pinned page (system scope, sticky until the host clears it).
// The host looks at that word wherever it takes a gated kernel's result (collect_reduction_spin) and once more before pk_prove
// returns (gate_timed_out: the closing fold has no result of its own), and abandons the proof with PK_ERR_HIP.  The device bound
// (2^24 polls of >= 1 us) is well above the host's own 10 s tolerance, so the host normally notices first and releases the gate itself.
#define PK_GATE_SPINS_LEADER (1u << 24)
#define PK_GATE_SPINS_FOLLOWER (1u << 27)
// Layout of the gate (48 bytes, 64-byte aligned, the same in the pinned page and in the device mirror): three 16-byte chunks
// [c0 c1 c2 seq] [c3 c4 c5 seq] [c6 c7 seq seq] -- every chunk carries the sequence number in its LAST word, which the writer stores last,
// so a reader that finds the expected number in all three chunks of ONE set of 16-byte loads holds the whole challenge: a poll is a single
// trip over the host link (three loads in flight), not nine dependent ones (a first version read word by word: +10 us per round).
typedef unsigned gate_v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool gate_try(const gate_v4* g, unsigned seq, fe& out) {
    // three system-coherent 16-byte loads in flight, one wait: volatile asm, so the poll loop really re-reads (a plain load is hoisted)
    gate_v4 a, b, c;
    asm volatile(
        "global_load_dwordx4 %0, %3, off sc0 sc1\n\t"
        "global_load_dwordx4 %1, %3, off offset:16 sc0 sc1\n\t"
        "global_load_dwordx4 %2, %3, off offset:32 sc0 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(a), "=&v"(b), "=&v"(c)
        : "v"(g)
        : "memory");
    if (a.w != seq || b.w != seq || c.z != seq || c.w != seq) return false;
    out.v[0] = a.x; out.v[1] = a.y; out.v[2] = a.z;
    out.v[3] = b.x; out.v[4] = b.y; out.v[5] = b.z;
    out.v[6] = c.x; out.v[7] = c.y;
    return true;
}
__device__ __forceinline__ fe gate_wait(const gate_args& g) {
    __shared__ unsigned s_chal[8];
    if (threadIdx.x == 0) {
        fe c = fe_zero();
        unsigned spins = 0;
        if (blockIdx.x == 0 && blockIdx.y == 0) {
            bool got = false;
            while (!(got = gate_try(reinterpret_cast<const gate_v4*>(g.host), g.seq, c)) && ++spins < g.leader_spins) __builtin_amdgcn_s_sleep(1);
            if (!got) {  // report, then release the grid with a zero challenge (c is still zero: gate_try writes it only on success)
                unsigned* pin_err = const_cast<unsigned*>(g.host) + (PK_PIN_GATE_TIMEOUT - PK_PIN_GATE) / 4;
                __hip_atomic_store(pin_err, g.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            // mirror: the eight data words, then ONE release store of the sequence number (agent scope: the other workgroups' L2)
#pragma unroll
            for (int i = 0; i < 8; i++) __hip_atomic_store(g.dev + i, c.v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(g.dev + 11, g.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            // thousands of wavefronts wait here on one L2 line: one word per poll and a real pause between polls, or the channel that
            // holds the line is saturated and workgroup (0,0)'s own stores queue behind the readers (measured: +10 us per round)
            while (__hip_atomic_load(g.dev + 11, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != g.seq && ++spins < PK_GATE_SPINS_FOLLOWER) __builtin_amdgcn_s_sleep(8);
            __atomic_thread_fence(__ATOMIC_ACQUIRE);  // once, after the word arrived (an acquire per poll invalidates caches thousands of times)
#pragma unroll
            for (int i = 0; i < 8; i++) c.v[i] = __hip_atomic_load(g.dev + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) s_chal[i] = c.v[i];
    }
    __syncthreads();
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = s_chal[i];
    return r;
}
inline unsigned gate_next(pk_ctx* ctx) {
    if (++ctx->gate_seq == 0) ctx->gate_seq = 1;
    return ctx->gate_seq;
}
inline gate_args gate_none() { return gate_args{nullptr, nullptr, 0, 0}; }
inline gate_args gate_for(pk_ctx* ctx, unsigned seq) {  // after reduction_scratch(ctx)
    const long hook = test_hook(PK_HOOK_GATE_SPINS);  // the test-suite's way to reach the give-up path in milliseconds
    const unsigned spins = hook > 0 ? (unsigned)hook : PK_GATE_SPINS_LEADER;
    return gate_args{(const unsigned*)((char*)ctx->h_pinned + PK_PIN_GATE), red_ticket(ctx) + 16, seq, spins};  // both 64-byte aligned
}
// the host's half: the challenge first, the sequence number last
inline void gate_publish(pk_ctx* ctx, unsigned seq, const fe& c) {
    unsigned* g = (unsigned*)((char*)ctx->h_pinned + PK_PIN_GATE);
    const unsigned w[12] = {c.v[0], c.v[1], c.v[2], seq, c.v[3], c.v[4], c.v[5], seq, c.v[6], c.v[7], seq, seq};
    for (int i = 0; i < 12; i++)
        if (i != 3 && i != 7 && i < 10) __atomic_store_n(g + i, w[i], __ATOMIC_RELAXED);
    __atomic_store_n(g + 3, seq, __ATOMIC_RELEASE);  // the tails last (x86 keeps the order of stores; a 16-byte read of a chunk sees a prefix)
    __atomic_store_n(g + 7, seq, __ATOMIC_RELEASE);
    __atomic_store_n(g + 10, seq, __ATOMIC_RELEASE);
    __atomic_store_n(g + 11, seq, __ATOMIC_RELEASE);
}
// did a gated kernel of this context give up on its challenge since the last call?  (clears the word)
inline bool gate_timed_out(pk_ctx* ctx) {
    if (!ctx->h_pinned) return false;
    unsigned* w = (unsigned*)((char*)ctx->h_pinned + PK_PIN_GATE_TIMEOUT);
    return __atomic_exchange_n(w, 0u, __ATOMIC_ACQ_REL) != 0;
}
#define PK_GATE_TIMEOUT_MSG "a gated sumcheck kernel gave up waiting for its challenge (the host thread was stalled for tens of seconds) and ran with a zero challenge: the proof is abandoned"
// wait for the reduction launched with sequence number `seq` WITHOUT draining the stream (a gated kernel may already sit behind it):
// spin on the completion word its finishing workgroup publishes after the K results
template <int K>
inline int collect_reduction_spin(pk_ctx* ctx, unsigned seq, uint64_t* host_out) {
    const unsigned* flag = (const unsigned*)ctx->h_pinned + PK_FLAG_WORD;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned polls = 0;
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
        if ((++polls & 0xfffff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 10.0)
            return set_err(ctx, PK_ERR_HIP, "a gated reduction did not complete within 10 s");
    }
    if (gate_timed_out(ctx)) return set_err(ctx, PK_ERR_HIP, "%s", PK_GATE_TIMEOUT_MSG);
    memcpy(host_out, ctx->h_pinned, 32 * K);
    return PK_OK;
}

// Work items per thread of a reduction kernel.  With the shuffle-tree epilogue (~1,200 instructions per thread for three sums) one
// item per thread made a quarter of such a kernel's vector work epilogue: four items per thread gave +1.6 % on the headline.  With
// the LDS column sums (block_reduce_fe) the epilogue is an eighth of that and the choice hardly matters any more (same box,
// alternating: 1 item 293.2 proofs/s / 8.77 ms one proof at a time, 2: 292.3 / 8.81, 4: 293.9 / 8.88); four stays for the
// many-prover mode, one for latency mode.  profiles/r06_reduction_items_ab.txt
inline unsigned reduction_blocks(const pk_ctx* ctx, size_t work_items) {
    const size_t per_thread = ctx->latency_mode ? 1 : 4;
    size_t need = (work_items + RED_THREADS * per_thread - 1) / (RED_THREADS * per_thread);
    size_t cap = (size_t)ctx->num_cus * 4;
    if (cap > RED_MAX_BLOCKS) cap = RED_MAX_BLOCKS;
    if (need < 1) need = 1;
    return (unsigned)(need < cap ? need : cap);
}

}  // namespace pk
