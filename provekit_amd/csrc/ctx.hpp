// ctx.hpp -- library context shared by all translation units of libprovekit_hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/provekit_hip.h"

struct pk_prof_rec {
    const char* name;
    hipEvent_t e0, e1;
};

struct pk_comm;  // comm.hip

struct pk_ctx {
    int device = 0;
    pk_comm* comm = nullptr;  // optional: this context's rank in a sharded commit (comm.hip); null = single GPU
    // optional per-kernel timing (pk_profile_*): hipEvent pairs recorded on the work stream
    bool prof_on = false;
    std::vector<pk_prof_rec> prof;
    std::vector<hipEvent_t> ev_pool;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // the stream work is enqueued on (own or borrowed)
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    int hash_version = 2;
    int num_cus = 256;
    char err[512] = {0};
    int err_code = 0;  // the status the last set_err returned
    // small device scratch for reductions (partials + results)
    void* d_scratch = nullptr;
    size_t scratch_bytes = 0;
    void* h_pinned = nullptr;  // pinned host staging for small results
    size_t pinned_bytes = 0;
    std::map<unsigned, void*> twiddles;  // log2(N) -> device table of w_N^e, N entries (owned; ntt.hip)
    std::map<unsigned, void*> twiddles29[2];  // [0] of twiddles, [1] of twiddles_scaled: 18 x u32 per entry, the Shoup multiplier form (ntt_regs.hpp tw29s)
    std::map<unsigned, void*> twiddles_pass;   // inter-pass twiddles in access order, per (size, pass, variant) (ntt.hip get_pass_table)
    std::map<unsigned, void*> twiddles_scaled;  // log2(N) -> 32 * w_N^e as plain integers: the hash-ready output scaling (ntt.hip)
    unsigned red_seq = 0;  // sequence number of the last reduction launch (completion flag in h_pinned)
    // Latency mode (pk_ctx_set_latency_mode; ONE proof at a time on this context, the chip otherwise idle): the rounds of a sumcheck
    // are enqueued one ahead -- round k+1's kernel is already in the queue, gated on a word of the pinned page, while the host
    // absorbs round k's result and squeezes the challenge -- so a Fiat-Shamir round trip costs the link latency (~2 us) instead of a
    // kernel launch + stream synchronisation (~13 us; profiles/r04_roundtrip.json).  Off by default: a gated kernel occupies its
    // workgroup slots while it waits, which is wasted capacity when other provers share the chip.
    bool latency_mode = false;
    unsigned gate_seq = 0;  // sequence number of the last gate handed out (reduce.hpp)
    void* d_ws = nullptr;  // large reusable workspace (NTT scratch); grows, never shrinks
    void* d_ztab = nullptr;  // 256 field elements: z^t for the point of the univariate evaluation in flight (mle.hip)
    size_t ws_bytes = 0;
    // "mailbox": device-visible pinned host memory the kernels read small inputs from and write small outputs to, so
    // the Fiat-Shamir round trips need no copy operations (each hipMemcpyAsync is a blit dispatch for the command processor).
    // Bump-allocated; the offset rewinds at every stream synchronisation (pk::sync_stream).
    char* h_mail = nullptr;
    size_t mail_bytes = 0, mail_off = 0;
    bool pow_armed = false;  // pow.hip: device-side best/ticket words initialised
    bool red_armed = false;  // reduce.hpp: ticket word zeroed
    // One proof sharded over a device set (prover.hip): while `red_across` is set, the operands of the reduction kernels
    // (sumcheck rounds, dot products, Horner) are this rank's shard, their results partial sums -- the finishing workgroup
    // writes them to the device block d_xred instead of the pinned page, and collect_reduction all-gathers the K x 32 bytes,
    // adds the ranks' partials mod p (each optionally multiplied by red_scales[rank] first) and only then hands them to the host.
    bool red_across = false;
    void* d_xred = nullptr;             // [own results, 2 KiB | gathered blocks of the ranks]
    const void* red_scales = nullptr;   // host, comm_world(ctx) field elements (Montgomery), or null
};
#define PK_XRED_OWN_BYTES 2048

// fixed slots in the 4 KiB h_pinned page: [0,1024) reduction results, word 256 completion flag (reduce.hpp)
#define PK_PIN_ROOT 2048 /* 32 B: the root of the last Merkle tree built on this context (hash.hip) */
#define PK_PIN_POW 2112  /* 8 B: the nonce found by the last proof-of-work launch (pow.hip) */
#define PK_PIN_GATE 2304 /* 48 B, 64-byte aligned: the gate through which the host publishes a round's challenge (reduce.hpp) */
#define PK_PIN_GATE_TIMEOUT 2368 /* 4 B: sequence number of a gate whose kernel gave up waiting (reduce.hpp); 0 = none */

// A launch too small to fill the chip is latency-bound, and it sits on some prover's Fiat-Shamir critical path while the
// chip-filling kernels of the other provers share its SIMDs: let its wavefronts issue ahead of theirs (s_setprio 3).  The
// chip-filling launches keep the default priority 0, so among themselves nothing changes.
#ifndef PK_BASE_PRIO
#define PK_BASE_PRIO 0
#endif
#define PK_LATENCY_PRIO()                                                        \
    do {                                                                         \
        if (gridDim.x * gridDim.y <= 128u) __builtin_amdgcn_s_setprio(3);        \
        else if (PK_BASE_PRIO) __builtin_amdgcn_s_setprio(PK_BASE_PRIO);         \
    } while (0)

namespace pk {

inline int set_err(pk_ctx* ctx, int code, const char* fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
        va_end(ap);
        ctx->err_code = code;
    }
    return code;
}

#define PK_HIP(ctx, expr)                                                                          \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return pk::set_err(ctx, _e == hipErrorOutOfMemory ? PK_ERR_OOM : PK_ERR_HIP,           \
                               "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,    \
                               __LINE__);                                                          \
    } while (0)

// wait for the context's stream (with the collective deadline of comm.hip comm_wait while an RCCL collective is pending on it)
#define PK_WAIT(ctx)                      \
    do {                                  \
        int _w = pk::wait_ctx_rc(ctx);    \
        if (_w) return _w;                \
    } while (0)

#define PK_REQUIRE(ctx, cond, msg)                                                \
    do {                                                                          \
        if (!(cond)) return pk::set_err(ctx, PK_ERR_BAD_ARG, "bad argument: %s", msg); \
    } while (0)

#define PK_LAUNCH_CHECK(ctx) PK_HIP(ctx, hipGetLastError())

// first line of every entry point that touches the device: the HIP current device is per host thread, and callers (one
// prover thread per context, one context per GPU) need not have selected this context's device on the calling thread
#define PK_ENTER(ctx)                                                                              \
    do {                                                                                           \
        if (!(ctx)) return PK_ERR_BAD_ARG;                                                         \
        if (hipSetDevice((ctx)->device) != hipSuccess)                                             \
            return pk::set_err(ctx, PK_ERR_HIP, "hipSetDevice(%d) failed", (ctx)->device);         \
    } while (0)

inline bool is_pow2(size_t x) { return x && !(x & (x - 1)); }
inline unsigned ilog2(size_t x) {
    unsigned l = 0;
    while (x > 1) {
        x >>= 1;
        l++;
    }
    return l;
}

// grid size for a flat elementwise / grid-stride launch
inline unsigned grid_for(const pk_ctx* ctx, size_t n, unsigned block, unsigned max_blocks_per_cu = 8) {
    size_t need = (n + block - 1) / block;
    size_t cap = (size_t)ctx->num_cus * max_blocks_per_cu;
    if (need < 1) need = 1;
    return (unsigned)(need < cap ? need : cap);
}

// RAII bracket around a kernel launch (or a group of launches) for pk_profile_*; free when profiling is off
struct ProfScope {
    pk_ctx* c;
    hipEvent_t e1 = nullptr;
    ProfScope(pk_ctx* ctx, const char* name) : c(ctx) {
        if (!c->prof_on) return;
        hipEvent_t ev[2];
        for (int i = 0; i < 2; i++) {
            if (!c->ev_pool.empty()) {
                ev[i] = c->ev_pool.back();
                c->ev_pool.pop_back();
            } else if (hipEventCreate(&ev[i]) != hipSuccess) {
                return;
            }
        }
        (void)hipEventRecord(ev[0], c->stream);
        e1 = ev[1];
        c->prof.push_back({name, ev[0], ev[1]});
    }
    ~ProfScope() {
        if (e1) (void)hipEventRecord(e1, c->stream);
    }
};

int ensure_scratch(pk_ctx* ctx, size_t bytes);
int ensure_pinned(pk_ctx* ctx);                               // the 4 KiB result page
// Every wait of the library for a stream goes through here (pk_device_set_host_wait): the runtime's hipStreamSynchronize -- spinning or
// blocking, whichever the device's scheduling flag says -- or, in PK_WAIT_POLL, the library's own loop: hipStreamQuery with short sleeps.
// test hooks, settable only through pk_selftest_set_hook (tools/probes/pk_selftest.h); 0 = off: a gated kernel's spin bound, microseconds the
// host sleeps before it publishes a gate's challenge, take the RCCL branch for a repeated device
enum { PK_HOOK_GATE_SPINS = 0, PK_HOOK_GATE_STALL_US = 1, PK_HOOK_RCCL_SAME_DEVICE = 2, PK_HOOK_COUNT = 3 };
long test_hook(int which);
hipError_t wait_stream(int device, hipStream_t stream);
hipError_t wait_ctx(pk_ctx* ctx);  // wait_stream on the context's stream; with a deadline while an RCCL collective is pending on it (comm.hip comm_wait)
int wait_ctx_rc(pk_ctx* ctx);      // the same as a status: PK_OK, PK_ERR_RCCL (the collective failed or timed out; pk_last_error says which) or PK_ERR_HIP
bool comm_collective_pending(const pk_ctx* ctx);
bool comm_rccl(const pk_ctx* ctx);
hipError_t comm_wait(pk_ctx* ctx);
int sync_stream(pk_ctx* ctx);                                 // wait_stream + rewind the mailbox
int mail_alloc(pk_ctx* ctx, size_t bytes, void** out);        // 64-B aligned; valid until the next sync_stream
int read_root(pk_ctx* ctx, const uint64_t* d_nodes, size_t n_leaves, uint64_t root[4]);  // hash.hip: after pk_merkle_*
int ensure_ws(pk_ctx* ctx, size_t bytes);
// comm.hip: rank / size of the context's communicator (0 / 1 without one) and its two collectives, enqueued on ctx->stream
int comm_rank(const pk_ctx* ctx);
int comm_world(const pk_ctx* ctx);
int comm_all_gather(pk_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_rank);
int comm_all_reduce_sum_u64(pk_ctx* ctx, uint64_t* d_buf, size_t count);
int comm_collect_fe(pk_ctx* ctx, int K, uint64_t* host_out);  // comm.hip: the cross-rank half of collect_reduction
int red_across_begin(pk_ctx* ctx);                             // allocate d_xred if needed and set red_across
void comm_turn_begin(pk_ctx* ctx);  // measurement aid of the in-process transport (comm.hip LocalGroup::turnstile)
void comm_turn_end(pk_ctx* ctx);
unsigned long long comm_collectives_issued(const pk_ctx* ctx);  // collectives this context's communicator has enqueued so far
void comm_abort(pk_ctx* ctx);  // this rank will not reach a collective its peers wait in: LOCAL wakes them; RCCL aborts its OWN communicator (the peers time out, comm_wait)
void comm_release(pk_ctx* ctx);
int eval_univariate_multi(pk_ctx* ctx, const uint64_t* const* d_polys, unsigned np, size_t n, const uint64_t z[4], uint64_t* out);  // mle.hip
int dot_rows(pk_ctx* ctx, const uint64_t* d_w, size_t row_stride, unsigned nrows, const uint64_t* d_f, const uint64_t* d_g, size_t n, uint64_t* out);
int pow_solve_x(pk_ctx* ctx, const uint8_t challenge[32], double bits, uint64_t* nonce, bool striped);  // pow.hip
void ntt_retain_ctx(pk_ctx* ctx);   // ntt.hip: one more context on this device shares its twiddle tables
void ntt_release_ctx(pk_ctx* ctx);  // ntt.hip: the context lets go of the device's twiddle tables (freed with the last context)

}  // namespace pk
