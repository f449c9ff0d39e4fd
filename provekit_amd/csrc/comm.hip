// comm.hip -- the exchange step of a commit sharded over the GPUs of a node, behind the C ABI.
//
// The reference is a single process on one host: there is nothing to mirror.  What the path needs is ONE
// collective per commit -- an all-gather of the 32-byte leaf digests -- and one per opening (the opened rows, each held by
// exactly one rank).  A pk_ctx optionally carries a communicator with two operations:
//     all_gather(send, recv, bytes_per_rank)            recv[r*bytes .. (r+1)*bytes) = rank r's send
//     all_reduce_sum_u64(buf, count)                    element-wise wrapping sum over ranks (used where exactly one rank
//                                                       contributes a non-zero value, so the "sum" is a gather)
// and three transports:
//   RCCL   one rank per GPU over xGMI: ncclAllGather / ncclAllReduce on the context's stream, in-order with the kernels that
//          produce / consume the buffers (no host synchronisation).  librccl is resolved at run time (dlopen + dlsym), so the
//          library has no link-time dependency on it; a missing or failing RCCL is reported as PK_ERR_RCCL.
//          Multi-process: pk_comm_unique_id on rank 0, broadcast the 128 bytes out of band, pk_comm_init_rank everywhere.
//          Single process: pk_ctx_create_set(devices, n) -> n contexts joined by ncclCommInitAll, one host thread per rank.
//   LOCAL  ranks = contexts of ONE process (any devices, the same device allowed): device-to-device copies between the ranks'
//          buffers with a host barrier.  This is what a one-GPU box can run, so it is the transport the GPU test-suite drives
//          the sharded prover with; RCCL refuses two ranks on one device.
//   (world == 1: no communicator, every call degenerates to a copy.)
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <mutex>

#include "ctx.hpp"
#include "fe29.hpp"

using namespace pk;

namespace {

// ------------------------------------------------------------------ RCCL, resolved at run time
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;  // optional
    ncclResult_t (*GetVersion)(int*) = nullptr;       // optional
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error, path;
};

RcclApi* rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // PK_RCCL_LIB names the library outright (a site build of RCCL; the test-suite's in-process stand-in,
        // tests/stub_rccl).  Otherwise a copy already in the process (e.g. the one a PyTorch wheel bundles) is preferred: one
        // RCCL per process
        const char* forced = getenv("PK_RCCL_LIB");
        if (forced && *forced) {
            api.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) api.path = forced;
        } else {
            const char* names[] = {"librccl.so.1", "librccl.so"};
            for (int pass = 0; pass < 2 && !api.handle; pass++)
                for (const char* n : names) {
                    api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
                    if (api.handle) {
                        api.path = n;
                        break;
                    }
                }
        }
        if (!api.handle) {
            const char* e = dlerror();  // one call: dlerror() clears the message it returns
            api.error = std::string("librccl not found: ") + (e ? e : "dlopen failed");
            return;
        }
#define PK_SYM(field, name)                                                  \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, name)); \
    if (!api.field) api.error = std::string("librccl lacks ") + name
        PK_SYM(GetUniqueId, "ncclGetUniqueId");
        PK_SYM(CommInitRank, "ncclCommInitRank");
        PK_SYM(CommInitAll, "ncclCommInitAll");
        PK_SYM(CommDestroy, "ncclCommDestroy");
        PK_SYM(AllGather, "ncclAllGather");
        PK_SYM(AllReduce, "ncclAllReduce");
        PK_SYM(GetErrorString, "ncclGetErrorString");
#undef PK_SYM
        api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(api.handle, "ncclCommAbort"));
        api.CommGetAsyncError = reinterpret_cast<decltype(api.CommGetAsyncError)>(dlsym(api.handle, "ncclCommGetAsyncError"));
        api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(dlsym(api.handle, "ncclGetVersion"));
    });
    return &api;
}

int rccl_fail(pk_ctx* ctx, const char* what, ncclResult_t r) {
    RcclApi* a = rccl();
    return set_err(ctx, PK_ERR_RCCL, "%s failed: %s", what, a->GetErrorString ? a->GetErrorString(r) : "rccl error");
}
#define PK_RCCL(ctx, expr)                                   \
    do {                                                     \
        ncclResult_t _r = (expr);                            \
        if (_r != ncclSuccess) return rccl_fail(ctx, #expr, _r); \
    } while (0)

// ------------------------------------------------------------------ LOCAL transport
struct LocalGroup {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long generation = 0;
    const void* send[PK_MAX_RANKS] = {};
    hipEvent_t ready[PK_MAX_RANKS] = {};
    int refs = 0;
    bool aborted = false;  // sticky: a rank failed before or inside a collective; every later collective fails on every rank
    // measurement aid (PK_LOCAL_TURNSTILE=1 at pk_comm_init_local): the ranks of this group share ONE GPU in the test-suite, so
    // their kernels overlap and per-kernel timings mean little.  With the turnstile a rank holds a token while it enqueues and
    // runs work inside pk_prove and hands it over at every collective, so the ranks' segments run one after another and
    // pk_profile_* reports each rank's kernels as if it had the chip to itself.
    bool turnstile = false, token_busy = false;
    void token_acquire() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !token_busy || aborted; });
        token_busy = true;
    }
    void token_release() {
        std::lock_guard<std::mutex> lk(mu);
        token_busy = false;
        cv.notify_all();
    }
    // false = the group was aborted (by this or another rank): nobody waits for a rank that will not come
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const unsigned long long gen = generation;
        if (++arrived == world) {
            arrived = 0;
            generation++;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen || aborted; });
        }
        return !aborted;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(mu);
        aborted = true;
        cv.notify_all();
    }
};

__global__ __launch_bounds__(256) void sum_ranks_u64_kernel(const unsigned long long* __restrict__ parts, unsigned long long* __restrict__ out,
                                                            size_t count, unsigned world) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        unsigned long long s = 0;
        for (unsigned r = 0; r < world; r++) s += parts[(size_t)r * count + i];
        out[i] = s;
    }
}

// the K partial results of every rank (block r = rank r's K field elements), optionally scaled per rank, summed mod p and
// written to the pinned result page the host reads after the stream synchronisation
struct rank_scales {
    u32 v[PK_MAX_RANKS][8];
};
__global__ void sum_ranks_fe_kernel(const fe* __restrict__ gathered, unsigned world, unsigned K, rank_scales sc, int scaled, fe* __restrict__ host_out) {
    const unsigned k = threadIdx.x;
    if (k >= K) return;
    fe acc = fe_zero();
    for (unsigned r = 0; r < world; r++) {
        fe x = fe_load(gathered + (size_t)r * K + k);
        if (scaled) {
            fe s;
#pragma unroll
            for (int i = 0; i < 8; i++) s.v[i] = sc.v[r][i];
            x = fe_mulx(x, s);
        }
        acc = fe_add(acc, x);
    }
    unsigned* out = reinterpret_cast<unsigned*>(host_out + k);
#pragma unroll
    for (int i = 0; i < 8; i++) __hip_atomic_store(out + i, acc.v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

struct pk_comm {
    int kind = PK_COMM_NONE;
    int rank = 0, world = 1;
    ncclComm_t nccl = nullptr;
    LocalGroup* grp = nullptr;
    void* d_tmp = nullptr;  // LOCAL / HOST all-reduce staging, grow-only
    size_t tmp_bytes = 0;
    // HOST transport: the caller's all-gather over host buffers and the pinned staging it runs between
    pk_host_all_gather_fn host_fn = nullptr;
    void* host_user = nullptr;
    char* h_stage = nullptr;
    size_t stage_bytes = 0;
    bool failed = false;  // sticky, like LocalGroup::aborted
    unsigned long long issued = 0;  // collectives enqueued so far (any transport)
    bool pending = false;  // RCCL: a collective was enqueued on the stream and has not been seen complete yet (comm_wait)
    hipEvent_t done = nullptr;  // RCCL: recorded on the stream right behind the latest collective: the deadline is the collective's, not the stream's
    bool holds_token = false;  // LOCAL turnstile: between comm_turn_begin and comm_turn_end
};

static void rccl_abort_own(pk_comm* c) {
    c->failed = true;
    c->pending = false;
    if (c->nccl) {
        // once a collective has failed or timed out ncclCommDestroy may itself block on the stuck kernel: abort, never destroy
        if (rccl()->CommAbort) (void)rccl()->CommAbort(c->nccl);  // frees the communicator
        c->nccl = nullptr;  // (without ncclCommAbort the handle is leaked on purpose: destroying it could hang)
    }
}
// an event on the stream right behind the collective just enqueued: comm_wait's deadline watches THIS, so that hours of honest work
// queued behind a collective are never mistaken for a collective that hangs
static int rccl_mark(pk_ctx* ctx, pk_comm* c) {
    if (!c->done) PK_HIP(ctx, hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
    PK_HIP(ctx, hipEventRecord(c->done, ctx->stream));
    return PK_OK;
}

namespace pk {

int comm_rank(const pk_ctx* ctx) { return ctx->comm ? ctx->comm->rank : 0; }
int comm_world(const pk_ctx* ctx) { return ctx->comm ? ctx->comm->world : 1; }

int comm_all_gather(pk_ctx* ctx, const void* d_send, void* d_recv, size_t bytes) {
    pk_comm* c = ctx->comm;
    if (!c) {  // no communicator = one rank
        if (d_send != d_recv && bytes) PK_HIP(ctx, hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        return PK_OK;
    }
    ProfScope prof(ctx, "comm_all_gather");
    c->issued++;
    if (c->kind == PK_COMM_RCCL) {
        if (c->failed || !c->nccl) return set_err(ctx, PK_ERR_RCCL, "the communicator was aborted by an earlier failure of this rank");
        c->pending = true;  // whoever waits for this stream next waits with a deadline (comm_wait)
        const ncclResult_t r = rccl()->AllGather(d_send, d_recv, bytes, ncclUint8, c->nccl, ctx->stream);
        if (r != ncclSuccess) {
            rccl_abort_own(c);  // the ranks are out of step from here on: no later collective may be attempted on this communicator
            return rccl_fail(ctx, "ncclAllGather", r);
        }
        return rccl_mark(ctx, c);
    }
    if (c->kind == PK_COMM_HOST) {
        if (c->failed) return set_err(ctx, PK_ERR_RCCL, "the host transport failed earlier; the communicator is unusable");
        const size_t need = bytes * ((size_t)c->world + 1);
        if (c->stage_bytes < need) {
            if (c->h_stage) (void)hipHostFree(c->h_stage);
            c->h_stage = nullptr;
            c->stage_bytes = 0;
            PK_HIP(ctx, hipHostMalloc((void**)&c->h_stage, need, hipHostMallocDefault));
            c->stage_bytes = need;
        }
        char* h_send = c->h_stage;
        char* h_recv = c->h_stage + bytes;
        PK_HIP(ctx, hipMemcpyAsync(h_send, d_send, bytes, hipMemcpyDeviceToHost, ctx->stream));
        PK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const int frc = c->host_fn(c->host_user, h_send, h_recv, bytes);
        if (frc) {
            c->failed = true;
            return set_err(ctx, PK_ERR_RCCL, "the host transport's all-gather returned %d", frc);
        }
        PK_HIP(ctx, hipMemcpyAsync(d_recv, h_recv, bytes * (size_t)c->world, hipMemcpyHostToDevice, ctx->stream));
        PK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the staging area is reused by the next collective
        return PK_OK;
    }
    // a rank that fails must not leave the others waiting at the barrier: it aborts the group, which wakes them with an error
    LocalGroup* g = c->grp;
    auto fail = [&](hipError_t e, const char* what) {
        g->abort();
        return set_err(ctx, PK_ERR_HIP, "%s failed in the in-process all-gather: %s", what, hipGetErrorString(e));
    };
    const char* peer_failed = "a rank of the device set failed; the communicator is unusable";
    hipError_t e = hipEventRecord(g->ready[c->rank], ctx->stream);  // the send buffer is complete at this point of the stream
    if (e != hipSuccess) return fail(e, "hipEventRecord");
    g->send[c->rank] = d_send;
    const bool turn = g->turnstile && c->holds_token;
    if (turn) {  // this rank's segment ends here: drain it, then let the next rank run
        if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(e, "hipStreamSynchronize");
        g->token_release();
    }
    const bool arrived = g->barrier();
    if (turn) g->token_acquire();
    if (!arrived) return set_err(ctx, PK_ERR_RCCL, "%s", peer_failed);
    for (int p = 0; p < c->world; p++) {
        if ((e = hipStreamWaitEvent(ctx->stream, g->ready[p], 0)) != hipSuccess) return fail(e, "hipStreamWaitEvent");
        if ((e = hipMemcpyAsync((char*)d_recv + (size_t)p * bytes, g->send[p], bytes, hipMemcpyDefault, ctx->stream)) != hipSuccess)
            return fail(e, "hipMemcpyAsync");
    }
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(e, "hipStreamSynchronize");
    if (turn) g->token_release();
    const bool done = g->barrier();  // every rank has read every send buffer: they may be reused
    if (turn) g->token_acquire();
    if (!done) return set_err(ctx, PK_ERR_RCCL, "%s", peer_failed);
    return PK_OK;
}

// A rank that fails BEFORE reaching a collective (e.g. its encode ran out of memory) calls comm_abort.  What that does for the ranks
// already waiting in the collective depends on the transport:
//   LOCAL  the group's sticky flag wakes them at once with PK_ERR_RCCL.
//   RCCL   nothing reaches the peers: ncclCommAbort is LOCAL -- it tears down THIS rank's communicator (and kills its own pending
//          collective kernel); the peers' collective kernels keep waiting on xGMI for a rank that will not come.  They are rescued by
//          their own deadline: every wait behind a collective (comm_wait below) polls an event recorded right after that collective
//          and ncclCommGetAsyncError, and after PK_COMM_TIMEOUT_S seconds (default 120) aborts ITS OWN communicator -- which ends its
//          stuck kernel -- and returns PK_ERR_RCCL.  The communicator is unusable afterwards on every rank (pk_comm_destroy + a fresh
//          pk_comm_init_rank to continue).
//   HOST   this rank fails fast, its peers are the caller's transport's to time out.
// pk_prove brackets itself with these; no-ops unless the context's in-process group was created with the turnstile on
void comm_turn_begin(pk_ctx* ctx) {
    pk_comm* c = ctx->comm;
    if (c && c->kind == PK_COMM_LOCAL && c->grp && c->grp->turnstile && !c->holds_token) {
        c->grp->token_acquire();
        c->holds_token = true;
    }
}
void comm_turn_end(pk_ctx* ctx) {
    pk_comm* c = ctx->comm;
    if (c && c->holds_token) {
        (void)hipStreamSynchronize(ctx->stream);
        c->holds_token = false;
        c->grp->token_release();
    }
}
void comm_abort(pk_ctx* ctx) {
    pk_comm* c = ctx->comm;
    if (!c) return;
    if (c->kind == PK_COMM_LOCAL && c->grp) c->grp->abort();
    if (c->kind == PK_COMM_HOST) c->failed = true;  // this rank's later collectives fail fast; its peers are the caller's transport's to time out
    if (c->kind == PK_COMM_RCCL && !c->failed) rccl_abort_own(c);
}

static double comm_timeout_s() {
    static const double t = [] {
        const char* e = getenv("PK_COMM_TIMEOUT_S");
        const double v = e ? atof(e) : 0.0;
        return v > 0.0 ? v : 120.0;
    }();
    return t;
}
unsigned long long comm_collectives_issued(const pk_ctx* ctx) { return ctx->comm ? ctx->comm->issued : 0; }
bool comm_rccl(const pk_ctx* ctx) { return ctx->comm && ctx->comm->kind == PK_COMM_RCCL; }
bool comm_collective_pending(const pk_ctx* ctx) { return ctx->comm && ctx->comm->kind == PK_COMM_RCCL && ctx->comm->pending; }
// The wait for a stream that carries an RCCL collective: a peer that died or aborted never arrives, and the collective kernel would
// spin for ever.  Poll the stream; between polls ask RCCL for an asynchronous error; past the deadline abort this rank's own
// communicator (which ends its kernel) and report.  hipSuccess = the stream drained and the collective completed.
hipError_t comm_wait(pk_ctx* ctx) {
    pk_comm* c = ctx->comm;
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    double waited = 0.0;
    for (unsigned polls = 1; c->pending; polls++) {
        const hipError_t e = c->done ? hipEventQuery(c->done) : hipStreamQuery(ctx->stream);
        if (e == hipSuccess) {
            c->pending = false;  // the collective itself is through; whatever is queued behind it is ordinary work
            break;
        }
        if (e != hipErrorNotReady) return e;
        // a healthy collective is over within a millisecond: the first 5 ms are a plain spin (a sharded proof waits ~65 times and
        // must not pay a timer for each); after that nothing is urgent any more -- sleep between polls
        if (waited >= 5e-3) {
            struct timespec ts = {0, 200000};
            (void)nanosleep(&ts, nullptr);
        } else if ((polls & 255) != 0) {
            continue;
        }
        struct timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        waited = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        if (waited < 5e-3) continue;
        ncclResult_t async = ncclSuccess;
        const bool broken = c->nccl && rccl()->CommGetAsyncError && rccl()->CommGetAsyncError(c->nccl, &async) == ncclSuccess && async != ncclSuccess &&
                            async != ncclInProgress;
        if (broken || waited > comm_timeout_s()) {
            rccl_abort_own(c);
            (void)hipStreamSynchronize(ctx->stream);  // the aborted kernel exits; drain what was queued behind it
            set_err(ctx, PK_ERR_RCCL, broken ? "RCCL reported an asynchronous error (%s) while a collective was pending; this rank's communicator was aborted"
                                             : "a collective did not complete within %s seconds (a rank of the device set failed or never arrived); this rank's communicator was aborted",
                    broken ? (rccl()->GetErrorString ? rccl()->GetErrorString(async) : "rccl error") : std::to_string((int)comm_timeout_s()).c_str());
            return hipErrorUnknown;
        }
    }
    return wait_stream(ctx->device, ctx->stream);
}

int comm_all_reduce_sum_u64(pk_ctx* ctx, uint64_t* d_buf, size_t count) {
    pk_comm* c = ctx->comm;
    if (!c || !count) return PK_OK;
    ProfScope prof(ctx, "comm_all_reduce");
    if (c->kind == PK_COMM_RCCL) {
        if (c->failed || !c->nccl) return set_err(ctx, PK_ERR_RCCL, "the communicator was aborted by an earlier failure of this rank");
        c->issued++;
        c->pending = true;
        const ncclResult_t r = rccl()->AllReduce(d_buf, d_buf, count, ncclUint64, ncclSum, c->nccl, ctx->stream);
        if (r != ncclSuccess) {
            rccl_abort_own(c);
            return rccl_fail(ctx, "ncclAllReduce", r);
        }
        return rccl_mark(ctx, c);
    }
    const size_t need = (size_t)c->world * count * 8;
    if (c->tmp_bytes < need) {
        if (c->d_tmp) {
            PK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            PK_HIP(ctx, hipFree(c->d_tmp));
            c->d_tmp = nullptr;
            c->tmp_bytes = 0;
        }
        PK_HIP(ctx, hipMalloc(&c->d_tmp, need));
        c->tmp_bytes = need;
    }
    int rc = comm_all_gather(ctx, d_buf, c->d_tmp, count * 8);
    if (rc) return rc;
    sum_ranks_u64_kernel<<<grid_for(ctx, count, 256), 256, 0, ctx->stream>>>((const unsigned long long*)c->d_tmp, (unsigned long long*)d_buf, count,
                                                                             (unsigned)c->world);
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}

int red_across_begin(pk_ctx* ctx) {
    if (!ctx->d_xred) PK_HIP(ctx, hipMalloc(&ctx->d_xred, PK_XRED_OWN_BYTES + (size_t)PK_MAX_RANKS * 8 * 32));
    int rc = ensure_pinned(ctx);
    if (rc) return rc;
    ctx->red_across = true;
    ctx->red_scales = nullptr;
    return PK_OK;
}

int comm_collect_fe(pk_ctx* ctx, int K, uint64_t* host_out) {
    const unsigned world = (unsigned)comm_world(ctx);
    fe* own = (fe*)ctx->d_xred;
    fe* gathered = (fe*)((char*)ctx->d_xred + PK_XRED_OWN_BYTES);
    int rc = comm_all_gather(ctx, own, gathered, (size_t)K * 32);
    if (rc) return rc;
    rank_scales sc{};
    if (ctx->red_scales) memcpy(sc.v, ctx->red_scales, (size_t)world * 32);
    sum_ranks_fe_kernel<<<1, 64, 0, ctx->stream>>>(gathered, world, (unsigned)K, sc, ctx->red_scales != nullptr, (fe*)ctx->h_pinned);
    PK_LAUNCH_CHECK(ctx);
    rc = sync_stream(ctx);
    if (rc) return rc;
    memcpy(host_out, ctx->h_pinned, 32 * (size_t)K);
    return PK_OK;
}

void comm_release(pk_ctx* ctx) {
    pk_comm* c = ctx->comm;
    if (!c) return;
    if (c->kind == PK_COMM_RCCL && c->nccl) {
        if (c->failed || c->pending) rccl_abort_own(c);  // a stuck collective would block ncclCommDestroy too
        else if (rccl()->CommDestroy) (void)rccl()->CommDestroy(c->nccl);
    }
    if (c->kind == PK_COMM_LOCAL && c->grp) {
        LocalGroup* g = c->grp;
        bool last;
        {
            std::lock_guard<std::mutex> lk(g->mu);
            if (g->ready[c->rank]) (void)hipEventDestroy(g->ready[c->rank]);
            g->ready[c->rank] = nullptr;
            last = --g->refs == 0;
        }
        if (last) delete g;
    }
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->d_tmp) (void)hipFree(c->d_tmp);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    delete c;
    ctx->comm = nullptr;
}

}  // namespace pk

extern "C" {

int pk_comm_unique_id(uint8_t id[PK_COMM_ID_BYTES]) {
    if (!id) return PK_ERR_BAD_ARG;
    RcclApi* a = rccl();
    if (!a->error.empty()) return PK_ERR_RCCL;
    static_assert(PK_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId u;
    if (a->GetUniqueId(&u) != ncclSuccess) return PK_ERR_RCCL;
    memcpy(id, u.internal, PK_COMM_ID_BYTES);
    return PK_OK;
}

int pk_comm_init_rank(pk_ctx* ctx, const uint8_t id[PK_COMM_ID_BYTES], int world, int rank) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, id && world >= 1 && world <= PK_MAX_RANKS && rank >= 0 && rank < world, "bad rank / world");
    PK_REQUIRE(ctx, is_pow2((size_t)world), "the number of ranks must be a power of two (leaf-index sharding)");
    PK_REQUIRE(ctx, !ctx->comm, "context already has a communicator");
    RcclApi* a = rccl();
    if (!a->error.empty()) return set_err(ctx, PK_ERR_RCCL, "%s", a->error.c_str());
    pk_comm* c = new (std::nothrow) pk_comm();
    if (!c) return PK_ERR_OOM;
    c->kind = PK_COMM_RCCL;
    c->rank = rank;
    c->world = world;
    ncclUniqueId u;
    memcpy(u.internal, id, PK_COMM_ID_BYTES);
    ncclResult_t r = a->CommInitRank(&c->nccl, world, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return rccl_fail(ctx, "ncclCommInitRank", r);
    }
    ctx->comm = c;
    return PK_OK;
}

int pk_comm_init_host(pk_ctx* ctx, int world, int rank, pk_host_all_gather_fn fn, void* user) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, fn && world >= 1 && world <= PK_MAX_RANKS && rank >= 0 && rank < world, "bad rank / world / callback");
    PK_REQUIRE(ctx, is_pow2((size_t)world), "the number of ranks must be a power of two (leaf-index sharding)");
    PK_REQUIRE(ctx, !ctx->comm, "context already has a communicator");
    pk_comm* c = new (std::nothrow) pk_comm();
    if (!c) return PK_ERR_OOM;
    c->kind = PK_COMM_HOST;
    c->rank = rank;
    c->world = world;
    c->host_fn = fn;
    c->host_user = user;
    ctx->comm = c;
    return PK_OK;
}

int pk_comm_init_local(pk_ctx* const* ctxs, int n) {
    if (!ctxs || n < 1 || n > PK_MAX_RANKS || !is_pow2((size_t)n)) return PK_ERR_BAD_ARG;
    for (int i = 0; i < n; i++)
        if (!ctxs[i] || ctxs[i]->comm) return PK_ERR_BAD_ARG;
    LocalGroup* g = new (std::nothrow) LocalGroup();
    if (!g) return PK_ERR_OOM;
    g->world = n;
    g->refs = n;
    {
        const char* ts = getenv("PK_LOCAL_TURNSTILE");
        g->turnstile = ts && ts[0] == '1';
    }
    pk_comm* cs[PK_MAX_RANKS] = {};
    bool ok = true;
    int caller_device = 0;
    const bool have_caller_device = hipGetDevice(&caller_device) == hipSuccess;
    for (int i = 0; i < n && ok; i++) {
        cs[i] = new (std::nothrow) pk_comm();
        ok = cs[i] && hipSetDevice(ctxs[i]->device) == hipSuccess && hipEventCreateWithFlags(&g->ready[i], hipEventDisableTiming) == hipSuccess;
    }
    if (have_caller_device) (void)hipSetDevice(caller_device);  // the caller's current device is not ours to change
    if (!ok) {
        for (int i = 0; i < n; i++) {
            delete cs[i];
            if (g->ready[i]) (void)hipEventDestroy(g->ready[i]);
        }
        delete g;
        return PK_ERR_HIP;
    }
    for (int i = 0; i < n; i++) {
        cs[i]->kind = PK_COMM_LOCAL;
        cs[i]->rank = i;
        cs[i]->world = n;
        cs[i]->grp = g;
        ctxs[i]->comm = cs[i];
    }
    return PK_OK;
}

int pk_ctx_create_set(const int* devices, int n, pk_ctx** out) {
    if (!devices || !out || n < 1 || n > PK_MAX_RANKS || !is_pow2((size_t)n)) return PK_ERR_BAD_ARG;
    for (int i = 0; i < n; i++) out[i] = nullptr;
    int rc = PK_OK;
    bool distinct = true;
    for (int i = 0; i < n && !rc; i++) {
        rc = pk_ctx_create(devices[i], &out[i]);
        for (int j = 0; j < i; j++) distinct = distinct && devices[i] != devices[j];
    }
    if (!rc && n > 1) {
        // test hook (pk_selftest_set_hook, together with PK_RCCL_LIB = the in-process stand-in): take the RCCL branch even for a
        // repeated device -- real RCCL refuses two ranks on one GPU
        if (!distinct && !test_hook(PK_HOOK_RCCL_SAME_DEVICE)) {
            rc = pk_comm_init_local(out, n);  // several ranks on one device: RCCL cannot, the in-process transport can
        } else {
            RcclApi* a = rccl();
            ncclComm_t comms[PK_MAX_RANKS];
            if (!a->error.empty()) {
                rc = set_err(out[0], PK_ERR_RCCL, "%s", a->error.c_str());
            } else {
                ncclResult_t r = a->CommInitAll(comms, n, devices);
                if (r != ncclSuccess) rc = rccl_fail(out[0], "ncclCommInitAll", r);
            }
            const bool have_comms = !rc;
            int adopted = 0;  // communicators now owned by a context (released with it)
            for (int i = 0; i < n && !rc; i++) {
                pk_comm* c = new (std::nothrow) pk_comm();
                if (!c) {
                    rc = PK_ERR_OOM;
                    break;
                }
                c->kind = PK_COMM_RCCL;
                c->rank = i;
                c->world = n;
                c->nccl = comms[i];
                out[i]->comm = c;
                adopted = i + 1;
            }
            if (rc && have_comms)
                for (int i = adopted; i < n; i++) (void)a->CommDestroy(comms[i]);  // not adopted: would leak
        }
    }
    if (rc) {
        for (int i = 0; i < n; i++)
            if (out[i]) {
                pk_ctx_destroy(out[i]);
                out[i] = nullptr;
            }
    }
    return rc;
}

int pk_comm_rccl_version(int* version, char* path, size_t cap) {
    RcclApi* a = rccl();
    if (!a->error.empty()) return PK_ERR_RCCL;
    if (version) {
        *version = 0;
        if (a->GetVersion && a->GetVersion(version) != ncclSuccess) return PK_ERR_RCCL;
    }
    if (path && cap) snprintf(path, cap, "%s", a->path.c_str());
    return PK_OK;
}

int pk_comm_info(const pk_ctx* ctx, int* rank, int* world, int* kind) {
    if (!ctx) return PK_ERR_BAD_ARG;
    if (rank) *rank = pk::comm_rank(ctx);
    if (world) *world = pk::comm_world(ctx);
    if (kind) *kind = ctx->comm ? ctx->comm->kind : PK_COMM_NONE;
    return PK_OK;
}

int pk_comm_reset(pk_ctx* ctx) {
    PK_ENTER(ctx);
    pk_comm* c = ctx->comm;
    if (!c) return PK_OK;
    if (c->kind == PK_COMM_RCCL) {
        if (c->failed || !c->nccl) return set_err(ctx, PK_ERR_RCCL, "an aborted RCCL communicator cannot be reset: pk_comm_destroy, then join again");
        return PK_OK;
    }
    if (c->kind == PK_COMM_HOST) c->failed = false;
    if (c->kind == PK_COMM_LOCAL && c->grp) {
        std::lock_guard<std::mutex> lk(c->grp->mu);
        c->grp->aborted = false;
        c->grp->arrived = 0;  // ranks that were woken out of a barrier by the abort never completed it
        c->grp->token_busy = false;
    }
    c->holds_token = false;
    return PK_OK;
}

int pk_comm_destroy(pk_ctx* ctx) {
    PK_ENTER(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    pk::comm_release(ctx);
    return PK_OK;
}

int pk_comm_all_gather(pk_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_rank) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, bytes_per_rank == 0 || (d_send && d_recv), "null pointer");
    return pk::comm_all_gather(ctx, d_send, d_recv, bytes_per_rank);
}

int pk_comm_all_reduce_sum_u64(pk_ctx* ctx, uint64_t* d_buf, size_t count) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, count == 0 || d_buf, "null pointer");
    return pk::comm_all_reduce_sum_u64(ctx, d_buf, count);
}

}  // extern "C"
