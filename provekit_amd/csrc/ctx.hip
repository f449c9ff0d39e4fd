// ctx.hip -- context, memory and timing entry points of the C ABI, plus the
// elementwise field kernels (rows A1/A2).
#include "ctx.hpp"
#include "fe29.hpp"

#include <sys/prctl.h>
#include <time.h>

#include <atomic>

using namespace pk;

#define PK_MAX_DEVICES 64
static std::atomic<int> g_poll_wait[PK_MAX_DEVICES];
static std::atomic<long> g_test_hooks[pk::PK_HOOK_COUNT];
// the runtime wait mode this library last set on a device (PK_WAIT_SPIN / PK_WAIT_BLOCK); -1 = never touched: HIP's hipDeviceScheduleAuto
static std::atomic<int> g_runtime_wait[PK_MAX_DEVICES];
static const bool g_runtime_wait_init = [] {
    for (auto& w : g_runtime_wait) w.store(-1, std::memory_order_relaxed);
    return true;
}();

namespace pk {
long test_hook(int which) { return which >= 0 && which < PK_HOOK_COUNT ? g_test_hooks[which].load(std::memory_order_relaxed) : 0; }
// PK_WAIT_POLL: a few immediate queries (work that ends within microseconds), then sleeps that lengthen from 20 us to 100 us.  A waiting
// thread costs a query and a timer wake-up per interval instead of a core (spinning) or the runtime's spin-then-block (~0.2 ms of spinning
// per wait: 65 waits per proof); it learns of completion up to one interval late, which a prover that shares the chip with others does not
// notice and a lone prover does (keep PK_WAIT_SPIN for latency).
hipError_t wait_stream(int device, hipStream_t stream) {
    if (device < 0 || device >= PK_MAX_DEVICES || !g_poll_wait[device].load(std::memory_order_relaxed)) return hipStreamSynchronize(stream);
    long slack = -1;  // the calling thread's timer slack (default 50 us) would round every short sleep up: lowered for the sleeps, then put back
    hipError_t e;
    for (unsigned polls = 0;; polls++) {
        e = hipStreamQuery(stream);
        if (e != hipErrorNotReady) break;
        if (polls < 4) continue;
        if (slack < 0) {
            slack = prctl(PR_GET_TIMERSLACK, 0, 0, 0, 0);
            if (slack > 2000) (void)prctl(PR_SET_TIMERSLACK, 2000UL, 0, 0, 0);
        }
        const long ns = polls < 12 ? 20000 : (polls < 40 ? 50000 : 100000);
        struct timespec ts = {0, ns};
        (void)nanosleep(&ts, nullptr);
    }
    if (slack > 2000) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)slack, 0, 0, 0);
    return e;
}
hipError_t wait_ctx(pk_ctx* ctx) { return comm_collective_pending(ctx) ? comm_wait(ctx) : wait_stream(ctx->device, ctx->stream); }
int wait_ctx_rc(pk_ctx* ctx) {
    const bool collective = comm_collective_pending(ctx);
    const hipError_t e = wait_ctx(ctx);
    if (e == hipSuccess) return PK_OK;
    if (collective && ctx->err_code == PK_ERR_RCCL && e == hipErrorUnknown) return PK_ERR_RCCL;  // comm_wait gave up on the collective and said why
    return set_err(ctx, e == hipErrorOutOfMemory ? PK_ERR_OOM : PK_ERR_HIP, "waiting for the stream failed: %s", hipGetErrorString(e));
}
}  // namespace pk

extern "C" {

int pk_abi_version(void) { return 2; }

// test infrastructure (tools/probes/pk_selftest.h): the hooks the GPU suite uses to reach rare paths; nothing reads the environment
int pk_selftest_set_hook(int which, long value) {
    if (which < 0 || which >= pk::PK_HOOK_COUNT || value < 0) return PK_ERR_BAD_ARG;
    g_test_hooks[which].store(value, std::memory_order_relaxed);
    return PK_OK;
}

int pk_device_count(int* n) {
    if (!n) return PK_ERR_BAD_ARG;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *n = 0;
        return PK_ERR_NO_DEVICE;
    }
    *n = c;
    return PK_OK;
}

int pk_device_set_host_wait(int device, int mode) {
    if (mode != PK_WAIT_SPIN && mode != PK_WAIT_BLOCK && mode != PK_WAIT_POLL) return PK_ERR_BAD_ARG;
    if (device < 0 || device >= PK_MAX_DEVICES) return PK_ERR_BAD_ARG;
    if (mode == PK_WAIT_POLL) {  // the library's own loop: nothing of the runtime changes, so this one may be switched at any time
        g_poll_wait[device].store(1, std::memory_order_relaxed);
        return PK_OK;
    }
    g_poll_wait[device].store(0, std::memory_order_relaxed);
    // The runtime's own modes.  Its flag is touched only when it has to change: PK_WAIT_SPIN on a device whose flag this library never
    // set (HIP's default, hipDeviceScheduleAuto: spin, then yield) just leaves polling -- safe with work in flight, unlike a flag change
    // (a wait that blocks on a signal created for polling never wakes: choose PK_WAIT_BLOCK before the device has streams).
    const int cur_flag = g_runtime_wait[device].load(std::memory_order_relaxed);
    if (cur_flag == mode || (mode == PK_WAIT_SPIN && cur_flag < 0)) return PK_OK;
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    if (hipSetDevice(device) != hipSuccess) return PK_ERR_BAD_ARG;
    const hipError_t e = hipSetDeviceFlags(mode == PK_WAIT_BLOCK ? hipDeviceScheduleBlockingSync : hipDeviceScheduleSpin);
    if (have) (void)hipSetDevice(cur);
    if (e != hipSuccess) return PK_ERR_HIP;
    g_runtime_wait[device].store(mode, std::memory_order_relaxed);
    return PK_OK;
}

int pk_ctx_create(int device, pk_ctx** out) {
    if (!out) return PK_ERR_BAD_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return PK_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return PK_ERR_BAD_ARG;
    {   // PK_HOST_WAIT=spin|block|poll: pk_device_set_host_wait for callers that cannot reach the API (A/B runs, tools/cpuuse.py); applied
        // BEFORE this context's stream exists (a blocking wait must not meet a stream created for spinning)
        const char* w = getenv("PK_HOST_WAIT");
        if (w && *w) {
            const int mode = !strcmp(w, "spin") ? PK_WAIT_SPIN : (!strcmp(w, "block") ? PK_WAIT_BLOCK : (!strcmp(w, "poll") ? PK_WAIT_POLL : -1));
            if (mode < 0) return PK_ERR_BAD_ARG;  // an unknown value is refused, not read as "spin"
            const int rc = pk_device_set_host_wait(device, mode);
            if (rc) return rc;
        }
    }
    pk_ctx* ctx = new (std::nothrow) pk_ctx();
    if (!ctx) return PK_ERR_OOM;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev_start) != hipSuccess || hipEventCreate(&ctx->ev_stop) != hipSuccess) {
        delete ctx;
        return PK_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    pk::ntt_retain_ctx(ctx);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        ctx->num_cus = prop.multiProcessorCount;
    *out = ctx;
    return PK_OK;
}

int pk_ctx_destroy(pk_ctx* ctx) {
    PK_ENTER(ctx);
    (void)hipSetDevice(ctx->device);
    (void)wait_ctx(ctx);
    pk::comm_release(ctx);
    pk::ntt_release_ctx(ctx);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->d_xred) (void)hipFree(ctx->d_xred);
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);
    if (ctx->d_ztab) (void)hipFree(ctx->d_ztab);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    if (ctx->h_mail) (void)hipHostFree(ctx->h_mail);
    for (auto& r : ctx->prof) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return PK_OK;
}

const char* pk_last_error(const pk_ctx* ctx) { return ctx ? ctx->err : "null context"; }

int pk_ctx_set_stream(pk_ctx* ctx, void* hip_stream) {
    PK_ENTER(ctx);
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return PK_OK;
}

int pk_ctx_sync(pk_ctx* ctx) {
    PK_ENTER(ctx);
    return sync_stream(ctx);
}

int pk_ctx_set_hash_version(pk_ctx* ctx, int version) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, version == 1 || version == 2, "hash version must be 1 or 2");
    ctx->hash_version = version;
    return PK_OK;
}

int pk_malloc(pk_ctx* ctx, size_t bytes, void** d_ptr) {
    if (!ctx || !d_ptr) return PK_ERR_BAD_ARG;
    *d_ptr = nullptr;
    PK_HIP(ctx, hipSetDevice(ctx->device));
    PK_HIP(ctx, hipMalloc(d_ptr, bytes ? bytes : 1));
    return PK_OK;
}
int pk_free(pk_ctx* ctx, void* d_ptr) {
    PK_ENTER(ctx);
    if (!d_ptr) return PK_OK;
    PK_WAIT(ctx);
    PK_HIP(ctx, hipFree(d_ptr));
    return PK_OK;
}
int pk_memcpy_h2d(pk_ctx* ctx, void* d_dst, const void* src, size_t bytes) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, bytes == 0 || (d_dst && src), "null pointer");
    if (!bytes) return PK_OK;
    PK_HIP(ctx, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    PK_WAIT(ctx);
    return PK_OK;
}
int pk_memcpy_d2h(pk_ctx* ctx, void* dst, const void* d_src, size_t bytes) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, bytes == 0 || (dst && d_src), "null pointer");
    if (!bytes) return PK_OK;
    PK_HIP(ctx, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PK_WAIT(ctx);
    return PK_OK;
}
int pk_memcpy_d2d(pk_ctx* ctx, void* d_dst, const void* d_src, size_t bytes) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, bytes == 0 || (d_dst && d_src), "null pointer");
    if (!bytes) return PK_OK;
    PK_HIP(ctx, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return PK_OK;
}
int pk_memset_zero(pk_ctx* ctx, void* d_dst, size_t bytes) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, bytes == 0 || d_dst, "null pointer");
    if (!bytes) return PK_OK;
    PK_HIP(ctx, hipMemsetAsync(d_dst, 0, bytes, ctx->stream));
    return PK_OK;
}
int pk_timer_start(pk_ctx* ctx) {
    PK_ENTER(ctx);
    PK_HIP(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    return PK_OK;
}
int pk_timer_stop(pk_ctx* ctx, float* ms) {
    if (!ctx || !ms) return PK_ERR_BAD_ARG;
    PK_ENTER(ctx);
    PK_HIP(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
    PK_HIP(ctx, hipEventSynchronize(ctx->ev_stop));
    PK_HIP(ctx, hipEventElapsedTime(ms, ctx->ev_start, ctx->ev_stop));
    return PK_OK;
}

int pk_profile_enable(pk_ctx* ctx, int on) {
    PK_ENTER(ctx);
    ctx->prof_on = on != 0;
    return PK_OK;
}
int pk_profile_reset(pk_ctx* ctx) {
    PK_ENTER(ctx);
    PK_WAIT(ctx);
    for (auto& r : ctx->prof) {
        ctx->ev_pool.push_back(r.e0);
        ctx->ev_pool.push_back(r.e1);
    }
    ctx->prof.clear();
    return PK_OK;
}
int pk_profile_read(pk_ctx* ctx, const char* name, uint64_t* launches, double* total_ms) {
    if (!ctx || !name || !launches || !total_ms) return PK_ERR_BAD_ARG;
    PK_WAIT(ctx);
    *launches = 0;
    *total_ms = 0.0;
    for (auto& r : ctx->prof) {
        if (strcmp(r.name, name) != 0) continue;
        float ms = 0.f;
        PK_HIP(ctx, hipEventElapsedTime(&ms, r.e0, r.e1));
        *launches += 1;
        *total_ms += ms;
    }
    return PK_OK;
}
int pk_profile_names(pk_ctx* ctx, char* buf, size_t cap) {
    if (!ctx || !buf || !cap) return PK_ERR_BAD_ARG;
    std::string s;
    std::vector<const char*> seen;
    for (auto& r : ctx->prof) {
        bool dup = false;
        for (auto p : seen) dup |= strcmp(p, r.name) == 0;
        if (dup) continue;
        seen.push_back(r.name);
        if (!s.empty()) s += ",";
        s += r.name;
    }
    snprintf(buf, cap, "%s", s.c_str());
    return PK_OK;
}

int pk_ctx_set_latency_mode(pk_ctx* ctx, int on) {
    if (!ctx) return PK_ERR_BAD_ARG;
    ctx->latency_mode = on != 0;
    return PK_OK;
}

}  // extern "C"

namespace pk {
int ensure_scratch(pk_ctx* ctx, size_t bytes) {
    if (ctx->scratch_bytes >= bytes) return PK_OK;
    if (ctx->d_scratch) {
        PK_WAIT(ctx);
        PK_HIP(ctx, hipFree(ctx->d_scratch));
        ctx->d_scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    size_t sz = bytes < (1u << 20) ? (1u << 20) : bytes;
    PK_HIP(ctx, hipMalloc(&ctx->d_scratch, sz));
    ctx->scratch_bytes = sz;
    ctx->red_armed = ctx->pow_armed = false;  // the ticket / best words live in this buffer
    return PK_OK;
}
int ensure_pinned(pk_ctx* ctx) {
    if (ctx->h_pinned) return PK_OK;
    PK_HIP(ctx, hipHostMalloc(&ctx->h_pinned, 4096, hipHostMallocMapped | hipHostMallocCoherent));
    memset(ctx->h_pinned, 0, 4096);
    ctx->pinned_bytes = 4096;
    return PK_OK;
}
int sync_stream(pk_ctx* ctx) {
    PK_WAIT(ctx);
    ctx->mail_off = 0;  // nothing in flight reads or writes the mailbox any more
    return PK_OK;
}
int mail_alloc(pk_ctx* ctx, size_t bytes, void** out) {
    bytes = (bytes + 63) & ~(size_t)63;
    if (ctx->mail_off + bytes > ctx->mail_bytes) {
        int rc = sync_stream(ctx);  // in-flight kernels may still use earlier allocations
        if (rc) return rc;
        if (bytes > ctx->mail_bytes) {
            size_t cap = (size_t)1 << 20;
            while (cap < bytes) cap <<= 1;
            if (ctx->h_mail) PK_HIP(ctx, hipHostFree(ctx->h_mail));
            ctx->h_mail = nullptr;
            ctx->mail_bytes = 0;
            PK_HIP(ctx, hipHostMalloc((void**)&ctx->h_mail, cap, hipHostMallocMapped | hipHostMallocCoherent));
            ctx->mail_bytes = cap;
        }
    }
    *out = ctx->h_mail + ctx->mail_off;
    ctx->mail_off += bytes;
    return PK_OK;
}
int ensure_ws(pk_ctx* ctx, size_t bytes) {
    if (ctx->ws_bytes >= bytes) return PK_OK;
    if (ctx->d_ws) {
        PK_WAIT(ctx);
        PK_HIP(ctx, hipFree(ctx->d_ws));
        ctx->d_ws = nullptr;
        ctx->ws_bytes = 0;
    }
    PK_HIP(ctx, hipMalloc(&ctx->d_ws, bytes));
    ctx->ws_bytes = bytes;
    return PK_OK;
}
}  // namespace pk

// ------------------------------------------------------------------ elementwise field kernels
enum { OP_ADD, OP_SUB, OP_MUL, OP_TO_MONT, OP_FROM_MONT };

template <int OP>
__global__ __launch_bounds__(256) void fe_elementwise_kernel(const fe* __restrict__ a, const fe* __restrict__ b,
                                                             fe* __restrict__ out, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        fe x = fe_load(a + i);
        fe r;
        if (OP == OP_ADD) r = fe_add(x, fe_load(b + i));
        if (OP == OP_SUB) r = fe_sub(x, fe_load(b + i));
        if (OP == OP_MUL) r = fe_mulx(x, fe_load(b + i));
        if (OP == OP_TO_MONT) r = fe_to_montx(fe_reduce_any(x));
        if (OP == OP_FROM_MONT) r = fe_from_montx(x);
        fe_store(out + i, r);
    }
}

template <int OP>
static int launch_elementwise(pk_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    PK_ENTER(ctx);
    PK_REQUIRE(ctx, n == 0 || (a && out && (b || OP >= OP_TO_MONT)), "null pointer");
    if (!n) return PK_OK;
    fe_elementwise_kernel<OP><<<grid_for(ctx, n, 256), 256, 0, ctx->stream>>>((const fe*)a, (const fe*)b, (fe*)out, n);
    PK_LAUNCH_CHECK(ctx);
    return PK_OK;
}

extern "C" {
int pk_fe_add(pk_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) { return launch_elementwise<OP_ADD>(ctx, a, b, o, n); }
int pk_fe_sub(pk_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) { return launch_elementwise<OP_SUB>(ctx, a, b, o, n); }
int pk_fe_mul(pk_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) { return launch_elementwise<OP_MUL>(ctx, a, b, o, n); }
int pk_fe_to_mont(pk_ctx* ctx, const uint64_t* a, uint64_t* o, size_t n) { return launch_elementwise<OP_TO_MONT>(ctx, a, nullptr, o, n); }
int pk_fe_from_mont(pk_ctx* ctx, const uint64_t* a, uint64_t* o, size_t n) { return launch_elementwise<OP_FROM_MONT>(ctx, a, nullptr, o, n); }
}
