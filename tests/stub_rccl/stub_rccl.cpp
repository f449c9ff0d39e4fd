// stub_rccl.cpp -- TEST INFRASTRUCTURE, never shipped: an in-process stand-in for librccl that lets csrc/comm.hip's RCCL
// branch run on a box with ONE GPU (real RCCL refuses two ranks on one device).  Selected with PK_RCCL_LIB=<this .so>
// (+ PK_RCCL_SAME_DEVICE=1 so that pk_ctx_create_set takes ncclCommInitAll for a repeated device).
//
// What it is for: everything comm.hip does on the RCCL side of its `kind` switch -- the dlsym'd entry points, counts given in
// ELEMENTS of the stated datatype (not bytes), the in-place all-reduce (send == recv), collectives ordered on the caller's
// stream, ncclCommInitAll / ncclGetUniqueId + ncclCommInitRank, ncclCommAbort -- so that a sharded commit / opening / proof
// driven through THAT branch can be compared with the lone prover's transcript (tests/test_gpu_rccl_stub.py).
// What it is not: a transport.  Ranks are threads of one process; a collective drains the caller's stream, meets its peers at
// a host barrier and copies device to device.  It also RECORDS every call (ncclStubCalls) so a test can assert what the
// library asked for.
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace {

constexpr int kMaxRanks = 64;
// how long a rank waits for its peers before the collective fails: the real library would wait for ever (a collective kernel spinning on
// xGMI), the stand-in gives up so that a test cannot hang.  PK_STUB_RCCL_TIMEOUT_S shortens it for the tests that make a rank stay away.
static const auto kTimeout = std::chrono::milliseconds([] {
    const char* e = getenv("PK_STUB_RCCL_TIMEOUT_S");
    const double v = e ? atof(e) : 0.0;
    return (long long)(1000.0 * (v > 0.0 ? v : 120.0));
}());

struct Group {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0, joined = 0, refs = 0;
    unsigned long long generation = 0;
    bool aborted = false;
    const void* send[kMaxRanks] = {};
    size_t bytes[kMaxRanks] = {};
    // false: aborted or timed out (a peer never came)
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const unsigned long long gen = generation;
        if (++arrived == world) {
            arrived = 0;
            generation++;
            cv.notify_all();
            return true;
        }
        if (!cv.wait_for(lk, kTimeout, [&] { return generation != gen || aborted; })) aborted = true;
        return !aborted;
    }
};

struct Comm {
    Group* grp;
    int rank, world, device;
    unsigned* hang_flag = nullptr;  // set while a "collective" of this communicator hangs on the stream (ncclStubHangNext)
};
int g_hang_next = 0;  // the next that-many collectives are enqueued as a stream wait nobody satisfies -- until the communicator is aborted

std::mutex g_mu;
std::map<unsigned long long, Group*> g_groups;  // by unique id
unsigned long long g_next_id = 1;
struct Counters {
    unsigned long long all_gather = 0, all_reduce = 0, init_rank = 0, init_all = 0, abort = 0, destroy = 0, bytes_gathered = 0, in_place_reduce = 0;
} g_calls;

size_t dtype_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8:
        case ncclUint8: return 1;
        case ncclFloat16: return 2;
        case ncclInt32:
        case ncclUint32:
        case ncclFloat32: return 4;
        case ncclInt64:
        case ncclUint64:
        case ncclFloat64: return 8;
        default: return 0;
    }
}

Comm* as_comm(ncclComm_t c) { return reinterpret_cast<Comm*>(c); }

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int* v) {
    if (!v) return ncclInvalidArgument;
    *v = 9990000;  // recognisable: no RCCL release carries it
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error (stub)";
        case ncclInvalidArgument: return "invalid argument (stub)";
        case ncclInvalidUsage: return "invalid usage (stub)";
        case ncclInternalError: return "a rank of the group aborted or never arrived (stub)";
        case ncclUnhandledCudaError: return "HIP error (stub)";
        default: return "error (stub)";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    memset(id->internal, 0, sizeof id->internal);
    const unsigned long long v = g_next_id++;
    memcpy(id->internal, "PKSTUBID", 8);
    memcpy(id->internal + 8, &v, 8);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
    if (!out || world < 1 || world > kMaxRanks || rank < 0 || rank >= world) return ncclInvalidArgument;
    if (memcmp(id.internal, "PKSTUBID", 8) != 0) return ncclInvalidArgument;  // an id that did not come from ncclGetUniqueId
    unsigned long long key;
    memcpy(&key, id.internal + 8, 8);
    Group* g;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_calls.init_rank++;
        auto it = g_groups.find(key);
        if (it == g_groups.end()) {
            g = new Group();
            g->world = world;
            g_groups[key] = g;
        } else {
            g = it->second;
        }
        if (g->world != world) return ncclInvalidArgument;
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ncclUnhandledCudaError;
    {
        std::unique_lock<std::mutex> lk(g->mu);  // like the real call: returns once every rank of the communicator has joined
        g->joined++;
        g->refs++;
        g->cv.notify_all();
        if (!g->cv.wait_for(lk, kTimeout, [&] { return g->joined >= world || g->aborted; }) || g->aborted) {
            g->aborted = true;
            return ncclInternalError;
        }
    }
    *out = reinterpret_cast<ncclComm_t>(new Comm{g, rank, world, dev});
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int* devices) {
    if (!comms || n < 1 || n > kMaxRanks) return ncclInvalidArgument;
    Group* g = new Group();
    g->world = n;
    g->joined = n;
    g->refs = n;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_calls.init_all++;
        g_groups[g_next_id++] = g;
    }
    for (int i = 0; i < n; i++) comms[i] = reinterpret_cast<ncclComm_t>(new Comm{g, i, n, devices ? devices[i] : i});
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    g_calls.destroy++;
    delete as_comm(c);  // groups are leaked on purpose: a few hundred bytes per test communicator, no teardown races
    return ncclSuccess;
}

// LOCAL, like the real call: this rank's communicator is torn down, nothing reaches the peers -- a rank waiting for this one in a
// collective keeps waiting (here: until the stand-in's timeout; with the real library: until its own deadline aborts it, comm.hip)
ncclResult_t ncclCommAbort(ncclComm_t c) {
    if (!c) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    g_calls.abort++;
    if (as_comm(c)->hang_flag) __atomic_store_n(as_comm(c)->hang_flag, 1u, __ATOMIC_RELEASE);  // like the real call: it ends this rank's own stuck kernel
    delete as_comm(c);  // (the flag's page is leaked: the stream may still be reading it)
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t stream) {
    Comm* c = as_comm(comm);
    const size_t es = dtype_size(dt);
    if (!c || !es || (count && (!send || !recv))) return ncclInvalidArgument;
    const size_t bytes = count * es;
    Group* g = c->grp;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_hang_next > 0) {  // a collective a peer never joins, the way the real library shows it: enqueued fine, then the stream never gets past it
            g_hang_next--;
            if (!c->hang_flag) {
                if (hipHostMalloc((void**)&c->hang_flag, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return ncclUnhandledCudaError;
                *c->hang_flag = 0;
            }
            if (hipStreamWaitValue32(stream, c->hang_flag, 1u, hipStreamWaitValueEq, 0xffffffffu) != hipSuccess) return ncclUnhandledCudaError;
            return ncclSuccess;
        }
    }
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;  // everything that produced `send` is done
    g->send[c->rank] = send;
    g->bytes[c->rank] = bytes;
    if (!g->barrier()) return ncclInternalError;
    for (int p = 0; p < c->world; p++) {
        if (g->bytes[p] != bytes) return ncclInvalidArgument;  // ranks disagree about the count: the real library would hang or corrupt
        char* dst = static_cast<char*>(recv) + (size_t)p * bytes;
        if (bytes && dst != g->send[p] && hipMemcpyAsync(dst, g->send[p], bytes, hipMemcpyDefault, stream) != hipSuccess) return ncclUnhandledCudaError;
    }
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (!g->barrier()) return ncclInternalError;  // every rank has read every send buffer
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_calls.all_gather++;
        g_calls.bytes_gathered += bytes * (size_t)c->world;
    }
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    Comm* c = as_comm(comm);
    if (!c || (count && (!send || !recv))) return ncclInvalidArgument;
    if (dt != ncclUint64 || op != ncclSum) return ncclInvalidUsage;  // the one reduction the library asks for
    Group* g = c->grp;
    const size_t bytes = count * 8;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    g->send[c->rank] = send;
    g->bytes[c->rank] = bytes;
    if (!g->barrier()) return ncclInternalError;
    std::vector<unsigned long long> acc(count, 0), part(count);
    for (int p = 0; p < c->world; p++) {
        if (g->bytes[p] != bytes) return ncclInvalidArgument;
        if (bytes && hipMemcpy(part.data(), g->send[p], bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        for (size_t i = 0; i < count; i++) acc[i] += part[i];
    }
    if (!g->barrier()) return ncclInternalError;  // everyone has read everyone's input: in-place outputs may be written now
    if (bytes && hipMemcpy(recv, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    if (!g->barrier()) return ncclInternalError;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_calls.all_reduce++;
        if (send == recv) g_calls.in_place_reduce++;
    }
    return ncclSuccess;
}

// test hook: the next n all-gathers hang on their stream (until their communicator is aborted), as a collective does whose peer never comes
void ncclStubHangNext(int n) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_hang_next = n;
}

// test hook: what the library asked of "RCCL" so far
void ncclStubCalls(unsigned long long out[8]) {
    std::lock_guard<std::mutex> lk(g_mu);
    out[0] = g_calls.all_gather;
    out[1] = g_calls.all_reduce;
    out[2] = g_calls.init_rank;
    out[3] = g_calls.init_all;
    out[4] = g_calls.abort;
    out[5] = g_calls.destroy;
    out[6] = g_calls.bytes_gathered;
    out[7] = g_calls.in_place_reduce;
}

}  // extern "C"
