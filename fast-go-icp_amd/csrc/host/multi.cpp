// Multi-GPU part of the C ABI (include/fgoicp_amd.h): the exchange of the sharded outer branch-and-bound inside the library.
//
//   fgoicp_rccl_*    one communicator per rank (ncclCommInitRank: one process per GPU, the id comes from rank 0 by whatever
//                    channel the launcher has; or one host thread per GPU in one process) — per expansion round ONE
//                    ncclAllReduce(ncclMin) of the best error and ONE small ncclAllGather on device buffers, over xGMI.
//   fgoicp_multi_*   one process, one host thread + one solver per device: what `fast-go-icp --gpus N` runs.
//                    Transport RCCL (distinct devices) or an in-process rendezvous (any device list, e.g. {0, 0, 0, 0} to
//                    rehearse four ranks on one GPU); optional recording of what every exchange returned and replay of ONE
//                    rank alone against the recording (tools/scale_replay.py: strong-scaling estimate on one GPU).
//
// The reference is single-GPU (no NCCL/MPI call sites); nothing here replaces reference code.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is loaded on first use (RcclApi below), not linked
#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../../include/fgoicp_amd.h"
#include "abi_guard.hpp"

namespace fgoicp {
void set_error(const std::string& s);
int ctx_icp_coop(fgoicp_ctx* c, int rank, int world, int (*gather)(void* dev_buf, size_t bytes_per_rank, void* user), void* user, const float* R0,
                 const float* t0, size_t max_iter, float thr, float* sse_out, float* R_out9, float* t_out3, int* iters_out);  // csrc/device/ctx.hip
}
using fgoicp::set_error;

// ---------------------------------------------------------------------------------------------------------------------
// RCCL transport
// ---------------------------------------------------------------------------------------------------------------------
struct fgoicp_rccl {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    float *d_send = nullptr, *d_recv = nullptr;   // device buffers of the collectives
    float* h_pin = nullptr;                       // pinned staging (send | recv)
    size_t cap = 0;                               // floats per rank the buffers hold
    int rank = 0, world = 1, device = 0;
    uint64_t calls = 0;
    // Failure handling.  `dead` may be set by ANY thread (fgoicp_rccl_abort: another rank has failed); the communicator itself is only
    // ever touched by its owner — the thread that runs this rank's collectives — which polls `dead` while it waits for a collective
    // and then calls ncclCommAbort itself (`released`).  (Round 2 aborted from the peer's thread: ncclCommAbort frees the
    // communicator, and the owner could be between its check and its ncclAllReduce — a use after free.)
    std::atomic<bool> dead{false};
    bool released = false;                        // owner only: the communicator has been aborted / destroyed
};

namespace {

// RCCL is needed by the multi-GPU entry points only, so libfgoicp_amd.so does not link it: a single-GPU installation without
// RCCL loads and runs; the first fgoicp_rccl_* call resolves the handful of entry points it uses.
struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitRankConfig) CommInitRankConfig = nullptr;
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
    bool ok = false;
};
const RcclApi& rccl_api() {
    static const RcclApi api = [] {
        RcclApi a;
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) { a.error = std::string("RCCL is not installed (dlopen librccl.so.1: ") + dlerror() + ")"; return a; }
        bool all = true;
        auto sym = [&](auto& fn, const char* name) { fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(h, name)); all = all && fn != nullptr; };
        sym(a.GetUniqueId, "ncclGetUniqueId"); sym(a.CommInitRank, "ncclCommInitRank"); sym(a.CommInitRankConfig, "ncclCommInitRankConfig");
        sym(a.CommGetAsyncError, "ncclCommGetAsyncError"); sym(a.CommDestroy, "ncclCommDestroy"); sym(a.CommAbort, "ncclCommAbort");
        sym(a.CommCount, "ncclCommCount"); sym(a.AllReduce, "ncclAllReduce"); sym(a.AllGather, "ncclAllGather"); sym(a.GetErrorString, "ncclGetErrorString");
        if (!all) { a.error = "librccl.so lacks an entry point this library uses"; return a; }
        a.ok = true;
        return a;
    }();
    return api;
}

#define RCCL_HIP(expr)                                                                                     \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) { set_error(std::string(#expr) + " failed: " + hipGetErrorString(e_)); return 1; } \
    } while (0)
#define RCCL_NCCL(expr)                                                                                    \
    do {                                                                                                   \
        ncclResult_t r_ = (expr);                                                                          \
        if (r_ != ncclSuccess) { set_error(std::string(#expr) + " failed: " + rccl_api().GetErrorString(r_)); return 1; } \
    } while (0)

int rccl_reserve(fgoicp_rccl* x, size_t n) {
    if (n <= x->cap) return 0;
    size_t cap = x->cap ? x->cap : 64;
    while (cap < n) cap *= 2;
    (void)hipFree(x->d_send); (void)hipFree(x->d_recv);
    if (x->h_pin) (void)hipHostFree(x->h_pin);
    x->d_send = x->d_recv = x->h_pin = nullptr;
    x->cap = 0;
    RCCL_HIP(hipMalloc(&x->d_send, sizeof(float) * cap));
    RCCL_HIP(hipMalloc(&x->d_recv, sizeof(float) * cap * x->world));
    RCCL_HIP(hipHostMalloc((void**)&x->h_pin, sizeof(float) * cap * (x->world + 1), hipHostMallocDefault));
    x->cap = cap;
    return 0;
}

// Owner only: give the communicator up (idempotent).  After it every collective of this rank fails at once.
void rccl_release(fgoicp_rccl* x) {
    if (x->released) return;
    x->released = true;
    if (x->comm) (void)rccl_api().CommAbort(x->comm);
    x->comm = nullptr;
}
int rccl_dead(fgoicp_rccl* x) {
    rccl_release(x);
    set_error("exchange aborted: another rank failed");
    return 1;
}
// Waits for the stream without blocking in the runtime: the collective of a rank whose peer has died never completes, so the
// owner polls — stream, `dead`, the communicator's asynchronous error — and aborts its own communicator when it has to.
int rccl_wait(fgoicp_rccl* x) {
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = hipStreamQuery(x->stream);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) { set_error(std::string("hipStreamQuery failed: ") + hipGetErrorString(q)); rccl_release(x); return 1; }
        if (x->dead.load(std::memory_order_acquire)) return rccl_dead(x);
        if ((spins & 255u) == 255u) {
            ncclResult_t async = ncclSuccess;
            if (rccl_api().CommGetAsyncError(x->comm, &async) != ncclSuccess || (async != ncclSuccess && async != ncclInProgress)) {
                set_error(std::string("RCCL reported an asynchronous error: ") + rccl_api().GetErrorString(async));
                rccl_release(x);
                return 1;
            }
            std::this_thread::yield();
        }
    }
}

int rccl_allreduce_min(float* buf, size_t n, void* user) {
    fgoicp_rccl* x = static_cast<fgoicp_rccl*>(user);
    if (x->released || x->dead.load(std::memory_order_acquire)) return rccl_dead(x);
    RCCL_HIP(hipSetDevice(x->device));
    if (rccl_reserve(x, n)) return 1;
    std::memcpy(x->h_pin, buf, sizeof(float) * n);
    RCCL_HIP(hipMemcpyAsync(x->d_send, x->h_pin, sizeof(float) * n, hipMemcpyHostToDevice, x->stream));
    RCCL_NCCL(rccl_api().AllReduce(x->d_send, x->d_recv, n, ncclFloat, ncclMin, x->comm, x->stream));
    RCCL_HIP(hipMemcpyAsync(x->h_pin, x->d_recv, sizeof(float) * n, hipMemcpyDeviceToHost, x->stream));
    if (rccl_wait(x)) return 1;
    std::memcpy(buf, x->h_pin, sizeof(float) * n);
    x->calls++;
    return 0;
}

int rccl_allgather(const float* send, float* recv, size_t n, void* user) {
    fgoicp_rccl* x = static_cast<fgoicp_rccl*>(user);
    if (x->released || x->dead.load(std::memory_order_acquire)) return rccl_dead(x);
    RCCL_HIP(hipSetDevice(x->device));
    if (rccl_reserve(x, n)) return 1;
    std::memcpy(x->h_pin, send, sizeof(float) * n);
    RCCL_HIP(hipMemcpyAsync(x->d_send, x->h_pin, sizeof(float) * n, hipMemcpyHostToDevice, x->stream));
    RCCL_NCCL(rccl_api().AllGather(x->d_send, x->d_recv, n, ncclFloat, x->comm, x->stream));
    RCCL_HIP(hipMemcpyAsync(x->h_pin + x->cap, x->d_recv, sizeof(float) * n * x->world, hipMemcpyDeviceToHost, x->stream));
    if (rccl_wait(x)) return 1;
    std::memcpy(recv, x->h_pin + x->cap, sizeof(float) * n * x->world);
    x->calls++;
    return 0;
}

// In place on the caller's device memory (cooperative ICP: 4 B per source point, twice per iteration): chunk `rank` of buf is this
// rank's contribution.  The caller's stream is idle; ours is waited for before returning.
int rccl_allgather_device(void* buf, size_t bytes, void* user) {
    fgoicp_rccl* x = static_cast<fgoicp_rccl*>(user);
    if (x->released || x->dead.load(std::memory_order_acquire)) return rccl_dead(x);
    RCCL_HIP(hipSetDevice(x->device));
    RCCL_NCCL(rccl_api().AllGather(static_cast<char*>(buf) + bytes * (size_t)x->rank, buf, bytes, ncclInt8, x->comm, x->stream));
    if (rccl_wait(x)) return 1;
    x->calls++;
    return 0;
}

}  // namespace

extern "C" {

int fgoicp_rccl_unique_id(unsigned char* id128) {
    if (!id128) return FGOICP_ERR_INVALID_ARG;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    if (!rccl_api().ok) { set_error(rccl_api().error); return FGOICP_ERR_EXCHANGE; }
    ncclUniqueId id;
    ncclResult_t r = rccl_api().GetUniqueId(&id);
    if (r != ncclSuccess) { set_error(std::string("ncclGetUniqueId failed: ") + rccl_api().GetErrorString(r)); return FGOICP_ERR_EXCHANGE; }
    std::memcpy(id128, &id, 128);
    return FGOICP_OK;
}

void fgoicp_rccl_destroy(fgoicp_rccl* x) {
    if (!x) return;
    (void)hipSetDevice(x->device);
    // the caller is the owner now (no collective of this rank is running): a communicator marked dead by a peer and not yet given
    // up is aborted here, a healthy one is drained and destroyed
    if (x->dead.load() && !x->released) rccl_release(x);
    if (x->stream && !x->released) (void)hipStreamSynchronize(x->stream);
    if (x->comm && !x->released) (void)rccl_api().CommDestroy(x->comm);
    (void)hipFree(x->d_send); (void)hipFree(x->d_recv);
    if (x->h_pin) (void)hipHostFree(x->h_pin);
    if (x->stream) (void)hipStreamDestroy(x->stream);
    delete x;
}

// Everything of a rank's transport that can fail WITHOUT its peers: device, stream, buffers.  fgoicp_multi_create runs this for all
// ranks before any of them enters ncclCommInitRank (which blocks until every rank has joined), so a rank that cannot even set
// itself up does not leave the others waiting for it.
static int rccl_prepare(int rank, int world, int device, fgoicp_rccl** out) {
    *out = nullptr;
    if (!rccl_api().ok) { set_error(rccl_api().error); return FGOICP_ERR_EXCHANGE; }
    if (world < 1 || rank < 0 || rank >= world) { set_error("fgoicp_rccl_create: invalid argument"); return FGOICP_ERR_INVALID_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { set_error("fgoicp_rccl_create: no such HIP device"); return FGOICP_ERR_NO_DEVICE; }
    auto x = std::make_unique<fgoicp_rccl>();
    x->rank = rank; x->world = world; x->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("fgoicp_rccl_create: stream creation failed");
        return FGOICP_ERR_HIP;
    }
    if (rccl_reserve(x.get(), 64)) { fgoicp_rccl_destroy(x.release()); return FGOICP_ERR_HIP; }
    *out = x.release();
    return FGOICP_OK;
}
// ... and the part that needs all ranks: the communicator.  `give_up` (optional) is polled while the non-blocking initialisation is
// in progress — set by the caller when another rank of the same process has failed.
static int rccl_join(fgoicp_rccl* x, const unsigned char* id128, const std::atomic<bool>* give_up) {
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    if (hipSetDevice(x->device) != hipSuccess) { set_error("fgoicp_rccl_create: hipSetDevice failed"); return FGOICP_ERR_HIP; }
    ncclResult_t r;
    if (give_up) {
        ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
        cfg.blocking = 0;
        r = rccl_api().CommInitRankConfig(&x->comm, x->world, id, x->rank, &cfg);
        while (r == ncclInProgress || r == ncclSuccess) {
            ncclResult_t st = ncclSuccess;
            if (!x->comm || rccl_api().CommGetAsyncError(x->comm, &st) != ncclSuccess) { r = ncclInternalError; break; }
            if (st != ncclInProgress) { r = st; break; }
            if (give_up->load(std::memory_order_acquire)) { r = ncclInvalidUsage; set_error("communicator set-up abandoned: another rank failed"); break; }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    } else {
        r = rccl_api().CommInitRank(&x->comm, x->world, id, x->rank);
    }
    if (r != ncclSuccess) {
        if (!(give_up && give_up->load())) set_error(std::string("ncclCommInitRank failed: ") + rccl_api().GetErrorString(r));
        if (x->comm) rccl_release(x);
        x->released = true;
        return FGOICP_ERR_EXCHANGE;
    }
    return FGOICP_OK;
}

int fgoicp_rccl_create(int rank, int world, const unsigned char* id128, int device, fgoicp_rccl** out) {
    if (!out) return FGOICP_ERR_INVALID_ARG;
    *out = nullptr;
    if (!id128) { set_error("fgoicp_rccl_create: invalid argument"); return FGOICP_ERR_INVALID_ARG; }
    fgoicp_rccl* x = nullptr;
    int rc = rccl_prepare(rank, world, device, &x);  // one process per GPU: a rank that fails here is its launcher's to report
    if (rc) return rc;
    rc = rccl_join(x, id128, nullptr);
    if (rc) { fgoicp_rccl_destroy(x); return rc; }
    *out = x;
    return FGOICP_OK;
}

// Another rank has failed: every collective of this communicator that is in flight or follows must end instead of waiting for
// it.  Callable from any thread, once or more — it only raises a flag; the communicator's owner sees it (rccl_wait polls it) and
// aborts the communicator itself.
int fgoicp_rccl_abort(fgoicp_rccl* x) {
    if (!x) return FGOICP_ERR_INVALID_ARG;
    x->dead.store(true, std::memory_order_release);
    return FGOICP_OK;
}

// Ranks the communicator itself reports (ncclCommCount): what a scaling run prints to show that RCCL really joined N ranks.
int fgoicp_rccl_comm_count(fgoicp_rccl* x, int* count) {
    if (!x || !count) return FGOICP_ERR_INVALID_ARG;
    if (x->released || !x->comm) { set_error("fgoicp_rccl_comm_count: the communicator has been released"); return FGOICP_ERR_EXCHANGE; }
    ncclResult_t r = rccl_api().CommCount(x->comm, count);
    if (r != ncclSuccess) { set_error(std::string("ncclCommCount failed: ") + rccl_api().GetErrorString(r)); return FGOICP_ERR_EXCHANGE; }
    return FGOICP_OK;
}

int fgoicp_rccl_exchange(fgoicp_rccl* x, fgoicp_exchange* out) {
    if (!x || !out) return FGOICP_ERR_INVALID_ARG;
    out->rank = x->rank;
    out->world_size = x->world;
    out->allreduce_min = rccl_allreduce_min;
    out->allgather = rccl_allgather;
    out->user = x;
    out->allgather_device = rccl_allgather_device;
    return FGOICP_OK;
}

int fgoicp_rccl_calls(const fgoicp_rccl* x, uint64_t* calls) {
    if (!x || !calls) return FGOICP_ERR_INVALID_ARG;
    *calls = x->calls;
    return FGOICP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// One process, one host thread and one solver per device
// ---------------------------------------------------------------------------------------------------------------------
namespace {

// In-process rendezvous of `world` threads: min-all-reduce and all-gather through shared memory (two generations of buffers,
// so a fast rank may enter the next collective while a slow one still reads the last result).
struct Rendezvous {
    std::mutex m;
    std::condition_variable cv;
    int world = 1, arrived = 0;
    uint64_t gen = 0;
    bool aborted = false;  // a rank failed: nobody waits for it (reset by fgoicp_multi_run)
    std::vector<float> acc[2];
    void abort() {
        std::lock_guard<std::mutex> lk(m);
        aborted = true;
        cv.notify_all();
    }
    void reset() {
        std::lock_guard<std::mutex> lk(m);
        aborted = false;
        arrived = 0;
    }
    // false: aborted
    bool run(size_t total, const std::function<void(std::vector<float>&, bool first)>& contribute, const std::function<void(const std::vector<float>&)>& collect) {
        std::unique_lock<std::mutex> lk(m);
        if (aborted) return false;
        const uint64_t g = gen;
        std::vector<float>& a = acc[g & 1];
        const bool first = arrived == 0;
        if (first) a.assign(total, 0.f);
        contribute(a, first);
        if (++arrived == world) {
            arrived = 0;
            ++gen;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return gen != g || aborted; });
            if (gen == g) return false;
        }
        collect(a);
        return true;
    }
};

struct DeviceGather {    // in-process all-gather on device memory: where every rank's buffer lives (written before the first barrier)
    std::vector<void*> ptr;
    std::vector<int> device;
    // recorded gathers (whole buffers, kept in DEVICE memory by rank 0): what a replayed rank receives.  The replay leaves the
    // transfer out, like the replay of the host-side exchanges: its copies are on-device (microseconds); tools/scale_replay.py
    // charges the collective's measured software path and a modelled wire time per gather instead.
    struct Rec { void* d = nullptr; size_t bytes = 0; int device = 0; };
    std::vector<Rec> log;
    void clear_log() {
        for (Rec& r : log) { (void)hipSetDevice(r.device); (void)hipFree(r.d); }
        log.clear();
    }
    ~DeviceGather() { clear_log(); }
};

struct RankLink {        // what one rank's exchange callbacks see
    int rank = 0, world = 1;
    Rendezvous* rv = nullptr;
    DeviceGather* dg = nullptr;
    int device = 0;
    size_t dev_replay_pos = 0;
    fgoicp_exchange inner{};                      // transport underneath (RCCL) when rv == nullptr
    bool record = false;
    std::vector<std::vector<float>>* log = nullptr;   // results of every exchange, in order
    size_t replay_pos = 0;
    bool replay = false;
    long fail_at = -1, calls = 0;                 // test hook (fgoicp_multi_test_fault): that exchange of that rank fails, once
};

int link_allreduce_min(float* buf, size_t n, void* user) {
    RankLink* l = static_cast<RankLink*>(user);
    if (l->replay) {
        if (l->replay_pos >= l->log->size() || (*l->log)[l->replay_pos].size() != n) return 1;
        std::memcpy(buf, (*l->log)[l->replay_pos++].data(), sizeof(float) * n);
        return 0;
    }
    if (l->calls++ == l->fail_at) { l->fail_at = -1; set_error("injected exchange fault (fgoicp_multi_test_fault)"); return 1; }
    int rc = 0;
    if (l->rv) {
        rc = l->rv->run(n,
                        [&](std::vector<float>& a, bool first) { for (size_t i = 0; i < n; ++i) a[i] = first ? buf[i] : (buf[i] < a[i] ? buf[i] : a[i]); },
                        [&](const std::vector<float>& a) { std::memcpy(buf, a.data(), sizeof(float) * n); }) ? 0 : 1;
        if (rc) set_error("exchange aborted: another rank failed");
    } else {
        rc = l->inner.allreduce_min(buf, n, l->inner.user);
    }
    if (!rc && l->record) l->log->emplace_back(buf, buf + n);
    return rc;
}

int link_allgather(const float* send, float* recv, size_t n, void* user) {
    RankLink* l = static_cast<RankLink*>(user);
    if (l->replay) {
        if (l->replay_pos >= l->log->size() || (*l->log)[l->replay_pos].size() != n * (size_t)l->world) return 1;
        std::memcpy(recv, (*l->log)[l->replay_pos++].data(), sizeof(float) * n * l->world);
        return 0;
    }
    if (l->calls++ == l->fail_at) { l->fail_at = -1; set_error("injected exchange fault (fgoicp_multi_test_fault)"); return 1; }
    int rc = 0;
    if (l->rv) {
        rc = l->rv->run(n * (size_t)l->world,
                        [&](std::vector<float>& a, bool) { std::memcpy(a.data() + n * (size_t)l->rank, send, sizeof(float) * n); },
                        [&](const std::vector<float>& a) { std::memcpy(recv, a.data(), sizeof(float) * n * l->world); }) ? 0 : 1;
        if (rc) set_error("exchange aborted: another rank failed");
    } else {
        rc = l->inner.allgather(send, recv, n, l->inner.user);
    }
    if (!rc && l->record) l->log->emplace_back(recv, recv + n * (size_t)l->world);
    return rc;
}

// Cooperative ICP's all-gather.  In process: every rank publishes its buffer, waits for the others, copies their chunks into its
// own buffer (peer copies between devices, plain copies when the ranks share one), and waits again before anybody overwrites its
// chunk.  Recording keeps the gathered buffer (rank 0's copy; they are equal); a replayed rank, alone on the device, computes its own
// chunk and takes the others from the recording (on-device copies: the transfer itself is left out of a replay and charged by the caller).
int link_allgather_device(void* buf, size_t bytes, void* user) {
    RankLink* l = static_cast<RankLink*>(user);
    const size_t total = bytes * (size_t)l->world;
    if (l->replay) {
        if (!l->dg || l->dev_replay_pos >= l->dg->log.size() || l->dg->log[l->dev_replay_pos].bytes != total) return 1;
        const DeviceGather::Rec& rec = l->dg->log[l->dev_replay_pos++];
        const size_t lo = bytes * (size_t)l->rank, hi = lo + bytes;  // everything but this rank's own chunk: two contiguous ranges
        auto copy = [&](size_t from, size_t to) {
            if (from >= to) return hipSuccess;
            char* dst = static_cast<char*>(buf) + from;
            const char* src = static_cast<const char*>(rec.d) + from;
            return rec.device == l->device ? hipMemcpyAsync(dst, src, to - from, hipMemcpyDeviceToDevice, nullptr) : hipMemcpyPeerAsync(dst, l->device, src, rec.device, to - from, nullptr);
        };
        if (copy(0, lo) != hipSuccess || copy(hi, total) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) return 1;
        return 0;
    }
    if (l->calls++ == l->fail_at) { l->fail_at = -1; set_error("injected exchange fault (fgoicp_multi_test_fault)"); return 1; }
    int rc = 0;
    if (l->rv) {
        auto barrier = [&] { return l->rv->run(0, [](std::vector<float>&, bool) {}, [](const std::vector<float>&) {}); };
        l->dg->ptr[l->rank] = buf;
        l->dg->device[l->rank] = l->device;
        if (!barrier()) { set_error("exchange aborted: another rank failed"); return 1; }
        for (int p = 0; p < l->world && !rc; ++p) {
            if (p == l->rank) continue;
            char* dst = static_cast<char*>(buf) + bytes * p;
            const char* src = static_cast<const char*>(l->dg->ptr[p]) + bytes * p;
            const hipError_t e = l->dg->device[p] == l->device ? hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice) : hipMemcpyPeer(dst, l->device, src, l->dg->device[p], bytes);
            if (e != hipSuccess) { set_error(std::string("in-process device all-gather: ") + hipGetErrorString(e)); rc = 1; }
        }
        // device-to-device copies may return before they have landed, and the contexts' streams do not wait for the null stream
        if (!rc && hipStreamSynchronize(nullptr) != hipSuccess) { set_error("in-process device all-gather: hipStreamSynchronize failed"); rc = 1; }
        if (!barrier() && !rc) { set_error("exchange aborted: another rank failed"); rc = 1; }
    } else {
        if (!l->inner.allgather_device) { set_error("the transport has no device all-gather"); return 1; }
        rc = l->inner.allgather_device(buf, bytes, l->inner.user);
    }
    if (!rc && l->record && l->rank == 0 && l->dg) {
        DeviceGather::Rec rec;
        rec.bytes = total;
        rec.device = l->device;
        if (hipMalloc(&rec.d, total) != hipSuccess) { set_error("recording a device all-gather: out of device memory"); return 1; }
        if (hipMemcpyAsync(rec.d, buf, total, hipMemcpyDeviceToDevice, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) { (void)hipFree(rec.d); return 1; }
        l->dg->log.push_back(rec);
    }
    return rc;
}

}  // namespace

struct fgoicp_multi {
    ~fgoicp_multi();  // solvers and communicators go with the object (fgoicp_multi_destroy, a failed or throwing create)
    std::vector<int> devices;
    std::vector<fgoicp_solver*> solvers;
    std::vector<fgoicp_rccl*> rccl;
    std::vector<std::unique_ptr<RankLink>> links;
    std::vector<std::vector<std::vector<float>>> logs;
    Rendezvous rv;
    DeviceGather dg;
    int transport = FGOICP_TRANSPORT_RCCL;
    std::vector<double> seconds;   // wall-clock of every rank's last run
};
fgoicp_multi::~fgoicp_multi() {
    for (fgoicp_solver* s : solvers) fgoicp_solver_destroy(s);
    for (fgoicp_rccl* x : rccl) fgoicp_rccl_destroy(x);
}

extern "C" {

void fgoicp_multi_destroy(fgoicp_multi* m) { delete m; }

static int multi_create_impl(const float* tgt_xyz, size_t nt, const float* src_xyz, size_t ns, float lut_resolution, float mse_threshold, const fgoicp_solver_opts* opts,
                             const int* devices, int ndev, int transport, fgoicp_multi** out);
int fgoicp_multi_create(const float* tgt_xyz, size_t nt, const float* src_xyz, size_t ns, float lut_resolution, float mse_threshold,
                        const fgoicp_solver_opts* opts, const int* devices, int ndev, int transport, fgoicp_multi** out) {
    if (!out) return FGOICP_ERR_INVALID_ARG;
    *out = nullptr;
    // (the object under construction is held by a unique_ptr inside; fgoicp_multi's destructor frees what it owns)
    return fgoicp::abi_guard("fgoicp_multi_create", [&] { return multi_create_impl(tgt_xyz, nt, src_xyz, ns, lut_resolution, mse_threshold, opts, devices, ndev, transport, out); });
}
static int multi_create_impl(const float* tgt_xyz, size_t nt, const float* src_xyz, size_t ns, float lut_resolution, float mse_threshold, const fgoicp_solver_opts* opts,
                             const int* devices, int ndev, int transport, fgoicp_multi** out) {
    if (!devices || ndev < 1 || (transport != FGOICP_TRANSPORT_RCCL && transport != FGOICP_TRANSPORT_IN_PROCESS)) {
        set_error("fgoicp_multi_create: invalid argument");
        return FGOICP_ERR_INVALID_ARG;
    }
    auto m = std::make_unique<fgoicp_multi>();
    m->devices.assign(devices, devices + ndev);
    m->transport = transport;
    m->rv.world = ndev;
    m->logs.resize(ndev);
    m->dg.ptr.assign(ndev, nullptr);
    m->dg.device.assign(ndev, 0);
    m->seconds.assign(ndev, 0.0);
    fgoicp_solver_opts o{FGOICP_SCHEDULE_ROUND, 0, 0u, 0, 0.0f};
    if (opts) o = *opts;
    // both schedules shard: ROUND deals a round's children over the ranks, SERIAL (the reference's exact order) deals the tasks of
    // every speculative evaluation (driver.hpp: run_task_list_sharded); opts == NULL: ROUND, adaptive width
    auto fail = [&](int rc) { fgoicp_multi_destroy(m.release()); return rc; };
    for (int r = 0; r < ndev; ++r) {
        o.device = devices[r];
        fgoicp_solver* s = nullptr;
        int rc = fgoicp_solver_create(tgt_xyz, nt, src_xyz, ns, lut_resolution, mse_threshold, &o, &s);
        if (rc) return fail(rc);
        m->solvers.push_back(s);
    }
    if (ndev > 1 && transport == FGOICP_TRANSPORT_RCCL) {
        for (int a = 0; a < ndev; ++a)
            for (int b = a + 1; b < ndev; ++b)
                if (devices[a] == devices[b]) { set_error("fgoicp_multi_create: the RCCL transport needs distinct devices (use FGOICP_TRANSPORT_IN_PROCESS to rehearse)"); return fail(FGOICP_ERR_INVALID_ARG); }
        unsigned char id[128];
        int rc = fgoicp_rccl_unique_id(id);
        if (rc) return fail(rc);
        m->rccl.assign(ndev, nullptr);
        for (int r = 0; r < ndev; ++r) {  // phase 1, on this thread: what a rank can fail at on its own
            rc = rccl_prepare(r, ndev, devices[r], &m->rccl[r]);
            if (rc) { set_error("rank " + std::to_string(r) + ": " + fgoicp_last_error()); return fail(rc); }
        }
        std::vector<int> rcs(ndev, 0);
        std::vector<std::string> errs(ndev);
        std::atomic<bool> give_up{false};
        std::vector<std::thread> th;  // phase 2: the communicator forms when every rank has joined — one thread per rank, non-blocking init
        for (int r = 0; r < ndev; ++r)
            th.emplace_back([&, r] {
                rcs[r] = rccl_join(m->rccl[r], id, &give_up);
                if (rcs[r]) { errs[r] = fgoicp_last_error(); give_up.store(true, std::memory_order_release); }
            });
        for (auto& t : th) t.join();
        for (int r = 0; r < ndev; ++r)
            if (rcs[r] && errs[r].find("abandoned") == std::string::npos) { set_error("rank " + std::to_string(r) + ": " + errs[r]); return fail(rcs[r]); }
        for (int r = 0; r < ndev; ++r)
            if (rcs[r]) { set_error("rank " + std::to_string(r) + ": " + errs[r]); return fail(rcs[r]); }
    }
    for (int r = 0; r < ndev; ++r) {
        auto l = std::make_unique<RankLink>();
        l->rank = r;
        l->world = ndev;
        l->log = &m->logs[r];
        if (ndev > 1 && transport == FGOICP_TRANSPORT_RCCL) fgoicp_rccl_exchange(m->rccl[r], &l->inner);
        else l->rv = &m->rv;
        l->dg = &m->dg;
        l->device = devices[r];
        fgoicp_exchange ex{r, ndev, link_allreduce_min, link_allgather, l.get(), link_allgather_device};
        int rc = fgoicp_solver_set_exchange(m->solvers[r], ndev > 1 ? &ex : nullptr);
        if (rc) return fail(rc);
        m->links.push_back(std::move(l));
    }
    *out = m.release();
    return FGOICP_OK;
}

// TEST HOOK (tests/test_gpu_multi.py): the `call`-th exchange of `rank` in the next run fails, once.  Nothing in the product calls it.
int fgoicp_multi_test_fault(fgoicp_multi* m, int rank, long call) {
    if (!m || rank < 0 || rank >= (int)m->links.size()) return FGOICP_ERR_INVALID_ARG;
    m->links[(size_t)rank]->fail_at = call;
    return FGOICP_OK;
}

// What the last recorded run exchanged: host-side collectives of `rank` (all-reduces and all-gathers) and device all-gathers.
int fgoicp_multi_recorded(const fgoicp_multi* m, int rank, uint64_t* host_exchanges, uint64_t* device_allgathers) {
    if (!m || rank < 0 || rank >= (int)m->logs.size()) return FGOICP_ERR_INVALID_ARG;
    if (host_exchanges) *host_exchanges = m->logs[(size_t)rank].size();
    if (device_allgathers) *device_allgathers = m->dg.log.size();
    return FGOICP_OK;
}

int fgoicp_multi_set_record(fgoicp_multi* m, int on) {
    if (!m) return FGOICP_ERR_INVALID_ARG;
    for (size_t r = 0; r < m->links.size(); ++r) {
        m->links[r]->record = on != 0;
        m->links[r]->replay = false;
        if (on) m->logs[r].clear();
        if (on) m->dg.clear_log();
    }
    return FGOICP_OK;
}

// Every rank's run() on its own host thread; the result is rank 0's (all ranks hold the same incumbent after the last exchange).
static int multi_run_impl(fgoicp_multi* m, float* R_out9, float* t_out3);
int fgoicp_multi_run(fgoicp_multi* m, float* R_out9, float* t_out3) {
    if (!m || !R_out9 || !t_out3) return FGOICP_ERR_INVALID_ARG;
    return fgoicp::abi_guard("fgoicp_multi_run", [&] { return multi_run_impl(m, R_out9, t_out3); });
}
static int multi_run_impl(fgoicp_multi* m, float* R_out9, float* t_out3) {
    const int n = (int)m->solvers.size();
    std::vector<int> rcs(n, 0);
    std::vector<std::string> errs(n);
    std::vector<float> R(9 * (size_t)n), t(3 * (size_t)n);
    for (auto& l : m->links) { l->replay = false; l->calls = 0; if (l->record) { l->log->clear(); m->dg.clear_log(); } }
    m->rv.reset();
    std::atomic<int> first_failed{-1};
    std::vector<std::thread> th;
    for (int r = 0; r < n; ++r)
        th.emplace_back([&, r] {
            const auto t0 = std::chrono::steady_clock::now();
            rcs[r] = fgoicp_solver_run(m->solvers[r], &R[9 * (size_t)r], &t[3 * (size_t)r]);
            if (rcs[r]) {  // the others would wait for this rank in their next collective for ever: end the exchange for all
                errs[r] = fgoicp_last_error();
                int none = -1;
                first_failed.compare_exchange_strong(none, r);
                m->rv.abort();
                for (fgoicp_rccl* x : m->rccl) (void)fgoicp_rccl_abort(x);
            }
            m->seconds[r] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        });
    for (auto& x : th) x.join();
    if (const int r = first_failed.load(); r >= 0) {  // the rank that failed on its own, not the ones it took down
        set_error("rank " + std::to_string(r) + ": " + errs[r]);
        return rcs[r];
    }
    for (int r = 1; r < n; ++r)
        if (std::memcmp(&R[0], &R[9 * (size_t)r], 36) != 0 || std::memcmp(&t[0], &t[3 * (size_t)r], 12) != 0) {
            set_error("fgoicp_multi_run: ranks ended with different incumbents");
            return FGOICP_ERR_EXCHANGE;
        }
    std::memcpy(R_out9, R.data(), 36);
    std::memcpy(t_out3, t.data(), 12);
    return FGOICP_OK;
}

// ONE rank alone against the recording of the last recorded run: what that rank would do on a GPU of its own (everything but
// the latency of the collectives).  seconds_out = wall-clock of its run().
int fgoicp_multi_replay_rank(fgoicp_multi* m, int rank, double* seconds_out) {
    if (!m || rank < 0 || rank >= (int)m->solvers.size()) return FGOICP_ERR_INVALID_ARG;
    RankLink* l = m->links[rank].get();
    if (m->solvers.size() > 1 && l->log->empty()) { set_error("fgoicp_multi_replay_rank: nothing recorded (fgoicp_multi_set_record, then fgoicp_multi_run)"); return FGOICP_ERR_INVALID_ARG; }
    const bool was_recording = l->record;
    l->record = false;
    l->replay = true;
    l->replay_pos = 0;
    l->dev_replay_pos = 0;
    float R[9], t[3];
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = fgoicp_solver_run(m->solvers[rank], R, t);
    if (seconds_out) *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    l->replay = false;
    l->record = was_recording;
    return rc;
}

// ONE IterativeClosestPoint3D::run() executed by all ranks together (what a cooperative round does for every triggered refinement):
// every rank thread scans its share of the source, the per-query results are all-gathered on device memory.  The result is rank 0's;
// every rank must end with the same bits (checked).
int fgoicp_multi_icp(fgoicp_multi* m, const float* R0, const float* t0, size_t max_iter, float convergence_threshold, float* sse_out, float* R_out9, float* t_out3,
                     int* iterations_out) {
    if (!m || !R0 || !t0 || !sse_out || !R_out9 || !t_out3) return FGOICP_ERR_INVALID_ARG;
    const int n = (int)m->solvers.size();
    std::vector<int> rcs(n, 0), its(n, 0);
    std::vector<std::string> errs(n);
    std::vector<float> sse(n), R(9 * (size_t)n), t(3 * (size_t)n);
    for (auto& l : m->links) { l->replay = false; l->calls = 0; }
    m->rv.reset();
    std::vector<std::thread> th;
    for (int r = 0; r < n; ++r)
        th.emplace_back([&, r] {
            RankLink* l = m->links[r].get();
            rcs[r] = fgoicp::ctx_icp_coop(fgoicp_solver_ctx(m->solvers[r]), r, n, n > 1 ? link_allgather_device : nullptr, l, R0, t0, max_iter, convergence_threshold, &sse[r],
                                          &R[9 * (size_t)r], &t[3 * (size_t)r], &its[r]);
            if (rcs[r]) {
                errs[r] = fgoicp_last_error();
                m->rv.abort();
                for (fgoicp_rccl* x : m->rccl) (void)fgoicp_rccl_abort(x);
            }
        });
    for (auto& x : th) x.join();
    for (int r = 0; r < n; ++r)
        if (rcs[r]) { set_error("rank " + std::to_string(r) + ": " + errs[r]); return rcs[r]; }
    for (int r = 1; r < n; ++r)
        if (std::memcmp(&sse[0], &sse[r], 4) != 0 || std::memcmp(&R[0], &R[9 * (size_t)r], 36) != 0 || std::memcmp(&t[0], &t[3 * (size_t)r], 12) != 0 || its[r] != its[0]) {
            set_error("fgoicp_multi_icp: ranks ended with different results");
            return FGOICP_ERR_EXCHANGE;
        }
    *sse_out = sse[0];
    std::memcpy(R_out9, R.data(), 36);
    std::memcpy(t_out3, t.data(), 12);
    if (iterations_out) *iterations_out = its[0];
    return FGOICP_OK;
}

int fgoicp_multi_world(const fgoicp_multi* m) { return m ? (int)m->solvers.size() : 0; }
fgoicp_solver* fgoicp_multi_solver(fgoicp_multi* m, int rank) { return m && rank >= 0 && rank < (int)m->solvers.size() ? m->solvers[rank] : nullptr; }
int fgoicp_multi_seconds(const fgoicp_multi* m, int rank, double* seconds) {
    if (!m || !seconds || rank < 0 || rank >= (int)m->seconds.size()) return FGOICP_ERR_INVALID_ARG;
    *seconds = m->seconds[rank];
    return FGOICP_OK;
}

}  // extern "C"
