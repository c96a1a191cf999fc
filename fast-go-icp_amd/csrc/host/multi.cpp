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
#include <link.h>   // dl_iterate_phdr: is a librccl already mapped into the process?

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
#include "multi_link.hpp"
#include "knobs.hpp"

namespace fgoicp {
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
    int test_inprogress = 0;                      // test hook (fgoicp_rccl_test_inprogress): that many collectives report ncclInProgress once
    uint64_t settled = 0;                         // polls of ncclCommGetAsyncError that rccl_settle has made
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
    std::string error, path;
    bool ok = false;
};
// Which librccl?  A process may already hold one: PyTorch-ROCm bundles its own librccl.so and librocm_smi64.so under torch/lib.  Loading the
// system's /opt/rocm/lib/librccl.so.1 next to it puts a SECOND librocm_smi64 into the process, both copies export the same C++ globals
// (amd::smi::...), the dynamic linker binds both to the copy that was loaded first, and that copy's objects are then constructed and
// DESTROYED TWICE: glibc aborts with "double free or corruption (!prev)" in the exit handlers, after main() has returned (round 3's and round
// 4's recorded heap abort: the destructor of a std::map<amd::smi::DevInfoTypes, const char*> of librocm_smi64.so.1 — profiles/
// r04_exit_abort_rocm_smi.txt; it needs both libraries in one process, which is why only the replays that also imported torch hit it).
// So: (1) if the process already has a librccl mapped, that instance is used — one RCCL, one SMI; (2) otherwise the system library is loaded
// RTLD_LOCAL | RTLD_DEEPBIND (round 3: RTLD_GLOBAL), so that its dependencies bind to themselves and a copy that something else loads later
// neither sees their symbols nor overrides them.
static int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* out) {
    const char* name = info->dlpi_name;
    if (!name || !*name) return 0;
    const char* base = std::strrchr(name, '/');
    base = base ? base + 1 : name;
    if (std::strncmp(base, "librccl.so", 10) != 0) return 0;
    *static_cast<std::string*>(out) = name;
    return 1;
}
const RcclApi& rccl_api() {
    static const RcclApi api = [] {
        RcclApi a;
        void* h = nullptr;
        std::string loaded;
        const char* legacy = fgoicp::dev_env("FGOICP_RCCL_LOAD_GLOBAL");  // development build only: round 3's loading (system library, RTLD_GLOBAL) — the A/B that shows the exit abort
        const bool old_way = legacy && std::atoi(legacy) != 0;
        if (!old_way) (void)dl_iterate_phdr(find_loaded_rccl, &loaded);
        if (!loaded.empty()) h = dlopen(loaded.c_str(), RTLD_NOW | RTLD_NOLOAD);  // the instance the process already runs (e.g. PyTorch's)
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (h) break;
            h = dlopen(name, old_way ? (RTLD_NOW | RTLD_GLOBAL) : (RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND));
        }
        if (!h) { a.error = std::string("RCCL is not installed (dlopen librccl.so.1: ") + dlerror() + ")"; return a; }
        bool all = true;
        auto sym = [&](auto& fn, const char* name) { fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(h, name)); all = all && fn != nullptr; };
        sym(a.GetUniqueId, "ncclGetUniqueId"); sym(a.CommInitRank, "ncclCommInitRank"); sym(a.CommInitRankConfig, "ncclCommInitRankConfig");
        sym(a.CommGetAsyncError, "ncclCommGetAsyncError"); sym(a.CommDestroy, "ncclCommDestroy"); sym(a.CommAbort, "ncclCommAbort");
        sym(a.CommCount, "ncclCommCount"); sym(a.AllReduce, "ncclAllReduce"); sym(a.AllGather, "ncclAllGather"); sym(a.GetErrorString, "ncclGetErrorString");
        if (!all) { a.error = "librccl.so lacks an entry point this library uses"; return a; }
        a.path = !loaded.empty() ? loaded + " (already in the process)" : old_way ? "librccl.so.1 (loaded by libfgoicp_amd, RTLD_GLOBAL: round 3's way)" : "librccl.so.1 (loaded by libfgoicp_amd, RTLD_LOCAL | RTLD_DEEPBIND)";
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
// (RCCL_NCCL: see rccl_settle below — a collective on a non-blocking communicator may return ncclInProgress)

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

// fgoicp_multi_create forms its communicators NON-blocking (a rank must be able to abandon the set-up when a peer fails), and a
// communicator stays what it was created as: every later call on it — the collectives included — may return ncclInProgress, which
// means "not yet on the stream".  Nothing more may be enqueued on x->stream, and the buffers may not be touched, until
// ncclCommGetAsyncError reports ncclSuccess (round 3 treated ncclInProgress as a failure and enqueued the copy back at
// once).  On a blocking communicator (fgoicp_rccl_create, one process per GPU) the first status is final.
int rccl_settle(fgoicp_rccl* x, ncclResult_t r, const char* what) {
    if (x->test_inprogress > 0 && r == ncclSuccess) { --x->test_inprogress; r = ncclInProgress; }  // test hook: walk the polling path
    for (unsigned spins = 0; r == ncclInProgress; ++spins) {
        if (x->dead.load(std::memory_order_acquire)) return rccl_dead(x);
        ncclResult_t st = ncclSuccess;
        if (rccl_api().CommGetAsyncError(x->comm, &st) != ncclSuccess) { r = ncclInternalError; break; }
        r = st;
        if (r == ncclInProgress) { if ((spins & 63u) == 63u) std::this_thread::sleep_for(std::chrono::microseconds(20)); else std::this_thread::yield(); }
        x->settled++;
    }
    if (r != ncclSuccess) {
        set_error(std::string(what) + " failed: " + rccl_api().GetErrorString(r));
        rccl_release(x);
        return 1;
    }
    return 0;
}

int rccl_allreduce_min(float* buf, size_t n, void* user) {
    fgoicp_rccl* x = static_cast<fgoicp_rccl*>(user);
    if (x->released || x->dead.load(std::memory_order_acquire)) return rccl_dead(x);
    RCCL_HIP(hipSetDevice(x->device));
    if (rccl_reserve(x, n)) return 1;
    std::memcpy(x->h_pin, buf, sizeof(float) * n);
    RCCL_HIP(hipMemcpyAsync(x->d_send, x->h_pin, sizeof(float) * n, hipMemcpyHostToDevice, x->stream));
    if (rccl_settle(x, rccl_api().AllReduce(x->d_send, x->d_recv, n, ncclFloat, ncclMin, x->comm, x->stream), "ncclAllReduce")) return 1;
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
    if (rccl_settle(x, rccl_api().AllGather(x->d_send, x->d_recv, n, ncclFloat, x->comm, x->stream), "ncclAllGather")) return 1;
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
    if (rccl_settle(x, rccl_api().AllGather(static_cast<char*>(buf) + bytes * (size_t)x->rank, buf, bytes, ncclInt8, x->comm, x->stream), "ncclAllGather (device)")) return 1;
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

// nonblocking != 0: the communicator is created with ncclConfig_t.blocking = 0, as fgoicp_multi_create does for its rank threads —
// every collective then goes through rccl_settle's polling (tests/test_gpu_multi.py runs this on a one-rank communicator).
int fgoicp_rccl_create_ex(int rank, int world, const unsigned char* id128, int device, int nonblocking, fgoicp_rccl** out) {
    if (!out) return FGOICP_ERR_INVALID_ARG;
    *out = nullptr;
    if (!id128) { set_error("fgoicp_rccl_create: invalid argument"); return FGOICP_ERR_INVALID_ARG; }
    fgoicp_rccl* x = nullptr;
    int rc = rccl_prepare(rank, world, device, &x);  // one process per GPU: a rank that fails here is its launcher's to report
    if (rc) return rc;
    static const std::atomic<bool> never{false};
    rc = rccl_join(x, id128, nonblocking ? &never : nullptr);
    if (rc) { fgoicp_rccl_destroy(x); return rc; }
    *out = x;
    return FGOICP_OK;
}
int fgoicp_rccl_create(int rank, int world, const unsigned char* id128, int device, fgoicp_rccl** out) {
    return fgoicp_rccl_create_ex(rank, world, id128, device, 0, out);
}

// TEST HOOK, not part of the drop-in surface: the next n collectives of x report ncclInProgress once before their real status, so
// that the polling path of a non-blocking communicator runs on a box where RCCL itself answers at once.  settled_out (optional):
// polls of ncclCommGetAsyncError made so far.
int fgoicp_rccl_test_inprogress(fgoicp_rccl* x, int n, uint64_t* settled_out) {
    if (!x || n < 0) return FGOICP_ERR_INVALID_ARG;
    x->test_inprogress = n;
    if (settled_out) *settled_out = x->settled;
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
    out->struct_size = sizeof(fgoicp_exchange);
    out->rank = x->rank;
    out->world_size = x->world;
    out->allreduce_min = rccl_allreduce_min;
    out->allgather = rccl_allgather;
    out->user = x;
    out->allgather_device = rccl_allgather_device;
    return FGOICP_OK;
}

// Which librccl the transport uses (diagnostic; bench.py prints it): the instance already mapped into the process, or the one this library loaded.
const char* fgoicp_rccl_library(void) { return rccl_api().ok ? rccl_api().path.c_str() : rccl_api().error.c_str(); }

int fgoicp_rccl_calls(const fgoicp_rccl* x, uint64_t* calls) {
    if (!x || !calls) return FGOICP_ERR_INVALID_ARG;
    *calls = x->calls;
    return FGOICP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// One process, one host thread and one solver per device.  The transport-independent core — rendezvous, exchange callbacks with
// recording / replay / fault injection, the per-rank threads — is multi_link.hpp (also instantiated, with host memory and the CPU
// oracle's operators, by tests/host_harness/multi_asan.cpp under AddressSanitizer); here: the HIP runtime and RCCL underneath it.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

int hip_mem_alloc(int device, size_t bytes, void** out) {
    if (hipSetDevice(device) != hipSuccess || hipMalloc(out, bytes) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; set_error("device all-gather: hipMalloc failed"); return 1; }
    return 0;
}
void hip_mem_release(int device, void* p) {
    if (!p) return;
    (void)hipSetDevice(device);
    (void)hipFree(p);
}
int hip_mem_copy(void* dst, int dst_device, const void* src, int src_device, size_t bytes) {
    const hipError_t e = dst_device == src_device ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, nullptr) : hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes, nullptr);
    if (e != hipSuccess) { set_error(std::string("device all-gather: copy failed: ") + hipGetErrorString(e)); return 1; }
    return 0;
}
int hip_mem_sync(int /*device*/) {
    if (hipStreamSynchronize(nullptr) != hipSuccess) { set_error("device all-gather: hipStreamSynchronize failed"); return 1; }
    return 0;
}
const fgoicp::DeviceMemApi kHipMem{hip_mem_alloc, hip_mem_release, hip_mem_copy, hip_mem_sync};

struct HipBackend {
    using Solver = fgoicp_solver;
    static int run(Solver* s, float* R9, float* t3) { return fgoicp_solver_run(s, R9, t3); }
    static int set_exchange(Solver* s, const fgoicp_exchange* ex) { return fgoicp_solver_set_exchange(s, ex); }
    static void destroy(Solver* s) { fgoicp_solver_destroy(s); }
    static int icp_coop(Solver* s, int rank, int world, int (*gather)(void*, size_t, void*), void* user, const float* R0, const float* t0, size_t max_iter, float thr, float* sse,
                        float* R9, float* t3, int* iters) {
        return fgoicp::ctx_icp_coop(fgoicp_solver_ctx(s), rank, world, gather, user, R0, t0, max_iter, thr, sse, R9, t3, iters);
    }
    static const char* last_error() { return fgoicp_last_error(); }
    static const fgoicp::DeviceMemApi* mem() { return &kHipMem; }
};

}  // namespace

struct fgoicp_multi : fgoicp::MultiCore<HipBackend> {
    std::vector<fgoicp_rccl*> rccl;
    int transport = FGOICP_TRANSPORT_RCCL;
    // solvers, then the recordings (device memory), then the communicators the links point into — each exactly once, whatever
    // path ends the object (fgoicp_multi_destroy, a failed or throwing create)
    ~fgoicp_multi() override {
        destroy_solvers();
        dg.clear_log();
        links.clear();
        for (fgoicp_rccl* x : rccl) fgoicp_rccl_destroy(x);
        rccl.clear();
    }
};

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
    m->init(devices, ndev);
    m->transport = transport;
    fgoicp_solver_opts o{FGOICP_SCHEDULE_ROUND, 0, 0u, 0, 0.0f};
    if (opts) o = *opts;
    // both schedules shard: ROUND deals a round's children over the ranks, SERIAL (the reference's exact order) deals the tasks of
    // every speculative evaluation (driver.hpp: run_task_list_sharded); opts == NULL: ROUND, adaptive width
    for (int r = 0; r < ndev; ++r) {
        o.device = devices[r];
        fgoicp_solver* s = nullptr;
        int rc = fgoicp_solver_create(tgt_xyz, nt, src_xyz, ns, lut_resolution, mse_threshold, &o, &s);
        if (rc) return rc;  // (m's destructor frees what has been built)
        m->solvers.push_back(s);
    }
    const bool use_rccl = ndev > 1 && transport == FGOICP_TRANSPORT_RCCL;
    if (use_rccl) {
        for (int a = 0; a < ndev; ++a)
            for (int b = a + 1; b < ndev; ++b)
                if (devices[a] == devices[b]) { set_error("fgoicp_multi_create: the RCCL transport needs distinct devices (use FGOICP_TRANSPORT_IN_PROCESS to rehearse)"); return FGOICP_ERR_INVALID_ARG; }
        unsigned char id[128];
        int rc = fgoicp_rccl_unique_id(id);
        if (rc) return rc;
        m->rccl.assign((size_t)ndev, nullptr);
        for (int r = 0; r < ndev; ++r) {  // phase 1, on this thread: what a rank can fail at on its own
            rc = rccl_prepare(r, ndev, devices[r], &m->rccl[(size_t)r]);
            if (rc) { set_error("rank " + std::to_string(r) + ": " + fgoicp_last_error()); return rc; }
        }
        std::vector<int> rcs((size_t)ndev, 0);
        std::vector<std::string> errs((size_t)ndev);
        std::atomic<bool> give_up{false};
        std::vector<std::thread> th;  // phase 2: the communicator forms when every rank has joined — one thread per rank, non-blocking init
        fgoicp_multi* mp = m.get();
        for (int r = 0; r < ndev; ++r)
            th.emplace_back([&, mp, r] {
                rcs[(size_t)r] = rccl_join(mp->rccl[(size_t)r], id, &give_up);
                if (rcs[(size_t)r]) { errs[(size_t)r] = fgoicp_last_error(); give_up.store(true, std::memory_order_release); }
            });
        for (auto& t : th) t.join();
        for (int r = 0; r < ndev; ++r)
            if (rcs[(size_t)r] && errs[(size_t)r].find("abandoned") == std::string::npos) { set_error("rank " + std::to_string(r) + ": " + errs[(size_t)r]); return rcs[(size_t)r]; }
        for (int r = 0; r < ndev; ++r)
            if (rcs[(size_t)r]) { set_error("rank " + std::to_string(r) + ": " + errs[(size_t)r]); return rcs[(size_t)r]; }
        fgoicp_multi* mraw = m.get();
        m->abort_transport = [mraw] { for (fgoicp_rccl* x : mraw->rccl) (void)fgoicp_rccl_abort(x); };
    }
    for (int r = 0; r < ndev; ++r) {
        fgoicp_exchange inner{};
        if (use_rccl) fgoicp_rccl_exchange(m->rccl[(size_t)r], &inner);
        int rc = m->connect(r, use_rccl ? &inner : nullptr);
        if (rc) return rc;
    }
    *out = m.release();
    return FGOICP_OK;
}

// TEST HOOK (tests/test_gpu_multi.py): the `call`-th exchange of `rank` in the next run fails, once.  Nothing in the product calls it.
int fgoicp_multi_test_fault(fgoicp_multi* m, int rank, long call) { return m ? m->test_fault(rank, call) : FGOICP_ERR_INVALID_ARG; }

// What the last recorded run exchanged: host-side collectives of `rank` (all-reduces and all-gathers) and device all-gathers.
int fgoicp_multi_recorded(const fgoicp_multi* m, int rank, uint64_t* host_exchanges, uint64_t* device_allgathers) {
    return m ? m->recorded(rank, host_exchanges, device_allgathers) : FGOICP_ERR_INVALID_ARG;
}

int fgoicp_multi_set_record(fgoicp_multi* m, int on) { return m ? m->set_record(on) : FGOICP_ERR_INVALID_ARG; }

// Every rank's run() on its own host thread; the result is rank 0's (all ranks hold the same incumbent after the last exchange).
int fgoicp_multi_run(fgoicp_multi* m, float* R_out9, float* t_out3) {
    if (!m || !R_out9 || !t_out3) return FGOICP_ERR_INVALID_ARG;
    return fgoicp::abi_guard("fgoicp_multi_run", [&] { return m->run(R_out9, t_out3); });
}

// ONE rank alone against the recording of the last recorded run: what that rank would do on a GPU of its own (everything but
// the latency of the collectives).  seconds_out = wall-clock of its run().
int fgoicp_multi_replay_rank(fgoicp_multi* m, int rank, double* seconds_out) {
    if (!m) return FGOICP_ERR_INVALID_ARG;
    return fgoicp::abi_guard("fgoicp_multi_replay_rank", [&] { return m->replay_rank(rank, seconds_out); });
}

// ONE IterativeClosestPoint3D::run() executed by all ranks together (what a cooperative round does for every triggered refinement).
int fgoicp_multi_icp(fgoicp_multi* m, const float* R0, const float* t0, size_t max_iter, float convergence_threshold, float* sse_out, float* R_out9, float* t_out3,
                     int* iterations_out) {
    if (!m || !R0 || !t0 || !sse_out || !R_out9 || !t_out3) return FGOICP_ERR_INVALID_ARG;
    return fgoicp::abi_guard("fgoicp_multi_icp", [&] { return m->icp(R0, t0, max_iter, convergence_threshold, sse_out, R_out9, t_out3, iterations_out); });
}

int fgoicp_multi_world(const fgoicp_multi* m) { return m ? (int)m->solvers.size() : 0; }
fgoicp_solver* fgoicp_multi_solver(fgoicp_multi* m, int rank) { return m && rank >= 0 && rank < (int)m->solvers.size() ? m->solvers[(size_t)rank] : nullptr; }
int fgoicp_multi_seconds(const fgoicp_multi* m, int rank, double* seconds) {
    if (!m || !seconds || rank < 0 || rank >= (int)m->seconds.size()) return FGOICP_ERR_INVALID_ARG;
    *seconds = m->seconds[(size_t)rank];
    return FGOICP_OK;
}

}  // extern "C"
