// The transport-independent core of fgoicp_multi (include/fgoicp_amd.h, "Multi-GPU inside the library"): the in-process rendezvous
// of the rank threads, the per-rank exchange callbacks with recording / replay / fault injection, and the runner that owns one
// solver per rank.  No HIP here: everything the core needs from the device runtime comes through `DeviceMemApi`, everything it
// needs from a solver through the `Backend` policy — multi.cpp instantiates it with the HIP runtime and fgoicp_solver, and
// tests/host_harness/multi_asan.cpp with host memory and the CPU oracle's operators, so that the same record / run / replay /
// destroy sequence that runs on the GPU runs under AddressSanitizer on the CPU.
//
// The reference is single-GPU (no NCCL/MPI call sites); nothing here replaces reference code.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/fgoicp_amd.h"

namespace fgoicp {
void set_error(const std::string& s);

// What the device all-gather of the cooperative refinements needs from the device runtime.  `copy` may return before the bytes have
// landed; `sync` waits for every copy the calling thread has issued.  All return 0 on success and describe a failure with set_error.
struct DeviceMemApi {
    int (*alloc)(int device, size_t bytes, void** out) = nullptr;
    void (*release)(int device, void* p) = nullptr;
    int (*copy)(void* dst, int dst_device, const void* src, int src_device, size_t bytes) = nullptr;
    int (*sync)(int device) = nullptr;
};

// In-process rendezvous of `world` threads: min-all-reduce and all-gather through shared memory (two generations of buffers,
// so a fast rank may enter the next collective while a slow one still reads the last result).
struct Rendezvous {
    std::mutex m;
    std::condition_variable cv;
    int world = 1, arrived = 0;
    uint64_t gen = 0;
    bool aborted = false;  // a rank failed: nobody waits for it (reset before every run)
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

// In-process all-gather on device memory: where every rank's buffer lives (written before the first barrier), and the recorded
// gathers — whole buffers, kept in DEVICE memory by rank 0: what a replayed rank receives.  The replay leaves the transfer out,
// like the replay of the host-side exchanges: its copies are on-device (microseconds); tools/scale_replay.py charges the
// collective's measured software path and a modelled wire time per gather instead.
// A recording OWNS its device buffer: move-only, freed exactly once by the object that holds it (round 3 kept plain structs that
// were copied by value into the log; nothing freed one twice, but nothing stopped it either).
class DeviceGather {
public:
    struct Rec {
        void* d = nullptr;
        size_t bytes = 0;
        int device = 0;
        const DeviceMemApi* api = nullptr;
        Rec() = default;
        Rec(const Rec&) = delete;
        Rec& operator=(const Rec&) = delete;
        Rec(Rec&& o) noexcept : d(o.d), bytes(o.bytes), device(o.device), api(o.api) { o.d = nullptr; }
        Rec& operator=(Rec&& o) noexcept {
            if (this != &o) { reset(); d = o.d; bytes = o.bytes; device = o.device; api = o.api; o.d = nullptr; }
            return *this;
        }
        ~Rec() { reset(); }
        void reset() {
            if (d && api) api->release(device, d);
            d = nullptr;
        }
    };
    DeviceGather() = default;
    DeviceGather(const DeviceGather&) = delete;
    DeviceGather& operator=(const DeviceGather&) = delete;
    std::vector<void*> ptr;
    std::vector<int> device;
    std::vector<Rec> log;
    void clear_log() { log.clear(); }
};

struct RankLink {        // what one rank's exchange callbacks see
    int rank = 0, world = 1;
    Rendezvous* rv = nullptr;
    DeviceGather* dg = nullptr;
    const DeviceMemApi* mem = nullptr;
    int device = 0;
    size_t dev_replay_pos = 0;
    fgoicp_exchange inner{};                      // transport underneath (RCCL) when rv == nullptr
    bool record = false;
    std::vector<std::vector<float>>* log = nullptr;   // results of every exchange, in order
    size_t replay_pos = 0;
    bool replay = false;
    long fail_at = -1, calls = 0;                 // test hook (fgoicp_multi_test_fault): that exchange of that rank fails, once
};

inline int link_allreduce_min(float* buf, size_t n, void* user) {
    RankLink* l = static_cast<RankLink*>(user);
    if (l->replay) {
        if (l->replay_pos >= l->log->size() || (*l->log)[l->replay_pos].size() != n) { set_error("replay: the recording has no all-reduce of this size at this point"); return 1; }
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

inline int link_allgather(const float* send, float* recv, size_t n, void* user) {
    RankLink* l = static_cast<RankLink*>(user);
    if (l->replay) {
        if (l->replay_pos >= l->log->size() || (*l->log)[l->replay_pos].size() != n * (size_t)l->world) { set_error("replay: the recording has no all-gather of this size at this point"); return 1; }
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
inline int link_allgather_device(void* buf, size_t bytes, void* user) {
    RankLink* l = static_cast<RankLink*>(user);
    const size_t total = bytes * (size_t)l->world;
    if (l->replay) {
        if (!l->dg || !l->mem || l->dev_replay_pos >= l->dg->log.size() || l->dg->log[l->dev_replay_pos].bytes != total) { set_error("replay: the recording has no device all-gather of this size at this point"); return 1; }
        const DeviceGather::Rec& rec = l->dg->log[l->dev_replay_pos++];
        const size_t lo = bytes * (size_t)l->rank, hi = lo + bytes;  // everything but this rank's own chunk: two contiguous ranges
        auto copy = [&](size_t from, size_t to) {
            if (from >= to) return 0;
            return l->mem->copy(static_cast<char*>(buf) + from, l->device, static_cast<const char*>(rec.d) + from, rec.device, to - from);
        };
        if (copy(0, lo) || copy(hi, total) || l->mem->sync(l->device)) return 1;
        return 0;
    }
    if (l->calls++ == l->fail_at) { l->fail_at = -1; set_error("injected exchange fault (fgoicp_multi_test_fault)"); return 1; }
    int rc = 0;
    if (l->rv) {
        if (!l->dg || !l->mem) { set_error("in-process device all-gather: no device memory interface"); return 1; }
        auto barrier = [&] { return l->rv->run(0, [](std::vector<float>&, bool) {}, [](const std::vector<float>&) {}); };
        l->dg->ptr[(size_t)l->rank] = buf;
        l->dg->device[(size_t)l->rank] = l->device;
        if (!barrier()) { set_error("exchange aborted: another rank failed"); return 1; }
        for (int p = 0; p < l->world && !rc; ++p) {
            if (p == l->rank) continue;
            rc = l->mem->copy(static_cast<char*>(buf) + bytes * (size_t)p, l->device, static_cast<const char*>(l->dg->ptr[(size_t)p]) + bytes * (size_t)p, l->dg->device[(size_t)p], bytes);
        }
        // device-to-device copies may return before they have landed, and the contexts' streams do not wait for the null stream
        if (!rc && l->mem->sync(l->device)) rc = 1;
        if (!barrier() && !rc) { set_error("exchange aborted: another rank failed"); rc = 1; }
    } else {
        if (!l->inner.allgather_device) { set_error("the transport has no device all-gather"); return 1; }
        rc = l->inner.allgather_device(buf, bytes, l->inner.user);
    }
    if (!rc && l->record && l->rank == 0 && l->dg && l->mem) {
        DeviceGather::Rec rec;
        rec.bytes = total;
        rec.device = l->device;
        rec.api = l->mem;
        if (l->mem->alloc(l->device, total, &rec.d)) { rec.d = nullptr; set_error("recording a device all-gather: out of device memory"); return 1; }
        if (l->mem->copy(rec.d, l->device, buf, l->device, total) || l->mem->sync(l->device)) return 1;  // (rec frees its buffer)
        l->dg->log.push_back(std::move(rec));
    }
    return rc;
}

// One solver per rank, every rank's run() on its own host thread.  Backend:
//     using Solver = ...;
//     static int  run(Solver*, float* R9, float* t3);                 status, message through set_error / last_error()
//     static int  set_exchange(Solver*, const fgoicp_exchange*);
//     static void destroy(Solver*);
//     static int  icp_coop(Solver*, int rank, int world, int (*gather)(void*, size_t, void*), void* user, const float* R0, const float* t0,
//                          size_t max_iter, float thr, float* sse, float* R9, float* t3, int* iters);
//     static const char* last_error();
//     static const DeviceMemApi* mem();
template <class Backend>
class MultiCore {
public:
    using Solver = typename Backend::Solver;
    MultiCore() = default;
    MultiCore(const MultiCore&) = delete;
    MultiCore& operator=(const MultiCore&) = delete;
    // Order of destruction (named, not left to the member order): the solvers first — nothing of a solver is used by the links, and
    // the recordings (device memory of rank 0's device) are released while the runtime the solvers used is still up —, then the
    // recorded device buffers, then the links and host logs.  A derived class that owns a transport destroys it AFTER destroy_solvers().
    virtual ~MultiCore() { destroy_solvers(); dg.clear_log(); links.clear(); }
    void destroy_solvers() {
        for (Solver* s : solvers) Backend::destroy(s);
        solvers.clear();
    }

    std::vector<int> devices;
    std::vector<Solver*> solvers;
    std::vector<std::unique_ptr<RankLink>> links;
    std::vector<std::vector<std::vector<float>>> logs;
    Rendezvous rv;
    DeviceGather dg;
    std::vector<double> seconds;                  // wall-clock of every rank's last run
    std::function<void()> abort_transport;        // a rank failed: end the transport's collectives for all (RCCL: raise every communicator's flag)

    void init(const int* devs, int ndev) {
        devices.assign(devs, devs + ndev);
        rv.world = ndev;
        logs.resize((size_t)ndev);
        dg.ptr.assign((size_t)ndev, nullptr);
        dg.device.assign((size_t)ndev, 0);
        seconds.assign((size_t)ndev, 0.0);
    }
    // installs rank r's link on its solver (solvers[r] must exist); inner == nullptr: the in-process rendezvous
    int connect(int r, const fgoicp_exchange* inner) {
        const int ndev = (int)devices.size();
        auto l = std::make_unique<RankLink>();
        l->rank = r;
        l->world = ndev;
        l->log = &logs[(size_t)r];
        if (inner) l->inner = *inner;
        else l->rv = &rv;
        l->dg = &dg;
        l->mem = Backend::mem();
        l->device = devices[(size_t)r];
        fgoicp_exchange ex{sizeof(fgoicp_exchange), r, ndev, link_allreduce_min, link_allgather, l.get(), link_allgather_device};
        const int rc = Backend::set_exchange(solvers[(size_t)r], ndev > 1 ? &ex : nullptr);
        if (rc) return rc;
        links.push_back(std::move(l));
        return FGOICP_OK;
    }

    int test_fault(int rank, long call) {
        if (rank < 0 || rank >= (int)links.size()) return FGOICP_ERR_INVALID_ARG;
        links[(size_t)rank]->fail_at = call;
        return FGOICP_OK;
    }
    int recorded(int rank, uint64_t* host_exchanges, uint64_t* device_allgathers) const {
        if (rank < 0 || rank >= (int)logs.size()) return FGOICP_ERR_INVALID_ARG;
        if (host_exchanges) *host_exchanges = logs[(size_t)rank].size();
        if (device_allgathers) *device_allgathers = dg.log.size();
        return FGOICP_OK;
    }
    int set_record(int on) {
        for (size_t r = 0; r < links.size(); ++r) {
            links[r]->record = on != 0;
            links[r]->replay = false;
            if (on) logs[r].clear();
        }
        if (on) dg.clear_log();
        return FGOICP_OK;
    }

    // Every rank's run() on its own host thread; the result is rank 0's (all ranks hold the same incumbent after the last exchange).
    int run(float* R_out9, float* t_out3) {
        const int n = (int)solvers.size();
        std::vector<int> rcs((size_t)n, 0);
        std::vector<std::string> errs((size_t)n);
        std::vector<float> R(9 * (size_t)n), t(3 * (size_t)n);
        bool recording = false;
        for (auto& l : links) { l->replay = false; l->calls = 0; if (l->record) { l->log->clear(); recording = true; } }
        if (recording) dg.clear_log();
        rv.reset();
        std::atomic<int> first_failed{-1};
        std::vector<std::thread> th;
        for (int r = 0; r < n; ++r)
            th.emplace_back([&, r] {
                const auto t0 = std::chrono::steady_clock::now();
                rcs[(size_t)r] = Backend::run(solvers[(size_t)r], &R[9 * (size_t)r], &t[3 * (size_t)r]);
                if (rcs[(size_t)r]) {  // the others would wait for this rank in their next collective for ever: end the exchange for all
                    errs[(size_t)r] = Backend::last_error();
                    int none = -1;
                    first_failed.compare_exchange_strong(none, r);
                    rv.abort();
                    if (abort_transport) abort_transport();
                }
                seconds[(size_t)r] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            });
        for (auto& x : th) x.join();
        if (const int r = first_failed.load(); r >= 0) {  // the rank that failed on its own, not the ones it took down
            set_error("rank " + std::to_string(r) + ": " + errs[(size_t)r]);
            return rcs[(size_t)r];
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
    int replay_rank(int rank, double* seconds_out) {
        if (rank < 0 || rank >= (int)solvers.size()) return FGOICP_ERR_INVALID_ARG;
        RankLink* l = links[(size_t)rank].get();
        if (solvers.size() > 1 && l->log->empty()) { set_error("fgoicp_multi_replay_rank: nothing recorded (fgoicp_multi_set_record, then fgoicp_multi_run)"); return FGOICP_ERR_INVALID_ARG; }
        const bool was_recording = l->record;
        l->record = false;
        l->replay = true;
        l->replay_pos = 0;
        l->dev_replay_pos = 0;
        float R[9], t[3];
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = Backend::run(solvers[(size_t)rank], R, t);
        if (seconds_out) *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        l->replay = false;
        l->record = was_recording;
        return rc;
    }

    // ONE IterativeClosestPoint3D::run() executed by all ranks together (what a cooperative round does for every triggered refinement):
    // every rank thread scans its share of the source, the per-query results are all-gathered on device memory.  The result is rank 0's;
    // every rank must end with the same bits (checked).
    int icp(const float* R0, const float* t0, size_t max_iter, float convergence_threshold, float* sse_out, float* R_out9, float* t_out3, int* iterations_out) {
        const int n = (int)solvers.size();
        std::vector<int> rcs((size_t)n, 0), its((size_t)n, 0);
        std::vector<std::string> errs((size_t)n);
        std::vector<float> sse((size_t)n), R(9 * (size_t)n), t(3 * (size_t)n);
        for (auto& l : links) { l->replay = false; l->calls = 0; }
        rv.reset();
        std::vector<std::thread> th;
        for (int r = 0; r < n; ++r)
            th.emplace_back([&, r] {
                RankLink* l = links[(size_t)r].get();
                rcs[(size_t)r] = Backend::icp_coop(solvers[(size_t)r], r, n, n > 1 ? link_allgather_device : nullptr, l, R0, t0, max_iter, convergence_threshold, &sse[(size_t)r],
                                                   &R[9 * (size_t)r], &t[3 * (size_t)r], &its[(size_t)r]);
                if (rcs[(size_t)r]) {
                    errs[(size_t)r] = Backend::last_error();
                    rv.abort();
                    if (abort_transport) abort_transport();
                }
            });
        for (auto& x : th) x.join();
        for (int r = 0; r < n; ++r)
            if (rcs[(size_t)r]) { set_error("rank " + std::to_string(r) + ": " + errs[(size_t)r]); return rcs[(size_t)r]; }
        for (int r = 1; r < n; ++r)
            if (std::memcmp(&sse[0], &sse[(size_t)r], 4) != 0 || std::memcmp(&R[0], &R[9 * (size_t)r], 36) != 0 || std::memcmp(&t[0], &t[3 * (size_t)r], 12) != 0 || its[(size_t)r] != its[0]) {
                set_error("fgoicp_multi_icp: ranks ended with different results");
                return FGOICP_ERR_EXCHANGE;
            }
        *sse_out = sse[0];
        std::memcpy(R_out9, R.data(), 36);
        std::memcpy(t_out3, t.data(), 12);
        if (iterations_out) *iterations_out = its[0];
        return FGOICP_OK;
    }
};

}  // namespace fgoicp
