// Host side of the Go-ICP search: node types, pre-processing and the outer SO(3) / inner R^3
// branch-and-bound of icp::FastGoICP (reference fgoicp/fgoicp.hpp:13-110, fgoicp/fgoicp.cpp:10-287,
// fgoicp/common.hpp:30-128), written against an abstract operator backend `Ops`:
//
//     int    Ops::bounds_multi(G, R9, rot_span, fix_rot, offsets, tnodes4, lb, ub, cut_above)   (fgoicp_bounds_multi)
//     int    Ops::bounds_submit(slot, G, ...tnodes4, twin, cut_above) / bounds_collect(slot, lb, ub)  (fgoicp_bounds_submit_cut / _collect)
//     bool   Ops::async()                                                            two slots available?
//     bool   Ops::twins()                                                            does bounds_submit evaluate a twin pair once? (then the memo is free)
//     int    Ops::icp(R0, t0, max_iter, thr, &sse, R9, t3, &iters)                   (fgoicp_icp)
//
// The product instantiates it with the HIP context only (solver.cpp).  tests/ instantiate it with
// the CPU oracle's operators to check the host logic without a GPU; no such instantiation is
// compiled into libfgoicp_amd.so.
//
// Two schedules:
//   SERIAL — the reference's exact exploration order: one rotation child at a time, UB inner BnB,
//            optional ICP, LB inner BnB, `best_sse` updated between siblings (fgoicp.cpp:51-97).
//   ROUND  — expansion rounds: pop K cubes, evaluate the UB and LB inner BnBs of all children
//            concurrently (one fused operator submission per tick), children sharded over ranks,
//            one min-all-reduce of the best error + one all-gather per round.  Bounds stay valid
//            (a stale, larger `best_sse` only prunes less); the exploration order differs from the
//            reference, the final optimum is asserted equal in tests.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <array>
#include <functional>
#include <iterator>
#include <map>
#include <thread>
#include <tuple>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <queue>
#include <vector>

#include "math3.hpp"
#include "knobs.hpp"

namespace fgoicp {

constexpr float kHostInf = 1E+10f;  // M_INF, common.hpp:18

// ---- node types (common.hpp:30-128) ---------------------------------------------------------
struct RotationQ {
    float x, y, z, r;
    Mat3f R;
    RotationQ() : RotationQ(0.f, 0.f, 0.f) {}
    RotationQ(float x_, float y_, float z_) : x(x_), y(y_), z(z_), r(x_ * x_ + y_ * y_ + z_ * z_), R(Mat3f::identity()) {
        if (r > 1.0f) return;  // not a rotation: R stays identity, r stays the squared norm (common.hpp:42)
        const float ww = 1.0f - r;
        const float w = std::sqrt(ww);
        const float wx = w * x, xx = x * x;
        const float wy = w * y, xy = x * y, yy = y * y;
        const float wz = w * z, xz = x * z, yz = y * z, zz = z * z;
        const float cols[9] = {ww + xx - yy - zz, 2 * (xy - wz),     2 * (xz + wy),      // column 0
                               2 * (xy + wz),     ww - xx + yy - zz, 2 * (yz - wx),      // column 1
                               2 * (xz - wy),     2 * (yz + wx),     ww - xx - yy + zz}; // column 2
        std::memcpy(R.m, cols, sizeof(cols));
        r = std::sqrt(r);
    }
    bool in_SO3() const { return r <= 1.0f; }
};

struct RotCube {
    RotationQ q;
    float span, lb, ub;
    RotCube(float x, float y, float z, float span_, float lb_, float ub_) : q(x, y, z), span(span_), lb(lb_), ub(ub_) {}
    // std::priority_queue pops the smallest lb first, ties → larger span (common.hpp:85-92)
    friend bool operator<(const RotCube& a, const RotCube& b) { return a.lb == b.lb ? a.span < b.span : a.lb > b.lb; }
    bool overlaps_SO3() const {  // common.hpp:99-103
        return q.r - 2 * span * (std::fabs(q.x) + std::fabs(q.y) + std::fabs(q.z)) + 3 * span * span <= 1;
    }
};

struct TransCube {
    Vec3f t;
    float span, lb, ub;
    TransCube(float x, float y, float z, float span_, float lb_, float ub_) : t{x, y, z}, span(span_), lb(lb_), ub(ub_) {}
    friend bool operator<(const TransCube& a, const TransCube& b) { return a.lb == b.lb ? a.span < b.span : a.lb > b.lb; }
};

// ---- pre-processing (fgoicp.cpp:176-287; serial fp32, the omp pragmas there are inert) ----------
inline Vec3f center_point_cloud(std::vector<Vec3f>& pc) {
    Vec3f c{0.f, 0.f, 0.f};
    for (const Vec3f& p : pc) c = c + p;
    c = c / static_cast<float>(pc.size());
    for (Vec3f& p : pc) p = p - c;
    return -c;
}
inline float scale_point_clouds(std::vector<Vec3f>& pct, std::vector<Vec3f>& pcs) {
    float max_abs = std::numeric_limits<float>::lowest();
    for (const Vec3f& p : pcs) max_abs = std::max(max_abs, std::max(std::fabs(p.x), std::max(std::fabs(p.y), std::fabs(p.z))));
    const float s = 1.0f / max_abs;
    for (Vec3f& p : pcs) p = p * s;
    for (Vec3f& p : pct) p = p * s;
    return s;
}
inline void point_cloud_ranges(const std::vector<Vec3f>& pc, float bounds6[6]) {
    for (int a = 0; a < 3; ++a) {
        bounds6[2 * a] = std::numeric_limits<float>::max();
        bounds6[2 * a + 1] = std::numeric_limits<float>::lowest();
    }
    for (const Vec3f& p : pc) {
        const float v[3] = {p.x, p.y, p.z};
        for (int a = 0; a < 3; ++a) {
            bounds6[2 * a] = std::min(bounds6[2 * a], v[a]);
            bounds6[2 * a + 1] = std::max(bounds6[2 * a + 1], v[a]);
        }
    }
}

// ---- exchange hook (mirrors fgoicp_exchange of the C ABI) --------------------------------------
struct Exchange {
    int rank = 0, world = 1;
    int (*allreduce_min)(float*, size_t, void*) = nullptr;
    int (*allgather)(const float*, float*, size_t, void*) = nullptr;
    void* user = nullptr;
    int (*allgather_device)(void*, size_t, void*) = nullptr;  // optional: in-place all-gather on device memory -> cooperative refinements
};

struct DriverStats {
    uint64_t trans_cubes = 0, bounds_calls = 0, rot_cubes = 0, icp_runs = 0, icp_iters = 0, inner_bnb = 0, rounds = 0;
    double seconds_total = 0, seconds_bnb = 0, seconds_icp = 0, initial_icp_sse = 0;
};

enum { kScheduleSerial = 0, kScheduleRound = 1 };
enum { kDriverOk = 0, kDriverExchangeFailed = 6 };

// ---- one inner (translation) BnB as a resumable task — fgoicp.cpp:102-174 -----------------------
struct InnerTask {
    bool fix_rot = true;
    float best_error = 0.f;
    Vec3f best_t{0.f, 0.f, 0.f};
    float best_ub = kHostInf;
    float start_sse = 0.f;  // the job's best error when the task started (:104)
    uint64_t count = 0;
    std::priority_queue<TransCube> cand;
    std::vector<TransCube> batch;
    size_t batch_cap = 32;  // nodes per operator call (fgoicp.cpp:122).  ROUND uses 48 (FGOICP_ROUND_BATCH): a task's batches are a chain of device
                            // round trips, a third fewer of them costs ~1 % more subcubes and shortens every latency-bound phase

    void start(bool fix, float best_sse, float rnode_ub) {
        fix_rot = fix;
        best_error = best_sse;  // :104
        start_sse = best_sse;
        best_t = Vec3f{0.f, 0.f, 0.f};
        best_ub = kHostInf;
        count = 0;
        cand = std::priority_queue<TransCube>();
        cand.push(TransCube(0.f, 0.f, 0.f, 1.0f, 0.f, rnode_ub));  // :113
        batch.clear();
    }
    // pops the next batch of <= 32 nodes; false when the search is over (:116-130)
    bool next_batch(float sse_threshold) {
        batch.clear();
        while (batch.empty()) {
            if (cand.empty()) return false;
            if (best_error - cand.top().lb < sse_threshold) return false;  // :120
            while (!cand.empty() && batch.size() < batch_cap) {
                TransCube tn = cand.top();
                cand.pop();
                if (tn.lb < best_error) batch.push_back(tn);
            }
        }
        count += batch.size();  // :132
        return true;
    }
    // What this task does not need to know exactly (fgoicp_bounds_submit_cut): a node whose lower bound is >= the value returned here
    // may come back as {T, T} instead of {lb, ub}.  Every use of a node's bounds is a comparison that both answers decide alike:
    //   * the node itself: dropped when lb >= best_error (:151), its ub counted only when < best_error (:143) — and T >= best_error,
    //     ub >= lb >= T; best_error only falls while the task runs, so the threshold of the submission also holds for a node that
    //     is looked up LATER (memo rows, look-ahead rows);
    //   * the task's result best_ub = min of all ub (:142): a pass with rotation uncertainty (fix_rot = 0) hands it to the outer loop as
    //     the cube's lower bound, which is compared with best_sse <= start_sse (:92) and stored only when below — T = best_error;
    //     the pass with the rotation fixed hands it to the trigger rule `ub < best_sse * 1.8` (:74) — T = 1.8 * start_sse, rounded up.
    // Both are exact whenever the outer loop looks at more than the comparison.  The child nodes (:163-166) inherit exact bounds:
    // they exist only below the threshold.
    float cut_above() const {
        if (!fix_rot) return best_error;
        return std::nextafter((float)((double)start_sse * 1.8), kHostInf);
    }
    // consumes the operator's {lb, ub} of the current batch (:139-169)
    void consume(const float* lb, const float* ub) {
        const size_t n = batch.size();
        size_t idx_min = 0;
        for (size_t i = 1; i < n; ++i)
            if (ub[i] < ub[idx_min]) idx_min = i;  // std::min_element: first minimum
        best_ub = best_ub < ub[idx_min] ? best_ub : ub[idx_min];
        if (ub[idx_min] < best_error) {
            best_error = ub[idx_min];
            best_t = batch[idx_min].t;
        }
        for (size_t i = 0; i < n; ++i) {
            if (lb[i] >= best_error) continue;  // :151
            const TransCube& tn = batch[i];
            if (tn.span < 0.1f) continue;       // :155
            const float span = tn.span / 2.0f;
            for (char j = 0; j < 8; ++j)
                cand.push(TransCube(tn.t.x - span + (j >> 0 & 1) * tn.span, tn.t.y - span + (j >> 1 & 1) * tn.span,
                                    tn.t.z - span + (j >> 2 & 1) * tn.span, span, lb[i], ub[i]));
        }
    }
};

using NodeKey = std::array<uint32_t, 4>;  // bit pattern of a translation node {t.x, t.y, t.z, span}
inline NodeKey node_key(const TransCube& c) {
    NodeKey k;
    std::memcpy(k.data(), &c.t.x, 12);
    std::memcpy(&k[3], &c.span, 4);
    return k;
}
struct NodeKeyHash {
    size_t operator()(const NodeKey& b) const {
        uint32_t x = b[0] * 0x9E3779B1u ^ b[1] * 0x85EBCA77u ^ b[2] * 0xC2B2AE3Du ^ b[3] * 0x27D4EB2Fu;
        return x ^ (x >> 15);
    }
};

// Flat open-addressing table NodeKey -> {lb, ub} (a task's memo holds ~100 entries; no allocation per insert).
class NodeMemo {
public:
    bool empty() const { return n_ == 0; }
    const std::pair<float, float>* find(const NodeKey& k) const {
        if (n_ == 0) return nullptr;
        for (size_t i = NodeKeyHash()(k) & mask_;; i = (i + 1) & mask_) {
            if (!used_[i]) return nullptr;
            if (keys_[i] == k) return &vals_[i];
        }
    }
    void insert(const NodeKey& k, float lb, float ub) {
        if ((n_ + 1) * 2 > keys_.size()) grow();
        size_t i = NodeKeyHash()(k) & mask_;
        while (used_[i]) { if (keys_[i] == k) return; i = (i + 1) & mask_; }
        used_[i] = 1; keys_[i] = k; vals_[i] = {lb, ub}; ++n_;
    }
    void clear() { keys_.clear(); vals_.clear(); used_.clear(); n_ = 0; mask_ = 0; }
private:
    void grow() {
        std::vector<NodeKey> ok; std::vector<std::pair<float, float>> ov; std::vector<char> ou;
        ok.swap(keys_); ov.swap(vals_); ou.swap(used_);
        const size_t cap = ok.empty() ? 64 : ok.size() * 2;
        keys_.resize(cap); vals_.resize(cap); used_.assign(cap, 0); mask_ = cap - 1; n_ = 0;
        for (size_t i = 0; i < ok.size(); ++i) if (ou[i]) insert(ok[i], ov[i].first, ov[i].second);
    }
    std::vector<NodeKey> keys_;
    std::vector<std::pair<float, float>> vals_;
    std::vector<char> used_;
    size_t n_ = 0, mask_ = 0;
};

struct Task : InnerTask {
    std::vector<NodeKey> seen;  // FGOICP_OVERLAP_STATS only: every node this task had evaluated
    // --- memo of the twin task's evaluations (LB tasks, see prepare_half) ---
    NodeMemo memo;                   // node -> {lb, ub} of THIS task's variant, computed when the UB twin evaluated the node
    std::vector<float> blb, bub;     // bounds of the current batch, batch order (memo hits filled in at pop time)
    std::vector<int> brow;           // per batch node: its row inside the task's group of the submission, or -1 = served from the memo
    int nrows = 0;                   // rows of the group that belong to the batch
    std::vector<TransCube> phantom;  // rows after them: nodes of the twin's batch evaluated for the memo
    bool done = false;
    bool has_batch = false;
    uint64_t batches = 0;  // operator calls this task took part in (= compute_sse_error calls of the reference for it)
};

// A few host threads for the per-task queue work of a tick (the inner BnBs of a round are independent:
// popping the next batch and pushing the children of the evaluated one touch only the task's own
// priority queue).  The calling thread takes part; results do not depend on the thread count.
// A tick needs three or four of these fork-joins and lasts 0.1-1 ms, so the hand-off has to cost microseconds: the workers
// spin on a generation counter for a short while after each job (the next one usually follows within that time) before they
// sleep on a condition variable, and every participant takes a fixed slice of the index range (no shared counter).
class WorkerPool {
public:
    explicit WorkerPool(int nthreads) {
        live_threads().fetch_add(nthreads, std::memory_order_relaxed);
        nthreads_ = nthreads;
        for (int i = 1; i < nthreads; ++i) threads_.emplace_back([this, i] { worker((size_t)i); });
    }
    ~WorkerPool() {
        live_threads().fetch_sub(nthreads_, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
            generation_.fetch_add(1, std::memory_order_seq_cst);
        }
        cv_start_.notify_all();
        for (auto& t : threads_) t.join();
    }
    size_t size() const { return threads_.size() + 1; }
    void parallel_for(size_t n, const std::function<void(size_t)>& f) {
        if (threads_.empty() || n < 2) {
            for (size_t i = 0; i < n; ++i) f(i);
            return;
        }
        fn_ = &f;
        n_ = n;
        active_.store(threads_.size(), std::memory_order_relaxed);
        generation_.fetch_add(1, std::memory_order_seq_cst);  // publishes the job to spinning workers
        if (sleepers_.load(std::memory_order_seq_cst) > 0) {
            std::lock_guard<std::mutex> g(m_);  // pairs with the sleeper's predicate check: no lost wake-up
            cv_start_.notify_all();
        }
        drain(0);
        for (int spin = 0; active_.load(std::memory_order_acquire) != 0; ++spin)
            if (spin > 2000) std::this_thread::yield();
        fn_ = nullptr;
    }

private:
    // Static blocks: participant p always takes the p-th slice of the index range.  The per-task state (priority queue, batch,
    // memo: a few KB) then stays in the cache of the core that touched it in the previous tick; handing indices out dynamically
    // moved it between cores every tick and scaled worse than it looked (4 threads: 1.3x; static: see DESIGN.md §5).
    void drain(size_t p) {
        const size_t parts = threads_.size() + 1;
        const size_t a = n_ * p / parts, b = n_ * (p + 1) / parts;
        for (size_t i = a; i < b; ++i) (*fn_)(i);
    }
    void worker(size_t id) {
        uint64_t seen = 0;
        for (;;) {
            bool got = false;
            // ~50-100 us of polling before going to sleep — unless the pools of this process (one per solver: N ranks of an
            // fgoicp_multi in one process) already hold more threads than the machine has cores: spinning workers would then keep each
            // other's solvers off the CPU, so they sleep at once
            const int hw = (int)std::thread::hardware_concurrency();
            const int budget = hw > 0 && live_threads().load(std::memory_order_relaxed) > hw ? 0 : spin_budget_;
            for (int spin = 0; spin < budget; ++spin) {
                if (generation_.load(std::memory_order_seq_cst) != seen) { got = true; break; }
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#endif
            }
            if (!got) {
                std::unique_lock<std::mutex> lk(m_);
                sleepers_.fetch_add(1, std::memory_order_seq_cst);  // store-buffering pair with the poster's (generation_++, sleepers_ load): both sides sequentially consistent, or a wake-up can be lost
                cv_start_.wait(lk, [&] { return generation_.load(std::memory_order_seq_cst) != seen; });
                sleepers_.fetch_sub(1, std::memory_order_seq_cst);
            }
            seen = generation_.load(std::memory_order_seq_cst);
            if (stop_) return;
            drain(id);
            active_.fetch_sub(1, std::memory_order_release);
        }
    }
    static std::atomic<int>& live_threads() { static std::atomic<int> n{0}; return n; }  // pool threads alive in this process
    int nthreads_ = 1;
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_start_;
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t n_ = 0;
    std::atomic<size_t> active_{0};
    std::atomic<uint64_t> generation_{0};
    std::atomic<int> sleepers_{0};
    const int spin_budget_ = [] { const char* e = std::getenv("FGOICP_HOST_SPIN"); return e ? std::atoi(e) : 20000; }();  // tuning knob: 0 = workers sleep at once
    std::atomic<bool> stop_{false};
};

template <class Ops>
class GoIcpDriver {
public:
    GoIcpDriver(Ops& ops, size_t ns, float mse_threshold, int schedule, int round_width)
        : ops_(ops), sse_threshold_(ns * mse_threshold), ns_(ns), schedule_(schedule), round_width_(round_width < 0 ? 0 : round_width) {
        int nthreads = 4;
        if (const char* e = std::getenv("FGOICP_HOST_THREADS")) nthreads = std::atoi(e);
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0 && nthreads > hw) nthreads = hw;
        if (nthreads < 1) nthreads = 1;
        pool_.reset(new WorkerPool(nthreads));
    }

    void set_exchange(const Exchange& ex) { ex_ = ex; }
    // The reference's log lines as events (fgoicp.cpp:15-17 "Initial ICP best error", :85-87 "New best error" after EVERY triggered ICP,
    // improved or not): 0 = the initial ICP's (sse, R, t) as it returned them, 1 = the incumbent after a triggered ICP.  Called on the
    // thread that runs run(); under SERIAL in the reference's order.
    enum { kLogInitialIcp = 0, kLogNewBest = 1 };
    void set_log(std::function<void(int, float, const Mat3f&, const Vec3f&)> fn) { log_ = std::move(fn); }
    void set_use_cut(bool on) { use_cut_ = on; }
    const DriverStats& stats() const { return stats_; }

    float best_sse() const { std::lock_guard<std::mutex> g(mu_); return best_sse_; }
    void best_transform(Mat3f& R, Vec3f& t) const { std::lock_guard<std::mutex> g(mu_); R = best_R_; t = best_t_; }
    void last_transform(Mat3f& R, Vec3f& t) const { std::lock_guard<std::mutex> g(mu_); R = last_R_; t = last_t_; }

    // FastGoICP::run(), fgoicp.cpp:10-30 (without restore_translation: the caller owns the offsets)
    int run() {
        const auto t_start = clock::now();
        stats_ = DriverStats{};
        {
            std::lock_guard<std::mutex> g(mu_);
            best_sse_ = kHostInf;
            best_R_ = Mat3f::identity();
            best_t_ = Vec3f{0.f, 0.f, 0.f};
        }
        float sse;
        Mat3f R;
        Vec3f t;
        int rc = icp(Mat3f::identity(), Vec3f{0.f, 0.f, 0.f}, 0.05f, sse, R, t);  // :12-13
        if (rc) return rc;
        set_best_sse_only(sse);  // :14 — only the error is adopted
        stats_.initial_icp_sse = sse;
        if (log_) log_(kLogInitialIcp, sse, R, t);  // :15-17
        const auto t_bnb = clock::now();
        rc = schedule_ == kScheduleSerial ? bnb_so3_serial() : bnb_so3_round();
        stats_.seconds_bnb = seconds_since(t_bnb);
        if (rc) return rc;
        Mat3f bR;
        Vec3f bt;
        best_transform(bR, bt);
        rc = icp(bR, bt, 0.0005f, sse, R, t);  // :22-23
        if (rc) return rc;
        {
            std::lock_guard<std::mutex> g(mu_);
            best_sse_ = sse;
            best_R_ = R;
            best_t_ = t;
        }
        stats_.seconds_total = seconds_since(t_start);
        if (timing_)
            std::fprintf(stderr, "[fgoicp timing] run %.3f s: pop+pack %.3f s, operator %.3f s, push %.3f s, icp %.3f s, calls %llu\n", stats_.seconds_total, t_pop_, t_ops_,
                         t_push_, stats_.seconds_icp, (unsigned long long)stats_.bounds_calls);
        if (timing_) std::fprintf(stderr, "[fgoicp timing] prepare: pops %.3f s, pairs %.3f s, pack %.3f s\n", t_prep_[0], t_prep_[1], t_prep_[2]);
        t_pop_ = t_ops_ = t_push_ = 0;
        t_prep_[0] = t_prep_[1] = t_prep_[2] = 0;
        if (overlap_stats_) {
            std::fprintf(stderr, "[fgoicp overlap] UB-task nodes %llu, LB-task nodes %llu, in both %llu (%.1f %% of all subcubes could be served from the twin's evaluation)\n",
                         (unsigned long long)ov_ub_, (unsigned long long)ov_lb_, (unsigned long long)ov_both_, 100.0 * ov_both_ / std::max<double>(1.0, (double)(ov_ub_ + ov_lb_)));
            ov_ub_ = ov_lb_ = ov_both_ = 0;
        }
        return kDriverOk;
    }

private:
    using clock = std::chrono::steady_clock;
    static double seconds_since(clock::time_point t0) { return std::chrono::duration<double>(clock::now() - t0).count(); }

    int icp(const Mat3f& R0, const Vec3f& t0, float thr, float& sse, Mat3f& R, Vec3f& t, bool background = false) {
        const auto t_icp = clock::now();
        const float t03[3] = {t0.x, t0.y, t0.z};
        float t3[3];
        int iters = 0;
        int rc = background ? ops_.icp_background(R0.m, t03, 100, thr, &sse, R.m, t3, &iters)
               : coop()     ? ops_.icp_coop(ex_.rank, ex_.world, ex_.allgather_device, ex_.user, R0.m, t03, 100, thr, &sse, R.m, t3, &iters)
                            : ops_.icp(R0.m, t03, 100, thr, &sse, R.m, t3, &iters);
        t = Vec3f{t3[0], t3[1], t3[2]};
        stats_.icp_runs++;
        stats_.icp_iters += (uint64_t)iters;
        stats_.seconds_icp += seconds_since(t_icp);
        if (timing_) std::fprintf(stderr, "[fgoicp timing] icp thr %g: %d iterations, %.3f ms, sse %g\n", (double)thr, iters, seconds_since(t_icp) * 1e3, (double)sse);
        return rc;
    }
    // Cooperative refinements (world > 1 and the exchange can all-gather device memory): EVERY ICP of a run — the initial one, a
    // round's triggers, the final refinement — is then one run that all ranks execute together (Ops::icp_coop), at the same points of
    // the replicated control flow.  FGOICP_COOP_ICP: 0 = the round-2 flow (a rank refines its own children alone, against its own
    // running best), 1 = always cooperative, unset = by size: clouds of at least FGOICP_COOP_MIN_POINTS (131072) source points.
    // Measured on the 8-rank replay (profiles/r03_scale_replay.jsonl): the cooperative flow runs the one-GPU run's refinements and no
    // others (dragon shape: 41 ms of ICP in all, against 196 ms summed over the ranks of the private flow, whose slowest rank spends
    // 56 ms) but runs them on every rank one after another; the private flow runs more of them, in parallel.  437k points: 6.0x ->
    // 6.4x; 40k points (ICP iterations are latency chains of 50 us whatever the share): 5.6x -> 4.9x, hence the size rule.
    bool coop() const {
        if (ex_.world <= 1 || ex_.allgather_device == nullptr) return false;
        return coop_icp_ < 0 ? ns_ >= coop_min_points_ : coop_icp_ != 0;
    }
    void set_best_sse_only(float sse) { std::lock_guard<std::mutex> g(mu_); best_sse_ = sse; }
    void log_new_best() {
        if (!log_) return;
        float e; Mat3f R; Vec3f t;
        { std::lock_guard<std::mutex> g(mu_); e = best_sse_; R = best_R_; t = best_t_; }
        log_(kLogNewBest, e, R, t);
    }
    void set_last(const Mat3f& R, const Vec3f& t) { std::lock_guard<std::mutex> g(mu_); last_R_ = R; last_t_ = t; }

    // -------------------------------------------------------------------------------------------
    // SERIAL: fgoicp.cpp:32-100 verbatim in behaviour — same pops, same pushes, same counters — but not
    // verbatim in execution.  A node's evaluation (the UB and LB inner BnBs of its children) depends on
    // the search state only through `best_sse`, and that changes only when an ICP improves it (a handful
    // of times per run).  So the children of the node being popped AND of the nodes that top the queue
    // behind it are evaluated SPECULATIVELY in lock-step with the current `best_sse` (one operator
    // submission per tick for thousands of tasks instead of one per 32 subcubes), kept in a cache, and
    // committed when — and in the order in which — the reference pops them.  The moment an ICP improves
    // `best_sse` every cached result that has not been committed is discarded and re-evaluated.
    // Committed work is exactly the reference's; discarded work is not counted.  The speculation width
    // adapts like ROUND's: one node while the incumbent improves, doubling (up to 256) while it stands.
    // FGOICP_SERIAL_SPECULATE=1 speculates inside the popped node only, =0 executes literally.
    // -------------------------------------------------------------------------------------------
    struct SpecNode {
        std::vector<RotCube> kids;   // the children in the reference's j order
        std::vector<int> kind;       // 0 = push unevaluated (centre outside the ball), 1 = evaluate
        std::vector<Task> tk;        // per evaluated child: its UB task [2k] and LB task [2k+1]
        std::vector<char> valid;     // computed with the best_sse of `epoch`
        uint64_t epoch = 0;
    };
    using SpecKey = std::array<uint32_t, 4>;
    static SpecKey spec_key(const RotCube& n) {
        SpecKey k;
        std::memcpy(&k[0], &n.q.x, 4); std::memcpy(&k[1], &n.q.y, 4); std::memcpy(&k[2], &n.q.z, 4); std::memcpy(&k[3], &n.span, 4);
        return k;
    }
    static void spec_make_kids(const RotCube& rnode, SpecNode& sp) {
        sp = SpecNode();
        const float span = rnode.span / 2.0f;
        for (char j = 0; j < 8; ++j) {
            if (span < 0.05f) continue;  // :53
            RotCube child(rnode.q.x - span + (j >> 0 & 1) * rnode.span, rnode.q.y - span + (j >> 1 & 1) * rnode.span,
                          rnode.q.z - span + (j >> 2 & 1) * rnode.span, span, rnode.lb, rnode.ub);
            if (!child.overlaps_SO3()) continue;
            sp.kids.push_back(child);
            sp.kind.push_back(child.q.in_SO3() ? 1 : 0);
        }
        sp.tk.assign(2 * sp.kids.size(), Task());
        sp.valid.assign(2 * sp.kids.size(), 0);
    }

    int bnb_so3_serial() {
        static const int mode = [] { const char* e = dev_env("FGOICP_SERIAL_SPECULATE"); return e ? std::atoi(e) : 2; }();
        // width cap: 256 nodes per rank (the evaluations of a speculation are dealt over the ranks: the same tick sizes per rank at any world size)
        static const int spec_cap_env = [] { const char* e = dev_env("FGOICP_SERIAL_WIDTH"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : 0; }();  // tuning knob
        const int spec_cap = spec_cap_env > 0 ? spec_cap_env : 256 * (serial_sharded() ? ex_.world : 1);
        // how the width grows while the incumbent stands and where it restarts when it improves (tuning knobs; the trajectory does not depend on them)
        static const int spec_grow = [] { const char* e = dev_env("FGOICP_SERIAL_GROW"); const int v = e ? std::atoi(e) : 0; return v >= 2 && v <= 16 ? v : 2; }();
        static const int spec_start = [] { const char* e = dev_env("FGOICP_SERIAL_START"); const int v = e ? std::atoi(e) : 0; return v >= 1 && v <= 256 ? v : 1; }();
        std::priority_queue<RotCube> rcand;
        rcand.push(RotCube(0.f, 0.f, 0.f, 1.0f, 0.f, best_sse()));
        std::map<SpecKey, SpecNode> cache;
        uint64_t epoch = 1, epoch_of_last_speculation = 0;
        int width = 1;
        // (re)evaluates, in ONE task list, every evaluable task >= first_task[i] of batch[i] that is not valid
        auto evaluate = [&](const std::vector<SpecNode*>& batch, const std::vector<size_t>& first_task) -> int {
            std::vector<Task*> tasks;
            std::vector<const RotCube*> cubes;
            const float snap = best_sse();
            for (size_t b = 0; b < batch.size() && (mode != 0 || tasks.empty()); ++b) {
                SpecNode& sp = *batch[b];
                for (size_t t = first_task[b]; t < sp.tk.size(); ++t) {
                    if (sp.kind[t / 2] != 1 || sp.valid[t]) continue;
                    sp.tk[t] = Task();
                    sp.tk[t].start((t & 1) == 0, snap, sp.kids[t / 2].ub);
                    tasks.push_back(&sp.tk[t]);
                    cubes.push_back(&sp.kids[t / 2]);
                    sp.valid[t] = 1;
                    if (mode == 0) break;  // literal execution: one task at a time
                }
            }
            if (tasks.empty()) return kDriverOk;
            account_submissions_ = false;
            const int rc = serial_sharded() ? run_task_list_sharded(tasks, cubes) : run_task_list(tasks, cubes);
            account_submissions_ = true;
            return rc;
        };
        auto commit = [&](const Task& t) { stats_.inner_bnb++; stats_.trans_cubes += t.count; stats_.bounds_calls += t.batches; };
        while (!rcand.empty()) {
            const RotCube rnode = rcand.top();
            rcand.pop();
            stats_.rounds++;
            if (best_sse() - rnode.lb <= sse_threshold_) break;  // :44
            const SpecKey key = spec_key(rnode);
            {
                auto it = cache.find(key);
                if (it == cache.end() || it->second.epoch != epoch) {  // not evaluated yet, or evaluated against a stale best_sse
                    if (mode >= 2) width = epoch_of_last_speculation == epoch ? std::min(width * spec_grow, spec_cap) : spec_start;
                    epoch_of_last_speculation = epoch;
                    std::vector<SpecNode*> batch;
                    SpecNode& mine = cache[key];
                    spec_make_kids(rnode, mine);
                    mine.epoch = epoch;
                    batch.push_back(&mine);
                    if (mode >= 2 && width > 1) {  // the nodes the reference pops next if nothing changes: the tops of the queue
                        // A COPY of the queue is popped, on purpose: popping the real queue and pushing the nodes back would re-seat nodes
                        // with equal (lb, span) — unevaluated children inherit their parent's bounds, so such ties are common — and change
                        // the order in which the reference's heap pops them.  The copy is one memcpy of the heap's vector per cache miss
                        // (a miss every `width` pops; 1e5 nodes = 7.6 MB, ~1 ms), not a per-pop cost.
                        std::priority_queue<RotCube> peek = rcand;
                        while ((int)batch.size() < width && !peek.empty()) {
                            const RotCube n = peek.top();
                            peek.pop();
                            if (best_sse() - n.lb <= sse_threshold_) break;
                            const SpecKey kn = spec_key(n);
                            auto jt = cache.find(kn);
                            if (jt != cache.end() && jt->second.epoch == epoch) continue;
                            SpecNode& sp = cache[kn];
                            spec_make_kids(n, sp);
                            sp.epoch = epoch;
                            batch.push_back(&sp);
                        }
                    }
                    const int rc = evaluate(batch, std::vector<size_t>(batch.size(), 0));
                    if (rc) return rc;
                }
            }
            SpecNode& sp = cache[key];
            for (size_t k = 0; k < sp.kids.size(); ++k) {
                RotCube& child = sp.kids[k];
                if (sp.kind[k] == 0) { rcand.push(child); continue; }  // :62-66
                stats_.rot_cubes++;
                if (!sp.valid[2 * k]) { int rc = evaluate({&sp}, {2 * k}); if (rc) return rc; }
                commit(sp.tk[2 * k]);
                const float ub = sp.tk[2 * k].best_ub;  // :69
                const Vec3f bt = sp.tk[2 * k].best_t;
                set_last(child.q.R, bt);  // :71-72
                if (ub < best_sse() * 1.8) {  // :74, double compare
                    float sse; Mat3f R; Vec3f t;
                    int rc = icp(child.q.R, bt, 0.005f, sse, R, t);
                    if (rc) return rc;
                    if (sse < best_sse()) {
                        { std::lock_guard<std::mutex> g(mu_); best_sse_ = sse; best_R_ = R; best_t_ = t; }
                        ++epoch;  // every uncommitted result saw a stale best_sse: this node's later tasks ...
                        sp.epoch = epoch;
                        for (size_t t2 = 2 * k + 1; t2 < sp.valid.size(); ++t2) sp.valid[t2] = 0;
                        for (auto it = cache.begin(); it != cache.end();)  // ... and every other cached node
                            it = it->second.epoch != epoch ? cache.erase(it) : std::next(it);
                    }
                    log_new_best();  // :85-87
                }
                if (!sp.valid[2 * k + 1]) { int rc = evaluate({&sp}, {2 * k + 1}); if (rc) return rc; }
                commit(sp.tk[2 * k + 1]);
                const float lb = sp.tk[2 * k + 1].best_ub;  // :90 — the LB pass' best_ub is the cube's lower bound
                if (lb >= best_sse()) continue;            // :92
                child.lb = lb;
                child.ub = ub;
                rcand.push(child);
            }
            cache.erase(key);
        }
        return kDriverOk;
    }

    // SERIAL ON N RANKS (round 3).  Everything the reference's trajectory consists of — the queue, the incumbent, the cache, the
    // commit order, the counters — is replicated and stays identical on every rank by construction; what is SHARDED is the only
    // expensive part, the speculative evaluation: the tasks of one `evaluate` call are dealt over the ranks by child (a child's UB and
    // LB task stay on one device: twins and the memo need both), every rank runs its share, and ONE all-gather returns each task's
    // outcome — exactly the fields a commit reads (best_ub, best_t, count, batches; fgoicp.cpp:69-72, :90, :132).  A task's inner BnB
    // does not depend on the company it keeps (SERIAL tasks pop the reference's 32 nodes whatever the list), so its outcome is the
    // one-GPU run's bit for bit, and so are all pops, pushes, counters and the result.  The refinements (:74-88) are cooperative runs
    // of all ranks (Ops::icp_coop) when the exchange can all-gather device memory, replicated runs otherwise: same bits either way.
    // The integer counters travel as 16-bit pieces in floats (exact; no transport may touch them).  FGOICP_SERIAL_SHARD=0: every rank
    // evaluates everything (replicated run).
    bool serial_sharded() const { return serial_shard_ && ex_.world > 1 && ex_.allgather != nullptr; }
    int run_task_list_sharded(std::vector<Task*>& tasks, const std::vector<const RotCube*>& cubes) {
        const int rank = ex_.rank, world = ex_.world;
        constexpr size_t F = 10;  // floats per task outcome
        std::vector<int> owner(tasks.size());
        std::vector<size_t> n_of((size_t)world, 0);
        int child = -1;
        const RotCube* prev = nullptr;
        for (size_t i = 0; i < tasks.size(); ++i) {
            if (cubes[i] != prev) { ++child; prev = cubes[i]; }
            owner[i] = child % world;
            n_of[(size_t)owner[i]]++;
        }
        const size_t cap = *std::max_element(n_of.begin(), n_of.end());
        std::vector<Task*> mine_t;
        std::vector<const RotCube*> mine_c;
        for (size_t i = 0; i < tasks.size(); ++i)
            if (owner[i] == rank) { mine_t.push_back(tasks[i]); mine_c.push_back(cubes[i]); }
        if (!mine_t.empty()) {
            const int rc = run_task_list(mine_t, mine_c);
            if (rc) return rc;
        }
        std::vector<float> send(cap * F, 0.f), recv(cap * F * (size_t)world, 0.f);
        for (size_t k = 0; k < mine_t.size(); ++k) {
            const Task& t = *mine_t[k];
            float* p = &send[F * k];
            p[0] = t.best_ub; p[1] = t.best_t.x; p[2] = t.best_t.y; p[3] = t.best_t.z;
            for (int j = 0; j < 3; ++j) {
                p[4 + j] = (float)((t.count >> (16 * j)) & 0xffffu);
                p[7 + j] = (float)((t.batches >> (16 * j)) & 0xffffu);
            }
        }
        if (ex_.allgather(send.data(), recv.data(), cap * F, ex_.user)) return kDriverExchangeFailed;
        std::vector<size_t> pos((size_t)world, 0);
        for (size_t i = 0; i < tasks.size(); ++i) {
            const int o = owner[i];
            const float* p = &recv[(size_t)o * cap * F + F * pos[(size_t)o]++];
            if (o == rank) continue;
            Task& t = *tasks[i];
            t.best_ub = p[0];
            t.best_t = Vec3f{p[1], p[2], p[3]};
            t.count = t.batches = 0;
            for (int j = 0; j < 3; ++j) {
                t.count |= (uint64_t)p[4 + j] << (16 * j);
                t.batches |= (uint64_t)p[7 + j] << (16 * j);
            }
            t.cand = std::priority_queue<TransCube>();
            t.done = true;
        }
        return kDriverOk;
    }

    // -------------------------------------------------------------------------------------------
    // ROUND: K parents per round, all children's UB+LB inner BnBs concurrently, children sharded
    // over ranks, one all-reduce(min) + one all-gather per round.
    // -------------------------------------------------------------------------------------------
    int bnb_so3_round() {
        std::priority_queue<RotCube> rcand;
        rcand.push(RotCube(0.f, 0.f, 0.f, 1.0f, 0.f, best_sse()));
        const int rank = ex_.rank, world = ex_.world < 1 ? 1 : ex_.world;
        std::vector<RotCube> children;
        // LATE-JOINING REFINEMENT (FGOICP_LATE_ICP = 1; built in round 3, measured, OFF by default).  On the 8-rank replay it LOSES —
        // bunny shape 5.55x -> 4.17x, dragon shape 6.12x -> 3.33x (profiles/r03_scale_replay.jsonl): a certify run has 5-8 rounds, the
        // first of which finds the incumbent; one round of lag means round 2 prunes and triggers against the initial ICP's error
        // (+20-25 % subcubes, 4-10x the ICP time: `ub < 1.8 best` holds for almost every child while `best` is stale).  The ICP runs a round triggers (fgoicp.cpp:74-88) fall on
        // the ranks whose children trigger them — one or two single runs of tens of milliseconds per round, on ONE rank, while the
        // others wait in the exchange (the named serial term of the 8-rank estimate).  With this on, a round's
        // triggers run in the background (own host thread, own ICP lane and streams of the context) while the rank goes on to the
        // exchange and to the NEXT round's bounds work; their result enters the exchange one round late.  Price: one round of weaker
        // pruning (tasks and pushes of round r+1 see the incumbent without round r's refinements).  The replicated state stays
        // identical on all ranks — refinements enter it only through exchanges, a last exchange after the loop collects the final
        // round's — and the result is an epsilon-optimal solution as before, reached through a different sequence of incumbents.
        struct LateJob {
            std::thread th;
            std::vector<std::tuple<Mat3f, Vec3f, float>> cands;  // (rotation, best translation, ub) in child order
            float sse = 0.f; Mat3f R{}; Vec3f t{0, 0, 0};
            int rc = kDriverOk;
            bool active = false;
            ~LateJob() { if (th.joinable()) th.join(); }
        } late;
        // folds a finished background job into (loc_sse, loc_R, loc_t)
        auto join_late = [&](float& ls, Mat3f& lR, Vec3f& lt) -> int {
            if (!late.active) return kDriverOk;
            late.th.join();
            late.active = false;
            if (late.rc) return late.rc;
            if (late.sse < ls) { ls = late.sse; lR = late.R; lt = late.t; }
            return kDriverOk;
        };
        // round_width 0 = adaptive: start at 32 cubes per rank, double after every round that leaves the incumbent
        // standing (the search is certifying: every cube below the threshold gap has to be expanded anyway, wide rounds
        // waste nothing and feed the device bigger ticks), fall back to the base width when the incumbent improves.
        const bool adaptive = round_width_ <= 0;
        const int base_width = adaptive ? 32 * world : round_width_;
        int width = base_width;
        while (!rcand.empty()) {
            const auto t_iter = clock::now();
            children.clear();
            int popped = 0;
            while (popped < width && !rcand.empty()) {
                if (best_sse() - rcand.top().lb <= sse_threshold_) break;  // :44 (the top is the global min lb)
                const RotCube rnode = rcand.top();
                rcand.pop();
                ++popped;
                const float span = rnode.span / 2.0f;
                for (char j = 0; j < 8; ++j) {
                    if (span < 0.05f) continue;
                    RotCube child(rnode.q.x - span + (j >> 0 & 1) * rnode.span, rnode.q.y - span + (j >> 1 & 1) * rnode.span,
                                  rnode.q.z - span + (j >> 2 & 1) * rnode.span, span, rnode.lb, rnode.ub);
                    if (!child.overlaps_SO3()) continue;
                    if (!child.q.in_SO3()) { rcand.push(child); continue; }
                    children.push_back(child);
                }
            }
            if (popped == 0) break;
            stats_.rounds++;
            const size_t nchild = children.size();
            // Children are dealt in BLOCKS of deal_block_ consecutive children (siblings: `children` holds a parent's children next to each other):
            // block b goes to rank b % world.  Block 1 = plain round-robin.  Siblings turn the cloud by nearly the same rotation, so their items
            // land in the same LUT cells of a tick's locality sort — what the one-GPU run gets for free from having every child in one tick.
            const size_t B = deal_block_, nblocks = (nchild + B - 1) / B;
            const size_t slots = ((nblocks + world - 1) / world) * B;  // per-rank child slots
            auto owner_of = [&](size_t i) { return (int)((i / B) % (size_t)world); };
            auto slot_of = [&](size_t i) { return (i / (B * (size_t)world)) * B + i % B; };
            const float snapshot = best_sse();

            // my children, in child order (slot_of is increasing along it)
            std::vector<size_t> mine;
            for (size_t i = 0; i < nchild; ++i)
                if (owner_of(i) == rank) mine.push_back(i);
            std::vector<Task> boxes(2 * mine.size());
            std::vector<Task*> tasks;
            std::vector<const RotCube*> cubes;
            for (size_t k = 0; k < mine.size(); ++k) {
                boxes[2 * k].batch_cap = boxes[2 * k + 1].batch_cap = round_batch_;
                boxes[2 * k].start(true, snapshot, children[mine[k]].ub);
                boxes[2 * k + 1].start(false, snapshot, children[mine[k]].ub);
                tasks.push_back(&boxes[2 * k]); cubes.push_back(&children[mine[k]]);
                tasks.push_back(&boxes[2 * k + 1]); cubes.push_back(&children[mine[k]]);
                stats_.inner_bnb += 2;
                stats_.rot_cubes++;
            }
            const auto t_round = clock::now();
            const uint64_t calls_before = stats_.bounds_calls, cubes_before = stats_.trans_cubes;
            const double icp_before = stats_.seconds_icp;
            // OVERLAPPED REFINEMENTS (round 4).  The triggers of a round (fgoicp.cpp:74-88) read nothing but the UB tasks' results — child by child,
            // against the running best — and the LB tasks of a ROUND round read nothing of the refinements (they prune against the round-start error,
            // see above).  So the refinements do not have to wait for the LB tasks, which are five times the work: a second host thread applies the
            // rule in child order as the UB tasks end and runs each triggered ICP on an ICP lane of its own (own scratch and streams) while this
            // thread keeps ticking the remaining tasks.  Same rule, same order, same inputs: the bits of the sequential flow; the round ends when both
            // have ended.  (Not to be confused with round 3's late-joining variant, which let the refinements LAG a round and lost.)
            // MEASURED (profiles/r04_ab_overlap_icp.txt): bit-identical, and no gain — default-threshold bunny step 15.3 -> 15.1 ms, headline 0.351 -> 0.350 s,
            // trimmed 1M and dragon within noise.  The timing lines say why: with the tail batches the UB tasks of a small round need as many device round trips as
            // the LB tasks (8 of the round's 8.5 per half), so the first trigger is known 5.4 ms into a 5.7-ms task phase, and in big rounds the refinements are
            // 3 % of the time.  OFF by default (development knob FGOICP_OVERLAP_ICP=1).
            struct Overlapped {
                std::thread th;
                std::atomic<size_t> ub_ready{0};   // UB tasks of my children [0, ub_ready) have ended
                std::atomic<bool> stop{false};
                float sse = 0.f; Mat3f R{}; Vec3f t{0, 0, 0};
                int rc = kDriverOk;
                bool active = false;
                ~Overlapped() { stop.store(true); if (th.joinable()) th.join(); }
            } ov;
            const bool overlapped = overlap_icp_ && !coop() && !(late_icp_ > 0 && (world > 1 || late_icp_ > 1)) && !mine.empty();
            if (overlapped) {
                { std::lock_guard<std::mutex> g(mu_); ov.sse = best_sse_; ov.R = best_R_; ov.t = best_t_; }
                ov.active = true;
                tick_hook_ = [&] {
                    size_t k = ov.ub_ready.load(std::memory_order_relaxed);
                    while (k < mine.size() && boxes[2 * k].done) ++k;
                    ov.ub_ready.store(k, std::memory_order_release);
                };
                ov.th = std::thread([&] {
                    for (size_t k = 0; k < mine.size(); ++k) {
                        for (unsigned spins = 0; ov.ub_ready.load(std::memory_order_acquire) <= k; ++spins) {
                            if (ov.stop.load(std::memory_order_acquire)) return;
                            if (spins < 64) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(10));
                        }
                        const RotCube& ch = children[mine[k]];
                        const float ub = boxes[2 * k].best_ub;
                        const Vec3f bt = boxes[2 * k].best_t;
                        set_last(ch.q.R, bt);
                        if (timing_ && mine.size() <= 64) std::fprintf(stderr, "[fgoicp timing] overlapped: UB task of child %zu ended %.3f ms into the round (%llu batches)\n", k, seconds_since(t_round) * 1e3, (unsigned long long)boxes[2 * k].batches);
                        if (ub < ov.sse * 1.8) {  // fgoicp.cpp:74
                            float sse; Mat3f R; Vec3f t;
                            const int r = icp(ch.q.R, bt, 0.005f, sse, R, t, true);
                            if (r) { ov.rc = r; return; }
                            if (sse < ov.sse) { ov.sse = sse; ov.R = R; ov.t = t; }
                            if (log_) log_(kLogNewBest, ov.sse, ov.R, ov.t);  // :85-87, this rank's running best
                        }
                    }
                });
            }
            // UB TASKS FIRST in a small round (round 4, FGOICP_UB_FIRST, development knob — see the measurement below): a round of a few cubes is a chain of
            // device round trips, every UB task needs one per level of its translation tree (5-8), and next to the LB tasks a round trip carries five times the
            // rows.  Run alone, the UB tasks end in a fraction of the task phase; the triggers are then known and the refinements run on the second thread
            // (above) next to the LB tasks.  The twin / memo sharing between a cube's two tasks is given up for such a round (a few hundred subcubes).
            // MEASURED (profiles/r04_ab_ub_first.txt): nothing — default-threshold step 15.3 -> 15.3-15.5 ms, headline and trimmed within noise.  A tick of a few
            // hundred subcubes is not shorter than one of a thousand: its 245 us are the chain of eight dependent launches (upload, keys, fold, scan, scatter, check,
            // bounds, finalize), so the UB tasks alone still need their 5-8 round trips of that length and the LB tasks theirs AFTER them.
            const bool ub_first = overlapped && ub_first_max_ > 0 && mine.size() <= ub_first_max_;
            int rc = kDriverOk;
            if (ub_first) {
                std::vector<Task*> ubt, lbt;
                std::vector<const RotCube*> ubc, lbc;
                for (size_t k = 0; k < mine.size(); ++k) {
                    ubt.push_back(tasks[2 * k]); ubc.push_back(cubes[2 * k]);
                    lbt.push_back(tasks[2 * k + 1]); lbc.push_back(cubes[2 * k + 1]);
                }
                tick_hook_ = nullptr;  // (the hook is keyed to the combined task list)
                rc = run_task_list(ubt, ubc);
                ov.ub_ready.store(rc ? 0 : mine.size(), std::memory_order_release);
                if (!rc) rc = run_task_list(lbt, lbc);
            } else {
                rc = run_task_list(tasks, cubes);
            }
            if (overlapped) {
                if (!ub_first) tick_hook_();   // every task has ended
                tick_hook_ = nullptr;
                if (rc) ov.stop.store(true);
                ov.th.join();
                if (!rc) rc = ov.rc;
            }
            if (rc) return rc;
            const double s_tasks = seconds_since(t_round);

            // ICP triggers in child order against the running local best (fgoicp.cpp:74-88), one run after another: a successful run
            // tightens the trigger of the next child, and after the first success most children no longer qualify.  (Refining every
            // child that COULD trigger in one concurrent batch — fgoicp_icp_batch — and replaying the rule over the results was
            // measured and lost: the default-threshold bunny step 38.1 -> 40.5 ms, ICP 17.5 -> 24.3 ms; the batch runs the
            // candidates the rule would have skipped, and its length is that of its longest run.)
            float loc_sse; Mat3f loc_R; Vec3f loc_t;
            { std::lock_guard<std::mutex> g(mu_); loc_sse = best_sse_; loc_R = best_R_; loc_t = best_t_; }
            std::vector<float> lbs(nchild), ubs(nchild);
            if (coop()) {
                // COOPERATIVE ROUND (round 3): the bounds are exchanged FIRST — {lb, ub, best translation} per child, one all-gather —, then
                // every rank applies the trigger rule to ALL children in the single-GPU child order against the same running best, and
                // each triggered refinement is one ICP run that all ranks execute together (ctx_icp_coop: every rank scans 1 / world of
                // the source, results identical on every rank).  No rank waits for another rank's private ICP (the serial term of the
                // round-2 flow: one 35-40 ms run on one of eight ranks), no candidate needs to be exchanged, and the sequence of
                // incumbents is the one-rank run's.
                const size_t per = 5 * slots;
                std::vector<float> send(per, 0.f), recv(per * world, 0.f);
                for (size_t k = 0; k < mine.size(); ++k) {
                    float* p = &send[5 * slot_of(mine[k])];
                    p[0] = boxes[2 * k + 1].best_ub;  // LB pass: its best_ub is the cube's lower bound (:90)
                    p[1] = boxes[2 * k].best_ub;      // UB pass
                    p[2] = boxes[2 * k].best_t.x; p[3] = boxes[2 * k].best_t.y; p[4] = boxes[2 * k].best_t.z;
                }
                if (!ex_.allgather) return kDriverExchangeFailed;
                if (nchild > 0 && ex_.allgather(send.data(), recv.data(), per, ex_.user)) return kDriverExchangeFailed;  // (a round may consist of cubes pushed unevaluated)
                for (size_t i = 0; i < nchild; ++i) {
                    const float* p = &recv[per * (size_t)owner_of(i)] + 5 * slot_of(i);
                    lbs[i] = p[0];
                    ubs[i] = p[1];
                    const Vec3f bt{p[2], p[3], p[4]};
                    set_last(children[i].q.R, bt);
                    if (ubs[i] < loc_sse * 1.8) {  // fgoicp.cpp:74
                        float sse; Mat3f R; Vec3f t;
                        rc = icp(children[i].q.R, bt, 0.005f, sse, R, t);
                        if (rc) return rc;
                        if (sse < loc_sse) { loc_sse = sse; loc_R = R; loc_t = t; }
                        if (log_) log_(kLogNewBest, loc_sse, loc_R, loc_t);  // :85-87, the running best of the round
                    }
                }
                if (loc_sse < best_sse()) { std::lock_guard<std::mutex> g(mu_); best_sse_ = loc_sse; best_R_ = loc_R; best_t_ = loc_t; }
            } else {
            if (late_icp_ > 0 && (world > 1 || late_icp_ > 1)) {
                rc = join_late(loc_sse, loc_R, loc_t);  // the refinements of the round before: they join THIS exchange
                if (rc) return rc;
                late.cands.clear();
                for (size_t k = 0; k < mine.size(); ++k) {
                    const RotCube& ch = children[mine[k]];
                    set_last(ch.q.R, boxes[2 * k].best_t);
                    if (boxes[2 * k].best_ub < loc_sse * 1.8) late.cands.emplace_back(ch.q.R, boxes[2 * k].best_t, boxes[2 * k].best_ub);  // could trigger at all
                }
                if (!late.cands.empty()) {
                    late.sse = loc_sse; late.R = loc_R; late.t = loc_t; late.rc = kDriverOk;
                    late.active = true;
                    late.th = std::thread([this, &late] {  // the rule of fgoicp.cpp:74-88, in child order against the job's running best
                        for (const auto& c : late.cands) {
                            if (!(std::get<2>(c) < late.sse * 1.8)) continue;
                            float sse; Mat3f R; Vec3f t;
                            const int r = icp(std::get<0>(c), std::get<1>(c), 0.005f, sse, R, t, true);
                            if (r) { late.rc = r; return; }
                            if (sse < late.sse) { late.sse = sse; late.R = R; late.t = t; }
                        }
                    });
                }
            } else if (ov.active) {
                loc_sse = ov.sse; loc_R = ov.R; loc_t = ov.t;  // the refinements ran next to the tasks (above)
            } else
            for (size_t k = 0; k < mine.size(); ++k) {
                const RotCube& ch = children[mine[k]];
                const float ub = boxes[2 * k].best_ub;
                const Vec3f bt = boxes[2 * k].best_t;
                set_last(ch.q.R, bt);
                if (ub < loc_sse * 1.8) {
                    float sse; Mat3f R; Vec3f t;
                    rc = icp(ch.q.R, bt, 0.005f, sse, R, t);
                    if (rc) return rc;
                    if (sse < loc_sse) { loc_sse = sse; loc_R = R; loc_t = t; }
                    if (log_) log_(kLogNewBest, loc_sse, loc_R, loc_t);  // :85-87, this rank's running best
                }
            }

            // exchange: best error (min-all-reduce) + {candidate transform, child bounds} (all-gather)
            if (world > 1) {
                float gmin = loc_sse;
                if (!ex_.allreduce_min || !ex_.allgather) return kDriverExchangeFailed;
                if (ex_.allreduce_min(&gmin, 1, ex_.user)) return kDriverExchangeFailed;
                const size_t per = 13 + 2 * slots;
                std::vector<float> send(per, 0.f), recv(per * world, 0.f);
                send[0] = loc_sse;
                std::memcpy(&send[1], loc_R.m, sizeof(float) * 9);
                send[10] = loc_t.x; send[11] = loc_t.y; send[12] = loc_t.z;
                for (size_t k = 0; k < mine.size(); ++k) {
                    send[13 + 2 * slot_of(mine[k])] = boxes[2 * k + 1].best_ub;  // LB pass: its best_ub is the cube's lower bound (:90)
                    send[13 + 2 * slot_of(mine[k]) + 1] = boxes[2 * k].best_ub;  // UB pass
                }
                if (ex_.allgather(send.data(), recv.data(), per, ex_.user)) return kDriverExchangeFailed;
                for (int r = 0; r < world; ++r) {  // lowest rank holding the global minimum wins
                    const float* p = &recv[per * r];
                    if (p[0] == gmin && gmin < best_sse()) {
                        std::lock_guard<std::mutex> g(mu_);
                        best_sse_ = gmin;
                        std::memcpy(best_R_.m, p + 1, sizeof(float) * 9);
                        best_t_ = Vec3f{p[10], p[11], p[12]};
                        break;
                    }
                }
                for (size_t i = 0; i < nchild; ++i) {
                    const float* p = &recv[per * (size_t)owner_of(i)] + 13 + 2 * slot_of(i);
                    lbs[i] = p[0];
                    ubs[i] = p[1];
                }
            } else {
                if (loc_sse < best_sse()) { std::lock_guard<std::mutex> g(mu_); best_sse_ = loc_sse; best_R_ = loc_R; best_t_ = loc_t; }
                for (size_t k = 0; k < mine.size(); ++k) {
                    lbs[mine[k]] = boxes[2 * k + 1].best_ub;
                    ubs[mine[k]] = boxes[2 * k].best_ub;
                }
            }
            }  // !coop()
            const float now = best_sse();
            if (timing_)
                std::fprintf(stderr, "[fgoicp timing] round %llu: popped %d, children %zu (mine %zu), submissions %llu, subcubes %llu, tasks %.3f ms, icp %.3f ms, round %.3f ms, setup %.3f ms\n",
                             (unsigned long long)stats_.rounds, popped, nchild, mine.size(), (unsigned long long)(stats_.bounds_calls - calls_before),
                             (unsigned long long)(stats_.trans_cubes - cubes_before), s_tasks * 1e3, (stats_.seconds_icp - icp_before) * 1e3, seconds_since(t_round) * 1e3,
                             std::chrono::duration<double>(t_round - t_iter).count() * 1e3);
            if (adaptive) width = now < snapshot ? base_width : std::min(width * 2, 1 << 14);
            for (size_t i = 0; i < nchild; ++i) {
                if (lbs[i] >= now) continue;  // :92
                children[i].lb = lbs[i];
                children[i].ub = ubs[i];
                rcand.push(children[i]);
            }
        }
        if (!coop() && late_icp_ > 0 && (world > 1 || late_icp_ > 1)) {
            // the last round's refinements: joined and agreed on by one more exchange (every rank gets here after the same round —
            // the loop's decisions depend on the replicated state only)
            float loc_sse; Mat3f loc_R; Vec3f loc_t;
            { std::lock_guard<std::mutex> g(mu_); loc_sse = best_sse_; loc_R = best_R_; loc_t = best_t_; }
            int rc = join_late(loc_sse, loc_R, loc_t);
            if (rc) return rc;
            if (world > 1) {
                float gmin = loc_sse;
                if (ex_.allreduce_min(&gmin, 1, ex_.user)) return kDriverExchangeFailed;
                std::vector<float> send(13, 0.f), recv((size_t)13 * world, 0.f);
                send[0] = loc_sse;
                std::memcpy(&send[1], loc_R.m, sizeof(float) * 9);
                send[10] = loc_t.x; send[11] = loc_t.y; send[12] = loc_t.z;
                if (ex_.allgather(send.data(), recv.data(), 13, ex_.user)) return kDriverExchangeFailed;
                for (int r = 0; r < world; ++r) {
                    const float* p = &recv[(size_t)13 * r];
                    if (p[0] == gmin && gmin < best_sse()) {
                        std::lock_guard<std::mutex> g(mu_);
                        best_sse_ = gmin;
                        std::memcpy(best_R_.m, p + 1, sizeof(float) * 9);
                        best_t_ = Vec3f{p[10], p[11], p[12]};
                        break;
                    }
                }
            } else if (loc_sse < best_sse()) {
                std::lock_guard<std::mutex> g(mu_);
                best_sse_ = loc_sse; best_R_ = loc_R; best_t_ = loc_t;
            }
        }
        return kDriverOk;
    }

    // Advance a set of inner tasks in lock-step: every tick submits the current batch of every live
    // task in ONE operator call (G groups) and hands each task its slice of the results.
    // One half of a round's tasks: its current batches packed for one operator submission.
    struct Half {
        std::vector<size_t> members;   // indices into tasks
        std::vector<int> live;         // members with a batch in this submission
        std::vector<float> R9, spans, tn4, lb, ub;
        std::vector<int> fix, offsets;
        std::vector<float> cut;        // per group: InnerTask::cut_above() of its task
        std::vector<int> twin;         // per subcube: the same translation node in the paired task's batch (or -1)
        std::vector<std::vector<std::pair<int, int>>> pair_twins;  // per (UB, LB) pair: {row in the UB group, row in the LB group}
        bool inflight = false;
    };

    // Pops the next batch of every task of the half and packs them; false if none is left.
    //
    // The UB task and the LB task of one child (tasks 2c, 2c+1, neighbours in `live`) walk the same translation tree under the same
    // rotation; 85 % of the UB task's nodes are also evaluated by the (five times larger) LB task.  Two savings, both leaving every
    // task's own sequence of batches, bounds and counters untouched:
    //   twins   a node both hold in THIS submission is evaluated once — one lookup per point, both variants of the bound formulae
    //           (fgoicp_bounds_submit_twins);
    //   memo    every other node of the UB batch is evaluated that way too: a "phantom" row in the LB task's group receives the LB
    //           variant's sums and goes into the LB task's memo; when the LB task pops that node later it is served from the memo
    //           (a batch served entirely is consumed on the spot, without a device round trip).  The sums depend on the node and the
    //           rotation only, and the dual evaluation is bit-identical to a separate one, so the LB task cannot tell.
    bool prepare_half(Half& h, std::vector<Task*>& tasks, const std::vector<const RotCube*>& cubes, bool par, bool twins_honoured) {
        const bool memo_on = use_memo_ && use_twins_ && twins_honoured;  // phantom rows cost nothing only where the twin hint is used
        const auto tp0 = clock::now();
        // ROUND, tail of a round: a few long-running tasks are left and every tick is a latency-bound device round trip that fills a
        // fraction of the GPU.  Such a half takes bigger batches (the extra nodes are the next-best of the task's own queue: some
        // would have been pruned by the results of the batch before — paid with idle capacity — and the round ends in fewer ticks).
        // the fewer tasks a half holds, the more of a tick is round-trip latency and the bigger the batch may be: 128 nodes from 32 tasks
        // down, 256 from 16, 512 from 8 (measured: default-threshold bunny step 17.3 -> 15.6 ms, the 16 tasks of its one round; certify runs
        // unchanged; profiles/r03_ab_tail_batch.txt) — for clouds of up to 200 000 points: on the dragon shape every extra node is 437k
        // point evaluations on a device that is full anyway (512: +2 % subcubes and time).  FGOICP_TAIL_BATCH fixes the size.
        size_t tail_cap = 0;
        if (tail_batch_ && h.members.size() <= tail_tasks_) {
            tail_cap = tail_batch_;
            if (!tail_batch_fixed_ && ns_ <= 200000) tail_cap = h.members.size() <= tail_tasks_ / 4 ? 512 : h.members.size() <= tail_tasks_ / 2 ? 256 : tail_batch_;
        }
        // the same rule made continuous (FGOICP_TICK_ROWS = R > 0): a half's batches are sized so that a tick carries about R rows — batch =
        // R / live tasks, between ROUND's 48 and 512 — which also reaches the middle of a round on a rank of many (a few hundred tasks per
        // half: ticks of 4 000 rows at 275 ns per evaluation where the one-GPU run's big ticks cost 207)
        if (tick_rows_ > 0 && tail_batch_ && !tail_batch_fixed_ && ns_ <= 200000 && !h.members.empty()) {
            const size_t b = std::min<size_t>(512, tick_rows_ / h.members.size());
            tail_cap = b > round_batch_ ? b : 0;
        }
        const std::function<void(size_t)> pop_fn = [&](size_t k) {
            Task& tk = *tasks[h.members[k]];
            if (tk.batch_cap != 32) tk.batch_cap = tail_cap ? tail_cap : round_batch_;  // SERIAL tasks keep the reference's 32 (fgoicp.cpp:122)
            tk.has_batch = false;
            tk.phantom.clear();
            if (tk.done) return;
            for (;;) {
                if (!tk.next_batch(sse_threshold_)) { tk.done = true; return; }
                tk.batches++;
                if (overlap_stats_)
                    for (const TransCube& c : tk.batch) tk.seen.push_back(node_key(c));
                const size_t n = tk.batch.size();
                tk.brow.resize(n); tk.blb.resize(n); tk.bub.resize(n);
                tk.nrows = 0;
                for (size_t q = 0; q < n; ++q) {
                    if (!tk.memo.empty()) {
                        if (const auto* hit = tk.memo.find(node_key(tk.batch[q]))) { tk.blb[q] = hit->first; tk.bub[q] = hit->second; tk.brow[q] = -1; continue; }
                    }
                    tk.brow[q] = tk.nrows++;
                }
                if (tk.nrows > 0) break;
                tk.consume(tk.blb.data(), tk.bub.data());  // the whole batch came from the memo
            }
            tk.has_batch = true;
            // SERIAL, tail of an evaluation (round 3): a SERIAL task pops the reference's 32 nodes per operator call (fgoicp.cpp:122) — that is part
            // of the trajectory — so the tail of an evaluation is a long chain of round trips with a few tasks x 32 nodes each (tick log:
            // 36 % of the kernel time of a SERIAL certify run sits in ticks of 256-1024 evaluations at 350 instead of 217 ns each).
            // LOOK-AHEAD: the nodes the task pops NEXT if nothing changes — the tops of a COPY of its queue (the real heap is not touched:
            // re-seating equal keys would change the pop order) — ride along as phantom rows of its own group; their bounds go into the
            // task's memo, and a later batch that the memo serves entirely is consumed without a device round trip (the mechanism the LB
            // tasks use for their twins' nodes).  The task's pops, pushes, counters and bounds are what they were: a node's sums do not
            // depend on the tick it is evaluated in; look-ahead nodes that are never popped are wasted work, not counted.
            // Measured (profiles/r03_ab_serial_ahead.txt): bunny-shape certify run 450 -> 377 ms (launches 1298 -> 410, evaluations + 5 %), the same run
            // on 8 ranks 3.5x -> 4.6x; look-ahead up to 512 live tasks per half (480 / 224 / 96 nodes from a quarter / half / all of that down) —
            // beyond that the wasted evaluations cost more than the round trips saved (2048 tasks: 414 ms), and on the dragon shape, where a node
            // is 437k point evaluations on a device that is full anyway, 32 tasks is the limit (1.63 -> 1.59 s; 512: 1.78 s).
            // (clouds below 16 384 points: off — such runs are bound by the host's queue work, not by device round trips: on the shape of
            // test/bunny.toml, 3 037 source points, the queue copies and memo look-ups add 15 ms of pops to a 100-ms run and save nothing)
            const size_t ahead_tasks = serial_ahead_tasks_ ? serial_ahead_tasks_ : (ns_ < 16384 ? 0 : ns_ <= 200000 ? 512 : 32);
            if (serial_ahead_ > 0 && tk.batch_cap == 32 && h.members.size() <= ahead_tasks && !tk.cand.empty()) {
                const size_t want = h.members.size() <= ahead_tasks / 4 ? 15 * 32 : h.members.size() <= ahead_tasks / 2 ? 7 * 32 : 3 * 32;
                std::priority_queue<TransCube> peek = tk.cand;
                for (size_t got = 0; got < std::min(want, (size_t)serial_ahead_) && !peek.empty();) {
                    const TransCube tn = peek.top();
                    peek.pop();
                    if (tk.best_error - tn.lb < sse_threshold_) break;  // the task would stop here (:120)
                    if (!(tn.lb < tk.best_error)) continue;             // popped and dropped (:126)
                    if (tk.memo.find(node_key(tn))) continue;
                    tk.phantom.push_back(tn);
                    ++got;
                }
            }
        };
        if (par) pool_->parallel_for(h.members.size(), pop_fn);
        else for (size_t k = 0; k < h.members.size(); ++k) pop_fn(k);
        const auto tp1 = clock::now();
        h.live.clear();
        h.members.erase(std::remove_if(h.members.begin(), h.members.end(), [&](size_t i) { return tasks[i]->done; }), h.members.end());
        for (size_t i : h.members) h.live.push_back((int)i);  // a member that is not done has a batch
        const size_t G = h.live.size();
        // pairs: (UB task, LB task) of one child, neighbours in `live`.  Per pair: the twins (local rows) and the LB group's phantoms.
        std::vector<size_t> pairs;
        if (use_twins_)
            for (size_t a = 0; a + 1 < G; ++a)
                if ((h.live[a] ^ 1) == h.live[a + 1] && cubes[h.live[a]] == cubes[h.live[a + 1]] && tasks[h.live[a]]->fix_rot && !tasks[h.live[a + 1]]->fix_rot) pairs.push_back(a);
        h.pair_twins.assign(pairs.size(), {});
        const std::function<void(size_t)> match_fn = [&](size_t q) {
            Task &ub = *tasks[h.live[pairs[q]]], &lb = *tasks[h.live[pairs[q] + 1]];
            auto& tw = h.pair_twins[q];
            const size_t n0 = ub.batch.size(), n1 = lb.batch.size();
            if (n0 > 512 || n1 > 512) return;  // batches hold <= 512 nodes (32 in the reference, fgoicp.cpp:122); the table below assumes it
            int table[1024];
            for (int& x : table) x = -1;
            for (size_t j = 0; j < n1; ++j) {
                if (lb.brow[j] < 0) continue;
                uint32_t sl = (uint32_t)NodeKeyHash()(node_key(lb.batch[j])) & 1023u;
                while (table[sl] >= 0) sl = (sl + 1) & 1023u;
                table[sl] = (int)j;
            }
            for (size_t i = 0; i < n0; ++i) {
                if (ub.brow[i] < 0) continue;  // served from the UB task's own memo (look-ahead): no row of this submission to pair with
                const NodeKey ki = node_key(ub.batch[i]);
                int hit = -1;
                for (uint32_t sl = (uint32_t)NodeKeyHash()(ki) & 1023u; table[sl] >= 0; sl = (sl + 1) & 1023u)
                    if (node_key(lb.batch[(size_t)table[sl]]) == ki) { hit = table[sl]; break; }
                if (hit >= 0) tw.push_back({ub.brow[i], lb.brow[(size_t)hit]});
                else if (memo_on && !lb.memo.find(ki)) {
                    tw.push_back({ub.brow[i], lb.nrows + (int)lb.phantom.size()});
                    lb.phantom.push_back(ub.batch[i]);
                }
            }
        };
        if (par && pairs.size() >= 64) pool_->parallel_for(pairs.size(), match_fn);
        else for (size_t q = 0; q < pairs.size(); ++q) match_fn(q);
        const auto tp2 = clock::now();
        h.offsets.assign(1, 0);
        for (size_t a = 0; a < G; ++a) h.offsets.push_back(h.offsets.back() + tasks[h.live[a]]->nrows + (int)tasks[h.live[a]]->phantom.size());
        const size_t total = (size_t)h.offsets.back();
        h.R9.resize(9 * G); h.spans.resize(G); h.fix.resize(G); h.tn4.resize(4 * total);
        h.cut.resize(G);
        const std::function<void(size_t)> pack_fn = [&](size_t a) {  // groups are independent: packed in parallel
            const int i = h.live[a];
            const Task& tk = *tasks[i];
            std::memcpy(&h.R9[9 * a], cubes[i]->q.R.m, 9 * sizeof(float));
            h.spans[a] = cubes[i]->span;
            h.fix[a] = tk.fix_rot ? 1 : 0;
            h.cut[a] = tk.cut_above();
            float* base = &h.tn4[4 * (size_t)h.offsets[a]];
            for (size_t q = 0; q < tk.batch.size(); ++q) {
                if (tk.brow[q] < 0) continue;
                const TransCube& c = tk.batch[q];
                float* out = base + 4 * (size_t)tk.brow[q];
                out[0] = c.t.x; out[1] = c.t.y; out[2] = c.t.z; out[3] = c.span;
            }
            float* out = base + 4 * (size_t)tk.nrows;
            for (const TransCube& c : tk.phantom) { out[0] = c.t.x; out[1] = c.t.y; out[2] = c.t.z; out[3] = c.span; out += 4; }
        };
        if (par && G >= 256) pool_->parallel_for(G, pack_fn);
        else for (size_t a = 0; a < G; ++a) pack_fn(a);
        h.lb.resize(total);
        h.ub.resize(total);
        h.twin.assign(total, -1);
        for (size_t q = 0; q < pairs.size(); ++q) {
            const int o0 = h.offsets[pairs[q]], o1 = h.offsets[pairs[q] + 1];
            for (const auto& pr : h.pair_twins[q]) { h.twin[(size_t)(o0 + pr.first)] = o1 + pr.second; h.twin[(size_t)(o1 + pr.second)] = o0 + pr.first; }
        }
        if (timing_) {
            const auto tp3 = clock::now();
            t_prep_[0] += std::chrono::duration<double>(tp1 - tp0).count();
            t_prep_[1] += std::chrono::duration<double>(tp2 - tp1).count();
            t_prep_[2] += std::chrono::duration<double>(tp3 - tp2).count();
        }
        return !h.live.empty();
    }
    // Tasks finish at very different times (an UB task may need 5 batches, its neighbour 50): when the idle half `from`
    // holds many more unfinished tasks than the other, it hands over the surplus so that both slots stay busy and the
    // device never waits for the host.  A task's own sequence of batches does not depend on the half it sits in.
    static void rebalance(Half& from, Half& to) {
        if (from.members.size() < to.members.size() + 2 + to.members.size() / 4) return;
        const size_t move = (from.members.size() - to.members.size()) / 2;
        // every other UNIT from the back; a unit is a (UB, LB) pair of one child (indices 2c, 2c+1) that is still together, or a
        // single task: pairs stay in one half, where their common subcubes can be evaluated once
        std::vector<size_t> keep;
        keep.reserve(from.members.size());
        size_t moved = 0, unit = 0;
        for (size_t k = from.members.size(); k-- > 0;) {
            const bool pair = k > 0 && (from.members[k] ^ 1) == from.members[k - 1];
            const size_t n = pair ? 2 : 1;
            const bool go = moved + n <= move && (unit & 1) == 0;  // never more than the surplus: a lone pair stays where it is
            for (size_t q = 0; q < n; ++q) {
                if (go) { to.members.push_back(from.members[k - (n - 1) + q]); ++moved; }
                else keep.push_back(from.members[k - q]);
            }
            if (pair) --k;
            ++unit;
        }
        std::reverse(keep.begin(), keep.end());
        from.members.swap(keep);
    }
    void consume_half(Half& h, std::vector<Task*>& tasks, bool par) {
        const std::function<void(size_t)> push_fn = [&](size_t k) {
            Task& tk = *tasks[h.live[k]];
            const float *lb = h.lb.data() + h.offsets[k], *ub = h.ub.data() + h.offsets[k];
            for (size_t q = 0; q < tk.batch.size(); ++q)
                if (tk.brow[q] >= 0) { tk.blb[q] = lb[tk.brow[q]]; tk.bub[q] = ub[tk.brow[q]]; }
            for (size_t j = 0; j < tk.phantom.size(); ++j)  // the twin's nodes, evaluated for later
                tk.memo.insert(node_key(tk.phantom[j]), lb[tk.nrows + (int)j], ub[tk.nrows + (int)j]);
            tk.consume(tk.blb.data(), tk.bub.data());
        };
        if (par) pool_->parallel_for(h.live.size(), push_fn);
        else for (size_t k = 0; k < h.live.size(); ++k) push_fn(k);
        if (account_submissions_) stats_.bounds_calls++;  // ROUND: one operator submission serves many tasks
    }

    // Advance a set of inner tasks to completion.  Every submission carries the current batch of every
    // live task of a half in ONE operator call (G groups).  With an asynchronous backend the tasks are
    // split into two halves on two slots: while the device evaluates one half, the host consumes the
    // results of the other and pops its next batches.  With the tail batches OFF (FGOICP_TAIL_BATCH=0) a task's own sequence of
    // batches is the same either way, so results do not depend on the mode.  With them on (ROUND's default) a task's batch size
    // depends on how many tasks share its half — on the mode, the rebalancing, the world size, the memo: the inner BnB stops on
    // `best_error - top.lb < threshold` (fgoicp.cpp:120), so best_ub, the counters and the accepted incumbent may then differ WITHIN
    // THE THRESHOLD between configurations (same epsilon-optimal answer, not the same bits).
    void overlap_report(std::vector<Task*>& tasks, const std::vector<const RotCube*>& cubes) {
        for (size_t i = 0; i + 1 < tasks.size(); i += 2) {
            if (cubes[i] != cubes[i + 1]) continue;
            auto &a = tasks[i]->seen, &b = tasks[i + 1]->seen;
            std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
            std::vector<std::array<uint32_t, 4>> both;
            std::set_intersection(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(both));
            ov_ub_ += a.size(); ov_lb_ += b.size(); ov_both_ += both.size();
        }
    }
    int run_task_list(std::vector<Task*>& tasks, const std::vector<const RotCube*>& cubes) {
        const int rc_ = run_task_list_impl(tasks, cubes);
        if (overlap_stats_) overlap_report(tasks, cubes);
        if (account_submissions_)  // ROUND: subcubes = what the tasks consumed (`count`, fgoicp.cpp:132), however they were served
            for (const Task* t : tasks) stats_.trans_cubes += t->count;
        return rc_;
    }
    int run_task_list_impl(std::vector<Task*>& tasks, const std::vector<const RotCube*>& cubes) {
        const bool par = tasks.size() >= 8 && pool_->size() > 1;
        const bool timing = timing_;
        if (tasks.size() >= 4 && ops_.async()) {
            Half h[2];
            // tasks come in (UB, LB) pairs per child: deal whole pairs so both halves hold both kinds
            for (size_t i = 0; i < tasks.size(); ++i) h[(i >> 1) & 1].members.push_back(i);
            for (;;) {
                bool any = false;
                for (int k = 0; k < 2; ++k) {
                    const auto ta = clock::now();
                    if (h[k].inflight) {
                        int rc = ops_.bounds_collect(k, h[k].lb.data(), h[k].ub.data());
                        if (rc) return rc;
                        h[k].inflight = false;
                    }
                    const auto tb = clock::now();
                    if (!h[k].live.empty()) consume_half(h[k], tasks, par);
                    h[k].live.clear();
                    const auto tc = clock::now();
                    rebalance(h[k], h[1 - k]);
                    const bool more = !h[k].members.empty() && prepare_half(h[k], tasks, cubes, par, ops_.twins());
                    if (tick_hook_) tick_hook_();  // (tasks that have just ended are marked done by now)
                    if (more) {
                        int rc = ops_.bounds_submit(k, (int)h[k].live.size(), h[k].R9.data(), h[k].spans.data(), h[k].fix.data(), h[k].offsets.data(), h[k].tn4.data(), h[k].twin.data(), use_cut_ ? h[k].cut.data() : nullptr);
                        if (rc) return rc;
                        h[k].inflight = true;
                    }
                    any = any || h[k].inflight;
                    if (timing) {
                        t_ops_ += std::chrono::duration<double>(tb - ta).count();
                        t_push_ += std::chrono::duration<double>(tc - tb).count();
                        t_pop_ += std::chrono::duration<double>(clock::now() - tc).count();
                    }
                }
                if (!any && h[0].members.empty() && h[1].members.empty()) return kDriverOk;
            }
        }
        Half h;
        for (size_t i = 0; i < tasks.size(); ++i) h.members.push_back(i);
        for (;;) {
            const auto ta = clock::now();
            const bool more = prepare_half(h, tasks, cubes, par, false);
            if (tick_hook_) tick_hook_();
            if (!more) return kDriverOk;
            const auto tb = clock::now();
            int rc = ops_.bounds_multi((int)h.live.size(), h.R9.data(), h.spans.data(), h.fix.data(), h.offsets.data(), h.tn4.data(), h.lb.data(), h.ub.data(), use_cut_ ? h.cut.data() : nullptr);
            if (rc) return rc;
            const auto tc = clock::now();
            consume_half(h, tasks, par);
            if (timing) {
                t_pop_ += std::chrono::duration<double>(tb - ta).count();
                t_ops_ += std::chrono::duration<double>(tc - tb).count();
                t_push_ += std::chrono::duration<double>(clock::now() - tc).count();
            }
        }
    }

    Ops& ops_;
    double t_pop_ = 0, t_ops_ = 0, t_push_ = 0;
    double t_prep_[3] = {0, 0, 0};  // FGOICP_TIMING: pops / pair matching / packing inside prepare_half
    const bool timing_ = dev_env("FGOICP_TIMING") != nullptr;  // host-side timing lines on stderr
    const bool serial_shard_ = [] { const char* e = dev_env("FGOICP_SERIAL_SHARD"); return !e || std::atoi(e) != 0; }();  // tuning knob / A-B: 0 = SERIAL on N ranks runs replicated
    const int coop_icp_ = [] { const char* e = dev_env("FGOICP_COOP_ICP"); return e ? (std::atoi(e) != 0 ? 1 : 0) : -1; }();  // tuning knob / A-B: 0 = every rank refines its own children alone, 1 = cooperative rounds, unset = by size
    const size_t coop_min_points_ = [] { const char* e = dev_env("FGOICP_COOP_MIN_POINTS"); return e ? (size_t)std::max(0L, std::atol(e)) : (size_t)131072; }();  // tuning knob
    const int late_icp_ = [] { const char* e = dev_env("FGOICP_LATE_ICP"); return e ? std::atoi(e) : 0; }();  // tuning knob (ROUND): 0 = off (default: measured slower, see above), 1 = with an exchange (world > 1), 2 = always
    bool use_cut_ = true;   // hand every task's cut_above() to the operator (set_use_cut: the A/B and the exact-rows mode of the tests)
    bool use_twins_ = [] { const char* e = dev_env("FGOICP_TWINS"); return !e || std::atoi(e) != 0; }();  // tuning knob
    const size_t round_batch_ = [] { const char* e = dev_env("FGOICP_ROUND_BATCH"); const int v = e ? std::atoi(e) : 48; return (size_t)(v >= 8 && v <= 64 ? v : 48); }();  // tuning knob (ROUND only; SERIAL keeps the reference's 32)
    const int serial_ahead_ = [] { const char* e = dev_env("FGOICP_SERIAL_AHEAD"); return e ? std::max(0, std::atoi(e)) : 480; }();  // tuning knob: look-ahead nodes of a SERIAL task in the tail of an evaluation (0 = off)
    const size_t serial_ahead_tasks_ = [] { const char* e = dev_env("FGOICP_SERIAL_AHEAD_TASKS"); const int v = e ? std::atoi(e) : 0; return (size_t)(v > 0 ? v : 0); }();  // ... while its half holds at most this many tasks (0 = by cloud size: 512 / 32)
    const size_t tick_rows_ = [] { const char* e = dev_env("FGOICP_TICK_ROWS"); const int v = e ? std::atoi(e) : 0; return (size_t)(v > 0 ? v : 0); }();  // tuning knob (ROUND): rows a tick should carry (0 = the stepwise tail rule)
    const size_t deal_block_ = [] { const char* e = dev_env("FGOICP_DEAL_BLOCK"); const int v = e ? std::atoi(e) : 0; return (size_t)(v >= 1 && v <= 64 ? v : 1); }();  // tuning knob (ROUND on N ranks): consecutive children dealt to one rank
    const bool tail_batch_fixed_ = dev_env("FGOICP_TAIL_BATCH") != nullptr;
    const size_t tail_batch_ = [] { const char* e = dev_env("FGOICP_TAIL_BATCH"); const int v = e ? std::atoi(e) : 128; return (size_t)(v >= 8 && v <= 512 ? v : 0); }();  // tuning knob (ROUND): batch of a half with few tasks left (0 = off)
    const size_t tail_tasks_ = [] { const char* e = dev_env("FGOICP_TAIL_TASKS"); const int v = e ? std::atoi(e) : 32; return (size_t)(v >= 0 ? v : 32); }();  // ... "few" = at most this many
    bool use_memo_ = [] { const char* e = dev_env("FGOICP_MEMO"); return !e || std::atoi(e) != 0; }();    // tuning knob: memo of the twin task's evaluations
    const bool overlap_stats_ = dev_env("FGOICP_OVERLAP_STATS") != nullptr;  // diagnostic: how many nodes both tasks of a rotation cube evaluate
    uint64_t ov_ub_ = 0, ov_lb_ = 0, ov_both_ = 0;
    bool account_submissions_ = true;   // false while SERIAL speculates: work is accounted per committed task instead
    std::function<void()> tick_hook_;   // called by the task loop's thread after every batch preparation (ROUND: the overlapped refinements watch the UB tasks end)
    const size_t ub_first_max_ = [] { const char* e = dev_env("FGOICP_UB_FIRST"); const int v = e ? std::atoi(e) : 0; return (size_t)(v > 0 ? v : 0); }();  // development knob: rounds of at most this many children run their UB tasks first (needs FGOICP_OVERLAP_ICP=1)
    const bool overlap_icp_ = [] { const char* e = dev_env("FGOICP_OVERLAP_ICP"); return e && std::atoi(e) != 0; }();  // development knob: 1 = a round's refinements next to its tasks (measured: no gain, see bnb_so3_round)
    std::unique_ptr<WorkerPool> pool_;
    float sse_threshold_;
    size_t ns_;
    int schedule_, round_width_;
    Exchange ex_;
    std::function<void(int, float, const Mat3f&, const Vec3f&)> log_;
    DriverStats stats_;
    mutable std::mutex mu_;
    float best_sse_ = kHostInf;
    Mat3f best_R_ = Mat3f::identity(), last_R_ = Mat3f::identity();
    Vec3f best_t_{0.f, 0.f, 0.f}, last_t_{0.f, 0.f, 0.f};
};

}  // namespace fgoicp
