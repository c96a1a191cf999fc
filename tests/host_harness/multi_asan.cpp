// TEST-ONLY: the multi-rank core of the product (fast-go-icp_amd/csrc/host/multi_link.hpp: rendezvous, exchange callbacks with
// recording / replay / fault injection, one host thread per rank) and the host driver (driver.hpp: ROUND and SERIAL, sharded) over
// the CPU oracle's operators, built with -fsanitize=address,undefined and run as a program: create -> record -> run -> replay every
// rank -> destroy, for the configurations fgoicp_multi runs on the GPU (a heap abort was recorded in the teardown of
// an 8-rank SERIAL replay on the GPU box; this is the same sequence with every host-side allocation watched).
// "Device" memory is host memory here; OracleOps::icp_coop exercises the device all-gather hook with a pattern buffer.
//
//   multi_asan [world] [ns] [nt] [mse_threshold] [angle_deg]      exit status 0 = every scenario ran, every rank agreed, every replay returned the recorded run
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

#include "../../fast-go-icp_amd/csrc/host/multi_link.hpp"
#include "oracle_ops.hpp"

namespace fgoicp {
static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
}  // namespace fgoicp

namespace {
using host_harness::Harness;

int host_alloc(int, size_t bytes, void** out) { *out = std::malloc(bytes ? bytes : 1); return *out ? 0 : 1; }
void host_release(int, void* p) { std::free(p); }
int host_copy(void* dst, int, const void* src, int, size_t bytes) { std::memcpy(dst, src, bytes); return 0; }
int host_sync(int) { return 0; }
const fgoicp::DeviceMemApi kHostMem{host_alloc, host_release, host_copy, host_sync};

struct CpuBackend {
    using Solver = Harness;
    static int run(Solver* h, float* R9, float* t3) {
        const int rc = h->drv->run();
        if (rc) { fgoicp::set_error("driver status " + std::to_string(rc) + (fgoicp::g_err.empty() ? "" : ": " + fgoicp::g_err)); return rc; }
        Mat3f R; Vec3f t;
        h->drv->best_transform(R, t);
        std::memcpy(R9, R.m, sizeof(R.m));
        t3[0] = t.x; t3[1] = t.y; t3[2] = t.z;
        return 0;
    }
    static int set_exchange(Solver* h, const fgoicp_exchange* ex) {
        Exchange e;
        if (ex) { e.rank = ex->rank; e.world = ex->world_size; e.allreduce_min = ex->allreduce_min; e.allgather = ex->allgather; e.user = ex->user; e.allgather_device = ex->allgather_device; }
        h->drv->set_exchange(e);
        return 0;
    }
    static void destroy(Solver* h) { delete h; }
    static int icp_coop(Solver* h, int rank, int world, int (*gather)(void*, size_t, void*), void* user, const float* R0, const float* t0, size_t max_iter, float thr, float* sse,
                        float* R9, float* t3, int* iters) {
        return h->ops.icp_coop(rank, world, gather, user, R0, t0, max_iter, thr, sse, R9, t3, iters);
    }
    static const char* last_error() { return fgoicp::g_err.c_str(); }
    static const fgoicp::DeviceMemApi* mem() { return &kHostMem; }
};
using Multi = fgoicp::MultiCore<CpuBackend>;

// a closed bumpy surface and a rotated, shifted, noisy part of it (the shape family of fgoicp_amd.synth, small)
void make_clouds(size_t nt, size_t ns, float angle_deg, std::vector<float>& tgt, std::vector<float>& src) {
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::normal_distribution<float> N(0.f, 1.f);
    auto surface = [&](float* p) {
        float v[3] = {N(rng), N(rng), N(rng)};
        const float n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + 1e-9f;
        for (float& x : v) x /= n;
        const float r = 1.0f + 0.25f * std::sin(3.0f * v[0]) * std::cos(2.0f * v[1]) + 0.15f * std::sin(5.0f * v[2]);
        p[0] = 0.9f * r * v[0]; p[1] = 0.7f * r * v[1]; p[2] = 0.5f * r * v[2];
    };
    tgt.resize(3 * nt);
    for (size_t i = 0; i < nt; ++i) surface(&tgt[3 * i]);
    // rotation by 100 degrees about (1, 2, 3) / |.|, translation (0.1, -0.05, 0.08)
    const float ax[3] = {0.2672612f, 0.5345225f, 0.8017837f}, ang = angle_deg * 3.14159265f / 180.0f, c = std::cos(ang), s = std::sin(ang);
    float R[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i][j] = (i == j ? c : 0.f) + (1 - c) * ax[i] * ax[j];
    R[0][1] -= s * ax[2]; R[0][2] += s * ax[1]; R[1][0] += s * ax[2]; R[1][2] -= s * ax[0]; R[2][0] -= s * ax[1]; R[2][1] += s * ax[0];
    src.resize(3 * ns);
    for (size_t i = 0; i < ns; ++i) {
        const float* q = &tgt[3 * (size_t)(U(rng) * (float)(nt - 1))];
        // source = R^T (q - t): the registration has to find (R, t)
        const float d[3] = {q[0] - 0.1f, q[1] + 0.05f, q[2] - 0.08f};
        for (int k = 0; k < 3; ++k) src[3 * i + k] = R[0][k] * d[0] + R[1][k] * d[1] + R[2][k] * d[2] + 0.01f * N(rng);
    }
}

struct Outcome { float R[9], t[3], sse; unsigned long long subcubes, icp_runs; };

int fail(const char* what, const std::string& detail = "") {
    std::fprintf(stderr, "multi_asan: FAILED: %s %s (%s)\n", what, detail.c_str(), fgoicp::g_err.c_str());
    return 1;
}

// schedule: 2 = ROUND, 3 = SERIAL (two-slot task loop), 4 / 5 = the same with the twin-task memo (what the HIP backend claims)
int scenario(const char* name, int world, int schedule, int round_width, bool coop, bool fault, const std::vector<float>& tgt, const std::vector<float>& src, float res, float mse,
             const Outcome* expect, Outcome* out) {
    const size_t nt = tgt.size() / 3, ns = src.size() / 3;
    auto m = std::make_unique<Multi>();
    std::vector<int> devs((size_t)world, 0);
    m->init(devs.data(), world);
    for (int r = 0; r < world; ++r) {
        Harness* h = host_harness::make_harness(tgt.data(), nt, src.data(), ns, res, mse, schedule, round_width, 0.0f, 1, 0);
        h->ops.exercise_gather = coop;
        m->solvers.push_back(h);
    }
    for (int r = 0; r < world; ++r)
        if (m->connect(r, nullptr)) return fail(name, "connect");
    if (!coop)  // the private flow: the exchange offers no device all-gather
        for (int r = 0; r < world; ++r) {
            fgoicp_exchange ex{sizeof(fgoicp_exchange), r, world, fgoicp::link_allreduce_min, fgoicp::link_allgather, m->links[(size_t)r].get(), nullptr};
            CpuBackend::set_exchange(m->solvers[(size_t)r], world > 1 ? &ex : nullptr);
        }
    float R[9], t[3];
    if (fault && world > 1) {  // a failing exchange ends the run for every rank; the same object then runs again
        m->test_fault(world - 1, 1);
        if (m->run(R, t) == 0) return fail(name, "the injected fault went unnoticed");
    }
    m->set_record(1);
    if (m->run(R, t)) return fail(name, "run");
    Outcome o{};
    std::memcpy(o.R, R, sizeof(R)); std::memcpy(o.t, t, sizeof(t));
    o.sse = m->solvers[0]->drv->best_sse();
    o.subcubes = m->solvers[0]->drv->stats().trans_cubes;
    o.icp_runs = m->solvers[0]->drv->stats().icp_runs;
    uint64_t hx = 0, dgath = 0;
    m->recorded(0, &hx, &dgath);
    std::fprintf(stderr, "multi_asan: %-34s world %d: sse %.6g, subcubes (rank 0) %llu, icp runs %llu, host exchanges %llu, device gathers %llu\n", name, world, (double)o.sse,
                 o.subcubes, o.icp_runs, (unsigned long long)hx, (unsigned long long)dgath);
    if (world > 1 && hx == 0) return fail(name, "nothing was recorded");
    if (coop && world > 1 && dgath == 0) return fail(name, "no device all-gather was recorded");
    for (int r = 0; r < world; ++r) {  // every rank alone against the recording: the same incumbent, the same counters
        const auto before = m->solvers[(size_t)r]->drv->stats();
        double sec = 0;
        if (m->replay_rank(r, &sec)) return fail(name, "replay of rank " + std::to_string(r));
        const auto& after = m->solvers[(size_t)r]->drv->stats();
        Mat3f Rr; Vec3f tr;
        m->solvers[(size_t)r]->drv->best_transform(Rr, tr);
        if (std::memcmp(Rr.m, o.R, sizeof(o.R)) != 0 || m->solvers[(size_t)r]->drv->best_sse() != o.sse) return fail(name, "replayed rank " + std::to_string(r) + " ended elsewhere");
        if (after.trans_cubes != before.trans_cubes || after.icp_runs != before.icp_runs) return fail(name, "replayed rank " + std::to_string(r) + " did other work");
    }
    if (m->replay_rank(0, nullptr)) return fail(name, "second replay of rank 0");  // a recording can be replayed more than once
    if (expect) {
        if (schedule == 3 || schedule == 5) {  // SERIAL: the one-rank run's bits and counters on any number of ranks
            if (std::memcmp(expect->R, o.R, sizeof(o.R)) != 0 || std::memcmp(expect->t, o.t, sizeof(o.t)) != 0 || expect->sse != o.sse || expect->icp_runs != o.icp_runs)
                return fail(name, "differs from the one-rank run");
        } else if (!(std::fabs(expect->sse - o.sse) <= 2e-3f * expect->sse)) {
            return fail(name, "another optimum than the one-rank run");
        }
    }
    if (out) *out = o;
    m.reset();  // teardown: solvers, recordings, links — the sequence under test
    return 0;
}
}  // namespace

int main(int argc, char** argv) {
    const int world = argc > 1 ? std::atoi(argv[1]) : 8;
    const size_t ns = argc > 2 ? (size_t)std::atol(argv[2]) : 160, nt = argc > 3 ? (size_t)std::atol(argv[3]) : 260;
    // the look-ahead of SERIAL tasks and ROUND's cooperative flow are sized for big clouds: force them on for this small one
    setenv("FGOICP_SERIAL_AHEAD_TASKS", "512", 1);
    setenv("FGOICP_COOP_MIN_POINTS", "1", 1);
    setenv("FGOICP_HOST_THREADS", "2", 1);
    setenv("FGOICP_HOST_SPIN", "0", 1);
    std::vector<float> tgt, src;
    const float mse = argc > 4 ? (float)std::atof(argv[4]) : 1e-4f, angle = argc > 5 ? (float)std::atof(argv[5]) : 100.0f;
    make_clouds(nt, ns, angle, tgt, src);
    const float res = 0.08f;
    Outcome serial1{}, round1{};
    int rc = 0;
    rc = rc || scenario("SERIAL, one rank", 1, 5, 1, false, false, tgt, src, res, mse, nullptr, &serial1);
    rc = rc || scenario("ROUND, one rank", 1, 4, 0, false, false, tgt, src, res, mse, nullptr, &round1);
    rc = rc || scenario("SERIAL sharded, cooperative", world, 5, 1, true, false, tgt, src, res, mse, &serial1, nullptr);
    rc = rc || scenario("SERIAL sharded, private, fault", world, 5, 1, false, true, tgt, src, res, mse, &serial1, nullptr);
    rc = rc || scenario("SERIAL sharded, no memo", 3, 3, 1, true, false, tgt, src, res, mse, &serial1, nullptr);
    rc = rc || scenario("ROUND sharded, cooperative", world, 4, 0, true, true, tgt, src, res, mse, &round1, nullptr);
    rc = rc || scenario("ROUND sharded, private", world, 4, 0, false, false, tgt, src, res, mse, &round1, nullptr);
    rc = rc || scenario("ROUND sharded, fixed width, 2 ranks", 2, 2, 4, true, false, tgt, src, res, mse, &round1, nullptr);
    if (!rc) std::fprintf(stderr, "multi_asan: all scenarios passed\n");
    return rc;
}
