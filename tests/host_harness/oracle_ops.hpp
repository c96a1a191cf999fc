// TEST-ONLY: the CPU oracle's operators behind the product's driver template (fast-go-icp_amd/csrc/host/driver.hpp), shared by
// harness.cpp (the ctypes harness of the CPU tests and bench.py's cpu_baseline) and multi_asan.cpp (the AddressSanitizer run of the
// multi-rank core).  Lives under tests/; never part of libfgoicp_amd.so.
#pragma once
#include <cstring>
#include <memory>
#include <vector>

#include "../../fast-go-icp_amd/csrc/device/morton.hpp"
#include "../../fast-go-icp_amd/csrc/host/driver.hpp"
#include "../../oracle/goicp_oracle.hpp"

namespace orc = goicp_oracle;
using namespace fgoicp;

namespace host_harness {
struct OracleOps {
    const orc::Registration* reg = nullptr;
    const orc::PointCloud* pct = nullptr;
    const orc::PointCloud* pcs = nullptr;
    // cut_above (fgoicp_bounds_submit_cut): the oracle always evaluates every subcube in full; with apply_cut it then reports a row whose
    // lower bound has reached its group's threshold T as {T, T}, like the device — the driver must not be able to tell (tests/test_host_logic.py)
    bool apply_cut = false;
    int bounds_multi(int G, const float* R9, const float* rot_span, const int* fix_rot, const int* offsets, const float* tn4, float* lb,
                     float* ub, const float* cut_above = nullptr) {
        for (int g = 0; g < G; ++g) {
            orc::RotNode rn(0, 0, 0, rot_span[g], 0, 0);
            std::memcpy(rn.q.R.c, R9 + 9 * g, sizeof(float) * 9);
            std::vector<orc::TransNode> tns;
            for (int i = offsets[g]; i < offsets[g + 1]; ++i) tns.emplace_back(tn4[4 * i], tn4[4 * i + 1], tn4[4 * i + 2], tn4[4 * i + 3], 0.f, 0.f);
            auto [l, u] = reg->compute_sse_error(rn, tns, fix_rot[g] != 0);
            for (size_t k = 0; k < tns.size(); ++k) { lb[offsets[g] + k] = l[k]; ub[offsets[g] + k] = u[k]; }
            if (apply_cut && cut_above)
                for (size_t k = 0; k < tns.size(); ++k)
                    if (lb[offsets[g] + k] >= cut_above[g]) lb[offsets[g] + k] = ub[offsets[g] + k] = cut_above[g];
        }
        return 0;
    }
    // "asynchronous" slots for the pipelined driver path: evaluated at submit, handed out at collect
    std::vector<float> slot_lb[2], slot_ub[2];
    bool use_async = false, claim_twins = false;
    bool async() const { return use_async; }
    bool twins() const { return claim_twins; }  // the oracle evaluates every row; claiming twins only switches the driver's memo logic on (schedule 4, 5)
    int bounds_submit(int slot, int G, const float* R9, const float* rot_span, const int* fix_rot, const int* offsets, const float* tn4,
                      const int* /*twin: a device-side saving, the oracle evaluates every subcube*/, const float* cut_above = nullptr) {
        slot_lb[slot].assign(offsets[G], 0.f);
        slot_ub[slot].assign(offsets[G], 0.f);
        return bounds_multi(G, R9, rot_span, fix_rot, offsets, tn4, slot_lb[slot].data(), slot_ub[slot].data(), cut_above);
    }
    int bounds_collect(int slot, float* lb, float* ub) {
        std::memcpy(lb, slot_lb[slot].data(), slot_lb[slot].size() * sizeof(float));
        std::memcpy(ub, slot_ub[slot].data(), slot_ub[slot].size() * sizeof(float));
        return 0;
    }
    // cooperative refinement: the CPU backend has nothing to split — every rank runs the whole (deterministic) loop: same result on every rank.
    // exercise_gather (multi_asan.cpp): before that, one in-place all-gather of a "device" buffer (host memory there) through the hook,
    // each rank's chunk filled with a pattern of its own and every chunk checked afterwards — the record / replay path of the gathers.
    bool exercise_gather = false;
    int icp_coop(int rank, int world, int (*gather)(void*, size_t, void*), void* user, const float* R0, const float* t0, size_t max_iter, float thr, float* sse, float* R9, float* t3,
                 int* iters) {
        if (exercise_gather && gather && world > 1) {
            const size_t per = 96;
            std::vector<unsigned char> buf(per * (size_t)world, 0xEE);
            for (size_t i = 0; i < per; ++i) buf[per * (size_t)rank + i] = (unsigned char)(17 * rank + (int)i);
            if (gather(buf.data(), per, user)) return 6;
            for (int r = 0; r < world; ++r)
                for (size_t i = 0; i < per; ++i)
                    if (buf[per * (size_t)r + i] != (unsigned char)(17 * r + (int)i)) { std::fprintf(stderr, "device all-gather returned a wrong chunk\n"); return 6; }
        }
        return icp(R0, t0, max_iter, thr, sse, R9, t3, iters);
    }
    int icp_background(const float* R0, const float* t0, size_t max_iter, float thr, float* sse, float* R9, float* t3, int* iters) {
        return icp(R0, t0, max_iter, thr, sse, R9, t3, iters);  // the oracle's ICP object is local to the call: safe next to the bounds operator
    }
    int icp(const float* R0, const float* t0, size_t max_iter, float thr, float* sse, float* R9, float* t3, int* iters) {
        orc::Mat3 R;
        std::memcpy(R.c, R0, sizeof(float) * 9);
        orc::IterativeClosestPoint3D icp3d(*reg, *pct, *pcs, max_iter, thr, R, orc::Vec3{t0[0], t0[1], t0[2]});
        auto [s, Ro, to] = icp3d.run();
        *sse = s;
        std::memcpy(R9, Ro.c, sizeof(float) * 9);
        t3[0] = to.x; t3[1] = to.y; t3[2] = to.z;
        *iters = (int)icp3d.iterations();
        return 0;
    }
};

struct Harness {
    std::vector<Vec3f> pcs, pct;
    Vec3f off_s, off_t;
    float scale;
    float bounds6[6];
    orc::PointCloud opct, opcs;
    std::unique_ptr<orc::Registration> reg;
    OracleOps ops;
    std::unique_ptr<GoIcpDriver<OracleOps>> drv;
};
// product pre-processing + the oracle's Registration + the driver template over OracleOps (what harness_create_ex hands to ctypes)
inline Harness* make_harness(const float* tgt, size_t nt, const float* src, size_t ns, float lut_res, float mse_thr, int schedule, int round_width, float trim_fraction,
                             int build_lut, int use_grid) {
    auto h = std::make_unique<Harness>();
    h->pcs.resize(ns); h->pct.resize(nt);
    std::memcpy(h->pcs.data(), src, sizeof(Vec3f) * ns);
    std::memcpy(h->pct.data(), tgt, sizeof(Vec3f) * nt);
    h->off_s = center_point_cloud(h->pcs);   // product pre-processing
    h->off_t = center_point_cloud(h->pct);
    h->scale = scale_point_clouds(h->pct, h->pcs);
    point_cloud_ranges(h->pct, h->bounds6);
    h->opct.resize(nt); h->opcs.resize(ns);
    std::memcpy(h->opct.data(), h->pct.data(), sizeof(Vec3f) * nt);
    std::memcpy(h->opcs.data(), h->pcs.data(), sizeof(Vec3f) * ns);
    orc::Bounds b{std::make_pair(h->bounds6[0], h->bounds6[1]), std::make_pair(h->bounds6[2], h->bounds6[3]), std::make_pair(h->bounds6[4], h->bounds6[5])};
    h->reg.reset(new orc::Registration(h->opct, h->opcs, b, lut_res, build_lut != 0));
    if (use_grid) h->reg->use_grid(true);
    h->ops.reg = h->reg.get(); h->ops.pct = &h->opct; h->ops.pcs = &h->opcs;
    h->ops.use_async = schedule >= 2;  // schedule 2 = ROUND, 3 = SERIAL, both with the two-slot pipelined task loop; 4, 5 = the same with the twin-task memo
    h->ops.claim_twins = schedule >= 4;
    schedule = (schedule == 2 || schedule == 4) ? 1 : (schedule == 3 || schedule == 5) ? 0 : schedule;
    size_t n_thr = ns;  // as solver.cpp: the threshold runs over the inliers when trimming
    if (trim_fraction > 0.0f) {
        size_t k = (size_t)((double)ns * (1.0 - (double)trim_fraction));
        if (k < 1) k = 1;
        if (k < ns) { h->reg->inliers = k; n_thr = k; }
    }
    h->drv.reset(new GoIcpDriver<OracleOps>(h->ops, n_thr, mse_thr, schedule, round_width));
    return h.release();
}
}  // namespace host_harness

