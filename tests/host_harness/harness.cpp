// TEST-ONLY: instantiates the product's host driver template (fast-go-icp_amd/csrc/host/driver.hpp)
// with the CPU oracle's operators, so that the branch-and-bound host logic (both schedules, the
// multi-rank exchange) can be checked without a GPU.  This translation unit lives under tests/ and
// is never part of libfgoicp_amd.so; the product instantiates the template with HIP only.
#include <cstring>
#include <memory>
#include <vector>

#include "../../fast-go-icp_amd/csrc/device/morton.hpp"
#include "../../fast-go-icp_amd/csrc/host/driver.hpp"
#include "../../oracle/goicp_oracle.hpp"

namespace orc = goicp_oracle;
using namespace fgoicp;

namespace {
struct OracleOps {
    const orc::Registration* reg = nullptr;
    const orc::PointCloud* pct = nullptr;
    const orc::PointCloud* pcs = nullptr;
    int bounds_multi(int G, const float* R9, const float* rot_span, const int* fix_rot, const int* offsets, const float* tn4, float* lb,
                     float* ub) {
        for (int g = 0; g < G; ++g) {
            orc::RotNode rn(0, 0, 0, rot_span[g], 0, 0);
            std::memcpy(rn.q.R.c, R9 + 9 * g, sizeof(float) * 9);
            std::vector<orc::TransNode> tns;
            for (int i = offsets[g]; i < offsets[g + 1]; ++i) tns.emplace_back(tn4[4 * i], tn4[4 * i + 1], tn4[4 * i + 2], tn4[4 * i + 3], 0.f, 0.f);
            auto [l, u] = reg->compute_sse_error(rn, tns, fix_rot[g] != 0);
            for (size_t k = 0; k < tns.size(); ++k) { lb[offsets[g] + k] = l[k]; ub[offsets[g] + k] = u[k]; }
        }
        return 0;
    }
    // "asynchronous" slots for the pipelined driver path: evaluated at submit, handed out at collect
    std::vector<float> slot_lb[2], slot_ub[2];
    bool use_async = false, claim_twins = false;
    bool async() const { return use_async; }
    bool twins() const { return claim_twins; }  // the oracle evaluates every row; claiming twins only switches the driver's memo logic on (schedule 4, 5)
    int bounds_submit(int slot, int G, const float* R9, const float* rot_span, const int* fix_rot, const int* offsets, const float* tn4,
                      const int* /*twin: a device-side saving, the oracle evaluates every subcube*/) {
        slot_lb[slot].assign(offsets[G], 0.f);
        slot_ub[slot].assign(offsets[G], 0.f);
        return bounds_multi(G, R9, rot_span, fix_rot, offsets, tn4, slot_lb[slot].data(), slot_ub[slot].data());
    }
    int bounds_collect(int slot, float* lb, float* ub) {
        std::memcpy(lb, slot_lb[slot].data(), slot_lb[slot].size() * sizeof(float));
        std::memcpy(ub, slot_ub[slot].data(), slot_ub[slot].size() * sizeof(float));
        return 0;
    }
    // cooperative refinement: the CPU backend has nothing to split — every rank runs the whole (deterministic) loop: same result on every rank
    int icp_coop(int, int, int (*)(void*, size_t, void*), void*, const float* R0, const float* t0, size_t max_iter, float thr, float* sse, float* R9, float* t3,
                 int* iters) {
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
}  // namespace

extern "C" {

typedef int (*ar_fn)(float*, size_t, void*);
typedef int (*ag_fn)(const float*, float*, size_t, void*);

// build_lut = 0: the LUT is left empty and must be filled with harness_lut_set (bench.py's cpu_baseline: the O(nodes * nt) CPU build of
// a 5e7-node LUT would take hours; the device LUT is bit-identical, tests/test_gpu_ops.py::test_lut_nodes_bit_exact)
void* harness_create_ex(const float* tgt, size_t nt, const float* src, size_t ns, float lut_res, float mse_thr, int schedule, int round_width,
                        float trim_fraction, int build_lut, int use_grid) {
    auto* h = new Harness;
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
    return h;
}
void* harness_create_trim(const float* tgt, size_t nt, const float* src, size_t ns, float lut_res, float mse_thr, int schedule, int round_width,
                          float trim_fraction) {
    return harness_create_ex(tgt, nt, src, ns, lut_res, mse_thr, schedule, round_width, trim_fraction, 1, 0);
}
void harness_lut_dims(void* p, int* dims3) { for (int a = 0; a < 3; ++a) dims3[a] = static_cast<Harness*>(p)->reg->nnlut.dims[a]; }
int harness_lut_set(void* p, const float* data, size_t count) {
    auto& lut = static_cast<Harness*>(p)->reg->nnlut;
    if (count != lut.size()) return 1;
    lut.data.assign(data, data + count);
    return 0;
}
double harness_seconds(void* p, int which) {
    const DriverStats& s = static_cast<Harness*>(p)->drv->stats();
    return which == 0 ? s.seconds_total : which == 1 ? s.seconds_bnb : s.seconds_icp;
}
void* harness_create(const float* tgt, size_t nt, const float* src, size_t ns, float lut_res, float mse_thr, int schedule, int round_width) {
    return harness_create_trim(tgt, nt, src, ns, lut_res, mse_thr, schedule, round_width, 0.0f);
}
void harness_destroy(void* p) { delete static_cast<Harness*>(p); }
void harness_set_exchange(void* p, int rank, int world, ar_fn ar, ag_fn ag) {
    Exchange e; e.rank = rank; e.world = world; e.allreduce_min = ar; e.allgather = ag; e.user = nullptr;
    static_cast<Harness*>(p)->drv->set_exchange(e);
}
// coop != 0: the exchange also offers a device all-gather, which switches the driver to COOPERATIVE rounds (bounds exchanged first,
// triggers in the single-GPU child order on every rank).  The CPU backend never calls the hook (OracleOps::icp_coop runs the whole loop).
static int harness_no_device_gather(void*, size_t, void*) { return 1; }
void harness_set_exchange_coop(void* p, int rank, int world, ar_fn ar, ag_fn ag, int coop) {
    Exchange e; e.rank = rank; e.world = world; e.allreduce_min = ar; e.allgather = ag; e.user = nullptr;
    e.allgather_device = coop ? harness_no_device_gather : nullptr;
    static_cast<Harness*>(p)->drv->set_exchange(e);
}
void harness_preproc(void* p, float* offs6, float* scale, float* bounds6) {
    auto* h = static_cast<Harness*>(p);
    offs6[0] = h->off_s.x; offs6[1] = h->off_s.y; offs6[2] = h->off_s.z; offs6[3] = h->off_t.x; offs6[4] = h->off_t.y; offs6[5] = h->off_t.z;
    *scale = h->scale;
    std::memcpy(bounds6, h->bounds6, sizeof(h->bounds6));
}
// returns driver status; t3 = restored translation, ts3 = translation in the scaled frame
int harness_run(void* p, float* R9, float* t3, float* ts3, float* best_sse, unsigned long long* stats7) {
    auto* h = static_cast<Harness*>(p);
    int rc = h->drv->run();
    if (rc) return rc;
    Mat3f R; Vec3f t;
    h->drv->best_transform(R, t);
    const Vec3f tr = t / h->scale + R * h->off_s - h->off_t;  // fgoicp.hpp:87-90
    std::memcpy(R9, R.m, sizeof(R.m));
    t3[0] = tr.x; t3[1] = tr.y; t3[2] = tr.z;
    ts3[0] = t.x; ts3[1] = t.y; ts3[2] = t.z;
    *best_sse = h->drv->best_sse();
    const DriverStats& s = h->drv->stats();
    stats7[0] = s.trans_cubes; stats7[1] = s.bounds_calls; stats7[2] = s.rot_cubes; stats7[3] = s.icp_runs; stats7[4] = s.icp_iters;
    stats7[5] = s.inner_bnb; stats7[6] = s.rounds;
    return 0;
}
// product-side pieces exercised directly
void harness_rotation(float x, float y, float z, float* R9, float* r, int* in_so3) {
    RotationQ q(x, y, z);
    std::memcpy(R9, q.R.m, sizeof(q.R.m));
    *r = q.r;
    *in_so3 = q.in_SO3();
}
int harness_overlaps(float x, float y, float z, float span) { return RotCube(x, y, z, span, 0, 0).overlaps_SO3(); }
void harness_closest_orthogonal(const float* ABt9, float* R9) {
    Mat3f r = closest_orthogonal_approximation(Mat3f::from(ABt9));
    std::memcpy(R9, r.m, sizeof(r.m));
}
void harness_svd3(const double* A9, double* U9, double* S3, double* V9) {
    double A[3][3], U[3][3], V[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = A9[3 * i + j];
    svd3_jacobi(A, U, S3, V);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { U9[3 * i + j] = U[i][j]; V9[3 * i + j] = V[i][j]; }
}
// the point orders of csrc/device/morton.hpp (host code): mode 1 = curve, 2 = k-d cells of `leaf` points (+ in-leaf order), 3 = density split
void harness_point_order(const float* xyz, size_t n, size_t leaf, int mode, int fine, uint32_t* perm_out) {
    const std::vector<uint32_t> p = mode == 3 ? fgoicp::mixed_order(xyz, n, 3, leaf, fine != 0) : mode == 2 ? fgoicp::kd_order(xyz, n, 3, leaf, fine != 0) : fgoicp::morton_order(xyz, n, 3);
    std::memcpy(perm_out, p.data(), sizeof(uint32_t) * n);
}

}  // extern "C"
