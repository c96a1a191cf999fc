// TEST-ONLY: instantiates the product's host driver template (fast-go-icp_amd/csrc/host/driver.hpp)
// with the CPU oracle's operators, so that the branch-and-bound host logic (both schedules, the
// multi-rank exchange) can be checked without a GPU.  This translation unit lives under tests/ and
// is never part of libfgoicp_amd.so; the product instantiates the template with HIP only.
#include "oracle_ops.hpp"
#include "../../fast-go-icp_amd/csrc/device/slab.hpp"

using namespace host_harness;

extern "C" {

typedef int (*ar_fn)(float*, size_t, void*);
typedef int (*ag_fn)(const float*, float*, size_t, void*);

// build_lut = 0: the LUT is left empty and must be filled with harness_lut_set (bench.py's cpu_baseline: the O(nodes * nt) CPU build of
// a 5e7-node LUT would take hours; the device LUT is bit-identical, tests/test_gpu_ops.py::test_lut_nodes_bit_exact)
void* harness_create_ex(const float* tgt, size_t nt, const float* src, size_t ns, float lut_res, float mse_thr, int schedule, int round_width,
                        float trim_fraction, int build_lut, int use_grid) {
    return make_harness(tgt, nt, src, ns, lut_res, mse_thr, schedule, round_width, trim_fraction, build_lut, use_grid);
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
// early exit (fgoicp_bounds_submit_cut): does the driver hand its tasks' thresholds to the operator, and does the oracle operator then
// answer {T, T} for a row at or above its threshold as the device does (it evaluates every row in full either way)
void harness_set_cut(void* p, int driver_passes_thresholds, int oracle_applies_them) {
    auto* h = static_cast<Harness*>(p);
    h->drv->set_use_cut(driver_passes_thresholds != 0);
    h->ops.apply_cut = oracle_applies_them != 0;
}
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
// the leaf-slab test of the exact scans (csrc/device/slab.hpp), n queries against one slab
void harness_slab_d2(const float* n3, float a, float b, const float* q_xyz, size_t n, float* out) {
    for (size_t i = 0; i < n; ++i) out[i] = fgoicp::slab_d2(n3[0], n3[1], n3[2], a, b, q_xyz[3 * i], q_xyz[3 * i + 1], q_xyz[3 * i + 2]);
}
// the point orders of csrc/device/morton.hpp (host code): mode 1 = curve, 2 = k-d cells of `leaf` points (+ in-leaf order), 3 = density split
void harness_point_order(const float* xyz, size_t n, size_t leaf, int mode, int fine, uint32_t* perm_out) {
    const std::vector<uint32_t> p = mode == 3 ? fgoicp::mixed_order(xyz, n, 3, leaf, fine != 0) : mode == 2 ? fgoicp::kd_order(xyz, n, 3, leaf, fine != 0) : fgoicp::morton_order(xyz, n, 3);
    std::memcpy(perm_out, p.data(), sizeof(uint32_t) * n);
}

}  // extern "C"
