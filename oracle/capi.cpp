// extern "C" surface of the CPU oracle for ctypes — TEST INFRASTRUCTURE ONLY (see goicp_oracle.hpp).
// PARITY UNPINNED (no reference outputs exist).
// All matrices are 9 floats in glm::mat3 memory order (column-major, m[col*3+row]).
#include <cstring>
#include <memory>

#include "goicp_oracle.hpp"
#include <queue>
#include <string>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace goicp_oracle;

namespace goicp_oracle {
void svd3_rowmajor(const double* A9, double* U9, double* S3, double* V9);
}

namespace {
Mat3 to_mat(const float* m) { Mat3 r; std::memcpy(r.c, m, sizeof(r.c)); return r; }
void from_mat(const Mat3& m, float* o) { std::memcpy(o, m.c, sizeof(m.c)); }
PointCloud to_cloud(const float* xyz, size_t n) {
    PointCloud pc(n);
    for (size_t i = 0; i < n; ++i) pc[i] = Vec3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    return pc;
}
Bounds to_bounds(const float* b6) {
    return Bounds{std::make_pair(b6[0], b6[1]), std::make_pair(b6[2], b6[3]), std::make_pair(b6[4], b6[5])};
}

struct RegHandle {
    PointCloud pct, pcs;
    std::unique_ptr<Registration> reg;
};
struct GoicpHandle {
    std::unique_ptr<FastGoICP> g;
};
}  // namespace

extern "C" {

int orc_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

// ---- conventions (goicp_oracle.hpp: Conventions) — process-global switches for tools/convention_flips.py ---------
// returns 0, or -1 for an unknown name; value < -100 only reads.  *old receives the previous value when not null.
int orc_convention(const char* name, int value, int* old) {
    Conventions& c = conventions();
    int* slot = nullptr;
    const std::string n(name);
    if (n == "fma_matvec") slot = &c.fma_matvec;
    else if (n == "fma_dist") slot = &c.fma_dist;
    else if (n == "fma_rot_sub") slot = &c.fma_rot_sub;
    else if (n == "fma_trans_sub") slot = &c.fma_trans_sub;
    else if (n == "tex_weight") slot = &c.tex_weight;
    else if (n == "tex_blend") slot = &c.tex_blend;
    else if (n == "sum_mode") slot = &c.sum_mode;
    else if (n == "sin_ulps") slot = &c.sin_ulps;
    else if (n == "svd_r2_two_sided") slot = &c.svd_r2_two_sided;
    if (!slot) return -1;
    if (old) *old = *slot;
    if (value >= -100) *slot = value;
    return 0;
}
void orc_conventions_reset() { conventions() = Conventions(); }

// ---- types -------------------------------------------------------------------------------
void orc_rotation(float x, float y, float z, float* R9, float* r, int* in_so3) {
    Rotation q(x, y, z);
    from_mat(q.R, R9);
    *r = q.r;
    *in_so3 = q.in_SO3() ? 1 : 0;
}
int orc_rotnode_overlaps(float x, float y, float z, float span) { return RotNode(x, y, z, span, 0, 0).overlaps_SO3() ? 1 : 0; }
// pops the given nodes in std::priority_queue order; out_order[i] = index of the i-th popped node
void orc_transnode_pop_order(const float* lb, const float* span, int n, int* out_order) {
    struct Tagged { TransNode n; int id; bool operator<(const Tagged& o) const { return n < o.n; } };
    std::priority_queue<Tagged> pq;
    for (int i = 0; i < n; ++i) pq.push(Tagged{TransNode(0, 0, 0, span[i], lb[i], 0), i});
    for (int i = 0; i < n; ++i) { out_order[i] = pq.top().id; pq.pop(); }
}

// ---- registration operator -------------------------------------------------------------
void* orc_reg_create(const float* tgt, size_t nt, const float* src, size_t ns, const float* bounds6, float res, int build_lut,
                     int quantize) {
    auto* h = new RegHandle;
    h->pct = to_cloud(tgt, nt);
    h->pcs = to_cloud(src, ns);
    h->reg.reset(new Registration(h->pct, h->pcs, to_bounds(bounds6), res, build_lut != 0));
    h->reg->nnlut.quantize_weights = quantize != 0;
    return h;
}
void orc_reg_destroy(void* p) { delete static_cast<RegHandle*>(p); }
void orc_reg_set_inliers(void* p, size_t k) { static_cast<RegHandle*>(p)->reg->inliers = k; }
void orc_reg_use_grid(void* p, int on) { static_cast<RegHandle*>(p)->reg->use_grid(on != 0); }  // CPU baseline only
void orc_reg_lut_dims(void* p, int* dims3) {
    auto* h = static_cast<RegHandle*>(p);
    for (int i = 0; i < 3; ++i) dims3[i] = h->reg->nnlut.dims[i];
}
void orc_reg_lut_get(void* p, float* out) {
    auto* h = static_cast<RegHandle*>(p);
    std::memcpy(out, h->reg->nnlut.data.data(), h->reg->nnlut.data.size() * sizeof(float));
}
void orc_reg_lut_set(void* p, const float* in) {
    auto* h = static_cast<RegHandle*>(p);
    h->reg->nnlut.data.assign(in, in + h->reg->nnlut.size());
}
void orc_reg_lut_search(void* p, const float* q, size_t n, float* out) {
    auto* h = static_cast<RegHandle*>(p);
    for (size_t i = 0; i < n; ++i) out[i] = h->reg->nnlut.search(Vec3{q[3 * i], q[3 * i + 1], q[3 * i + 2]});
}
void orc_reg_bounds(void* p, const float* R9, float rot_span, const float* tn4, int B, int fix_rot, float* lb, float* ub) {
    auto* h = static_cast<RegHandle*>(p);
    RotNode rn(0, 0, 0, rot_span, 0, 0);
    rn.q.R = to_mat(R9);
    std::vector<TransNode> tns;
    for (int b = 0; b < B; ++b) tns.emplace_back(tn4[4 * b], tn4[4 * b + 1], tn4[4 * b + 2], tn4[4 * b + 3], 0.f, 0.f);
    auto [l, u] = h->reg->compute_sse_error(rn, tns, fix_rot != 0);
    for (int b = 0; b < B; ++b) { lb[b] = l[b]; ub[b] = u[b]; }
}
float orc_reg_sse(void* p, const float* R9, const float* t3) {
    auto* h = static_cast<RegHandle*>(p);
    return h->reg->compute_sse_error(to_mat(R9), Vec3{t3[0], t3[1], t3[2]});
}
void orc_reg_icp(void* p, const float* R9, const float* t3, size_t max_iter, float thr, float* sse, float* Rout, float* tout, int* iters) {
    auto* h = static_cast<RegHandle*>(p);
    IterativeClosestPoint3D icp(*h->reg, h->pct, h->pcs, max_iter, thr, to_mat(R9), Vec3{t3[0], t3[1], t3[2]});
    auto [s, R, t] = icp.run();
    *sse = s;
    from_mat(R, Rout);
    tout[0] = t.x; tout[1] = t.y; tout[2] = t.z;
    *iters = (int)icp.iterations();
}
// one Procrustes step on an explicit working cloud (ns x 3)
void orc_reg_procrustes(void* p, const float* working, float* R9, float* t3, float* centroids6, float* ABt9, int* corr_idx) {
    auto* h = static_cast<RegHandle*>(p);
    PointCloud w = to_cloud(working, h->pcs.size());
    IterativeClosestPoint3D icp(*h->reg, h->pct, w, 1, 0.f, mat3_identity(), Vec3{0, 0, 0});
    ProcrustesDebug dbg;
    auto [R, t] = icp.procrustes(&dbg);
    from_mat(R, R9);
    t3[0] = t.x; t3[1] = t.y; t3[2] = t.z;
    if (centroids6) {
        centroids6[0] = dbg.src_centroid.x; centroids6[1] = dbg.src_centroid.y; centroids6[2] = dbg.src_centroid.z;
        centroids6[3] = dbg.cor_centroid.x; centroids6[4] = dbg.cor_centroid.y; centroids6[5] = dbg.cor_centroid.z;
    }
    if (ABt9) from_mat(dbg.ABt, ABt9);
    if (corr_idx) std::memcpy(corr_idx, icp.last_corr_index().data(), h->pcs.size() * sizeof(int));
}
void orc_closest_orthogonal(const float* ABt9, float* R9) { from_mat(closest_orthogonal_approximation(to_mat(ABt9)), R9); }
void orc_svd3(const double* A9, double* U9, double* S3, double* V9) { svd3_rowmajor(A9, U9, S3, V9); }

// ---- driver -------------------------------------------------------------------------------
void* orc_goicp_create(const float* tgt, size_t nt, const float* src, size_t ns, float lut_res, float mse_thr) {
    auto* h = new GoicpHandle;
    h->g.reset(new FastGoICP(to_cloud(tgt, nt), to_cloud(src, ns), lut_res, mse_thr));
    return h;
}
void* orc_goicp_create_trim(const float* tgt, size_t nt, const float* src, size_t ns, float lut_res, float mse_thr, float trim) {
    auto* h = new GoicpHandle;
    h->g.reset(new FastGoICP(to_cloud(tgt, nt), to_cloud(src, ns), lut_res, mse_thr, trim));
    return h;
}
void orc_goicp_destroy(void* p) { delete static_cast<GoicpHandle*>(p); }
// exact NN through the uniform grid instead of the O(ns*nt) loops (bit-identical results; tools/convention_flips.py, to finish in minutes)
void orc_goicp_use_grid(void* p, int on) { static_cast<GoicpHandle*>(p)->g->registration.use_grid(on != 0); }
// offs6 = offset_pcs, offset_pct; bounds6 as (minx,maxx,miny,maxy,minz,maxz); clouds may be null
void orc_goicp_preproc(void* p, float* offs6, float* scale, float* bounds6, float* tgt_scaled, float* src_scaled, int* lut_dims3) {
    auto& g = *static_cast<GoicpHandle*>(p)->g;
    offs6[0] = g.offset_pcs.x; offs6[1] = g.offset_pcs.y; offs6[2] = g.offset_pcs.z;
    offs6[3] = g.offset_pct.x; offs6[4] = g.offset_pct.y; offs6[5] = g.offset_pct.z;
    *scale = g.scaling_factor;
    for (int i = 0; i < 3; ++i) { bounds6[2 * i] = g.target_bounds[i].first; bounds6[2 * i + 1] = g.target_bounds[i].second; }
    if (tgt_scaled) std::memcpy(tgt_scaled, g.pct.data(), g.nt * sizeof(Vec3));
    if (src_scaled) std::memcpy(src_scaled, g.pcs.data(), g.ns * sizeof(Vec3));
    if (lut_dims3) for (int i = 0; i < 3; ++i) lut_dims3[i] = g.registration.nnlut.dims[i];
}
// stats7: trans_cubes, bounds_calls, rot_cubes, icp_runs, icp_iters, inner_bnb, reserved
void orc_goicp_run(void* p, float* R9, float* t3, float* best_sse, float* t_scaled3, unsigned long long* stats7) {
    auto& g = *static_cast<GoicpHandle*>(p)->g;
    auto [R, t] = g.run();
    from_mat(R, R9);
    t3[0] = t.x; t3[1] = t.y; t3[2] = t.z;
    *best_sse = g.get_best_error();
    t_scaled3[0] = g.best_translation.x; t_scaled3[1] = g.best_translation.y; t_scaled3[2] = g.best_translation.z;
    const RunStats& s = g.stats();
    stats7[0] = s.trans_cubes; stats7[1] = s.bounds_calls; stats7[2] = s.rot_cubes; stats7[3] = s.icp_runs;
    stats7[4] = s.icp_iters; stats7[5] = s.inner_bnb; stats7[6] = 0;
}

}  // extern "C"
