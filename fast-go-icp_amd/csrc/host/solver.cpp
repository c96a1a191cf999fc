// Driver-level C ABI: icp::FastGoICP (reference fgoicp/fgoicp.hpp:13-43) over the HIP operator
// context.  The driver template is instantiated with the HIP backend ONLY — there is no CPU
// backend in this library.
#include <cstddef>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/fgoicp_amd.h"
#include "../device/ctx.hpp"
#include "driver.hpp"

namespace fgoicp {

struct HipOps {
    fgoicp_ctx* ctx = nullptr;
    int bounds_multi(int G, const float* R9, const float* rot_span, const int* fix_rot, const int* offsets, const float* tn4, float* lb,
                     float* ub, const float* cut_above) {
        return ctx_bounds_multi(ctx, G, R9, rot_span, fix_rot, offsets, tn4, lb, ub, cut_above);
    }
    int bounds_submit(int slot, int G, const float* R9, const float* rot_span, const int* fix_rot, const int* offsets, const float* tn4,
                      const int* twin, const float* cut_above) {
        return ctx_bounds_submit(ctx, slot, G, R9, rot_span, fix_rot, offsets, tn4, twin, cut_above);
    }
    int bounds_collect(int slot, float* lb, float* ub) { return ctx_bounds_collect(ctx, slot, lb, ub); }
    bool async() const { return ctx->sorted_bounds && pipeline; }
    bool twins() const { return true; }
    int icp(const float* R0, const float* t0, size_t max_iter, float thr, float* sse, float* R9, float* t3, int* iters) {
        return ctx_icp(ctx, R0, t0, max_iter, thr, sse, R9, t3, iters);
    }
    // one ICP run executed by all ranks together (cooperative refinements of the sharded ROUND schedule)
    int icp_coop(int rank, int world, int (*gather)(void*, size_t, void*), void* user, const float* R0, const float* t0, size_t max_iter, float thr, float* sse,
                 float* R9, float* t3, int* iters) {
        return ctx_icp_coop(ctx, rank, world, gather, user, R0, t0, max_iter, thr, sse, R9, t3, iters);
    }
    // a refinement that runs next to the bounds work (late-joining ICP of the ROUND schedule): ICP lane 1, its own streams
    int icp_background(const float* R0, const float* t0, size_t max_iter, float thr, float* sse, float* R9, float* t3, int* iters) {
        return ctx_icp_lane(ctx, 1, R0, t0, max_iter, thr, sse, R9, t3, iters);
    }
    bool pipeline = [] { const char* e = dev_env("FGOICP_PIPELINE"); return e ? std::atoi(e) != 0 : true; }();  // tuning knob
};

}  // namespace fgoicp

using namespace fgoicp;

struct fgoicp_solver {
    // FastGoICP members in declaration order (fgoicp.hpp:47-58)
    std::vector<Vec3f> pcs, pct;
    size_t ns = 0, nt = 0;
    Vec3f offset_pcs{0, 0, 0}, offset_pct{0, 0, 0};
    float scaling_factor = 1.f;
    float bounds6[6] = {0, 0, 0, 0, 0, 0};
    fgoicp_ctx* ctx = nullptr;  // "registration"
    HipOps ops;
    std::unique_ptr<GoIcpDriver<HipOps>> driver;
    fgoicp_exchange ex{};
    bool has_ex = false;
    ~fgoicp_solver() {  // the context goes with the solver however the solver goes (fgoicp_solver_destroy, a failed or throwing create)
        driver.reset();
        fgoicp_ctx_destroy(ctx);
    }
};

extern "C" {

static int solver_create_impl(const float* tgt_xyz, size_t nt, const float* src_xyz, size_t ns, float lut_resolution, float mse_threshold, const fgoicp_solver_opts* opts,
                              fgoicp_solver** out);
int fgoicp_solver_create(const float* tgt_xyz, size_t nt, const float* src_xyz, size_t ns, float lut_resolution, float mse_threshold,
                         const fgoicp_solver_opts* opts, fgoicp_solver** out) {
    if (!out) return FGOICP_ERR_INVALID_ARG;
    *out = nullptr;
    // (the solver under construction is held by a unique_ptr inside: an exception unwinds it, and its destructor frees the context)
    return fgoicp::abi_guard("fgoicp_solver_create", [&] { return solver_create_impl(tgt_xyz, nt, src_xyz, ns, lut_resolution, mse_threshold, opts, out); });
}
static int solver_create_impl(const float* tgt_xyz, size_t nt, const float* src_xyz, size_t ns, float lut_resolution, float mse_threshold, const fgoicp_solver_opts* opts,
                              fgoicp_solver** out) {
    if (!tgt_xyz || !src_xyz || nt == 0 || ns == 0 || !(lut_resolution > 0) || !(mse_threshold >= 0)) {
        set_error("fgoicp_solver_create: invalid argument");
        return FGOICP_ERR_INVALID_ARG;
    }
    fgoicp_solver_opts o{FGOICP_SCHEDULE_SERIAL, 1, 0u, 0, 0.0f};
    if (opts) o = *opts;
    auto s = std::make_unique<fgoicp_solver>();
    s->ns = ns;
    s->nt = nt;
    s->pcs.resize(ns);
    s->pct.resize(nt);
    std::memcpy(s->pcs.data(), src_xyz, sizeof(Vec3f) * ns);
    std::memcpy(s->pct.data(), tgt_xyz, sizeof(Vec3f) * nt);
    // member-initialiser order of the reference ctor (fgoicp.hpp:13-19)
    s->offset_pcs = center_point_cloud(s->pcs);
    s->offset_pct = center_point_cloud(s->pct);
    s->scaling_factor = scale_point_clouds(s->pct, s->pcs);
    point_cloud_ranges(s->pct, s->bounds6);
    int rc = fgoicp_ctx_create(reinterpret_cast<const float*>(s->pct.data()), nt, reinterpret_cast<const float*>(s->pcs.data()), ns,
                               s->bounds6, lut_resolution, o.device, o.ctx_flags | (o.trim_fraction > 0.0f ? (unsigned)FGOICP_FLAG_CURVE_ORDER : 0u), &s->ctx);
    if (rc) return rc;
    s->ops.ctx = s->ctx;
    size_t n_thr = ns;  // sse_threshold = n * mse_threshold (fgoicp.hpp:23); over the inliers when trimming
    if (o.trim_fraction > 0.0f) {  // inlierNum = (int)(Nd * (1 - trimFraction)), as in Go-ICP
        size_t k = (size_t)((double)ns * (1.0 - (double)o.trim_fraction));
        if (k < 1) k = 1;
        if (k < ns) {
            rc = ctx_set_inliers(s->ctx, k);
            if (rc) return rc;
            n_thr = k;
        }
    }
    s->driver.reset(new GoIcpDriver<HipOps>(s->ops, n_thr, mse_threshold, o.schedule, o.round_width));
    *out = s.release();
    return FGOICP_OK;
}

void fgoicp_solver_destroy(fgoicp_solver* s) {
    if (!s) return;
    delete s;
}

int fgoicp_solver_set_exchange(fgoicp_solver* s, const fgoicp_exchange* ex_in) {
    if (!s) return FGOICP_ERR_INVALID_ARG;
    Exchange e;
    if (ex_in) {
        // only the bytes the caller's struct has are read (struct_size, ABI 2): members it does not know stay NULL
        fgoicp_exchange x{};
        const size_t n = ex_in->struct_size;
        if (n < offsetof(fgoicp_exchange, user) + sizeof(void*) || n > 4096) { set_error("fgoicp_solver_set_exchange: set struct_size = sizeof(fgoicp_exchange)"); return FGOICP_ERR_INVALID_ARG; }
        std::memcpy(&x, ex_in, n < sizeof(x) ? n : sizeof(x));
        const fgoicp_exchange* ex = &x;
        if (ex->world_size < 1 || ex->rank < 0 || ex->rank >= ex->world_size || (ex->world_size > 1 && (!ex->allreduce_min || !ex->allgather))) {
            set_error("fgoicp_solver_set_exchange: invalid exchange");
            return FGOICP_ERR_INVALID_ARG;
        }
        s->ex = *ex;
        s->has_ex = true;
        e.rank = ex->rank;
        e.world = ex->world_size;
        e.allreduce_min = ex->allreduce_min;
        e.allgather = ex->allgather;
        e.user = ex->user;
        e.allgather_device = ex->allgather_device;
    } else {
        s->has_ex = false;
    }
    s->driver->set_exchange(e);
    return FGOICP_OK;
}

int fgoicp_solver_set_log(fgoicp_solver* s, fgoicp_log_fn cb, void* user) {
    if (!s) return FGOICP_ERR_INVALID_ARG;
    if (!cb) { s->driver->set_log(nullptr); return FGOICP_OK; }
    s->driver->set_log([s, cb, user](int event, float sse, const Mat3f& R, const Vec3f& t) {
        // the initial ICP's translation is printed as returned (fgoicp.cpp:17), the incumbent's restored (:87, fgoicp.hpp:87-90)
        const Vec3f tr = event == FGOICP_LOG_NEW_BEST ? t / s->scaling_factor + R * s->offset_pcs - s->offset_pct : t;
        const float t3[3] = {tr.x, tr.y, tr.z};
        cb(event, sse, R.m, t3, user);
    });
    return FGOICP_OK;
}

int fgoicp_solver_set_early_exit(fgoicp_solver* s, int on) {
    if (!s) return FGOICP_ERR_INVALID_ARG;
    s->driver->set_use_cut(on != 0);
    return FGOICP_OK;
}

static int solver_run_impl(fgoicp_solver* s, float* R_out9, float* t_out3);
int fgoicp_solver_run(fgoicp_solver* s, float* R_out9, float* t_out3) {
    if (!s || !R_out9 || !t_out3) return FGOICP_ERR_INVALID_ARG;
    return fgoicp::abi_guard("fgoicp_solver_run", [&] { return solver_run_impl(s, R_out9, t_out3); });  // the host driver allocates (queues, tick buffers)
}
static int solver_run_impl(fgoicp_solver* s, float* R_out9, float* t_out3) {
    int rc = s->driver->run();
    if (rc == kDriverExchangeFailed) set_error("fgoicp_solver_run: exchange callback failed");
    if (rc) return rc;
    Mat3f R;
    Vec3f t;
    s->driver->best_transform(R, t);
    // restore_translation, fgoicp.hpp:87-90
    const Vec3f tr = t / s->scaling_factor + R * s->offset_pcs - s->offset_pct;
    std::memcpy(R_out9, R.m, sizeof(R.m));
    t_out3[0] = tr.x; t_out3[1] = tr.y; t_out3[2] = tr.z;
    return FGOICP_OK;
}

int fgoicp_solver_best_error(const fgoicp_solver* s, float* sse_out) {
    if (!s || !sse_out) return FGOICP_ERR_INVALID_ARG;
    *sse_out = s->driver->best_sse();
    return FGOICP_OK;
}

int fgoicp_solver_best_transform(const fgoicp_solver* s, float* R9, float* t3) {
    if (!s || !R9 || !t3) return FGOICP_ERR_INVALID_ARG;
    Mat3f R; Vec3f t;
    s->driver->best_transform(R, t);
    std::memcpy(R9, R.m, sizeof(R.m));
    t3[0] = t.x; t3[1] = t.y; t3[2] = t.z;
    return FGOICP_OK;
}

int fgoicp_solver_last_transform(const fgoicp_solver* s, float* R9, float* t3) {
    if (!s || !R9 || !t3) return FGOICP_ERR_INVALID_ARG;
    Mat3f R; Vec3f t;
    s->driver->last_transform(R, t);
    std::memcpy(R9, R.m, sizeof(R.m));
    t3[0] = t.x; t3[1] = t.y; t3[2] = t.z;
    return FGOICP_OK;
}

int fgoicp_solver_stats(const fgoicp_solver* s, fgoicp_run_stats* out) {
    if (!s || !out) return FGOICP_ERR_INVALID_ARG;
    const DriverStats& d = s->driver->stats();
    out->trans_cubes = d.trans_cubes; out->bounds_calls = d.bounds_calls; out->rot_cubes = d.rot_cubes;
    out->icp_runs = d.icp_runs; out->icp_iters = d.icp_iters; out->inner_bnb = d.inner_bnb; out->rounds = d.rounds;
    out->seconds_total = d.seconds_total; out->seconds_bnb = d.seconds_bnb; out->seconds_icp = d.seconds_icp; out->initial_icp_sse = d.initial_icp_sse;
    return FGOICP_OK;
}

int fgoicp_solver_preproc(const fgoicp_solver* s, float* offs6, float* scale, float* bounds6) {
    if (!s) return FGOICP_ERR_INVALID_ARG;
    if (offs6) {
        offs6[0] = s->offset_pcs.x; offs6[1] = s->offset_pcs.y; offs6[2] = s->offset_pcs.z;
        offs6[3] = s->offset_pct.x; offs6[4] = s->offset_pct.y; offs6[5] = s->offset_pct.z;
    }
    if (scale) *scale = s->scaling_factor;
    if (bounds6) std::memcpy(bounds6, s->bounds6, sizeof(s->bounds6));
    return FGOICP_OK;
}

int fgoicp_cloud_stats(const float* xyz, size_t n, fgoicp_cloud_stats_t* out) {
    if (!out || (!xyz && n)) return FGOICP_ERR_INVALID_ARG;
    *out = fgoicp_cloud_stats_t{};
    out->n = n;
    if (n == 0) return FGOICP_OK;
    double sum[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) { out->min[k] = xyz[k]; out->max[k] = xyz[k]; }
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            const float v = xyz[3 * i + k];
            sum[k] += (double)v;
            out->min[k] = v < out->min[k] ? v : out->min[k];
            out->max[k] = v > out->max[k] ? v : out->max[k];
        }
    double c[3];
    for (int k = 0; k < 3; ++k) { c[k] = sum[k] / (double)n; out->centroid[k] = (float)c[k]; }
    double r2 = 0.0, mx = 0.0;
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            const double d = (double)xyz[3 * i + k] - c[k];
            r2 += d * d;
            mx = std::fabs(d) > mx ? std::fabs(d) : mx;
        }
    out->max_abs_centred = (float)mx;
    out->rms_radius = (float)std::sqrt(r2 / (double)n);
    return FGOICP_OK;
}

fgoicp_ctx* fgoicp_solver_ctx(fgoicp_solver* s) { return s ? s->ctx : nullptr; }

}  // extern "C"
