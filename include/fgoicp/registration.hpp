// icp::Registration / icp::NearestNeighborLUT for C++ callers — same constructor arguments, method
// names and return order as the reference's fgoicp/registration.hpp:18-98; every call forwards to the
// HIP context behind include/fgoicp_amd.h.  Non-copyable (the reference's versions double-free on copy).
#pragma once
#include <array>
#include <tuple>
#include <utility>

#include "common.hpp"

namespace icp {

class Registration;

// registration.hpp:18-42.  The LUT lives inside the Registration's device context; this view exposes
// its geometry/contents.  `search` is a device function in the reference (registration.cu:320-328);
// here it is a host call evaluating the same lookup on the device for n points.
class NearestNeighborLUT {
public:
    std::array<int, 3> dims() const {
        std::array<int, 3> d{};
        check_status(fgoicp_lut_dims(ctx_, d.data()), "fgoicp_lut_dims");
        return d;
    }
    std::vector<float> data() const {
        auto d = dims();
        std::vector<float> out((size_t)d[0] * d[1] * d[2]);
        check_status(fgoicp_lut_read(ctx_, out.data(), out.size()), "fgoicp_lut_read");
        return out;
    }
    std::vector<float> search(const PointCloud& queries) const {
        std::vector<float> out(queries.size());
        check_status(fgoicp_lut_search(ctx_, &queries.data()->x, queries.size(), out.data()), "fgoicp_lut_search");
        return out;
    }
private:
    friend class Registration;
    fgoicp_ctx* ctx_ = nullptr;
};

class Registration {
public:
    // registration.hpp:68
    Registration(const PointCloud& pct, const PointCloud& pcs, const std::array<std::pair<float, float>, 3> target_bounds,
                 float lut_resolution, int device = 0, unsigned flags = 0)
        : nt(pct.size()), ns(pcs.size()) {
        const float b[6] = {target_bounds[0].first, target_bounds[0].second, target_bounds[1].first,
                            target_bounds[1].second, target_bounds[2].first, target_bounds[2].second};
        check_status(fgoicp_ctx_create(&pct.data()->x, nt, &pcs.data()->x, ns, b, lut_resolution, device, flags, &ctx_), "fgoicp_ctx_create");
        nnlut.ctx_ = ctx_;
    }
    ~Registration() { fgoicp_ctx_destroy(ctx_); }
    Registration(const Registration&) = delete;
    Registration& operator=(const Registration&) = delete;

    using BoundsResult_t = std::tuple<std::vector<float>, std::vector<float>>;

    // registration.hpp:96
    float compute_sse_error(mat3 R, vec3 t) const {
        float sse = 0.f;
        check_status(fgoicp_sse(ctx_, R.data(), &t.x, &sse), "fgoicp_sse");
        return sse;
    }
    // registration.hpp:97 — returns {lower, upper} (registration.cu:151)
    BoundsResult_t compute_sse_error(RotNode& rnode, std::vector<TransNode>& tnodes, bool fix_rot, StreamPool&) const {
        const size_t B = tnodes.size();
        std::vector<float> tn4(4 * B), lb(B), ub(B);
        for (size_t i = 0; i < B; ++i) { tn4[4 * i] = tnodes[i].t.x; tn4[4 * i + 1] = tnodes[i].t.y; tn4[4 * i + 2] = tnodes[i].t.z; tn4[4 * i + 3] = tnodes[i].span; }
        check_status(fgoicp_bounds_batch(ctx_, rnode.q.R.data(), rnode.span, tn4.data(), (int)B, fix_rot ? 1 : 0, lb.data(), ub.data()),
                     "fgoicp_bounds_batch");
        return {lb, ub};
    }

    fgoicp_ctx* handle() const { return ctx_; }
    const size_t nt, ns;
    NearestNeighborLUT nnlut;
private:
    fgoicp_ctx* ctx_ = nullptr;
};

}  // namespace icp
