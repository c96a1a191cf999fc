// icp::IterativeClosestPoint3D for C++ callers — reference fgoicp/icp3d.hpp:9-41.
#pragma once
#include <tuple>

#include "registration.hpp"

namespace icp {

class IterativeClosestPoint3D {
public:
    // icp3d.hpp:30 — the clouds already live in `reg`'s device context; pct/pcs are accepted for
    // signature compatibility and must be the clouds `reg` was built from.
    IterativeClosestPoint3D(const Registration& reg, const PointCloud& pct, const PointCloud& pcs, size_t max_iter,
                            float convergence_threshold, mat3 R, vec3 t)
        : reg_(reg), max_iter_(max_iter), thr_(convergence_threshold), R_(R), t_(t) {
        if (pct.size() != reg.nt || pcs.size() != reg.ns) throw std::runtime_error("IterativeClosestPoint3D: clouds do not match the Registration");
    }
    using Result_t = std::tuple<float, mat3, vec3>;
    Result_t run() {  // icp3d.hpp:35
        float sse = 0.f;
        mat3 R;
        vec3 t;
        check_status(fgoicp_icp(reg_.handle(), R_.data(), &t_.x, max_iter_, thr_, &sse, R.data(), &t.x, &iterations_), "fgoicp_icp");
        return {sse, R, t};
    }
    int iterations() const { return iterations_; }
private:
    const Registration& reg_;
    size_t max_iter_;
    float thr_;
    mat3 R_;
    vec3 t_;
    int iterations_ = 0;
};

}  // namespace icp
