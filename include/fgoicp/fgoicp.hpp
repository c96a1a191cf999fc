// icp::FastGoICP for C++ callers — reference fgoicp/fgoicp.hpp:8-110.  The branch-and-bound runs in
// host C++ inside libfgoicp_amd.so (fast-go-icp_amd/csrc/host/driver.hpp) and calls the HIP operators.
#pragma once
#include <tuple>
#include <vector>

#include "common.hpp"

namespace icp {

class FastGoICP {
public:
    // fgoicp.hpp:13 (+ optional schedule: FGOICP_SCHEDULE_SERIAL reproduces the reference's order)
    FastGoICP(std::vector<vec3> pct, std::vector<vec3> pcs, float lut_resolution, float mse_threshold,
              int schedule = FGOICP_SCHEDULE_SERIAL, int round_width = 1, int device = 0, float trim_fraction = 0.0f) {
        fgoicp_solver_opts o{schedule, round_width, 0u, device, trim_fraction};
        check_status(fgoicp_solver_create(&pct.data()->x, pct.size(), &pcs.data()->x, pcs.size(), lut_resolution, mse_threshold, &o, &s_),
                     "fgoicp_solver_create");
        check_status(fgoicp_solver_set_log(s_, &FastGoICP::log_line, nullptr), "fgoicp_solver_set_log");
    }
    ~FastGoICP() { fgoicp_solver_destroy(s_); }
    FastGoICP(const FastGoICP&) = delete;
    FastGoICP& operator=(const FastGoICP&) = delete;

    using Result_t = std::tuple<mat3, vec3>;
    Result_t run() {  // fgoicp.hpp:28
        mat3 R;
        vec3 t;
        check_status(fgoicp_solver_run(s_, R.data(), &t.x), "fgoicp_solver_run");
        Logger(LogLevel::Info) << "Searching over! Best Error: " << get_best_error() << "\n\tRotation:\n" << R << "\n\tTranslation: " << t;  // fgoicp.cpp:25-27
        return {R, t};
    }
    // interfaces for visualisation, fgoicp.hpp:31-43 (safe to poll from another thread)
    float get_best_error() const { float v = 0; check_status(fgoicp_solver_best_error(s_, &v), "fgoicp_solver_best_error"); return v; }
    Result_t get_best_transform() const { mat3 R; vec3 t; check_status(fgoicp_solver_best_transform(s_, R.data(), &t.x), "fgoicp_solver_best_transform"); return {R, t}; }
    Result_t get_last_transform() const { mat3 R; vec3 t; check_status(fgoicp_solver_last_transform(s_, R.data(), &t.x), "fgoicp_solver_last_transform"); return {R, t}; }

    // not in the reference: false = every subcube is evaluated in full, as kernComputeBounds does (default: the inner BnBs tell the bounds
    // operator what they do not need to know, fgoicp_bounds_submit_cut — same trajectory, counters and result)
    void set_early_exit(bool on) { check_status(fgoicp_solver_set_early_exit(s_, on ? 1 : 0), "fgoicp_solver_set_early_exit"); }
    fgoicp_run_stats stats() const { fgoicp_run_stats st{}; check_status(fgoicp_solver_stats(s_, &st), "fgoicp_solver_stats"); return st; }
    fgoicp_solver* handle() const { return s_; }
    // the reference's own lines while the search runs (fgoicp.cpp:15-17 Info, :85-87 Debug), from the driver's log events
    static void log_line(int event, float sse, const float* R9, const float* t3, void*) {
        mat3 R;
        for (int k = 0; k < 9; ++k) R.data()[k] = R9[k];
        const vec3 t{t3[0], t3[1], t3[2]};
        if (event == FGOICP_LOG_INITIAL_ICP) Logger(LogLevel::Info) << "Initial ICP best error: " << sse << "\n\tRotation:\n" << R << "\n\tTranslation: " << t;
        else Logger(LogLevel::Debug) << "New best error: " << sse << "\n\tRotation:\n" << R << "\n\tTranslation: " << t;
    }
private:
    fgoicp_solver* s_ = nullptr;
};

}  // namespace icp
