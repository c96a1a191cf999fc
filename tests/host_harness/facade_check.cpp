// Test-only: the header-only C++ façades of include/fgoicp/*.hpp used the way the reference's fgoicp.cpp uses its classes
// (Registration + compute_sse_error overloads, NearestNeighborLUT, IterativeClosestPoint3D, FastGoICP).  Reads two clouds
// (count, then x y z per line, the reference's TXT format) and a parameter line, prints the results as one JSON object.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>

#include "../../include/fgoicp/fgoicp.hpp"
#include "../../include/fgoicp/icp3d.hpp"
#include "../../include/fgoicp/registration.hpp"

static icp::PointCloud read_txt(const std::string& path) {
    std::ifstream f(path);
    size_t n = 0;
    f >> n;
    icp::PointCloud pc(n);
    for (size_t i = 0; i < n; ++i) f >> pc[i].x >> pc[i].y >> pc[i].z;
    return pc;
}

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    // operator level: pre-processed clouds + bounds
    icp::PointCloud pct = read_txt(argv[1]), pcs = read_txt(argv[2]);
    const float res = std::stof(argv[3]);
    std::array<std::pair<float, float>, 3> bounds;
    {
        std::ifstream f(argv[4]);
        for (auto& b : bounds) f >> b.first >> b.second;
    }
    icp::Registration reg(pct, pcs, bounds, res);
    auto dims = reg.nnlut.dims();
    icp::RotNode rn(0.25f, -0.125f, 0.375f, 0.125f, 0.f, 0.f);
    std::vector<icp::TransNode> tn;
    for (int i = 0; i < 5; ++i) tn.emplace_back(0.1f * i - 0.2f, 0.05f * i, -0.03f * i, 0.25f, 0.f, 0.f);
    icp::StreamPool pool(32);
    auto [lb, ub] = reg.compute_sse_error(rn, tn, false, pool);
    const float sse = reg.compute_sse_error(rn.q.R, icp::vec3(0.01f, -0.02f, 0.005f));
    icp::IterativeClosestPoint3D icp3d(reg, pct, pcs, 100, 0.005f, rn.q.R, icp::vec3(0.01f, -0.02f, 0.005f));
    auto [icp_sse, icp_R, icp_t] = icp3d.run();
    // driver level first (raw clouds: argv[5], argv[6]) — FastGoICP::run() prints the reference's log lines (fgoicp.cpp:15-17, :25-27)
    // on stdout while it runs; the JSON record follows as the last line
    bool have_run = false;
    float run_sse = 0.f;
    icp::mat3 run_R;
    icp::vec3 run_t;
    if (argc >= 7) {
        icp::FastGoICP solver(read_txt(argv[5]), read_txt(argv[6]), res, 1e-3f);
        std::tie(run_R, run_t) = solver.run();
        run_sse = solver.get_best_error();
        have_run = true;
    }
    std::cout.flush();
    std::printf("{\"dims\": [%d, %d, %d], \"lb\": [", dims[0], dims[1], dims[2]);
    for (size_t i = 0; i < lb.size(); ++i) std::printf("%s%.9g", i ? ", " : "", lb[i]);
    std::printf("], \"ub\": [");
    for (size_t i = 0; i < ub.size(); ++i) std::printf("%s%.9g", i ? ", " : "", ub[i]);
    std::printf("], \"sse\": %.9g, \"icp_sse\": %.9g, \"icp_iters\": %d, \"icp_R\": [", sse, icp_sse, icp3d.iterations());
    for (int i = 0; i < 9; ++i) std::printf("%s%.9g", i ? ", " : "", icp_R.data()[i]);
    std::printf("], \"icp_t\": [%.9g, %.9g, %.9g]", icp_t.x, icp_t.y, icp_t.z);
    if (have_run) {
        std::printf(", \"run_sse\": %.9g, \"run_R\": [", run_sse);
        for (int i = 0; i < 9; ++i) std::printf("%s%.9g", i ? ", " : "", run_R.data()[i]);
        std::printf("], \"run_t\": [%.9g, %.9g, %.9g]", run_t.x, run_t.y, run_t.z);
    }
    std::printf("}\n");
    return 0;
}
