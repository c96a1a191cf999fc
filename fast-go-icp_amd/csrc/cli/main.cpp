// fast-go-icp — the reference's CLI (src/main.cpp:8-58) over libfgoicp_amd.so:
//     fast-go-icp -c config.toml [-v]
// Same flags, same TOML keys / defaults / clamps, same log lines; the search runs on the MI355X.
#include <chrono>
#include <cstdlib>
#include <filesystem>
#include <string>

#include "../../../include/fgoicp/fgoicp.hpp"
#include "config.hpp"

static std::string usage(const std::string& exe) {
    return "Fast Go-ICP: an MI355X (HIP) implementation of Go-ICP\nUsage: " + exe + " [OPTIONS]\n\nOptions:\n"
           "  -h,--help                   Print this help message and exit\n"
           "  -c,--config TEXT REQUIRED   Path to the TOML configuration file\n"
           "  -v,--verbose                Enable verbose logging\n"
           "  -g,--gpus N                 Shard the search over N GPUs of this node (default: params.gpus, 1)\n\nExample Usage:\n  " + exe + " -c config.toml --verbose\n  " + exe + " --config=config.toml\n";
}

int main(int argc, char* argv[]) {
    std::string config_file;
    bool verbose = false;
    int gpus_flag = 0;
    const std::string exe = std::filesystem::path(argv[0]).filename().string();
    auto fail = [&](const std::string& what, int code) {
        icp::Logger(icp::LogLevel::Error) << what;
        icp::Logger(icp::LogLevel::Info) << usage(exe);
        std::exit(code);
    };
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-h" || a == "--help") { std::cout << usage(exe); return 0; }
        else if (a == "-v" || a == "--verbose") verbose = true;
        else if (a == "-g" || a == "--gpus") { if (i + 1 >= argc) fail("--gpus: 1 required INT missing", 114); gpus_flag = std::atoi(argv[++i]); }
        else if (a.rfind("--gpus=", 0) == 0) gpus_flag = std::atoi(a.substr(7).c_str());
        else if (a == "-c" || a == "--config") { if (i + 1 >= argc) fail("--config: 1 required TEXT missing", 114); config_file = argv[++i]; }
        else if (a.rfind("--config=", 0) == 0) config_file = a.substr(9);
        else if (a.rfind("-c", 0) == 0 && a.size() > 2) config_file = a.substr(2);
        else fail("The following argument was not expected: " + a, 109);
    }
    if (config_file.empty()) fail("--config is required", 106);
    icp::Logger::set_verbose(verbose);

    cli::Config config(config_file);
    std::vector<icp::vec3> pct, pcs;
    cli::load_cloud(config.io.target, config.params.target_subsample, pct, config.params.seed);
    icp::Logger(icp::LogLevel::Info) << "Target point cloud (" << pct.size() << ") loaded from " << config.io.target;
    cli::load_cloud(config.io.source, config.params.source_subsample, pcs, config.params.seed < 0 ? -1 : config.params.seed + 1);
    icp::Logger(icp::LogLevel::Info) << "Source point cloud (" << pcs.size() << ") loaded from " << config.io.source;
    const std::vector<icp::vec3> pct_in = pct, pcs_in = pcs;
    for (const auto* pc : {&pct, &pcs}) {  // verbose: the statistics the pre-processing normalises by (TODO.md:7 of the reference)
        fgoicp_cloud_stats_t cs{};
        if (fgoicp_cloud_stats(&pc->data()->x, pc->size(), &cs) == FGOICP_OK)
            icp::Logger(icp::LogLevel::Debug) << (pc == &pct ? "Target" : "Source") << " statistics: centroid " << icp::vec3{cs.centroid[0], cs.centroid[1], cs.centroid[2]}
                                              << ", box [" << cs.min[0] << ", " << cs.max[0] << "] x [" << cs.min[1] << ", " << cs.max[1] << "] x [" << cs.min[2] << ", " << cs.max[2]
                                              << "], largest centred coordinate " << cs.max_abs_centred << ", RMS radius " << cs.rms_radius;
    }

    const int schedule = config.params.schedule == "round" ? FGOICP_SCHEDULE_ROUND : FGOICP_SCHEDULE_SERIAL;
    const int gpus = gpus_flag > 0 ? gpus_flag : config.params.gpus;
    icp::mat3 R;
    icp::vec3 t;
    fgoicp_run_stats st{};
    float best_error = 0.f;
    std::chrono::duration<double> elapsed_seconds{};
    if (gpus > 1) {
        // EXTENSION: one host thread + one solver per GPU (include/fgoicp_amd.h, fgoicp_multi_*).  params.schedule = "serial" (the
        // default) keeps the reference's exact trajectory and deals the inner BnBs of every speculative evaluation over the GPUs;
        // "round" deals the children of every expansion round (one RCCL all-gather per round; fastest).
        std::vector<int> devices;
        for (int d = 0; d < gpus; ++d) devices.push_back(d);
        int transport = FGOICP_TRANSPORT_RCCL;
        if (const char* e = std::getenv("FGOICP_MULTI_DEVICES")) {  // e.g. "0,0": rehearse two ranks on one GPU (in-process transport)
            devices.clear();
            for (const char* p = e; *p;) { devices.push_back(std::atoi(p)); while (*p && *p != ',') ++p; if (*p) ++p; }
            transport = FGOICP_TRANSPORT_IN_PROCESS;
        }
        icp::Logger(icp::LogLevel::Info) << "Sharding the search over " << devices.size() << " GPUs (schedule: "
                                         << (schedule == FGOICP_SCHEDULE_ROUND ? "expansion rounds" : "the reference's order, evaluations sharded") << ")";
        fgoicp_solver_opts o{schedule, schedule == FGOICP_SCHEDULE_ROUND ? config.params.round_width : 1, 0u, 0, config.params.trim_fraction};
        fgoicp_multi* m = nullptr;
        icp::check_status(fgoicp_multi_create(&pct.data()->x, pct.size(), &pcs.data()->x, pcs.size(), config.params.lut_resolution, config.params.mse_threshold, &o,
                                              devices.data(), (int)devices.size(), transport, &m), "fgoicp_multi_create");
        // rank 0's log events: every rank holds the same incumbents (SERIAL and the cooperative flow: the same refinements too)
        icp::check_status(fgoicp_solver_set_log(fgoicp_multi_solver(m, 0), &icp::FastGoICP::log_line, nullptr), "fgoicp_solver_set_log");
        auto start = std::chrono::high_resolution_clock::now();
        icp::check_status(fgoicp_multi_run(m, R.data(), &t.x), "fgoicp_multi_run");
        elapsed_seconds = std::chrono::high_resolution_clock::now() - start;
        // ROUND: counters summed over the ranks; SERIAL: every rank holds the whole trajectory's counters (rank 0's are reported).
        // Rank 0's incumbent is every rank's.
        for (int r = 0; r < (schedule == FGOICP_SCHEDULE_ROUND ? (int)devices.size() : 1); ++r) {
            fgoicp_run_stats s1{};
            icp::check_status(fgoicp_solver_stats(fgoicp_multi_solver(m, r), &s1), "fgoicp_solver_stats");
            st.trans_cubes += s1.trans_cubes; st.rot_cubes += s1.rot_cubes; st.icp_runs += s1.icp_runs; st.icp_iters += s1.icp_iters;
            st.bounds_calls += s1.bounds_calls; st.inner_bnb += s1.inner_bnb;
            if (r == 0) { st.rounds = s1.rounds; st.initial_icp_sse = s1.initial_icp_sse; }
        }
        icp::check_status(fgoicp_solver_best_error(fgoicp_multi_solver(m, 0), &best_error), "fgoicp_solver_best_error");
        fgoicp_multi_destroy(m);
        icp::Logger(icp::LogLevel::Info) << "Searching over! Best Error: " << best_error << "\n\tRotation:\n" << R << "\n\tTranslation: " << t;  // fgoicp.cpp:25-27
    } else {
        icp::FastGoICP fgoicp(std::move(pct), std::move(pcs), config.params.lut_resolution, config.params.mse_threshold, schedule,
                              config.params.round_width, 0, config.params.trim_fraction);
        auto start = std::chrono::high_resolution_clock::now();
        std::tie(R, t) = fgoicp.run();
        elapsed_seconds = std::chrono::high_resolution_clock::now() - start;
        st = fgoicp.stats();
        best_error = fgoicp.get_best_error();
    }
    icp::Logger(icp::LogLevel::Debug) << "Subcubes: " << st.trans_cubes << ", rotation cubes: " << st.rot_cubes << ", ICP runs: " << st.icp_runs;
    icp::Logger(icp::LogLevel::Info) << "Fast Go-ICP finished, time elapsed: " << std::fixed << std::setprecision(3) << elapsed_seconds.count() << " seconds";
    if (!config.io.output.empty()) cli::write_result_toml(config.io.output, R, t, best_error, pcs_in.size(), elapsed_seconds.count(), st);
    if (!config.io.visualization.empty()) cli::write_visualization_ply(config.io.visualization, pct_in, pcs_in, R, t);
    return 0;
}
