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
           "  -v,--verbose                Enable verbose logging\n\nExample Usage:\n  " + exe + " -c config.toml --verbose\n  " + exe + " --config=config.toml\n";
}

int main(int argc, char* argv[]) {
    std::string config_file;
    bool verbose = false;
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

    const int schedule = config.params.schedule == "round" ? FGOICP_SCHEDULE_ROUND : FGOICP_SCHEDULE_SERIAL;
    icp::FastGoICP fgoicp(std::move(pct), std::move(pcs), config.params.lut_resolution, config.params.mse_threshold, schedule,
                          config.params.round_width, 0, config.params.trim_fraction);

    auto start = std::chrono::high_resolution_clock::now();
    auto [R, t] = fgoicp.run();
    auto end = std::chrono::high_resolution_clock::now();
    std::chrono::duration<double> elapsed_seconds = end - start;
    const fgoicp_run_stats st = fgoicp.stats();
    icp::Logger(icp::LogLevel::Info) << "Initial ICP best error: " << st.initial_icp_sse;
    icp::Logger(icp::LogLevel::Info) << "Searching over! Best Error: " << fgoicp.get_best_error() << "\n\tRotation:\n" << R << "\n\tTranslation: " << t;
    icp::Logger(icp::LogLevel::Debug) << "Subcubes: " << st.trans_cubes << ", rotation cubes: " << st.rot_cubes << ", ICP runs: " << st.icp_runs;
    icp::Logger(icp::LogLevel::Info) << "Fast Go-ICP finished, time elapsed: " << std::fixed << std::setprecision(3) << elapsed_seconds.count() << " seconds";
    if (!config.io.output.empty()) cli::write_result_toml(config.io.output, R, t, fgoicp.get_best_error(), pcs_in.size(), elapsed_seconds.count(), st);
    if (!config.io.visualization.empty()) cli::write_visualization_ply(config.io.visualization, pct_in, pcs_in, R, t);
    return 0;
}
