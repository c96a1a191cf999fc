// TEST-ONLY: exposes the CLI's host-side parsers (fast-go-icp_amd/csrc/cli/config.hpp: TOML subset,
// TXT/PLY loaders) through a C interface so that pytest can exercise them without a GPU.
#include <cstring>

#include "../../fast-go-icp_amd/csrc/cli/config.hpp"

extern "C" const char* fgoicp_last_error(void) { return ""; }  // icp::check_status is never reached here

extern "C" {

struct CliConfigOut {
    char target[512], source[512], output[512], visualization[512], schedule[64];
    int trim;
    float target_subsample, source_subsample, lut_resolution, mse_threshold;
    long long seed;
    int round_width;
};

int cli_parse_config(const char* path, CliConfigOut* out) {
    try {
        cli::Config c(path);
        std::snprintf(out->target, sizeof(out->target), "%s", c.io.target.c_str());
        std::snprintf(out->source, sizeof(out->source), "%s", c.io.source.c_str());
        std::snprintf(out->output, sizeof(out->output), "%s", c.io.output.c_str());
        std::snprintf(out->visualization, sizeof(out->visualization), "%s", c.io.visualization.c_str());
        std::snprintf(out->schedule, sizeof(out->schedule), "%s", c.params.schedule.c_str());
        out->trim = c.params.trim;
        out->target_subsample = c.params.target_subsample;
        out->source_subsample = c.params.source_subsample;
        out->lut_resolution = c.params.lut_resolution;
        out->mse_threshold = c.params.mse_threshold;
        out->seed = c.params.seed;
        out->round_width = c.params.round_width;
        return 0;
    } catch (const std::exception&) {
        return 1;
    }
}

// returns the number of points, or -1 and the message in err
long cli_load_cloud(const char* path, float subsample, long long seed, float* out_xyz, long capacity, char* err, int err_len) {
    try {
        std::vector<icp::vec3> cloud;
        size_t n = cli::load_cloud(path, subsample, cloud, seed);
        if ((long)cloud.size() > capacity) { std::snprintf(err, err_len, "capacity"); return -1; }
        if (n != cloud.size()) { std::snprintf(err, err_len, "count mismatch"); return -1; }
        std::memcpy(out_xyz, cloud.data(), cloud.size() * sizeof(icp::vec3));
        return (long)cloud.size();
    } catch (const std::exception& e) {
        std::snprintf(err, err_len, "%s", e.what());
        return -1;
    }
}
}
