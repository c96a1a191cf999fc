// CLI-side configuration and cloud loaders — the caller side of the drop-in (reference
// src/utilities.hpp:18-260, test/bunny.toml).  Own minimal parsers: a TOML subset (tables, string /
// bool / integer / float values, comments) and PLY (ascii, binary little/big endian; any scalar
// property types; list properties and foreign elements are skipped) — the reference vendors toml++,
// CLI11 and tinyply for these, none of which are copied here.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/fgoicp/common.hpp"

namespace cli {

// ------------------------------------------------------------------------------------------
// TOML subset
// ------------------------------------------------------------------------------------------
struct TomlValue {
    enum Kind { String, Bool, Number } kind = String;
    std::string s;
    bool b = false;
    double d = 0;
};
using TomlTable = std::map<std::string, std::map<std::string, TomlValue>>;  // [section][key]

inline std::string trim(const std::string& x) {
    size_t a = x.find_first_not_of(" \t\r\n"), b = x.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? "" : x.substr(a, b - a + 1);
}

inline TomlTable parse_toml(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("Error parsing file '" + path + "': could not open file");
    TomlTable t;
    std::string line, section;
    int lineno = 0;
    while (std::getline(f, line)) {
        ++lineno;
        // strip comments outside strings
        bool in_str = false;
        char q = 0;
        std::string clean;
        for (size_t i = 0; i < line.size(); ++i) {
            char ch = line[i];
            if (in_str) {
                clean += ch;
                if (ch == '\\' && q == '"' && i + 1 < line.size()) { clean += line[++i]; continue; }
                if (ch == q) in_str = false;
            } else if (ch == '"' || ch == '\'') { in_str = true; q = ch; clean += ch; }
            else if (ch == '#') break;
            else clean += ch;
        }
        clean = trim(clean);
        if (clean.empty()) continue;
        auto err = [&](const std::string& what) { return std::runtime_error("Error parsing file '" + path + "': line " + std::to_string(lineno) + ": " + what); };
        if (clean.front() == '[') {
            if (clean.back() != ']') throw err("unterminated table header");
            section = trim(clean.substr(1, clean.size() - 2));
            t[section];
            continue;
        }
        size_t eq = clean.find('=');
        if (eq == std::string::npos) throw err("expected key = value");
        std::string key = trim(clean.substr(0, eq)), val = trim(clean.substr(eq + 1));
        if (key.empty() || val.empty()) throw err("expected key = value");
        TomlValue v;
        if (val.front() == '"' || val.front() == '\'') {
            if (val.size() < 2 || val.back() != val.front()) throw err("unterminated string");
            v.kind = TomlValue::String;
            std::string raw = val.substr(1, val.size() - 2), out;
            if (val.front() == '"')
                for (size_t i = 0; i < raw.size(); ++i) {
                    if (raw[i] == '\\' && i + 1 < raw.size()) {
                        char n = raw[++i];
                        out += n == 'n' ? '\n' : n == 't' ? '\t' : n;
                    } else out += raw[i];
                }
            else out = raw;
            v.s = out;
        } else if (val == "true" || val == "false") {
            v.kind = TomlValue::Bool;
            v.b = val == "true";
        } else {
            std::string num;
            for (char ch : val) if (ch != '_') num += ch;
            if (num == "inf" || num == "+inf") v.d = std::numeric_limits<double>::infinity();
            else if (num == "-inf") v.d = -std::numeric_limits<double>::infinity();
            else {
                char* end = nullptr;
                v.d = std::strtod(num.c_str(), &end);
                if (end == num.c_str() || *end != '\0') throw err("unsupported value '" + val + "'");
            }
            v.kind = TomlValue::Number;
        }
        t[section][key] = v;
    }
    return t;
}

// ------------------------------------------------------------------------------------------
// Config — src/utilities.hpp:18-107 (same keys, defaults and clamps; `seed` is an addition)
// ------------------------------------------------------------------------------------------
struct Config {
    struct IO { std::string target, source, output, visualization; } io;
    struct Params {
        bool trim = false;
        float target_subsample = 1.0f, source_subsample = 1.0f, lut_resolution = 0.005f, mse_threshold = 1e-3f;
        long long seed = -1;       // < 0: std::random_device, as the reference (utilities.hpp:149-150)
        std::string schedule = "serial";
        int round_width = 1;
        float trim_fraction = 0.0f;  // EXTENSION: > 0 enables trimmed Go-ICP; `trim` itself stays parsed-and-ignored as upstream
        int gpus = 1;                // EXTENSION: > 1 shards the outer BnB over that many GPUs of this node (one host thread each, RCCL)
    } params;

    explicit Config(const std::string& toml_filepath) {
        std::string base = toml_filepath.substr(toml_filepath.find_last_of("/\\") + 1);
        icp::Logger(icp::LogLevel::Info) << "Reading configurations from " << base;
        TomlTable tbl;
        try {
            tbl = parse_toml(toml_filepath);
        } catch (const std::exception& e) {
            icp::Logger(icp::LogLevel::Error) << e.what() << "\n";
            std::exit(1);  // utilities.hpp:69-78
        }
        auto str = [&](const char* sec, const char* key, const std::string& def) {
            auto s = tbl.find(sec);
            if (s == tbl.end()) return def;
            auto k = s->second.find(key);
            return (k == s->second.end() || k->second.kind != TomlValue::String) ? def : k->second.s;
        };
        auto num = [&](const char* sec, const char* key, double def) {
            auto s = tbl.find(sec);
            if (s == tbl.end()) return def;
            auto k = s->second.find(key);
            return (k == s->second.end() || k->second.kind != TomlValue::Number) ? def : k->second.d;
        };
        auto boolean = [&](const char* sec, const char* key, bool def) {
            auto s = tbl.find(sec);
            if (s == tbl.end()) return def;
            auto k = s->second.find(key);
            return (k == s->second.end() || k->second.kind != TomlValue::Bool) ? def : k->second.b;
        };
        io.target = str("io", "target", "");
        io.source = str("io", "source", "");
        io.output = str("io", "output", "");                // declared in test/bunny.toml:10, unparsed upstream
        io.visualization = str("io", "visualization", "");  // declared in test/bunny.toml:11, unparsed upstream
        auto clampf0 = [](float x) { return x < 0.0f ? 0.0f : (x > 0.9f ? 0.9f : x); };
        if (tbl.count("params")) {
            params.trim = boolean("params", "trim", false);
            params.target_subsample = (float)num("params", "target_subsample", 1.0);
            params.source_subsample = (float)num("params", "source_subsample", 1.0);
            params.lut_resolution = (float)num("params", "lut_resolution", 0.005);
            params.mse_threshold = (float)num("params", "mse_threshold", 1e-3);
            params.seed = (long long)num("params", "seed", -1);
            params.schedule = str("params", "schedule", "serial");
            params.round_width = (int)num("params", "round_width", 1);
            params.trim_fraction = clampf0((float)num("params", "trim_fraction", 0.0));
            params.gpus = std::max(1, (int)num("params", "gpus", 1));
            auto clampf = [](float x, float lo, float hi) { return x < hi ? (x > lo ? x : lo) : hi; };
            params.target_subsample = clampf(params.target_subsample, 1e-5f, 1.0f);  // utilities.hpp:101-104
            params.source_subsample = clampf(params.source_subsample, 1e-5f, 1.0f);
            params.source_subsample = clampf(params.source_subsample, 1e-5f, 0.5f);
            params.mse_threshold = clampf(params.mse_threshold, 1e-12f, INFINITY);
        }
        icp::Logger(icp::LogLevel::Info) << *this;
    }

    friend std::ostream& operator<<(std::ostream& os, const Config& c) {  // utilities.hpp:45-58
        os << "Fast Go-ICP Configurations\n"
           << "\tIO Configuration:\n"
           << "\t\tTarget: " << c.io.target << "\n"
           << "\t\tSource: " << c.io.source << "\n"
           << "\tParameters:\n"
           << "\t\tTrim: " << (c.params.trim ? "true" : "false") << "\n"
           << "\t\tTarget Subsample: " << c.params.target_subsample << "\n"
           << "\t\tSource Subsample: " << c.params.source_subsample << "\n"
           << "\t\tLUT Resolution: " << c.params.lut_resolution << "\n"
           << "\t\tMSE Threshold: " << c.params.mse_threshold;
        return os;
    }
};

// ------------------------------------------------------------------------------------------
// Loaders — src/utilities.hpp:113-260
// ------------------------------------------------------------------------------------------
struct Subsampler {  // keep each point with probability `subsample` until floor(total*subsample) are kept
    std::mt19937 gen;
    std::uniform_real_distribution<float> dis{0.0f, 1.0f};
    float subsample;
    size_t budget, kept = 0;
    Subsampler(size_t total, float s, long long seed) : subsample(s), budget(static_cast<size_t>(total * s)) {
        if (seed < 0) { std::random_device rd; gen.seed(rd()); } else gen.seed((uint32_t)seed);
    }
    bool keep() { return dis(gen) <= subsample; }
};

inline size_t load_cloud_txt(const std::string& path, float subsample, std::vector<icp::vec3>& cloud, long long seed) {
    std::ifstream f(path);
    if (!f.is_open()) throw std::runtime_error("Error reading TXT file: Unable to open TXT file: " + path);
    int total = 0;
    f >> total;
    if (total <= 0) throw std::runtime_error("Error reading TXT file: Invalid number of points in the TXT file: " + path);
    Subsampler ss((size_t)total, subsample, seed);
    cloud.reserve(ss.budget);
    for (int i = 0; i < total; ++i) {
        float x, y, z;
        if (!(f >> x >> y >> z)) throw std::runtime_error("Error reading TXT file: Error reading point data from TXT file: " + path);
        if (ss.keep() && ss.kept < ss.budget) {  // utilities.hpp:217 — the RNG is drawn for every point
            cloud.emplace_back(x, y, z);
            ++ss.kept;
        }
    }
    return ss.kept;
}

namespace ply {
struct Property { std::string name, type, count_type; bool is_list = false; };
struct Element { std::string name; size_t count = 0; std::vector<Property> props; };
inline size_t type_size(const std::string& t) {
    if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
    if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
    if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
    if (t == "double" || t == "float64") return 8;
    throw std::runtime_error("unsupported PLY property type '" + t + "'");
}
inline double read_scalar(std::istream& f, const std::string& t, bool swap) {
    unsigned char b[8];
    const size_t n = type_size(t);
    f.read(reinterpret_cast<char*>(b), (std::streamsize)n);
    if (!f) throw std::runtime_error("unexpected end of PLY data");
    if (swap) std::reverse(b, b + n);
    if (t == "char" || t == "int8") { int8_t v; std::memcpy(&v, b, 1); return v; }
    if (t == "uchar" || t == "uint8") { uint8_t v; std::memcpy(&v, b, 1); return v; }
    if (t == "short" || t == "int16") { int16_t v; std::memcpy(&v, b, 2); return v; }
    if (t == "ushort" || t == "uint16") { uint16_t v; std::memcpy(&v, b, 2); return v; }
    if (t == "int" || t == "int32") { int32_t v; std::memcpy(&v, b, 4); return v; }
    if (t == "uint" || t == "uint32") { uint32_t v; std::memcpy(&v, b, 4); return v; }
    if (t == "float" || t == "float32") { float v; std::memcpy(&v, b, 4); return v; }
    double v; std::memcpy(&v, b, 8); return v;
}
}  // namespace ply

inline size_t load_cloud_ply(const std::string& path, float subsample, std::vector<icp::vec3>& cloud, long long seed) {
    try {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("Unable to open file: " + path);
        std::string line;
        std::getline(f, line);
        if (trim(line) != "ply") throw std::runtime_error("not a PLY file");
        std::string format;
        std::vector<ply::Element> elements;
        while (std::getline(f, line)) {
            std::istringstream ls(trim(line));
            std::string tok;
            ls >> tok;
            if (tok == "format") ls >> format;
            else if (tok == "element") { ply::Element e; ls >> e.name >> e.count; elements.push_back(e); }
            else if (tok == "property") {
                if (elements.empty()) throw std::runtime_error("property before element");
                ply::Property p;
                std::string t;
                ls >> t;
                if (t == "list") { p.is_list = true; ls >> p.count_type >> p.type >> p.name; }
                else { p.type = t; ls >> p.name; }
                elements.back().props.push_back(p);
            } else if (tok == "end_header") break;
        }
        const bool ascii = format == "ascii";
        const bool big = format == "binary_big_endian";
        if (!ascii && !big && format != "binary_little_endian") throw std::runtime_error("unsupported PLY format '" + format + "'");
        const uint16_t probe = 1;
        const bool host_little = *reinterpret_cast<const uint8_t*>(&probe) == 1;
        const bool swap = !ascii && (big == host_little);
        for (const ply::Element& e : elements) {
            const bool is_vertex = e.name == "vertex";
            int ix = -1, iy = -1, iz = -1;
            for (size_t k = 0; k < e.props.size(); ++k) {
                if (e.props[k].name == "x") ix = (int)k;
                if (e.props[k].name == "y") iy = (int)k;
                if (e.props[k].name == "z") iz = (int)k;
            }
            if (is_vertex && (ix < 0 || iy < 0 || iz < 0)) throw std::runtime_error("PLY file missing 'x', 'y', or 'z' vertex properties.");
            if (is_vertex && e.count == 0) throw std::runtime_error("No vertices found in the PLY file.");
            Subsampler ss(e.count, subsample, seed);
            if (is_vertex) cloud.reserve(ss.budget);
            std::vector<double> vals(e.props.size());
            for (size_t i = 0; i < e.count; ++i) {
                if (is_vertex && ss.kept >= ss.budget) return ss.kept;  // utilities.hpp:154 — the PLY loop ends early
                for (size_t k = 0; k < e.props.size(); ++k) {
                    const ply::Property& p = e.props[k];
                    if (p.is_list) {
                        double n;
                        if (ascii) { if (!(f >> n)) throw std::runtime_error("unexpected end of PLY data"); }
                        else n = ply::read_scalar(f, p.count_type, swap);
                        for (long long j = 0; j < (long long)n; ++j) {
                            double dummy;
                            if (ascii) { if (!(f >> dummy)) throw std::runtime_error("unexpected end of PLY data"); }
                            else (void)ply::read_scalar(f, p.type, swap);
                        }
                    } else if (ascii) {
                        if (!(f >> vals[k])) throw std::runtime_error("unexpected end of PLY data");
                    } else {
                        vals[k] = ply::read_scalar(f, p.type, swap);
                    }
                }
                if (is_vertex && ss.keep()) {
                    cloud.emplace_back((float)vals[ix], (float)vals[iy], (float)vals[iz]);
                    ++ss.kept;
                }
            }
            if (is_vertex) return ss.kept;
        }
        throw std::runtime_error("No vertices found in the PLY file.");
    } catch (const std::exception& err) {
        throw std::runtime_error(std::string("Error reading PLY file: ") + err.what());
    }
}

inline size_t load_cloud(const std::string& filepath, float subsample, std::vector<icp::vec3>& cloud, long long seed = -1) {
    auto dot = filepath.find_last_of('.');
    if (dot == std::string::npos) throw std::runtime_error("Filepath does not have a valid extension: " + filepath);
    std::string ext = filepath.substr(dot + 1);
    std::transform(ext.begin(), ext.end(), ext.begin(), ::tolower);
    if (ext == "ply") return load_cloud_ply(filepath, subsample, cloud, seed);
    if (ext == "txt") return load_cloud_txt(filepath, subsample, cloud, seed);
    throw std::runtime_error("Unsupported file extension: " + ext);
}

// io.output / io.visualization (declared by test/bunny.toml:10-11, unimplemented upstream)
inline void write_result_toml(const std::string& path, const icp::mat3& R, const icp::vec3& t, float sse, size_t ns, double seconds,
                              const fgoicp_run_stats& st) {
    std::ofstream f(path);
    if (!f) throw std::runtime_error("Unable to write " + path);
    f.precision(9);
    f << "# Fast Go-ICP result (fgoicp_amd)\n[result]\n";
    f << "rotation = [\n";
    for (int r = 0; r < 3; ++r) f << "  [" << R[0][r] << ", " << R[1][r] << ", " << R[2][r] << "],\n";
    f << "]\ntranslation = [" << t.x << ", " << t.y << ", " << t.z << "]\n";
    f << "sse = " << sse << "\nmse = " << sse / (float)ns << "\nseconds = " << seconds << "\n";
    f << "\n[stats]\nsubcubes = " << st.trans_cubes << "\nrotation_cubes = " << st.rot_cubes << "\nicp_runs = " << st.icp_runs
      << "\nicp_iterations = " << st.icp_iters << "\nrounds = " << st.rounds << "\n";
}

inline void write_visualization_ply(const std::string& path, const std::vector<icp::vec3>& tgt, const std::vector<icp::vec3>& src,
                                    const icp::mat3& R, const icp::vec3& t) {
    std::ofstream f(path);
    if (!f) throw std::runtime_error("Unable to write " + path);
    f << "ply\nformat ascii 1.0\ncomment target = blue, registered source = red\nelement vertex " << tgt.size() + src.size()
      << "\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n";
    for (const auto& p : tgt) f << p.x << " " << p.y << " " << p.z << " 40 90 220\n";
    for (const auto& p : src) { icp::vec3 q = R * p + t; f << q.x << " " << q.y << " " << q.z << " 220 60 40\n"; }
}

}  // namespace cli
