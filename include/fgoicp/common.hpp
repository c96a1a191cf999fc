// icp:: common types for C++ callers of libfgoicp_amd.so — the names and semantics of the reference's
// fgoicp/common.hpp:15-269, without CUDA or GLM.  Header-only; everything computes on the host except
// what forwards to the C ABI (include/fgoicp_amd.h).
#pragma once
#include <chrono>
#include <cmath>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../fgoicp_amd.h"

#ifndef M_INF
#define M_INF 1E+10f            // common.hpp:18
#endif
#define M_SQRT3 1.732050807568877f  // common.hpp:19

namespace icp {

// Layout-identical stand-ins for glm::vec3 (12 B) and glm::mat3 (36 B, column-major, m[col][row]).
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }

struct mat3 {
    vec3 col[3];
    mat3() : mat3(1.0f) {}
    explicit mat3(float d) { col[0] = vec3(d, 0, 0); col[1] = vec3(0, d, 0); col[2] = vec3(0, 0, d); }
    // nine scalars fill COLUMNS, as glm::mat3 does (common.hpp:50-54)
    mat3(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2) {
        col[0] = vec3(a0, a1, a2); col[1] = vec3(b0, b1, b2); col[2] = vec3(c0, c1, c2);
    }
    vec3& operator[](int c) { return col[c]; }
    const vec3& operator[](int c) const { return col[c]; }
    const float* data() const { return &col[0].x; }
    float* data() { return &col[0].x; }
};
inline vec3 operator*(const mat3& m, vec3 v) {
    return {m[0].x * v.x + m[1].x * v.y + m[2].x * v.z, m[0].y * v.x + m[1].y * v.y + m[2].y * v.z, m[0].z * v.x + m[1].z * v.y + m[2].z * v.z};
}
static_assert(sizeof(vec3) == 12 && sizeof(mat3) == 36, "glm layout");

typedef vec3 Point3D;                       // common.hpp:130
using PointCloud = std::vector<Point3D>;    // common.hpp:132

// common.hpp:30-69
struct Rotation {
    float x, y, z, r;
    mat3 R;
    Rotation() : Rotation(0.0f, 0.0f, 0.0f) {}
    Rotation(float x_, float y_, float z_) : x(x_), y(y_), z(z_), r(x_ * x_ + y_ * y_ + z_ * z_), R(1.0f) {
        if (r > 1.0f) return;  // not a rotation
        float ww = 1.0f - r, w = std::sqrt(ww);
        float wx = w * x, xx = x * x, wy = w * y, xy = x * y, yy = y * y, wz = w * z, xz = x * z, yz = y * z, zz = z * z;
        R = mat3(ww + xx - yy - zz, 2 * (xy - wz), 2 * (xz + wy), 2 * (xy + wz), ww - xx + yy - zz, 2 * (yz - wx), 2 * (xz - wy),
                 2 * (yz + wx), ww - xx - yy + zz);
        r = std::sqrt(r);
    }
    bool in_SO3() const { return r <= 1.0f; }
};

// common.hpp:75-104
struct RotNode {
    Rotation q;
    float span, lb, ub;
    RotNode(float x, float y, float z, float span_, float lb_, float ub_) : q(x, y, z), span(span_), lb(lb_), ub(ub_) {}
    friend bool operator<(const RotNode& a, const RotNode& b) { return a.lb == b.lb ? a.span < b.span : a.lb > b.lb; }
    bool overlaps_SO3() const { return q.r - 2 * span * (std::fabs(q.x) + std::fabs(q.y) + std::fabs(q.z)) + 3 * span * span <= 1; }
};

// common.hpp:110-128
struct TransNode {
    vec3 t;
    float span, lb, ub;
    TransNode(float x, float y, float z, float span_, float lb_, float ub_) : t(x, y, z), span(span_), lb(lb_), ub(ub_) {}
    friend bool operator<(const TransNode& a, const TransNode& b) { return a.lb == b.lb ? a.span < b.span : a.lb > b.lb; }
};

// common.hpp:138-164 — kept for signature compatibility.  The HIP context owns its stream and a
// batch is one fused launch, so the pool carries no streams.
class StreamPool {
public:
    explicit StreamPool(size_t size) : size_(size) {}
    size_t size() const { return size_; }
private:
    size_t size_;
};

// common.hpp:170-269
enum class LogLevel { Debug, Info, Warning, Error };
class Logger {
public:
    explicit Logger(LogLevel level) : level_(level) {}
    Logger() : Logger(LogLevel::Debug) {}
    template <typename T> Logger& operator<<(const T& msg) { buffer_ << msg; return *this; }
    Logger& operator<<(const vec3& v) {
        buffer_ << std::fixed << std::setprecision(6) << v.x << "\t" << v.y << "\t" << v.z;
        return *this;
    }
    Logger& operator<<(const mat3& m) {  // prints the mathematical matrix row by row (common.hpp:202-209)
        buffer_ << std::fixed << std::setprecision(4);
        buffer_ << "\t" << m[0][0] << "\t" << m[1][0] << "\t" << m[2][0] << "\n";
        buffer_ << "\t" << m[0][1] << "\t" << m[1][1] << "\t" << m[2][1] << "\n";
        buffer_ << "\t" << m[0][2] << "\t" << m[1][2] << "\t" << m[2][2];
        return *this;
    }
    ~Logger() {
        if (level_ == LogLevel::Debug && !verbose()) return;
        const char* color = "\033[34m"; const char* name = "Debug";
        if (level_ == LogLevel::Info) { color = "\033[32m"; name = "Info"; }
        if (level_ == LogLevel::Warning) { color = "\033[33m"; name = "Warning"; }
        if (level_ == LogLevel::Error) { color = "\033[31m"; name = "Error"; }
        auto now = std::chrono::system_clock::to_time_t(std::chrono::system_clock::now());
        std::tm buf{};
        localtime_r(&now, &buf);
        std::cout << color << "[" << name << " " << std::put_time(&buf, "%H:%M:%S") << "] " << buffer_.str() << "\033[0m" << "\n";
    }
    static void set_verbose(bool v) { verbose() = v; }
private:
    static bool& verbose() { static bool v = false; return v; }
    LogLevel level_;
    std::ostringstream buffer_;
};

// Every C-ABI failure surfaces as the exception type the reference's CLI path already uses
// (std::runtime_error, src/utilities.hpp:122) after an Error log line.
inline void check_status(int status, const char* where) {
    if (status == FGOICP_OK) return;
    std::string msg = std::string(where) + " failed (status " + std::to_string(status) + "): " + fgoicp_last_error();
    Logger(LogLevel::Error) << msg;
    throw std::runtime_error(msg);
}

}  // namespace icp
