// Host-side 3-vector / 3x3 types of the product, layout-identical to glm::vec3 / glm::mat3
// (12 B / 36 B, column-major; reference fgoicp/common.hpp:12-13), plus the 3x3 SVD
// that stands in for Eigen::JacobiSVD<Matrix3d> (fgoicp/icp3d.cu:110-138).
// Host arithmetic is plain mul/add, left to right (this library is built with -ffp-contract=off),
// as the reference's host translation units evaluate it.
#pragma once
#include <cmath>
#include <cstring>
#include <limits>
#include <utility>

namespace fgoicp {

struct Vec3f {
    float x, y, z;
};
inline Vec3f operator+(Vec3f a, Vec3f b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3f operator-(Vec3f a, Vec3f b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3f operator*(Vec3f a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3f operator/(Vec3f a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline Vec3f operator-(Vec3f a) { return {-a.x, -a.y, -a.z}; }

struct Mat3f {
    float m[9];  // m[col*3 + row]
    float& at(int col, int row) { return m[col * 3 + row]; }
    float at(int col, int row) const { return m[col * 3 + row]; }
    static Mat3f identity() {
        Mat3f r{};
        r.m[0] = r.m[4] = r.m[8] = 1.0f;
        return r;
    }
    static Mat3f from(const float* p) {
        Mat3f r;
        std::memcpy(r.m, p, sizeof(r.m));
        return r;
    }
};

// glm: m[0]*v.x + m[1]*v.y + m[2]*v.z
inline Vec3f operator*(const Mat3f& a, Vec3f v) {
    return {a.m[0] * v.x + a.m[3] * v.y + a.m[6] * v.z, a.m[1] * v.x + a.m[4] * v.y + a.m[7] * v.z,
            a.m[2] * v.x + a.m[5] * v.y + a.m[8] * v.z};
}
// glm: Result[j][i] = A[0][i]*B[j][0] + A[1][i]*B[j][1] + A[2][i]*B[j][2]
inline Mat3f operator*(const Mat3f& a, const Mat3f& b) {
    Mat3f r;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) r.m[j * 3 + i] = a.m[0 + i] * b.m[j * 3 + 0] + a.m[3 + i] * b.m[j * 3 + 1] + a.m[6 + i] * b.m[j * 3 + 2];
    return r;
}

// ---- 3x3 SVD, one-sided (Hestenes) Jacobi in double --------------------------------------
// A = U diag(S) V^T with S sorted descending; row-major double[3][3] in and out.
inline void svd3_hestenes(const double Ain[3][3], double U[3][3], double S[3], double V[3][3]) {
    double A[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            A[i][j] = Ain[i][j];
            V[i][j] = (i == j) ? 1.0 : 0.0;
        }
    const double eps = std::numeric_limits<double>::epsilon();
    for (int sweep = 0; sweep < 80; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int r = 0; r < 3; ++r) {
                    alpha += A[r][p] * A[r][p];
                    beta += A[r][q] * A[r][q];
                    gamma += A[r][p] * A[r][q];
                }
                if (gamma == 0.0 || std::fabs(gamma) <= eps * std::sqrt(alpha * beta)) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < 3; ++r) {
                    const double ap = A[r][p], aq = A[r][q];
                    A[r][p] = c * ap - s * aq;
                    A[r][q] = s * ap + c * aq;
                    const double vp = V[r][p], vq = V[r][q];
                    V[r][p] = c * vp - s * vq;
                    V[r][q] = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    int order[3] = {0, 1, 2};
    double norm[3];
    for (int j = 0; j < 3; ++j) norm[j] = std::sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (norm[order[j]] > norm[order[i]]) std::swap(order[i], order[j]);
    double Vs[3][3];
    for (int k = 0; k < 3; ++k) {
        const int j = order[k];
        S[k] = norm[j];
        for (int r = 0; r < 3; ++r) {
            Vs[r][k] = V[r][j];
            U[r][k] = norm[j] > 0 ? A[r][j] / norm[j] : 0.0;
        }
    }
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) V[r][k] = Vs[r][k];
    // complete U for (numerically) vanishing singular values so that it stays orthonormal
    const double tiny = S[0] * 1e-14;
    auto col = [&](int k, double out[3]) { for (int r = 0; r < 3; ++r) out[r] = U[r][k]; };
    auto set = [&](int k, const double in[3]) {
        double n = std::sqrt(in[0] * in[0] + in[1] * in[1] + in[2] * in[2]);
        for (int r = 0; r < 3; ++r) U[r][k] = in[r] / n;
    };
    if (S[0] <= 0) {
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) U[i][j] = (i == j) ? 1.0 : 0.0;
        return;
    }
    if (S[1] <= tiny) {  // rank 1: any unit vector orthogonal to u0
        double u0[3]; col(0, u0);
        int m = std::fabs(u0[0]) < std::fabs(u0[1]) ? (std::fabs(u0[0]) < std::fabs(u0[2]) ? 0 : 2) : (std::fabs(u0[1]) < std::fabs(u0[2]) ? 1 : 2);
        double e[3] = {0, 0, 0}; e[m] = 1.0;
        double u1[3] = {u0[1] * e[2] - u0[2] * e[1], u0[2] * e[0] - u0[0] * e[2], u0[0] * e[1] - u0[1] * e[0]};
        set(1, u1);
    }
    if (S[2] <= tiny) {
        double u0[3], u1[3]; col(0, u0); col(1, u1);
        double u2[3] = {u0[1] * u1[2] - u0[2] * u1[1], u0[2] * u1[0] - u0[0] * u1[2], u0[0] * u1[1] - u0[1] * u1[0]};
        set(2, u2);
    }
}

// closest_orthogonal_approximation, fgoicp/icp3d.cu:110-138: H(r,c) = ABt[c][r]; H = U S V^T;
// R = V diag(1, 1, det(V U^T)) U^T in double, cast to fp32, returned in glm order.
inline Mat3f closest_orthogonal_approximation(const Mat3f& ABt) {
    double H[3][3], U[3][3], V[3][3], S[3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) H[r][c] = (double)ABt.at(c, r);
    svd3_hestenes(H, U, S, V);
    double VUt[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) VUt[i][j] = V[i][0] * U[j][0] + V[i][1] * U[j][1] + V[i][2] * U[j][2];
    const double det = VUt[0][0] * (VUt[1][1] * VUt[2][2] - VUt[1][2] * VUt[2][1]) - VUt[0][1] * (VUt[1][0] * VUt[2][2] - VUt[1][2] * VUt[2][0]) +
                       VUt[0][2] * (VUt[1][0] * VUt[2][1] - VUt[1][1] * VUt[2][0]);
    Mat3f out;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double rij = V[i][0] * U[j][0] + V[i][1] * U[j][1] + det * V[i][2] * U[j][2];
            out.at(j, i) = (float)rij;  // out[col j][row i] = R(i, j)
        }
    return out;
}

}  // namespace fgoicp
