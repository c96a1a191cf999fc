// Host-side 3-vector / 3x3 types of the product, layout-identical to glm::vec3 / glm::mat3
// (12 B / 36 B, column-major; reference fgoicp/common.hpp:12-13), plus the 3x3 SVD
// that stands in for Eigen::JacobiSVD<Matrix3d> (fgoicp/icp3d.cu:110-138).
// Host arithmetic is plain mul/add, left to right (this library is built with -ffp-contract=off),
// as the reference's host translation units evaluate it.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>
#include <utility>

// The 3x3 types and the SVD below are also compiled for the device (csrc/device/kernels.hip: the ICP loop's step kernel runs
// the same source on one GPU thread, so the host loop and the device-resident loop return the same bits); g++-only
// translation units (tests/host_harness) see plain functions.
#if defined(__HIPCC__)
#define FGOICP_HD __host__ __device__
#else
#define FGOICP_HD
#endif

namespace fgoicp {

struct Vec3f {
    float x, y, z;
};
FGOICP_HD inline Vec3f operator+(Vec3f a, Vec3f b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
FGOICP_HD inline Vec3f operator-(Vec3f a, Vec3f b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
FGOICP_HD inline Vec3f operator*(Vec3f a, float s) { return {a.x * s, a.y * s, a.z * s}; }
FGOICP_HD inline Vec3f operator/(Vec3f a, float s) { return {a.x / s, a.y / s, a.z / s}; }
FGOICP_HD inline Vec3f operator-(Vec3f a) { return {-a.x, -a.y, -a.z}; }

struct Mat3f {
    float m[9];  // m[col*3 + row]
    FGOICP_HD float& at(int col, int row) { return m[col * 3 + row]; }
    FGOICP_HD float at(int col, int row) const { return m[col * 3 + row]; }
    FGOICP_HD static Mat3f identity() {
        Mat3f r{};
        r.m[0] = r.m[4] = r.m[8] = 1.0f;
        return r;
    }
    FGOICP_HD static Mat3f from(const float* p) {
        Mat3f r;
        for (int i = 0; i < 9; ++i) r.m[i] = p[i];
        return r;
    }
};

// glm: m[0]*v.x + m[1]*v.y + m[2]*v.z
FGOICP_HD inline Vec3f operator*(const Mat3f& a, Vec3f v) {
    return {a.m[0] * v.x + a.m[3] * v.y + a.m[6] * v.z, a.m[1] * v.x + a.m[4] * v.y + a.m[7] * v.z,
            a.m[2] * v.x + a.m[5] * v.y + a.m[8] * v.z};
}
// glm: Result[j][i] = A[0][i]*B[j][0] + A[1][i]*B[j][1] + A[2][i]*B[j][2]
FGOICP_HD inline Mat3f operator*(const Mat3f& a, const Mat3f& b) {
    Mat3f r;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) r.m[j * 3 + i] = a.m[0 + i] * b.m[j * 3 + 0] + a.m[3 + i] * b.m[j * 3 + 1] + a.m[6 + i] * b.m[j * 3 + 2];
    return r;
}

// ---- 3x3 SVD in double: Eigen::JacobiSVD<Matrix3d>'s algorithm ----------------------------------------
// The reference takes its Procrustes rotation from Eigen::JacobiSVD<Eigen::Matrix3d>(H, ComputeFullU | ComputeFullV)
// (fgoicp/icp3d.cu:118-121; Eigen3 >= 3.3, fgoicp/CMakeLists.txt:19).  On a full-rank H any SVD yields the same
// R = V diag(1, 1, det(V U^T)) U^T; on a rank-deficient H (all correspondences on one or two target points, collinear
// or coplanar clouds) the null-space columns of U and V are the algorithm's choice, so R is too.  Round 2 used a
// one-sided Hestenes Jacobi here and differed from the checker on such inputs (fuzz seed 7, cases 103 / 505); this is
// Eigen 3.3 / 3.4's two-sided Jacobi as published (JacobiSVD::compute, real_2x2_jacobi_svd, JacobiRotation::makeJacobi),
// written out for n = 3:
//   W = H / max|H_ij|; pairs in the order (1,0), (2,0), (2,1); a pair is worked on while |W(p,q)| or |W(q,p)| exceeds
//   max(DBL_MIN, 2 eps maxdiag) with maxdiag the running maximum of the |diagonal|; per pair a left rotation that makes the
//   2x2 block symmetric, then the symmetric Jacobi rotation; sigma_i = |W(i,i)| * scale with U's column negated for a
//   negative diagonal; descending order by swapping with the first maximum of the tail, stopping at a zero.
// A plane rotation (c, s) is the matrix [[c, s], [-s, c]]; `turn(x, y, c, s)` is (x, y) <- (c x + s y, -s x + c y) and
// leaves the operands alone for the identity (as Eigen's apply_rotation_in_the_plane does).
// A = U diag(S) V^T, row-major double[3][3] in and out.
namespace detail {
FGOICP_HD inline void turn(double& x, double& y, double c, double s) {
    if (c == 1.0 && s == 0.0) return;
    const double x0 = x, y0 = y;
    x = c * x0 + s * y0;
    y = -s * x0 + c * y0;
}
}  // namespace detail

FGOICP_HD inline void svd3_jacobi(const double Ain[3][3], double U[3][3], double S[3], double V[3][3]) {
    const double tiny = DBL_MIN;
    const double two_eps = 2.0 * DBL_EPSILON;
    double scale = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) scale = fabs(Ain[i][j]) > scale ? fabs(Ain[i][j]) : scale;
    if (scale == 0.0) scale = 1.0;
    double W[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            W[i][j] = Ain[i][j] / scale;
            U[i][j] = V[i][j] = (i == j) ? 1.0 : 0.0;
        }
    double maxdiag = fabs(W[0][0]);
    if (fabs(W[1][1]) > maxdiag) maxdiag = fabs(W[1][1]);
    if (fabs(W[2][2]) > maxdiag) maxdiag = fabs(W[2][2]);
    for (int sweep = 0; sweep < 1000; ++sweep) {  // Eigen sweeps until nothing rotates (3-6 sweeps); the cap only bounds a defect
        bool rotated = false;
        for (int pair = 0; pair < 3; ++pair) {
            const int p = pair == 0 ? 1 : 2, q = pair == 2 ? 1 : 0;  // (1,0), (2,0), (2,1)
            const double limit = two_eps * maxdiag > tiny ? two_eps * maxdiag : tiny;
            if (!(fabs(W[p][q]) > limit || fabs(W[q][p]) > limit)) continue;
            rotated = true;
            // the 2x2 block, p first
            double m00 = W[p][p], m01 = W[p][q], m10 = W[q][p], m11 = W[q][q];
            // left rotation (c1, s1) that makes it symmetric
            double c1 = 1.0, s1 = 0.0;
            const double tsum = m00 + m11, diff = m10 - m01;
            if (!(fabs(diff) < tiny)) {
                const double u = tsum / diff;
                const double h = sqrt(1.0 + u * u);
                s1 = 1.0 / h;
                c1 = u / h;
            }
            detail::turn(m00, m10, c1, s1);
            detail::turn(m01, m11, c1, s1);
            // symmetric Jacobi rotation (cr, sr) of [[m00, m01], [m01, m11]]
            double cr = 1.0, sr = 0.0;
            const double deno = 2.0 * fabs(m01);
            if (!(deno < tiny)) {
                const double tau = (m00 - m11) / deno;
                const double w = sqrt(tau * tau + 1.0);
                const double t = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
                const double sign_t = t > 0.0 ? 1.0 : -1.0;
                const double n = 1.0 / sqrt(t * t + 1.0);
                sr = -sign_t * (m01 / fabs(m01)) * fabs(t) * n;
                cr = n;
            }
            // left = rot1 * right^T
            const double cl = c1 * cr - s1 * (-sr);
            const double sl = c1 * (-sr) + s1 * cr;
            for (int k = 0; k < 3; ++k) detail::turn(W[p][k], W[q][k], cl, sl);   // W <- L W      (rows p, q)
            for (int k = 0; k < 3; ++k) detail::turn(U[k][p], U[k][q], cl, sl);   // U <- U L^T    (columns p, q)
            for (int k = 0; k < 3; ++k) detail::turn(W[k][p], W[k][q], cr, -sr);  // W <- W R      (columns p, q)
            for (int k = 0; k < 3; ++k) detail::turn(V[k][p], V[k][q], cr, -sr);  // V <- V R
            const double dp = fabs(W[p][p]), dq = fabs(W[q][q]);
            if (dp > maxdiag) maxdiag = dp;
            if (dq > maxdiag) maxdiag = dq;
        }
        if (!rotated) break;
    }
    for (int i = 0; i < 3; ++i) {
        const double d = W[i][i];
        S[i] = fabs(d);
        if (d < 0.0)
            for (int r = 0; r < 3; ++r) U[r][i] = -U[r][i];
        S[i] *= scale;
    }
    for (int i = 0; i < 3; ++i) {
        int pos = i;
        for (int j = i + 1; j < 3; ++j)
            if (S[j] > S[pos]) pos = j;
        if (S[pos] == 0.0) break;
        if (pos == i) continue;
        double sw = S[i]; S[i] = S[pos]; S[pos] = sw;
        for (int r = 0; r < 3; ++r) {
            sw = U[r][i]; U[r][i] = U[r][pos]; U[r][pos] = sw;
            sw = V[r][i]; V[r][i] = V[r][pos]; V[r][pos] = sw;
        }
    }
}

// closest_orthogonal_approximation, fgoicp/icp3d.cu:110-138: H(r,c) = ABt[c][r]; H = U S V^T;
// R = V diag(1, 1, det(V U^T)) U^T in double, cast to fp32, returned in glm order.  Products and the determinant are
// evaluated in Eigen's order for fixed 3x3 operands: a row-by-column product as x0 + (x1 + x2), the determinant as
// m00 (m11 m22 - m12 m21) - m01 (m10 m22 - m12 m20) + m02 (m10 m21 - m11 m20).
FGOICP_HD inline Mat3f closest_orthogonal_approximation(const Mat3f& ABt) {
    double H[3][3], U[3][3], V[3][3], S[3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) H[r][c] = (double)ABt.at(c, r);
    svd3_jacobi(H, U, S, V);
    double VUt[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) VUt[i][j] = V[i][0] * U[j][0] + (V[i][1] * U[j][1] + V[i][2] * U[j][2]);
    const double det = VUt[0][0] * (VUt[1][1] * VUt[2][2] - VUt[1][2] * VUt[2][1]) - VUt[0][1] * (VUt[1][0] * VUt[2][2] - VUt[1][2] * VUt[2][0]) +
                       VUt[0][2] * (VUt[1][0] * VUt[2][1] - VUt[1][1] * VUt[2][0]);
    // D = diag(1, 1, det); R = (V D) U^T as two full 3x3 products, as icp3d.cu:129-133 forms them (the products with D's
    // zeros and ones are kept: they decide the sign of a zero entry)
    const double D[3][3] = {{1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, 0.0, det}};
    double VD[3][3];
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) VD[i][k] = V[i][0] * D[0][k] + (V[i][1] * D[1][k] + V[i][2] * D[2][k]);
    Mat3f out;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double rij = VD[i][0] * U[j][0] + (VD[i][1] * U[j][1] + VD[i][2] * U[j][2]);
            out.at(j, i) = (float)rij;  // out[col j][row i] = R(i, j)
        }
    return out;
}

}  // namespace fgoicp
