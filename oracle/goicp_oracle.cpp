// CPU restatement of the reference hot path — TEST INFRASTRUCTURE ONLY (see goicp_oracle.hpp).
// PARITY UNPINNED (no reference outputs exist; see header).
// Build with -ffp-contract=off: every fused multiply-add below is written explicitly.
#include "goicp_oracle.hpp"

#include <algorithm>
#include <cfloat>
#include <cstring>
#include <limits>
#include <queue>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace goicp_oracle {

// ---------------------------------------------------------------------------------------------
// small math (GLM semantics: column-major mat3)
// ---------------------------------------------------------------------------------------------
Conventions& conventions() {
    static Conventions c;
    return c;
}

// a*x + b*y + c*z under the three contraction choices of Conventions
static inline float dot3_conv(int mode, float a, float x, float b, float y, float c, float z) {
    if (mode == 1) return std::fmaf(c, z, std::fmaf(b, y, a * x));
    if (mode == 2) return std::fmaf(c, z, std::fmaf(a, x, b * y));
    return a * x + b * y + c * z;
}

Mat3 mat3_identity() {
    Mat3 m;
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) m.c[c][r] = (c == r) ? 1.0f : 0.0f;
    return m;
}

// glm mat3*vec3: m[0][r]*v.x + m[1][r]*v.y + m[2][r]*v.z, device build → fma chain.
Vec3 dev_mul(const Mat3& m, const Vec3& v) {
    Vec3 o;
    const int f = conventions().fma_matvec;
    o.x = dot3_conv(f, m.c[0][0], v.x, m.c[1][0], v.y, m.c[2][0], v.z);
    o.y = dot3_conv(f, m.c[0][1], v.x, m.c[1][1], v.y, m.c[2][1], v.z);
    o.z = dot3_conv(f, m.c[0][2], v.x, m.c[1][2], v.y, m.c[2][2], v.z);
    return o;
}

Vec3 host_mul(const Mat3& m, const Vec3& v) {
    Vec3 o;
    o.x = m.c[0][0] * v.x + m.c[1][0] * v.y + m.c[2][0] * v.z;
    o.y = m.c[0][1] * v.x + m.c[1][1] * v.y + m.c[2][1] * v.z;
    o.z = m.c[0][2] * v.x + m.c[1][2] * v.y + m.c[2][2] * v.z;
    return o;
}

// glm mat3*mat3: Result[j][i] = A[0][i]*B[j][0] + A[1][i]*B[j][1] + A[2][i]*B[j][2]
Mat3 host_mul(const Mat3& a, const Mat3& b) {
    Mat3 o;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i)
            o.c[j][i] = a.c[0][i] * b.c[j][0] + a.c[1][i] * b.c[j][1] + a.c[2][i] * b.c[j][2];
    return o;
}

// device-side squared distance, registration.cu:154-160 / :250-256 (fma convention)
static inline float dev_dist_sq(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    return dot3_conv(conventions().fma_dist, dx, dx, dy, dy, dz, dz);
}
// registration.cu:39-41: x*x + y*y + z*z of the source point (the "radius")
static inline float dev_norm_sq(const Vec3& p) { return dot3_conv(conventions().fma_dist, p.x, p.x, p.y, p.y, p.z, p.z); }
// device sin(float) of registration.cu:43, optionally moved by whole ulps (Conventions::sin_ulps)
static inline float dev_sin(float a) {
    float v = std::sin(a);
    int k = conventions().sin_ulps;
    for (; k > 0; --k) v = std::nextafterf(v, 2.0f);
    for (; k < 0; ++k) v = std::nextafterf(v, -2.0f);
    return v;
}
// Thrust reduce(float, plus) under Conventions::sum_mode 1 / 2 (mode 0 keeps the chunked fp64 sums of the callers)
static float reduce_f32(const float* v, long n, int mode) {
    if (n <= 0) return 0.0f;
    if (mode == 2) { float s = 0.0f; for (long i = 0; i < n; ++i) s += v[i]; return s; }
    if (n <= 8) { float s = v[0]; for (long i = 1; i < n; ++i) s += v[i]; return s; }
    const long h = n / 2;
    return reduce_f32(v, h, mode) + reduce_f32(v + h, n - h, mode);
}

// ---------------------------------------------------------------------------------------------
// Rotation / RotNode — fgoicp/common.hpp:30-104 (host code: plain arithmetic)
// ---------------------------------------------------------------------------------------------
Rotation::Rotation(float x_, float y_, float z_) : x(x_), y(y_), z(z_), r(x_ * x_ + y_ * y_ + z_ * z_), R(mat3_identity()) {
    if (r > 1.0f) return;  // common.hpp:42 — r keeps the *squared* norm, R stays identity
    float ww = 1.0f - r;
    float w = std::sqrt(ww);
    float wx = w * x, xx = x * x;
    float wy = w * y, xy = x * y, yy = y * y;
    float wz = w * z, xz = x * z, yz = y * z, zz = z * z;
    // glm::mat3(9 scalars) fills COLUMNS (common.hpp:50-54)
    R.c[0][0] = ww + xx - yy - zz; R.c[0][1] = 2 * (xy - wz);     R.c[0][2] = 2 * (xz + wy);
    R.c[1][0] = 2 * (xy + wz);     R.c[1][1] = ww - xx + yy - zz; R.c[1][2] = 2 * (yz - wx);
    R.c[2][0] = 2 * (xz - wy);     R.c[2][1] = 2 * (yz + wx);     R.c[2][2] = ww - xx - yy + zz;
    r = std::sqrt(r);
}

bool RotNode::overlaps_SO3() const {  // common.hpp:99-103
    return q.r - 2 * span * (std::fabs(q.x) + std::fabs(q.y) + std::fabs(q.z)) + 3 * span * span <= 1;
}

// ---------------------------------------------------------------------------------------------
// NearestNeighborLUT — registration.cu:180-207 (ctor), :258-318 (build), :320-328 (search)
// ---------------------------------------------------------------------------------------------
NearestNeighborLUT::NearestNeighborLUT(float res, const Bounds& b, const PointCloud& pc, bool build_now) : resolution(res) {
    dims[0] = (int)std::ceil((b[0].second - b[0].first) / resolution);  // :186-188
    dims[1] = (int)std::ceil((b[1].second - b[1].first) / resolution);
    dims[2] = (int)std::ceil((b[2].second - b[2].first) / resolution);
    scale = 1.0f / resolution;  // :201
    offset = Vec3{-b[0].first, -b[1].first, -b[2].first};  // :202-204
    if (build_now) build(pc);
}

void NearestNeighborLUT::build(const PointCloud& pc) {
    const int np = (int)pc.size();
    std::vector<Vec3> pts(np);
    for (int i = 0; i < np; ++i)  // :289-296 (host)
        pts[i] = Vec3{pc[i].x + offset.x, pc[i].y + offset.y, pc[i].z + offset.z};
    data.assign(size(), 0.f);
    const int dx = dims[0], dy = dims[1], dz = dims[2];
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
    for (int z = 0; z < dz; ++z)
        for (int y = 0; y < dy; ++y)
            for (int x = 0; x < dx; ++x) {  // buildLUTKernel :258-278
                float cx = x * resolution, cy = y * resolution, cz = z * resolution;
                float minDist = FLT_MAX;
                for (int i = 0; i < np; ++i) {
                    float d = dev_dist_sq(cx, cy, cz, pts[i].x, pts[i].y, pts[i].z);
                    minDist = minDist < d ? minDist : d;
                }
                data[((size_t)z * dy + y) * dx + x] = minDist;
            }
}

// CUDA linear filtering, unnormalised coordinates, clamp addressing (CUDA C
// Programming Guide "Texture Fetching / Linear Filtering"): xB = x - 0.5, i = floor(xB),
// alpha = frac(xB) stored in 9-bit fixed point with 8 fractional bits.
static inline void tex_axis(float u, int dim, bool quant, int& i0, int& i1, float& w) {
    float ub = u - 0.5f;
    float fl = std::floor(ub);
    w = ub - fl;
    const int wmode = conventions().tex_weight;
    if (quant && wmode == 0) w = std::floor(w * 256.0f + 0.5f) * (1.0f / 256.0f);
    else if (quant && wmode == 1) w = std::floor(w * 256.0f) * (1.0f / 256.0f);
    // clamp in float first so that huge |u| cannot overflow the int conversion
    float lo = fl < -1.0f ? -1.0f : (fl > (float)dim ? (float)dim : fl);
    int i = (int)lo;
    i0 = std::min(std::max(i, 0), dim - 1);
    i1 = std::min(std::max(i + 1, 0), dim - 1);
}

float NearestNeighborLUT::search(const Vec3& q) const {
    float x = (q.x + offset.x) * scale;  // :323-325
    float y = (q.y + offset.y) * scale;
    float z = (q.z + offset.z) * scale;
    int x0, x1, y0, y1, z0, z1;
    float a, b, c;
    tex_axis(x, dims[0], quantize_weights, x0, x1, a);
    tex_axis(y, dims[1], quantize_weights, y0, y1, b);
    tex_axis(z, dims[2], quantize_weights, z0, z1, c);
    const size_t dx = dims[0], dy = dims[1];
    auto T = [&](int xi, int yi, int zi) { return data[((size_t)zi * dy + yi) * dx + xi]; };
    if (conventions().tex_blend == 1) {
        // CUDA C Programming Guide, "Linear Filtering", 3-D: (1-a)(1-b)(1-c) T[i,j,k] + a(1-b)(1-c) T[i+1,j,k] + ... + abc T[i+1,j+1,k+1]
        const float na = 1.0f - a, nb = 1.0f - b, nc = 1.0f - c;
        float r = na * nb * nc * T(x0, y0, z0);
        r += a * nb * nc * T(x1, y0, z0);
        r += na * b * nc * T(x0, y1, z0);
        r += a * b * nc * T(x1, y1, z0);
        r += na * nb * c * T(x0, y0, z1);
        r += a * nb * c * T(x1, y0, z1);
        r += na * b * c * T(x0, y1, z1);
        r += a * b * c * T(x1, y1, z1);
        return r;
    }
    auto lerp = [](float p, float q_, float w) { return std::fmaf(w, q_ - p, p); };
    float c00 = lerp(T(x0, y0, z0), T(x1, y0, z0), a);
    float c10 = lerp(T(x0, y1, z0), T(x1, y1, z0), a);
    float c01 = lerp(T(x0, y0, z1), T(x1, y0, z1), a);
    float c11 = lerp(T(x0, y1, z1), T(x1, y1, z1), a);
    float c0 = lerp(c00, c10, b);
    float c1 = lerp(c01, c11, b);
    return lerp(c0, c1, c);
}

// ---------------------------------------------------------------------------------------------
// Registration — registration.hpp:68-98, registration.cu:14-174
// ---------------------------------------------------------------------------------------------
Registration::Registration(const PointCloud& pct_, const PointCloud& pcs_, const Bounds& bounds, float lut_res, bool build_lut)
    : pct(pct_), pcs(pcs_), nnlut(lut_res, bounds, pct_, build_lut) {}

float brute_force_find_nearest_neighbor(const Vec3& q, const PointCloud& pct) {  // :162-174
    float best = kInf;
    const size_t nt = pct.size();
    for (size_t i = 0; i < nt; ++i) {
        float d = dev_dist_sq(q.x, q.y, q.z, pct[i].x, pct[i].y, pct[i].z);
        if (d < best) best = d;
    }
    return best;
}

// ---------------------------------------------------------------------------------------------
// GridNN — CPU baseline only (see header).  Cells of side h hold the target indices (counting sort, ascending inside a cell);
// a query walks Chebyshev shells around its (clamped) cell until everything outside the walked block is provably farther.
// ---------------------------------------------------------------------------------------------
void GridNN::build(const PointCloud& cloud) {
    pc = &cloud;
    const size_t nt = cloud.size();
    lo = hi = cloud[0];
    for (const Vec3& p : cloud) {
        lo.x = std::min(lo.x, p.x); lo.y = std::min(lo.y, p.y); lo.z = std::min(lo.z, p.z);
        hi.x = std::max(hi.x, p.x); hi.y = std::max(hi.y, p.y); hi.z = std::max(hi.z, p.z);
    }
    const double ex = std::max(1e-9, (double)hi.x - lo.x), ey = std::max(1e-9, (double)hi.y - lo.y), ez = std::max(1e-9, (double)hi.z - lo.z);
    // surfaces: ~2 points per occupied cell when the cell side is ~sqrt(2 * area / nt); the box area stands in for the surface's
    const double area = 2.0 * (ex * ey + ey * ez + ex * ez);
    h = (float)std::max(std::sqrt(2.0 * area / (double)nt), std::max(ex, std::max(ey, ez)) / 512.0);
    n[0] = std::max(1, (int)std::ceil(ex / h)); n[1] = std::max(1, (int)std::ceil(ey / h)); n[2] = std::max(1, (int)std::ceil(ez / h));
    auto cell_of = [&](const Vec3& p, int a) {
        const float v = a == 0 ? (p.x - lo.x) : a == 1 ? (p.y - lo.y) : (p.z - lo.z);
        return std::min(n[a] - 1, std::max(0, (int)std::floor(v / h)));
    };
    const size_t ncell = (size_t)n[0] * n[1] * n[2];
    start.assign(ncell + 1, 0);
    std::vector<int> cid(nt);
    for (size_t i = 0; i < nt; ++i) {
        cid[i] = (cell_of(cloud[i], 2) * n[1] + cell_of(cloud[i], 1)) * n[0] + cell_of(cloud[i], 0);
        start[cid[i] + 1]++;
    }
    for (size_t c = 0; c < ncell; ++c) start[c + 1] += start[c];
    items.resize(nt);
    std::vector<int> cur(start.begin(), start.end() - 1);
    for (size_t i = 0; i < nt; ++i) items[cur[cid[i]]++] = (int)i;
}

// visit(first, last) is called with item ranges; after every shell `done(lb2)` is asked whether points at squared distance >= lb2
// can still matter (lb2 = squared distance from q to everything outside the walked block of cells, deflated for rounding).
template <class F>
void GridNN::rings(const Vec3& q, F& f) const {
    int c[3];
    const float qv[3] = {q.x, q.y, q.z}, lov[3] = {lo.x, lo.y, lo.z}, hiv[3] = {hi.x, hi.y, hi.z};
    for (int a = 0; a < 3; ++a) c[a] = std::min(n[a] - 1, std::max(0, (int)std::floor((qv[a] - lov[a]) / h)));
    const int rmax = std::max(n[0], std::max(n[1], n[2]));
    for (int r = 0; r <= rmax; ++r) {
        const int x0 = std::max(0, c[0] - r), x1 = std::min(n[0] - 1, c[0] + r), y0 = std::max(0, c[1] - r), y1 = std::min(n[1] - 1, c[1] + r);
        const int z0 = std::max(0, c[2] - r), z1 = std::min(n[2] - 1, c[2] + r);
        for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y) {
                const bool edge_zy = (z == c[2] - r || z == c[2] + r || y == c[1] - r || y == c[1] + r);
                if (edge_zy) {
                    const size_t row = ((size_t)z * n[1] + y) * n[0];
                    f.visit(start[row + x0], start[row + x1 + 1]);
                } else {  // only the two end cells of the row lie on the shell
                    const size_t row = ((size_t)z * n[1] + y) * n[0];
                    if (c[0] - r >= 0) f.visit(start[row + c[0] - r], start[row + c[0] - r + 1]);
                    if (r > 0 && c[0] + r < n[0]) f.visit(start[row + c[0] + r], start[row + c[0] + r + 1]);
                }
            }
        // everything not walked yet lies in one of up to six slabs of the grid's box outside the block [x0..x1] x [y0..y1] x [z0..z1]
        const int b0[3] = {x0, y0, z0}, b1[3] = {x1, y1, z1};
        double lb2 = -1.0;
        for (int a = 0; a < 3; ++a)
            for (int side = 0; side < 2; ++side) {
                if (side == 0 ? b0[a] == 0 : b1[a] == n[a] - 1) continue;
                double d2 = 0.0;
                for (int k = 0; k < 3; ++k) {
                    double slo = lov[k], shi = hiv[k];
                    if (k == a) {  // the slab, widened by a hair towards the block (points are binned with fp32 arithmetic)
                        if (side == 0) shi = (double)lov[k] + (double)b0[k] * h + 1e-3 * h;
                        else slo = (double)lov[k] + (double)(b1[k] + 1) * h - 1e-3 * h;
                    }
                    const double d = std::max(std::max(slo - qv[k], qv[k] - shi), 0.0);
                    d2 += d * d;
                }
                lb2 = lb2 < 0.0 || d2 < lb2 ? d2 : lb2;
            }
        if (lb2 < 0.0) return;                       // the block is the whole grid
        if (f.done((float)(lb2 * 0.9999))) return;
    }
}

float GridNN::min_d2(const Vec3& q) const {
    struct V {
        const GridNN& g; const Vec3& q; float best;
        void visit(int a, int b) {
            for (int k = a; k < b; ++k) {
                const Vec3& p = (*g.pc)[g.items[k]];
                const float d = dev_dist_sq(q.x, q.y, q.z, p.x, p.y, p.z);
                if (d < best) best = d;
            }
        }
        bool done(float lb2) const { return lb2 > best; }
    } v{*this, q, kInf};
    rings(q, v);
    return v.best;
}

static float sqrt_tie_threshold(float best) {  // largest float whose correctly rounded sqrt equals sqrt(best)
    uint32_t b;
    std::memcpy(&b, &best, 4);
    const float s = std::sqrt(best);
    for (int it = 0; it < 8; ++it) {
        const uint32_t nb = b + 1;
        float x;
        std::memcpy(&x, &nb, 4);
        if (std::sqrt(x) == s) b = nb; else break;
    }
    float out;
    std::memcpy(&out, &b, 4);
    return out;
}

int GridNN::first_min_sqrt_index(const Vec3& q) const {
    const float thr = sqrt_tie_threshold(min_d2(q));  // icp3d.cu:20-25 compares sqrt distances with a strict '>': the first index of the tie set wins
    struct V {
        const GridNN& g; const Vec3& q; float thr; int idx;
        void visit(int a, int b) {
            for (int k = a; k < b; ++k) {
                const int j = g.items[k];
                const Vec3& p = (*g.pc)[j];
                if (dev_dist_sq(p.x, p.y, p.z, q.x, q.y, q.z) <= thr && j < idx) idx = j;
            }
        }
        bool done(float lb2) const { return lb2 > thr; }
    } v{*this, q, thr, 0x7fffffff};
    rings(q, v);
    return v.idx;
}

void Registration::use_grid(bool on) {
    if (!on) { grid.reset(); return; }
    grid = std::make_shared<GridNN>();
    grid->build(pct);
}

// EXTENSION: trimmed sums (see header).  The k smallest terms, added in ascending order in fp64.
float trimmed_sum(std::vector<float>& v, size_t k) {
    if (k == 0 || k >= v.size()) {
        double s = 0.0;
        for (float x : v) s += (double)x;
        return (float)s;
    }
    std::nth_element(v.begin(), v.begin() + (k - 1), v.end());
    std::sort(v.begin(), v.begin() + k);
    double s = 0.0;
    for (size_t i = 0; i < k; ++i) s += (double)v[i];
    return (float)s;
}

// Sums over points: fixed 1024-point chunks, one fp64 partial per chunk, chunks combined in index
// order — independent of the OpenMP thread count, so fixtures reproduce on any host.
static constexpr long kChunk = 1024;

float Registration::compute_sse_error(const Mat3& R, const Vec3& t) const {  // :62-86 + kernel :14-25
    const long ns = (long)pcs.size();
    const int sum_mode = conventions().sum_mode;
    const bool trimmed = inliers > 0 && inliers < (size_t)ns;
    if (trimmed || sum_mode != 0) {  // per-point values kept: EXTENSION trimmed SSE, or a Thrust-order flip
        std::vector<float> e(ns);
#pragma omp parallel for schedule(static)
        for (long i = 0; i < ns; ++i) {
            Vec3 rp = dev_mul(R, pcs[i]);
            const Vec3 q{rp.x + t.x, rp.y + t.y, rp.z + t.z};
            e[i] = grid ? grid->min_d2(q) : brute_force_find_nearest_neighbor(q, pct);
        }
        return trimmed ? trimmed_sum(e, inliers) : reduce_f32(e.data(), ns, sum_mode);
    }
    const long nchunk = (ns + kChunk - 1) / kChunk;
    std::vector<double> part(nchunk, 0.0);
#pragma omp parallel for schedule(dynamic, 1)
    for (long c = 0; c < nchunk; ++c) {
        double s = 0.0;
        for (long i = c * kChunk; i < std::min(ns, (c + 1) * kChunk); ++i) {
            Vec3 rp = dev_mul(R, pcs[i]);
            Vec3 q{rp.x + t.x, rp.y + t.y, rp.z + t.z};
            s += (double)(grid ? grid->min_d2(q) : brute_force_find_nearest_neighbor(q, pct));
        }
        part[c] = s;
    }
    double sum = 0.0;  // thrust::reduce(float, plus) restated as a double sum (header, conventions)
    for (long c = 0; c < nchunk; ++c) sum += part[c];
    return (float)sum;
}

// kernComputeBounds :27-60 for one point: {ub_i, lb_i}.  `sin_half` and `trans_uncertain_radius` are the per-kernel constants
// of :42-43 and :33, hoisted by the callers (they do not depend on the point).
static inline void bounds_point(const Registration& reg, const Mat3& Rc, const Vec3& tc, const Vec3& p, bool fix_rot, float sin_half,
                                float span_t, float& ubv, float& lbv) {
    const Conventions& cv = conventions();
    const float trans_uncertain_radius = kSqrt3 * span_t;             // :33
    Vec3 rp = dev_mul(Rc, p);
    Vec3 q{rp.x + tc.x, rp.y + tc.y, rp.z + tc.z};                    // :34
    const float dsq = reg.nnlut.search(q);                            // :46
    float d = std::sqrt(dsq);                                         // :48
    if (!fix_rot) {
        const float radius = dev_norm_sq(p);                          // :39-41 (squared norm: reference quirk)
        if (cv.fma_rot_sub) d = std::fmaf(-(2.0f * radius), sin_half, d);
        else d -= 2.0f * radius * sin_half;                           // :43, :51
    }
    ubv = d > 0.0f ? d * d : 0.0f;                                    // :54
    const float l = cv.fma_trans_sub ? std::fmaf(-kSqrt3, span_t, d) : d - trans_uncertain_radius;  // :57
    lbv = l > 0.0f ? l * l : 0.0f;                                    // :58
}

std::tuple<std::vector<float>, std::vector<float>>
Registration::compute_sse_error(const RotNode& rnode, const std::vector<TransNode>& tnodes, bool fix_rot) const {
    const long B = (long)tnodes.size();
    const long ns = (long)pcs.size();
    std::vector<float> upper(B), lower(B);
    float half_angle = rnode.span * kSqrt3 * kPi / 2.0f;  // :42
    float sin_half = dev_sin(half_angle);                 // float overload, as device sin(float)
    const int sum_mode = conventions().sum_mode;
    const bool trimmed = inliers > 0 && inliers < (size_t)ns;
    if (trimmed || sum_mode != 0) {  // per-point values kept: EXTENSION trimmed bounds (the k smallest ub / lb terms), or a Thrust-order flip
#pragma omp parallel for schedule(dynamic, 1)
        for (long b = 0; b < B; ++b) {
            const TransNode& tn = tnodes[b];
            std::vector<float> vu(ns), vl(ns);
            for (long i = 0; i < ns; ++i) bounds_point(*this, rnode.q.R, tn.t, pcs[i], fix_rot, sin_half, tn.span, vu[i], vl[i]);
            upper[b] = trimmed ? trimmed_sum(vu, inliers) : reduce_f32(vu.data(), ns, sum_mode);
            lower[b] = trimmed ? trimmed_sum(vl, inliers) : reduce_f32(vl.data(), ns, sum_mode);
        }
        return {lower, upper};
    }
    const long nchunk = (ns + kChunk - 1) / kChunk;
    std::vector<double> part_ub(B * nchunk, 0.0), part_lb(B * nchunk, 0.0);
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (long b = 0; b < B; ++b)
        for (long c = 0; c < nchunk; ++c) {
            const TransNode& tn = tnodes[b];
            double sum_ub = 0.0, sum_lb = 0.0;
            for (long i = c * kChunk; i < std::min(ns, (c + 1) * kChunk); ++i) {
                float ubv, lbv;
                bounds_point(*this, rnode.q.R, tn.t, pcs[i], fix_rot, sin_half, tn.span, ubv, lbv);
                sum_ub += (double)ubv;
                sum_lb += (double)lbv;
            }
            part_ub[b * nchunk + c] = sum_ub;
            part_lb[b * nchunk + c] = sum_lb;
        }
    for (long b = 0; b < B; ++b) {
        double su = 0.0, sl = 0.0;
        for (long c = 0; c < nchunk; ++c) { su += part_ub[b * nchunk + c]; sl += part_lb[b * nchunk + c]; }
        upper[b] = (float)su;
        lower[b] = (float)sl;
    }
    return {lower, upper};  // :151 — lower first
}

// ---------------------------------------------------------------------------------------------
// 3x3 SVD in double — a restatement of the PUBLISHED ALGORITHM of Eigen::JacobiSVD<Matrix3d> (Eigen 3.3 / 3.4,
// Eigen/src/SVD/JacobiSVD.h `compute`, Eigen/src/Jacobi/Jacobi.h `makeJacobi`, `real_2x2_jacobi_svd`), which is what
// the reference calls at icp3d.cu:118-121 (Eigen3 >= 3.3, fgoicp/CMakeLists.txt:19; not vendored, version unpinned).
// Eigen is absent from this image, so this is written from the algorithm as published, statement by statement:
//   * work matrix W = A / max|A_ij| (scale 1 for the zero matrix), U = V = I;
//   * sweeps over the index pairs (p, q) = (1,0), (2,0), (2,1) — p is the LARGER index — until a sweep rotates nothing;
//     a pair is rotated when |W(p,q)| or |W(q,p)| exceeds max(DBL_MIN, 2 eps * maxDiag), maxDiag = the running maximum of
//     |diagonal| (initialised from the scaled input, raised after every rotation);
//   * the 2x2 block [[W(p,p) W(p,q)] [W(q,p) W(q,q)]] is first symmetrised by a left rotation rot1 (t = m00 + m11,
//     d = m10 - m01, u = t/d, rot1 = (u, 1)/sqrt(1 + u^2)), then diagonalised by the symmetric Jacobi rotation j_right of
//     makeJacobi; j_left = rot1 * j_right^T;  W <- j_left W j_right, U <- U j_left^T, V <- V j_right;
//   * singular values |W(i,i)| * scale, U column negated where W(i,i) < 0; sorted descending by swapping with the FIRST
//     maximum of the tail, stopping at the first zero.
// A rotation (c, s) stands for the matrix [[c, s], [-s, c]] (Eigen's JacobiRotation).
// On a rank-deficient matrix the null-space columns of U and V — and with them R = V diag(1,1,det) U^T — are whatever
// this sequence of rotations leaves; round 2 restated "a two-sided Jacobi" here and a one-sided Hestenes Jacobi in the
// product, which picked different members of the solution family (fuzz seed 7, cases 103 and 505).
// ---------------------------------------------------------------------------------------------
namespace {
struct M3d { double a[3][3]; };  // row-major math matrix a[row][col]

M3d m3d_identity() { M3d m{}; for (int i = 0; i < 3; ++i) m.a[i][i] = 1.0; return m; }
M3d m3d_T(const M3d& x) { M3d o{}; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o.a[i][j] = x.a[j][i]; return o; }
// Eigen's coefficient-based 3x3 product: row . column through the unrolled redux, x0 + (x1 + x2)
M3d m3d_mul(const M3d& x, const M3d& y) {
    M3d o{};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o.a[i][j] = x.a[i][0] * y.a[0][j] + (x.a[i][1] * y.a[1][j] + x.a[i][2] * y.a[2][j]);
    return o;
}
// Eigen's 3x3 determinant (bruteforce_det3_helper): m(0,a) * (m(1,b) m(2,c) - m(1,c) m(2,b)), terms (0,1,2) - (1,0,2) + (2,0,1)
double m3d_det(const M3d& m) {
    auto h = [&](int a, int b, int c) { return m.a[0][a] * (m.a[1][b] * m.a[2][c] - m.a[1][c] * m.a[2][b]); };
    return h(0, 1, 2) - h(1, 0, 2) + h(2, 0, 1);
}

struct JacobiRot {  // [[c, s], [-s, c]]
    double c = 1.0, s = 0.0;
    JacobiRot transpose() const { return JacobiRot{c, -s}; }
    JacobiRot operator*(const JacobiRot& o) const { return JacobiRot{c * o.c - s * o.s, c * o.s + s * o.c}; }
    // makeJacobi(x, y, z): J with J^T [[x, y], [y, z]] J diagonal
    void make_jacobi(double x, double y, double z) {
        const double deno = 2.0 * std::fabs(y);
        if (deno < std::numeric_limits<double>::min()) { c = 1.0; s = 0.0; return; }
        const double tau = (x - z) / deno;
        const double w = std::sqrt(tau * tau + 1.0);
        const double t = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
        const double sign_t = t > 0.0 ? 1.0 : -1.0;
        const double n = 1.0 / std::sqrt(t * t + 1.0);
        s = -sign_t * (y / std::fabs(y)) * std::fabs(t) * n;
        c = n;
    }
};
// apply_rotation_in_the_plane: (x, y) <- (c x + s y, -s x + c y); the identity rotation is skipped
inline void rot_apply(double& x, double& y, const JacobiRot& j) {
    if (j.c == 1.0 && j.s == 0.0) return;
    const double xi = x, yi = y;
    x = j.c * xi + j.s * yi;
    y = -j.s * xi + j.c * yi;
}
void apply_on_the_left(M3d& m, int p, int q, const JacobiRot& j) { for (int k = 0; k < 3; ++k) rot_apply(m.a[p][k], m.a[q][k], j); }
void apply_on_the_right(M3d& m, int p, int q, const JacobiRot& j) { const JacobiRot jt = j.transpose(); for (int k = 0; k < 3; ++k) rot_apply(m.a[k][p], m.a[k][q], jt); }

void real_2x2_jacobi_svd(const M3d& W, int p, int q, JacobiRot& j_left, JacobiRot& j_right) {
    double m[2][2] = {{W.a[p][p], W.a[p][q]}, {W.a[q][p], W.a[q][q]}};
    JacobiRot rot1;
    const double t = m[0][0] + m[1][1];
    const double d = m[1][0] - m[0][1];
    if (std::fabs(d) < std::numeric_limits<double>::min()) {
        rot1.s = 0.0; rot1.c = 1.0;
    } else {
        const double u = t / d;
        const double tmp = std::sqrt(1.0 + u * u);
        rot1.s = 1.0 / tmp;
        rot1.c = u / tmp;
    }
    for (int k = 0; k < 2; ++k) rot_apply(m[0][k], m[1][k], rot1);  // m.applyOnTheLeft(0, 1, rot1)
    j_right.make_jacobi(m[0][0], m[0][1], m[1][1]);
    j_left = rot1 * j_right.transpose();
}

// A = U * diag(S) * V^T, S sorted descending, S >= 0.  JacobiSVD<Matrix3d>::compute, full U and V.
void svd3_eigen_jacobi(const M3d& Ain, M3d& U, double S[3], M3d& V) {
    const double precision = 2.0 * std::numeric_limits<double>::epsilon();
    const double consider_as_zero = std::numeric_limits<double>::min();
    double scale = 0.0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) scale = std::max(scale, std::fabs(Ain.a[i][j]));
    if (scale == 0.0) scale = 1.0;
    M3d W;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) W.a[i][j] = Ain.a[i][j] / scale;
    U = m3d_identity();
    V = m3d_identity();
    double max_diag = std::max(std::fabs(W.a[0][0]), std::max(std::fabs(W.a[1][1]), std::fabs(W.a[2][2])));
    bool finished = false;
    int guard = 0;  // Eigen loops until finished; NaN input ends it there as here (comparisons false); the guard only bounds a defect
    while (!finished && guard++ < 1000) {
        finished = true;
        for (int p = 1; p < 3; ++p)
            for (int q = 0; q < p; ++q) {
                const double threshold = std::max(consider_as_zero, precision * max_diag);
                if (std::fabs(W.a[p][q]) > threshold || std::fabs(W.a[q][p]) > threshold) {
                    finished = false;
                    JacobiRot j_left, j_right;
                    real_2x2_jacobi_svd(W, p, q, j_left, j_right);
                    apply_on_the_left(W, p, q, j_left);
                    apply_on_the_right(U, p, q, j_left.transpose());
                    apply_on_the_right(W, p, q, j_right);
                    apply_on_the_right(V, p, q, j_right);
                    max_diag = std::max(max_diag, std::max(std::fabs(W.a[p][p]), std::fabs(W.a[q][q])));
                }
            }
    }
    for (int i = 0; i < 3; ++i) {
        const double a = W.a[i][i];
        S[i] = std::fabs(a);
        if (a < 0.0) for (int r = 0; r < 3; ++r) U.a[r][i] = -U.a[r][i];
    }
    for (int i = 0; i < 3; ++i) S[i] *= scale;
    for (int i = 0; i < 3; ++i) {  // sort: first maximum of the tail; stop at the first zero
        int pos = i;
        for (int j = i + 1; j < 3; ++j) if (S[j] > S[pos]) pos = j;
        if (S[pos] == 0.0) break;
        if (pos != i) {
            std::swap(S[i], S[pos]);
            for (int r = 0; r < 3; ++r) { std::swap(U.a[r][i], U.a[r][pos]); std::swap(V.a[r][i], V.a[r][pos]); }
        }
    }
}

// Round 2's stand-in ("a two-sided Jacobi", own sweep order (0,1),(0,2),(1,2), own per-pair threshold, selection sort), kept
// ONLY as a convention flip (Conventions::svd_r2_two_sided): on full-rank H it must give the same R, on rank-deficient H it
// shows how much of R is the SVD's choice.
// A = U * diag(S) * V^T, S sorted descending, S >= 0.
void svd3_two_sided_r2(const M3d& Ain, M3d& U, double S[3], M3d& V) {
    M3d A = Ain;
    U = m3d_identity();
    V = m3d_identity();
    double scale = 0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) scale = std::max(scale, std::fabs(A.a[i][j]));
    if (scale == 0) { S[0] = S[1] = S[2] = 0; return; }
    const double eps = std::numeric_limits<double>::epsilon();
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool done = true;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double thr = std::max(std::fabs(A.a[p][p]), std::fabs(A.a[q][q])) * 2 * eps;
                thr = std::max(thr, std::numeric_limits<double>::min());
                if (std::fabs(A.a[p][q]) <= thr && std::fabs(A.a[q][p]) <= thr) continue;
                done = false;
                // 2x2 block
                double a = A.a[p][p], b = A.a[p][q], c = A.a[q][p], d = A.a[q][q];
                // left rotation G=[[cg,sg],[-sg,cg]] with G*M symmetric: cg*b + sg*d = -sg*a + cg*c
                double cg = 1, sg = 0;
                double tt = a + d, dd = c - b;
                if (std::fabs(dd) > std::numeric_limits<double>::min()) {
                    double u = tt / dd;
                    double h = std::sqrt(1.0 + u * u);
                    sg = 1.0 / h;
                    cg = u / h;
                }
                // S = G*M
                double s00 = cg * a + sg * c, s01 = cg * b + sg * d, s11 = -sg * b + cg * d;
                // Jacobi rotation J=[[cj,sj],[-sj,cj]] diagonalising symmetric [[s00,s01],[s01,s11]]
                double cj = 1, sj = 0;
                if (std::fabs(s01) > std::numeric_limits<double>::min()) {
                    double tau = (s11 - s00) / (2.0 * s01);
                    double tj = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
                    cj = 1.0 / std::sqrt(1.0 + tj * tj);
                    sj = tj * cj;
                }
                // M = G^T S, S = J D J^T  =>  A <- (J^T G) A J,  U <- U (J^T G)^T,  V <- V J
                double G[2][2] = {{cg, sg}, {-sg, cg}};
                double J[2][2] = {{cj, sj}, {-sj, cj}};
                double L[2][2];  // L = J^T * G
                for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) L[i][j] = J[0][i] * G[0][j] + J[1][i] * G[1][j];
                // A <- L * A (rows p,q)
                for (int col = 0; col < 3; ++col) {
                    double r0 = A.a[p][col], r1 = A.a[q][col];
                    A.a[p][col] = L[0][0] * r0 + L[0][1] * r1;
                    A.a[q][col] = L[1][0] * r0 + L[1][1] * r1;
                }
                // A <- A * J (cols p,q)
                for (int row = 0; row < 3; ++row) {
                    double c0 = A.a[row][p], c1 = A.a[row][q];
                    A.a[row][p] = c0 * J[0][0] + c1 * J[1][0];
                    A.a[row][q] = c0 * J[0][1] + c1 * J[1][1];
                }
                // U <- U * L^T, V <- V * J
                for (int row = 0; row < 3; ++row) {
                    double u0 = U.a[row][p], u1 = U.a[row][q];
                    U.a[row][p] = u0 * L[0][0] + u1 * L[0][1];
                    U.a[row][q] = u0 * L[1][0] + u1 * L[1][1];
                    double v0 = V.a[row][p], v1 = V.a[row][q];
                    V.a[row][p] = v0 * J[0][0] + v1 * J[1][0];
                    V.a[row][q] = v0 * J[0][1] + v1 * J[1][1];
                }
            }
        if (done) break;
    }
    for (int i = 0; i < 3; ++i) {
        S[i] = A.a[i][i];
        if (S[i] < 0) { S[i] = -S[i]; for (int r = 0; r < 3; ++r) U.a[r][i] = -U.a[r][i]; }
    }
    // sort descending (selection, swapping columns of U and V)
    for (int i = 0; i < 2; ++i) {
        int m = i;
        for (int j = i + 1; j < 3; ++j) if (S[j] > S[m]) m = j;
        if (m != i) {
            std::swap(S[i], S[m]);
            for (int r = 0; r < 3; ++r) { std::swap(U.a[r][i], U.a[r][m]); std::swap(V.a[r][i], V.a[r][m]); }
        }
    }
}

void svd3(const M3d& A, M3d& U, double S[3], M3d& V) {
    if (conventions().svd_r2_two_sided) svd3_two_sided_r2(A, U, S, V);
    else svd3_eigen_jacobi(A, U, S, V);
}
}  // namespace

// icp3d.cu:110-138
Mat3 closest_orthogonal_approximation(const Mat3& ABt) {
    M3d H;  // matrix(r,c) = ABt[c][r]  (:113-116)
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) H.a[r][c] = (double)ABt.c[c][r];
    M3d U, V;
    double S[3];
    svd3(H, U, S, V);
    M3d VUt = m3d_mul(V, m3d_T(U));
    double det = m3d_det(VUt);
    M3d D = m3d_identity();
    D.a[2][2] = det;
    M3d R = m3d_mul(m3d_mul(V, D), m3d_T(U));
    Mat3 out;  // glm::mat3{R(0,0),R(1,0),R(2,0), ...}: out[c][r] = R(r,c)  (:135-137)
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) out.c[c][r] = (float)R.a[r][c];
    return out;
}

// exposed for tests through the C API
void svd3_rowmajor(const double* A9, double* U9, double* S3, double* V9) {
    M3d A, U, V;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A.a[i][j] = A9[i * 3 + j];
    svd3(A, U, S3, V);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { U9[i * 3 + j] = U.a[i][j]; V9[i * 3 + j] = V.a[i][j]; }
}

// ---------------------------------------------------------------------------------------------
// IterativeClosestPoint3D — icp3d.cu:11-172
// ---------------------------------------------------------------------------------------------
IterativeClosestPoint3D::IterativeClosestPoint3D(const Registration& reg, const PointCloud& pct, const PointCloud& pcs, size_t max_iter,
                                                 float conv_thr, const Mat3& R, const Vec3& t)
    : reg_(reg), pct_(pct), pcs_buf_(pcs), R_(R), t_(t), max_iter_(max_iter), thr_(conv_thr) {}

static void rotate_translate_inplace(PointCloud& pc, const Mat3& R, const Vec3& t) {  // kernRotateTranslateInplace :30-36
    const long n = (long)pc.size();
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        Vec3 rp = dev_mul(R, pc[i]);
        pc[i] = Vec3{rp.x + t.x, rp.y + t.y, rp.z + t.z};
    }
}

std::tuple<Mat3, Vec3> IterativeClosestPoint3D::procrustes(ProcrustesDebug* dbg) {  // :140-172
    const long ns = (long)pcs_buf_.size();
    const long nt = (long)pct_.size();
    PointCloud corrs(ns);
    corr_idx_.assign(ns, -1);
    // kernFindNearestNeighbor :11-28 — sqrt distance, strict '>' so the first minimum wins
#pragma omp parallel for schedule(static)
    for (long i = 0; i < ns; ++i) {
        float dist_min = kInf;
        Vec3 corr{0.f, 0.f, 0.f};
        int best = -1;
        const Vec3 s = pcs_buf_[i];
        if (reg_.grid) {  // CPU baseline only: the same index through the grid
            best = reg_.grid->first_min_sqrt_index(s);
            corrs[i] = pct_[best];
            corr_idx_[i] = best;
            continue;
        }
        for (long j = 0; j < nt; ++j) {
            float dist = std::sqrt(dev_dist_sq(pct_[j].x, pct_[j].y, pct_[j].z, s.x, s.y, s.z));
            if (dist_min > dist) { dist_min = dist; corr = pct_[j]; best = (int)j; }
        }
        corrs[i] = corr;
        corr_idx_[i] = best;
    }
    // EXTENSION: trimmed ICP — only the `inliers` correspondences with the smallest squared distance take part
    // (ties at the cut: lowest index first); every mean below is then over the inliers.
    std::vector<char> use(ns, 1);
    long nuse = ns;
    if (reg_.inliers > 0 && reg_.inliers < (size_t)ns) {
        std::vector<std::pair<float, long>> d(ns);
        for (long i = 0; i < ns; ++i)
            d[i] = {dev_dist_sq(pcs_buf_[i].x, pcs_buf_[i].y, pcs_buf_[i].z, corrs[i].x, corrs[i].y, corrs[i].z), i};
        std::sort(d.begin(), d.end());
        std::fill(use.begin(), use.end(), 0);
        nuse = (long)reg_.inliers;
        for (long k = 0; k < nuse; ++k) use[d[k].second] = 1;
    }
    // thrust::reduce of Point3D x2 (:152-153) → double sums rounded to fp32 (Conventions::sum_mode 1 / 2: fp32 tree / serial)
    const int sum_mode = conventions().sum_mode;
    auto reduce_col = [&](auto&& value_of) {  // Σ over the used points of one scalar per point
        if (sum_mode == 0) {
            double acc = 0.0;
            for (long i = 0; i < ns; ++i) if (use[i]) acc += (double)value_of(i);
            return (float)acc;
        }
        std::vector<float> v;
        v.reserve(ns);
        for (long i = 0; i < ns; ++i) if (use[i]) v.push_back(value_of(i));
        return reduce_f32(v.data(), (long)v.size(), sum_mode);
    };
    Vec3 src_centroid{reduce_col([&](long i) { return pcs_buf_[i].x; }), reduce_col([&](long i) { return pcs_buf_[i].y; }),
                      reduce_col([&](long i) { return pcs_buf_[i].z; })};
    Vec3 cor_centroid{reduce_col([&](long i) { return corrs[i].x; }), reduce_col([&](long i) { return corrs[i].y; }),
                      reduce_col([&](long i) { return corrs[i].z; })};
    float fn = static_cast<float>(nuse);
    src_centroid = Vec3{src_centroid.x / fn, src_centroid.y / fn, src_centroid.z / fn};  // :155-156
    cor_centroid = Vec3{cor_centroid.x / fn, cor_centroid.y / fn, cor_centroid.z / fn};
    // kernCentralize x2 (:38-44), kernOuterProduct (:46-52), reduce mat3 (:165-166)
    // glm::outerProduct(c=a, r=b): m[col][row] = a[row]*b[col]
    auto a_of = [&](long i, int k) { return (k == 0 ? pcs_buf_[i].x : k == 1 ? pcs_buf_[i].y : pcs_buf_[i].z) - (k == 0 ? src_centroid.x : k == 1 ? src_centroid.y : src_centroid.z); };
    auto b_of = [&](long i, int k) { return (k == 0 ? corrs[i].x : k == 1 ? corrs[i].y : corrs[i].z) - (k == 0 ? cor_centroid.x : k == 1 ? cor_centroid.y : cor_centroid.z); };
    Mat3 ABt;
    for (int col = 0; col < 3; ++col)
        for (int row = 0; row < 3; ++row) ABt.c[col][row] = reduce_col([&](long i) { return a_of(i, row) * b_of(i, col); });
    Mat3 Rn = closest_orthogonal_approximation(ABt);  // :168
    Vec3 rs = host_mul(Rn, src_centroid);
    Vec3 tn{cor_centroid.x - rs.x, cor_centroid.y - rs.y, cor_centroid.z - rs.z};  // :169
    if (dbg) { dbg->src_centroid = src_centroid; dbg->cor_centroid = cor_centroid; dbg->ABt = ABt; }
    return {Rn, tn};
}

std::tuple<float, Mat3, Vec3> IterativeClosestPoint3D::run() {  // :80-108
    rotate_translate_inplace(pcs_buf_, R_, t_);  // :85
    size_t iter = 0;
    float sse = kInf;
    float last_sse = 2.0f * kInf;
    Mat3 last_R = mat3_identity();
    Vec3 last_t{0, 0, 0};
    iters_ = 0;
    while (iter++ < max_iter_ && (last_sse - sse) > thr_ * last_sse) {
        last_sse = sse;
        last_R = R_;
        last_t = t_;
        auto [Rn, tn] = procrustes();
        rotate_translate_inplace(pcs_buf_, Rn, tn);  // :100
        R_ = host_mul(Rn, R_);                        // :101
        Vec3 rt = host_mul(Rn, t_);
        t_ = Vec3{rt.x + tn.x, rt.y + tn.y, rt.z + tn.z};  // :102
        sse = reg_.compute_sse_error(R_, t_);         // :103
        ++iters_;
    }
    if (sse < last_sse) return {sse, R_, t_};
    return {last_sse, last_R, last_t};
}

// ---------------------------------------------------------------------------------------------
// Pre-processing — fgoicp.cpp:176-287 (the omp pragmas there are inert; serial fp32 sums)
// ---------------------------------------------------------------------------------------------
Vec3 center_point_cloud(PointCloud& pc) {
    Vec3 c{0, 0, 0};
    for (size_t i = 0; i < pc.size(); ++i) { c.x += pc[i].x; c.y += pc[i].y; c.z += pc[i].z; }
    float fn = static_cast<float>(pc.size());
    c = Vec3{c.x / fn, c.y / fn, c.z / fn};
    for (size_t i = 0; i < pc.size(); ++i) { pc[i].x -= c.x; pc[i].y -= c.y; pc[i].z -= c.z; }
    return Vec3{-c.x, -c.y, -c.z};
}

float get_scaling_factor(const PointCloud& pc) {
    float max_abs = std::numeric_limits<float>::lowest();
    for (const auto& p : pc) max_abs = std::max(max_abs, std::max(std::fabs(p.x), std::max(std::fabs(p.y), std::fabs(p.z))));
    return 1.0f / max_abs;
}

float scale_point_clouds(PointCloud& pct, PointCloud& pcs) {
    float s = get_scaling_factor(pcs);
    for (auto& p : pcs) { p.x *= s; p.y *= s; p.z *= s; }
    for (auto& p : pct) { p.x *= s; p.y *= s; p.z *= s; }
    return s;
}

Bounds get_point_cloud_ranges(const PointCloud& pc) {
    Bounds r = {std::make_pair(std::numeric_limits<float>::max(), std::numeric_limits<float>::lowest()),
                std::make_pair(std::numeric_limits<float>::max(), std::numeric_limits<float>::lowest()),
                std::make_pair(std::numeric_limits<float>::max(), std::numeric_limits<float>::lowest())};
    for (const auto& p : pc) {
        r[0].first = std::min(r[0].first, p.x); r[0].second = std::max(r[0].second, p.x);
        r[1].first = std::min(r[1].first, p.y); r[1].second = std::max(r[1].second, p.y);
        r[2].first = std::min(r[2].first, p.z); r[2].second = std::max(r[2].second, p.z);
    }
    return r;
}

// ---------------------------------------------------------------------------------------------
// FastGoICP — fgoicp.hpp:13-25 (member init order :47-58), fgoicp.cpp:10-174
// ---------------------------------------------------------------------------------------------
FastGoICP::FastGoICP(PointCloud pct_, PointCloud pcs_, float lut_resolution, float mse_thr, float trim_fraction)
    : pcs(std::move(pcs_)), pct(std::move(pct_)), ns(pcs.size()), nt(pct.size()),
      offset_pcs(center_point_cloud(pcs)), offset_pct(center_point_cloud(pct)),
      scaling_factor(scale_point_clouds(pct, pcs)), target_bounds(get_point_cloud_ranges(pct)),
      registration(pct, pcs, target_bounds, lut_resolution),
      best_sse(kInf), best_rotation(mat3_identity()), best_translation{0, 0, 0},
      mse_threshold(mse_thr), sse_threshold(ns * mse_thr) {
    if (trim_fraction > 0.0f) {  // EXTENSION: inlierNum = (int)(Nd * (1 - trimFraction)) as in Go-ICP; threshold over the inliers
        size_t k = (size_t)((double)ns * (1.0 - (double)trim_fraction));
        if (k < 1) k = 1;
        if (k < ns) {
            registration.inliers = k;
            sse_threshold = k * mse_thr;
        }
    }
}

Vec3 FastGoICP::restore_translation(const Mat3& R, const Vec3& t) const {  // fgoicp.hpp:87-90
    Vec3 ro = host_mul(R, offset_pcs);
    return Vec3{t.x / scaling_factor + ro.x - offset_pct.x, t.y / scaling_factor + ro.y - offset_pct.y,
                t.z / scaling_factor + ro.z - offset_pct.z};
}

std::tuple<Mat3, Vec3> FastGoICP::run() {  // fgoicp.cpp:10-30
    {
        IterativeClosestPoint3D icp3d(registration, pct, pcs, 100, 0.05f, mat3_identity(), Vec3{0, 0, 0});
        auto [icp_sse, icp_R, icp_t] = icp3d.run();
        (void)icp_R; (void)icp_t;
        best_sse = icp_sse;  // :14 — only the error is adopted, not (R,t)
        stats_.icp_runs++; stats_.icp_iters += icp3d.iterations();
    }
    branch_and_bound_SO3();
    IterativeClosestPoint3D icp3d_best(registration, pct, pcs, 100, 0.0005f, best_rotation, best_translation);
    std::tie(best_sse, best_rotation, best_translation) = icp3d_best.run();
    stats_.icp_runs++; stats_.icp_iters += icp3d_best.iterations();
    return {best_rotation, restore_translation(best_rotation, best_translation)};
}

float FastGoICP::branch_and_bound_SO3() {  // fgoicp.cpp:32-100
    std::priority_queue<RotNode> rcandidates;
    rcandidates.push(RotNode(0.0f, 0.0f, 0.0f, 1.0f, 0.0f, best_sse));
    while (!rcandidates.empty()) {
        RotNode rnode = rcandidates.top();
        rcandidates.pop();
        if (best_sse - rnode.lb <= sse_threshold) break;  // :44
        float span = rnode.span / 2.0f;
        for (char j = 0; j < 8; ++j) {
            if (span < 0.05f) continue;  // :53
            RotNode child(rnode.q.x - span + (j >> 0 & 1) * rnode.span, rnode.q.y - span + (j >> 1 & 1) * rnode.span,
                          rnode.q.z - span + (j >> 2 & 1) * rnode.span, span, rnode.lb, rnode.ub);
            if (!child.overlaps_SO3()) continue;
            if (!child.q.in_SO3()) { rcandidates.push(child); continue; }
            stats_.rot_cubes++;
            auto [ub, best_t] = branch_and_bound_R3(child, true);  // :69
            if (ub < best_sse * 1.8) {                              // :74 (double compare)
                IterativeClosestPoint3D icp3d(registration, pct, pcs, 100, 0.005f, child.q.R, best_t);
                auto [icp_sse, icp_R, icp_t] = icp3d.run();
                stats_.icp_runs++; stats_.icp_iters += icp3d.iterations();
                if (icp_sse < best_sse) { best_sse = icp_sse; best_rotation = icp_R; best_translation = icp_t; }
            }
            auto [lb, unused_t] = branch_and_bound_R3(child, false);  // :90
            (void)unused_t;
            if (lb >= best_sse) continue;
            child.lb = lb;
            child.ub = ub;
            rcandidates.push(child);
        }
    }
    return best_sse;
}

std::tuple<float, Vec3> FastGoICP::branch_and_bound_R3(RotNode& rnode, bool fix_rot) {  // fgoicp.cpp:102-174
    float best_error = best_sse;
    Vec3 best_t{0, 0, 0};
    float best_ub = kInf;
    stats_.inner_bnb++;
    std::priority_queue<TransNode> tcandidates;
    tcandidates.push(TransNode(0.0f, 0.0f, 0.0f, 1.0f, 0.0f, rnode.ub));
    while (!tcandidates.empty()) {
        std::vector<TransNode> tnodes;
        if (best_error - tcandidates.top().lb < sse_threshold) break;  // :120
        while (!tcandidates.empty() && tnodes.size() < 32) {
            TransNode tn = tcandidates.top();
            tcandidates.pop();
            if (tn.lb < best_error) tnodes.push_back(tn);
        }
        stats_.trans_cubes += tnodes.size();
        stats_.bounds_calls++;
        auto [lb, ub] = registration.compute_sse_error(rnode, tnodes, fix_rot);
        size_t idx_min = std::distance(ub.begin(), std::min_element(ub.begin(), ub.end()));
        best_ub = best_ub < ub[idx_min] ? best_ub : ub[idx_min];
        if (ub[idx_min] < best_error) { best_error = ub[idx_min]; best_t = tnodes[idx_min].t; }
        for (size_t i = 0; i < tnodes.size(); ++i) {
            if (lb[i] >= best_error) continue;
            TransNode& tn = tnodes[i];
            if (tn.span < 0.1f) continue;  // :155
            float span = tn.span / 2.0f;
            for (char j = 0; j < 8; ++j)
                tcandidates.push(TransNode(tn.t.x - span + (j >> 0 & 1) * tn.span, tn.t.y - span + (j >> 1 & 1) * tn.span,
                                           tn.t.z - span + (j >> 2 & 1) * tn.span, span, lb[i], ub[i]));
        }
    }
    return {best_ub, best_t};
}

}  // namespace goicp_oracle
