// Host-side construction and upload of the implicit bounding-volume tree (bvh.hpp).
#include "bvh.hpp"
#include "../host/knobs.hpp"

#include <cfloat>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "../host/math3.hpp"
#include "morton.hpp"

namespace fgoicp {

bool bvh_kd_order() {
    static const bool kd = [] { const char* e = dev_env("FGOICP_BVH_ORDER"); return !e || std::atoi(e) != 0; }();  // tuning knob / A-B
    return kd;
}

BvhHost bvh_build_host(const float4* p, size_t n, std::vector<uint32_t>* order) {
    BvhHost h;
    const size_t nleaf_needed = (n + kBvhLeaf - 1) / kBvhLeaf;
    int depth = 0;
    while (((size_t)1 << depth) < nleaf_needed) ++depth;
    const size_t nleaf = (size_t)1 << depth;
    h.depth = depth;
    h.first_leaf = (int)(nleaf - 1);
    const size_t nnodes = 2 * nleaf - 1;
    // leaves = the cells of a k-d tree (FGOICP_BVH_ORDER=1, default) or runs of the space-filling curve (0): morton.hpp
    const bool kd = bvh_kd_order();
    std::vector<uint32_t> own;
    std::vector<uint32_t>& perm = order ? *order : own;
    if (perm.size() != n) perm = kd ? kd_order(reinterpret_cast<const float*>(p), n, 4, (size_t)kBvhLeaf) : morton_order(reinterpret_cast<const float*>(p), n, 4);
    h.pts.assign(nleaf * kBvhLeaf, make_float4(FLT_MAX, FLT_MAX, FLT_MAX, 0.f));
    for (size_t i = 0; i < nleaf * kBvhLeaf; ++i) {
        uint32_t idx = 0x7fffffffu;
        if (i < n) {
            idx = perm[i];
            h.pts[i].x = p[idx].x; h.pts[i].y = p[idx].y; h.pts[i].z = p[idx].z;
        }
        std::memcpy(&h.pts[i].w, &idx, sizeof(idx));
    }
    h.box.assign(2 * nnodes, make_float4(0, 0, 0, 0));
    auto set_empty = [&](size_t node) {
        h.box[2 * node] = make_float4(FLT_MAX, FLT_MAX, FLT_MAX, 0.f);
        h.box[2 * node + 1] = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, 0.f);
    };
    for (size_t l = 0; l < nleaf; ++l) {
        const size_t node = h.first_leaf + l;
        set_empty(node);
        for (size_t k = 0; k < (size_t)kBvhLeaf; ++k) {
            const size_t i = l * kBvhLeaf + k;
            if (i >= n) break;
            float4& lo = h.box[2 * node];
            float4& hi = h.box[2 * node + 1];
            lo.x = std::min(lo.x, h.pts[i].x); lo.y = std::min(lo.y, h.pts[i].y); lo.z = std::min(lo.z, h.pts[i].z);
            hi.x = std::max(hi.x, h.pts[i].x); hi.y = std::max(hi.y, h.pts[i].y); hi.z = std::max(hi.z, h.pts[i].z);
        }
    }
    // LEAF SLABS (round 3): a leaf of a surface is flat, its axis-aligned box is not — a large ball around a far query cuts the box
    // of every leaf inside a cap of radius sqrt(2 d s) (s = the box's size), while the leaf's points lie between two parallel planes a few
    // noise amplitudes apart.  Per leaf: n = the direction of least variance of its points (|n| <= 1 after rounding), [a, b] = the range of
    // n.p over them, widened by the rounding of both sides.  max(box distance, slab distance) is still a lower bound of the distance to
    // every point of the leaf (kernels.hip box_walk refines its per-query test with it).
    static const bool want_slab = [] { const char* e = dev_env("FGOICP_BVH_SLAB"); return !e || std::atoi(e) != 0; }();  // tuning knob / A-B
    if (want_slab) {
        h.slab.assign(2 * nleaf, make_float4(0.f, 0.f, 0.f, 0.f));
        const size_t nreal = (n + kBvhLeaf - 1) / kBvhLeaf;
        auto slab_range = [&](size_t l0, size_t l1) {
        for (size_t l = l0; l < l1; ++l) {
            const size_t b0 = l * kBvhLeaf, b1 = std::min(n, b0 + kBvhLeaf);
            if (b1 <= b0 + 2) continue;  // (0, 0, 0 | 0, 0): a slab that never rejects
            double c[3] = {0, 0, 0};
            for (size_t i = b0; i < b1; ++i) { c[0] += h.pts[i].x; c[1] += h.pts[i].y; c[2] += h.pts[i].z; }
            for (double& v : c) v /= (double)(b1 - b0);
            double C[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            for (size_t i = b0; i < b1; ++i) {
                const double d[3] = {h.pts[i].x - c[0], h.pts[i].y - c[1], h.pts[i].z - c[2]};
                for (int r = 0; r < 3; ++r)
                    for (int q = 0; q < 3; ++q) C[r][q] += d[r] * d[q];
            }
            double U[3][3], S[3], V[3][3];
            svd3_jacobi(C, U, S, V);  // symmetric positive semi-definite: singular values descending, the last column = least variance
            double nn[3] = {U[0][2], U[1][2], U[2][2]};
            const double len = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
            if (!(len > 0.5) || !std::isfinite(len)) continue;
            const float nf[3] = {(float)(nn[0] / len * (1.0 - 1e-6)), (float)(nn[1] / len * (1.0 - 1e-6)), (float)(nn[2] / len * (1.0 - 1e-6))};
            double lo = 1e300, hi = -1e300;
            for (size_t i = b0; i < b1; ++i) {
                const double pr = (double)nf[0] * h.pts[i].x + (double)nf[1] * h.pts[i].y + (double)nf[2] * h.pts[i].z;
                lo = std::min(lo, pr);
                hi = std::max(hi, pr);
            }
            const double m = 4e-7 * std::max(std::fabs(lo), std::fabs(hi)) + 1e-30;
            h.slab[2 * l] = make_float4(nf[0], nf[1], nf[2], (float)(lo - m));
            h.slab[2 * l + 1] = make_float4((float)(hi + m), 0.f, 0.f, 0.f);
            // float conversion may round inwards: step outwards once more
            h.slab[2 * l].w = std::nextafter(h.slab[2 * l].w, -FLT_MAX);
            h.slab[2 * l + 1].x = std::nextafter(h.slab[2 * l + 1].x, FLT_MAX);
        }
        };
        // one 3 x 3 eigen-decomposition per leaf: a few threads for big clouds (31 250 leaves at 1M points)
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t nth = nreal >= 4096 ? std::min<size_t>(8, hw > 1 ? hw : 1) : 1;
        if (nth <= 1) {
            slab_range(0, nreal);
        } else {
            std::vector<std::thread> th;
            try {
                for (size_t k = 0; k < nth; ++k) th.emplace_back(slab_range, nreal * k / nth, nreal * (k + 1) / nth);
            } catch (...) {  // a thread that cannot start: the rest is done here
                const size_t done = th.size();
                for (auto& x : th) x.join();
                th.clear();
                slab_range(nreal * done / nth, nreal);
            }
            for (auto& x : th) x.join();
        }
    }
    for (long node = (long)h.first_leaf - 1; node >= 0; --node) {
        const size_t l = 2 * (size_t)node + 1, r = l + 1;
        float4& lo = h.box[2 * node];
        float4& hi = h.box[2 * node + 1];
        lo = make_float4(std::min(h.box[2 * l].x, h.box[2 * r].x), std::min(h.box[2 * l].y, h.box[2 * r].y), std::min(h.box[2 * l].z, h.box[2 * r].z), 0.f);
        hi = make_float4(std::max(h.box[2 * l + 1].x, h.box[2 * r + 1].x), std::max(h.box[2 * l + 1].y, h.box[2 * r + 1].y),
                         std::max(h.box[2 * l + 1].z, h.box[2 * r + 1].z), 0.f);
    }
    return h;
}

hipError_t bvh_upload(const BvhHost& h, BvhDevice* d) {
    d->depth = h.depth;
    d->first_leaf = h.first_leaf;
    hipError_t e = hipMalloc(&d->box, h.box.size() * sizeof(float4));
    if (e != hipSuccess) return e;
    e = hipMalloc(&d->pts, h.pts.size() * sizeof(float4));
    if (e != hipSuccess) return e;
    e = hipMemcpy(d->box, h.box.data(), h.box.size() * sizeof(float4), hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    if (!h.slab.empty()) {
        e = hipMalloc(&d->slab, h.slab.size() * sizeof(float4));
        if (e != hipSuccess) return e;
        e = hipMemcpy(d->slab, h.slab.data(), h.slab.size() * sizeof(float4), hipMemcpyHostToDevice);
        if (e != hipSuccess) return e;
    }
    return hipMemcpy(d->pts, h.pts.data(), h.pts.size() * sizeof(float4), hipMemcpyHostToDevice);
}

void bvh_free(BvhDevice* d) {
    if (d->box) (void)hipFree(d->box);
    if (d->pts) (void)hipFree(d->pts);
    if (d->slab) (void)hipFree(d->slab);
    d->box = nullptr;
    d->pts = nullptr;
    d->slab = nullptr;
}

}  // namespace fgoicp
