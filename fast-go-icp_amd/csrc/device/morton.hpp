// 30-bit space-filling-curve order of a point set inside its own AABB (host).  Used for the source cloud
// (neighbouring lanes stay in neighbouring LUT voxels under any rigid motion) and for the BVH.
// The curve is Hilbert's by default (every run of consecutive points is one connected patch; a Z-order
// run that straddles a power-of-two boundary is two patches far apart), FGOICP_POINT_CURVE=0 selects
// Z-order.  Only locality depends on the order, results do not.
#pragma once
#include "../host/knobs.hpp"
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

namespace fgoicp {

inline uint32_t morton_expand10(uint32_t v) {
    v &= 0x3ff;
    v = (v | (v << 16)) & 0x030000FF;
    v = (v | (v << 8)) & 0x0300F00F;
    v = (v | (v << 4)) & 0x030C30C3;
    v = (v | (v << 2)) & 0x09249249;
    return v;
}

// 30-bit Hilbert index (10 bits per axis), Skilling's transpose algorithm
inline uint32_t hilbert30(uint32_t x, uint32_t y, uint32_t z) {
    uint32_t X[3] = {x & 1023u, y & 1023u, z & 1023u};
    for (uint32_t Q = 512u; Q > 1u; Q >>= 1) {
        const uint32_t P = Q - 1u;
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) X[0] ^= P;
            else { const uint32_t t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    uint32_t t = 0;
    for (uint32_t Q = 512u; Q > 1u; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1u;
    X[0] ^= t; X[1] ^= t; X[2] ^= t;
    return morton_expand10(X[2]) | (morton_expand10(X[1]) << 1) | (morton_expand10(X[0]) << 2);
}

// xyz: n points with the given float stride (3 for packed xyz, 4 for float4)
inline std::vector<uint32_t> morton_order(const float* xyz, size_t n, size_t stride) {
    static const bool hilbert = [] { const char* e = dev_env("FGOICP_POINT_CURVE"); return e ? std::atoi(e) != 0 : true; }();  // tuning knob
    std::vector<uint32_t> perm(n);
    std::iota(perm.begin(), perm.end(), 0u);
    if (n == 0) return perm;
    float lo[3] = {xyz[0], xyz[1], xyz[2]}, hi[3] = {xyz[0], xyz[1], xyz[2]};
    for (size_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], xyz[stride * i + a]);
            hi[a] = std::max(hi[a], xyz[stride * i + a]);
        }
    const float ext = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
    if (!(ext > 0)) return perm;
    std::vector<uint32_t> code(n);
    for (size_t i = 0; i < n; ++i) {
        uint32_t c[3];
        for (int a = 0; a < 3; ++a) {
            const float f = (xyz[stride * i + a] - lo[a]) / ext * 1023.0f;
            c[a] = (uint32_t)std::min(1023.0f, std::max(0.0f, f));
        }
        code[i] = hilbert ? hilbert30(c[0], c[1], c[2]) : morton_expand10(c[0]) | (morton_expand10(c[1]) << 1) | (morton_expand10(c[2]) << 2);
    }
    std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return code[a] < code[b]; });
    return perm;
}

// K-D ORDER (round 3).  The same implicit complete binary tree as a sorted curve gives — node at level L owns a run of
// leaf << (depth - L) consecutive points — but the runs are the cells of a k-d tree: a node's points are split at the position
// the implicit layout prescribes (the left child takes as many whole leaves as it can hold) along the LONGEST axis of their
// bounding box.  Every node's box is then as tight as a box over that many points gets, whereas a run of a space-filling curve
// through a SURFACE straddles cell boundaries of the curve: its box is the union of two or three patches.  Measured on the
// dragon-shape pair at the optimum (tools/kd_sim.py, the scan's own rules): leaf boxes 0.0079 -> 0.0052 across, leaves a
// 64-query group has to scan 10.5 -> 7.0 (-> 5.5 when the queries are grouped the same way).  Ties are broken by index, so the
// order is a function of the input alone.  Only locality depends on the order, results do not.
struct KdItem { float c[3]; uint32_t idx; };  // the points themselves are moved (16 bytes each): every pass streams through memory
inline void kd_order_rec(KdItem* it, size_t n, size_t cap_leaves, size_t leaf, int spawn_depth);
inline int kd_longest_axis(const KdItem* it, size_t n) {
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) lo[a] = hi[a] = it[0].c[a];
    for (size_t i = 1; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], it[i].c[a]);
            hi[a] = std::max(hi[a], it[i].c[a]);
        }
    int ax = 0;
    if (hi[1] - lo[1] > hi[ax] - lo[ax]) ax = 1;
    if (hi[2] - lo[2] > hi[ax] - lo[ax]) ax = 2;
    return ax;
}
// compared through an order-preserving integer image of the float: a TOTAL order whatever the input holds (a NaN would make `<` on
// floats an inconsistent comparator, which std::nth_element may answer with out-of-range accesses)
inline void kd_split(KdItem* it, size_t n, size_t at, int ax) {
    auto key = [ax](const KdItem& p) {
        uint32_t u;
        std::memcpy(&u, &p.c[ax], 4);
        return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
    };
    std::nth_element(it, it + at, it + n, [&](const KdItem& a, const KdItem& b) {
        const uint32_t ka = key(a), kb = key(b);
        return ka < kb || (ka == kb && a.idx < b.idx);
    });
}
// inside a leaf: median splits down to single points, so that neighbouring positions (lanes) hold neighbouring points
inline void kd_order_fine(KdItem* it, size_t n) {
    while (n > 2) {
        const size_t half = n / 2;
        kd_split(it, n, half, kd_longest_axis(it, n));
        kd_order_fine(it, half);
        it += half;
        n -= half;
    }
}
inline std::vector<uint32_t> kd_order(const float* xyz, size_t n, size_t stride, size_t leaf, bool fine = false) {
    std::vector<uint32_t> perm(n);
    std::iota(perm.begin(), perm.end(), 0u);
    if (n <= leaf) return perm;
    size_t cap = 1;
    while (cap * leaf < n) cap *= 2;
    std::vector<KdItem> it(n);
    for (size_t i = 0; i < n; ++i) it[i] = KdItem{{xyz[stride * i], xyz[stride * i + 1], xyz[stride * i + 2]}, (uint32_t)i};
    kd_order_rec(it.data(), n, cap, leaf, n >= 100000 ? 3 : 0);
    if (fine)
        for (size_t l = 0; l * leaf < n; ++l) kd_order_fine(it.data() + l * leaf, std::min(leaf, n - l * leaf));
    for (size_t i = 0; i < n; ++i) perm[i] = it[i].idx;
    return perm;
}

}  // namespace fgoicp

#include <future>

namespace fgoicp {

inline void kd_order_rec(KdItem* it, size_t n, size_t cap_leaves, size_t leaf, int spawn_depth) {
    std::vector<std::future<void>> left;
    while (cap_leaves > 1 && n > leaf) {
        const size_t half = cap_leaves / 2 * leaf;  // what the left child holds when it is full
        if (n <= half) { cap_leaves /= 2; continue; }  // everything goes left: the right subtree stays empty
        kd_split(it, n, half, kd_longest_axis(it, n));
        if (spawn_depth > 0) left.push_back(std::async(std::launch::async, kd_order_rec, it, half, cap_leaves / 2, leaf, spawn_depth - 1));
        else kd_order_rec(it, half, cap_leaves / 2, leaf, 0);
        it += half;
        n -= half;
        cap_leaves /= 2;
        if (spawn_depth > 0) --spawn_depth;
    }
    for (auto& f : left) f.get();
}

// MIXED ORDER (mode 3): a cloud with outliers scattered through the volume (trimmed Go-ICP's input) is two populations — a surface
// sample and a sparse volume sample — and a k-d cell over both is a surface patch plus the outliers above and below it (the tails that
// make the k-d order lose there, ctx.hip).  Here the points are split by LOCAL DENSITY first: the grid level of the Hilbert code is
// chosen so that the median point shares its cell with at least 8 others, a point alone in its cell is "scattered"; the dense
// points come first in k-d order (pure surface patches), the scattered ones follow along the curve.  Fewer than 1 % scattered: plain
// k-d order.  Locality only — results do not depend on the order.
// MEASURED (profiles/r03_ab_kd_order.txt, third pass), 1M points with 20 % outliers, trimmed run: 1.40 s against 1.25 s with the plain
// Hilbert order and 1.36 s with the plain k-d order (ICP 543 / 459 / 514 ms, bounds kernel 5 640 / 5 290 / 5 250 us per launch) — a
// recorded negative: the trimmed kernels do better when a wave's queries mix both populations than when whole waves are outliers
// (the row sample of the trimmed bounds, every 32nd point, also stops being a sample of the row).  Kept as FGOICP_POINT_CURVE=3.
inline std::vector<uint32_t> mixed_order(const float* xyz, size_t n, size_t stride, size_t leaf, bool fine, size_t* n_dense_out = nullptr) {
    if (n_dense_out) *n_dense_out = n;
    if (n <= 4 * leaf) return kd_order(xyz, n, stride, leaf, fine);
    float lo[3] = {xyz[0], xyz[1], xyz[2]}, hi[3] = {xyz[0], xyz[1], xyz[2]};
    for (size_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], xyz[stride * i + a]);
            hi[a] = std::max(hi[a], xyz[stride * i + a]);
        }
    const float ext = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
    if (!(ext > 0)) return kd_order(xyz, n, stride, leaf, fine);
    std::vector<uint64_t> keyed(n);  // (code, index): one flat sort
    for (size_t i = 0; i < n; ++i) {
        uint32_t c[3];
        for (int a = 0; a < 3; ++a) c[a] = (uint32_t)std::min(1023.0f, std::max(0.0f, (xyz[stride * i + a] - lo[a]) / ext * 1023.0f));
        keyed[i] = ((uint64_t)hilbert30(c[0], c[1], c[2]) << 32) | (uint64_t)i;
    }
    std::sort(keyed.begin(), keyed.end());
    std::vector<uint32_t> code(n), by_code(n);  // by position in curve order
    for (size_t k = 0; k < n; ++k) { code[k] = (uint32_t)(keyed[k] >> 32); by_code[k] = (uint32_t)keyed[k]; }
    std::vector<uint32_t> cnt(n);  // per position of by_code: the population of its cell at the level under test
    int level = 0;
    for (int L = 10; L >= 1; --L) {
        const int sh = 3 * (10 - L);
        for (size_t b = 0; b < n;) {
            size_t e = b + 1;
            while (e < n && (code[e] >> sh) == (code[b] >> sh)) ++e;
            for (size_t k = b; k < e; ++k) cnt[k] = (uint32_t)(e - b);
            b = e;
        }
        std::vector<uint32_t> tmp(cnt);
        std::nth_element(tmp.begin(), tmp.begin() + n / 2, tmp.end());
        if (tmp[n / 2] >= 9) { level = L; break; }
    }
    if (level == 0) return kd_order(xyz, n, stride, leaf, fine);
    std::vector<uint32_t> dense, scattered;
    for (size_t k = 0; k < n; ++k) (cnt[k] <= 1 ? scattered : dense).push_back(by_code[k]);  // both stay in curve order
    if (scattered.size() * 100 < n) return kd_order(xyz, n, stride, leaf, fine);
    std::vector<float> dp(3 * dense.size());
    for (size_t i = 0; i < dense.size(); ++i)
        for (int a = 0; a < 3; ++a) dp[3 * i + a] = xyz[stride * dense[i] + a];
    const std::vector<uint32_t> kd = kd_order(dp.data(), dense.size(), 3, leaf, fine);
    std::vector<uint32_t> perm;
    perm.reserve(n);
    for (uint32_t j : kd) perm.push_back(dense[j]);
    for (uint32_t j : scattered) perm.push_back(j);
    if (n_dense_out) *n_dense_out = dense.size();
    return perm;
}

// The order of a point set for device work: 1 = Hilbert curve, 0 = Z-order, 2 = k-d order with runs of `leaf` points, 3 = mixed.
inline std::vector<uint32_t> point_order(const float* xyz, size_t n, size_t stride, size_t leaf, int mode) {
    // inside a 64-point run: median splits down to pairs, so that neighbouring lanes hold neighbouring points (their gathers share LUT
    // lines): dragon-shape bounds kernel 5240 -> 5084 us per launch, bunny shape unchanged (profiles/r03_ab_kd_order.txt)
    static const bool fine = [] { const char* e = dev_env("FGOICP_KD_FINE"); return !e || std::atoi(e) != 0; }();  // tuning knob / A-B
    return mode == 3 ? mixed_order(xyz, n, stride, leaf, fine) : mode == 2 ? kd_order(xyz, n, stride, leaf, fine) : morton_order(xyz, n, stride);
}

}  // namespace fgoicp
