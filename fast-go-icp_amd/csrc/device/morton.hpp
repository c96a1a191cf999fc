// 30-bit space-filling-curve order of a point set inside its own AABB (host).  Used for the source cloud
// (neighbouring lanes stay in neighbouring LUT voxels under any rigid motion) and for the BVH.
// The curve is Hilbert's by default (every run of consecutive points is one connected patch; a Z-order
// run that straddles a power-of-two boundary is two patches far apart), FGOICP_POINT_CURVE=0 selects
// Z-order.  Only locality depends on the order, results do not.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
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
    static const bool hilbert = [] { const char* e = std::getenv("FGOICP_POINT_CURVE"); return e ? std::atoi(e) != 0 : true; }();  // tuning knob
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

}  // namespace fgoicp
