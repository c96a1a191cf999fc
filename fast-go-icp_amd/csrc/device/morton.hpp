// 30-bit Morton order of a point set inside its own AABB (host).  Used for the source cloud
// (neighbouring lanes stay in neighbouring LUT voxels under any rigid motion) and for the BVH.
#pragma once
#include <algorithm>
#include <cstdint>
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

// xyz: n points with the given float stride (3 for packed xyz, 4 for float4)
inline std::vector<uint32_t> morton_order(const float* xyz, size_t n, size_t stride) {
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
        code[i] = morton_expand10(c[0]) | (morton_expand10(c[1]) << 1) | (morton_expand10(c[2]) << 2);
    }
    std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return code[a] < code[b]; });
    return perm;
}

}  // namespace fgoicp
