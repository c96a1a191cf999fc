// Exact nearest-neighbour search over a point set: an implicit, heap-ordered bounding-volume tree
// over the Morton-sorted points (leaves of kBvhLeaf points).  The device code (kernels.hip, box_scan)
// uses two of its levels — the leaves and the "super-leaves" 5 levels up — as a flat two-level scan.
//
// It replaces the reference's brute-force loops (fgoicp/registration.cu:162-174, :258-278,
// fgoicp/icp3d.cu:11-28) with a search that returns BIT-IDENTICAL results: every candidate distance
// is evaluated with the same fp32 expression as the brute force (dist_sq, kernels.hip), and a
// subtree is skipped only when its box is provably farther than the incumbent — the fp32 box
// distance is shrunk by 1e-6 relative, which covers the <=4e-7 relative rounding of both the box
// distance and the point distances.  min is order-independent, so pruning cannot change the value.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

namespace fgoicp {

#ifndef FGOICP_BVH_LEAF   // development builds (tools/ab_bvh_leaf.sh): points per leaf
#define FGOICP_BVH_LEAF 32
#endif
constexpr int kBvhLeaf = FGOICP_BVH_LEAF;  // big leaves: leaf points are throughput work, tree levels are latency

// Device view.  Node i (heap order: children 2i+1, 2i+2) has box[2i] = {lo.xyz, -}, box[2i+1] = {hi.xyz, -}.
// Leaves are the last level: leaf l = node first_leaf + l holds pts[kBvhLeaf*l .. kBvhLeaf*(l+1)) = {x, y, z, bits(original index)};
// padding points sit at +FLT_MAX (distance +inf), empty leaves have an inverted box (distance +inf).
struct BvhView {
    const float4* box;
    const float4* pts;
    const float4* slab;  // optional (nullptr: none), 2 per leaf: {n.x, n.y, n.z, a}, {b, 0, 0, 0} — every point p of the leaf has a <= n.p <= b, |n| <= 1
    int depth;       // leaves live at this depth (root = 0)
    int first_leaf;  // (1 << depth) - 1
};

struct BvhHost {
    std::vector<float4> box;
    std::vector<float4> pts;
    std::vector<float4> slab;  // 2 per leaf (see BvhView); empty = none
    int depth = 0;
    int first_leaf = 0;
};

// Builds the tree over n points (xyz float4, w ignored); original indices are the positions in `p`.
// `order`: empty = computed here (and returned through it); otherwise the order to use — a second tree over the same points up to a
// common shift (the LUT build's) is as tight under the first one's order and skips the sort.
BvhHost bvh_build_host(const float4* p, size_t n, std::vector<uint32_t>* order = nullptr);
bool bvh_kd_order();  // leaves = k-d cells (default) or runs of the space-filling curve (FGOICP_BVH_ORDER=0)

struct BvhDevice {
    float4* box = nullptr;
    float4* pts = nullptr;
    float4* slab = nullptr;
    int depth = 0, first_leaf = 0;
    BvhView view() const { return BvhView{box, pts, slab, depth, first_leaf}; }
};
hipError_t bvh_upload(const BvhHost& h, BvhDevice* d);
void bvh_free(BvhDevice* d);

}  // namespace fgoicp
