// Launch-side declarations of the gfx950 kernels (kernels.hip).  Host-callable wrappers only;
// everything takes an explicit hipStream_t and never allocates or synchronises.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace fgoicp {

struct BvhView;

constexpr int kMaxBatch = 32;          // translation nodes per bounds launch (kernarg-resident)
constexpr int kBlock = 256;            // 4 wave64 per workgroup everywhere
constexpr float kSqrt3 = 1.732050807568877f;   // fgoicp/common.hpp:19
constexpr float kPi = 3.141592653589793f;      // fgoicp/common.hpp:17
constexpr float kInf = 1E+10f;                 // fgoicp/common.hpp:18

// Geometry of the nearest-squared-distance LUT (fgoicp/registration.cu:180-207).  The device
// copy is stored with a one-voxel replicated border on every side, so that CUDA's clamp
// addressing (registration.cu:226-228) needs no per-texel branch: padded[k] = T[clamp(k-1)].
struct LutGeom {
    float off_x, off_y, off_z;   // offset = -min_bound (registration.cu:202-204)
    float scale;                 // 1/resolution (registration.cu:201)
    float resolution;
    int dx, dy, dz;              // reference dims (registration.cu:186-188)
    int px, py, pz;              // padded dims = d + 2
    int quantize;                // 1: interpolation weights in 1.8 fixed point (CUDA linear filtering)
    const uint32_t* idx;         // optional (nullptr: none): per padded node the caller index of A nearest target point — the exact scans seed their
                                 // pruning bound with the distance to it (round 3, kernels.hip lut_upper_bound_d2); not read by the bounds kernels
};

// One bounds launch = one rotation node + up to kMaxBatch translation nodes, passed by value in
// the kernarg segment (the reference passes RotNode/TransNode by value too, registration.cu:111).
struct BoundsArgs {
    float R[9];        // glm::mat3 order (column-major)
    float sin_half;    // sin(rot_span * sqrt3 * pi / 2), registration.cu:42-43, hoisted to the host
    int fix_rot;
    int B;
    int out_base;      // first row of `partials` this launch writes
    int pad_;
    float4 tn[kMaxBatch];   // t.x, t.y, t.z, span
};

// bounds: partials[(out_base + b) * nchunk + chunk] = {sum_ub, sum_lb} over the chunk's points
void launch_bounds(const float4* src, int ns, const float* lut, const LutGeom& g, const BoundsArgs& a, double2* partials,
                   int nchunk, int pts_per_thread, hipStream_t s);
// Whole-tick variant: all subcubes of all rotation nodes in ONE launch, work items ordered by LUT locality
// (kernels.hip).  groups/subs are device arrays of TickGroup (48 B) / TickSub (48 B); partials is indexed
// [s * nchunk + chunk] with 256-point chunks; events (optional) bracket the bounds kernel only.
struct TickGroup {   // one rotation node
    float R[9];
    float sin_half;
    int fix_rot;
    int pad_;
};
struct TickSub {     // one EVALUATION: a translation node + its rotation node, and where its sums go
    float tx, ty, tz, span;
    int group;
    int out0;        // output row of the result (dual: of the fix_rot = 1 variant)
    int out1;        // dual only: output row of the fix_rot = 0 variant
    int dual;        // 1 = the UB task and the LB task of one rotation cube both hold this translation node in the same
                     // submission: one lookup per point, both variants of the bound formulae (registration.cu:39-58)
    float cut0;      // cut_above of out0's group (fgoicp_bounds_submit_cut; +inf = none): once the lower-bound sums of the evaluation's
    float cut1;      // finished items reach it, the remaining items are not evaluated (dual: both variants must have reached theirs)
    int pad_[2];
};
static_assert(sizeof(TickSub) == 48, "TickSub is copied in 16-byte units");
// Early exit (fgoicp_bounds_submit_cut): acc = 2 doubles per evaluation (sum of the lower-bound partials of its finished items, per
// variant; zero on entry, re-zeroed by bounds_finalize_kernel), row_cut = the threshold of every output row (written by the
// bounds kernel, applied by bounds_finalize_kernel), stat = {items not evaluated} (optional).  acc == nullptr: off.
constexpr int kCutStatSlots = 64;  // the counter is spread over this many words (one atomic per output row with skipped items)
struct TickCut {
    double* acc = nullptr;
    unsigned* done = nullptr;             // per evaluation: 1 = an item has seen the running sums at their thresholds (a cached hint, zero between windows)
    float* row_cut = nullptr;
    unsigned long long* stat = nullptr;   // [kCutStatSlots]
    const unsigned* tier_split = nullptr; // items of the first tier of `sorted` (launch_tick_sort with tier_lut): the grid walks them before the others
    int probe = 0;                        // development build, FGOICP_CUT_PROBE: 1 = the running sums are not read (nothing is ever cut), 2 = not added to, 4 = no `done` hint
};
constexpr int kTickNumKeys = 1 << 15;
void launch_tick_sort(const LutGeom& g, const float4* chunk_cen, int nchunk, const TickGroup* groups, const TickSub* subs, int nsub, int cell_shift,
                      unsigned short* keys, unsigned* ranks /* per item, like keys */, unsigned* hist /* kTickNumKeys, zero on entry and on exit */,
                      unsigned* hist_xcd /* 16 x kTickNumKeys, zero on entry and on exit (optional) */, unsigned* xoff /* 16 x kTickNumKeys (optional) */,
                      unsigned* block_sums /* 64 */, unsigned* cursor, unsigned* sorted,
                      int allow_xcd /* 0: device-scope histogram atomics */, int prefill /* 1: `sorted` is filled with 0xFFFFFFFF first, for the permutation check in launch_bounds_sorted */,
                      unsigned* check_err /* development build, A/B only: the check as a launch of its own behind the scatter */,
                      int inject_fault /* test hook */, hipStream_t s,
                      int nunits = 0, int unit_m = 1 /* sibling units: the first nunits * unit_m evaluations form nunits items per chunk (bounds_units_kernel) */,
                      const float* tier_lut = nullptr /* windows with thresholds: the plain LUT — items likely to carry much of their evaluation's lower bound are sorted
                                                         in front of the others (two tiers; cursor[kTickTierSplit] = items of the first) */,
                      float tier_level = 0.0f /* ... those whose per-point term at the patch centre reaches tier_level * T */);
constexpr int kTickTierSplit = 1 << 14;
// descriptors of a tick: pinned staging (device-visible addresses) -> device arrays, one launch
void launch_tick_upload(const TickGroup* hd_groups, TickGroup* d_groups, int ngroups, const TickSub* hd_subs, TickSub* d_subs, int nsubs, hipStream_t s);
#ifdef FGOICP_DEV_KNOBS
bool bounds_dev_variant_selected(const float2* packed_or_null, int layout, int unit_m);
#endif
bool launch_bounds_sorted(const float4* src, int ns, const float* lut, const float2* packed_or_null, int layout /* 1 z-pair, 2 yz-quad */, const LutGeom& g, int nchunk,
                          int chunk_pts /* 256 .. 2048 points per item */, const TickGroup* groups, const TickSub* subs, int nsub, const unsigned* sorted, double2* partials,
                          float* evals_or_null /* trimmed mode: row r = the per-point e = max(d, 0) of output row r */, size_t erow /* floats per row, multiple of 4 */,
                          int samp_shift /* trimmed mode: > 0 = every 2^samp_shift-th point once more in the sample behind the row (offset: ns rounded up to 64 floats) */,
                          unsigned* sort_err /* optional, host-visible: set to 1 unless `sorted` (prefilled, see launch_tick_sort) is a permutation of the items */,
                          const TickCut& cut /* early exit of evaluations whose lower bound has reached its group's cut_above */,
                          int span /* chunks per work item: `sorted` then orders nsub * ceil(nchunk / span) items (1 unless the item kernel runs: bounds_dev_variant_selected) */,
                          hipEvent_t ev_start, hipEvent_t ev_stop, hipStream_t s, int nunits = 0, int unit_m = 1);
// EXTENSION (trimmed Go-ICP): per output row the sums of ub = e*e and lb = max(e - sqrt3*span, 0)^2 over the row's k smallest e
// (one exact selection per row, kernels.hip trim_rows_kernel); row_span[r] = translation span of row r (device-readable)
// samp_shift > 0: one pass per row (trim_rows_sampled_kernel) — the bracket of the cut comes from the row's sample, `margin` sample ranks
// either side of the expected rank, verified exactly, two-pass fallback inside the kernel; stat (optional): {rows, fallbacks, members}
void launch_trim_rows(const float* evals, size_t erow, int n, int k, int rows, const float* row_span, float* out_ub, float* out_lb, hipStream_t s,
                      int samp_shift = 0, int margin = 0, unsigned long long* stat = nullptr);
// one row: out[0] (optional) = sum of the k smallest of vals[0..n), sel_info (optional) = {bits of the k-th smallest, copies of it needed}
void launch_trim_select(const float* vals, int n, int k, float* out, uint32_t* sel_info,
                        uint32_t* wide_scratch /* 64 KiB, optional: rows of n >= 32768 are then selected by the whole device */, hipStream_t s);
void launch_icp_inliers(const float4* work, const float4* tgt, const uint32_t* idx, int n, int nt, int k, float* d2, uint32_t* sel_info,
                        uint32_t* equal_count, const uint32_t* orig_of_slot, unsigned char* use, uint32_t* wide_scratch, hipStream_t s);
// out_lb[i], out_ub[i] = float(sum over chunks), fixed order → bit-reproducible
void launch_bounds_finalize(const double2* partials, int nchunk, int total, float* out_lb, float* out_ub, const TickCut& cut, hipStream_t s);

void launch_lut_build(const float4* tgt_shifted, int nt, const LutGeom& g, float* lut_padded, hipStream_t s);
// zp[o] = {lut[o], lut[o + one z-slice]}: the z-paired copy the sorted bounds kernel gathers from (kernels.hip)
void launch_lut_zpair(const float* lut_padded, const LutGeom& g, float2* zp, hipStream_t s);
void launch_lut_quad(const float* lut_padded, const LutGeom& g, float4* qd, hipStream_t s);
void launch_lut_quad_apron(const float* lut_padded, const LutGeom& g, float4* qd /* ceil(px/3)*ceil(py/2)*pz*8 quads */, hipStream_t s);
void launch_lut_quad_bricked(const float* lut_padded, const LutGeom& g, float4* qd /* ceil(px/4)*ceil(py/4)*ceil(pz/4)*64 quads */, hipStream_t s);
void launch_lut_unpad(const float* lut_padded, const LutGeom& g, float* out, hipStream_t s);
void launch_lut_search(const float* lut, const LutGeom& g, const float* q_xyz, size_t n, float* out, hipStream_t s);
void launch_lut_nodes(const float* lut_padded, const LutGeom& g, const int* xyz /* device, n node indices (clamped into the grid) */, size_t n, float* out, hipStream_t s);

// Exact nearest neighbour (brute force, tiled through LDS).
//   queries: if `apply` q_i = R*pts_i + t (fma convention) else q_i = pts_i.
//   min_bits[i] = bit pattern of min_j |q_i - tgt_j|^2 (must be pre-filled with bits(1e10f)).
void launch_fill_u32(uint32_t* p, uint32_t v, size_t n, hipStream_t s);
void launch_nn_min(const float4* pts, int n, const float4* tgt, int nt, const float* R9, const float* t3, int apply,
                   uint32_t* min_bits, hipStream_t s);
// thr_bits[i] = largest float x with sqrtf(x) == sqrtf(min_i): every target within it ties under
// glm::distance (icp3d.cu:20-25); first_idx must be pre-filled with 0x7fffffff.
void launch_nn_tie_threshold(const uint32_t* min_bits, int n, uint32_t* thr_bits, hipStream_t s);
void launch_nn_first_index(const float4* pts, int n, const float4* tgt, int nt, const uint32_t* thr_bits, uint32_t* first_idx,
                           hipStream_t s);

// Exact NN through the two-level box scan (bvh.hpp) — bit-identical to the brute-force kernels above.
//   want_index = 0: out[i] = bits(min squared distance);  1: out[i] = lowest index in the sqrt-tie set
// seed_idx (optional, may alias out): per query the caller-order index of some target point, e.g. the correspondence of the
// previous ICP pass; its distance tightens the pruning bound, the result is the same exact minimum.
void launch_nn_scan(const float4* pts, int n, const BvhView& t, const float* lut, const LutGeom& g, const float* R9, const float* t3, int apply,
                    int want_index, const float4* tgt, int nt, const uint32_t* seed_idx,
                    const float* skip_lb /* optional (trimmed): queries with skip_lb[i] > float(skip_u[0]) are left out */, const uint32_t* skip_u, uint32_t* out, hipStream_t s,
                    float4* writeback = nullptr /* optional, with apply: the moved queries are stored here (may be `pts`: kernRotateTranslateInplace folded in) */,
                    const float* rt_dev = nullptr /* optional: the motion (R[9], t[3]) is read from device memory instead of R9 / t3 */,
                    const int* done = nullptr /* optional: the kernel returns at once when *done != 0 */,
                    double* wsum = nullptr /* optional, n <= 262144: per group of 64 queries the wave-level sums of the reduction that follows the scan —
                                              index mode {sum query xyz, sum correspondence xyz} (6), distance mode the sum of the minima (1) */);
// The two scans of an ICP iteration in ONE walk (kernels.hip nn_scan_dual_kernel): set A = ptsA (moved by (RA, tA) when applyA; written back to
// `writeback`) -> lowest index of the sqrt-tie set in out_idx; set B = ptsB under (RB, tB) -> bits of the minimum in out_min.  Same results as
// launch_nn_scan with want_index = 1 / 0.  skip_*: trimmed mode, per set.  wsumA / wsumB: as launch_nn_scan's wsum.
void launch_nn_scan_dual(const float4* ptsA, const float* RA9, const float* tA3, int applyA, const float4* ptsB, const float* RB9, const float* tB3, int n, const BvhView& t,
                         const float* lut, const LutGeom& g, const float4* tgt, int nt, const uint32_t* seed_idx, const float* skip_lbA, const uint32_t* skip_uA,
                         const float* skip_lbB, const uint32_t* skip_uB, uint32_t* out_idx, uint32_t* out_min, float4* writeback, double* wsumA, double* wsumB, hipStream_t s);
// EXTENSION (trimmed Go-ICP): per query a rigorous bracket [lb, ub] of its nearest squared distance from the LUT (kernels.hip, nn_prep_kernel);
// box6 = the target's bounding box {minx,maxx,miny,maxy,minz,maxz}
void launch_nn_prep(const float4* pts, int n, const float* lut, const LutGeom& g, const float* R9, const float* t3, int apply, const float4* tgt, int nt,
                    const uint32_t* seed_idx, const float* box6, float* ub_out, float* lb_out, hipStream_t s);
void launch_lut_build_scan(const BvhView& shifted_targets, const LutGeom& g, float* scratch_padded, float* lut_padded, hipStream_t s, uint32_t* lut_idx_padded = nullptr);  // lut_idx: optional, per node the caller index of a nearest target

// deterministic double sums: out[k] = sum_i vals[i*stride + k]  (k < width <= 16)
void launch_sum_f32_as_f64(const uint32_t* bits, int n, double* block_partials, int nblocks, hipStream_t s, const int* done = nullptr);
void launch_sum_partials(const double* block_partials, int nblocks, int width, double* out, hipStream_t s);

// ICP pieces (fgoicp/icp3d.cu:30-52)
void launch_transform_inplace(float4* pts, int n, const float* R9, const float* t3, hipStream_t s);
void launch_icp_sums(const float4* work, const float4* tgt, const uint32_t* idx, int n, int nt, const unsigned char* use_or_null,
                     double* block_partials, int nblocks, hipStream_t s, const int* done = nullptr);  // width 6: sum src xyz, sum corr xyz
void launch_icp_centroids(const double* block_partials, int nblocks, int ns, float* cen_dev, float* cen_host, hipStream_t s);
void launch_icp_cov(const float4* work, const float4* tgt, const uint32_t* idx, int n, int nt, const float* cen_dev, const unsigned char* use_or_null,
                    double* block_partials, int nblocks, hipStream_t s);  // width 9: glm mat3 order


// Device-resident ICP loop (kernels.hip): the loop state of IterativeClosestPoint3D::run (icp3d.cu:88-107) in device memory,
// advanced by icp_step_kernel; the scans read their motion from it (R/t at float offset 0, Rn/tn at 12).
struct IcpDevState {
    float R[9], t[3];            // composed transform of the current iteration (icp3d.cu:101-102)
    float Rn[9], tn[3];          // (R_, t_) of the last Procrustes step: the move of the working cloud (:100)
    float last_R[9], last_t[3];  // :97-98
    float sse, last_sse;
    int iters, done, max_iter;
    float thr;
};
struct IcpHostResult {           // pinned, written by the step kernel
    float sse, R[9], t[3];
    int iters;
    int iters_done;              // progress: iterations whose Procrustes step has run
    int done;                    // set after everything above
};
void launch_icp_init(IcpDevState* st, const float* R9, const float* t3, int max_iter, float thr, IcpHostResult* res, hipStream_t s);
void launch_icp_step(IcpDevState* st, const double* bp_cov, int nb_cov, const double* bp_sse, int nb_sse, const float* cen, IcpHostResult* res, hipStream_t s);
// icp_cov with the centroid kernel folded in (same bits); cen_out receives the six centroid components
// sums_bp: block partials of icp_sums_kernel (from_waves = 0) or the per-wave sums of the scan's epilogue (from_waves = their count)
void launch_icp_cov_cen(const float4* work, const float4* tgt, const uint32_t* idx, int n, int nt, const double* sums_bp, int sums_nblocks, int from_waves,
                        float* cen_out, double* block_partials, int nblocks, hipStream_t s, const int* done = nullptr);

int reduce_blocks_for(int n);

}  // namespace fgoicp
