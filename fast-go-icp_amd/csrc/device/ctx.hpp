// Internal layout of the opaque fgoicp_ctx (include/fgoicp_amd.h).  Shared by the operator ABI
// (ctx.hip) and the driver ABI (solver.cpp); not installed.
#pragma once
#include <exception>
#include <new>
#include <string>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../host/math3.hpp"
#include "../host/abi_guard.hpp"
#include "bvh.hpp"
#include "kernels.hpp"

struct fgoicp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    size_t ns = 0, nt = 0;
    bool profile = false;

    // HBM-resident state
    float4* d_src = nullptr;     // ns  x {x,y,z,|p|^2}, Morton order (pristine source, registration.hpp:63)
    float4* d_tgt = nullptr;     // nt  x {x,y,z,0}, caller order (registration.hpp:61)
    float* d_lut = nullptr;      // (dx+2)(dy+2)(dz+2) floats, x fastest, replicated border
    int lut_layout = 1;          // 1: d_lut_zp is the z-paired copy (float2), 2: it is the yz-quad copy (float4)
    int source_order = 0, tree_order = 0;  // what ctx_create chose (fgoicp_ctx_get_info)
    uint32_t* d_lut_idx = nullptr;  // index LUT (geom.idx): per padded node the caller index of a nearest target point; seeds of the exact scans (FGOICP_LUT_INDEX=0: none)
    float2* d_lut_zp = nullptr;  // z-paired copy {T[o], T[o + z-slice]} for the bounds kernel (2x the bytes, half the gathers)
    fgoicp::LutGeom geom{};
    fgoicp::BvhDevice bvh_tgt;   // exact-NN tree over the target (caller indices in pts[].w)
    bool brute_force_nn = false; // FGOICP_FLAG_BRUTE_FORCE_NN: O(ns*nt) kernels instead of the tree
    std::vector<uint32_t> perm;  // device slot i holds caller point perm[i]

    // bounds-operator scratch (persistent: the reference mallocs/frees per call, registration.cu:95-149)
    int pts_per_thread = 1, nchunk = 0, max_subcubes = 0;
    double2* d_partials = nullptr;           // [max_subcubes][nchunk] {sum_ub, sum_lb}
    float *h_lb = nullptr, *h_ub = nullptr;  // pinned, device-visible result rows
    float *hd_lb = nullptr, *hd_ub = nullptr;

    // locality-sorted whole-tick path (kernels.hip, bounds_sorted_kernel): 256-point chunks.
    // Two tick slots: the host prepares / consumes one half of a round's inner BnBs while the device
    // evaluates the other (fgoicp_bounds_submit / _collect).
    struct TickSlot {
        hipStream_t stream = nullptr;            // both slots queue on the context stream: their kernels run back to back, never
                                                 // concurrently (two sorted kernels at once would thrash each other's L2 neighbourhoods)
        hipEvent_t done = nullptr;               // recorded behind the slot's last kernel; collect waits on it, not on the stream
        hipStream_t sort_stream = nullptr;       // descriptors upload + locality sort of THIS slot run here, next to the other
        hipEvent_t bounds_ev = nullptr;          // bounds kernel of this slot finished (the finalize on the side stream waits for it)
        hipEvent_t sorted_ev = nullptr;          //   slot's bounds kernel on the main stream, which then waits for sorted_ev
        fgoicp::TickGroup *d_groups = nullptr, *h_groups = nullptr, *hd_groups = nullptr;   // device / pinned staging / its device alias
        fgoicp::TickSub *d_subs = nullptr, *h_subs = nullptr, *hd_subs = nullptr;
        unsigned short* d_keys = nullptr;
        unsigned* d_ranks = nullptr;             // place of every item inside its key's bin (returned by the histogram atomic)
        unsigned *d_hist = nullptr, *d_block_sums = nullptr, *d_cursor = nullptr, *d_sorted = nullptr;
        unsigned *d_hist_xcd = nullptr, *d_xoff = nullptr;   // per-XCD histograms of the tick sort and their offsets inside a bin
        double2* d_partials = nullptr;           // [max_subcubes][nchunk1]
        double* d_cut_acc = nullptr;             // early exit (fgoicp_bounds_submit_cut): 2 running sums per evaluation, zero between windows
        float* d_row_cut = nullptr;              // ... and the threshold of every output row
        unsigned* d_cut_done = nullptr;          // ... and the cached "finished" hint per evaluation
        bool win_cut = false;                    // the window in flight carries thresholds
        float *h_lb = nullptr, *h_ub = nullptr, *hd_lb = nullptr, *hd_ub = nullptr;  // pinned results of the window in flight
        float* d_evals = nullptr;                // trimmed mode: per-point e = max(d, 0) of every output row, [vals_rows][erow]
        float *h_row_span = nullptr, *hd_row_span = nullptr;   // translation span of every output row of the window (pinned)
        unsigned *h_sort_err = nullptr, *hd_sort_err = nullptr; // pinned: set by tick_check_kernel when `sorted` is no permutation
        int win_groups = 0, win_evals = 0;       // the window in flight: groups and evaluations in the staging buffers
        int win_units = 0;                       // ... of which the first win_units * unit_m evaluations are sibling units (bounds_units_kernel)
        std::vector<fgoicp::TickSub> sub_tmp;    // packing scratch of the unit detection
        std::vector<int> unit_of;
        std::vector<float> lb, ub;
        std::vector<int> row_group;              // window-local group of every output row (packing scratch)               // results of the whole submission
        int total = 0, win_pos = 0, win_rows = 0;
        bool inflight = false;
    };
    unsigned long long* d_cut_stat = nullptr;    // items the early exit did not evaluate, since creation (device counter)
    uint64_t cut_items_offered = 0;              // items of the windows submitted with thresholds
    uint64_t cut_verify_windows = 0, cut_verify_rows = 0, cut_verify_above = 0, cut_verify_bad = 0;  // development build, FGOICP_CUT_VERIFY
    uint64_t cut_stat_base = 0;                  // value of *d_cut_stat at the last reset
    bool sorted_bounds = true;
    bool sort_xcd = true;                    // XCD-private histograms for the tick sort (cleared for good if a permutation check fails)
    bool sort_check = true;                  // verify on the device that every tick's `sorted` is a permutation
    int sort_fault_tick = 0;                 // test hook
    uint64_t sorted_ticks = 0, sort_fallbacks = 0;
    int nchunk1 = 0, max_groups = 0, cell_shift = 4;
    int chunk_pts = 256;                     // points per (subcube, chunk) work item of the sorted path
    int unit_m = 0;                          // siblings per work item (0 / 1 = off): the children of one translation node share the point loads and the rotation
    uint64_t unit_evals = 0, unit_total = 0; // evaluations that went into units / all evaluations (statistics)
    bool finalize_on_side = true;
    int small_tick_items = 4096;             // ticks of at most this many items skip the descriptor copies and the locality sort
    size_t coop_split_min = (size_t)-1;      // cooperative ICP: source clouds of at least this many points split the two scans of an iteration over the ranks (FGOICP_COOP_SPLIT_MIN); default: never —
                                             // the loop runs replicated on every rank (same bits).  Measured at 437k points on 8 ranks: a 55k-query share is a latency chain like the whole scan, 2 gathers and 4 syncs per
                                             // iteration on top: 25-44 ms of ICP per rank split against 41 ms replicated, 6.2x against 6.4x with the gathers charged
    size_t coop_split_trim_min = 262144;     // ... trimmed contexts split from this size on: a trimmed iteration is long (1.5 ms at 1M points, 85 % of it the walk) and splitting pays — 8-rank replay of the 1M trimmed run 1.84x -> 2.69x (2.45x with the 632 gathers charged)
    bool icp_seeding = true;                 // ICP passes seed their exact NN search with the previous pass's correspondences
    float4* d_chunk_cen = nullptr;           // centroid of every chunk (source frame)
    float cut_tier_level = 0.5f;             // windows with thresholds: items whose per-point term at the patch centre reaches this multiple of T / ns go first (0: one tier)
    int cut_span = 1;                        // chunks per work item in windows with thresholds (2 on sparse clouds: see bounds_item_kernel)
    float4* d_span_cen = nullptr;            // centroid of every run of cut_span chunks
    TickSlot slots[2];

    // EXTENSION: trimmed Go-ICP (sum of the `inliers` smallest per-point terms; 0 = off)
    size_t inliers = 0;
    int vals_rows = 0;                       // subcubes per window in trimmed mode (memory budget)
    size_t erow = 0;                         // floats per row of d_evals (ns rounded up to a multiple of 4, plus the row's sample)
    int trim_samp_shift = 5;                 // trimmed mode: every 2^shift-th point goes into the row's sample (0: none, two-pass selection)
    float trim_margin_sd = 1.0f;             // bracket half-width in standard deviations of a binomial sample rank
    int trim_margin = 0;                     // ... in sample ranks (set with the inlier count)
    uint32_t* d_coop = nullptr;              // cooperative ICP: {correspondence indices | bits of the minima}, coop_cap entries each (all ranks' shares)
    size_t coop_cap = 0;
    unsigned long long* d_trim_stat = nullptr;        // {rows selected, rows that fell back to two passes, bracket members}
    uint64_t trim_stat_acc[3] = {0, 0, 0};
    bool trim_ready = false;                 // trimmed-mode buffers allocated
    bool trim_skip = true;                   // exact NN only for queries that can be among the k smallest (nn_prep_kernel)
    float bounds6[6] = {0, 0, 0, 0, 0, 0};   // target_bounds as passed to fgoicp_ctx_create (they place the LUT)
    float tgt_box6[6] = {0, 0, 0, 0, 0, 0};  // the target's bounding box computed from the points (trimmed search: nn_prep_kernel)
    uint32_t* d_orig_of_slot = nullptr;      // caller index of every device slot (ties at the inlier cut)

    // exact-NN / ICP scratch, one set per lane: ICP runs on different lanes may be in flight together (ctx_icp_batch).  Lane 0 is
    // the lane of fgoicp_sse / fgoicp_icp / fgoicp_procrustes and queues on the context's main stream.
    struct IcpLane {
        hipStream_t stream = nullptr;            // main stream of the lane (SSE pass, working-cloud transforms)
        hipStream_t icp_stream = nullptr;        // side stream (correspondence + covariance pass of the NEXT iteration)
        hipEvent_t icp_ev_w = nullptr, icp_ev_b = nullptr;  // working cloud transformed / side-stream pass finished
        float4* d_work = nullptr;                // ns x {x,y,z,-}: ICP working copy (icp3d.hpp:24)
        uint32_t *d_min_bits = nullptr, *d_thr_bits = nullptr, *d_first_idx = nullptr, *d_first_idx2 = nullptr;
        double *d_bp = nullptr, *d_bp2 = nullptr, *d_bp3 = nullptr;   // per-block partial sums (sums / covariance / SSE)
        double *h_sums = nullptr, *hd_sums = nullptr;                 // pinned result of the last reduction (<= 16 doubles)
        float* d_cen = nullptr;                  // centroids {src, corr} on the device
        float *h_cen = nullptr, *hd_cen = nullptr;  // ... and their pinned host copy
        // trimmed mode
        float* d_d2 = nullptr;                   // squared correspondence distances
        float *d_nn_lb = nullptr, *d_nn_ub = nullptr, *d_nn_lb2 = nullptr, *d_nn_ub2 = nullptr;  // LUT brackets of the nearest distance (SSE pass / correspondence pass)
        uint32_t *d_sel = nullptr, *d_eq = nullptr, *d_sel_wide = nullptr, *d_sel_wide2 = nullptr;
        unsigned char* d_use = nullptr;          // inlier mask of the current Procrustes step
        float *h_trim = nullptr, *hd_trim = nullptr;   // pinned trimmed SSE
        // fused reductions of small clouds (ns <= 262144; ctx.hip icp_fused): wave sums from the scans' epilogues, final folds on the host
        double* d_wsum = nullptr;                // [groups][6] wave-level sums {query xyz, correspondence xyz} of the last correspondence scan
        double *h_wsse = nullptr, *hd_wsse = nullptr;     // pinned: [groups] wave-level sums of the last SSE scan
        double *h_covbp = nullptr, *hd_covbp = nullptr;   // pinned: [blocks][9] block partials of the covariance
        bool cov_on_host = false, sse_on_host = false;    // where the result of the last enqueued pass lands
        int cov_blocks = 0;
        // gated loop (ctx.hip lane_icp_gated): the next iteration's kernels are enqueued BEFORE their motion is known, behind a stream
        // wait on a signal word the host raises once it has written the motion into pinned memory
        uint64_t *sig_b = nullptr, *sig_a = nullptr;   // hipMallocSignalMemory, 8 bytes each
        uint64_t gate_seq = 0;                         // last value the gates have been raised to
        float *h_rt = nullptr, *hd_rt = nullptr;       // pinned: 2 slots x {R_[9], t_[3], R[9], t[3]}
        int *h_done = nullptr, *hd_done = nullptr;     // pinned: kernels enqueued behind a gate return at once when it is set
        // device-resident ICP loop (ctx.hip lane_icp_device)
        fgoicp::IcpDevState* d_icp = nullptr;    // loop state in device memory
        fgoicp::IcpHostResult *h_res = nullptr, *hd_res = nullptr;   // pinned result + progress words
        static constexpr int kRing = 8;
        hipEvent_t ev_step[kRing] = {}, ev_sse[kRing] = {};   // step kernel j done (stream B) / SSE partials of iteration j ready (stream A)
    };
    std::vector<IcpLane> lanes;
    bool icp_overlap = true;
    bool icp_fuse = true;                    // small clouds: reductions started in the scans' epilogues, folded on the host (FGOICP_ICP_FUSE=0: separate kernels)
    bool icp_gated = false;                  // FGOICP_ICP_GATED=1: small clouds, iterations pre-enqueued behind stream gates (built in round 3; measured slower, off)
    bool icp_gate_ok = false;                // stream wait-value operations work on this device (probed at context creation)
    int icp_dual_env = -1;                   // FGOICP_ICP_DUAL: one walk serves the two scans of an ICP iteration (nn_scan_dual_kernel); -1 = by cloud size
    bool icp_device = false;                 // ICP loop advanced on the device (FGOICP_ICP_DEVICE=0: the host loop, for A/B and as the bit reference)
    int icp_ahead = 2;                       // iterations the host may enqueue ahead of the device's progress

    // HIP-event profile of the bounds kernel
    std::vector<hipEvent_t> ev_start, ev_stop, ev_sel_start, ev_sel_stop;   // bounds kernel / trimmed selection kernel of the same window
    std::vector<char> ev_has_sel;
    std::vector<int> ev_evals;          // evaluations of the launch an event pair brackets (FGOICP_TICK_LOG)
    double prof_sel_ms = 0.0, prof_sel_ms_last = 0.0;
    int ev_used = 0;
    double prof_ms = 0.0;
    uint64_t prof_launches = 0, prof_subcubes = 0, prof_evals = 0;
};

namespace fgoicp {
void set_error(const std::string& s);

int ctx_bounds_multi(fgoicp_ctx* c, int G, const float* R9, const float* rot_span, const int* fix_rot, const int* offsets,
                     const float* tn4, float* lb_out, float* ub_out, const float* cut_above = nullptr);
int ctx_cut_stats(fgoicp_ctx* c, uint64_t* items_offered, uint64_t* items_cut, int reset);
int ctx_bounds_submit(fgoicp_ctx* c, int slot, int G, const float* R9, const float* rot_span, const int* fix_rot, const int* offsets, const float* tn4,
                      const int* twin = nullptr, const float* cut_above = nullptr);
int ctx_bounds_collect(fgoicp_ctx* c, int slot, float* lb_out, float* ub_out);
int ctx_set_inliers(fgoicp_ctx* c, size_t k);
int ctx_sse(fgoicp_ctx* c, const float* R9, const float* t3, float* sse_out, const uint32_t* seed_idx = nullptr);
int ctx_icp(fgoicp_ctx* c, const float* R0, const float* t0, size_t max_iter, float thr, float* sse_out, float* R_out9, float* t_out3,
            int* iters_out);
int ctx_icp_coop(fgoicp_ctx* c, int rank, int world, int (*gather)(void* dev_buf, size_t bytes_per_rank, void* user), void* user, const float* R0,
                 const float* t0, size_t max_iter, float thr, float* sse_out, float* R_out9, float* t_out3, int* iters_out);
int ctx_icp_lane(fgoicp_ctx* c, int lane, const float* R0, const float* t0, size_t max_iter, float thr, float* sse_out, float* R_out9, float* t_out3,
                 int* iters_out);
int ctx_icp_batch(fgoicp_ctx* c, int n, const float* R0s, const float* t0s, size_t max_iter, float thr, float* sse_out, float* R_out9s, float* t_out3s,
                  int* iters_out);
}  // namespace fgoicp
