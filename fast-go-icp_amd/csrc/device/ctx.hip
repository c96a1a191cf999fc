// Operator-level C ABI (include/fgoicp_amd.h): one context = one HIP device + one stream + all
// device memory of icp::Registration / icp::NearestNeighborLUT / icp::IterativeClosestPoint3D.
// No CPU fallback: without a HIP device fgoicp_ctx_create fails with FGOICP_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <limits>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/fgoicp_amd.h"
#include "../host/math3.hpp"
#include "../host/knobs.hpp"
#include "ctx.hpp"
#include "kernels.hpp"
#include "morton.hpp"

namespace fgoicp {

thread_local std::string g_last_error;
void set_error(const std::string& s) { g_last_error = s; }

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            set_error(std::string(#expr) + " failed: " + hipGetErrorString(e_));                        \
            return e_ == hipErrorOutOfMemory ? FGOICP_ERR_OOM : FGOICP_ERR_HIP;                         \
        }                                                                                              \
    } while (0)

// Waiting for the device with the host's latency in mind.  hipEventSynchronize / hipStreamSynchronize park the thread in the runtime (the wake-up
// costs 5-8 us on this image); the ICP loop pays that twice per iteration and an iteration is 45-55 us at 40k points.  These poll the event /
// stream for a bounded time first (an iteration's kernels are tens of microseconds) and fall back to the blocking call for long waits.
// MEASURED (profiles/r04_ab_spin_sync.txt): nothing — 54.7-55.9 us per ICP iteration blocking against 54.7-60.3 polling, every bench leg within noise: the
// runtime's own wait is not what an iteration's 18-us host turn-around consists of.  OFF (0) by default; FGOICP_SPIN_SYNC_US=<us> is the development knob.
static const int g_spin_sync_us = [] { const char* e = dev_env("FGOICP_SPIN_SYNC_US"); return e ? std::atoi(e) : 0; }();
static hipError_t wait_event(hipEvent_t ev) {
    if (g_spin_sync_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; ++spins) {
            const hipError_t q = hipEventQuery(ev);
            if (q != hipErrorNotReady) return q;
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
            if ((spins & 15u) == 15u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(g_spin_sync_us)) break;
        }
    }
    return hipEventSynchronize(ev);
}
static hipError_t wait_stream(hipStream_t st) {
    if (g_spin_sync_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; ++spins) {
            const hipError_t q = hipStreamQuery(st);
            if (q != hipErrorNotReady) return q;
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
            if ((spins & 15u) == 15u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(g_spin_sync_us)) break;
        }
    }
    return hipStreamSynchronize(st);
}

int ctx_flush_profile(fgoicp_ctx* c) {
    for (int i = 0; i < c->ev_used; ++i) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_start[i], c->ev_stop[i]));
        c->prof_ms += ms;
        static const bool tick_log = dev_env("FGOICP_TICK_LOG") != nullptr;  // debugging aid: one line per bounds launch on stderr
        if (tick_log && i < (int)c->ev_evals.size()) std::fprintf(stderr, "[tick] evals %d us %.1f\n", c->ev_evals[i], ms * 1e3);
        if (c->ev_has_sel[i]) {  // trimmed mode: the selection kernel of the same window (side stream)
            HIPCHK(hipEventSynchronize(c->ev_sel_stop[i]));
            HIPCHK(hipEventElapsedTime(&ms, c->ev_sel_start[i], c->ev_sel_stop[i]));
            c->prof_sel_ms += ms;
            c->ev_has_sel[i] = 0;
        }
    }
    c->ev_used = 0;
    return FGOICP_OK;
}

// -------------------------------------------------------------------------------------------
// Registration::compute_sse_error(RotNode&, vector<TransNode>&, bool, StreamPool&) for G groups,
// locality-sorted whole-tick path: descriptors -> device, sort the (subcube, chunk) items by LUT
// cell, one bounds launch, one finalize, one host sync per window of <= max_subcubes subcubes.
// -------------------------------------------------------------------------------------------
namespace {
struct TickTiming {  // FGOICP_TIMING=1: where a tick's wall time goes (host side), printed at context destruction
    bool on = dev_env("FGOICP_TIMING") != nullptr;
    double pack = 0, enqueue = 0, wait = 0, copyout = 0;
    uint64_t ticks = 0;
};
TickTiming g_tt;
constexpr float kNoCut = std::numeric_limits<float>::infinity();  // TickSub::cut0 / cut1 of a group without a threshold
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

// Device half of a window whose descriptors sit in the slot's staging buffers (sl.win_groups groups, sl.win_evals evaluations,
// sl.win_rows output rows): descriptors -> device, locality sort of the (evaluation, chunk) items, one bounds launch, one
// finalize (or, trimmed, one selection).  Called again by tick_wait_window if the sort check failed.
static int tick_launch_window(fgoicp_ctx* c, fgoicp_ctx::TickSlot& sl) {
    const int ng = sl.win_groups, neval = sl.win_evals, rows = sl.win_rows;
    // A small tick (the tail of a round: a few long-running tasks left, or one of many ranks) is pure latency: its bounds
    // kernel reads the descriptors straight from the pinned staging buffers and takes the items in submission order —
    // two copies and four sort launches fewer on the critical path.  Results do not depend on the item order.
    const int um = sl.win_units > 0 ? c->unit_m : 1;
    // windows with thresholds: an item may span several chunks (fewer workgroups to dispatch for the items that end early)
    int span = sl.win_cut ? c->cut_span : 1;
#ifdef FGOICP_DEV_KNOBS
    if (span > 1 && bounds_dev_variant_selected(c->d_lut_zp, c->lut_layout, c->unit_m)) span = 1;  // round 3's kernels take one chunk per item
#endif
    const int per_eval = (c->nchunk1 + span - 1) / span;  // work items per evaluation
    bool tiers = sl.win_cut && c->cut_tier_level > 0.0f;   // ... and its items in two tiers, the heavy ones first (launch_tick_sort)
#ifdef FGOICP_DEV_KNOBS
    if (tiers && bounds_dev_variant_selected(c->d_lut_zp, c->lut_layout, c->unit_m)) tiers = false;
#endif
    const size_t nitems = (size_t)(neval - sl.win_units * (um - 1)) * (size_t)per_eval;
    const bool small = nitems <= (size_t)c->small_tick_items;
    const TickGroup* dev_groups = small ? sl.hd_groups : sl.d_groups;
    const TickSub* dev_subs = small ? sl.hd_subs : sl.d_subs;
    *sl.h_sort_err = 0u;
    unsigned* fused_err = !small && c->sort_check ? sl.hd_sort_err : nullptr;  // the bounds kernel checks the permutation it walks
    if (!small) {
        // descriptors + locality sort on the slot's side stream (overlaps the other slot's bounds kernel); the main stream joins behind it
        static const int upload_kernel = [] { const char* e = dev_env("FGOICP_UPLOAD_KERNEL"); return e ? std::atoi(e) : 1; }();  // tuning knob: 0 = two hipMemcpyAsync
        if (upload_kernel) {
            launch_tick_upload(sl.hd_groups, sl.d_groups, ng, sl.hd_subs, sl.d_subs, neval, sl.sort_stream);
        } else {
            HIPCHK(hipMemcpyAsync(sl.d_groups, sl.h_groups, sizeof(TickGroup) * ng, hipMemcpyHostToDevice, sl.sort_stream));
            HIPCHK(hipMemcpyAsync(sl.d_subs, sl.h_subs, sizeof(TickSub) * neval, hipMemcpyHostToDevice, sl.sort_stream));
        }
        ++c->sorted_ticks;
        const int fault = c->sort_fault_tick && c->sorted_ticks == (uint64_t)c->sort_fault_tick;
        // FGOICP_SEPARATE_CHECK=1 (development build): the permutation check as round 3's launch behind the scatter instead of inside the bounds kernel
        static const bool separate_check = [] { const char* e = dev_env("FGOICP_SEPARATE_CHECK"); return e && std::atoi(e) != 0; }();
        launch_tick_sort(c->geom, span > 1 ? c->d_span_cen : c->d_chunk_cen, per_eval, sl.d_groups, sl.d_subs, neval, c->cell_shift, sl.d_keys, sl.d_ranks, sl.d_hist, sl.d_hist_xcd, sl.d_xoff, sl.d_block_sums, sl.d_cursor, sl.d_sorted,
                         c->sort_xcd ? 1 : 0, c->sort_check ? 1 : 0, separate_check && c->sort_check ? sl.hd_sort_err : nullptr, fault, sl.sort_stream, sl.win_units, um,
                         tiers ? c->d_lut : nullptr, c->cut_tier_level / (float)c->ns);
        if (separate_check) fused_err = nullptr;
        HIPCHK(hipEventRecord(sl.sorted_ev, sl.sort_stream));
        HIPCHK(hipStreamWaitEvent(sl.stream, sl.sorted_ev, 0));
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->profile) {
        if (c->ev_used == (int)c->ev_start.size()) {  // drain both slots before recycling events
            HIPCHK(hipStreamSynchronize(c->stream));
            int rc = ctx_flush_profile(c);
            if (rc) return rc;
        }
        e0 = c->ev_start[c->ev_used];
        e1 = c->ev_stop[c->ev_used];
        if ((int)c->ev_evals.size() <= c->ev_used) c->ev_evals.resize(c->ev_start.size(), 0);
        c->ev_evals[c->ev_used] = neval;
        c->ev_used++;
        c->prof_launches++;
        c->prof_subcubes += rows;
        c->prof_evals += neval;  // a twin pair is two subcubes and one evaluation
    }
    TickCut cut;  // early exit of evaluations that have reached their group's threshold (fgoicp_bounds_submit_cut)
    if (sl.win_cut) { cut.acc = sl.d_cut_acc; cut.done = sl.d_cut_done; cut.row_cut = sl.d_row_cut; cut.stat = c->d_cut_stat; }
    if (tiers && !small) cut.tier_split = sl.d_cursor + kTickTierSplit;
    static const int cut_probe = [] { const char* e = dev_env("FGOICP_CUT_PROBE"); return e ? std::atoi(e) : 0; }();  // measurement of the early exit's own cost (tools/op_bench.py)
    cut.probe = cut_probe;
    const bool cut_on = launch_bounds_sorted(c->d_src, (int)c->ns, c->d_lut, c->d_lut_zp, c->lut_layout, c->geom, c->nchunk1, c->chunk_pts, dev_groups, dev_subs, neval, small ? nullptr : sl.d_sorted,
                                             sl.d_partials, c->inliers ? sl.d_evals : nullptr, c->erow, c->trim_samp_shift, fused_err, cut, span, e0, e1, sl.stream, sl.win_units, um);
    if (cut_on) c->cut_items_offered += (size_t)neval * c->nchunk1;  // (counted in chunks, like the skipped ones: bounds_finalize_kernel)
    // the per-subcube sums run on the slot's side stream, so the main stream holds nothing but bounds kernels back to back
    hipStream_t fin = c->finalize_on_side ? sl.sort_stream : sl.stream;
    if (fin != sl.stream) {
        HIPCHK(hipEventRecord(sl.bounds_ev, sl.stream));
        HIPCHK(hipStreamWaitEvent(fin, sl.bounds_ev, 0));
    }
    if (c->inliers) {  // trimmed: per row one selection of the k smallest e, both sums from it
        const int pi = e0 ? c->ev_used - 1 : -1;
        if (pi >= 0) { HIPCHK(hipEventRecord(c->ev_sel_start[pi], fin)); c->ev_has_sel[pi] = 1; }
        launch_trim_rows(sl.d_evals, c->erow, (int)c->ns, (int)c->inliers, rows, sl.hd_row_span, sl.hd_ub, sl.hd_lb, fin, c->trim_samp_shift, c->trim_margin, c->d_trim_stat);
        if (pi >= 0) HIPCHK(hipEventRecord(c->ev_sel_stop[pi], fin));
    } else
        launch_bounds_finalize(sl.d_partials, c->nchunk1, rows, sl.hd_lb, sl.hd_ub, cut_on ? cut : TickCut(), fin);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(sl.done, fin));
    return FGOICP_OK;
}

// Enqueues the window [pos, pos+rows) of a submission on its slot: packs the descriptors (host), then tick_launch_window.
// Returns the window end.
static int tick_enqueue_window(fgoicp_ctx* c, fgoicp_ctx::TickSlot& sl, int G, const float* R9, const float* rot_span, const int* fix_rot,
                               const int* offsets, const float* tn4, const int* twin, const float* cut_above, int pos, int* end_out) {
    const double t0 = g_tt.on ? now_s() : 0;
    int g = 0;
    while (g < G && offsets[g + 1] <= pos) ++g;
    const int cap = c->inliers ? std::min(c->max_subcubes, c->vals_rows) : c->max_subcubes;  // subcubes per window
    int end = pos, ng = 0;
    sl.row_group.resize((size_t)cap);
    for (int gg = g; gg < G && ng < c->max_groups && end - pos < cap; ++gg) {
        TickGroup& tg = sl.h_groups[ng];
        std::memcpy(tg.R, R9 + 9 * gg, sizeof(tg.R));
        const float half_angle = rot_span[gg] * kSqrt3 * kPi / 2.0f;  // registration.cu:42
        tg.sin_half = std::sin(half_angle);
        tg.fix_rot = fix_rot[gg] ? 1 : 0;
        tg.pad_ = gg;  // the submission's group index (twin validation below)
        const int first = std::max(offsets[gg], pos);
        const int last = std::min(offsets[gg + 1], pos + cap);
        for (int i = first; i < last; ++i) sl.row_group[(size_t)(i - pos)] = ng;
        end = std::max(end, last);
        ++ng;
        if (last < offsets[gg + 1]) break;  // window full in the middle of a group
    }
    const int rows = end - pos;  // output rows of this window
    *end_out = end;
    if (rows <= 0) return FGOICP_OK;
    // evaluations: one per output row, or one per twin pair (both inside this window, same rotation, same translation node,
    // one in a fix_rot group and one not — checked here, the caller's hint is not trusted)
    int neval = 0;
    for (int r = 0; r < rows; ++r) {
        const int i = pos + r;
        const int gi = sl.row_group[(size_t)r];
        sl.h_row_span[r] = tn4[4 * (size_t)i + 3];
        int j = twin ? twin[i] : -1;
        if (j >= pos && j < end && j != i && twin[j] == i) {
            const int gj = sl.row_group[(size_t)(j - pos)];
            const TickGroup &a = sl.h_groups[gi], &b = sl.h_groups[gj];
            const bool ok = a.fix_rot != b.fix_rot && a.sin_half == b.sin_half && std::memcmp(a.R, b.R, sizeof(a.R)) == 0 &&
                            std::memcmp(tn4 + 4 * (size_t)i, tn4 + 4 * (size_t)j, 4 * sizeof(float)) == 0;
            if (ok) {
                if (j < i) continue;  // evaluated with its twin
                TickSub& ts = sl.h_subs[neval++];
                ts.tx = tn4[4 * (size_t)i]; ts.ty = tn4[4 * (size_t)i + 1]; ts.tz = tn4[4 * (size_t)i + 2]; ts.span = tn4[4 * (size_t)i + 3];
                ts.group = gi;
                ts.out0 = a.fix_rot ? r : j - pos;  // the fix_rot = 1 variant belongs to the fix_rot group's row
                ts.out1 = a.fix_rot ? j - pos : r;
                ts.dual = 1;
                const float ca = cut_above ? cut_above[a.pad_] : kNoCut, cb = cut_above ? cut_above[b.pad_] : kNoCut;  // (pad_: the submission's group index)
                ts.cut0 = a.fix_rot ? ca : cb;
                ts.cut1 = a.fix_rot ? cb : ca;
                ts.pad_[0] = ts.pad_[1] = 0;
                continue;
            }
        }
        TickSub& ts = sl.h_subs[neval++];
        ts.tx = tn4[4 * (size_t)i]; ts.ty = tn4[4 * (size_t)i + 1]; ts.tz = tn4[4 * (size_t)i + 2]; ts.span = tn4[4 * (size_t)i + 3];
        ts.group = gi;
        ts.out0 = r;
        ts.out1 = r;
        ts.dual = 0;
        ts.cut0 = ts.cut1 = cut_above ? cut_above[sl.h_groups[gi].pad_] : kNoCut;
        ts.pad_[0] = ts.pad_[1] = 0;
    }
    sl.win_cut = cut_above != nullptr && !c->inliers;
    for (int k = 0; k < ng; ++k) sl.h_groups[k].pad_ = 0;
    // Sibling units (bounds_units_kernel): the eight children of a translation node carry one queue key (fgoicp.cpp:157-168), so the
    // inner BnB pops them together and they sit next to each other here.  A run of 8 evaluations of one group, one span and one
    // kind whose centres are the 8 corners of a cube of side 2 * span is taken as an octet (checked, not assumed) and moved to
    // the front of the descriptor list, as 8 / unit_m units; everything else follows as one-sibling items.
    sl.win_units = 0;
    if (c->unit_m > 1 && neval >= 8) {
        const int M = c->unit_m;
        std::vector<TickSub>& tmp = sl.sub_tmp;
        tmp.assign(sl.h_subs, sl.h_subs + neval);
        std::vector<int>& mark = sl.unit_of;
        mark.assign((size_t)neval, 0);
        int nocts = 0;
        for (int i = 0; i + 8 <= neval;) {
            const TickSub& a = tmp[(size_t)i];
            bool ok = c->inliers || !a.dual;   // the untrimmed kernel keeps its accumulators for single-kind, non-dual units
            float lo[3] = {a.tx, a.ty, a.tz}, hi[3] = {a.tx, a.ty, a.tz};
            for (int j = 1; j < 8 && ok; ++j) {
                const TickSub& b = tmp[(size_t)(i + j)];
                ok = b.group == a.group && b.span == a.span && b.dual == a.dual;
                lo[0] = std::min(lo[0], b.tx); lo[1] = std::min(lo[1], b.ty); lo[2] = std::min(lo[2], b.tz);
                hi[0] = std::max(hi[0], b.tx); hi[1] = std::max(hi[1], b.ty); hi[2] = std::max(hi[2], b.tz);
            }
            unsigned corners = 0;
            for (int j = 0; j < 8 && ok; ++j) {
                const TickSub& b = tmp[(size_t)(i + j)];
                const float v[3] = {b.tx, b.ty, b.tz};
                unsigned bits = 0;
                for (int ax = 0; ax < 3; ++ax) {
                    ok = ok && hi[ax] - lo[ax] == 2.0f * a.span && (v[ax] == lo[ax] || v[ax] == hi[ax]);
                    bits |= (v[ax] == hi[ax] ? 1u : 0u) << ax;
                }
                corners |= 1u << bits;
            }
            if (ok && corners == 0xFFu) {
                for (int j = 0; j < 8; ++j) mark[(size_t)(i + j)] = 1;
                ++nocts;
                i += 8;
            } else {
                ++i;
            }
        }
        if (nocts > 0) {
            int w = 0;
            for (int i = 0; i < neval; ++i) if (mark[(size_t)i]) sl.h_subs[w++] = tmp[(size_t)i];
            for (int i = 0; i < neval; ++i) if (!mark[(size_t)i]) sl.h_subs[w++] = tmp[(size_t)i];
            sl.win_units = nocts * (8 / M);
        }
        c->unit_evals += (uint64_t)nocts * 8;
    }
    c->unit_total += (uint64_t)neval;
    const double t1 = g_tt.on ? now_s() : 0;
    sl.win_pos = pos;
    sl.win_rows = rows;
    sl.win_groups = ng;
    sl.win_evals = neval;
    const int rc = tick_launch_window(c, sl);
    if (g_tt.on) { g_tt.pack += t1 - t0; g_tt.enqueue += now_s() - t1; g_tt.ticks++; }
    return rc;
}

static int tick_wait_window(fgoicp_ctx* c, fgoicp_ctx::TickSlot& sl) {
    const double t2 = g_tt.on ? now_s() : 0;
    HIPCHK(wait_event(sl.done));
    if (*sl.h_sort_err) {
        // The tick's `sorted` was not a permutation: the XCD-private histogram (workgroup-scope atomics, see kernels.hip) is
        // not a single point of coherence on this device.  Switch this context to device-scope atomics for good and repeat
        // the window (its descriptors are still in the staging buffers); a second failure is an error.
        c->sort_xcd = false;
        c->sort_fallbacks++;
        HIPCHK(hipStreamSynchronize(sl.sort_stream));
        HIPCHK(hipMemsetAsync(sl.d_hist, 0, sizeof(unsigned) * kTickNumKeys, sl.sort_stream));   // state a failed sort may have left
        HIPCHK(hipMemsetAsync(sl.d_hist_xcd, 0, sizeof(unsigned) * 16 * kTickNumKeys, sl.sort_stream));
        int rc = tick_launch_window(c, sl);
        if (rc) return rc;
        HIPCHK(hipEventSynchronize(sl.done));
        if (*sl.h_sort_err) { set_error("tick sort did not produce a permutation of the work items (device-scope atomics)"); return FGOICP_ERR_HIP; }
    }
#ifdef FGOICP_DEV_KNOBS
    // FGOICP_CUT_VERIFY=1 (development build): every window that carried thresholds is evaluated once more WITHOUT them and each row is checked against
    // the contract of fgoicp_bounds_submit_cut — at or above its threshold T in the exact evaluation: {T, T} was reported; below: the exact bits.
    // The answers handed on are the first run's, so the search goes on as it would; the tally is printed when the context is destroyed.
    const bool cut_verify = [] { const char* e = dev_env("FGOICP_CUT_VERIFY"); return e && std::atoi(e) != 0; }();  // (read per window: a test toggles it)
    if (cut_verify && sl.win_cut) {
        const int rows = sl.win_rows;
        std::vector<float> lb(sl.h_lb, sl.h_lb + rows), ub(sl.h_ub, sl.h_ub + rows);
        sl.win_cut = false;
        int rc = tick_launch_window(c, sl);
        sl.win_cut = true;
        if (rc) return rc;
        HIPCHK(hipEventSynchronize(sl.done));
        if (*sl.h_sort_err) { set_error("FGOICP_CUT_VERIFY: the repeated window's sort failed"); return FGOICP_ERR_HIP; }
        for (int e = 0; e < sl.win_evals; ++e) {
            const TickSub& ts = sl.h_subs[e];
            for (int v = 0; v < (ts.dual ? 2 : 1); ++v) {
                const int r = v ? ts.out1 : ts.out0;
                const float T = v ? ts.cut1 : ts.cut0;
                const bool above = sl.h_lb[r] >= T;
                const bool ok = above ? (lb[(size_t)r] == T && ub[(size_t)r] == T)
                                      : (std::memcmp(&lb[(size_t)r], &sl.h_lb[r], 4) == 0 && std::memcmp(&ub[(size_t)r], &sl.h_ub[r], 4) == 0);
                c->cut_verify_rows++;
                c->cut_verify_above += above ? 1 : 0;
                if (!ok) {
                    if (c->cut_verify_bad++ < 5)
                        std::fprintf(stderr, "[fgoicp cut verify] row %d: threshold %.9g, exact {%.9g, %.9g}, reported {%.9g, %.9g}\n", r, (double)T, (double)sl.h_lb[r], (double)sl.h_ub[r],
                                     (double)lb[(size_t)r], (double)ub[(size_t)r]);
                }
            }
        }
        c->cut_verify_windows++;
        std::memcpy(sl.h_lb, lb.data(), sizeof(float) * rows);
        std::memcpy(sl.h_ub, ub.data(), sizeof(float) * rows);
    }
#endif
    const double t3 = g_tt.on ? now_s() : 0;
    std::memcpy(sl.lb.data() + sl.win_pos, sl.h_lb, sizeof(float) * sl.win_rows);
    std::memcpy(sl.ub.data() + sl.win_pos, sl.h_ub, sizeof(float) * sl.win_rows);
    if (g_tt.on) { g_tt.wait += t3 - t2; g_tt.copyout += now_s() - t3; }
    return FGOICP_OK;
}

// fgoicp_bounds_submit: all windows but the last are completed here, the last one stays in flight.
int ctx_bounds_submit(fgoicp_ctx* c, int slot, int G, const float* R9, const float* rot_span, const int* fix_rot, const int* offsets,
                      const float* tn4, const int* twin, const float* cut_above) {
    HIPCHK(hipSetDevice(c->device));
    fgoicp_ctx::TickSlot& sl = c->slots[slot];
    if (sl.inflight) { set_error("fgoicp_bounds_submit: slot still in flight (collect it first)"); return FGOICP_ERR_INVALID_ARG; }
    sl.total = offsets[G];
    sl.lb.resize(sl.total);
    sl.ub.resize(sl.total);
    sl.win_rows = 0;
    int pos = 0;
    while (pos < sl.total) {
        int end = pos;
        int rc = tick_enqueue_window(c, sl, G, R9, rot_span, fix_rot, offsets, tn4, twin, cut_above, pos, &end);
        if (rc) return rc;
        if (end <= pos) break;
        pos = end;
        if (pos < sl.total) {  // more windows follow: this one has to be drained first (its buffers are reused)
            rc = tick_wait_window(c, sl);
            if (rc) return rc;
            sl.win_rows = 0;
        }
    }
    sl.inflight = true;
    return FGOICP_OK;
}

// Items (evaluation x chunk of source points) of the windows submitted with thresholds since the last reset, and how many of
// them the early exit did not evaluate.  Call between submissions (it synchronises the context's streams).
int ctx_cut_stats(fgoicp_ctx* c, uint64_t* items_offered, uint64_t* items_cut, int reset) {
    HIPCHK(hipSetDevice(c->device));
    unsigned long long now = 0;
    if (c->d_cut_stat) {
        unsigned long long part[kCutStatSlots];
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(part, c->d_cut_stat, sizeof(part), hipMemcpyDeviceToHost));
        for (unsigned long long v : part) now += v;
    }
    if (items_offered) *items_offered = c->cut_items_offered;
    if (items_cut) *items_cut = (uint64_t)now - c->cut_stat_base;
    if (reset) { c->cut_items_offered = 0; c->cut_stat_base = (uint64_t)now; }
    return FGOICP_OK;
}

int ctx_bounds_collect(fgoicp_ctx* c, int slot, float* lb_out, float* ub_out) {
    HIPCHK(hipSetDevice(c->device));
    fgoicp_ctx::TickSlot& sl = c->slots[slot];
    if (!sl.inflight) { set_error("fgoicp_bounds_collect: nothing submitted on this slot"); return FGOICP_ERR_INVALID_ARG; }
    if (sl.win_rows > 0) {
        int rc = tick_wait_window(c, sl);
        if (rc) return rc;
    }
    sl.inflight = false;
    std::memcpy(lb_out, sl.lb.data(), sizeof(float) * sl.total);
    std::memcpy(ub_out, sl.ub.data(), sizeof(float) * sl.total);
    if (c->profile && !c->slots[0].inflight && !c->slots[1].inflight) {
        int rc = ctx_flush_profile(c);
        if (rc) return rc;
    }
    return FGOICP_OK;
}

static int ctx_bounds_multi_sorted(fgoicp_ctx* c, int G, const float* R9, const float* rot_span, const int* fix_rot, const int* offsets,
                                   const float* tn4, float* lb_out, float* ub_out, const float* cut_above) {
    int rc = ctx_bounds_submit(c, 0, G, R9, rot_span, fix_rot, offsets, tn4, nullptr, cut_above);
    if (rc) return rc;
    return ctx_bounds_collect(c, 0, lb_out, ub_out);
}

// -------------------------------------------------------------------------------------------
// Registration::compute_sse_error(RotNode&, vector<TransNode>&, bool, StreamPool&) for G groups.
// -------------------------------------------------------------------------------------------
int ctx_bounds_multi(fgoicp_ctx* c, int G, const float* R9, const float* rot_span, const int* fix_rot, const int* offsets,
                     const float* tn4, float* lb_out, float* ub_out, const float* cut_above) {
    HIPCHK(hipSetDevice(c->device));
    if (c->sorted_bounds) return ctx_bounds_multi_sorted(c, G, R9, rot_span, fix_rot, offsets, tn4, lb_out, ub_out, cut_above);  // (the per-node path below ignores cut_above: exact rows)
    if (c->inliers) { set_error("trimmed bounds need the sorted bounds path (FGOICP_BOUNDS_SORTED=0 is set)"); return FGOICP_ERR_INVALID_ARG; }
    struct Piece { int g, pos, B; };
    std::vector<Piece> pieces;
    for (int g = 0; g < G; ++g)
        for (int pos = offsets[g]; pos < offsets[g + 1]; pos += kMaxBatch) pieces.push_back({g, pos, std::min(kMaxBatch, offsets[g + 1] - pos)});
    size_t pi = 0;
    while (pi < pieces.size()) {
        // one window = at most max_subcubes rows of `partials`, one finalize, one host sync
        const int first = pieces[pi].pos;
        int rows = 0;
        while (pi < pieces.size() && rows + pieces[pi].B <= c->max_subcubes) {
            const Piece& pc = pieces[pi];
            BoundsArgs a;
            std::memcpy(a.R, R9 + 9 * pc.g, sizeof(a.R));
            const float half_angle = rot_span[pc.g] * kSqrt3 * kPi / 2.0f;  // registration.cu:42
            a.sin_half = std::sin(half_angle);
            a.fix_rot = fix_rot[pc.g] ? 1 : 0;
            a.B = pc.B;
            a.out_base = rows;
            a.pad_ = 0;
            std::memcpy(a.tn, tn4 + 4 * (size_t)pc.pos, sizeof(float) * 4 * pc.B);
            const bool prof = c->profile;
            if (prof) {
                if (c->ev_used == (int)c->ev_start.size()) {
                    HIPCHK(hipStreamSynchronize(c->stream));
                    int rc = ctx_flush_profile(c);
                    if (rc) return rc;
                }
                HIPCHK(hipEventRecord(c->ev_start[c->ev_used], c->stream));
            }
            launch_bounds(c->d_src, (int)c->ns, c->d_lut, c->geom, a, c->d_partials, c->nchunk, c->pts_per_thread, c->stream);
            if (prof) {
                HIPCHK(hipEventRecord(c->ev_stop[c->ev_used], c->stream));
                c->ev_used++;
                c->prof_launches++;
                c->prof_subcubes += pc.B;
                c->prof_evals += pc.B;
            }
            rows += pc.B;
            ++pi;
        }
        launch_bounds_finalize(c->d_partials, c->nchunk, rows, c->hd_lb, c->hd_ub, TickCut(), c->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->stream));
        std::memcpy(lb_out + first, c->h_lb, sizeof(float) * rows);
        std::memcpy(ub_out + first, c->h_ub, sizeof(float) * rows);
        if (c->profile) {
            int rc = ctx_flush_profile(c);
            if (rc) return rc;
        }
    }
    return FGOICP_OK;
}

// ---- host-side final folds of the fused reductions (small clouds) -----------------------------------------------------------------
// The device reduces in three levels: the shuffle tree of a wave (kernels.hip wave_sum: v[i] += v[i + off], off = 32 .. 1), the four
// waves of a block in order, then sum_partials_kernel / icp_centroids_kernel over the blocks (lane-strided partial sums + the same
// tree).  With the first level done in the scan's epilogue, the rest are a few hundred additions: done here, in the device's
// order, on values the kernels wrote straight into pinned memory — the same bits, two launches less per pass.
static double tree64(double (&v)[64]) {
    for (int off = 32; off > 0; off >>= 1)
        for (int i = 0; i < off; ++i) v[i] += v[i + off];
    return v[0];
}
static double fold_blocks(const double* bp, int nblocks, int width, int k) {  // sum_partials_kernel
    double lane[64] = {0};
    for (int b = 0; b < nblocks; ++b) lane[b & 63] += bp[(size_t)b * width + k];
    return tree64(lane);
}
static double fold_wave_sums(const double* w, int nwaves) {  // block_sum<1> of sum_f32_kernel over the waves, then sum_partials_kernel
    double lane[64] = {0};
    const int nblocks = (nwaves + 3) / 4;
    for (int b = 0; b < nblocks; ++b) {
        double blk = w[4 * b];
        for (int k = 1; k < 4; ++k) blk += 4 * b + k < nwaves ? w[4 * b + k] : 0.0;
        lane[b & 63] += blk;
    }
    return tree64(lane);
}
static bool icp_fused(const fgoicp_ctx* c) { return c->icp_fuse && !c->brute_force_nn && !c->inliers && c->ns <= 262144; }

// float Registration::compute_sse_error(glm::mat3, glm::vec3) — registration.cu:62-86.  Enqueue only: the result lands in
// pinned memory (sse_result) once `st` has drained.
// `part`: 0 = everything, 1 = the search only, 2 = the sum only (the ICP loop interleaves the launches of its two streams: both long
// scans first — a launch costs the submitting thread ~5 us, and whatever is enqueued last starts that much later).
static int sse_enqueue(fgoicp_ctx* c, fgoicp_ctx::IcpLane& L, const float* R9, const float* t3, const uint32_t* seed_idx, hipStream_t st, int part = 0) {
    const int ns = (int)c->ns;
    if (icp_fused(c)) {  // one launch: the scan leaves the wave-level sums of its minima in pinned memory, sse_result() folds them
        if (part != 2) launch_nn_scan(c->d_src, ns, c->bvh_tgt.view(), c->d_lut, c->geom, R9, t3, 1, 0, c->d_tgt, (int)c->nt, seed_idx, nullptr, nullptr, L.d_min_bits, st,
                                      nullptr, nullptr, nullptr, L.hd_wsse);
        L.sse_on_host = true;
        HIPCHK(hipGetLastError());
        return FGOICP_OK;
    }
    L.sse_on_host = false;
    if (part == 2) {
    } else if (c->brute_force_nn) {
        launch_fill_u32(L.d_min_bits, 0x501502F9u /* bits(1e10f) */, c->ns, st);
        launch_nn_min(c->d_src, ns, c->d_tgt, (int)c->nt, R9, t3, 1, L.d_min_bits, st);
    } else {
        const float* skip_lb = nullptr;
        const uint32_t* skip_u = nullptr;
        if (c->inliers && c->trim_skip) {  // trimmed: queries provably beyond the k-th smallest distance are left out of the exact search
            launch_nn_prep(c->d_src, ns, c->d_lut, c->geom, R9, t3, 1, c->d_tgt, (int)c->nt, seed_idx, c->tgt_box6, L.d_nn_ub, L.d_nn_lb, st);
            launch_trim_select(L.d_nn_ub, ns, (int)c->inliers, nullptr, L.d_sel + 8, L.d_sel_wide, st);
            skip_lb = L.d_nn_lb;
            skip_u = L.d_sel + 8;
        }
        launch_nn_scan(c->d_src, ns, c->bvh_tgt.view(), c->d_lut, c->geom, R9, t3, 1, 0, c->d_tgt, (int)c->nt, seed_idx, skip_lb, skip_u, L.d_min_bits, st);
    }
    if (part == 1) {
    } else if (c->inliers) {  // trimmed SSE: the k smallest nearest-neighbour terms
        launch_trim_select(reinterpret_cast<const float*>(L.d_min_bits), ns, (int)c->inliers, L.hd_trim, nullptr, L.d_sel_wide, st);
    } else {
        const int nb = reduce_blocks_for(ns);
        launch_sum_f32_as_f64(L.d_min_bits, ns, L.d_bp3, nb, st);
        launch_sum_partials(L.d_bp3, nb, 1, L.hd_sums + 12, st);
    }
    HIPCHK(hipGetLastError());
    return FGOICP_OK;
}
static float sse_result(const fgoicp_ctx* c, const fgoicp_ctx::IcpLane& L) {
    if (L.sse_on_host) return (float)fold_wave_sums(L.h_wsse, (int)((c->ns + 63) / 64));
    return c->inliers ? L.h_trim[0] : (float)L.h_sums[12];
}

static int lane_sse(fgoicp_ctx* c, fgoicp_ctx::IcpLane& L, const float* R9, const float* t3, float* sse_out, const uint32_t* seed_idx) {
    int rc = sse_enqueue(c, L, R9, t3, seed_idx, L.stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(L.stream));
    *sse_out = sse_result(c, L);
    return FGOICP_OK;
}
int ctx_sse(fgoicp_ctx* c, const float* R9, const float* t3, float* sse_out, const uint32_t* seed_idx) {
    HIPCHK(hipSetDevice(c->device));
    return lane_sse(c, c->lanes[0], R9, t3, sse_out, seed_idx);
}

// IterativeClosestPoint3D::procrustes() on L.d_work — icp3d.cu:140-172.  The device half (enqueue only): correspondences
// into `idx`, centroids and covariance into pinned memory; `wide` is the selection scratch of the trimmed variant.
// move9 / move3 (optional): the working cloud is first moved by this (R_, t_) — icp3d.cu:100 of the iteration before — inside the
// correspondence scan where that is possible (one launch less on the iteration's critical chain), by its own kernel otherwise.
// `part` as for sse_enqueue: 1 = up to and including the correspondence search, 2 = the inlier cut and the sums.
static int procrustes_enqueue(fgoicp_ctx* c, fgoicp_ctx::IcpLane& L, const uint32_t* seed_idx, uint32_t* idx, uint32_t* wide, hipStream_t st,
                              const float* move9 = nullptr, const float* move3 = nullptr, int part = 0) {
    const int ns = (int)c->ns, nt = (int)c->nt;
    if (icp_fused(c)) {  // two launches: scan (+ move, + wave-level sums of points and correspondences), covariance (+ centroids) into pinned memory
        const int nb = reduce_blocks_for(ns);
        if (part != 2)
            launch_nn_scan(L.d_work, ns, c->bvh_tgt.view(), c->d_lut, c->geom, move9, move3, move9 ? 1 : 0, 1, c->d_tgt, nt, seed_idx, nullptr, nullptr, idx, st,
                           move9 ? L.d_work : nullptr, nullptr, nullptr, L.d_wsum);
        if (part != 1) launch_icp_cov_cen(L.d_work, c->d_tgt, idx, ns, nt, L.d_wsum, nb, (ns + 63) / 64, L.hd_cen, L.hd_covbp, nb, st);
        L.cov_on_host = true;
        L.cov_blocks = nb;
        HIPCHK(hipGetLastError());
        return FGOICP_OK;
    }
    L.cov_on_host = false;
    if (part != 2) {
        static const bool fold_move = [] { const char* e = dev_env("FGOICP_ICP_FOLD_MOVE"); return !e || std::atoi(e) != 0; }();  // tuning knob / A-B
        const bool fold = move9 && fold_move && !c->brute_force_nn && !(c->inliers && c->trim_skip);
        if (move9 && !fold) launch_transform_inplace(L.d_work, ns, move9, move3, st);
        // kernFindNearestNeighbor (icp3d.cu:11-28): min distance, tie set, lowest index
        if (c->brute_force_nn) {
            launch_fill_u32(L.d_min_bits, 0x501502F9u, c->ns, st);
            launch_fill_u32(idx, 0x7fffffffu, c->ns, st);
            launch_nn_min(L.d_work, ns, c->d_tgt, nt, nullptr, nullptr, 0, L.d_min_bits, st);
            launch_nn_tie_threshold(L.d_min_bits, ns, L.d_thr_bits, st);
            launch_nn_first_index(L.d_work, ns, c->d_tgt, nt, L.d_thr_bits, idx, st);
        } else {
            const float* skip_lb = nullptr;
            const uint32_t* skip_u = nullptr;
            if (c->inliers && c->trim_skip) {  // trimmed: points provably outside the inlier set get no correspondence
                launch_nn_prep(L.d_work, ns, c->d_lut, c->geom, nullptr, nullptr, 0, c->d_tgt, nt, seed_idx, c->tgt_box6, L.d_nn_ub2, L.d_nn_lb2, st);
                launch_trim_select(L.d_nn_ub2, ns, (int)c->inliers, nullptr, L.d_sel + 4, wide, st);
                skip_lb = L.d_nn_lb2;
                skip_u = L.d_sel + 4;
            }
            launch_nn_scan(L.d_work, ns, c->bvh_tgt.view(), c->d_lut, c->geom, fold ? move9 : nullptr, fold ? move3 : nullptr, fold ? 1 : 0, 1, c->d_tgt, nt, seed_idx, skip_lb,
                           skip_u, idx, st, fold ? L.d_work : nullptr);
        }
    }
    if (part == 1) { HIPCHK(hipGetLastError()); return FGOICP_OK; }
    const int nb = reduce_blocks_for(ns);
    const unsigned char* use = nullptr;
    int ncount = ns;
    if (c->inliers) {  // trimmed ICP: only the k closest correspondences enter the Procrustes sums
        launch_icp_inliers(L.d_work, c->d_tgt, idx, ns, nt, (int)c->inliers, L.d_d2, L.d_sel, L.d_eq, c->d_orig_of_slot, L.d_use, wide, st);
        use = L.d_use;
        ncount = (int)c->inliers;
    }
    // (Folding the block partials in the grid's last block instead of a follow-up one-block kernel — the threadfence reduction —
    // was measured and lost, 62 -> 91 us per iteration: an agent-scope fence per block writes back the XCD's L2 on this 8-XCD part.)
    launch_icp_sums(L.d_work, c->d_tgt, idx, ns, nt, use, L.d_bp, nb, st);
    launch_icp_centroids(L.d_bp, nb, ncount, L.d_cen, L.hd_cen, st);  // icp3d.cu:152-156, no host round trip
    launch_icp_cov(L.d_work, c->d_tgt, idx, ns, nt, L.d_cen, use, L.d_bp2, nb, st);
    launch_sum_partials(L.d_bp2, nb, 9, L.hd_sums, st);
    HIPCHK(hipGetLastError());
    return FGOICP_OK;
}
// ... and the host half, once the stream has drained: 3x3 SVD, icp3d.cu:168-169
static void procrustes_finish(const fgoicp_ctx::IcpLane& L, Mat3f* R_out, Vec3f* t_out, float* centroids6_out, Mat3f* ABt_out) {
    float cen[6];
    std::memcpy(cen, L.h_cen, sizeof(cen));
    Mat3f ABt;
    for (int k = 0; k < 9; ++k) ABt.m[k] = L.cov_on_host ? (float)fold_blocks(L.h_covbp, L.cov_blocks, 9, k) : (float)L.h_sums[k];
    const Mat3f Rn = closest_orthogonal_approximation(ABt);  // icp3d.cu:168
    const Vec3f sc{cen[0], cen[1], cen[2]}, cc{cen[3], cen[4], cen[5]};
    *R_out = Rn;
    *t_out = cc - Rn * sc;  // icp3d.cu:169
    if (centroids6_out) std::memcpy(centroids6_out, cen, sizeof(cen));
    if (ABt_out) *ABt_out = ABt;
}

int ctx_procrustes_device(fgoicp_ctx* c, fgoicp_ctx::IcpLane& L, Mat3f* R_out, Vec3f* t_out, float* centroids6_out, Mat3f* ABt_out, bool seeded) {
    int rc = procrustes_enqueue(c, L, seeded ? L.d_first_idx : nullptr, L.d_first_idx, L.d_sel_wide, L.stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(L.stream));
    procrustes_finish(L, R_out, t_out, centroids6_out, ABt_out);
    return FGOICP_OK;
}

// IterativeClosestPoint3D ctor + run() — icp3d.cu:55-108.
// Two streams: the exact SSE of iteration k (pristine source under the composed transform, main stream) and the
// correspondence + covariance pass of iteration k+1 (working cloud, side stream) depend on nothing but the transform of
// iteration k, so they run next to each other — both are latency chains that fill half the device at 40k points.  The pass of
// iteration k+1 is speculative (the loop may end on the SSE of iteration k); it is drained before returning.  Same kernels,
// same arithmetic, same order of every sum as the one-stream loop (FGOICP_ICP_OVERLAP=0).
#ifdef FGOICP_DEV_KNOBS
static int lane_icp_device(fgoicp_ctx* c, fgoicp_ctx::IcpLane& L, const float* R0, const float* t0, size_t max_iter, float thr, float* sse_out, float* R_out9,
                           float* t_out3, int* iters_out);
static int lane_icp_gated(fgoicp_ctx* c, fgoicp_ctx::IcpLane& L, const float* R0, const float* t0, size_t max_iter, float thr, float* sse_out, float* R_out9,
                          float* t_out3, int* iters_out);
#endif
static int lane_icp_dual(fgoicp_ctx* c, fgoicp_ctx::IcpLane& L, const float* R0, const float* t0, size_t max_iter, float thr, float* sse_out, float* R_out9,
                         float* t_out3, int* iters_out);
static int lane_icp(fgoicp_ctx* c, fgoicp_ctx::IcpLane& L, const float* R0, const float* t0, size_t max_iter, float thr, float* sse_out, float* R_out9, float* t_out3,
                    int* iters_out) {
#ifdef FGOICP_DEV_KNOBS
    if (c->icp_device && c->icp_overlap && !c->brute_force_nn && !c->inliers) return lane_icp_device(c, L, R0, t0, max_iter, thr, sse_out, R_out9, t_out3, iters_out);
#endif
    // one walk for both scans where the device is full anyway (clouds beyond 262 144 points, trimmed runs: -3 ... -5 % of the ICP time); below
    // that the two scans of an iteration overlap on two streams and a wave carrying both query sets only lengthens the chain (40k points:
    // 51-53 -> 55-56 us per iteration) — FGOICP_ICP_DUAL = 1 / 0 forces either
#ifdef FGOICP_DEV_KNOBS
    if (c->icp_gated && c->icp_gate_ok && c->icp_dual_env <= 0 && c->icp_overlap && icp_fused(c) && L.sig_b && L.sig_a)
        return lane_icp_gated(c, L, R0, t0, max_iter, thr, sse_out, R_out9, t_out3, iters_out);
#endif
    const bool dual = c->icp_dual_env >= 0 ? c->icp_dual_env != 0 : !icp_fused(c);
    if (dual && c->icp_overlap && !c->brute_force_nn) return lane_icp_dual(c, L, R0, t0, max_iter, thr, sse_out, R_out9, t_out3, iters_out);
    HIPCHK(hipSetDevice(c->device));  // per host thread
    const int ns = (int)c->ns;
    const bool overlap = c->icp_overlap && !c->brute_force_nn;
    const bool seeding = c->icp_seeding && !c->brute_force_nn;
    hipStream_t A = L.stream, B = overlap ? L.icp_stream : L.stream;
    uint32_t* idx[2] = {L.d_first_idx, overlap ? L.d_first_idx2 : L.d_first_idx};
    int cur = 0;
    HIPCHK(hipMemcpyAsync(L.d_work, c->d_src, sizeof(float4) * c->ns, hipMemcpyDeviceToDevice, A));
    launch_transform_inplace(L.d_work, ns, R0, t0, A);  // icp3d.cu:85
    Mat3f R = Mat3f::from(R0);
    Vec3f t{t0[0], t0[1], t0[2]};
    size_t iter = 0;
    float sse = kInf, last_sse = 2.0f * kInf;
    Mat3f last_R = Mat3f::identity();
    Vec3f last_t{0, 0, 0};
    int iters = 0;
    bool chain_pending = false;  // a correspondence pass is in flight on B
    if (overlap && max_iter > 0) {
        HIPCHK(hipEventRecord(L.icp_ev_w, A));
        HIPCHK(hipStreamWaitEvent(B, L.icp_ev_w, 0));
        int rc = procrustes_enqueue(c, L, nullptr, idx[0], L.d_sel_wide2, B);
        if (rc) return rc;
        HIPCHK(hipEventRecord(L.icp_ev_b, B));
        chain_pending = true;
    }
    while (iter++ < max_iter && (last_sse - sse) > thr * last_sse) {  // icp3d.cu:94
        last_sse = sse;
        last_R = R;
        last_t = t;
        Mat3f Rn;
        Vec3f tn;
        if (overlap) {
            HIPCHK(wait_event(L.icp_ev_b));
            chain_pending = false;
            procrustes_finish(L, &Rn, &tn, nullptr, nullptr);
        } else {
            // every pass after the first seeds its exact search with the correspondences of the pass before it
            int rc = ctx_procrustes_device(c, L, &Rn, &tn, nullptr, nullptr, iters > 0 && seeding);
            if (rc) return rc;
        }
        const float tn3[3] = {tn.x, tn.y, tn.z};
        R = Rn * R;                                              // :101
        t = Rn * t + tn;                                         // :102
        const float t3[3] = {t.x, t.y, t.z};
        if (overlap) {
            // The working cloud and the next iteration's pass are the critical path (transform -> correspondences -> sums -> host):
            // they go to the side stream FIRST; this iteration's SSE, which needs nothing but (R, t), is enqueued behind them on the
            // main stream.  (Enqueued after the SSE chain, the correspondence scan started 25 us late: five launches of host time.)
            const uint32_t* seed = seeding ? idx[cur] : nullptr;  // the pass the host has just consumed
            const bool next = iter < max_iter;  // the next iteration's pass (it moves the cloud first, :100), next to this iteration's SSE
            int rc = FGOICP_OK;
            if (next) rc = procrustes_enqueue(c, L, seed, idx[cur ^ 1], L.d_sel_wide2, B, Rn.m, tn3, 1);
            else launch_transform_inplace(L.d_work, ns, Rn.m, tn3, B);  // :100 (B is in order behind the pass that read d_work)
            if (rc) return rc;
            rc = sse_enqueue(c, L, R.m, t3, seed, A, 1);  // :103
            if (rc) return rc;
            if (next) {
                rc = procrustes_enqueue(c, L, seed, idx[cur ^ 1], L.d_sel_wide2, B, nullptr, nullptr, 2);
                if (rc) return rc;
                cur ^= 1;
            }
            HIPCHK(hipEventRecord(L.icp_ev_b, B));
            chain_pending = true;
            rc = sse_enqueue(c, L, R.m, t3, seed, A, 2);
            if (rc) return rc;
            HIPCHK(wait_stream(A));
            sse = sse_result(c, L);
        } else {
            launch_transform_inplace(L.d_work, ns, Rn.m, tn3, A);  // :100
            int rc = lane_sse(c, L, R.m, t3, &sse, seeding ? L.d_first_idx : nullptr);  // :103
            if (rc) return rc;
        }
        ++iters;
    }
    if (chain_pending) HIPCHK(hipEventSynchronize(L.icp_ev_b));  // the speculative pass: drained, not used
    const bool cur_best = sse < last_sse;  // :106-107
    *sse_out = cur_best ? sse : last_sse;
    const Mat3f& Ro = cur_best ? R : last_R;
    const Vec3f& to = cur_best ? t : last_t;
    std::memcpy(R_out9, Ro.m, sizeof(Ro.m));
    t_out3[0] = to.x; t_out3[1] = to.y; t_out3[2] = to.z;
    if (iters_out) *iters_out = iters;
    return FGOICP_OK;
}

#ifdef FGOICP_DEV_KNOBS
// The same loop advanced ON THE DEVICE (kernels.hip, icp_step_kernel): no host round trip inside the loop.  Per iteration j the
// host enqueues, without waiting for anything,
//   stream B:  step_j   (waits for the SSE partials of iteration j-1: loop test of :94, then SVD, compose -> state)
//   stream A:  E_j      exact SSE of the composed (R, t) of the state on the pristine source   (scan + block sums)
//   stream B:  P_{j+1}  correspondences of the working cloud moved by the state's (R_, t_) (+ write-back), sums, covariance
// and paces itself by the progress word the step kernel leaves in pinned memory (`icp_ahead` iterations ahead at most).  Once a
// step kernel has ended the loop everything enqueued behind it returns at once.  Same kernels, same sums in the same order as the
// host loop: bit-identical (sse, R, t, iterations) (tests/test_gpu_ops.py).  Untrimmed tree path only.
static int lane_icp_device(fgoicp_ctx* c, fgoicp_ctx::IcpLane& L, const float* R0, const float* t0, size_t max_iter, float thr, float* sse_out, float* R_out9,
                           float* t_out3, int* iters_out) {
    HIPCHK(hipSetDevice(c->device));
    const int ns = (int)c->ns, nt = (int)c->nt;
    const int nb = reduce_blocks_for(ns);
    const int mi = (int)std::min<size_t>(max_iter, (size_t)1 << 30);
    constexpr int kRing = fgoicp_ctx::IcpLane::kRing;
    hipStream_t A = L.stream, B = L.icp_stream;
    const bool seeding = c->icp_seeding;
    uint32_t* idx[2] = {L.d_first_idx, L.d_first_idx2};
    const float* st_f = reinterpret_cast<const float*>(L.d_icp);
    const int* done = &L.d_icp->done;
    volatile IcpHostResult* res = L.h_res;
    res->done = 0;
    res->iters_done = 0;
    launch_icp_init(L.d_icp, R0, t0, mi, thr, L.hd_res, A);
    HIPCHK(hipEventRecord(L.icp_ev_w, A));
    HIPCHK(hipStreamWaitEvent(B, L.icp_ev_w, 0));
    // P_1: the pristine source moved by (R0, t0) becomes the working cloud (icp3d.cu:85) inside the first correspondence scan
    launch_nn_scan(c->d_src, ns, c->bvh_tgt.view(), c->d_lut, c->geom, R0, t0, 1, 1, c->d_tgt, nt, nullptr, nullptr, nullptr, idx[0], B, L.d_work);
    launch_icp_sums(L.d_work, c->d_tgt, idx[0], ns, nt, nullptr, L.d_bp, nb, B);
    launch_icp_cov_cen(L.d_work, c->d_tgt, idx[0], ns, nt, L.d_bp, nb, 0, L.d_cen, L.d_bp2, nb, B);
    int cur = 0, enq = 0;
    for (int j = 1; j <= mi; ++j) {
        // pace: at most icp_ahead iterations ahead of the step kernels that have run.  The progress word is only a pacing hint;
        // if it does not move for 20 ms (it always has), the host falls back to draining the stream — never a hang.
        if (!res->done && j - res->iters_done > c->icp_ahead) {
            const auto t_spin = std::chrono::steady_clock::now();
            unsigned spins = 0;
            while (!res->done && j - res->iters_done > c->icp_ahead) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
                if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t_spin > std::chrono::milliseconds(20)) {
                    HIPCHK(hipStreamSynchronize(B));
                    break;
                }
            }
        }
        if (res->done) break;
        if (j > 1) HIPCHK(hipStreamWaitEvent(B, L.ev_sse[(j - 1) % kRing], 0));
        launch_icp_step(L.d_icp, L.d_bp2, nb, L.d_bp3, nb, L.d_cen, L.hd_res, B);
        HIPCHK(hipEventRecord(L.ev_step[j % kRing], B));
        const uint32_t* seed = seeding ? idx[cur] : nullptr;
        // P_{j+1} first: it is the head of the next iteration's critical chain (scan -> sums -> covariance -> step)
        if (j < mi) {
            launch_nn_scan(L.d_work, ns, c->bvh_tgt.view(), c->d_lut, c->geom, nullptr, nullptr, 1, 1, c->d_tgt, nt, seed, nullptr, nullptr, idx[cur ^ 1], B, L.d_work,
                           st_f + 12, done);
        }
        HIPCHK(hipStreamWaitEvent(A, L.ev_step[j % kRing], 0));
        launch_nn_scan(c->d_src, ns, c->bvh_tgt.view(), c->d_lut, c->geom, nullptr, nullptr, 1, 0, c->d_tgt, nt, seed, nullptr, nullptr, L.d_min_bits, A, nullptr, st_f, done);
        if (j < mi) {
            launch_icp_sums(L.d_work, c->d_tgt, idx[cur ^ 1], ns, nt, nullptr, L.d_bp, nb, B, done);
            launch_icp_cov_cen(L.d_work, c->d_tgt, idx[cur ^ 1], ns, nt, L.d_bp, nb, 0, L.d_cen, L.d_bp2, nb, B, done);
            cur ^= 1;
        }
        launch_sum_f32_as_f64(L.d_min_bits, ns, L.d_bp3, nb, A, done);
        HIPCHK(hipEventRecord(L.ev_sse[j % kRing], A));
        enq = j;
    }
    // the deciding step: behind the last SSE (returns at once if an earlier step already ended the loop)
    if (enq > 0) HIPCHK(hipStreamWaitEvent(B, L.ev_sse[enq % kRing], 0));
    launch_icp_step(L.d_icp, L.d_bp2, nb, L.d_bp3, nb, L.d_cen, L.hd_res, B);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(B));
    HIPCHK(hipStreamSynchronize(A));  // passes enqueued ahead of the decision: drained (they return at once)
    if (!res->done) { set_error("device-resident ICP loop ended without a result"); return FGOICP_ERR_HIP; }
    *sse_out = res->sse;
    for (int k = 0; k < 9; ++k) R_out9[k] = res->R[k];
    for (int k = 0; k < 3; ++k) t_out3[k] = res->t[k];
    if (iters_out) *iters_out = res->iters;
    return FGOICP_OK;
}

#endif  // FGOICP_DEV_KNOBS (device-resident loop)

int ctx_icp(fgoicp_ctx* c, const float* R0, const float* t0, size_t max_iter, float thr, float* sse_out, float* R_out9, float* t_out3, int* iters_out) {
    return lane_icp(c, c->lanes[0], R0, t0, max_iter, thr, sse_out, R_out9, t_out3, iters_out);
}

// COOPERATIVE ICP (round 3): `world` ranks that hold the same clouds run ONE IterativeClosestPoint3D::run() (icp3d.cu:80-108)
// together.  The two exact scans of an iteration — what an ICP run consists of (600-1900 us per iteration at 437k points) — are
// split by query range: rank r scans the Hilbert-consecutive queries [r * per, (r + 1) * per), `gather` all-gathers the results
// (correspondence indices / bits of the minima: 4 B per query) IN PLACE on device memory, and every rank then runs the cheap
// reductions (sums, centroids, covariance, SSE) over the WHOLE cloud with the single-GPU kernels and the 3x3 SVD on its host.  A
// query's nearest neighbour does not depend on which queries are scanned next to it and the reductions see the same arrays in the
// same order on every rank, so (sse, R, t, iterations) are the single-GPU loop's bits on every rank — no result needs to be
// exchanged, and an N-rank run refines exactly like a one-rank run.  Brute-force contexts, and every context below
// coop_split_min source points (the default: never split), run the whole loop on every rank instead (replicated: same bits).  `gather(buf, bytes_per_rank, user)`: the caller's stream is idle when it is
// called; on return chunk r of buf holds rank r's results, for every r.
int ctx_icp_coop(fgoicp_ctx* c, int rank, int world, int (*gather)(void* dev_buf, size_t bytes_per_rank, void* user), void* user, const float* R0,
                 const float* t0, size_t max_iter, float thr, float* sse_out, float* R_out9, float* t_out3, int* iters_out) {
    const size_t split_min = c->inliers ? std::min(c->coop_split_min, c->coop_split_trim_min) : c->coop_split_min;
    if (world <= 1 || !gather || c->brute_force_nn || c->ns < split_min || !c->icp_overlap) return ctx_icp(c, R0, t0, max_iter, thr, sse_out, R_out9, t_out3, iters_out);
    if (rank < 0 || rank >= world) { set_error("ctx_icp_coop: rank out of range"); return FGOICP_ERR_INVALID_ARG; }
    HIPCHK(hipSetDevice(c->device));
    fgoicp_ctx::IcpLane& L = c->lanes[0];
    const int ns = (int)c->ns, nt = (int)c->nt;
    const size_t per = ((((size_t)ns + world - 1) / world) + 255) & ~(size_t)255;  // whole 256-query blocks per rank
    if (c->coop_cap < per * world) {
        (void)hipFree(c->d_coop);
        c->d_coop = nullptr;
        c->coop_cap = 0;
        HIPCHK(hipMalloc(&c->d_coop, sizeof(uint32_t) * 3 * per * world));
        HIPCHK(hipMemset(c->d_coop, 0, sizeof(uint32_t) * 3 * per * world));
        c->coop_cap = per * world;
    }
    // the gathered correspondences are double-buffered, like idx[cur] / idx[cur ^ 1] of the one-GPU loop: a pass reads its seeds from the
    // buffer the pass before it wrote and writes the other one (round 3 used ONE buffer as seed and output of the same kernel: with skip
    // lists the kernel wrote 0x7fffffff to a slot before every wave of the block had read its seed — results stayed exact, a seed only
    // tightens a bound, but the work done varied from run to run)
    uint32_t* idx_buf[2] = {c->d_coop, c->d_coop + c->coop_cap};
    int cur = 0;
    uint32_t* idx = idx_buf[cur];
    uint32_t* mins = c->d_coop + 2 * c->coop_cap;
    const int qb = (int)std::min<size_t>((size_t)ns, per * rank), nq = (int)std::min<size_t>((size_t)ns, per * (rank + 1)) - qb;
    const bool seeding = c->icp_seeding;
    hipStream_t S = L.stream;
    const int nb = reduce_blocks_for(ns);
    HIPCHK(hipMemcpyAsync(L.d_work, c->d_src, sizeof(float4) * c->ns, hipMemcpyDeviceToDevice, S));
    launch_transform_inplace(L.d_work, ns, R0, t0, S);  // icp3d.cu:85
    Mat3f R = Mat3f::from(R0);
    Vec3f t{t0[0], t0[1], t0[2]};
    size_t iter = 0;
    float sse = kInf, last_sse = 2.0f * kInf;
    Mat3f last_R = Mat3f::identity();
    Vec3f last_t{0, 0, 0};
    int iters = 0;
    // The loop is lane_icp_dual's (one walk per iteration serves the exact SSE of iteration k and the correspondences of iteration k + 1),
    // with the walk cut down to this rank's queries and two in-place all-gathers behind it: ONE scan, two gathers and two host syncs
    // per iteration (a first version ran the two scans one after the other: two scans, two gathers, four syncs).  Trimmed contexts
    // (c->inliers): the LUT brackets, the cuts and the inlier selection run over the whole cloud on every rank (cheap, replicated), the
    // walk — 85 % of a trimmed iteration at 1M points — on the share; the reductions are the single-GPU loop's own enqueue functions.
    const bool trimmed = c->inliers != 0, skip = trimmed && c->trim_skip;
    struct MinBitsGuard {  // sse_enqueue's trimmed selection reads L.d_min_bits: it is pointed at the gathered buffer for the duration of the run
        fgoicp_ctx::IcpLane& L; uint32_t* saved;
        MinBitsGuard(fgoicp_ctx::IcpLane& l, uint32_t* m) : L(l), saved(l.d_min_bits) { L.d_min_bits = m; }
        ~MinBitsGuard() { L.d_min_bits = saved; }
    } guard(L, mins);
    auto reduce_pass = [&]() -> int {  // inlier cut (trimmed), sums, centroids, covariance of the correspondences in idx[] (all ranks' shares)
        if (trimmed) return procrustes_enqueue(c, L, nullptr, idx, L.d_sel_wide2, S, nullptr, nullptr, 2);
        launch_icp_sums(L.d_work, c->d_tgt, idx, ns, nt, nullptr, L.d_bp, nb, S);
        launch_icp_centroids(L.d_bp, nb, ns, L.d_cen, L.hd_cen, S);
        launch_icp_cov(L.d_work, c->d_tgt, idx, ns, nt, L.d_cen, nullptr, L.d_bp2, nb, S);
        launch_sum_partials(L.d_bp2, nb, 9, L.hd_sums, S);
        L.cov_on_host = false;
        return FGOICP_OK;
    };
    auto sse_sum = [&]() -> int {  // compute_sse_error's sum over the gathered minima (trimmed: the k smallest)
        if (trimmed) return sse_enqueue(c, L, nullptr, nullptr, nullptr, S, 2);
        launch_sum_f32_as_f64(mins, ns, L.d_bp3, nb, S);
        launch_sum_partials(L.d_bp3, nb, 1, L.hd_sums + 12, S);
        L.sse_on_host = false;
        return FGOICP_OK;
    };
    bool pending = false;
    if (max_iter > 0) {  // pass 1 (icp3d.cu:140-172): correspondences of my share, gathered; sums over everything
        if (skip) {
            launch_nn_prep(L.d_work, ns, c->d_lut, c->geom, nullptr, nullptr, 0, c->d_tgt, nt, nullptr, c->tgt_box6, L.d_nn_ub2, L.d_nn_lb2, S);
            launch_trim_select(L.d_nn_ub2, ns, (int)c->inliers, nullptr, L.d_sel + 4, L.d_sel_wide2, S);
        }
        if (nq > 0)
            launch_nn_scan(L.d_work + qb, nq, c->bvh_tgt.view(), c->d_lut, c->geom, nullptr, nullptr, 0, 1, c->d_tgt, nt, nullptr, skip ? L.d_nn_lb2 + qb : nullptr,
                           skip ? L.d_sel + 4 : nullptr, idx + qb, S);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(S));
        if (gather(idx, sizeof(uint32_t) * per, user)) return FGOICP_ERR_EXCHANGE;
        int rc = reduce_pass();
        if (rc) return rc;
        pending = true;
    }
    while (iter++ < max_iter && (last_sse - sse) > thr * last_sse) {  // icp3d.cu:94
        last_sse = sse;
        last_R = R;
        last_t = t;
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(S));  // first iteration: pass 1; later: a no-op (the SSE's sync below drained the stream)
        pending = false;
        Mat3f Rn;
        Vec3f tn;
        procrustes_finish(L, &Rn, &tn, nullptr, nullptr);
        const float tn3[3] = {tn.x, tn.y, tn.z};
        R = Rn * R;                                              // :101
        t = Rn * t + tn;                                         // :102
        const float t3[3] = {t.x, t.y, t.z};
        launch_transform_inplace(L.d_work, ns, Rn.m, tn3, S);    // :100 (the whole cloud: the reductions read all of it)
        const bool next = iter < max_iter;
        const uint32_t* seed = seeding ? idx : nullptr;          // the gathered correspondences of the pass the host has just consumed
        uint32_t* idx_out = idx_buf[cur ^ 1];                    // where the pass riding along writes (never the buffer the seeds come from)
        const float *lbA = nullptr, *lbB = nullptr;
        const uint32_t *uA = nullptr, *uB = nullptr;
        if (skip) {  // trimmed: brackets and cuts of both query sets, whole cloud, as lane_icp_dual
            if (next) {
                launch_nn_prep(L.d_work, ns, c->d_lut, c->geom, nullptr, nullptr, 0, c->d_tgt, nt, seed, c->tgt_box6, L.d_nn_ub2, L.d_nn_lb2, S);
                launch_trim_select(L.d_nn_ub2, ns, (int)c->inliers, nullptr, L.d_sel + 4, L.d_sel_wide2, S);
                lbA = L.d_nn_lb2 + qb; uA = L.d_sel + 4;
            }
            launch_nn_prep(c->d_src, ns, c->d_lut, c->geom, R.m, t3, 1, c->d_tgt, nt, seed, c->tgt_box6, L.d_nn_ub, L.d_nn_lb, S);
            launch_trim_select(L.d_nn_ub, ns, (int)c->inliers, nullptr, L.d_sel + 8, L.d_sel_wide, S);
            lbB = L.d_nn_lb + qb; uB = L.d_sel + 8;
        }
        if (nq > 0) {
            if (next)  // compute_sse_error(R, t) (:103) of this iteration and the correspondences of the next: one walk, my share
                launch_nn_scan_dual(L.d_work + qb, nullptr, nullptr, 0, c->d_src + qb, R.m, t3, nq, c->bvh_tgt.view(), c->d_lut, c->geom, c->d_tgt, nt, seed ? seed + qb : nullptr,
                                    lbA, uA, lbB, uB, idx_out + qb, mins + qb, nullptr, nullptr, nullptr, S);
            else       // the last iteration the loop can make: no pass rides along
                launch_nn_scan(c->d_src + qb, nq, c->bvh_tgt.view(), c->d_lut, c->geom, R.m, t3, 1, 0, c->d_tgt, nt, seed ? seed + qb : nullptr, lbB, uB, mins + qb, S);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(S));
        if (next && gather(idx_out, sizeof(uint32_t) * per, user)) return FGOICP_ERR_EXCHANGE;
        if (gather(mins, sizeof(uint32_t) * per, user)) return FGOICP_ERR_EXCHANGE;
        int rc = FGOICP_OK;
        if (next) { cur ^= 1; idx = idx_buf[cur]; rc = reduce_pass(); pending = true; }  // speculative, as on one GPU (the loop may end on this iteration's SSE)
        if (rc) return rc;
        rc = sse_sum();
        if (rc) return rc;
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(S));
        sse = sse_result(c, L);
        ++iters;
    }
    if (pending) HIPCHK(hipStreamSynchronize(S));  // the speculative pass: drained, not used
    const bool cur_best = sse < last_sse;  // :106-107
    *sse_out = cur_best ? sse : last_sse;
    const Mat3f& Ro = cur_best ? R : last_R;
    const Vec3f& to = cur_best ? t : last_t;
    std::memcpy(R_out9, Ro.m, sizeof(Ro.m));
    t_out3[0] = to.x; t_out3[1] = to.y; t_out3[2] = to.z;
    if (iters_out) *iters_out = iters;
    return FGOICP_OK;
}

// The loop with ONE walk per iteration (kernels.hip nn_scan_dual_kernel; default).  Once the host has (R_, t_) of iteration k, the exact
// SSE of iteration k and the correspondences of iteration k+1 are two query sets of the same walk, so the lane needs one stream and the
// host one sync per iteration: [trimmed: move, LUT brackets and cuts of both sets] -> dual scan -> inlier cut / sums / covariance of pass
// k+1 -> sum (or trimmed selection) of the SSE -> sync.  Pass k+1 is speculative as before (the loop may end on the SSE of iteration
// k).  Same per-query results, same sums in the same order: the bits of the two-stream loop (FGOICP_ICP_DUAL=0; tests).
static int lane_icp_dual(fgoicp_ctx* c, fgoicp_ctx::IcpLane& L, const float* R0, const float* t0, size_t max_iter, float thr, float* sse_out, float* R_out9,
                         float* t_out3, int* iters_out) {
    HIPCHK(hipSetDevice(c->device));
    const int ns = (int)c->ns, nt = (int)c->nt;
    const bool seeding = c->icp_seeding, fused = icp_fused(c), trimmed = c->inliers != 0, skip = trimmed && c->trim_skip;
    hipStream_t S = L.stream;
    uint32_t* idx[2] = {L.d_first_idx, L.d_first_idx2};
    int cur = 0;
    HIPCHK(hipMemcpyAsync(L.d_work, c->d_src, sizeof(float4) * c->ns, hipMemcpyDeviceToDevice, S));
    launch_transform_inplace(L.d_work, ns, R0, t0, S);  // icp3d.cu:85
    Mat3f R = Mat3f::from(R0);
    Vec3f t{t0[0], t0[1], t0[2]};
    size_t iter = 0;
    float sse = kInf, last_sse = 2.0f * kInf;
    Mat3f last_R = Mat3f::identity();
    Vec3f last_t{0, 0, 0};
    int iters = 0;
    bool pending = false;
    if (max_iter > 0) {
        int rc = procrustes_enqueue(c, L, nullptr, idx[0], L.d_sel_wide2, S);
        if (rc) return rc;
        pending = true;
    }
    while (iter++ < max_iter && (last_sse - sse) > thr * last_sse) {  // icp3d.cu:94
        last_sse = sse;
        last_R = R;
        last_t = t;
        HIPCHK(wait_stream(S));  // first iteration: pass 1; later: a no-op (the SSE's sync below drained the stream)
        pending = false;
        Mat3f Rn;
        Vec3f tn;
        procrustes_finish(L, &Rn, &tn, nullptr, nullptr);
        const float tn3[3] = {tn.x, tn.y, tn.z};
        R = Rn * R;                                              // :101
        t = Rn * t + tn;                                         // :102
        const float t3[3] = {t.x, t.y, t.z};
        const uint32_t* seed = seeding ? idx[cur] : nullptr;
        const bool next = iter < max_iter;
        int rc = FGOICP_OK;
        if (next) {
            const float *lbA = nullptr, *lbB = nullptr;
            const uint32_t *uA = nullptr, *uB = nullptr;
            if (skip) {  // trimmed: the brackets need the moved cloud, so the move is its own kernel here (as in the two-stream loop)
                launch_transform_inplace(L.d_work, ns, Rn.m, tn3, S);  // :100
                launch_nn_prep(L.d_work, ns, c->d_lut, c->geom, nullptr, nullptr, 0, c->d_tgt, nt, seed, c->tgt_box6, L.d_nn_ub2, L.d_nn_lb2, S);
                launch_nn_prep(c->d_src, ns, c->d_lut, c->geom, R.m, t3, 1, c->d_tgt, nt, seed, c->tgt_box6, L.d_nn_ub, L.d_nn_lb, S);
                launch_trim_select(L.d_nn_ub2, ns, (int)c->inliers, nullptr, L.d_sel + 4, L.d_sel_wide2, S);
                launch_trim_select(L.d_nn_ub, ns, (int)c->inliers, nullptr, L.d_sel + 8, L.d_sel_wide, S);
                lbA = L.d_nn_lb2; uA = L.d_sel + 4; lbB = L.d_nn_lb; uB = L.d_sel + 8;
            }
            launch_nn_scan_dual(L.d_work, skip ? nullptr : Rn.m, skip ? nullptr : tn3, skip ? 0 : 1, c->d_src, R.m, t3, ns, c->bvh_tgt.view(), c->d_lut, c->geom, c->d_tgt, nt, seed,
                                lbA, uA, lbB, uB, idx[cur ^ 1], L.d_min_bits, skip ? nullptr : L.d_work, fused ? L.d_wsum : nullptr, fused ? L.hd_wsse : nullptr, S);
            rc = procrustes_enqueue(c, L, seed, idx[cur ^ 1], L.d_sel_wide2, S, nullptr, nullptr, 2);  // inlier cut, sums, covariance of pass k+1
            if (rc) return rc;
            rc = sse_enqueue(c, L, R.m, t3, seed, S, 2);                                                // sum / trimmed selection of the SSE
            if (rc) return rc;
            cur ^= 1;
            pending = true;
        } else {  // the last iteration the loop can make: no pass rides along
            launch_transform_inplace(L.d_work, ns, Rn.m, tn3, S);  // :100
            rc = sse_enqueue(c, L, R.m, t3, seed, S);              // :103
            if (rc) return rc;
        }
        HIPCHK(hipGetLastError());
        HIPCHK(wait_stream(S));
        sse = sse_result(c, L);
        ++iters;
    }
    if (pending) HIPCHK(hipStreamSynchronize(S));  // the speculative pass: drained, not used
    const bool cur_best = sse < last_sse;  // :106-107
    *sse_out = cur_best ? sse : last_sse;
    const Mat3f& Ro = cur_best ? R : last_R;
    const Vec3f& to = cur_best ? t : last_t;
    std::memcpy(R_out9, Ro.m, sizeof(Ro.m));
    t_out3[0] = to.x; t_out3[1] = to.y; t_out3[2] = to.z;
    if (iters_out) *iters_out = iters;
    return FGOICP_OK;
}

#ifdef FGOICP_DEV_KNOBS
// (FGOICP_ICP_GATED=1; measured SLOWER than paying the launches — 50-53 -> 53-56 us per iteration at 40k points: the command processor's
// wait-value poll and the two extra stream operations per iteration cost more than the launch latency they hide — so this is a knob, off by default.)
// The two-stream loop of small clouds with the launch latency taken off the iteration's chain (round 3).  An iteration's kernels depend on the
// host only through 24 floats (R_, t_, R, t): they are enqueued one iteration AHEAD, behind a stream wait (hipStreamWaitValue64) on a signal
// word, and read their motion from pinned memory; when the SVD is done the host writes the motion and raises the signal — the command
// processor releases kernels that are already queued instead of the host paying three launches (5 us each to submit, 6-8 us until the first
// one starts) between the SVD and the scan.  Same kernels, same arithmetic as lane_icp's fused path (the scans' `rt_dev` / `done` arguments of the
// device-resident loop are reused): bit-identical (tests).  The set enqueued behind a gate the loop never opens is released with `done` set.
static int lane_icp_gated(fgoicp_ctx* c, fgoicp_ctx::IcpLane& L, const float* R0, const float* t0, size_t max_iter, float thr, float* sse_out, float* R_out9,
                          float* t_out3, int* iters_out) {
    HIPCHK(hipSetDevice(c->device));
    const int ns = (int)c->ns, nt = (int)c->nt, nb = reduce_blocks_for(ns), groups = (ns + 63) / 64;
    constexpr int kRing = fgoicp_ctx::IcpLane::kRing;
    hipStream_t A = L.stream, B = L.icp_stream;
    const bool seeding = c->icp_seeding;
    uint32_t* idx[2] = {L.d_first_idx, L.d_first_idx2};
    const uint64_t base = L.gate_seq;  // gate j of this run opens at base + j
    *(volatile int*)L.h_done = 0;
    L.sse_on_host = true;              // where the gated kernels leave their results (as the fused enqueue functions would record)
    L.cov_on_host = true;
    L.cov_blocks = nb;
    HIPCHK(hipMemcpyAsync(L.d_work, c->d_src, sizeof(float4) * c->ns, hipMemcpyDeviceToDevice, A));
    launch_transform_inplace(L.d_work, ns, R0, t0, A);  // icp3d.cu:85
    Mat3f R = Mat3f::from(R0);
    Vec3f t{t0[0], t0[1], t0[2]};
    size_t iter = 0;
    float sse = kInf, last_sse = 2.0f * kInf;
    Mat3f last_R = Mat3f::identity();
    Vec3f last_t{0, 0, 0};
    int iters = 0;
    uint64_t enq = 0;     // highest gate with kernels queued behind it
    uint64_t opened = 0;  // highest gate raised
    // pass j+1 and the SSE of iteration j, queued behind gate j; pass j writes idx[j & 1] and is seeded by idx[(j - 1) & 1]
    auto enqueue_gated = [&](uint64_t j) -> int {
        const float* rt = L.hd_rt + 24 * (j & 1);
        const uint32_t* seed = seeding ? idx[j & 1] : nullptr;
        enq = j;  // from here on something may be waiting behind gate j: release_all() has to raise it whatever happens below
        HIPCHK(hipStreamWaitValue64(B, L.sig_b, base + j, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull));
        launch_nn_scan(L.d_work, ns, c->bvh_tgt.view(), c->d_lut, c->geom, nullptr, nullptr, 1, 1, c->d_tgt, nt, seed, nullptr, nullptr, idx[(j + 1) & 1], B, L.d_work, rt, L.hd_done, L.d_wsum);
        launch_icp_cov_cen(L.d_work, c->d_tgt, idx[(j + 1) & 1], ns, nt, L.d_wsum, nb, groups, L.hd_cen, L.hd_covbp, nb, B, L.hd_done);
        HIPCHK(hipEventRecord(L.ev_step[(j + 1) % kRing], B));
        HIPCHK(hipStreamWaitValue64(A, L.sig_a, base + j, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull));
        launch_nn_scan(c->d_src, ns, c->bvh_tgt.view(), c->d_lut, c->geom, nullptr, nullptr, 1, 0, c->d_tgt, nt, seed, nullptr, nullptr, L.d_min_bits, A, nullptr, rt + 12, L.hd_done, L.hd_wsse);
        HIPCHK(hipEventRecord(L.ev_sse[j % kRing], A));
        HIPCHK(hipGetLastError());
        return FGOICP_OK;
    };
    auto open_gate = [&](uint64_t j) {
        __atomic_store_n(L.sig_b, base + j, __ATOMIC_RELEASE);
        __atomic_store_n(L.sig_a, base + j, __ATOMIC_RELEASE);
        opened = j;
    };
    int rc = FGOICP_OK;
    if (max_iter > 0) {
        HIPCHK(hipEventRecord(L.icp_ev_w, A));
        HIPCHK(hipStreamWaitEvent(B, L.icp_ev_w, 0));
        rc = procrustes_enqueue(c, L, nullptr, idx[1], L.d_sel_wide2, B);  // pass 1 (no gate: its motion is (R0, t0), applied above)
        if (rc == FGOICP_OK) HIPCHK(hipEventRecord(L.ev_step[1 % kRing], B));
        if (rc == FGOICP_OK && max_iter > 1) rc = enqueue_gated(1);
    }
    // every exit below has to leave no kernel waiting behind a closed gate
    auto release_all = [&]() {
        if (enq > opened) { *(volatile int*)L.h_done = 1; __atomic_thread_fence(__ATOMIC_SEQ_CST); open_gate(enq); }
        (void)hipStreamSynchronize(B);
        (void)hipStreamSynchronize(A);
        L.gate_seq = base + (enq > opened ? enq : opened) + 1;
        *(volatile uint64_t*)L.sig_b = L.gate_seq;
        *(volatile uint64_t*)L.sig_a = L.gate_seq;
        *(volatile int*)L.h_done = 0;
    };
    if (rc) { release_all(); return rc; }
    while (iter++ < max_iter && (last_sse - sse) > thr * last_sse) {  // icp3d.cu:94
        last_sse = sse;
        last_R = R;
        last_t = t;
        if (hipEventSynchronize(L.ev_step[iter % kRing]) != hipSuccess) { release_all(); set_error("hipEventSynchronize failed in the ICP loop"); return FGOICP_ERR_HIP; }
        Mat3f Rn;
        Vec3f tn;
        procrustes_finish(L, &Rn, &tn, nullptr, nullptr);
        const float tn3[3] = {tn.x, tn.y, tn.z};
        R = Rn * R;                                              // :101
        t = Rn * t + tn;                                         // :102
        const float t3[3] = {t.x, t.y, t.z};
        if (iter < max_iter) {
            float* slot = L.h_rt + 24 * (iter & 1);
            std::memcpy(slot, Rn.m, 36); std::memcpy(slot + 9, tn3, 12);
            std::memcpy(slot + 12, R.m, 36); std::memcpy(slot + 21, t3, 12);
            open_gate(iter);                                      // pass iter+1 and SSE iter start now: they were queued an iteration ago
            if (iter + 1 < max_iter) { rc = enqueue_gated(iter + 1); if (rc) { release_all(); return rc; } }
        } else {  // the last iteration the loop can make: nothing was queued for it
            launch_transform_inplace(L.d_work, ns, Rn.m, tn3, B);  // :100
            rc = sse_enqueue(c, L, R.m, t3, seeding ? idx[iter & 1] : nullptr, A);
            if (rc) { release_all(); return rc; }
            HIPCHK(hipEventRecord(L.ev_sse[iter % kRing], A));
        }
        if (hipEventSynchronize(L.ev_sse[iter % kRing]) != hipSuccess) { release_all(); set_error("hipEventSynchronize failed in the ICP loop"); return FGOICP_ERR_HIP; }
        sse = sse_result(c, L);
        ++iters;
    }
    release_all();  // the speculative pass (and a set behind a gate the loop never opened): drained, not used
    const bool cur_best = sse < last_sse;  // :106-107
    *sse_out = cur_best ? sse : last_sse;
    const Mat3f& Ro = cur_best ? R : last_R;
    const Vec3f& to = cur_best ? t : last_t;
    std::memcpy(R_out9, Ro.m, sizeof(Ro.m));
    t_out3[0] = to.x; t_out3[1] = to.y; t_out3[2] = to.z;
    if (iters_out) *iters_out = iters;
    return FGOICP_OK;
}

#endif  // FGOICP_DEV_KNOBS (gated loop)

// One ICP run on a lane of its own (lane >= 1: own scratch, own two streams — nothing of it queues on the context's main stream,
// where the bounds kernels run): the late-joining refinement of the ROUND schedule (driver.hpp) calls this from a background
// host thread while the main thread keeps submitting bounds ticks.  Same kernels and sums as ctx_icp.
int ctx_icp_lane(fgoicp_ctx* c, int lane, const float* R0, const float* t0, size_t max_iter, float thr, float* sse_out, float* R_out9, float* t_out3, int* iters_out) {
    if (lane < 0 || lane >= (int)c->lanes.size() || c->brute_force_nn) lane = 0;  // the brute-force kernels keep their scratch on lane 0
    return lane_icp(c, c->lanes[(size_t)lane], R0, t0, max_iter, thr, sse_out, R_out9, t_out3, iters_out);
}

// Several ICP runs at once (the triggers of one expansion round, driver.hpp): run i goes to lane i % lanes, every lane has its
// own scratch and streams and a host thread of its own (an ICP iteration is a chain of small kernels with two host syncs — at
// 40k points one run fills a fraction of the device).  Each run is exactly the run ctx_icp would do: same kernels, same sums.
int ctx_icp_batch(fgoicp_ctx* c, int n, const float* R0s, const float* t0s, size_t max_iter, float thr, float* sse_out, float* R_out9s, float* t_out3s, int* iters_out) {
    const int nl = std::min<int>(n, (int)c->lanes.size());
    if (n <= 0) return FGOICP_OK;
    if (nl <= 1 || c->brute_force_nn) {
        for (int i = 0; i < n; ++i) {
            int rc = lane_icp(c, c->lanes[0], R0s + 9 * i, t0s + 3 * i, max_iter, thr, sse_out + i, R_out9s + 9 * i, t_out3s + 3 * i, iters_out ? iters_out + i : nullptr);
            if (rc) return rc;
        }
        return FGOICP_OK;
    }
    std::vector<int> rcs(nl, FGOICP_OK);
    std::vector<std::string> errs(nl);
    std::vector<std::thread> th;
    for (int l = 0; l < nl; ++l)
        th.emplace_back([&, l] {
            for (int i = l; i < n && rcs[l] == FGOICP_OK; i += nl)
                rcs[l] = lane_icp(c, c->lanes[l], R0s + 9 * i, t0s + 3 * i, max_iter, thr, sse_out + i, R_out9s + 9 * i, t_out3s + 3 * i, iters_out ? iters_out + i : nullptr);
            if (rcs[l]) errs[l] = g_last_error;
        });
    for (auto& t : th) t.join();
    for (int l = 0; l < nl; ++l)
        if (rcs[l]) { set_error(errs[l]); return rcs[l]; }
    return FGOICP_OK;
}

// EXTENSION: trimmed Go-ICP.  k = 0 (or k >= ns) switches trimming off.
int ctx_set_inliers(fgoicp_ctx* c, size_t k) {
    HIPCHK(hipSetDevice(c->device));
    if (k >= c->ns) k = 0;
    if (k && !c->sorted_bounds) { set_error("trimming needs the sorted bounds path (FGOICP_BOUNDS_SORTED=0 is set)"); return FGOICP_ERR_INVALID_ARG; }
    if (c->slots[0].inflight || c->slots[1].inflight) { set_error("fgoicp_ctx_set_inliers: a bounds submission is in flight"); return FGOICP_ERR_INVALID_ARG; }
    if (k && !c->trim_ready) {
        // per-point e of one window, per slot: up to 12 GiB (a sixth of what is free): 3000 subcubes of a 1M-point cloud per
        // window — the selection launches one workgroup per row and wants several hundred of them
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        const size_t budget = std::max<size_t>((size_t)3 << 29, std::min<size_t>((size_t)12 << 30, free_b / 6));
        if (const char* e = dev_env("FGOICP_TRIM_SAMPLE")) c->trim_samp_shift = std::max(0, std::min(10, std::atoi(e)));  // tuning knob: 0 = two-pass selection, no sample
        if (const char* e = dev_env("FGOICP_TRIM_MARGIN")) c->trim_margin_sd = (float)std::atof(e);                          // tuning knob: bracket half-width (standard deviations)
        c->erow = (c->ns + 3) & ~(size_t)3;  // rows start 16-byte aligned
        if (c->trim_samp_shift > 0)  // + the row's sample (trim_store); rows and samples then start on 256-byte boundaries
            c->erow = ((c->ns + 63) & ~(size_t)63) + (((((c->ns + ((size_t)1 << c->trim_samp_shift) - 1) >> c->trim_samp_shift)) + 63) & ~(size_t)63);
        if (!c->d_trim_stat) { HIPCHK(hipMalloc(&c->d_trim_stat, sizeof(unsigned long long) * 4)); HIPCHK(hipMemset(c->d_trim_stat, 0, sizeof(unsigned long long) * 4)); }
        size_t rows = budget / (sizeof(float) * c->erow);
        rows = std::max<size_t>(1, std::min<size_t>(rows, (size_t)c->max_subcubes));
        c->vals_rows = (int)rows;
        // every pointer is guarded on its own: a call that failed half-way (out of memory) can be repeated without leaking
        for (auto& sl : c->slots) if (!sl.d_evals) HIPCHK(hipMalloc(&sl.d_evals, sizeof(float) * c->erow * rows));
        for (auto& L : c->lanes) {
            if (!L.d_d2) HIPCHK(hipMalloc(&L.d_d2, sizeof(float) * c->ns));
            if (!L.d_nn_lb) HIPCHK(hipMalloc(&L.d_nn_lb, sizeof(float) * c->ns));
            if (!L.d_nn_ub) HIPCHK(hipMalloc(&L.d_nn_ub, sizeof(float) * c->ns));
            if (!L.d_nn_lb2) HIPCHK(hipMalloc(&L.d_nn_lb2, sizeof(float) * c->ns));
            if (!L.d_nn_ub2) HIPCHK(hipMalloc(&L.d_nn_ub2, sizeof(float) * c->ns));
            if (!L.d_sel) HIPCHK(hipMalloc(&L.d_sel, sizeof(uint32_t) * 16));
            if (!L.d_eq) HIPCHK(hipMalloc(&L.d_eq, sizeof(uint32_t)));
            const char* e = dev_env("FGOICP_SELECT_WIDE");  // tuning knob: 0 = always the one-block selection for single rows
            if (!(e && std::atoi(e) == 0)) {
                if (!L.d_sel_wide) HIPCHK(hipMalloc(&L.d_sel_wide, 65536));
                if (!L.d_sel_wide2) HIPCHK(hipMalloc(&L.d_sel_wide2, 65536));
            }
            if (!L.d_use) HIPCHK(hipMalloc(&L.d_use, c->ns));
            if (!L.h_trim) {
                HIPCHK(hipHostMalloc((void**)&L.h_trim, sizeof(float) * 4, hipHostMallocMapped));
                HIPCHK(hipHostGetDevicePointer((void**)&L.hd_trim, L.h_trim, 0));
            }
        }
        if (!c->d_orig_of_slot) {
            HIPCHK(hipMalloc(&c->d_orig_of_slot, sizeof(uint32_t) * c->ns));
            HIPCHK(hipMemcpy(c->d_orig_of_slot, c->perm.data(), sizeof(uint32_t) * c->ns, hipMemcpyHostToDevice));
        }
        c->trim_ready = true;
    }
    c->inliers = k;
    if (k && c->trim_samp_shift > 0) {
        // the cut's rank in a sample of m points: binomial standard deviation sqrt(m p (1 - p)) if the sample were random (a systematic
        // sample of a Hilbert-ordered surface is tighter); the bracket spans trim_margin_sd of them either side, at least 2 ranks
        const double m = (double)((c->ns + ((size_t)1 << c->trim_samp_shift) - 1) >> c->trim_samp_shift), p = (double)k / (double)c->ns;
        c->trim_margin = 2 + (int)std::ceil((double)c->trim_margin_sd * std::sqrt(m * p * (1.0 - p)));
    }
    return FGOICP_OK;
}

}  // namespace fgoicp

using namespace fgoicp;

extern "C" {

const char* fgoicp_last_error(void) { return g_last_error.c_str(); }
const char* fgoicp_version(void) { return kDevKnobs ? "fgoicp_amd 0.4 (gfx950, development build)" : "fgoicp_amd 0.4 (gfx950)"; }
int fgoicp_dev_knobs(void) { return kDevKnobs ? 1 : 0; }
int fgoicp_abi_version(void) { return FGOICP_ABI_VERSION; }

static int ctx_create_impl(const float* tgt_xyz, size_t nt, const float* src_xyz, size_t ns, const float* bounds6, float lut_resolution, int device, unsigned flags,
                           fgoicp_ctx** out, fgoicp_ctx** partial);
int fgoicp_ctx_create(const float* tgt_xyz, size_t nt, const float* src_xyz, size_t ns, const float* bounds6, float lut_resolution,
                      int device, unsigned flags, fgoicp_ctx** out) {
    if (!out) return FGOICP_ERR_INVALID_ARG;
    *out = nullptr;
    fgoicp_ctx* partial = nullptr;  // what an exception (std::bad_alloc in a host-side table, a thread that cannot start) leaves half-built
    const int rc = fgoicp::abi_guard("fgoicp_ctx_create", [&] { return ctx_create_impl(tgt_xyz, nt, src_xyz, ns, bounds6, lut_resolution, device, flags, out, &partial); });
    if (rc != FGOICP_OK && partial && !*out) fgoicp_ctx_destroy(partial);
    return rc;
}
static int ctx_create_impl(const float* tgt_xyz, size_t nt, const float* src_xyz, size_t ns, const float* bounds6, float lut_resolution, int device, unsigned flags,
                           fgoicp_ctx** out, fgoicp_ctx** partial) {
    if (!tgt_xyz || !src_xyz || !bounds6 || nt == 0 || ns == 0 || !(lut_resolution > 0) || nt > 0x7ffffffeull || ns > 0x7ffffffeull) {
        set_error("fgoicp_ctx_create: invalid argument");
        return FGOICP_ERR_INVALID_ARG;
    }
    // non-finite coordinates: the reference would carry them into every sum (NaN bounds, a search that never prunes); refused here
    auto finite_cloud = [](const float* p, size_t n) {
        float acc = 0.0f;
        for (size_t i = 0; i < 3 * n; ++i) acc += p[i] * 0.0f;  // stays 0 unless some coordinate is NaN or infinite
        return acc == 0.0f;
    };
    if (!finite_cloud(tgt_xyz, nt) || !finite_cloud(src_xyz, ns) || !finite_cloud(bounds6, 2)) {
        set_error("fgoicp_ctx_create: a cloud (or the target bounds) holds a non-finite coordinate");
        return FGOICP_ERR_INVALID_ARG;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_error(std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                  "); fgoicp_amd has no CPU path");
        return FGOICP_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        set_error("fgoicp_ctx_create: device ordinal out of range");
        return FGOICP_ERR_INVALID_ARG;
    }
    HIPCHK(hipSetDevice(device));
    fgoicp_ctx* c = new fgoicp_ctx();
    *partial = c;
    c->device = device;
    c->ns = ns;
    c->nt = nt;
    c->profile = (flags & FGOICP_FLAG_PROFILE) != 0;
    c->brute_force_nn = (flags & FGOICP_FLAG_BRUTE_FORCE_NN) != 0;
    std::memcpy(c->bounds6, bounds6, sizeof(c->bounds6));
    // The trimmed search drops queries whose distance to the target's box already exceeds the cut (nn_prep_kernel): that box is taken
    // from the DATA, not from the caller's `bounds6` — the reference's Registration accepts any target_bounds (they only place the
    // LUT, registration.hpp:68), and with cropped bounds "every target point lies inside the box" would be false.
    for (int a = 0; a < 3; ++a) { c->tgt_box6[2 * a] = tgt_xyz[a]; c->tgt_box6[2 * a + 1] = tgt_xyz[a]; }
    for (size_t i = 1; i < nt; ++i)
        for (int a = 0; a < 3; ++a) {
            const float v = tgt_xyz[3 * i + a];
            c->tgt_box6[2 * a] = std::min(c->tgt_box6[2 * a], v);
            c->tgt_box6[2 * a + 1] = std::max(c->tgt_box6[2 * a + 1], v);
        }
    if (const char* e = dev_env("FGOICP_TRIM_SKIP")) c->trim_skip = std::atoi(e) != 0;        // tuning knob
    if (const char* e = dev_env("FGOICP_SORT_XCD")) c->sort_xcd = std::atoi(e) != 0;          // tuning knob
    if (const char* e = dev_env("FGOICP_SORT_CHECK")) c->sort_check = std::atoi(e) != 0;      // tuning knob: 0 = no permutation check of the tick sort
    auto fail = [&](int rc) { fgoicp_ctx_destroy(c); *partial = nullptr; return rc; };
#define CHK(expr) do { hipError_t e2_ = (expr); if (e2_ != hipSuccess) { set_error(std::string(#expr) + " failed: " + hipGetErrorString(e2_)); return fail(e2_ == hipErrorOutOfMemory ? FGOICP_ERR_OOM : FGOICP_ERR_HIP); } } while (0)
    CHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));

    // LUT geometry — NearestNeighborLUT ctor, registration.cu:186-204
    LutGeom& g = c->geom;
    g.resolution = lut_resolution;
    g.dx = (int)std::ceil((bounds6[1] - bounds6[0]) / lut_resolution);
    g.dy = (int)std::ceil((bounds6[3] - bounds6[2]) / lut_resolution);
    g.dz = (int)std::ceil((bounds6[5] - bounds6[4]) / lut_resolution);
    if (g.dx < 1 || g.dy < 1 || g.dz < 1 || g.dx > 4094 || g.dy > 4094 || g.dz > 4094) {
        set_error("fgoicp_ctx_create: LUT dims out of range (" + std::to_string(g.dx) + "," + std::to_string(g.dy) + "," + std::to_string(g.dz) + ")");
        return fail(FGOICP_ERR_INVALID_ARG);
    }
    g.px = g.dx + 2; g.py = g.dy + 2; g.pz = g.dz + 2;
    g.scale = 1.0f / lut_resolution;
    g.off_x = -bounds6[0]; g.off_y = -bounds6[2]; g.off_z = -bounds6[4];
    g.quantize = (flags & FGOICP_FLAG_NO_WEIGHT_QUANT) ? 0 : 1;
    g.idx = nullptr;  // set once the index LUT has been built (below)

    // clouds: float4 on the device.  Source: Morton order, w = x*x+y*y+z*z in the device
    // contraction order (registration.cu:39-41).  Target: caller order (index tie rule), w = 0.
    c->perm.resize(ns);
    if (flags & FGOICP_FLAG_NO_MORTON) std::iota(c->perm.begin(), c->perm.end(), 0u);
    else {
        // k-d order with runs of 64 points (a wave of the exact scan; any chunk of 64 << k consecutive points is one k-d cell,
        // morton.hpp) unless the caller asks for the Hilbert curve (FGOICP_FLAG_CURVE_ORDER).  Measured against the curve (round 3,
        // profiles/r03_ab_kd_order.txt): bounds kernel 1625 -> 1478 us per launch on the bunny shape (a chunk's LUT footprint is one
        // compact cell instead of a run that straddles curve cells), 5540 -> 5324 us on the dragon shape, ICP iteration 222 -> 169 us
        // at 437k points; with 20 % volume outliers (1M trimmed) the cells grow tails along the surface normal and the trimmed ICP is
        // 13 % SLOWER — hence the flag.  FGOICP_POINT_CURVE overrides: 2 = k-d, 1 = Hilbert, 0 = Z-order.
        static const int forced = [] { const char* e = dev_env("FGOICP_POINT_CURVE"); return e ? std::atoi(e) : -1; }();  // tuning knob
        const int mode = forced >= 0 ? forced : ((flags & FGOICP_FLAG_CURVE_ORDER) ? 1 : 2);
        c->perm = point_order(src_xyz, ns, 3, 64, mode);
        c->source_order = mode;
    }
    {
        std::vector<float4> h(ns);
        for (size_t i = 0; i < ns; ++i) {
            const float* p = src_xyz + 3 * (size_t)c->perm[i];
            h[i] = make_float4(p[0], p[1], p[2], std::fmaf(p[2], p[2], std::fmaf(p[1], p[1], p[0] * p[0])));
        }
        CHK(hipMalloc(&c->d_src, sizeof(float4) * ns));
        CHK(hipMemcpy(c->d_src, h.data(), sizeof(float4) * ns, hipMemcpyHostToDevice));
    }
    // target on the device (caller order) + exact-NN tree; LUT build — buildLUTKernel, registration.cu:258-318
    {
        std::vector<float4> h(nt), hs(nt);
        for (size_t i = 0; i < nt; ++i) {
            const float* p = tgt_xyz + 3 * i;
            h[i] = make_float4(p[0], p[1], p[2], 0.f);
            hs[i] = make_float4(p[0] + g.off_x, p[1] + g.off_y, p[2] + g.off_z, 0.f);  // registration.cu:289-296
        }
        CHK(hipMalloc(&c->d_tgt, sizeof(float4) * nt));
        CHK(hipMemcpy(c->d_tgt, h.data(), sizeof(float4) * nt, hipMemcpyHostToDevice));
        const size_t total = (size_t)g.px * g.py * g.pz;
        CHK(hipMalloc(&c->d_lut, total * sizeof(float)));
        hipError_t e3 = hipSuccess;
        if (c->brute_force_nn) {
            float4* d_tgt_shift = nullptr;
            e3 = hipMalloc(&d_tgt_shift, sizeof(float4) * nt);
            if (e3 == hipSuccess) e3 = hipMemcpy(d_tgt_shift, hs.data(), sizeof(float4) * nt, hipMemcpyHostToDevice);
            if (e3 == hipSuccess) {
                launch_lut_build(d_tgt_shift, (int)nt, g, c->d_lut, c->stream);
                e3 = hipGetLastError();
                if (e3 == hipSuccess) e3 = hipStreamSynchronize(c->stream);
            }
            (void)hipFree(d_tgt_shift);
        } else {
            std::vector<uint32_t> order;  // one sort for both trees
            c->tree_order = bvh_kd_order() ? 1 : 0;
            CHK(bvh_upload(bvh_build_host(h.data(), nt, &order), &c->bvh_tgt));
            BvhDevice shifted;  // the LUT is built from the SHIFTED targets (pc + offset in fp32), its own tree
            e3 = bvh_upload(bvh_build_host(hs.data(), nt, &order), &shifted);
            float* scratch = nullptr;
            if (e3 == hipSuccess) e3 = hipMalloc(&scratch, total * sizeof(float));
            // the index LUT (4 B per node; an allocation that fails just leaves the feature off).  Built in round 3, bit-identical (146 GPU
            // tests with it on), measured — and it buys NOTHING (profiles/r03_ab_lut_index.txt: ICP iteration 44.7 -> 46.4 us at 40k points,
            // 169 -> 170 us at 437k, trimmed 1M ICP 463 -> 465 ms): the triangle bound's slack is not what makes the scans of a far state
            // expensive — the leaf BOXES a large ball cuts are (tools/kd_sim.py-style count: 5.4 -> 16 leaves per query group 20 degrees
            // off the optimum even with exact bounds).  OFF by default; FGOICP_LUT_INDEX=1 builds and uses it.
            static const bool want_idx = [] { const char* e = dev_env("FGOICP_LUT_INDEX"); return e && std::atoi(e) != 0; }();  // tuning knob / A-B
            if (e3 == hipSuccess && want_idx && hipMalloc(&c->d_lut_idx, total * sizeof(uint32_t)) != hipSuccess) { c->d_lut_idx = nullptr; (void)hipGetLastError(); }
            if (e3 == hipSuccess) {
                launch_lut_build_scan(shifted.view(), g, scratch, c->d_lut, c->stream, c->d_lut_idx);
                e3 = hipGetLastError();
                if (e3 == hipSuccess) e3 = hipStreamSynchronize(c->stream);
            }
            (void)hipFree(scratch);
            bvh_free(&shifted);
        }
        if (e3 != hipSuccess) { set_error(std::string("LUT build failed: ") + hipGetErrorString(e3)); return fail(e3 == hipErrorOutOfMemory ? FGOICP_ERR_OOM : FGOICP_ERR_HIP); }
        g.idx = c->d_lut_idx;  // from here on the exact scans seed their bounds with it (kernels.hip lut_upper_bound_d2)
        // Packed copy for the bounds kernel: 0 none, 1 z-pair (2x bytes, two rows per lookup), 2 yz-quad (4x bytes, one
        // line per lookup).  Measured: the quad wins on sparse clouds (every lane-gather its own line; +8 % at 40k
        // points), loses on dense ones (lanes share lines and the 4x footprint falls out of cache; -5 % at 437k), so it
        // is chosen by the number of source points per voxel of the LUT's projected faces.
        const double face_voxels = (double)g.dx * g.dy + (double)g.dy * g.dz + (double)g.dx * g.dz;
        int layout = ((double)ns / face_voxels < 0.5 && total * sizeof(float4) <= ((size_t)16 << 30)) ? 2 : 1;
        // sparse clouds, round 3: the apron-bricked quads (kernels.hip apron_index: every lookup inside one line, a line serves a 3 x 2
        // patch of base voxels; 21.3 instead of 16 B per node) — bounds kernel -1.1 % on the bunny shape, three A/B pairs
        const size_t apron_bytes = (size_t)((g.px + 2) / 3) * ((g.py + 1) / 2) * g.pz * 8 * sizeof(float4);
        if (layout == 2 && apron_bytes <= ((size_t)16 << 30)) layout = 4;
        if (const char* e = dev_env("FGOICP_LUT_ZPAIR")) layout = std::atoi(e);  // tuning knob
        const char* units_env = dev_env("FGOICP_UNITS");
        const bool units_on = units_env && (std::atoi(units_env) == 4 || std::atoi(units_env) == 8);
        if (layout == 3 && (g.px > 1023 || g.py > 1023 || g.pz > 1023 || c->inliers)) layout = 2;  // the bricked copy packs indices in 10 bits
        if (layout == 4 && (g.px > 1023 || g.py > 1023 || g.pz > 1023 || units_on)) layout = 2;     // the apron copy too; no sibling-unit kernel for it
        c->lut_layout = layout;
        if (layout == 4) {
            const size_t lines = (size_t)((g.px + 2) / 3) * ((g.py + 1) / 2) * g.pz;
            CHK(hipMalloc(&c->d_lut_zp, lines * 8 * sizeof(float4)));
            launch_lut_quad_apron(c->d_lut, g, reinterpret_cast<float4*>(c->d_lut_zp), c->stream);
        } else if (layout == 3) {
            const size_t bricks = (size_t)((g.px + 3) / 4) * ((g.py + 3) / 4) * ((g.pz + 3) / 4);
            CHK(hipMalloc(&c->d_lut_zp, bricks * 64 * sizeof(float4)));
            launch_lut_quad_bricked(c->d_lut, g, reinterpret_cast<float4*>(c->d_lut_zp), c->stream);
        } else if (layout == 2) {
            CHK(hipMalloc(&c->d_lut_zp, total * sizeof(float4)));
            launch_lut_quad(c->d_lut, g, reinterpret_cast<float4*>(c->d_lut_zp), c->stream);
        } else if (layout == 1) {
            CHK(hipMalloc(&c->d_lut_zp, total * sizeof(float2)));
            launch_lut_zpair(c->d_lut, g, c->d_lut_zp, c->stream);
        }
        if (layout) {
            CHK(hipGetLastError());
            CHK(hipStreamSynchronize(c->stream));
        }
    }
    // bounds scratch: P points per thread so that one 32-subcube launch has >= ~2048 blocks
    {
        int P = 8;
        while (P > 1 && ((ns + (size_t)kBlock * P - 1) / ((size_t)kBlock * P)) * kMaxBatch < 2048) P >>= 1;
        if (const char* e = dev_env("FGOICP_PTS_PER_THREAD")) {  // tuning knob: 1, 2, 4 or 8
            const int v = std::atoi(e);
            if (v == 1 || v == 2 || v == 4 || v == 8) P = v;
        }
        c->pts_per_thread = P;
        c->nchunk = (int)((ns + (size_t)kBlock * P - 1) / ((size_t)kBlock * P));
        // subcubes per window: as many as 4 GiB of per-(subcube, chunk) scratch per slot hold, 4096..131072 (wide rounds submit
        // tens of thousands per tick; bigger launches sort into longer runs of items per LUT cell and re-use the L2 better)
        {
            // points per work item of the sorted path: the patch of 256 Morton-consecutive points of a sparse cloud spans
            // ~28 voxels; in a dense cloud (>= 1 point per voxel of the LUT's faces) it shrinks to ~6, so the items grow to
            // 2048 points — 8x fewer items to sort, 8x fewer partials, same locality.  Measured (437k points, density 1.8,
            // one certify run): 256 / 512 / 1024 / 2048 / 4096 points per item -> 4.68 / 3.34 / 2.70 / 2.48 / 2.60 s;
            // 40k points, density 0.1: 256 -> 512 points 8 % slower.
            {
                const double face_voxels = (double)g.dx * g.dy + (double)g.dy * g.dz + (double)g.dx * g.dz;
                const double density = (double)ns / face_voxels;
                c->chunk_pts = density >= 1.0 ? 2048 : density >= 0.5 ? 1024 : density >= 0.25 ? 512 : 256;
                if (const char* e = dev_env("FGOICP_CHUNK_PTS")) {  // tuning knob
                    const int v = std::atoi(e);
                    if (v == 64 || v == 128 || v == 256 || v == 512 || v == 1024 || v == 2048 || v == 4096) c->chunk_pts = v;  // 64 / 128 need FGOICP_BOUNDS_VARIANT 4 / 3
                }
            }
            const size_t nchunk1 = (ns + c->chunk_pts - 1) / c->chunk_pts;
            const size_t fit = ((size_t)4 << 30) / (nchunk1 * (sizeof(double2) + 2 * sizeof(unsigned) + sizeof(unsigned short)));
            c->max_subcubes = (int)std::max<size_t>(4096, std::min<size_t>(131072, fit));
            if (const char* e = dev_env("FGOICP_MAX_SUBCUBES")) c->max_subcubes = std::max(kMaxBatch, std::min(1 << 18, std::atoi(e)));  // tuning knob: subcubes per window
            // one launch = one workgroup per (subcube, chunk) item: keep items x 256 threads inside the 32-bit grid
            const size_t launch_fit = (((size_t)1 << 24) - 1) / nchunk1;
            if (launch_fit < (size_t)kMaxBatch) {
                set_error("fgoicp_ctx_create: source cloud too large for one bounds launch per batch (ns > ~134M)");
                return fail(FGOICP_ERR_TOO_LARGE);
            }
            c->max_subcubes = (int)std::min<size_t>((size_t)c->max_subcubes, launch_fit);
        }
        CHK(hipMalloc(&c->d_partials, sizeof(double2) * (size_t)c->max_subcubes * c->nchunk));
        CHK(hipHostMalloc((void**)&c->h_lb, sizeof(float) * c->max_subcubes, hipHostMallocMapped));
        CHK(hipHostMalloc((void**)&c->h_ub, sizeof(float) * c->max_subcubes, hipHostMallocMapped));
        CHK(hipHostGetDevicePointer((void**)&c->hd_lb, c->h_lb, 0));
        CHK(hipHostGetDevicePointer((void**)&c->hd_ub, c->h_ub, 0));
    }
    // locality-sorted whole-tick path
    {
        if (const char* e = dev_env("FGOICP_BOUNDS_SORTED")) c->sorted_bounds = std::atoi(e) != 0;
        c->nchunk1 = (int)((ns + c->chunk_pts - 1) / c->chunk_pts);
        c->max_groups = std::max(512, c->max_subcubes / 8);
        if (const char* e = dev_env("FGOICP_FINALIZE_SIDE")) c->finalize_on_side = std::atoi(e) != 0;  // tuning knob
        if (const char* e = dev_env("FGOICP_ICP_SEED")) c->icp_seeding = std::atoi(e) != 0;             // tuning knob
        if (const char* e = dev_env("FGOICP_COOP_SPLIT_MIN")) c->coop_split_min = c->coop_split_trim_min = (size_t)std::max(0L, std::atol(e));  // tuning knob (both thresholds)
        if (const char* e = dev_env("FGOICP_UNITS")) { const int v = std::atoi(e); c->unit_m = (v == 4 || v == 8) ? v : 0; }  // tuning knob: siblings per work item
        if (c->lut_layout == 4) c->unit_m = 0;  // the apron layout has no sibling-unit kernel
        if (const char* e = dev_env("FGOICP_SMALL_TICK")) c->small_tick_items = std::max(0, std::atoi(e));  // tuning knob: items
        int maxd = std::max(g.dx, std::max(g.dy, g.dz));
        c->cell_shift = 0;
        while ((maxd >> c->cell_shift) > 32) ++c->cell_shift;  // 5 bits per axis
        // [0, n): centres of the runs of `pts` consecutive points; [n, 2 n): their normals (direction of least variance)
        const auto run_centres = [&](size_t pts) {
            const size_t n = (ns + pts - 1) / pts;
            std::vector<float4> cen(2 * n);
            for (size_t k = 0; k < n; ++k) {
                double sx = 0, sy = 0, sz = 0;
                const size_t a = k * pts, b = std::min(ns, a + pts);
                for (size_t i = a; i < b; ++i) {
                    const float* p = src_xyz + 3 * (size_t)c->perm[i];
                    sx += p[0]; sy += p[1]; sz += p[2];
                }
                const double inv = 1.0 / (double)(b - a);
                const double m[3] = {sx * inv, sy * inv, sz * inv};
                cen[k] = make_float4((float)m[0], (float)m[1], (float)m[2], 0.f);
                double H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, U[3][3], S[3], V[3][3];
                for (size_t i = a; i < b; ++i) {
                    const float* p = src_xyz + 3 * (size_t)c->perm[i];
                    const double d[3] = {p[0] - m[0], p[1] - m[1], p[2] - m[2]};
                    for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) H[r][q] += d[r] * d[q];
                }
                svd3_jacobi(H, U, S, V);  // symmetric positive semi-definite: the last column belongs to the smallest eigenvalue
                cen[n + k] = make_float4((float)V[0][2], (float)V[1][2], (float)V[2][2], 0.f);
            }
            return cen;
        };
        {
            const std::vector<float4> cen = run_centres((size_t)c->chunk_pts);
            CHK(hipMalloc(&c->d_chunk_cen, sizeof(float4) * cen.size()));
            CHK(hipMemcpy(c->d_chunk_cen, cen.data(), sizeof(float4) * cen.size(), hipMemcpyHostToDevice));
        }
        // Windows with thresholds (fgoicp_bounds_submit_cut) on sparse clouds: two chunks per work item.  Half of such a window's items end
        // after three loads, and what they cost is their workgroup's dispatch (profiles/r04_dispatch_rate.txt); measured on the bunny
        // shape with the early exit, 256 / 512 / 1024 points per item: 243 / 220 / 245 ms per certify run — while without thresholds
        // 512 points are 8 % slower than 256 (above).  The sums stay per chunk, so both kinds of window return the same bits.
        c->cut_span = c->chunk_pts == 256 ? 2 : 1;
        if (const char* e = dev_env("FGOICP_CUT_TIERS")) c->cut_tier_level = (float)std::atof(e);  // tuning knob: 0 = one tier
        if (const char* e = dev_env("FGOICP_CUT_SPAN")) { const int v = std::atoi(e); if (v == 1 || v == 2 || v == 4) c->cut_span = v; }  // tuning knob
        if (c->cut_span > 1) {
            const std::vector<float4> cen = run_centres((size_t)c->chunk_pts * c->cut_span);
            CHK(hipMalloc(&c->d_span_cen, sizeof(float4) * cen.size()));
            CHK(hipMemcpy(c->d_span_cen, cen.data(), sizeof(float4) * cen.size(), hipMemcpyHostToDevice));
        }
        const size_t max_items = (size_t)c->max_subcubes * c->nchunk1;
        for (int k = 0; k < 2; ++k) {
            fgoicp_ctx::TickSlot& sl = c->slots[k];
            sl.stream = c->stream;
            CHK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
            {
                const char* e = dev_env("FGOICP_SORT_STREAM");  // tuning knob: 0 = sort on the main stream
                if (e && std::atoi(e) == 0) sl.sort_stream = c->stream;
                else {
                    const char* pe = dev_env("FGOICP_SIDE_PRIORITY");  // tuning knob: -1 = the side streams (sort, finalize / selection) above the bounds kernels' stream, 1 = below
                    const int pri = pe ? std::atoi(pe) : 0;
                    int least = 0, greatest = 0;
                    if (pri) CHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
                    if (pri) CHK(hipStreamCreateWithPriority(&sl.sort_stream, hipStreamNonBlocking, pri < 0 ? greatest : least));
                    else CHK(hipStreamCreateWithFlags(&sl.sort_stream, hipStreamNonBlocking));
                }
            }
            CHK(hipEventCreateWithFlags(&sl.sorted_ev, hipEventDisableTiming));
            CHK(hipEventCreateWithFlags(&sl.bounds_ev, hipEventDisableTiming));
            CHK(hipMalloc(&sl.d_groups, sizeof(TickGroup) * c->max_groups));
            CHK(hipMalloc(&sl.d_subs, sizeof(TickSub) * c->max_subcubes));
            CHK(hipHostMalloc((void**)&sl.h_groups, sizeof(TickGroup) * c->max_groups, hipHostMallocMapped));
            CHK(hipHostMalloc((void**)&sl.h_subs, sizeof(TickSub) * c->max_subcubes, hipHostMallocMapped));
            CHK(hipHostGetDevicePointer((void**)&sl.hd_groups, sl.h_groups, 0));
            CHK(hipHostGetDevicePointer((void**)&sl.hd_subs, sl.h_subs, 0));
            CHK(hipHostMalloc((void**)&sl.h_row_span, sizeof(float) * c->max_subcubes, hipHostMallocMapped));
            CHK(hipHostGetDevicePointer((void**)&sl.hd_row_span, sl.h_row_span, 0));
            CHK(hipHostMalloc((void**)&sl.h_sort_err, sizeof(unsigned) * 4, hipHostMallocMapped));
            CHK(hipHostGetDevicePointer((void**)&sl.hd_sort_err, sl.h_sort_err, 0));
            *sl.h_sort_err = 0u;
            CHK(hipMalloc(&sl.d_keys, sizeof(unsigned short) * max_items));
            CHK(hipMalloc(&sl.d_ranks, sizeof(unsigned) * max_items));
            CHK(hipMalloc(&sl.d_hist, sizeof(unsigned) * kTickNumKeys));
            CHK(hipMemset(sl.d_hist, 0, sizeof(unsigned) * kTickNumKeys));  // the scan kernel re-zeroes it after every tick
            CHK(hipMalloc(&sl.d_block_sums, sizeof(unsigned) * 64));
            CHK(hipMalloc(&sl.d_hist_xcd, sizeof(unsigned) * 16 * kTickNumKeys));
            CHK(hipMemset(sl.d_hist_xcd, 0, sizeof(unsigned) * 16 * kTickNumKeys));
            CHK(hipMalloc(&sl.d_xoff, sizeof(unsigned) * 16 * kTickNumKeys));
            CHK(hipMalloc(&sl.d_cursor, sizeof(unsigned) * kTickNumKeys));
            CHK(hipMalloc(&sl.d_sorted, sizeof(unsigned) * max_items));
            CHK(hipMalloc(&sl.d_partials, sizeof(double2) * max_items));
            CHK(hipMalloc(&sl.d_cut_acc, sizeof(double) * 2 * (size_t)c->max_subcubes));
            CHK(hipMemset(sl.d_cut_acc, 0, sizeof(double) * 2 * (size_t)c->max_subcubes));  // bounds_finalize_kernel re-zeroes what a window used
            CHK(hipMalloc(&sl.d_row_cut, sizeof(float) * (size_t)c->max_subcubes));
            CHK(hipMalloc(&sl.d_cut_done, sizeof(unsigned) * (size_t)c->max_subcubes));
            CHK(hipMemset(sl.d_cut_done, 0, sizeof(unsigned) * (size_t)c->max_subcubes));
            if (k == 0 && !c->d_cut_stat) {
                CHK(hipMalloc(&c->d_cut_stat, sizeof(unsigned long long) * kCutStatSlots));
                CHK(hipMemset(c->d_cut_stat, 0, sizeof(unsigned long long) * kCutStatSlots));
            }
            if (k == 0) { sl.h_lb = c->h_lb; sl.h_ub = c->h_ub; sl.hd_lb = c->hd_lb; sl.hd_ub = c->hd_ub; }
            else {
                CHK(hipHostMalloc((void**)&sl.h_lb, sizeof(float) * c->max_subcubes, hipHostMallocMapped));
                CHK(hipHostMalloc((void**)&sl.h_ub, sizeof(float) * c->max_subcubes, hipHostMallocMapped));
                CHK(hipHostGetDevicePointer((void**)&sl.hd_lb, sl.h_lb, 0));
                CHK(hipHostGetDevicePointer((void**)&sl.hd_ub, sl.h_ub, 0));
            }
        }
    }
    // exact-NN / ICP scratch
    {
        int nl = 4;  // concurrent ICP runs (ctx_icp_batch); lane 0 shares the context's main stream
        if (const char* e = dev_env("FGOICP_ICP_LANES")) nl = std::max(1, std::min(16, std::atoi(e)));  // tuning knob
        if (const char* e = dev_env("FGOICP_ICP_OVERLAP")) c->icp_overlap = std::atoi(e) != 0;       // tuning knob
        if (const char* e = dev_env("FGOICP_ICP_DEVICE")) c->icp_device = std::atoi(e) != 0;         // tuning knob / A-B: 1 = loop advanced on the device (measured slower)
        if (const char* e = dev_env("FGOICP_ICP_DUAL")) c->icp_dual_env = std::atoi(e) != 0 ? 1 : 0; // tuning knob / A-B: 1 = one walk for both scans of an iteration, 0 = two scans on two streams
        if (const char* e = dev_env("FGOICP_ICP_GATED")) c->icp_gated = std::atoi(e) != 0;           // tuning knob / A-B: 1 = iterations pre-enqueued behind stream gates
        if (const char* e = dev_env("FGOICP_ICP_FUSE")) c->icp_fuse = std::atoi(e) != 0;             // tuning knob / A-B: 0 = separate reduction kernels
        if (const char* e = dev_env("FGOICP_ICP_AHEAD")) c->icp_ahead = std::max(1, std::min(6, std::atoi(e)));  // tuning knob
        c->lanes.resize((size_t)nl);
        for (int l = 0; l < nl; ++l) {
            fgoicp_ctx::IcpLane& L = c->lanes[(size_t)l];
            if (l == 0) L.stream = c->stream;
            else CHK(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
            CHK(hipMalloc(&L.d_work, sizeof(float4) * ns));
            CHK(hipMalloc(&L.d_min_bits, sizeof(uint32_t) * ns));
            if (l == 0) CHK(hipMalloc(&L.d_thr_bits, sizeof(uint32_t) * ns));  // brute-force kernels only, which run on lane 0
            CHK(hipMalloc(&L.d_first_idx, sizeof(uint32_t) * ns));
            CHK(hipMalloc(&L.d_first_idx2, sizeof(uint32_t) * ns));
            CHK(hipMalloc(&L.d_bp, sizeof(double) * 1024 * 16));
            CHK(hipMalloc(&L.d_bp2, sizeof(double) * 1024 * 16));
            CHK(hipMalloc(&L.d_bp3, sizeof(double) * 1024 * 16));
            CHK(hipStreamCreateWithFlags(&L.icp_stream, hipStreamNonBlocking));
            CHK(hipEventCreateWithFlags(&L.icp_ev_w, hipEventDisableTiming));
            CHK(hipEventCreateWithFlags(&L.icp_ev_b, hipEventDisableTiming));
            CHK(hipMalloc(&L.d_cen, sizeof(float) * 8));
            CHK(hipHostMalloc((void**)&L.h_cen, sizeof(float) * 8, hipHostMallocMapped));
            CHK(hipHostGetDevicePointer((void**)&L.hd_cen, L.h_cen, 0));
            CHK(hipHostMalloc((void**)&L.h_sums, sizeof(double) * 16, hipHostMallocMapped));
            CHK(hipHostGetDevicePointer((void**)&L.hd_sums, L.h_sums, 0));
            CHK(hipMalloc(&L.d_wsum, sizeof(double) * 4096 * 6));
            CHK(hipHostMalloc((void**)&L.h_wsse, sizeof(double) * 4096, hipHostMallocMapped));
            CHK(hipHostGetDevicePointer((void**)&L.hd_wsse, L.h_wsse, 0));
            CHK(hipHostMalloc((void**)&L.h_covbp, sizeof(double) * 1024 * 9, hipHostMallocMapped));
            CHK(hipHostGetDevicePointer((void**)&L.hd_covbp, L.h_covbp, 0));
            CHK(hipHostMalloc((void**)&L.h_rt, sizeof(float) * 48, hipHostMallocMapped));
            CHK(hipHostGetDevicePointer((void**)&L.hd_rt, L.h_rt, 0));
            CHK(hipHostMalloc((void**)&L.h_done, sizeof(int) * 4, hipHostMallocMapped));
            CHK(hipHostGetDevicePointer((void**)&L.hd_done, L.h_done, 0));
            *L.h_done = 0;
            // signal words for hipStreamWaitValue64; without them (or without the stream operation) the gated loop is simply not used
            if (hipExtMallocWithFlags((void**)&L.sig_b, 8, hipMallocSignalMemory) != hipSuccess) L.sig_b = nullptr;
            if (hipExtMallocWithFlags((void**)&L.sig_a, 8, hipMallocSignalMemory) != hipSuccess) L.sig_a = nullptr;
            (void)hipGetLastError();
            CHK(hipMalloc(&L.d_icp, sizeof(IcpDevState)));
            CHK(hipHostMalloc((void**)&L.h_res, sizeof(IcpHostResult), hipHostMallocMapped));
            CHK(hipHostGetDevicePointer((void**)&L.hd_res, L.h_res, 0));
            for (int k = 0; k < fgoicp_ctx::IcpLane::kRing; ++k) {
                CHK(hipEventCreateWithFlags(&L.ev_step[k], hipEventDisableTiming));
                CHK(hipEventCreateWithFlags(&L.ev_sse[k], hipEventDisableTiming));
            }
        }
    }
    {   // probe the stream wait-value operation once: a wait that is already satisfied, on lane 0's side stream
        fgoicp_ctx::IcpLane& L0 = c->lanes[0];
        c->icp_gate_ok = false;
        if (L0.sig_b && L0.sig_a) {
            *(volatile uint64_t*)L0.sig_b = 1;
            if (hipStreamWaitValue64(L0.icp_stream, L0.sig_b, 1, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull) == hipSuccess && hipStreamSynchronize(L0.icp_stream) == hipSuccess)
                c->icp_gate_ok = true;
            (void)hipGetLastError();
            for (auto& L : c->lanes) { if (L.sig_b) *(volatile uint64_t*)L.sig_b = 1; if (L.sig_a) *(volatile uint64_t*)L.sig_a = 1; L.gate_seq = 1; }
        }
    }
#undef CHK
    if (c->profile) {
        c->profile = false;
        int rc = fgoicp_ctx_set_profile(c, 1);
        if (rc) { fgoicp_ctx_destroy(c); return rc; }
    }
    *out = c;
    return FGOICP_OK;
}

void fgoicp_ctx_destroy(fgoicp_ctx* c) {
    if (!c) return;
    if (c->cut_verify_windows)
        std::fprintf(stderr, "[fgoicp cut verify] %llu windows with thresholds evaluated again without them: %llu rows, %llu of them at or above their threshold, %llu violations of the contract\n",
                     (unsigned long long)c->cut_verify_windows, (unsigned long long)c->cut_verify_rows, (unsigned long long)c->cut_verify_above, (unsigned long long)c->cut_verify_bad);
    if (g_tt.on && g_tt.ticks) {
        std::fprintf(stderr, "[fgoicp timing] ticks %llu: pack %.1f us, enqueue %.1f us, wait %.1f us, copyout %.1f us per tick\n", (unsigned long long)g_tt.ticks,
                     1e6 * g_tt.pack / g_tt.ticks, 1e6 * g_tt.enqueue / g_tt.ticks, 1e6 * g_tt.wait / g_tt.ticks, 1e6 * g_tt.copyout / g_tt.ticks);
        g_tt = TickTiming{};
    }
    if (c->unit_m > 1 && dev_env("FGOICP_UNITS_STATS"))
        std::fprintf(stderr, "[fgoicp units] M = %d: %llu of %llu evaluations in sibling units (%.1f %%)\n", c->unit_m, (unsigned long long)c->unit_evals,
                     (unsigned long long)c->unit_total, c->unit_total ? 100.0 * (double)c->unit_evals / (double)c->unit_total : 0.0);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& e : c->ev_start) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_stop) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_sel_start) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_sel_stop) if (e) (void)hipEventDestroy(e);
    (void)hipFree(c->d_src); (void)hipFree(c->d_tgt); (void)hipFree(c->d_lut); (void)hipFree(c->d_lut_idx); (void)hipFree(c->d_lut_zp);
    (void)hipFree(c->d_partials);
    (void)hipFree(c->d_cut_stat);
    for (auto& L : c->lanes) {
        if (L.stream && L.stream != c->stream) { (void)hipStreamSynchronize(L.stream); (void)hipStreamDestroy(L.stream); }
        if (L.icp_stream) { (void)hipStreamSynchronize(L.icp_stream); (void)hipStreamDestroy(L.icp_stream); }
        if (L.icp_ev_w) (void)hipEventDestroy(L.icp_ev_w);
        if (L.icp_ev_b) (void)hipEventDestroy(L.icp_ev_b);
        for (int k = 0; k < fgoicp_ctx::IcpLane::kRing; ++k) {
            if (L.ev_step[k]) (void)hipEventDestroy(L.ev_step[k]);
            if (L.ev_sse[k]) (void)hipEventDestroy(L.ev_sse[k]);
        }
        (void)hipFree(L.d_icp);
        if (L.sig_b) (void)hipFree(L.sig_b);
        if (L.sig_a) (void)hipFree(L.sig_a);
        if (L.h_rt) (void)hipHostFree(L.h_rt);
        if (L.h_done) (void)hipHostFree(L.h_done);
        (void)hipFree(L.d_wsum);
        if (L.h_wsse) (void)hipHostFree(L.h_wsse);
        if (L.h_covbp) (void)hipHostFree(L.h_covbp);
        if (L.h_res) (void)hipHostFree(L.h_res);
        (void)hipFree(L.d_work); (void)hipFree(L.d_min_bits); (void)hipFree(L.d_thr_bits); (void)hipFree(L.d_first_idx); (void)hipFree(L.d_first_idx2);
        (void)hipFree(L.d_bp); (void)hipFree(L.d_bp2); (void)hipFree(L.d_bp3); (void)hipFree(L.d_cen);
        (void)hipFree(L.d_d2); (void)hipFree(L.d_nn_lb); (void)hipFree(L.d_nn_ub); (void)hipFree(L.d_nn_lb2); (void)hipFree(L.d_nn_ub2);
        (void)hipFree(L.d_sel); (void)hipFree(L.d_eq); (void)hipFree(L.d_use); (void)hipFree(L.d_sel_wide); (void)hipFree(L.d_sel_wide2);
        if (L.h_cen) (void)hipHostFree(L.h_cen);
        if (L.h_sums) (void)hipHostFree(L.h_sums);
        if (L.h_trim) (void)hipHostFree(L.h_trim);
    }
    bvh_free(&c->bvh_tgt);
    (void)hipFree(c->d_chunk_cen);
    (void)hipFree(c->d_span_cen);
    (void)hipFree(c->d_orig_of_slot);
    for (int k = 0; k < 2; ++k) {
        fgoicp_ctx::TickSlot& sl = c->slots[k];
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.sorted_ev) (void)hipEventDestroy(sl.sorted_ev);
        if (sl.bounds_ev) (void)hipEventDestroy(sl.bounds_ev);
        if (sl.sort_stream && sl.sort_stream != c->stream) { (void)hipStreamSynchronize(sl.sort_stream); (void)hipStreamDestroy(sl.sort_stream); }
        (void)hipFree(sl.d_evals);
        (void)hipFree(c->d_trim_stat); c->d_trim_stat = nullptr;
        (void)hipFree(c->d_coop); c->d_coop = nullptr; c->coop_cap = 0;
        if (sl.h_row_span) (void)hipHostFree(sl.h_row_span);
        if (sl.h_sort_err) (void)hipHostFree(sl.h_sort_err);
        (void)hipFree(sl.d_groups); (void)hipFree(sl.d_subs); (void)hipFree(sl.d_keys); (void)hipFree(sl.d_ranks); (void)hipFree(sl.d_hist);
        (void)hipFree(sl.d_cursor); (void)hipFree(sl.d_block_sums); (void)hipFree(sl.d_hist_xcd); (void)hipFree(sl.d_xoff); (void)hipFree(sl.d_sorted); (void)hipFree(sl.d_partials); (void)hipFree(sl.d_cut_acc); (void)hipFree(sl.d_row_cut); (void)hipFree(sl.d_cut_done);
        if (sl.h_groups) (void)hipHostFree(sl.h_groups);
        if (sl.h_subs) (void)hipHostFree(sl.h_subs);
        if (k == 1 && sl.h_lb) (void)hipHostFree(sl.h_lb);
        if (k == 1 && sl.h_ub) (void)hipHostFree(sl.h_ub);
    }
    if (c->h_lb) (void)hipHostFree(c->h_lb);
    if (c->h_ub) (void)hipHostFree(c->h_ub);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int fgoicp_lut_dims(const fgoicp_ctx* c, int* dims3) {
    if (!c || !dims3) return FGOICP_ERR_INVALID_ARG;
    dims3[0] = c->geom.dx; dims3[1] = c->geom.dy; dims3[2] = c->geom.dz;
    return FGOICP_OK;
}

int fgoicp_ctx_get_info(const fgoicp_ctx* c, fgoicp_ctx_info* out) {
    if (!c || !out) return FGOICP_ERR_INVALID_ARG;
    // the caller says how large ITS struct is; everything is assembled in a full-size local and only that many bytes are copied out
    const size_t caller = out->struct_size;
    if (caller < offsetof(fgoicp_ctx_info, lut_layout)) { set_error("fgoicp_ctx_get_info: set struct_size = sizeof(fgoicp_ctx_info) before the call"); return FGOICP_ERR_INVALID_ARG; }
    fgoicp_ctx_info full{};
    fgoicp_ctx_info* const user_out = out;
    out = &full;
    const LutGeom& g = c->geom;
    *out = fgoicp_ctx_info{};
    out->lut_dims[0] = g.dx; out->lut_dims[1] = g.dy; out->lut_dims[2] = g.dz;
    out->lut_layout = c->d_lut_zp ? c->lut_layout : 0;
    out->lut_nodes = (uint64_t)g.dx * g.dy * g.dz;
    const uint64_t padded = (uint64_t)g.px * g.py * g.pz;
    size_t packed = 0;  // the bricked yz-quad copy (layout 3) holds whole 4 x 4 x 4 bricks of float4
    if (c->d_lut_zp) {
        if (c->lut_layout == 4) packed = (size_t)((c->geom.px + 2) / 3) * ((c->geom.py + 1) / 2) * c->geom.pz * 8 * sizeof(float4);
        else if (c->lut_layout == 3) packed = (size_t)((c->geom.px + 3) / 4) * ((c->geom.py + 3) / 4) * ((c->geom.pz + 3) / 4) * 64 * sizeof(float4);
        else packed = padded * (c->lut_layout == 2 ? sizeof(float4) : sizeof(float2));
    }
    out->lut_bytes = padded * sizeof(float) + packed + (c->d_lut_idx ? padded * sizeof(uint32_t) : 0);  // + the index LUT of the exact scans
    out->source_points_per_face_voxel = (double)c->ns / ((double)g.dx * g.dy + (double)g.dy * g.dz + (double)g.dx * g.dz);
    out->points_per_item = c->chunk_pts;
    out->items_per_evaluation = c->nchunk1;
    out->max_subcubes_per_window = c->max_subcubes;
    out->source_order = c->source_order;
    out->tree_order = c->tree_order;
    out->chunks_per_item_with_thresholds = c->cut_span;
    full.struct_size = caller < sizeof(full) ? caller : sizeof(full);
    std::memcpy(user_out, &full, full.struct_size);
    return FGOICP_OK;
}

int fgoicp_lut_read(fgoicp_ctx* c, float* out, size_t capacity) {
    if (!c || !out) return FGOICP_ERR_INVALID_ARG;
    const size_t total = (size_t)c->geom.dx * c->geom.dy * c->geom.dz;
    if (capacity < total) { set_error("fgoicp_lut_read: buffer too small"); return FGOICP_ERR_INVALID_ARG; }
    HIPCHK(hipSetDevice(c->device));
    float* d = nullptr;
    HIPCHK(hipMalloc(&d, total * sizeof(float)));
    launch_lut_unpad(c->d_lut, c->geom, d, c->stream);
    hipError_t e = hipMemcpyAsync(out, d, total * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    HIPCHK(e);
    return FGOICP_OK;
}

int fgoicp_lut_nodes(fgoicp_ctx* c, const int* xyz, size_t n, float* out) {
    if (!c || !xyz || !out) return FGOICP_ERR_INVALID_ARG;
    if (n == 0) return FGOICP_OK;
    HIPCHK(hipSetDevice(c->device));
    int* dq = nullptr;
    float* dout = nullptr;
    HIPCHK(hipMalloc(&dq, 3 * n * sizeof(int)));
    hipError_t e = hipMalloc(&dout, n * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(dq, xyz, 3 * n * sizeof(int), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) { launch_lut_nodes(c->d_lut, c->geom, dq, n, dout, c->stream); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dq); (void)hipFree(dout);
    HIPCHK(e);
    return FGOICP_OK;
}

int fgoicp_lut_search(fgoicp_ctx* c, const float* q, size_t n, float* out) {
    if (!c || !q || !out) return FGOICP_ERR_INVALID_ARG;
    if (n == 0) return FGOICP_OK;
    HIPCHK(hipSetDevice(c->device));
    float *dq = nullptr, *dout = nullptr;
    HIPCHK(hipMalloc(&dq, 3 * n * sizeof(float)));
    hipError_t e = hipMalloc(&dout, n * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(dq, q, 3 * n * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) { launch_lut_search(c->d_lut, c->geom, dq, n, dout, c->stream); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dq); (void)hipFree(dout);
    HIPCHK(e);
    return FGOICP_OK;
}

int fgoicp_bounds_multi(fgoicp_ctx* c, int G, const float* R9, const float* rot_span, const int* fix_rot, const int* offsets,
                        const float* tn4, float* lb_out, float* ub_out) {
    if (!c || G < 0 || (G > 0 && (!R9 || !rot_span || !fix_rot || !offsets || !tn4 || !lb_out || !ub_out))) return FGOICP_ERR_INVALID_ARG;
    if (G == 0) return FGOICP_OK;
    for (int g = 0; g < G; ++g)
        if (offsets[g + 1] < offsets[g] || offsets[0] != 0) { set_error("fgoicp_bounds_multi: offsets must start at 0 and be non-decreasing"); return FGOICP_ERR_INVALID_ARG; }
    if (offsets[G] == 0) return FGOICP_OK;
    return ctx_bounds_multi(c, G, R9, rot_span, fix_rot, offsets, tn4, lb_out, ub_out);
}

int fgoicp_bounds_submit_twins(fgoicp_ctx* c, int slot, int G, const float* R9, const float* rot_span, const int* fix_rot, const int* offsets,
                               const float* tn4, const int* twin) {
    return fgoicp_bounds_submit_cut(c, slot, G, R9, rot_span, fix_rot, offsets, tn4, twin, nullptr);
}

int fgoicp_bounds_submit_cut(fgoicp_ctx* c, int slot, int G, const float* R9, const float* rot_span, const int* fix_rot, const int* offsets,
                             const float* tn4, const int* twin, const float* cut_above) {
    if (!c || slot < 0 || slot > 1 || G < 0 || (G > 0 && (!R9 || !rot_span || !fix_rot || !offsets || !tn4))) return FGOICP_ERR_INVALID_ARG;
    if (!c->sorted_bounds) { set_error("fgoicp_bounds_submit needs the sorted bounds path (FGOICP_BOUNDS_SORTED=0 is set)"); return FGOICP_ERR_INVALID_ARG; }
    static const int zero[1] = {0};
    if (G == 0) offsets = zero;
    for (int g = 0; g < G; ++g)
        if (offsets[g + 1] < offsets[g] || offsets[0] != 0) { set_error("fgoicp_bounds_submit: offsets must start at 0 and be non-decreasing"); return FGOICP_ERR_INVALID_ARG; }
    return ctx_bounds_submit(c, slot, G, R9, rot_span, fix_rot, offsets, tn4, twin, cut_above);
}

int fgoicp_ctx_cut_stats(fgoicp_ctx* c, uint64_t* items_offered, uint64_t* items_cut, int reset) {
    if (!c) return FGOICP_ERR_INVALID_ARG;
    return ctx_cut_stats(c, items_offered, items_cut, reset);
}

int fgoicp_bounds_submit(fgoicp_ctx* c, int slot, int G, const float* R9, const float* rot_span, const int* fix_rot, const int* offsets,
                         const float* tn4) {
    return fgoicp_bounds_submit_twins(c, slot, G, R9, rot_span, fix_rot, offsets, tn4, nullptr);
}

int fgoicp_bounds_collect(fgoicp_ctx* c, int slot, float* lb_out, float* ub_out) {
    if (!c || slot < 0 || slot > 1 || !lb_out || !ub_out) return FGOICP_ERR_INVALID_ARG;
    return ctx_bounds_collect(c, slot, lb_out, ub_out);
}

int fgoicp_bounds_batch(fgoicp_ctx* c, const float* R9, float rot_span, const float* tn4, int B, int fix_rot, float* lb_out, float* ub_out) {
    if (B < 0) return FGOICP_ERR_INVALID_ARG;
    const int offsets[2] = {0, B};
    return fgoicp_bounds_multi(c, 1, R9, &rot_span, &fix_rot, offsets, tn4, lb_out, ub_out);
}

int fgoicp_bounds_point_distances(fgoicp_ctx* c, const float* R9, float rot_span, const float* tnode4, int fix_rot, float* e_out) {
    if (!c || !R9 || !tnode4 || !e_out) return FGOICP_ERR_INVALID_ARG;
    if (!c->inliers || !c->sorted_bounds) { set_error("fgoicp_bounds_point_distances: trimming is off (fgoicp_ctx_set_inliers)"); return FGOICP_ERR_INVALID_ARG; }
    const int offsets[2] = {0, 1};
    float lb = 0.f, ub = 0.f;
    int rc = ctx_bounds_submit(c, 0, 1, R9, &rot_span, &fix_rot, offsets, tnode4, nullptr);
    if (rc) return rc;
    rc = ctx_bounds_collect(c, 0, &lb, &ub);
    if (rc) return rc;
    std::vector<float> row(c->ns);
    HIPCHK(hipMemcpy(row.data(), c->slots[0].d_evals, sizeof(float) * c->ns, hipMemcpyDeviceToHost));  // row 0 of the window just collected
    for (size_t i = 0; i < c->ns; ++i) e_out[c->perm[i]] = row[i];
    return FGOICP_OK;
}

// TEST HOOK (tests/test_gpu_fullsize.py): the n-th sorted tick from now on gets a slot of `sorted` spoiled, so that the on-device
// permutation check has something to find.  Nothing in the product calls it.
int fgoicp_ctx_test_sort_fault(fgoicp_ctx* c, int nth_tick) {
    if (!c || nth_tick < 0) return FGOICP_ERR_INVALID_ARG;
    c->sort_fault_tick = nth_tick ? (int)c->sorted_ticks + nth_tick : 0;
    return FGOICP_OK;
}

int fgoicp_ctx_sort_fallbacks(const fgoicp_ctx* c, uint64_t* sorted_ticks, uint64_t* fallbacks) {
    if (!c) return FGOICP_ERR_INVALID_ARG;
    if (sorted_ticks) *sorted_ticks = c->sorted_ticks;
    if (fallbacks) *fallbacks = c->sort_fallbacks;
    return FGOICP_OK;
}

int fgoicp_sse(fgoicp_ctx* c, const float* R9, const float* t3, float* sse_out) {
    if (!c || !R9 || !t3 || !sse_out) return FGOICP_ERR_INVALID_ARG;
    return ctx_sse(c, R9, t3, sse_out, nullptr);
}

int fgoicp_icp(fgoicp_ctx* c, const float* R0, const float* t0, size_t max_iter, float thr, float* sse_out, float* R_out9, float* t_out3,
               int* iters_out) {
    if (!c || !R0 || !t0 || !sse_out || !R_out9 || !t_out3) return FGOICP_ERR_INVALID_ARG;
    return ctx_icp(c, R0, t0, max_iter, thr, sse_out, R_out9, t_out3, iters_out);
}

int fgoicp_icp_batch(fgoicp_ctx* c, int n, const float* R0s_9, const float* t0s_3, size_t max_iter, float thr, float* sse_out, float* R_out9s, float* t_out3s,
                     int* iters_out) {
    if (!c || n < 0 || (n > 0 && (!R0s_9 || !t0s_3 || !sse_out || !R_out9s || !t_out3s))) return FGOICP_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(c->device));
    return ctx_icp_batch(c, n, R0s_9, t0s_3, max_iter, thr, sse_out, R_out9s, t_out3s, iters_out);
}

int fgoicp_procrustes(fgoicp_ctx* c, const float* working_xyz, float* R_out9, float* t_out3, float* centroids6, float* ABt9, int* corr_idx) {
    if (!c || !working_xyz || !R_out9 || !t_out3) return FGOICP_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(c->device));
    std::vector<float4> h(c->ns);
    for (size_t i = 0; i < c->ns; ++i) {
        const float* p = working_xyz + 3 * (size_t)c->perm[i];
        h[i] = make_float4(p[0], p[1], p[2], 0.f);
    }
    HIPCHK(hipMemcpy(c->lanes[0].d_work, h.data(), sizeof(float4) * c->ns, hipMemcpyHostToDevice));
    Mat3f R, ABt;
    Vec3f t;
    int rc = ctx_procrustes_device(c, c->lanes[0], &R, &t, centroids6, &ABt, false);
    if (rc) return rc;
    std::memcpy(R_out9, R.m, sizeof(R.m));
    t_out3[0] = t.x; t_out3[1] = t.y; t_out3[2] = t.z;
    if (ABt9) std::memcpy(ABt9, ABt.m, sizeof(ABt.m));
    if (corr_idx) {
        std::vector<uint32_t> idx(c->ns);
        HIPCHK(hipMemcpy(idx.data(), c->lanes[0].d_first_idx, sizeof(uint32_t) * c->ns, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < c->ns; ++i) corr_idx[c->perm[i]] = (int)idx[i];
    }
    return FGOICP_OK;
}

int fgoicp_ctx_set_inliers(fgoicp_ctx* c, size_t k) {
    if (!c) return FGOICP_ERR_INVALID_ARG;
    return ctx_set_inliers(c, k);
}

int fgoicp_ctx_profile(fgoicp_ctx* c, double* kernel_ms, uint64_t* launches, uint64_t* subcubes, int reset) {
    if (!c) return FGOICP_ERR_INVALID_ARG;
    if (c->ev_used) {
        HIPCHK(hipStreamSynchronize(c->stream));
        int rc = ctx_flush_profile(c);
        if (rc) return rc;
    }
    if (kernel_ms) *kernel_ms = c->prof_ms;
    if (launches) *launches = c->prof_launches;
    if (subcubes) *subcubes = c->prof_subcubes;
    if (reset) { c->prof_ms = 0; c->prof_launches = 0; c->prof_subcubes = 0; c->prof_evals = 0; c->prof_sel_ms_last = c->prof_sel_ms; c->prof_sel_ms = 0; }
    return FGOICP_OK;
}

int fgoicp_ctx_profile_select_ms(fgoicp_ctx* c, double* select_ms) {
    if (!c || !select_ms) return FGOICP_ERR_INVALID_ARG;
    if (c->ev_used) {
        HIPCHK(hipStreamSynchronize(c->stream));
        int rc = ctx_flush_profile(c);
        if (rc) return rc;
    }
    *select_ms = c->prof_sel_ms;
    return FGOICP_OK;
}

int fgoicp_ctx_trim_stats(fgoicp_ctx* c, uint64_t* out3, int reset) {
    if (!c || !out3) return FGOICP_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(c->device));
    if (c->d_trim_stat) {
        unsigned long long h[4] = {0, 0, 0, 0};
        for (auto& sl : c->slots) if (sl.inflight) { set_error("fgoicp_ctx_trim_stats: a bounds submission is in flight"); return FGOICP_ERR_INVALID_ARG; }
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(h, c->d_trim_stat, sizeof(h), hipMemcpyDeviceToHost));
        HIPCHK(hipMemset(c->d_trim_stat, 0, sizeof(h)));
        for (int i = 0; i < 3; ++i) c->trim_stat_acc[i] += h[i];
    }
    for (int i = 0; i < 3; ++i) out3[i] = c->trim_stat_acc[i];
    if (reset) for (auto& v : c->trim_stat_acc) v = 0;
    return FGOICP_OK;
}

int fgoicp_ctx_profile_evaluations(fgoicp_ctx* c, uint64_t* evaluations) {
    if (!c || !evaluations) return FGOICP_ERR_INVALID_ARG;
    *evaluations = c->prof_evals;
    return FGOICP_OK;
}

int fgoicp_ctx_set_coop_split(fgoicp_ctx* c, size_t min_points_untrimmed, size_t min_points_trimmed) {
    if (!c) return FGOICP_ERR_INVALID_ARG;
    c->coop_split_min = min_points_untrimmed;
    c->coop_split_trim_min = min_points_trimmed;
    return FGOICP_OK;
}

int fgoicp_ctx_set_profile(fgoicp_ctx* c, int enabled) {
    if (!c) return FGOICP_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(c->device));
    if (enabled && c->ev_start.empty()) {
        c->ev_start.assign(1024, nullptr);
        c->ev_stop.assign(1024, nullptr);
        c->ev_sel_start.assign(1024, nullptr);
        c->ev_sel_stop.assign(1024, nullptr);
        c->ev_has_sel.assign(1024, 0);
        for (size_t i = 0; i < c->ev_start.size(); ++i) {
            HIPCHK(hipEventCreate(&c->ev_start[i]));
            HIPCHK(hipEventCreate(&c->ev_stop[i]));
            HIPCHK(hipEventCreate(&c->ev_sel_start[i]));
            HIPCHK(hipEventCreate(&c->ev_sel_stop[i]));
        }
    }
    if (!enabled && c->ev_used) {
        HIPCHK(hipStreamSynchronize(c->stream));
        int rc = ctx_flush_profile(c);
        if (rc) return rc;
    }
    c->profile = enabled != 0;
    return FGOICP_OK;
}

size_t fgoicp_ctx_ns(const fgoicp_ctx* c) { return c ? c->ns : 0; }
size_t fgoicp_ctx_nt(const fgoicp_ctx* c) { return c ? c->nt : 0; }

}  // extern "C"
