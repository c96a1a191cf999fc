// The bounds kernel of the sorted whole-tick path, round 4 (included by kernels.hip inside namespace fgoicp::{anonymous}; uses its helpers).
//
// kernComputeBounds + the two reductions (fgoicp/registration.cu:27-60, :126-140): a work item is (evaluation s, chunk c of 256 ... 2048
// consecutive source points) = ONE wave, four points per lane and pass.  Same per-point fp32 values and the same fp64 sums, in the same
// order, as round 3's bounds_sorted_kernel<64, 4, ...> (kept in the development build as the A/B and bit reference) — rewritten against
// the instruction stream (120 -> 77 VALU instructions per point-evaluation; the launches did not get shorter for it: the texture
// addresser / L1 path binds both the sparse and the dense leg):
//   * everything wave-uniform is read once (rotation, sin, kind of the item: round 3 re-read `sin_half` and `fix_rot` through the scalar
//     cache for every point — the compiler could not prove that the stores of the kernel do not alias the descriptors);
//   * the three kinds of item (fix_rot = 1, fix_rot = 0, dual) and "every point of the pass exists" are decided per pass, outside the
//     per-point code: no select on `valid` (four v_cndmask per point), no branch per point;
//   * the clamp of CUDA's clamp addressing is one v_med3_f32 per axis, the "+ 1" of the padded layout is folded into one constant, the
//     texel offset is 32-bit arithmetic (v_mad_i32_i24) on a scalar base unless the packed LUT is larger than 4 GiB (WIDE);
//   * pair-wise packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: IEEE per element, so no bit changes) where two elements of ONE
//     point sit in an aligned register pair by construction: the x/y components of rotate -> translate -> LUT coordinates -> weights, and
//     the two z-columns of the trilinear blend (a 16-byte gather lands in v[n:n+3]: {x0z0, x0z1 | x1z0, x1z1}).  Measured on this chip
//     (tools/calib/valu_rate.hip, profiles/r04_valu_rate.txt): a packed instruction issues in the time of TWO plain ones (0.49-0.55 against
//     0.85-0.94 instructions per SIMD and ns), so packing buys 15-19 % of the packed instructions' issue time, not half of it —
//     the instruction COUNT falls further than the time does.
// What did shorten the runs is not evaluating what the search drops anyway: the early exit further down (fgoicp_bounds_submit_cut).
#pragma once

constexpr float kCutNone = 3.0e38f;       // thresholds at or above this (fgoicp_bounds_submit_cut: +inf) switch the early exit off
constexpr double kCutMargin = 1.000001;   // see bounds_item_kernel

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f2v pk_fma(f2v a, f2v b, f2v c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2v splat2(float v) { return f2v{v, v}; }

// what a wave needs of the LUT geometry, derived once from the kernel arguments (scalar registers)
struct ItemGeom {
    f2v off_xy, dmax_xy;     // -min_bound (x, y); (float)(d - 1) (x, y): upper clamp of floor(u - 0.5)
    float off_z, dmax_z, scale;
    float pyf;               // (float)py: the row number (z + 1) * py + (y + 1) is exact in fp32 (< 2^24)
    int px;
    int c0;                  // (py + 1) * px + 1: what the "+ 1" of the three padded indices adds to the offset
    unsigned nbx3, nby2;     // apron layout
};
__device__ __forceinline__ ItemGeom item_geom(const LutGeom& g) {
    ItemGeom G;
    G.off_xy = f2v{g.off_x, g.off_y};
    G.dmax_xy = f2v{(float)(g.dx - 1), (float)(g.dy - 1)};
    G.off_z = g.off_z;
    G.dmax_z = (float)(g.dz - 1);
    G.scale = g.scale;
    G.pyf = (float)g.py;
    G.px = g.px;
    G.c0 = (g.py + 1) * g.px + 1;
    G.nbx3 = (unsigned)(g.px + 2) / 3u;
    G.nby2 = (unsigned)(g.py + 1) >> 1;
    return G;
}

// NearestNeighborLUT::search up to the texel fetch (registration.cu:320-328; tex_axis / lut_address above, same arithmetic): the clamped
// floors of the three axes (integers held in fp32, -1 ... d - 1) and the three weights.
struct ItemTex {
    f2v fl_xy, w_xy;
    float fl_z, w_z;
};
template <bool QUANT>
__device__ __forceinline__ ItemTex item_tex(const ItemGeom& G, const float (&R)[9], f2v t_xy, float t_z, const float4& p) {
    // glm::mat3 * vec3 in the device contraction order (rotate() above): fma(R[6 + r], z, fma(R[3 + r], y, R[r] * x))
    const f2v r_xy = pk_fma(f2v{R[6], R[7]}, splat2(p.z), pk_fma(f2v{R[3], R[4]}, splat2(p.y), f2v{R[0], R[1]} * splat2(p.x)));
    const float r_z = fma_(R[8], p.z, fma_(R[5], p.y, R[2] * p.x));
    // (R p + t + offset) * scale - 0.5, registration.cu:34, :323-325 and the texel-centre shift of linear filtering
    const f2v ub_xy = ((r_xy + t_xy) + G.off_xy) * splat2(G.scale) - splat2(0.5f);
    const float ub_z = ((r_z + t_z) + G.off_z) * G.scale - 0.5f;
    f2v fl_xy = f2v{floorf(ub_xy.x), floorf(ub_xy.y)};
    float fl_z = floorf(ub_z);
    ItemTex t;
    t.w_xy = ub_xy - fl_xy;
    t.w_z = ub_z - fl_z;
    if (QUANT) {  // 1.8 fixed point, round to nearest: floor(w * 256 + 0.5) / 256 (w * 256 is exact, so the fused form has the same bits)
        const f2v q = pk_fma(t.w_xy, splat2(256.0f), splat2(0.5f));
        t.w_xy = f2v{floorf(q.x), floorf(q.y)} * splat2(1.0f / 256.0f);
        t.w_z = floorf(fma_(t.w_z, 256.0f, 0.5f)) * (1.0f / 256.0f);
    }
    // clamp addressing: fmin(fmax(fl, -1), d - 1) — as one median (fl is never NaN for finite input; a NaN ends in index 0 either way)
    t.fl_xy = f2v{__builtin_amdgcn_fmed3f(fl_xy.x, -1.0f, G.dmax_xy.x), __builtin_amdgcn_fmed3f(fl_xy.y, -1.0f, G.dmax_xy.y)};
    t.fl_z = __builtin_amdgcn_fmed3f(fl_z, -1.0f, G.dmax_z);
    return t;
}
// element index of the texel (x0, y0, z0) in the padded LUT = ((iz * py) + iy) * px + ix with i = (int)fl + 1.
// NARROW: the launch guarantees py * pz <= 2^23 (the row number fits the signed 24-bit multiply) and a packed copy below 4 GiB (the
// byte offset fits 32 bits: one scalar base + one 32-bit lane offset per gather); otherwise 64-bit arithmetic.
template <bool WIDE>
__device__ __forceinline__ size_t item_index(const ItemGeom& G, const ItemTex& t) {
    const float rowf = fma_(t.fl_z, G.pyf, t.fl_xy.y);  // exact: |value| < 2^24
    const int row = (int)rowf, ix = (int)t.fl_xy.x;
    if (WIDE) return (size_t)((long long)row * (long long)G.px + (long long)(ix + G.c0));
    return (size_t)(unsigned)(__mul24(row, G.px) + ix + G.c0);
}
template <bool WIDE, int SHIFT>
__device__ __forceinline__ const char* item_address(const char* __restrict__ base, size_t index) {
    if (WIDE) return base + (index << SHIFT);
    return base + (size_t)((unsigned)index << SHIFT);  // < 4 GiB by the launch's choice of WIDE
}
__device__ __forceinline__ unsigned item_packed_index(const ItemTex& t) {  // x | y << 10 | z << 20 of the padded indices (apron layout; dims <= 1023)
    return (unsigned)((int)t.fl_xy.x + 1) | ((unsigned)((int)t.fl_xy.y + 1) << 10) | ((unsigned)((int)t.fl_z + 1) << 20);
}

// trilinear blend of the 2 x 2 x 2 footprint (lut_blend above: lerps along x, then y, then z, each fma(w, q - p, p)).
// lo / hi: the footprint's x0 and x1 faces, each as {(y0, z0), (y0, z1)}, {(y1, z0), (y1, z1)}.
__device__ __forceinline__ float item_blend(const ItemTex& t, f2v lo_y0, f2v hi_y0, f2v lo_y1, f2v hi_y1) {
    const f2v wa = splat2(t.w_xy.x), wb = splat2(t.w_xy.y);
    const f2v c_y0 = pk_fma(wa, hi_y0 - lo_y0, lo_y0);   // {c00, c01}
    const f2v c_y1 = pk_fma(wa, hi_y1 - lo_y1, lo_y1);   // {c10, c11}
    const f2v c_z = pk_fma(wb, c_y1 - c_y0, c_y0);       // {lerp(c00, c10, b), lerp(c01, c11, b)}
    return fma_(t.w_z, c_z.y - c_z.x, c_z.x);
}

// MODE: 0 = fix_rot item, 1 = rotation-uncertainty item, 2 = dual (both variants from one lookup: acc[0..1] fix_rot = 1, acc[2..3] fix_rot = 0)
// LAYOUT: 1 = z-pair copy (float2 {T[o], T[o + slice]}), 3 = yz-quad runs, 5 = apron-bricked yz-quads (both fetched by lane pairs)
template <int LAYOUT, int TRIM, bool WIDE, bool QUANT, int MODE, bool FULL>
__device__ __forceinline__ void item_pass(const float4* __restrict__ src, int ns, const char* __restrict__ lutp, const ItemGeom& G, const float (&R)[9], f2v t_xy, float t_z,
                                          float sin_half, float trans_radius, int first, int odd, double (&acc)[4], float* __restrict__ row0, float* __restrict__ row1,
                                          int samp_shift) {
    constexpr int P = 4;
    float4 p[P];
    ItemTex tx[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const int i = first + 64 * k;
        p[k] = src[FULL ? i : (i < ns ? i : ns - 1)];
        tx[k] = item_tex<QUANT>(G, R, t_xy, t_z, p[k]);
    }
    // all gathers of the pass in flight before the first use
    f4v ga[P], gb[P];
    if (LAYOUT == 1) {  // z-pair: rows y0 and y0 + 1, each {T[x0, z0], T[x0, z1], T[x1, z0], T[x1, z1]}
        const char* row1 = lutp + (size_t)G.px * sizeof(float2);  // a second scalar base: both rows of a lookup share one 32-bit lane offset
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const size_t idx = item_index<WIDE>(G, tx[k]);
            ga[k] = *reinterpret_cast<const float4u*>(item_address<WIDE, 3>(lutp, idx));
            gb[k] = *reinterpret_cast<const float4u*>(item_address<WIDE, 3>(row1, idx));
        }
    } else {            // yz-quads (at most 2^30 of them), the two 16-byte halves of a lookup fetched by a pair of neighbouring lanes (quad_pair_issue above)
        QuadPairLoads qp[P];
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int own = LAYOUT == 5 ? (int)apron_index(item_packed_index(tx[k]), G.nbx3, G.nby2) : (int)item_index<WIDE>(G, tx[k]);
            const int other = swap_lane_pair(own);
            const int o_even = odd ? other : own, o_odd = odd ? own : other;
            qp[k].r1 = *reinterpret_cast<const float4a*>(item_address<WIDE, 4>(lutp, (size_t)(unsigned)(o_even + odd)));
            qp[k].r2 = *reinterpret_cast<const float4a*>(item_address<WIDE, 4>(lutp, (size_t)(unsigned)(o_odd + odd)));
        }
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const float4a send = odd ? qp[k].r1 : qp[k].r2;
            float4a recv;
            recv.x = swap_lane_pair(send.x); recv.y = swap_lane_pair(send.y); recv.z = swap_lane_pair(send.z); recv.w = swap_lane_pair(send.w);
            const float4a a = odd ? recv : qp[k].r1, b = odd ? qp[k].r2 : recv;  // quad of x0, quad of x1: {y0z0, y0z1, y1z0, y1z1}
            ga[k] = f4v{a.x, a.y, a.z, a.w};
            gb[k] = f4v{b.x, b.y, b.z, b.w};
        }
    }
    float te0[TRIM ? P : 1], te1[TRIM ? P : 1];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        float dsq;
        if (LAYOUT == 1) dsq = item_blend(tx[k], f2v{ga[k].x, ga[k].y}, f2v{ga[k].z, ga[k].w}, f2v{gb[k].x, gb[k].y}, f2v{gb[k].z, gb[k].w});   // :46
        else dsq = item_blend(tx[k], f2v{ga[k].x, ga[k].y}, f2v{gb[k].x, gb[k].y}, f2v{ga[k].z, ga[k].w}, f2v{gb[k].z, gb[k].w});
        const float d1 = sqrtf(dsq);                                              // :48
        const float d0 = d1 - 2.0f * p[k].w * sin_half;                           // :39-43, :49-52 (fix_rot = 0)
        const int i = first + 64 * k;
        if (TRIM) {
            // trimmed Go-ICP: both bounds are non-decreasing in e = max(d, 0) (ub = e * e, lb = max(e - r_t, 0)^2: the values of :54-58),
            // so one row of e per variant carries both selections (trim_rows_sampled_kernel)
            const float e0 = fmaxf(MODE == 1 ? d0 : d1, 0.0f), e1 = fmaxf(d0, 0.0f);
            if (FULL || i < ns) {
                row0[i] = e0;
                if (MODE == 2) row1[i] = e1;
            }
            te0[k] = e0;
            te1[k] = e1;
            continue;
        }
        if (MODE != 1) {  // fix_rot = 1: :54, :57-58
            const float m = fmaxf(d1, 0.0f), l = fmaxf(d1 - trans_radius, 0.0f);  // d > 0 ? d * d : 0 == max(d, 0)^2 (bit for bit, -0 and NaN included)
            const float ubv = m * m, lbv = l * l;
            acc[0] += (FULL || i < ns) ? (double)ubv : 0.0;
            acc[1] += (FULL || i < ns) ? (double)lbv : 0.0;
        }
        if (MODE != 0) {  // fix_rot = 0
            const float m = fmaxf(d0, 0.0f), l = fmaxf(d0 - trans_radius, 0.0f);
            const float ubv = m * m, lbv = l * l;
            acc[MODE == 2 ? 2 : 0] += (FULL || i < ns) ? (double)ubv : 0.0;
            acc[MODE == 2 ? 3 : 1] += (FULL || i < ns) ? (double)lbv : 0.0;
        }
    }
    if (TRIM && samp_shift > 0) {  // the row's sample (trim_store / trim_is_sample above): one point of every run of 2^samp_shift once more behind the row
        const size_t off = trim_sample_offset(ns);
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int i = first + 64 * k;
            if ((FULL || i < ns) && trim_is_sample(i, samp_shift, ns)) {
                row0[off + (size_t)(i >> samp_shift)] = te0[k];
                if (MODE == 2) row1[off + (size_t)(i >> samp_shift)] = te1[k];
            }
        }
    }
}

template <int LAYOUT, int TRIM, bool WIDE, bool QUANT, int MODE>
__device__ __forceinline__ void item_walk(const float4* __restrict__ src, int ns, const char* __restrict__ lutp, const ItemGeom& G, const float (&R)[9], f2v t_xy, float t_z,
                                          float sin_half, float trans_radius, int base, int chunk_pts, int lane, double (&acc)[4], float* __restrict__ row0,
                                          float* __restrict__ row1, int samp_shift) {
    const int odd = lane & 1;
    for (int pass = 0; pass < chunk_pts; pass += 256) {
        const int first = base + pass + lane;
        if (base + pass + 256 <= ns)  // wave-uniform: every point of the pass exists (all passes but the cloud's last)
            item_pass<LAYOUT, TRIM, WIDE, QUANT, MODE, true>(src, ns, lutp, G, R, t_xy, t_z, sin_half, trans_radius, first, odd, acc, row0, row1, samp_shift);
        else
            item_pass<LAYOUT, TRIM, WIDE, QUANT, MODE, false>(src, ns, lutp, G, R, t_xy, t_z, sin_half, trans_radius, first, odd, acc, row0, row1, samp_shift);
    }
}

template <int LAYOUT, int TRIM, bool WIDE, bool QUANT, bool SPAN /* several chunks per item: its own instantiation, the loop costs the one-chunk kernel 1-2 % */>
__global__ __launch_bounds__(64) void bounds_item_kernel(const float4* __restrict__ src, int ns, const char* __restrict__ lutp, LutGeom g,
                                                         const TickGroup* __restrict__ groups, const TickSub* __restrict__ subs, const unsigned* __restrict__ sorted,
                                                         int nchunk, int chunk_pts, double2* __restrict__ partials, float* __restrict__ evals, size_t erow, int samp_shift,
                                                         unsigned nitems, unsigned* __restrict__ sort_err, TickCut cut, int span /* chunks per work item (1: the rule) */) {
    // two tiers (windows with thresholds): workgroups are handed out in the order of their ids, so the first n0 of them take the first tier —
    // each tier divided among the XCDs in contiguous runs like the whole list otherwise (a bijection either way: both parts are)
    unsigned slot;
    if (!TRIM && cut.tier_split && sorted) {
        const unsigned n0 = min(*cut.tier_split, gridDim.x);
        slot = blockIdx.x < n0 ? xcd_remap(blockIdx.x, n0) : n0 + xcd_remap(blockIdx.x - n0, gridDim.x - n0);
    } else {
        slot = xcd_remap(blockIdx.x, gridDim.x);
    }
    const unsigned item = sorted ? sorted[slot] : slot;  // small ticks come unsorted
    // The sort's check, folded into its only consumer.  `sorted` was filled with 0xFFFFFFFF before the scatter and every in-range rank
    // is written by exactly the item that drew it, so if no slot still holds an out-of-range value every slot was written, hence written
    // once: `sorted` is a permutation of the items.  The grid reads every slot exactly once (xcd_remap is a bijection), so a slot that
    // was never written (two items drew the same rank: the XCD-private histogram's workgroup-scope atomics did not behave as one point
    // of coherence) is seen here; *sort_err is pinned host memory and the context then repeats the window with device-scope atomics.
    if (item >= nitems) {
        if (sort_err && threadIdx.x == 0) *sort_err = 1u;
        return;
    }
    // A work item is `span` consecutive chunks of one evaluation (one chunk, except in windows with thresholds on sparse clouds: an item
    // that ends early costs a workgroup dispatch — 0.21 ns each on this chip, profiles/r04_dispatch_rate.txt — and little else, so
    // such windows take two chunks per wave).  The sums stay per chunk: same partials, same bits, whatever the span.
    const unsigned per_eval = SPAN ? ((unsigned)nchunk + (unsigned)span - 1u) / (unsigned)span : (unsigned)nchunk;
    const int s = (int)(item / per_eval);
    const int chunk0 = (int)(item - (unsigned)s * per_eval) * (SPAN ? span : 1);
    const int nsub = SPAN ? min(span, nchunk - chunk0) : 1;
    // (read next to the descriptor, not behind it: an item that ends early is three dependent L2 round trips — slot, descriptor + hint, partial)
    const unsigned done_hint = (!TRIM && cut.acc && !(cut.probe & 4)) ? __builtin_nontemporal_load(&cut.done[s]) : 0u;
    const TickSub sb = subs[s];
    const int lane = (int)threadIdx.x;
    // Early exit (fgoicp_bounds_submit_cut).  Every term of the lower-bound sum is >= 0, so once the partial sums of the evaluation's
    // FINISHED items have reached the caller's threshold the whole sum has, and the caller has said that it only needs to know that
    // much (the inner BnB drops such a node whatever its exact bounds are: fgoicp.cpp:151): the remaining items leave a partial that
    // keeps the row at or above the threshold and end.  Which items get here early depends on timing; what is reported does not —
    // bounds_finalize_kernel returns {T, T} for EVERY row whose lower bound is >= T, cut short or not.  The margin keeps the decision
    // on the safe side of the two summation orders (this running sum: finished items in any order; the reported one: the fixed tree).
    bool cutting = false;
    if (!TRIM && cut.acc) {
        if (chunk0 == 0 && lane == 0) {
            cut.row_cut[sb.out0] = sb.cut0;
            if (sb.dual) cut.row_cut[sb.out1] = sb.cut1;
        }
        cutting = sb.cut0 < kCutNone && (!sb.dual || sb.cut1 < kCutNone);
        if (cutting) {
            // The running sums live at the device's point of coherence (the eight XCDs' L2s are not coherent with each other): reading
            // them costs a trip to memory, ~2 us.  Once an item has found its evaluation finished it says so in `done`, an ordinary
            // cached word: the later items of that evaluation ON THE SAME XCD (whose L2 holds that store) end after an L2 hit instead.
            // A stale 0 — another XCD's L2, a line not refreshed yet — only sends the item down the slow path.
            bool reached = done_hint != 0u;
            if (!__builtin_amdgcn_readfirstlane((int)reached) && !(cut.probe & 1)) {
                const double a0 = __hip_atomic_load(&cut.acc[2 * (size_t)s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const double a1 = sb.dual ? __hip_atomic_load(&cut.acc[2 * (size_t)s + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
                reached = a0 >= (double)sb.cut0 * kCutMargin && (!sb.dual || a1 >= (double)sb.cut1 * kCutMargin);
                if (reached && lane == 0) cut.done[s] = 1u;
            }
            if (__builtin_amdgcn_readfirstlane((int)reached)) {
                if (lane < nsub) {  // (the negative upper-bound partial marks a chunk as not evaluated: bounds_finalize_kernel counts them)
                    partials[(size_t)sb.out0 * nchunk + chunk0 + lane] = make_double2(-1.0, (double)sb.cut0);
                    if (sb.dual) partials[(size_t)sb.out1 * nchunk + chunk0 + lane] = make_double2(0.0, (double)sb.cut1);
                }
                return;
            }
        }
    }
    const TickGroup gr = groups[sb.group];  // by value: the rotation node of the item, read once
    float R[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = gr.R[k];
    const ItemGeom G = item_geom(g);
    const float trans_radius = kSqrt3 * sb.span;  // registration.cu:33
    const f2v t_xy = f2v{sb.tx, sb.ty};
    float* row0 = TRIM ? evals + (size_t)sb.out0 * erow : nullptr;
    float* row1 = TRIM ? evals + (size_t)sb.out1 * erow : nullptr;
    double lb_fix = 0.0, lb_rot = 0.0;  // what this item adds to the evaluation's running sums
    for (int sub = 0; sub < (SPAN ? nsub : 1); ++sub) {
        const int chunk = chunk0 + sub;
        const int base = chunk * chunk_pts;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        if (sb.dual) item_walk<LAYOUT, TRIM, WIDE, QUANT, 2>(src, ns, lutp, G, R, t_xy, sb.tz, gr.sin_half, trans_radius, base, chunk_pts, lane, acc, row0, row1, samp_shift);
        else if (gr.fix_rot) item_walk<LAYOUT, TRIM, WIDE, QUANT, 0>(src, ns, lutp, G, R, t_xy, sb.tz, gr.sin_half, trans_radius, base, chunk_pts, lane, acc, row0, row1, samp_shift);
        else item_walk<LAYOUT, TRIM, WIDE, QUANT, 1>(src, ns, lutp, G, R, t_xy, sb.tz, gr.sin_half, trans_radius, base, chunk_pts, lane, acc, row0, row1, samp_shift);
        if (TRIM) continue;
        // the wave tree of block_sum with one wave (bounds_sorted_kernel<64, ...>: same operands, same order), lane 0 writes
        const double r0 = wave_sum(acc[0]), r1 = wave_sum(acc[1]);
        lb_fix += r1;
        if (sb.dual) {
            const double r2 = wave_sum(acc[2]), r3 = wave_sum(acc[3]);
            lb_rot += r3;
            if (lane == 0) {
                partials[(size_t)sb.out0 * nchunk + chunk] = make_double2(r0, r1);
                partials[(size_t)sb.out1 * nchunk + chunk] = make_double2(r2, r3);
            }
        } else if (lane == 0) {
            partials[(size_t)sb.out0 * nchunk + chunk] = make_double2(r0, r1);
        }
    }
    if (!TRIM && cutting && lane == 0 && !(cut.probe & 2)) {
        unsafeAtomicAdd(&cut.acc[2 * (size_t)s], lb_fix);
        if (sb.dual) unsafeAtomicAdd(&cut.acc[2 * (size_t)s + 1], lb_rot);
    }
}
