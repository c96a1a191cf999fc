// gfx950 (CDNA4, wave64) kernels of the Go-ICP hot path.  Written for MI355X only.
//
// Arithmetic contract (checked bit-for-bit against oracle/ in tests/): the file is compiled with
// -ffp-contract=off and every fused multiply-add is spelled out, in the order the reference's
// device code contracts to under nvcc's default -fmad=true:
//     a*x + b*y + c*z  ->  fma(c, z, fma(b, y, a*x)).
// Per-point values are fp32; every cross-point sum is accumulated in fp64 in a fixed order
// (wave shuffle tree -> waves in order -> blocks in order), so results are reproducible run to
// run and lie within one fp32 rounding of the exact sum the reference's Thrust reductions
// approximate in an unspecified order (fgoicp/registration.cu:126-140).
#include "kernels.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "../host/math3.hpp"
#include "../host/knobs.hpp"
#include "slab.hpp"
#include "bvh.hpp"

namespace fgoicp {
namespace {

typedef float float2u __attribute__((ext_vector_type(2), aligned(4)));  // dword-aligned 8-byte load

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

struct Rt {  // rigid motion by value in the kernarg segment
    float R[9];
    float t[3];
};

// glm::mat3 * vec3 (column-major), device contraction order
__device__ __forceinline__ void rotate(const float* R, float x, float y, float z, float& ox, float& oy, float& oz) {
    ox = fma_(R[6], z, fma_(R[3], y, R[0] * x));
    oy = fma_(R[7], z, fma_(R[4], y, R[1] * x));
    oz = fma_(R[8], z, fma_(R[5], y, R[2] * x));
}

// distance_squared, fgoicp/registration.cu:154-160 / :250-256
__device__ __forceinline__ float dist_sq(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    return fma_(dz, dz, fma_(dy, dy, dx * dx));
}

// Sum over the wave, valid in lane 0.  The tree is v[i] += v[i + off] for off = 32, 16, 8, 4, 2, 1; the two far steps go
// through the LDS crossbar (ds_bpermute), the four near ones stay inside a row of 16 lanes and use DPP row_shl on the two
// halves of the double — same operands, same order, no LDS round trip (lanes whose partner lies outside their row add 0
// instead of themselves; they do not feed lane 0).
template <int OFF>
__device__ __forceinline__ double dpp_row_shl(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), 0x100 + OFF, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x100 + OFF, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += __shfl_down(v, 32, 64);
    v += __shfl_down(v, 16, 64);
    v += dpp_row_shl<8>(v);
    v += dpp_row_shl<4>(v);
    v += dpp_row_shl<2>(v);
    v += dpp_row_shl<1>(v);
    return v;
}

// Sum K doubles per thread over the 256-thread block; the result is valid in threads 0..K-1
// (thread k holds component k).  Fixed order: shuffle tree inside a wave, then waves 0..3.
template <int K, int NW = 4>
__device__ __forceinline__ double block_sum(const double (&v)[K], double* lds /* [NW*K] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double r = wave_sum(v[k]);
        if (lane == 0) lds[wave * K + k] = r;
    }
    __syncthreads();
    double out = 0.0;
    if (threadIdx.x < K) {
        out = lds[threadIdx.x];
#pragma unroll
        for (int w = 1; w < NW; ++w) out += lds[w * K + threadIdx.x];
    }
    __syncthreads();
    return out;
}

// ---------------------------------------------------------------------------------------------
// NearestNeighborLUT::search — fgoicp/registration.cu:320-328 with tex3D<float> linear filtering
// restated in software (gfx950 has no image unit).  CUDA semantics (unnormalised coordinates,
// cudaFilterModeLinear, cudaAddressModeClamp): xB = x - 0.5, i = floor(xB), alpha = frac(xB) held
// in 9-bit fixed point with 8 fractional bits, texels i and i+1 clamped to [0, D-1].
// On the padded LUT texel k lives at k+1, so clamping floor(xB) to [-1, D-1] is enough.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void tex_axis(float u, int d, int quant, int& i, float& w) {
    float ub = u - 0.5f;
    float fl = floorf(ub);
    w = ub - fl;
    if (quant) w = floorf(w * 256.0f + 0.5f) * (1.0f / 256.0f);
    fl = fminf(fmaxf(fl, -1.0f), (float)(d - 1));
    i = (int)fl + 1;
}

__device__ __forceinline__ float lerp(float p, float q, float w) { return fma_(w, q - p, p); }

// ---------------------------------------------------------------------------------------------
// kernComputeBounds (+ the two thrust::reduce calls) — fgoicp/registration.cu:27-60, :126-140.
// grid = chunks of 256*P points x translation nodes of the batch (XCD-aware order); one block
// owns one (chunk, subcube) pair and emits one {sum_ub, sum_lb} partial.  The source cloud is float4
// {x, y, z, x*x+y*y+z*z}: one coalesced 16-byte load per point (TODO.md:14 of the reference).
// ---------------------------------------------------------------------------------------------
// lut_search split in two so that a thread can put the gathers of several points in flight before
// it consumes any of them (memory-level parallelism is what bounds this kernel, not arithmetic).
struct TexAddr {
    size_t o;         // element offset of the first texel (x0, y0, z0) in the padded LUT
    float a, b, c;    // interpolation weights
    unsigned pk;      // the same texel as packed padded indices x | y << 10 | z << 20 (bricked layout; dims <= 1023)
};
__device__ __forceinline__ TexAddr lut_address(const LutGeom& g, float qx, float qy, float qz) {
    const float x = (qx + g.off_x) * g.scale;
    const float y = (qy + g.off_y) * g.scale;
    const float z = (qz + g.off_z) * g.scale;
    int ix, iy, iz;
    TexAddr t;
    tex_axis(x, g.dx, g.quantize, ix, t.a);
    tex_axis(y, g.dy, g.quantize, iy, t.b);
    tex_axis(z, g.dz, g.quantize, iz, t.c);
    // indices are >= 0 and < 4096: the row number fits 24 + bits, one 32 x 32 -> 64 multiply-add finishes the offset (signed size_t
    // arithmetic cost seven instructions here, a point is ~120)
    t.o = (size_t)((unsigned long long)((unsigned)iz * (unsigned)g.py + (unsigned)iy) * (unsigned)g.px + (unsigned)ix);
    t.pk = (unsigned)ix | ((unsigned)iy << 10) | ((unsigned)iz << 20);
    return t;
}
__device__ __forceinline__ float lut_blend(const TexAddr& t, float2u v00, float2u v10, float2u v01, float2u v11) {
    const float c00 = lerp(v00.x, v00.y, t.a);
    const float c10 = lerp(v10.x, v10.y, t.a);
    const float c01 = lerp(v01.x, v01.y, t.a);
    const float c11 = lerp(v11.x, v11.y, t.a);
    return lerp(lerp(c00, c10, t.b), lerp(c01, c11, t.b), t.c);
}

__device__ __forceinline__ float lut_search(const float* __restrict__ lut, const LutGeom& g, float qx, float qy, float qz) {
    const TexAddr t = lut_address(g, qx, qy, qz);
    const size_t sy = (size_t)g.px, sz = (size_t)g.px * g.py;
    const float* p = lut + t.o;
    return lut_blend(t, *(const float2u*)(p), *(const float2u*)(p + sy), *(const float2u*)(p + sz), *(const float2u*)(p + sz + sy));
}

// Z-paired copy of the LUT: zp[o] = {T[o], T[o + one z-slice]} as float2, same (padded, x-fastest) indexing.
// The 2x2x2 footprint of a lookup is then TWO 16-byte gathers — {T[x0,y,z0], T[x0,y,z1], T[x1,y,z0], T[x1,y,z1]}
// for y = y0 and y0+1 — instead of four 8-byte ones: the bounds kernel is bound by per-lane address
// processing of divergent gathers, not by bytes (measured 1.6x on the sparse 40k cloud).  Same texels,
// same blend order -> bit-identical values.
typedef float float4u __attribute__((ext_vector_type(4), aligned(8)));
__device__ __forceinline__ void zpair_gather(const float2* __restrict__ zp, const LutGeom& g, const TexAddr& t, float2u& v00, float2u& v10,
                                             float2u& v01, float2u& v11) {
    const float4u a = *(const float4u*)(zp + t.o);
    const float4u b = *(const float4u*)(zp + t.o + (size_t)g.px);
    v00 = float2u{a.x, a.z};  // (x0, x1) at (y0, z0)
    v01 = float2u{a.y, a.w};  // (x0, x1) at (y0, z1)
    v10 = float2u{b.x, b.z};  // (x0, x1) at (y1, z0)
    v11 = float2u{b.y, b.w};  // (x0, x1) at (y1, z1)
}

// yz-quad copy: q[o] = {T[o], T[o + z-slice], T[o + row], T[o + row + z-slice]} — a lookup is then 32 CONTIGUOUS bytes
// (q[o], q[o+1]: x0 and x1), i.e. one cache line (two when x0 % 8 == 7) instead of two rows.  4x the bytes of the LUT.
typedef float float4a __attribute__((ext_vector_type(4), aligned(16)));
__device__ __forceinline__ void quad_gather(const float4* __restrict__ qd, const TexAddr& t, float2u& v00, float2u& v10, float2u& v01, float2u& v11) {
    const float4a a = *(const float4a*)(qd + t.o);
    const float4a b = *(const float4a*)(qd + t.o + 1);
    v00 = float2u{a.x, b.x};  // (x0, x1) at (y0, z0)
    v01 = float2u{a.y, b.y};  // (y0, z1)
    v10 = float2u{a.z, b.z};  // (y1, z0)
    v11 = float2u{a.w, b.w};  // (y1, z1)
}
// The same lookup with the two 16-byte halves fetched by a PAIR of neighbouring lanes in ONE instruction: load 1 brings
// q[o], q[o+1] of the even lane's point (even lane: q[o], odd lane: q[o+1]), load 2 those of the odd lane's point, and the
// lanes swap what the other one needs (DPP quad_perm, no LDS).  The texture addresser merges the two lanes' accesses to the
// same line, so a lookup costs the L1 one tag access instead of a miss plus a hit on the still-pending line (PMC: the L1
// spends half its cycles in pending stalls with the per-lane form).
__device__ __forceinline__ int swap_lane_pair(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true); }
__device__ __forceinline__ float swap_lane_pair(float v) { return __int_as_float(swap_lane_pair(__float_as_int(v))); }
struct QuadPairLoads { float4a r1, r2; };
__device__ __forceinline__ QuadPairLoads quad_pair_issue(const float4* __restrict__ qd, const TexAddr& t, int odd) {
    const int own = (int)t.o, other = swap_lane_pair(own);
    const int o_even = odd ? other : own, o_odd = odd ? own : other;
    QuadPairLoads q;
    q.r1 = *(const float4a*)(qd + (size_t)(o_even + odd));
    q.r2 = *(const float4a*)(qd + (size_t)(o_odd + odd));
    return q;
}
__device__ __forceinline__ void quad_pair_finish(const QuadPairLoads& q, int odd, float2u& v00, float2u& v10, float2u& v01, float2u& v11) {
    float4a send = odd ? q.r1 : q.r2, recv;
    recv.x = swap_lane_pair(send.x); recv.y = swap_lane_pair(send.y); recv.z = swap_lane_pair(send.z); recv.w = swap_lane_pair(send.w);
    const float4a a = odd ? recv : q.r1, b = odd ? q.r2 : recv;
    v00 = float2u{a.x, b.x};
    v01 = float2u{a.y, b.y};
    v10 = float2u{a.z, b.z};
    v11 = float2u{a.w, b.w};
}
// Bricked yz-quad copy (experimental, FGOICP_LUT_ZPAIR=3): the quads of a 4 x 4 x 4 block of nodes are contiguous (1 KiB) and
// Morton-ordered inside it, so a 128-byte line holds a 2 x 2 x 2 block of quads instead of a run of 8 along x: a surface patch of
// any orientation then uses ~4 of the 8 quads of a line it touches, an x-run ~3 on average (1 + ln 8).
__device__ __forceinline__ unsigned spread2(unsigned v) { return (v & 1u) | ((v & 2u) << 2); }
__device__ __forceinline__ size_t brick_index(unsigned x, unsigned y, unsigned z, unsigned nbx, unsigned nby) {
    const size_t b = ((size_t)(z >> 2) * nby + (y >> 2)) * nbx + (x >> 2);
    return b * 64 + (spread2(x & 3u) | (spread2(y & 3u) << 1) | (spread2(z & 3u) << 2));
}
__device__ __forceinline__ QuadPairLoads quad_pair_issue_bricked(const float4* __restrict__ qd, const TexAddr& t, int odd, unsigned nbx, unsigned nby) {
    const int own = (int)t.pk, other = swap_lane_pair(own);
    const unsigned p_even = (unsigned)(odd ? other : own), p_odd = (unsigned)(odd ? own : other);
    QuadPairLoads q;
    q.r1 = *(const float4a*)(qd + brick_index((p_even & 1023u) + odd, (p_even >> 10) & 1023u, p_even >> 20, nbx, nby));
    q.r2 = *(const float4a*)(qd + brick_index((p_odd & 1023u) + odd, (p_odd >> 10) & 1023u, p_odd >> 20, nbx, nby));
    return q;
}
__global__ __launch_bounds__(kBlock) void lut_quad_bricked_kernel(const float* __restrict__ lut, LutGeom g, float4* __restrict__ qd) {
    const unsigned nbx = (unsigned)(g.px + 3) >> 2, nby = (unsigned)(g.py + 3) >> 2;
    const size_t total = (size_t)g.px * g.py * g.pz, sy = (size_t)g.px, sz = (size_t)g.px * g.py;
    for (size_t n = (size_t)blockIdx.x * kBlock + threadIdx.x; n < total; n += (size_t)gridDim.x * kBlock) {
        const unsigned x = (unsigned)(n % g.px), y = (unsigned)((n / g.px) % g.py), z = (unsigned)(n / sz);
        const size_t nz = n + sz < total ? n + sz : n, ny = n + sy < total ? n + sy : n, nyz = n + sy + sz < total ? n + sy + sz : n;
        qd[brick_index(x, y, z, nbx, nby)] = make_float4(lut[n], lut[nz], lut[ny], lut[nyz]);
    }
}

// Apron-bricked yz-quad copy (round 3, FGOICP_LUT_ZPAIR=4): a 128-byte line holds the quads of 4 consecutive x at 2 consecutive y
// (one z), and consecutive lines OVERLAP by one x: line (xb, yb, z) = quads x in [3 xb, 3 xb + 3], y in {2 yb, 2 yb + 1}.  A lookup
// with base voxel (x0, y0, z0) needs the quads at x0 and x0 + 1 of row y0: slots s, s + 1 of line (x0 / 3, y0 / 2, z0) with
// s = (y0 & 1) * 4 + x0 % 3 — ALWAYS one line (the x-run layout straddles two lines when x0 % 8 == 7, the 2 x 2 x 2 brick for every odd
// x0), and a line serves a 3 x 2 patch of base voxels instead of a run of 8: a surface of any orientation crosses
// (|nx| + |ny| + |nz|) / (|nx| / 3 + |ny| / 2 + |nz|) = 1.6 base voxels of a line it touches against 1.4 / 1.125 lines per lookup for the
// run.  21.3 B per node (the run: 16).  Same texels, same 32 contiguous bytes per lookup, same blend: bit-identical values.
__device__ __forceinline__ unsigned apron_index(unsigned pk /* x | y << 10 | z << 20 */, unsigned nbx3, unsigned nby2) {
    const unsigned x = pk & 1023u, y = (pk >> 10) & 1023u, z = pk >> 20;
    const unsigned q = (x * 43691u) >> 17;  // x / 3 (x < 1024)
    return ((z * nby2 + (y >> 1)) * nbx3 + q) * 8u + ((y & 1u) << 2) + (x - 3u * q);
}
__device__ __forceinline__ QuadPairLoads quad_pair_issue_apron(const float4* __restrict__ qd, const TexAddr& t, int odd, unsigned nbx3, unsigned nby2) {
    const int own = (int)apron_index(t.pk, nbx3, nby2), other = swap_lane_pair(own);
    const int o_even = odd ? other : own, o_odd = odd ? own : other;
    QuadPairLoads q;
    q.r1 = *(const float4a*)(qd + (size_t)(o_even + odd));
    q.r2 = *(const float4a*)(qd + (size_t)(o_odd + odd));
    return q;
}
__global__ __launch_bounds__(kBlock) void lut_quad_apron_kernel(const float* __restrict__ lut, LutGeom g, float4* __restrict__ qd) {
    const unsigned nbx3 = (unsigned)(g.px + 2) / 3u, nby2 = (unsigned)(g.py + 1) >> 1;
    const size_t slots = (size_t)nbx3 * nby2 * g.pz * 8, total = (size_t)g.px * g.py * g.pz, sy = (size_t)g.px, sz = (size_t)g.px * g.py;
    for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < slots; e += (size_t)gridDim.x * kBlock) {
        const unsigned s = (unsigned)(e & 7u);
        const size_t line = e >> 3;
        const unsigned xb = (unsigned)(line % nbx3), yb = (unsigned)((line / nbx3) % nby2), z = (unsigned)(line / ((size_t)nbx3 * nby2));
        const unsigned x = 3u * xb + (s & 3u), y = 2u * yb + (s >> 2);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x < (unsigned)g.px && y < (unsigned)g.py) {
            const size_t n = ((size_t)z * g.py + y) * g.px + x;
            const size_t nz = n + sz < total ? n + sz : n, ny = n + sy < total ? n + sy : n, nyz = n + sy + sz < total ? n + sy + sz : n;
            v = make_float4(lut[n], lut[nz], lut[ny], lut[nyz]);
        }
        qd[e] = v;
    }
}

__global__ __launch_bounds__(kBlock) void lut_quad_kernel(const float* __restrict__ lut, LutGeom g, float4* __restrict__ qd) {
    const size_t total = (size_t)g.px * g.py * g.pz, sy = (size_t)g.px, sz = (size_t)g.px * g.py;
    for (size_t n = (size_t)blockIdx.x * kBlock + threadIdx.x; n < total; n += (size_t)gridDim.x * kBlock) {
        const size_t nz = n + sz < total ? n + sz : n, ny = n + sy < total ? n + sy : n, nyz = n + sy + sz < total ? n + sy + sz : n;
        qd[n] = make_float4(lut[n], lut[nz], lut[ny], lut[nyz]);  // out-of-range neighbours belong to texels no lookup uses as (y0, z0)
    }
}

__global__ __launch_bounds__(kBlock) void lut_zpair_kernel(const float* __restrict__ lut, LutGeom g, float2* __restrict__ zp) {
    const size_t total = (size_t)g.px * g.py * g.pz, sz = (size_t)g.px * g.py;
    for (size_t n = (size_t)blockIdx.x * kBlock + threadIdx.x; n < total; n += (size_t)gridDim.x * kBlock)
        zp[n] = make_float2(lut[n], n + sz < total ? lut[n + sz] : lut[n]);  // the last padded slice is never a z0
}

// XCD-aware block order.  Workgroups are dealt round-robin over the 8 XCDs (each with a private
// 4 MiB L2), so linear ids congruent mod 8 share an L2.  Map them to CONSECUTIVE virtual ids: an
// XCD then walks whole subcubes chunk by chunk, and Morton-adjacent chunks — which share the LUT
// lines along their common border — hit in the same L2.  Bijective for any grid size.
__device__ __forceinline__ unsigned xcd_remap(unsigned id, unsigned total) {
    const unsigned q = total >> 3, r = total & 7u, xcd = id & 7u, k = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

template <int P>
__global__ __launch_bounds__(kBlock) void bounds_kernel(const float4* __restrict__ src, int ns, const float* __restrict__ lut,
                                                        LutGeom g, BoundsArgs a, double2* __restrict__ partials, int nchunk) {
    __shared__ double red[8];
    const unsigned v = xcd_remap(blockIdx.x, gridDim.x);
    const int b = (int)(v / (unsigned)nchunk);
    const int chunk = (int)(v - (unsigned)b * (unsigned)nchunk);
    const float4 tn = a.tn[b];
    const float trans_uncertain_radius = kSqrt3 * tn.w;  // :33
    const int base = chunk * (kBlock * P) + threadIdx.x;
    const size_t sy = (size_t)g.px, sz = (size_t)g.px * g.py;

    // phase 1: the points (coalesced 16-byte loads; out-of-range lanes re-read the last point)
    float4 p[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const int i = base + k * kBlock;
        p[k] = src[i < ns ? i : ns - 1];
    }
    // phase 2: all 4*P gathers in flight
    TexAddr ta[P];
    float2u v00[P], v10[P], v01[P], v11[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        float rx, ry, rz;
        rotate(a.R, p[k].x, p[k].y, p[k].z, rx, ry, rz);
        ta[k] = lut_address(g, rx + tn.x, ry + tn.y, rz + tn.z);  // :34, :323-325
    }
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const float* q = lut + ta[k].o;
        v00[k] = *(const float2u*)(q);
        v10[k] = *(const float2u*)(q + sy);
        v01[k] = *(const float2u*)(q + sz);
        v11[k] = *(const float2u*)(q + sz + sy);
    }
    // phase 3: blend, bounds, fp64 accumulation
    double acc[2] = {0.0, 0.0};
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const float dsq = lut_blend(ta[k], v00[k], v10[k], v01[k], v11[k]);  // :46
        float d = sqrtf(dsq);                                                 // :48
        if (!a.fix_rot) d -= 2.0f * p[k].w * a.sin_half;                      // :39-43, :49-52
        const float ubv = d > 0.0f ? d * d : 0.0f;                            // :54
        const float l = d - trans_uncertain_radius;                           // :57
        const float lbv = l > 0.0f ? l * l : 0.0f;                            // :58
        const bool valid = base + k * kBlock < ns;
        acc[0] += valid ? (double)ubv : 0.0;
        acc[1] += valid ? (double)lbv : 0.0;
    }
    const double r = block_sum<2>(acc, red);
    // threads 0 and 1 hold sum_ub and sum_lb
    double* out = reinterpret_cast<double*>(partials + ((size_t)(a.out_base + b) * nchunk + chunk));
    if (threadIdx.x < 2) out[threadIdx.x] = r;
}

// ---------------------------------------------------------------------------------------------
// Locality-sorted variant for a whole tick (many rotation nodes, hundreds of subcubes).
// A work item is (subcube s, chunk c of 256 Morton-consecutive points): a compact surface patch that
// reads a compact region of the LUT.  One tick's items together want every LUT line ~25 times, but
// in submission order two users of a line run far apart in time, the 203 MB LUT streams through the
// 4 MiB L2s over and over, and the kernel runs at the fabric rate instead of the L2 rate.  So the
// items are ordered by the Hilbert index of the LUT cell their patch centre lands in (counting sort on
// the device, 15-bit keys; Hilbert rather than Z-order: +5 % kernel throughput, no long jumps between
// consecutive cells), and the XCD-aware remap hands each XCD one contiguous run of that order:
// blocks resident together on an XCD read the same neighbourhood of the LUT.  Every item writes the
// same partial it would write in any order, and the finalize sum is ordered by (s, c) — results
// are bit-identical to the plain kernel.
// ---------------------------------------------------------------------------------------------
constexpr int kKeyBits = 15;
constexpr int kNumKeys = 1 << kKeyBits;

__device__ __forceinline__ unsigned part1by2_5(unsigned v) {  // spread 5 bits to every third position
    v &= 31u;
    v = (v | (v << 8)) & 0x100fu;
    v = (v | (v << 4)) & 0x10c3u;
    v = (v | (v << 2)) & 0x1249u;
    return v;
}

// 15-bit Hilbert index of a cell (5 bits per axis, Skilling's transpose algorithm): consecutive keys are always face
// neighbours, without the long jumps a Z-order curve makes at every power-of-two boundary.
__device__ __forceinline__ unsigned hilbert15(unsigned x, unsigned y, unsigned z) {
    unsigned X[3] = {x & 31u, y & 31u, z & 31u};
#pragma unroll
    for (unsigned Q = 16u; Q > 1u; Q >>= 1) {
        const unsigned P = Q - 1u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) X[0] ^= P;
            else { const unsigned t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    unsigned t = 0;
#pragma unroll
    for (unsigned Q = 16u; Q > 1u; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1u;
    X[0] ^= t; X[1] ^= t; X[2] ^= t;
    return part1by2_5(X[2]) | (part1by2_5(X[1]) << 1) | (part1by2_5(X[0]) << 2);
}

// XCD = 1: the histogram is private to the XCD the wave runs on (hist points at kTickXcds histograms; HW_REG_XCC_ID picks
// one) and the atomic has workgroup scope, i.e. it executes in that XCD's L2.  A device-scope atomic leaves the L2 for the
// memory side, and next to the bounds kernel every such atomic per item cost that kernel ~4 % (measured by adding dummy
// ones).  An address is only ever touched from one XCD, so the L2 is a sufficient point of coherence; the dirty lines reach
// memory at the end of the kernel like any other store.  The rank then carries the XCD in its top bits.
constexpr int kTickXcds = 16;  // the XCC_ID field is 4 bits wide
template <int HILBERT, int XCD>
__global__ __launch_bounds__(64) void tick_keys_kernel(const float4* __restrict__ chunk_cen, int nchunk, const TickGroup* __restrict__ groups,
                                                           const TickSub* __restrict__ subs, int nsub, LutGeom g, int cell_shift,
                                                           unsigned short* __restrict__ keys, unsigned* __restrict__ ranks, unsigned* __restrict__ hist,
                                                           unsigned* __restrict__ prefill /* optional: `sorted`, filled with 0xFFFFFFFF for the permutation check of the bounds kernel */,
                                                           int nunits, int unit_m /* sibling units: the first nunits * unit_m evaluations, unit_m per item */,
                                                           int orient /* experimental: 1 = 12-bit cell index + 3 bits of the rotated patch normal (chunk_cen[nchunk + c]) */,
                                                           const float* __restrict__ tier_lut /* windows with thresholds: the plain LUT; nullptr = one tier */, float tier_level) {
    const size_t unit_items = (size_t)nunits * nchunk;
    const size_t nitems = unit_items + (size_t)(nsub - nunits * unit_m) * nchunk;
    for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < nitems; i += (size_t)gridDim.x * 64) {
        if (prefill) prefill[i] = 0xFFFFFFFFu;
        int s, c;
        TickSub sb;
        if (i < unit_items) {  // a sibling unit is keyed by the mean of its translation nodes (the parent's centre for a whole octet)
            const int u = (int)(i / nchunk);
            c = (int)(i - (size_t)u * nchunk);
            s = u * unit_m;
            sb = subs[s];
            for (int j = 1; j < unit_m; ++j) { const TickSub o = subs[s + j]; sb.tx += o.tx; sb.ty += o.ty; sb.tz += o.tz; }
            const float inv = 1.0f / (float)unit_m;
            sb.tx *= inv; sb.ty *= inv; sb.tz *= inv;
        } else {
            const size_t r = i - unit_items;
            s = nunits * unit_m + (int)(r / nchunk);
            c = (int)(r - (size_t)(s - nunits * unit_m) * nchunk);
            sb = subs[s];
        }
        const TickGroup& gr = groups[sb.group];
        const float4 cc = chunk_cen[c];
        float rx, ry, rz;
        rotate(gr.R, cc.x, cc.y, cc.z, rx, ry, rz);
        const int vx = (int)fminf(fmaxf((rx + sb.tx + g.off_x) * g.scale, 0.0f), (float)(g.dx - 1)) >> cell_shift;
        const int vy = (int)fminf(fmaxf((ry + sb.ty + g.off_y) * g.scale, 0.0f), (float)(g.dy - 1)) >> cell_shift;
        const int vz = (int)fminf(fmaxf((rz + sb.tz + g.off_z) * g.scale, 0.0f), (float)(g.dz - 1)) >> cell_shift;
        unsigned key = HILBERT ? hilbert15((unsigned)vx, (unsigned)vy, (unsigned)vz)
                               : part1by2_5((unsigned)vx) | (part1by2_5((unsigned)vy) << 1) | (part1by2_5((unsigned)vz) << 2);
        if (orient) {
            // two patches around the same LUT cell share lines only if they lie in (nearly) the same plane: cells twice as wide, and
            // inside a cell the items grouped by the direction of the rotated patch normal (hemisphere: 4 quadrants x {pole cap, rim})
            const float4 nn = chunk_cen[nchunk + c];
            float nx, ny, nz;
            rotate(gr.R, nn.x, nn.y, nn.z, nx, ny, nz);
            if (nz < 0.0f) { nx = -nx; ny = -ny; }
            const unsigned oc = (nx >= 0.0f ? 1u : 0u) | (ny >= 0.0f ? 2u : 0u) | (nz * nz > 0.5f ? 4u : 0u);
            key = (hilbert15((unsigned)vx >> 1, (unsigned)vy >> 1, (unsigned)vz >> 1) << 3) | oc;
        }
        if (tier_lut) {
            // Windows with thresholds (fgoicp_bounds_submit_cut): an evaluation is over as soon as the lower-bound sums of its finished items
            // reach its threshold T, so the items likely to carry much of the sum go FIRST (tier 0: the top key bit clear; inside a tier the
            // Hilbert order at half the cell resolution).  The guess: the bound's per-point term at the patch centre, from the nearest LUT
            // node, against the level a point contributes on average when the sum is T — tier_level * T / ns.  Only the order depends
            // on it; a wrong guess costs time, never a bit.
            const int ix = (int)fminf(fmaxf((rx + sb.tx + g.off_x) * g.scale, 0.0f), (float)(g.dx - 1));
            const int iy = (int)fminf(fmaxf((ry + sb.ty + g.off_y) * g.scale, 0.0f), (float)(g.dy - 1));
            const int iz = (int)fminf(fmaxf((rz + sb.tz + g.off_z) * g.scale, 0.0f), (float)(g.dz - 1));
            const float d = sqrtf(tier_lut[((size_t)(iz + 1) * g.py + (size_t)(iy + 1)) * g.px + (size_t)(ix + 1)]);
            const float e_fix = fmaxf(d - kSqrt3 * sb.span, 0.0f);
            const float e_rot = fmaxf(d - kSqrt3 * sb.span - 2.0f * gr.sin_half * sqrtf(cc.x * cc.x + cc.y * cc.y + cc.z * cc.z), 0.0f);
            bool heavy;
            if (sb.dual) heavy = e_fix * e_fix >= tier_level * sb.cut0 && e_rot * e_rot >= tier_level * sb.cut1;
            else heavy = (gr.fix_rot ? e_fix * e_fix : e_rot * e_rot) >= tier_level * sb.cut0;
            key = (key >> 1) | (heavy ? 0u : 1u << 14);
        }
        keys[i] = (unsigned short)key;
        if (XCD) {
            const unsigned x = __builtin_amdgcn_s_getreg(20 /* HW_REG_XCC_ID */ | (0 << 6) | ((4 - 1) << 11));
            ranks[i] = __hip_atomic_fetch_add(&hist[(size_t)x * kNumKeys + key], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) | (x << 28);
        } else {
            ranks[i] = atomicAdd(&hist[key], 1u);  // the item's place inside its bin: the scatter pass needs no atomics of its own
        }
    }
}

// The sort kernels run next to the other slot's bounds kernel, which keeps every wave slot of every CU taken by one-wave
// workgroups: a multi-wave workgroup has to wait until several slots of ONE CU are free at the same time (measured: a
// 16-wave scan block 380 us, a 4-wave block 250 us, against ~10 us on an idle device), a one-wave workgroup takes the
// next free slot.  So all three sort kernels use 64-thread workgroups.
//
// exclusive scan of the 32768-bin histogram in two one-round-trip kernels of 64 one-wave blocks (512 bins per block,
// 8 per lane): block sums, then every block scans the 64 block sums and its own bins.  Next to a saturated memory system a
// dependent load costs several microseconds, so the chain is kept to two loads deep (a single block walking all bins took
// 250-600 us there).  The second kernel leaves the histogram zeroed for the next tick (it must be zero before the first).
constexpr int kScanBlocks = kNumKeys / 512;

__global__ __launch_bounds__(64) void tick_scan_sums_kernel(const unsigned* __restrict__ hist, unsigned* __restrict__ block_sums) {
    const int lane = threadIdx.x;
    const uint4* h4 = reinterpret_cast<const uint4*>(hist) + (size_t)blockIdx.x * 128 + 2 * lane;
    const uint4 a = h4[0], b = h4[1];
    unsigned sum = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) block_sums[blockIdx.x] = sum;
}

__global__ __launch_bounds__(64) void tick_scan_apply_kernel(unsigned* __restrict__ hist, const unsigned* __restrict__ block_sums,
                                                             unsigned* __restrict__ cursor) {
    static_assert(kScanBlocks == 64, "one lane per block sum");
    const int lane = threadIdx.x;
    uint4* h4 = reinterpret_cast<uint4*>(hist) + (size_t)blockIdx.x * 128 + 2 * lane;
    const uint4 a = h4[0], b = h4[1];
    const unsigned bs = lane < (int)blockIdx.x ? block_sums[lane] : 0u;
    h4[0] = make_uint4(0u, 0u, 0u, 0u);
    h4[1] = make_uint4(0u, 0u, 0u, 0u);
    unsigned base = bs;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) base += __shfl_xor(base, off, 64);
    const unsigned sum = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
    unsigned incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned u = __shfl_up(incl, off, 64);
        if (lane >= off) incl += u;
    }
    unsigned run = base + incl - sum;
    uint4 o0, o1;
    o0.x = run; run += a.x;
    o0.y = run; run += a.y;
    o0.z = run; run += a.z;
    o0.w = run; run += a.w;
    o1.x = run; run += b.x;
    o1.y = run; run += b.y;
    o1.z = run; run += b.z;
    o1.w = run;
    uint4* c4 = reinterpret_cast<uint4*>(cursor) + (size_t)blockIdx.x * 128 + 2 * lane;
    c4[0] = o0;
    c4[1] = o1;
}

// The per-XCD histograms folded AND the first scan level, one kernel of kScanBlocks one-wave blocks (512 keys per block, 8 per lane):
// hist[k] = sum over XCDs (input of tick_scan_apply_kernel), xoff[x][k] = items of key k on XCDs before x, block_sums[b] = items of the
// block's keys; leaves the per-XCD histograms zeroed for the next tick.
__global__ __launch_bounds__(64) void tick_fold_sums_kernel(unsigned* __restrict__ hist_xcd, unsigned* __restrict__ xoff, unsigned* __restrict__ hist,
                                                            unsigned* __restrict__ block_sums) {
    const int lane = threadIdx.x;
    const size_t k0 = (size_t)blockIdx.x * 512 + 8 * lane;  // this lane's 8 consecutive keys (two 16-byte accesses per XCD histogram)
    uint4 ra = make_uint4(0u, 0u, 0u, 0u), rb = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll 4
    for (int x = 0; x < kTickXcds; ++x) {
        uint4* h4 = reinterpret_cast<uint4*>(hist_xcd + (size_t)x * kNumKeys + k0);
        uint4* o4 = reinterpret_cast<uint4*>(xoff + (size_t)x * kNumKeys + k0);
        const uint4 a = h4[0], b = h4[1];
        o4[0] = ra; o4[1] = rb;
        ra.x += a.x; ra.y += a.y; ra.z += a.z; ra.w += a.w;
        rb.x += b.x; rb.y += b.y; rb.z += b.z; rb.w += b.w;
        h4[0] = make_uint4(0u, 0u, 0u, 0u);
        h4[1] = make_uint4(0u, 0u, 0u, 0u);
    }
    uint4* d4 = reinterpret_cast<uint4*>(hist + k0);
    d4[0] = ra; d4[1] = rb;
    unsigned sum = ra.x + ra.y + ra.z + ra.w + rb.x + rb.y + rb.z + rb.w;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) block_sums[blockIdx.x] = sum;
}
__global__ __launch_bounds__(64) void tick_scatter_xcd_kernel(const unsigned short* __restrict__ keys, const unsigned* __restrict__ ranks, size_t nitems,
                                                              const unsigned* __restrict__ cursor, const unsigned* __restrict__ xoff, unsigned* __restrict__ sorted) {
    for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < nitems; i += (size_t)gridDim.x * 64) {
        const unsigned k = keys[i], r = ranks[i];
        sorted[cursor[k] + xoff[(size_t)(r >> 28) * kNumKeys + k] + (r & 0x0FFFFFFFu)] = (unsigned)i;
    }
}
#ifdef FGOICP_DEV_KNOBS
// The permutation check as a launch of its own, for round 3's bounds kernels (bounds_item_kernel carries it itself: see there).
__global__ __launch_bounds__(64) void tick_check_kernel(const unsigned* __restrict__ sorted, size_t nitems, unsigned* __restrict__ err) {
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < nitems; i += (size_t)gridDim.x * 64) bad = bad || sorted[i] >= nitems;
    if (__any(bad) && threadIdx.x == 0) *err = 1u;
}
#endif
// The tick's descriptors from the pinned staging buffers into device memory: one launch instead of two hipMemcpyAsync calls (each
// costs the submitting thread ~10 us; the bytes — <= 150 KB — cross PCIe either way).
__global__ __launch_bounds__(64) void tick_upload_kernel(const uint4* __restrict__ hg, uint4* __restrict__ dg, unsigned ng16, const uint4* __restrict__ hs,
                                                         uint4* __restrict__ ds, unsigned ns16) {
    for (unsigned i = blockIdx.x * 64 + threadIdx.x; i < ng16 + ns16; i += gridDim.x * 64) {
        if (i < ng16) dg[i] = hg[i];
        else ds[i - ng16] = hs[i - ng16];
    }
}
__global__ void tick_fault_kernel(unsigned* sorted) { sorted[0] = 0xFFFFFFFFu; }  // test hook (FGOICP_SORT_FAULT_TICK): a slot no item was scattered to
// A/B only (FGOICP_SORT_RANKS=0): the classic scatter with its own atomic per item
__global__ __launch_bounds__(64) void tick_scatter_atomic_kernel(const unsigned short* __restrict__ keys, size_t nitems, unsigned* __restrict__ cursor,
                                                                 unsigned* __restrict__ sorted) {
    for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < nitems; i += (size_t)gridDim.x * 64)
        sorted[atomicAdd(&cursor[keys[i]], 1u)] = (unsigned)i;
}
__global__ __launch_bounds__(64) void tick_scatter_kernel(const unsigned short* __restrict__ keys, const unsigned* __restrict__ ranks, size_t nitems,
                                                          const unsigned* __restrict__ cursor, unsigned* __restrict__ sorted) {
    for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < nitems; i += (size_t)gridDim.x * 64)
        sorted[cursor[keys[i]] + ranks[i]] = (unsigned)i;  // order inside a bin is irrelevant (scheduling only)
}

// One pass = THREADS x P points.  Measured on MI355X, bunny shape, whole benchmark step, yz-quad layout, wide rounds:
// 64x4 (default: one wave per item, no block-level reduction) 2.94 M subcubes/s, 128x2 2.81, 256x1 2.54; 128- and 64-point
// items (64x2, 64x1) 2.77 / 2.26 — smaller items do not buy locality, they only add items.  Dragon shape: 64x4 = 128x2.
// Fewer resident blocks per CU (LDS padding) only hurts — the kernel wants every wave slot and many gathers in flight.
// WPG > 1 (THREADS == 64 only): a workgroup is WPG independent one-wave items that are NEIGHBOURS in the sorted order, so they run
// on one CU at the same time and share its L1 (the XCD remap alone spreads neighbours over the 32 CUs of an XCD: they share
// the L2 only).  NT = 1: the source points are loaded non-temporally (they stream through once per item; kept out of the L1
// they leave it to the LUT lines).
// Trimmed mode: the per-point e of an output row, and — samp_shift > 0 — one point of every run of 2^samp_shift points of the stored cloud
// once more in a compact SAMPLE behind the row (offset: ns rounded up to 64 floats): a systematic sample of the row that the selection
// reads first (125 KB instead of 4 MB at 1M points) to bracket the cut, so that it needs ONE pass over the row instead of two
// (trim_rows_sampled_kernel).  The sample only steers; the selection verifies the bracket exactly and falls back if it is wrong.
__host__ __device__ __forceinline__ size_t trim_sample_offset(int ns) { return ((size_t)ns + 63) & ~(size_t)63; }  // the sample starts on a 256-byte boundary of its row
// Which point of run r (2^samp_shift consecutive points) is its sample: a hashed position, not the first one.  The stored order
// is structured (k-d order: position 0 of every 64-point run is a corner of its cell), and a sample that always takes the same
// position is a biased sample of the row — the bracket then misses and the selection falls back to two passes (13 % of the rows
// of a surface cloud in k-d order against 1-2 % in Hilbert order; with the hashed position both orders are at 1-2 %).  One per
// run either way (stratified); the last, partial run takes its first point.
__device__ __forceinline__ bool trim_is_sample(int i, int samp_shift, int ns) {
    const int mask = (1 << samp_shift) - 1, run = i >> samp_shift;
    int pos = (int)(((unsigned)run * 0x9E3779B1u) >> 16) & mask;
    if (((run << samp_shift) | pos) >= ns) pos = 0;
    return (i & mask) == pos;
}
__device__ __forceinline__ void trim_store(float* __restrict__ evals, size_t row_base, int i, float e, int samp_shift, int ns) {
    evals[row_base + (size_t)i] = e;
    if (samp_shift > 0 && trim_is_sample(i, samp_shift, ns)) evals[row_base + trim_sample_offset(ns) + (size_t)(i >> samp_shift)] = e;
}

#include "bounds_item.hpp"   // round 4: the bounds kernel of the sorted path (bounds_item_kernel)

typedef float v4f __attribute__((ext_vector_type(4)));
// FGOICP_BOUNDS_WAVES (development builds, tools/ab_waves.sh): ask the register allocator for that many resident waves per SIMD
#ifdef FGOICP_BOUNDS_WAVES
#define FGOICP_BOUNDS_OCC __attribute__((amdgpu_waves_per_eu(FGOICP_BOUNDS_WAVES, FGOICP_BOUNDS_WAVES)))
#else
#define FGOICP_BOUNDS_OCC
#endif
template <int THREADS, int P, int ZPAIR, int TRIM, int WPG = 1, int NT = 0>
__global__ __launch_bounds__(THREADS * WPG) FGOICP_BOUNDS_OCC void bounds_sorted_kernel(const float4* __restrict__ src, int ns, const float* __restrict__ lut,
                                                                const float2* __restrict__ zp, LutGeom g,
                                                                const TickGroup* __restrict__ groups, const TickSub* __restrict__ subs,
                                                                const unsigned* __restrict__ sorted, int nchunk, int chunk_pts,
                                                                double2* __restrict__ partials, float* __restrict__ evals, size_t erow, int samp_shift, unsigned nitems) {
    static_assert(THREADS % 64 == 0 && THREADS * P <= kBlock, "one pass covers THREADS * P points");
    static_assert(WPG == 1 || THREADS == 64, "several items per workgroup: one wave each");
    __shared__ double red[4 * (THREADS / 64)];
    const unsigned slot = xcd_remap(blockIdx.x, gridDim.x) * WPG + (WPG > 1 ? threadIdx.x >> 6 : 0);
    if (WPG > 1 && slot >= nitems) return;
    const unsigned item = sorted ? sorted[slot] : slot;  // small ticks come unsorted
    if (item >= nitems) return;  // never taken when `sorted` is a permutation (tick_check_kernel verifies that on the device)
    const unsigned tix = WPG > 1 ? (threadIdx.x & 63u) : threadIdx.x;  // thread index inside the item
    const int s = (int)(item / (unsigned)nchunk);
    const int chunk = (int)(item - (unsigned)s * (unsigned)nchunk);
    const TickSub sb = subs[s];
    const TickGroup& gr = groups[sb.group];
    const float trans_uncertain_radius = kSqrt3 * sb.span;  // registration.cu:33
    const size_t sy = (size_t)g.px, sz = (size_t)g.px * g.py;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};  // {ub, lb} of the item's variant; a dual item: {ub, lb} with fix_rot = 1, then with fix_rot = 0
    const bool dual = sb.dual != 0;
    // an item is chunk_pts (256 .. 2048) Morton-consecutive points, walked in passes of THREADS * P: dense clouds take bigger
    // items (the patch of 256 points is only a few voxels wide there), which divides the items to sort and the partials
    for (int pass = 0; pass < chunk_pts; pass += THREADS * P) {
        float4 p[P];
        TexAddr ta[P];
        float2u v00[P], v10[P], v01[P], v11[P];
        const int first = chunk * chunk_pts + pass + (int)tix;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int i = first + k * THREADS;
            if (NT) {
                const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src + (i < ns ? i : ns - 1)));
                p[k] = make_float4(v.x, v.y, v.z, v.w);
            } else {
                p[k] = src[i < ns ? i : ns - 1];
            }
            float rx, ry, rz;
#if defined(FGOICP_ABLATE) && (FGOICP_ABLATE & 1)   // timing-only build (tools/ablate.sh): what the per-evaluation rotation costs
            rx = p[k].x; ry = p[k].y; rz = p[k].z;
#else
            rotate(gr.R, p[k].x, p[k].y, p[k].z, rx, ry, rz);
#endif
            ta[k] = lut_address(g, rx + sb.tx, ry + sb.ty, rz + sb.tz);  // :34, :323-325
#if defined(FGOICP_ABLATE) && (FGOICP_ABLATE & 4)   // timing-only build: every gather hits a 64 KiB corner of the LUT (address unit + VALU, no misses)
            ta[k].o &= (size_t)4095;
#endif
        }
        QuadPairLoads qp[(ZPAIR == 3 || ZPAIR == 4 || ZPAIR == 5) ? P : 1];
        const int odd = (int)tix & 1;
        if (ZPAIR == 3 || ZPAIR == 4 || ZPAIR == 5) {
            const unsigned nbx = (unsigned)(g.px + 3) >> 2, nby = (unsigned)(g.py + 3) >> 2;
            const unsigned nbx3 = (unsigned)(g.px + 2) / 3u, nby2 = (unsigned)(g.py + 1) >> 1;
#pragma unroll
            for (int k = 0; k < P; ++k)
                qp[k] = ZPAIR == 4 ? quad_pair_issue_bricked(reinterpret_cast<const float4*>(zp), ta[k], odd, nbx, nby)
                      : ZPAIR == 5 ? quad_pair_issue_apron(reinterpret_cast<const float4*>(zp), ta[k], odd, nbx3, nby2)
                                   : quad_pair_issue(reinterpret_cast<const float4*>(zp), ta[k], odd);
#pragma unroll
            for (int k = 0; k < P; ++k) quad_pair_finish(qp[k], odd, v00[k], v10[k], v01[k], v11[k]);
        }
#pragma unroll
        for (int k = 0; k < P; ++k) {
            if (ZPAIR == 3 || ZPAIR == 4 || ZPAIR == 5) {
            } else if (ZPAIR == 2) {
                quad_gather(reinterpret_cast<const float4*>(zp), ta[k], v00[k], v10[k], v01[k], v11[k]);
            } else if (ZPAIR == 1) {
                zpair_gather(zp, g, ta[k], v00[k], v10[k], v01[k], v11[k]);
            } else {
                const float* q = lut + ta[k].o;
                v00[k] = *(const float2u*)(q);  // default cache policy: non-temporal loads measured 2x slower here
                v10[k] = *(const float2u*)(q + sy);
                v01[k] = *(const float2u*)(q + sz);
                v11[k] = *(const float2u*)(q + sz + sy);
            }
        }
        float te0[TRIM ? P : 1], te1[TRIM ? P : 1];
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const float dsq = lut_blend(ta[k], v00[k], v10[k], v01[k], v11[k]);  // :46
            float d = sqrtf(dsq);                                                 // :48
            const int i = first + k * THREADS;
            const bool valid = i < ns;
            if (TRIM) {
                // trimmed Go-ICP: both bounds are non-decreasing functions of e = max(d, 0) (ub = e*e, lb = max(e - r_t, 0)^2 —
                // the same fp32 values as :54-58), so ONE row of e per variant carries both selections (trim_rows_kernel)
                float e0, e1 = 0.0f;
                if (dual) {
                    e0 = d > 0.0f ? d : 0.0f;
                    d -= 2.0f * p[k].w * gr.sin_half;
                    e1 = d > 0.0f ? d : 0.0f;
                } else {
                    if (!gr.fix_rot) d -= 2.0f * p[k].w * gr.sin_half;
                    e0 = d > 0.0f ? d : 0.0f;
                }
                if (valid) {
                    evals[(size_t)sb.out0 * erow + i] = e0;
                    if (dual) evals[(size_t)sb.out1 * erow + i] = e1;
                }
                te0[k] = e0;
                te1[k] = e1;
                continue;
            }
            if (dual) {  // wave-uniform
                const float ub1 = d > 0.0f ? d * d : 0.0f;                        // fix_rot = 1: :54
                const float l1 = d - trans_uncertain_radius;                      // :57
                const float lb1 = l1 > 0.0f ? l1 * l1 : 0.0f;                     // :58
                d -= 2.0f * p[k].w * gr.sin_half;                                 // fix_rot = 0: :39-43, :49-52
                const float ub0 = d > 0.0f ? d * d : 0.0f;
                const float l0 = d - trans_uncertain_radius;
                const float lb0 = l0 > 0.0f ? l0 * l0 : 0.0f;
                acc[0] += valid ? (double)ub1 : 0.0;
                acc[1] += valid ? (double)lb1 : 0.0;
                acc[2] += valid ? (double)ub0 : 0.0;
                acc[3] += valid ? (double)lb0 : 0.0;
                continue;
            }
            if (!gr.fix_rot) d -= 2.0f * p[k].w * gr.sin_half;                    // :39-43, :49-52
            const float ubv = d > 0.0f ? d * d : 0.0f;                            // :54
            const float l = d - trans_uncertain_radius;                           // :57
            const float lbv = l > 0.0f ? l * l : 0.0f;                            // :58
            acc[0] += valid ? (double)ubv : 0.0;
            acc[1] += valid ? (double)lbv : 0.0;
        }
        if (TRIM && samp_shift > 0) {
            // the row's sample (see trim_store / trim_is_sample): one point of every run of 2^samp_shift once more behind the row
            const size_t off = trim_sample_offset(ns);
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const int i = first + k * THREADS;
                if (i < ns && trim_is_sample(i, samp_shift, ns)) {
                    evals[(size_t)sb.out0 * erow + off + (size_t)(i >> samp_shift)] = te0[k];
                    if (dual) evals[(size_t)sb.out1 * erow + off + (size_t)(i >> samp_shift)] = te1[k];
                }
            }
        }
    }
    if (!TRIM && WPG > 1) {  // one wave per item, several items per workgroup: the wave tree only (= block_sum with one wave), no barrier
        const double r0 = wave_sum(acc[0]), r1 = wave_sum(acc[1]);
        if (dual) {
            const double r2 = wave_sum(acc[2]), r3 = wave_sum(acc[3]);
            if (tix == 0) {
                partials[(size_t)sb.out0 * nchunk + chunk] = make_double2(r0, r1);
                partials[(size_t)sb.out1 * nchunk + chunk] = make_double2(r2, r3);
            }
        } else if (tix == 0) {
            partials[(size_t)sb.out0 * nchunk + chunk] = make_double2(r0, r1);
        }
    } else if (!TRIM) {
        // sums 0, 1 land in threads 0, 1 (dual: sums 2, 3 in threads 2, 3); same reduction tree per sum either way
        if (dual) {
            const double r = block_sum<4, THREADS / 64>(acc, red);
            if (threadIdx.x < 4) {
                double* out = reinterpret_cast<double*>(partials + ((size_t)(threadIdx.x < 2 ? sb.out0 : sb.out1) * nchunk + chunk));
                out[threadIdx.x & 1] = r;
            }
        } else {
            const double a2[2] = {acc[0], acc[1]};
            const double r = block_sum<2, THREADS / 64>(a2, red);
            double* out = reinterpret_cast<double*>(partials + ((size_t)sb.out0 * nchunk + chunk));
            if (threadIdx.x < 2) out[threadIdx.x] = r;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Sibling units (round 3; dense clouds).  The inner BnB (fgoicp.cpp:157-168) pushes the eight children of a translation node with
// one key, so they are popped together: almost every evaluation of a tick has its seven siblings next to it — the same rotation,
// the same points, translations 2 * span apart.  bounds_sorted_kernel re-loads and re-rotates the chunk's points for each of them
// (16 B per point-evaluation against the 0.375 B of SURVEY 8d; at 1M points the cloud is 16 MB and no L2 holds it).  Here an item is
// (unit of M siblings, chunk): the wave loads and rotates its 4 points per lane ONCE per pass (the reference's TODO.md:11, "avoid
// repeated rotation computation") and walks the M translations with them — per sibling the same lookups, the same per-point
// expressions and the same per-lane accumulation order as bounds_sorted_kernel, one partial per (subcube, chunk) as before, so
// every sum keeps its bits.  Items are keyed by the unit's mean translation (tick_keys_kernel); evaluations that are not in a
// unit follow as one-sibling items.  M = 8 for the trimmed kernel (no accumulators), 4 or 8 otherwise (2 fp64 sums per sibling in
// registers).  Dual (twin) evaluations are grouped only in the trimmed kernel.
// ---------------------------------------------------------------------------------------------
template <int ZPAIR /* 0 plain, 1 z-pair, 2 yz-quad (lane-paired) */, int TRIM, int M>
__global__ __launch_bounds__(64) void bounds_units_kernel(const float4* __restrict__ src, int ns, const float* __restrict__ lut, const float2* __restrict__ zp, LutGeom g,
                                                          const TickGroup* __restrict__ groups, const TickSub* __restrict__ subs, const unsigned* __restrict__ sorted,
                                                          int nchunk, int chunk_pts, double2* __restrict__ partials, float* __restrict__ evals, size_t erow, int samp_shift,
                                                          unsigned nitems, int nunits) {
    constexpr int P = 4;
    const unsigned slot = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned item = sorted ? sorted[slot] : slot;
    if (item >= nitems) return;
    const unsigned unit_items = (unsigned)nunits * (unsigned)nchunk;
    int s0, chunk, count;
    if (item < unit_items) {
        const int u = (int)(item / (unsigned)nchunk);
        chunk = (int)(item - (unsigned)u * (unsigned)nchunk);
        s0 = u * M;
        count = M;
    } else {
        const unsigned r = item - unit_items;
        const int q = (int)(r / (unsigned)nchunk);
        chunk = (int)(r - (unsigned)q * (unsigned)nchunk);
        s0 = nunits * M + q;
        count = 1;
    }
    const int tix = (int)threadIdx.x;
    const TickSub sb0 = subs[s0];
    const TickGroup& gr = groups[sb0.group];  // one rotation node per unit
    const bool dual = sb0.dual != 0;           // ... and one kind (the host groups only evaluations of the same kind)
    const size_t sy = (size_t)g.px, sz = (size_t)g.px * g.py;
    double acc[TRIM ? 1 : 2 * M];
#pragma unroll
    for (int k = 0; k < (TRIM ? 1 : 2 * M); ++k) acc[k] = 0.0;
    double accd[4] = {0.0, 0.0, 0.0, 0.0};  // a dual one-sibling item of the untrimmed kernel (as bounds_sorted_kernel)
    const int odd = tix & 1;
    for (int pass = 0; pass < chunk_pts; pass += 64 * P) {
        float4 p[P];
        float rx[P], ry[P], rz[P];
        const int first = chunk * chunk_pts + pass + tix;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int i = first + k * 64;
            p[k] = src[i < ns ? i : ns - 1];
            rotate(gr.R, p[k].x, p[k].y, p[k].z, rx[k], ry[k], rz[k]);
        }
#pragma unroll
        for (int j = 0; j < M; ++j) {
            if (j >= count) break;  // wave-uniform
            const TickSub sb = subs[s0 + j];
            const float trans_uncertain_radius = kSqrt3 * sb.span;  // registration.cu:33
            TexAddr ta[P];
            float2u v00[P], v10[P], v01[P], v11[P];
#pragma unroll
            for (int k = 0; k < P; ++k) ta[k] = lut_address(g, rx[k] + sb.tx, ry[k] + sb.ty, rz[k] + sb.tz);  // :34, :323-325
            if (ZPAIR == 2) {
                QuadPairLoads qp[P];
#pragma unroll
                for (int k = 0; k < P; ++k) qp[k] = quad_pair_issue(reinterpret_cast<const float4*>(zp), ta[k], odd);
#pragma unroll
                for (int k = 0; k < P; ++k) quad_pair_finish(qp[k], odd, v00[k], v10[k], v01[k], v11[k]);
            } else {
#pragma unroll
                for (int k = 0; k < P; ++k) {
                    if (ZPAIR == 1) {
                        zpair_gather(zp, g, ta[k], v00[k], v10[k], v01[k], v11[k]);
                    } else {
                        const float* q = lut + ta[k].o;
                        v00[k] = *(const float2u*)(q);
                        v10[k] = *(const float2u*)(q + sy);
                        v01[k] = *(const float2u*)(q + sz);
                        v11[k] = *(const float2u*)(q + sz + sy);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const float dsq = lut_blend(ta[k], v00[k], v10[k], v01[k], v11[k]);  // :46
                float d = sqrtf(dsq);                                                 // :48
                const int i = first + k * 64;
                const bool valid = i < ns;
                if (TRIM) {
                    if (valid) {
                        if (dual) {
                            trim_store(evals, (size_t)sb.out0 * erow, i, d > 0.0f ? d : 0.0f, samp_shift, ns);
                            d -= 2.0f * p[k].w * gr.sin_half;
                            trim_store(evals, (size_t)sb.out1 * erow, i, d > 0.0f ? d : 0.0f, samp_shift, ns);
                        } else {
                            if (!gr.fix_rot) d -= 2.0f * p[k].w * gr.sin_half;
                            trim_store(evals, (size_t)sb.out0 * erow, i, d > 0.0f ? d : 0.0f, samp_shift, ns);
                        }
                    }
                    continue;
                }
                if (dual) {  // wave-uniform; one-sibling items only (the host does not group dual evaluations for this kernel)
                    const float ub1 = d > 0.0f ? d * d : 0.0f;                        // fix_rot = 1: :54
                    const float l1 = d - trans_uncertain_radius;                      // :57
                    const float lb1 = l1 > 0.0f ? l1 * l1 : 0.0f;                     // :58
                    d -= 2.0f * p[k].w * gr.sin_half;                                 // fix_rot = 0: :39-43, :49-52
                    const float ub0 = d > 0.0f ? d * d : 0.0f;
                    const float l0 = d - trans_uncertain_radius;
                    const float lb0 = l0 > 0.0f ? l0 * l0 : 0.0f;
                    accd[0] += valid ? (double)ub1 : 0.0;
                    accd[1] += valid ? (double)lb1 : 0.0;
                    accd[2] += valid ? (double)ub0 : 0.0;
                    accd[3] += valid ? (double)lb0 : 0.0;
                    continue;
                }
                if (!gr.fix_rot) d -= 2.0f * p[k].w * gr.sin_half;                    // :39-43, :49-52
                const float ubv = d > 0.0f ? d * d : 0.0f;                            // :54
                const float l = d - trans_uncertain_radius;                           // :57
                const float lbv = l > 0.0f ? l * l : 0.0f;                            // :58
                acc[TRIM ? 0 : 2 * j] += valid ? (double)ubv : 0.0;
                acc[TRIM ? 0 : 2 * j + 1] += valid ? (double)lbv : 0.0;
            }
        }
    }
    if (TRIM) return;
    if (dual) {  // count == 1
        const double r0 = wave_sum(accd[0]), r1 = wave_sum(accd[1]), r2 = wave_sum(accd[2]), r3 = wave_sum(accd[3]);
        if (tix == 0) {
            partials[(size_t)sb0.out0 * nchunk + chunk] = make_double2(r0, r1);
            partials[(size_t)sb0.out1 * nchunk + chunk] = make_double2(r2, r3);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < M; ++j) {
        if (j >= count) break;
        const double r0 = wave_sum(acc[TRIM ? 0 : 2 * j]), r1 = wave_sum(acc[TRIM ? 0 : 2 * j + 1]);
        if (tix == 0) partials[(size_t)subs[s0 + j].out0 * nchunk + chunk] = make_double2(r0, r1);
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-staged LUT tiles (round 3; north_star names them, the reference wished for them: TODO.md:12).  One wave per item as in
// bounds_sorted_kernel (64 x 4).  Per pass of 256 Hilbert-consecutive points the wave
//   1. computes the points' first texels and takes their bounding brick (wave min / max of the three padded indices),
//   2. if the brick is at most 16 nodes wide and ROWS rows (y x z) high, copies it from the plain fp32 LUT into LDS — 16 lanes per
//      row, four rows per load instruction, rows padded to 16 floats, so the staging loads are row-coalesced —
//   3. and reads the 2 x 2 x 2 footprints of its points from LDS (four 8-byte reads per point) instead of gathering them from
//      global memory; a pass whose brick does not fit gathers from the plain LUT as before.
// Same texels, same blend, same per-lane accumulation order: bit-identical sums (tests).  Whether it pays is a question of how
// many lookups share a staged node: 256 points of a surface patch touch a brick of (patch extent + 2)^2 x depth nodes.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}
template <int TRIM, int ROWS>
__global__ __launch_bounds__(64) void bounds_lds_kernel(const float4* __restrict__ src, int ns, const float* __restrict__ lut, LutGeom g,
                                                        const TickGroup* __restrict__ groups, const TickSub* __restrict__ subs, const unsigned* __restrict__ sorted,
                                                        int nchunk, int chunk_pts, double2* __restrict__ partials, float* __restrict__ evals, size_t erow, int samp_shift,
                                                        unsigned nitems, unsigned* __restrict__ stat /* optional: [0] staged passes, [1] all passes */) {
    constexpr int P = 4;
    __shared__ float tile[ROWS * 16];
    const unsigned slot = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned item = sorted ? sorted[slot] : slot;
    if (item >= nitems) return;
    const int s = (int)(item / (unsigned)nchunk);
    const int chunk = (int)(item - (unsigned)s * (unsigned)nchunk);
    const int tix = (int)threadIdx.x;
    const TickSub sb = subs[s];
    const TickGroup& gr = groups[sb.group];
    const float trans_uncertain_radius = kSqrt3 * sb.span;  // registration.cu:33
    const bool dual = sb.dual != 0;
    const size_t sy = (size_t)g.px, sz = (size_t)g.px * g.py;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    unsigned staged_passes = 0, all_passes = 0;
    for (int pass = 0; pass < chunk_pts; pass += 64 * P) {
        float4 p[P];
        TexAddr ta[P];
        float2u v00[P], v10[P], v01[P], v11[P];
        const int first = chunk * chunk_pts + pass + tix;
        int lo[3] = {1 << 20, 1 << 20, 1 << 20}, hi[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int i = first + k * 64;
            p[k] = src[i < ns ? i : ns - 1];
            float rx, ry, rz;
            rotate(gr.R, p[k].x, p[k].y, p[k].z, rx, ry, rz);
            ta[k] = lut_address(g, rx + sb.tx, ry + sb.ty, rz + sb.tz);  // :34, :323-325
            const int ix = (int)(ta[k].pk & 1023u), iy = (int)((ta[k].pk >> 10) & 1023u), iz = (int)(ta[k].pk >> 20);
            lo[0] = min(lo[0], ix); lo[1] = min(lo[1], iy); lo[2] = min(lo[2], iz);
            hi[0] = max(hi[0], ix); hi[1] = max(hi[1], iy); hi[2] = max(hi[2], iz);
        }
        // the pass's brick: [lo, hi + 1] per axis (wave-uniform after the reductions)
        int bx, by, bz;
        {
            const int x0 = wave_min_i(lo[0]), y0 = wave_min_i(lo[1]), z0 = wave_min_i(lo[2]);
            bx = wave_max_i(hi[0]) - x0 + 2; by = wave_max_i(hi[1]) - y0 + 2; bz = wave_max_i(hi[2]) - z0 + 2;
            lo[0] = x0; lo[1] = y0; lo[2] = z0;
        }
        const int rows = by * bz;
        const bool fits = bx <= 16 && rows <= ROWS && g.px < 1024 && g.py < 1024 && g.pz < 1024;  // `pk` holds 10 bits per index
        ++all_passes;
        if (fits) {  // wave-uniform
            ++staged_passes;
            const int sub = tix >> 4, xl = tix & 15;
            int y = sub % by, z = sub / by;  // row r = y + by * z, rows r0 + sub for r0 = 0, 4, 8, ...
            const int xg = min(lo[0] + xl, g.px - 1);  // lanes beyond the brick's width copy in-range neighbours nobody reads
            for (int r = sub; r < rows; r += 4) {
                tile[r * 16 + xl] = lut[((size_t)(lo[2] + z) * g.py + (size_t)(lo[1] + y)) * g.px + (size_t)xg];
                y += 4;
                while (y >= by) { y -= by; ++z; }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const int ix = (int)(ta[k].pk & 1023u) - lo[0], iy = (int)((ta[k].pk >> 10) & 1023u) - lo[1], iz = (int)(ta[k].pk >> 20) - lo[2];
                const float* t0 = tile + ((iz * by + iy) * 16 + ix);
                v00[k] = float2u{t0[0], t0[1]};                       // (x0, x1) at (y0, z0)
                v10[k] = float2u{t0[16], t0[17]};                     // (y1, z0)
                v01[k] = float2u{t0[by * 16], t0[by * 16 + 1]};       // (y0, z1)
                v11[k] = float2u{t0[by * 16 + 16], t0[by * 16 + 17]}; // (y1, z1)
            }
            __syncthreads();  // the next pass overwrites the tile
        } else {
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const float* q = lut + ta[k].o;
                v00[k] = *(const float2u*)(q);
                v10[k] = *(const float2u*)(q + sy);
                v01[k] = *(const float2u*)(q + sz);
                v11[k] = *(const float2u*)(q + sz + sy);
            }
        }
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const float dsq = lut_blend(ta[k], v00[k], v10[k], v01[k], v11[k]);  // :46
            float d = sqrtf(dsq);                                                 // :48
            const int i = first + k * 64;
            const bool valid = i < ns;
            if (TRIM) {
                if (valid) {
                    if (dual) {
                        trim_store(evals, (size_t)sb.out0 * erow, i, d > 0.0f ? d : 0.0f, samp_shift, ns);
                        d -= 2.0f * p[k].w * gr.sin_half;
                        trim_store(evals, (size_t)sb.out1 * erow, i, d > 0.0f ? d : 0.0f, samp_shift, ns);
                    } else {
                        if (!gr.fix_rot) d -= 2.0f * p[k].w * gr.sin_half;
                        trim_store(evals, (size_t)sb.out0 * erow, i, d > 0.0f ? d : 0.0f, samp_shift, ns);
                    }
                }
                continue;
            }
            if (dual) {
                const float ub1 = d > 0.0f ? d * d : 0.0f;
                const float l1 = d - trans_uncertain_radius;
                const float lb1 = l1 > 0.0f ? l1 * l1 : 0.0f;
                d -= 2.0f * p[k].w * gr.sin_half;
                const float ub0 = d > 0.0f ? d * d : 0.0f;
                const float l0 = d - trans_uncertain_radius;
                const float lb0 = l0 > 0.0f ? l0 * l0 : 0.0f;
                acc[0] += valid ? (double)ub1 : 0.0;
                acc[1] += valid ? (double)lb1 : 0.0;
                acc[2] += valid ? (double)ub0 : 0.0;
                acc[3] += valid ? (double)lb0 : 0.0;
                continue;
            }
            if (!gr.fix_rot) d -= 2.0f * p[k].w * gr.sin_half;
            const float ubv = d > 0.0f ? d * d : 0.0f;
            const float l = d - trans_uncertain_radius;
            const float lbv = l > 0.0f ? l * l : 0.0f;
            acc[0] += valid ? (double)ubv : 0.0;
            acc[1] += valid ? (double)lbv : 0.0;
        }
    }
    if (stat && tix == 0) { atomicAdd(&stat[0], staged_passes); atomicAdd(&stat[1], all_passes); }
    if (TRIM) return;
    const double r0 = wave_sum(acc[0]), r1 = wave_sum(acc[1]);
    if (dual) {
        const double r2 = wave_sum(acc[2]), r3 = wave_sum(acc[3]);
        if (tix == 0) {
            partials[(size_t)sb.out0 * nchunk + chunk] = make_double2(r0, r1);
            partials[(size_t)sb.out1 * nchunk + chunk] = make_double2(r2, r3);
        }
    } else if (tix == 0) {
        partials[(size_t)sb.out0 * nchunk + chunk] = make_double2(r0, r1);
    }
}

// ---------------------------------------------------------------------------------------------
// EXTENSION — trimmed Go-ICP (no reference behaviour: `params.trim` is parsed and ignored upstream).
// Per output row (one subcube variant) the bounds kernel leaves n values e_i = max(d_i, 0) >= 0; with r_t = sqrt3 * span
//     ub_i = e_i * e_i,   lb_i = max(e_i - r_t, 0)^2        (registration.cu:54-58, same fp32 values)
// are both non-decreasing in e_i, so the k smallest ub terms and the k smallest lb terms belong to the k smallest e_i:
// ONE exact selection per row serves both sums.  One 1024-thread workgroup per row, normally two passes over the row:
//   1. histogram of e over 8192 bins, 512 per octave on [2^-14, 4) (exact zeros — d <= rotation radius — are only counted:
//      if the k smallest are all zero both sums are zero and the row is done after this pass);
//   2. fp64 sums of ub_i, lb_i over everything below the cut's bin, its own members (a few hundred) gathered into LDS,
//      sorted there (bitonic) and the first k - below of them added in sorted order.
// A bin that holds more than kTrimCap members is refined (one more pass per 13 bits) before the gather.  Every sum has a
// fixed order (thread-strided, wave tree, waves in order; gathered members in sorted order): bit-reproducible.
// ---------------------------------------------------------------------------------------------
constexpr int kTrimThreads = 1024;
constexpr int kTrimBins = 8192;
constexpr int kTrimCap = 8192;
constexpr unsigned kTrimLo = 0x38800000u;  // bits(2^-14f); level-0 bins are 2^14 bit patterns wide (1/512 octave)

__device__ __forceinline__ unsigned trim_bin0(unsigned u) {
    const unsigned b = ((u > kTrimLo ? u : kTrimLo) - kTrimLo) >> 14;
    return b < (unsigned)(kTrimBins - 1) ? b : (unsigned)(kTrimBins - 1);
}

// The bin holding the element of rank `need` (1-based) of an 8192-bin LDS histogram and how many elements lie in the bins
// before it; every thread returns the same pair.  s_pick: two words of LDS.
__device__ __forceinline__ void trim_pick(const unsigned* hist, unsigned need, unsigned* wsum, unsigned* s_pick, unsigned& bin, unsigned& below) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int per = kTrimBins / kTrimThreads;  // 8 consecutive bins per thread
    unsigned mine = 0;
#pragma unroll
    for (int b = 0; b < per; ++b) mine += hist[tid * per + b];
    unsigned incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned u = __shfl_up(incl, off, 64);
        if (lane >= off) incl += u;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    const unsigned excl = base + incl - mine;
    if (excl < need && need <= excl + mine) {  // exactly one thread
        unsigned bl = excl;
        int b = 0;
        for (; b < per - 1; ++b) {
            const unsigned c = hist[tid * per + b];
            if (need <= bl + c) break;
            bl += c;
        }
        s_pick[0] = (unsigned)(tid * per + b);
        s_pick[1] = bl;
    }
    __syncthreads();
    bin = s_pick[0];
    below = s_pick[1];
    __syncthreads();
}

template <class F>
__device__ __forceinline__ void trim_for_each(const float* __restrict__ v, int n, F f) {  // rows start 16-byte aligned
    const int n4 = n >> 2;
    const float4* v4 = reinterpret_cast<const float4*>(v);
    int i = threadIdx.x;
    for (; i + kTrimThreads < n4; i += 2 * kTrimThreads) {  // two 16-byte loads in flight per lane
        const float4 a = v4[i], b = v4[i + kTrimThreads];
        f(a.x); f(a.y); f(a.z); f(a.w);
        f(b.x); f(b.y); f(b.z); f(b.w);
    }
    if (i < n4) { const float4 a = v4[i]; f(a.x); f(a.y); f(a.z); f(a.w); }
    const int t = (n4 << 2) + threadIdx.x;
    if (t < n) f(v[t]);
}

template <int CAP>
__device__ __forceinline__ void trim_row_two_pass(const float* __restrict__ v, int n, int k, float rt, unsigned* hist /* kTrimBins */, unsigned* list /* CAP */,
                                                  unsigned* wsum, unsigned* s_pick, unsigned* s_count_p, double* red, float* __restrict__ out_ub,
                                                  float* __restrict__ out_lb, int row) {
    static_assert((CAP & (CAP - 1)) == 0, "the bitonic sort pads to a power of two inside the buffer");
    unsigned& s_count = *s_count_p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // pass 1: zeros counted, the rest into the level-0 histogram
    for (int b = tid; b < kTrimBins; b += kTrimThreads) hist[b] = 0;
    __syncthreads();
    unsigned nz = 0;
    trim_for_each(v, n, [&](float x) {
        const unsigned u = __float_as_uint(x);
        if (u == 0u) ++nz; else atomicAdd(&hist[trim_bin0(u)], 1u);
    });
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nz += __shfl_xor(nz, off, 64);
    if (lane == 0) wsum[wave] = nz;
    __syncthreads();
    unsigned zeros = 0;
    for (int w = 0; w < kTrimThreads / 64; ++w) zeros += wsum[w];
    __syncthreads();
    if ((unsigned)k <= zeros) {  // the k smallest terms are all zero
        if (tid == 0) { out_ub[row] = 0.0f; out_lb[row] = 0.0f; }
        return;
    }
    unsigned bin, before;
    trim_pick(hist, (unsigned)k - zeros, wsum, s_pick, bin, before);
    unsigned below = zeros + before;  // elements strictly below the cut's range [a, b)
    unsigned cnt = hist[bin];
    unsigned long long a = bin == 0 ? 1ull : (unsigned long long)kTrimLo + ((unsigned long long)bin << 14);
    unsigned long long b = bin == kTrimBins - 1 ? 0x100000000ull : (unsigned long long)kTrimLo + ((unsigned long long)(bin + 1) << 14);
    __syncthreads();
    // refinement (rare): more members than the gather buffer holds
    while (cnt > (unsigned)CAP && b - a > 1ull) {
        int sh = 0;
        while (((b - a - 1ull) >> sh) >= (unsigned long long)kTrimBins) ++sh;
        for (int q = tid; q < kTrimBins; q += kTrimThreads) hist[q] = 0;
        __syncthreads();
        const unsigned lo = (unsigned)a;
        const unsigned long long hi = b;
        trim_for_each(v, n, [&](float x) {
            const unsigned u = __float_as_uint(x);
            if (u >= lo && (unsigned long long)u < hi) atomicAdd(&hist[(u - lo) >> sh], 1u);
        });
        __syncthreads();
        trim_pick(hist, (unsigned)k - below, wsum, s_pick, bin, before);
        below += before;
        cnt = hist[bin];
        const unsigned long long a2 = a + ((unsigned long long)bin << sh);
        const unsigned long long b2 = a2 + (1ull << sh);
        a = a2;
        b = b2 < b ? b2 : b;
        __syncthreads();
    }
    // pass 2: sums below the range, members of the range into LDS
    const bool gather = cnt <= (unsigned)CAP;
    if (tid == 0) s_count = 0;
    __syncthreads();
    double acc[2] = {0.0, 0.0};
    {
        const unsigned lo = (unsigned)a;
        const unsigned long long hi = b;
        trim_for_each(v, n, [&](float x) {
            const unsigned u = __float_as_uint(x);
            if (u < lo) {
                const float l = x - rt;
                acc[0] += (double)(x * x);
                acc[1] += (double)(l > 0.0f ? l * l : 0.0f);
            } else if (gather && (unsigned long long)u < hi) {
                list[atomicAdd(&s_count, 1u)] = u;
            }
        });
    }
    __syncthreads();
    const unsigned m = (unsigned)k - below;  // members of the range among the k smallest (1 <= m <= cnt)
    if (gather) {
        unsigned P = 64;
        while (P < cnt) P <<= 1;
        for (unsigned q = cnt + tid; q < P; q += kTrimThreads) list[q] = 0xFFFFFFFFu;
        __syncthreads();
        for (unsigned size = 2; size <= P; size <<= 1) {
            for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
                for (unsigned idx = tid; idx < (P >> 1); idx += kTrimThreads) {
                    const unsigned i = 2 * idx - (idx & (stride - 1)), j = i + stride;
                    const unsigned x = list[i], y = list[j];
                    const bool up = (i & size) == 0;
                    if ((x > y) == up) { list[i] = y; list[j] = x; }
                }
                __syncthreads();
            }
        }
        for (unsigned q = tid; q < m; q += kTrimThreads) {
            const float x = __uint_as_float(list[q]);
            const float l = x - rt;
            acc[0] += (double)(x * x);
            acc[1] += (double)(l > 0.0f ? l * l : 0.0f);
        }
    }
    const double r = block_sum<2, kTrimThreads / 64>(acc, red);
    if (tid < 2) {
        double extra = 0.0;
        if (!gather) {  // b - a == 1: every member of the range is the same value
            const float x = __uint_as_float((unsigned)a);
            const float l = x - rt;
            extra = (double)m * (double)(tid == 0 ? x * x : (l > 0.0f ? l * l : 0.0f));
        }
        (tid == 0 ? out_ub : out_lb)[row] = (float)(r + extra);
    }
}


__global__ __launch_bounds__(kTrimThreads) void trim_rows_kernel(const float* __restrict__ evals, size_t erow, int n, int k, const float* __restrict__ row_span,
                                                                 float* __restrict__ out_ub, float* __restrict__ out_lb) {
    __shared__ unsigned hist[kTrimBins];
    __shared__ unsigned list[kTrimCap];
    __shared__ unsigned wsum[kTrimThreads / 64];
    __shared__ unsigned s_pick[2];
    __shared__ unsigned s_count;
    __shared__ double red[2 * (kTrimThreads / 64)];
    const int row = blockIdx.x;
    trim_row_two_pass<kTrimCap>(evals + (size_t)row * erow, n, k, kSqrt3 * row_span[row] /* registration.cu:33 */, hist, list, wsum, s_pick, &s_count, red, out_ub, out_lb, row);
}

// ---------------------------------------------------------------------------------------------
// One pass per row (round 3).  trim_rows_kernel reads a row twice (histogram, then sums) = 8 B per point-row on top of the 4 B the
// bounds kernel wrote.  Here the bounds kernel also leaves a systematic SAMPLE of the row behind it (trim_store: every 2^samp_shift-th
// point of the Hilbert-ordered cloud, 1/32 of the row), and the selection
//   (0) histograms the sample (level-0 bins) and takes the values at sample ranks r -+ margin around the cut's expected rank as a
//       BRACKET [a, bmax] of bit patterns;
//   (1) streams the row ONCE: zeros counted, everything below a summed (fp64, per thread) and counted, the members of the bracket
//       compacted into LDS — per wave in a segment of its own, in the order (load slot, lane), so the list does not depend on timing;
//   (2) verifies the bracket EXACTLY: with z zeros and c elements in (0, a), the k-th smallest lies in the bracket iff
//       1 <= k - z - c <= members, and no segment overflowed.  If not (a poor sample, a wide bracket), the row is done again by
//       the two-pass algorithm — the sample steers, it never decides;
//   (3) radix-selects the (k - z - c)-th smallest member in LDS (8 bits per round over the bracket's width) and adds the members
//       below it in list order plus the needed copies of it.
// Every sum has a fixed order (thread-strided row, own segment lane-strided, wave tree, waves in order): bit-reproducible.  The
// value can differ from trim_rows_kernel's in the last bits of the fp64 sum (other grouping of the same terms): tests compare both
// with the oracle at 1e-6.
// ---------------------------------------------------------------------------------------------
constexpr int kTrimSegCap = 960;                       // members per wave segment: 16 x 960 x 4 B = 60 KB of LDS
constexpr int kTrimFallbackCap = 4096;                 // gather buffer of the in-kernel fallback (hist 32 KB + list 16 KB share that LDS)

__global__ __launch_bounds__(kTrimThreads) void trim_rows_sampled_kernel(const float* __restrict__ evals, size_t erow, int n, int k, int samp_shift, int margin,
                                                                         const float* __restrict__ row_span, float* __restrict__ out_ub, float* __restrict__ out_lb,
                                                                         unsigned long long* __restrict__ stat /* optional: [0] rows, [1] fallbacks, [2] members */) {
    constexpr int kWaves = kTrimThreads / 64;
    __shared__ unsigned buf[kWaves * kTrimSegCap];  // sample histogram (8192 bins), then the member segments; fallback: hist + list
    static_assert(kWaves * kTrimSegCap >= kTrimBins + kTrimFallbackCap, "the fallback's histogram and gather buffer live in the same LDS");
    __shared__ unsigned h256[256];
    __shared__ unsigned wsum[kWaves];
    __shared__ unsigned wcount[kWaves];
    __shared__ unsigned s_pick[2];
    __shared__ unsigned s_count;
    __shared__ unsigned s_tot[4];
    __shared__ double red[2 * kWaves];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x;
    const float* v = evals + (size_t)row * erow;
    const float rt = kSqrt3 * row_span[row];  // registration.cu:33
    const int nsamp = (n + (1 << samp_shift) - 1) >> samp_shift;
    const float* sv = v + trim_sample_offset(n);

    // (0) the sample: zeros counted, the rest into the level-0 histogram
    unsigned* hist = buf;
    for (int b = tid; b < kTrimBins; b += kTrimThreads) hist[b] = 0;
    __syncthreads();
    unsigned nzs = 0;
    for (int j = tid; j < nsamp; j += kTrimThreads) {
        const unsigned u = __float_as_uint(sv[j]);
        if (u == 0u) ++nzs; else atomicAdd(&hist[trim_bin0(u)], 1u);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nzs += __shfl_xor(nzs, off, 64);
    if (lane == 0) wsum[wave] = nzs;
    __syncthreads();
    unsigned zeros_s = 0;
    for (int w = 0; w < kWaves; ++w) zeros_s += wsum[w];
    __syncthreads();
    const unsigned pos_s = (unsigned)nsamp - zeros_s;  // positive sample values
    unsigned a = 1u, bmax = 0xFFFFFFFFu;               // the bracket, inclusive, in bit patterns (a >= 1: zeros are never members)
    if (pos_s > 0u) {
        const long long r = ((long long)k * nsamp + n - 1) / n;  // the cut's expected rank in the sample
        long long lo_p = r - margin - (long long)zeros_s, hi_p = r + margin - (long long)zeros_s;
        if (hi_p < 1) hi_p = 1;
        unsigned bin, before;
        if (lo_p >= 1) {
            if (lo_p > (long long)pos_s) lo_p = pos_s;
            trim_pick(hist, (unsigned)lo_p, wsum, s_pick, bin, before);
            a = bin == 0 ? 1u : kTrimLo + (bin << 14);
        }
        if (hi_p <= (long long)pos_s) {
            trim_pick(hist, (unsigned)hi_p, wsum, s_pick, bin, before);
            if (bin != kTrimBins - 1) bmax = kTrimLo + ((bin + 1u) << 14) - 1u;
        }
    }
    __syncthreads();  // the histogram is dead: its LDS becomes the segments

    // (1) one pass over the row
    unsigned* seg = buf + wave * kTrimSegCap;
    unsigned nz = 0, nbelow = 0, wcnt = 0;  // wcnt: wave-uniform
    double acc[2] = {0.0, 0.0};
    auto below_a = [&](float x, unsigned u, bool valid) {
        const bool zero = valid && u == 0u, low = valid && u != 0u && u < a;
        nz += zero ? 1u : 0u;  // branch-free counters (as `if / else if` the compiler turned the two into a scratch array indexed by the case)
        nbelow += low ? 1u : 0u;
        if (low) {
            const float l = x - rt;
            acc[0] += (double)(x * x);
            acc[1] += (double)(l > 0.0f ? l * l : 0.0f);
        }
    };
    auto member = [&](unsigned u, bool in) {  // wave-level compaction in the order (call, lane)
        const unsigned long long m = __ballot(in);
        if (in) {
            const unsigned pos = wcnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (pos < (unsigned)kTrimSegCap) seg[pos] = u;
        }
        wcnt += (unsigned)__popcll(m);
    };
    auto visit4 = [&](const float4& p, bool valid) {
        const unsigned u0 = __float_as_uint(p.x), u1 = __float_as_uint(p.y), u2 = __float_as_uint(p.z), u3 = __float_as_uint(p.w);
        below_a(p.x, u0, valid); below_a(p.y, u1, valid); below_a(p.z, u2, valid); below_a(p.w, u3, valid);
        const bool i0 = valid && u0 >= a && u0 <= bmax, i1 = valid && u1 >= a && u1 <= bmax, i2 = valid && u2 >= a && u2 <= bmax, i3 = valid && u3 >= a && u3 <= bmax;
        if (__ballot(i0 | i1 | i2 | i3)) {  // rare: a few per cent of the row lie in the bracket
            member(u0, i0); member(u1, i1); member(u2, i2); member(u3, i3);
        }
    };
    {
        const int n4 = n >> 2;
        const float4* v4 = reinterpret_cast<const float4*>(v);
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int base = 0; base < n4; base += 4 * kTrimThreads) {  // every lane of a wave runs the same trips (the ballots need that)
            const int i0 = base + tid, i1 = i0 + kTrimThreads, i2 = i1 + kTrimThreads, i3 = i2 + kTrimThreads;
            const bool ok0 = i0 < n4, ok1 = i1 < n4, ok2 = i2 < n4, ok3 = i3 < n4;
            const float4 p0 = ok0 ? v4[i0] : zero4, p1 = ok1 ? v4[i1] : zero4, p2 = ok2 ? v4[i2] : zero4, p3 = ok3 ? v4[i3] : zero4;  // four 16-byte loads in flight per lane
            visit4(p0, ok0); visit4(p1, ok1); visit4(p2, ok2); visit4(p3, ok3);
        }
        const int t = (n4 << 2) + tid;
        if (wave == 0) {  // the row's last n % 4 elements
            const bool ok = t < n;
            const float x = ok ? v[t] : 0.f;
            const unsigned u = __float_as_uint(x);
            below_a(x, u, ok);
            member(u, ok && u >= a && u <= bmax);
        }
    }
    // totals
    unsigned t0 = nz, t1 = nbelow;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { t0 += __shfl_xor(t0, off, 64); t1 += __shfl_xor(t1, off, 64); }
    if (lane == 0) { wsum[wave] = t0; wcount[wave] = wcnt; h256[wave] = t1; }
    __syncthreads();
    if (tid == 0) {
        unsigned z = 0, c = 0, mem = 0, over = 0;
        for (int w = 0; w < kWaves; ++w) { z += wsum[w]; c += h256[w]; mem += wcount[w]; over |= wcount[w] > (unsigned)kTrimSegCap ? 1u : 0u; }
        s_tot[0] = z; s_tot[1] = c; s_tot[2] = mem; s_tot[3] = over;
    }
    __syncthreads();
    const unsigned zeros = s_tot[0], below = s_tot[0] + s_tot[1], members = s_tot[2], over = s_tot[3];
    const unsigned mycnt = wcount[wave] < (unsigned)kTrimSegCap ? wcount[wave] : (unsigned)kTrimSegCap;
    __syncthreads();
    if ((unsigned)k <= zeros) {  // the k smallest terms are all zero
        if (tid == 0) { out_ub[row] = 0.0f; out_lb[row] = 0.0f; if (stat) atomicAdd(&stat[0], 1ull); }
        return;
    }
    // (2) the exact check of the bracket
    if (over || (unsigned)k <= below || (unsigned)k - below > members) {
        if (tid == 0 && stat) { atomicAdd(&stat[0], 1ull); atomicAdd(&stat[1], 1ull); }
        trim_row_two_pass<kTrimFallbackCap>(v, n, k, rt, buf, buf + kTrimBins, wsum, s_pick, &s_count, red, out_ub, out_lb, row);
        return;
    }
    if (tid == 0 && stat) { atomicAdd(&stat[0], 1ull); atomicAdd(&stat[2], (unsigned long long)members); }
    // (3) the need-th smallest member: radix select on w = u - a, 8 bits per round
    unsigned need = (unsigned)k - below;  // 1 <= need <= members
    const unsigned width = bmax - a;      // w in [0, width]
    int nbits = 32 - __clz(width | 1u);
    int shift = ((nbits + 7) / 8) * 8 - 8;
    unsigned prefix = 0;                  // the bits of the answer above shift + 8
    for (; shift >= 0; shift -= 8) {
        if (tid < 256) h256[tid] = 0;
        __syncthreads();
        for (unsigned j = lane; j < mycnt; j += 64) {
            const unsigned w = seg[j] - a;
            if (shift + 8 >= 32 || (w >> (shift + 8)) == prefix) atomicAdd(&h256[(w >> shift) & 255u], 1u);
        }
        __syncthreads();
        unsigned mine = 0, incl = 0;
        if (tid < 256) {  // waves 0..3: which digit holds rank `need`
            mine = h256[tid];
            incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned u = __shfl_up(incl, off, 64);
                if (lane >= off) incl += u;
            }
            if (lane == 63) wsum[wave] = incl;
        }
        __syncthreads();
        if (tid < 256) {
            unsigned base = 0;
            for (int w = 0; w < wave; ++w) base += wsum[w];
            const unsigned excl = base + incl - mine;
            if (excl < need && need <= excl + mine) { s_pick[0] = (unsigned)tid; s_pick[1] = excl; }  // exactly one thread
        }
        __syncthreads();
        prefix = (prefix << 8) | s_pick[0];
        need -= s_pick[1];
        __syncthreads();
    }
    const unsigned wk = prefix;  // w of the need-th smallest member; `need` is now its rank among its copies: that many copies count
    for (unsigned j = lane; j < mycnt; j += 64) {
        const unsigned u = seg[j];
        if (u - a < wk) {
            const float x = __uint_as_float(u);
            const float l = x - rt;
            acc[0] += (double)(x * x);
            acc[1] += (double)(l > 0.0f ? l * l : 0.0f);
        }
    }
    const double r = block_sum<2, kWaves>(acc, red);
    if (tid < 2) {
        const float x = __uint_as_float(a + wk);
        const float l = x - rt;
        const double extra = (double)need * (double)(tid == 0 ? x * x : (l > 0.0f ? l * l : 0.0f));
        (tid == 0 ? out_ub : out_lb)[row] = (float)(r + extra);
    }
}

// Sum of the k smallest of n non-negative floats by ONE 256-thread block: a 3-level radix select on the bit pattern (11 + 11 + 10
// bits, LDS histograms) finds the k-th smallest value v_k and how many copies of it are needed, then
// sum = sum_{v < v_k} v + need * v_k in fp64, fixed order.  Used for single short rows (trimmed SSE and the ICP inlier cut of
// clouds below 32768 points) and as the A/B of the device-wide selection below (FGOICP_SELECT_WIDE=0).
__global__ __launch_bounds__(kBlock) void trim_select_kernel(const float* __restrict__ vals, size_t row_stride, int ncols, int elem_stride, int n, int k,
                                                             float* __restrict__ out0, float* __restrict__ out1, uint32_t* __restrict__ sel_info) {
    __shared__ unsigned hist[2048];
    __shared__ unsigned scan[kBlock];
    __shared__ unsigned s_bin, s_below;
    __shared__ double red[8];
    const int tid = threadIdx.x;
    const int row = blockIdx.x, col = blockIdx.y;
    const float* v = vals + (size_t)row * row_stride + col;
    unsigned prefix = 0, prevmask = 0;
    unsigned need = (unsigned)k;  // rank (1-based) of the wanted element inside the current candidate set
    const int shifts[3] = {21, 10, 0}, nbits[3] = {11, 11, 10};
    for (int lvl = 0; lvl < 3; ++lvl) {
        const int shift = shifts[lvl], nb = 1 << nbits[lvl];
        for (int b = tid; b < 2048; b += kBlock) hist[b] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += kBlock) {
            const unsigned u = __float_as_uint(v[(size_t)i * elem_stride]);
            if ((u & prevmask) == prefix) atomicAdd(&hist[(u >> shift) & (nb - 1)], 1u);
        }
        __syncthreads();
        const int per = nb / kBlock;  // 8 or 4 consecutive bins per thread
        unsigned mine = 0;
        for (int b = 0; b < per; ++b) mine += hist[tid * per + b];
        scan[tid] = mine;
        __syncthreads();
        for (int off = 1; off < kBlock; off <<= 1) {  // inclusive Hillis-Steele scan
            const unsigned add = tid >= off ? scan[tid - off] : 0u;
            __syncthreads();
            scan[tid] += add;
            __syncthreads();
        }
        const unsigned excl = scan[tid] - mine;
        if (excl < need && need <= excl + mine) {  // exactly one thread
            unsigned below = excl;
            int b = 0;
            for (; b < per; ++b) {
                const unsigned c = hist[tid * per + b];
                if (need <= below + c) break;
                below += c;
            }
            s_bin = (unsigned)(tid * per + b);
            s_below = below;
        }
        __syncthreads();
        prefix |= s_bin << shift;
        prevmask |= (unsigned)(nb - 1) << shift;
        need -= s_below;
        __syncthreads();
    }
    // prefix = bits of v_k, need = copies of v_k among the k smallest
    double acc[1] = {0.0};
    for (int i = tid; i < n; i += kBlock) {
        const float x = v[(size_t)i * elem_stride];
        if (__float_as_uint(x) < prefix) acc[0] += (double)x;
    }
    const double ssum = block_sum<1>(acc, red);
    if (tid == 0) {
        const float r = (float)(ssum + (double)need * (double)__uint_as_float(prefix));
        if (col == 0 && out0) out0[row] = r;
        if (col == 1 && out1) out1[row] = r;
        if (sel_info) { sel_info[2 * (row * ncols + col)] = prefix; sel_info[2 * (row * ncols + col) + 1] = need; }
    }
}

// The same selection for ONE long row (trimmed SSE and the ICP inlier cut: n = ns values, a single (row, column)), spread over
// the whole device: per level a histogram kernel (LDS histograms merged with global atomics) and a one-block pick, then ordered
// block partial sums.  A single block needs 3-4 ms for 1M values; this takes nine small launches.
// scratch layout (uint32): [0, 6144) three 2048-bin histograms, [6144] prefix, [6145] prevmask, [6146] need; doubles from byte 32768.
constexpr int kSelWideBlocks = 512;
__device__ __forceinline__ double* sel_partials(uint32_t* scratch) { return reinterpret_cast<double*>(scratch + 8192); }

__global__ __launch_bounds__(kBlock) void select_wide_init_kernel(uint32_t* __restrict__ scratch, int k) {
    for (int i = threadIdx.x; i < 6144; i += kBlock) scratch[i] = 0u;
    if (threadIdx.x == 0) { scratch[6144] = 0u; scratch[6145] = 0u; scratch[6146] = (uint32_t)k; }
}
__global__ __launch_bounds__(kBlock) void select_wide_hist_kernel(const float* __restrict__ v, int n, int lvl, uint32_t* __restrict__ scratch) {
    __shared__ unsigned hist[2048];
    const int shift = lvl == 0 ? 21 : lvl == 1 ? 10 : 0, nb = lvl == 2 ? 1024 : 2048;
    const unsigned prefix = scratch[6144], prevmask = scratch[6145];
    for (int b = threadIdx.x; b < 2048; b += kBlock) hist[b] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * kBlock) {
        const unsigned u = __float_as_uint(v[i]);
        if ((u & prevmask) == prefix) atomicAdd(&hist[(u >> shift) & (nb - 1)], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += kBlock)
        if (hist[b]) atomicAdd(&scratch[lvl * 2048 + b], hist[b]);
}
__global__ __launch_bounds__(kBlock) void select_wide_pick_kernel(int lvl, uint32_t* __restrict__ scratch) {
    __shared__ unsigned scan[kBlock];
    const int tid = threadIdx.x;
    const int shift = lvl == 0 ? 21 : lvl == 1 ? 10 : 0, nb = lvl == 2 ? 1024 : 2048;
    const unsigned* hist = scratch + lvl * 2048;
    const unsigned need = scratch[6146];
    const int per = nb / kBlock;
    unsigned mine = 0;
    for (int b = 0; b < per; ++b) mine += hist[tid * per + b];
    scan[tid] = mine;
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {
        const unsigned add = tid >= off ? scan[tid - off] : 0u;
        __syncthreads();
        scan[tid] += add;
        __syncthreads();
    }
    const unsigned excl = scan[tid] - mine;
    if (excl < need && need <= excl + mine) {  // exactly one thread
        unsigned below = excl;
        int b = 0;
        for (; b < per; ++b) {
            const unsigned c = hist[tid * per + b];
            if (need <= below + c) break;
            below += c;
        }
        scratch[6144] |= (unsigned)(tid * per + b) << shift;
        scratch[6145] |= (unsigned)(nb - 1) << shift;
        scratch[6146] = need - below;
    }
}
__global__ __launch_bounds__(kBlock) void select_wide_sum_kernel(const float* __restrict__ v, int n, uint32_t* __restrict__ scratch) {
    __shared__ double red[8];
    const unsigned vk = scratch[6144];
    const size_t per = ((size_t)n + gridDim.x - 1) / gridDim.x;  // contiguous slice per block: the order of the sum is fixed
    const size_t a = per * blockIdx.x, b = a + per < (size_t)n ? a + per : (size_t)n;
    double acc[1] = {0.0};
    for (size_t i = a + threadIdx.x; i < b; i += kBlock) {
        const float x = v[i];
        if (__float_as_uint(x) < vk) acc[0] += (double)x;
    }
    const double r = block_sum<1>(acc, red);
    if (threadIdx.x == 0) sel_partials(scratch)[blockIdx.x] = r;
}
__global__ __launch_bounds__(64) void select_wide_final_kernel(uint32_t* __restrict__ scratch, int nblocks, float* __restrict__ out, uint32_t* __restrict__ sel_info) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 64) s += sel_partials(scratch)[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) {
        const unsigned vk = scratch[6144], need = scratch[6146];
        if (out) out[0] = (float)(s + (double)need * (double)__uint_as_float(vk));
        if (sel_info) { sel_info[0] = vk; sel_info[1] = need; }
    }
}

// ICP with trimming: squared distance of every working point to its correspondence ...
__global__ __launch_bounds__(kBlock) void icp_corr_d2_kernel(const float4* __restrict__ work, const float4* __restrict__ tgt, const uint32_t* __restrict__ idx,
                                                             int n, int nt, float* __restrict__ d2, uint32_t* __restrict__ equal_count) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i == 0) *equal_count = 0u;  // counted by icp_inlier_mask_kernel, two launches later on the same stream
    if (i >= n) return;
    const float4 a = work[i];
    const uint32_t j = idx[i];
    const float4 c = tgt[min(j, (uint32_t)(nt - 1))];
    d2[i] = j < (uint32_t)nt ? dist_sq(a.x, a.y, a.z, c.x, c.y, c.z) : 3.0e38f;  // no correspondence: provably beyond the cut (nn_prep_kernel)
}
// ... and the inlier mask: d2 < v_k, plus `need` of the points with d2 == v_k.  When all copies of v_k are needed
// (always, unless distances tie exactly at the cut) the mask is complete here; otherwise icp_inlier_ties_kernel admits the
// lowest ORIGINAL (caller) indices.
__global__ __launch_bounds__(kBlock) void icp_inlier_mask_kernel(const float* __restrict__ d2, int n, const uint32_t* __restrict__ sel_info,
                                                                 unsigned char* __restrict__ use, uint32_t* __restrict__ equal_count) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t u = __float_as_uint(d2[i]), vk = sel_info[0];
    use[i] = u < vk ? 1 : (u == vk ? 2 : 0);  // 2 = at the cut, resolved below
    if (u == vk) atomicAdd(equal_count, 1u);
}
__global__ __launch_bounds__(1024) void icp_inlier_ties_kernel(int n, const uint32_t* __restrict__ sel_info, const uint32_t* __restrict__ equal_count,
                                                               const uint32_t* __restrict__ orig_of_slot, unsigned char* __restrict__ use) {
    const uint32_t need = sel_info[1];
    if (*equal_count == need) {  // every point at the cut is an inlier
        for (int i = blockIdx.x * 1024 + threadIdx.x; i < n; i += gridDim.x * 1024)
            if (use[i] == 2) use[i] = 1;
        return;
    }
    if (blockIdx.x != 0) return;
    // exact distance ties at the cut (rare): the `need` lowest CALLER indices among the tied points, by a radix select on the
    // index (11 + 11 + 10 bits) in one workgroup
    __shared__ unsigned hist[2048];
    __shared__ unsigned s_bin, s_below;
    const int tid = threadIdx.x;
    unsigned prefix = 0, mask = 0, want = need;
    for (int lvl = 0; lvl < 3; ++lvl) {
        const int shift = lvl == 0 ? 21 : lvl == 1 ? 10 : 0, nb = lvl == 2 ? 1024 : 2048;
        for (int q = tid; q < 2048; q += 1024) hist[q] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += 1024) {
            if (use[i] != 2) continue;
            const unsigned o = orig_of_slot[i];
            if ((o & mask) == prefix) atomicAdd(&hist[(o >> shift) & (nb - 1)], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned below = 0;
            int q = 0;
            for (; q < nb - 1; ++q) {
                if (want <= below + hist[q]) break;
                below += hist[q];
            }
            s_bin = (unsigned)q;
            s_below = below;
        }
        __syncthreads();
        prefix |= s_bin << shift;
        mask |= (unsigned)(nb - 1) << shift;
        want -= s_below;
        __syncthreads();
    }
    for (int i = tid; i < n; i += 1024)
        if (use[i] == 2) use[i] = orig_of_slot[i] <= prefix ? 1 : 0;
}

// One block (one wave) per subcube: fixed-order sum of its chunk partials, rounded once to fp32.
__global__ __launch_bounds__(64) void bounds_finalize_kernel(const double2* __restrict__ partials, int nchunk, int total,
                                                             float* __restrict__ out_lb, float* __restrict__ out_ub, TickCut cut) {
    const int s = blockIdx.x;
    if (s >= total) return;
    const double2* row = partials + (size_t)s * nchunk;
    double u = 0.0, l = 0.0;
    unsigned skipped = 0;  // work items the early exit did not evaluate (their upper-bound partial is -1; such a row reports its threshold below)
    for (int c = threadIdx.x; c < nchunk; c += 64) {
        const double2 v = row[c];
        skipped += v.x < 0.0 ? 1u : 0u;
        u += v.x;
        l += v.y;
    }
    u = wave_sum(u);
    l = wave_sum(l);
    if (cut.stat) {
        for (int off = 32; off > 0; off >>= 1) skipped += __shfl_down(skipped, off);
        if (threadIdx.x == 0 && skipped) atomicAdd(&cut.stat[s & (kCutStatSlots - 1)], (unsigned long long)skipped);
    }
    if (threadIdx.x == 0) {
        float ubf = (float)u, lbf = (float)l;
        if (cut.row_cut) {  // fgoicp_bounds_submit_cut: a row at or above its threshold T reports {T, T}, whether the bounds kernel cut it short or not
            const float T = cut.row_cut[s];
            if (lbf >= T) lbf = ubf = T;
        }
        out_ub[s] = ubf;
        out_lb[s] = lbf;
    }
    if (cut.acc && threadIdx.x == 2) cut.done[s] = 0u;
    if (cut.acc && threadIdx.x < 2) cut.acc[2 * (size_t)s + threadIdx.x] = 0.0;  // evaluations <= rows: the running sums are zero again for the slot's next window
}

// ---------------------------------------------------------------------------------------------
// buildLUTKernel — fgoicp/registration.cu:258-278.  Exact brute force over all targets, tiled
// through LDS (one broadcast ds_read_b128 feeds kNodes*7 VALU ops), written straight into the
// padded layout; border nodes recompute their clamped neighbour.
// ---------------------------------------------------------------------------------------------
constexpr int kLutNodes = 4;
constexpr int kTile = 1024;

__global__ __launch_bounds__(kBlock) void lut_build_kernel(const float4* __restrict__ tgt, int nt, LutGeom g, float* __restrict__ lut) {
    __shared__ float4 tile[kTile];
    const size_t total = (size_t)g.px * g.py * g.pz;
    const size_t base = (size_t)blockIdx.x * (kBlock * kLutNodes) + threadIdx.x;
    float cx[kLutNodes], cy[kLutNodes], cz[kLutNodes], m[kLutNodes];
#pragma unroll
    for (int k = 0; k < kLutNodes; ++k) {
        size_t n = base + (size_t)k * kBlock;
        if (n >= total) n = total - 1;
        const int x = (int)(n % g.px);
        const size_t r = n / g.px;
        const int y = (int)(r % g.py);
        const int z = (int)(r / g.py);
        cx[k] = (float)min(max(x - 1, 0), g.dx - 1) * g.resolution;  // :265
        cy[k] = (float)min(max(y - 1, 0), g.dy - 1) * g.resolution;
        cz[k] = (float)min(max(z - 1, 0), g.dz - 1) * g.resolution;
        m[k] = 3.402823466e+38f;  // FLT_MAX, :266
    }
    for (int t0 = 0; t0 < nt; t0 += kTile) {
        const int cnt = min(kTile, nt - t0);
        for (int j = threadIdx.x; j < cnt; j += kBlock) tile[j] = tgt[t0 + j];
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            const float4 p = tile[j];
#pragma unroll
            for (int k = 0; k < kLutNodes; ++k) {
                const float d = dist_sq(cx[k], cy[k], cz[k], p.x, p.y, p.z);
                m[k] = m[k] < d ? m[k] : d;  // :272
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < kLutNodes; ++k) {
        const size_t n = base + (size_t)k * kBlock;
        if (n < total) lut[n] = m[k];
    }
}

__global__ __launch_bounds__(kBlock) void lut_unpad_kernel(const float* __restrict__ lut, LutGeom g, float* __restrict__ out) {
    const size_t total = (size_t)g.dx * g.dy * g.dz;
    for (size_t n = (size_t)blockIdx.x * kBlock + threadIdx.x; n < total; n += (size_t)gridDim.x * kBlock) {
        const int x = (int)(n % g.dx);
        const size_t r = n / g.dx;
        const int y = (int)(r % g.dy);
        const int z = (int)(r / g.dy);
        out[n] = lut[((size_t)(z + 1) * g.py + (y + 1)) * g.px + (x + 1)];
    }
}

__global__ __launch_bounds__(kBlock) void lut_nodes_kernel(const float* __restrict__ lut, LutGeom g, const int* __restrict__ xyz, size_t n, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const int x = min(max(xyz[3 * i], 0), g.dx - 1), y = min(max(xyz[3 * i + 1], 0), g.dy - 1), z = min(max(xyz[3 * i + 2], 0), g.dz - 1);
        out[i] = lut[((size_t)(z + 1) * g.py + (y + 1)) * g.px + (x + 1)];
    }
}

__global__ __launch_bounds__(kBlock) void lut_search_kernel(const float* __restrict__ lut, LutGeom g, const float* __restrict__ q,
                                                            size_t n, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
        out[i] = lut_search(lut, g, q[3 * i], q[3 * i + 1], q[3 * i + 2]);
}

// ---------------------------------------------------------------------------------------------
// Exact nearest neighbour, brute force — kernComputeClosestError / brute_force_find_nearest_neighbor
// (fgoicp/registration.cu:14-25, :162-174) and the distance half of kernFindNearestNeighbor
// (fgoicp/icp3d.cu:11-28).  grid = (query tiles, target slices); slices merge with an integer
// atomicMin on the (non-negative) float bit pattern, which is order-independent and exact.
// ---------------------------------------------------------------------------------------------
constexpr int kNnQ = 4;

__device__ __forceinline__ void load_queries(const float4* __restrict__ pts, int n, const Rt& rt, int apply, float (&qx)[kNnQ],
                                             float (&qy)[kNnQ], float (&qz)[kNnQ], int (&qi)[kNnQ]) {
#pragma unroll
    for (int k = 0; k < kNnQ; ++k) {
        const int i = (blockIdx.x * kNnQ + k) * kBlock + threadIdx.x;
        qi[k] = i;
        const float4 p = pts[i < n ? i : n - 1];
        if (apply) {
            float rx, ry, rz;
            rotate(rt.R, p.x, p.y, p.z, rx, ry, rz);
            qx[k] = rx + rt.t[0];
            qy[k] = ry + rt.t[1];
            qz[k] = rz + rt.t[2];
        } else {
            qx[k] = p.x;
            qy[k] = p.y;
            qz[k] = p.z;
        }
    }
}

__global__ __launch_bounds__(kBlock) void nn_min_kernel(const float4* __restrict__ pts, int n, const float4* __restrict__ tgt, int nt,
                                                        Rt rt, int apply, int slice_len, uint32_t* __restrict__ min_bits) {
    __shared__ float4 tile[kTile];
    float qx[kNnQ], qy[kNnQ], qz[kNnQ], best[kNnQ];
    int qi[kNnQ];
    load_queries(pts, n, rt, apply, qx, qy, qz, qi);
#pragma unroll
    for (int k = 0; k < kNnQ; ++k) best[k] = kInf;  // M_INF, registration.cu:164
    const int s0 = blockIdx.y * slice_len;
    const int s1 = min(nt, s0 + slice_len);
    for (int t0 = s0; t0 < s1; t0 += kTile) {
        const int cnt = min(kTile, s1 - t0);
        for (int j = threadIdx.x; j < cnt; j += kBlock) tile[j] = tgt[t0 + j];
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            const float4 p = tile[j];
#pragma unroll
            for (int k = 0; k < kNnQ; ++k) {
                const float d = dist_sq(qx[k], qy[k], qz[k], p.x, p.y, p.z);
                best[k] = d < best[k] ? d : best[k];  // strict '<', :168
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < kNnQ; ++k)
        if (qi[k] < n) atomicMin(&min_bits[qi[k]], __float_as_uint(best[k]));
}

// glm::distance takes the square root before comparing (icp3d.cu:20), so every target whose
// squared distance rounds to the same fp32 sqrt as the minimum ties, and the lowest index wins
// (strict '>' at icp3d.cu:21).  thr = largest float whose correctly rounded sqrt equals
// sqrt(min); the square root is monotone, so the tie set is exactly {j : d2_j <= thr}.
__global__ __launch_bounds__(kBlock) void nn_tie_threshold_kernel(const uint32_t* __restrict__ min_bits, int n,
                                                                  uint32_t* __restrict__ thr_bits) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint32_t b = min_bits[i];
    const float s = sqrtf(__uint_as_float(b));
    for (int it = 0; it < 8; ++it) {
        const uint32_t nb = b + 1;
        if (sqrtf(__uint_as_float(nb)) == s) b = nb; else break;
    }
    thr_bits[i] = b;
}

__global__ __launch_bounds__(kBlock) void nn_first_index_kernel(const float4* __restrict__ pts, int n, const float4* __restrict__ tgt,
                                                                int nt, int slice_len, const uint32_t* __restrict__ thr_bits,
                                                                uint32_t* __restrict__ first_idx) {
    __shared__ float4 tile[kTile];
    float qx[kNnQ], qy[kNnQ], qz[kNnQ], thr[kNnQ];
    int qi[kNnQ];
    uint32_t idx[kNnQ];
    Rt dummy{};
    load_queries(pts, n, dummy, 0, qx, qy, qz, qi);
#pragma unroll
    for (int k = 0; k < kNnQ; ++k) {
        thr[k] = __uint_as_float(thr_bits[qi[k] < n ? qi[k] : n - 1]);
        idx[k] = 0x7fffffffu;
    }
    const int s0 = blockIdx.y * slice_len;
    const int s1 = min(nt, s0 + slice_len);
    for (int t0 = s0; t0 < s1; t0 += kTile) {
        const int cnt = min(kTile, s1 - t0);
        for (int j = threadIdx.x; j < cnt; j += kBlock) tile[j] = tgt[t0 + j];
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            const float4 p = tile[j];
#pragma unroll
            for (int k = 0; k < kNnQ; ++k) {
                const float d = dist_sq(qx[k], qy[k], qz[k], p.x, p.y, p.z);
                const uint32_t cand = d <= thr[k] ? (uint32_t)(t0 + j) : 0x7fffffffu;
                idx[k] = min(idx[k], cand);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < kNnQ; ++k)
        if (qi[k] < n && idx[k] != 0x7fffffffu) atomicMin(&first_idx[qi[k]], idx[k]);
}

// ---------------------------------------------------------------------------------------------
// Exact nearest neighbour by a three-level box scan over the Morton-sorted targets (bvh.hpp):
// "leaves" of kBvhLeaf = 32 consecutive points, "super-leaves" of 32 leaves (1024 points) and "top boxes" of 32 super-leaves, each
// with its bounding box (three levels of the implicit tree).  Every query walks the SAME sequence
//     for each top box:  box test  ->  for each of its super-leaves:  box test  ->  for each of its leaves:  box test  ->  its 32 points
// so control flow is wave-uniform (a branch is taken when ANY lane needs it) and every box and point
// load has a wave-uniform address (one broadcast transaction, scalar-cacheable) — no dependent-load
// chains, no stack, no divergence beyond the exec mask.  The queries of a wave are Morton neighbours
// and each starts with a provable upper bound of its nearest distance from the distance LUT
// (lut_upper_bound_d2), so a wave touches a handful of leaves; the worst case (queries equidistant
// from the whole target) degrades to the cost of the brute-force sweep, never beyond.
// Candidate distances use the brute force's fp32 expression and min is order-independent: results
// are bit-identical to kernComputeClosestError / kernFindNearestNeighbor / buildLUTKernel.
// ---------------------------------------------------------------------------------------------
constexpr float kBoxShrink = 0.999999f;  // see bvh.hpp: covers fp32 rounding of box and point distances
constexpr float kMasked = 3.4028234e38f; // the "distance" a lane sees for a leaf that cannot matter to it: beyond every bound and every initial value
// Fan-out of the two box levels above the leaves: 2^kSuperShift leaves per super-leaf, as many super-leaves per top box.  A wave walks its
// candidate super-leaves one after another, each a DEPENDENT load of the leaf boxes, and a scan lasts as long as its slowest wave
// (tools/scan_stats.sh: bunny shape, far from convergence, up to 25 super-leaf steps and 16 leaf scans in one wave).  64 instead of 32 uses
// every lane of the box tests and halves the chain.
constexpr int kSuperShift = 6;

__device__ __forceinline__ float box_d2(const float4 lo, const float4 hi, float qx, float qy, float qz) {
    const float dx = fmaxf(fmaxf(lo.x - qx, qx - hi.x), 0.0f);
    const float dy = fmaxf(fmaxf(lo.y - qy, qy - hi.y), 0.0f);
    const float dz = fmaxf(fmaxf(lo.z - qz, qz - hi.z), 0.0f);
    return fma_(dz, dz, fma_(dy, dy, dx * dx));
}

__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float bcast(float v, int lane) {  // lane is wave-uniform -> v_readlane_b32
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// squared distance between a box and the wave's query region [wl, wh] — a lower bound of box_d2 for every lane
__device__ __forceinline__ float boxbox_d2(const float4 lo, const float4 hi, const float (&wl)[3], const float (&wh)[3]) {
    const float dx = fmaxf(fmaxf(lo.x - wh[0], wl[0] - hi.x), 0.0f);
    const float dy = fmaxf(fmaxf(lo.y - wh[1], wl[1] - hi.y), 0.0f);
    const float dz = fmaxf(fmaxf(lo.z - wh[2], wl[2] - hi.z), 0.0f);
    return fma_(dz, dz, fma_(dy, dy, dx * dx));
}

// Calls leaf(p) for every target point p whose leaf box is not provably farther than bound() (which
// may shrink as points are consumed).  `active` = false lanes only tag along.  Wave-cooperative:
//   1. lanes test 64 super-leaf boxes at a time against the wave's query region inflated by the
//      largest bound in the wave (one coalesced load per 64 boxes) -> ballot of candidate super-leaves;
//   2. per candidate, lanes 0..31 test its 32 leaf boxes the same way -> ballot of candidate leaves;
//   3. per candidate leaf, its box is broadcast from the lane that loaded it (readlane), every query
//      tests it against its OWN bound, and if any query needs it lanes 0..31 load its 32 points once
//      and broadcast them one by one.
// When `nparts` waves share the same 64 queries (small clouds: more waves in flight), every wave runs
// steps 1-2 identically (the wave radius r2 is NOT tightened then, so all see the same candidate list)
// and takes every nparts-th candidate leaf; the caller min-combines their results.
// Steps 1-2 are supersets of what each query needs (box-to-region distance <= box-to-query distance,
// wave radius >= own bound), step 3 applies the exact per-query rule, so no needed point is skipped.
#ifdef FGOICP_SCAN_STATS   // development builds only (tools/scan_stats.sh): what a scan's waves spend their steps on
__device__ unsigned long long g_scan_stats[8];  // walks, top candidates, super candidates, leaf candidates (dealt to this part), leaves scanned
__device__ unsigned long long g_scan_times[4096 * 4];  // per block of the last index-mode scan: s_memtime at entry, after the seeds, after the first walk, at exit
#define SCAN_TIME(slot) do { if (WANT_INDEX && threadIdx.x == 0 && blockIdx.x < 4096) g_scan_times[blockIdx.x * 4 + (slot)] = __builtin_readcyclecounter(); } while (0)
#if FGOICP_SCAN_STATS == 2   // stamps only: the counters' atomics sit inside the loops and would stretch what is being timed
#define SCAN_STAT(i, n) do { } while (0)
#else
#define SCAN_STAT(i, n) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_scan_stats[i], (unsigned long long)(n)); } while (0)
#endif
#define SCAN_STAT_DECL unsigned stat_supers = 0, stat_scanned = 0
#define SCAN_STAT_END do { if ((threadIdx.x & 63) == 0) { atomicMax(&g_scan_stats[5], (unsigned long long)stat_scanned); atomicMax(&g_scan_stats[6], (unsigned long long)stat_supers); \
                            if (stat_scanned > 16) atomicAdd(&g_scan_stats[7], 1ull); } } while (0)
#define SCAN_STAT_LOCAL(v) (++(v))
#else
#define SCAN_TIME(slot) do { } while (0)
#define SCAN_STAT(i, n) do { } while (0)
#define SCAN_STAT_DECL do { } while (0)
#define SCAN_STAT_END do { } while (0)
#define SCAN_STAT_LOCAL(v) do { } while (0)
#endif
// The walk itself, independent of what a lane asks of a leaf: `wl` / `wh` = the wave's query region, radius() = the wave-uniform squared
// pruning radius (re-evaluated after every leaf when one wave holds the queries alone), test(lo, hi) = the lane's flags for a leaf box
// (0: the leaf cannot matter to this lane), leaf(point, flags) = what to do with each of the leaf's points.
// refine(flags, n.x, n.y, n.z, a, b) = the lane's flags again after the leaf's SLAB test (bvh.hpp: a <= n.p <= b for every point p of the leaf) —
// only called for a leaf whose box some lane could not rule out, with the slab read through the scalar unit next to the leaf's points.
// slab_d2: slab.hpp (shared with the host so that its rounding allowance can be tested against exact arithmetic)
template <class RadiusFn, class TestFn, class LeafFn, class RefineFn>
__device__ __forceinline__ void box_walk(const BvhView t, const float (&wl)[3], const float (&wh)[3], int part, int nparts, RadiusFn radius, TestFn test, LeafFn leaf, RefineFn refine,
                                         int* claim_ctr /* LDS word, zero on entry, when nparts > 1 (nullptr: round-robin) */,
                                         unsigned long long* flat_mask = nullptr /* LDS, 32 words: every wave of the block holds the SAME queries (nn_scan kernels) */) {
    const int lane = threadIdx.x & 63;
    // The candidate leaves of these 64 queries are shared out among the `nparts` waves of the block DYNAMICALLY: every wave enumerates the
    // same candidates in the same order (cand) and works on the one it has claimed from an LDS counter, claiming the next when it is
    // done.  (Round 2 dealt them round-robin: a candidate the per-query test drops costs 20 instructions, one that is scanned 700, so
    // the slowest of 8 waves carried twice the mean — tools/scan_stats.sh — and the block waits for it at the combine.)
    // claim_ctr == nullptr: round-robin (candidate c goes to wave c mod nparts) — FGOICP_NN_CLAIM=0, the A/B.
    int cand = 0, next_claim = part;
    auto claim = [&]() {
        if (!claim_ctr) return next_claim + nparts;
        int v = 0;
        if (lane == 0) v = atomicAdd(claim_ctr, 1);
        return __builtin_amdgcn_readfirstlane(v);
    };
    if (nparts > 1 && claim_ctr) next_claim = claim();
    const float big = 3.0e38f;
    float r2 = radius();
    SCAN_STAT(0, 1);
    SCAN_STAT_DECL;
    // The leaf step shared by both forms of the walk: claim, per-query test against the leaf's box, then its 32 points through the
    // scalar unit (a wave-uniform address in the constant address space is what the backend selects s_load for; `leaf` takes the lane's
    // flags and stays branch-free: a lane the leaf cannot matter to sees a distance beyond every bound).
    auto visit = [&](int leaf_id, const float4 lo, const float4 hi) {
        if (nparts > 1) {
            const bool mine = cand == next_claim;
            ++cand;
            if (!mine) return;
        }
        int pl = test(lo, hi);
        SCAN_STAT(3, 1);
        typedef float v4f_c __attribute__((ext_vector_type(4)));
        if (t.slab && __any(pl != 0)) {  // the box could not rule the leaf out for some lane: its slab may (a far query's ball cuts many boxes, few slabs)
            const __attribute__((address_space(4))) v4f_c* sp = (const __attribute__((address_space(4))) v4f_c*)(t.slab + 2 * (size_t)__builtin_amdgcn_readfirstlane(leaf_id));
            const v4f_c s0 = sp[0], s1 = sp[1];
            pl = refine(pl, s0.x, s0.y, s0.z, s0.w, s1.x);
        }
        if (__any(pl != 0)) {
            SCAN_STAT(4, 1);
            SCAN_STAT_LOCAL(stat_scanned);
            const __attribute__((address_space(4))) v4f_c* lp = (const __attribute__((address_space(4))) v4f_c*)(t.pts + (size_t)__builtin_amdgcn_readfirstlane(leaf_id) * kBvhLeaf);
#pragma unroll 4   // 4 points in flight keep the index-mode kernel at 61 VGPRs = 8 waves per SIMD (16: 91 VGPRs, 5 waves — the scan
            // lives on resident waves hiding each other's dependent loads; measured slower)
            for (int k = 0; k < kBvhLeaf; ++k) {
                const v4f_c c = lp[k];
                leaf(make_float4(c.x, c.y, c.z, c.w), pl);
            }
            if (nparts == 1) r2 = radius();  // the wave radius only shrinks
        }
        if (nparts > 1) next_claim = claim();
    };
    // FLAT form (FGOICP_NN_FLAT=1; built in round 3, measured SLOWER — 49-53 -> 53-58 us per ICP iteration at 40k points — and off by default;
    // targets of at most 2048 leaves = 65 536 points, block-cooperative scans): the box levels above the leaves cost a
    // wave one DEPENDENT load per candidate super-leaf (tools/scan_stats.sh: 6 on average, 15-25 in the slowest wave, at 40k points) — the
    // chain the scan's duration consists of.  Here the block tests EVERY leaf box once, 64 per step, the steps dealt to its waves and
    // independent of each other (1 258 leaves / 8 waves = 3 loads per wave, issued together), publishes the candidate masks in LDS,
    // and every wave then walks the same candidate list, fetching a claimed leaf's box through the scalar unit.
    if (flat_mask && t.depth <= 11) {
        const int nleaf = 1 << t.depth, steps = (nleaf + 63) >> 6;
        for (int j = part; j < steps; j += nparts) {
            const int ln = j * 64 + lane;
            bool c = false;
            if (ln < nleaf) {
                const int node = t.first_leaf + ln;
                c = !(boxbox_d2(t.box[2 * node], t.box[2 * node + 1], wl, wh) * kBoxShrink > r2);
            }
            const unsigned long long m = __ballot(c);
            if (lane == 0) flat_mask[j] = m;
        }
        __syncthreads();
        typedef float v4f_b __attribute__((ext_vector_type(4)));
        const __attribute__((address_space(4))) v4f_b* bx = (const __attribute__((address_space(4))) v4f_b*)t.box;
        for (int j = 0; j < steps; ++j) {
            const unsigned long long mv = flat_mask[j];
            unsigned long long m = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(mv >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(mv & 0xffffffffull));
            while (m) {
                const int leaf_id = j * 64 + __ffsll((long long)m) - 1;
                m &= m - 1;
                if (nparts > 1 && cand != next_claim) { ++cand; continue; }  // not this wave's: no box fetch
                const v4f_b lo4 = bx[2 * (size_t)(t.first_leaf + leaf_id)], hi4 = bx[2 * (size_t)(t.first_leaf + leaf_id) + 1];
                visit(leaf_id, make_float4(lo4.x, lo4.y, lo4.z, 0.f), make_float4(hi4.x, hi4.y, hi4.z, 0.f));
            }
        }
        SCAN_STAT_END;
        return;
    }
    const int sdepth = t.depth > kSuperShift ? t.depth - kSuperShift : 0;
    const int nsuper = 1 << sdepth;
    const int lps = 1 << (t.depth - sdepth);  // leaves per super-leaf (<= 64)
    const int first_super = nsuper - 1;
    // a third level above the super-leaves ("top boxes"): without it every wave tests every super-leaf box — 977 of them for a
    // million targets (at 32 per level), 16 dependent rounds before the first leaf
    const int tdepth = sdepth > kSuperShift ? sdepth - kSuperShift : 0;
    const int ntop = 1 << tdepth;
    const int spt = 1 << (sdepth - tdepth);   // super-leaves per top box (<= 64)
    const int first_top = ntop - 1;
    for (int tb = 0; tb < ntop; tb += 64) {
        bool tc = false;
        if (tb + lane < ntop) {
            const int tn = first_top + tb + lane;
            tc = !(boxbox_d2(t.box[2 * tn], t.box[2 * tn + 1], wl, wh) * kBoxShrink > r2);
        }
        unsigned long long tmask = __ballot(tc);
        SCAN_STAT(1, __popcll(tmask));
        while (tmask) {
            const int top = tb + __ffsll((long long)tmask) - 1;
            tmask &= tmask - 1;
            bool sc = false;
            if (lane < spt) {
                const int sn = first_super + top * spt + lane;
                sc = !(boxbox_d2(t.box[2 * sn], t.box[2 * sn + 1], wl, wh) * kBoxShrink > r2);
            }
            unsigned long long smask = __ballot(sc);
            SCAN_STAT(2, __popcll(smask));
            while (smask) {
                const int s = top * spt + __ffsll((long long)smask) - 1;
                smask &= smask - 1;
                SCAN_STAT_LOCAL(stat_supers);
                float4 llo = make_float4(big, big, big, 0.f), lhi = make_float4(-big, -big, -big, 0.f);
                bool lc = false;
                if (lane < lps) {
                    const int ln = t.first_leaf + s * lps + lane;
                    llo = t.box[2 * ln];
                    lhi = t.box[2 * ln + 1];
                    lc = !(boxbox_d2(llo, lhi, wl, wh) * kBoxShrink > r2);
                }
                unsigned long long lmask = __ballot(lc);
                while (lmask) {
                    const int l = __ffsll((long long)lmask) - 1;
                    lmask &= lmask - 1;
                    if (nparts > 1 && cand != next_claim) { ++cand; continue; }
                    const float4 lo = make_float4(bcast(llo.x, l), bcast(llo.y, l), bcast(llo.z, l), 0.f);
                    const float4 hi = make_float4(bcast(lhi.x, l), bcast(lhi.y, l), bcast(lhi.z, l), 0.f);
                    visit(s * lps + l, lo, hi);
                }
            }
        }
    }
    SCAN_STAT_END;
}

template <class LeafFn, class BoundFn>
__device__ __forceinline__ void box_scan(const BvhView t, float qx, float qy, float qz, bool active, int part, int nparts, LeafFn leaf, BoundFn bound,
                                         int* claim_ctr = nullptr /* LDS word, zero on entry, when nparts > 1 */, unsigned long long* flat_mask = nullptr) {
    const float big = 3.0e38f;
    const float wl[3] = {wave_min_f(active ? qx : big), wave_min_f(active ? qy : big), wave_min_f(active ? qz : big)};
    const float wh[3] = {wave_max_f(active ? qx : -big), wave_max_f(active ? qy : -big), wave_max_f(active ? qz : -big)};
    box_walk(t, wl, wh, part, nparts, [&]() { return wave_max_f(active ? bound() : 0.0f); },
             [&](const float4 lo, const float4 hi) { return (int)(active && !(box_d2(lo, hi, qx, qy, qz) * kBoxShrink > bound())); },
             [&](const float4 c, int pl) { leaf(c, pl != 0); },
             [&](int pl, float nx, float ny, float nz, float a, float b) { return (pl && slab_d2(nx, ny, nz, a, b, qx, qy, qz) * kBoxShrink > bound()) ? 0 : pl; }, claim_ctr, flat_mask);
}

// Minimum squared distance.  `ub` is any value >= the true minimum (or +huge): it only seeds the
// pruning bound; the result is the minimum over VISITED points, and the true nearest point is never
// pruned while the bound stays >= its distance.  If fp32 rounding made a seed a hair too small and
// some lane found nothing, the wave repeats the scan unseeded — the result is exact either way.
// `arg` (optional): receives the caller index (the leaf point's w) of a point that attains the minimum.
__device__ __forceinline__ float scan_min_d2(const BvhView t, float qx, float qy, float qz, float ub, float init, bool active, uint32_t* arg = nullptr) {
    float best = ub < init ? ub : init;
    float found = init;
    uint32_t who = 0x7fffffffu;
    box_scan(t, qx, qy, qz, active, 0, 1,
             [&](const float4 p, bool on) {
                 const float d = on ? dist_sq(qx, qy, qz, p.x, p.y, p.z) : kMasked;
                 if (arg) who = d < found ? __float_as_uint(p.w) : who;
                 found = d < found ? d : found;
                 best = d < best ? d : best;
             },
             [&]() { return best; });
    const bool redo = active && found > best;  // cannot happen with a valid seed
    if (__any(redo)) {
        if (redo) found = init;
        box_scan(t, qx, qy, qz, redo, 0, 1,
                 [&](const float4 p, bool on) {
                     const float d = on ? dist_sq(qx, qy, qz, p.x, p.y, p.z) : kMasked;
                     if (arg) who = d < found ? __float_as_uint(p.w) : who;
                     found = d < found ? d : found;
                 },
                 [&]() { return found; });
    }
    if (arg) *arg = who;
    return found;
}

// Upper bound on the nearest-target squared distance of q from the LUT: for ANY LUT node c,
// min_j |q - tgt_j| <= |q - c| + min_j |c - tgt_j| = |q - c| + sqrt(T[c])  (triangle inequality).  With the
// node nearest to q (clamped into the grid, so it also works outside the target's box) the slack is
// at most half a voxel diagonal.  Inflated by 1e-4 relative + 1e-6 absolute: T was computed from
// the shifted targets in fp32.
// Round 3 (g.idx, the "index LUT"): the node also knows WHICH target point is nearest to it, and the distance from q to that very
// point is a bound too — an exact one (it is one of the scan's candidates, same fp32 expression), and a much tighter one for a
// query far from the surface: the triangle bound is loose by up to a voxel diagonal whatever the distance, so the ball it admits
// cuts a cap of radius sqrt(2 d diag) out of the surface (43 leaves for d = 0.3 at resolution 0.005), while the nearest point of a
// node half a voxel away is almost always the query's own nearest point or its neighbour.
__device__ __forceinline__ float lut_upper_bound_d2(const float* __restrict__ lut, const LutGeom& g, float qx, float qy, float qz, const float4* __restrict__ tgt = nullptr,
                                                    int nt = 0) {
    const float sx = qx + g.off_x, sy = qy + g.off_y, sz = qz + g.off_z;
    const float fx = fminf(fmaxf(rintf(sx * g.scale), 0.0f), (float)(g.dx - 1));
    const float fy = fminf(fmaxf(rintf(sy * g.scale), 0.0f), (float)(g.dy - 1));
    const float fz = fminf(fmaxf(rintf(sz * g.scale), 0.0f), (float)(g.dz - 1));
    const size_t node = ((size_t)((int)fz + 1) * g.py + ((int)fy + 1)) * (size_t)g.px + ((int)fx + 1);
    const float T = lut[node];
    const float dx = sx - fx * g.resolution, dy = sy - fy * g.resolution, dz = sz - fz * g.resolution;
    const float u = sqrtf(T) + sqrtf(dx * dx + dy * dy + dz * dz);
    float ub = u * u * 1.0001f + 1e-6f;
    if (g.idx && tgt) {
        const uint32_t j = g.idx[node];
        if (j < (uint32_t)nt) {
            const float4 c = tgt[j];
            const float d = dist_sq(qx, qy, qz, c.x, c.y, c.z);
            ub = d < ub ? d : ub;
        }
    }
    return ub;
}

__device__ __forceinline__ float tie_threshold(float best) {  // see nn_tie_threshold_kernel
    uint32_t b = __float_as_uint(best);
    const float s = sqrtf(best);
    for (int it = 0; it < 8; ++it) {
        const uint32_t nb = b + 1;
        if (sqrtf(__uint_as_float(nb)) == s) b = nb; else break;
    }
    return __uint_as_float(b);
}

// EXTENSION (trimmed Go-ICP) — which queries can be left out of the exact search.  Only the k smallest nearest-neighbour
// distances enter a trimmed sum or the inlier set.  For every query the LUT gives a rigorous bracket of its nearest
// distance: ub (lut_upper_bound_d2, or the distance to the previous correspondence) and
//     lb = max( (sqrt(T[c]) - |q - c|)^2 ,  squared distance from q to the target's bounding box )   (deflated for rounding)
// for the LUT node c nearest to q (triangle inequality the other way round).  With U = the k-th smallest ub, at least k
// queries have a nearest distance <= U, so a query with lb > U is not among the k smallest and is not in a tie at the cut:
// it needs neither its exact distance nor a correspondence.  On a cloud with uniform far outliers those are exactly the
// queries whose searches are expensive (a large ball around the query cuts many leaves) and that widen the region their
// wave has to scan.  This kernel writes ub and lb; U comes from the selection kernels; nn_scan_kernel drops lb > U.
__global__ __launch_bounds__(kBlock) void nn_prep_kernel(const float4* __restrict__ pts, int n, const float* __restrict__ lut, LutGeom g, Rt rt, int apply,
                                                         const float4* __restrict__ tgt, int nt, const uint32_t* __restrict__ seed_idx, float4 box_lo, float4 box_hi,
                                                         float* __restrict__ ub_out, float* __restrict__ lb_out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    float qx = p.x, qy = p.y, qz = p.z;
    if (apply) {
        rotate(rt.R, p.x, p.y, p.z, qx, qy, qz);
        qx += rt.t[0]; qy += rt.t[1]; qz += rt.t[2];
    }
    float ub = lut_upper_bound_d2(lut, g, qx, qy, qz, tgt, nt);
    if (seed_idx) {
        const uint32_t j = seed_idx[i];
        if (j < (uint32_t)nt) {
            const float4 c = tgt[j];
            const float d = dist_sq(qx, qy, qz, c.x, c.y, c.z);
            ub = d < ub ? d : ub;
        }
    }
    const float sx = qx + g.off_x, sy = qy + g.off_y, sz = qz + g.off_z;
    const float fx = fminf(fmaxf(rintf(sx * g.scale), 0.0f), (float)(g.dx - 1));
    const float fy = fminf(fmaxf(rintf(sy * g.scale), 0.0f), (float)(g.dy - 1));
    const float fz = fminf(fmaxf(rintf(sz * g.scale), 0.0f), (float)(g.dz - 1));
    const float T = lut[((size_t)((int)fz + 1) * g.py + ((int)fy + 1)) * (size_t)g.px + ((int)fx + 1)];
    const float dx = sx - fx * g.resolution, dy = sy - fy * g.resolution, dz = sz - fz * g.resolution;
    const float l = sqrtf(T) - sqrtf(dx * dx + dy * dy + dz * dz);
    float lb = l > 0.0f ? l * l * 0.9999f - 1e-6f : 0.0f;
    const float bd = box_d2(box_lo, box_hi, qx, qy, qz) * 0.9999f - 1e-6f;  // every target point lies inside its bounding box
    lb = fmaxf(fmaxf(lb, bd), 0.0f);
    ub_out[i] = ub < kInf ? ub : kInf;
    lb_out[i] = lb;
}

//   want_index = 0: out[i] = bits(min_j |q_i - tgt_j|^2)                       (registration.cu:162-174)
//   want_index = 1: out[i] = lowest j inside the sqrt-tie set of the minimum      (icp3d.cu:11-28)
// One block = 64 queries x `nparts` waves (blockDim = 64 * nparts): each wave scans its share of the
// candidate leaves, the shares are min-combined through LDS.
constexpr int kMaxParts = 16;

template <int WANT_INDEX>
__global__ __launch_bounds__(64 * kMaxParts) void nn_scan_kernel(const float4* pts, int n, BvhView t, const float* __restrict__ lut,
                                                                 LutGeom g, Rt rt, int apply, const float4* __restrict__ tgt, int nt,
                                                                 const uint32_t* seed_idx, const float* __restrict__ skip_lb, const uint32_t* __restrict__ skip_u, uint32_t* out,
                                                                 float4* writeback, const float* __restrict__ rt_dev, const int* __restrict__ done,
                                                                 double* __restrict__ wsum, int dynamic_claim, int flat_walk) {
    if (done && *done) return;  // device-resident ICP loop: the run has ended, this pass was enqueued ahead of the decision
    if (rt_dev) {               // ... and the motion is the one the step kernel left in device memory (12 floats: R, t)
#pragma unroll
        for (int k = 0; k < 9; ++k) rt.R[k] = rt_dev[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) rt.t[k] = rt_dev[9 + k];
    }
    SCAN_TIME(0);
    __shared__ int claim_ctr[3];  // one per walk (first, redo, tie): zeroed here, used once each
    if (threadIdx.x < 3) claim_ctr[threadIdx.x] = 0;
    __shared__ unsigned long long leaf_mask[32];  // flat form of the walk (box_walk): the block's candidate leaves
    unsigned long long* const flat = flat_walk ? leaf_mask : nullptr;
    __shared__ uint32_t comb[kMaxParts][64];
    __shared__ uint32_t comb_i[WANT_INDEX ? kMaxParts : 1][64];
    __shared__ uint32_t comb_2[WANT_INDEX ? kMaxParts : 1][64];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6, nparts = blockDim.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    bool active = i < n;
    if (skip_lb && active && skip_lb[i] > __uint_as_float(skip_u[0])) {  // trimmed: provably beyond the k-th smallest distance (nn_prep_kernel)
        active = false;
        if (part == 0) out[i] = WANT_INDEX ? 0x7fffffffu : __float_as_uint(skip_lb[i]);  // any value above the cut / no correspondence
    }
    const float4 p = pts[i < n ? i : n - 1];
    float qx = p.x, qy = p.y, qz = p.z;
    if (apply) {
        rotate(rt.R, p.x, p.y, p.z, qx, qy, qz);
        qx += rt.t[0]; qy += rt.t[1]; qz += rt.t[2];
    }
    // pass 1: minimum.  `found` = min over the points this wave visited, `best` additionally seeded.
    float ub = lut_upper_bound_d2(lut, g, qx, qy, qz, tgt, nt);
    if (seed_idx) {  // ICP: the previous pass's correspondence — the distance to ANY target point bounds the minimum, and
        const uint32_t j = seed_idx[i < n ? i : n - 1];  // one ICP step later it usually still IS the minimum (may alias `out`:
        if (j < (uint32_t)nt) {                          // every lane reads its own slot here and writes it at the very end)
            const float4 c = tgt[j];
            const float d = dist_sq(qx, qy, qz, c.x, c.y, c.z);
            ub = d < ub ? d : ub;
        }
    }
    float best = ub < kInf ? ub : kInf, found = kInf;
    // index mode: the same walk also keeps the lowest index attaining the minimum (i1) and the smallest distance above it
    // (second).  The walk prunes against best * (1 + 1.5e-6) >= tie_threshold(best), so every point of the sqrt-tie set of the
    // final minimum is visited; if `second` lies outside that set — always, but for a genuine tie of two different squared
    // distances — i1 is the answer and the second walk below is skipped.
    float second = kInf;
    uint32_t i1 = 0x7fffffffu;
    SCAN_TIME(1);
    if (nparts > 1) __syncthreads();  // the claim counters are zero for every wave
    box_scan(t, qx, qy, qz, active, part, nparts,
             [&](const float4 c, bool on) {
                 const float d = on ? dist_sq(qx, qy, qz, c.x, c.y, c.z) : kMasked;
                 if (WANT_INDEX) {
                     // branch-free form of:  d < found: second = found, i1 = j;  d == found: i1 = min(i1, j);  else second = min(second, d)
                     // (`second` > `found` always, so min(second, max(found, d)) covers both outer cases)
                     const uint32_t j = __float_as_uint(c.w);
                     const bool lt = d < found, eq = d == found;
                     const float s2 = fminf(second, fmaxf(found, d));
                     second = eq ? second : s2;
                     const uint32_t cand = d <= found ? j : 0x7fffffffu;
                     i1 = lt ? j : min(i1, cand);
                 }
                 found = fminf(found, d);
                 best = fminf(best, d);
             },
             [&]() { return WANT_INDEX ? best * 1.0000015f : best; }, dynamic_claim ? &claim_ctr[0] : nullptr, flat);
    SCAN_TIME(2);
    if (nparts > 1) {
        comb[part][lane] = __float_as_uint(found);  // non-negative floats order like their bit patterns
        __syncthreads();
        uint32_t m = comb[0][lane];
        for (int k = 1; k < nparts; ++k) m = min(m, comb[k][lane]);
        if (WANT_INDEX) {  // the parts' (minimum, index, second) folded: index among the parts that hold the global minimum
            const float gm = __uint_as_float(m);
            comb_i[part][lane] = found == gm ? i1 : 0x7fffffffu;
            comb_2[part][lane] = __float_as_uint(found > gm ? found : second);
            __syncthreads();
            uint32_t mi = comb_i[0][lane], m2 = comb_2[0][lane];
            for (int k = 1; k < nparts; ++k) { mi = min(mi, comb_i[k][lane]); m2 = min(m2, comb_2[k][lane]); }
            i1 = mi;
            second = __uint_as_float(m2);
        }
        found = __uint_as_float(m);
        __syncthreads();
    }
    const bool redo = active && found > (ub < kInf ? ub : kInf);  // seed too small by rounding: cannot happen, stay exact
    if (__syncthreads_or(redo)) {
        float f2 = kInf;
        box_scan(t, qx, qy, qz, redo, part, nparts,
                 [&](const float4 c, bool on) {
                     const float d = on ? dist_sq(qx, qy, qz, c.x, c.y, c.z) : kMasked;
                     f2 = d < f2 ? d : f2;
                 },
                 [&]() { return f2; }, dynamic_claim ? &claim_ctr[1] : nullptr, flat);
        if (nparts > 1) {
            comb[part][lane] = __float_as_uint(f2);
            __syncthreads();
            uint32_t m = comb[0][lane];
            for (int k = 1; k < nparts; ++k) m = min(m, comb[k][lane]);
            f2 = __uint_as_float(m);
            __syncthreads();
        }
        if (redo) found = f2;
    }
    uint32_t result = __float_as_uint(found);
    if (WANT_INDEX) {
        const float thr = tie_threshold(found);
        // a second walk only for lanes whose tie set may hold a different squared distance (or whose first walk was redone)
        const bool again = active && (redo || !(second > thr));
        uint32_t idx = i1;
        if (__syncthreads_or(again)) {
            uint32_t idx2 = 0x7fffffffu;
            box_scan(t, qx, qy, qz, again, part, nparts,
                     [&](const float4 c, bool on) {
                         const float d = on ? dist_sq(qx, qy, qz, c.x, c.y, c.z) : kMasked;
                         idx2 = min(idx2, d <= thr ? __float_as_uint(c.w) : 0x7fffffffu);
                     },
                     [&]() { return thr; }, dynamic_claim ? &claim_ctr[2] : nullptr, flat);
            if (nparts > 1) {
                comb[part][lane] = idx2;
                __syncthreads();
                for (int k = 0; k < nparts; ++k) idx2 = min(idx2, comb[k][lane]);
            }
            if (again) idx = idx2;
        }
        result = idx;
    }
    if (active && part == 0) out[i] = result;
    // ICP: the queries are the working cloud moved by this iteration's (R_, t_) — kernRotateTranslateInplace (icp3d.cu:30-36, :100)
    // folded into the pass that needs its result first; the moved points go back to the cloud (`writeback` may be `pts`: every
    // wave of the block read its point before the barriers above, only wave 0 writes).
    if (writeback && part == 0 && i < n) writeback[i] = make_float4(qx, qy, qz, p.w);
    SCAN_TIME(3);
    // The reductions that follow a scan, started here (small clouds: one launch less on the ICP iteration's chain).  These 64 queries
    // are one wave of icp_sums_kernel / sum_f32_kernel when every thread of those kernels holds at most one point (n <= gridDim *
    // 256 there), so the shuffle tree below is THEIR first reduction level with the same operands: wsum[group] is the value they
    // would park in LDS, and whoever folds four consecutive groups in order, then the blocks, gets their bits (icp_cov_cen_kernel,
    // fold_wave_sums on the host).
    if (wsum && part == 0) {
        if (WANT_INDEX) {
            float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n) c = tgt[min(result, (uint32_t)(nt - 1))];
            const double v[6] = {i < n ? (double)qx : 0.0, i < n ? (double)qy : 0.0, i < n ? (double)qz : 0.0,
                                 i < n ? (double)c.x : 0.0, i < n ? (double)c.y : 0.0, i < n ? (double)c.z : 0.0};
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double r = wave_sum(v[k]);
                if (lane == 0) wsum[(size_t)blockIdx.x * 6 + k] = r;
            }
        } else {
            const double r = wave_sum(i < n ? (double)__uint_as_float(result) : 0.0);
            if (lane == 0) wsum[blockIdx.x] = r;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// ONE walk for the two scans of an ICP iteration (round 3).  Iteration k needs, once its (R_, t_) is known, the correspondences of the
// working cloud moved by (R_, t_) (kernFindNearestNeighbor for iteration k+1, icp3d.cu:146 after :100) and the exact SSE of the pristine
// source under the composed (R, t) (registration.cu:62-86 for :103).  Query i of the one is query i of the other up to the fp32 error
// the in-place moves have accumulated (~1e-6): the same region, the same candidate leaves.  Round 2 ran them as two kernels on two streams
// — 10 000 waves on 8 192 slots at 40k points, twice the box loads and leaf reads; here every lane carries BOTH queries through one
// walk (set A: index rule, set B: minimum), so the traversal is paid once.  Each set keeps its own exact rule (own seeds, own bounds,
// own skip list in trimmed mode, own combine), and the results are those of nn_scan_kernel<1> / <0> bit for bit (tests).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * kMaxParts) void nn_scan_dual_kernel(const float4* ptsA, Rt rtA, int applyA, const float4* __restrict__ ptsB, Rt rtB, int n, BvhView t,
                                                                      const float* __restrict__ lut, LutGeom g, const float4* __restrict__ tgt, int nt, const uint32_t* seed_idx,
                                                                      const float* __restrict__ skip_lbA, const uint32_t* __restrict__ skip_uA,
                                                                      const float* __restrict__ skip_lbB, const uint32_t* __restrict__ skip_uB, uint32_t* out_idx,
                                                                      uint32_t* __restrict__ out_min, float4* writeback, double* __restrict__ wsumA, double* __restrict__ wsumB,
                                                                      int dynamic_claim) {
    __shared__ int claim_ctr[3];
    if (threadIdx.x < 3) claim_ctr[threadIdx.x] = 0;
    __shared__ uint32_t comb[kMaxParts][64];     // set A: minimum
    __shared__ uint32_t comb_i[kMaxParts][64];   //        index
    __shared__ uint32_t comb_2[kMaxParts][64];   //        second distance
    __shared__ uint32_t comb_b[kMaxParts][64];   // set B: minimum
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6, nparts = blockDim.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    bool actA = i < n, actB = i < n;
    if (skip_lbA && actA && skip_lbA[i] > __uint_as_float(skip_uA[0])) {  // trimmed: provably beyond the k-th smallest distance (nn_prep_kernel)
        actA = false;
        if (part == 0) out_idx[i] = 0x7fffffffu;
    }
    if (skip_lbB && actB && skip_lbB[i] > __uint_as_float(skip_uB[0])) {
        actB = false;
        if (part == 0) out_min[i] = __float_as_uint(skip_lbB[i]);
    }
    const float4 pA = ptsA[i < n ? i : n - 1], pB = ptsB[i < n ? i : n - 1];
    float ax = pA.x, ay = pA.y, az = pA.z, bx, by, bz;
    if (applyA) {
        rotate(rtA.R, pA.x, pA.y, pA.z, ax, ay, az);
        ax += rtA.t[0]; ay += rtA.t[1]; az += rtA.t[2];
    }
    rotate(rtB.R, pB.x, pB.y, pB.z, bx, by, bz);
    bx += rtB.t[0]; by += rtB.t[1]; bz += rtB.t[2];
    float ubA = lut_upper_bound_d2(lut, g, ax, ay, az, tgt, nt), ubB = lut_upper_bound_d2(lut, g, bx, by, bz, tgt, nt);
    if (seed_idx) {
        const uint32_t j = seed_idx[i < n ? i : n - 1];
        if (j < (uint32_t)nt) {
            const float4 c = tgt[j];
            const float dA = dist_sq(ax, ay, az, c.x, c.y, c.z), dB = dist_sq(bx, by, bz, c.x, c.y, c.z);
            ubA = dA < ubA ? dA : ubA;
            ubB = dB < ubB ? dB : ubB;
        }
    }
    float bestA = ubA < kInf ? ubA : kInf, foundA = kInf, secondA = kInf;
    uint32_t i1 = 0x7fffffffu;
    float bestB = ubB < kInf ? ubB : kInf, foundB = kInf;
    const float big = 3.0e38f;
    const bool any_act = actA || actB;
    // the wave's region: both query sets (an inactive set contributes nothing)
    auto lo3 = [&](float a, float b) { return wave_min_f(fminf(actA ? a : big, actB ? b : big)); };
    auto hi3 = [&](float a, float b) { return wave_max_f(fmaxf(actA ? a : -big, actB ? b : -big)); };
    const float wl[3] = {lo3(ax, bx), lo3(ay, by), lo3(az, bz)};
    const float wh[3] = {hi3(ax, bx), hi3(ay, by), hi3(az, bz)};
    if (nparts > 1) __syncthreads();  // the claim counters are zero for every wave
    box_walk(t, wl, wh, part, nparts,
             [&]() { return wave_max_f(fmaxf(actA ? bestA * 1.0000015f : 0.0f, actB ? bestB : 0.0f)); },
             [&](const float4 lo, const float4 hi) {
                 const int fa = actA && !(box_d2(lo, hi, ax, ay, az) * kBoxShrink > bestA * 1.0000015f);
                 const int fb = actB && !(box_d2(lo, hi, bx, by, bz) * kBoxShrink > bestB);
                 return fa | (fb << 1);
             },
             [&](const float4 c, int fl) {
                 const float dA = (fl & 1) ? dist_sq(ax, ay, az, c.x, c.y, c.z) : kMasked;
                 const float dB = (fl & 2) ? dist_sq(bx, by, bz, c.x, c.y, c.z) : kMasked;
                 const uint32_t j = __float_as_uint(c.w);
                 const bool lt = dA < foundA, eq = dA == foundA;
                 const float s2 = fminf(secondA, fmaxf(foundA, dA));
                 secondA = eq ? secondA : s2;
                 const uint32_t cnd = dA <= foundA ? j : 0x7fffffffu;
                 i1 = lt ? j : min(i1, cnd);
                 foundA = fminf(foundA, dA);
                 bestA = fminf(bestA, dA);
                 foundB = fminf(foundB, dB);
                 bestB = fminf(bestB, dB);
             },
             [&](int fl, float nx, float ny, float nz, float a, float b) {
                 if ((fl & 1) && slab_d2(nx, ny, nz, a, b, ax, ay, az) * kBoxShrink > bestA * 1.0000015f) fl &= ~1;
                 if ((fl & 2) && slab_d2(nx, ny, nz, a, b, bx, by, bz) * kBoxShrink > bestB) fl &= ~2;
                 return fl;
             },
             dynamic_claim ? &claim_ctr[0] : nullptr);
    (void)any_act;
    if (nparts > 1) {
        comb[part][lane] = __float_as_uint(foundA);
        comb_b[part][lane] = __float_as_uint(foundB);
        __syncthreads();
        uint32_t m = comb[0][lane], mb = comb_b[0][lane];
        for (int k = 1; k < nparts; ++k) { m = min(m, comb[k][lane]); mb = min(mb, comb_b[k][lane]); }
        const float gm = __uint_as_float(m);
        comb_i[part][lane] = foundA == gm ? i1 : 0x7fffffffu;
        comb_2[part][lane] = __float_as_uint(foundA > gm ? foundA : secondA);
        __syncthreads();
        uint32_t mi = comb_i[0][lane], m2 = comb_2[0][lane];
        for (int k = 1; k < nparts; ++k) { mi = min(mi, comb_i[k][lane]); m2 = min(m2, comb_2[k][lane]); }
        i1 = mi;
        secondA = __uint_as_float(m2);
        foundA = gm;
        foundB = __uint_as_float(mb);
        __syncthreads();
    }
    // seed too small by rounding (cannot happen: stay exact): the lanes concerned walk again unseeded, each set on its own
    const bool redoA = actA && foundA > (ubA < kInf ? ubA : kInf), redoB = actB && foundB > (ubB < kInf ? ubB : kInf);
    if (__syncthreads_or(redoA || redoB)) {
        float fA = kInf, fB = kInf;
        const float rl[3] = {wave_min_f(fminf(redoA ? ax : big, redoB ? bx : big)), wave_min_f(fminf(redoA ? ay : big, redoB ? by : big)), wave_min_f(fminf(redoA ? az : big, redoB ? bz : big))};
        const float rh[3] = {wave_max_f(fmaxf(redoA ? ax : -big, redoB ? bx : -big)), wave_max_f(fmaxf(redoA ? ay : -big, redoB ? by : -big)), wave_max_f(fmaxf(redoA ? az : -big, redoB ? bz : -big))};
        box_walk(t, rl, rh, part, nparts, [&]() { return wave_max_f(fmaxf(redoA ? fA : 0.0f, redoB ? fB : 0.0f)); },
                 [&](const float4 lo, const float4 hi) {
                     const int fa = redoA && !(box_d2(lo, hi, ax, ay, az) * kBoxShrink > fA);
                     const int fb = redoB && !(box_d2(lo, hi, bx, by, bz) * kBoxShrink > fB);
                     return fa | (fb << 1);
                 },
                 [&](const float4 c, int fl) {
                     const float dA = (fl & 1) ? dist_sq(ax, ay, az, c.x, c.y, c.z) : kMasked;
                     const float dB = (fl & 2) ? dist_sq(bx, by, bz, c.x, c.y, c.z) : kMasked;
                     fA = fminf(fA, dA);
                     fB = fminf(fB, dB);
                 },
                 [&](int fl, float nx, float ny, float nz, float a, float b) {
                     if ((fl & 1) && slab_d2(nx, ny, nz, a, b, ax, ay, az) * kBoxShrink > fA) fl &= ~1;
                     if ((fl & 2) && slab_d2(nx, ny, nz, a, b, bx, by, bz) * kBoxShrink > fB) fl &= ~2;
                     return fl;
                 },
                 dynamic_claim ? &claim_ctr[1] : nullptr);
        if (nparts > 1) {
            comb[part][lane] = __float_as_uint(fA);
            comb_b[part][lane] = __float_as_uint(fB);
            __syncthreads();
            uint32_t m = comb[0][lane], mb = comb_b[0][lane];
            for (int k = 1; k < nparts; ++k) { m = min(m, comb[k][lane]); mb = min(mb, comb_b[k][lane]); }
            fA = __uint_as_float(m);
            fB = __uint_as_float(mb);
            __syncthreads();
        }
        if (redoA) foundA = fA;
        if (redoB) foundB = fB;
    }
    // set A: the lowest index inside the sqrt-tie set of the minimum (icp3d.cu:20-25), as nn_scan_kernel<1>
    const float thr = tie_threshold(foundA);
    const bool again = actA && (redoA || !(secondA > thr));
    uint32_t idx = i1;
    if (__syncthreads_or(again)) {
        uint32_t idx2 = 0x7fffffffu;
        box_scan(t, ax, ay, az, again, part, nparts,
                 [&](const float4 c, bool on) {
                     const float d = on ? dist_sq(ax, ay, az, c.x, c.y, c.z) : kMasked;
                     idx2 = min(idx2, d <= thr ? __float_as_uint(c.w) : 0x7fffffffu);
                 },
                 [&]() { return thr; }, dynamic_claim ? &claim_ctr[2] : nullptr);
        if (nparts > 1) {
            comb[part][lane] = idx2;
            __syncthreads();
            for (int k = 0; k < nparts; ++k) idx2 = min(idx2, comb[k][lane]);
        }
        if (again) idx = idx2;
    }
    if (part == 0) {
        if (actA) out_idx[i] = idx;
        if (actB) out_min[i] = __float_as_uint(foundB);
        if (writeback && i < n) writeback[i] = make_float4(ax, ay, az, pA.w);
    }
    // the first level of the reductions that follow (see nn_scan_kernel): wave sums of {moved point, correspondence} and of the minima
    if (part == 0 && wsumA) {
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n) c = tgt[min(idx, (uint32_t)(nt - 1))];
        const double v[6] = {i < n ? (double)ax : 0.0, i < n ? (double)ay : 0.0, i < n ? (double)az : 0.0,
                             i < n ? (double)c.x : 0.0, i < n ? (double)c.y : 0.0, i < n ? (double)c.z : 0.0};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double r = wave_sum(v[k]);
            if (lane == 0) wsumA[(size_t)blockIdx.x * 6 + k] = r;
        }
    }
    if (part == 0 && wsumB) {
        const double r = wave_sum(i < n ? (double)foundB : 0.0);
        if (lane == 0) wsumB[blockIdx.x] = r;
    }
}

// buildLUTKernel (registration.cu:258-278) through the box scan of the shifted targets, coarse to fine:
// pass 0 evaluates the nodes whose (clamped) coordinates are multiples of kLutCoarse unseeded, pass 1
// evaluates every node seeded with the triangle-inequality bound from its coarse neighbour.  Both
// passes return exact minima, pass 1 recomputes the coarse nodes identically.  Threads are mapped to
// 4x4x4 node bricks so that a wave's 64 queries are spatially compact.
constexpr int kLutCoarse = 4;

__global__ __launch_bounds__(kBlock) void lut_build_scan_coarse_kernel(BvhView t, LutGeom g, float* __restrict__ lut) {
    const int ncx = (g.dx + kLutCoarse - 1) / kLutCoarse, ncy = (g.dy + kLutCoarse - 1) / kLutCoarse, ncz = (g.dz + kLutCoarse - 1) / kLutCoarse;
    // 4x4x4 bricks of coarse nodes per wave
    const int bx = (ncx + 3) / 4, by = (ncy + 3) / 4, bz = (ncz + 3) / 4;
    const size_t wave = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= (size_t)bx * by * bz) return;
    const int wx = (int)(wave % bx), wy = (int)((wave / bx) % by), wz = (int)(wave / ((size_t)bx * by));
    const int jx = wx * 4 + (lane & 3), jy = wy * 4 + ((lane >> 2) & 3), jz = wz * 4 + (lane >> 4);
    const bool active = jx < ncx && jy < ncy && jz < ncz;
    const int ix = min(jx, ncx - 1) * kLutCoarse, iy = min(jy, ncy - 1) * kLutCoarse, iz = min(jz, ncz - 1) * kLutCoarse;
    const float cx = (float)ix * g.resolution, cy = (float)iy * g.resolution, cz = (float)iz * g.resolution;  // :265
    const float v = scan_min_d2(t, cx, cy, cz, 3.402823466e+38f, 3.402823466e+38f, active);
    if (active) lut[((size_t)(iz + 1) * g.py + (iy + 1)) * (size_t)g.px + (ix + 1)] = v;
}

__global__ __launch_bounds__(kBlock) void lut_build_scan_kernel(BvhView t, LutGeom g, const float* __restrict__ coarse, float* __restrict__ lut, uint32_t* __restrict__ lut_idx) {
    const int bx = (g.px + 3) / 4, by = (g.py + 3) / 4, bz = (g.pz + 3) / 4;
    const size_t wave = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= (size_t)bx * by * bz) return;
    const int wx = (int)(wave % bx), wy = (int)((wave / bx) % by), wz = (int)(wave / ((size_t)bx * by));
    const int x = wx * 4 + (lane & 3), y = wy * 4 + ((lane >> 2) & 3), z = wz * 4 + (lane >> 4);  // padded node index
    const bool active = x < g.px && y < g.py && z < g.pz;
    const int ix = min(max(x - 1, 0), g.dx - 1), iy = min(max(y - 1, 0), g.dy - 1), iz = min(max(z - 1, 0), g.dz - 1);
    const float cx = (float)ix * g.resolution, cy = (float)iy * g.resolution, cz = (float)iz * g.resolution;  // :265
    const int kx = ix / kLutCoarse * kLutCoarse, ky = iy / kLutCoarse * kLutCoarse, kz = iz / kLutCoarse * kLutCoarse;
    const float T = coarse[((size_t)(kz + 1) * g.py + (ky + 1)) * (size_t)g.px + (kx + 1)];
    const float ddx = (float)(ix - kx) * g.resolution, ddy = (float)(iy - ky) * g.resolution, ddz = (float)(iz - kz) * g.resolution;
    const float u = sqrtf(T) + sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
    uint32_t who = 0x7fffffffu;
    const float v = lut_idx ? scan_min_d2(t, cx, cy, cz, u * u * 1.0001f + 1e-6f, 3.402823466e+38f, active, &who)
                            : scan_min_d2(t, cx, cy, cz, u * u * 1.0001f + 1e-6f, 3.402823466e+38f, active);  // FLT_MAX, :266
    if (active) {
        lut[((size_t)z * g.py + y) * (size_t)g.px + x] = v;
        if (lut_idx) lut_idx[((size_t)z * g.py + y) * (size_t)g.px + x] = who;
    }
}

__global__ __launch_bounds__(kBlock) void fill_u32_kernel(uint32_t* p, uint32_t v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) p[i] = v;
}

// ---------------------------------------------------------------------------------------------
// Deterministic fp64 sums (stand-ins for thrust::reduce at registration.cu:79-80, icp3d.cu:152-166)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void sum_f32_kernel(const uint32_t* __restrict__ bits, int n, double* __restrict__ bp, const int* __restrict__ done) {
    if (done && *done) return;
    __shared__ double red[4];
    double acc[1] = {0.0};
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) acc[0] += (double)__uint_as_float(bits[i]);
    const double r = block_sum<1>(acc, red);
    if (threadIdx.x == 0) bp[blockIdx.x] = r;
}

// single block, one wave per component: out[k] = sum over blocks of bp[b*width + k]
// (lane-strided partial sums + shuffle tree: a fixed order, hence reproducible)
__global__ __launch_bounds__(1024) void sum_partials_kernel(const double* __restrict__ bp, int nblocks, int width, double* __restrict__ out) {
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (k >= width) return;
    double s = 0.0;
    for (int b = lane; b < nblocks; b += 64) s += bp[(size_t)b * width + k];
    s = wave_sum(s);
    if (lane == 0) out[k] = s;
}

// kernRotateTranslateInplace — fgoicp/icp3d.cu:30-36
__global__ __launch_bounds__(kBlock) void transform_inplace_kernel(float4* __restrict__ pts, int n, Rt rt) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 p = pts[i];
    float rx, ry, rz;
    rotate(rt.R, p.x, p.y, p.z, rx, ry, rz);
    p.x = rx + rt.t[0];
    p.y = ry + rt.t[1];
    p.z = rz + rt.t[2];
    pts[i] = p;
}

// Sum of the working cloud and of its correspondences (icp3d.cu:152-153).
__global__ __launch_bounds__(kBlock) void icp_sums_kernel(const float4* __restrict__ work, const float4* __restrict__ tgt,
                                                          const uint32_t* __restrict__ idx, int n, int nt, const unsigned char* __restrict__ use,
                                                          double* __restrict__ bp, const int* __restrict__ done) {
    if (done && *done) return;
    __shared__ double red[24];
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        if (use && !use[i]) continue;  // trimmed ICP: inliers only
        const float4 a = work[i];
        const float4 c = tgt[min(idx[i], (uint32_t)(nt - 1))];  // an index is always found for finite clouds; never read out of bounds
        acc[0] += (double)a.x; acc[1] += (double)a.y; acc[2] += (double)a.z;
        acc[3] += (double)c.x; acc[4] += (double)c.y; acc[5] += (double)c.z;
    }
    const double r = block_sum<6>(acc, red);
    if (threadIdx.x < 6) bp[(size_t)blockIdx.x * 6 + threadIdx.x] = r;
}

// kernCentralize x2 + kernOuterProduct + reduce (icp3d.cu:38-52, :158-166), fused: per-point fp32
// centring and products exactly as the reference, fp64 accumulation.  Output in glm::mat3 order:
// ABt[col][row] = sum a[row]*b[col]  (glm::outerProduct(c, r): m[i] = c * r[i]).
__global__ __launch_bounds__(kBlock) void icp_cov_kernel(const float4* __restrict__ work, const float4* __restrict__ tgt,
                                                         const uint32_t* __restrict__ idx, int n, int nt, const float* __restrict__ cen,
                                                         const unsigned char* __restrict__ use, double* __restrict__ bp) {
    __shared__ double red[36];
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        if (use && !use[i]) continue;
        const float4 p = work[i];
        const float4 q = tgt[min(idx[i], (uint32_t)(nt - 1))];
        const float a[3] = {p.x - cen[0], p.y - cen[1], p.z - cen[2]};
        const float b[3] = {q.x - cen[3], q.y - cen[4], q.z - cen[5]};
#pragma unroll
        for (int col = 0; col < 3; ++col)
#pragma unroll
            for (int row = 0; row < 3; ++row) acc[col * 3 + row] += (double)(a[row] * b[col]);
    }
    const double r = block_sum<9>(acc, red);
    if (threadIdx.x < 9) bp[(size_t)blockIdx.x * 9 + threadIdx.x] = r;
}

// The same with icp_centroids_kernel folded in (device-resident ICP loop: one launch less on the iteration's chain): every block
// reduces the block partials of icp_sums_kernel itself — per component one wave, lane-strided partial sums + the shuffle tree,
// rounded to fp32, divided by float(ns): the arithmetic and order of icp_centroids_kernel, hence the same bits — and block 0
// leaves the centroids in device memory for the step kernel.
__global__ __launch_bounds__(kBlock) void icp_cov_cen_kernel(const float4* __restrict__ work, const float4* __restrict__ tgt,
                                                             const uint32_t* __restrict__ idx, int n, int nt, const double* __restrict__ sums_bp,
                                                             int sums_nblocks, int from_waves, float* __restrict__ cen_out, double* __restrict__ bp,
                                                             const int* __restrict__ done) {
    if (done && *done) return;
    __shared__ double red[36];
    __shared__ float cen[6];
    {
        // from_waves: sums_bp holds the per-WAVE sums the correspondence scan left (nn_scan_kernel's epilogue), `from_waves` of them;
        // a block partial of icp_sums_kernel is its four waves in order
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int k = wave; k < 6; k += kBlock / 64) {
            double sacc = 0.0;
            for (int b = lane; b < sums_nblocks; b += 64) {
                if (from_waves) {
                    double blk = sums_bp[(size_t)(4 * b) * 6 + k];
#pragma unroll
                    for (int w = 1; w < 4; ++w) blk += 4 * b + w < from_waves ? sums_bp[(size_t)(4 * b + w) * 6 + k] : 0.0;
                    sacc += blk;
                } else {
                    sacc += sums_bp[(size_t)b * 6 + k];
                }
            }
            sacc = wave_sum(sacc);
            if (lane == 0) cen[k] = (float)sacc / (float)n;
        }
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x < 6) cen_out[threadIdx.x] = cen[threadIdx.x];
    }
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 p = work[i];
        const float4 q = tgt[min(idx[i], (uint32_t)(nt - 1))];
        const float a[3] = {p.x - cen[0], p.y - cen[1], p.z - cen[2]};
        const float b[3] = {q.x - cen[3], q.y - cen[4], q.z - cen[5]};
#pragma unroll
        for (int col = 0; col < 3; ++col)
#pragma unroll
            for (int row = 0; row < 3; ++row) acc[col * 3 + row] += (double)(a[row] * b[col]);
    }
    const double r = block_sum<9>(acc, red);
    if (threadIdx.x < 9) bp[(size_t)blockIdx.x * 9 + threadIdx.x] = r;
}

// ---------------------------------------------------------------------------------------------
// Device-resident ICP loop (IterativeClosestPoint3D::run, icp3d.cu:88-107).  The host loop needs the covariance on the host for
// the 3x3 SVD and the SSE for the loop test — two host round trips per iteration, 18 us of a 90-us iteration at 40k points.
// Here the state of the loop lives in device memory (IcpDevState) and ONE thread advances it:
//   step j  =  [loop test of icp3d.cu:94 with the SSE of iteration j-1]  +  [Procrustes finish of iteration j: cross-covariance
//               from the block partials, closest_orthogonal_approximation (math3.hpp, the host's source compiled for the device),
//               t_ = c_corr - R_ c_src, R = R_ R, t = R_ t + t_]
// so the passes of iteration j+1 can be enqueued before iteration j has been decided: every kernel of the loop returns at once
// when `done` is set, and the scans take their motion from the state instead of the kernarg segment.  Sums are folded exactly
// as sum_partials_kernel folds them (one wave per component, lane-strided, shuffle tree), so (sse, R, t, iterations) are the bits
// of the host loop (FGOICP_ICP_DEVICE=0; tests/test_gpu_ops.py).
// ---------------------------------------------------------------------------------------------
__global__ void icp_init_kernel(IcpDevState* __restrict__ st, Rt rt0, int max_iter, float thr, IcpHostResult* __restrict__ res) {
    if (threadIdx.x != 0) return;
    for (int k = 0; k < 9; ++k) { st->R[k] = rt0.R[k]; st->Rn[k] = rt0.R[k]; st->last_R[k] = (k % 4 == 0) ? 1.0f : 0.0f; }
    for (int k = 0; k < 3; ++k) { st->t[k] = rt0.t[k]; st->tn[k] = rt0.t[k]; st->last_t[k] = 0.0f; }
    st->sse = kInf;                 // icp3d.cu:89-90
    st->last_sse = 2.0f * kInf;
    st->iters = 0;
    st->done = 0;
    st->max_iter = max_iter;
    st->thr = thr;
    res->iters_done = 0;
    res->done = 0;
}

__global__ __launch_bounds__(640) void icp_step_kernel(IcpDevState* __restrict__ st, const double* __restrict__ bp_cov, int nb_cov,
                                                       const double* __restrict__ bp_sse, int nb_sse, const float* __restrict__ cen,
                                                       IcpHostResult* __restrict__ res) {
    if (st->done) return;
    __shared__ double red[10];
    {   // sum_partials_kernel's fold: wave k < 9 the covariance component k, wave 9 the SSE of the iteration before
        const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const double* bp = k < 9 ? bp_cov : bp_sse;
        const int nb = k < 9 ? nb_cov : nb_sse, width = k < 9 ? 9 : 1, col = k < 9 ? k : 0;
        double s = 0.0;
        for (int b = lane; b < nb; b += 64) s += bp[(size_t)b * width + col];
        s = wave_sum(s);
        if (lane == 0) red[k] = s;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int it = st->iters;                       // iterations completed = `iter` of icp3d.cu:94 before its post-increment
    float sse = st->sse, last_sse = st->last_sse;
    if (it > 0) sse = (float)red[9];                // sse = reg.compute_sse_error(R, t) of iteration `it` (:103)
    Mat3f R = Mat3f::from(st->R);
    Vec3f t{st->t[0], st->t[1], st->t[2]};
    const bool go_on = it < st->max_iter && (last_sse - sse) > st->thr * last_sse;   // :94
    if (!go_on) {
        const bool cur_best = sse < last_sse;       // :106-107
        res->sse = cur_best ? sse : last_sse;
        for (int k = 0; k < 9; ++k) res->R[k] = cur_best ? R.m[k] : st->last_R[k];
        for (int k = 0; k < 3; ++k) res->t[k] = cur_best ? st->t[k] : st->last_t[k];
        res->iters = it;
        st->sse = sse;
        st->done = 1;
        __threadfence_system();
        *(volatile int*)&res->done = 1;
        return;
    }
    for (int k = 0; k < 9; ++k) st->last_R[k] = R.m[k];   // :96-98
    for (int k = 0; k < 3; ++k) st->last_t[k] = st->t[k];
    st->last_sse = sse;
    st->sse = sse;
    Mat3f ABt;
    for (int k = 0; k < 9; ++k) ABt.m[k] = (float)red[k];
    const Mat3f Rn = closest_orthogonal_approximation(ABt);  // :168
    const Vec3f sc{cen[0], cen[1], cen[2]}, cc{cen[3], cen[4], cen[5]};
    const Vec3f tn = cc - Rn * sc;                             // :169
    R = Rn * R;                                                // :101
    t = Rn * t + tn;                                           // :102
    for (int k = 0; k < 9; ++k) { st->R[k] = R.m[k]; st->Rn[k] = Rn.m[k]; }
    st->t[0] = t.x; st->t[1] = t.y; st->t[2] = t.z;
    st->tn[0] = tn.x; st->tn[1] = tn.y; st->tn[2] = tn.z;
    st->iters = it + 1;
    *(volatile int*)&res->iters_done = it + 1;  // progress hint for the host's look-ahead (no ordering needed: the host only paces itself by it)
}

// Centroids on the device (icp3d.cu:152-156): ordered fp64 sum of the block partials, rounded to fp32,
// divided by float(ns) in fp32 (correctly rounded division, as on the host).  Written both to device
// memory (for icp_cov_kernel) and to pinned host memory (for t_ = c_corr - R_ * c_src).
__global__ __launch_bounds__(384) void icp_centroids_kernel(const double* __restrict__ bp, int nblocks, int ns, float* __restrict__ cen_dev,
                                                            float* __restrict__ cen_host) {
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;  // one wave per component
    double s = 0.0;
    for (int b = lane; b < nblocks; b += 64) s += bp[(size_t)b * 6 + k];
    s = wave_sum(s);
    if (lane == 0) {
        const float c = (float)s / (float)ns;
        cen_dev[k] = c;
        cen_host[k] = c;
    }
}

Rt make_rt(const float* R9, const float* t3) {
    Rt rt{};
    for (int i = 0; i < 9; ++i) rt.R[i] = R9 ? R9[i] : (i % 4 == 0 ? 1.0f : 0.0f);
    for (int i = 0; i < 3; ++i) rt.t[i] = t3 ? t3[i] : 0.0f;
    return rt;
}

int nn_slices(int nq, int nt) {
    // enough (query tile x target slice) blocks to put >= 8 blocks on each of the 256 CUs
    const int qtiles = (nq + kBlock * kNnQ - 1) / (kBlock * kNnQ);
    int want = (2048 + qtiles - 1) / qtiles;
    const int max_slices = (nt + kTile - 1) / kTile;
    want = want < 1 ? 1 : want;
    return want > max_slices ? max_slices : want;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------
void launch_bounds(const float4* src, int ns, const float* lut, const LutGeom& g, const BoundsArgs& a, double2* partials, int nchunk,
                   int P, hipStream_t s) {
#ifndef FGOICP_DEV_KNOBS
    (void)src; (void)ns; (void)lut; (void)g; (void)a; (void)partials; (void)nchunk; (void)P; (void)s;  // the per-rotation-node kernel is the development build's A/B reference (FGOICP_BOUNDS_SORTED=0)
#else
    dim3 grid((unsigned)nchunk * (unsigned)a.B), block(kBlock);
    switch (P) {
        case 1: hipLaunchKernelGGL(bounds_kernel<1>, grid, block, 0, s, src, ns, lut, g, a, partials, nchunk); break;
        case 2: hipLaunchKernelGGL(bounds_kernel<2>, grid, block, 0, s, src, ns, lut, g, a, partials, nchunk); break;
        case 4: hipLaunchKernelGGL(bounds_kernel<4>, grid, block, 0, s, src, ns, lut, g, a, partials, nchunk); break;
        default: hipLaunchKernelGGL(bounds_kernel<8>, grid, block, 0, s, src, ns, lut, g, a, partials, nchunk); break;
    }
#endif
}

// The locality sort of one tick (descriptors must already be on the device): keys + histogram, scan, scatter.
void launch_tick_sort(const LutGeom& g, const float4* chunk_cen, int nchunk, const TickGroup* groups, const TickSub* subs, int nsub, int cell_shift,
                      unsigned short* keys, unsigned* ranks, unsigned* hist, unsigned* hist_xcd, unsigned* xoff, unsigned* block_sums, unsigned* cursor, unsigned* sorted,
                      int allow_xcd, int prefill, unsigned* check_err, int inject_fault, hipStream_t s, int nunits, int unit_m, const float* tier_lut, float tier_level) {
    const size_t nitems = (size_t)(nsub - nunits * (unit_m - 1)) * nchunk;
    const unsigned kb = (unsigned)std::min<size_t>((nitems + 63) / 64, 8192);  // `hist` / `hist_xcd` are zero here: the scan / fold kernels re-zero them
#ifdef FGOICP_DEV_KNOBS
    static const int hilbert = [] { const char* e = dev_env("FGOICP_SORT_CURVE"); return e ? std::atoi(e) : 1; }();  // tuning knob: 1 = Hilbert (default), 0 = Z-order
    static const int use_ranks = [] { const char* e = dev_env("FGOICP_SORT_RANKS"); return e ? std::atoi(e) : 1; }();  // tuning knob
    static const int orient = [] { const char* e = dev_env("FGOICP_SORT_ORIENT"); return e ? std::atoi(e) : 0; }();  // tuning knob (experimental): orientation bits in the sort key
#else
    constexpr int orient = 0;  // shipped: Hilbert keys, ranks from the histogram atomic (the scatter needs no atomics of its own)
#endif
    const bool xcd = allow_xcd != 0 && hist_xcd && xoff
#ifdef FGOICP_DEV_KNOBS
                     && use_ranks != 0
#endif
        ;  // allow_xcd: per context, cleared by a failed permutation check
    if (xcd) {
#ifdef FGOICP_DEV_KNOBS
        if (!hilbert) hipLaunchKernelGGL((tick_keys_kernel<0, 1>), dim3(kb), dim3(64), 0, s, chunk_cen, nchunk, groups, subs, nsub, g, cell_shift, keys, ranks, hist_xcd, prefill ? sorted : nullptr, nunits, unit_m, orient, tier_lut, tier_level);
        else
#endif
        hipLaunchKernelGGL((tick_keys_kernel<1, 1>), dim3(kb), dim3(64), 0, s, chunk_cen, nchunk, groups, subs, nsub, g, cell_shift, keys, ranks, hist_xcd, prefill ? sorted : nullptr, nunits, unit_m, orient, tier_lut, tier_level);
        hipLaunchKernelGGL(tick_fold_sums_kernel, dim3(kScanBlocks), dim3(64), 0, s, hist_xcd, xoff, hist, block_sums);
    } else {
#ifdef FGOICP_DEV_KNOBS
        if (!hilbert) hipLaunchKernelGGL((tick_keys_kernel<0, 0>), dim3(kb), dim3(64), 0, s, chunk_cen, nchunk, groups, subs, nsub, g, cell_shift, keys, ranks, hist, prefill ? sorted : nullptr, nunits, unit_m, orient, tier_lut, tier_level);
        else
#endif
        hipLaunchKernelGGL((tick_keys_kernel<1, 0>), dim3(kb), dim3(64), 0, s, chunk_cen, nchunk, groups, subs, nsub, g, cell_shift, keys, ranks, hist, prefill ? sorted : nullptr, nunits, unit_m, orient, tier_lut, tier_level);
    }
    if (!xcd) hipLaunchKernelGGL(tick_scan_sums_kernel, dim3(kScanBlocks), dim3(64), 0, s, hist, block_sums);
    hipLaunchKernelGGL(tick_scan_apply_kernel, dim3(kScanBlocks), dim3(64), 0, s, hist, block_sums, cursor);
    if (xcd) hipLaunchKernelGGL(tick_scatter_xcd_kernel, dim3(kb), dim3(64), 0, s, keys, ranks, nitems, cursor, xoff, sorted);
#ifdef FGOICP_DEV_KNOBS
    else if (!use_ranks) hipLaunchKernelGGL(tick_scatter_atomic_kernel, dim3(kb), dim3(64), 0, s, keys, nitems, cursor, sorted);
#endif
    else hipLaunchKernelGGL(tick_scatter_kernel, dim3(kb), dim3(64), 0, s, keys, ranks, nitems, cursor, sorted);
    if (inject_fault) hipLaunchKernelGGL(tick_fault_kernel, dim3(1), dim3(1), 0, s, sorted);
#ifdef FGOICP_DEV_KNOBS
    if (check_err) hipLaunchKernelGGL(tick_check_kernel, dim3(kb), dim3(64), 0, s, sorted, nitems, check_err);  // FGOICP_SEPARATE_CHECK=1 (ctx.hip)
#endif
}

#ifdef FGOICP_DEV_KNOBS
// development build: does a knob select one of round 3's kernels (which know neither thresholds nor chunk spans) for this context's windows?
bool bounds_dev_variant_selected(const float2* zp, int layout, int unit_m) {
    const int lds_rows = [] { const char* e = dev_env("FGOICP_LDS_TILES"); return e ? std::atoi(e) : 0; }();
    const int item_kernel = [] { const char* e = dev_env("FGOICP_BOUNDS_ITEM"); return e ? std::atoi(e) : 1; }();
    return lds_rows == 128 || lds_rows == 192 || unit_m > 1 || !item_kernel || dev_env("FGOICP_BOUNDS_VARIANT") || dev_env("FGOICP_ITEMS_PER_WG") || dev_env("FGOICP_NT_SOURCE") ||
           dev_env("FGOICP_TRIM_VARIANT") || dev_env("FGOICP_QUAD_PAIRED") || dev_env("FGOICP_LDS_PAD") || !zp || layout == 3;
}
// Round 3's launch logic with every variant behind its knob; false = nothing launched (the shipped kernel follows).
static bool launch_bounds_sorted_dev(const float4* src, int ns, const float* lut, const float2* zp, int layout, const LutGeom& g, int nchunk, int chunk_pts,
                                     const TickGroup* groups, const TickSub* subs, int nsub, const unsigned* sorted, double2* partials, float* evals, size_t erow, int samp_shift,
                                     hipStream_t s, int nunits, int unit_m) {
    const hipEvent_t ev_start = nullptr, ev_stop = nullptr;  // (recorded by the caller)
    const size_t nitems = (size_t)(nsub - nunits * (unit_m - 1)) * nchunk;
    const int lds_rows = [] { const char* e = dev_env("FGOICP_LDS_TILES"); return e ? std::atoi(e) : 0; }();  // tuning knob / A-B (read per launch: tests toggle it): 128 or 192 rows of 16 floats per wave
    if (nunits == 0 && (lds_rows == 128 || lds_rows == 192) && lut) {
        static unsigned* d_stat = [] { unsigned* p = nullptr; if (dev_env("FGOICP_LDS_STATS")) { (void)hipMalloc(&p, 8); (void)hipMemset(p, 0, 8); } return p; }();
        if (ev_start) (void)hipEventRecord(ev_start, s);
        const dim3 lgrid((unsigned)nitems);
        if (evals) {
            if (lds_rows == 128) hipLaunchKernelGGL((bounds_lds_kernel<1, 128>), lgrid, dim3(64), 0, s, src, ns, lut, g, groups, subs, sorted, nchunk, chunk_pts, partials, evals, erow, samp_shift, (unsigned)nitems, d_stat);
            else hipLaunchKernelGGL((bounds_lds_kernel<1, 192>), lgrid, dim3(64), 0, s, src, ns, lut, g, groups, subs, sorted, nchunk, chunk_pts, partials, evals, erow, samp_shift, (unsigned)nitems, d_stat);
        } else {
            if (lds_rows == 128) hipLaunchKernelGGL((bounds_lds_kernel<0, 128>), lgrid, dim3(64), 0, s, src, ns, lut, g, groups, subs, sorted, nchunk, chunk_pts, partials, evals, erow, samp_shift, (unsigned)nitems, d_stat);
            else hipLaunchKernelGGL((bounds_lds_kernel<0, 192>), lgrid, dim3(64), 0, s, src, ns, lut, g, groups, subs, sorted, nchunk, chunk_pts, partials, evals, erow, samp_shift, (unsigned)nitems, d_stat);
        }
        if (ev_stop) (void)hipEventRecord(ev_stop, s);
        if (d_stat) {
            static int calls = 0;
            if ((++calls & 63) == 0) {
                unsigned h[2] = {0, 0};
                (void)hipStreamSynchronize(s);
                (void)hipMemcpy(h, d_stat, 8, hipMemcpyDeviceToHost);
                std::fprintf(stderr, "[fgoicp lds tiles] %u of %u passes staged (%.1f %%)\n", h[0], h[1], h[1] ? 100.0 * h[0] / h[1] : 0.0);
            }
        }
        return true;
    }
    if (nunits > 0) {  // sibling units (dense clouds): the z-pair or plain layouts, one wave per item, 4 points per lane
        if (ev_start) (void)hipEventRecord(ev_start, s);
        const dim3 ugrid((unsigned)nitems);
#define FGOICP_LAUNCH_UNITS(Z, TR, M) \
        hipLaunchKernelGGL((bounds_units_kernel<Z, TR, M>), ugrid, dim3(64), 0, s, src, ns, lut, zp, g, groups, subs, sorted, nchunk, chunk_pts, partials, evals, erow, samp_shift, (unsigned)nitems, nunits)
        const int z = (zp && layout == 2) ? 2 : zp ? 1 : 0;
        if (evals) {
            if (unit_m == 8) { if (z == 2) FGOICP_LAUNCH_UNITS(2, 1, 8); else if (z == 1) FGOICP_LAUNCH_UNITS(1, 1, 8); else FGOICP_LAUNCH_UNITS(0, 1, 8); }
            else { if (z == 2) FGOICP_LAUNCH_UNITS(2, 1, 4); else if (z == 1) FGOICP_LAUNCH_UNITS(1, 1, 4); else FGOICP_LAUNCH_UNITS(0, 1, 4); }
        } else {
            if (unit_m == 8) { if (z == 2) FGOICP_LAUNCH_UNITS(2, 0, 8); else if (z == 1) FGOICP_LAUNCH_UNITS(1, 0, 8); else FGOICP_LAUNCH_UNITS(0, 0, 8); }
            else { if (z == 2) FGOICP_LAUNCH_UNITS(2, 0, 4); else if (z == 1) FGOICP_LAUNCH_UNITS(1, 0, 4); else FGOICP_LAUNCH_UNITS(0, 0, 4); }
        }
#undef FGOICP_LAUNCH_UNITS
        if (ev_stop) (void)hipEventRecord(ev_stop, s);
        return true;
    }
    // FGOICP_BOUNDS_ITEM (default 1): 0 = round 3's kernel family below instead of bounds_item_kernel; the variants further down imply it
    const int item_kernel = [] { const char* e = dev_env("FGOICP_BOUNDS_ITEM"); return e ? std::atoi(e) : 1; }();  // (read per launch: tests toggle it)
    const bool other_variant = dev_env("FGOICP_BOUNDS_VARIANT") || dev_env("FGOICP_ITEMS_PER_WG") || dev_env("FGOICP_NT_SOURCE") || dev_env("FGOICP_TRIM_VARIANT") || dev_env("FGOICP_QUAD_PAIRED") ||
                               dev_env("FGOICP_LDS_PAD") || !zp || layout == 3;
    if (item_kernel && !other_variant) return false;
    const TickGroup* gp = groups;
    const TickSub* sp = subs;
    static const int variant = [] { const char* e = dev_env("FGOICP_BOUNDS_VARIANT"); return e ? std::atoi(e) : 2; }();  // tuning knob (2 = default: one wave, 4 points per lane)
    const dim3 grid((unsigned)nitems);
    static const int wpg = [] { const char* e = dev_env("FGOICP_ITEMS_PER_WG"); return e ? std::atoi(e) : 1; }();   // tuning knob: 1, 2 or 4 one-wave items per workgroup
    static const int nt_src = [] { const char* e = dev_env("FGOICP_NT_SOURCE"); return e ? std::atoi(e) : 0; }();   // tuning knob: non-temporal source loads
    static const unsigned lds_pad = [] { const char* e = dev_env("FGOICP_LDS_PAD"); return e ? (unsigned)std::atoi(e) : 0u; }();  // tuning knob: unused dynamic LDS per workgroup = fewer resident waves per CU
#define FGOICP_LAUNCH_SORTED(T, PP, Z, TR) \
    hipLaunchKernelGGL((bounds_sorted_kernel<T, PP, Z, TR>), grid, dim3(T), lds_pad, s, src, ns, lut, zp, g, gp, sp, sorted, nchunk, chunk_pts, partials, evals, erow, samp_shift, (unsigned)nitems)
#define FGOICP_LAUNCH_WPG(Z, W, N) \
    hipLaunchKernelGGL((bounds_sorted_kernel<64, 4, Z, 0, W, N>), dim3((unsigned)((nitems + W - 1) / W)), dim3(64 * W), 0, s, src, ns, lut, zp, g, gp, sp, sorted, nchunk, chunk_pts, partials, evals, erow, samp_shift, (unsigned)nitems)
    if (!evals && variant == 2 && (wpg > 1 || nt_src)) {  // experimental variants of the default 64 x 4 kernel
        const int z = (zp && layout == 2) ? 3 : zp ? 1 : 0;
        bool done = true;
        if (z == 3) { if (wpg == 4 && nt_src) FGOICP_LAUNCH_WPG(3, 4, 1); else if (wpg == 4) FGOICP_LAUNCH_WPG(3, 4, 0); else if (wpg == 2 && nt_src) FGOICP_LAUNCH_WPG(3, 2, 1); else if (wpg == 2) FGOICP_LAUNCH_WPG(3, 2, 0); else if (nt_src) FGOICP_LAUNCH_WPG(3, 1, 1); else done = false; }
        else if (z == 1) { if (wpg == 4 && nt_src) FGOICP_LAUNCH_WPG(1, 4, 1); else if (wpg == 4) FGOICP_LAUNCH_WPG(1, 4, 0); else if (wpg == 2 && nt_src) FGOICP_LAUNCH_WPG(1, 2, 1); else if (wpg == 2) FGOICP_LAUNCH_WPG(1, 2, 0); else if (nt_src) FGOICP_LAUNCH_WPG(1, 1, 1); else done = false; }
        else done = false;
        if (done) return true;
    }
    if (evals) {
        static const int trim_variant = [] { const char* e = dev_env("FGOICP_TRIM_VARIANT"); return e ? std::atoi(e) : 2; }();  // tuning knob (2 = 64x4, default)
        if (trim_variant == 2) {
            if (zp && layout == 4) FGOICP_LAUNCH_SORTED(64, 4, 5, 1); else
            if (zp && layout == 2) FGOICP_LAUNCH_SORTED(64, 4, 3, 1); else if (zp) FGOICP_LAUNCH_SORTED(64, 4, 1, 1); else FGOICP_LAUNCH_SORTED(64, 4, 0, 1);
        } else if (zp && layout == 2) FGOICP_LAUNCH_SORTED(128, 2, 2, 1); else if (zp) FGOICP_LAUNCH_SORTED(128, 2, 1, 1); else FGOICP_LAUNCH_SORTED(128, 2, 0, 1);
    } else if (zp && layout == 3) {
        FGOICP_LAUNCH_SORTED(64, 4, 4, 0);
    } else if (zp && layout == 4) {
        FGOICP_LAUNCH_SORTED(64, 4, 5, 0);
    } else if (zp && layout == 2) {
        static const int paired = [] { const char* e = dev_env("FGOICP_QUAD_PAIRED"); return e ? std::atoi(e) : 1; }();  // tuning knob (1 = default)
        if (variant == 2 && paired) FGOICP_LAUNCH_SORTED(64, 4, 3, 0); else
        if (variant == 0) FGOICP_LAUNCH_SORTED(256, 1, 2, 0); else if (variant == 2) FGOICP_LAUNCH_SORTED(64, 4, 2, 0);
        else if (variant == 3) FGOICP_LAUNCH_SORTED(64, 2, 2, 0); else if (variant == 4) FGOICP_LAUNCH_SORTED(64, 1, 2, 0); else FGOICP_LAUNCH_SORTED(128, 2, 2, 0);
    } else if (zp) {
        if (variant == 0) FGOICP_LAUNCH_SORTED(256, 1, 1, 0); else if (variant == 2) FGOICP_LAUNCH_SORTED(64, 4, 1, 0);
        else if (variant == 3) FGOICP_LAUNCH_SORTED(64, 2, 1, 0); else if (variant == 4) FGOICP_LAUNCH_SORTED(64, 1, 1, 0); else FGOICP_LAUNCH_SORTED(128, 2, 1, 0);
    } else {
        if (variant == 0) FGOICP_LAUNCH_SORTED(256, 1, 0, 0); else if (variant == 2) FGOICP_LAUNCH_SORTED(64, 4, 0, 0); else FGOICP_LAUNCH_SORTED(128, 2, 0, 0);
    }
#undef FGOICP_LAUNCH_SORTED
#undef FGOICP_LAUNCH_WPG
    return true;
}

#endif  // FGOICP_DEV_KNOBS

// The bounds kernel of a window.  Shipped: bounds_item_kernel (bounds_item.hpp), one instantiation per packed layout x trimmed x wide
// addressing x weight quantisation.  Development build: FGOICP_BOUNDS_ITEM=0 runs round 3's bounds_sorted_kernel family instead (the
// bit reference of tests/test_gpu_ops.py::test_item_kernel_keeps_every_bit), and the variants that were measured and rejected —
// sibling units, LDS tiles, several items per workgroup, other thread / point shapes — stay selectable by their knobs (NOTES.md).
template <int LAYOUT, int TRIM>
static void launch_item(const float4* src, int ns, const char* lutp, const LutGeom& g, bool wide, const TickGroup* groups, const TickSub* subs, const unsigned* sorted, int nchunk,
                        int chunk_pts, double2* partials, float* evals, size_t erow, int samp_shift, unsigned nitems, unsigned* sort_err, const TickCut& cut, int span, hipStream_t s) {
    const dim3 grid(nitems), block(64);
#define FGOICP_ITEM(W, Q, S) hipLaunchKernelGGL((bounds_item_kernel<LAYOUT, TRIM, W, Q, S>), grid, block, 0, s, src, ns, lutp, g, groups, subs, sorted, nchunk, chunk_pts, partials, evals, erow, samp_shift, nitems, sort_err, cut, span)
    if (!TRIM && span > 1) {  // (trimmed windows carry no thresholds, hence no spans)
        if (wide) { if (g.quantize) FGOICP_ITEM(true, true, !TRIM); else FGOICP_ITEM(true, false, !TRIM); }
        else { if (g.quantize) FGOICP_ITEM(false, true, !TRIM); else FGOICP_ITEM(false, false, !TRIM); }
    } else if (wide) { if (g.quantize) FGOICP_ITEM(true, true, false); else FGOICP_ITEM(true, false, false); }
    else { if (g.quantize) FGOICP_ITEM(false, true, false); else FGOICP_ITEM(false, false, false); }
#undef FGOICP_ITEM
}

bool launch_bounds_sorted(const float4* src, int ns, const float* lut, const float2* zp, int layout, const LutGeom& g, int nchunk, int chunk_pts,
                          const TickGroup* groups, const TickSub* subs, int nsub, const unsigned* sorted, double2* partials, float* evals, size_t erow, int samp_shift,
                          unsigned* sort_err, const TickCut& cut, int span, hipEvent_t ev_start, hipEvent_t ev_stop, hipStream_t s, int nunits, int unit_m) {
    const size_t nitems = (size_t)(nsub - nunits * (unit_m - 1)) * (size_t)((nchunk + span - 1) / span);
    if (ev_start) (void)hipEventRecord(ev_start, s);
    bool done = false, item_kernel = false;
#ifdef FGOICP_DEV_KNOBS
    done = launch_bounds_sorted_dev(src, ns, lut, zp, layout, g, nchunk, chunk_pts, groups, subs, nsub, sorted, partials, evals, erow, samp_shift, s, nunits, unit_m);
    // round 3's kernels do not look at the values they read from `sorted`: their check is a launch of its own
    if (done && sorted && sort_err) hipLaunchKernelGGL(tick_check_kernel, dim3((unsigned)std::min<size_t>((nitems + 63) / 64, 4096)), dim3(64), 0, s, sorted, nitems, sort_err);
#endif
    if (!done && zp && (layout == 1 || layout == 2 || layout == 4) && nunits == 0) {
        const size_t nodes = (size_t)g.px * g.py * g.pz;
        // 32-bit texel addressing: the row number (z * py + y) must fit the signed 24-bit multiply and the byte offsets of the packed copy 32 bits
        const size_t bytes = layout == 4 ? (size_t)((g.px + 2) / 3) * ((g.py + 1) / 2) * g.pz * 8 * sizeof(float4) : nodes * (layout == 2 ? sizeof(float4) : sizeof(float2));
        const bool wide = (size_t)g.py * g.pz > ((size_t)1 << 23) || bytes + 64 > ((size_t)1 << 32);
        const char* lutp = reinterpret_cast<const char*>(zp);
        if (evals) {
            if (layout == 1) launch_item<1, 1>(src, ns, lutp, g, wide, groups, subs, sorted, nchunk, chunk_pts, partials, evals, erow, samp_shift, (unsigned)nitems, sort_err, cut, span, s);
            else if (layout == 2) launch_item<3, 1>(src, ns, lutp, g, wide, groups, subs, sorted, nchunk, chunk_pts, partials, evals, erow, samp_shift, (unsigned)nitems, sort_err, cut, span, s);
            else launch_item<5, 1>(src, ns, lutp, g, wide, groups, subs, sorted, nchunk, chunk_pts, partials, evals, erow, samp_shift, (unsigned)nitems, sort_err, cut, span, s);
        } else {
            if (layout == 1) launch_item<1, 0>(src, ns, lutp, g, wide, groups, subs, sorted, nchunk, chunk_pts, partials, evals, erow, samp_shift, (unsigned)nitems, sort_err, cut, span, s);
            else if (layout == 2) launch_item<3, 0>(src, ns, lutp, g, wide, groups, subs, sorted, nchunk, chunk_pts, partials, evals, erow, samp_shift, (unsigned)nitems, sort_err, cut, span, s);
            else launch_item<5, 0>(src, ns, lutp, g, wide, groups, subs, sorted, nchunk, chunk_pts, partials, evals, erow, samp_shift, (unsigned)nitems, sort_err, cut, span, s);
        }
        done = item_kernel = true;
    }
    if (!done) std::fprintf(stderr, "fgoicp: no bounds kernel for packed layout %d in this build\n", layout);  // (ctx_create only chooses layouts 1, 2, 4 outside the development build)
    if (ev_stop) (void)hipEventRecord(ev_stop, s);
    return item_kernel && cut.acc && !evals;  // the early exit was in force (round 3's kernels in the development build do not know it: exact rows then)
}

void launch_tick_upload(const TickGroup* h_groups, TickGroup* d_groups, int ngroups, const TickSub* h_subs, TickSub* d_subs, int nsubs, hipStream_t s) {
    static_assert(sizeof(TickGroup) % 16 == 0 && sizeof(TickSub) % 16 == 0, "descriptors are copied in 16-byte units");
    const unsigned ng16 = (unsigned)ngroups * (sizeof(TickGroup) / 16), ns16 = (unsigned)nsubs * (sizeof(TickSub) / 16);
    hipLaunchKernelGGL(tick_upload_kernel, dim3(std::min(1024u, (ng16 + ns16 + 63) / 64)), dim3(64), 0, s, reinterpret_cast<const uint4*>(h_groups),
                       reinterpret_cast<uint4*>(d_groups), ng16, reinterpret_cast<const uint4*>(h_subs), reinterpret_cast<uint4*>(d_subs), ns16);
}

void launch_bounds_finalize(const double2* partials, int nchunk, int total, float* out_lb, float* out_ub, const TickCut& cut, hipStream_t s) {
    hipLaunchKernelGGL(bounds_finalize_kernel, dim3(total), dim3(64), 0, s, partials, nchunk, total, out_lb, out_ub, cut);
}

void launch_lut_build(const float4* tgt_shifted, int nt, const LutGeom& g, float* lut_padded, hipStream_t s) {
    const size_t total = (size_t)g.px * g.py * g.pz;
    const size_t per_block = (size_t)kBlock * kLutNodes;
    const unsigned blocks = (unsigned)((total + per_block - 1) / per_block);
    hipLaunchKernelGGL(lut_build_kernel, dim3(blocks), dim3(kBlock), 0, s, tgt_shifted, nt, g, lut_padded);
}

void launch_lut_quad_bricked(const float* lut_padded, const LutGeom& g, float4* qd, hipStream_t s) {
#ifdef FGOICP_DEV_KNOBS   // layout 3 (2 x 2 x 2 quad bricks): measured slower, development build only
    hipLaunchKernelGGL(lut_quad_bricked_kernel, dim3(8192), dim3(kBlock), 0, s, lut_padded, g, qd);
#else
    (void)lut_padded; (void)g; (void)qd; (void)s;
#endif
}

void launch_lut_quad_apron(const float* lut_padded, const LutGeom& g, float4* qd, hipStream_t s) {
    hipLaunchKernelGGL(lut_quad_apron_kernel, dim3(4096), dim3(kBlock), 0, s, lut_padded, g, qd);
}
void launch_lut_quad(const float* lut_padded, const LutGeom& g, float4* qd, hipStream_t s) {
    hipLaunchKernelGGL(lut_quad_kernel, dim3(8192), dim3(kBlock), 0, s, lut_padded, g, qd);
}

void launch_lut_zpair(const float* lut_padded, const LutGeom& g, float2* zp, hipStream_t s) {
    hipLaunchKernelGGL(lut_zpair_kernel, dim3(8192), dim3(kBlock), 0, s, lut_padded, g, zp);
}

void launch_lut_unpad(const float* lut_padded, const LutGeom& g, float* out, hipStream_t s) {
    hipLaunchKernelGGL(lut_unpad_kernel, dim3(4096), dim3(kBlock), 0, s, lut_padded, g, out);
}

void launch_lut_nodes(const float* lut, const LutGeom& g, const int* xyz, size_t n, float* out, hipStream_t s) {
    const unsigned blocks = (unsigned)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(lut_nodes_kernel, dim3(blocks < 4096 ? (blocks ? blocks : 1) : 4096), dim3(kBlock), 0, s, lut, g, xyz, n, out);
}

void launch_lut_search(const float* lut, const LutGeom& g, const float* q_xyz, size_t n, float* out, hipStream_t s) {
    const unsigned blocks = (unsigned)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(lut_search_kernel, dim3(blocks < 4096 ? (blocks ? blocks : 1) : 4096), dim3(kBlock), 0, s, lut, g, q_xyz, n, out);
}

void launch_fill_u32(uint32_t* p, uint32_t v, size_t n, hipStream_t s) {
    const unsigned blocks = (unsigned)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(fill_u32_kernel, dim3(blocks < 2048 ? (blocks ? blocks : 1) : 2048), dim3(kBlock), 0, s, p, v, n);
}

void launch_nn_min(const float4* pts, int n, const float4* tgt, int nt, const float* R9, const float* t3, int apply,
                   uint32_t* min_bits, hipStream_t s) {
    const int qtiles = (n + kBlock * kNnQ - 1) / (kBlock * kNnQ);
    const int slices = nn_slices(n, nt);
    const int slice_len = ((nt + slices - 1) / slices + kTile - 1) / kTile * kTile;
    const int nsl = (nt + slice_len - 1) / slice_len;
    hipLaunchKernelGGL(nn_min_kernel, dim3(qtiles, nsl), dim3(kBlock), 0, s, pts, n, tgt, nt, make_rt(R9, t3), apply, slice_len, min_bits);
}

void launch_nn_tie_threshold(const uint32_t* min_bits, int n, uint32_t* thr_bits, hipStream_t s) {
    hipLaunchKernelGGL(nn_tie_threshold_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, min_bits, n, thr_bits);
}

void launch_nn_first_index(const float4* pts, int n, const float4* tgt, int nt, const uint32_t* thr_bits, uint32_t* first_idx,
                           hipStream_t s) {
    const int qtiles = (n + kBlock * kNnQ - 1) / (kBlock * kNnQ);
    const int slices = nn_slices(n, nt);
    const int slice_len = ((nt + slices - 1) / slices + kTile - 1) / kTile * kTile;
    const int nsl = (nt + slice_len - 1) / slice_len;
    hipLaunchKernelGGL(nn_first_index_kernel, dim3(qtiles, nsl), dim3(kBlock), 0, s, pts, n, tgt, nt, slice_len, thr_bits, first_idx);
}

void launch_nn_scan(const float4* pts, int n, const BvhView& t, const float* lut, const LutGeom& g, const float* R9, const float* t3, int apply,
                    int want_index, const float4* tgt, int nt, const uint32_t* seed_idx, const float* skip_lb, const uint32_t* skip_u, uint32_t* out, hipStream_t s,
                    float4* writeback, const float* rt_dev, const int* done, double* wsum) {
    const int groups = (n + 63) / 64;
    // waves per 64 queries.  The scan is a chain of dependent steps per wave (boxes -> leaf boxes -> points), so its run time is
    // that chain's latency: splitting the candidate leaves of a query group over 4-8 waves shortens the chain even when the
    // chip is already full (measured: 437k queries 1 -> 4 waves 1.75x faster, 40k queries 4 -> 8 waves +4 %, 16 waves slower).
    int nparts = 4;
    while (nparts < 8 && groups * nparts < 4096) nparts <<= 1;
    static const int forced = [] { const char* e = dev_env("FGOICP_NN_PARTS"); const int v = e ? std::atoi(e) : 0; return v; }();  // tuning knob
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8 || forced == 16) nparts = forced;
    // candidate leaves claimed dynamically by the waves of a block while the grid does not fill the device (the scan then lasts as long as its
    // slowest block: 40k points, -3 %), dealt round-robin when it does (437k / 1M points: the LDS claims cost 1-2 % and buy nothing)
    static const int dyn_env = [] { const char* e = dev_env("FGOICP_NN_CLAIM"); return e ? std::atoi(e) : -1; }();  // tuning knob / A-B: 0 / 1 force
    const int dyn = dyn_env >= 0 ? dyn_env : (groups * nparts <= 8192 ? 1 : 0);
    const int flat = [] { const char* e = dev_env("FGOICP_NN_FLAT"); return e ? std::atoi(e) : 0; }();  // tuning knob / A-B (read per launch): 1 = flat form of the walk for targets of <= 2048 leaves (measured slower: off)
    if (want_index) hipLaunchKernelGGL(nn_scan_kernel<1>, dim3(groups), dim3(64 * nparts), 0, s, pts, n, t, lut, g, make_rt(R9, t3), apply, tgt, nt, seed_idx, skip_lb, skip_u, out, writeback, rt_dev, done, wsum, dyn, flat);
    else hipLaunchKernelGGL(nn_scan_kernel<0>, dim3(groups), dim3(64 * nparts), 0, s, pts, n, t, lut, g, make_rt(R9, t3), apply, tgt, nt, seed_idx, skip_lb, skip_u, out, writeback, rt_dev, done, wsum, dyn, flat);
}

void launch_nn_scan_dual(const float4* ptsA, const float* RA9, const float* tA3, int applyA, const float4* ptsB, const float* RB9, const float* tB3, int n, const BvhView& t,
                         const float* lut, const LutGeom& g, const float4* tgt, int nt, const uint32_t* seed_idx, const float* skip_lbA, const uint32_t* skip_uA,
                         const float* skip_lbB, const uint32_t* skip_uB, uint32_t* out_idx, uint32_t* out_min, float4* writeback, double* wsumA, double* wsumB, hipStream_t s) {
    const int groups = (n + 63) / 64;
    int nparts = 4;
    while (nparts < 8 && groups * nparts < 4096) nparts <<= 1;
    static const int forced = [] { const char* e = dev_env("FGOICP_NN_PARTS"); const int v = e ? std::atoi(e) : 0; return v; }();  // tuning knob
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8 || forced == 16) nparts = forced;
    static const int dyn_env = [] { const char* e = dev_env("FGOICP_NN_CLAIM"); return e ? std::atoi(e) : -1; }();  // tuning knob / A-B
    const int dyn = dyn_env >= 0 ? dyn_env : (groups * nparts <= 8192 ? 1 : 0);
    hipLaunchKernelGGL(nn_scan_dual_kernel, dim3(groups), dim3(64 * nparts), 0, s, ptsA, make_rt(RA9, tA3), applyA, ptsB, make_rt(RB9, tB3), n, t, lut, g, tgt, nt, seed_idx,
                       skip_lbA, skip_uA, skip_lbB, skip_uB, out_idx, out_min, writeback, wsumA, wsumB, dyn);
}

void launch_nn_prep(const float4* pts, int n, const float* lut, const LutGeom& g, const float* R9, const float* t3, int apply, const float4* tgt, int nt,
                    const uint32_t* seed_idx, const float* box6, float* ub_out, float* lb_out, hipStream_t s) {
    const float4 lo = make_float4(box6[0], box6[2], box6[4], 0.f), hi = make_float4(box6[1], box6[3], box6[5], 0.f);
    hipLaunchKernelGGL(nn_prep_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, pts, n, lut, g, make_rt(R9, t3), apply, tgt, nt, seed_idx, lo, hi, ub_out, lb_out);
}

// `scratch` must hold as many floats as the padded LUT; it receives the coarse pass.
void launch_lut_build_scan(const BvhView& t, const LutGeom& g, float* scratch, float* lut_padded, hipStream_t s, uint32_t* lut_idx) {
    auto cdiv = [](size_t a, size_t b) { return (a + b - 1) / b; };
    const size_t ncx = cdiv(g.dx, kLutCoarse), ncy = cdiv(g.dy, kLutCoarse), ncz = cdiv(g.dz, kLutCoarse);
    const size_t cwaves = cdiv(ncx, 4) * cdiv(ncy, 4) * cdiv(ncz, 4);
    hipLaunchKernelGGL(lut_build_scan_coarse_kernel, dim3((unsigned)cdiv(cwaves, kBlock / 64)), dim3(kBlock), 0, s, t, g, scratch);
    const size_t waves = cdiv(g.px, 4) * cdiv(g.py, 4) * cdiv(g.pz, 4);
    hipLaunchKernelGGL(lut_build_scan_kernel, dim3((unsigned)cdiv(waves, kBlock / 64)), dim3(kBlock), 0, s, t, g, scratch, lut_padded, lut_idx);
}

#ifdef FGOICP_SCAN_STATS
}  // namespace fgoicp
extern "C" int fgoicp_debug_scan_times(unsigned long long* out, int nblocks) {
    using namespace fgoicp;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scan_times), sizeof(unsigned long long) * 4 * (size_t)nblocks) != hipSuccess;
}
extern "C" int fgoicp_debug_scan_stats(unsigned long long* out8, int reset) {
    using namespace fgoicp;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_scan_stats), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_scan_stats), z, sizeof(z)); }
    return 0;
}
namespace fgoicp {
#endif

int reduce_blocks_for(int n) {
    int b = (n + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    return b > 1024 ? 1024 : b;
}

void launch_sum_f32_as_f64(const uint32_t* bits, int n, double* bp, int nblocks, hipStream_t s, const int* done) {
    hipLaunchKernelGGL(sum_f32_kernel, dim3(nblocks), dim3(kBlock), 0, s, bits, n, bp, done);
}

void launch_sum_partials(const double* bp, int nblocks, int width, double* out, hipStream_t s) {
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64 * width), 0, s, bp, nblocks, width, out);
}

void launch_transform_inplace(float4* pts, int n, const float* R9, const float* t3, hipStream_t s) {
    hipLaunchKernelGGL(transform_inplace_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, pts, n, make_rt(R9, t3));
}

void launch_icp_sums(const float4* work, const float4* tgt, const uint32_t* idx, int n, int nt, const unsigned char* use, double* bp, int nblocks,
                     hipStream_t s, const int* done) {
    hipLaunchKernelGGL(icp_sums_kernel, dim3(nblocks), dim3(kBlock), 0, s, work, tgt, idx, n, nt, use, bp, done);
}

void launch_icp_centroids(const double* bp, int nblocks, int ns, float* cen_dev, float* cen_host, hipStream_t s) {
    hipLaunchKernelGGL(icp_centroids_kernel, dim3(1), dim3(384), 0, s, bp, nblocks, ns, cen_dev, cen_host);
}

void launch_icp_cov(const float4* work, const float4* tgt, const uint32_t* idx, int n, int nt, const float* cen_dev, const unsigned char* use,
                    double* bp, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL(icp_cov_kernel, dim3(nblocks), dim3(kBlock), 0, s, work, tgt, idx, n, nt, cen_dev, use, bp);
}

void launch_icp_cov_cen(const float4* work, const float4* tgt, const uint32_t* idx, int n, int nt, const double* sums_bp, int sums_nblocks, int from_waves,
                        float* cen_out, double* bp, int nblocks, hipStream_t s, const int* done) {
    hipLaunchKernelGGL(icp_cov_cen_kernel, dim3(nblocks), dim3(kBlock), 0, s, work, tgt, idx, n, nt, sums_bp, sums_nblocks, from_waves, cen_out, bp, done);
}

#ifdef FGOICP_DEV_KNOBS   // the device-resident ICP loop (measured slower than the host loop): development build only
void launch_icp_init(IcpDevState* st, const float* R9, const float* t3, int max_iter, float thr, IcpHostResult* res, hipStream_t s) {
    hipLaunchKernelGGL(icp_init_kernel, dim3(1), dim3(64), 0, s, st, make_rt(R9, t3), max_iter, thr, res);
}

void launch_icp_step(IcpDevState* st, const double* bp_cov, int nb_cov, const double* bp_sse, int nb_sse, const float* cen, IcpHostResult* res, hipStream_t s) {
    hipLaunchKernelGGL(icp_step_kernel, dim3(1), dim3(640), 0, s, st, bp_cov, nb_cov, bp_sse, nb_sse, cen, res);
}
#endif

// trimmed bounds of a window: out_ub[row] / out_lb[row] from the row's k smallest e (trim_rows_kernel)
void launch_trim_rows(const float* evals, size_t erow, int n, int k, int rows, const float* row_span, float* out_ub, float* out_lb, hipStream_t s,
                      int samp_shift, int margin, unsigned long long* stat) {
    if (samp_shift > 0)  // one pass per row, steered by the sample the bounds kernel left behind each row (trim_store)
        hipLaunchKernelGGL(trim_rows_sampled_kernel, dim3(rows), dim3(kTrimThreads), 0, s, evals, erow, n, k, samp_shift, margin, row_span, out_ub, out_lb, stat);
    else
        hipLaunchKernelGGL(trim_rows_kernel, dim3(rows), dim3(kTrimThreads), 0, s, evals, erow, n, k, row_span, out_ub, out_lb);
}

// ONE row: out[0] = sum of the k smallest of vals[0..n), sel_info = {bits of the k-th smallest, copies of it among the k}
void launch_trim_select(const float* vals, int n, int k, float* out, uint32_t* sel_info, uint32_t* wide_scratch, hipStream_t s) {
    if (wide_scratch && n >= 32768) {  // a long row: spread the selection over the device
        const int nb = std::min(kSelWideBlocks, (n + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(select_wide_init_kernel, dim3(1), dim3(kBlock), 0, s, wide_scratch, k);
        for (int lvl = 0; lvl < 3; ++lvl) {
            hipLaunchKernelGGL(select_wide_hist_kernel, dim3(nb), dim3(kBlock), 0, s, vals, n, lvl, wide_scratch);
            hipLaunchKernelGGL(select_wide_pick_kernel, dim3(1), dim3(kBlock), 0, s, lvl, wide_scratch);
        }
        hipLaunchKernelGGL(select_wide_sum_kernel, dim3(nb), dim3(kBlock), 0, s, vals, n, wide_scratch);
        hipLaunchKernelGGL(select_wide_final_kernel, dim3(1), dim3(64), 0, s, wide_scratch, nb, out, sel_info);
        return;
    }
    hipLaunchKernelGGL(trim_select_kernel, dim3(1, 1), dim3(kBlock), 0, s, vals, (size_t)0, 1, 1, n, k, out, (float*)nullptr, sel_info);
}

void launch_icp_inliers(const float4* work, const float4* tgt, const uint32_t* idx, int n, int nt, int k, float* d2, uint32_t* sel_info,
                        uint32_t* equal_count, const uint32_t* orig_of_slot, unsigned char* use, uint32_t* wide_scratch, hipStream_t s) {
    const int nb = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(icp_corr_d2_kernel, dim3(nb), dim3(kBlock), 0, s, work, tgt, idx, n, nt, d2, equal_count);
    launch_trim_select(d2, n, k, nullptr, sel_info, wide_scratch, s);
    hipLaunchKernelGGL(icp_inlier_mask_kernel, dim3(nb), dim3(kBlock), 0, s, d2, n, sel_info, use, equal_count);
    hipLaunchKernelGGL(icp_inlier_ties_kernel, dim3(std::max(1, std::min(256, (n + 1023) / 1024))), dim3(1024), 0, s, n, sel_info, equal_count, orig_of_slot, use);
}

}  // namespace fgoicp
