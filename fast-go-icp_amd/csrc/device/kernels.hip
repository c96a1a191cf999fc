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

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Sum K doubles per thread over the 256-thread block; the result is valid in threads 0..K-1
// (thread k holds component k).  Fixed order: shuffle tree inside a wave, then waves 0..3.
template <int K>
__device__ __forceinline__ double block_sum(const double (&v)[K], double* lds /* [4*K] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double r = wave_sum(v[k]);
        if (lane == 0) lds[wave * K + k] = r;
    }
    __syncthreads();
    double out = 0.0;
    if (threadIdx.x < K) out = ((lds[threadIdx.x] + lds[K + threadIdx.x]) + lds[2 * K + threadIdx.x]) + lds[3 * K + threadIdx.x];
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
    const float* p;   // first texel (x0, y0, z0) in the padded LUT
    float a, b, c;    // interpolation weights
};
__device__ __forceinline__ TexAddr lut_address(const float* __restrict__ lut, const LutGeom& g, float qx, float qy, float qz) {
    const float x = (qx + g.off_x) * g.scale;
    const float y = (qy + g.off_y) * g.scale;
    const float z = (qz + g.off_z) * g.scale;
    int ix, iy, iz;
    TexAddr t;
    tex_axis(x, g.dx, g.quantize, ix, t.a);
    tex_axis(y, g.dy, g.quantize, iy, t.b);
    tex_axis(z, g.dz, g.quantize, iz, t.c);
    t.p = lut + ((size_t)iz * g.py + iy) * (size_t)g.px + ix;
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
    const TexAddr t = lut_address(lut, g, qx, qy, qz);
    const size_t sy = (size_t)g.px, sz = (size_t)g.px * g.py;
    return lut_blend(t, *(const float2u*)(t.p), *(const float2u*)(t.p + sy), *(const float2u*)(t.p + sz), *(const float2u*)(t.p + sz + sy));
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
        ta[k] = lut_address(lut, g, rx + tn.x, ry + tn.y, rz + tn.z);  // :34, :323-325
    }
#pragma unroll
    for (int k = 0; k < P; ++k) {
        v00[k] = *(const float2u*)(ta[k].p);
        v10[k] = *(const float2u*)(ta[k].p + sy);
        v01[k] = *(const float2u*)(ta[k].p + sz);
        v11[k] = *(const float2u*)(ta[k].p + sz + sy);
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

// One block (one wave) per subcube: fixed-order sum of its chunk partials, rounded once to fp32.
__global__ __launch_bounds__(64) void bounds_finalize_kernel(const double2* __restrict__ partials, int nchunk, int total,
                                                             float* __restrict__ out_lb, float* __restrict__ out_ub) {
    const int s = blockIdx.x;
    if (s >= total) return;
    const double2* row = partials + (size_t)s * nchunk;
    double u = 0.0, l = 0.0;
    for (int c = threadIdx.x; c < nchunk; c += 64) {
        const double2 v = row[c];
        u += v.x;
        l += v.y;
    }
    u = wave_sum(u);
    l = wave_sum(l);
    if (threadIdx.x == 0) {
        out_ub[s] = (float)u;
        out_lb[s] = (float)l;
    }
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

__global__ __launch_bounds__(kBlock) void fill_u32_kernel(uint32_t* p, uint32_t v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) p[i] = v;
}

// ---------------------------------------------------------------------------------------------
// Deterministic fp64 sums (stand-ins for thrust::reduce at registration.cu:79-80, icp3d.cu:152-166)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void sum_f32_kernel(const uint32_t* __restrict__ bits, int n, double* __restrict__ bp) {
    __shared__ double red[4];
    double acc[1] = {0.0};
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) acc[0] += (double)__uint_as_float(bits[i]);
    const double r = block_sum<1>(acc, red);
    if (threadIdx.x == 0) bp[blockIdx.x] = r;
}

// single block: out[k] = sum over blocks (in order) of bp[b*width + k]
__global__ __launch_bounds__(64) void sum_partials_kernel(const double* __restrict__ bp, int nblocks, int width, double* __restrict__ out) {
    const int k = threadIdx.x;
    if (k >= width) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += bp[(size_t)b * width + k];
    out[k] = s;
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
                                                          const uint32_t* __restrict__ idx, int n, double* __restrict__ bp) {
    __shared__ double red[24];
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 a = work[i];
        const float4 c = tgt[idx[i]];
        acc[0] += (double)a.x; acc[1] += (double)a.y; acc[2] += (double)a.z;
        acc[3] += (double)c.x; acc[4] += (double)c.y; acc[5] += (double)c.z;
    }
    const double r = block_sum<6>(acc, red);
    if (threadIdx.x < 6) bp[(size_t)blockIdx.x * 6 + threadIdx.x] = r;
}

// kernCentralize x2 + kernOuterProduct + reduce (icp3d.cu:38-52, :158-166), fused: per-point fp32
// centring and products exactly as the reference, fp64 accumulation.  Output in glm::mat3 order:
// ABt[col][row] = sum a[row]*b[col]  (glm::outerProduct(c, r): m[i] = c * r[i]).
struct Centroids { float s[3]; float c[3]; };
__global__ __launch_bounds__(kBlock) void icp_cov_kernel(const float4* __restrict__ work, const float4* __restrict__ tgt,
                                                         const uint32_t* __restrict__ idx, int n, Centroids cen,
                                                         double* __restrict__ bp) {
    __shared__ double red[36];
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 p = work[i];
        const float4 q = tgt[idx[i]];
        const float a[3] = {p.x - cen.s[0], p.y - cen.s[1], p.z - cen.s[2]};
        const float b[3] = {q.x - cen.c[0], q.y - cen.c[1], q.z - cen.c[2]};
#pragma unroll
        for (int col = 0; col < 3; ++col)
#pragma unroll
            for (int row = 0; row < 3; ++row) acc[col * 3 + row] += (double)(a[row] * b[col]);
    }
    const double r = block_sum<9>(acc, red);
    if (threadIdx.x < 9) bp[(size_t)blockIdx.x * 9 + threadIdx.x] = r;
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
    dim3 grid((unsigned)nchunk * (unsigned)a.B), block(kBlock);
    switch (P) {
        case 1: hipLaunchKernelGGL(bounds_kernel<1>, grid, block, 0, s, src, ns, lut, g, a, partials, nchunk); break;
        case 2: hipLaunchKernelGGL(bounds_kernel<2>, grid, block, 0, s, src, ns, lut, g, a, partials, nchunk); break;
        case 4: hipLaunchKernelGGL(bounds_kernel<4>, grid, block, 0, s, src, ns, lut, g, a, partials, nchunk); break;
        default: hipLaunchKernelGGL(bounds_kernel<8>, grid, block, 0, s, src, ns, lut, g, a, partials, nchunk); break;
    }
}

void launch_bounds_finalize(const double2* partials, int nchunk, int total, float* out_lb, float* out_ub, hipStream_t s) {
    hipLaunchKernelGGL(bounds_finalize_kernel, dim3(total), dim3(64), 0, s, partials, nchunk, total, out_lb, out_ub);
}

void launch_lut_build(const float4* tgt_shifted, int nt, const LutGeom& g, float* lut_padded, hipStream_t s) {
    const size_t total = (size_t)g.px * g.py * g.pz;
    const size_t per_block = (size_t)kBlock * kLutNodes;
    const unsigned blocks = (unsigned)((total + per_block - 1) / per_block);
    hipLaunchKernelGGL(lut_build_kernel, dim3(blocks), dim3(kBlock), 0, s, tgt_shifted, nt, g, lut_padded);
}

void launch_lut_unpad(const float* lut_padded, const LutGeom& g, float* out, hipStream_t s) {
    hipLaunchKernelGGL(lut_unpad_kernel, dim3(4096), dim3(kBlock), 0, s, lut_padded, g, out);
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

int reduce_blocks_for(int n) {
    int b = (n + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    return b > 1024 ? 1024 : b;
}

void launch_sum_f32_as_f64(const uint32_t* bits, int n, double* bp, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL(sum_f32_kernel, dim3(nblocks), dim3(kBlock), 0, s, bits, n, bp);
}

void launch_sum_partials(const double* bp, int nblocks, int width, double* out, hipStream_t s) {
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, s, bp, nblocks, width, out);
}

void launch_transform_inplace(float4* pts, int n, const float* R9, const float* t3, hipStream_t s) {
    hipLaunchKernelGGL(transform_inplace_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, pts, n, make_rt(R9, t3));
}

void launch_icp_sums(const float4* work, const float4* tgt, const uint32_t* idx, int n, double* bp, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL(icp_sums_kernel, dim3(nblocks), dim3(kBlock), 0, s, work, tgt, idx, n, bp);
}

void launch_icp_cov(const float4* work, const float4* tgt, const uint32_t* idx, int n, const float* centroids6, double* bp,
                    int nblocks, hipStream_t s) {
    Centroids cen;
    for (int i = 0; i < 3; ++i) { cen.s[i] = centroids6[i]; cen.c[i] = centroids6[3 + i]; }
    hipLaunchKernelGGL(icp_cov_kernel, dim3(nblocks), dim3(kBlock), 0, s, work, tgt, idx, n, cen, bp);
}

}  // namespace fgoicp
