// What a SIMD of gfx950 issues per clock for the instruction kinds the dense bounds kernel is made of: plain fp32 fma / mul / add,
// their PACKED forms (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two fp32 lanes of a 64-bit register pair per instruction), fp64 add,
// v_sqrt_f32, v_cvt.  One number decides whether pair-wise packing of the kernel's arithmetic can pay: if a packed
// instruction issues at the rate of a plain one, the packed kernel does the same flops in about half the VALU issue slots.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

constexpr int kIters = 4096;

// 16 independent accumulators per lane in every variant, 16 "element operations" per loop trip
template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float* out, float seed) {
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = seed + (float)(threadIdx.x + i);
    const float m = 1.0000001f, c = 1e-9f;
    float2v p[8], mm = {m, m}, cc = {c, c};
    for (int i = 0; i < 8; ++i) p[i] = float2v{a[2 * i], a[2 * i + 1]};
    double d[8];
    for (int i = 0; i < 8; ++i) d[i] = (double)a[i];
    for (int it = 0; it < kIters; ++it) {
        if (KIND == 0) {  // 16 x v_fma_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        } else if (KIND == 1) {  // 8 x v_pk_fma_f32 = 16 element fmas
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(mm), "v"(cc));
        } else if (KIND == 2) {  // 16 x v_mul_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        } else if (KIND == 3) {  // 8 x v_pk_mul_f32
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(mm));
        } else if (KIND == 4) {  // 16 x v_add_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (KIND == 5) {  // 8 x v_pk_add_f32
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cc));
        } else if (KIND == 6) {  // 8 x v_add_f64 (counted as 8 element operations)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"((double)c));
        } else if (KIND == 7) {  // 16 x v_sqrt_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
        } else if (KIND == 8) {  // 16 x v_cvt_f64_f32 + nothing else: conversions feeding the fp64 accumulation (8 per trip)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
        } else if (KIND == 9) {  // 16 x v_max_f32
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (KIND == 10) {  // 16 x v_floor_f32 (1.8 weights: floor / fract / rndne live on this unit)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
        } else if (KIND == 11) {  // 16 x v_cvt_i32_f32 + back: address arithmetic
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
        } else if (KIND == 12) {  // 16 x v_mad_u32_u24 (addresses)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a[i];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y + (float)d[i];
    if (s == 123.456f) out[0] = s;
}

template <int KIND>
void run(const char* name, int elems_per_trip, int instr_per_trip, float* out, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd;  // 256 CUs x 4 SIMDs x waves_per_simd waves = blocks x 4 waves
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double s = ms * 1e-3 / 5.0;
    const double waves = (double)blocks * 4.0;
    const double winstr = waves * kIters * instr_per_trip;           // wave-instructions
    const double elem = waves * 64.0 * kIters * elems_per_trip;      // lane-level element operations
    // per SIMD: 1024 SIMDs
    std::printf("%-14s %d waves/SIMD: %8.1f us  %7.2f G wave-instr/s  = %.3f instr / SIMD / ns   %8.2f T elem-op/s\n", name, waves_per_simd, s * 1e6, winstr / s * 1e-9, winstr / s * 1e-9 / 1024.0,
                elem / s * 1e-12);
}

int main() {
    float* out;
    CHK(hipMalloc(&out, 4));
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", 16, 16, out, w);
        run<1>("v_pk_fma_f32", 16, 8, out, w);
        run<2>("v_mul_f32", 16, 16, out, w);
        run<3>("v_pk_mul_f32", 16, 8, out, w);
        run<4>("v_add_f32", 16, 16, out, w);
        run<5>("v_pk_add_f32", 16, 8, out, w);
        run<6>("v_add_f64", 8, 8, out, w);
        run<7>("v_sqrt_f32", 16, 16, out, w);
        run<8>("v_cvt_f64_f32", 8, 8, out, w);
        run<9>("v_max_f32", 16, 16, out, w);
        run<10>("v_floor_f32", 16, 16, out, w);
        run<11>("v_cvt_i32_f32", 16, 16, out, w);
        run<12>("v_mad_u32_u24", 16, 16, out, w);
    }
    return 0;
}
