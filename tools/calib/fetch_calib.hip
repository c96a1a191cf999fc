// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access pattern of the bounds kernel (divergent 16-byte gathers), as
// MI355X_MICROARCH.md §HBM asks before trusting an absolute.  Known byte counts:
//   stream:      every byte of a 1 GiB table once, 16 B per lane, coalesced              (guide: FETCH_SIZE reads 1/2 of the bytes)
//   gather128:   ONE 16-byte load from every 128-byte line of the table, random order     (8 Mi lines touched once)
//   gather64:    ONE 16-byte load from every 64-byte half line, random order              (16 Mi half lines touched once)
//   gather32:    ONE 16-byte load from every 32-byte sector, random order
// If the memory side moves whole 128-B lines, gather64 and gather32 fetch what gather128 fetches; if it moves 64-B (32-B)
// sectors, FETCH_SIZE doubles from gather128 to gather64 (and again to gather32).
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void stream_kernel(const float4* __restrict__ t, size_t n16, float* out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const float4 v = t[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
template <int GRAN>  // one 16-byte load per GRAN bytes of the table
__global__ void gather_kernel(const char* __restrict__ t, size_t nunits, float* out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nunits; i += (size_t)gridDim.x * blockDim.x) {
        const size_t v = (i * 2654435761ull) & (nunits - 1);  // odd multiplier mod 2^k: a bijection, every unit exactly once
        const float4 x = *reinterpret_cast<const float4*>(t + v * GRAN);
        acc += x.x + x.y + x.z + x.w;
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    char* t; float* out;
    CHK(hipMalloc(&t, bytes)); CHK(hipMalloc(&out, 4));
    CHK(hipMemset(t, 1, bytes));
    char* evict; CHK(hipMalloc(&evict, bytes));
    for (int rep = 0; rep < 2; ++rep) {
        CHK(hipMemset(evict, rep, bytes));  // push the table out of the 256 MiB Infinity Cache between kernels
        hipLaunchKernelGGL(stream_kernel, dim3(8192), dim3(256), 0, 0, (const float4*)t, bytes / 16, out);
        CHK(hipMemset(evict, rep + 2, bytes));
        hipLaunchKernelGGL(gather_kernel<128>, dim3(8192), dim3(256), 0, 0, t, bytes / 128, out);
        CHK(hipMemset(evict, rep + 4, bytes));
        hipLaunchKernelGGL(gather_kernel<64>, dim3(8192), dim3(256), 0, 0, t, bytes / 64, out);
        CHK(hipMemset(evict, rep + 6, bytes));
        hipLaunchKernelGGL(gather_kernel<32>, dim3(8192), dim3(256), 0, 0, t, bytes / 32, out);
        CHK(hipDeviceSynchronize());
    }
    std::printf("table %zu bytes: stream reads all of it; gather128/64/32 issue %zu / %zu / %zu 16-byte loads\n", bytes, bytes / 128, bytes / 64, bytes / 32);
    return 0;
}
