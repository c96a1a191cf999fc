// How fast gfx950 gets through workgroups that have (almost) nothing to do: the floor under a work item of the bounds kernel that ends
// early (fgoicp_bounds_submit_cut).  Kernels: EMPTY (s_endpgm after the kernarg load), ONE dependent scalar load, TWO dependent scalar
// loads + a vector load + a store (the shape of the early-exit path), for workgroups of 64 / 256 threads.
//   hipcc --offload-arch=gfx950 -O3 -o dispatch_rate dispatch_rate.hip && ./dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int KIND, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const unsigned* __restrict__ a, const unsigned* __restrict__ b, unsigned* __restrict__ out, unsigned n) {
    const unsigned slot = blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    if (KIND == 0) { if (n == 0xFFFFFFFFu) out[0] = slot; return; }
    const unsigned i = a[slot];                       // scalar: block-uniform
    if (KIND == 1) { if (i == 0xFFFFFFFFu) out[0] = slot; return; }
    const unsigned j = b[i % n];                      // dependent
    const unsigned h = __builtin_nontemporal_load(&out[n + (i % n)]);
    if (KIND == 2) { if ((j | h) == 0xFFFFFFFFu) out[0] = slot; return; }
    if ((threadIdx.x & 63) == 0) out[slot % n] = j + h;  // KIND 3: + a store
}

template <int KIND, int THREADS>
static void run(const char* name, const unsigned* a, const unsigned* b, unsigned* out, unsigned items, unsigned n) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const unsigned grid = items / (THREADS / 64);
    hipLaunchKernelGGL((k<KIND, THREADS>), dim3(grid), dim3(THREADS), 0, 0, a, b, out, n);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<KIND, THREADS>), dim3(grid), dim3(THREADS), 0, 0, a, b, out, n);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("%-44s threads/WG %3d: %8.1f us per launch of %u waves = %.3f ns per wave = %.2f waves per ns\n", name, THREADS, ms * 1e3 / 5, items, ms * 1e6 / 5 / items, items / (ms * 1e6 / 5));
}

int main() {
    const unsigned items = 1u << 20, n = 8192;
    unsigned *a, *b, *out;
    CHK(hipMalloc(&a, sizeof(unsigned) * items)); CHK(hipMalloc(&b, sizeof(unsigned) * n)); CHK(hipMalloc(&out, sizeof(unsigned) * (items + 2 * n)));
    unsigned* h = (unsigned*)std::malloc(sizeof(unsigned) * items);
    for (unsigned i = 0; i < items; ++i) h[i] = (i * 2654435761u) >> 8;
    CHK(hipMemcpy(a, h, sizeof(unsigned) * items, hipMemcpyHostToDevice));
    CHK(hipMemset(b, 0, sizeof(unsigned) * n)); CHK(hipMemset(out, 0, sizeof(unsigned) * (items + 2 * n)));
    run<0, 64>("empty", a, b, out, items, n);
    run<1, 64>("one scalar load", a, b, out, items, n);
    run<2, 64>("two dependent scalar loads + vector load", a, b, out, items, n);
    run<3, 64>("... + a store", a, b, out, items, n);
    run<0, 256>("empty", a, b, out, items, n);
    run<1, 256>("one scalar load", a, b, out, items, n);
    run<2, 256>("two dependent scalar loads + vector load", a, b, out, items, n);
    run<3, 256>("... + a store", a, b, out, items, n);
    run<3, 512>("... + a store", a, b, out, items, n);
    run<3, 1024>("... + a store", a, b, out, items, n);
    return 0;
}
