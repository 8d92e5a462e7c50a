// ubench_hbm.hip -- the HBM bandwidth this GPU actually delivers: the second denominator of bench.py's rooflines
// ("confirm on the box with a device-to-device copy/triad micro-bench and use the measured peak too").
//   copy   b[i] = a[i]                 16 B read + 16 B written per float4
//   triad  c[i] = a[i] + s * b[i]      32 B read + 16 B written per float4
//   read   sum of a[i]                 16 B read per float4 (what K1's geometry fetch looks like)
// Arrays are 1 GiB each (far beyond the 256 MiB Infinity Cache); 16-byte accesses, grid-stride, 2048 workgroups.
// Build (hipcc cross-compiles without a GPU) and run:
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_hbm tools/ubench_hbm.hip && tools/ubench_hbm [--json]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ void __launch_bounds__(256) k_triad(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, float s, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 x = a[i], y = b[i];
        c[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
    }
}
__global__ void __launch_bounds__(256) k_read(const float4* __restrict__ a, float* __restrict__ out, size_t n)
{
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 x = a[i];
        acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

int main(int argc, char** argv)
{
    const bool json = argc > 1 && !strcmp(argv[1], "--json");
    const size_t n = (size_t)1 << 26;   // float4s per array: 1 GiB
    float4 *a = nullptr, *b = nullptr, *c = nullptr;
    float* out = nullptr;
    CHK(hipMalloc(reinterpret_cast<void**>(&a), n * 16)); CHK(hipMalloc(reinterpret_cast<void**>(&b), n * 16));
    CHK(hipMalloc(reinterpret_cast<void**>(&c), n * 16)); CHK(hipMalloc(reinterpret_cast<void**>(&out), 4));
    CHK(hipMemset(a, 0, n * 16)); CHK(hipMemset(b, 0, n * 16)); CHK(hipMemset(c, 0, n * 16));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int grid = 2048, reps = 10;
    double best[3] = {0, 0, 0};
    for (int kind = 0; kind < 3; ++kind)
        for (int r = 0; r < reps + 2; ++r) {
            CHK(hipEventRecord(e0, nullptr));
            if (kind == 0) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, nullptr, a, b, n);
            else if (kind == 1) hipLaunchKernelGGL(k_triad, dim3(grid), dim3(256), 0, nullptr, a, b, c, 0.5f, n);
            else hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, nullptr, a, out, n);
            CHK(hipEventRecord(e1, nullptr));
            CHK(hipEventSynchronize(e1));
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            const double bytes = (kind == 0 ? 32.0 : kind == 1 ? 48.0 : 16.0) * (double)n;
            if (r >= 2) best[kind] = std::max(best[kind], bytes / (ms * 1e-3) / 1e9);
        }
    if (json) printf("{\"copy_GBps\": %.1f, \"triad_GBps\": %.1f, \"read_GBps\": %.1f, \"bytes_per_array\": %zu}\n", best[0], best[1], best[2], n * 16);
    else printf("HBM bandwidth (best of %d, 1 GiB arrays, 16-byte accesses): copy %.0f GB/s, triad %.0f GB/s, read-only %.0f GB/s\n", reps, best[0], best[1], best[2]);
    return 0;
}
