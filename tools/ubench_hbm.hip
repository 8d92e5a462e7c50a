// ubench_hbm.hip -- the HBM bandwidth this GPU actually delivers: the second denominator of bench.py's rooflines
// ("confirm on the box with a device-to-device copy/triad micro-bench and use the measured peak too").
//   copy     b[i] = a[i], plain grid-stride loop, one 16-byte load in flight per lane (the round-3 yardstick: it
//            under-reads the part -- a lane that has one load outstanding cannot cover the HBM latency)
//   copy4    the same bytes with FOUR independent 16-byte loads in flight per lane (64 B) and non-temporal stores,
//            a workgroup walking contiguous 16-KiB chunks: what a streaming kernel written for this GPU looks like
//   copy8    eight loads in flight per lane (128 B)
//   triad    c[i] = a[i] + s * b[i]      32 B read + 16 B written per float4 (4 x unrolled)
//   read     sum of a[i]                 16 B read per float4, 4 x unrolled (what K1's geometry fetch looks like)
// "peak" = the best copy variant (read + write bytes per second).  Arrays are 1 GiB each (far beyond the 256 MiB Infinity
// Cache); grids of 8 / 16 / 32 workgroups per CU are tried and the best time is kept.
// Build (hipcc cross-compiles without a GPU) and run:
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_hbm tools/ubench_hbm.hip && tools/ubench_hbm [--json]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_copy(const v4f* __restrict__ a, v4f* __restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
// U independent loads per lane, then U non-temporal stores; a workgroup owns chunks of 256 * U consecutive float4s
template <int U>
__global__ void __launch_bounds__(256) k_copy_u(const v4f* __restrict__ a, v4f* __restrict__ b, size_t n)
{
    const size_t chunk = (size_t)256 * U;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            v[u] = i < n ? __builtin_nontemporal_load(&a[i]) : (v4f)(0.0f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (i < n) __builtin_nontemporal_store(v[u], &b[i]);
        }
    }
}
__global__ void __launch_bounds__(256) k_triad(const v4f* __restrict__ a, const v4f* __restrict__ b, v4f* __restrict__ c, float s, size_t n)
{
    const size_t chunk = (size_t)256 * 4;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
        v4f x[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            x[u] = i < n ? __builtin_nontemporal_load(&a[i]) : (v4f)(0.0f);
            y[u] = i < n ? __builtin_nontemporal_load(&b[i]) : (v4f)(0.0f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (i < n) __builtin_nontemporal_store(x[u] + s * y[u], &c[i]);
        }
    }
}
__global__ void __launch_bounds__(256) k_read(const v4f* __restrict__ a, float* __restrict__ out, size_t n)
{
    v4f acc = (v4f)(0.0f);
    const size_t chunk = (size_t)256 * 4;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
        v4f x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            x[u] = i < n ? __builtin_nontemporal_load(&a[i]) : (v4f)(0.0f);
        }
        acc += (x[0] + x[1]) + (x[2] + x[3]);
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

int main(int argc, char** argv)
{
    const bool json = argc > 1 && !strcmp(argv[1], "--json");
    const size_t n = (size_t)1 << 26;   // float4s per array: 1 GiB
    v4f *a = nullptr, *b = nullptr, *c = nullptr;
    float* out = nullptr;
    CHK(hipMalloc(reinterpret_cast<void**>(&a), n * 16)); CHK(hipMalloc(reinterpret_cast<void**>(&b), n * 16));
    CHK(hipMalloc(reinterpret_cast<void**>(&c), n * 16)); CHK(hipMalloc(reinterpret_cast<void**>(&out), 4));
    CHK(hipMemset(a, 0, n * 16)); CHK(hipMemset(b, 0, n * 16)); CHK(hipMemset(c, 0, n * 16));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const int grids[3] = {cus * 8, cus * 16, cus * 32};
    const int reps = 6;
    enum { COPY, COPY4, COPY8, TRIAD, READ, KINDS };
    double best[KINDS] = {0, 0, 0, 0, 0};
    for (int kind = 0; kind < KINDS; ++kind)
        for (int gi = 0; gi < 3; ++gi)
            for (int r = 0; r < reps + 2; ++r) {
                const int grid = grids[gi];
                CHK(hipEventRecord(e0, nullptr));
                switch (kind) {
                case COPY: hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, nullptr, a, b, n); break;
                case COPY4: hipLaunchKernelGGL(k_copy_u<4>, dim3(grid), dim3(256), 0, nullptr, a, b, n); break;
                case COPY8: hipLaunchKernelGGL(k_copy_u<8>, dim3(grid), dim3(256), 0, nullptr, a, b, n); break;
                case TRIAD: hipLaunchKernelGGL(k_triad, dim3(grid), dim3(256), 0, nullptr, a, b, c, 0.5f, n); break;
                default: hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, nullptr, a, out, n); break;
                }
                CHK(hipEventRecord(e1, nullptr));
                CHK(hipEventSynchronize(e1));
                float ms = 0;
                CHK(hipEventElapsedTime(&ms, e0, e1));
                const double bytes = (kind == TRIAD ? 48.0 : kind == READ ? 16.0 : 32.0) * (double)n;
                if (r >= 2) best[kind] = std::max(best[kind], bytes / (ms * 1e-3) / 1e9);
            }
    const double peak = std::max(best[COPY], std::max(best[COPY4], best[COPY8]));
    if (json)
        printf("{\"peak_GBps\": %.1f, \"copy_GBps\": %.1f, \"copy4_GBps\": %.1f, \"copy8_GBps\": %.1f, \"triad_GBps\": %.1f, \"read_GBps\": %.1f, "
               "\"bytes_per_array\": %zu, \"note\": \"copy = one 16-byte load in flight per lane (the round-3 yardstick); copy4 / copy8 = 64 / 128 bytes in flight per "
               "lane, non-temporal stores; peak = the best copy\"}\n",
               peak, best[COPY], best[COPY4], best[COPY8], best[TRIAD], best[READ], n * 16);
    else
        printf("HBM bandwidth (best of %d x 3 grids, 1 GiB arrays): copy %.0f GB/s (1 load in flight), copy4 %.0f, copy8 %.0f, triad %.0f, read-only %.0f\n",
               reps, best[COPY], best[COPY4], best[COPY8], best[TRIAD], best[READ]);
    return 0;
}
