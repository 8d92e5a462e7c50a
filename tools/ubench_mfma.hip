// ubench_mfma.hip -- what v_mfma_f32_4x4x1_16B_f32 is, lane for lane, on this GPU, and what it costs beside FP32 vector work.
//
// k_blend (csrc/k_blend.h, -DBL_MFMA) forms the two affine forms of four records for a wave's 64 pixels, and adds a record's
// weighted colour to the pixels' accumulators, on the matrix pipe -- bit-identical to the fmaf chains of the arithmetic contract
//  or not at all.  This program pins the three facts that rests on:
//   1. the operand layout: D[b][i][j] = A[b][i] * B[b][j] + C[b][i][j], 16 blocks b of 4 x 4; A: lane 4b+i; B: lane 4b+j;
//      C / D: lane 4b+j, register i;
//   2. every element is ONE IEEE fused multiply-add (compared bit for bit with fmaf on random operands, including
//      products that cancel against the addend and denormal results);
//   3. the issue cost: a loop of FP32 vector work per wave with / without MFMAs beside it, at 1..8 waves per SIMD.
// Build + run (GPU box): hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/ubench_mfma tools/ubench_mfma.hip && tools/ubench_mfma
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void k_layout(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c /* [64][4] */, float* __restrict__ d /* [64][4] */)
{
    const int l = threadIdx.x;
    v4f acc = {c[l * 4 + 0], c[l * 4 + 1], c[l * 4 + 2], c[l * 4 + 3]};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[l * 4 + i] = acc[i];
}

// the issue-cost loop: per trip `VALU` dependent-free fma pairs on 8 accumulators, and `MF` MFMAs on one accumulator chain
template <int VALU, int MF>
__global__ void __launch_bounds__(256) k_mix(float* __restrict__ out, int trips, float seed)
{
    float x[8];
    for (int k = 0; k < 8; ++k) x[k] = seed + (float)(threadIdx.x + k);
    v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
    const float a = seed * 0.5f, b = 1.0f + seed;
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int v = 0; v < VALU; ++v)
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = __builtin_fmaf(x[k], b, a);
#pragma unroll
        for (int m = 0; m < MF; ++m) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(x[m & 7], b, acc, 0, 0, 0);
    }
    float s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int k = 0; k < 8; ++k) s += x[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VALU, int MF>
static double time_mix(int waves_per_simd, float* d_out)
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int grid = p.multiProcessorCount * waves_per_simd;   // 256 threads = one wave per SIMD of a CU
    const int trips = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mix<VALU, MF>), dim3(grid), dim3(256), 0, 0, d_out, 100, 0.001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mix<VALU, MF>), dim3(grid), dim3(256), 0, 0, d_out, trips, 0.001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)ms * 1e6 / trips / waves_per_simd;   // ns per trip and wave
}

int main()
{
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> U(-4.0f, 4.0f);
    float *da, *db, *dc, *dd;
    hipMalloc(&da, 64 * 4); hipMalloc(&db, 64 * 4); hipMalloc(&dc, 64 * 16); hipMalloc(&dd, 64 * 16);
    long bad_layout = 0, bad_bits = 0, checked = 0;
    for (int it = 0; it < 2000; ++it) {
        std::vector<float> a(64), b(64), c(256), d(256);
        for (auto& v : a) v = U(rng);
        for (auto& v : b) v = U(rng);
        for (auto& v : c) v = U(rng);
        if (it % 5 == 1) for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) c[l * 4 + i] = -a[(l & ~3) + i] * b[l] * (1.0f + 1e-7f * (float)i);   // cancellation
        if (it % 5 == 2) { for (auto& v : a) v *= 1e-20f; for (auto& v : b) v *= 1e-19f; for (auto& v : c) v *= 1e-39f; }                               // denormal results
        if (it % 5 == 3) for (auto& v : c) v *= 1e6f;                                                                                                    // addend dominates
        hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
        hipMemcpy(dc, c.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) {
                const int blk = l >> 2;
                const float want = std::fmaf(a[blk * 4 + i], b[l], c[l * 4 + i]);
                uint32_t w, g;
                std::memcpy(&w, &want, 4); std::memcpy(&g, &d[l * 4 + i], 4);
                ++checked;
                if (w != g) { ++bad_bits; if (std::fabs(want - d[l * 4 + i]) > 1e-3f * (1.0f + std::fabs(want))) ++bad_layout; if (bad_bits < 6) printf("  it %d lane %d reg %d: fmaf %.9g (%08x)  mfma %.9g (%08x)\n", it, l, i, want, w, d[l * 4 + i], g); }
            }
    }
    printf("layout D[lane 4b+j][reg i] = A[lane 4b+i] * B[lane 4b+j] + C[lane 4b+j][reg i]: %ld of %ld elements off by more than rounding\n", bad_layout, checked);
    printf("bitwise equal to fmaf: %ld of %ld elements differ\n", bad_bits, checked);
    float* d_out;
    hipMalloc(&d_out, (size_t)256 * 8 * 256 * 4 * 2);
    printf("ns per loop trip and wave (256 CUs x 4 SIMDs busy); a trip = V x 8 v_fma_f32 [+ M v_mfma_f32_4x4x1]\n");
    printf("%-18s %8s %8s %8s %8s\n", "waves per SIMD", "1", "2", "4", "6");
#define ROW(V, M) printf("V=%d M=%d %9s %8.2f %8.2f %8.2f %8.2f\n", V, M, "", time_mix<V, M>(1, d_out), time_mix<V, M>(2, d_out), time_mix<V, M>(4, d_out), time_mix<V, M>(6, d_out));
    ROW(4, 0) ROW(4, 1) ROW(4, 2) ROW(4, 4) ROW(0, 4) ROW(2, 0) ROW(2, 2) ROW(2, 4)
    return (bad_layout || bad_bits) ? 1 : 0;
}
