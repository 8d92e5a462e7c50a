// host_stage_ref.cpp -- the reference's PER-CAMERA-MOVE HOST STAGE, restated for timing only.
//
// TEST/BENCH INFRASTRUCTURE (bench.py's cpu_baseline leg and tests/ only), never product code.
//
// What the reference does on the CPU every time the camera translates
// (/root/reference/gsplat_plugin/src/GSplatRenderer.C:188-208, argsortByDistance): one squared
// distance per splat, an iota, and a PARALLEL COMPARISON SORT of the int indices through an indirect
// float compare (tbb::parallel_sort there; __gnu_parallel::sort stands in for it here -- TBB headers
// are not in this image).  Like the reference's, the sort is unstable.  The GPU path replaces this
// stage with k_preprocess's key + the device radix sort.
#include <parallel/algorithm>
#include <omp.h>
#include <cstdint>
#include <numeric>
#include <vector>

extern "C" int gso_reference_host_stage(const float* P, int64_t n, const float cam_pos[3], int32_t* perm, int threads)
{
    if (!P || !perm || n < 0) return -1;
    if (threads > 0) omp_set_num_threads(threads);
    std::vector<float> dist((size_t)n);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float dx = P[3 * i] - cam_pos[0], dy = P[3 * i + 1] - cam_pos[1], dz = P[3 * i + 2] - cam_pos[2];
        dist[(size_t)i] = dx * dx + dy * dy + dz * dz;
    }
    std::iota(perm, perm + n, 0);
    const float* d = dist.data();
    __gnu_parallel::sort(perm, perm + n, [d](int32_t a, int32_t b) { return d[a] < d[b]; });
    return 0;
}
