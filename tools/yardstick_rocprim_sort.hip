// Yardstick only (not part of the product, not built by build()): rocPRIM's device radix sort on the depth sort's shapes (u32 keys of
// 25 bits + u64 payloads).  hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/rp_sort tools/yardstick_rocprim_sort.hip
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>
int main() {
    for (size_t n : {310000ul, 1300000ul, 2000000ul, 4340000ul}) {
        std::vector<uint32_t> k(n); std::vector<uint64_t> v(n);
        std::mt19937 g(1); for (size_t i = 0; i < n; ++i) { k[i] = g() & ((1u << 25) - 1); v[i] = i; }
        uint32_t *ka, *kb; uint64_t *va, *vb;
        hipMalloc(&ka, n * 4); hipMalloc(&kb, n * 4); hipMalloc(&va, n * 8); hipMalloc(&vb, n * 8);
        hipMemcpy(ka, k.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(va, v.data(), n * 8, hipMemcpyHostToDevice);
        size_t tmp = 0; void* d = nullptr;
        rocprim::radix_sort_pairs(nullptr, tmp, ka, kb, va, vb, n, 0, 25);
        hipMalloc(&d, tmp);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) rocprim::radix_sort_pairs(d, tmp, ka, kb, va, vb, n, 0, 25);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) rocprim::radix_sort_pairs(d, tmp, ka, kb, va, vb, n, 0, 25);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("rocprim radix_sort_pairs n=%zu (25 bits, 12 B per item): %.1f us, temp %zu KB\n", n, ms / 20 * 1e3, tmp / 1024);
        hipFree(ka); hipFree(kb); hipFree(va); hipFree(vb); hipFree(d);
    }
    return 0;
}
