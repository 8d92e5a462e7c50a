// ubench_h2d.hip -- how fast can 0.8 GB of pageable host arrays reach HBM on this box?  (the upload path's first stage: gsr_upload_append)
//   pageable hipMemcpyAsync (ROCclr stages through its own pinned buffers) | pinned | hipHostRegister + copy + unregister |
//   own staging: T threads memcpy into two pinned chunks while the previous chunk's DMA runs
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench_h2d tools/ubench_h2d.hip -pthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static void par_copy(char* d, const char* s, size_t n, int T)
{
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([=] { const size_t a = n * t / T, b = n * (t + 1) / T; std::memcpy(d + a, s + a, b - a); });
    for (auto& x : th) x.join();
}
int main(int argc, char** argv)
{
    const size_t N = (size_t)((argc > 1 ? atof(argv[1]) : 0.79) * 1.0e9);
    char* src = (char*)malloc(N);
    for (size_t i = 0; i < N; i += 4096) src[i] = (char)i;     // touch
    memset(src, 1, N);
    char* dev; CK(hipMalloc((void**)&dev, N));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        CK(hipMemcpyAsync(dev, src, N, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
        printf("pageable hipMemcpyAsync           %7.2f ms  %6.1f GB/s\n", (now() - t0) * 1e3, N / (now() - t0) / 1e9);
    }
    char* pin; CK(hipHostMalloc((void**)&pin, N, 0));
    memcpy(pin, src, N);
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        CK(hipMemcpyAsync(dev, pin, N, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
        printf("pinned hipMemcpyAsync             %7.2f ms  %6.1f GB/s\n", (now() - t0) * 1e3, N / (now() - t0) / 1e9);
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        CK(hipHostRegister(src, N, hipHostRegisterDefault));
        double t1 = now();
        CK(hipMemcpyAsync(dev, src, N, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
        double t2 = now();
        CK(hipHostUnregister(src));
        printf("register %.2f + copy %.2f + unregister %.2f = %7.2f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (now() - t2) * 1e3, (now() - t0) * 1e3);
    }
    for (int T : {1, 2, 4, 8, 16}) {
        for (size_t chunk : {(size_t)8 << 20, (size_t)32 << 20}) {
            char* stg[2]; hipEvent_t ev[2];
            for (int k = 0; k < 2; ++k) { CK(hipHostMalloc((void**)&stg[k], chunk, 0)); CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming)); memset(stg[k], 0, chunk); }
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                double t0 = now();
                int k = 0;
                for (size_t off = 0; off < N; off += chunk, k ^= 1) {
                    const size_t b = N - off < chunk ? N - off : chunk;
                    if (off >= 2 * chunk) CK(hipEventSynchronize(ev[k]));
                    par_copy(stg[k], src + off, b, T);
                    CK(hipMemcpyAsync(dev + off, stg[k], b, hipMemcpyHostToDevice, s));
                    CK(hipEventRecord(ev[k], s));
                }
                CK(hipStreamSynchronize(s));
                best = std::min(best, now() - t0);
            }
            printf("own staging, %2d threads, %2zu MB chunks %7.2f ms  %6.1f GB/s\n", T, chunk >> 20, best * 1e3, N / best / 1e9);
            for (int k = 0; k < 2; ++k) { hipHostFree(stg[k]); hipEventDestroy(ev[k]); }
        }
    }
    return 0;
}
