// what device memory costs to take and to give back on this box: hipMalloc / first use / hipFree by size.  hipcc -O2 -o /tmp/malloc_cost tools/exp/malloc_cost.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    double t0 = now();
    hipSetDevice(0);
    hipFree(nullptr);
    printf("context %.1f ms\n", (now() - t0) * 1e3);
    for (int round = 0; round < 2; ++round)
        for (size_t gb10 : {1, 5, 10, 40, 80}) {
            const size_t bytes = gb10 * (1ull << 30) / 10;
            void *p = nullptr;
            t0 = now();
            hipError_t e = hipMalloc(&p, bytes);
            const double tm = (now() - t0) * 1e3;
            t0 = now();
            hipMemsetAsync(p, 0xFF, bytes, 0);
            hipStreamSynchronize(0);
            const double ts = (now() - t0) * 1e3;
            t0 = now();
            hipMemsetAsync(p, 0x00, bytes, 0);
            hipStreamSynchronize(0);
            const double ts2 = (now() - t0) * 1e3;
            t0 = now();
            hipFree(p);
            const double tf = (now() - t0) * 1e3;
            printf("%5.1f GB: hipMalloc %7.2f ms (%s), first memset %7.2f ms, second %7.2f ms, hipFree %7.2f ms\n", bytes / 1073741824.0, tm, hipGetErrorString(e), ts, ts2, tf);
        }
    // many buffers held at exit: what the process pays after main
    std::vector<void *> held;
    t0 = now();
    for (int i = 0; i < 25; ++i) { void *p; hipMalloc(&p, 1ull << 30); held.push_back(p); }
    printf("25 x 1 GB hipMalloc %.1f ms\n", (now() - t0) * 1e3);
    fflush(stdout);
    return 0;
}
