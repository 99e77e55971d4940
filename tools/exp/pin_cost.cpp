// what page-locked host memory costs to take, and whether another thread's runtime calls wait for it.  hipcc -O2 -o /tmp/pin_cost tools/exp/pin_cost.cpp -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipSetDevice(0);
    hipFree(nullptr);
    void *d = nullptr;
    hipMalloc(&d, 1 << 20);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (size_t mb : {64, 256, 512}) {
        void *h = nullptr;
        double t0 = now();
        hipHostMalloc(&h, mb << 20, hipHostMallocNonCoherent);
        const double ta = (now() - t0) * 1e3;
        t0 = now();
        hipHostFree(h);
        printf("hipHostMalloc %4zu MB: %7.2f ms (%.2f GB/s), hipHostFree %6.2f ms\n", mb, ta, (double)(mb << 20) / ta / 1e6, (now() - t0) * 1e3);
    }
    for (size_t mb : {64, 256, 512}) {   // the same through mmap + touch + hipHostRegister
        double t0 = now();
        void *h = mmap(nullptr, mb << 20, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
        const double tm = (now() - t0) * 1e3;
        t0 = now();
        hipError_t e = hipHostRegister(h, mb << 20, hipHostRegisterDefault);
        const double tr = (now() - t0) * 1e3;
        t0 = now();
        hipHostUnregister(h);
        munmap(h, mb << 20);
        printf("mmap(populate) %4zu MB: %7.2f ms, hipHostRegister %7.2f ms (%s), unregister + munmap %6.2f ms\n", mb, tm, tr, hipGetErrorString(e), (now() - t0) * 1e3);
    }
    // another thread's small calls while 512 MB are being page-locked
    for (int how = 0; how < 2; ++how) {
        std::atomic<bool> stop{false};
        double worst_cpy = 0, worst_malloc = 0;
        std::thread other([&] {
            char buf[8];
            while (!stop.load()) {
                double t0 = now();
                hipMemcpy(buf, d, 8, hipMemcpyDeviceToHost);
                worst_cpy = std::max(worst_cpy, (now() - t0) * 1e3);
                void *p = nullptr;
                t0 = now();
                hipMalloc(&p, 1 << 20);
                worst_malloc = std::max(worst_malloc, (now() - t0) * 1e3);
                hipFree(p);
            }
        });
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        const double w0c = worst_cpy, w0m = worst_malloc;
        worst_cpy = worst_malloc = 0;
        void *h = nullptr;
        double t0 = now();
        if (how == 0) hipHostMalloc(&h, 512u << 20, hipHostMallocNonCoherent);
        else { h = mmap(nullptr, 512u << 20, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0); hipHostRegister(h, 512u << 20, hipHostRegisterDefault); }
        const double ta = (now() - t0) * 1e3;
        stop = true;
        other.join();
        printf("%s of 512 MB took %.1f ms; meanwhile the other thread's worst hipMemcpy(8 B) %.2f ms (before: %.2f), worst hipMalloc(1 MB) %.2f ms (before: %.2f)\n",
               how == 0 ? "hipHostMalloc" : "mmap + hipHostRegister", ta, worst_cpy, w0c, worst_malloc, w0m);
    }
    return 0;
}
