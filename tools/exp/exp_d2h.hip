// How fast do K-BFS's records reach the host?  hipMemcpy vs a kernel that stores into pinned host memory (GPU box).
//   hipcc --offload-arch=gfx950 -O2 -o exp_d2h exp_d2h.hip && ./exp_d2h
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t bytes = 64ull << 20;
    void *d = nullptr, *h1 = nullptr, *h2 = nullptr, *h3 = nullptr;
    CK(hipMalloc(&d, bytes));
    CK(hipMemset(d, 7, bytes));
    CK(hipHostMalloc(&h1, bytes, hipHostMallocNonCoherent));
    CK(hipHostMalloc(&h2, bytes, hipHostMallocDefault));
    CK(hipHostMalloc(&h3, bytes, hipHostMallocNonCoherent | hipHostMallocNumaUser));
    memset(h1, 0, bytes); memset(h2, 0, bytes); memset(h3, 0, bytes);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const char *names[3] = {"noncoherent", "default(coherent)", "noncoherent+numa_user"};
    void *hs[3] = {h1, h2, h3};
    for (int v = 0; v < 3; ++v) {
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now();
            CK(hipMemcpy(hs[v], d, bytes, hipMemcpyDeviceToHost));
            double t1 = now();
            CK(hipMemcpyAsync(hs[v], d, bytes, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            double t2 = now();
            k_copy<<<1024, 256, 0, st>>>((const uint4 *)d, (uint4 *)hs[v], bytes / 16);
            CK(hipStreamSynchronize(st));
            double t3 = now();
            k_copy<<<256, 256, 0, st>>>((const uint4 *)d, (uint4 *)hs[v], bytes / 16);
            CK(hipStreamSynchronize(st));
            double t4 = now();
            if (rep == 2)
                printf("%-22s hipMemcpy %.1f GB/s | async+sync %.1f GB/s | kernel 1024 blocks %.1f GB/s | kernel 256 blocks %.1f GB/s\n", names[v], bytes / (t1 - t0) / 1e9,
                       bytes / (t2 - t1) / 1e9, bytes / (t3 - t2) / 1e9, bytes / (t4 - t3) / 1e9);
        }
    }
    // host -> device for comparison (the state upload)
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        CK(hipMemcpy(d, h1, bytes, hipMemcpyHostToDevice));
        double t1 = now();
        if (rep == 2) printf("H2D hipMemcpy from pinned %.1f GB/s\n", bytes / (t1 - t0) / 1e9);
    }
    // two copies at once on two streams (records and pool)
    hipStream_t st2;
    CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    double t0 = now();
    CK(hipMemcpyAsync(h1, d, bytes / 2, hipMemcpyDeviceToHost, st));
    CK(hipMemcpyAsync((char *)h2, (char *)d + bytes / 2, bytes / 2, hipMemcpyDeviceToHost, st2));
    CK(hipStreamSynchronize(st));
    CK(hipStreamSynchronize(st2));
    double t1 = now();
    printf("two async halves on two streams %.1f GB/s\n", bytes / (t1 - t0) / 1e9);
    return 0;
}
