// does a large device-to-host copy on one stream hold up a short kernel on another?  hipcc -O2 -o /tmp/cvk tools/exp/copy_vs_kernel.cpp
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void k_touch(const uint32_t *in, uint32_t *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + 1;
}
int main(int argc, char **argv) {
    const size_t copy_bytes = (argc > 1 ? atol(argv[1]) : 90) << 20, n = 5000000;
    const bool registered = argc > 2 && atoi(argv[2]);
    hipSetDevice(0);
    void *dsrc, *h;
    uint32_t *a, *b;
    hipMalloc(&dsrc, copy_bytes);
    hipMalloc((void **)&a, n * 4);
    hipMalloc((void **)&b, n * 4);
    if (registered) { h = mmap(nullptr, copy_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0); hipHostRegister(h, copy_bytes, hipHostRegisterDefault); }
    else hipHostMalloc(&h, copy_bytes, hipHostMallocNonCoherent);
    hipStream_t sc, sk;
    hipStreamCreateWithFlags(&sc, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sk, hipStreamNonBlocking);
    hipEvent_t e0, e1, c0, c1;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&c0); hipEventCreate(&c1);
    for (int round = 0; round < 4; ++round) {
        const bool with_copy = round >= 2;
        hipDeviceSynchronize();
        if (with_copy) { hipEventRecord(c0, sc); hipMemcpyAsync(h, dsrc, copy_bytes, hipMemcpyDeviceToHost, sc); hipEventRecord(c1, sc); }
        hipEventRecord(e0, sk);
        for (int r = 0; r < 5; ++r) k_touch<<<(unsigned)((n + 255) / 256), 256, 0, sk>>>(a, b, n);
        hipEventRecord(e1, sk);
        hipDeviceSynchronize();
        float km = 0, cm = 0;
        hipEventElapsedTime(&km, e0, e1);
        if (with_copy) hipEventElapsedTime(&cm, c0, c1);
        printf("%s host memory, %zu MB copy %s: five 20 MB kernels %.3f ms%s\n", registered ? "registered" : "hipHostMalloc", copy_bytes >> 20, with_copy ? "beside them" : "absent     ", km,
               with_copy ? (std::string(", the copy ") + std::to_string(cm) + " ms = " + std::to_string(copy_bytes / cm / 1e6) + " GB/s").c_str() : "");
    }
    return 0;
}
