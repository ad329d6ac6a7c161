// Does hipExtLaunchKernel(..., hipExtAnyOrderLaunch) let two kernels of ONE stream overlap on gfx950?
// Two small grids (64 groups each: a quarter of the chip), each spinning ~200 us; in order they
// take 2 x, overlapped 1 x.  Build: hipcc --offload-arch=gfx950 -O2 -o anyorder anyorder.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

__global__ void spin(unsigned long long ticks, int *out)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < ticks) { }
    if (threadIdx.x == 0 && out) out[blockIdx.x] = 1;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipStream_t s;
    hipStreamCreate(&s);
    int *d;
    hipMalloc(&d, 4096);
    unsigned long long ticks = 20000; // s_memtime / cycle counter ticks (100 MHz: 200 us)
    void *args[] = { &ticks, &d };
    for (int flags = 0; flags < 2; ++flags) {
        for (int rep = 0; rep < 3; ++rep) {
            hipStreamSynchronize(s);
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < 4; ++k)
                hipExtLaunchKernel((const void *)spin, dim3(64), dim3(256), args, 0, s, nullptr, nullptr, flags);
            hipStreamSynchronize(s);
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("flags %d: 4 kernels %.1f us\n", flags, us);
        }
    }
    return 0;
}
