// What does a d16 LDS load do to the other half of its destination on this GPU?
// (sramecc targets: LLVM assumes "not preserved".)  hipcc --offload-arch=gfx950 d16_probe.hip && ./a.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint32_t *out)
{
    extern __shared__ uint8_t tab[]; // the only LDS of the kernel: starts at LDS address 0
    uint32_t a = threadIdx.x, v = threadIdx.x + 1, hi = 0xdeadbeefu, lo = 0xdeadbeefu;
    asm volatile("ds_write_b8 %2, %3\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n\t"
                 "ds_read_u8_d16_hi %0, %2\n\tds_read_u8_d16 %1, %2\n\ts_waitcnt lgkmcnt(0)"
                 : "+v"(hi), "+v"(lo) : "v"(a), "v"(v) : "memory");
    out[threadIdx.x * 2] = hi;
    out[threadIdx.x * 2 + 1] = lo;
    if (threadIdx.x == 0)
        out[128] = (uint32_t)(uintptr_t)tab;
}
int main()
{
    uint32_t *d, h[129];
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 1;
    hipMemset(d, 0xff, sizeof h);
    probe<<<1, 64, 64>>>(d);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("%s  d16_hi: %08x %08x   d16: %08x %08x  base %u\n", hipGetErrorString(e), h[0], h[10], h[1], h[11], h[128]);
    return 0;
}
