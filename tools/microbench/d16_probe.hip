// What does a d16 LDS load do to the other half of its destination on this GPU?
// (sramecc targets: LLVM assumes "not preserved".)  hipcc --offload-arch=gfx950 d16_probe.hip && ./a.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint32_t *out)
{
    __shared__ uint8_t tab[64];
    tab[threadIdx.x] = (uint8_t)(threadIdx.x + 1);
    __syncthreads();
    uint32_t a = threadIdx.x, hi = 0xdeadbeefu, lo = 0xdeadbeefu;
    asm volatile("ds_read_u8_d16_hi %0, %2\n\tds_read_u8_d16 %1, %2\n\ts_waitcnt lgkmcnt(0)"
                 : "+v"(hi), "+v"(lo) : "v"(a) : "memory");
    out[threadIdx.x * 2] = hi;
    out[threadIdx.x * 2 + 1] = lo;
}
int main()
{
    uint32_t *d, h[128];
    hipMalloc(&d, sizeof h);
    probe<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("d16_hi: %08x %08x   d16: %08x %08x\n", h[0], h[10], h[1], h[11]);
    return 0;
}
