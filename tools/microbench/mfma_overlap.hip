// Does a wave's VALU work (the top-5 insert chain) overlap its own fp32 MFMAs on gfx950?
// Three loops with 4 waves per SIMD: MFMA only, VALU only, both interleaved.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
#define REP 512

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, float s)
{
    v16f acc0 = {}, acc1 = {};
    float a = threadIdx.x * 0.001f + s, b = 1.0f + s;
    float L0 = threadIdx.x, L1 = 2, L2 = 3, L3 = 4, L4 = 5, key = s;
    for (int r = 0; r < REP; ++r) {
        if (MODE != 1) {
#pragma unroll
            for (int i = 0; i < 7; ++i) { // 14 MFMAs = one 32-density tile of K = 28
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
            }
        }
        if (MODE != 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { // 16 key inserts (and_or + 4 med3 + max)
                asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(key), "v"(L1), "v"(L2));
                asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(L4) : "v"(L3), "v"(key));
                asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(L3) : "v"(L2), "v"(key));
                asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(L2) : "v"(L1), "v"(key));
                asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(L1) : "v"(L0), "v"(key));
                asm volatile("v_max_f32 %0, %0, %1" : "+v"(L0) : "v"(key));
            }
        }
    }
    float t = L0 + L1 + L2 + L3 + L4;
    for (int i = 0; i < 16; ++i) t += acc0[i] + acc1[i];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int MODE> static void run(const char *name, float *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * 4), dim3(256), 0, 0, out, 0.5f);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves x REP x 14 MFMAs
    printf("%-28s %8.1f us  -> %.1f ns per 14-MFMA tile + 16 inserts per wave on the SIMD\n", name, ms * 1e3,
           ms * 1e6 / (REP * 4.0));
}

int main()
{
    float *out; hipMalloc(&out, 256 * 4 * 256 * 4);
    run<0>("mfma only", out);
    run<1>("valu only", out);
    run<2>("mfma + valu interleaved", out);
    return 0;
}
