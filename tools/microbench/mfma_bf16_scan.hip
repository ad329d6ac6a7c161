// Feasibility of a bf16-MFMA speculative scan ("path to 55 M"):
//  (a) throughput: one (32 densities x 32 frames) tile = 12 x v_mfma_f32_32x32x16_bf16 (K = 32 in
//      two steps, six split-precision products) + 16 top-5 inserts per lane (and_or + 4 med3 +
//      max); MFMA only, VALU only, both, with 2/3/4 waves per SIMD.
//  (b) accuracy: what the bf16 MFMA's fp32 accumulation does to a 16-term sum with a C input,
//      against an exact (double) sum, on random data with heavy cancellation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
#define REP 256

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, float s)
{
    v16f acc0 = {};
    v8bf a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(threadIdx.x * 0.001f + s + i);
        b[i] = (__bf16)(1.0f + s * i);
    }
    float L0 = threadIdx.x, L1 = 2, L2 = 3, L3 = 4, L4 = 5, key = s;
    for (int r = 0; r < REP; ++r) {
        if (MODE != 1) {
#pragma unroll
            for (int i = 0; i < 12; ++i)
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        }
        if (MODE != 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float kk = MODE == 2 ? acc0[i] : key;
                asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(kk), "v"(L1), "v"(L2));
                asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(L4) : "v"(L3), "v"(key));
                asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(L3) : "v"(L2), "v"(key));
                asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(L2) : "v"(L1), "v"(key));
                asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(L1) : "v"(L0), "v"(key));
                asm volatile("v_max_f32 %0, %0, %1" : "+v"(L0) : "v"(key));
            }
            if (MODE == 2)
                acc0 = (v16f){}; // the next tile starts from zero, as the real loop would
        }
    }
    float t = L0 + L1 + L2 + L3 + L4;
    for (int i = 0; i < 16; ++i) t += acc0[i];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int MODE> static void run(const char *name, float *out, int wg_per_cu)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, 0.5f);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    // wg_per_cu workgroups of 4 waves per CU = wg_per_cu waves per SIMD
    printf("%-26s %d waves/SIMD %8.1f us -> %.0f ns per tile per wave-slot, %.0f ns per tile of SIMD time\n",
           name, wg_per_cu, ms * 1e3, ms * 1e6 / REP, ms * 1e6 / (REP * (double)wg_per_cu));
}

// accuracy: D = A(32x16) * B(16x32) + C, A/B bf16, C fp32; compare with double
__global__ void acc_kernel(const __bf16 *A, const __bf16 *B, const float *C, float *D)
{
    // fragment layout of v_mfma_f32_32x32x16_bf16: A: lane l holds row l%32, k = 8*(l/32) .. +7;
    // B: lane l holds column l%32, the same k; C/D: lane l holds column l%32, rows
    // 8*(i/4) + 4*(l/32)... = (i%4) + 4*(l/32)*... (see below: D index r -> row 8*(r/4) + 4*(l/32) + r%4)
    int l = threadIdx.x;
    v8bf a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = A[(l % 32) * 16 + 8 * (l / 32) + i];
        b[i] = B[(8 * (l / 32) + i) * 32 + (l % 32)];
    }
    v16f c;
    for (int r = 0; r < 16; ++r)
        c[r] = C[(8 * (r / 4) + 4 * (l / 32) + r % 4) * 32 + (l % 32)];
    v16f d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r)
        D[(8 * (r / 4) + 4 * (l / 32) + r % 4) * 32 + (l % 32)] = d[r];
}

static float bf16_round(float x) { __bf16 b = (__bf16)x; return (float)b; }

int main()
{
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w = 2; w <= 4; ++w) {
        run<0>("mfma only", out, w);
        run<1>("valu only", out, w);
        run<2>("mfma + valu (dependent)", out, w);
    }
    // accuracy
    const int trials = 2000;
    std::vector<__bf16> hA(32 * 16), hB(16 * 32);
    std::vector<float> hC(32 * 32), hD(32 * 32);
    __bf16 *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2);
    hipMalloc(&dC, hC.size() * 4); hipMalloc(&dD, hD.size() * 4);
    double worst_rel_terms = 0, worst_ulps = 0;
    int layout_ok = 1;
    srand(12345);
    for (int t = 0; t < trials; ++t) {
        int mode = t % 4; // 0 random, 1 large cancellation, 2 wide exponent range, 3 C dominates
        for (auto &v : hA) { float x = (rand() / (float)RAND_MAX - 0.5f) * (mode == 2 ? ldexpf(1.0f, rand() % 24 - 12) : 4.0f); v = (__bf16)x; }
        for (auto &v : hB) { float x = (rand() / (float)RAND_MAX - 0.5f) * (mode == 2 ? ldexpf(1.0f, rand() % 24 - 12) : 4.0f); v = (__bf16)x; }
        for (int i = 0; i < 32 * 32; ++i) hC[i] = (rand() / (float)RAND_MAX - 0.5f) * (mode == 3 ? 1e6f : 1.0f);
        if (mode == 1)   // make every dot product cancel against C
            for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
                double s = 0; for (int k = 0; k < 16; ++k) s += (double)(float)hA[i * 16 + k] * (double)(float)hB[k * 32 + j];
                hC[i * 32 + j] = (float)(-s * (1.0 + 1e-3 * (rand() / (double)RAND_MAX)));
            }
        hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dC, hC.data(), hC.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(acc_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD.data(), dD, hD.size() * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = hC[i * 32 + j], sa = fabs((double)hC[i * 32 + j]);
            for (int k = 0; k < 16; ++k) {
                double p = (double)(float)hA[i * 16 + k] * (double)(float)hB[k * 32 + j];
                s += p; sa += fabs(p);
            }
            double err = fabs((double)hD[i * 32 + j] - s);
            if (sa > 0 && err / sa > worst_rel_terms) worst_rel_terms = err / sa;
            double ulp = fabs(s) > 0 ? ldexp(1.0, (int)floor(log2(fabs(s))) - 23) : 0;
            if (ulp > 0 && mode != 1 && err / ulp > worst_ulps) worst_ulps = err / ulp;
            if (t == 0 && err > 1e-2 * (sa + 1)) layout_ok = 0;
        }
    }
    printf("accuracy over %d tiles: layout %s; worst |D - exact| / sum|terms| = %.3g (= 2^%.2f); worst error in ulps of the result (no-cancellation cases) %.2f\n",
           trials, layout_ok ? "ok" : "WRONG", worst_rel_terms, log2(worst_rel_terms), worst_ulps);
    return 0;
}
