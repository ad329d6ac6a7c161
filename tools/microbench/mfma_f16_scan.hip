// Round 4: the scan on v_mfma_f32_32x32x16_f16 with operands cut into TWO fp16 parts (22
// significand bits, three part products (1,1) (1,2) (2,1) = 6 MFMAs per 32 x 32 tile) instead of
// three bf16 parts and six part products (12 MFMAs).
//  (a) throughput of one tile: N MFMAs + 16 top-5 inserts per lane, N = 6 (f16) and 12 (bf16),
//      4 and 5 waves per SIMD;
//  (b) accuracy of the f16 MFMA's fp32 accumulation against an exact (double) sum on random,
//      cancelling and wide-range data (the eps of csrc/ssw_model.c);
//  (c) fp16 SUBNORMAL inputs: are they honoured or flushed by the matrix pipe, and by
//      v_cvt_pk_f16_f32 with the kernel's default mode?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
#define REP 256

template <int NMFMA, bool F16, int MODE>
__global__ void __launch_bounds__(256) k(float *out, float s)
{
    v16f acc0 = {};
    v8bf a, b;
    v8h ah, bh;
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(threadIdx.x * 0.001f + s + i);
        b[i] = (__bf16)(1.0f + s * i);
        ah[i] = (_Float16)(threadIdx.x * 0.001f + s + i);
        bh[i] = (_Float16)(1.0f + s * i);
    }
    float L0 = threadIdx.x, L1 = 2, L2 = 3, L3 = 4, L4 = 5, key = s;
    for (int r = 0; r < REP; ++r) {
        if (MODE != 1) {
#pragma unroll
            for (int i = 0; i < NMFMA; ++i)
                acc0 = F16 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0, 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
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
                acc0 = (v16f){};
        }
    }
    float t = L0 + L1 + L2 + L3 + L4;
    for (int i = 0; i < 16; ++i) t += acc0[i];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int NMFMA, bool F16, int MODE> static void run(const char *name, float *out, int wg_per_cu)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<NMFMA, F16, MODE>), dim3(256 * wg_per_cu), dim3(256), 0, 0, out, 0.5f);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-34s %d waves/SIMD %8.1f us -> %.0f ns per tile of SIMD time\n", name, wg_per_cu,
           ms * 1e3, ms * 1e6 / (REP * (double)wg_per_cu));
}

__global__ void acc_kernel(const _Float16 *A, const _Float16 *B, const float *C, float *D)
{
    int l = threadIdx.x;
    v8h a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = A[(l % 32) * 16 + 8 * (l / 32) + i];
        b[i] = B[(8 * (l / 32) + i) * 32 + (l % 32)];
    }
    v16f c;
    for (int r = 0; r < 16; ++r)
        c[r] = C[(8 * (r / 4) + 4 * (l / 32) + r % 4) * 32 + (l % 32)];
    v16f d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r)
        D[(8 * (r / 4) + 4 * (l / 32) + r % 4) * 32 + (l % 32)] = d[r];
}

// (c): float -> two fp16 parts with the compiler's conversion (what the scan kernel will do)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void cvt_kernel(const float *x, unsigned *bits, float *resid)
{
    int i = threadIdx.x;
    f2 v = { x[2 * i], x[2 * i + 1] };
    h2 p = __builtin_convertvector(v, h2);
    unsigned pk = __builtin_bit_cast(unsigned, p);
    bits[i] = pk;
    resid[2 * i] = v.x - (float)p.x;
    resid[2 * i + 1] = v.y - (float)p.y;
}

int main()
{
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w = 4; w <= 5; ++w) {
        run<12, false, 0>("12 bf16 mfma only", out, w);
        run<6, true, 0>("6 f16 mfma only", out, w);
        run<6, true, 1>("valu only", out, w);
        run<12, false, 2>("12 bf16 mfma + inserts (dependent)", out, w);
        run<6, true, 2>("6 f16 mfma + inserts (dependent)", out, w);
    }
    const int trials = 2000;
    std::vector<_Float16> hA(32 * 16), hB(16 * 32);
    std::vector<float> hC(32 * 32), hD(32 * 32);
    _Float16 *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2);
    hipMalloc(&dC, hC.size() * 4); hipMalloc(&dD, hD.size() * 4);
    double worst_rel_terms = 0, worst_ulps = 0;
    int layout_ok = 1;
    srand(12345);
    for (int t = 0; t < trials; ++t) {
        int mode = t % 4;
        for (auto &v : hA) { float x = (rand() / (float)RAND_MAX - 0.5f) * (mode == 2 ? ldexpf(1.0f, rand() % 20 - 10) : 4.0f); v = (_Float16)x; }
        for (auto &v : hB) { float x = (rand() / (float)RAND_MAX - 0.5f) * (mode == 2 ? ldexpf(1.0f, rand() % 20 - 10) : 4.0f); v = (_Float16)x; }
        for (int i = 0; i < 32 * 32; ++i) hC[i] = (rand() / (float)RAND_MAX - 0.5f) * (mode == 3 ? 1e6f : 1.0f);
        if (mode == 1)
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
    printf("f16 accuracy over %d tiles: layout %s; worst |D - exact| / sum|terms| = %.3g (= %.2f u, u = 2^-24); worst error in ulps of the result (no-cancellation cases) %.2f\n",
           trials, layout_ok ? "ok" : "WRONG", worst_rel_terms, worst_rel_terms * 16777216.0, worst_ulps);
    // (c) subnormal inputs through the matrix pipe: A = 2^-20 (subnormal fp16) in one k slot,
    // B = 2^10 -> the exact product is 2^-10; a flushing pipe returns C
    for (int side = 0; side < 3; ++side) {
        for (auto &v : hA) v = (_Float16)0.0f;
        for (auto &v : hB) v = (_Float16)0.0f;
        for (auto &v : hC) v = 0.0f;
        const float tiny = ldexpf(1.0f, -20), big = ldexpf(1.0f, 10);
        for (int i = 0; i < 32; ++i) {
            hA[i * 16 + 3] = (_Float16)(side == 1 ? big : tiny);
            hB[3 * 32 + i] = (_Float16)(side == 0 ? big : tiny);
        }
        hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dC, hC.data(), hC.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(acc_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD.data(), dD, hD.size() * 4, hipMemcpyDeviceToHost);
        const double want = side == 2 ? ldexp(1.0, -40) : ldexp(1.0, -10);
        printf("subnormal fp16 %s: D[0][0] = %.6g, exact %.6g -> %s\n",
               side == 0 ? "A operand (2^-20 x 2^10)" : side == 1 ? "B operand (2^10 x 2^-20)" : "both (2^-20 x 2^-20)",
               (double)hD[0], want, hD[0] == (float)want ? "honoured" : "FLUSHED / inexact");
    }
    {
        const int n = 64;
        std::vector<float> hx(2 * n), hr(2 * n);
        std::vector<unsigned> hb(n);
        for (int i = 0; i < 2 * n; ++i)
            hx[i] = ldexpf(1.0f + 0.37f * (i % 7), -8 - i / 4) * (i & 1 ? -1.0f : 1.0f); // 2^-8 .. 2^-40
        float *dx, *dr; unsigned *db;
        hipMalloc(&dx, 8 * n); hipMalloc(&dr, 8 * n); hipMalloc(&db, 4 * n);
        hipMemcpy(dx, hx.data(), 8 * n, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(n), 0, 0, dx, db, dr);
        hipMemcpy(hb.data(), db, 4 * n, hipMemcpyDeviceToHost);
        hipMemcpy(hr.data(), dr, 8 * n, hipMemcpyDeviceToHost);
        int bad = 0; double worst = 0;
        for (int i = 0; i < 2 * n; ++i) {
            unsigned short h = (unsigned short)(hb[i / 2] >> (16 * (i & 1)));
            _Float16 hv; memcpy(&hv, &h, 2);
            _Float16 want = (_Float16)hx[i]; // host: round to nearest even, subnormals kept
            unsigned short hw; memcpy(&hw, &want, 2);
            if (hw != h) { if (bad < 6) printf("  cvt x = %.6g: device 0x%04x host 0x%04x\n", hx[i], h, hw); ++bad; }
            double r = fabs((double)hx[i] - (double)(float)hv);
            double lim = fmax(ldexp(fabs((double)hx[i]), -11), ldexp(1.0, -25));
            if (r / lim > worst) worst = r / lim;
            if ((double)hr[i] != (double)hx[i] - (double)(float)hv) { printf("  residual of %.6g not exact\n", hx[i]); ++bad; }
        }
        printf("v_cvt to fp16 on %d values down to 2^-40: %d differ from host RNE; worst residual / max(2^-11 |x|, 2^-25) = %.3f\n",
               2 * n, bad, worst);
    }
    return 0;
}
