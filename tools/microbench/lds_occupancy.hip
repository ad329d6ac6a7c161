// How many 256-thread workgroups does a CU hold at once, as a function of their LDS size?
// (Round 4: the scan kernel holds 32 KB and 94 VGPRs -- five per CU by the arithmetic, four by
// the time stamps.)  Every workgroup records s_memtime at start and end and its HW_ID / XCC_ID;
// the host counts the largest number of workgroups alive at once on one CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <map>

template <int VG>
__global__ void __launch_bounds__(256) spin(unsigned long long *rec, int iters)
{
    extern __shared__ float lds[];
    float acc[VG];
    for (int i = 0; i < VG; ++i) acc[i] = threadIdx.x * 0.5f + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    lds[threadIdx.x] = acc[0];
    __syncthreads();
    for (int r = 0; r < iters; ++r)
        for (int i = 0; i < VG; ++i) acc[i] = acc[i] * 1.0001f + lds[(threadIdx.x + i) & 255];
    float s = 0;
    for (int i = 0; i < VG; ++i) s += acc[i];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        rec[blockIdx.x * 4 + 0] = t0;
        rec[blockIdx.x * 4 + 1] = t1;
        rec[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg(63492);                 // HW_ID
        rec[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg(20 | (31 << 11));       // XCC_ID
        if (s == 12345.678f) rec[0] = 0;
    }
}

template <int VG> static void run(size_t lds_bytes, unsigned long long *d, int nwg)
{
    hipFuncSetAttribute((const void *)spin<VG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(spin<VG>, dim3(nwg), dim3(256), lds_bytes, 0, d, 3000);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nwg * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::map<long, std::vector<std::pair<unsigned long long, int>>> ev;
    for (int i = 0; i < nwg; ++i) {
        long cu = (long)(h[i * 4 + 3] & 0xf) * 1000 + (long)((h[i * 4 + 2] >> 13) & 7) * 100 + (long)((h[i * 4 + 2] >> 8) & 15);
        ev[cu].push_back({h[i * 4], 1});
        ev[cu].push_back({h[i * 4 + 1], -1});
    }
    int best = 0;
    for (auto &kv : ev) {
        std::sort(kv.second.begin(), kv.second.end());
        int c = 0;
        for (auto &e : kv.second) { c += e.second; best = std::max(best, c); }
    }
    printf("VGPR-ish %3d  LDS %6zu B: %zu CUs seen, at most %d workgroups of 256 threads alive on one CU\n",
           VG, lds_bytes, ev.size(), best);
}

int main()
{
    const int nwg = 256 * 12;
    unsigned long long *d;
    hipMalloc(&d, nwg * 4 * 8);
    for (size_t kb : {8, 16, 20, 24, 26, 28, 30, 31, 32, 40, 64})
        run<24>(kb * 1024, d, nwg);
    for (size_t kb : {16, 24, 30, 32})
        run<80>(kb * 1024, d, nwg);
    return 0;
}
