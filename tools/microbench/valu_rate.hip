// Issue-rate microbenchmark for the instruction mix of ptm_topn_frames_kernel on gfx950.
// Each variant runs REP x 64 instructions per wave; blocks of 256 threads, grid sized so every
// SIMD holds W waves.  Prints cycles per wave-instruction per SIMD (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));
#define REP 256

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *cyc, float s0, float s1)
{
    float2v acc[4], x[4];
    for (int i = 0; i < 4; ++i) {
        acc[i] = float2v{ (float)threadIdx.x, 1.0f };
        x[i] = float2v{ 1.0f + i, 0.5f };
    }
    float L0 = threadIdx.x, L1 = 2, L2 = 3, L3 = 4, L4 = 5;
    int iL0 = threadIdx.x, iL1 = 0x01020304, iL2 = 7, iL3 = 9, iL4 = 11;
    __shared__ unsigned char tab[256];
    tab[threadIdx.x & 255] = (unsigned char)threadIdx.x;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) { // 4 independent chains, pk_fma VGPR only
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(x[i & 3]), "v"(x[(i + 1) & 3]));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[(i + 1) & 3]) : "v"(x[i & 3]), "v"(x[(i + 1) & 3]));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[(i + 2) & 3]) : "v"(x[i & 3]), "v"(x[(i + 1) & 3]));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[(i + 3) & 3]) : "v"(x[i & 3]), "v"(x[(i + 1) & 3]));
            } else if (MODE == 1) { // one dependent chain, pk_fma with SGPR pair broadcast
                float2v sp = { s0, s1 };
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[0]) : "s"(sp), "v"(x[i & 3]));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[0]) : "s"(sp), "v"(x[(i + 1) & 3]));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[0]) : "s"(sp), "v"(x[(i + 2) & 3]));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[0]) : "s"(sp), "v"(x[(i + 3) & 3]));
            } else if (MODE == 2) { // plain v_fma_f32 dependent chain
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(L0) : "v"(L1), "v"(L2));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(L0) : "v"(L1), "v"(L2));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(L0) : "v"(L1), "v"(L2));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(L0) : "v"(L1), "v"(L2));
            } else if (MODE == 3) { // med3 chain as in the key insert
                asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(L4) : "v"(L3), "v"(L0));
                asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(L3) : "v"(L2), "v"(L0));
                asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(L2) : "v"(L1), "v"(L0));
                asm volatile("v_max_f32 %0, %0, %1" : "+v"(L1) : "v"(L0));
            } else if (MODE == 4) { // v_pk_mul + v_pk_add (the old scan)
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(acc[1]) : "v"(acc[0]), "v"(x[i & 3]));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc[0]) : "v"(acc[1]), "v"(x[(i + 1) & 3]));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(acc[1]) : "v"(acc[0]), "v"(x[i & 3]));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc[0]) : "v"(acc[1]), "v"(x[(i + 1) & 3]));
            } else if (MODE == 6) { // SDWA byte add
                asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "+v"(iL2) : "v"(iL1));
                asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "+v"(iL3) : "v"(iL1));
                asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(iL4) : "v"(iL1));
            } else if (MODE == 7) { // v_sad_u16
                asm volatile("v_sad_u16 %0, %1, %2, 0" : "=v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_sad_u16 %0, %1, %2, 0" : "=v"(iL3) : "v"(iL1), "v"(iL2));
                asm volatile("v_sad_u16 %0, %1, %2, 0" : "=v"(iL4) : "v"(iL1), "v"(iL2));
                asm volatile("v_sad_u16 %0, %1, %2, 0" : "=v"(iL0) : "v"(iL1), "v"(iL2));
            } else if (MODE == 8) { // v_min_i32 / v_sub_u32
                asm volatile("v_min_i32 %0, %1, %2" : "=v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(iL3) : "v"(iL0), "v"(iL2));
                asm volatile("v_min_i32 %0, %1, %2" : "=v"(iL4) : "v"(iL3), "v"(iL2));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(iL1) : "v"(iL4), "v"(iL2));
            } else if (MODE == 9) { // the senone inner step: add_sdwa, sad, ds_read_u8, min, sub
                asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_sad_u16 %0, %1, %2, 0" : "=v"(iL3) : "v"(iL0), "v"(iL4));
                asm volatile("ds_read_u8 %0, %1" : "=v"(iL2) : "v"(iL3 & 255));
                asm volatile("v_min_i32 %0, %1, %2" : "=v"(iL4) : "v"(iL0), "v"(iL4));
                asm volatile("s_waitcnt lgkmcnt(0)\n\tv_sub_u32 %0, %1, %2" : "=v"(iL4) : "v"(iL4), "v"(iL2));
            } else if (MODE == 10) { // and_or + med3
                asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(iL0) : "v"(iL1), "v"(iL2), "v"(iL3));
                asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(iL4) : "v"(iL1), "v"(iL2), "v"(iL3));
                asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(iL0) : "v"(iL1), "v"(iL2), "v"(iL3));
                asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(iL4) : "v"(iL1), "v"(iL2), "v"(iL3));
            } else if (MODE == 11) { // max/min pair instead of med3
                asm volatile("v_min_f32 %0, %1, %2" : "=v"(L4) : "v"(L3), "v"(L0));
                asm volatile("v_max_f32 %0, %0, %1" : "+v"(L2) : "v"(L4));
                asm volatile("v_min_f32 %0, %1, %2" : "=v"(L4) : "v"(L1), "v"(L0));
                asm volatile("v_max_f32 %0, %0, %1" : "+v"(L3) : "v"(L4));
            } else if (MODE == 12) { // v_add_u32 rotating dst
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(iL3) : "v"(iL0), "v"(iL2));
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(iL4) : "v"(iL3), "v"(iL2));
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(iL1) : "v"(iL4), "v"(iL2));
            } else if (MODE == 13) { // v_sub_u32 in place
                asm volatile("v_sub_u32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_sub_u32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_sub_u32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_sub_u32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
            } else if (MODE == 14) { // v_min_i32 in place
                asm volatile("v_min_i32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_min_i32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_min_i32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_min_i32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
            } else if (MODE == 15) { // v_min_f32 in place
                asm volatile("v_min_f32 %0, %0, %1" : "+v"(L0) : "v"(L1));
                asm volatile("v_min_f32 %0, %0, %1" : "+v"(L0) : "v"(L1));
                asm volatile("v_min_f32 %0, %0, %1" : "+v"(L0) : "v"(L1));
                asm volatile("v_min_f32 %0, %0, %1" : "+v"(L0) : "v"(L1));
            } else if (MODE == 16) { // v_mul_f32 in place
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(L0) : "v"(L1));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(L0) : "v"(L1));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(L0) : "v"(L1));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(L0) : "v"(L1));
            } else if (MODE == 17) { // v_xor_b32 in place
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
            } else if (MODE == 18) { // v_sad_u16 in place
                asm volatile("v_sad_u16 %0, %0, %1, 0" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_sad_u16 %0, %0, %1, 0" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_sad_u16 %0, %0, %1, 0" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_sad_u16 %0, %0, %1, 0" : "+v"(iL0) : "v"(iL1));
            } else if (MODE == 19) { // v_med3_f32 in place
                asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(L0) : "v"(L1), "v"(L2));
                asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(L0) : "v"(L1), "v"(L2));
                asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(L0) : "v"(L1), "v"(L2));
                asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(L0) : "v"(L1), "v"(L2));
            } else if (MODE == 20) { // v_and_or_b32 in place
                asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(iL0) : "v"(iL1), "v"(iL2));
            } else if (MODE == 21) { // v_fma_f32 rotating dst
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(L0) : "v"(L1), "v"(L2), "v"(L3));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(L4) : "v"(L0), "v"(L2), "v"(L3));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(L1) : "v"(L4), "v"(L2), "v"(L3));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(L3) : "v"(L1), "v"(L2), "v"(L0));
            } else if (MODE == 22) { // v_bfe_u32 in place
                asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(iL0) : "v"(iL1));
            } else if (MODE == 23) { // v_lshrrev_b32 in place
                asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(iL0));
                asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(iL0));
                asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(iL0));
                asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(iL0));
            } else if (MODE == 24) { // v_lshlrev_b32 in place
                asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(iL0));
                asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(iL0));
                asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(iL0));
                asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(iL0));
            } else if (MODE == 25) { // v_and_b32 in place
                asm volatile("v_and_b32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_and_b32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_and_b32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_and_b32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
            } else if (MODE == 26) { // v_or_b32 in place
                asm volatile("v_or_b32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_or_b32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_or_b32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_or_b32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
            } else if (MODE == 27) { // v_cndmask_b32 (vcc)
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(iL0) : "v"(iL1));
            } else if (MODE == 28) { // v_cmp_gt_i32 (to vcc)
                asm volatile("v_cmp_gt_i32 vcc, %0, %1" :: "v"(iL0), "v"(iL1) : "vcc");
                asm volatile("v_cmp_gt_i32 vcc, %0, %1" :: "v"(iL0), "v"(iL1) : "vcc");
                asm volatile("v_cmp_gt_i32 vcc, %0, %1" :: "v"(iL0), "v"(iL1) : "vcc");
                asm volatile("v_cmp_gt_i32 vcc, %0, %1" :: "v"(iL0), "v"(iL1) : "vcc");
            } else if (MODE == 29) { // v_lshl_add_u32 in place
                asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(iL0) : "v"(iL1));
            } else if (MODE == 30) { // v_add3_u32 in place
                asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(iL0) : "v"(iL1), "v"(iL2));
            } else if (MODE == 31) { // v_mul_u32_u24 in place
                asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
            } else if (MODE == 32) { // v_mad_u32_u24 in place
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(iL0) : "v"(iL1), "v"(iL2));
            } else if (MODE == 33) { // v_perm_b32 in place
                asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(iL0) : "v"(iL1), "v"(iL2));
            } else if (MODE == 34) { // v_max_i32 in place
                asm volatile("v_max_i32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_max_i32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_max_i32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
                asm volatile("v_max_i32 %0, %0, %1" : "+v"(iL0) : "v"(iL1));
            } else if (MODE == 35) { // v_min3_i32 in place
                asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(iL0) : "v"(iL1), "v"(iL2));
            } else if (MODE == 36) { // v_sub_u32_sdwa byte-byte
                asm volatile("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "=v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "=v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "=v"(iL0) : "v"(iL1), "v"(iL2));
                asm volatile("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "=v"(iL0) : "v"(iL1), "v"(iL2));
            } else if (MODE == 37) { // v_mov_b32
                asm volatile("v_mov_b32 %0, %1" : "=v"(iL0) : "v"(iL1));
                asm volatile("v_mov_b32 %0, %1" : "=v"(iL1) : "v"(iL0));
                asm volatile("v_mov_b32 %0, %1" : "=v"(iL0) : "v"(iL1));
                asm volatile("v_mov_b32 %0, %1" : "=v"(iL1) : "v"(iL0));
            } else if (MODE == 5) { // v_add_u32 (plain integer)
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(L0) : "v"(L1));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(L0) : "v"(L1));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(L0) : "v"(L1));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(L0) : "v"(L1));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = acc[0].x + acc[1].y + acc[2].x + acc[3].y + L0 + L1 + L2 + L3 + L4 + (float)(iL0 + iL1 + iL2 + iL3 + iL4 + tab[3]);
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0)
        cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
static void run(const char *name, float *out, unsigned long long *cyc)
{
    for (int w = 4; w <= 4; w *= 2) {
        int blocks = 256 * w; // 256 CUs x w blocks of 4 waves -> w waves per SIMD
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0f, 2.0f);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0;
        for (auto v : h) sum += (double)v;
        double per_wave = sum / h.size();
        // cycles per instruction as seen by the SIMD: wave time / (instrs per wave * waves sharing)
        printf("%-34s waves/SIMD %d: wave time %8.0f ticks, %.2f ticks per wave-instr on the SIMD; kernel %.1f us "
               "-> %.2f ns per wave-instr on the SIMD\n", name, w, per_wave, per_wave / (REP * 64.0 * w),
               ms * 1e3, ms * 1e6 / (REP * 64.0 * w));
    }
}

int main()
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMalloc(&cyc, 256 * 8 * 4 * 8);
    run<0>("pk_fma vgpr, 4 chains", out, cyc);
    run<1>("pk_fma sgpr-pair bcast, 1 chain", out, cyc);
    run<2>("v_fma_f32, 1 chain", out, cyc);
    run<3>("med3 x3 + max", out, cyc);
    run<4>("pk_mul + pk_add", out, cyc);
    run<5>("v_add_u32", out, cyc);
    run<6>("v_add_u32_sdwa", out, cyc);
    run<7>("v_sad_u16", out, cyc);
    run<8>("v_min_i32 / v_sub_u32", out, cyc);
    run<9>("senone step (4 valu + ds_read_u8)", out, cyc);
    run<10>("v_and_or_b32", out, cyc);
    run<11>("v_min_f32 / v_max_f32", out, cyc);
    run<22>("v_bfe_u32 in place", out, cyc);
    run<23>("v_lshrrev_b32 in place", out, cyc);
    run<24>("v_lshlrev_b32 in place", out, cyc);
    run<25>("v_and_b32 in place", out, cyc);
    run<26>("v_or_b32 in place", out, cyc);
    run<27>("v_cndmask_b32 (vcc)", out, cyc);
    run<28>("v_cmp_gt_i32 (to vcc)", out, cyc);
    run<29>("v_lshl_add_u32 in place", out, cyc);
    run<30>("v_add3_u32 in place", out, cyc);
    run<31>("v_mul_u32_u24 in place", out, cyc);
    run<32>("v_mad_u32_u24 in place", out, cyc);
    run<33>("v_perm_b32 in place", out, cyc);
    run<34>("v_max_i32 in place", out, cyc);
    run<35>("v_min3_i32 in place", out, cyc);
    run<36>("v_sub_u32_sdwa byte-byte", out, cyc);
    run<37>("v_mov_b32", out, cyc);
    run<12>("v_add_u32 rotating dst", out, cyc);
    run<13>("v_sub_u32 in place", out, cyc);
    run<14>("v_min_i32 in place", out, cyc);
    run<15>("v_min_f32 in place", out, cyc);
    run<16>("v_mul_f32 in place", out, cyc);
    run<17>("v_xor_b32 in place", out, cyc);
    run<18>("v_sad_u16 in place", out, cyc);
    run<19>("v_med3_f32 in place", out, cyc);
    run<20>("v_and_or_b32 in place", out, cyc);
    run<21>("v_fma_f32 rotating dst", out, cyc);
    return 0;
}
