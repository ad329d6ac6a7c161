// Is  r = v - (float)h  (v_cvt_f32_f16 + v_sub_f32) the same as ONE v_fma_mix_f32 (h as a binary16
// source, times -1.0, plus v) -- for every binary16 h, subnormals included, either half of the
// register, and v = h plus everything a round-to-nearest conversion leaves behind?  The scan's
// operand split (ssw_k1a_mfma.inc: split2_f16) uses the mix form if and only if this prints 0.
//   hipcc --offload-arch=gfx950 -o fma_mix_probe fma_mix_probe.hip && ./fma_mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
__global__ void probe(uint32_t *n_diff, uint32_t *first)
{
    const uint32_t hbits = blockIdx.x * 256 + threadIdx.x;          // every binary16 pattern
    _Float16 h = __builtin_bit_cast(_Float16, (uint16_t)hbits);
    const float hf = (float)h;
    if (hf != hf || hf - hf != 0.0f)                                  // NaN, infinity
        return;
    // v: h itself and h +- k quarter-ulps of ITS conversion cell (what cvt(v) == h allows), as
    // fp32 neighbours; plus the float just inside the cell's ends
    const float up = (float)__builtin_bit_cast(_Float16, (uint16_t)(hbits + 1));
    const float cell = (hbits & 0x7fff) == 0x7bff ? 32.0f : (up - hf) * ((hbits & 0x8000) ? -1.0f : 1.0f);
    for (int k = -32; k <= 32; ++k) {
        float v = hf + cell * (float)k * (1.0f / 64.0f);
        for (int nudge = -1; nudge <= 1; ++nudge) {
            float vv = __uint_as_float(__float_as_uint(v) + nudge);
            if (vv != vv)                       // (0 - 1 ulp as bits: a NaN; payloads may differ)
                continue;
            const float2v pr = { vv, vv };
            const f16x2 cv = __builtin_convertvector(pr, f16x2);      // as the kernel converts
            const uint32_t packed = __builtin_bit_cast(uint32_t, cv);
            const float a = vv - (float)cv.x;
            float lo, hi;
            asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(packed), "v"(vv));
            asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(packed), "v"(vv));
            if (__float_as_uint(a) != __float_as_uint(lo) || __float_as_uint(a) != __float_as_uint(hi)) {
                if (atomicAdd(n_diff, 1u) == 0) {
                    first[0] = hbits, first[1] = __float_as_uint(vv), first[2] = __float_as_uint(a);
                    first[3] = __float_as_uint(lo), first[4] = __float_as_uint(hi);
                }
            }
        }
    }
}
int main()
{
    uint32_t *d, h[8] = { 0 };
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 1;
    hipMemset(d, 0, sizeof h);
    probe<<<256, 256>>>(d, d + 1);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("%s  differences: %u  (first: h %04x v %08x sub %08x mix.lo %08x mix.hi %08x)\n",
           hipGetErrorString(e), h[0], h[1], h[2], h[3], h[4], h[5]);
    return h[0] != 0;
}
