// Two addressing questions the round-3 senone kernel depends on (gfx950):
//  (1) LDS: does `ds_read_u8 v, vaddr offset:512` with a NEGATIVE vaddr wrap to 512 - |vaddr|
//      (a table mirrored around offset 512, indexed by a signed difference)?
//  (2) buffer loads with ADD_TID_ENABLE in the resource word 3: address = base + voffset +
//      stride * lane, no per-lane address arithmetic (what range check applies?).
// hipcc --offload-arch=gfx950 addr_probe.hip -o addr_probe && ./addr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void lds_probe(uint32_t *out)
{
    extern __shared__ uint8_t tab[]; // the kernel's only LDS: starts at address 0
    for (int i = threadIdx.x; i < 1024; i += 64)
        tab[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    int a = (int)threadIdx.x * 13 - 400; // -400 .. 419
    uint32_t lo = 0xdeadbeefu, hi = 0xdeadbeefu;
    asm volatile("ds_read_u8 %0, %2 offset:512\n\tds_read_u8_d16_hi %1, %2 offset:512\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(lo), "=&v"(hi) : "v"(a) : "memory");
    out[threadIdx.x * 2] = lo;
    out[threadIdx.x * 2 + 1] = hi;
}

typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void buf_probe(const uint32_t *data, uint32_t nrec, uint32_t w3, uint32_t *out)
{
    // word0/1: base + stride 4 in bits 48..61; word2: num_records; word3: flags
    const uint64_t b = (uint64_t)data;
    v4i r;
    r.x = (int)(uint32_t)b;
    r.y = (int)(((uint32_t)(b >> 32) & 0xffffu) | (4u << 16));
    r.z = (int)nrec;
    r.w = (int)w3;
    // make the descriptor wave-uniform SGPRs
    r.x = __builtin_amdgcn_readfirstlane(r.x);
    r.y = __builtin_amdgcn_readfirstlane(r.y);
    r.z = __builtin_amdgcn_readfirstlane(r.z);
    r.w = __builtin_amdgcn_readfirstlane(r.w);
    uint32_t voff = (threadIdx.x & 3u) * 4000u; // bytes
    uint32_t v0, v1;
    asm volatile("buffer_load_dword %0, %2, %3, 0 offen\n\t"
                 "buffer_load_dword %1, %2, %3, 0 offen offset:8\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1) : "v"(voff), "s"(r) : "memory");
    out[threadIdx.x * 2] = v0;
    out[threadIdx.x * 2 + 1] = v1;
}

int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    uint32_t *d;
    hipMalloc(&d, 4096);
    std::vector<uint32_t> h(1024);
    lds_probe<<<1, 64, 1024>>>(d);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
    int bad_lo = 0, bad_hi = 0;
    for (int l = 0; l < 64; ++l) {
        int a = l * 13 - 400;
        uint32_t want = (uint8_t)((512 + a) * 7 + 3);
        bad_lo += h[l * 2] != want;
        bad_hi += h[l * 2 + 1] != (want << 16);
    }
    printf("lds signed address: %s  mismatches lo %d hi %d   lane0 (a=-400): %08x %08x want %02x\n",
           hipGetErrorString(e), bad_lo, bad_hi, h[0], h[1], (uint8_t)((512 - 400) * 7 + 3));

    const int N = 1 << 24; // 64 MB: a wrong stride guess stays inside the allocation
    uint32_t *data;
    hipMalloc(&data, N * 4);
    std::vector<uint32_t> src(N);
    for (int i = 0; i < N; ++i)
        src[i] = i;
    hipMemcpy(data, src.data(), N * 4, hipMemcpyHostToDevice);
    const uint32_t w3s[] = { 0x00020000u, (1u << 23), (1u << 23) | 0x00020000u };
    // (with ADD_TID_ENABLE the DATA_FORMAT bits may extend the stride: 4 << 14 more bytes per lane)
    const uint32_t nrecs[] = { 0xffffffffu, (uint32_t)N, 64u, (uint32_t)N * 4u };
    for (uint32_t w3 : w3s)
        for (uint32_t nrec : nrecs) {
            hipMemset(d, 0xee, 4096);
            buf_probe<<<1, 64>>>(data, nrec, w3, d);
            e = hipDeviceSynchronize();
            hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
            int ok_tid = 0, ok_plain = 0, ok_tid2 = 0;
            for (int l = 0; l < 64; ++l) {
                uint32_t base = (l & 3) * 1000;
                ok_tid += h[l * 2] == base + l;
                ok_plain += h[l * 2] == base;
                ok_tid2 += h[l * 2 + 1] == base + l + 2;
            }
            printf("w3 %08x nrec %10u: %s  lanes matching base+voff+4*tid: %d (offset:8 form %d), "
                   "matching plain base+voff: %d   lane5: %u %u\n",
                   w3, nrec, hipGetErrorString(e), ok_tid, ok_tid2, ok_plain, h[10], h[11]);
        }
    return 0;
}
