// Can the buffer unit hand the senone kernel its mixture weights already spread?  A typed buffer
// load with data format 8_8_8_8 / UINT and D16 packing should return four bytes as two
// registers (b0 | b1 << 16), (b2 | b3 << 16) -- the layout the packed log-add chain wants, which
// costs a v_perm_b32 per pair today.  Also with ADD_TID_ENABLE (stride 4) as in the kernel.
// hipcc --offload-arch=gfx950 tbuf_probe.hip -o tbuf_probe && ./tbuf_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

__global__ void probe(const uint8_t *data, uint32_t w3, int idx, uint32_t *out)
{
    const uint64_t b = (uint64_t)data;
    v4i r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    r.y = __builtin_amdgcn_readfirstlane((int)(((uint32_t)(b >> 32) & 0xffffu) | (4u << 16)));
    r.z = __builtin_amdgcn_readfirstlane(idx ? (int)0x7fffffff : 64);
    r.w = __builtin_amdgcn_readfirstlane((int)w3);
    uint32_t voff = (threadIdx.x & 3u) * 1024u;
    u2 v;
    if (idx) { /* the row offset as an INDEX (dwords): address = base + stride (index + lane) */
        uint32_t vidx = voff >> 2;
        asm volatile("tbuffer_load_format_d16_xyzw %0, %1, %2, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_UINT] idxen\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(v) : "v"(vidx), "s"(r) : "memory");
    } else
    asm volatile("tbuffer_load_format_d16_xyzw %0, %1, %2, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_UINT] offen\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v) : "v"(voff), "s"(r) : "memory");
    out[threadIdx.x * 2] = v.x;
    out[threadIdx.x * 2 + 1] = v.y;
}

int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    const int N = 1 << 20;
    uint8_t *data;
    uint32_t *d;
    hipMalloc(&data, N);
    hipMalloc(&d, 4096);
    std::vector<uint8_t> src(N);
    for (int i = 0; i < N; ++i)
        src[i] = (uint8_t)(i * 37 + 11);
    hipMemcpy(data, src.data(), N, hipMemcpyHostToDevice);
    std::vector<uint32_t> h(128);
    const uint32_t w3s[] = { (1u << 23), (1u << 23) | (4u << 12) /* nfmt uint */, 0x00020000u };
    for (int idx = 0; idx < 2; ++idx)
    for (uint32_t w3 : w3s) {
        hipMemset(d, 0xee, 4096);
        probe<<<1, 64>>>(data, w3, idx, d);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
        int ok = 0, ok_notid = 0;
        for (int l = 0; l < 64; ++l) {
            for (int tid_on = 0; tid_on < 2; ++tid_on) {
                const uint8_t *p = &src[(l & 3) * 1024 + (tid_on ? 4 * l : 0)];
                const uint32_t w0 = p[0] | (uint32_t)p[1] << 16, w1 = p[2] | (uint32_t)p[3] << 16;
                if (h[2 * l] == w0 && h[2 * l + 1] == w1)
                    (tid_on ? ok : ok_notid)++;
            }
        }
        printf("idxen %d w3 %08x: %s  lanes spread+tid %d, spread without tid %d   lane5: %08x %08x (bytes %02x %02x %02x %02x)\n",
               idx, w3, hipGetErrorString(e), ok, ok_notid, h[10], h[11], src[1024 + 20], src[1024 + 21],
               src[1024 + 22], src[1024 + 23]);
    }
    return 0;
}
