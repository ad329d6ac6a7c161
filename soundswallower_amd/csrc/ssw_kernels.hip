/*
 * ssw_kernels.hip -- gfx950 (MI355X, CDNA4) kernels for the SoundSwallower acoustic hot path
 * and the C ABI of include/ssw_amd.h on top of them.
 *
 * All arithmetic that the reference does in float32 is done here in float32 with one rounding
 * per operation (no FMA contraction: this file is built with -ffp-contract=off and carries the
 * pragma below), in the reference's operation order; everything else is int32.  Citations are
 * file:line in the SoundSwallower tree.
 *
 * Kernels
 *   ptm_topn_chain_kernel   exact frame-sequential top-N (eval_topn + eval_cb,
 *                           src/ptm_mgau.c:86-225): one wave64 per (utterance, codebook,
 *                           stream) chain, two densities per lane held in registers
 *   ptm_senone_kernel       codebook_norm + senone_eval (src/ptm_mgau.c:264-403): one
 *                           workgroup per frame, top-N block + log-add table in LDS
 *   viterbi_align_kernel    state_align_search step/finish + hmm_vit_eval_3st_lr
 *                           (src/state_align_search.c:177-268, src/hmm.c:482-567): one wave64
 *                           per utterance, HMM state in LDS
 */
#pragma clang fp contract(off)

#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ssw_internal.h"

#define HIP_OK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            ssw_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,   \
                          __LINE__);                                                         \
            return -1;                                                                       \
        }                                                                                    \
    } while (0)

namespace {

constexpr int WAVE = 64;

/* ---------------------------------------------------------------------------------- */
/* small device helpers                                                                */
/* ---------------------------------------------------------------------------------- */

/* (int32)d with the reference's clamp (src/ptm_mgau.c:128-131).  v_cvt_i32_f32 saturates, so
 * the explicit compare only documents intent. */
__device__ __forceinline__ int
dens2int(float d)
{
    return d < -2147483648.0f ? INT_MIN : (int)d;
}

/* senone_eval's density term (src/ms_senone.c:332-335): INT32_MIN >> 10 below the int range,
 * else ((int32)dist + 1023) >> 10 */
__device__ __forceinline__ int
ms_fden(float d)
{
    return d < -2147483648.0f ? (INT_MIN >> SSW_SENSCR_SHIFT)
                              : (((int)d + ((1 << SSW_SENSCR_SHIFT) - 1)) >> SSW_SENSCR_SHIFT);
}

/* d = det - sum_j (x_j - mu_j)^2 v_j: sub, mul, mul, sub, each rounded, j ascending
 * (src/ptm_mgau.c:63-68, src/ms_gauden.c:410-416). */
template <int VECLEN>
__device__ __forceinline__ float
density(const float (&x)[VECLEN], const float (&mean)[VECLEN], const float (&var)[VECLEN],
        float det)
{
    float d = det;
#pragma unroll
    for (int j = 0; j < VECLEN; ++j) {
        float diff = x[j] - mean[j];
        float sq = diff * diff;
        float c = sq * var[j];
        d = d - c;
    }
    return d;
}

__device__ __forceinline__ int
wave_max_i32(int v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        int o = __shfl_xor(v, off, WAVE);
        v = o > v ? o : v;
    }
    return v;
}

/* workgroup barrier that orders LDS traffic only (no wait for outstanding global loads/stores) */
__device__ __forceinline__ void
lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

/* wave-wide maximum with DPP row shifts and row broadcasts (no LDS round trips); uniform result */
__device__ __forceinline__ int
wave_max_dpp(int v)
{
#define SSW_DPP_MAX(ctrl, rmask)                                                             \
    {                                                                                        \
        int o = __builtin_amdgcn_update_dpp(INT_MIN, v, ctrl, rmask, 0xf, false);            \
        v = o > v ? o : v;                                                                   \
    }
    SSW_DPP_MAX(0x111, 0xf) /* row_shr:1 */
    SSW_DPP_MAX(0x112, 0xf) /* row_shr:2 */
    SSW_DPP_MAX(0x114, 0xf) /* row_shr:4 */
    SSW_DPP_MAX(0x118, 0xf) /* row_shr:8: lane 15 of every row holds the row's maximum */
    SSW_DPP_MAX(0x142, 0xa) /* row_bcast:15 into rows 1 and 3 */
    SSW_DPP_MAX(0x143, 0xc) /* row_bcast:31 into rows 2 and 3 */
#undef SSW_DPP_MAX
    return __builtin_amdgcn_readlane(v, 63);
}

/* ---------------------------------------------------------------------------------- */
/* K1a: exact frame-sequential top-N for one (utterance, codebook, stream) chain        */
/* ---------------------------------------------------------------------------------- */

struct ChainParams {
    const float *rec;     /* [n_cb][n_feat][n_density][SSW_REC_FLOATS] */
    const float *feats;   /* [n_frames][featdim] */
    const int *utt_off;   /* [n_utts+1] (chain mode) */
    const uint32_t *work; /* fix-up mode: list of flagged pairs, entry = t*n_cbf + cbf */
    const unsigned *work_count; /* fix-up mode: entries in work[] (device side) */
    const uint32_t *utt_start;  /* fix-up mode: bit per frame, set at an utterance's first frame */
    const uint32_t *carry_pk; /* optional [n_utts][n_cb*n_feat] packed cw order to start from */
    const uint8_t *cb_active; /* optional [n_cb]: 0 = only re-score the carried codewords
                               * (ptm_mgau_codebook_eval skips eval_cb, src/ptm_mgau.c:245-251) */
    uint32_t *topn_cw;    /* [n_frames][n_cb*n_feat] 4 x uint8 packed */
    int4 *topn_sc;        /* [n_frames][n_cb*n_feat] raw scores */
    const uint32_t *flags;/* fix-up mode: bit per (frame, cbf) */
    int n_utts, n_cbf, n_feat, featdim, ds, n_frames, frame_base;
    int featoff[SSW_MAX_FEAT];
};

template <int NDL>
struct LaneDens {
    float d[NDL];
    int i[NDL];
};

/* value of iv[] for density cw (cw wave-uniform) */
template <int NDL>
__device__ __forceinline__ int
read_density_int(const int (&iv)[NDL], int cw)
{
    int lane = cw & 63, half = cw >> 6, v = 0;
#pragma unroll
    for (int h = 0; h < NDL; ++h) {
        int t = __builtin_amdgcn_readlane(iv[h], lane);
        v = (half == h) ? t : v;
    }
    return v;
}

/* One frame of the reference's top-N state machine on a wave that holds this frame's 64*NDL
 * float densities (dv) and their truncated ints (iv), given last frame's codeword order Lc.
 * Restates eval_topn + eval_cb (src/ptm_mgau.c:86-225) exactly:
 *   1. re-score the carried codewords in carried order, each placed AFTER equal scores;
 *   2. scan codewords ascending; admit when d >= (float)worst.score and not present; place
 *      BEFORE equal scores; the worst drops off.  The scan is run as "find the next admissible
 *      codeword with a ballot", so it costs one iteration per insertion, not per density. */
template <int NDL, int TOPN>
__device__ __forceinline__ void
topn_exact_step(const float (&dv)[NDL], const int (&iv)[NDL], int (&Lc)[TOPN], int (&Ls)[TOPN],
                bool do_scan)
{
    int nc[TOPN], ns[TOPN];
#pragma unroll
    for (int i = 0; i < TOPN; ++i) {
        int c = Lc[i];
        int s = read_density_int<NDL>(iv, c);
        int pos = 0;
#pragma unroll
        for (int k = 0; k < i; ++k)
            pos += (ns[k] >= s) ? 1 : 0;
#pragma unroll
        for (int k = i; k >= 1; --k)
            if (k > pos) {
                ns[k] = ns[k - 1];
                nc[k] = nc[k - 1];
            }
#pragma unroll
        for (int k = 0; k <= i; ++k)
            if (k == pos) {
                ns[k] = s;
                nc[k] = c;
            }
    }
#pragma unroll
    for (int i = 0; i < TOPN; ++i) {
        Lc[i] = nc[i];
        Ls[i] = ns[i];
    }
    if (!do_scan)
        return;

    unsigned long long rem[NDL];
#pragma unroll
    for (int h = 0; h < NDL; ++h)
        rem[h] = ~0ull;
    for (;;) {
        float thr = (float)Ls[TOPN - 1];
        unsigned long long m[NDL];
#pragma unroll
        for (int h = 0; h < NDL; ++h)
            m[h] = __ballot(dv[h] >= thr) & rem[h];
#pragma unroll
        for (int k = 0; k < TOPN; ++k) {
            int c = Lc[k];
#pragma unroll
            for (int h = 0; h < NDL; ++h)
                if ((c >> 6) == h)
                    m[h] &= ~(1ull << (c & 63));
        }
        int cw = -1;
#pragma unroll
        for (int h = NDL - 1; h >= 0; --h)
            if (m[h] != 0)
                cw = h * 64 + __builtin_ctzll(m[h]);
        if (cw < 0)
            break;
        /* everything up to and including cw has now been scanned */
#pragma unroll
        for (int h = 0; h < NDL; ++h) {
            if (h < (cw >> 6))
                rem[h] = 0;
            else if (h == (cw >> 6))
                rem[h] &= ~((2ull << (cw & 63)) - 1ull);
        }
        int s = read_density_int<NDL>(iv, cw);
        int pos = 0;
#pragma unroll
        for (int k = 0; k < TOPN - 1; ++k)
            pos += (Ls[k] > s) ? 1 : 0;
#pragma unroll
        for (int k = TOPN - 1; k >= 1; --k)
            if (k > pos) {
                Ls[k] = Ls[k - 1];
                Lc[k] = Lc[k - 1];
            }
#pragma unroll
        for (int k = 0; k < TOPN; ++k)
            if (k == pos) {
                Ls[k] = s;
                Lc[k] = cw;
            }
    }
}

template <int VECLEN, int NDL>
__device__ __forceinline__ void
load_lane_gaussians(const float *rec_cbf, int lane, float (&mean)[NDL][VECLEN],
                    float (&var)[NDL][VECLEN], float (&det)[NDL])
{
#pragma unroll
    for (int h = 0; h < NDL; ++h) {
        const float4 *r = reinterpret_cast<const float4 *>(rec_cbf
                                                           + (size_t)(h * 64 + lane)
                                                               * SSW_REC_FLOATS);
        float buf[SSW_REC_FLOATS];
#pragma unroll
        for (int q = 0; q < SSW_REC_FLOATS / 4; ++q) {
            float4 v = r[q];
            buf[q * 4 + 0] = v.x;
            buf[q * 4 + 1] = v.y;
            buf[q * 4 + 2] = v.z;
            buf[q * 4 + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < VECLEN; ++j) {
            mean[h][j] = buf[j];
            var[h][j] = buf[SSW_REC_VAR + j];
        }
        det[h] = buf[SSW_REC_DET];
    }
}

/* Densities of one frame (feature sub-vector x) for the wave's 64*NDL codewords, then one exact
 * top-N step, then the frame's packed result is stored by lane 0. */
template <int VECLEN, int NDL, int TOPN>
__device__ __forceinline__ void
chain_frame_x(const ChainParams &P, int t, int cbf, const float (&x)[VECLEN], bool do_scan,
              int lane, const float (&mean)[NDL][VECLEN], const float (&var)[NDL][VECLEN],
              const float (&det)[NDL], int (&Lc)[TOPN], int (&Ls)[TOPN])
{
    float dv[NDL];
    int iv[NDL];
#pragma unroll
    for (int h = 0; h < NDL; ++h) {
        dv[h] = density<VECLEN>(x, mean[h], var[h], det[h]);
        iv[h] = dens2int(dv[h]);
    }
    topn_exact_step<NDL, TOPN>(dv, iv, Lc, Ls, do_scan);
    if (lane == 0) {
        uint32_t pk = 0;
#pragma unroll
        for (int k = 0; k < TOPN; ++k)
            pk |= (uint32_t)(Lc[k] & 0xff) << (8 * k);
        P.topn_cw[(size_t)t * P.n_cbf + cbf] = pk;
        static_assert(TOPN == 4, "score store is an int4");
        P.topn_sc[(size_t)t * P.n_cbf + cbf] = make_int4(Ls[0], Ls[1], Ls[2], Ls[3]);
    }
}

template <int VECLEN, int NDL, int TOPN>
__device__ __forceinline__ void
chain_frame(const ChainParams &P, int t, int cbf, int f, bool do_scan, int lane,
            const float (&mean)[NDL][VECLEN], const float (&var)[NDL][VECLEN],
            const float (&det)[NDL], int (&Lc)[TOPN], int (&Ls)[TOPN])
{
    const float *xp = P.feats + (size_t)t * P.featdim + P.featoff[f];
    float x[VECLEN];
#pragma unroll
    for (int j = 0; j < VECLEN; ++j)
        x[j] = xp[j];
    chain_frame_x<VECLEN, NDL, TOPN>(P, t, cbf, x, do_scan, lane, mean, var, det, Lc, Ls);
}

/* Exact path: one wave per (utterance, cbf) chain, all frames in order, history reset (or
 * taken from `carry_pk`).  Used when ds != 1 and by the one-frame frame_eval path. */
template <int VECLEN, int NDL, int TOPN>
__global__ void __launch_bounds__(256)
ptm_topn_chain_kernel(ChainParams P)
{
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (wid >= P.n_utts * P.n_cbf)
        return;
    const int u = wid / P.n_cbf;
    const int cbf = wid - u * P.n_cbf;
    const int t0 = P.utt_off[u], t1 = P.utt_off[u + 1];
    const int f = cbf % P.n_feat;
    const float *rec_cbf = P.rec + (size_t)cbf * (NDL * 64) * SSW_REC_FLOATS;

    float mean[NDL][VECLEN], var[NDL][VECLEN], det[NDL];
    load_lane_gaussians<VECLEN, NDL>(rec_cbf, lane, mean, var, det);

    int Lc[TOPN], Ls[TOPN];
    if (P.carry_pk != nullptr) {
        uint32_t pk = P.carry_pk[(size_t)u * P.n_cbf + cbf];
#pragma unroll
        for (int k = 0; k < TOPN; ++k)
            Lc[k] = (pk >> (8 * k)) & 0xff;
    } else {
#pragma unroll
        for (int k = 0; k < TOPN; ++k)
            Lc[k] = k; /* reset history: cw = m (src/ptm_mgau.c:709) */
    }
#pragma unroll
    for (int k = 0; k < TOPN; ++k)
        Ls[k] = INT_MIN;
    for (int t = t0; t < t1; ++t) {
        bool do_scan = ((t - t0 + P.frame_base) % P.ds) == 0; /* src/ptm_mgau.c:241 */
        if (P.cb_active != nullptr && P.cb_active[cbf / P.n_feat] == 0)
            do_scan = false;
        chain_frame<VECLEN, NDL, TOPN>(P, t, cbf, f, do_scan, lane, mean, var, det, Lc, Ls);
    }
}

/* Fix-up pass behind ptm_topn_frames_kernel.  The frames kernel appends every pair it could not
 * prove order-independent to a work list (and sets bit t*n_cbf + cbf of the flag bitset); one
 * wave takes one list entry.  An entry that heads a run of consecutive flagged frames of its
 * chain re-does the run exactly, in frame order: carried order = the previous frame's (final)
 * result, or the reset state at an utterance start (utt_start: bit per frame).  Entries inside
 * a run are skipped, their head covers them.  ds == 1 only. */
__device__ __forceinline__ bool
bit_test(const uint32_t *bits, long long i)
{
    return (bits[i >> 5] >> (i & 31)) & 1u;
}

template <int VECLEN, int NDL, int TOPN>
__global__ void __launch_bounds__(64)
ptm_topn_fixup_kernel(ChainParams P, unsigned long long *n_fixed)
{
    const int lane = threadIdx.x;
    float mean[NDL][VECLEN], var[NDL][VECLEN], det[NDL];
    unsigned long long fixed = 0;
    const unsigned n_work = *P.work_count;
    for (unsigned i = blockIdx.x; i < n_work; i += gridDim.x) {
        const uint32_t bit = P.work[i];
        const int t = (int)(bit / (uint32_t)P.n_cbf);
        const int cbf = (int)(bit - (uint32_t)t * (uint32_t)P.n_cbf);
        const int f = cbf % P.n_feat;
        /* everything the item may need is requested at once; the head test comes after */
        const uint32_t pbit = bit >= (uint32_t)P.n_cbf ? bit - (uint32_t)P.n_cbf : bit;
        const uint32_t w_start = P.utt_start[t >> 5];
        const uint32_t w_prev = P.flags[pbit >> 5];
        const uint32_t pk = P.topn_cw[pbit];
        const float *xp = P.feats + (size_t)t * P.featdim + P.featoff[f];
        float x[VECLEN];
#pragma unroll
        for (int j = 0; j < VECLEN; ++j)
            x[j] = xp[j];
        load_lane_gaussians<VECLEN, NDL>(P.rec + (size_t)cbf * (NDL * 64) * SSW_REC_FLOATS, lane,
                                         mean, var, det);
        const bool at_start = (w_start >> (t & 31)) & 1u;
        if (!at_start && ((w_prev >> (pbit & 31)) & 1u))
            continue; /* only run heads start a walk */
        int Lc[TOPN], Ls[TOPN];
#pragma unroll
        for (int k = 0; k < TOPN; ++k) {
            Lc[k] = at_start ? k : (int)((pk >> (8 * k)) & 0xff);
            Ls[k] = INT_MIN;
        }
        chain_frame_x<VECLEN, NDL, TOPN>(P, t, cbf, x, true, lane, mean, var, det, Lc, Ls);
        ++fixed;
        for (int tt = t + 1; tt < P.n_frames; ++tt) {
            if (bit_test(P.utt_start, tt) || !bit_test(P.flags, (long long)tt * P.n_cbf + cbf))
                break;
            chain_frame<VECLEN, NDL, TOPN>(P, tt, cbf, f, true, lane, mean, var, det, Lc, Ls);
            ++fixed;
        }
    }
    if (lane == 0 && fixed)
        atomicAdd(n_fixed, fixed);
}

/* ---------------------------------------------------------------------------------- */
/* K1a fast path: history-free top-N, one lane per frame                                 */
/* ---------------------------------------------------------------------------------- */
/*
 * The reference's top-N list depends on the previous frame only through tie order and
 * boundary membership among EQUAL truncated scores.  If the four best
 * densities of a frame truncate to four distinct ints, all greater than the int of every other
 * density, the reference's list is exactly "the four best, best first" whatever it carried in
 * (proof in DESIGN.md).  So every (frame, chain) pair is first evaluated independently:
 *
 *   - a wave owns one (codebook, stream) and 64*FPL frames, one frame per lane (FPL packed);
 *     the 128 Gaussians stream through SGPRs (wave-uniform scalar loads), the lane's feature
 *     vector stays in VGPRs; the four fp32 ops per dimension run as packed v_pk_* ops over
 *     the lane's FPL frames, each op rounded on its own, in the reference's order;
 *   - a running top-5 is kept as 5 floats per frame whose low 7 mantissa bits carry the
 *     codeword (v_and_or + 5 x v_med3): 6 VALU ops per density, no cross-lane traffic;
 *   - the 4 best codewords are then recomputed exactly (gather of 4 records) and sorted; the
 *     5th key bounds every other density from above.  When "4 distinct ints > bound" cannot
 *     be shown the pair is flagged and ptm_topn_fixup_kernel redoes it with the exact
 *     sequential state machine.
 */
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4u __attribute__((ext_vector_type(4), aligned(4))); /* 4-byte aligned float4 */
typedef short short4u __attribute__((ext_vector_type(4), aligned(2))); /* 2-byte aligned 4 x int16 */
typedef float v16f __attribute__((ext_vector_type(16)));

struct FramesParams {
    uint32_t *topn_cw;
    int4 *topn_sc;
    uint32_t *flags;
    uint32_t *work;       /* flagged pairs, appended as found; entry = t*n_cbf + cbf */
    unsigned *work_count;
    int n_frames, n_cbf, n_feat, featdim, tile_groups;
    int featoff[SSW_MAX_FEAT];
};

__device__ __forceinline__ float
med3f(float a, float b, float c)
{
    return __builtin_amdgcn_fmed3f(a, b, c);
}

#define SSW_REC_LDS_STRIDE 36 /* dwords per exact record in LDS (32 + 4 of padding) */
#define SSW_EXLIST_STRIDE 132 /* [0] count, [1..] codewords the scan leaves to the exact form */

#if defined(SSW_TIMELINE) || defined(SSW_TIMELINE_SEN)
__device__ unsigned long long g_timeline[16384 * 6];
#endif
#ifdef SSW_TIMELINE
#define SSW_TL(k)                                                                            \
    if (lane == 0) {                                                                         \
        int wv = blockIdx.x * 4 + (threadIdx.x >> 6);                                        \
        if (wv < 8192)                                                                       \
            g_timeline[wv * 6 + (k)] = __builtin_amdgcn_s_memtime();                        \
    }
#else
#define SSW_TL(k)
#endif

template <int VECLEN, int FPL, bool MS>
__global__ void __launch_bounds__(256)
ptm_topn_frames_kernel(const float *__restrict__ rec, const float *__restrict__ recq,
                       const float *__restrict__ recmax, const uint32_t *__restrict__ exlist,
                       const float *__restrict__ feats, FramesParams P)
{
    static_assert(FPL == 1 || FPL == 2, "one or two frames per lane");
    const int lane = threadIdx.x & 63;
    /* Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one).  The (codebook,
     * stream) x tile-group pairs are cut into 8 contiguous chunks, one per XCD, so that the
     * workgroups that stream the same 16 KB of records sit behind the same L2. */
    const int n_pairs = P.n_cbf * P.tile_groups;
    const int chunk = (n_pairs + 7) >> 3;
    const int pair = (int)(blockIdx.x & 7u) * chunk + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= chunk || pair >= n_pairs)
        return;
    const int cbf = pair / P.tile_groups;
    const int tile = (pair - cbf * P.tile_groups) * 4 + (threadIdx.x >> 6);
    const int t_base = tile * 64 * FPL;
    /* The workgroup's four waves share the codebook: its 128 exact records (the epilogue
     * re-evaluates 4 of them per frame) are staged in LDS once, rows padded to 36 dwords so that
     * the per-lane row gathers spread over the banks.  From L1 those gathers ran at 64 B/clk
     * per CU and were the epilogue's whole cost. */
    __shared__ __align__(16) float s_rec[128 * SSW_REC_LDS_STRIDE];
    {
        const float4 *src = reinterpret_cast<const float4 *>(rec + (size_t)cbf * 128 * SSW_REC_FLOATS);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = i * 256 + (int)threadIdx.x; /* float4 index in the 128 x 32 table */
            const float4 v = src[e];
            *reinterpret_cast<float4 *>(&s_rec[(e >> 3) * SSW_REC_LDS_STRIDE + (e & 7) * 4]) = v;
        }
    }
    __syncthreads();
    if (t_base >= P.n_frames)
        return;
    SSW_TL(0)
#ifdef SSW_TIMELINE
    if (lane == 0) {
        int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (wv < 8192) {
            g_timeline[wv * 6 + 4] = __builtin_amdgcn_s_getreg(63492); /* HW_ID */
            g_timeline[wv * 6 + 5] = __builtin_amdgcn_s_getreg(20 | (31 << 11)); /* XCC_ID */
        }
    }
#endif
    const int f = cbf % P.n_feat;
    const float *rec_cbf = rec + (size_t)cbf * 128 * SSW_REC_FLOATS;
    const float *rq_cbf = recq + (size_t)cbf * 128 * SSW_REC_FLOATS;
    const float d0 = recmax[(size_t)cbf * SSW_REC_FLOATS]; /* the keys are relative to this */
    /* touch every 128-byte line of this codebook's scan records now, so that the scalar loads
     * of the scan find them in L2 */
    float touch = rq_cbf[lane * SSW_REC_FLOATS] + rq_cbf[(64 + lane) * SSW_REC_FLOATS];

    int tt[FPL];
    float x[FPL][VECLEN];
    float2v xv[VECLEN], xq[VECLEN]; /* x and x*x of the lane's frames, packed per dimension */
#pragma unroll
    for (int h = 0; h < FPL; ++h) {
        tt[h] = t_base + h * 64 + lane;
        int tl = tt[h] < P.n_frames ? tt[h] : P.n_frames - 1;
        const float *xp = feats + (size_t)tl * P.featdim + P.featoff[f];
        /* every lane reads its own row: 16-byte loads (rows are only 4-byte aligned, which
         * global loads allow) cut the number of line look-ups per wave by three */
#pragma unroll
        for (int j = 0; j + 4 <= VECLEN; j += 4) {
            float4u v = *reinterpret_cast<const float4u *>(xp + j);
            x[h][j] = v.x;
            x[h][j + 1] = v.y;
            x[h][j + 2] = v.z;
            x[h][j + 3] = v.w;
        }
#pragma unroll
        for (int j = VECLEN & ~3; j < VECLEN; ++j)
            x[h][j] = xp[j];
    }
#pragma unroll
    for (int j = 0; j < VECLEN; ++j) {
        xv[j].x = x[0][j];
        xv[j].y = x[FPL - 1][j];
        xq[j] = xv[j] * xv[j];
    }

    const float NEG_INF = -__builtin_huge_valf(), POS_INF = __builtin_huge_valf();
    float L[FPL][5];
#pragma unroll
    for (int h = 0; h < FPL; ++h)
#pragma unroll
        for (int k = 0; k < 5; ++k)
            L[h][k] = NEG_INF;

    /* The 128 records stream through SGPRs, double-buffered by hand: the loads of record
     * cw+1 are issued before the arithmetic on record cw and waited for after it.  They are
     * inline asm because SMEM returns out of order (only lgkmcnt(0) is meaningful) and the
     * compiler would otherwise issue and wait in one place; the operand ties keep issue, use
     * and wait in this order, and keep every register of a tuple reserved while its load is
     * in flight. */
    uint32_t keymask;
    asm volatile("v_mov_b32 %0, 0xffffff80" : "=v"(keymask));
    v16f a_lo, a_hi, b_lo, b_hi;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40"
                 : "=&s"(a_lo), "=&s"(a_hi)
                 : "s"(rq_cbf));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a_lo), "+s"(a_hi));

#define SSW_REC_ISSUE(lo, hi, ptr, tie)                                                      \
    asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx16 %1, %3, 0x40"              \
                 : "=&s"(lo), "=&s"(hi), "+s"(tie)                                           \
                 : "s"(ptr))
#define SSW_REC_WAIT(lo, hi, vtie) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(lo), "+s"(hi), "+v"(vtie))
#define SSW_KEY_INSERT(d, cwv)                                                               \
    _Pragma("unroll") for (int h = 0; h < FPL; ++h)                                          \
    {                                                                                        \
        float key = __uint_as_float((__float_as_uint(d[h]) & keymask) | (uint32_t)(cwv));    \
        L[h][4] = med3f(L[h][3], L[h][4], key);                                              \
        L[h][3] = med3f(L[h][2], L[h][3], key);                                              \
        L[h][2] = med3f(L[h][1], L[h][2], key);                                              \
        L[h][1] = med3f(L[h][0], L[h][1], key);                                              \
        asm("v_max_f32 %0, %1, %2" : "=v"(L[h][0]) : "v"(L[h][0]), "v"(key));               \
    }
#define SSW_REC_SCAN(lo, hi, cwv)                                                            \
    {                                                                                        \
        float d[FPL];                                                                        \
        if (FPL == 2) {                                                                      \
            float2v dd = { lo[SSW_REC_DET], lo[SSW_REC_DET] };                               \
            _Pragma("unroll") for (int j = 0; j < VECLEN; ++j)                               \
            {                                                                                \
                float2v aa = { lo[j], lo[j] };                                               \
                float2v bb = { hi[j], hi[j] };                                               \
                dd = __builtin_elementwise_fma(aa, xv[j], dd);                               \
                dd = __builtin_elementwise_fma(bb, xq[j], dd);                               \
            }                                                                                \
            d[0] = dd.x;                                                                     \
            d[FPL - 1] = dd.y;                                                               \
        } else {                                                                             \
            float dd = lo[SSW_REC_DET];                                                      \
            _Pragma("unroll") for (int j = 0; j < VECLEN; ++j)                               \
            {                                                                                \
                dd = __builtin_fmaf(lo[j], xv[j].x, dd);                                     \
                dd = __builtin_fmaf(hi[j], xq[j].x, dd);                                     \
            }                                                                                \
            d[0] = dd;                                                                       \
        }                                                                                    \
        SSW_KEY_INSERT(d, cwv)                                                               \
    }

    static_assert(SSW_REC_VAR == 16 && SSW_REC_FLOATS == 32, "record = two 16-dword halves");
    SSW_TL(1)
    for (int cw = 0; cw < 128; cw += 2) {
        const float *rb = rq_cbf + (cw + 1) * SSW_REC_FLOATS;
        SSW_REC_ISSUE(b_lo, b_hi, rb, a_lo);
        SSW_REC_SCAN(a_lo, a_hi, cw);
        SSW_REC_WAIT(b_lo, b_hi, L[FPL - 1][0]);
        const float *ra = rq_cbf + (cw + 2 < 128 ? cw + 2 : 127) * SSW_REC_FLOATS;
        SSW_REC_ISSUE(a_lo, a_hi, ra, b_lo);
        SSW_REC_SCAN(b_lo, b_hi, cw + 1);
        SSW_REC_WAIT(a_lo, a_hi, L[FPL - 1][0]);
    }
#undef SSW_REC_ISSUE
#undef SSW_REC_WAIT
#undef SSW_REC_SCAN
    /* The few ill-conditioned densities of this codebook (their scan records are inert: key
     * -3e38) are evaluated the reference's way from the exact records; their keys need no bias. */
    {
        const uint32_t *xl = exlist + (size_t)cbf * SSW_EXLIST_STRIDE;
        const int n_ex = (int)xl[0];
        for (int i = 0; i < n_ex; ++i) {
            const int cwx = (int)xl[1 + i];
            const float *r = rec_cbf + cwx * SSW_REC_FLOATS;
            float2v dd = { r[SSW_REC_DET], r[SSW_REC_DET] };
#pragma unroll
            for (int j = 0; j < VECLEN; ++j) {
                float2v mm = { r[j], r[j] };
                float2v vv = { r[SSW_REC_VAR + j], r[SSW_REC_VAR + j] };
                float2v diff = xv[j] - mm;
                float2v sq = diff * diff;
                float2v c = sq * vv;
                dd = dd - c;
            }
            float d[FPL];
            d[0] = dd.x - d0;
            d[FPL - 1] = dd.y - d0;
            SSW_KEY_INSERT(d, cwx)
        }
    }
#undef SSW_KEY_INSERT
    SSW_TL(2)
    asm volatile("" ::"v"(touch));

#pragma unroll
    for (int h = 0; h < FPL; ++h) {
        int c[4], s[4];
        float dv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            c[k] = (int)(__float_as_uint(L[h][k]) & 127u);
            const float4 *rp = reinterpret_cast<const float4 *>(&s_rec[c[k] * SSW_REC_LDS_STRIDE]);
            float buf[SSW_REC_FLOATS];
#pragma unroll
            for (int q = 0; q < SSW_REC_FLOATS / 4; ++q) {
                float4 v = rp[q];
                buf[q * 4 + 0] = v.x;
                buf[q * 4 + 1] = v.y;
                buf[q * 4 + 2] = v.z;
                buf[q * 4 + 3] = v.w;
            }
            float dd = buf[SSW_REC_DET];
#pragma unroll
            for (int j = 0; j < VECLEN; ++j) {
                float diff = x[h][j] - buf[j];
                float sq = diff * diff;
                float cc = sq * buf[SSW_REC_VAR + j];
                dd = dd - cc;
            }
            dv[k] = dd;
            /* PTM keeps the truncated density (src/ptm_mgau.c:128-131); the ms scorer keeps the
             * float and later uses ((int32)dist + 1023) >> 10 (src/ms_senone.c:332-335) */
            s[k] = MS ? ms_fden(dd) : dens2int(dd);
        }
        /* sort the four best first: PTM by truncated score, ms by the float itself */
#define CSWAP(a, b)                                                                          \
    {                                                                                        \
        bool sw = MS ? (dv[b] > dv[a]) : (s[b] > s[a]);                                      \
        int ts = sw ? s[b] : s[a], tc = sw ? c[b] : c[a];                                    \
        float td = sw ? dv[b] : dv[a];                                                       \
        s[b] = sw ? s[a] : s[b];                                                             \
        c[b] = sw ? c[a] : c[b];                                                             \
        dv[b] = sw ? dv[a] : dv[b];                                                          \
        s[a] = ts;                                                                           \
        c[a] = tc;                                                                           \
        dv[a] = td;                                                                          \
    }
        CSWAP(0, 1) CSWAP(2, 3) CSWAP(0, 2) CSWAP(1, 3) CSWAP(1, 2)
#undef CSWAP
        /* upper bound on the true value of every density outside the four: the 5th key with
         * its 7 borrowed bits pushed towards +inf */
        uint32_t kb = __float_as_uint(L[h][4]);
        float ub = __uint_as_float((kb & 0x80000000u) ? (kb & ~127u) : (kb | 127u));
        /* the scan's keys are upper bounds, relative to d0, up to a term proportional to
         * their own size: v + 104 * 2^-24 |v| + 0.001 is increasing in v,
         * so the 5th key bounds every density outside the four; back to absolute, rounded up */
        ub = ub + __builtin_fabsf(ub) * 6.198883056640625e-06f + 1.0e-3f;
        ub = ub + d0;
        ub = ub + __builtin_fabsf(ub) * 2.384185791015625e-07f;
        bool proven;
        if (MS) /* compute_dist orders by float; exact ties are what needs the exact pass */
            proven = dv[0] > dv[1] && dv[1] > dv[2] && dv[2] > dv[3] && dv[3] > ub;
        else
            proven = s[0] > s[1] && s[1] > s[2] && s[2] > s[3] && ub == ub
                && s[3] > dens2int(ub);
        if (tt[h] < P.n_frames) {
            size_t idx = (size_t)tt[h] * P.n_cbf + cbf;
            P.topn_cw[idx] = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16)
                | ((uint32_t)c[3] << 24);
            P.topn_sc[idx] = make_int4(s[0], s[1], s[2], s[3]);
            if (!proven) {
                atomicOr(&P.flags[idx >> 5], 1u << (idx & 31));
                P.work[atomicAdd(P.work_count, 1u)] = (uint32_t)idx;
            }
        }
    }
    SSW_TL(3)
}

/* Exact pass of the ms scorer for flagged pairs: compute_dist (src/ms_gauden.c:384-432) is
 * history-free, so every flagged (frame, codebook, stream) is independent.  The list starts at
 * dist = (float)INT32_MIN; codeword d is admitted when dval >= worst.dist and placed at the
 * first position i with dval >= dist[i]. */
template <int VECLEN, int NDL, int TOPN>
__global__ void __launch_bounds__(64)
ms_topn_fixup_kernel(ChainParams P, unsigned long long *n_fixed)
{
    const int lane = threadIdx.x;
    float mean[NDL][VECLEN], var[NDL][VECLEN], det[NDL];
    int loaded_cbf = -1;
    unsigned long long fixed = 0;
    const unsigned n_work = *P.work_count;
    for (unsigned i = blockIdx.x; i < n_work; i += gridDim.x) {
        const uint32_t bit = P.work[i];
        const int t = (int)(bit / (uint32_t)P.n_cbf);
        const int cbf = (int)(bit - (uint32_t)t * (uint32_t)P.n_cbf);
        if (cbf != loaded_cbf) {
            load_lane_gaussians<VECLEN, NDL>(
                P.rec + (size_t)cbf * (NDL * 64) * SSW_REC_FLOATS, lane, mean, var, det);
            loaded_cbf = cbf;
        }
        const int f = cbf % P.n_feat;
        const float *xp = P.feats + (size_t)t * P.featdim + P.featoff[f];
        float x[VECLEN];
#pragma unroll
        for (int j = 0; j < VECLEN; ++j)
            x[j] = xp[j];
        float dvl[NDL];
#pragma unroll
        for (int h = 0; h < NDL; ++h)
            dvl[h] = density<VECLEN>(x, mean[h], var[h], det[h]);
        float Ld[TOPN];
        int Lc[TOPN];
#pragma unroll
        for (int k = 0; k < TOPN; ++k) {
            Ld[k] = -2147483648.0f;
            Lc[k] = 0;
        }
        unsigned long long rem[NDL];
#pragma unroll
        for (int h = 0; h < NDL; ++h)
            rem[h] = ~0ull;
        for (;;) {
            float thr = Ld[TOPN - 1];
            int cw = -1;
#pragma unroll
            for (int h = NDL - 1; h >= 0; --h) {
                unsigned long long m = __ballot(dvl[h] >= thr) & rem[h];
                if (m != 0)
                    cw = h * 64 + __builtin_ctzll(m);
            }
            if (cw < 0)
                break;
#pragma unroll
            for (int h = 0; h < NDL; ++h) {
                if (h < (cw >> 6))
                    rem[h] = 0;
                else if (h == (cw >> 6))
                    rem[h] &= ~((2ull << (cw & 63)) - 1ull);
            }
            float dval = 0.0f;
#pragma unroll
            for (int h = 0; h < NDL; ++h) {
                float tv = __builtin_bit_cast(
                    float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dvl[h]),
                                                     cw & 63));
                dval = ((cw >> 6) == h) ? tv : dval;
            }
            int pos = 0;
#pragma unroll
            for (int k = 0; k < TOPN - 1; ++k)
                pos += (Ld[k] > dval) ? 1 : 0;
#pragma unroll
            for (int k = TOPN - 1; k >= 1; --k)
                if (k > pos) {
                    Ld[k] = Ld[k - 1];
                    Lc[k] = Lc[k - 1];
                }
#pragma unroll
            for (int k = 0; k < TOPN; ++k)
                if (k == pos) {
                    Ld[k] = dval;
                    Lc[k] = cw;
                }
        }
        if (lane == 0) {
            uint32_t pk = 0;
#pragma unroll
            for (int k = 0; k < TOPN; ++k)
                pk |= (uint32_t)(Lc[k] & 0xff) << (8 * k);
            P.topn_cw[(size_t)t * P.n_cbf + cbf] = pk;
            P.topn_sc[(size_t)t * P.n_cbf + cbf]
                = make_int4(ms_fden(Ld[0]), ms_fden(Ld[1]), ms_fden(Ld[2]), ms_fden(Ld[3]));
        }
        ++fixed;
    }
    if (lane == 0 && fixed)
        atomicAdd(n_fixed, fixed);
}

/* ---------------------------------------------------------------------------------- */
/* K1b: normalise the top-N block and combine it with the mixture weights              */
/* ---------------------------------------------------------------------------------- */

struct SenoneParams {
    const uint32_t *topn_cw; /* [n_frames][n_cbf] 4 codewords packed */
    const int4 *topn_sc;     /* [n_frames][n_cbf] raw scores */
    const uint8_t *mixw;     /* [n_feat][n_density][slot_stride], slot order */
    const uint8_t *quad_cb;  /* [n_quads] codebook of slots 4q..4q+3 */
    const short4 *slot_sen;  /* [n_quads] senone ids of the 4 slots, -1 = padding */
    const uint8_t *logadd8;  /* [256] */
    uint32_t *flags;         /* optional: flag words of this frame are cleared for the next call */
    unsigned long long *nfixed; /* optional: [0] running count of the fix-up pass, [1] last batch,
                                 * [2] fill count of the fix-up work list */
    int16_t *out;            /* [n_frames][n_sen] */
    int n_frames, n_cb, n_feat, n_density, n_sen, slot_stride, n_quads;
    int aw, zero; /* ms scorer: acoustic weight divisor, logmath zero at shift 10 */
    int raw;      /* ms scorer: 1 = skip the frame normalisation (the caller applies it over an
                   * active set, src/ms_mgau.c:342-364) */
};

constexpr int SEN_MAX_THREADS = 1024;

#define SSW_LOGADD_LDS 512 /* 8-bit log-add table in LDS: 256 entries + zero padding */

/* fast_logmath_add (tied_mgau_common.h:100-117): min(x, y) - table[|x - y|] */
__device__ __forceinline__ int
fast_logadd(int x, int y, const uint8_t *tab)
{
    int d = (int)__builtin_amdgcn_sad_u16((unsigned)x, (unsigned)y, 0u); /* |x - y|, both < 2^16 */
    int r = x < y ? x : y;
    return r - (int)tab[d];
}

/* One workgroup per FPB consecutive frames.  Senones are visited in "slot" order: grouped by
 * codebook, each group padded to a multiple of 4, so one lane owns 4 consecutive slots that
 * share their top-N block and fetches the 4 mixture weights of a (stream, codeword) row with one
 * dword load.  R = quads per thread.  Several frames per workgroup amortise the prologue, the
 * barriers and the launch of 704-thread groups, which is what bounded the one-frame version. */
template <int TOPN, int R, int FPB, int NF>
__global__ void __launch_bounds__(SEN_MAX_THREADS)
ptm_senone_kernel(SenoneParams P)
{
#ifdef SSW_TIMELINE_SEN
#define STL(k)                                                                               \
    if ((threadIdx.x & 63) == 0) {                                                           \
        int wv = blockIdx.x * 11 + (threadIdx.x >> 6);                                       \
        if (wv < 16384)                                                                      \
            g_timeline[wv * 6 + (k)] = __builtin_amdgcn_s_memtime();                        \
    }
#else
#define STL(k)
#endif
    STL(0)
    /* set-up, block minimum and output are short latency-bound phases that the rest of the
     * workgroup (or its successor) waits for: let them issue ahead of other groups' main loops */
    __builtin_amdgcn_s_setprio(2);
    /* NF = number of streams when known at compile time (0 = read it from P): with a constant
     * trip count the 12 row loads of a quad are all issued before the first log-add */
    const int n_feat = NF ? NF : P.n_feat;
    static_assert(TOPN == 4, "top-N block is packed 4 x 8 bit");
    extern __shared__ __align__(16) unsigned char smem[];
    const int n_cbf = P.n_cb * P.n_feat;
    /* LDS carve: logadd[512] | norm[FPB][8] | ns4[FPB][n_cbf] | cw4[FPB][n_cbf] | red[FPB][16].
     * The table is indexed by |x - y| without a bound, as in the reference; x, y <= 255 + 96 and
     * the running value can dip below zero, so it is padded with zeros to 512 entries. */
    uint8_t *s_tab = smem;
    int *s_norm = reinterpret_cast<int *>(smem + SSW_LOGADD_LDS);
    uint32_t *s_ns4
        = reinterpret_cast<uint32_t *>(smem + SSW_LOGADD_LDS + 4 * SSW_MAX_FEAT * FPB);
    uint32_t *s_cw4 = s_ns4 + FPB * n_cbf;
    int *s_red = reinterpret_cast<int *>(s_cw4 + FPB * n_cbf);
    /* byte offsets of the 4 mixture-weight rows of every (frame, codebook, stream) */
    uint4 *s_roff = reinterpret_cast<uint4 *>(
        smem + ((reinterpret_cast<unsigned char *>(s_red + FPB * 16) - smem + 15) & ~(size_t)15));

    const int t0 = blockIdx.x * FPB;
    const int nfr = (P.n_frames - t0) < FPB ? (P.n_frames - t0) : FPB;
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;

    for (int i = tid; i < SSW_LOGADD_LDS; i += nthr)
        s_tab[i] = i < 256 ? P.logadd8[i] : (uint8_t)0;
    if (tid < SSW_MAX_FEAT * FPB)
        s_norm[tid] = SSW_WORST_SCORE;
    if (P.flags != nullptr) { /* these frames' flag bits have been consumed by the fix-up pass */
        long long b0 = (long long)t0 * n_cbf, b1 = b0 + (long long)nfr * n_cbf - 1;
        int w0 = (int)(b0 >> 5), w1 = (int)(b1 >> 5);
        for (int w = w0 + tid; w <= w1; w += nthr)
            P.flags[w] = 0u;
        if (blockIdx.x == 0 && tid == 0) {
            P.nfixed[1] = P.nfixed[0];
            P.nfixed[0] = 0ull;
            P.nfixed[2] = 0ull; /* the work list's fill count */
        }
    }
    __syncthreads();
    /* per-stream normaliser: max over codebooks of (best >> 10), src/ptm_mgau.c:271-278;
     * one thread per (frame, codebook, stream), combined with LDS atomics */
    const int n_items = nfr * n_cbf;
    for (int it = tid; it < n_items; it += nthr) {
        const int fr = it / n_cbf, i = it - fr * n_cbf;
        const int top = P.topn_sc[(size_t)(t0 + fr) * n_cbf + i].x;
        atomicMax(&s_norm[fr * SSW_MAX_FEAT + i % n_feat], top >> SSW_SENSCR_SHIFT);
    }
    __syncthreads();
    /* s = min(96, -((s >> 10) - norm)), src/ptm_mgau.c:284-290 */
    for (int it = tid; it < n_items; it += nthr) {
        const int fr = it / n_cbf, i = it - fr * n_cbf;
        const int4 sc = P.topn_sc[(size_t)(t0 + fr) * n_cbf + i];
        const uint32_t cw = P.topn_cw[(size_t)(t0 + fr) * n_cbf + i];
        const int norm = s_norm[fr * SSW_MAX_FEAT + i % n_feat];
        const int v[4] = { sc.x, sc.y, sc.z, sc.w };
        uint32_t pk = 0;
#pragma unroll
        for (int k = 0; k < TOPN; ++k) {
            int q = -((v[k] >> SSW_SENSCR_SHIFT) - norm);
            q = q > SSW_MAX_NEG_ASCR ? SSW_MAX_NEG_ASCR : q;
            pk |= (uint32_t)(q & 0xff) << (8 * k);
        }
        s_ns4[it] = pk;
        s_cw4[it] = cw;
        const uint32_t row0 = (uint32_t)(i % n_feat) * (uint32_t)P.n_density;
        s_roff[it] = make_uint4(__umul24(row0 + (cw & 0xffu), (uint32_t)P.slot_stride),
                                __umul24(row0 + ((cw >> 8) & 0xffu), (uint32_t)P.slot_stride),
                                __umul24(row0 + ((cw >> 16) & 0xffu), (uint32_t)P.slot_stride),
                                __umul24(row0 + (cw >> 24), (uint32_t)P.slot_stride));
    }
    __syncthreads();

    STL(1)
    __builtin_amdgcn_s_setprio(0);
    /* senone combine, src/ptm_mgau.c:342-395 */
    int asc[R][FPB][4];
    int best[FPB];
#pragma unroll
    for (int fr = 0; fr < FPB; ++fr)
        best[fr] = INT_MAX;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int q = r * nthr + tid;
#pragma unroll
        for (int fr = 0; fr < FPB; ++fr)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                asc[r][fr][j] = 0;
        if (q < P.n_quads) {
            const int cb = P.quad_cb[q];
            const uint8_t *mq = P.mixw + (size_t)q * 4;
            const short4 sen = P.slot_sen[q];
            const int sj[4] = { sen.x, sen.y, sen.z, sen.w };
#pragma unroll
            for (int fr = 0; fr < FPB; ++fr) {
                if (fr < nfr) {
                    if (NF) {
                        uint32_t mw[NF ? NF : 1][TOPN], ns4[NF ? NF : 1];
#pragma unroll
                        for (int f = 0; f < NF; ++f) {
                            const uint4 ro = s_roff[fr * n_cbf + cb * NF + f];
                            const uint32_t q4 = (uint32_t)q * 4u;
                            ns4[f] = s_ns4[fr * n_cbf + cb * NF + f];
                            mw[f][0] = *reinterpret_cast<const uint32_t *>(P.mixw + (ro.x + q4));
                            mw[f][1] = *reinterpret_cast<const uint32_t *>(P.mixw + (ro.y + q4));
                            mw[f][2] = *reinterpret_cast<const uint32_t *>(P.mixw + (ro.z + q4));
                            mw[f][3] = *reinterpret_cast<const uint32_t *>(P.mixw + (ro.w + q4));
                        }
#pragma unroll
                        for (int f = 0; f < NF; ++f)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                int fden = (int)((mw[f][0] >> (8 * j)) & 0xffu)
                                    + (int)(ns4[f] & 0xffu);
#pragma unroll
                                for (int k = 1; k < TOPN; ++k) {
                                    int y = (int)((mw[f][k] >> (8 * j)) & 0xffu)
                                        + (int)((ns4[f] >> (8 * k)) & 0xffu);
                                    fden = fast_logadd(fden, y, s_tab);
                                }
                                asc[r][fr][j] += fden;
                            }
                    } else
                    for (int f = 0; f < n_feat; ++f) {
                        const uint32_t cw4 = s_cw4[fr * n_cbf + cb * n_feat + f];
                        const uint32_t ns4 = s_ns4[fr * n_cbf + cb * n_feat + f];
                        uint32_t mw[TOPN];
#pragma unroll
                        for (int k = 0; k < TOPN; ++k) {
                            const uint32_t cw = (cw4 >> (8 * k)) & 0xffu;
                            mw[k] = *reinterpret_cast<const uint32_t *>(
                                mq + ((size_t)f * P.n_density + cw) * P.slot_stride);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            int fden = (int)((mw[0] >> (8 * j)) & 0xffu) + (int)(ns4 & 0xffu);
#pragma unroll
                            for (int k = 1; k < TOPN; ++k) {
                                int y = (int)((mw[k] >> (8 * j)) & 0xffu)
                                    + (int)((ns4 >> (8 * k)) & 0xffu);
                                fden = fast_logadd(fden, y, s_tab);
                            }
                            asc[r][fr][j] += fden;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (sj[j] >= 0)
                            best[fr] = asc[r][fr][j] < best[fr] ? asc[r][fr][j] : best[fr];
                }
            }
        }
    }
    STL(2)
    __builtin_amdgcn_s_setprio(2);
    /* block minimum of every frame: wave minimum by DPP, one LDS word per (frame, wave), one
     * barrier, then every wave folds the partials itself */
#pragma unroll
    for (int fr = 0; fr < FPB; ++fr) {
        const int wmin = -wave_max_dpp(-best[fr]); /* best >= INT_MIN + 1 */
        if ((tid & 63) == 0)
            s_red[fr * 16 + (tid >> 6)] = wmin;
    }
    __syncthreads();
#pragma unroll
    for (int fr = 0; fr < FPB; ++fr) {
        int b = INT_MAX;
        for (int k = 0; k < (nthr >> 6); ++k) {
            const int o = s_red[fr * 16 + k];
            b = o < b ? o : b;
        }
        best[fr] = b;
    }
    STL(3)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int q = r * nthr + tid;
        if (q < P.n_quads) {
            const short4 sen = P.slot_sen[q];
            const int sj[4] = { sen.x, sen.y, sen.z, sen.w };
            const bool run4 = sj[0] >= 0 && sj[1] == sj[0] + 1 && sj[2] == sj[0] + 2
                && sj[3] == sj[0] + 3;
#pragma unroll
            for (int fr = 0; fr < FPB; ++fr) {
                if (fr < nfr) {
                    const int b = best[fr];
                    int16_t *orow = P.out + (size_t)(t0 + fr) * P.n_sen;
                    int16_t v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) /* int16 arithmetic as in src/ptm_mgau.c:394-400 */
                        v[j] = (int16_t)((int16_t)asc[r][fr][j] - (int16_t)b);
                    if (run4) { /* 4 consecutive senone ids (19 quads in 20): one 8-byte store,
                                 * neighbouring lanes write neighbouring chunks */
                        short4u pk = { v[0], v[1], v[2], v[3] };
                        *reinterpret_cast<short4u *>(orow + sj[0]) = pk;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (sj[j] >= 0)
                                orow[sj[j]] = v[j];
                    }
                }
            }
        }
    }
    STL(4)
#undef STL
}

/* One frame with an explicit active set: the compallsen = no half of ptm_mgau_frame_eval
 * (src/ptm_mgau.c:264-403).  cb_active is the activity the top-N block was computed with
 * (the history slot's mgau_active): the per-stream normaliser runs over active codebooks only and
 * senones of inactive codebooks see 96 for all their scores (:353-364).  sen_active marks the
 * senones of the delta list (bridge entries included); the best score is taken over them, and,
 * as in the reference, it is subtracted from ALL n_sen entries, the others starting from 0. */
struct SenoneFrameParams {
    const uint32_t *topn_cw;
    const int4 *topn_sc;
    const uint8_t *mixw, *quad_cb, *logadd8, *cb_active, *sen_active;
    const short4 *slot_sen;
    int16_t *out;
    int n_cb, n_feat, n_density, n_sen, slot_stride, n_quads;
};

__global__ void __launch_bounds__(1024)
ptm_senone_frame_kernel(SenoneFrameParams P)
{
    __shared__ uint8_t s_tab[SSW_LOGADD_LDS];
    __shared__ int s_norm[SSW_MAX_FEAT];
    __shared__ int s_red[16];
    extern __shared__ __align__(16) unsigned char smem[];
    const int n_cbf = P.n_cb * P.n_feat;
    uint32_t *s_ns4 = reinterpret_cast<uint32_t *>(smem);
    uint32_t *s_cw4 = s_ns4 + n_cbf;
    const int tid = threadIdx.x, nthr = blockDim.x;

    for (int i = tid; i < SSW_LOGADD_LDS; i += nthr)
        s_tab[i] = i < 256 ? P.logadd8[i] : (uint8_t)0;
    if (tid < SSW_MAX_FEAT)
        s_norm[tid] = SSW_WORST_SCORE;
    __syncthreads();
    for (int i = tid; i < n_cbf; i += nthr)
        if (P.cb_active[i / P.n_feat])
            atomicMax(&s_norm[i % P.n_feat], P.topn_sc[i].x >> SSW_SENSCR_SHIFT);
    __syncthreads();
    for (int i = tid; i < n_cbf; i += nthr) {
        const int4 sc = P.topn_sc[i];
        const int v[4] = { sc.x, sc.y, sc.z, sc.w };
        const int norm = s_norm[i % P.n_feat];
        uint32_t pk = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int q = -((v[k] >> SSW_SENSCR_SHIFT) - norm);
            q = q > SSW_MAX_NEG_ASCR ? SSW_MAX_NEG_ASCR : q;
            if (!P.cb_active[i / P.n_feat])
                q = SSW_MAX_NEG_ASCR;
            pk |= (uint32_t)(q & 0xff) << (8 * k);
        }
        s_ns4[i] = pk;
        s_cw4[i] = P.topn_cw[i];
    }
    __syncthreads();
    int best = INT_MAX;
    for (int sl = tid; sl < P.n_quads * 4; sl += nthr) {
        const short4 s4 = P.slot_sen[sl >> 2];
        const int sj[4] = { s4.x, s4.y, s4.z, s4.w };
        const int sen = sj[sl & 3];
        if (sen < 0 || !P.sen_active[sen])
            continue;
        const int cb = P.quad_cb[sl >> 2];
        int a = 0;
        for (int f = 0; f < P.n_feat; ++f) {
            const uint32_t cw4 = s_cw4[cb * P.n_feat + f], ns4 = s_ns4[cb * P.n_feat + f];
            const uint8_t *mw = P.mixw + (size_t)f * P.n_density * P.slot_stride + sl;
            int fden = (int)mw[(size_t)(cw4 & 0xffu) * P.slot_stride] + (int)(ns4 & 0xffu);
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                int y = (int)mw[(size_t)((cw4 >> (8 * k)) & 0xffu) * P.slot_stride]
                    + (int)((ns4 >> (8 * k)) & 0xffu);
                fden = fast_logadd(fden, y, s_tab);
            }
            a += fden;
        }
        P.out[sen] = (int16_t)a; /* raw; normalised below */
        best = a < best ? a : best;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        int o = __shfl_xor(best, off, WAVE);
        best = o < best ? o : best;
    }
    if ((tid & 63) == 0)
        s_red[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        int b = INT_MAX;
        for (int w = 0; w < (nthr >> 6); ++w)
            b = s_red[w] < b ? s_red[w] : b;
        s_red[0] = b;
    }
    __syncthreads();
    best = s_red[0];
    __threadfence_block();
    for (int sen = tid; sen < P.n_sen; sen += nthr) {
        /* int16 -= int32 exactly as `senone_scores[i] -= bestscore` converts (src/ptm_mgau.c:399) */
        int16_t cur = P.sen_active[sen] ? P.out[sen] : (int16_t)0;
        P.out[sen] = (int16_t)((int)cur - best);
    }
}

/* logmath_add on the shift-10 table (src/logmath.c:228-272): max(x, y) + table[|x - y|], with
 * the "zero" short-cuts; the table has exactly 256 entries for the bases the loader accepts. */
__device__ __forceinline__ int
ms_logadd(int x, int y, int zero, const uint8_t *tab)
{
    int d = x > y ? x - y : y - x;
    int r = x > y ? x : y;
    int add = d < 256 ? (int)tab[d < 256 ? d : 255] : 0;
    int v = r + add;
    v = (y <= zero) ? x : v;
    v = (x <= zero) ? y : v;
    return v;
}

/* K3b: senone_eval + frame normalisation of the ms scorer (src/ms_senone.c:314-362,
 * src/ms_mgau.c:299-321).  Same slot order and quad-per-lane layout as ptm_senone_kernel; the
 * top-N block holds fden = ((int32)dist + 1023) >> 10 per codeword. */
template <int R>
__global__ void __launch_bounds__(SEN_MAX_THREADS)
ms_senone_kernel(SenoneParams P)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int n_cbf = P.n_cb * P.n_feat;
    /* LDS carve: logadd[256] | fd[n_cbf] int4 | cw4[n_cbf] | red[16] */
    uint8_t *s_tab = smem;
    int4 *s_fd = reinterpret_cast<int4 *>(smem + 256);
    uint32_t *s_cw4 = reinterpret_cast<uint32_t *>(s_fd + n_cbf);
    int *s_red = reinterpret_cast<int *>(s_cw4 + n_cbf);

    const int t = blockIdx.x;
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    if (tid < 256)
        s_tab[tid] = P.logadd8[tid];
    if (tid < n_cbf) {
        s_fd[tid] = P.topn_sc[(size_t)t * n_cbf + tid];
        s_cw4[tid] = P.topn_cw[(size_t)t * n_cbf + tid];
    }
    if (P.flags != nullptr) {
        long long b0 = (long long)t * n_cbf, b1 = b0 + n_cbf - 1;
        int w0 = (int)(b0 >> 5), w1 = (int)(b1 >> 5);
        if (tid <= w1 - w0)
            P.flags[w0 + tid] = 0u;
        if (t == 0 && tid == 0) {
            P.nfixed[1] = P.nfixed[0];
            P.nfixed[0] = 0ull;
            P.nfixed[2] = 0ull; /* the work list's fill count */
        }
    }
    __syncthreads();

    int scr[R][4];
    int best = INT_MAX;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int q = r * nthr + tid;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            scr[r][j] = 0;
        if (q < P.n_quads) {
            const int cb = P.quad_cb[q];
            const uint8_t *mq = P.mixw + (size_t)q * 4;
            for (int f = 0; f < P.n_feat; ++f) {
                const uint32_t cw4 = s_cw4[cb * P.n_feat + f];
                const int4 fd4 = s_fd[cb * P.n_feat + f];
                const int fd[4] = { fd4.x, fd4.y, fd4.z, fd4.w };
                uint32_t mw[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t cw = (cw4 >> (8 * k)) & 0xffu;
                    mw[k] = *reinterpret_cast<const uint32_t *>(
                        mq + ((size_t)f * P.n_density + cw) * P.slot_stride);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int fscr = fd[0] - (int)((mw[0] >> (8 * j)) & 0xffu);
#pragma unroll
                    for (int k = 1; k < 4; ++k) {
                        int fw = fd[k] - (int)((mw[k] >> (8 * j)) & 0xffu);
                        fscr = ms_logadd(fscr, fw, P.zero, s_tab);
                    }
                    scr[r][j] -= fscr;
                }
            }
            const short4 sen = P.slot_sen[q];
            const int sj[4] = { sen.x, sen.y, sen.z, sen.w };
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int v = scr[r][j] / P.aw; /* C division truncates toward zero */
                v = v > 32767 ? 32767 : v;
                v = v < -32768 ? -32768 : v;
                scr[r][j] = v;
                if (sj[j] >= 0)
                    best = v < best ? v : best;
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        int o = __shfl_xor(best, off, WAVE);
        best = o < best ? o : best;
    }
    if ((tid & 63) == 0)
        s_red[tid >> 6] = best;
    __syncthreads();
    if (tid < 64) {
        int b = tid < (nthr >> 6) ? s_red[tid] : INT_MAX;
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            int o = __shfl_xor(b, off, WAVE);
            b = o < b ? o : b;
        }
        if (tid == 0)
            s_red[0] = b;
    }
    __syncthreads();
    best = P.raw ? 0 : s_red[0];
    int16_t *orow = P.out + (size_t)t * P.n_sen;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int q = r * nthr + tid;
        if (q < P.n_quads) {
            const short4 sen = P.slot_sen[q];
            const int sj[4] = { sen.x, sen.y, sen.z, sen.w };
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (sj[j] >= 0) {
                    int bs = scr[r][j] - best; /* src/ms_mgau.c:314-320 */
                    bs = bs > 32767 ? 32767 : bs;
                    bs = bs < -32768 ? -32768 : bs;
                    orow[sj[j]] = (int16_t)bs;
                }
        }
    }
}

/* ---------------------------------------------------------------------------------- */
/* K4: dynamic features, whole utterances: batch CMN + 1s_c_d_dd                        */
/* ---------------------------------------------------------------------------------- */
/* feat_s2mfc2feat_block_utt for feat = 1s_c_d_dd, cmn = batch (src/feat.c:977-1008, 589-632;
 * src/cmn.c:168-200).  One wave per utterance, lane = cepstral dimension for the mean (the
 * reference accumulates sum[i] += mfc[f][i] in float32, frame after frame, skipping frames with
 * c0 < 0: that order is kept, 13 independent chains), then lanes sweep (frame, dimension)
 * pairs for the subtraction and the deltas:
 *   d[i]  = c[t+2][i] - c[t-2][i],   dd[i] = (c[t+3][i] - c[t-1][i]) - (c[t+1][i] - c[t-3][i])
 * with the first / last (mean-subtracted) frame replicated 3 times at the edges. */
struct FeatParams {
    const float *cep; /* [n_frames][ncep] */
    const int *utt_off;
    float *out;       /* [n_frames][3*ncep] */
    int n_utts, ncep;
};

__global__ void __launch_bounds__(64)
feat_1s_c_d_dd_kernel(FeatParams P)
{
    __shared__ float s_mean[64];
    const int u = blockIdx.x, lane = threadIdx.x, C = P.ncep;
    const int t0 = P.utt_off[u], n = P.utt_off[u + 1] - t0;
    if (n <= 0)
        return;
    const float *cep = P.cep + (size_t)t0 * C;
    if (lane < C) {
        float sum = 0.0f;
        int nframe = 0;
        for (int f = 0; f < n; ++f) {
            const float *row = cep + (size_t)f * C;
            if (row[0] < 0) /* "skip zero energy frames", src/cmn.c:186-188 */
                continue;
            sum += row[lane];
            ++nframe;
        }
        s_mean[lane] = sum / nframe; /* float / int, src/cmn.c:197 */
    }
    __syncthreads();
    float *out = P.out + (size_t)t0 * 3 * C;
    const int total = n * C;
    for (int idx = lane; idx < total; idx += 64) {
        const int t = idx / C, i = idx - t * C;
        const float mean = s_mean[i];
#define CEP(tt) (cep[(size_t)((tt) < 0 ? 0 : ((tt) >= n ? n - 1 : (tt))) * C + i] - mean)
        const float c0 = CEP(t);
        const float d = CEP(t + 2) - CEP(t - 2);
        const float d1 = CEP(t + 3) - CEP(t - 1);
        const float d2 = CEP(t + 1) - CEP(t - 3);
#undef CEP
        out[(size_t)t * 3 * C + i] = c0;
        out[(size_t)t * 3 * C + C + i] = d;
        out[(size_t)t * 3 * C + 2 * C + i] = d1 - d2;
    }
}

/* ---------------------------------------------------------------------------------- */
/* K2: forced-alignment Viterbi, one wave per utterance                                 */
/* ---------------------------------------------------------------------------------- */

struct AlignUtt {
    int frame_off, n_frames, phone_off, n_phones;
    long long tok_off; /* offset (in tokens) of this utterance's token stack */
};

struct AlignParams {
    const int16_t *senscr; /* [total_frames][n_sen] */
    const AlignUtt *utts;
    const uint16_t *senid; /* [total_phones][3] */
    const int16_t *tmatid; /* [total_phones] */
    const int32_t *sf, *ef;
    const uint8_t *tp; /* [n_tmat][12] */
    int2 *tokens;
    ssw_align_entry_t *state_io; /* [total_phones*3] */
    int32_t *status;
    int n_sen, n_utts, max_phones;
};

/* hmm_vit_eval_3st_lr, src/hmm.c:482-567.  n0..n2 are the NEGATED senone scores.  Written
 * with selects instead of the reference's nested ifs (a lone wave pays for every branch with
 * scalar exec-mask bookkeeping); the decision tree is the same, including the t2 that the state-2
 * block inherits from the exit block when it has no 0->2 arc of its own (:496,501-502,519-520):
 *   exit   only if s1 + n1 > WORST:  t1 = a2 + tp23, t2 = a1 + tp13 if that arc exists else
 *          INT_MIN;  take t1 iff t1 > t2 (history of state 2), else t2 (history of state 1)
 *   state2 t0 = a2 + tp22, t1 = a1 + tp12, t2 = a0 + tp02 if that arc exists, else the exit
 *          block's t2;  if t0 > t1: (t2 > t0 ? t2/h0 : t0/h2) else (t2 > t1 ? t2/h0 : t1/h1)
 *   state1 t0 = a1 + tp11, t1 = a0 + tp01;  t0 > t1 ? t0/h1 : t1/h0
 *   state0 a0 + tp00;  every new score clamped to WORST, best = max over them and the exit. */
__device__ __forceinline__ int
vit_eval_3st(int &s0, int &s1, int &s2, int &h0, int &h1, int &h2, int &os, int &oh, int n0,
             int n1, int n2, uint32_t tpa, uint32_t tpb, uint32_t tpc)
{
#define TPQ(word, j) (-(int)(((word) >> (8 * (j))) & 0xffu))
    const int tp00 = TPQ(tpa, 0), tp01 = TPQ(tpa, 1), tp02 = TPQ(tpa, 2);
    const int tp11 = TPQ(tpb, 1), tp12 = TPQ(tpb, 2), tp13 = TPQ(tpb, 3);
    const int tp22 = TPQ(tpc, 2), tp23 = TPQ(tpc, 3);
#undef TPQ
    const int a2 = s2 + n2, a1 = s1 + n1, a0 = s0 + n0;
    const int W = SSW_WORST_SCORE;

    /* exit */
    const bool a1_live = a1 > W;
    const int e1 = a2 + tp23;
    const int e2 = (a1_live && tp13 > -255) ? a1 + tp13 : INT_MIN;
    const bool from2 = e1 > e2;
    int s3 = from2 ? e1 : e2;
    s3 = s3 < W ? W : s3;
    const int oh_new = from2 ? h2 : h1;
    os = a1_live ? s3 : os;
    oh = a1_live ? oh_new : oh;
    int best = a1_live ? s3 : W;

    /* state 2 (uses h1, h2 as they were) */
    const int t0 = a2 + tp22, t1 = a1 + tp12;
    const int t2 = (tp02 > -255) ? a0 + tp02 : e2;
    const bool self2 = t0 > t1;
    const int base2 = self2 ? t0 : t1;
    const int hb2 = self2 ? h2 : h1;
    const bool skip2 = t2 > base2;
    int ns2 = skip2 ? t2 : base2;
    h2 = skip2 ? h0 : hb2;
    ns2 = ns2 < W ? W : ns2;
    best = ns2 > best ? ns2 : best;

    /* state 1 */
    const int u0 = a1 + tp11, u1 = a0 + tp01;
    const bool self1 = u0 > u1;
    int ns1 = self1 ? u0 : u1;
    h1 = self1 ? h1 : h0;
    ns1 = ns1 < W ? W : ns1;
    best = ns1 > best ? ns1 : best;

    /* state 0 */
    int ns0 = a0 + tp00;
    ns0 = ns0 < W ? W : ns0;
    best = ns0 > best ? ns0 : best;
    s0 = ns0;
    s1 = ns1;
    s2 = ns2;
    return best;
}

/* state_align_search_finish (state_align_search.c:215-268): one lane walks the token stack
 * back from frame n_frames - 2. */
__device__ __forceinline__ void
align_backtrace(const AlignParams &P, const AlignUtt &U, int u, const int2 *tok, int n_states,
                int final_id, int final_score)
{
    ssw_align_entry_t *st = P.state_io + (size_t)U.phone_off * 3;
    int last_id = final_id, cur_id = last_id;
    int last_score = final_score;
    int status = 0;
    if (last_id == -1) {
        status = -1;
    } else {
        int last_frame = U.n_frames;
        for (int cf = U.n_frames - 2; cf >= 0; --cf) {
            int2 cur = tok[(size_t)cf * n_states + cur_id];
            cur_id = cur.x;
            if (cur_id == -1) {
                status = -(2 + cf);
                break;
            }
            if (cur_id != last_id) {
                st[last_id].start = cf + 1;
                st[last_id].duration = last_frame - (cf + 1);
                st[last_id].score = last_score - cur.y;
                last_id = cur_id;
                last_score = cur.y;
                last_frame = cf + 1;
            }
        }
        if (status == 0) {
            st[0].start = 0;
            st[0].duration = last_frame;
        }
    }
    P.status[u] = status;
}

/* LDS layout: 16 int arrays of `P` (padded phone count) entries each. */
enum { A_S0, A_S1, A_S2, A_H0, A_H1, A_H2, A_OS, A_OH, A_FR, A_TPA, A_TPB, A_TPC, A_SEN01,
       A_SEN2, A_SF, A_EF, A_COUNT };

/* WMAX > 0: utterances of up to 64 * WMAX phones; every lane keeps the senone scores of its
 * phones for the current frame in registers and requests the next frame's at the top of each
 * frame, so the scattered 2-byte gathers from the score rows (DRAM latency: the rows were
 * written by another kernel long ago) are a whole frame step ahead of their use.  WMAX == 0:
 * any phone count, scores fetched where they are used. */
/* The alignment kernel runs one wave64 per workgroup: LDS operations of a wave are processed in
 * issue order, so ordering between lanes needs only a compiler-level fence, not s_barrier —
 * whose __syncthreads() form would also drain the outstanding token stores and score
 * prefetches every time. */
__device__ __forceinline__ void
wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int WMAX>
__global__ void __launch_bounds__(64)
viterbi_align_kernel(AlignParams P)
{
    extern __shared__ int lds[];
    const int u = blockIdx.x;
    const int lane = threadIdx.x;
    const AlignUtt U = P.utts[u];
    const int NP = U.n_phones;
    const int PP = P.max_phones + 1; /* +1: slot NP is a never-active sentinel neighbour */
    const int W = (NP + 63) >> 6;
    const int n_states = NP * 3;
#define L(arr, p) lds[(arr)*PP + (p)]

    for (int p = lane; p <= NP; p += 64) {
        bool real = p < NP;
        int gp = U.phone_off + p;
        L(A_S0, p) = SSW_WORST_SCORE; /* hmm_clear, src/hmm.c:124-140 */
        L(A_S1, p) = SSW_WORST_SCORE;
        L(A_S2, p) = SSW_WORST_SCORE;
        L(A_H0, p) = -1;
        L(A_H1, p) = -1;
        L(A_H2, p) = -1;
        L(A_OS, p) = SSW_WORST_SCORE;
        L(A_OH, p) = -1;
        L(A_FR, p) = -1;
        if (real) {
            const uint32_t *tp = reinterpret_cast<const uint32_t *>(P.tp)
                + (size_t)P.tmatid[gp] * 3;
            L(A_TPA, p) = (int)tp[0];
            L(A_TPB, p) = (int)tp[1];
            L(A_TPC, p) = (int)tp[2];
            L(A_SEN01, p) = (int)P.senid[gp * 3] | ((int)P.senid[gp * 3 + 1] << 16);
            L(A_SEN2, p) = (int)P.senid[gp * 3 + 2];
            L(A_SF, p) = P.sf[gp];
            L(A_EF, p) = P.ef[gp];
        } else {
            L(A_SF, p) = INT_MAX; /* nothing ever transitions into the sentinel */
            L(A_EF, p) = INT_MAX;
        }
    }
    wave_sync();
    if (lane == 0) { /* state_align_search_start: hmm_enter(hmms, 0, 0, 0) */
        L(A_S0, 0) = 0;
        L(A_H0, 0) = 0;
        L(A_FR, 0) = 0;
    }
    wave_sync();

    int2 *tok = P.tokens + U.tok_off;
    int best_score = 0;
    constexpr int WR = WMAX ? WMAX : 1;
    int sid01[WR], sid2[WR], cur01[WR], cur2[WR];
    if (WMAX) {
        const int16_t *row0 = P.senscr + (size_t)U.frame_off * P.n_sen;
#pragma unroll
        for (int w = 0; w < WR; ++w) {
            int p = w * 64 + lane;
            bool real = p < NP && U.n_frames > 0;
            sid01[w] = real ? L(A_SEN01, p) : 0;
            sid2[w] = real ? L(A_SEN2, p) : 0;
            cur01[w] = real ? ((int)(uint16_t)row0[sid01[w] & 0xffff]
                               | ((int)(uint16_t)row0[(sid01[w] >> 16) & 0xffff] << 16))
                            : 0;
            cur2[w] = real ? (int)row0[sid2[w]] : 0;
        }
    }
    for (int t = 0; t < U.n_frames; ++t) {
        const int16_t *row = P.senscr + (size_t)(U.frame_off + t) * P.n_sen;
        const int nf = t + 1;
        int nxt01[WR], nxt2[WR];
        if (WMAX) { /* next frame's scores (the last frame re-reads its own row) */
            const int16_t *rown = row + (nf < U.n_frames ? (size_t)P.n_sen : 0);
#pragma unroll
            for (int w = 0; w < WR; ++w) {
                bool real = w * 64 + lane < NP;
                nxt01[w] = real ? ((int)(uint16_t)rown[sid01[w] & 0xffff]
                                   | ((int)(uint16_t)rown[(sid01[w] >> 16) & 0xffff] << 16))
                                : 0;
                nxt2[w] = real ? (int)rown[sid2[w]] : 0;
            }
        }
        const bool renorm = (best_score - 0x300000) < SSW_WORST_SCORE;
        int bs = SSW_WORST_SCORE;

        /* renormalize_hmms + evaluate_hmms + prune_hmms (state_align_search.c:57-106) */
#pragma unroll
        for (int w = 0; w < (WMAX ? WMAX : W); ++w) {
            int p = w * 64 + lane;
            if (p < NP) {
                int s0 = L(A_S0, p), s1 = L(A_S1, p), s2 = L(A_S2, p), os = L(A_OS, p);
                int fr = L(A_FR, p);
                if (renorm) { /* hmm_normalize, src/hmm.c:150-161 */
                    if (s0 > SSW_WORST_SCORE)
                        s0 -= best_score;
                    if (s1 > SSW_WORST_SCORE)
                        s1 -= best_score;
                    if (s2 > SSW_WORST_SCORE)
                        s2 -= best_score;
                    if (os > SSW_WORST_SCORE)
                        os -= best_score;
                }
                if (fr >= t) {
                    int h0 = L(A_H0, p), h1 = L(A_H1, p), h2 = L(A_H2, p), oh = L(A_OH, p);
                    int n0, n1, n2;
                    if (WMAX) {
                        n0 = -(int)(int16_t)(cur01[w < WR ? w : 0] & 0xffff);
                        n1 = -(cur01[w < WR ? w : 0] >> 16);
                        n2 = -cur2[w < WR ? w : 0];
                    } else {
                        int sen01 = L(A_SEN01, p), sen2 = L(A_SEN2, p);
                        n0 = -(int)row[sen01 & 0xffff];
                        n1 = -(int)row[(sen01 >> 16) & 0xffff];
                        n2 = -(int)row[sen2];
                    }
                    int b = vit_eval_3st(s0, s1, s2, h0, h1, h2, os, oh, n0, n1, n2,
                                         (uint32_t)L(A_TPA, p), (uint32_t)L(A_TPB, p),
                                         (uint32_t)L(A_TPC, p));
                    bs = b > bs ? b : bs;
                    L(A_H1, p) = h1;
                    L(A_H2, p) = h2;
                    L(A_OH, p) = oh;
                    if (nf <= L(A_EF, p))
                        L(A_FR, p) = nf;
                }
                L(A_S0, p) = s0;
                L(A_S1, p) = s1;
                L(A_S2, p) = s2;
                L(A_OS, p) = os;
            }
        }
        best_score = wave_max_i32(bs);
        wave_sync();

        /* phone_transition (state_align_search.c:108-133) as a carry chain, then
         * record_transitions (:149-175).  entered(i+1) = C_i & (A_i | entered(i)) is the carry
         * recurrence of the binary sum X + Y with X = C, Y = A & C. */
        unsigned long long cin = 0;
        int2 *tkrow = tok + (size_t)t * n_states;
        for (int w = 0; w < W; ++w) {
            int p = w * 64 + lane;
            bool valid = p < NP;
            int fr = valid ? L(A_FR, p) : -1;
            bool a_bit = valid && fr == nf;
            bool c_bit = false;
            if (valid && p + 1 < NP) {
                int nfr = L(A_FR, p + 1);
                c_bit = (nf >= L(A_SF, p + 1)) && (nfr < t || L(A_OS, p) > L(A_S0, p + 1));
            }
            unsigned long long A = __ballot(a_bit), Cm = __ballot(c_bit);
            unsigned long long X = Cm, Y = A & Cm;
            unsigned long long S = X + Y + cin;
            unsigned long long E = S ^ X ^ Y; /* bit i: phone (w*64+i) is entered */
            unsigned long long cout = ((X & Y) | ((X | Y) & ~S)) >> 63;
            bool entered = (E >> lane) & 1ull;
            int src_os = 0, src_oh = 0;
            if (valid && entered) { /* p >= 1 whenever entered */
                src_os = L(A_OS, p - 1);
                src_oh = L(A_OH, p - 1);
            }
            wave_sync(); /* all reads of neighbours done before this word's writes */
            if (valid) {
                if (entered) { /* hmm_enter, src/hmm.c:142-148 */
                    L(A_S0, p) = src_os;
                    L(A_H0, p) = src_oh;
                    L(A_FR, p) = nf;
                    fr = nf;
                }
                int2 k0 = make_int2(-1, -1), k1 = k0, k2 = k0;
                if (fr >= t) {
                    k0 = make_int2(L(A_H0, p), L(A_S0, p));
                    k1 = make_int2(L(A_H1, p), L(A_S1, p));
                    k2 = make_int2(L(A_H2, p), L(A_S2, p));
                    L(A_H0, p) = p * 3;
                    L(A_H1, p) = p * 3 + 1;
                    L(A_H2, p) = p * 3 + 2;
                }
                tkrow[p * 3] = k0;
                tkrow[p * 3 + 1] = k1;
                tkrow[p * 3 + 2] = k2;
            }
            cin = cout;
            wave_sync();
        }
        if (WMAX) {
#pragma unroll
            for (int w = 0; w < WR; ++w) {
                cur01[w] = nxt01[w];
                cur2[w] = nxt2[w];
            }
        }
    }

    /* state_align_search_finish (state_align_search.c:215-268) */
    /* the token stack is read back by this workgroup only: workgroup-scope ordering (an
     * agent-scope fence would write back and invalidate the XCD's L2 under the other
     * utterances) */
    __syncthreads();
    if (lane == 0)
        align_backtrace(P, U, u, tok, n_states, L(A_OH, NP - 1), L(A_OS, NP - 1));
#undef L
}

/* The same search with every phone's HMM in registers (lane = phone, WMAX words of 64 phones):
 * nothing of the frame step goes through LDS.  Neighbour values move by one lane with DPP wave
 * shifts, whose `old` operand supplies the value that crosses a 64-phone word boundary.  All of
 * phone_transition's reads see the state left by evaluate/prune (as in the reference's loop,
 * where hmm i+1 is examined before it is entered); the enters are applied afterwards. */
__device__ __forceinline__ int
lane_from_next(int v, int edge) /* lane i <- lane i+1, lane 63 <- edge */
{
    return __builtin_amdgcn_update_dpp(edge, v, 0x130, 0xf, 0xf, false);
}

__device__ __forceinline__ int
lane_from_prev(int v, int edge) /* lane i <- lane i-1, lane 0 <- edge */
{
    return __builtin_amdgcn_update_dpp(edge, v, 0x138, 0xf, 0xf, false);
}

template <int WMAX>
__global__ void __launch_bounds__(64)
viterbi_align_reg_kernel(AlignParams P)
{
    const int u = blockIdx.x;
    const int lane = threadIdx.x;
    const AlignUtt U = P.utts[u];
    const int NP = U.n_phones;
    const int n_states = NP * 3;

    int s0[WMAX], s1[WMAX], s2[WMAX], h0[WMAX], h1[WMAX], h2[WMAX], os[WMAX], oh[WMAX], fr[WMAX];
    uint32_t tpa[WMAX], tpb[WMAX], tpc[WMAX];
    int sid01[WMAX], sid2[WMAX], sf_next[WMAX], ef[WMAX], cur01[WMAX], cur2[WMAX];
    const int16_t *row0 = P.senscr + (size_t)U.frame_off * P.n_sen;
#pragma unroll
    for (int w = 0; w < WMAX; ++w) {
        const int p = w * 64 + lane;
        const bool real = p < NP;
        const int gp = U.phone_off + (real ? p : 0);
        s0[w] = s1[w] = s2[w] = os[w] = SSW_WORST_SCORE; /* hmm_clear, src/hmm.c:124-140 */
        h0[w] = h1[w] = h2[w] = oh[w] = -1;
        fr[w] = -1;
        const uint32_t *tp = reinterpret_cast<const uint32_t *>(P.tp) + (size_t)P.tmatid[gp] * 3;
        tpa[w] = real ? tp[0] : 0u;
        tpb[w] = real ? tp[1] : 0u;
        tpc[w] = real ? tp[2] : 0u;
        sid01[w] = real ? ((int)P.senid[gp * 3] | ((int)P.senid[gp * 3 + 1] << 16)) : 0;
        sid2[w] = real ? (int)P.senid[gp * 3 + 2] : 0;
        ef[w] = real ? P.ef[gp] : INT_MAX;
        sf_next[w] = (p + 1 < NP) ? P.sf[gp + 1] : INT_MAX; /* nothing enters past the end */
        const bool have = real && U.n_frames > 0;
        cur01[w] = have ? ((int)(uint16_t)row0[sid01[w] & 0xffff]
                           | ((int)(uint16_t)row0[(sid01[w] >> 16) & 0xffff] << 16))
                        : 0;
        cur2[w] = have ? (int)row0[sid2[w]] : 0;
    }
    if (lane == 0) { /* state_align_search_start: hmm_enter(hmms, 0, 0, 0) */
        s0[0] = 0;
        h0[0] = 0;
        fr[0] = 0;
    }

    int2 *tok = P.tokens + U.tok_off;
    int best_score = 0;
    for (int t = 0; t < U.n_frames; ++t) {
        const int nf = t + 1;
        const int16_t *rown
            = P.senscr + (size_t)(U.frame_off + (nf < U.n_frames ? nf : t)) * P.n_sen;
        int nxt01[WMAX], nxt2[WMAX];
#pragma unroll
        for (int w = 0; w < WMAX; ++w) {
            const bool real = w * 64 + lane < NP;
            nxt01[w] = real ? ((int)(uint16_t)rown[sid01[w] & 0xffff]
                               | ((int)(uint16_t)rown[(sid01[w] >> 16) & 0xffff] << 16))
                            : 0;
            nxt2[w] = real ? (int)rown[sid2[w]] : 0;
        }
        const bool renorm = (best_score - 0x300000) < SSW_WORST_SCORE;
        int bs = SSW_WORST_SCORE;

        /* renormalize_hmms + evaluate_hmms + prune_hmms (state_align_search.c:57-106) */
#pragma unroll
        for (int w = 0; w < WMAX; ++w) {
            if (w * 64 + lane < NP) {
                if (renorm) { /* hmm_normalize, src/hmm.c:150-161 */
                    if (s0[w] > SSW_WORST_SCORE)
                        s0[w] -= best_score;
                    if (s1[w] > SSW_WORST_SCORE)
                        s1[w] -= best_score;
                    if (s2[w] > SSW_WORST_SCORE)
                        s2[w] -= best_score;
                    if (os[w] > SSW_WORST_SCORE)
                        os[w] -= best_score;
                }
                if (fr[w] >= t) {
                    const int n0 = -(int)(int16_t)(cur01[w] & 0xffff);
                    const int n1 = -(cur01[w] >> 16);
                    const int n2 = -cur2[w];
                    int b = vit_eval_3st(s0[w], s1[w], s2[w], h0[w], h1[w], h2[w], os[w], oh[w],
                                         n0, n1, n2, tpa[w], tpb[w], tpc[w]);
                    bs = b > bs ? b : bs;
                    if (nf <= ef[w])
                        fr[w] = nf;
                }
            }
        }
        best_score = wave_max_i32(bs);

        /* phone_transition (state_align_search.c:108-133) as a carry chain, then
         * record_transitions (:149-175).  entered(i+1) = C_i & (A_i | entered(i)) is the carry
         * recurrence of the binary sum X + Y with X = C, Y = A & C. */
        unsigned long long Am[WMAX], Cm[WMAX];
#pragma unroll
        for (int w = 0; w < WMAX; ++w) {
            const int p = w * 64 + lane;
            /* frame and entry score of phone p + 1 */
            const int efr = (w + 1 < WMAX) ? __builtin_amdgcn_readlane(fr[w + 1 < WMAX ? w + 1 : w], 0) : -1;
            const int es0 = (w + 1 < WMAX) ? __builtin_amdgcn_readlane(s0[w + 1 < WMAX ? w + 1 : w], 0) : 0;
            const int nfr = lane_from_next(fr[w], efr);
            const int ns0 = lane_from_next(s0[w], es0);
            const bool a_bit = p < NP && fr[w] == nf;
            const bool c_bit = p + 1 < NP && nf >= sf_next[w] && (nfr < t || os[w] > ns0);
            Am[w] = __ballot(a_bit);
            Cm[w] = __ballot(c_bit);
        }
        unsigned long long cin = 0;
        int2 *tkrow = tok + (size_t)t * n_states;
        int prev_os = 0, prev_oh = 0; /* exit score/history of the last phone of the previous word */
#pragma unroll
        for (int w = 0; w < WMAX; ++w) {
            const int p = w * 64 + lane;
            const unsigned long long X = Cm[w], Y = Am[w] & Cm[w];
            const unsigned long long S = X + Y + cin;
            const unsigned long long E = S ^ X ^ Y; /* bit i: phone (w*64+i) is entered */
            cin = ((X & Y) | ((X | Y) & ~S)) >> 63;
            const bool entered = (E >> lane) & 1ull;
            const int src_os = lane_from_prev(os[w], prev_os);
            const int src_oh = lane_from_prev(oh[w], prev_oh);
            prev_os = __builtin_amdgcn_readlane(os[w], 63);
            prev_oh = __builtin_amdgcn_readlane(oh[w], 63);
            if (p < NP) {
                if (entered) { /* hmm_enter, src/hmm.c:142-148 */
                    s0[w] = src_os;
                    h0[w] = src_oh;
                    fr[w] = nf;
                }
                int2 k0 = make_int2(-1, -1), k1 = k0, k2 = k0;
                if (fr[w] >= t) {
                    k0 = make_int2(h0[w], s0[w]);
                    k1 = make_int2(h1[w], s1[w]);
                    k2 = make_int2(h2[w], s2[w]);
                    h0[w] = p * 3;
                    h1[w] = p * 3 + 1;
                    h2[w] = p * 3 + 2;
                }
                tkrow[p * 3] = k0;
                tkrow[p * 3 + 1] = k1;
                tkrow[p * 3 + 2] = k2;
            }
        }
#pragma unroll
        for (int w = 0; w < WMAX; ++w) {
            cur01[w] = nxt01[w];
            cur2[w] = nxt2[w];
        }
    }

    /* state_align_search_finish (state_align_search.c:215-268) */
    const int lw = (NP - 1) >> 6, ll = (NP - 1) & 63;
    int fin_oh = -1, fin_os = SSW_WORST_SCORE;
#pragma unroll
    for (int w = 0; w < WMAX; ++w)
        if (w == lw) {
            fin_oh = __shfl(oh[w], ll, WAVE);
            fin_os = __shfl(os[w], ll, WAVE);
        }
    /* the token stack is read back by this workgroup only: workgroup-scope ordering (an
     * agent-scope fence would write back and invalidate the XCD's L2 under the other
     * utterances) */
    __syncthreads();
    if (lane == 0)
        align_backtrace(P, U, u, tok, n_states, fin_oh, fin_os);
}

/* One wave per 64-phone word of the utterance (workgroup = n_words waves, up to 16): the frame
 * step of every word runs in parallel, HMMs in registers as above.  Per frame the waves meet
 * twice at an LDS-only barrier (no drain of the outstanding token stores): once to publish their
 * boundary values (frame / entry score of their first phone, exit score / history of their last)
 * and their best score, once to publish the A and C masks of phone_transition, after which every
 * wave folds the carry chain up to its own word.  The exchange slots are double-buffered by
 * frame parity: a wave can be at most one barrier ahead of the slowest one. */
#define SSW_ALIGN_MAX_WAVES 16

__global__ void __launch_bounds__(64 * SSW_ALIGN_MAX_WAVES)
viterbi_align_mw_kernel(AlignParams P)
{
    __shared__ int x_bs[2][SSW_ALIGN_MAX_WAVES], x_fr0[2][SSW_ALIGN_MAX_WAVES],
        x_s00[2][SSW_ALIGN_MAX_WAVES], x_os63[2][SSW_ALIGN_MAX_WAVES],
        x_oh63[2][SSW_ALIGN_MAX_WAVES];
    __shared__ unsigned long long x_A[2][SSW_ALIGN_MAX_WAVES], x_C[2][SSW_ALIGN_MAX_WAVES];
    __shared__ int x_fin[2];
    const int u = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
    const AlignUtt U = P.utts[u];
    const int NP = U.n_phones;
    const int n_states = NP * 3;
    const int p = w * 64 + lane;
    const bool real = p < NP;
    const int gp = U.phone_off + (real ? p : 0);

    int s0 = SSW_WORST_SCORE, s1 = SSW_WORST_SCORE, s2 = SSW_WORST_SCORE, os = SSW_WORST_SCORE;
    int h0 = -1, h1 = -1, h2 = -1, oh = -1, fr = -1; /* hmm_clear, src/hmm.c:124-140 */
    const uint32_t *tp = reinterpret_cast<const uint32_t *>(P.tp) + (size_t)P.tmatid[gp] * 3;
    const uint32_t tpa = real ? tp[0] : 0u, tpb = real ? tp[1] : 0u, tpc = real ? tp[2] : 0u;
    const int sid01 = real ? ((int)P.senid[gp * 3] | ((int)P.senid[gp * 3 + 1] << 16)) : 0;
    const int sid2 = real ? (int)P.senid[gp * 3 + 2] : 0;
    const int ef = real ? P.ef[gp] : INT_MAX;
    const int sf_next = (p + 1 < NP) ? P.sf[gp + 1] : INT_MAX;
    /* senone scores of this lane's phone: three rotating register sets, so that the scattered
     * 2-byte gathers of frame t + 2 are requested at the top of frame t (a frame step is shorter
     * than a DRAM round trip) */
    const int last = U.n_frames - 1;
    auto fetch = [&](int t, int &v01, int &v2) {
        const int16_t *row = P.senscr + (size_t)(U.frame_off + (t < last ? t : last)) * P.n_sen;
        const bool have = real && U.n_frames > 0;
        v01 = have ? ((int)(uint16_t)row[sid01 & 0xffff]
                      | ((int)(uint16_t)row[(sid01 >> 16) & 0xffff] << 16))
                   : 0;
        v2 = have ? (int)row[sid2] : 0;
    };
    int a01, a2, b01, b2, c01, c2;
    fetch(0, a01, a2);
    fetch(1, b01, b2);
    if (p == 0) { /* state_align_search_start: hmm_enter(hmms, 0, 0, 0) */
        s0 = 0;
        h0 = 0;
        fr = 0;
    }

    int2 *tok = P.tokens + U.tok_off;
    int best_score = 0;
    /* The frame step is written with selects rather than branches: a lone wave pays for every
     * divergent `if` with scalar exec-mask bookkeeping, and lanes without a phone hold an inert
     * HMM (scores WORST, frame -1) that the arithmetic leaves inert.  Only stores are masked. */
    int2 *tkrow = tok + (real ? p * 3 : 0);
    auto frame = [&](const int t, const int cur01, const int cur2, int &fut01, int &fut2) {
        const int nf = t + 1, par = t & 1;
        const int W = SSW_WORST_SCORE;
        fetch(t + 2, fut01, fut2);
        /* renormalize_hmms (hmm_normalize, src/hmm.c:150-161) */
        const bool renorm = (best_score - 0x300000) < W;
        s0 = (renorm && s0 > W) ? s0 - best_score : s0;
        s1 = (renorm && s1 > W) ? s1 - best_score : s1;
        s2 = (renorm && s2 > W) ? s2 - best_score : s2;
        os = (renorm && os > W) ? os - best_score : os;
        /* evaluate_hmms + prune_hmms (state_align_search.c:57-106): evaluated for every lane,
         * kept for the phones that are active in this frame */
        const bool active = fr >= t;
        int e0 = s0, e1 = s1, e2 = s2, g0 = h0, g1 = h1, g2 = h2, eos = os, eoh = oh;
        const int b = vit_eval_3st(e0, e1, e2, g0, g1, g2, eos, eoh,
                                   -(int)(int16_t)(cur01 & 0xffff), -(cur01 >> 16), -cur2, tpa,
                                   tpb, tpc);
        s0 = active ? e0 : s0;
        s1 = active ? e1 : s1;
        s2 = active ? e2 : s2;
        h1 = active ? g1 : h1;
        h2 = active ? g2 : h2;
        os = active ? eos : os;
        oh = active ? eoh : oh;
        fr = (active && nf <= ef) ? nf : fr;
        int bs = wave_max_dpp(active ? b : W);
        if (lane == 0) {
            x_bs[par][w] = bs;
            x_fr0[par][w] = fr;
            x_s00[par][w] = s0;
        }
        if (lane == 63) {
            x_os63[par][w] = os;
            x_oh63[par][w] = oh;
        }
        lds_barrier();
        best_score = W;
        for (int k = 0; k < nw; ++k) {
            int v = x_bs[par][k];
            best_score = v > best_score ? v : best_score;
        }

        /* phone_transition (state_align_search.c:108-133): A/C masks of this word */
        const int efr = w + 1 < nw ? x_fr0[par][w + 1] : -1;
        const int es0 = w + 1 < nw ? x_s00[par][w + 1] : 0;
        const int nfr = lane_from_next(fr, efr);
        const int ns0 = lane_from_next(s0, es0);
        const bool a_bit = fr == nf; /* lanes without a phone keep frame -1 */
        const bool c_bit = p + 1 < NP && nf >= sf_next && (nfr < t || os > ns0);
        const unsigned long long Am = __ballot(a_bit), Cm = __ballot(c_bit);
        if (lane == 0) {
            x_A[par][w] = Am;
            x_C[par][w] = Cm;
        }
        lds_barrier();
        /* carry chain over the words before this one, then this word's enters */
        unsigned long long cin = 0;
        for (int k = 0; k < w; ++k) {
            const unsigned long long X = x_C[par][k], Y = x_A[par][k] & X;
            const unsigned long long S = X + Y + cin;
            cin = ((X & Y) | ((X | Y) & ~S)) >> 63;
        }
        const unsigned long long X = Cm, Y = Am & Cm;
        const unsigned long long E = (X + Y + cin) ^ X ^ Y; /* bit i: phone (w*64+i) is entered */
        const bool entered = real && ((E >> lane) & 1ull);
        const int src_os = lane_from_prev(os, w > 0 ? x_os63[par][w > 0 ? w - 1 : 0] : 0);
        const int src_oh = lane_from_prev(oh, w > 0 ? x_oh63[par][w > 0 ? w - 1 : 0] : 0);
        /* hmm_enter (src/hmm.c:142-148), then record_transitions (:149-175) */
        s0 = entered ? src_os : s0;
        h0 = entered ? src_oh : h0;
        fr = entered ? nf : fr;
        const bool rec = fr >= t;
        const int2 k0 = make_int2(rec ? h0 : -1, rec ? s0 : -1);
        const int2 k1 = make_int2(rec ? h1 : -1, rec ? s1 : -1);
        const int2 k2 = make_int2(rec ? h2 : -1, rec ? s2 : -1);
        h0 = rec ? p * 3 : h0;
        h1 = rec ? p * 3 + 1 : h1;
        h2 = rec ? p * 3 + 2 : h2;
        if (real) {
            tkrow[0] = k0;
            tkrow[1] = k1;
            tkrow[2] = k2;
        }
        tkrow += n_states;
    };
    for (int t = 0; t < U.n_frames; t += 3) { /* the same trip count in every wave: barriers */
        frame(t, a01, a2, c01, c2);
        if (t + 1 < U.n_frames)
            frame(t + 1, b01, b2, a01, a2);
        if (t + 2 < U.n_frames)
            frame(t + 2, c01, c2, b01, b2);
    }

    /* state_align_search_finish (state_align_search.c:215-268) */
    if (p == NP - 1) {
        x_fin[0] = oh;
        x_fin[1] = os;
    }
    /* the token stack is read back by this workgroup only: workgroup-scope ordering (an
     * agent-scope fence would write back and invalidate the XCD's L2 under the other
     * utterances) */
    __syncthreads();
    if (threadIdx.x == 0)
        align_backtrace(P, U, u, tok, n_states, x_fin[0], x_fin[1]);
}

} // namespace

/* ==================================================================================== */
/* Host side: device model + C ABI                                                      */
/* ==================================================================================== */

struct ssw_model_s {
    ssw_host_model_t *h;
    int device;
    int n_cbf, sen_stride;
    float *d_rec;
    float *d_recq;   /* the same densities as a quadratic form in x (the speculative scan) */
    float *d_recmax; /* [n_cbf][SSW_REC_FLOATS] per-(codebook, stream) constants of the scan: [0] = d0 */
    uint32_t *d_exlist; /* [n_cbf][SSW_EXLIST_STRIDE] densities the scan leaves to the exact form */
    int n_exact_form;
    uint8_t *d_mixw, *d_ms_pdf, *d_sen2cb, *d_logadd8, *d_tp, *d_quad_cb;
    short4 *d_slot_sen;
    int n_quads, slot_stride;
    /* scoring workspace */
    uint32_t *d_topn_cw;
    int4 *d_topn_sc;
    int *d_utt_off;
    uint32_t *d_flags;          /* bit per (frame, cbf): needs the exact pass */
    uint32_t *d_work;           /* the same pairs as a list (fix-up work items) */
    uint32_t *d_utt_start;      /* bit per frame: first frame of an utterance */
    unsigned long long *d_nfixed; /* [0],[1] exact-pass counters, [2] fill count of d_work */
    size_t ws_frames, ws_utts;
    std::vector<int32_t> *utt_cache; /* last uploaded utterance offsets */
    int force_exact;            /* SSW_PTM_EXACT=1: always run the sequential chain kernel */
    int ms_raw;                 /* next ms launch leaves its scores un-normalised */
    int stats_pending;
    int last_n_frames;
    int64_t stats[2];
    /* optional per-kernel event timing */
    int timing;
    hipEvent_t ev[3];
    /* alignment workspace (ssw_align_batch) */
    unsigned char *d_align_ws;
    size_t align_ws_bytes;
    /* host-API staging */
    float *d_feats;
    int16_t *d_out;
    size_t st_frames;
};

template <typename T>
static int
dev_alloc(T **p, size_t n)
{
    HIP_OK(hipMalloc(reinterpret_cast<void **>(p), n * sizeof(T)));
    return 0;
}

static int
upload_model(ssw_model_s *m)
{
    const ssw_host_model_t *h = m->h;
    const int ncbf = h->n_cb * h->n_feat;
    m->n_cbf = ncbf;
    /* Gaussian records */
    std::vector<float> rec((size_t)ncbf * h->n_density * SSW_REC_FLOATS, 0.0f);
    const float *mp = h->mean, *vp = h->var;
    for (int c = 0; c < h->n_cb; ++c)
        for (int f = 0; f < h->n_feat; ++f)
            for (int d = 0; d < h->n_density; ++d) {
                float *r = rec.data()
                    + (((size_t)c * h->n_feat + f) * h->n_density + d) * SSW_REC_FLOATS;
                for (int j = 0; j < h->veclen[f]; ++j) {
                    r[j] = *mp++;
                    r[SSW_REC_VAR + j] = *vp++;
                }
                r[SSW_REC_DET] = h->det[((size_t)c * h->n_feat + f) * h->n_density + d];
            }
    if (dev_alloc(&m->d_rec, rec.size()) < 0)
        return -1;
    HIP_OK(hipMemcpy(m->d_rec, rec.data(), rec.size() * sizeof(float), hipMemcpyHostToDevice));
    {
        /* The speculative scan evaluates  det - sum var (x - mean)^2  as a quadratic form in x,
         *   key = c + sum_j (a_j x_j + b_j x_j^2),  a = 2 var mean,  b = -var,
         *   c = (det - d0) - R + bias,  R = sum var mean^2,  d0 = the codebook's median det,
         * with 26 fused multiply-adds.  With u = 2^-24, S = sum var (x - mean)^2 and
         * M = |det - d0| + R + sum |a x| + sum |b| x^2 <= |det - d0| + 6 R + 3 S
         * (Cauchy-Schwarz), the form is within 27 u M of the real number and the reference's
         * fp32 value within 13 u |det| + 17 u S of it (13 subtractions whose partial sums lie
         * between det and the result; (1+u)^4 on every product); with
         * S <= |det - d0| + |value - d0| that is u (125 |det - d0| + 162 R + 13 |det|) -- folded
         * into c as `bias` with a few per cent of slack, so the key is an upper bound -- plus
         * 98 u |value - d0|, which the kernel adds (104 u |key| + 0.001: |value| and |key| differ
         * by at most the bias) to the one key it uses as a bound.  Densities whose bias would exceed 4 score units (floored variances far from
         * the origin) get an inert scan record and go on the codebook's exact-form list: the
         * kernel evaluates them the reference's way after the scan. */
        std::vector<float> rq(rec.size(), 0.0f), rmax((size_t)ncbf * SSW_REC_FLOATS, 0.0f);
        std::vector<uint32_t> exl((size_t)ncbf * SSW_EXLIST_STRIDE, 0u);
        const double u24 = 1.0 / 16777216.0;
        int n_exact_form = 0;
        for (int cbf = 0; cbf < ncbf; ++cbf) {
            std::vector<float> dets((size_t)h->n_density);
            for (int d = 0; d < h->n_density; ++d)
                dets[d] = rec[((size_t)cbf * h->n_density + d) * SSW_REC_FLOATS + SSW_REC_DET];
            std::nth_element(dets.begin(), dets.begin() + dets.size() / 2, dets.end());
            const float d0 = dets[dets.size() / 2];
            rmax[(size_t)cbf * SSW_REC_FLOATS] = d0;
            for (int d = 0; d < h->n_density; ++d) {
                const float *r = rec.data() + ((size_t)cbf * h->n_density + d) * SSW_REC_FLOATS;
                float *q = rq.data() + ((size_t)cbf * h->n_density + d) * SSW_REC_FLOATS;
                const double det = r[SSW_REC_DET], delta = det - (double)d0;
                double R = 0.0;
                bool finite = std::isfinite(det);
                for (int j = 0; j < SSW_MAX_VECLEN; ++j) {
                    double mean = r[j], var = r[SSW_REC_VAR + j];
                    R += fabs(var) * mean * mean;
                    finite = finite && std::isfinite(mean) && std::isfinite(var) && var >= 0.0;
                }
                const double bias = 1.05 * u24 * (126.0 * fabs(delta) + 164.0 * R + 14.0 * fabs(det));
                if (!finite || !(bias <= 4.0)) {
                    uint32_t *xl = exl.data() + (size_t)cbf * SSW_EXLIST_STRIDE;
                    xl[1 + xl[0]++] = (uint32_t)d;
                    q[SSW_REC_DET] = -3.0e38f; /* a = b = 0: the key stays out of the way */
                    ++n_exact_form;
                    continue;
                }
                for (int j = 0; j < SSW_MAX_VECLEN; ++j) {
                    q[j] = (float)(2.0 * (double)r[SSW_REC_VAR + j] * (double)r[j]);
                    q[SSW_REC_VAR + j] = -r[SSW_REC_VAR + j];
                }
                const double c = delta - R + bias;
                float cf = (float)c;
                if ((double)cf < c)
                    cf = nextafterf(cf, INFINITY);
                q[SSW_REC_DET] = cf;
            }
        }
        m->n_exact_form = n_exact_form;
        if (dev_alloc(&m->d_recq, rq.size()) < 0 || dev_alloc(&m->d_recmax, rmax.size()) < 0
            || dev_alloc(&m->d_exlist, exl.size()) < 0)
            return -1;
        HIP_OK(hipMemcpy(m->d_exlist, exl.data(), exl.size() * sizeof(uint32_t),
                         hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(m->d_recq, rq.data(), rq.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(m->d_recmax, rmax.data(), rmax.size() * sizeof(float),
                         hipMemcpyHostToDevice));
    }
    HIP_OK(hipMalloc((void **)&m->d_logadd8, 256));
    HIP_OK(hipMemcpy(m->d_logadd8, h->logadd8, 256, hipMemcpyHostToDevice));
    if (h->n_sen) {
        std::vector<uint8_t> s2c((size_t)h->n_sen);
        for (int i = 0; i < h->n_sen; ++i)
            s2c[i] = (uint8_t)h->sen2cb[i];
        HIP_OK(hipMalloc((void **)&m->d_sen2cb, s2c.size()));
        HIP_OK(hipMemcpy(m->d_sen2cb, s2c.data(), s2c.size(), hipMemcpyHostToDevice));
    }
    if (h->ptm_mixw || h->ms_pdf) {
        /* slot order: senones grouped by codebook (ascending id inside a group), every group
         * padded to a multiple of 4 slots */
        std::vector<int16_t> slot_sen;
        std::vector<uint8_t> quad_cb;
        for (int c = 0; c < h->n_cb; ++c) {
            size_t start = slot_sen.size();
            for (int i = 0; i < h->n_sen; ++i)
                if (h->sen2cb[i] == c)
                    slot_sen.push_back((int16_t)i);
            while ((slot_sen.size() - start) % 4)
                slot_sen.push_back(-1);
            for (size_t q = start / 4; q < slot_sen.size() / 4; ++q)
                quad_cb.push_back((uint8_t)c);
        }
        m->n_quads = (int)quad_cb.size();
        m->slot_stride = ((int)slot_sen.size() + 127) & ~127;
        m->sen_stride = m->slot_stride;
        const size_t rows = (size_t)h->n_feat * h->n_density;
        if (h->ptm_mixw) {
            std::vector<uint8_t> mw(rows * m->slot_stride, 0);
            for (size_t r = 0; r < rows; ++r)
                for (size_t sl = 0; sl < slot_sen.size(); ++sl)
                    if (slot_sen[sl] >= 0)
                        mw[r * m->slot_stride + sl] = h->ptm_mixw[r * h->n_sen + slot_sen[sl]];
            HIP_OK(hipMalloc((void **)&m->d_mixw, mw.size()));
            HIP_OK(hipMemcpy(m->d_mixw, mw.data(), mw.size(), hipMemcpyHostToDevice));
        }
        if (h->ms_pdf) {
            /* pdf[sen][feat][cw] (src/ms_senone.c:145-148) transposed to [feat][cw][slot] */
            std::vector<uint8_t> mw(rows * m->slot_stride, 0);
            for (size_t sl = 0; sl < slot_sen.size(); ++sl) {
                if (slot_sen[sl] < 0)
                    continue;
                const uint8_t *src = h->ms_pdf + (size_t)slot_sen[sl] * rows;
                for (size_t r = 0; r < rows; ++r)
                    mw[r * m->slot_stride + sl] = src[r];
            }
            HIP_OK(hipMalloc((void **)&m->d_ms_pdf, mw.size()));
            HIP_OK(hipMemcpy(m->d_ms_pdf, mw.data(), mw.size(), hipMemcpyHostToDevice));
        }
        HIP_OK(hipMalloc((void **)&m->d_quad_cb, quad_cb.size()));
        HIP_OK(hipMemcpy(m->d_quad_cb, quad_cb.data(), quad_cb.size(), hipMemcpyHostToDevice));
        HIP_OK(hipMalloc((void **)&m->d_slot_sen, slot_sen.size() * sizeof(int16_t)));
        HIP_OK(hipMemcpy(m->d_slot_sen, slot_sen.data(), slot_sen.size() * sizeof(int16_t),
                         hipMemcpyHostToDevice));
    }
    if (h->tp) {
        size_t n = (size_t)h->tp_n_tmat * h->tp_n_state * (h->tp_n_state + 1);
        HIP_OK(hipMalloc((void **)&m->d_tp, n + 16));
        HIP_OK(hipMemcpy(m->d_tp, h->tp, n, hipMemcpyHostToDevice));
    }
    return 0;
}

extern "C" ssw_model_t *
ssw_model_load(const char *mdef, const char *means, const char *variances, const char *sendump,
               const char *mixw, const char *tmat, const ssw_config_t *cfg)
{
    ssw_host_model_t *h = ssw_host_model_load(mdef, means, variances, sendump, mixw, tmat, cfg);
    if (h == NULL)
        return NULL;
    for (int f = 0; f < h->n_feat; ++f)
        if (h->veclen[f] > SSW_MAX_VECLEN) {
            ssw_set_error("stream %d has %d dimensions; the gfx950 kernels handle <= %d", f,
                          h->veclen[f], SSW_MAX_VECLEN);
            ssw_host_model_free(h);
            return NULL;
        }
    if (h->cfg.device == SSW_DEVICE_NONE) {
        /* host tables only (loader checks without a GPU); every compute entry point refuses */
        ssw_model_s *hm = new ssw_model_s();
        memset(hm, 0, sizeof(*hm));
        hm->h = h;
        hm->device = SSW_DEVICE_NONE;
        hm->n_cbf = h->n_cb * h->n_feat;
        return hm;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        ssw_set_error("no HIP device: the MI355X path has no CPU fallback");
        ssw_host_model_free(h);
        return NULL;
    }
    ssw_model_s *m = new ssw_model_s();
    memset(m, 0, sizeof(*m));
    m->h = h;
    if (h->cfg.device >= 0) {
        if (hipSetDevice(h->cfg.device) != hipSuccess) {
            ssw_set_error("hipSetDevice(%d) failed", h->cfg.device);
            ssw_model_free(m);
            return NULL;
        }
        m->device = h->cfg.device;
    } else
        (void)hipGetDevice(&m->device);
    if (upload_model(m) < 0) {
        ssw_model_free(m);
        return NULL;
    }
    {
        const char *e = getenv("SSW_PTM_EXACT");
        m->force_exact = (e != NULL && e[0] == '1');
    }
    return m;
}

extern "C" void
ssw_model_free(ssw_model_t *m)
{
    if (m == NULL)
        return;
    if (m->device == SSW_DEVICE_NONE) {
        ssw_host_model_free(m->h);
        delete m;
        return;
    }
    (void)hipFree(m->d_rec);
    (void)hipFree(m->d_recq);
    (void)hipFree(m->d_recmax);
    (void)hipFree(m->d_exlist);
    (void)hipFree(m->d_align_ws);
    (void)hipFree(m->d_mixw);
    (void)hipFree(m->d_ms_pdf);
    (void)hipFree(m->d_sen2cb);
    (void)hipFree(m->d_quad_cb);
    (void)hipFree(m->d_slot_sen);
    (void)hipFree(m->d_logadd8);
    (void)hipFree(m->d_tp);
    (void)hipFree(m->d_topn_cw);
    (void)hipFree(m->d_topn_sc);
    (void)hipFree(m->d_utt_off);
    (void)hipFree(m->d_flags);
    (void)hipFree(m->d_work);
    (void)hipFree(m->d_utt_start);
    (void)hipFree(m->d_nfixed);
    delete m->utt_cache;
    (void)hipFree(m->d_feats);
    (void)hipFree(m->d_out);
    if (m->timing)
        for (int i = 0; i < 3; ++i)
            (void)hipEventDestroy(m->ev[i]);
    ssw_host_model_free(m->h);
    delete m;
}

extern "C" const ssw_host_model_t *
ssw_model_host(const ssw_model_t *m)
{
    return m->h;
}

extern "C" int
ssw_model_info(const ssw_model_t *m, ssw_model_info_t *o)
{
    const ssw_host_model_t *h = m->h;
    memset(o, 0, sizeof(*o));
    o->n_cb = h->n_cb;
    o->n_feat = h->n_feat;
    o->n_density = h->n_density;
    o->veclen_total = h->veclen_total;
    o->n_sen = h->n_sen;
    o->n_ci_sen = h->n_ci_sen;
    o->n_ciphone = h->n_ciphone;
    o->n_phone = h->n_phone;
    o->n_emit_state = h->n_emit_state;
    o->n_tmat = h->tp ? h->tp_n_tmat : h->n_tmat;
    o->n_sseq = h->n_sseq;
    o->sil = h->sil;
    o->n_floored = h->n_floored;
    o->topn = h->cfg.topn;
    o->has_ptm = h->ptm_mixw != NULL;
    o->has_ms = h->ms_pdf != NULL;
    o->device = m->device;
    for (int f = 0; f < h->n_feat && f < 8; ++f)
        o->veclen[f] = h->veclen[f];
    return 0;
}

extern "C" const void *
ssw_model_table(const ssw_model_t *m, int which, size_t *nbytes)
{
    const ssw_host_model_t *h = m->h;
    size_t gau = (size_t)h->n_cb * h->n_density * h->veclen_total * sizeof(float);
    size_t n = 0;
    const void *p = NULL;
    switch (which) {
    case SSW_TAB_MEAN: p = h->mean; n = gau; break;
    case SSW_TAB_VAR: p = h->var; n = gau; break;
    case SSW_TAB_DET: p = h->det; n = (size_t)h->n_cb * h->n_feat * h->n_density * 4; break;
    case SSW_TAB_PTM_MIXW: p = h->ptm_mixw; n = (size_t)h->n_feat * h->n_density * h->n_sen; break;
    case SSW_TAB_MS_PDF: p = h->ms_pdf; n = (size_t)h->n_feat * h->n_density * h->n_sen; break;
    case SSW_TAB_TP: p = h->tp; n = (size_t)h->tp_n_tmat * h->tp_n_state * (h->tp_n_state + 1); break;
    case SSW_TAB_SSEQ: p = h->sseq; n = (size_t)h->n_sseq * h->n_emit_state * 2; break;
    case SSW_TAB_SEN2CB: p = h->sen2cb; n = (size_t)h->n_sen * 2; break;
    case SSW_TAB_LOGADD8: p = h->logadd8; n = 256; break;
    case SSW_TAB_PHONE_SSID: p = h->phone_ssid; n = (size_t)h->n_phone * 4; break;
    case SSW_TAB_PHONE_TMAT: p = h->phone_tmat; n = (size_t)h->n_phone * 4; break;
    default: break;
    }
    if (p == NULL)
        n = 0;
    if (nbytes)
        *nbytes = n;
    return p;
}

/* ---------------------------------------------------------------------------------- */
static int
ensure_score_ws(ssw_model_s *m, int n_frames, int n_utts)
{
    if ((size_t)n_frames > m->ws_frames) {
        (void)hipFree(m->d_topn_cw);
        (void)hipFree(m->d_topn_sc);
        (void)hipFree(m->d_flags);
        (void)hipFree(m->d_work);
        (void)hipFree(m->d_utt_start);
        m->d_topn_cw = NULL;
        m->d_topn_sc = NULL;
        m->d_flags = NULL;
        m->d_work = NULL;
        m->d_utt_start = NULL;
        m->ws_frames = 0;
        if (m->utt_cache)
            m->utt_cache->clear(); /* d_utt_start has to be rebuilt */
        if ((uint64_t)n_frames * (uint64_t)m->n_cbf > 0xffffffffull) {
            ssw_set_error("batch of %d frames is too large (frames x codebooks x streams must fit 32 bits)",
                          n_frames);
            return -1;
        }
        if (dev_alloc(&m->d_topn_cw, (size_t)n_frames * m->n_cbf) < 0
            || dev_alloc(&m->d_topn_sc, (size_t)n_frames * m->n_cbf) < 0
            || dev_alloc(&m->d_flags, ((size_t)n_frames * m->n_cbf + 31) / 32 + 64) < 0
            || dev_alloc(&m->d_work, (size_t)n_frames * m->n_cbf) < 0
            || dev_alloc(&m->d_utt_start, ((size_t)n_frames + 31) / 32 + 2) < 0)
            return -1;
        HIP_OK(hipMemset(m->d_flags, 0,
                         sizeof(uint32_t) * (((size_t)n_frames * m->n_cbf + 31) / 32 + 64)));
        m->ws_frames = (size_t)n_frames;
    }
    if (m->d_nfixed == NULL) {
        if (dev_alloc(&m->d_nfixed, 4) < 0)
            return -1;
        HIP_OK(hipMemset(m->d_nfixed, 0, 4 * sizeof(unsigned long long)));
    }
    if ((size_t)n_utts + 1 > m->ws_utts) {
        (void)hipFree(m->d_utt_off);
        m->d_utt_off = NULL;
        m->ws_utts = 0;
        if (dev_alloc(&m->d_utt_off, (size_t)n_utts + 1) < 0)
            return -1;
        m->ws_utts = (size_t)n_utts + 1;
    }
    return 0;
}

static int
check_scorer_shape(const ssw_model_s *m, int scorer)
{
    const ssw_host_model_t *h = m->h;
    if (m->device == SSW_DEVICE_NONE) {
        ssw_set_error("model was loaded with device = SSW_DEVICE_NONE (tables only): no GPU, "
                      "no scoring -- there is no CPU fallback");
        return -1;
    }
    if (scorer == SSW_SCORER_PTM && h->ptm_mixw == NULL) {
        ssw_set_error("model has no PTM mixture weights (sendump / mixw)");
        return -1;
    }
    if (scorer == SSW_SCORER_MS && h->ms_pdf == NULL) {
        ssw_set_error("the ms scorer needs a mixture_weights file (src/ms_mgau.c:208-212)");
        return -1;
    }
    if (scorer == SSW_SCORER_MS && (h->logadd8_size != 256 || h->cfg.aw == 0)) {
        ssw_set_error("ms scorer: unsupported log base (add table of %d entries) or aw = 0",
                      h->logadd8_size);
        return -1;
    }
    if (scorer != SSW_SCORER_PTM && scorer != SSW_SCORER_MS) {
        ssw_set_error("unknown scorer %d", scorer);
        return -1;
    }
    if (h->n_density != 128 || h->cfg.topn != 4 || h->n_cb > 255) {
        ssw_set_error("PTM kernels are built for 128 densities, top-4, <= 255 codebooks "
                      "(model: %d densities, topn %d, %d codebooks)",
                      h->n_density, h->cfg.topn, h->n_cb);
        return -1;
    }
    for (int f = 0; f < h->n_feat; ++f)
        if (h->veclen[f] != 13) {
            ssw_set_error("PTM kernels are built for 13-dimensional streams");
            return -1;
        }
    return 0;
}

static void
fill_chain_params(const ssw_model_s *m, ChainParams &P, const float *d_feats)
{
    const ssw_host_model_t *h = m->h;
    memset(&P, 0, sizeof(P));
    P.rec = m->d_rec;
    P.feats = d_feats;
    P.utt_off = m->d_utt_off;
    P.topn_cw = m->d_topn_cw;
    P.topn_sc = m->d_topn_sc;
    P.n_cbf = m->n_cbf;
    P.n_feat = h->n_feat;
    P.featdim = h->veclen_total;
    P.ds = h->cfg.ds < 1 ? 1 : h->cfg.ds;
    for (int f = 0; f < h->n_feat; ++f)
        P.featoff[f] = h->featoff[f];
}

static int
launch_senone(ssw_model_s *m, int scorer, int n_frames, const uint32_t *cw, const int4 *sc,
              int16_t *d_out, uint32_t *flags, hipStream_t st)
{
    const ssw_host_model_t *h = m->h;
    SenoneParams S;
    S.topn_cw = cw;
    S.topn_sc = sc;
    S.mixw = scorer == SSW_SCORER_MS ? m->d_ms_pdf : m->d_mixw;
    S.aw = h->cfg.aw;
    S.zero = h->zero8;
    S.raw = m->ms_raw;
    S.quad_cb = m->d_quad_cb;
    S.slot_sen = m->d_slot_sen;
    S.logadd8 = m->d_logadd8;
    S.flags = flags;
    S.nfixed = m->d_nfixed;
    S.out = d_out;
    S.n_frames = n_frames;
    S.n_cb = h->n_cb;
    S.n_feat = h->n_feat;
    S.n_density = h->n_density;
    S.n_sen = h->n_sen;
    S.slot_stride = m->slot_stride;
    S.n_quads = m->n_quads;
    /* frames per workgroup: 4 when the batch still leaves >= 2 workgroups per CU, and the
     * prologue can give every (frame, codebook, stream) its own thread (measured on MI355X,
     * en-us, 4096 frames: 1 -> 65 us, 2 -> 58 us, 4 -> 53.5 us, 8 -> 68 us) */
    int fpb = n_frames >= 4 * 512 ? 4 : 1;
    size_t lds = SSW_LOGADD_LDS + (4 * SSW_MAX_FEAT + 24 * (size_t)m->n_cbf + 16 * sizeof(int)) * fpb + 16;
    /* quads per thread: as few as a 1024-thread workgroup allows (measured on MI355X, en-us:
     * R = 2 -> 58 us, 3 -> 59 us, 4 -> 78 us per 4096 frames; more quads per thread only adds
     * register pressure) */
    int R = (m->n_quads + SEN_MAX_THREADS - 1) / SEN_MAX_THREADS;
    {
        const char *e = getenv("SSW_SEN_R");
        if (e != NULL && e[0] >= '1' && e[0] <= '4')
            R = e[0] - '0';
    }
    int threads = ((m->n_quads + R - 1) / R + 63) & ~63;
    if (threads < 256)
        threads = 256; /* the prologue copies the 256-entry table with the first 256 threads */
    if (threads > SEN_MAX_THREADS || (scorer == SSW_SCORER_MS && threads < m->n_cbf)) {
        ssw_set_error("unsupported senone/codebook shape (%d quads, %d codebook x stream)",
                      m->n_quads, m->n_cbf);
        return -1;
    }
    if (scorer == SSW_SCORER_MS) {
        lds = 256 + 20 * (size_t)m->n_cbf + 16 * sizeof(int);
        switch (R) {
        case 1: hipLaunchKernelGGL((ms_senone_kernel<1>), dim3(n_frames), dim3(threads), lds, st, S); break;
        case 2: hipLaunchKernelGGL((ms_senone_kernel<2>), dim3(n_frames), dim3(threads), lds, st, S); break;
        case 3: hipLaunchKernelGGL((ms_senone_kernel<3>), dim3(n_frames), dim3(threads), lds, st, S); break;
        case 4: hipLaunchKernelGGL((ms_senone_kernel<4>), dim3(n_frames), dim3(threads), lds, st, S); break;
        default:
            ssw_set_error("too many senones (%d quads)", m->n_quads);
            return -1;
        }
        HIP_OK(hipGetLastError());
        return 0;
    }
    const dim3 grid((n_frames + fpb - 1) / fpb);
#define SSW_SEN_LAUNCH(RR)                                                                   \
    if (fpb == 4 && h->n_feat == 3)                                                          \
        hipLaunchKernelGGL((ptm_senone_kernel<4, RR, 4, 3>), grid, dim3(threads), lds, st, S); \
    else if (fpb == 4)                                                                       \
        hipLaunchKernelGGL((ptm_senone_kernel<4, RR, 4, 0>), grid, dim3(threads), lds, st, S); \
    else if (h->n_feat == 3)                                                                 \
        hipLaunchKernelGGL((ptm_senone_kernel<4, RR, 1, 3>), grid, dim3(threads), lds, st, S); \
    else                                                                                     \
        hipLaunchKernelGGL((ptm_senone_kernel<4, RR, 1, 0>), grid, dim3(threads), lds, st, S)
    switch (R) {
    case 1: SSW_SEN_LAUNCH(1); break;
    case 2: SSW_SEN_LAUNCH(2); break;
    case 3: SSW_SEN_LAUNCH(3); break;
    case 4: SSW_SEN_LAUNCH(4); break;
    default:
        ssw_set_error("too many senones (%d quads)", m->n_quads);
        return -1;
    }
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int
ssw_score_batch(ssw_model_t *m, int scorer, const float *d_feats, int32_t n_frames,
                const int32_t *utt_off, int32_t n_utts, int16_t *d_out, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n_frames == 0 || n_utts == 0)
        return 0;
    if (n_frames < 0 || n_utts < 0 || utt_off == NULL || utt_off[0] != 0
        || utt_off[n_utts] != n_frames) {
        ssw_set_error("bad utterance offsets");
        return -1;
    }
    for (int u = 0; u < n_utts; ++u)
        if (utt_off[u + 1] < utt_off[u]) {
            ssw_set_error("utterance offsets must be non-decreasing");
            return -1;
        }
    if (check_scorer_shape(m, scorer) < 0)
        return -1;
    const bool ms = scorer == SSW_SCORER_MS;
    HIP_OK(hipSetDevice(m->device));
    if (ensure_score_ws(m, n_frames, n_utts) < 0)
        return -1;
    /* the offsets rarely change between calls of one job: upload only when they do */
    if (m->utt_cache == NULL)
        m->utt_cache = new std::vector<int32_t>();
    if (m->utt_cache->size() != (size_t)n_utts + 1
        || memcmp(m->utt_cache->data(), utt_off, sizeof(int32_t) * ((size_t)n_utts + 1)) != 0) {
        m->utt_cache->assign(utt_off, utt_off + n_utts + 1);
        HIP_OK(hipStreamSynchronize(st)); /* earlier launches may still read the old offsets */
        HIP_OK(hipMemcpy(m->d_utt_off, utt_off, sizeof(int32_t) * ((size_t)n_utts + 1),
                         hipMemcpyHostToDevice));
        std::vector<uint32_t> starts(((size_t)n_frames + 31) / 32 + 1, 0u);
        for (int u = 0; u < n_utts; ++u)
            if (utt_off[u] < n_frames)
                starts[(size_t)utt_off[u] >> 5] |= 1u << (utt_off[u] & 31);
        HIP_OK(hipMemcpy(m->d_utt_start, starts.data(), starts.size() * sizeof(uint32_t),
                         hipMemcpyHostToDevice));
    }
    const ssw_host_model_t *h = m->h;
    ChainParams P;
    fill_chain_params(m, P, d_feats);
    P.n_utts = n_utts;
    P.n_frames = n_frames;
    P.flags = m->d_flags;
    P.work = m->d_work;
    P.work_count = (const unsigned *)(m->d_nfixed + 2);
    P.utt_start = m->d_utt_start;
    const int64_t pairs = (int64_t)n_frames * m->n_cbf;
    if (m->timing)
        HIP_OK(hipEventRecord(m->ev[0], st));
    if (!ms && (h->cfg.ds != 1 || m->force_exact)) {
        /* frame down-sampling makes every frame depend on its predecessor: exact chains */
        int n_chain = n_utts * m->n_cbf;
        hipLaunchKernelGGL((ptm_topn_chain_kernel<13, 2, 4>), dim3((n_chain + 3) / 4),
                           dim3(256), 0, st, P);
        HIP_OK(hipGetLastError());
        m->stats[0] = pairs;
        m->stats_pending = 0;
    } else {
        /* flags are all-zero and the work list is empty here: zeroed at allocation, and the senone kernel clears each
         * frame's words once the fix-up pass has consumed them */
        FramesParams F;
        memset(&F, 0, sizeof(F));
        F.topn_cw = m->d_topn_cw;
        F.topn_sc = m->d_topn_sc;
        F.flags = m->d_flags;
        F.work = m->d_work;
        F.work_count = (unsigned *)(m->d_nfixed + 2);
        F.n_frames = n_frames;
        F.n_cbf = m->n_cbf;
        F.n_feat = h->n_feat;
        F.featdim = h->veclen_total;
        for (int f = 0; f < h->n_feat; ++f)
            F.featoff[f] = h->featoff[f];
        /* two frames per lane once that still fills the chip with >= 4 waves per SIMD */
        const bool two = (int64_t)((n_frames + 127) / 128) * m->n_cbf >= 2048;
        const int fpl = two ? 2 : 1;
        const int tiles = (n_frames + 64 * fpl - 1) / (64 * fpl);
        F.tile_groups = (tiles + 3) / 4;
        dim3 grid((unsigned)((((int64_t)F.tile_groups * m->n_cbf + 7) / 8) * 8));
        if (two && ms)
            hipLaunchKernelGGL((ptm_topn_frames_kernel<13, 2, true>), grid, dim3(256), 0, st,
                               m->d_rec, m->d_recq, m->d_recmax, m->d_exlist, d_feats, F);
        else if (two)
            hipLaunchKernelGGL((ptm_topn_frames_kernel<13, 2, false>), grid, dim3(256), 0, st,
                               m->d_rec, m->d_recq, m->d_recmax, m->d_exlist, d_feats, F);
        else if (ms)
            hipLaunchKernelGGL((ptm_topn_frames_kernel<13, 1, true>), grid, dim3(256), 0, st,
                               m->d_rec, m->d_recq, m->d_recmax, m->d_exlist, d_feats, F);
        else
            hipLaunchKernelGGL((ptm_topn_frames_kernel<13, 1, false>), grid, dim3(256), 0, st,
                               m->d_rec, m->d_recq, m->d_recmax, m->d_exlist, d_feats, F);
        HIP_OK(hipGetLastError());
        /* one work item per wave; the list is normally far shorter than the grid */
        int64_t fb = pairs / 64 + 1;
        const int fix_blocks = (int)(fb > 2048 ? 2048 : fb);
        if (ms)
            hipLaunchKernelGGL((ms_topn_fixup_kernel<13, 2, 4>), dim3(fix_blocks), dim3(64), 0,
                               st, P, m->d_nfixed);
        else
            hipLaunchKernelGGL((ptm_topn_fixup_kernel<13, 2, 4>), dim3(fix_blocks), dim3(64), 0,
                               st, P, m->d_nfixed);
        HIP_OK(hipGetLastError());
        m->stats_pending = 1;
    }
    if (m->timing)
        HIP_OK(hipEventRecord(m->ev[1], st));
    if (launch_senone(m, scorer, n_frames, m->d_topn_cw, m->d_topn_sc, d_out,
                      m->stats_pending ? m->d_flags : NULL, st) < 0)
        return -1;
    if (m->timing)
        HIP_OK(hipEventRecord(m->ev[2], st));
#if defined(SSW_TIMELINE) || defined(SSW_TIMELINE_SEN)
    if (getenv("SSW_TIMELINE_OUT")) {
        std::vector<unsigned long long> tl(16384 * 6);
        HIP_OK(hipStreamSynchronize(st));
        HIP_OK(hipMemcpyFromSymbol(tl.data(), HIP_SYMBOL(g_timeline), tl.size() * 8));
        FILE *fp = fopen(getenv("SSW_TIMELINE_OUT"), "wb");
        if (fp) {
            fwrite(tl.data(), 8, tl.size(), fp);
            fclose(fp);
        }
    }
#endif
    m->last_n_frames = n_frames;
    m->stats[1] = pairs;
    return 0;
}

extern "C" int
ssw_score_batch_host(ssw_model_t *m, int scorer, const float *feats, int32_t n_frames,
                     const int32_t *utt_off, int32_t n_utts, int16_t *out)
{
    const ssw_host_model_t *h = m->h;
    if (n_frames <= 0)
        return 0;
    if (check_scorer_shape(m, scorer) < 0)
        return -1;
    HIP_OK(hipSetDevice(m->device));
    if ((size_t)n_frames > m->st_frames) {
        (void)hipFree(m->d_feats);
        (void)hipFree(m->d_out);
        m->d_feats = NULL;
        m->d_out = NULL;
        m->st_frames = 0;
        if (dev_alloc(&m->d_feats, (size_t)n_frames * h->veclen_total) < 0
            || dev_alloc(&m->d_out, (size_t)n_frames * h->n_sen) < 0)
            return -1;
        m->st_frames = (size_t)n_frames;
    }
    HIP_OK(hipMemcpy(m->d_feats, feats, sizeof(float) * (size_t)n_frames * h->veclen_total,
                     hipMemcpyHostToDevice));
    if (ssw_score_batch(m, scorer, m->d_feats, n_frames, utt_off, n_utts, m->d_out, NULL) < 0)
        return -1;
    HIP_OK(hipMemcpy(out, m->d_out, sizeof(int16_t) * (size_t)n_frames * h->n_sen,
                     hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int
ssw_score_batch_topn(ssw_model_t *m, int32_t n_frames, uint8_t *cw, int32_t *score)
{
    if (n_frames > m->last_n_frames) {
        ssw_set_error("only %d frames were scored", m->last_n_frames);
        return -1;
    }
    HIP_OK(hipSetDevice(m->device));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(cw, m->d_topn_cw, (size_t)n_frames * m->n_cbf * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(score, m->d_topn_sc, (size_t)n_frames * m->n_cbf * 16,
                     hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int
ssw_set_kernel_timing(ssw_model_t *m, int enable)
{
    HIP_OK(hipSetDevice(m->device));
    if (enable && !m->timing) {
        for (int i = 0; i < 3; ++i)
            HIP_OK(hipEventCreate(&m->ev[i]));
        m->timing = 1;
    } else if (!enable && m->timing) {
        for (int i = 0; i < 3; ++i)
            (void)hipEventDestroy(m->ev[i]);
        m->timing = 0;
    }
    return 0;
}

extern "C" int
ssw_get_kernel_timing(ssw_model_t *m, float *ms, int n)
{
    if (!m->timing) {
        ssw_set_error("kernel timing is off");
        return -1;
    }
    HIP_OK(hipEventSynchronize(m->ev[2]));
    int k = 0;
    for (; k < 2 && k < n; ++k)
        HIP_OK(hipEventElapsedTime(&ms[k], m->ev[k], m->ev[k + 1]));
    return k;
}

extern "C" int
ssw_score_batch_stats(ssw_model_t *m, int64_t stats[2])
{
    if (m->stats_pending) {
        unsigned long long n = 0;
        HIP_OK(hipSetDevice(m->device));
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemcpy(&n, m->d_nfixed + 1, sizeof(n), hipMemcpyDeviceToHost));
        m->stats[0] = (int64_t)n; /* moved there by the last batch's senone kernel */
        m->stats_pending = 0;
    }
    stats[0] = m->stats[0];
    stats[1] = m->stats[1];
    return 0;
}

/* ---------------------------------------------------------------------------------- */
/* alignment                                                                            */
/* ---------------------------------------------------------------------------------- */
extern "C" int
ssw_align_batch(ssw_model_t *m, const int16_t *d_senscr, int32_t n_utts,
                const int32_t *frame_off, const int32_t *phone_off, const uint16_t *senid,
                const int16_t *tmatid, const int32_t *sf, const int32_t *ef,
                ssw_align_entry_t *state_io, int32_t *status, void *stream)
{
    const ssw_host_model_t *h = m->h;
    hipStream_t st = (hipStream_t)stream;
    if (n_utts <= 0)
        return 0;
    if (m->device == SSW_DEVICE_NONE) {
        ssw_set_error("model was loaded with device = SSW_DEVICE_NONE (tables only): no GPU, "
                      "no alignment -- there is no CPU fallback");
        return -1;
    }
    if (h->tp == NULL || h->tp_n_state != 3) {
        ssw_set_error("alignment kernel needs 3-state transition matrices");
        return -1;
    }
    HIP_OK(hipSetDevice(m->device));
    const int total_phones = phone_off[n_utts];
    std::vector<AlignUtt> utts((size_t)n_utts);
    long long tok_total = 0;
    int max_phones = 0;
    for (int u = 0; u < n_utts; ++u) {
        AlignUtt &U = utts[u];
        U.frame_off = frame_off[u];
        U.n_frames = frame_off[u + 1] - frame_off[u];
        U.phone_off = phone_off[u];
        U.n_phones = phone_off[u + 1] - phone_off[u];
        U.tok_off = tok_total;
        if (U.n_phones < 1 || U.n_frames < 0) {
            ssw_set_error("utterance %d: %d phones, %d frames", u, U.n_phones, U.n_frames);
            return -1;
        }
        tok_total += (long long)U.n_frames * U.n_phones * 3;
        max_phones = U.n_phones > max_phones ? U.n_phones : max_phones;
    }
    for (int p = 0; p < total_phones; ++p)
        if (tmatid[p] < 0 || tmatid[p] >= h->tp_n_tmat) {
            ssw_set_error("phone %d: transition matrix %d out of range", p, tmatid[p]);
            return -1;
        }
    size_t lds = (size_t)A_COUNT * (max_phones + 1) * sizeof(int);
    if (max_phones > 64 * SSW_ALIGN_MAX_WAVES && lds > 160 * 1024) {
        ssw_set_error("utterance with %d phones exceeds the LDS-resident limit", max_phones);
        return -1;
    }
    /* one grow-only device arena per model for the call's inputs, token stacks and results */
    size_t off = 0;
    auto carve = [&off](size_t bytes) {
        size_t at = off;
        off += (bytes + 255) & ~(size_t)255;
        return at;
    };
    const size_t o_utts = carve(sizeof(AlignUtt) * n_utts);
    const size_t o_senid = carve(sizeof(uint16_t) * 3 * (size_t)total_phones + 16);
    const size_t o_tmatid = carve(sizeof(int16_t) * (size_t)total_phones + 16);
    const size_t o_sf = carve(sizeof(int32_t) * (size_t)total_phones);
    const size_t o_ef = carve(sizeof(int32_t) * (size_t)total_phones);
    const size_t o_status = carve(sizeof(int32_t) * n_utts);
    const size_t o_state = carve(sizeof(ssw_align_entry_t) * 3 * (size_t)total_phones);
    const size_t o_tok = carve(sizeof(int2) * (size_t)(tok_total > 0 ? tok_total : 1));
    int rv = -1;
#define TRY(expr)                                                                            \
    if ((expr) != hipSuccess) {                                                              \
        ssw_set_error("%s failed: %s", #expr, hipGetErrorString(hipGetLastError()));         \
        goto out;                                                                            \
    }
    if (off > m->align_ws_bytes) {
        (void)hipFree(m->d_align_ws);
        m->d_align_ws = NULL;
        m->align_ws_bytes = 0;
        TRY(hipMalloc((void **)&m->d_align_ws, off));
        m->align_ws_bytes = off;
    }
    {
    unsigned char *ws = m->d_align_ws;
    AlignUtt *d_utts = reinterpret_cast<AlignUtt *>(ws + o_utts);
    uint16_t *d_senid = reinterpret_cast<uint16_t *>(ws + o_senid);
    int16_t *d_tmatid = reinterpret_cast<int16_t *>(ws + o_tmatid);
    int32_t *d_sf = reinterpret_cast<int32_t *>(ws + o_sf);
    int32_t *d_ef = reinterpret_cast<int32_t *>(ws + o_ef);
    int32_t *d_status = reinterpret_cast<int32_t *>(ws + o_status);
    ssw_align_entry_t *d_state = reinterpret_cast<ssw_align_entry_t *>(ws + o_state);
    int2 *d_tok = reinterpret_cast<int2 *>(ws + o_tok);
    TRY(hipMemcpyAsync(d_utts, utts.data(), sizeof(AlignUtt) * n_utts, hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(d_senid, senid, sizeof(uint16_t) * 3 * (size_t)total_phones,
                       hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(d_tmatid, tmatid, sizeof(int16_t) * (size_t)total_phones,
                       hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(d_sf, sf, sizeof(int32_t) * (size_t)total_phones, hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(d_ef, ef, sizeof(int32_t) * (size_t)total_phones, hipMemcpyHostToDevice, st));
    TRY(hipMemcpyAsync(d_state, state_io, sizeof(ssw_align_entry_t) * 3 * (size_t)total_phones,
                       hipMemcpyHostToDevice, st));
    {
        AlignParams A;
        A.senscr = d_senscr;
        A.utts = d_utts;
        A.senid = d_senid;
        A.tmatid = d_tmatid;
        A.sf = d_sf;
        A.ef = d_ef;
        A.tp = m->d_tp;
        A.tokens = d_tok;
        A.state_io = d_state;
        A.status = d_status;
        A.n_sen = h->n_sen;
        A.n_utts = n_utts;
        A.max_phones = max_phones;
        const int words = (max_phones + 63) / 64;
        /* HMMs in registers: one wave per 64-phone word when the batch leaves SIMDs idle (the
         * usual case: a frame step is a short dependent chain), one wave per utterance for very
         * large batches; utterances beyond 1024 phones go through LDS */
        const char *mode = getenv("SSW_ALIGN_KERNEL"); /* "lds", "reg", "mw": tests and tuning */
        if (words <= SSW_ALIGN_MAX_WAVES && !(mode && (!strcmp(mode, "lds") || !strcmp(mode, "reg")))
            && (words > 4 || (int64_t)n_utts * words <= 8192 || (mode && !strcmp(mode, "mw")))) {
            hipLaunchKernelGGL(viterbi_align_mw_kernel, dim3(n_utts), dim3(64 * words), 0, st, A);
            TRY(hipGetLastError());
        } else {
            const bool in_regs = words <= 4 && !(mode && !strcmp(mode, "lds"));
            void (*kern)(AlignParams) = words <= 1 ? viterbi_align_reg_kernel<1>
                : words <= 2                       ? viterbi_align_reg_kernel<2>
                : words <= 4                       ? viterbi_align_reg_kernel<4>
                : words <= 8                       ? viterbi_align_kernel<8>
                                                   : viterbi_align_kernel<0>;
            if (!in_regs && words <= 4)
                kern = viterbi_align_kernel<4>;
            if (!in_regs && lds > 64 * 1024)
                TRY(hipFuncSetAttribute((const void *)kern,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(n_utts), dim3(64), in_regs ? 0 : lds, st, A);
            TRY(hipGetLastError());
        }
    }
    TRY(hipMemcpyAsync(state_io, d_state, sizeof(ssw_align_entry_t) * 3 * (size_t)total_phones,
                       hipMemcpyDeviceToHost, st));
    TRY(hipMemcpyAsync(status, d_status, sizeof(int32_t) * n_utts, hipMemcpyDeviceToHost, st));
    TRY(hipStreamSynchronize(st));
    rv = 0;
    }
out:
#undef TRY
    return rv;
}

extern "C" int
ssw_alignment_propagate(const ssw_align_entry_t *child, const int32_t *parent, int32_t n_child,
                        ssw_align_entry_t *parent_out, int32_t n_parent)
{
    int last = -1;
    for (int i = 0; i < n_child; ++i) {
        int p = parent[i];
        if (p < 0 || p >= n_parent) {
            ssw_set_error("child %d has parent %d of %d", i, p, n_parent);
            return -1;
        }
        if (p != last) { /* src/ps_alignment.c:326-330 */
            parent_out[p].start = child[i].start;
            parent_out[p].duration = 0;
            parent_out[p].score = 0;
        }
        parent_out[p].duration += child[i].duration;
        parent_out[p].score += child[i].score;
        last = p;
    }
    return 0;
}

/* ---------------------------------------------------------------------------------- */
/* scorer object: mgau_t / mgaufuncs_t drop-in                                          */
/* ---------------------------------------------------------------------------------- */
struct ssw_mgau_impl {
    ssw_mgau_s base; /* must be first: {vt, frame_idx} as in acmod.h:108-111 */
    ssw_model_s *m;
    int scorer;
    /* one-frame path: the reference's 2-deep history ring (src/ptm_mgau.c:425-448) */
    uint32_t *d_hist_cw[2];
    int4 *d_hist_sc[2];
    int *d_utt1;
    float *d_feat1;
    int16_t *d_out1;
    uint8_t *d_cb_active[2]; /* mgau_active of each history slot */
    uint8_t *d_sen_active;
    int cb_all[2], sen_all;  /* the device copy currently says "everything active" */
    /* pinned, device-mapped staging of the one-frame call: the kernels read the feature row from
     * host memory and write the scores to it, so a frame costs two launches and one stream
     * synchronise instead of four blocking copies */
    float *h_feat1, *dh_feat1;
    int16_t *h_out1, *dh_out1;
    /* whole-utterance cache filled by ssw_mgau_prescore */
    std::vector<int16_t> cache;
    int cache_frames;
};

static int mgau_frame_eval(ssw_mgau_t *mg, int16_t *senscr, uint8_t *senone_active,
                           int32_t n_senone_active, float **feat, int32_t frame,
                           int32_t compallsen);
static int mgau_transform(ssw_mgau_t *mg, void *mllr);
static void mgau_free(ssw_mgau_t *mg);

static ssw_mgaufuncs_t g_ptm_funcs = { "ptm", mgau_frame_eval, mgau_transform, mgau_free };

static int
mgau_alloc_pinned(ssw_mgau_impl *g)
{
    const ssw_host_model_t *h = g->m->h;
    g->h_feat1 = NULL;
    g->h_out1 = NULL;
    g->cb_all[0] = g->cb_all[1] = g->sen_all = 0;
    HIP_OK(hipHostMalloc((void **)&g->h_feat1, sizeof(float) * SSW_MAX_FEAT * SSW_MAX_VECLEN,
                         hipHostMallocMapped));
    HIP_OK(hipHostMalloc((void **)&g->h_out1, sizeof(int16_t) * (size_t)h->n_sen,
                         hipHostMallocMapped));
    HIP_OK(hipHostGetDevicePointer((void **)&g->dh_feat1, g->h_feat1, 0));
    HIP_OK(hipHostGetDevicePointer((void **)&g->dh_out1, g->h_out1, 0));
    return 0;
}

static int
mgau_reset_device_hist(ssw_mgau_impl *g)
{
    std::vector<uint32_t> init((size_t)g->m->n_cbf, 0x03020100u); /* cw = m */
    std::vector<uint8_t> ones((size_t)g->m->h->n_cb, 1);         /* all codebooks active */
    for (int i = 0; i < 2; ++i) {
        HIP_OK(hipMemcpy(g->d_hist_cw[i], init.data(), init.size() * 4, hipMemcpyHostToDevice));
        if (g->d_cb_active[i] != NULL) {
            HIP_OK(hipMemcpy(g->d_cb_active[i], ones.data(), ones.size(), hipMemcpyHostToDevice));
            g->cb_all[i] = 1;
        }
    }
    return 0;
}

extern "C" ssw_mgau_t *
ssw_ptm_mgau_init(ssw_model_t *m)
{
    if (check_scorer_shape(m, SSW_SCORER_PTM) < 0)
        return NULL;
    if (hipSetDevice(m->device) != hipSuccess) {
        ssw_set_error("hipSetDevice failed");
        return NULL;
    }
    ssw_mgau_impl *g = new ssw_mgau_impl();
    g->base.vt = &g_ptm_funcs;
    g->base.frame_idx = 0;
    g->m = m;
    g->scorer = SSW_SCORER_PTM;
    g->cache_frames = 0;
    g->d_utt1 = NULL;
    g->d_feat1 = NULL;
    g->d_out1 = NULL;
    for (int i = 0; i < 2; ++i) {
        g->d_hist_cw[i] = NULL;
        g->d_hist_sc[i] = NULL;
        g->d_cb_active[i] = NULL;
    }
    g->d_sen_active = NULL;
    bool ok = true;
    for (int i = 0; i < 2 && ok; ++i)
        ok = dev_alloc(&g->d_hist_cw[i], (size_t)m->n_cbf) == 0
            && dev_alloc(&g->d_hist_sc[i], (size_t)m->n_cbf) == 0;
    ok = ok && dev_alloc(&g->d_utt1, 2) == 0 && dev_alloc(&g->d_feat1, (size_t)m->h->veclen_total) == 0
        && dev_alloc(&g->d_out1, (size_t)m->h->n_sen) == 0
        && dev_alloc(&g->d_sen_active, (size_t)m->h->n_sen) == 0
        && dev_alloc(&g->d_cb_active[0], (size_t)m->h->n_cb) == 0
        && dev_alloc(&g->d_cb_active[1], (size_t)m->h->n_cb) == 0;
    if (ok) { /* every codebook starts active (src/ptm_mgau.c:716-718) */
        std::vector<uint8_t> ones((size_t)m->h->n_cb, 1);
        for (int i = 0; i < 2 && ok; ++i)
            ok = hipMemcpy(g->d_cb_active[i], ones.data(), ones.size(), hipMemcpyHostToDevice)
                == hipSuccess;
    }
    int one[2] = { 0, 1 };
    ok = ok && hipMemcpy(g->d_utt1, one, sizeof(one), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && mgau_alloc_pinned(g) == 0;
    ok = ok && mgau_reset_device_hist(g) == 0;
    if (!ok) {
        mgau_free(&g->base);
        return NULL;
    }
    return &g->base;
}

static ssw_mgaufuncs_t g_ms_funcs = { "ms", mgau_frame_eval, mgau_transform, mgau_free };

/* ms_mgau_init(acmod_t *) (src/ms_mgau.c:165): needs a model loaded with a mixture_weights file */
extern "C" ssw_mgau_t *
ssw_ms_mgau_init(ssw_model_t *m)
{
    if (check_scorer_shape(m, SSW_SCORER_MS) < 0)
        return NULL;
    if (hipSetDevice(m->device) != hipSuccess) {
        ssw_set_error("hipSetDevice failed");
        return NULL;
    }
    ssw_mgau_impl *g = new ssw_mgau_impl();
    g->base.vt = &g_ms_funcs;
    g->base.frame_idx = 0;
    g->m = m;
    g->scorer = SSW_SCORER_MS;
    g->cache_frames = 0;
    g->d_utt1 = NULL;
    g->d_feat1 = NULL;
    g->d_out1 = NULL;
    for (int i = 0; i < 2; ++i) {
        g->d_hist_cw[i] = NULL;
        g->d_hist_sc[i] = NULL;
        g->d_cb_active[i] = NULL;
    }
    g->d_sen_active = NULL;
    if (dev_alloc(&g->d_feat1, (size_t)m->h->veclen_total) < 0
        || dev_alloc(&g->d_out1, (size_t)m->h->n_sen) < 0) {
        mgau_free(&g->base);
        return NULL;
    }
    return &g->base;
}

extern "C" void
ssw_mgau_reset_hist(ssw_mgau_t *mg)
{
    ssw_mgau_impl *g = reinterpret_cast<ssw_mgau_impl *>(mg);
    (void)hipSetDevice(g->m->device);
    if (g->scorer == SSW_SCORER_PTM)
        (void)mgau_reset_device_hist(g);
    g->cache_frames = 0;
}

extern "C" int
ssw_mgau_prescore(ssw_mgau_t *mg, const float *feats, int32_t n_frames)
{
    ssw_mgau_impl *g = reinterpret_cast<ssw_mgau_impl *>(mg);
    g->cache_frames = 0;
    if (n_frames <= 0)
        return 0;
    g->cache.resize((size_t)n_frames * g->m->h->n_sen);
    int32_t off[2] = { 0, n_frames };
    if (ssw_score_batch_host(g->m, g->scorer, feats, n_frames, off, 1, g->cache.data()) < 0)
        return -1;
    g->cache_frames = n_frames;
    return 0;
}

/* uint8 delta list (acmod_flags2list, src/acmod.c:947-999) -> per-senone and per-codebook
 * activity, exactly as ptm_mgau_calc_cb_active / ms_cont_mgau_frame_eval walk it */
static int
decode_active(const ssw_host_model_t *h, const uint8_t *list, int32_t n, std::vector<uint8_t> &sen,
              std::vector<uint8_t> &cb)
{
    sen.assign((size_t)h->n_sen, 0);
    cb.assign((size_t)h->n_cb, 0);
    int last = 0;
    for (int32_t i = 0; i < n; ++i) {
        int s = list[i] + last;
        if (s >= h->n_sen) {
            ssw_set_error("active list runs past the last senone (%d >= %d)", s, h->n_sen);
            return -1;
        }
        sen[s] = 1;
        cb[h->sen2cb[s]] = 1;
        last = s;
    }
    return 0;
}

/* frame_eval slot of mgaufuncs_t (acmod.h:96-102); semantics of ptm_mgau_frame_eval
 * (src/ptm_mgau.c:408-454) and ms_cont_mgau_frame_eval (src/ms_mgau.c:278-368), both for
 * compallsen = yes and for an active-senone list. */
static int
mgau_frame_eval(ssw_mgau_t *mg, int16_t *senscr, uint8_t *senone_active,
                int32_t n_senone_active, float **feat, int32_t frame, int32_t compallsen)
{
    ssw_mgau_impl *g = reinterpret_cast<ssw_mgau_impl *>(mg);
    ssw_model_s *m = g->m;
    const ssw_host_model_t *h = m->h;
    if (frame < 0) {
        ssw_set_error("negative frame");
        return -1;
    }
    if (!compallsen && (senone_active == NULL || n_senone_active < 0)) {
        ssw_set_error("compallsen=no needs the active senone list");
        return -1;
    }
    if (compallsen && frame < g->cache_frames) {
        memcpy(senscr, g->cache.data() + (size_t)frame * h->n_sen, sizeof(int16_t) * h->n_sen);
        return 0;
    }
    HIP_OK(hipSetDevice(m->device));
    std::vector<uint8_t> sen_act, cb_act;
    if (!compallsen && decode_active(h, senone_active, n_senone_active, sen_act, cb_act) < 0)
        return -1;
    float row[SSW_MAX_FEAT * SSW_MAX_VECLEN];
    for (int f = 0; f < h->n_feat; ++f)
        memcpy(row + h->featoff[f], feat[f], sizeof(float) * h->veclen[f]);

    if (g->scorer == SSW_SCORER_MS) { /* history-free: one frame is a batch of one */
        int32_t off[2] = { 0, 1 };
        HIP_OK(hipMemcpy(g->d_feat1, row, sizeof(float) * h->veclen_total, hipMemcpyHostToDevice));
        m->ms_raw = compallsen ? 0 : 1;
        int rv = ssw_score_batch(m, SSW_SCORER_MS, g->d_feat1, 1, off, 1, g->d_out1, NULL);
        m->ms_raw = 0;
        if (rv < 0)
            return -1;
        if (compallsen) {
            HIP_OK(hipMemcpy(senscr, g->d_out1, sizeof(int16_t) * h->n_sen, hipMemcpyDeviceToHost));
            return 0;
        }
        /* only active senones are written; they are normalised by the best active one
         * (src/ms_mgau.c:342-364).  Densities of inactive codebooks are simply not used. */
        std::vector<int16_t> raw((size_t)h->n_sen);
        HIP_OK(hipMemcpy(raw.data(), g->d_out1, sizeof(int16_t) * h->n_sen, hipMemcpyDeviceToHost));
        int best = INT_MAX;
        for (int s = 0; s < h->n_sen; ++s)
            if (sen_act[s] && raw[s] < best)
                best = raw[s];
        for (int s = 0; s < h->n_sen; ++s)
            if (sen_act[s]) {
                int bs = raw[s] - best;
                bs = bs > 32767 ? 32767 : bs;
                bs = bs < -32768 ? -32768 : bs;
                senscr[s] = (int16_t)bs;
            }
        return 0;
    }

    const int slot = frame % 2;
    if (frame >= g->base.frame_idx) {
        memcpy(g->h_feat1, row, sizeof(float) * h->veclen_total);
        if (!(compallsen && g->cb_all[slot])) {
            if (compallsen)
                cb_act.assign((size_t)h->n_cb, 1);
            HIP_OK(hipMemcpy(g->d_cb_active[slot], cb_act.data(), (size_t)h->n_cb,
                             hipMemcpyHostToDevice));
            g->cb_all[slot] = compallsen ? 1 : 0;
        }
        ChainParams P;
        fill_chain_params(m, P, g->dh_feat1);
        P.utt_off = g->d_utt1;
        P.n_utts = 1;
        P.carry_pk = g->d_hist_cw[slot ^ 1]; /* lastf, src/ptm_mgau.c:435-441 */
        P.cb_active = g->d_cb_active[slot];
        P.topn_cw = g->d_hist_cw[slot];
        P.topn_sc = g->d_hist_sc[slot];
        P.frame_base = frame;
        hipLaunchKernelGGL((ptm_topn_chain_kernel<13, 2, 4>), dim3((m->n_cbf + 3) / 4),
                           dim3(256), 0, 0, P);
        HIP_OK(hipGetLastError());
    }
    if (!(compallsen && g->sen_all)) {
        if (compallsen)
            sen_act.assign((size_t)h->n_sen, 1);
        HIP_OK(hipMemcpy(g->d_sen_active, sen_act.data(), (size_t)h->n_sen,
                         hipMemcpyHostToDevice));
        g->sen_all = compallsen ? 1 : 0;
    }
    {
        SenoneFrameParams F;
        F.topn_cw = g->d_hist_cw[slot];
        F.topn_sc = g->d_hist_sc[slot];
        F.mixw = m->d_mixw;
        F.quad_cb = m->d_quad_cb;
        F.logadd8 = m->d_logadd8;
        F.cb_active = g->d_cb_active[slot];
        F.sen_active = g->d_sen_active;
        F.slot_sen = m->d_slot_sen;
        F.out = g->dh_out1;
        F.n_cb = h->n_cb;
        F.n_feat = h->n_feat;
        F.n_density = h->n_density;
        F.n_sen = h->n_sen;
        F.slot_stride = m->slot_stride;
        F.n_quads = m->n_quads;
        hipLaunchKernelGGL(ptm_senone_frame_kernel, dim3(1), dim3(1024),
                           8 * (size_t)m->n_cbf, 0, F);
        HIP_OK(hipGetLastError());
    }
    HIP_OK(hipStreamSynchronize(0));
    memcpy(senscr, g->h_out1, sizeof(int16_t) * h->n_sen);
    return 0;
}

static int
mgau_transform(ssw_mgau_t *mg, void *mllr)
{
    (void)mg;
    (void)mllr;
    ssw_set_error("MLLR transforms are outside the accelerated path");
    return -1;
}

static void
mgau_free(ssw_mgau_t *mg)
{
    ssw_mgau_impl *g = reinterpret_cast<ssw_mgau_impl *>(mg);
    if (g == NULL)
        return;
    for (int i = 0; i < 2; ++i) {
        (void)hipFree(g->d_hist_cw[i]);
        (void)hipFree(g->d_hist_sc[i]);
        (void)hipFree(g->d_cb_active[i]);
    }
    (void)hipFree(g->d_sen_active);
    (void)hipFree(g->d_utt1);
    (void)hipFree(g->d_feat1);
    (void)hipFree(g->d_out1);
    (void)hipHostFree(g->h_feat1);
    (void)hipHostFree(g->h_out1);
    delete g;
}

/* ---------------------------------------------------------------------------------- */
/* search-module shaped forced aligner                                                  */
/* ---------------------------------------------------------------------------------- */
struct ssw_state_align_search_s {
    ssw_model_s *m;
    ssw_mgau_impl *mgau;
    int n_phones, n_frames, started, finished;
    std::vector<uint16_t> senid;
    std::vector<int16_t> tmatid;
    std::vector<int32_t> sf, ef;
    std::vector<ssw_align_entry_t> state0, states, phones;
    std::vector<float> feats;
};

extern "C" ssw_state_align_search_t *
ssw_state_align_search_init(ssw_model_t *m, ssw_mgau_t *mgau, int32_t n_phones,
                            const int32_t *ssid, const int32_t *tmatid, const int32_t *start,
                            const int32_t *duration)
{
    const ssw_host_model_t *h = m->h;
    if (h->sseq == NULL || h->n_emit_state != 3) {
        ssw_set_error("alignment needs a 3-state mdef");
        return NULL;
    }
    if (n_phones < 1 || n_phones * 3 > 0xffff) { /* alignment vectors cap at 65535 entries */
        ssw_set_error("bad phone count %d", n_phones);
        return NULL;
    }
    ssw_state_align_search_s *s = new ssw_state_align_search_s();
    s->m = m;
    s->mgau = reinterpret_cast<ssw_mgau_impl *>(mgau);
    s->n_phones = n_phones;
    s->n_frames = 0;
    s->started = s->finished = 0;
    s->senid.resize((size_t)n_phones * 3);
    s->tmatid.resize(n_phones);
    s->sf.resize(n_phones);
    s->ef.resize(n_phones);
    s->state0.resize((size_t)n_phones * 3);
    for (int p = 0; p < n_phones; ++p) {
        if (ssid[p] < 0 || ssid[p] >= h->n_sseq) {
            ssw_set_error("phone %d: senone sequence %d out of range", p, ssid[p]);
            delete s;
            return NULL;
        }
        for (int j = 0; j < 3; ++j) {
            s->senid[(size_t)p * 3 + j] = h->sseq[(size_t)ssid[p] * 3 + j];
            /* alignment_populate: states inherit the phone's window, score 0 */
            s->state0[(size_t)p * 3 + j].start = start ? start[p] : 0;
            s->state0[(size_t)p * 3 + j].duration = duration ? duration[p] : 0;
            s->state0[(size_t)p * 3 + j].score = 0;
        }
        s->tmatid[p] = (int16_t)tmatid[p];
        int st = start ? start[p] : 0, du = duration ? duration[p] : 0;
        s->sf[p] = st > 0 ? st : 0;               /* state_align_search.c:464-467 */
        s->ef[p] = du > 0 ? st + du : INT_MAX;    /* :468-471 */
    }
    return s;
}

extern "C" int
ssw_state_align_search_start(ssw_state_align_search_t *s)
{
    s->n_frames = 0;
    s->feats.clear();
    s->started = 1;
    s->finished = 0;
    return 0;
}

extern "C" int
ssw_state_align_search_step(ssw_state_align_search_t *s, const float *feat, int frame_idx)
{
    if (!s->started || frame_idx != s->n_frames) {
        ssw_set_error("step(%d) out of order (next frame is %d)", frame_idx, s->n_frames);
        return -1;
    }
    const int dim = s->m->h->veclen_total;
    s->feats.insert(s->feats.end(), feat, feat + dim);
    s->n_frames++;
    return 0;
}

extern "C" int
ssw_state_align_search_finish(ssw_state_align_search_t *s)
{
    ssw_model_s *m = s->m;
    const ssw_host_model_t *h = m->h;
    const int n = s->n_frames;
    s->states = s->state0;
    s->phones.assign(s->n_phones, ssw_align_entry_t{ 0, 0, 0 });
    int16_t *d_scr = NULL;
    float *d_feats = NULL;
    int rv = -1, status = 0;
    int32_t foff[2] = { 0, n }, poff[2] = { 0, s->n_phones };
    std::vector<int32_t> parent((size_t)s->n_phones * 3);
    if (hipSetDevice(m->device) != hipSuccess
        || hipMalloc((void **)&d_scr, sizeof(int16_t) * (size_t)(n ? n : 1) * h->n_sen) != hipSuccess
        || hipMalloc((void **)&d_feats, sizeof(float) * (size_t)(n ? n : 1) * h->veclen_total)
            != hipSuccess) {
        ssw_set_error("device allocation failed");
        goto out;
    }
    if (n > 0) {
        if (hipMemcpy(d_feats, s->feats.data(), sizeof(float) * s->feats.size(),
                      hipMemcpyHostToDevice) != hipSuccess) {
            ssw_set_error("feature upload failed");
            goto out;
        }
        if (ssw_score_batch(m, s->mgau ? s->mgau->scorer : SSW_SCORER_PTM, d_feats, n, foff, 1,
                            d_scr, NULL) < 0)
            goto out;
    }
    if (ssw_align_batch(m, d_scr, 1, foff, poff, s->senid.data(), s->tmatid.data(), s->sf.data(),
                        s->ef.data(), s->states.data(), &status, NULL) < 0)
        goto out;
    if (status != 0) {
        if (status == -1)
            ssw_set_error("Failed to reach final state in alignment");
        else
            ssw_set_error("Alignment failed in frame %d", -status - 2);
        goto out;
    }
    for (size_t i = 0; i < parent.size(); ++i)
        parent[i] = (int32_t)(i / 3);
    if (ssw_alignment_propagate(s->states.data(), parent.data(), (int32_t)parent.size(),
                                s->phones.data(), s->n_phones) < 0)
        goto out;
    s->finished = 1;
    rv = 0;
out:
    (void)hipFree(d_scr);
    (void)hipFree(d_feats);
    return rv;
}

extern "C" int32_t
ssw_state_align_search_n_frames(const ssw_state_align_search_t *s)
{
    return s->n_frames;
}

extern "C" const ssw_align_entry_t *
ssw_state_align_search_states(const ssw_state_align_search_t *s, int32_t *n)
{
    if (n)
        *n = (int32_t)s->states.size();
    return s->states.data();
}

extern "C" const ssw_align_entry_t *
ssw_state_align_search_phones(const ssw_state_align_search_t *s, int32_t *n)
{
    if (n)
        *n = (int32_t)s->phones.size();
    return s->phones.data();
}

extern "C" void
ssw_state_align_search_free(ssw_state_align_search_t *s)
{
    delete s;
}

/* ---------------------------------------------------------------------------------- */
/* dynamic features                                                 */
/* ---------------------------------------------------------------------------------- */
extern "C" int
ssw_feat_batch(ssw_model_t *m, const float *d_cep, int32_t n_frames, const int32_t *utt_off,
               int32_t n_utts, int32_t ncep, float *d_out, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n_frames == 0 || n_utts == 0)
        return 0;
    if (m->device == SSW_DEVICE_NONE) {
        ssw_set_error("model was loaded with device = SSW_DEVICE_NONE: no GPU, no CPU fallback");
        return -1;
    }
    if (ncep < 1 || ncep > 64 || n_frames < 0 || n_utts < 0 || utt_off == NULL || utt_off[0] != 0
        || utt_off[n_utts] != n_frames) {
        ssw_set_error("bad arguments to ssw_feat_batch");
        return -1;
    }
    HIP_OK(hipSetDevice(m->device));
    int *d_off = NULL;
    HIP_OK(hipMalloc((void **)&d_off, sizeof(int) * ((size_t)n_utts + 1)));
    hipError_t e = hipMemcpyAsync(d_off, utt_off, sizeof(int) * ((size_t)n_utts + 1),
                                  hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        FeatParams F;
        F.cep = d_cep;
        F.utt_off = d_off;
        F.out = d_out;
        F.n_utts = n_utts;
        F.ncep = ncep;
        hipLaunchKernelGGL(feat_1s_c_d_dd_kernel, dim3(n_utts), dim3(64), 0, st, F);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipStreamSynchronize(st);
    (void)hipFree(d_off);
    if (e != hipSuccess) {
        ssw_set_error("ssw_feat_batch: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

/* ---------------------------------------------------------------------------------- */
/* device-memory helpers                                                               */
/* ---------------------------------------------------------------------------------- */
extern "C" void *
ssw_device_malloc(size_t nbytes)
{
    void *p = NULL;
    if (hipMalloc(&p, nbytes ? nbytes : 1) != hipSuccess) {
        ssw_set_error("hipMalloc(%zu) failed", nbytes);
        return NULL;
    }
    return p;
}

extern "C" void
ssw_device_free(void *d_ptr)
{
    (void)hipFree(d_ptr);
}

extern "C" int
ssw_memcpy_h2d(void *d_dst, const void *src, size_t nbytes)
{
    HIP_OK(hipMemcpy(d_dst, src, nbytes, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int
ssw_memcpy_d2h(void *dst, const void *d_src, size_t nbytes)
{
    HIP_OK(hipMemcpy(dst, d_src, nbytes, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int
ssw_device_synchronize(void)
{
    HIP_OK(hipDeviceSynchronize());
    return 0;
}
