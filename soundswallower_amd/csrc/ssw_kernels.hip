/*
 * ssw_kernels.hip -- gfx950 (MI355X, CDNA4) kernels for the SoundSwallower acoustic hot path
 * and the C ABI of include/ssw_amd.h on top of them.
 *
 * All arithmetic that the reference does in float32 is done here in float32 with one rounding
 * per operation (no FMA contraction: this file is built with -ffp-contract=off and carries the
 * pragma below), in the reference's operation order; everything else is int32.  Citations are
 * file:line in the SoundSwallower tree.
 *
 * One translation unit, split over .inc files for reading (device code first, inside one
 * anonymous namespace, then the host side):
 *   ssw_dev_common.inc   truncation, density, wave reductions, LDS-only barrier
 *   ssw_k1a_chain.inc    ptm_topn_chain_kernel (exact frame-sequential top-N: eval_topn + eval_cb,
 *                        src/ptm_mgau.c:86-225) and topn_exact_step, the exact step the scans'
 *                        in-wave pass shares
 *   ssw_k1a_frames.inc   ptm_topn_frames_kernel (speculative history-free top-N on the vector
 *                        unit with a proof test per pair; the exact in-wave pass and its helpers)
 *   ssw_k1a_mfma.inc     ptm_topn_mfma_kernel: the same scan with its multiply-adds on the matrix
 *                        cores (keys from two-part binary16 operands, exact re-evaluation and
 *                        in-wave pass as above): what every batch takes
 *   ssw_k1b_senone.inc   ptm_senone_kernel (codebook_norm + senone_eval, src/ptm_mgau.c:264-403),
 *                        ptm_senone_frame_kernel (one frame, active sets), ms_senone_kernel
 *   ssw_k4_feat.inc      feat_1s_c_d_dd_kernel (batch CMN + 1s_c_d_dd, src/feat.c:271-326)
 *   ssw_k2_align.inc     viterbi_align_mw_kernel / _reg_kernel / viterbi_align_kernel
 *                        (state_align_search step/finish + hmm_vit_eval_3st_lr,
 *                        src/state_align_search.c:177-268, src/hmm.c:482-567)
 *   ssw_k2_anytopo.inc   viterbi_align_any_kernel: the same search over HMMs of 1, 2, 4 or 5 states
 *                        (hmm_vit_eval_5st_lr, hmm_vit_eval_anytopo, src/hmm.c:166-304, :671-739)
 *   ssw_k5_firstpass.inc first_pass_kernel (fsg_search start/step/finish over the linear
 *                        grammar's phone trees, src/fsg_search.c:665-925)
 *   ssw_k6_compact.inc   compact score rows: the plan of a batch of alignments (which of an
 *                        utterance's states share a senone, where each score goes), gather
 *   ssw_k7_fpactive.inc  the first pass in the default configuration (compallsen = no) as a
 *                        batch: per-frame listed sets from the search's exported HMM sets, the
 *                        senone kernel over them, the comparison that proves a trajectory
 *   ssw_host_*.inc       device model and loaders' upload, batched scoring, alignment, the
 *                        mgau_t / search-module shaped objects, features, device-memory helpers,
 *                        the RCCL gather of final alignments (ssw_host_comm.inc)
 */
#pragma clang fp contract(off)

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <functional>
#include <map>
#include <set>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
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
#include "ssw_dev_common.inc"
#include "ssw_k1a_chain.inc"
#include "ssw_k1a_frames.inc"
#include "ssw_k1a_mfma.inc"
#include "ssw_k1b_senone.inc"
#include "ssw_k4_feat.inc"
#include "ssw_k2_align.inc"
#include "ssw_k2_anytopo.inc"
#include "ssw_k5_firstpass.inc"
#include "ssw_k6_compact.inc"
#include "ssw_k7_fpactive.inc"

} // namespace

#include "ssw_host_model.inc"
#include "ssw_host_score.inc"
#include "ssw_host_align.inc"
#include "ssw_host_compact.inc"
#include "ssw_host_active.inc"
#include "ssw_host_mgau.inc"
#include "ssw_host_search.inc"
#include "ssw_host_feat.inc"
#include "ssw_host_firstpass.inc"
#include "ssw_host_fpactive.inc"
#include "ssw_host_devmem.inc"
#include "ssw_host_comm.inc"
