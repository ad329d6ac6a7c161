/*
 * ssw_amd.h -- C ABI of the MI355X-native SoundSwallower acoustic hot path.
 *
 * Everything here is `extern "C"`, plain pointers and sizes.  Each entry point names the
 * reference interface it stands in for (file:line relative to the SoundSwallower tree) so a
 * maintainer can bind it from the reference's C code; INTEGRATION.md shows the binding.
 *
 * Conventions follow the reference: int return, 0 on success, < 0 on error (the message is
 * kept per thread and returned by ssw_last_error()); constructors return NULL on error;
 * objects are freed by their *_free function.  Calls are synchronous unless a stream is passed.
 * Pointers named d_* are device (HIP) pointers, everything else is host memory.
 *
 * Threading: as the reference's decoder, a model is NOT re-entrant.  An ssw_model_t owns the
 * workspaces its calls use (top-N block, alignment and first-pass arenas, cached utterance
 * offsets, the score rows of ssw_align_text_batch): one call at a time per model, from any
 * thread; calls on different models -- one per host thread or per stream of work -- run side by
 * side.  Host-only calls (ssw_first_pass_prepare, ssw_alignment_populate, the dictionary and JSON
 * functions) only read the model and may run beside a device call on it.  The rule is enforced:
 * a compute call that finds another thread inside one on the same model fails with "the model is
 * in use by another thread" instead of sharing its workspaces.
 *
 * Inputs: non-finite features are not refused (checking every row would cost a pass over the
 * batch); every (codebook, stream) pair they touch fails the scan's proof test and is scored by
 * the exact in-wave pass.  PTM scorer (round 5): that pass does what the reference does with
 * them -- its (int32) casts of a NaN are x86's 0x80000000, and eval_cb's staged dimension walk
 * (src/ptm_mgau.c:150-225) drops every density of a stream with a NaN in dimensions 0..8 while
 * a NaN confined to the last four dimensions lets densities through with INT32_MIN -- so NaN
 * and +-Inf rows score as in the reference (tests/test_gpu_nonfinite.py, against the oracle
 * compiled on x86).  ms scorer: infinities likewise; with a NaN the reference's compute_dist
 * (src/ms_gauden.c:397-420) leaves its top-N entries as they were, i.e. reads the PREVIOUS
 * frame's ids from its buffer -- a result no batch can reproduce and none is claimed.
 * Range: any finite feature gives the reference's scores, but the fast path of the scoring
 * kernels (the matrix-core scan, binary16 operands) only trusts frames whose features lie
 * within +-255 -- cepstral features of the shipped front end are a tenth of that; a frame
 * beyond takes the exact in-wave pass, ~13 x the work.  Data in another scale altogether
 * (integer-scaled cepstra) is scored fastest with SSW_SCAN=fma, the vector-unit scan, which
 * has fp32's range.
 */
#ifndef SSW_AMD_H
#define SSW_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSW_ABI_VERSION 2 /* 2: ssw_model_info_t grew (scan_mode ..), round 5 */

/* ------------------------------------------------------------------------------------ */
/* Configuration: the scalar parameters the path reads from config_t                     */
/* (include/soundswallower/config_defs.h:94-97, 198-253).                                */
/* ------------------------------------------------------------------------------------ */
typedef struct ssw_config_s {
    double logbase;   /* "logbase"   1.0001 */
    double varfloor;  /* "varfloor"  1e-4   */
    double mixwfloor; /* "mixwfloor" 1e-7   */
    double tmatfloor; /* "tmatfloor" 1e-4   */
    int32_t topn;     /* "topn"      4      */
    int32_t ds;       /* "ds"        1      */
    int32_t aw;       /* "aw"        1      */
    int32_t device;   /* HIP device ordinal; -1 = current device; SSW_DEVICE_NONE = load the
                         host tables only (loader checks), every compute call then fails */
} ssw_config_t;
#define SSW_DEVICE_NONE (-2)

void ssw_config_defaults(ssw_config_t *cfg);
const char *ssw_last_error(void);
int ssw_abi_version(void);

/* ------------------------------------------------------------------------------------ */
/* Acoustic model: replaces the loading half of acmod_load_am (src/acmod.c:62-129):      */
/* bin_mdef_read (src/bin_mdef.c:310), tmat_init (src/tmat.c:107), gauden_init           */
/* (src/ms_gauden.c:304), read_sendump / read_mixw (src/ptm_mgau.c:456, 611),            */
/* senone_mixw_read (src/ms_senone.c:103).  Tables are derived on the host exactly as    */
/* the reference derives them, then laid out for the GPU and uploaded once.              */
/* ------------------------------------------------------------------------------------ */
typedef struct ssw_model_s ssw_model_t;

/* sendump or mixw may be NULL (one of them is required for scoring); mdef/tmat may be NULL
 * when only scoring tables are wanted and n_sen can be taken from mixw. */
ssw_model_t *ssw_model_load(const char *mdef, const char *means, const char *variances,
                            const char *sendump, const char *mixw, const char *tmat,
                            const ssw_config_t *cfg);
void ssw_model_free(ssw_model_t *m);

typedef struct ssw_model_info_s {
    int32_t n_cb, n_feat, n_density, veclen_total;
    int32_t n_sen, n_ci_sen, n_ciphone, n_phone, n_emit_state, n_tmat, n_sseq, sil;
    int32_t n_floored, topn, has_ptm, has_ms, device;
    int32_t veclen[8];
    /* ABI 2.  The speculative top-N scan runs on the matrix cores only on a device that has
     * passed the load-time self-test of the one hardware property its error bound assumes (the
     * accumulation error of v_mfma_f32_32x32x16_f16, csrc/ssw_model.c): ssw_model_load measures
     * it with adversarial and random cancelling tiles against fp64 / closed-form sums, and on a
     * failure (or a HIP error) every batch takes the vector-unit scan, whose bound is replayed on
     * the CPU.  scan_mode: 1 matrix cores, 0 vector unit.  mfma_selftest: 1 passed, -1 failed
     * (fell back), 0 not run (no device, no scan tables, SSW_MFMA_SELFTEST=off). */
    int32_t scan_mode, mfma_selftest;
    float mfma_selftest_worst_u; /* worst |D - exact| / sum |terms| seen, in u = 2^-24 */
    float mfma_selftest_ms;      /* what the self-test added to ssw_model_load */
} ssw_model_info_t;
int ssw_model_info(const ssw_model_t *m, ssw_model_info_t *out);
/* one line on the self-test's outcome ("" when it did not run); owned by the model */
const char *ssw_model_selftest_message(const ssw_model_t *m);

/* Host copies of the derived tables, for loader parity checks.  Returns a pointer owned by
 * the model and writes the byte size.  Layouts: MEAN/VAR float32 in s3 file order
 * [cb][feat][density][veclen]; DET float32 [cb][feat][density]; PTM_MIXW uint8
 * [feat][density][n_sen]; MS_PDF uint8 [sen][feat][density]; TP uint8 [tmat][n][n+1];
 * SSEQ uint16 [n_sseq][n_emit]; SEN2CB int16 [n_sen]; LOGADD8 uint8[256];
 * PHONE_SSID / PHONE_TMAT int32 [n_phone].  The last four are the device-layout Gaussian
 * tables (no reference counterpart; exposed so the scan's error bound can be replayed on
 * the CPU): REC float32 [cb*feat][density][32] = mean[0..15) | det[15] | scale[16..31);
 * SCAN_REC the same shape holding the quadratic form a[0..15) | c[15] | b[16..31);
 * SCAN_D0 float32 [cb*feat][32], element 0 = the codebook's reference det (and, for the
 * matrix-core scan, 1 = 2^s, 2 = 2^-s, 3 = 2^ec: its operands' scales, csrc/ssw_model.c);
 * SCAN_EXACT uint32 [cb*feat][132] = count, then the densities scored in the exact form;
 * SCAN_REC_MFMA / SCAN_EXACT_MFMA the same two for the matrix-core scan (its own error
 * constant; its c is what the two parts below add up to, unscaled), SCAN_WFRAG uint16
 * (binary16) [cb*feat][4][2][2][64][8]: those records, scaled by 2^-s (c by 2^-(s + ec)), cut
 * into two binary16 parts in MFMA A-fragment order. */
enum ssw_table {
    SSW_TAB_MEAN = 0, SSW_TAB_VAR, SSW_TAB_DET, SSW_TAB_PTM_MIXW, SSW_TAB_MS_PDF, SSW_TAB_TP,
    SSW_TAB_SSEQ, SSW_TAB_SEN2CB, SSW_TAB_LOGADD8, SSW_TAB_PHONE_SSID, SSW_TAB_PHONE_TMAT,
    SSW_TAB_REC, SSW_TAB_SCAN_REC, SSW_TAB_SCAN_D0, SSW_TAB_SCAN_EXACT,
    SSW_TAB_SCAN_REC_MFMA, SSW_TAB_SCAN_EXACT_MFMA, SSW_TAB_SCAN_WFRAG
};
const void *ssw_model_table(const ssw_model_t *m, int which, size_t *nbytes);

/* ------------------------------------------------------------------------------------ */
/* Batched senone scoring (NEW: the reference scores one frame per call).                */
/* One call scores every frame of a batch of utterances with compallsen = yes:           */
/*   acmod_score -> ptm_mgau_frame_eval   (src/acmod.c:822-860, src/ptm_mgau.c:408-454)   */
/*   acmod_score -> ms_cont_mgau_frame_eval (src/ms_mgau.c:278-368)                       */
/* feats: float32 [n_frames][veclen_total] (streams back to back, the row acmod keeps in  */
/* feat_buf, src/feat.c:386-395).  utt_off: int32 [n_utts+1], frame offsets; every         */
/* utterance starts from the reset top-N history {cw = m, score = INT32_MIN}              */
/* (src/ptm_mgau.c:706-713).  out: int16 [n_frames][n_sen].                               */
/* ------------------------------------------------------------------------------------ */
enum ssw_scorer { SSW_SCORER_PTM = 0, SSW_SCORER_MS = 1 };

/* device pointers, asynchronous on `stream` (a hipStream_t, NULL = default stream) */
int ssw_score_batch(ssw_model_t *m, int scorer, const float *d_feats, int32_t n_frames,
                    const int32_t *utt_off, int32_t n_utts, int16_t *d_out, void *stream);
/* The same with the reference's history semantics made available to batches.  The reference
 * never resets the PTM top-N history after start-up: it carries from utterance to utterance
 * (acmod_start_utt only resets frame_idx, src/acmod.c:367) and across acmod_rewind from the
 * first pass of forced alignment into the second (src/decoder.c:786-793, src/ptm_mgau.c:425-448);
 * only the codeword ORDER matters (tie-breaks among equal truncated scores).
 * The reference's history is a ring of TWO slots indexed by frame % 2 (n_fast_hist = 2,
 * src/ptm_mgau.c:425-437, :804) and frame numbers restart at 0 with every utterance and after
 * acmod_rewind: frame t > 0 starts from frame t - 1, but frame 0 copies slot 1 -- the final order
 * of the last ODD-numbered frame scored before it.  After an utterance of even length that is
 * its last frame; after one of odd length T >= 3 it is frame T - 2; an utterance of one frame
 * leaves slot 1 as it found it.
 *   flags      SSW_SCORE_CARRY_UTTS: no reset at the utterance boundaries of the batch: each
 *              utterance's frame 0 starts from slot 1 as the utterances before it left it
 *              SSW_SCORE_CARRY_OUT_REWIND: see carry_out
 *   carry_in   optional uint32 [n_cb * n_feat], 4 codewords packed best first: the order the
 *              batch's first frame starts from (NULL = the reset history)
 *   carry_out  optional uint32 [n_cb * n_feat] (makes the call synchronous).  Default: the order
 *              after the batch's last frame -- carry_in of a call that CONTINUES the same utterance
 *              (cut it after an even number of frames if an utterance boundary with
 *              SSW_SCORE_CARRY_UTTS follows, so that the parity of its frames is kept).  With
 *              SSW_SCORE_CARRY_OUT_REWIND: slot 1 after the batch -- carry_in of the next
 *              UTTERANCE, or of the second pass of decoder_alignment over the same utterance after
 *              acmod_rewind (with SSW_SCORE_CARRY_UTTS the batch's utterances are walked back to
 *              the last one with two frames; without it only the last utterance counts; if no
 *              frame wrote the slot, carry_in / the reset order comes back).
 * The ms scorer keeps no history: the three are ignored for it. */
#define SSW_SCORE_CARRY_UTTS 1u
/* the caller runs kernels of ANOTHER stream beside this call (the alignment of the previous
 * chunk beside the scoring of the next, section "Multi-GPU" / soundswallower_amd/jobs.py): the
 * persistent scoring workgroups then leave after 8 frame pairs instead of staying for the
 * whole batch, so the other stream finds free wave slots every ~100 us.  Same results. */
#define SSW_SCORE_SHARE_DEVICE 2u
#define SSW_SCORE_CARRY_OUT_REWIND 4u
int ssw_score_batch_ex(ssw_model_t *m, int scorer, const float *d_feats, int32_t n_frames,
                       const int32_t *utt_off, int32_t n_utts, int16_t *d_out, void *stream,
                       uint32_t flags, const uint32_t *carry_in, uint32_t *carry_out);
/* host pointers, synchronous: copies in, scores, copies out -- as a three-stage pipeline over
 * sub-batches of whole utterances (~1024 frames) through pinned staging.  History reset per
 * utterance.  An utterance of more than 16,384 frames (SSW_HOST_PIPE_CAP) is scored in pieces
 * of that many that hand the history on, so the staging the model keeps for this call stops at
 * 2 x 16,384 feature rows + score rows pinned on the host and as many on the device (en-us:
 * 2 x 171 MB each), however long the utterances are. */
int ssw_score_batch_host(ssw_model_t *m, int scorer, const float *feats, int32_t n_frames,
                         const int32_t *utt_off, int32_t n_utts, int16_t *out);
/* Debug/parity view of the PTM top-N state after normalisation for every frame of the last
 * ssw_score_batch* call: cw uint8 / score int32 laid out [n_frames][n_cb][n_feat][topn]. */
int ssw_score_batch_topn(ssw_model_t *m, int32_t n_frames, uint8_t *cw, int32_t *score);
/* Debug/parity view of the matrix core itself: D = A B + C, one v_mfma_f32_32x32x16_f16 per tile
 * (host pointers; A [n_tiles][32][16], B [n_tiles][16][32] binary16 bit patterns, C, D
 * [n_tiles][32][32] float).  The scan's error bound assumes an accumulation error of at most
 * 34 x 2^-24 of the sum of the |terms| for this instruction (csrc/ssw_model.c);
 * tests/test_gpu_mfma_bound.py measures it with this call. */
int ssw_debug_mfma_f16_tiles(ssw_model_t *m, const uint16_t *A, const uint16_t *B, const float *C,
                             float *D, int32_t n_tiles);
/* Debug/parity view of the matrix-core scan: the raw keys (upper bounds of the densities,
 * relative to the codebook's reference level SCAN_D0, before their widening) of codebook x
 * stream `cbf` for every frame of d_feats, computed exactly as the scan computes them.
 * keys: host float [n_frames][128]; inert rows (SCAN_EXACT_MFMA densities) read about -3e38. */
int ssw_debug_scan_keys(ssw_model_t *m, const float *d_feats, int32_t n_frames, int32_t cbf,
                        float *keys);
/* Counters of the last PTM batch: [0] = (chain,frame) pairs the history-free pass could not
 * prove order-independent and handed to the exact sequential pass, [1] = pairs total. */
int ssw_score_batch_stats(ssw_model_t *m, int64_t stats[2]);
/* In-process audit of the matrix-core scan (round 5).  With SSW_SCAN_AUDIT=k in the environment
 * when the model is loaded, every k-th wave of the scan hands ALL of its frames to the exact
 * in-wave pass -- proven or not -- and compares what the exact pass finds for a pair the scan had
 * proven with what the scan wrote for it.  stats[0] = proven pairs so audited, stats[1] = of
 * those, pairs that differ: 0 unless the scan's error bound does not hold on this device (the
 * load-time self-test of ssw_model_info_t checks the same assumption on synthetic tiles; the
 * audit checks the claim itself on the caller's data).  Running totals since ssw_model_load;
 * results are the same with and without the audit.  Cost at k = 1: the scan ~13 x. */
int ssw_scan_audit_stats(ssw_model_t *m, int64_t stats[2]);

/* Per-kernel timing with HIP events recorded on the launch stream (for bench.py's roofline
 * line).  When enabled every ssw_score_batch call brackets each kernel with events;
 * ssw_get_kernel_timing synchronises and returns the last call's milliseconds.  Return value 2:
 * ms[0] = top-N (density) kernel(s), ms[1] = senone kernel.  Return value 1 (a batch large
 * enough to be scored in pieces, scan and senone launches alternating: there is no split):
 * ms[0] = the whole call (ms[1] is set to 0 when n >= 2).  An event between two launches costs
 * ~2 us of its own.  -1 on error. */
int ssw_set_kernel_timing(ssw_model_t *m, int enable);
int ssw_get_kernel_timing(ssw_model_t *m, float *ms, int n);
/* measurement aid: `reps` ssw_score_batch calls on the same batch back to back, then one
 * synchronise (tools/bench_frame_sync.py) */
int ssw_debug_score_loop(ssw_model_t *m, int scorer, const float *d_feats, int32_t n_frames,
                         const int32_t *utt_off, int32_t n_utts, int16_t *d_out, int32_t reps,
                         void *stream);

/* ------------------------------------------------------------------------------------ */
/* Scorer object: drop-in for mgau_t / mgaufuncs_t (include/soundswallower/acmod.h:93-111).*/
/* The first two members mirror mgau_t so acmod can write frame_idx                       */
/* (src/acmod.c:367,748,760) and call through vt.                                         */
/* ------------------------------------------------------------------------------------ */
typedef struct ssw_mgau_s ssw_mgau_t;
typedef struct ssw_mgaufuncs_s {
    const char *name;
    int (*frame_eval)(ssw_mgau_t *mgau, int16_t *senscr, uint8_t *senone_active,
                      int32_t n_senone_active, float **feat, int32_t frame,
                      int32_t compallsen);
    int (*transform)(ssw_mgau_t *mgau, void *mllr); /* MLLR is out of scope: returns -1 */
    void (*free)(ssw_mgau_t *mgau);
} ssw_mgaufuncs_t;
struct ssw_mgau_s {
    ssw_mgaufuncs_t *vt;
    int frame_idx;
};

/* ptm_mgau_init(acmod_t *) (ptm_mgau.h:94) / ms_mgau_init(acmod_t *) (ms_mgau.h) */
ssw_mgau_t *ssw_ptm_mgau_init(ssw_model_t *m);
ssw_mgau_t *ssw_ms_mgau_init(ssw_model_t *m);
/* ptm_mgau_reset_fast_hist (src/ptm_mgau.c:694) without its reallocation */
void ssw_mgau_reset_hist(ssw_mgau_t *mgau);
/* NEW: score a whole utterance in one batch and cache it; frame_eval(frame) then copies row
 * `frame`.  Without it frame_eval scores one frame per call on the GPU. */
int ssw_mgau_prescore(ssw_mgau_t *mgau, const float *feats, int32_t n_frames);

/* ------------------------------------------------------------------------------------ */
/* Forced alignment: state_align_search (src/state_align_search.c) + hmm_vit_eval         */
/* (src/hmm.c:741) + the state level of alignment_t (include/soundswallower/alignment.h). */
/* ------------------------------------------------------------------------------------ */
typedef struct ssw_align_entry_s {
    int32_t start, duration, score; /* alignment_entry_t.{start,duration,score} */
} ssw_align_entry_t;

/* Batch of independent utterances.
 *   d_senscr  int16 [total_frames][n_sen]   scores as acmod_score returns them
 *   frame_off int32 [n_utts+1]              rows of d_senscr per utterance
 *   phone_off int32 [n_utts+1]              phones per utterance (offsets into per-phone arrays)
 *   senid     uint16 [total_phones][n_emit] mdef->sseq[ssid][j] (src/hmm.c:99)
 *   tmatid    int16 [total_phones]
 *   sf, ef    int32 [total_phones]          state_align_search.c:464-471 (0 / INT_MAX = free)
 *   state_io  [total_phones*n_emit]         in: what alignment_populate left in the state
 *                                           entries (src/ps_alignment.c:237-239); out: backtrace
 *   status    int32 [n_utts]                0 ok, -1 "Failed to reach final state",
 *                                           -(2+frame) "Alignment failed in frame"
 * All host pointers except d_senscr.  Synchronous.
 * n_emit is the transition matrices' (hmm_vit_eval, src/hmm.c:741-759): 3 takes
 * hmm_vit_eval_3st_lr in the kernels this library is built around; 5 (hmm_vit_eval_5st_lr) and
 * 1, 2, 4 (hmm_vit_eval_anytopo) take one plain kernel, a wave per utterance.  More than
 * HMM_MAX_NSTATE = 5 states is refused, as hmm_context_init does (src/hmm.c:60-64). */
int ssw_align_batch(ssw_model_t *m, const int16_t *d_senscr, int32_t n_utts,
                    const int32_t *frame_off, const int32_t *phone_off, const uint16_t *senid,
                    const int16_t *tmatid, const int32_t *sf, const int32_t *ef,
                    ssw_align_entry_t *state_io, int32_t *status, void *stream);
/* How the last alignments were made (round 5).  Utterances of up to 1024 phones first go through
 * a kernel that keeps, per state and frame, a 2-bit back-pointer instead of the reference's
 * {history, score} token (src/state_align_search.c:149-175) and recovers the scores the backtrace
 * needs by replaying the path; what that kernel cannot follow -- the reference's stale-exit-score
 * and left-over-t2 cases (src/hmm.c:496-520), a backtrace that runs into a missing token -- it
 * hands, untouched, to the kernel with the full tokens.  stats[0] = utterances that took the
 * byte-token kernel, stats[1] = of those, the ones that were handed on; running totals since
 * ssw_model_load.  Results are the same either way; SSW_ALIGN_BT=0 skips the first kernel. */
int ssw_align_stats(ssw_model_t *m, int64_t stats[2]);
/* The same search in the reference's DEFAULT configuration (compallsen = no), scoring included:
 * acmod scores only the senones of the HMMs the search has active (acmod_activate_hmm /
 * acmod_flags2list, src/acmod.c:905-999 -- a uint8 delta list whose gaps above 255 list extra
 * senones), the PTM scorer scans only the codebooks those senones belong to, normalises over
 * them and subtracts the best LISTED score from every entry (src/ptm_mgau.c:264-403), and this
 * search never clears acmod's set (src/state_align_search.c:186-189): frame t sees the seed --
 * what the first pass left active, seed_active uint32 [n_utts][(n_sen + 31) / 32], bit b of
 * word b / 32, or NULL -- plus the senones of every phone entered so far.  Which phones are
 * entered when follows from sf / ef alone when those do not decrease along an utterance (the
 * windows alignment_populate produces); otherwise the call is refused.  d_feats: feature rows
 * [total_frames][veclen_total]; d_senscr: optional int16 [total_frames][n_sen] that receives
 * the scores exactly as acmod->senone_scores would hold them frame by frame.  Other arguments
 * and results as ssw_align_batch.  PTM scorer, history reset per utterance.
 * Limits (the call aligns over compact score rows, see below): 3-state HMMs, at most 64
 * codebooks, utterances of at most 10,922 phones (a row's per-state indices are 16 bits) --
 * beyond them the call is refused with an error; ssw_score_batch + ssw_align_batch take any
 * length but are the compallsen = yes configuration, i.e. other scores. */
int ssw_align_batch_active(ssw_model_t *m, const float *d_feats, int32_t n_utts,
                           const int32_t *frame_off, const int32_t *phone_off,
                           const uint16_t *senid, const int16_t *tmatid, const int32_t *sf,
                           const int32_t *ef, const uint32_t *seed_active,
                           ssw_align_entry_t *state_io, int32_t *status, int16_t *d_senscr,
                           void *stream);
/* The same with the scorer named (round 3): SSW_SCORER_MS follows ms_cont_mgau_frame_eval's
 * active-list half (src/ms_mgau.c:322-365): only the listed senones are evaluated and written,
 * normalised by the best of them with the int16 clamp (bridge entries of the delta list are
 * listed entries).  d_senscr rows hold 0 wherever the frame's list has no entry: acmod's buffer
 * would still show the last value of an entry that has dropped out of the list -- only bridge
 * entries ever do, when a senone between their neighbours joins the set, and no HMM reads them. */
int ssw_align_batch_active_ex(ssw_model_t *m, int scorer, const float *d_feats, int32_t n_utts,
                              const int32_t *frame_off, const int32_t *phone_off,
                              const uint16_t *senid, const int16_t *tmatid, const int32_t *sf,
                              const int32_t *ef, const uint32_t *seed_active,
                              ssw_align_entry_t *state_io, int32_t *status, int16_t *d_senscr,
                              void *stream);
/* Compact score rows (NEW, round 5).  state_align_search only ever reads the scores of its own
 * phones' senones -- it activates just those (src/state_align_search.c:186-189), and in the
 * default configuration acmod then scores nothing else (src/acmod.c:905-945).  When the phone
 * strings of a batch are known before it is scored (BASELINE configs 3 and 5; the second pass of
 * decoder_alignment, whose phones come from alignment_populate), a PLAN lets the scorer store,
 * per utterance, rows of that utterance's states only -- [n_frames_u][3 n_phones_u rounded up to
 * even] int16 instead of [n_frames_u][n_sen]: 0.9 KB instead of 10 KB per frame for a 150-phone
 * utterance of en-us -- which the alignment kernel then reads coalesced instead of gathering
 * three scattered 2-byte values per phone from rows it touches nearly every sector of.  Every
 * senone is still evaluated (compallsen = yes: the frame's best score is taken over all of
 * them, src/ptm_mgau.c:394-400): the scores, and so the alignments, are those of
 * ssw_score_batch + ssw_align_batch bit for bit.  A senone several states of an utterance
 * share is stored once, at the first of them.
 *   ssw_compact_plan_create   frame_off / phone_off / senid as ssw_align_batch takes them (host);
 *                             builds the device tables on `stream` and returns when they are
 *                             complete.  Utterances of up to 10,922 phones.
 *   ssw_compact_plan_elems    int16 elements the compact buffer needs (the caller allocates it)
 *   ssw_compact_plan_rows     utterance u's rows: offset (elements) and row length in the buffer;
 *                             state k of the utterance (3 per phone) at column k, or at the
 *                             column of the first state with the same senone
 *   ssw_score_batch_compact   ssw_score_batch_ex for the plan's batch (utterance offsets = the
 *                             plan's frame_off; history reset per utterance), asynchronous on
 *                             `stream`; flags: SSW_SCORE_SHARE_DEVICE
 *   ssw_align_batch_compact   ssw_align_batch over those rows (tmatid, sf, ef, state_io, status
 *                             as there; the utterances, phones and senones are the plan's);
 *                             flags: SSW_ALIGN_STATE_OUT_ONLY -- state_io is output only, the
 *                             entries the backtrace does not write (states a path skips,
 *                             utterances without an alignment) come back as zeros instead of
 *                             what the caller passed in: saves the upload of 36 bytes per phone
 * A plan is bound to its model and can be used any number of times. */
typedef struct ssw_compact_plan_s ssw_compact_plan_t;
ssw_compact_plan_t *ssw_compact_plan_create(ssw_model_t *m, int32_t n_utts,
                                            const int32_t *frame_off, const int32_t *phone_off,
                                            const uint16_t *senid, void *stream);
void ssw_compact_plan_free(ssw_compact_plan_t *plan);
size_t ssw_compact_plan_elems(const ssw_compact_plan_t *plan);
int ssw_compact_plan_rows(const ssw_compact_plan_t *plan, int32_t utt, long long *row_off,
                          int32_t *stride);
int ssw_score_batch_compact(ssw_model_t *m, int scorer, const float *d_feats,
                            const ssw_compact_plan_t *plan, int16_t *d_compact, void *stream,
                            uint32_t flags);
int ssw_align_batch_compact(ssw_model_t *m, const ssw_compact_plan_t *plan,
                            const int16_t *d_compact, const int16_t *tmatid, const int32_t *sf,
                            const int32_t *ef, ssw_align_entry_t *state_io, int32_t *status,
                            void *stream, uint32_t flags);
#define SSW_ALIGN_STATE_OUT_ONLY 1u

/* alignment_propagate (src/ps_alignment.c:316-352): sums children into parents.
 * parent[i] = index of child i's parent; parents must appear in non-decreasing order. */
int ssw_alignment_propagate(const ssw_align_entry_t *child, const int32_t *parent,
                            int32_t n_child, ssw_align_entry_t *parent_out, int32_t n_parent);

/* Search-module shaped object: state_align_search_init/start/step/finish/free
 * (state_align_search.h:89-92; vtable slots of search_module.h:72-83).  The constructor takes
 * what state_align_search_init reads from the alignment's phone level
 * (src/state_align_search.c:458-471): per phone ssid, tmatid, start, duration. */
typedef struct ssw_state_align_search_s ssw_state_align_search_t;
ssw_state_align_search_t *ssw_state_align_search_init(ssw_model_t *m, ssw_mgau_t *mgau,
                                                      int32_t n_phones, const int32_t *ssid,
                                                      const int32_t *tmatid,
                                                      const int32_t *start,
                                                      const int32_t *duration);
int ssw_state_align_search_start(ssw_state_align_search_t *s);
/* feat = the frame's feature row; frames must be stepped in order from 0 */
int ssw_state_align_search_step(ssw_state_align_search_t *s, const float *feat, int frame_idx);
/* runs scoring + Viterbi + backtrace for all stepped frames; fills state/phone entries */
int ssw_state_align_search_finish(ssw_state_align_search_t *s);
int32_t ssw_state_align_search_n_frames(const ssw_state_align_search_t *s);
const ssw_align_entry_t *ssw_state_align_search_states(const ssw_state_align_search_t *s,
                                                       int32_t *n);
const ssw_align_entry_t *ssw_state_align_search_phones(const ssw_state_align_search_t *s,
                                                       int32_t *n);
void ssw_state_align_search_free(ssw_state_align_search_t *s);

/* ------------------------------------------------------------------------------------ */
/* Lexicon glue (host C): what decoder_alignment needs to go from       */
/* words + word windows to the per-phone rows state_align_search_init reads.               */
/* ------------------------------------------------------------------------------------ */
typedef struct ssw_dict_s ssw_dict_t;
/* dict_init (src/dict.c): text dictionaries "word PH PH ...", main then filler; <s>, </s>,
 * <sil> always exist.  Words with unknown phones are skipped. */
ssw_dict_t *ssw_dict_load(const ssw_model_t *m, const char *dict_path, const char *filler_path);
void ssw_dict_free(ssw_dict_t *d);
int32_t ssw_dict_size(const ssw_dict_t *d);
/* CI phone ids of a word's pronunciation; returns its length or -1 when unknown */
int32_t ssw_dict_pron(const ssw_dict_t *d, const char *word, int32_t *ciphones, int32_t max);
const char *ssw_ciphone_name(const ssw_model_t *m, int32_t ci);
/* bin_mdef_phone_id_nearest (src/bin_mdef.c:666-720); pos: 0 internal, 1 begin, 2 end, 3 single */
int32_t ssw_phone_id_nearest(const ssw_model_t *m, int32_t b, int32_t l, int32_t r, int32_t pos);
/* alignment_add_word per word + alignment_populate (src/ps_alignment.c:114-247): returns the
 * number of phones written (ssid, tmatid, and optionally cipid, parent word, and the word's
 * start/duration copied to each phone, which is what state_align_search_init turns into
 * sf/ef), or -1 (unknown word, too many phones). */
int32_t ssw_alignment_populate(const ssw_model_t *m, const ssw_dict_t *d, int32_t n_words,
                               const char *const *words, const int32_t *start,
                               const int32_t *duration, int32_t max_phones, int32_t *ssid,
                               int32_t *tmatid, int32_t *cipid, int32_t *parent,
                               int32_t *ph_start, int32_t *ph_duration);

/* Dictionary word ids (dict_wordid / dict_wordstr); alternates are words of their own,
 * "forward(2)".  ssw_dict_word returns NULL and ssw_dict_word_id -1 when unknown. */
const char *ssw_dict_word(const ssw_dict_t *d, int32_t wid);
int32_t ssw_dict_word_id(const ssw_dict_t *d, const char *word);
/* dict_basewid (the word an alternate belongs to; itself otherwise) and dict_filler_word
 * (from the filler dictionary, <s> and </s> excepted; src/dict.c:373-384) */
int32_t ssw_dict_base_id(const ssw_dict_t *d, int32_t wid);
int32_t ssw_dict_is_filler(const ssw_dict_t *d, int32_t wid);

/* ------------------------------------------------------------------------------------ */
/* First pass of forced alignment: which fillers and alternate          */
/* pronunciations the text is spoken with, and every word's frames -- the word windows the  */
/* second pass (ssw_align_batch) is constrained to.  Replaces, for the linear grammar        */
/* decoder_set_align_text builds (src/decoder.c:686-735):                                    */
/*   fsg_search_init: silence / filler loops, alternates     src/fsg_search.c:84-170, 172-253 */
/*   fsg_lextree_init: per-state phone trees                 src/fsg_lextree.c:85-276, 356-660 */
/*   fsg_search_start / _step / _finish: beam Viterbi        src/fsg_search.c:665-852          */
/*   fsg_history_entry_add: word exits                       src/fsg_history.c:129-205         */
/*   fsg_search_find_exit / _seg_iter: backtrace             src/fsg_search.c:854-925,1085-1143 */
/* The graphs are built on the host, the search runs on the GPU, one workgroup per          */
/* utterance, over senone scores already in HBM (ssw_score_batch).                          */
/* ------------------------------------------------------------------------------------ */
typedef struct ssw_first_pass_config_s {
    double beam, pbeam, wbeam; /* 1e-48, 1e-48, 7e-29 (config_defs.h) */
    double wip, pip;           /* 0.65, 1.0 */
    float lw;                  /* 6.5 */
    float silprob, fillprob;   /* 0.005, 1e-8 */
    int32_t use_filler;        /* fsgusefiller, 1 */
    int32_t use_altpron;       /* fsgusealtpron, 1 */
    /* ssw_align_text_batch only, PTM scorer: score the second pass as decoder_alignment does
     * (src/decoder.c:786-793, src/ptm_mgau.c:425-448) -- again, after the rewind, every
     * utterance starting from the top-N history its own first pass left -- instead of feeding
     * both passes with the scores made from the reset history.  The two differ only where
     * truncated densities tie; costs a second scoring pass.  0 by default. */
    int32_t two_pass_history;
} ssw_first_pass_config_t;
void ssw_first_pass_config_defaults(ssw_first_pass_config_t *cfg);

typedef struct ssw_word_seg_s {
    int32_t wid;      /* dictionary word id (ssw_dict_word): text word, alternate or filler */
    int32_t start;    /* first frame */
    int32_t duration; /* frames */
    int32_t score;    /* path score at the word's exit (fsg_hist_entry_score) */
} ssw_word_seg_t;

/* d_senscr: int16 [n_frames][n_sen] device scores of the whole batch (utterance u owns
 * frames utt_off[u] .. utt_off[u+1]); words: the texts, utterance u = words[word_off[u] ..
 * word_off[u+1]).  seg: [n_utts][max_seg]; n_seg[u] = segments written, or -1 when the
 * grammar's final state is not reached in the last frame that has word exits ("Final result
 * does not match the grammar", src/fsg_search.c:913-916), or -(2 + k) when the search succeeded
 * but its k segments do not fit max_seg (call again with max_seg >= k; ssw_forced_align_batch
 * and ssw_align_text_batch do that themselves).
 * cfg NULL = defaults.  Returns 0, or -1 (unknown word, graph too large, no device).
 * Synchronous on `stream`. */
int ssw_first_pass_batch(ssw_model_t *m, const ssw_dict_t *d, const ssw_first_pass_config_t *cfg,
                         const int16_t *d_senscr, int32_t n_frames, const int32_t *utt_off,
                         int32_t n_utts, const int32_t *word_off, const char *const *words,
                         int32_t max_seg, int32_t *n_seg, ssw_word_seg_t *seg, void *stream);

/* The first pass in two halves, for pipelines: ssw_first_pass_prepare builds the graphs of a
 * batch of texts on the host (no device call; safe on another host thread while the GPU works
 * on the previous batch), ssw_first_pass_run / ssw_forced_align_planned search them.  A plan
 * can be run any number of times (e.g. the same texts against several recordings) and is
 * freed by the caller. */
typedef struct ssw_first_pass_plan_s ssw_first_pass_plan_t;
ssw_first_pass_plan_t *ssw_first_pass_prepare(const ssw_model_t *m, const ssw_dict_t *d,
                                              const ssw_first_pass_config_t *cfg, int32_t n_utts,
                                              const int32_t *word_off, const char *const *words);
void ssw_first_pass_plan_free(ssw_first_pass_plan_t *plan);
int ssw_first_pass_run(ssw_model_t *m, const ssw_first_pass_plan_t *plan, const int16_t *d_senscr,
                       int32_t n_frames, const int32_t *utt_off, int32_t max_seg, int32_t *n_seg,
                       ssw_word_seg_t *seg, void *stream);

/* decoder_alignment (src/decoder.c:737-798) for a batch: first pass, then alignment_add_word
 * with the first pass's word windows + alignment_populate, the state alignment constrained to
 * those windows (state_align_search_init / step / finish), alignment_propagate.  The result
 * holds one alignment_t-shaped record per utterance -- word, phone and state entries with
 * (start, duration, score) -- read with the accessors below.  Arguments as
 * ssw_first_pass_batch.  Returns NULL on a call-level error; per-utterance outcomes are in
 * ssw_alignment_set_status: 0 aligned, 1 the first pass did not reach the end of the text,
 * 2 the second pass failed ("Failed to reach final state" / "Alignment failed in frame"),
 * 3 the utterance's text was rejected (decoder_set_align_text's "Unknown word ..."; the message
 * is in ssw_alignment_set_message) -- the other utterances of the batch are aligned all the same. */
typedef struct ssw_alignment_set_s ssw_alignment_set_t;
ssw_alignment_set_t *ssw_forced_align_batch(ssw_model_t *m, const ssw_dict_t *d,
                                            const ssw_first_pass_config_t *cfg,
                                            const int16_t *d_senscr, int32_t n_frames,
                                            const int32_t *utt_off, int32_t n_utts,
                                            const int32_t *word_off, const char *const *words,
                                            void *stream);
/* ssw_forced_align_batch with the graphs prepared beforehand (ssw_first_pass_prepare) */
ssw_alignment_set_t *ssw_forced_align_planned(ssw_model_t *m, const ssw_dict_t *d,
                                              const ssw_first_pass_plan_t *plan,
                                              const int16_t *d_senscr, int32_t n_frames,
                                              const int32_t *utt_off, void *stream);
/* The same from FEATURES: scores the batch's feature rows in HBM (d_feats float32
 * [n_frames][39], e.g. from ssw_feat_batch) with `scorer` into a workspace the model keeps,
 * then ssw_forced_align_batch -- decoder_start_utt .. decoder_alignment for a batch, minus the
 * MFCC front end, in one call. */
ssw_alignment_set_t *ssw_align_text_batch(ssw_model_t *m, const ssw_dict_t *d,
                                          const ssw_first_pass_config_t *cfg, int scorer,
                                          const float *d_feats, int32_t n_frames,
                                          const int32_t *utt_off, int32_t n_utts,
                                          const int32_t *word_off, const char *const *words,
                                          void *stream);
/* The first pass -- and decoder_alignment -- in the reference's DEFAULT configuration
 * (compallsen = no) for a batch (NEW, round 6).  There acmod scores, frame by frame, only the
 * senones of the HMMs the search holds active -- fsg_search_sen_active rebuilds the set before
 * every frame (src/fsg_search.c:310-325, :677-680), acmod_flags2list turns it into the uint8
 * delta list with its bridge entries (src/acmod.c:947-999), the PTM scorer scans the codebooks of
 * the listed senones, normalises over them and subtracts the best listed score
 * (src/ptm_mgau.c:264-403) -- so frame t can only be scored once frame t - 1 has been searched.
 * These calls break that dependency by speculation and proof: a trajectory of active sets is
 * assumed (the search over the compallsen = yes scores gives the first one), the whole batch is
 * scored with exactly those per-frame sets, the search runs again on those scores, and an
 * utterance whose search took the assumed sets frame for frame has, by induction over the
 * frames, read nothing but the scores the reference would have computed: it IS the reference's
 * search.  Any other utterance is right up to and including the set of its first differing
 * frame and goes round again from the sets it just took; the proven prefix grows every round.
 * Word segmentations and scores are those of the reference's default configuration (they differ
 * from the compallsen = yes ones: another normaliser per frame and, through its clamp, slightly
 * other pruning).  Texts of any length (beyond 1,024 phone-tree HMMs the long-text search
 * kernels export their sets, as ssw_first_pass_batch uses them); 3-state HMMs, <= 64 codebooks,
 * ds = 1; history reset per utterance.
 *   d_feats      feature rows [n_frames][veclen] in HBM (ssw_feat_batch)
 *   seed_active  NULL, or host uint32 [n_utts][(n_sen + 31) / 32]: acmod's flags after each
 *                utterance's last frame -- what ssw_align_batch_active takes as its seed
 *   d_senscr     NULL, or device int16 [n_frames][n_sen]: the score rows exactly as acmod's
 *                buffer would hold them frame by frame (costs one more scoring pass)
 *   rounds       NULL, or host int32 [n_utts]: searches over default-configuration scores the
 *                utterance took (1: its first assumption was proven)
 * other arguments and results as ssw_first_pass_batch.  Synchronous on `stream`.
 * Device memory the model keeps for these calls (grow-only, freed with the model): two tables
 * of exported sets (8 bytes per 64 phone-tree HMMs and frame), a bitmap of listed senones per
 * frame (n_sen / 8 bytes), the longest utterance's score rows once more, and -- when d_senscr
 * is NULL -- the batch's score rows [n_frames][n_sen] (shared with ssw_align_text_batch). */
int ssw_first_pass_batch_active(ssw_model_t *m, const ssw_dict_t *d,
                                const ssw_first_pass_config_t *cfg, int scorer,
                                const float *d_feats, int32_t n_frames, const int32_t *utt_off,
                                int32_t n_utts, const int32_t *word_off, const char *const *words,
                                int32_t max_seg, int32_t *n_seg, ssw_word_seg_t *seg,
                                uint32_t *seed_active, int16_t *d_senscr, int32_t *rounds,
                                void *stream);
/* running totals since ssw_model_load: [0] utterances searched that way, [1] their rounds
 * summed, [2] rounds of the last call, [3] utterances that needed more than one */
int ssw_first_pass_active_stats(ssw_model_t *m, int64_t stats[4]);
/* ssw_align_text_batch in the default configuration: that first pass, alignment_populate with
 * its word windows, the second pass as ssw_align_batch_active_ex runs it (acmod's set starts
 * from what the first pass's last frame left and only grows), alignment_propagate.
 * cfg->two_pass_history (PTM scorer): the second pass of every utterance starts, as
 * decoder_alignment's does after acmod_rewind, from the top-N history its own first pass left
 * in slot 1 -- every codebook's list after the last odd-numbered frame, the codebooks that
 * frame did not scan included; 0 (default): from the reset history.  The two differ only
 * where truncated densities tie; on the reference's recordings they do not (tests). */
ssw_alignment_set_t *ssw_align_text_batch_active(ssw_model_t *m, const ssw_dict_t *d,
                                                 const ssw_first_pass_config_t *cfg, int scorer,
                                                 const float *d_feats, int32_t n_frames,
                                                 const int32_t *utt_off, int32_t n_utts,
                                                 const int32_t *word_off, const char *const *words,
                                                 void *stream);
/* Debug / parity view: the top-N orders the last ssw_align_text_batch_active call with
 * two_pass_history handed from every utterance's first pass to its second: host uint32
 * [n_utts][n_cb * n_feat], 4 codewords packed best first. */
int ssw_first_pass_active_carry(ssw_model_t *m, int32_t n_utts, uint32_t *rows);
int32_t ssw_alignment_set_status(const ssw_alignment_set_t *a, int32_t utt);
const char *ssw_alignment_set_message(const ssw_alignment_set_t *a, int32_t utt);
/* each returns the number of entries and points the outputs (any may be NULL) at arrays owned
 * by the set: words -> dictionary ids; phones -> CI phone id and parent word index;
 * states -> senone id and parent phone = index / 3 */
int32_t ssw_alignment_set_words(const ssw_alignment_set_t *a, int32_t utt, const int32_t **wid,
                                const ssw_align_entry_t **al);
int32_t ssw_alignment_set_phones(const ssw_alignment_set_t *a, int32_t utt, const int32_t **cipid,
                                 const int32_t **parent, const ssw_align_entry_t **al);
int32_t ssw_alignment_set_states(const ssw_alignment_set_t *a, int32_t utt,
                                 const uint16_t **senid, const ssw_align_entry_t **al);
/* decoder_result_json(d, utt_start, align_level) for utterance `utt` of the set (align_level 1:
 * words and phones, 2: states too): "t" of the top level is the hypothesis string as
 * decoder_hyp builds it (base words, fillers left out), "d" the reference's decoder_n_frames
 * (the utterance's frames + 1) / frate, word texts are dictionary strings ("de(2)").  The
 * posterior of the first pass is not computed: top-level "p" is 1.000.  snprintf-style return;
 * -1 when the utterance has no alignment. */
int32_t ssw_alignment_set_json(const ssw_alignment_set_t *a, int32_t utt, double utt_start,
                               int32_t frate, int32_t align_level, char *out, int32_t out_len);
void ssw_alignment_set_free(ssw_alignment_set_t *a);

/* The graph ssw_first_pass_batch searches for one text, node by node (host only, works without
 * a device): the phone-tree HMMs of fsg_lextree_init with their entry penalty, predecessor,
 * FSG state and context sets.  flags: 1 word-initial, 2 word-final, 4 exit valid for every
 * right context, 8 / 16 / 32 member / first / last of a group of word-final HMMs that can never
 * differ (alternates pronounced alike).  beams[3] receives beam, pbeam, wbeam in score units.  Returns the node
 * count (which may exceed max_nodes: only max_nodes are written) or -1. */
typedef struct ssw_fp_node_s {
    uint16_t senid[3];
    int16_t tmat;
    int32_t pen, parent;
    uint32_t flags;
    int32_t ci_ext, state, to_state, wid;
    uint64_t ctxt;
} ssw_fp_node_t;
int32_t ssw_first_pass_graph(const ssw_model_t *m, const ssw_dict_t *d,
                             const ssw_first_pass_config_t *cfg, int32_t n_words,
                             const char *const *words, int32_t max_nodes, ssw_fp_node_t *nodes,
                             int32_t *beams);

/* decoder_result_json(d, utt_start, align_level >= 1) (src/decoder.c:1339-1593): the one-line
 * JSON the reference prints for an alignment, {"b","d","p","t","w":[...]} with b/d in seconds
 * (frames / frate) and p = logmath_exp(score).  hyp / hyp_logprob are the first pass's text
 * and posterior (decoder_hyp / decoder_prob: not part of this path, passed through).  Phones
 * must be grouped by ascending parent word, as ssw_alignment_populate writes them; pass
 * state_senid / state_al (3 per phone) for align_level 2, NULL for level 1.  Writes at most
 * out_len bytes (NUL-terminated) and returns the full length, snprintf-style; -1 on bad
 * arguments. */
int32_t ssw_alignment_json(const ssw_model_t *m, const char *hyp, int32_t hyp_logprob,
                           double utt_start, int32_t frate, int32_t n_frames, int32_t n_words,
                           const char *const *words, const ssw_align_entry_t *word_al,
                           int32_t n_phones, const int32_t *cipid, const int32_t *parent,
                           const ssw_align_entry_t *phone_al, const uint16_t *state_senid,
                           const ssw_align_entry_t *state_al, char *out, int32_t out_len);

/* ------------------------------------------------------------------------------------ */
/* Dynamic features on the device: feat_s2mfc2feat_live for whole    */
/* utterances with feat = 1s_c_d_dd, cmn = batch ("current"), no varnorm / agc / lda       */
/* (src/feat.c:977-1008, 589-632; src/cmn.c:168-200).  d_cep [n_frames][ncep] MFCC rows     */
/* -> d_out [n_frames][3*ncep] = cep | delta | delta-delta, i.e. the 0-12/13-25/26-38       */
/* stream split the scorers read.  Bit-exact float32, synchronous on `stream`.            */
/* ------------------------------------------------------------------------------------ */
int ssw_feat_batch(ssw_model_t *m, const float *d_cep, int32_t n_frames,
                   const int32_t *utt_off, int32_t n_utts, int32_t ncep, float *d_out,
                   void *stream);

/* ------------------------------------------------------------------------------------ */
/* Multi-GPU (NEW: the reference is single-process).  The path shards by utterance -- one   */
/* process per GPU, the model replicated, no exchange while scoring and aligning -- and the  */
/* only collective is ONE gather of the final alignment entries over RCCL (xGMI inside a      */
/* node).  What is gathered is the state level of alignment_t (include/soundswallower/         */
/* alignment.h:60-109) as ssw_align_batch returns it; alignment_propagate then runs wherever   */
/* the word / phone levels are wanted.  RCCL is bound at run time, so single-GPU hosts do not  */
/* need it installed.                                                                          */
/* ------------------------------------------------------------------------------------ */
typedef struct ssw_comm_s ssw_comm_t;
#define SSW_COMM_ID_BYTES 128
/* ncclGetUniqueId: called by ONE rank, which hands the 128 bytes to the others by whatever
 * channel the host has (MPI, a file, torch.distributed ...) */
int ssw_comm_unique_id(char id[SSW_COMM_ID_BYTES]);
/* ncclCommInitRank on `device`; collective over the n_ranks callers */
ssw_comm_t *ssw_comm_init(const char id[SSW_COMM_ID_BYTES], int32_t n_ranks, int32_t rank,
                          int32_t device);
/* wrap a communicator the host already has (an ncclComm_t); not destroyed by ssw_comm_free */
ssw_comm_t *ssw_comm_from_nccl(void *nccl_comm, int32_t n_ranks, int32_t rank, int32_t device);
/* A host with a transport of its own (MPI_Allgather, a test double): `fn` must deliver, on
 * every rank, the n_int32_per_rank words each rank passes in `send` to recv[r * n_int32_per_rank
 * ..] for r = 0 .. n_ranks - 1 (host memory on both sides) and return 0.  ssw_gather_alignments
 * then pads, exchanges and unpacks through it without touching RCCL or HIP. */
typedef int (*ssw_all_gather_fn)(void *ctx, const void *send, void *recv, size_t n_int32_per_rank);
ssw_comm_t *ssw_comm_from_transport(ssw_all_gather_fn fn, void *ctx, int32_t n_ranks, int32_t rank);
void ssw_comm_free(ssw_comm_t *c);
/* Collective.  local: this rank's n_local entries (the state entries of its utterances, in its
 * shard's order); counts [n_ranks]: entries of every rank -- known to all of them, being a
 * function of the transcripts and the shard plan; out [sum(counts)]: all entries in rank order,
 * on every rank (host memory).  One padded ncclAllGather; synchronous on `stream`. */
int ssw_gather_alignments(ssw_comm_t *c, const ssw_align_entry_t *local, int32_t n_local,
                          const int32_t *counts, ssw_align_entry_t *out, void *stream);
/* ncclCommCount of the communicator (the ranks RCCL itself says it spans); n_ranks of a host
 * transport.  -1 on error. */
int32_t ssw_comm_count(ssw_comm_t *c);

/* device-memory helpers so a C caller needs no HIP headers */
void *ssw_device_malloc(size_t nbytes);
void ssw_device_free(void *d_ptr);
/* free and total memory of the current device (hipMemGetInfo): a job sizes what it keeps
 * resident by it (soundswallower_amd/jobs.py) */
int ssw_device_mem_info(size_t *free_bytes, size_t *total_bytes);
int ssw_memcpy_h2d(void *d_dst, const void *src, size_t nbytes);
int ssw_memcpy_d2h(void *dst, const void *d_src, size_t nbytes);
int ssw_device_synchronize(void);
/* independent (non-blocking) streams for the `stream` arguments above, e.g. to score the next
 * chunk of a job while ssw_align_batch works through the previous one */
void *ssw_stream_create(void);
void ssw_stream_destroy(void *stream);
int ssw_stream_synchronize(void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SSW_AMD_H */
