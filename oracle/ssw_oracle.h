/*
 * ssw_oracle.h -- CPU restatement ("oracle") of SoundSwallower's acoustic hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under soundswallower_amd/ may include, link or call
 * this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker the HIP path is compared against.
 *
 * Every function restates one reference function in plain scalar C; the reference location
 * is cited as file:line relative to /root/reference.  No reference source is copied.
 *
 * PARITY PIN STATUS: pinned.  The reference cannot be built here under the round's rules (its
 * translation units need the cmake-generated config.h), so the oracle is
 * pinned against outputs of the real library instead: (a) the reference's known-answer test
 * tests/test_log_shifted.c, (b) its golden front-end table tests/_test_fe.res, (c) the values
 * SURVEY.md Appendix C records from the library (log-add tables, model counts, tmat row 0) and
 * (d) end to end: the reference's printed phone alignment of tests/data/goforward.wav -- 18
 * exact (start, duration, score) triples and 6 word scores -- which this oracle reproduces from
 * the PCM samples (tests/test_oracle_e2e_goforward.py).
 */
#ifndef SSW_ORACLE_H
#define SSW_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- logmath (src/logmath.c) ------------------------------------------------------ */
typedef struct orc_logmath_s {
    double base, log_of_base, log10_of_base, inv_log_of_base, inv_log10_of_base;
    int32_t zero;
    int shift;
    int width;          /* bytes per table entry: 1, 2 or 4; 0 if no table */
    uint32_t table_size;
    void *table;
} orc_logmath_t;

orc_logmath_t *orc_logmath_init(double base, int shift, int use_table);
void orc_logmath_free(orc_logmath_t *lm);
int orc_logmath_log(const orc_logmath_t *lm, double p);
int orc_logmath_ln_to_log(const orc_logmath_t *lm, double log_p);
double orc_logmath_exp(const orc_logmath_t *lm, int logb_p);
int orc_logmath_add(const orc_logmath_t *lm, int x, int y);
/* copy the add table out as uint32 per entry; returns table_size */
uint32_t orc_logmath_table(const orc_logmath_t *lm, uint32_t *out, uint32_t max);

/* ---- model ------------------------------------------------------------------------ */
typedef struct orc_config_s {
    double logbase;    /* config "logbase", default 1.0001 */
    double varfloor;   /* "varfloor" 1e-4 */
    double mixwfloor;  /* "mixwfloor" 1e-7 */
    double tmatfloor;  /* "tmatfloor" 1e-4 */
    int32_t topn;      /* "topn" 4 */
    int32_t ds;        /* "ds" 1 */
    int32_t aw;        /* "aw" 1 */
} orc_config_t;

void orc_config_defaults(orc_config_t *cfg);

typedef struct orc_model_s orc_model_t;

/* Any path may be NULL when that part is not needed.  Exactly one of sendump / mixw feeds the
 * PTM scorer (src/ptm_mgau.c:779-790); mixw also feeds the ms scorer (src/ms_senone.c:104). */
orc_model_t *orc_model_load(const char *mdef, const char *means, const char *vars,
                            const char *sendump, const char *mixw, const char *tmat,
                            const orc_config_t *cfg);
void orc_model_free(orc_model_t *m);
const char *orc_last_error(void);

/* dims[]: n_cb, n_feat, n_density, veclen_total, n_sen, n_ci_sen, n_ciphone, n_phone,
 *         n_emit_state, n_tmat, n_sseq, sil, n_floored, n_cd_tree, has_ptm_mixw, has_ms_pdf */
#define ORC_NDIMS 16
void orc_model_dims(const orc_model_t *m, int32_t *dims);
const int32_t *orc_model_veclen(const orc_model_t *m);
/* flat [cb][feat][density][veclen(feat)] exactly as the s3 file lays it out */
const float *orc_model_mean(const orc_model_t *m);
const float *orc_model_var(const orc_model_t *m);   /* after precompute */
const float *orc_model_det(const orc_model_t *m);   /* [cb][feat][density] */
const uint8_t *orc_model_ptm_mixw(const orc_model_t *m); /* [feat][density][n_sen] */
const uint8_t *orc_model_ms_pdf(const orc_model_t *m);   /* [sen][feat][density] */
const uint8_t *orc_model_tp(const orc_model_t *m);       /* [tmat][n_emit][n_emit+1] */
const uint16_t *orc_model_sseq(const orc_model_t *m);    /* [n_sseq][n_emit] */
const int16_t *orc_model_sen2cimap(const orc_model_t *m);
const int32_t *orc_model_phone_ssid(const orc_model_t *m);  /* [n_phone] */
const int32_t *orc_model_phone_tmat(const orc_model_t *m);  /* [n_phone] */
const orc_logmath_t *orc_model_lmath(const orc_model_t *m);     /* shift 0 */
const orc_logmath_t *orc_model_lmath_8b(const orc_model_t *m);  /* shift 10 */

/* ---- PTM scorer (src/ptm_mgau.c) --------------------------------------------------- */
/* Put both history slots back to {cw = m, score = INT32_MIN} (src/ptm_mgau.c:706-713) and
 * frame_idx to 0 (src/acmod.c:367). */
void orc_ptm_reset(orc_model_t *m);
void orc_ptm_set_frame_idx(orc_model_t *m, int frame_idx);
/* ptm_mgau_frame_eval (src/ptm_mgau.c:408-454).  feat = 39 contiguous floats (streams back to
 * back).  senone_active is the uint8 delta list (may be NULL when compallsen). */
int orc_ptm_frame_eval(orc_model_t *m, int16_t *senscr, const uint8_t *senone_active,
                       int32_t n_senone_active, const float *feat, int32_t frame,
                       int32_t compallsen);
/* top-N state of the slot used by `frame`: cw[n_cb][n_feat][topn], score likewise */
void orc_ptm_get_topn(const orc_model_t *m, int frame, int32_t *cw, int32_t *score);
/* Whole utterance, compallsen=yes: reset, then for t in [0,n): frame_eval(t); frame_idx++
 * exactly as acmod_score/acmod_advance drive it (src/acmod.c:822-860, 753-762).
 * out = int16 [n_frames][n_sen]; topn_cw/topn_score (may be NULL) receive the post-norm state of
 * every frame: [n_frames][n_cb][n_feat][topn]. */
int orc_ptm_score_utt(orc_model_t *m, const float *feats, int n_frames, int16_t *out,
                      int32_t *topn_cw, int32_t *topn_score);

/* ---- ms scorer (src/ms_mgau.c, ms_gauden.c, ms_senone.c) --------------------------- */
int orc_ms_frame_eval(orc_model_t *m, int16_t *senscr, const uint8_t *senone_active,
                      int32_t n_senone_active, const float *feat, int32_t frame,
                      int32_t compallsen);
int orc_ms_score_utt(orc_model_t *m, const float *feats, int n_frames, int16_t *out);

/* ---- active list (src/acmod.c:889-999) --------------------------------------------- */
/* bit-vector (uint32 words) -> uint8 delta list, returns n */
int orc_flags2list(const uint32_t *vec, int n_sen, uint8_t *out);

/* ---- HMM + state alignment (src/hmm.c, src/state_align_search.c, src/ps_alignment.c) */
typedef struct orc_align_entry_s {
    int32_t start, duration, score;
} orc_align_entry_t;

/* One forced alignment.  senscr: int16 [n_frames][n_sen] (what acmod_score returns per frame).
 * Per phone: senid[p*n_emit + j], tmatid[p], sf[p], ef[p] (state_align_search.c:456-471 gives
 * sf = start or 0, ef = start+duration or INT_MAX).  state_io: on entry the values
 * alignment_populate leaves in each state entry (src/ps_alignment.c:237-239), on return the
 * backtraced (start,duration,score) (state_align_search.c:215-268).  phone_out [n_phones]:
 * alignment_propagate's phone level (src/ps_alignment.c:316-334).  tp may override the model's
 * transition table (NULL = model).  Returns 0, or -1 with the reference's failure semantics
 * ("Failed to reach final state", "Alignment failed in frame"). */
int orc_state_align(const orc_model_t *m, const uint8_t *tp_override, const int16_t *senscr,
                    int n_sen, int n_frames, int n_phones, int n_emit,
                    const uint16_t *senid, const int16_t *tmatid, const int32_t *sf,
                    const int32_t *ef, orc_align_entry_t *state_io,
                    orc_align_entry_t *phone_out, int32_t *best_score_trace);

/* hmm_vit_eval (src/hmm.c:741-759) over HMMs held in flat arrays; see oracle/fsg_oracle.py */
void orc_hmm_vit_eval_many(const orc_model_t *m, const int16_t *senscr, int n, const int32_t *idx,
                           const uint16_t *senid, const int16_t *tmat, int32_t *score,
                           int32_t *hist, int32_t *out_score, int32_t *out_hist, int32_t *best);

/* test helper: the reference's density values next to the GPU scan's quadratic-form keys */
void orc_scan_replay(const float *rec, const float *recq, int n_density, int veclen,
                     const float *x, int n, int x_stride, float *ref_out, float *key_out);

/* One hmm_vit_eval step on a bare HMM (src/hmm.c:741-759) for unit tests.
 * score/history have n_emit entries; out[0]=out_score, out[1]=out_history. Returns bestscore. */
int32_t orc_hmm_vit_eval(int n_emit, const uint8_t *tp /* [n_emit][n_emit+1] */,
                         const int16_t *senscr, const uint16_t *senid, int32_t *score,
                         int32_t *history, int32_t *out);

/* ---- triphone lookup (src/bin_mdef.c:543-720); word positions: 0 internal, 1 begin, 2 end,
 * 3 single ------------------------------------------------------------------------------- */
int orc_mdef_ciphone_id(const orc_model_t *m, const char *name);
int orc_mdef_phone_id_nearest(const orc_model_t *m, int b, int l, int r, int pos);

/* ---- front end + dynamic features (oracle/ssw_oracle_fe.c): only to reproduce the
 * reference's recorded alignment of tests/data/goforward.wav end to end ---------------------- */
int orc_fe_mfcc(const int16_t *pcm, size_t n_samps, int nfilt, double lowerf, double upperf,
                int lifter, int remove_noise_flag, int legacy_transform, float *cep,
                int max_frames);
int orc_feat_1s_c_d_dd(float *cep, int n, float *out);
int orc_feat_1s_c_d_dd_ex(float *cep, int n, float *out, int cmn);

#ifdef __cplusplus
}
#endif
#endif
