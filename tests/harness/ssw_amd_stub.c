/* ssw_amd_stub.c -- a stand-in for libssw_amd.so WITHOUT a GPU, for tests/test_integration_shim.py:
 * every C-ABI symbol examples/integration_shim.c references, so that the shim can be LINKED with
 * the real reference library and RUN in the CPU container.  It is test infrastructure, not a CPU
 * fallback of the product: the searches return what the driver hands it (results the reference
 * itself produced a moment earlier), the scorer object forwards every call to a reference scorer
 * the driver lends it -- through ssw_mgaufuncs_t, i.e. relying on the layout claim of
 * include/ssw_amd.h in the opposite direction -- and everything records how it was driven. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ssw_amd.h"

/* ---- what the driver sets / reads -------------------------------------------------- */
ssw_align_entry_t *stub_canned_states; /* returned by ..._finish / ..._states */
int32_t stub_canned_n;
void *stub_backend_mgau;               /* a reference mgau_t * serving frame_eval */
int stub_init_calls, stub_start_calls, stub_finish_calls, stub_free_calls, stub_mgau_free_calls;
int stub_n_steps, stub_step_out_of_order;
int32_t stub_init_phones;
int32_t *stub_init_ssid, *stub_init_tmatid, *stub_init_start, *stub_init_dur;
double stub_feat_sum;                  /* sum of every float of every row handed to step */
float stub_rows[512][39];              /* the rows themselves (first 512 frames) */
int stub_frame_eval_calls, stub_frame_idx_mismatch;

struct ssw_model_s { int dummy; };
static struct ssw_model_s the_model;
static const char *last_err = "";

void ssw_config_defaults(ssw_config_t *cfg) { memset(cfg, 0, sizeof(*cfg)); cfg->device = -1; }
const char *ssw_last_error(void) { return last_err; }
ssw_model_t *ssw_model_load(const char *mdef, const char *means, const char *variances,
                            const char *sendump, const char *mixw, const char *tmat,
                            const ssw_config_t *cfg)
{
    (void)mdef; (void)means; (void)variances; (void)sendump; (void)mixw; (void)tmat; (void)cfg;
    return &the_model;
}

/* ---- scorer object: every slot forwards to the reference scorer the driver lent ---- */
typedef struct { ssw_mgau_t base; } stub_mgau_t;
static int
stub_frame_eval(ssw_mgau_t *mgau, int16_t *senscr, uint8_t *senone_active, int32_t n_senone_active,
                float **feat, int32_t frame, int32_t compallsen)
{
    ssw_mgau_t *b = (ssw_mgau_t *)stub_backend_mgau; /* really a reference mgau_t */
    ++stub_frame_eval_calls;
    b->frame_idx = mgau->frame_idx; /* what acmod wrote into OUR second member (src/acmod.c:367,748,760) */
    return b->vt->frame_eval(b, senscr, senone_active, n_senone_active, feat, frame, compallsen);
}
static int stub_transform(ssw_mgau_t *mgau, void *mllr) { (void)mgau; (void)mllr; return -1; }
static void stub_mgau_free(ssw_mgau_t *mgau) { ++stub_mgau_free_calls; free(mgau); }
static ssw_mgaufuncs_t stub_vt = { "ptm", stub_frame_eval, stub_transform, stub_mgau_free };
ssw_mgau_t *
ssw_ptm_mgau_init(ssw_model_t *m)
{
    stub_mgau_t *g = calloc(1, sizeof(*g));
    (void)m;
    g->base.vt = &stub_vt;
    return &g->base;
}
int ssw_mgau_prescore(ssw_mgau_t *mgau, const float *feats, int32_t n_frames)
{ (void)mgau; (void)feats; (void)n_frames; return 0; }

/* ---- search object ----------------------------------------------------------------- */
struct ssw_state_align_search_s { int started, next_frame; };
ssw_state_align_search_t *
ssw_state_align_search_init(ssw_model_t *m, ssw_mgau_t *mgau, int32_t n_phones, const int32_t *ssid,
                            const int32_t *tmatid, const int32_t *start, const int32_t *duration)
{
    size_t nb = sizeof(int32_t) * (size_t)n_phones;
    (void)m; (void)mgau;
    ++stub_init_calls;
    stub_init_phones = n_phones;
    free(stub_init_ssid); free(stub_init_tmatid); free(stub_init_start); free(stub_init_dur);
    stub_init_ssid = malloc(nb); stub_init_tmatid = malloc(nb);
    stub_init_start = malloc(nb); stub_init_dur = malloc(nb);
    memcpy(stub_init_ssid, ssid, nb); memcpy(stub_init_tmatid, tmatid, nb);
    memcpy(stub_init_start, start, nb); memcpy(stub_init_dur, duration, nb);
    return calloc(1, sizeof(ssw_state_align_search_t));
}
int ssw_state_align_search_start(ssw_state_align_search_t *s)
{ ++stub_start_calls; s->started = 1; s->next_frame = 0; stub_n_steps = 0; stub_feat_sum = 0.0; return 0; }
int
ssw_state_align_search_step(ssw_state_align_search_t *s, const float *feat, int frame_idx)
{
    int j;
    if (!s->started || frame_idx != s->next_frame)
        ++stub_step_out_of_order;
    s->next_frame = frame_idx + 1;
    for (j = 0; j < 39; ++j)
        stub_feat_sum += feat[j];
    if (frame_idx >= 0 && frame_idx < 512)
        memcpy(stub_rows[frame_idx], feat, sizeof(stub_rows[0]));
    ++stub_n_steps;
    return 0;
}
int ssw_state_align_search_finish(ssw_state_align_search_t *s)
{ (void)s; ++stub_finish_calls; return stub_canned_states ? 0 : -1; }
const ssw_align_entry_t *
ssw_state_align_search_states(const ssw_state_align_search_t *s, int32_t *n)
{ (void)s; *n = stub_canned_n; return stub_canned_states; }
void ssw_state_align_search_free(ssw_state_align_search_t *s) { ++stub_free_calls; free(s); }
void stub_release(void) /* the copies ..._init keeps for the driver's checks */
{
    free(stub_init_ssid); free(stub_init_tmatid); free(stub_init_start); free(stub_init_dur);
    stub_init_ssid = stub_init_tmatid = stub_init_start = stub_init_dur = NULL;
}

/* ---- referenced by the parts of the shim this test does not drive (they need a GPU) --- */
void *ssw_device_malloc(size_t n) { (void)n; last_err = "stub: no device"; return NULL; }
void ssw_device_free(void *p) { (void)p; }
int ssw_memcpy_h2d(void *d, const void *s, size_t n) { (void)d; (void)s; (void)n; return -1; }
int ssw_device_synchronize(void) { return -1; }
int ssw_score_batch(ssw_model_t *m, int scorer, const float *d_feats, int32_t n_frames,
                    const int32_t *utt_off, int32_t n_utts, int16_t *d_out, void *stream)
{ (void)m; (void)scorer; (void)d_feats; (void)n_frames; (void)utt_off; (void)n_utts; (void)d_out; (void)stream; return -1; }
ssw_alignment_set_t *
ssw_forced_align_batch(ssw_model_t *m, const ssw_dict_t *d, const ssw_first_pass_config_t *cfg,
                       const int16_t *d_senscr, int32_t n_frames, const int32_t *utt_off,
                       int32_t n_utts, const int32_t *word_off, const char *const *words, void *stream)
{ (void)m; (void)d; (void)cfg; (void)d_senscr; (void)n_frames; (void)utt_off; (void)n_utts; (void)word_off; (void)words; (void)stream; return NULL; }
ssw_alignment_set_t *
ssw_align_text_batch_active(ssw_model_t *m, const ssw_dict_t *d, const ssw_first_pass_config_t *cfg,
                            int scorer, const float *d_feats, int32_t n_frames, const int32_t *utt_off,
                            int32_t n_utts, const int32_t *word_off, const char *const *words, void *stream)
{ (void)m; (void)d; (void)cfg; (void)scorer; (void)d_feats; (void)n_frames; (void)utt_off; (void)n_utts; (void)word_off; (void)words; (void)stream; return NULL; }
int32_t ssw_alignment_set_status(const ssw_alignment_set_t *a, int32_t utt) { (void)a; (void)utt; return 1; }
int32_t ssw_alignment_set_words(const ssw_alignment_set_t *a, int32_t utt, const int32_t **wid,
                                const ssw_align_entry_t **al) { (void)a; (void)utt; (void)wid; (void)al; return 0; }
int32_t ssw_alignment_set_states(const ssw_alignment_set_t *a, int32_t utt, const uint16_t **senid,
                                 const ssw_align_entry_t **al) { (void)a; (void)utt; (void)senid; (void)al; return 0; }
void ssw_alignment_set_free(ssw_alignment_set_t *a) { (void)a; }
const char *ssw_dict_word(const ssw_dict_t *d, int32_t wid) { (void)d; (void)wid; return NULL; }
