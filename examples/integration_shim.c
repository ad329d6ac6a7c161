/* integration_shim.c -- the binding INTEGRATION.md describes, as a translation unit a
 * SoundSwallower maintainer could add to the reference's src/.  It is compiled (syntax and layout
 * checks only, never linked into this repository's product) by tests/test_integration_shim.py
 * against the reference's public headers where /root/reference is available:
 *
 *     gcc -fsyntax-only -I<reference>/include -Iinclude examples/integration_shim.c
 *
 * The static assertions are the drop-in claims of include/ssw_amd.h: ssw_mgau_t can be stored in
 * acmod->mgau because it starts with the same two members as mgau_t, and its vtable has
 * mgaufuncs_t's slots in mgaufuncs_t's order (include/soundswallower/acmod.h:93-111). */
#include <stddef.h>
#include <stdio.h>

#include <soundswallower/acmod.h>
#include <soundswallower/alignment.h>
#include <soundswallower/ckd_alloc.h>
#include <soundswallower/configuration.h>
#include <soundswallower/decoder.h>
#include <soundswallower/err.h>
#include <soundswallower/search_module.h>
#include <soundswallower/state_align_search.h>

#include "ssw_amd.h"

_Static_assert(offsetof(ssw_mgau_t, vt) == offsetof(mgau_t, vt), "vtable pointer first");
_Static_assert(offsetof(ssw_mgau_t, frame_idx) == offsetof(mgau_t, frame_idx), "frame_idx second");
_Static_assert(offsetof(ssw_mgaufuncs_t, name) == offsetof(mgaufuncs_t, name), "slot 0: name");
_Static_assert(offsetof(ssw_mgaufuncs_t, frame_eval) == offsetof(mgaufuncs_t, frame_eval),
               "slot 1: frame_eval");
_Static_assert(offsetof(ssw_mgaufuncs_t, transform) == offsetof(mgaufuncs_t, transform),
               "slot 2: transform");
_Static_assert(offsetof(ssw_mgaufuncs_t, free) == offsetof(mgaufuncs_t, free), "slot 3: free");
_Static_assert(sizeof(ssw_mgaufuncs_t) == sizeof(mgaufuncs_t), "no extra slots");

static ssw_model_t *gpu_model;

/* swap the CPU scorer for the GPU one after decoder_init() */
int
use_gpu_scorer(decoder_t *d, const char *hmmdir)
{
    ssw_config_t cfg;
    char mdef[512], mean[512], var[512], sendump[512], tmat[512];
    ssw_mgau_t *g;

    ssw_config_defaults(&cfg);
    cfg.logbase = config_float(d->config, "logbase");
    cfg.varfloor = config_float(d->config, "varfloor");
    cfg.mixwfloor = config_float(d->config, "mixwfloor");
    cfg.tmatfloor = config_float(d->config, "tmatfloor");
    cfg.topn = config_int(d->config, "topn");
    cfg.ds = config_int(d->config, "ds");
    cfg.aw = config_int(d->config, "aw");
    snprintf(mdef, sizeof mdef, "%s/mdef", hmmdir);
    snprintf(mean, sizeof mean, "%s/means", hmmdir);
    snprintf(var, sizeof var, "%s/variances", hmmdir);
    snprintf(sendump, sizeof sendump, "%s/sendump", hmmdir);
    snprintf(tmat, sizeof tmat, "%s/transition_matrices", hmmdir);
    gpu_model = ssw_model_load(mdef, mean, var, sendump, NULL, tmat, &cfg);
    if (gpu_model == NULL) {
        E_ERROR("GPU model: %s\n", ssw_last_error());
        return -1;
    }
    g = ssw_ptm_mgau_init(gpu_model); /* stands in for ptm_mgau_init(acmod) */
    if (g == NULL)
        return -1;
    ps_mgau_free(d->acmod->mgau);
    d->acmod->mgau = (mgau_t *)g; /* same leading layout, same vtable slots */
    d->acmod->compallsen = TRUE;  /* the GPU scorer computes all senones */
    return 0;
}

/* hand the whole utterance over once the features exist */
void
prescore_utterance(acmod_t *acmod)
{
    ssw_mgau_prescore((ssw_mgau_t *)acmod->mgau, acmod->feat_buf[0][0], acmod->n_feat_frame);
}

/* the second pass of decoder_alignment on the GPU */
int
gpu_second_pass(decoder_t *d, alignment_t *al, int n_frames)
{
    int n = alignment_n_phones(al), i = 0, rv = 0;
    int32 *ssid = ckd_calloc(n, sizeof(*ssid)), *tmatid = ckd_calloc(n, sizeof(*tmatid));
    int32 *start = ckd_calloc(n, sizeof(*start)), *dur = ckd_calloc(n, sizeof(*dur));
    alignment_iter_t *it;
    ssw_state_align_search_t *s;
    const ssw_align_entry_t *st;
    int32 ns;
    int f;

    for (it = alignment_phones(al); it; it = alignment_iter_next(it), ++i) {
        alignment_entry_t *e = alignment_iter_get(it);
        ssid[i] = e->id.pid.ssid;
        tmatid[i] = e->id.pid.tmatid;
        start[i] = e->start;
        dur[i] = e->duration;
    }
    s = ssw_state_align_search_init(gpu_model, (ssw_mgau_t *)d->acmod->mgau, n, ssid, tmatid,
                                    start, dur);
    ssw_state_align_search_start(s);
    for (f = 0; f < n_frames; ++f) /* the acmod_score/step loop of src/decoder.c:789-793 */
        ssw_state_align_search_step(s, d->acmod->feat_buf[f][0], f);
    if (ssw_state_align_search_finish(s) < 0) {
        E_ERROR("%s\n", ssw_last_error());
        rv = -1;
    }
    st = ssw_state_align_search_states(s, &ns);
    i = 0;
    for (it = alignment_states(al); rv == 0 && it; it = alignment_iter_next(it), ++i) {
        alignment_entry_t *e = alignment_iter_get(it);
        e->start = st[i].start;
        e->duration = st[i].duration;
        e->score = st[i].score;
    }
    if (rv == 0)
        alignment_propagate(al);
    ssw_state_align_search_free(s);
    ckd_free(ssid);
    ckd_free(tmatid);
    ckd_free(start);
    ckd_free(dur);
    return rv;
}

/* both passes on the GPU for one decoder -- what
 * decoder_set_align_text + the first pass + decoder_alignment (src/decoder.c:686-798) do,
 * starting from the features acmod has buffered.  gpu_dict is ssw_dict_load() of the same
 * dictionary files.  Returns a filled alignment_t (caller frees) or NULL. */
alignment_t *
gpu_text_alignment(decoder_t *d, ssw_dict_t *gpu_dict, const char *const *words, int n_words)
{
    acmod_t *acmod = d->acmod;
    const int n_frames = acmod->n_feat_frame, n_sen = bin_mdef_n_sen(acmod->mdef);
    const int32 utt_off[2] = { 0, n_frames }, word_off[2] = { 0, n_words };
    const size_t feat_bytes = (size_t)n_frames * 39 * sizeof(float);
    float *d_feat = ssw_device_malloc(feat_bytes);
    int16 *d_scr = ssw_device_malloc((size_t)n_frames * n_sen * sizeof(int16));
    ssw_alignment_set_t *set = NULL;
    alignment_t *al = NULL;
    const int32 *wid;
    const ssw_align_entry_t *wal, *sal;
    alignment_iter_t *it;
    int i, n;

    if (d_feat == NULL || d_scr == NULL
        || ssw_memcpy_h2d(d_feat, acmod->feat_buf[0][0], feat_bytes) < 0) {
        E_ERROR("%s\n", ssw_last_error());
        goto done;
    }
    if (!acmod->compallsen) {
        /* the reference's default: acmod scores what the searches hold active (src/acmod.c:905-999).
         * Both passes in that configuration, one call (round 6: the first pass by speculation
         * and proof, include/ssw_amd.h) -- the scores decoder_alignment reports are then the
         * ones this decoder would have computed frame by frame */
        set = ssw_align_text_batch_active(gpu_model, gpu_dict, NULL, SSW_SCORER_PTM, d_feat, n_frames,
                                          utt_off, 1, word_off, words, NULL);
    } else if (ssw_score_batch(gpu_model, SSW_SCORER_PTM, d_feat, n_frames, utt_off, 1, d_scr, NULL) == 0
               && ssw_device_synchronize() == 0) {
        set = ssw_forced_align_batch(gpu_model, gpu_dict, NULL, d_scr, n_frames, utt_off, 1,
                                     word_off, words, NULL);
    }
    if (set == NULL) {
        E_ERROR("%s\n", ssw_last_error());
        goto done;
    }
    if (ssw_alignment_set_status(set, 0) != 0) {
        E_ERROR("Final result does not match the grammar, or the alignment failed\n");
        goto done;
    }
    /* the words the first pass found (fillers and alternates included), as decoder_alignment
     * adds them from the seg_iter (src/decoder.c:758-768) */
    al = alignment_init(d->d2p);
    n = ssw_alignment_set_words(set, 0, &wid, &wal);
    for (i = 0; i < n; ++i)
        alignment_add_word(al, dict_wordid(d->dict, ssw_dict_word(gpu_dict, wid[i])),
                           wal[i].start, wal[i].duration);
    if (alignment_populate(al) < 0) {
        alignment_free(al);
        al = NULL;
        goto done;
    }
    ssw_alignment_set_states(set, 0, NULL, &sal);
    for (i = 0, it = alignment_states(al); it; it = alignment_iter_next(it), ++i) {
        alignment_entry_t *e = alignment_iter_get(it);
        e->start = sal[i].start;
        e->duration = sal[i].duration;
        e->score = sal[i].score;
    }
    alignment_propagate(al);
done:
    ssw_alignment_set_free(set);
    ssw_device_free(d_feat);
    ssw_device_free(d_scr);
    return al;
}

/* ------------------------------------------------------------------------------------ */
/* INTEGRATION.md section 2, vtable level: a search module decoder_alignment() can drive   */
/* unchanged.  Replace its call                                                           */
/*     d->align = state_align_search_init("_state_align", d->config, d->acmod, al);       */
/* (src/decoder.c:776) by gpu_state_align_search_init(...).  The object starts with the    */
/* reference's own state_align_search_t, because decoder_alignment reads ->frame and ->al   */
/* through that type (src/decoder.c:747-750); start/step/finish/free forward to the GPU     */
/* object, which buffers the feature rows and runs scoring + Viterbi + backtrace at finish. */
/* ------------------------------------------------------------------------------------ */
typedef struct gpu_state_align_search_s {
    state_align_search_t sas; /* base, al, n_phones, frame are the fields others read */
    ssw_state_align_search_t *gpu;
} gpu_state_align_search_t;

static int
gpu_sas_start(search_module_t *search)
{
    gpu_state_align_search_t *g = (gpu_state_align_search_t *)search;
    g->sas.frame = 0;
    return ssw_state_align_search_start(g->gpu);
}

static int
gpu_sas_step(search_module_t *search, int frame_idx)
{
    gpu_state_align_search_t *g = (gpu_state_align_search_t *)search;
    mfcc_t **feat = acmod_get_frame(search_module_acmod(search), &frame_idx);
    if (feat == NULL)
        return -1;
    /* feat[0] is the contiguous 39-float row of the frame (src/feat.c:386-395) */
    if (ssw_state_align_search_step(g->gpu, feat[0], frame_idx) < 0)
        return -1;
    g->sas.frame = frame_idx + 1;
    return 0;
}

static int
gpu_sas_finish(search_module_t *search)
{
    gpu_state_align_search_t *g = (gpu_state_align_search_t *)search;
    const ssw_align_entry_t *st;
    alignment_iter_t *it;
    int32 ns, i = 0;

    if (ssw_state_align_search_finish(g->gpu) < 0) {
        E_ERROR("%s\n", ssw_last_error()); /* the two messages of src/state_align_search.c:229-241 */
        return -1;
    }
    st = ssw_state_align_search_states(g->gpu, &ns);
    for (it = alignment_states(g->sas.al); it && i < ns; it = alignment_iter_next(it), ++i) {
        alignment_entry_t *e = alignment_iter_get(it);
        e->start = st[i].start;
        e->duration = st[i].duration;
        e->score = st[i].score;
    }
    alignment_propagate(g->sas.al);
    return 0;
}

static int
gpu_sas_reinit(search_module_t *search, dict_t *dict, dict2pid_t *d2p)
{
    (void)search;
    (void)dict;
    (void)d2p;
    return 0; /* as state_align_search_reinit (src/state_align_search.c:270-276) */
}

static void
gpu_sas_free(search_module_t *search)
{
    gpu_state_align_search_t *g = (gpu_state_align_search_t *)search;
    search_module_base_free(search);
    ssw_state_align_search_free(g->gpu);
    alignment_free(g->sas.al); /* consuming semantics, as the reference */
    ckd_free(g);
}

static searchfuncs_t gpu_sas_funcs = {
    /* start: */ gpu_sas_start,
    /* step: */ gpu_sas_step,
    /* finish: */ gpu_sas_finish,
    /* reinit: */ gpu_sas_reinit,
    /* free: */ gpu_sas_free,
    /* lattice: */ NULL,
    /* hyp: */ NULL,
    /* prob: */ NULL,
    /* seg_iter: */ NULL,
};

search_module_t *
gpu_state_align_search_init(const char *name, config_t *config, acmod_t *acmod, alignment_t *al)
{
    gpu_state_align_search_t *g = ckd_calloc(1, sizeof(*g));
    int n = alignment_n_phones(al), i = 0;
    int32 *ssid = ckd_calloc(n, sizeof(*ssid)), *tmatid = ckd_calloc(n, sizeof(*tmatid));
    int32 *start = ckd_calloc(n, sizeof(*start)), *dur = ckd_calloc(n, sizeof(*dur));
    alignment_iter_t *it;

    search_module_init(search_module_base(g), &gpu_sas_funcs, PS_SEARCH_TYPE_STATE_ALIGN, name,
                       config, acmod, al->d2p->dict, al->d2p);
    g->sas.al = al; /* consuming semantics */
    g->sas.n_phones = n;
    g->sas.n_emit_state = alignment_n_states(al);
    for (it = alignment_phones(al); it && i < n; it = alignment_iter_next(it), ++i) {
        alignment_entry_t *e = alignment_iter_get(it);
        ssid[i] = e->id.pid.ssid;
        tmatid[i] = e->id.pid.tmatid;
        start[i] = e->start;
        dur[i] = e->duration;
    }
    g->gpu = ssw_state_align_search_init(gpu_model, (ssw_mgau_t *)acmod->mgau, n, ssid, tmatid,
                                         start, dur);
    ckd_free(ssid);
    ckd_free(tmatid);
    ckd_free(start);
    ckd_free(dur);
    if (g->gpu == NULL) {
        E_ERROR("%s\n", ssw_last_error());
        search_module_base_free(search_module_base(g));
        ckd_free(g);
        return NULL;
    }
    return search_module_base(g);
}
