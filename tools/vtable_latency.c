/* Latency of the drop-in scorer call from C, the way acmod_score makes it: one
 * vt->frame_eval per frame (tools/bench_vtable.py compiles and runs this on the GPU box:
 *   gcc -O2 -I include tools/vtable_latency.c -L soundswallower_amd -lssw_amd -lm).
 * argv: model directory, frames.  compallsen = yes, then compallsen = no with an active list of
 * about 750 senones per frame (250 phone-tree HMMs), given as acmod_flags2list would (uint8
 * deltas). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ssw_amd.h"

static double
now_us(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

int
main(int argc, char **argv)
{
    char p[5][512];
    const char *dir = argc > 1 ? argv[1] : ".";
    const int n_frames = argc > 2 ? atoi(argv[2]) : 1000;
    const char *names[5] = { "mdef", "means", "variances", "sendump", "transition_matrices" };
    ssw_config_t cfg;
    ssw_model_info_t info;
    int i, f, t;
    for (i = 0; i < 5; ++i)
        snprintf(p[i], sizeof(p[i]), "%s/%s", dir, names[i]);
    ssw_config_defaults(&cfg);
    ssw_model_t *m = ssw_model_load(p[0], p[1], p[2], p[3], NULL, p[4], &cfg);
    if (m == NULL) {
        fprintf(stderr, "ssw_model_load: %s\n", ssw_last_error());
        return 1;
    }
    ssw_model_info(m, &info);
    ssw_mgau_t *g = ssw_ptm_mgau_init(m);
    if (g == NULL) {
        fprintf(stderr, "ssw_ptm_mgau_init: %s\n", ssw_last_error());
        return 1;
    }
    const int dim = info.veclen_total, n_sen = info.n_sen, n_feat = info.n_feat;
    float *feats = (float *)malloc(sizeof(float) * (size_t)n_frames * dim);
    int16_t *scr = (int16_t *)malloc(sizeof(int16_t) * (size_t)n_sen);
    uint8_t *act = (uint8_t *)malloc((size_t)n_sen);
    unsigned s = 12345u;
    for (i = 0; i < n_frames * dim; ++i) {
        s = s * 1664525u + 1013904223u;
        feats[i] = ((float)(s >> 8) / 16777216.0f - 0.5f) * 4.0f;
    }
    /* about 750 active senones, spread: deltas 6 or 7 */
    int n_act = 0, last = 0;
    for (i = 3; i < n_sen; i += 6 + (i % 3 == 0)) {
        act[n_act++] = (uint8_t)(i - last);
        last = i;
    }
    double best[2] = { 1e30, 1e30 }, again = 1e30;
    long long sum = 0;
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 2; ++mode) {
            ssw_mgau_reset_hist(g);
            g->frame_idx = 0;
            const double t0 = now_us();
            for (t = 0; t < n_frames; ++t) {
                float *streams[8];
                for (f = 0; f < n_feat; ++f)
                    streams[f] = feats + (size_t)t * dim + (size_t)f * (dim / n_feat);
                if (g->vt->frame_eval(g, scr, mode ? act : NULL, mode ? n_act : 0, streams, t,
                                      mode ? 0 : 1) < 0) {
                    fprintf(stderr, "frame_eval: %s\n", ssw_last_error());
                    return 1;
                }
                g->frame_idx = t + 1; /* acmod advances it (src/acmod.c:748) */
                sum += scr[t % n_sen];
            }
            const double us = (now_us() - t0) / n_frames;
            best[mode] = us < best[mode] ? us : best[mode];
        }
    { /* the same frame asked for again (frame < frame_idx): the senone part alone */
        float *streams[8];
        for (f = 0; f < n_feat; ++f)
            streams[f] = feats + (size_t)f * (dim / n_feat);
        for (int rep = 0; rep < 3; ++rep) {
            const double t0 = now_us();
            for (t = 0; t < n_frames; ++t)
                g->vt->frame_eval(g, scr, NULL, 0, streams, g->frame_idx - 1, 1);
            const double us = (now_us() - t0) / n_frames;
            again = us < again ? us : again;
        }
    }
    printf("{\"frames\": %d, \"active_senones\": %d, \"frame_eval_us_compallsen_yes\": %.2f, "
           "\"frame_eval_us_compallsen_no\": %.2f, \"same_frame_again_us\": %.2f, "
           "\"checksum\": %lld}\n",
           n_frames, n_act, best[0], best[1], again, sum);
    g->vt->free(g);
    ssw_model_free(m);
    return 0;
}
