/* shim_driver.c -- runs examples/integration_shim.c inside the REAL reference decoder (built by
 * tests/test_integration_shim.py from /root/reference in a scratch directory) with
 * tests/harness/ssw_amd_stub.c in place of libssw_amd.so.  Linked with
 * -Wl,--wrap=state_align_search_init, so that decoder_alignment's own call (src/decoder.c:776)
 * reaches gpu_state_align_search_init when the driver says so -- the one-line change
 * INTEGRATION.md section 2 asks a maintainer to make.
 *
 *   run 1  stock reference, compallsen=yes: the alignment of goforward at all three levels
 *   run 2  the search module of the shim around the stub search (which returns run 1's state
 *          entries): ->al / ->frame as decoder_alignment reads them, frames stepped 0 .. n-1
 *          once each with the rows acmod holds, the entries copied and propagated, the
 *          "reuse" branch, the consuming free
 *   run 3  use_gpu_scorer(): acmod->mgau replaced by an ssw_mgau_t whose slots forward to a
 *          second decoder's scorer; the reference calls name / frame_eval / free through its
 *          own mgaufuncs_t view and writes frame_idx through its own mgau_t view
 * Prints "SHIM-DRIVER OK" and exits 0, or says what differed. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <soundswallower/acmod.h>
#include <soundswallower/alignment.h>
#include <soundswallower/configuration.h>
#include <soundswallower/decoder.h>
#include <soundswallower/search_module.h>
#include <soundswallower/state_align_search.h>

#include "ssw_amd.h"

/* the shim */
int use_gpu_scorer(decoder_t *d, const char *hmmdir);
search_module_t *gpu_state_align_search_init(const char *name, config_t *config, acmod_t *acmod,
                                             alignment_t *al);
/* the stub */
extern ssw_align_entry_t *stub_canned_states;
extern int32_t stub_canned_n, stub_init_phones;
extern void *stub_backend_mgau;
extern int stub_init_calls, stub_start_calls, stub_finish_calls, stub_free_calls, stub_mgau_free_calls;
extern int stub_n_steps, stub_step_out_of_order, stub_frame_eval_calls;
extern int32_t *stub_init_ssid, *stub_init_tmatid, *stub_init_start, *stub_init_dur;
extern float stub_rows[512][39];
void stub_release(void);

static int use_adapter;
search_module_t *__real_state_align_search_init(const char *, config_t *, acmod_t *, alignment_t *);
search_module_t *
__wrap_state_align_search_init(const char *name, config_t *config, acmod_t *acmod, alignment_t *al)
{
    return use_adapter ? gpu_state_align_search_init(name, config, acmod, al)
                       : __real_state_align_search_init(name, config, acmod, al);
}

#define CHECK(c, ...) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); \
    fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } } while (0)

typedef struct { int n[3]; ssw_align_entry_t *e[3]; } levels_t;

static void
grab(alignment_t *al, levels_t *L) /* (L zero-initialised or filled by an earlier grab) */
{
    alignment_iter_t *(*start[3])(alignment_t *) = { alignment_words, alignment_phones, alignment_states };
    int n[3] = { alignment_n_words(al), alignment_n_phones(al), alignment_n_states(al) }, k, i;
    for (k = 0; k < 3; ++k) {
        alignment_iter_t *it;
        L->n[k] = n[k];
        free(L->e[k]);
        L->e[k] = calloc((size_t)n[k] + 1, sizeof(ssw_align_entry_t));
        for (i = 0, it = start[k](al); it; it = alignment_iter_next(it), ++i) {
            alignment_entry_t *e = alignment_iter_get(it);
            L->e[k][i].start = e->start;
            L->e[k][i].duration = e->duration;
            L->e[k][i].score = e->score;
        }
        CHECK(i == n[k], "level %d: %d entries, expected %d", k, i, n[k]);
    }
}

static void
same(const levels_t *a, const levels_t *b, const char *what)
{
    int k;
    for (k = 0; k < 3; ++k) {
        CHECK(a->n[k] == b->n[k], "%s: level %d has %d entries, expected %d", what, k, b->n[k], a->n[k]);
        CHECK(memcmp(a->e[k], b->e[k], sizeof(ssw_align_entry_t) * (size_t)a->n[k]) == 0,
              "%s: level %d entries differ", what, k);
    }
}

static decoder_t *
make_decoder(const char *hmm)
{
    config_t *c = config_init(NULL);
    decoder_t *d;
    config_set_str(c, "hmm", hmm);
    config_set_str(c, "compallsen", "yes");
    config_set_str(c, "loglevel", "ERROR");
    d = decoder_init(c);
    CHECK(d != NULL, "decoder_init");
    return d;
}

static alignment_t *
align_goforward(decoder_t *d, const int16 *pcm, size_t n)
{
    CHECK(decoder_set_align_text(d, "go forward ten meters") == 0, "decoder_set_align_text");
    CHECK(decoder_start_utt(d) == 0, "decoder_start_utt");
    CHECK(decoder_process_int16(d, (int16 *)pcm, n, FALSE, TRUE) >= 0, "decoder_process_int16");
    CHECK(decoder_end_utt(d) == 0, "decoder_end_utt");
    return decoder_alignment(d);
}

int
main(int argc, char **argv)
{
    FILE *fh;
    int16 *pcm;
    size_t n;
    decoder_t *d, *b;
    alignment_t *al, *again;
    levels_t ref = { { 0 }, { 0 } }, got = { { 0 }, { 0 } };
    int i, nfr;

    CHECK(argc == 3, "usage: %s <hmm dir> <goforward.raw>", argv[0]);
    fh = fopen(argv[2], "rb");
    CHECK(fh != NULL, "cannot open %s", argv[2]);
    pcm = malloc(1 << 20);
    n = fread(pcm, 2, 1 << 19, fh);
    fclose(fh);

    /* ---- run 1: the reference as it stands ---- */
    d = make_decoder(argv[1]);
    use_adapter = 0;
    al = align_goforward(d, pcm, n);
    CHECK(al != NULL, "run 1: no alignment");
    grab(al, &ref);
    nfr = d->acmod->output_frame;
    CHECK(ref.n[0] >= 4 && ref.n[2] == 3 * ref.n[1] && nfr > 100, "run 1: implausible alignment");

    /* ---- run 2: the shim's search module inside decoder_alignment ---- */
    stub_canned_states = ref.e[2];
    stub_canned_n = ref.n[2];
    use_adapter = 1;
    al = align_goforward(d, pcm, n);
    CHECK(al != NULL, "run 2: no alignment (%s)", ssw_last_error());
    CHECK(stub_init_calls == 1 && stub_start_calls == 1 && stub_finish_calls == 1, "run 2: init/start/finish called %d/%d/%d times",
          stub_init_calls, stub_start_calls, stub_finish_calls);
    CHECK(stub_n_steps == nfr && stub_step_out_of_order == 0, "run 2: %d steps for %d frames, %d out of order",
          stub_n_steps, nfr, stub_step_out_of_order);
    CHECK(stub_init_phones == ref.n[1], "run 2: %d phones handed over, %d expected", stub_init_phones, ref.n[1]);
    {   /* the per-phone rows: ssid / tmatid of the populated alignment, word windows as start / duration */
        alignment_iter_t *it;
        for (i = 0, it = alignment_phones(al); it; it = alignment_iter_next(it), ++i) {
            alignment_entry_t *e = alignment_iter_get(it);
            CHECK(stub_init_ssid[i] == e->id.pid.ssid && stub_init_tmatid[i] == e->id.pid.tmatid,
                  "run 2: phone %d ids", i);
        }
        /* windows are read BEFORE the search fills the entries: the first phone starts at 0 */
        CHECK(stub_init_start[0] == 0, "run 2: first window starts at %d", stub_init_start[0]);
    }
    grab(al, &got);
    same(&ref, &got, "run 2 (adapter) vs run 1 (reference)");
    {   /* ->frame and ->al through the reference's state_align_search_t view: the reuse branch */
        state_align_search_t *view = (state_align_search_t *)d->align;
        CHECK(view->frame == d->acmod->output_frame && view->al == al, "run 2: ->frame %d (output_frame %d) / ->al",
              view->frame, d->acmod->output_frame);
        again = decoder_alignment(d);
        CHECK(again == al && stub_init_calls == 1, "run 2: the existing alignment was not reused");
    }
    {   /* the rows handed to step() are the rows acmod holds */
        int fi, bad = 0;
        CHECK(acmod_rewind(d->acmod) == 0, "acmod_rewind");
        for (fi = 0; fi < nfr && fi < 512; ++fi) {
            int f2 = fi;
            mfcc_t **feat = acmod_get_frame(d->acmod, &f2);
            CHECK(feat != NULL, "acmod_get_frame(%d)", fi);
            bad += memcmp(feat[0], stub_rows[fi], sizeof(stub_rows[0])) != 0;
        }
        CHECK(bad == 0, "run 2: %d feature rows differ from acmod's", bad);
    }
    /* a new alignment request frees the module: base first, then the GPU object, then al */
    use_adapter = 0;
    al = align_goforward(d, pcm, n);
    CHECK(al != NULL && stub_free_calls == 1, "run 2: the module was freed %d times", stub_free_calls);
    grab(al, &got);
    same(&ref, &got, "run 2b (reference again after the adapter)");

    /* ---- run 3: the scorer swap ---- */
    b = make_decoder(argv[1]);               /* lends its scorer to the stub */
    stub_backend_mgau = b->acmod->mgau;
    {
        decoder_t *a = make_decoder(argv[1]);
        CHECK(use_gpu_scorer(a, argv[1]) == 0, "use_gpu_scorer");
        CHECK(strcmp(a->acmod->mgau->vt->name, "ptm") == 0, "run 3: slot 0 (name)");
        al = align_goforward(a, pcm, n);
        CHECK(al != NULL, "run 3: no alignment");
        CHECK(stub_frame_eval_calls >= 2 * nfr, "run 3: frame_eval called %d times for 2 x %d frames",
              stub_frame_eval_calls, nfr);
        grab(al, &got);
        same(&ref, &got, "run 3 (scorer swapped) vs run 1");
        decoder_free(a);
        CHECK(stub_mgau_free_calls == 1, "run 3: slot 3 (free) called %d times", stub_mgau_free_calls);
    }
    decoder_free(b);
    decoder_free(d);
    free(pcm);
    stub_release();
    for (i = 0; i < 3; ++i)
        free(got.e[i]);
    {
        int nw = ref.n[0], np = ref.n[1], ns = ref.n[2];
        for (i = 0; i < 3; ++i)
            free(ref.e[i]);
        ref.n[0] = nw, ref.n[1] = np, ref.n[2] = ns;
    }
    printf("SHIM-DRIVER OK: %d words, %d phones, %d states over %d frames; adapter and scorer swap "
           "reproduce the reference's alignment\n", ref.n[0], ref.n[1], ref.n[2], nfr);
    return 0;
}
