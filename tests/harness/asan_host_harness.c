/* Host-only sanitizer harness for the product's C code (model loaders, dictionary, triphone
 * lookup, first-pass graph builder incl. its threaded merge, alignment_populate, JSON writer):
 * built by tests/test_host_sanitizers.py with -fsanitize=address,undefined from
 * csrc/ssw_model.c + ssw_lexicon.c + ssw_fsg.c and this file, which supplies the one thing the
 * HIP translation unit normally provides to them (the model handle).  GPU AddressSanitizer is
 * not available on the pool, so this is how the host code gets sanitized.
 *
 *   usage: harness <model dir> <n_texts> word word word ...   (texts of 1..7 words drawn in turn) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ssw_internal.h"

struct ssw_model_s {
    ssw_host_model_t *h;
};

const ssw_host_model_t *
ssw_model_host(const ssw_model_t *m)
{
    return m->h;
}

int
main(int argc, char **argv)
{
    char p[6][600];
    const char *dir = argv[1];
    struct ssw_model_s m;
    ssw_dict_t *d;
    ssw_fp_graphs_t *g;
    int n_texts, n_vocab, u, k = 0;
    int32_t *word_off;
    const char **words;
    long check = 0;

    if (argc < 4)
        return 2;
    n_texts = atoi(argv[2]);
    n_vocab = argc - 3;
    snprintf(p[0], sizeof p[0], "%s/mdef", dir);
    snprintf(p[1], sizeof p[1], "%s/means", dir);
    snprintf(p[2], sizeof p[2], "%s/variances", dir);
    snprintf(p[3], sizeof p[3], "%s/sendump", dir);
    snprintf(p[4], sizeof p[4], "%s/transition_matrices", dir);
    m.h = ssw_host_model_load(p[0], p[1], p[2], p[3], NULL, p[4], NULL);
    if (m.h == NULL) {
        fprintf(stderr, "load: %s\n", ssw_last_error());
        return 1;
    }
    snprintf(p[0], sizeof p[0], "%s/dict.txt", dir);
    snprintf(p[1], sizeof p[1], "%s/noisedict.txt", dir);
    d = ssw_dict_load(&m, p[0], p[1]);
    if (d == NULL) {
        fprintf(stderr, "dict: %s\n", ssw_last_error());
        return 1;
    }
    word_off = (int32_t *)calloc((size_t)n_texts + 1, sizeof(int32_t));
    words = (const char **)calloc((size_t)n_texts * 7 + 1, sizeof(char *));
    for (u = 0; u < n_texts; ++u) {
        int n = 1 + u % 7, i;
        word_off[u] = k;
        for (i = 0; i < n; ++i, ++k)
            words[k] = argv[3 + (u * 5 + i * 3) % n_vocab];
    }
    word_off[n_texts] = k;
    /* the whole batch (threads when n_texts >= 32), then text by text: same totals */
    g = ssw_fp_graphs_build(&m, d, NULL, n_texts, word_off, words);
    if (g == NULL) {
        fprintf(stderr, "graphs: %s\n", ssw_last_error());
        return 1;
    }
    for (u = 0; u < n_texts; ++u) {
        ssw_fp_node_t nodes[16];
        int32_t beams[3];
        int n = ssw_first_pass_graph(&m, d, NULL, word_off[u + 1] - word_off[u], words + word_off[u],
                                     16, nodes, beams);
        if (n != g->node_off[u + 1] - g->node_off[u]) {
            fprintf(stderr, "text %d: %d nodes alone, %d in the batch\n", u, n,
                    g->node_off[u + 1] - g->node_off[u]);
            return 1;
        }
        check += n;
    }
    for (u = 0; u < g->n_nodes; ++u)
        check += g->pen[u] + g->parent[u] + (long)(g->ctxt[u] & 0xff) + g->twin_ref[u];
    for (u = 0; u < g->n_tw; ++u)
        check += g->tw[u];
    for (u = 0; u < g->n_in; ++u)
        check += g->in_leaf[u];
    /* alignment_populate + the JSON writer on the first text */
    {
        int32_t ssid[256], tmat[256], ci[256], par[256];
        ssw_align_entry_t wal[8], pal[256];
        char out[8192];
        int nw = word_off[1] - word_off[0], np, i;
        np = ssw_alignment_populate(&m, d, nw, words, NULL, NULL, 256, ssid, tmat, ci, par, NULL, NULL);
        if (np < 0)
            return 1;
        for (i = 0; i < np; ++i) {
            pal[i].start = i * 3;
            pal[i].duration = 3;
            pal[i].score = -10 * i;
        }
        for (i = 0; i < nw; ++i) {
            wal[i].start = 0;
            wal[i].duration = 3 * np;
            wal[i].score = -1;
        }
        if (ssw_alignment_json(&m, "x", 0, 0.0, 100, 3 * np, nw, words, wal, np, ci, par, pal, NULL,
                               NULL, out, sizeof(out)) < 0)
            return 1;
        check += (long)strlen(out);
    }
    printf("ok %d texts, %d nodes, %d leaves, %d twin ints, checksum %ld\n", n_texts, g->n_nodes,
           g->n_leaves, g->n_tw, check);
    ssw_fp_graphs_free(g);
    ssw_dict_free(d);
    ssw_host_model_free(m.h);
    free(word_off);
    free(words);
    return 0;
}
