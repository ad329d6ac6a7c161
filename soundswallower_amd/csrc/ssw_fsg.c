/*
 * ssw_fsg.c -- host C: the graphs the first pass of forced alignment searches (SURVEY.md
 * section 8(f) row 4).  For every text of a batch: the linear word grammar of
 * decoder_set_align_text (src/decoder.c:686-735), the silence / filler loops and alternate
 * pronunciations fsg_search_init adds (src/fsg_search.c:84-170), and the per-state phone trees
 * of fsg_lextree_init (src/fsg_lextree.c:85-214 context lists, 356-587 psubtree_add_trans),
 * flattened into the arrays the GPU kernel walks (ssw_k5_firstpass.inc).
 *
 * The trees keep the reference's sharing rules, because they decide which HMMs exist:
 *   - one word-initial node per (first phone, second phone) of a state, shared by every word
 *     and alternate that starts that way; it serves ALL left contexts with the senones of the
 *     FIRST (lowest-numbered) left-context phone (src/fsg_lextree.c:496-524: the search for an
 *     existing node never comes back empty-handed once one node exists);
 *   - word-internal nodes shared by senone sequence under the same predecessor (:528-540);
 *   - one word-final node per distinct right-context senone sequence, per word (:563-600);
 *   - one-phone words: one node per distinct left-context senone sequence, right context
 *     SIL, exits valid for every right context (:404-445, src/fsg_search.c:459-473);
 *   - fillers: context-independent, present SIL to their neighbours (:446-468).
 * Orders the reference takes from hash-table iteration only break exact score ties; here
 * links are taken as: the word, its alternates (newest first), <sil>, the other fillers.
 */
#include "ssw_internal.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

enum { POS_INTERNAL = 0, POS_BEGIN = 1, POS_END = 2, POS_SINGLE = 3 };

const ssw_host_model_t *ssw_model_host(const ssw_model_t *m);

void
ssw_first_pass_config_defaults(ssw_first_pass_config_t *cfg)
{
    memset(cfg, 0, sizeof(*cfg));
    cfg->beam = 1e-48;
    cfg->pbeam = 1e-48;
    cfg->wbeam = 7e-29;
    cfg->wip = 0.65;
    cfg->pip = 1.0;
    cfg->lw = 6.5f;
    cfg->silprob = 0.005f;
    cfg->fillprob = 1e-8f;
    cfg->use_filler = 1;
    cfg->use_altpron = 1;
}

/* logmath_log at shift 0 (src/logmath.c:282-290) */
static int32_t
ilog0(double base, double p)
{
    if (p <= 0)
        return (int32_t)0x80000000 >> 2;
    return (int32_t)(log(p) * (1.0 / log(base)));
}

static int
cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

typedef struct {
    int word;     /* dictionary id */
    int to;       /* destination state */
    int logp;     /* fsg_link_logs2prob */
    int filler;
} link_t;

/* growable arrays of the batch */
typedef struct {
    ssw_fp_graphs_t *g;
    int cap_nodes, cap_leaves, cap_states, cap_in, cap_tw;
    int rk_last; /* rank-buffer ints of the utterance just built */
} builder_t;

static int
grow(void **p, int *cap, int need, size_t elt)
{
    if (need > *cap) {
        int nc = *cap ? *cap : 1024;
        void *q;
        while (nc < need)
            nc *= 2;
        q = realloc(*p, (size_t)nc * elt);
        if (q == NULL)
            return -1;
        *p = q;
        *cap = nc;
    }
    return 0;
}

static int
reserve_nodes(builder_t *b, int need)
{
    ssw_fp_graphs_t *g = b->g;
    int c0 = b->cap_nodes, c;
    c = c0; if (grow((void **)&g->senid, &c, need, 4 * sizeof(uint16_t)) < 0) return -1;
    c = c0; if (grow((void **)&g->pen, &c, need, sizeof(int32_t)) < 0) return -1;
    c = c0; if (grow((void **)&g->parent, &c, need, sizeof(int32_t)) < 0) return -1;
    c = c0; if (grow((void **)&g->info, &c, need, sizeof(uint32_t)) < 0) return -1;
    c = c0; if (grow((void **)&g->ctxt, &c, need, sizeof(uint64_t)) < 0) return -1;
    c = c0; if (grow((void **)&g->leaf_ord, &c, need, sizeof(int32_t)) < 0) return -1;
    c = c0; if (grow((void **)&g->twin_ref, &c, need, sizeof(int32_t)) < 0) return -1;
    b->cap_nodes = c;
    return 0;
}

static int
reserve_leaves(builder_t *b, int need)
{
    ssw_fp_graphs_t *g = b->g;
    int c0 = b->cap_leaves, c;
    c = c0; if (grow((void **)&g->leaf_wid, &c, need, sizeof(int32_t)) < 0) return -1;
    c = c0; if (grow((void **)&g->leaf_to, &c, need, sizeof(int32_t)) < 0) return -1;
    c = c0; if (grow((void **)&g->leaf_node, &c, need, sizeof(int32_t)) < 0) return -1;
    b->cap_leaves = c;
    return 0;
}

#define INFO_ROOT 1u
#define INFO_LEAF 2u
#define INFO_ALLRC 4u
#define INFO_TWIN 8u        /* one of several word-final HMMs that can never differ */
#define INFO_TWIN_FIRST 16u /* the first of them in the predecessor's successor chain */
#define INFO_TWIN_LAST 32u

/* new node of the utterance being built; returns its local index or -1 */
static int
add_node(builder_t *b, const ssw_host_model_t *h, int base, int ssid, int ci, int pen, int parent,
         uint32_t flags, int ci_ext, int state, uint64_t ctxt)
{
    ssw_fp_graphs_t *g = b->g;
    int n = g->n_nodes, j;
    if (reserve_nodes(b, n + 1) < 0)
        return -1;
    for (j = 0; j < 3; ++j)
        g->senid[(size_t)n * 4 + j] = h->sseq[(size_t)ssid * h->n_emit_state + j];
    g->senid[(size_t)n * 4 + 3] = (uint16_t)h->phone_tmat[ci]; /* bin_mdef_pid2tmatid(mdef, ci) */
    g->pen[n] = pen;
    g->parent[n] = parent;
    g->info[n] = flags | ((uint32_t)ci_ext << 8) | ((uint32_t)state << 16);
    g->ctxt[n] = ctxt;
    g->leaf_ord[n] = -1;
    g->twin_ref[n] = -1;
    ++g->n_nodes;
    return n - base;
}

static int
mark_leaf(builder_t *b, int base, int leaf_base, int node, int wid, int to)
{
    ssw_fp_graphs_t *g = b->g;
    if (reserve_leaves(b, g->n_leaves + 1) < 0)
        return -1;
    g->leaf_ord[base + node] = g->n_leaves - leaf_base;
    g->leaf_wid[g->n_leaves] = wid;
    g->leaf_to[g->n_leaves] = to;
    g->leaf_node[g->n_leaves] = node;
    ++g->n_leaves;
    return 0;
}

static int
ssid_of(const ssw_model_t *m, const ssw_host_model_t *h, int b, int l, int r, int pos)
{
    int pid = ssw_phone_id_nearest(m, b, l, r, pos);
    if (pid < 0 || pid >= h->n_phone)
        return -1;
    return h->phone_ssid[pid];
}

/* the ssid a node was made with: compare its three senones (sseq rows are unique per ssid) */
static int
same_ssid(const ssw_fp_graphs_t *g, const ssw_host_model_t *h, int node, int ssid)
{
    int j;
    for (j = 0; j < 3; ++j)
        if (g->senid[(size_t)node * 4 + j] != h->sseq[(size_t)ssid * h->n_emit_state + j])
            return 0;
    return 1;
}

/* word-final nodes i and j (absolute indices) that can never differ: the same senones into the
 * same destination, under the same predecessor -- or, for one-phone words, hanging off the same
 * state with the same left contexts */
static int
alike(const ssw_fp_graphs_t *g, int leaf_base, int i, int j)
{
    const int is_root = (g->info[i] & INFO_ROOT) != 0;
    if (!(g->info[j] & INFO_LEAF) || ((g->info[j] & INFO_ROOT) != 0) != is_root)
        return 0;
    if (is_root ? ((g->info[j] >> 16) != (g->info[i] >> 16) || g->ctxt[j] != g->ctxt[i])
                : g->parent[j] != g->parent[i])
        return 0;
    return g->leaf_to[leaf_base + g->leaf_ord[j]] == g->leaf_to[leaf_base + g->leaf_ord[i]]
        && memcmp(&g->senid[(size_t)j * 4], &g->senid[(size_t)i * 4], 3 * sizeof(uint16_t)) == 0;
}

static int
build_one(builder_t *b, const ssw_model_t *m, const ssw_host_model_t *h, const ssw_dict_t *d,
          const ssw_first_pass_config_t *cfg, int n_words, const char *const *words, int wip,
          int pip, int logsil, int logfil)
{
    ssw_fp_graphs_t *g = b->g;
    const int n_state = n_words + 1, sil = h->sil;
    const int base = g->n_nodes, leaf_base = g->n_leaves;
    uint64_t *lcset = (uint64_t *)calloc((size_t)n_state, sizeof(uint64_t));
    uint64_t *rcset = (uint64_t *)calloc((size_t)n_state, sizeof(uint64_t));
    link_t *links = NULL;
    int *link_off = (int *)calloc((size_t)n_state + 1, sizeof(int));
    int n_links = 0, cap_links = 0, s, i, k, rc = -1;

    if (!lcset || !rcset || !link_off)
        goto oom;
    /* links per state: word + alternates to s+1, then the filler loops */
    for (s = 0; s < n_state; ++s) {
        link_off[s] = n_links;
        if (s < n_words) {
            int w = ssw_dict_find(d, words[s]);
            if (w < 0) {
                ssw_set_error("Unknown word %s", words[s]); /* src/decoder.c:699-703 */
                goto bad;
            }
            /* fsg_search_add_altpron walks the dict_nextalt chain from the word and
             * fsg_model_add_alt PREPENDS every alternate's link (src/fsg_model.c:430-444): the
             * state's list ends up w(2), w(3), ..., w(k), w.  The order matters: alternates with
             * identical pronunciations tie for ever and the list order picks the one reported */
            {
                int n_alt = 0, j;
                for (k = cfg->use_altpron ? d->alt[w] : -1; k >= 0; k = d->alt[k])
                    ++n_alt;
                if (grow((void **)&links, &cap_links, n_links + n_alt + 1, sizeof(link_t)) < 0)
                    goto oom;
                j = n_links + n_alt - 1;
                for (k = cfg->use_altpron ? d->alt[w] : -1; k >= 0; k = d->alt[k], --j)
                    links[j].word = k;
                links[n_links + n_alt].word = w;
                for (j = n_links; j <= n_links + n_alt; ++j) {
                    links[j].to = s + 1;
                    links[j].logp = 0;
                    links[j].filler = 0;
                }
                n_links += n_alt + 1;
            }
        }
        if (cfg->use_filler) {
            /* fsg_search_add_silences (src/fsg_search.c:84-119): <sil> with silprob, then the
             * filler words from dict_filler_start up to BUT NOT INCLUDING dict_filler_end
             * (the loop's `wid < dict_filler_end`), except <s> and </s>, with fillprob;
             * <sil> met again there keeps its larger probability (fsg_model_trans_add) */
            const int sw = ssw_dict_find(d, "<sil>"), start = ssw_dict_find(d, "<s>"),
                      fin = ssw_dict_find(d, "</s>");
            int f;
            for (f = -1; f < d->n_words - 1; f = (f < 0 ? d->filler_start : f + 1)) {
                const int w = f < 0 ? sw : f;
                if (w < 0 || (f >= 0 && (w == sw || w == start || w == fin)))
                    continue;
                for (k = w; k >= 0; k = (cfg->use_altpron ? d->alt[k] : -1)) {
                    if (grow((void **)&links, &cap_links, n_links + 1, sizeof(link_t)) < 0)
                        goto oom;
                    links[n_links].word = k;
                    links[n_links].to = s;
                    links[n_links].logp = (w == sw) ? logsil : logfil;
                    links[n_links].filler = 1;
                    ++n_links;
                }
            }
        }
    }
    link_off[n_state] = n_links;

    /* fsg_lextree_lc_rc: context phone sets per state (no null transitions to propagate) */
    for (s = 0; s < n_state; ++s) {
        lcset[s] |= 1ull << sil;
        rcset[s] |= 1ull << sil;
    }
    for (s = 0; s < n_state; ++s)
        for (i = link_off[s]; i < link_off[s + 1]; ++i) {
            const link_t *l = &links[i];
            if (l->filler)
                continue; /* SIL on both sides, already there */
            rcset[s] |= 1ull << d->pron[l->word][0];
            lcset[l->to] |= 1ull << d->pron[l->word][d->pronlen[l->word] - 1];
        }

    /* the phone trees */
    for (s = 0; s < n_state; ++s) {
        const int first_node = g->n_nodes - base; /* nodes of this state start here */
        for (i = link_off[s]; i < link_off[s + 1]; ++i) {
            const link_t *l = &links[i];
            const int16_t *pron = d->pron[l->word];
            const int len = d->pronlen[l->word], lp = l->logp >> SSW_SENSCR_SHIFT;
            int p, pred = -1;
            if (len == 1) {
                const int ci = pron[0];
                if (l->filler) {
                    int n = add_node(b, h, base, h->phone_ssid[ci], ci, lp + wip + pip, -1,
                                     INFO_ROOT | INFO_LEAF | INFO_ALLRC, sil, s, ~0ull);
                    if (n < 0 || mark_leaf(b, base, leaf_base, n, l->word, l->to) < 0)
                        goto oom;
                } else {
                    const int mine = g->n_nodes - base; /* nodes of this word start here */
                    int lc;
                    for (lc = 0; lc < h->n_ciphone; ++lc) {
                        int ssid, n;
                        if (!((lcset[s] >> lc) & 1))
                            continue;
                        ssid = ssid_of(m, h, ci, lc, sil, POS_SINGLE);
                        if (ssid < 0)
                            goto bad_phone;
                        for (n = mine; n < g->n_nodes - base; ++n)
                            if (same_ssid(g, h, base + n, ssid))
                                break;
                        if (n == g->n_nodes - base) {
                            n = add_node(b, h, base, ssid, ci, lp + wip + pip, -1,
                                         INFO_ROOT | INFO_LEAF | INFO_ALLRC, ci, s, 0);
                            if (n < 0 || mark_leaf(b, base, leaf_base, n, l->word, l->to) < 0)
                                goto oom;
                        }
                        g->ctxt[base + n] |= 1ull << lc;
                    }
                }
                continue;
            }
            for (p = 0; p < len; ++p) {
                const int ci = pron[p];
                if (p == 0) {
                    int n, lc0 = 0;
                    rc = pron[1];
                    /* an existing word-initial node of this state for (ci, rc)? */
                    for (n = first_node; n < g->n_nodes - base; ++n) {
                        uint32_t inf = g->info[base + n];
                        if ((inf & INFO_ROOT) && !(inf & INFO_LEAF)
                            && g->parent[base + n] == -(2 + ci * 256 + rc))
                            break;
                    }
                    if (n < g->n_nodes - base) {
                        pred = n;
                        continue;
                    }
                    while (!((lcset[s] >> lc0) & 1))
                        ++lc0;
                    {
                        int ssid = ssid_of(m, h, ci, lc0, rc, POS_BEGIN);
                        if (ssid < 0)
                            goto bad_phone;
                        n = add_node(b, h, base, ssid, ci, wip + pip, -1, INFO_ROOT, ci, s,
                                     lcset[s]);
                        if (n < 0)
                            goto oom;
                        /* roots have no predecessor: the slot remembers the diphone instead */
                        g->parent[base + n] = -(2 + ci * 256 + rc);
                        pred = n;
                    }
                } else if (p != len - 1) {
                    int ssid = ssid_of(m, h, ci, pron[p - 1], pron[p + 1], POS_INTERNAL), n;
                    if (ssid < 0)
                        goto bad_phone;
                    for (n = first_node; n < g->n_nodes - base; ++n)
                        if (g->parent[base + n] == pred && !(g->info[base + n] & INFO_LEAF)
                            && same_ssid(g, h, base + n, ssid))
                            break;
                    if (n == g->n_nodes - base) {
                        n = add_node(b, h, base, ssid, ci, pip, pred, 0, ci, s, 0);
                        if (n < 0)
                            goto oom;
                    }
                    pred = n;
                } else {
                    const int mine = g->n_nodes - base;
                    int r;
                    for (r = 0; r < h->n_ciphone; ++r) {
                        int ssid, n;
                        if (!((rcset[l->to] >> r) & 1))
                            continue;
                        ssid = ssid_of(m, h, ci, pron[p - 1], r, POS_END);
                        if (ssid < 0)
                            goto bad_phone;
                        for (n = mine; n < g->n_nodes - base; ++n)
                            if (same_ssid(g, h, base + n, ssid))
                                break;
                        if (n == g->n_nodes - base) {
                            n = add_node(b, h, base, ssid, ci, lp + pip, pred, INFO_LEAF, ci, s, 0);
                            if (n < 0 || mark_leaf(b, base, leaf_base, n, l->word, l->to) < 0)
                                goto oom;
                        }
                        g->ctxt[base + n] |= 1ull << r;
                    }
                }
            }
        }
    }
    /* twins: word-final nodes of different links under the same predecessor with the same
     * senones and destination (alternates pronounced alike).  They score alike for ever; the
     * reference keeps one exit of the two (src/fsg_history.c:164-170, the later arrival's right
     * contexts are all covered) and which one arrives first alternates with the frame, because
     * its active list is rebuilt by prepending.  Creation order here = the order of the
     * predecessor's successor chain (links in list order). */
    for (i = base; i < g->n_nodes; ++i) {
        int j, last = i;
        const int is_root = (g->info[i] & INFO_ROOT) != 0;
        if (!(g->info[i] & INFO_LEAF) || (g->info[i] & INFO_TWIN))
            continue;
        /* a state's nodes are contiguous and twins hang off the same state */
        for (j = i + 1; j < g->n_nodes && (g->info[j] >> 16) == (g->info[i] >> 16); ++j)
            if (alike(g, leaf_base, i, j)) {
                g->info[j] |= INFO_TWIN;
                last = j;
            }
        if (last != i) {
            /* word-final nodes hang off their predecessor in creation order; one-phone words
             * are word-initial nodes, which the reference links newest first (:421, :458) */
            g->info[i] |= INFO_TWIN | (is_root ? INFO_TWIN_LAST : INFO_TWIN_FIRST);
            g->info[last] |= is_root ? INFO_TWIN_FIRST : INFO_TWIN_LAST;
        }
    }
    /* twin records (see ssw_internal.h): what a member needs to follow the reference's list
     * order among its group and their ancestors */
    {
        const int tw_base = g->n_tw;
        int rk = 0;
        for (i = base; i < g->n_nodes; ++i) {
            int members[64], anc[64], n_mem = 0, n_anc = 0, j, q, L;
            if (!(g->info[i] & INFO_TWIN_FIRST) && !(g->info[i] & INFO_TWIN_LAST))
                continue;
            /* handle a group once, from its lowest-numbered member */
            {
                const int is_root = (g->info[i] & INFO_ROOT) != 0;
                int lowest = 1;
                for (j = i - 1; j >= base && (g->info[j] >> 16) == (g->info[i] >> 16); --j)
                    if ((g->info[j] & INFO_TWIN) && alike(g, leaf_base, i, j))
                        lowest = 0;
                if (!lowest)
                    continue;
                for (j = i; j < g->n_nodes && n_mem < 64 && (g->info[j] >> 16) == (g->info[i] >> 16); ++j)
                    if ((g->info[j] & INFO_TWIN) && alike(g, leaf_base, i, j))
                        members[n_mem++] = j - base;
                if (is_root) /* word-initial nodes are linked newest first */
                    for (j = 0; j < n_mem / 2; ++j) {
                        q = members[j];
                        members[j] = members[n_mem - 1 - j];
                        members[n_mem - 1 - j] = q;
                    }
                else
                    for (q = g->parent[i]; q >= 0 && n_anc < 64; q = g->parent[base + q] < -1 ? -1 : g->parent[base + q])
                        anc[n_anc++] = q;
            }
            if (n_mem >= 64 || n_anc >= 64) {
                ssw_set_error("a word of the text has 64+ phones or 64+ alternates pronounced alike");
                goto bad;
            }
            L = n_anc + n_mem;
            for (j = 0; j < n_mem; ++j) {
                int o;
                if (grow((void **)&g->tw, &b->cap_tw, g->n_tw + 4 + L, sizeof(int32_t)) < 0)
                    goto oom;
                o = g->n_tw;
                g->twin_ref[base + members[j]] = o - tw_base;
                g->tw[o] = L;
                g->tw[o + 1] = n_anc;
                g->tw[o + 2] = n_anc + j;
                g->tw[o + 3] = rk;
                for (q = 0; q < n_anc; ++q) /* root first */
                    g->tw[o + 4 + q] = anc[n_anc - 1 - q];
                for (q = 0; q < n_mem; ++q)
                    g->tw[o + 4 + n_anc + q] = members[q];
                g->n_tw += 4 + L;
                rk += 3 * L;
            }
        }
        b->rk_last = rk;
    }
    /* roots: forget the diphone note */
    for (i = base; i < g->n_nodes; ++i)
        if (g->parent[i] < -1)
            g->parent[i] = -1;

    /* leaves entering each state, by (left-context phone they present, ordinal) */
    {
        const int nl = g->n_leaves - leaf_base;
        uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(nl ? nl : 1));
        int st, k2 = 0;
        if (keys == NULL
            || grow((void **)&g->in_off, &b->cap_states, g->n_states + n_state + 1, sizeof(int32_t)) < 0
            || grow((void **)&g->in_leaf, &b->cap_in, g->n_in + nl, sizeof(int32_t)) < 0) {
            free(keys);
            goto oom;
        }
        for (i = 0; i < nl; ++i) {
            const int node = base + g->leaf_node[leaf_base + i];
            keys[i] = ((uint64_t)g->leaf_to[leaf_base + i] << 40)
                | ((uint64_t)((g->info[node] >> 8) & 0xff) << 32) | (uint64_t)i;
        }
        qsort(keys, (size_t)nl, sizeof(uint64_t), cmp_u64);
        for (st = 0; st < n_state; ++st) {
            g->in_off[g->n_states + st] = g->n_in;
            while (k2 < nl && (int)(keys[k2] >> 40) == st)
                g->in_leaf[g->n_in++] = (int32_t)(keys[k2++] & 0xffffffffu);
        }
        free(keys);
        g->n_states += n_state;
        g->in_off[g->n_states] = g->n_in;
    }
    free(lcset);
    free(rcset);
    free(links);
    free(link_off);
    return 0;
bad_phone:
    ssw_set_error("no triphone for a phone of the text (model and dictionary do not match)");
    goto bad;
oom:
    ssw_set_error("out of memory building the first-pass graphs");
bad:
    free(lcset);
    free(rcset);
    free(links);
    free(link_off);
    return -1;
}

static ssw_fp_graphs_t *
graphs_build_serial(const ssw_model_t *m, const ssw_dict_t *d, const ssw_first_pass_config_t *cfg_in,
                    int32_t n_utts, const int32_t *word_off, const char *const *words)
{
    const ssw_host_model_t *h = ssw_model_host(m);
    ssw_first_pass_config_t cfg;
    builder_t b;
    ssw_fp_graphs_t *g;
    int u, wip, pip, logsil, logfil;
    const double base = h->cfg.logbase;

    if (cfg_in)
        cfg = *cfg_in;
    else
        ssw_first_pass_config_defaults(&cfg);
    if (h->sil < 0 || h->sseq == NULL || h->cd_tree == NULL || h->n_emit_state != 3) {
        ssw_set_error("the first pass needs a 3-state model loaded with its mdef");
        return NULL;
    }
    if (h->n_ciphone > 64) { /* the reference's own limit is 128 (FSG_PNODE_CTXT_BVSZ) */
        ssw_set_error("%d CI phones: the first pass handles at most 64", h->n_ciphone);
        return NULL;
    }
    memset(&b, 0, sizeof(b));
    g = b.g = (ssw_fp_graphs_t *)calloc(1, sizeof(*g));
    if (g == NULL)
        return NULL;
    g->n_utts = n_utts;
    g->node_off = (int32_t *)calloc((size_t)n_utts + 1, sizeof(int32_t));
    g->leaf_off = (int32_t *)calloc((size_t)n_utts + 1, sizeof(int32_t));
    g->state_off = (int32_t *)calloc((size_t)n_utts + 1, sizeof(int32_t));
    g->tw_off = (int32_t *)calloc((size_t)n_utts + 1, sizeof(int32_t));
    g->tw_rk = (int32_t *)calloc((size_t)n_utts + 1, sizeof(int32_t));
    /* fsg_search_init, src/fsg_search.c:198-217 and fsg_model_add_silence, fsg_model.c:367 */
    g->beam = ilog0(base, cfg.beam) >> SSW_SENSCR_SHIFT;
    g->pbeam = ilog0(base, cfg.pbeam) >> SSW_SENSCR_SHIFT;
    g->wbeam = ilog0(base, cfg.wbeam) >> SSW_SENSCR_SHIFT;
    pip = (int32_t)((float)ilog0(base, cfg.pip) * cfg.lw) >> SSW_SENSCR_SHIFT;
    wip = (int32_t)((float)ilog0(base, cfg.wip) * cfg.lw) >> SSW_SENSCR_SHIFT;
    logsil = (int32_t)((float)ilog0(base, (double)cfg.silprob) * cfg.lw);
    logfil = (int32_t)((float)ilog0(base, (double)cfg.fillprob) * cfg.lw);
    for (u = 0; u < n_utts; ++u) {
        g->node_off[u] = g->n_nodes;
        g->leaf_off[u] = g->n_leaves;
        g->state_off[u] = g->n_states;
        g->tw_off[u] = g->n_tw;
        if (build_one(&b, m, h, d, &cfg, word_off[u + 1] - word_off[u],
                      words + (word_off[u] - word_off[0]), wip, pip, logsil, logfil) < 0) {
            ssw_fp_graphs_free(g);
            return NULL;
        }
        g->tw_rk[u] = b.rk_last;
    }
    g->tw_off[n_utts] = g->n_tw;
    g->node_off[n_utts] = g->n_nodes;
    g->leaf_off[n_utts] = g->n_leaves;
    g->state_off[n_utts] = g->n_states;
    return g;
}

/* host threads for per-utterance work: SSW_HOST_THREADS, default 8 */
int
ssw_host_threads(void)
{
    const char *e = getenv("SSW_HOST_THREADS");
    int n = e ? atoi(e) : 8;
    return n < 1 ? 1 : n;
}

/* A small persistent pool for the per-utterance host work of a batch (graph building here,
 * alignment_populate in ssw_host_firstpass.inc): starting and joining a dozen threads costs
 * more than the work they share at batch sizes of a few hundred utterances.  The workers are
 * created on first use (SSW_HOST_THREADS - 1 of them; the caller takes chunks too) and sleep on
 * a condition variable between calls.  One job at a time; fn must not call back into the pool. */
static struct {
    pthread_mutex_t job_mu, mu;
    pthread_cond_t cv_work, cv_done;
    int n_workers, started;
    void (*fn)(void *, int);
    void *arg;
    int n_chunks, next, done;
    unsigned gen;
} g_pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER,
             PTHREAD_COND_INITIALIZER, 0, 0, NULL, NULL, 0, 0, 0, 0u };

static void
pool_take_chunks(void)
{
    for (;;) {
        int k;
        pthread_mutex_lock(&g_pool.mu);
        k = g_pool.next < g_pool.n_chunks ? g_pool.next++ : -1;
        pthread_mutex_unlock(&g_pool.mu);
        if (k < 0)
            return;
        g_pool.fn(g_pool.arg, k);
        pthread_mutex_lock(&g_pool.mu);
        if (++g_pool.done == g_pool.n_chunks)
            pthread_cond_broadcast(&g_pool.cv_done);
        pthread_mutex_unlock(&g_pool.mu);
    }
}

static void *
pool_worker(void *unused)
{
    unsigned seen = 0;
    (void)unused;
    for (;;) {
        pthread_mutex_lock(&g_pool.mu);
        while (g_pool.gen == seen)
            pthread_cond_wait(&g_pool.cv_work, &g_pool.mu);
        seen = g_pool.gen;
        pthread_mutex_unlock(&g_pool.mu);
        pool_take_chunks();
    }
    return NULL;
}

void
ssw_parallel_for(int n_chunks, void (*fn)(void *, int), void *arg)
{
    int k;
    if (n_chunks <= 1 || ssw_host_threads() < 2) {
        for (k = 0; k < n_chunks; ++k)
            fn(arg, k);
        return;
    }
    pthread_mutex_lock(&g_pool.job_mu);
    if (!g_pool.started) {
        const int want = ssw_host_threads() - 1 > 31 ? 31 : ssw_host_threads() - 1;
        g_pool.started = 1;
        for (k = 0; k < want; ++k) {
            pthread_t t;
            if (pthread_create(&t, NULL, pool_worker, NULL) != 0)
                break;
            pthread_detach(t);
            ++g_pool.n_workers;
        }
    }
    pthread_mutex_lock(&g_pool.mu);
    g_pool.fn = fn;
    g_pool.arg = arg;
    g_pool.n_chunks = n_chunks;
    g_pool.next = g_pool.done = 0;
    ++g_pool.gen;
    pthread_cond_broadcast(&g_pool.cv_work);
    pthread_mutex_unlock(&g_pool.mu);
    pool_take_chunks();
    pthread_mutex_lock(&g_pool.mu);
    while (g_pool.done < g_pool.n_chunks)
        pthread_cond_wait(&g_pool.cv_done, &g_pool.mu);
    pthread_mutex_unlock(&g_pool.mu);
    pthread_mutex_unlock(&g_pool.job_mu);
}

/* Texts are independent: large batches are built by a few threads, each over a contiguous
 * range of utterances, and the pieces are concatenated (indices are local to an utterance;
 * only the offsets tables need rebasing). */
typedef struct {
    const ssw_model_t *m;
    const ssw_dict_t *d;
    const ssw_first_pass_config_t *cfg;
    int32_t n_utts;
    const int32_t *word_off;
    const char *const *words;
    ssw_fp_graphs_t *out;
    char err[512]; /* the error text is thread-local: carried back by hand */
} build_job_t;

static void
build_chunk(void *arg, int k)
{
    build_job_t *j = (build_job_t *)arg + k;
    j->out = graphs_build_serial(j->m, j->d, j->cfg, j->n_utts, j->word_off, j->words);
    j->err[0] = '\0';
    if (j->out == NULL) {
        strncpy(j->err, ssw_last_error(), sizeof(j->err) - 1);
        j->err[sizeof(j->err) - 1] = '\0';
    }
}

#define CAT(field, count_field, type)                                                        \
    do {                                                                                     \
        g->field = (type *)malloc(sizeof(type) * (size_t)(total.count_field ? total.count_field : 1)); \
        if (g->field == NULL)                                                                \
            ok = 0;                                                                          \
        else {                                                                               \
            size_t at = 0;                                                                   \
            for (t = 0; t < n_thr; ++t) {                                                    \
                if (job[t].out->count_field)                                                 \
                    memcpy(g->field + at, job[t].out->field,                                 \
                           sizeof(type) * (size_t)job[t].out->count_field);                  \
                at += (size_t)job[t].out->count_field;                                       \
            }                                                                                \
        }                                                                                    \
    } while (0)

ssw_fp_graphs_t *
ssw_fp_graphs_build(const ssw_model_t *m, const ssw_dict_t *d, const ssw_first_pass_config_t *cfg,
                    int32_t n_utts, const int32_t *word_off, const char *const *words)
{
    enum { MAX_THR = 32, MIN_PER_THR = 16 };
    build_job_t job[MAX_THR];
    ssw_fp_graphs_t *g, total;
    int n_thr = n_utts / MIN_PER_THR, t, ok = 1, u;
    const int cap = ssw_host_threads();
    if (n_thr > cap)
        n_thr = cap;
    if (n_thr > MAX_THR)
        n_thr = MAX_THR;
    if (n_thr < 2)
        return graphs_build_serial(m, d, cfg, n_utts, word_off, words);
    for (t = 0; t < n_thr; ++t) {
        const int u0 = (int)((long long)n_utts * t / n_thr), u1 = (int)((long long)n_utts * (t + 1) / n_thr);
        job[t].m = m;
        job[t].d = d;
        job[t].cfg = cfg;
        job[t].n_utts = u1 - u0;
        job[t].word_off = word_off + u0;
        job[t].words = words + word_off[u0];
        job[t].out = NULL;
    }
    ssw_parallel_for(n_thr, build_chunk, job);
    memset(&total, 0, sizeof(total));
    for (t = 0; t < n_thr; ++t) {
        if (job[t].out == NULL) {
            if (ok)
                ssw_set_error("%s", job[t].err);
            ok = 0;
        } else {
            total.n_nodes += job[t].out->n_nodes;
            total.n_leaves += job[t].out->n_leaves;
            total.n_states += job[t].out->n_states;
            total.n_in += job[t].out->n_in;
            total.n_tw += job[t].out->n_tw;
        }
    }
    g = ok ? (ssw_fp_graphs_t *)calloc(1, sizeof(*g)) : NULL;
    if (g) {
        g->n_utts = n_utts;
        g->n_nodes = total.n_nodes;
        g->n_leaves = total.n_leaves;
        g->n_states = total.n_states;
        g->n_in = total.n_in;
        g->n_tw = total.n_tw;
        g->beam = job[0].out->beam;
        g->pbeam = job[0].out->pbeam;
        g->wbeam = job[0].out->wbeam;
        g->node_off = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n_utts + 1));
        g->leaf_off = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n_utts + 1));
        g->state_off = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n_utts + 1));
        g->in_off = (int32_t *)malloc(sizeof(int32_t) * ((size_t)total.n_states + 1));
        g->tw_off = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n_utts + 1));
        g->tw_rk = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n_utts + 1));
        if (!g->node_off || !g->leaf_off || !g->state_off || !g->in_off || !g->tw_off || !g->tw_rk)
            ok = 0;
        else {
            int nb = 0, lb = 0, sb = 0, ib = 0, tb = 0;
            u = 0;
            for (t = 0; t < n_thr; ++t) {
                const ssw_fp_graphs_t *p = job[t].out;
                int k;
                for (k = 0; k < p->n_utts; ++k, ++u) {
                    g->node_off[u] = nb + p->node_off[k];
                    g->leaf_off[u] = lb + p->leaf_off[k];
                    g->state_off[u] = sb + p->state_off[k];
                    g->tw_off[u] = tb + p->tw_off[k];
                    g->tw_rk[u] = p->tw_rk[k];
                }
                for (k = 0; k < p->n_states; ++k)
                    g->in_off[sb + k] = ib + p->in_off[k];
                nb += p->n_nodes;
                lb += p->n_leaves;
                sb += p->n_states;
                ib += p->n_in;
                tb += p->n_tw;
            }
            g->tw_off[n_utts] = tb;
            g->node_off[n_utts] = nb;
            g->leaf_off[n_utts] = lb;
            g->state_off[n_utts] = sb;
            g->in_off[sb] = ib;
        }
        if (ok) {
            g->senid = (uint16_t *)malloc(sizeof(uint16_t) * 4 * (size_t)(total.n_nodes ? total.n_nodes : 1));
            if (g->senid == NULL)
                ok = 0;
            else {
                size_t at = 0;
                for (t = 0; t < n_thr; ++t) {
                    memcpy(g->senid + at, job[t].out->senid, sizeof(uint16_t) * 4 * (size_t)job[t].out->n_nodes);
                    at += 4 * (size_t)job[t].out->n_nodes;
                }
            }
            CAT(pen, n_nodes, int32_t);
            CAT(parent, n_nodes, int32_t);
            CAT(info, n_nodes, uint32_t);
            CAT(ctxt, n_nodes, uint64_t);
            CAT(leaf_ord, n_nodes, int32_t);
            CAT(leaf_wid, n_leaves, int32_t);
            CAT(leaf_to, n_leaves, int32_t);
            CAT(leaf_node, n_leaves, int32_t);
            CAT(in_leaf, n_in, int32_t);
            CAT(twin_ref, n_nodes, int32_t);
            CAT(tw, n_tw, int32_t);
        }
    }
    for (t = 0; t < n_thr; ++t)
        ssw_fp_graphs_free(job[t].out);
    if (!ok) {
        if (g)
            ssw_set_error("out of memory building the first-pass graphs");
        ssw_fp_graphs_free(g);
        return NULL;
    }
    {
        static uint64_t next_uid = 0; /* (graphs of several host threads: atomically) */
        g->uid = __atomic_add_fetch(&next_uid, 1, __ATOMIC_RELAXED);
    }
    return g;
}

void
ssw_fp_graphs_free(ssw_fp_graphs_t *g)
{
    if (g == NULL)
        return;
    free(g->node_off);
    free(g->leaf_off);
    free(g->state_off);
    free(g->senid);
    free(g->pen);
    free(g->parent);
    free(g->info);
    free(g->ctxt);
    free(g->leaf_ord);
    free(g->leaf_wid);
    free(g->leaf_to);
    free(g->leaf_node);
    free(g->in_off);
    free(g->in_leaf);
    free(g->tw);
    free(g->tw_off);
    free(g->twin_ref);
    free(g->tw_rk);
    free(g);
}

/* the graph of one text, node by node, for tests and tooling (no device needed) */
int32_t
ssw_first_pass_graph(const ssw_model_t *m, const ssw_dict_t *d, const ssw_first_pass_config_t *cfg,
                     int32_t n_words, const char *const *words, int32_t max_nodes,
                     ssw_fp_node_t *nodes, int32_t *beams)
{
    const int32_t word_off[2] = { 0, n_words };
    ssw_fp_graphs_t *g = ssw_fp_graphs_build(m, d, cfg, 1, word_off, words);
    int32_t n, i;
    if (g == NULL)
        return -1;
    n = g->n_nodes;
    if (beams) {
        beams[0] = g->beam;
        beams[1] = g->pbeam;
        beams[2] = g->wbeam;
    }
    for (i = 0; i < n && i < max_nodes; ++i) {
        ssw_fp_node_t *o = &nodes[i];
        const int lo = g->leaf_ord[i];
        memset(o, 0, sizeof(*o));
        o->senid[0] = g->senid[(size_t)i * 4];
        o->senid[1] = g->senid[(size_t)i * 4 + 1];
        o->senid[2] = g->senid[(size_t)i * 4 + 2];
        o->tmat = (int16_t)g->senid[(size_t)i * 4 + 3];
        o->pen = g->pen[i];
        o->parent = g->parent[i];
        o->flags = g->info[i] & 63u;
        o->ci_ext = (int32_t)((g->info[i] >> 8) & 0xff);
        o->state = (int32_t)(g->info[i] >> 16);
        o->to_state = lo >= 0 ? g->leaf_to[lo] : -1;
        o->wid = lo >= 0 ? g->leaf_wid[lo] : -1;
        o->ctxt = g->ctxt[i];
    }
    ssw_fp_graphs_free(g);
    return n;
}
