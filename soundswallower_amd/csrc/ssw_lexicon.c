/*
 * ssw_lexicon.c -- host C: pronunciation dictionary, triphone lookup and alignment_populate
 * (SURVEY.md section 8(f) row 1), so the batch aligner can be fed words + word windows instead
 * of pre-expanded senone sequences.
 *
 * Restates, for exactly what forced alignment needs:
 *   dict_init / dict_read (text dictionary "word PH PH ...", src/dict.c)
 *   bin_mdef_phone_id, bin_mdef_phone_id_nearest   src/bin_mdef.c:596-720
 *   dict2pid_internal, ldiph_lc, rssid, lrdiph_rc   src/dict2pid.c:262-376 (all of them reduce
 *       to ssid(bin_mdef_phone_id_nearest(b, l, r, position)))
 *   alignment_add_word, alignment_populate           src/ps_alignment.c:114-247
 */
#include "ssw_internal.h"

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* model accessors implemented next to the device model (ssw_kernels.hip) */
const ssw_host_model_t *ssw_model_host(const ssw_model_t *m);

enum { POS_INTERNAL = 0, POS_BEGIN = 1, POS_END = 2, POS_SINGLE = 3, POS_UNDEFINED = 4 };

static uint32_t
hash_str(const char *s)
{
    uint32_t h = 2166136261u;
    for (; *s; ++s)
        h = (h ^ (uint8_t)*s) * 16777619u;
    return h;
}

static int
ci_id(const ssw_host_model_t *h, const char *name)
{
    int i;
    for (i = 0; i < h->n_ciphone; ++i)
        if (strcmp(h->ciname[i], name) == 0)
            return i;
    return -1;
}

static int
dict_find(const ssw_dict_t *d, const char *w)
{
    uint32_t k;
    if (d->n_slots == 0)
        return -1;
    for (k = hash_str(w) % (uint32_t)d->n_slots; d->slot[k]; k = (k + 1) % (uint32_t)d->n_slots)
        if (strcmp(d->word[d->slot[k] - 1], w) == 0)
            return d->slot[k] - 1;
    return -1;
}

static void
dict_rehash(ssw_dict_t *d)
{
    int i;
    free(d->slot);
    d->n_slots = d->cap_words * 2 + 17;
    d->slot = (int *)calloc((size_t)d->n_slots, sizeof(int));
    for (i = 0; i < d->n_words; ++i) {
        uint32_t k = hash_str(d->word[i]) % (uint32_t)d->n_slots;
        while (d->slot[k])
            k = (k + 1) % (uint32_t)d->n_slots;
        d->slot[k] = i + 1;
    }
}

int
ssw_dict_find(const ssw_dict_t *d, const char *w)
{
    return dict_find(d, w);
}

/* dict_word2basestr (src/dict.c:400-418): length of the base of "base(...)", or -1 */
static int
base_len(const char *w)
{
    int len = (int)strlen(w), i;
    if (len > 0 && w[len - 1] == ')') {
        for (i = len - 2; i > 0 && w[i] != '('; --i)
            ;
        if (i > 0)
            return i;
    }
    return -1;
}

static int
dict_add(ssw_dict_t *d, const char *w, const int16_t *pron, int n)
{
    uint32_t k;
    int bl = base_len(w), base = -1;
    if (dict_find(d, w) >= 0)
        return 0; /* first definition wins, as dict_add_word refuses duplicates */
    if (bl > 0) { /* an alternate pronunciation needs its base word first (src/dict.c:92-107) */
        char *b = strdup(w);
        b[bl] = '\0';
        base = dict_find(d, b);
        free(b);
        if (base < 0)
            return 0;
    }
    if (d->n_words == d->cap_words) {
        d->cap_words = d->cap_words ? d->cap_words * 2 : 4096;
        d->word = (char **)realloc(d->word, sizeof(char *) * (size_t)d->cap_words);
        d->pron = (int16_t **)realloc(d->pron, sizeof(int16_t *) * (size_t)d->cap_words);
        d->pronlen = (int *)realloc(d->pronlen, sizeof(int) * (size_t)d->cap_words);
        d->alt = (int *)realloc(d->alt, sizeof(int) * (size_t)d->cap_words);
        d->base = (int *)realloc(d->base, sizeof(int) * (size_t)d->cap_words);
        dict_rehash(d);
    }
    if (base >= 0) { /* link into the base word's alt list, newest first */
        d->base[d->n_words] = base;
        d->alt[d->n_words] = d->alt[base];
        d->alt[base] = d->n_words;
    } else {
        d->base[d->n_words] = d->n_words;
        d->alt[d->n_words] = -1;
    }
    d->word[d->n_words] = strdup(w);
    d->pron[d->n_words] = (int16_t *)malloc(sizeof(int16_t) * (size_t)n);
    memcpy(d->pron[d->n_words], pron, sizeof(int16_t) * (size_t)n);
    d->pronlen[d->n_words] = n;
    k = hash_str(w) % (uint32_t)d->n_slots;
    while (d->slot[k])
        k = (k + 1) % (uint32_t)d->n_slots;
    d->slot[k] = ++d->n_words;
    return 1;
}

static int
dict_read(ssw_dict_t *d, const ssw_host_model_t *h, const char *path)
{
    FILE *fp = fopen(path, "r");
    char line[4096];
    int lineno = 0;
    if (fp == NULL) {
        ssw_set_error("%s: cannot open dictionary", path);
        return -1;
    }
    while (fgets(line, sizeof(line), fp)) {
        int16_t pron[256];
        int n = 0;
        char *save = NULL, *w, *tok;
        ++lineno;
        if (line[0] == '#' && line[1] == '#') /* comment lines, src/dict.c */
            continue;
        if (line[0] == ';' && line[1] == ';')
            continue;
        w = strtok_r(line, " \t\r\n", &save);
        if (w == NULL)
            continue;
        while ((tok = strtok_r(NULL, " \t\r\n", &save)) != NULL && n < 256) {
            int id = ci_id(h, tok);
            if (id < 0) {
                n = -1;
                break;
            }
            pron[n++] = (int16_t)id;
        }
        if (n <= 0)
            continue; /* the reference skips words with unknown phones with an error line */
        dict_add(d, w, pron, n);
    }
    fclose(fp);
    return 0;
}

ssw_dict_t *
ssw_dict_load(const ssw_model_t *m, const char *dict_path, const char *filler_path)
{
    const ssw_host_model_t *h = ssw_model_host(m);
    ssw_dict_t *d;
    if (h->ciname == NULL || h->cd_tree == NULL) {
        ssw_set_error("the dictionary needs a model loaded with its mdef");
        return NULL;
    }
    d = (ssw_dict_t *)calloc(1, sizeof(*d));
    d->cap_words = 0;
    if (dict_path && dict_read(d, h, dict_path) < 0) {
        ssw_dict_free(d);
        return NULL;
    }
    d->filler_start = d->n_words;
    if (filler_path && dict_read(d, h, filler_path) < 0) {
        ssw_dict_free(d);
        return NULL;
    }
    /* <s>, </s>, <sil> always exist (dict_init adds them when the filler file lacks them) */
    if (h->sil >= 0) {
        int16_t sil = (int16_t)h->sil;
        if (d->cap_words == 0) {
            d->cap_words = 16;
            d->word = (char **)calloc(16, sizeof(char *));
            d->pron = (int16_t **)calloc(16, sizeof(int16_t *));
            d->pronlen = (int *)calloc(16, sizeof(int));
            d->alt = (int *)calloc(16, sizeof(int));
            d->base = (int *)calloc(16, sizeof(int));
            dict_rehash(d);
        }
        dict_add(d, "<s>", &sil, 1);
        dict_add(d, "</s>", &sil, 1);
        dict_add(d, "<sil>", &sil, 1);
    }
    return d;
}

void
ssw_dict_free(ssw_dict_t *d)
{
    int i;
    if (d == NULL)
        return;
    for (i = 0; i < d->n_words; ++i) {
        free(d->word[i]);
        free(d->pron[i]);
    }
    free(d->word);
    free(d->pron);
    free(d->pronlen);
    free(d->alt);
    free(d->base);
    free(d->slot);
    free(d);
}

int32_t
ssw_dict_size(const ssw_dict_t *d)
{
    return d->n_words;
}

const char *
ssw_dict_word(const ssw_dict_t *d, int32_t wid)
{
    return (wid >= 0 && wid < d->n_words) ? d->word[wid] : NULL;
}

int32_t
ssw_dict_word_id(const ssw_dict_t *d, const char *word)
{
    return dict_find(d, word);
}

int32_t
ssw_dict_base_id(const ssw_dict_t *d, int32_t wid)
{
    return (wid >= 0 && wid < d->n_words) ? d->base[wid] : -1;
}

int32_t
ssw_dict_is_filler(const ssw_dict_t *d, int32_t wid)
{
    int b;
    if (wid < 0 || wid >= d->n_words)
        return 0;
    b = d->base[wid];
    if (strcmp(d->word[b], "<s>") == 0 || strcmp(d->word[b], "</s>") == 0)
        return 0;
    return b >= d->filler_start;
}

int32_t
ssw_dict_pron(const ssw_dict_t *d, const char *word, int32_t *ciphones, int32_t max)
{
    int w = dict_find(d, word), i;
    if (w < 0)
        return -1;
    for (i = 0; i < d->pronlen[w] && i < max; ++i)
        ciphones[i] = d->pron[w][i];
    return d->pronlen[w];
}

/* bin_mdef_phone_id: walk position -> base -> left -> right, fillers mapped to silence */
static int
phone_id(const ssw_host_model_t *h, int ci, int lc, int rc, int wpos)
{
    int ctx[4], level = 0, max = 4, i;
    const struct ssw_cd_node_s *node = h->cd_tree;
    if (lc < 0 && rc < 0 && wpos == POS_UNDEFINED)
        return ci;
    if (h->cd_tree == NULL || lc < 0 || rc < 0 || wpos == POS_UNDEFINED)
        return -1;
    ctx[0] = wpos;
    ctx[1] = ci;
    ctx[2] = (h->sil >= 0 && h->ci_filler[lc]) ? h->sil : lc;
    ctx[3] = (h->sil >= 0 && h->ci_filler[rc]) ? h->sil : rc;
    while (level < 4) {
        for (i = 0; i < max; ++i)
            if (node[i].ctx == ctx[level])
                break;
        if (i == max)
            return -1;
        if (node[i].n_down == 0)
            return node[i].down_or_pid;
        max = node[i].n_down;
        node = h->cd_tree + node[i].down_or_pid;
        ++level;
    }
    return -1;
}

static int32_t phone_id_nearest_walk(const ssw_host_model_t *h, int32_t b, int32_t l, int32_t r,
                                     int32_t pos);

/* bin_mdef_phone_id_nearest, memoised per model: the graph builder and alignment_populate ask for
 * the same few thousand triphones over and over */
int32_t
ssw_phone_id_nearest(const ssw_model_t *m, int32_t b, int32_t l, int32_t r, int32_t pos)
{
    ssw_host_model_t *h = (ssw_host_model_t *)ssw_model_host(m);
    const size_t n = (size_t)h->n_ciphone;
    int32_t *memo, p;
    if (b < 0 || b >= h->n_ciphone || l >= h->n_ciphone || r >= h->n_ciphone || pos < 0 || pos > 3) {
        ssw_set_error("phone ids out of range");
        return -1;
    }
    if (l < 0 || r < 0)
        return b;
    memo = __atomic_load_n(&h->pid_memo, __ATOMIC_ACQUIRE);
    if (memo == NULL) {
        int32_t *fresh = (int32_t *)malloc(sizeof(int32_t) * 4 * n * n * n), *expect = NULL;
        size_t i;
        if (fresh == NULL)
            return phone_id_nearest_walk(h, b, l, r, pos);
        for (i = 0; i < 4 * n * n * n; ++i)
            fresh[i] = -2;
        if (__atomic_compare_exchange_n(&h->pid_memo, &expect, fresh, 0, __ATOMIC_ACQ_REL,
                                        __ATOMIC_ACQUIRE))
            memo = fresh;
        else { /* another thread installed its table first */
            free(fresh);
            memo = expect;
        }
    }
    {
        int32_t *slot = &memo[(((size_t)pos * n + (size_t)b) * n + (size_t)l) * n + (size_t)r];
        p = __atomic_load_n(slot, __ATOMIC_RELAXED);
        if (p == -2) {
            p = phone_id_nearest_walk(h, b, l, r, pos);
            __atomic_store_n(slot, p, __ATOMIC_RELAXED);
        }
    }
    return p;
}

static int32_t
phone_id_nearest_walk(const ssw_host_model_t *h, int32_t b, int32_t l, int32_t r, int32_t pos)
{
    int p, t;
    if ((p = phone_id(h, b, l, r, pos)) >= 0)
        return p;
    for (t = 0; t < 4; ++t)
        if (t != pos && (p = phone_id(h, b, l, r, t)) >= 0)
            return p;
    if (h->sil >= 0) {
        int nl = l, nr = r;
        if (h->ci_filler[l] || pos == POS_BEGIN || pos == POS_SINGLE)
            nl = h->sil;
        if (h->ci_filler[r] || pos == POS_END || pos == POS_SINGLE)
            nr = h->sil;
        if (nl != l || nr != r) {
            if ((p = phone_id(h, b, nl, nr, pos)) >= 0)
                return p;
            for (t = 0; t < 4; ++t)
                if (t != pos && (p = phone_id(h, b, nl, nr, t)) >= 0)
                    return p;
        }
    }
    return b;
}

/* alignment_add_word x n + alignment_populate: words with their windows -> phone rows */
int32_t
ssw_alignment_populate(const ssw_model_t *m, const ssw_dict_t *d, int32_t n_words,
                       const char *const *words, const int32_t *start, const int32_t *duration,
                       int32_t max_phones, int32_t *ssid, int32_t *tmatid, int32_t *cipid,
                       int32_t *parent, int32_t *ph_start, int32_t *ph_duration)
{
    const ssw_host_model_t *h = ssw_model_host(m);
    int n = 0, i, j, lc;
    if (h->sil < 0) {
        ssw_set_error("model has no SIL phone");
        return -1;
    }
    lc = h->sil;
    for (i = 0; i < n_words; ++i) {
        int w = dict_find(d, words[i]), len, rc;
        const int16_t *p;
        if (w < 0) {
            ssw_set_error("word '%s' is not in the dictionary", words[i]);
            return -1;
        }
        p = d->pron[w];
        len = d->pronlen[w];
        if (i < n_words - 1) {
            int nw = dict_find(d, words[i + 1]);
            if (nw < 0) {
                ssw_set_error("word '%s' is not in the dictionary", words[i + 1]);
                return -1;
            }
            rc = d->pron[nw][0];
        } else
            rc = h->sil;
        if (n + len > max_phones || (n + len) * h->n_emit_state > 0xffff) {
            ssw_set_error("alignment of %d+ phones exceeds the limit", n + len); /* uint16 n_ent */
            return -1;
        }
        for (j = 0; j < len; ++j) {
            int pid;
            if (len == 1)
                pid = ssw_phone_id_nearest(m, p[0], lc, rc, POS_SINGLE);
            else if (j == 0)
                pid = ssw_phone_id_nearest(m, p[0], lc, p[1], POS_BEGIN);
            else if (j == len - 1)
                pid = ssw_phone_id_nearest(m, p[j], p[j - 1], rc, POS_END);
            else
                pid = ssw_phone_id_nearest(m, p[j], p[j - 1], p[j + 1], POS_INTERNAL);
            if (pid < 0 || pid >= h->n_phone)
                return -1;
            ssid[n] = h->phone_ssid[pid];
            tmatid[n] = h->phone_tmat[p[j]]; /* bin_mdef_pid2tmatid(mdef, cipid) */
            if (cipid)
                cipid[n] = p[j];
            if (parent)
                parent[n] = i;
            if (ph_start)
                ph_start[n] = start ? start[i] : 0;
            if (ph_duration)
                ph_duration[n] = duration ? duration[i] : 0;
            ++n;
        }
        lc = p[len - 1];
    }
    return n;
}

const char *
ssw_ciphone_name(const ssw_model_t *m, int32_t ci)
{
    const ssw_host_model_t *h = ssw_model_host(m);
    if (h->ciname == NULL || ci < 0 || ci >= h->n_ciphone)
        return NULL;
    return h->ciname[ci];
}

/* ------------------------------------------------------------------------------------ */
/* decoder_result_json at align_level >= 1 (src/decoder.c:1339-1593): one line,           */
/*   {"b":..,"d":..,"p":..,"t":"<hyp>","w":[ word objects ]}\n                            */
/* every object {"b":%.3f,"d":%.3f,"p":%.3f,"t":"%s"} with b = utt_start + start / frate,   */
/* d = duration / frate, p = logmath_exp(score) = base^score (src/logmath.c:292-295); words */
/* carry their phones in "w", phones their states (named by senone id) when state entries   */
/* are given (align_level 2).                                                              */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    char *out;
    size_t cap, len;
} json_buf_t;

static void
jb_put(json_buf_t *b, const char *s, size_t n)
{
    if (b->out != NULL && b->len < b->cap) {
        size_t room = b->cap - b->len - 1;
        memcpy(b->out + b->len, s, n < room ? n : room);
    }
    b->len += n;
}

static void
jb_obj(json_buf_t *b, double base, double utt_start, int frate, const ssw_align_entry_t *e,
       const char *name)
{
    char tmp[96];
    const double st = utt_start + (double)e->start / frate;
    const double dur = (double)e->duration / frate;
    const double prob = pow(base, (double)e->score);
    int n = snprintf(tmp, sizeof(tmp), "{\"b\":%.3f,\"d\":%.3f,\"p\":%.3f,\"t\":\"", st, dur, prob);
    jb_put(b, tmp, (size_t)n);
    jb_put(b, name, strlen(name));
    jb_put(b, "\"", 1);
}

int32_t
ssw_alignment_json(const ssw_model_t *m, const char *hyp, int32_t hyp_logprob, double utt_start,
                   int32_t frate, int32_t n_frames, int32_t n_words, const char *const *words,
                   const ssw_align_entry_t *word_al, int32_t n_phones, const int32_t *cipid,
                   const int32_t *parent, const ssw_align_entry_t *phone_al,
                   const uint16_t *state_senid, const ssw_align_entry_t *state_al, char *out,
                   int32_t out_len)
{
    const ssw_host_model_t *h = ssw_model_host(m);
    const double base = h->cfg.logbase;
    json_buf_t b = { out, out_len > 0 ? (size_t)out_len : 0, 0 };
    ssw_align_entry_t top;
    int32_t w, p = 0;

    if (frate <= 0 || n_words < 0 || n_phones < 0 || (n_words > 0 && (words == NULL || word_al == NULL))
        || (n_phones > 0 && (cipid == NULL || parent == NULL || phone_al == NULL))
        || (state_senid == NULL) != (state_al == NULL)) {
        ssw_set_error("ssw_alignment_json: inconsistent arguments");
        return -1;
    }
    top.start = 0;
    top.duration = n_frames;
    top.score = hyp_logprob;
    jb_obj(&b, base, utt_start, frate, &top, hyp ? hyp : "");
    jb_put(&b, ",\"w\":[", 6);
    for (w = 0; w < n_words; ++w) {
        int first = 1;
        if (w > 0)
            jb_put(&b, ",", 1);
        jb_obj(&b, base, utt_start, frate, &word_al[w], words[w] ? words[w] : "");
        jb_put(&b, ",\"w\":[", 6);
        for (; p < n_phones && parent[p] == w; ++p) {
            const char *nm = ssw_ciphone_name(m, cipid[p]);
            if (!first)
                jb_put(&b, ",", 1);
            first = 0;
            jb_obj(&b, base, utt_start, frate, &phone_al[p], nm ? nm : "");
            if (state_al != NULL) {
                int k;
                jb_put(&b, ",\"w\":[", 6);
                for (k = 0; k < h->n_emit_state; ++k) {
                    char nm2[16];
                    snprintf(nm2, sizeof(nm2), "%u", (unsigned)state_senid[p * h->n_emit_state + k]);
                    if (k > 0)
                        jb_put(&b, ",", 1);
                    jb_obj(&b, base, utt_start, frate, &state_al[p * h->n_emit_state + k], nm2);
                    jb_put(&b, "}", 1);
                }
                jb_put(&b, "]", 1);
            }
            jb_put(&b, "}", 1);
        }
        jb_put(&b, "]}", 2);
    }
    if (p != n_phones) {
        ssw_set_error("ssw_alignment_json: phones are not grouped by ascending parent word");
        return -1;
    }
    jb_put(&b, "]}\n", 3);
    if (b.out != NULL && b.cap > 0)
        b.out[b.len < b.cap ? b.len : b.cap - 1] = '\0';
    return (int32_t)b.len;
}
