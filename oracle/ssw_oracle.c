/*
 * ssw_oracle.c -- CPU restatement ("oracle") of SoundSwallower's acoustic hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see ssw_oracle.h).  Plain scalar C, written to be obviously
 * faithful to the reference's arithmetic, not fast.  Build with -ffp-contract=off and without
 * -march=native / -ffast-math so every float op rounds once, as in the reference's x86-64 build.
 *
 * Citations are file:line relative to /root/reference.
 */
#include "ssw_oracle.h"

#include <ctype.h>
#include <limits.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SENSCR_SHIFT 10                 /* include/soundswallower/hmm.h:70 */
#define WORST_SCORE ((int32_t)0xE0000000) /* hmm.h:81 */
#define TMAT_WORST_SCORE (-255)         /* hmm.h:87 */
#define MAX_NEG_INT32 ((int32_t)0x80000000)
#define MAX_NEG_MIXW 159                /* tied_mgau_common.h:81 */
#define MAX_NEG_ASCR 96                 /* tied_mgau_common.h:82 */

static char g_err[512];

static void
set_err(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char *
orc_last_error(void)
{
    return g_err;
}

/* ================================================================================== */
/* logmath: src/logmath.c                                                              */
/* ================================================================================== */

static uint32_t
tab_get(const orc_logmath_t *lm, uint32_t i)
{
    switch (lm->width) {
    case 1: return ((const uint8_t *)lm->table)[i];
    case 2: return ((const uint16_t *)lm->table)[i];
    default: return ((const uint32_t *)lm->table)[i];
    }
}

static void
tab_set(orc_logmath_t *lm, uint32_t i, uint32_t v)
{
    switch (lm->width) {
    case 1: ((uint8_t *)lm->table)[i] = (uint8_t)v; break;
    case 2: ((uint16_t *)lm->table)[i] = (uint16_t)v; break;
    default: ((uint32_t *)lm->table)[i] = v; break;
    }
}

/* logmath_init, src/logmath.c:60-164 */
orc_logmath_t *
orc_logmath_init(double base, int shift, int use_table)
{
    orc_logmath_t *lm;
    uint32_t maxyx, i;
    double byx;

    if (base <= 1.0) {
        set_err("logmath: base must be > 1.0");
        return NULL;
    }
    lm = calloc(1, sizeof(*lm));
    lm->base = base;
    lm->log_of_base = log(base);
    lm->log10_of_base = log10(base);
    lm->inv_log_of_base = 1.0 / lm->log_of_base;
    lm->inv_log10_of_base = 1.0 / lm->log10_of_base;
    lm->shift = shift;
    lm->zero = MAX_NEG_INT32 >> (shift + 2); /* :84 */
    if (!use_table)
        return lm;

    maxyx = (uint32_t)(log(2.0) / log(base) + 0.5) >> shift; /* :90 */
    if (maxyx < 256)
        lm->width = 1;
    else if (maxyx < 65536)
        lm->width = 2;
    else
        lm->width = 4;

    /* size pass, :101-119 */
    byx = 1.0;
    for (i = 0;; ++i) {
        double lobyx = log(1.0 + byx) * lm->inv_log_of_base;
        int32_t k = (int32_t)(lobyx + 0.5 * (1 << shift)) >> shift;
        if (k <= 0)
            break;
        byx /= base;
    }
    i >>= shift;
    if (i < 255)
        i = 255;
    lm->table = calloc(i + 1, lm->width);
    lm->table_size = i + 1;

    /* fill pass, :124-161: only the first (largest) value landing in a slot is kept */
    byx = 1.0;
    for (i = 0;; ++i) {
        double lobyx = log(1.0 + byx) * lm->inv_log_of_base;
        int32_t k = (int32_t)(lobyx + 0.5 * (1 << shift)) >> shift;
        uint32_t prev = tab_get(lm, i >> shift);
        if (prev == 0)
            tab_set(lm, i >> shift, (uint32_t)k);
        if (k <= 0)
            break;
        byx /= base;
    }
    return lm;
}

void
orc_logmath_free(orc_logmath_t *lm)
{
    if (lm == NULL)
        return;
    free(lm->table);
    free(lm);
}

/* logmath_log, src/logmath.c:282-289 */
int
orc_logmath_log(const orc_logmath_t *lm, double p)
{
    if (p <= 0)
        return lm->zero;
    return (int)(log(p) * lm->inv_log_of_base) >> lm->shift;
}

/* logmath_ln_to_log, src/logmath.c:297-301 */
int
orc_logmath_ln_to_log(const orc_logmath_t *lm, double log_p)
{
    return (int)(log_p * lm->inv_log_of_base) >> lm->shift;
}

/* logmath_exp, src/logmath.c:291-295 */
double
orc_logmath_exp(const orc_logmath_t *lm, int logb_p)
{
    return pow(lm->base, (double)(logb_p << lm->shift));
}

/* logmath_add, src/logmath.c:228-272 */
int
orc_logmath_add(const orc_logmath_t *lm, int x, int y)
{
    int d, r;

    if (x <= lm->zero)
        return y;
    if (y <= lm->zero)
        return x;
    if (lm->table == NULL) /* logmath_add_exact, :274-280 */
        return orc_logmath_log(lm, orc_logmath_exp(lm, x) + orc_logmath_exp(lm, y));
    if (x > y) {
        d = x - y;
        r = x;
    } else {
        d = y - x;
        r = y;
    }
    if (d < 0)
        return r;
    if ((size_t)d >= lm->table_size)
        return r;
    return r + (int)tab_get(lm, (uint32_t)d);
}

uint32_t
orc_logmath_table(const orc_logmath_t *lm, uint32_t *out, uint32_t max)
{
    uint32_t i;
    for (i = 0; i < lm->table_size && i < max; ++i)
        out[i] = tab_get(lm, i);
    return lm->table_size;
}

/* ================================================================================== */
/* s3 binary files: src/s3file.c                                                       */
/* ================================================================================== */

typedef struct s3buf_s {
    uint8_t *buf;
    size_t len, pos;
    int do_swap, do_chksum;
    uint32_t chksum;
} s3buf_t;

static int
s3_open(s3buf_t *s, const char *path)
{
    FILE *fp;
    long n;

    memset(s, 0, sizeof(*s));
    if ((fp = fopen(path, "rb")) == NULL) {
        set_err("cannot open %s", path);
        return -1;
    }
    fseek(fp, 0, SEEK_END);
    n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    s->buf = malloc(n > 0 ? (size_t)n : 1);
    s->len = (size_t)n;
    if (fread(s->buf, 1, s->len, fp) != s->len) {
        fclose(fp);
        set_err("short read on %s", path);
        return -1;
    }
    fclose(fp);
    return 0;
}

static void
s3_close(s3buf_t *s)
{
    free(s->buf);
    s->buf = NULL;
}

static uint32_t
bswap32(uint32_t v)
{
    return (v >> 24) | ((v >> 8) & 0xff00) | ((v << 8) & 0xff0000) | (v << 24);
}

static uint16_t
bswap16(uint16_t v)
{
    return (uint16_t)((v >> 8) | (v << 8));
}

/* s3file_get, src/s3file.c:424-445 with chksum_accum :366-394 (4-byte and 1-byte elements
 * are all this path reads) */
static size_t
s3_get(void *out, size_t el_sz, size_t n_el, s3buf_t *s)
{
    size_t avail = s->len - s->pos, i;

    if (avail < el_sz * n_el)
        n_el = avail / el_sz;
    if (n_el == 0)
        return 0;
    memcpy(out, s->buf + s->pos, el_sz * n_el);
    s->pos += el_sz * n_el;
    if (s->do_swap && el_sz == 4)
        for (i = 0; i < n_el; ++i)
            ((uint32_t *)out)[i] = bswap32(((uint32_t *)out)[i]);
    if (s->do_swap && el_sz == 2)
        for (i = 0; i < n_el; ++i)
            ((uint16_t *)out)[i] = bswap16(((uint16_t *)out)[i]);
    if (s->do_chksum) {
        uint32_t sum = s->chksum;
        if (el_sz == 4)
            for (i = 0; i < n_el; ++i)
                sum = (sum << 20 | sum >> 12) + ((uint32_t *)out)[i];
        else if (el_sz == 2)
            for (i = 0; i < n_el; ++i)
                sum = (sum << 10 | sum >> 22) + ((uint16_t *)out)[i];
        else
            for (i = 0; i < n_el; ++i)
                sum = (sum << 5 | sum >> 27) + ((uint8_t *)out)[i];
        s->chksum = sum;
    }
    return n_el;
}

/* s3file_parse_header, src/s3file.c:210-327: "s3\n", name/value lines, "endhdr", then the
 * byte-order magic 0x11223344.  Any line whose first word is "chksum0" turns checksumming on
 * (:289-290), whatever its value. */
static int
s3_parse_header(s3buf_t *s)
{
    int do_chksum = 0;
    uint32_t magic;

    if (s->len < 3 || memcmp(s->buf, "s3\n", 3) != 0) {
        set_err("not an s3 file (old headerless format unsupported by the oracle)");
        return -1;
    }
    s->pos = 3;
    for (;;) {
        size_t ls = s->pos, le, ws, we;
        if (ls >= s->len) {
            set_err("premature EOF in s3 header");
            return -1;
        }
        for (le = ls; le < s->len && s->buf[le] != '\n'; ++le)
            ;
        s->pos = (le < s->len) ? le + 1 : le;
        for (ws = ls; ws < le && isspace(s->buf[ws]); ++ws)
            ;
        if (ws == le) {
            set_err("missing header word");
            return -1;
        }
        for (we = ws; we < le && !isspace(s->buf[we]); ++we)
            ;
        if (s->buf[ws] == '#')
            continue;
        /* reference compares strncmp(word, "endhdr", wordlen): a prefix matches too */
        if (we - ws <= 6 && strncmp((char *)s->buf + ws, "endhdr", we - ws) == 0)
            break;
        if (we - ws == 7 && memcmp(s->buf + ws, "chksum0", 7) == 0)
            do_chksum = 1;
    }
    /* swap_check, :126-150 */
    if (s3_get(&magic, 4, 1, s) != 1) {
        set_err("cannot read byte-order magic");
        return -1;
    }
    if (magic != 0x11223344u) {
        if (bswap32(magic) != 0x11223344u) {
            set_err("bad byte-order magic %08x", magic);
            return -1;
        }
        s->do_swap = 1;
    }
    s->do_chksum = do_chksum;
    return 0;
}

/* s3file_verify_chksum, src/s3file.c:551-570 */
static int
s3_verify_chksum(s3buf_t *s)
{
    uint32_t file_sum, sum;

    if (!s->do_chksum)
        return 0;
    s->do_chksum = 0;
    sum = s->chksum;
    if (s3_get(&file_sum, 4, 1, s) != 1) {
        set_err("cannot read checksum");
        return -1;
    }
    if (file_sum != sum) {
        set_err("checksum error: file %08x computed %08x", file_sum, sum);
        return -1;
    }
    return 0;
}

/* ================================================================================== */
/* model                                                                               */
/* ================================================================================== */

typedef struct topn_s {
    int32_t cw, score;
} topn_t; /* ptm_topn_t, ptm_mgau.h:62-65 */

struct orc_model_s {
    orc_config_t cfg;
    orc_logmath_t *lmath;    /* logmath_init(base, 0, TRUE), src/decoder.c:267 */
    orc_logmath_t *lmath_8b; /* logmath_init(base, 10, TRUE), src/ptm_mgau.c:735 */

    /* gauden_t, ms_gauden.h:83-92 */
    int32_t n_cb, n_feat, n_density, veclen_total, n_floored;
    int32_t *veclen, *featoff; /* featoff[f] = offset of stream f inside a 39-float frame */
    float *mean, *var, *det;   /* flat, file order */
    size_t *cbf_off;           /* [cb][feat] -> offset (floats) of density 0 in mean/var */

    /* mdef */
    int32_t n_ciphone, n_phone, n_emit_state, n_ci_sen, n_sen, n_tmat, n_sseq, sil, n_cd_tree;
    uint16_t *sseq;
    int16_t *sen2cimap;
    int32_t *phone_ssid, *phone_tmat;
    uint8_t *cd_tree;     /* n_cd_tree x {int16 ctx, int16 n_down, int32 pid|down}, host order */
    uint8_t *ci_filler;   /* [n_ciphone] mdef_entry_t.info.ci.filler */
    char *ciname;         /* NUL-separated CI phone names */
    size_t ciname_len;

    /* tmat */
    uint8_t *tp; /* [n_tmat][n_emit][n_emit+1] */
    int32_t tp_n_tmat, tp_n_state;

    /* PTM */
    uint8_t *ptm_mixw; /* [feat][density][n_sen] */
    uint8_t *mixw_cb;  /* 4-bit cluster codebook or NULL */
    int mixw_bits;
    size_t mixw_row;   /* bytes per (feat,density) row */
    uint8_t *sen2cb;
    topn_t *hist[2]; /* [n_cb][n_feat][topn] each; n_fast_hist = 2, ptm_mgau.c:804 */
    uint8_t *cb_active[2];
    int frame_idx; /* mgau_t.frame_idx, acmod.h:110 */

    /* ms */
    uint8_t *ms_pdf; /* [sen][feat][density], n_gauden > 1 layout, ms_senone.c:145-148 */
};

void
orc_config_defaults(orc_config_t *cfg)
{
    /* include/soundswallower/config_defs.h defaults quoted in SURVEY.md section 5 */
    cfg->logbase = 1.0001;
    cfg->varfloor = 1e-4;
    cfg->mixwfloor = 1e-7;
    cfg->tmatfloor = 1e-4;
    cfg->topn = 4;
    cfg->ds = 1;
    cfg->aw = 1;
}

/* gauden_param_read, src/ms_gauden.c:105-202 */
static float *
gauden_param_read(const char *path, int32_t *n_cb, int32_t *n_feat, int32_t *n_density,
                  int32_t **veclen)
{
    s3buf_t s;
    int32_t n, blk = 0, i;
    float *buf = NULL;

    *veclen = NULL;
    if (s3_open(&s, path) < 0)
        return NULL;
    if (s3_parse_header(&s) < 0)
        goto fail;
    if (s3_get(n_cb, 4, 1, &s) != 1 || s3_get(n_feat, 4, 1, &s) != 1
        || s3_get(n_density, 4, 1, &s) != 1) {
        set_err("%s: truncated dimensions", path);
        goto fail;
    }
    if (*n_feat <= 0 || *n_feat > 64) {
        set_err("%s: bad n_feat %d", path, *n_feat);
        goto fail;
    }
    *veclen = calloc(*n_feat, sizeof(int32_t));
    if (s3_get(*veclen, 4, *n_feat, &s) != (size_t)*n_feat) {
        set_err("%s: truncated veclen", path);
        goto fail;
    }
    for (i = 0; i < *n_feat; ++i)
        blk += (*veclen)[i];
    if (s3_get(&n, 4, 1, &s) != 1 || n != *n_cb * *n_density * blk) {
        set_err("%s: parameter count mismatch", path);
        goto fail;
    }
    buf = malloc(sizeof(float) * (size_t)n);
    if (s3_get(buf, 4, n, &s) != (size_t)n) {
        set_err("%s: truncated data", path);
        goto fail;
    }
    if (s3_verify_chksum(&s) < 0)
        goto fail;
    s3_close(&s);
    return buf;
fail:
    free(buf);
    free(*veclen);
    *veclen = NULL;
    s3_close(&s);
    return NULL;
}

/* gauden_init_s3file + gauden_dist_precompute, src/ms_gauden.c:260-301, 217-258 */
static int
load_gauden(orc_model_t *m, const char *means, const char *vars)
{
    int32_t ncb2, nf2, nd2, *vl2 = NULL, c, f, d, i;
    size_t off;

    m->mean = gauden_param_read(means, &m->n_cb, &m->n_feat, &m->n_density, &m->veclen);
    if (m->mean == NULL)
        return -1;
    m->var = gauden_param_read(vars, &ncb2, &nf2, &nd2, &vl2);
    if (m->var == NULL)
        return -1;
    if (ncb2 != m->n_cb || nf2 != m->n_feat || nd2 != m->n_density) {
        free(vl2);
        set_err("means/variances dimensions differ");
        return -1;
    }
    for (i = 0; i < m->n_feat; ++i)
        if (vl2[i] != m->veclen[i]) {
            free(vl2);
            set_err("means/variances feature lengths differ");
            return -1;
        }
    free(vl2);

    m->featoff = calloc(m->n_feat, sizeof(int32_t));
    for (f = 0, m->veclen_total = 0; f < m->n_feat; ++f) {
        m->featoff[f] = m->veclen_total;
        m->veclen_total += m->veclen[f];
    }
    m->cbf_off = calloc((size_t)m->n_cb * m->n_feat, sizeof(size_t));
    for (c = 0, off = 0; c < m->n_cb; ++c)
        for (f = 0; f < m->n_feat; ++f) {
            m->cbf_off[c * m->n_feat + f] = off;
            off += (size_t)m->n_density * m->veclen[f];
        }

    /* precompute: per dimension, floor the variance, accumulate det in FLOAT, replace var */
    m->det = calloc((size_t)m->n_cb * m->n_feat * m->n_density, sizeof(float));
    m->n_floored = 0;
    for (c = 0; c < m->n_cb; ++c)
        for (f = 0; f < m->n_feat; ++f) {
            int flen = m->veclen[f];
            float *detp = m->det + ((size_t)c * m->n_feat + f) * m->n_density;
            for (d = 0; d < m->n_density; ++d) {
                float *varp = m->var + m->cbf_off[c * m->n_feat + f] + (size_t)d * flen;
                detp[d] = 0;
                for (i = 0; i < flen; ++i) {
                    if (varp[i] < (float)m->cfg.varfloor) { /* :241, float32 varfloor */
                        varp[i] = (float)m->cfg.varfloor;
                        ++m->n_floored;
                    }
                    /* :245-246  *detp += (mfcc_t)logmath_log(1/sqrt(var*2*pi)) */
                    detp[d] += (float)orc_logmath_log(m->lmath,
                                                      1.0 / sqrt(varp[i] * 2.0 * M_PI));
                    /* :248-249 */
                    varp[i] = (float)orc_logmath_ln_to_log(m->lmath, 1.0 / (varp[i] * 2.0));
                }
            }
        }
    return 0;
}

/* bin_mdef_read_s3file, src/bin_mdef.c:333-540 (the parts the path needs) */
static int
load_mdef(orc_model_t *m, const char *path)
{
    s3buf_t s;
    int32_t val, hdr[10], i, sseq_size;
    size_t names_start, p;
    uint8_t *ent;

    if (s3_open(&s, path) < 0)
        return -1;
    if (s3_get(&val, 4, 1, &s) != 1)
        goto trunc;
    if ((uint32_t)val == 0x424d4446u)
        s.do_swap = 1;
    else if ((uint32_t)val != 0x46444d42u) {
        set_err("%s: not a binary mdef", path);
        goto fail;
    }
    if (s3_get(&val, 4, 1, &s) != 1)
        goto trunc;
    if (val > 1) {
        set_err("%s: mdef format version %d too new", path, val);
        goto fail;
    }
    if (s3_get(&val, 4, 1, &s) != 1)
        goto trunc;
    s.pos += (size_t)val; /* skip format descriptor */
    if (s.pos > s.len)
        goto trunc;
    if (s3_get(hdr, 4, 10, &s) != 10)
        goto trunc;
    m->n_ciphone = hdr[0];
    m->n_phone = hdr[1];
    m->n_emit_state = hdr[2];
    m->n_ci_sen = hdr[3];
    m->n_sen = hdr[4];
    m->n_tmat = hdr[5];
    m->n_sseq = hdr[6];
    m->n_cd_tree = hdr[8];
    m->sil = hdr[9];
    if (m->n_emit_state <= 0) {
        set_err("%s: heterogeneous topologies unsupported by the oracle", path);
        goto fail;
    }
    /* NUL-separated CI names, padded to 4 (:405-416) */
    names_start = s.pos;
    p = s.pos;
    m->sil = -1;
    for (i = 0; i < m->n_ciphone; ++i) {
        size_t l = strlen((char *)s.buf + p);
        if (strcmp((char *)s.buf + p, "SIL") == 0) /* bin_mdef_ciphone_id(m,"SIL"), :520 */
            m->sil = i;
        p += l + 1;
        if (p > s.len)
            goto trunc;
    }
    m->ciname_len = p - names_start;
    m->ciname = malloc(m->ciname_len + 1);
    memcpy(m->ciname, s.buf + names_start, m->ciname_len);
    p = names_start + (((p - names_start) + 3) & ~(size_t)3);
    if (p + (size_t)m->n_cd_tree * 8 > s.len)
        goto trunc;
    m->cd_tree = malloc((size_t)m->n_cd_tree * 8 + 8);
    memcpy(m->cd_tree, s.buf + p, (size_t)m->n_cd_tree * 8);
    if (s.do_swap) /* bin_mdef.c:425-431 */
        for (i = 0; i < m->n_cd_tree; ++i) {
            uint16_t a, b;
            uint32_t c;
            memcpy(&a, m->cd_tree + (size_t)i * 8, 2);
            memcpy(&b, m->cd_tree + (size_t)i * 8 + 2, 2);
            memcpy(&c, m->cd_tree + (size_t)i * 8 + 4, 4);
            a = bswap16(a);
            b = bswap16(b);
            c = bswap32(c);
            memcpy(m->cd_tree + (size_t)i * 8, &a, 2);
            memcpy(m->cd_tree + (size_t)i * 8 + 2, &b, 2);
            memcpy(m->cd_tree + (size_t)i * 8 + 4, &c, 4);
        }
    p += (size_t)m->n_cd_tree * 8; /* cd_tree_t is 8 bytes */
    if (p + (size_t)m->n_phone * 12 + 4 > s.len)
        goto trunc;
    /* mdef_entry_t: packed {int32 ssid; int32 tmat; uint8 info[4]}, bin_mdef.h:75-93 */
    m->phone_ssid = calloc(m->n_phone, sizeof(int32_t));
    m->phone_tmat = calloc(m->n_phone, sizeof(int32_t));
    ent = s.buf + p;
    m->ci_filler = calloc((size_t)m->n_ciphone, 1);
    for (i = 0; i < m->n_ciphone; ++i)
        m->ci_filler[i] = ent[(size_t)i * 12 + 8];
    for (i = 0; i < m->n_phone; ++i) {
        uint32_t a, b;
        memcpy(&a, ent + (size_t)i * 12, 4);
        memcpy(&b, ent + (size_t)i * 12 + 4, 4);
        if (s.do_swap) {
            a = bswap32(a);
            b = bswap32(b);
        }
        m->phone_ssid[i] = (int32_t)a;
        m->phone_tmat[i] = (int32_t)b;
    }
    p += (size_t)m->n_phone * 12;
    memcpy(&sseq_size, s.buf + p, 4);
    if (s.do_swap)
        sseq_size = (int32_t)bswap32((uint32_t)sseq_size);
    p += 4;
    if (p + (size_t)sseq_size * 2 > s.len || sseq_size < m->n_sseq * m->n_emit_state)
        goto trunc;
    m->sseq = malloc(sizeof(uint16_t) * (size_t)sseq_size);
    memcpy(m->sseq, s.buf + p, sizeof(uint16_t) * (size_t)sseq_size);
    if (s.do_swap)
        for (i = 0; i < sseq_size; ++i)
            m->sseq[i] = bswap16(m->sseq[i]);

    /* sen2cimap: first CI phone seen for each senone, scanning all phones (:488-517) */
    m->sen2cimap = malloc(sizeof(int16_t) * (size_t)m->n_sen);
    for (i = 0; i < m->n_sen; ++i)
        m->sen2cimap[i] = -1;
    for (i = 0; i < m->n_phone; ++i) {
        int j, ci;
        ci = (i < m->n_ciphone) ? i : ent[(size_t)i * 12 + 9]; /* info.cd.ctx[0] */
        for (j = 0; j < m->n_emit_state; ++j) {
            int sen = m->sseq[(size_t)m->phone_ssid[i] * m->n_emit_state + j];
            if (sen < m->n_sen && m->sen2cimap[sen] == -1)
                m->sen2cimap[sen] = (int16_t)ci;
        }
    }
    s3_close(&s);
    return 0;
trunc:
    set_err("%s: truncated mdef", path);
fail:
    s3_close(&s);
    return -1;
}

/* vector_sum_norm / vector_floor / vector_nz_floor, src/vector.c:86-123 */
static double
vec_sum_norm(float *v, int n)
{
    double sum = 0.0, f;
    int i;
    for (i = 0; i < n; ++i)
        sum += v[i];
    if (sum != 0.0) {
        f = 1.0 / sum;
        for (i = 0; i < n; ++i)
            v[i] = (float)(v[i] * f);
    }
    return sum;
}

static void
vec_floor(float *v, int n, double flr)
{
    int i;
    for (i = 0; i < n; ++i)
        if (v[i] < flr)
            v[i] = (float)flr;
}

static void
vec_nz_floor(float *v, int n, double flr)
{
    int i;
    for (i = 0; i < n; ++i)
        if (v[i] != 0.0 && v[i] < flr)
            v[i] = (float)flr;
}

/* tmat_init_s3file, src/tmat.c:125-227 */
static int
load_tmat(orc_model_t *m, const char *path)
{
    s3buf_t s;
    int32_t n_tmat, n_src, n_dst, n, i, j, k;
    float *row = NULL;

    if (s3_open(&s, path) < 0)
        return -1;
    if (s3_parse_header(&s) < 0)
        goto fail;
    if (s3_get(&n_tmat, 4, 1, &s) != 1 || s3_get(&n_src, 4, 1, &s) != 1
        || s3_get(&n_dst, 4, 1, &s) != 1 || s3_get(&n, 4, 1, &s) != 1) {
        set_err("%s: truncated tmat header", path);
        goto fail;
    }
    if (n_dst != n_src + 1 || n != n_tmat * n_src * n_dst) {
        set_err("%s: unsupported tmat shape %d x %d x %d", path, n_tmat, n_src, n_dst);
        goto fail;
    }
    m->tp_n_tmat = n_tmat;
    m->tp_n_state = n_src;
    m->tp = calloc((size_t)n, 1);
    row = malloc(sizeof(float) * (size_t)n_dst);
    for (i = 0; i < n_tmat; ++i)
        for (j = 0; j < n_src; ++j) {
            if (s3_get(row, 4, n_dst, &s) != (size_t)n_dst) {
                set_err("%s: truncated tmat %d", path, i);
                goto fail;
            }
            vec_sum_norm(row, n_dst);
            vec_nz_floor(row, n_dst, m->cfg.tmatfloor);
            vec_sum_norm(row, n_dst);
            for (k = 0; k < n_dst; ++k) {
                /* :206  ltp = -logmath_log(lmath, p) >> SENSCR_SHIFT; clamp 255 */
                int ltp = -orc_logmath_log(m->lmath, row[k]) >> SENSCR_SHIFT;
                if (ltp > 255)
                    ltp = 255;
                m->tp[((size_t)i * n_src + j) * n_dst + k] = (uint8_t)ltp;
            }
        }
    if (s3_verify_chksum(&s) < 0)
        goto fail;
    free(row);
    s3_close(&s);
    return 0;
fail:
    free(row);
    s3_close(&s);
    return -1;
}

/* read_sendump, src/ptm_mgau.c:456-609.  The sendump has no s3 header. */
static int
load_sendump(orc_model_t *m, const char *path)
{
    s3buf_t s;
    int32_t n, r, c;
    int n_clust = 0, n_feat = m->n_feat, n_density = m->n_density, n_sen = m->n_sen, n_bits = 8;
    size_t step, need;

    if (s3_open(&s, path) < 0)
        return -1;
    if (s3_get(&n, 4, 1, &s) != 1)
        goto trunc;
    if (n < 1 || n > 999) { /* :474-481 */
        n = (int32_t)bswap32((uint32_t)n);
        if (n < 1 || n > 999) {
            set_err("%s: title length out of range", path);
            goto fail;
        }
        s.do_swap = 1;
    }
    if (s.pos + (size_t)n > s.len || s.buf[s.pos + n - 1] != '\0')
        goto trunc;
    s.pos += (size_t)n;
    if (s3_get(&n, 4, 1, &s) != 1)
        goto trunc;
    if (n < 1 || s.pos + (size_t)n > s.len || s.buf[s.pos + n - 1] != '\0')
        goto trunc;
    s.pos += (size_t)n;
    for (;;) {
        const char *str;
        if (s3_get(&n, 4, 1, &s) != 1)
            goto trunc;
        if (n == 0)
            break;
        if (n < 0 || s.pos + (size_t)n > s.len)
            goto trunc;
        str = (const char *)s.buf + s.pos;
        if (!strncmp(str, "feature_count ", 14))
            n_feat = atoi(str + 14);
        if (!strncmp(str, "mixture_count ", 14))
            n_density = atoi(str + 14);
        if (!strncmp(str, "model_count ", 12))
            n_sen = atoi(str + 12);
        if (!strncmp(str, "cluster_count ", 14))
            n_clust = atoi(str + 14);
        if (!strncmp(str, "cluster_bits ", 13))
            n_bits = atoi(str + 13);
        s.pos += (size_t)n;
    }
    c = n_sen;
    r = n_density;
    if (n_clust == 0) {
        if (s3_get(&r, 4, 1, &s) != 1 || s3_get(&c, 4, 1, &s) != 1)
            goto trunc;
    }
    if (n_feat != m->n_feat || n_density != m->n_density || n_sen != m->n_sen) {
        set_err("%s: sendump dimensions (%d,%d,%d) do not match model (%d,%d,%d)", path,
                n_feat, n_density, n_sen, m->n_feat, m->n_density, m->n_sen);
        goto fail;
    }
    if (!(n_clust == 0 || n_clust == 15 || n_clust == 16) || !(n_bits == 8 || n_bits == 4)) {
        set_err("%s: bad cluster_count/cluster_bits", path);
        goto fail;
    }
    if (n_clust == 15)
        ++n_clust;
    if (n_clust) {
        if (s.pos + (size_t)n_clust > s.len)
            goto trunc;
        m->mixw_cb = malloc(16);
        memcpy(m->mixw_cb, s.buf + s.pos, (size_t)n_clust);
        s.pos += (size_t)n_clust;
    }
    step = (size_t)c;
    if (n_bits == 4)
        step = (step + 1) / 2;
    need = (size_t)n_feat * (size_t)r * step;
    if (s.pos + need > s.len)
        goto trunc;
    if (r != n_density) {
        set_err("%s: %d rows != %d densities", path, r, n_density);
        goto fail;
    }
    m->mixw_bits = n_bits;
    m->mixw_row = step;
    m->ptm_mixw = malloc(need);
    memcpy(m->ptm_mixw, s.buf + s.pos, need);
    s3_close(&s);
    return 0;
trunc:
    set_err("%s: truncated sendump", path);
fail:
    s3_close(&s);
    return -1;
}

/* Mixture-weight file.  For PTM: read_mixw, src/ptm_mgau.c:611-692 (quantised with the
 * shift-10 logmath, clamp 159, layout [feat][cw][sen]).  For ms: senone_mixw_read,
 * src/ms_senone.c:103-198 (shift-0 logmath, +511, >>10, clamp 255, layout [sen][feat][cw]). */
static int
load_mixw(orc_model_t *m, const char *path, int want_ptm)
{
    s3buf_t s;
    int32_t n_sen, n_feat, n_comp, n, i, f, c;
    float *pdf = NULL;

    if (s3_open(&s, path) < 0)
        return -1;
    if (s3_parse_header(&s) < 0)
        goto fail;
    if (s3_get(&n_sen, 4, 1, &s) != 1 || s3_get(&n_feat, 4, 1, &s) != 1
        || s3_get(&n_comp, 4, 1, &s) != 1 || s3_get(&n, 4, 1, &s) != 1) {
        set_err("%s: truncated mixw header", path);
        goto fail;
    }
    if (n_feat != m->n_feat || n_comp != m->n_density || n != n_sen * n_feat * n_comp
        || (m->n_sen && n_sen != m->n_sen)) {
        set_err("%s: mixw dimensions %d x %d x %d do not match model", path, n_sen, n_feat,
                n_comp);
        goto fail;
    }
    if (m->n_sen == 0)
        m->n_sen = n_sen;
    pdf = malloc(sizeof(float) * (size_t)n_comp);
    if (want_ptm) {
        m->mixw_bits = 8;
        m->mixw_row = (size_t)n_sen;
        m->ptm_mixw = calloc((size_t)n_feat * n_comp * n_sen, 1);
    }
    m->ms_pdf = calloc((size_t)n_sen * n_feat * n_comp, 1);
    for (i = 0; i < n_sen; ++i)
        for (f = 0; f < n_feat; ++f) {
            if (s3_get(pdf, 4, n_comp, &s) != (size_t)n_comp) {
                set_err("%s: truncated mixw data", path);
                goto fail;
            }
            vec_sum_norm(pdf, n_comp);
            vec_floor(pdf, n_comp, (float)m->cfg.mixwfloor); /* float32 mixwfloor args */
            vec_sum_norm(pdf, n_comp);
            for (c = 0; c < n_comp; ++c) {
                int32_t p;
                if (want_ptm) {
                    int32_t q = -orc_logmath_log(m->lmath_8b, pdf[c]);
                    if (q > MAX_NEG_MIXW || q < 0)
                        q = MAX_NEG_MIXW;
                    m->ptm_mixw[((size_t)f * n_comp + c) * n_sen + i] = (uint8_t)q;
                }
                p = -orc_logmath_log(m->lmath, pdf[c]);
                p += (1 << (SENSCR_SHIFT - 1)) - 1;
                m->ms_pdf[((size_t)i * n_feat + f) * n_comp + c]
                    = (p < (255 << SENSCR_SHIFT)) ? (uint8_t)(p >> SENSCR_SHIFT) : 255;
            }
        }
    if (s3_verify_chksum(&s) < 0)
        goto fail;
    free(pdf);
    s3_close(&s);
    return 0;
fail:
    free(pdf);
    s3_close(&s);
    return -1;
}

static void
ptm_alloc_hist(orc_model_t *m)
{
    size_t n = (size_t)m->n_cb * m->n_feat * m->cfg.topn;
    int i;
    for (i = 0; i < 2; ++i) {
        m->hist[i] = calloc(n, sizeof(topn_t));
        m->cb_active[i] = calloc((size_t)m->n_cb, 1);
    }
    orc_ptm_reset(m);
}

orc_model_t *
orc_model_load(const char *mdef, const char *means, const char *vars, const char *sendump,
               const char *mixw, const char *tmat, const orc_config_t *cfg)
{
    orc_model_t *m = calloc(1, sizeof(*m));
    int i;

    if (cfg)
        m->cfg = *cfg;
    else
        orc_config_defaults(&m->cfg);
    g_err[0] = '\0';
    m->lmath = orc_logmath_init(m->cfg.logbase, 0, 1);
    m->lmath_8b = orc_logmath_init(m->cfg.logbase, SENSCR_SHIFT, 1);
    if (m->lmath == NULL || m->lmath_8b == NULL)
        goto fail;
    if (m->lmath_8b->width != 1) { /* src/ptm_mgau.c:739-743 */
        set_err("log base too small for an 8-bit add table");
        goto fail;
    }
    if (mdef && load_mdef(m, mdef) < 0)
        goto fail;
    if (means && vars && load_gauden(m, means, vars) < 0)
        goto fail;
    if (tmat && load_tmat(m, tmat) < 0)
        goto fail;
    if (sendump && load_sendump(m, sendump) < 0)
        goto fail;
    if (mixw && load_mixw(m, mixw, sendump == NULL) < 0)
        goto fail;
    if (m->mean && m->n_sen && m->sen2cimap) {
        /* src/ptm_mgau.c:797-799 / src/ms_senone.c:231-237: senone -> CI-phone codebook */
        m->sen2cb = malloc((size_t)m->n_sen);
        for (i = 0; i < m->n_sen; ++i)
            m->sen2cb[i] = (uint8_t)m->sen2cimap[i];
        if (m->cfg.topn > m->n_density || m->cfg.topn <= 0)
            m->cfg.topn = m->n_density;
        ptm_alloc_hist(m);
    }
    return m;
fail:
    orc_model_free(m);
    return NULL;
}

void
orc_model_free(orc_model_t *m)
{
    int i;
    if (m == NULL)
        return;
    orc_logmath_free(m->lmath);
    orc_logmath_free(m->lmath_8b);
    free(m->veclen);
    free(m->featoff);
    free(m->mean);
    free(m->var);
    free(m->det);
    free(m->cbf_off);
    free(m->sseq);
    free(m->sen2cimap);
    free(m->phone_ssid);
    free(m->phone_tmat);
    free(m->cd_tree);
    free(m->ci_filler);
    free(m->ciname);
    free(m->tp);
    free(m->ptm_mixw);
    free(m->mixw_cb);
    free(m->sen2cb);
    free(m->ms_pdf);
    for (i = 0; i < 2; ++i) {
        free(m->hist[i]);
        free(m->cb_active[i]);
    }
    free(m);
}

void
orc_model_dims(const orc_model_t *m, int32_t *d)
{
    d[0] = m->n_cb;
    d[1] = m->n_feat;
    d[2] = m->n_density;
    d[3] = m->veclen_total;
    d[4] = m->n_sen;
    d[5] = m->n_ci_sen;
    d[6] = m->n_ciphone;
    d[7] = m->n_phone;
    d[8] = m->n_emit_state;
    d[9] = m->n_tmat;
    d[10] = m->n_sseq;
    d[11] = m->sil;
    d[12] = m->n_floored;
    d[13] = m->n_cd_tree;
    d[14] = m->ptm_mixw != NULL;
    d[15] = m->ms_pdf != NULL;
}

const int32_t *orc_model_veclen(const orc_model_t *m) { return m->veclen; }
const float *orc_model_mean(const orc_model_t *m) { return m->mean; }
const float *orc_model_var(const orc_model_t *m) { return m->var; }
const float *orc_model_det(const orc_model_t *m) { return m->det; }
const uint8_t *orc_model_ptm_mixw(const orc_model_t *m) { return m->ptm_mixw; }
const uint8_t *orc_model_ms_pdf(const orc_model_t *m) { return m->ms_pdf; }
const uint8_t *orc_model_tp(const orc_model_t *m) { return m->tp; }
const uint16_t *orc_model_sseq(const orc_model_t *m) { return m->sseq; }
const int16_t *orc_model_sen2cimap(const orc_model_t *m) { return m->sen2cimap; }
const int32_t *orc_model_phone_ssid(const orc_model_t *m) { return m->phone_ssid; }
const int32_t *orc_model_phone_tmat(const orc_model_t *m) { return m->phone_tmat; }
const orc_logmath_t *orc_model_lmath(const orc_model_t *m) { return m->lmath; }
const orc_logmath_t *orc_model_lmath_8b(const orc_model_t *m) { return m->lmath_8b; }

/* ================================================================================== */
/* PTM scorer: src/ptm_mgau.c                                                          */
/* ================================================================================== */

/* One density: d = det - sum_j (x_j - mu_j)^2 * v_j, four separately rounded float ops per
 * dimension, subtracted in index order (src/ptm_mgau.c:63-68,106-127; the 4-way unroll does
 * not reorder the subtractions). */
static float
density(const float *x, const float *mean, const float *var, float det, int len)
{
    float d = det;
    int j;
    for (j = 0; j < len; ++j) {
        float diff = x[j] - mean[j];
        float sq = diff * diff;
        float c = sq * var[j];
        d = d - c;
    }
    return d;
}

/* Replay of the GPU scan's arithmetic, for tests/test_scan_bound.py (no reference
 * counterpart: the reference only has the exact form above).  For every frame x[i] and every
 * density of one codebook-stream: ref = the reference's fp32 density value from the exact
 * record (mean | det at [15] | scale at [16..]), key = the quadratic form of the scan record
 * (a | c at [15] | b at [16..]) accumulated the way the kernel does it: c, then for each
 * dimension fma(a, x, .) and fma(b, fl(x*x), .), each rounded once (fmaf). */
void
orc_scan_replay(const float *rec, const float *recq, int n_density, int veclen,
                const float *x, int n, int x_stride, float *ref_out, float *key_out)
{
    int i, d, j;
    for (i = 0; i < n; ++i) {
        const float *xi = x + (size_t)i * x_stride;
        for (d = 0; d < n_density; ++d) {
            const float *r = rec + (size_t)d * 32, *q = recq + (size_t)d * 32;
            float k = q[15];
            for (j = 0; j < veclen; ++j) {
                float xx = xi[j] * xi[j];
                k = fmaf(q[j], xi[j], k);
                k = fmaf(q[16 + j], xx, k);
            }
            ref_out[(size_t)i * n_density + d] = density(xi, r, r + 16, r[15], veclen);
            key_out[(size_t)i * n_density + d] = k;
        }
    }
}

/* (int32)d with the reference's clamp, src/ptm_mgau.c:128-131, 218-221 */
static int32_t
dens2int(float d)
{
    if (d < (float)MAX_NEG_INT32)
        return MAX_NEG_INT32;
    return (int32_t)d;
}

void
orc_ptm_reset(orc_model_t *m)
{
    int i, c, f, k;
    if (m->hist[0] == NULL)
        return;
    for (i = 0; i < 2; ++i) {
        for (c = 0; c < m->n_cb; ++c) {
            for (f = 0; f < m->n_feat; ++f)
                for (k = 0; k < m->cfg.topn; ++k) {
                    topn_t *t = m->hist[i] + ((size_t)c * m->n_feat + f) * m->cfg.topn + k;
                    t->cw = k;
                    t->score = MAX_NEG_INT32;
                }
            m->cb_active[i][c] = 1;
        }
    }
    m->frame_idx = 0;
}

void
orc_ptm_set_frame_idx(orc_model_t *m, int frame_idx)
{
    m->frame_idx = frame_idx;
}

/* eval_topn, src/ptm_mgau.c:86-135 (+ insertion_sort_topn :70-84) */
static void
ptm_eval_topn(const orc_model_t *m, topn_t *topn, int cb, int feat, const float *z)
{
    int len = m->veclen[feat], n = m->cfg.topn, i, j;
    size_t base = m->cbf_off[cb * m->n_feat + feat];
    const float *det = m->det + ((size_t)cb * m->n_feat + feat) * m->n_density;

    for (i = 0; i < n; ++i) {
        int32_t cw = topn[i].cw;
        float d = density(z, m->mean + base + (size_t)cw * len, m->var + base + (size_t)cw * len,
                          det[cw], len);
        int32_t s = dens2int(d);
        topn_t tmp;
        topn[i].score = s;
        tmp = topn[i];
        /* bubble towards the front while strictly better: new lands AFTER equal scores */
        for (j = i - 1; j >= 0 && s > topn[j].score; --j)
            topn[j + 1] = topn[j];
        topn[j + 1] = tmp;
    }
}

/* eval_cb, src/ptm_mgau.c:150-225 (+ insertion_sort_cb :139-148).  The reference walks the
 * dimensions in stages -- len % 4 single ones, then groups of four -- and leaves the walk as soon
 * as d < thresh at the head of a stage (:182, :191): "terminated early, so not in topn" (:206-
 * 211).  For finite features that only skips densities the final `d < thresh` test (:212) would
 * reject as well (every term is >= 0).  Restated stage by stage all the same (round 5), because
 * a NaN feature makes the two differ: `d >= thresh` is false for a NaN d, so a density whose d
 * turns NaN before the last stage is dropped, while one whose d turns NaN IN the last stage
 * passes `d < thresh` (false again) and is inserted with (int32)NaN -- 0x80000000 on x86. */
static void
ptm_eval_cb(const orc_model_t *m, topn_t *topn, int cb, int feat, const float *z)
{
    int len = m->veclen[feat], n = m->cfg.topn, cw, i, k;
    size_t base = m->cbf_off[cb * m->n_feat + feat];
    const float *det = m->det + ((size_t)cb * m->n_feat + feat) * m->n_density;

    for (cw = 0; cw < m->n_density; ++cw) {
        float thresh = (float)topn[n - 1].score;
        const float *mean = m->mean + base + (size_t)cw * len;
        const float *var = m->var + base + (size_t)cw * len;
        float d = det[cw];
        int32_t s;
        int j, q;
        for (j = 0; j < len % 4 && d >= thresh; ++j) {
            float diff = z[j] - mean[j];
            float sq = diff * diff;
            float c = sq * var[j];
            d = d - c;
        }
        for (; j < len && d >= thresh; j += 4)
            for (q = j; q < j + 4; ++q) {
                float diff = z[q] - mean[q];
                float sq = diff * diff;
                float c = sq * var[q];
                d = d - c;
            }
        if (j < len)
            continue;
        if (d < thresh)
            continue;
        for (i = 0; i < n; ++i)
            if (topn[i].cw == cw)
                break;
        if (i < n)
            continue;
        s = dens2int(d);
        /* shift while s >= entry: new lands BEFORE equal scores; the worst falls off */
        for (k = n - 2; k >= 0 && s >= topn[k].score; --k)
            topn[k + 1] = topn[k];
        topn[k + 1].cw = cw;
        topn[k + 1].score = s;
    }
}

/* ptm_mgau_frame_eval, src/ptm_mgau.c:408-454 */
int
orc_ptm_frame_eval(orc_model_t *m, int16_t *senscr, const uint8_t *senone_active,
                   int32_t n_senone_active, const float *feat, int32_t frame, int32_t compallsen)
{
    int slot = frame % 2, n = m->cfg.topn, c, f, k, i, lastsen;
    topn_t *cur = m->hist[slot];
    uint8_t *active = m->cb_active[slot];
    size_t cbf = (size_t)m->n_feat * n;
    int32_t best;

    if (frame >= m->frame_idx) {
        const topn_t *last = m->hist[slot == 0 ? 1 : 0];
        memcpy(cur, last, sizeof(topn_t) * (size_t)m->n_cb * cbf); /* :440 */
        /* ptm_mgau_calc_cb_active, :297-321 */
        if (compallsen) {
            memset(active, 1, (size_t)m->n_cb);
        } else {
            memset(active, 0, (size_t)m->n_cb);
            for (lastsen = i = 0; i < n_senone_active; ++i) {
                int sen = senone_active[i] + lastsen;
                active[m->sen2cb[sen]] = 1;
                lastsen = sen;
            }
        }
        /* ptm_mgau_codebook_eval, :230-253 */
        for (c = 0; c < m->n_cb; ++c)
            for (f = 0; f < m->n_feat; ++f)
                ptm_eval_topn(m, cur + c * cbf + (size_t)f * n, c, f, feat + m->featoff[f]);
        if (frame % m->cfg.ds == 0) {
            for (c = 0; c < m->n_cb; ++c) {
                if (!active[c])
                    continue;
                for (f = 0; f < m->n_feat; ++f)
                    ptm_eval_cb(m, cur + c * cbf + (size_t)f * n, c, f, feat + m->featoff[f]);
            }
        }
        /* ptm_mgau_codebook_norm, :264-295 */
        for (f = 0; f < m->n_feat; ++f) {
            int32_t norm = WORST_SCORE;
            for (c = 0; c < m->n_cb; ++c) {
                int32_t top;
                if (!active[c])
                    continue;
                top = cur[c * cbf + (size_t)f * n].score >> SENSCR_SHIFT;
                if (norm < top)
                    norm = top;
            }
            for (c = 0; c < m->n_cb; ++c) {
                if (!active[c])
                    continue;
                for (k = 0; k < n; ++k) {
                    topn_t *t = cur + c * cbf + (size_t)f * n + k;
                    t->score >>= SENSCR_SHIFT;
                    t->score -= norm;
                    t->score = -t->score;
                    if (t->score > MAX_NEG_ASCR)
                        t->score = MAX_NEG_ASCR;
                }
            }
        }
    }

    /* ptm_mgau_senone_eval, :326-403 */
    memset(senscr, 0, sizeof(int16_t) * (size_t)m->n_sen);
    if (compallsen)
        n_senone_active = m->n_sen;
    best = INT32_MAX;
    for (lastsen = i = 0; i < n_senone_active; ++i) {
        int sen = compallsen ? i : senone_active[i] + lastsen;
        int cb, ascore = 0;
        lastsen = sen;
        cb = m->sen2cb[sen];
        if (!active[cb]) /* :353-364 */
            for (f = 0; f < m->n_feat; ++f)
                for (k = 0; k < n; ++k)
                    cur[cb * cbf + (size_t)f * n + k].score = MAX_NEG_ASCR;
        for (f = 0; f < m->n_feat; ++f) {
            const topn_t *t = cur + cb * cbf + (size_t)f * n;
            int fden = 0;
            for (k = 0; k < n; ++k) {
                const uint8_t *row = m->ptm_mixw + ((size_t)f * m->n_density + t[k].cw) * m->mixw_row;
                int mixw;
                if (m->mixw_cb) { /* :375-378: tests dcw & 1, not sen & 1 */
                    int dcw = row[sen / 2];
                    dcw = (dcw & 1) ? dcw >> 4 : dcw & 0x0f;
                    mixw = m->mixw_cb[dcw];
                } else
                    mixw = row[sen];
                if (k == 0)
                    fden = mixw + t[k].score;
                else {
                    /* fast_logmath_add, tied_mgau_common.h:100-117 */
                    int mly = mixw + t[k].score, d, r;
                    if (fden > mly) {
                        d = fden - mly;
                        r = mly;
                    } else {
                        d = mly - fden;
                        r = fden;
                    }
                    /* the reference indexes its (>= 256-entry, :107-109) table without a bound;
                     * d can only pass 255 when fden has gone negative, where the reference reads
                     * past its allocation.  The restatement defines that read as 0, the value
                     * every entry from 29 up holds. */
                    fden = r - ((size_t)d < m->lmath_8b->table_size
                                    ? ((const uint8_t *)m->lmath_8b->table)[d] : 0);
                }
            }
            ascore += fden;
        }
        if (ascore < best)
            best = ascore;
        senscr[sen] = (int16_t)ascore;
    }
    for (i = 0; i < m->n_sen; ++i)
        senscr[i] = (int16_t)(senscr[i] - best); /* :398-400 */
    return 0;
}

void
orc_ptm_get_topn(const orc_model_t *m, int frame, int32_t *cw, int32_t *score)
{
    size_t n = (size_t)m->n_cb * m->n_feat * m->cfg.topn, i;
    const topn_t *t = m->hist[frame % 2];
    for (i = 0; i < n; ++i) {
        cw[i] = t[i].cw;
        score[i] = t[i].score;
    }
}

int
orc_ptm_score_utt(orc_model_t *m, const float *feats, int n_frames, int16_t *out,
                  int32_t *topn_cw, int32_t *topn_score)
{
    size_t n = (size_t)m->n_cb * m->n_feat * m->cfg.topn;
    int t;
    orc_ptm_reset(m);
    for (t = 0; t < n_frames; ++t) {
        orc_ptm_frame_eval(m, out + (size_t)t * m->n_sen, NULL, 0,
                           feats + (size_t)t * m->veclen_total, t, 1);
        if (topn_cw && topn_score)
            orc_ptm_get_topn(m, t, topn_cw + (size_t)t * n, topn_score + (size_t)t * n);
        ++m->frame_idx; /* acmod_advance, src/acmod.c:760 */
    }
    return 0;
}

/* ================================================================================== */
/* ms scorer: src/ms_gauden.c, src/ms_senone.c, src/ms_mgau.c                          */
/* ================================================================================== */

typedef struct gdist_s {
    int32_t id;
    float dist;
} gdist_t; /* gauden_dist_t, ms_gauden.h:71-74 */

/* compute_dist, src/ms_gauden.c:384-432 (n_top < n_density) and compute_dist_all :349-378 */
static void
ms_compute_dist(const orc_model_t *m, gdist_t *out, int n_top, int cb, int feat, const float *obs)
{
    int len = m->veclen[feat], d, i, j;
    size_t base = m->cbf_off[cb * m->n_feat + feat];
    const float *det = m->det + ((size_t)cb * m->n_feat + feat) * m->n_density;

    if (n_top >= m->n_density) {
        for (d = 0; d < m->n_density; ++d) {
            out[d].dist = density(obs, m->mean + base + (size_t)d * len,
                                  m->var + base + (size_t)d * len, det[d], len);
            out[d].id = d;
        }
        return;
    }
    for (i = 0; i < n_top; ++i) {
        out[i].dist = (float)MAX_NEG_INT32; /* WORST_DIST as float, :397-398 */
        out[i].id = 0;                      /* calloc'd in ms_mgau_init, src/ms_mgau.c:152 */
    }
    for (d = 0; d < m->n_density; ++d) {
        float dval = density(obs, m->mean + base + (size_t)d * len,
                             m->var + base + (size_t)d * len, det[d], len);
        if (dval < out[n_top - 1].dist)
            continue;
        for (i = 0; i < n_top && dval < out[i].dist; ++i)
            ;
        for (j = n_top - 1; j > i; --j)
            out[j] = out[j - 1];
        out[i].dist = dval;
        out[i].id = d;
    }
}

/* senone_eval, src/ms_senone.c:314-362 (n_gauden > 1: untransposed pdf[sen][feat][cw]) */
static int32_t
ms_senone_eval(const orc_model_t *m, int sen, const gdist_t *dist /* [feat][topn] */, int n_top)
{
    int32_t scr = 0;
    int f, t;
    for (f = 0; f < m->n_feat; ++f) {
        const gdist_t *fd = dist + (size_t)f * n_top;
        const uint8_t *pdf = m->ms_pdf + ((size_t)sen * m->n_feat + f) * m->n_density;
        int32_t fden, fscr;
        if (fd[0].dist < (float)MAX_NEG_INT32)
            fden = MAX_NEG_INT32 >> SENSCR_SHIFT;
        else
            fden = ((int32_t)fd[0].dist + ((1 << SENSCR_SHIFT) - 1)) >> SENSCR_SHIFT;
        fscr = fden + -(int32_t)pdf[fd[0].id];
        for (t = 1; t < n_top; ++t) {
            int32_t fwscr;
            if (fd[t].dist < (float)MAX_NEG_INT32)
                fden = MAX_NEG_INT32 >> SENSCR_SHIFT;
            else
                fden = ((int32_t)fd[t].dist + ((1 << SENSCR_SHIFT) - 1)) >> SENSCR_SHIFT;
            fwscr = fden + -(int32_t)pdf[fd[t].id];
            fscr = orc_logmath_add(m->lmath_8b, fscr, fwscr);
        }
        scr -= fscr;
    }
    scr /= m->cfg.aw;
    if (scr > 32767)
        scr = 32767;
    if (scr < -32768)
        scr = -32768;
    return scr;
}

/* ms_cont_mgau_frame_eval, src/ms_mgau.c:278-368 */
int
orc_ms_frame_eval(orc_model_t *m, int16_t *senscr, const uint8_t *senone_active,
                  int32_t n_senone_active, const float *feat, int32_t frame, int32_t compallsen)
{
    int n_top = m->cfg.topn, c, f, s, i, n;
    size_t per_cb = (size_t)m->n_feat * n_top;
    gdist_t *dist = calloc((size_t)m->n_cb * per_cb, sizeof(gdist_t));
    uint8_t *active = calloc((size_t)m->n_cb, 1);
    int32_t best = INT32_MAX;

    (void)frame;
    if (compallsen) {
        for (c = 0; c < m->n_cb; ++c)
            for (f = 0; f < m->n_feat; ++f)
                ms_compute_dist(m, dist + c * per_cb + (size_t)f * n_top, n_top, c, f,
                                feat + m->featoff[f]);
        for (s = 0; s < m->n_sen; ++s) {
            senscr[s] = (int16_t)ms_senone_eval(m, s, dist + m->sen2cb[s] * per_cb, n_top);
            if (best > senscr[s])
                best = senscr[s];
        }
        for (s = 0; s < m->n_sen; ++s) {
            int32_t bs = senscr[s] - best;
            if (bs > 32767)
                bs = 32767;
            if (bs < -32768)
                bs = -32768;
            senscr[s] = (int16_t)bs;
        }
    } else {
        for (n = 0, i = 0; i < n_senone_active; ++i) {
            s = senone_active[i] + n;
            active[m->sen2cb[s]] = 1;
            n = s;
        }
        for (c = 0; c < m->n_cb; ++c)
            if (active[c])
                for (f = 0; f < m->n_feat; ++f)
                    ms_compute_dist(m, dist + c * per_cb + (size_t)f * n_top, n_top, c, f,
                                    feat + m->featoff[f]);
        for (n = 0, i = 0; i < n_senone_active; ++i) {
            s = senone_active[i] + n;
            senscr[s] = (int16_t)ms_senone_eval(m, s, dist + m->sen2cb[s] * per_cb, n_top);
            if (best > senscr[s])
                best = senscr[s];
            n = s;
        }
        for (n = 0, i = 0; i < n_senone_active; ++i) {
            int32_t bs;
            s = senone_active[i] + n;
            bs = senscr[s] - best;
            if (bs > 32767)
                bs = 32767;
            if (bs < -32768)
                bs = -32768;
            senscr[s] = (int16_t)bs;
            n = s;
        }
    }
    free(dist);
    free(active);
    return 0;
}

int
orc_ms_score_utt(orc_model_t *m, const float *feats, int n_frames, int16_t *out)
{
    int t;
    for (t = 0; t < n_frames; ++t)
        orc_ms_frame_eval(m, out + (size_t)t * m->n_sen, NULL, 0,
                          feats + (size_t)t * m->veclen_total, t, 1);
    return 0;
}

/* ================================================================================== */
/* active list: src/acmod.c:947-999                                                    */
/* ================================================================================== */
int
orc_flags2list(const uint32_t *vec, int n_sen, uint8_t *out)
{
    int total_words = n_sen / 32, extra = n_sen % 32, w, b, n = 0, l = 0;
    for (w = 0; w < total_words; ++w) {
        if (vec[w] == 0)
            continue;
        for (b = 0; b < 32; ++b)
            if (vec[w] & (1UL << b)) {
                int sen = w * 32 + b, delta = sen - l;
                while (delta > 255) { /* lossy bridge: extra senones become active */
                    out[n++] = 255;
                    delta -= 255;
                }
                out[n++] = (uint8_t)delta;
                l = sen;
            }
    }
    for (b = 0; b < extra; ++b)
        if (vec[w] & (1UL << b)) {
            int sen = w * 32 + b, delta = sen - l;
            while (delta > 255) {
                out[n++] = 255;
                delta -= 255;
            }
            out[n++] = (uint8_t)delta;
            l = sen;
        }
    return n;
}

/* ================================================================================== */
/* HMM Viterbi step: src/hmm.c                                                         */
/* ================================================================================== */

#define HMM_MAX_NSTATE 5
typedef struct ohmm_s {
    int32_t score[HMM_MAX_NSTATE];
    int32_t history[HMM_MAX_NSTATE];
    int32_t out_score, out_history;
    uint16_t senid[HMM_MAX_NSTATE];
    int32_t bestscore;
    int16_t tmatid;
    int32_t frame;
    int n_emit;
} ohmm_t; /* hmm_t, hmm.h:164-180 (non-mpx fields) */

/* hmm_clear, src/hmm.c:124-140 */
static void
ohmm_clear(ohmm_t *h)
{
    int i;
    for (i = 0; i < h->n_emit; ++i) {
        h->score[i] = WORST_SCORE;
        h->history[i] = -1;
    }
    h->out_score = WORST_SCORE;
    h->out_history = -1;
    h->bestscore = WORST_SCORE;
    h->frame = -1;
}

/* hmm_vit_eval_3st_lr, src/hmm.c:482-567.  tp = row-major [3][4] uint8, negated on use. */
static int32_t
vit_3st(ohmm_t *h, const uint8_t *tp, const int16_t *senscr)
{
    int32_t s3, s2, s1, s0, t2, t1, t0, best;
#define TP(i, j) (-(int32_t)tp[(i) * 4 + (j)])
    s2 = h->score[2] + -(int32_t)senscr[h->senid[2]];
    s1 = h->score[1] + -(int32_t)senscr[h->senid[1]];
    s0 = h->score[0] + -(int32_t)senscr[h->senid[0]];
    best = WORST_SCORE;
    t2 = INT_MIN; /* only assigned when a skip arc exists; NOT reset between blocks */

    if (s1 > WORST_SCORE) {
        t1 = s2 + TP(2, 3);
        if (TP(1, 3) > TMAT_WORST_SCORE)
            t2 = s1 + TP(1, 3);
        if (t1 > t2) {
            s3 = t1;
            h->out_history = h->history[2];
        } else {
            s3 = t2;
            h->out_history = h->history[1];
        }
        if (s3 < WORST_SCORE)
            s3 = WORST_SCORE;
        h->out_score = s3;
        best = s3;
    }

    t0 = s2 + TP(2, 2);
    t1 = s1 + TP(1, 2);
    if (TP(0, 2) > TMAT_WORST_SCORE)
        t2 = s0 + TP(0, 2);
    if (t0 > t1) {
        if (t2 > t0) {
            s2 = t2;
            h->history[2] = h->history[0];
        } else
            s2 = t0;
    } else {
        if (t2 > t1) {
            s2 = t2;
            h->history[2] = h->history[0];
        } else {
            s2 = t1;
            h->history[2] = h->history[1];
        }
    }
    if (s2 < WORST_SCORE)
        s2 = WORST_SCORE;
    if (s2 > best)
        best = s2;
    h->score[2] = s2;

    t0 = s1 + TP(1, 1);
    t1 = s0 + TP(0, 1);
    if (t0 > t1) {
        s1 = t0;
    } else {
        s1 = t1;
        h->history[1] = h->history[0];
    }
    if (s1 < WORST_SCORE)
        s1 = WORST_SCORE;
    if (s1 > best)
        best = s1;
    h->score[1] = s1;

    s0 = s0 + TP(0, 0);
    if (s0 < WORST_SCORE)
        s0 = WORST_SCORE;
    if (s0 > best)
        best = s0;
    h->score[0] = s0;
#undef TP
    h->bestscore = best;
    return best;
}

/* hmm_vit_eval_5st_lr, src/hmm.c:166-304.  tp = [5][6]. */
static int32_t
vit_5st(ohmm_t *h, const uint8_t *tp, const int16_t *senscr)
{
    int32_t s5, s4, s3, s2, s1, s0, t2, t1, t0, best = WORST_SCORE;
#define TP(i, j) (-(int32_t)tp[(i) * 6 + (j)])
#define SS(i) (-(int32_t)senscr[h->senid[i]])
    s4 = h->score[4] + SS(4);
    s3 = h->score[3] + SS(3);
    if (s3 > WORST_SCORE) {
        t1 = s4 + TP(4, 5);
        t2 = s3 + TP(3, 5);
        if (t1 > t2) {
            s5 = t1;
            h->out_history = h->history[4];
        } else {
            s5 = t2;
            h->out_history = h->history[3];
        }
        if (s5 < WORST_SCORE)
            s5 = WORST_SCORE;
        h->out_score = s5;
        best = s5;
    }
    s2 = h->score[2] + SS(2);
    if (s2 > WORST_SCORE) {
        t0 = s4 + TP(4, 4);
        t1 = s3 + TP(3, 4);
        t2 = s2 + TP(2, 4);
        if (t0 > t1) {
            if (t2 > t0) {
                s4 = t2;
                h->history[4] = h->history[2];
            } else
                s4 = t0;
        } else {
            if (t2 > t1) {
                s4 = t2;
                h->history[4] = h->history[2];
            } else {
                s4 = t1;
                h->history[4] = h->history[3];
            }
        }
        if (s4 < WORST_SCORE)
            s4 = WORST_SCORE;
        if (s4 > best)
            best = s4;
        h->score[4] = s4;
    }
    s1 = h->score[1] + SS(1);
    if (s1 > WORST_SCORE) {
        t0 = s3 + TP(3, 3);
        t1 = s2 + TP(2, 3);
        t2 = s1 + TP(1, 3);
        if (t0 > t1) {
            if (t2 > t0) {
                s3 = t2;
                h->history[3] = h->history[1];
            } else
                s3 = t0;
        } else {
            if (t2 > t1) {
                s3 = t2;
                h->history[3] = h->history[1];
            } else {
                s3 = t1;
                h->history[3] = h->history[2];
            }
        }
        if (s3 < WORST_SCORE)
            s3 = WORST_SCORE;
        if (s3 > best)
            best = s3;
        h->score[3] = s3;
    }
    s0 = h->score[0] + SS(0);
    t0 = s2 + TP(2, 2);
    t1 = s1 + TP(1, 2);
    t2 = s0 + TP(0, 2);
    if (t0 > t1) {
        if (t2 > t0) {
            s2 = t2;
            h->history[2] = h->history[0];
        } else
            s2 = t0;
    } else {
        if (t2 > t1) {
            s2 = t2;
            h->history[2] = h->history[0];
        } else {
            s2 = t1;
            h->history[2] = h->history[1];
        }
    }
    if (s2 < WORST_SCORE)
        s2 = WORST_SCORE;
    if (s2 > best)
        best = s2;
    h->score[2] = s2;

    t0 = s1 + TP(1, 1);
    t1 = s0 + TP(0, 1);
    if (t0 > t1) {
        s1 = t0;
    } else {
        s1 = t1;
        h->history[1] = h->history[0];
    }
    if (s1 < WORST_SCORE)
        s1 = WORST_SCORE;
    if (s1 > best)
        best = s1;
    h->score[1] = s1;

    s0 = s0 + TP(0, 0);
    if (s0 < WORST_SCORE)
        s0 = WORST_SCORE;
    if (s0 > best)
        best = s0;
    h->score[0] = s0;
#undef TP
#undef SS
    h->bestscore = best;
    return best;
}

/* hmm_vit_eval_anytopo, src/hmm.c:671-739 (non-mpx) */
static int32_t
vit_any(ohmm_t *h, const uint8_t *tp, const int16_t *senscr)
{
    int n = h->n_emit, to, from, bestfrom, final = n;
    int32_t st[HMM_MAX_NSTATE], scr, newscr, bestscr;
#define TP(i, j) (-(int32_t)tp[(i) * (n + 1) + (j)])
    st[0] = h->score[0] + -(int32_t)senscr[h->senid[0]];
    for (from = 1; from < n; ++from) {
        st[from] = h->score[from] + -(int32_t)senscr[h->senid[from]];
        if (st[from] < WORST_SCORE)
            st[from] = WORST_SCORE;
    }
    to = final;
    scr = WORST_SCORE;
    bestfrom = -1;
    for (from = to - 1; from >= 0; --from)
        if (TP(from, to) > TMAT_WORST_SCORE && (newscr = st[from] + TP(from, to)) > scr) {
            scr = newscr;
            bestfrom = from;
        }
    h->out_score = scr;
    if (bestfrom >= 0)
        h->out_history = h->history[bestfrom];
    bestscr = scr;
    for (to = final - 1; to >= 0; --to) {
        scr = (TP(to, to) > TMAT_WORST_SCORE) ? st[to] + TP(to, to) : WORST_SCORE;
        bestfrom = -1;
        for (from = to - 1; from >= 0; --from)
            if (TP(from, to) > TMAT_WORST_SCORE && (newscr = st[from] + TP(from, to)) > scr) {
                scr = newscr;
                bestfrom = from;
            }
        h->score[to] = scr;
        if (bestfrom >= 0)
            h->history[to] = h->history[bestfrom];
        if (bestscr < scr)
            bestscr = scr;
    }
#undef TP
    h->bestscore = bestscr;
    return bestscr;
}

/* hmm_vit_eval, src/hmm.c:741-759 (non-mpx branch) */
static int32_t
ohmm_vit_eval(ohmm_t *h, const uint8_t *tp, const int16_t *senscr)
{
    if (h->n_emit == 5)
        return vit_5st(h, tp, senscr);
    if (h->n_emit == 3)
        return vit_3st(h, tp, senscr);
    return vit_any(h, tp, senscr);
}

int32_t
orc_hmm_vit_eval(int n_emit, const uint8_t *tp, const int16_t *senscr, const uint16_t *senid,
                 int32_t *score, int32_t *history, int32_t *out)
{
    ohmm_t h;
    int i;
    int32_t best;
    memset(&h, 0, sizeof(h));
    h.n_emit = n_emit;
    for (i = 0; i < n_emit; ++i) {
        h.score[i] = score[i];
        h.history[i] = history[i];
        h.senid[i] = senid[i];
    }
    h.out_score = out[0];
    h.out_history = out[1];
    best = ohmm_vit_eval(&h, tp, senscr);
    for (i = 0; i < n_emit; ++i) {
        score[i] = h.score[i];
        history[i] = h.history[i];
    }
    out[0] = h.out_score;
    out[1] = h.out_history;
    return best;
}

/* hmm_vit_eval over a set of HMMs held in flat arrays (the first-pass restatement in
 * oracle/fsg_oracle.py keeps its lextree nodes that way): nodes idx[0..n) are stepped in place
 * with the model's transition matrices; best[] gets hmm_bestscore.  src/hmm.c:741-759. */
void
orc_hmm_vit_eval_many(const orc_model_t *m, const int16_t *senscr, int n, const int32_t *idx,
                      const uint16_t *senid, const int16_t *tmat, int32_t *score,
                      int32_t *hist, int32_t *out_score, int32_t *out_hist, int32_t *best)
{
    int k, i;
    const int n_emit = m->n_emit_state;
    for (k = 0; k < n; ++k) {
        const int32_t p = idx[k];
        ohmm_t h;
        memset(&h, 0, sizeof(h));
        h.n_emit = n_emit;
        for (i = 0; i < n_emit; ++i) {
            h.score[i] = score[(size_t)p * n_emit + i];
            h.history[i] = hist[(size_t)p * n_emit + i];
            h.senid[i] = senid[(size_t)p * n_emit + i];
        }
        h.out_score = out_score[p];
        h.out_history = out_hist[p];
        best[p] = ohmm_vit_eval(&h, m->tp + (size_t)tmat[p] * n_emit * (n_emit + 1), senscr);
        for (i = 0; i < n_emit; ++i) {
            score[(size_t)p * n_emit + i] = h.score[i];
            hist[(size_t)p * n_emit + i] = h.history[i];
        }
        out_score[p] = h.out_score;
        out_hist[p] = h.out_history;
    }
}

/* ================================================================================== */
/* state alignment: src/state_align_search.c + alignment_propagate                     */
/* ================================================================================== */

typedef struct tok_s {
    int32_t id, score;
} tok_t; /* state_align_hist_t, state_align_search.h:61-64 */

int
orc_state_align(const orc_model_t *m, const uint8_t *tp_override, const int16_t *senscr,
                int n_sen, int n_frames, int n_phones, int n_emit, const uint16_t *senid,
                const int16_t *tmatid, const int32_t *sf, const int32_t *ef,
                orc_align_entry_t *state_io, orc_align_entry_t *phone_out,
                int32_t *best_score_trace)
{
    const uint8_t *tp = tp_override ? tp_override : m->tp;
    int n_states = n_phones * n_emit, tpsz = n_emit * (n_emit + 1);
    ohmm_t *hmms = calloc((size_t)n_phones, sizeof(ohmm_t));
    tok_t *tokens = malloc(sizeof(tok_t) * (size_t)n_states * (size_t)(n_frames > 0 ? n_frames : 1));
    int32_t best_score = 0; /* calloc'd search struct, state_align_search.c:439 */
    int frame = 0, i, j, t, rv = -1;
    tok_t last, cur;
    int last_frame, cur_frame, last_parent;

    if (n_emit > HMM_MAX_NSTATE || n_phones <= 0) {
        set_err("bad alignment shape");
        goto done;
    }
    for (i = 0; i < n_phones; ++i) { /* hmm_init non-mpx, src/hmm.c:85-103 */
        hmms[i].n_emit = n_emit;
        for (j = 0; j < n_emit; ++j)
            hmms[i].senid[j] = senid[i * n_emit + j];
        hmms[i].tmatid = tmatid[i];
        ohmm_clear(&hmms[i]);
    }
    /* state_align_search_start :46-55: hmm_enter(hmms, 0, 0, 0) */
    hmms[0].score[0] = 0;
    hmms[0].history[0] = 0;
    hmms[0].frame = 0;

    for (t = 0; t < n_frames; ++t) { /* state_align_search_step :177-213, frame_idx = t */
        const int16_t *scr = senscr + (size_t)t * n_sen;
        tok_t *tk = tokens + (size_t)t * n_states;
        int nf = t + 1;
        int32_t bs = WORST_SCORE;

        if (best_score - 0x300000 < WORST_SCORE) /* renormalize_hmms :57-64, hmm_normalize */
            for (i = 0; i < n_phones; ++i) {
                for (j = 0; j < n_emit; ++j)
                    if (hmms[i].score[j] > WORST_SCORE)
                        hmms[i].score[j] -= best_score;
                if (hmms[i].out_score > WORST_SCORE)
                    hmms[i].out_score -= best_score;
            }
        /* evaluate_hmms :66-86 */
        for (i = 0; i < n_phones; ++i) {
            int32_t s;
            if (hmms[i].frame < t)
                continue;
            s = ohmm_vit_eval(&hmms[i], tp + (size_t)hmms[i].tmatid * tpsz, scr);
            if (s > bs)
                bs = s;
        }
        best_score = bs;
        if (best_score_trace)
            best_score_trace[t] = bs;
        /* prune_hmms :88-106 */
        for (i = 0; i < n_phones; ++i) {
            if (hmms[i].frame < t)
                continue;
            if (nf > ef[i])
                continue;
            hmms[i].frame = nf;
        }
        /* phone_transition :108-133 */
        for (i = 0; i < n_phones - 1; ++i) {
            ohmm_t *h = &hmms[i], *nh = &hmms[i + 1];
            int32_t newphone;
            if (h->frame != nf)
                continue;
            if (nf < sf[i + 1])
                continue;
            newphone = h->out_score;
            if (nh->frame < t || newphone > nh->score[0]) {
                nh->score[0] = newphone; /* hmm_enter, src/hmm.c:142-148 */
                nh->history[0] = h->out_history;
                nh->frame = nf;
            }
        }
        /* record_transitions :149-175 (extend_tokenstack memsets the frame to 0xff) */
        memset(tk, 0xff, sizeof(tok_t) * (size_t)n_states);
        for (i = 0; i < n_phones; ++i) {
            if (hmms[i].frame < t)
                continue;
            for (j = 0; j < n_emit; ++j) {
                int idx = i * n_emit + j;
                tk[idx].id = hmms[i].history[j];
                tk[idx].score = hmms[i].score[j];
                hmms[i].history[j] = idx;
            }
        }
        ++frame;
    }

    /* state_align_search_finish :215-268 */
    last.id = cur.id = hmms[n_phones - 1].out_history;
    last.score = hmms[n_phones - 1].out_score;
    if (last.id == -1) {
        set_err("Failed to reach final state in alignment");
        goto done;
    }
    last_frame = frame;
    for (cur_frame = frame - 2; cur_frame >= 0; --cur_frame) {
        cur = tokens[(size_t)cur_frame * n_states + cur.id];
        if (cur.id == -1) {
            set_err("Alignment failed in frame %d", cur_frame);
            goto done;
        }
        if (cur.id != last.id) {
            orc_align_entry_t *ent = state_io + last.id;
            ent->start = cur_frame + 1;
            ent->duration = last_frame - ent->start;
            ent->score = last.score - cur.score;
            last = cur;
            last_frame = cur_frame + 1;
        }
    }
    state_io[0].start = 0;
    state_io[0].duration = last_frame;

    /* alignment_propagate, src/ps_alignment.c:316-334: states -> phones */
    if (phone_out) {
        last_parent = -1;
        for (i = 0; i < n_states; ++i) {
            int p = i / n_emit;
            if (p != last_parent) {
                phone_out[p].start = state_io[i].start;
                phone_out[p].duration = 0;
                phone_out[p].score = 0;
            }
            phone_out[p].duration += state_io[i].duration;
            phone_out[p].score += state_io[i].score;
            last_parent = p;
        }
    }
    rv = 0;
done:
    free(hmms);
    free(tokens);
    return rv;
}

/* ================================================================================== */
/* triphone lookup: src/bin_mdef.c:543-720 (only used to build the inputs of the       */
/* end-to-end alignment pin; alignment_populate itself is restated in tests/)         */
/* ================================================================================== */

/* bin_mdef_ciphone_id: linear scan is enough here (the reference bisects a sorted list) */
int
orc_mdef_ciphone_id(const orc_model_t *m, const char *name)
{
    size_t p = 0;
    int i;
    for (i = 0; i < m->n_ciphone && p < m->ciname_len; ++i) {
        if (strcmp(m->ciname + p, name) == 0)
            return i;
        p += strlen(m->ciname + p) + 1;
    }
    return -1;
}

/* bin_mdef_phone_id, src/bin_mdef.c:596-664: walk wpos -> base -> left -> right */
static int
mdef_phone_id(const orc_model_t *m, int ci, int lc, int rc, int wpos)
{
    int ctx[4], level = 0, max = 4 /* N_WORD_POSN */, i;
    size_t node = 0;
    if (lc < 0 && rc < 0 && wpos == 4)
        return ci;
    if (m->cd_tree == NULL || lc < 0 || rc < 0 || wpos == 4)
        return -1;
    ctx[0] = wpos;
    ctx[1] = ci;
    ctx[2] = (m->sil >= 0 && m->ci_filler[lc]) ? m->sil : lc;
    ctx[3] = (m->sil >= 0 && m->ci_filler[rc]) ? m->sil : rc;
    while (level < 4) {
        int16_t c, n_down;
        int32_t down;
        for (i = 0; i < max; ++i) {
            memcpy(&c, m->cd_tree + (node + (size_t)i) * 8, 2);
            if (c == ctx[level])
                break;
        }
        if (i == max)
            return -1;
        memcpy(&n_down, m->cd_tree + (node + (size_t)i) * 8 + 2, 2);
        memcpy(&down, m->cd_tree + (node + (size_t)i) * 8 + 4, 4);
        if (n_down == 0)
            return down; /* leaf: c.pid */
        max = n_down;
        node = (size_t)down;
        ++level;
    }
    return -1;
}

/* bin_mdef_phone_id_nearest, src/bin_mdef.c:666-720 */
int
orc_mdef_phone_id_nearest(const orc_model_t *m, int b, int l, int r, int pos)
{
    int p, tmppos;
    if (l < 0 || r < 0)
        return b;
    if ((p = mdef_phone_id(m, b, l, r, pos)) >= 0)
        return p;
    for (tmppos = 0; tmppos < 4; ++tmppos)
        if (tmppos != pos && (p = mdef_phone_id(m, b, l, r, tmppos)) >= 0)
            return p;
    if (m->sil >= 0) {
        int newl = l, newr = r;
        if (m->ci_filler[l] || pos == 1 /* BEGIN */ || pos == 3 /* SINGLE */)
            newl = m->sil;
        if (m->ci_filler[r] || pos == 2 /* END */ || pos == 3)
            newr = m->sil;
        if (newl != l || newr != r) {
            if ((p = mdef_phone_id(m, b, newl, newr, pos)) >= 0)
                return p;
            for (tmppos = 0; tmppos < 4; ++tmppos)
                if (tmppos != pos && (p = mdef_phone_id(m, b, newl, newr, tmppos)) >= 0)
                    return p;
        }
    }
    return b;
}
