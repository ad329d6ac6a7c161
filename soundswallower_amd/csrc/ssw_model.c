/*
 * ssw_model.c -- host-side model loading for the MI355X acoustic path (plain C).
 *
 * Reads the Sphinx-3 binary model files and derives every table exactly as SoundSwallower
 * derives it at load time (file:line citations are relative to the SoundSwallower tree):
 *   s3 container, byte order, checksum   src/s3file.c:210-327, 366-445, 551-570
 *   means / variances + precompute       src/ms_gauden.c:105-202, 217-258
 *   sendump / mixture_weights            src/ptm_mgau.c:456-692, src/ms_senone.c:103-198
 *   transition matrices                  src/tmat.c:125-227
 *   binary mdef                          src/bin_mdef.c:333-540
 *   integer log tables                   src/logmath.c:60-164, 282-301
 * Load-time arithmetic is double precision libm (log, sqrt) followed by integer truncation,
 * as in the reference; the results are data for the kernels.
 */
#include "ssw_internal.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char tls_err[512];

void
ssw_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tls_err, sizeof(tls_err), fmt, ap);
    va_end(ap);
}

const char *
ssw_last_error(void)
{
    return tls_err;
}

int
ssw_abi_version(void)
{
    return SSW_ABI_VERSION;
}

void
ssw_config_defaults(ssw_config_t *cfg)
{
    cfg->logbase = 1.0001;
    cfg->varfloor = 1e-4;
    cfg->mixwfloor = 1e-7;
    cfg->tmatfloor = 1e-4;
    cfg->topn = 4;
    cfg->ds = 1;
    cfg->aw = 1;
    cfg->device = -1;
}

/* ---------------------------------------------------------------------------------- */
/* Integer log domain (logmath_t).  Only what load time needs: log(), ln_to_log() at a  */
/* given shift, and the 8-bit add table at shift 10.                                   */
/* ---------------------------------------------------------------------------------- */
typedef struct {
    double inv_ln_base;
    double base;
} lbase_t;

static int
ilog(const lbase_t *b, int shift, double p)
{
    if (p <= 0)
        return (int)((int32_t)0x80000000 >> (shift + 2)); /* logmath zero, logmath.c:84 */
    return (int)(log(p) * b->inv_ln_base) >> shift;      /* logmath.c:288 */
}

static int
iln_to_log(const lbase_t *b, int shift, double ln_p)
{
    return (int)(ln_p * b->inv_ln_base) >> shift; /* logmath.c:300 */
}

/* Add table for (base, shift): entry d = round(log_base(1 + base^-d')) >> shift for the first
 * unshifted d' that lands in slot d (logmath.c:101-161).  Returns the entry count the
 * reference would allocate (>= 256) or -1 when entries need more than 8 bits. */
static int
build_logadd8(const lbase_t *b, int shift, uint8_t out[256])
{
    uint32_t maxyx = (uint32_t)(log(2.0) / log(b->base) + 0.5) >> shift;
    double byx = 1.0;
    uint32_t i, size;
    uint8_t seen[256];

    if (maxyx >= 256)
        return -1;
    memset(out, 0, 256);
    memset(seen, 0, sizeof(seen));
    for (i = 0;; ++i) {
        double lobyx = log(1.0 + byx) * b->inv_ln_base;
        int32_t k = (int32_t)(lobyx + 0.5 * (1 << shift)) >> shift;
        uint32_t slot = i >> shift;
        /* the reference keeps a slot's first value and treats a stored 0 as "empty" */
        if (slot < 256 && out[slot] == 0)
            out[slot] = (uint8_t)k;
        if (k <= 0)
            break;
        byx /= b->base;
    }
    size = (i >> shift);
    if (size < 255)
        size = 255;
    (void)seen;
    return (int)size + 1;
}

/* ---------------------------------------------------------------------------------- */
/* Byte cursor over a whole file                                                       */
/* ---------------------------------------------------------------------------------- */
typedef struct {
    uint8_t *base;
    size_t size, at;
    int swap, summing;
    uint32_t sum;
    const char *name;
} rd_t;

static int
rd_open(rd_t *r, const char *path)
{
    FILE *fp = fopen(path, "rb");
    long sz;
    memset(r, 0, sizeof(*r));
    r->name = path;
    if (fp == NULL) {
        ssw_set_error("%s: cannot open", path);
        return -1;
    }
    if (fseek(fp, 0, SEEK_END) != 0 || (sz = ftell(fp)) < 0) {
        fclose(fp);
        ssw_set_error("%s: cannot size", path);
        return -1;
    }
    rewind(fp);
    r->base = (uint8_t *)malloc(sz ? (size_t)sz : 1);
    r->size = (size_t)sz;
    if (r->base == NULL || fread(r->base, 1, r->size, fp) != r->size) {
        fclose(fp);
        free(r->base);
        r->base = NULL;
        ssw_set_error("%s: read failed", path);
        return -1;
    }
    fclose(fp);
    return 0;
}

static void
rd_close(rd_t *r)
{
    free(r->base);
    r->base = NULL;
}

static uint32_t
flip32(uint32_t v)
{
    return ((v & 0xffu) << 24) | ((v & 0xff00u) << 8) | ((v >> 8) & 0xff00u) | (v >> 24);
}

/* n 32-bit words -> dst, honouring byte order and the rolling checksum
 * (sum = rotl(sum, 20) + word, src/s3file.c:383-387) */
static int
rd_words(rd_t *r, void *dst, size_t n)
{
    uint32_t *w = (uint32_t *)dst;
    size_t i;
    if (r->size - r->at < 4 * n) {
        ssw_set_error("%s: truncated (wanted %zu words at offset %zu)", r->name, n, r->at);
        return -1;
    }
    memcpy(w, r->base + r->at, 4 * n);
    r->at += 4 * n;
    if (r->swap)
        for (i = 0; i < n; ++i)
            w[i] = flip32(w[i]);
    if (r->summing) {
        uint32_t s = r->sum;
        for (i = 0; i < n; ++i)
            s = ((s << 20) | (s >> 12)) + w[i];
        r->sum = s;
    }
    return 0;
}

static int
rd_i32(rd_t *r, int32_t *v)
{
    return rd_words(r, v, 1);
}

/* "s3\n" text header: `name value` lines up to `endhdr`, then the 0x11223344 marker.
 * A `chksum0` line (any value) switches the trailing checksum on (src/s3file.c:289-290). */
static int
rd_s3_header(rd_t *r)
{
    int want_sum = 0;
    uint32_t mark;
    if (r->size < 3 || memcmp(r->base, "s3\n", 3) != 0) {
        ssw_set_error("%s: missing s3 signature", r->name);
        return -1;
    }
    r->at = 3;
    for (;;) {
        const char *line = (const char *)r->base + r->at;
        const char *eol = memchr(line, '\n', r->size - r->at);
        size_t len, a, z;
        if (r->at >= r->size) {
            ssw_set_error("%s: header runs off the end", r->name);
            return -1;
        }
        len = eol ? (size_t)(eol - line) : r->size - r->at;
        r->at += eol ? len + 1 : len;
        for (a = 0; a < len && (line[a] == ' ' || line[a] == '\t' || line[a] == '\r'); ++a)
            ;
        for (z = a; z < len && !(line[z] == ' ' || line[z] == '\t' || line[z] == '\r'); ++z)
            ;
        if (z == a) {
            ssw_set_error("%s: empty header line", r->name);
            return -1;
        }
        if (line[a] == '#')
            continue;
        if (z - a <= 6 && memcmp(line + a, "endhdr", z - a) == 0)
            break;
        if (z - a == 7 && memcmp(line + a, "chksum0", 7) == 0)
            want_sum = 1;
    }
    if (rd_words(r, &mark, 1) < 0)
        return -1;
    if (mark != 0x11223344u) {
        if (flip32(mark) != 0x11223344u) {
            ssw_set_error("%s: bad byte-order marker %08x", r->name, mark);
            return -1;
        }
        r->swap = 1;
    }
    r->summing = want_sum;
    r->sum = 0;
    return 0;
}

static int
rd_s3_finish(rd_t *r)
{
    uint32_t expect, got;
    if (!r->summing)
        return 0;
    got = r->sum;
    r->summing = 0;
    if (rd_words(r, &expect, 1) < 0)
        return -1;
    if (expect != got) {
        ssw_set_error("%s: checksum mismatch (file %08x, computed %08x)", r->name, expect, got);
        return -1;
    }
    return 0;
}

/* ---------------------------------------------------------------------------------- */
/* Gaussians                                                                           */
/* ---------------------------------------------------------------------------------- */
static float *
read_gauss_file(const char *path, int32_t shape[3], int32_t veclen[SSW_MAX_FEAT])
{
    rd_t r;
    int32_t total, per_density = 0, i;
    float *data = NULL;

    if (rd_open(&r, path) < 0)
        return NULL;
    if (rd_s3_header(&r) < 0 || rd_words(&r, shape, 3) < 0)
        goto bad;
    if (shape[1] < 1 || shape[1] > SSW_MAX_FEAT || shape[0] < 1 || shape[2] < 1) {
        ssw_set_error("%s: unsupported shape %d x %d x %d", path, shape[0], shape[1], shape[2]);
        goto bad;
    }
    if (rd_words(&r, veclen, (size_t)shape[1]) < 0 || rd_i32(&r, &total) < 0)
        goto bad;
    for (i = 0; i < shape[1]; ++i)
        per_density += veclen[i];
    if (total != shape[0] * shape[2] * per_density) {
        ssw_set_error("%s: %d values do not match %d x %d x %d", path, total, shape[0],
                      shape[2], per_density);
        goto bad;
    }
    data = (float *)malloc(sizeof(float) * (size_t)total);
    if (data == NULL || rd_words(&r, data, (size_t)total) < 0 || rd_s3_finish(&r) < 0)
        goto bad;
    rd_close(&r);
    return data;
bad:
    free(data);
    rd_close(&r);
    return NULL;
}

static int
load_gaussians(ssw_host_model_t *h, const lbase_t *lb, const char *means, const char *vars)
{
    int32_t sm[3], sv[3], vlv[SSW_MAX_FEAT], c, f, d, j;
    const float vfloor = (float)h->cfg.varfloor; /* float32 parameter, ms_gauden.c:218 */
    float *mp, *vp, *dp;

    if ((h->mean = read_gauss_file(means, sm, h->veclen)) == NULL)
        return -1;
    if ((h->var = read_gauss_file(vars, sv, vlv)) == NULL)
        return -1;
    if (memcmp(sm, sv, sizeof(sm)) != 0 || memcmp(h->veclen, vlv, sizeof(int32_t) * sm[1])) {
        ssw_set_error("means and variances have different shapes");
        return -1;
    }
    h->n_cb = sm[0];
    h->n_feat = sm[1];
    h->n_density = sm[2];
    for (f = 0, h->veclen_total = 0; f < h->n_feat; ++f) {
        h->featoff[f] = h->veclen_total;
        h->veclen_total += h->veclen[f];
    }
    h->det = (float *)malloc(sizeof(float) * (size_t)h->n_cb * h->n_feat * h->n_density);
    if (h->det == NULL)
        return -1;

    /* Walk the file order once: per density, det = sum_j (float)ilog(1/sqrt(2 pi var_j))
     * accumulated in float32; var_j <- (float)iln_to_log(1/(2 var_j)); floor first. */
    mp = h->mean;
    vp = h->var;
    dp = h->det;
    (void)mp;
    for (c = 0; c < h->n_cb; ++c)
        for (f = 0; f < h->n_feat; ++f)
            for (d = 0; d < h->n_density; ++d) {
                float acc = 0;
                for (j = 0; j < h->veclen[f]; ++j, ++vp) {
                    if (*vp < vfloor) {
                        *vp = vfloor;
                        h->n_floored++;
                    }
                    acc += (float)ilog(lb, 0, 1.0 / sqrt(*vp * 2.0 * M_PI));
                    *vp = (float)iln_to_log(lb, 0, 1.0 / (*vp * 2.0));
                }
                *dp++ = acc;
            }
    return 0;
}

/* ---------------------------------------------------------------------------------- */
/* Transition matrices                                                                 */
/* ---------------------------------------------------------------------------------- */
static void
renorm(float *v, int n)
{
    double s = 0.0, k;
    int i;
    for (i = 0; i < n; ++i)
        s += v[i];
    if (s == 0.0)
        return;
    k = 1.0 / s;
    for (i = 0; i < n; ++i)
        v[i] = (float)(v[i] * k);
}

static int
load_tmat(ssw_host_model_t *h, const lbase_t *lb, const char *path)
{
    rd_t r;
    int32_t dims[4], t, i, k;
    float row[16];

    if (rd_open(&r, path) < 0)
        return -1;
    if (rd_s3_header(&r) < 0 || rd_words(&r, dims, 4) < 0)
        goto bad;
    if (dims[2] != dims[1] + 1 || dims[2] > 16 || dims[3] != dims[0] * dims[1] * dims[2]) {
        ssw_set_error("%s: unsupported transition shape %d x %d x %d", path, dims[0], dims[1],
                      dims[2]);
        goto bad;
    }
    h->tp_n_tmat = dims[0];
    h->tp_n_state = dims[1];
    h->tp = (uint8_t *)malloc((size_t)dims[3]);
    for (t = 0; t < dims[0]; ++t)
        for (i = 0; i < dims[1]; ++i) {
            if (rd_words(&r, row, (size_t)dims[2]) < 0)
                goto bad;
            renorm(row, dims[2]);
            for (k = 0; k < dims[2]; ++k) /* floor only non-zero arcs, vector.c:116-123 */
                if (row[k] != 0.0 && row[k] < h->cfg.tmatfloor)
                    row[k] = (float)h->cfg.tmatfloor;
            renorm(row, dims[2]);
            for (k = 0; k < dims[2]; ++k) {
                int q = (-ilog(lb, 0, row[k])) >> SSW_SENSCR_SHIFT; /* tmat.c:206 */
                h->tp[((size_t)t * dims[1] + i) * dims[2] + k] = (uint8_t)(q > 255 ? 255 : q);
            }
        }
    if (rd_s3_finish(&r) < 0)
        goto bad;
    rd_close(&r);
    return 0;
bad:
    rd_close(&r);
    return -1;
}

/* ---------------------------------------------------------------------------------- */
/* Binary model definition                                                             */
/* ---------------------------------------------------------------------------------- */
static uint32_t
peek32(const rd_t *r, size_t at)
{
    uint32_t v;
    memcpy(&v, r->base + at, 4);
    return r->swap ? flip32(v) : v;
}

static uint16_t
peek16(const rd_t *r, size_t at)
{
    uint16_t v;
    memcpy(&v, r->base + at, 2);
    return r->swap ? (uint16_t)((v << 8) | (v >> 8)) : v;
}

static int
load_mdef(ssw_host_model_t *h, const char *path)
{
    rd_t r;
    int32_t magic, ver, desc_len, hd[10], n_tree, i, j;
    size_t names, at, phones, n_sseq_words;

    if (rd_open(&r, path) < 0)
        return -1;
    if (rd_i32(&r, &magic) < 0)
        goto bad;
    if ((uint32_t)magic == 0x424d4446u)
        r.swap = 1;
    else if ((uint32_t)magic != 0x46444d42u) {
        ssw_set_error("%s: not a BMDF file", path);
        goto bad;
    }
    if (rd_i32(&r, &ver) < 0 || rd_i32(&r, &desc_len) < 0)
        goto bad;
    if (ver > 1 || desc_len < 0 || r.at + (size_t)desc_len > r.size) {
        ssw_set_error("%s: unsupported BMDF version/descriptor", path);
        goto bad;
    }
    r.at += (size_t)desc_len;
    if (rd_words(&r, hd, 10) < 0)
        goto bad;
    h->n_ciphone = hd[0];
    h->n_phone = hd[1];
    h->n_emit_state = hd[2];
    h->n_ci_sen = hd[3];
    h->n_sen = hd[4];
    h->n_tmat = hd[5];
    h->n_sseq = hd[6];
    n_tree = hd[8];
    if (h->n_emit_state < 1) {
        ssw_set_error("%s: mixed-topology mdef not supported", path);
        goto bad;
    }
    /* the counts size every table below: a damaged header must not become an allocation of
     * -1 elements or an index past a row */
    if (h->n_ciphone < 1 || h->n_ciphone > 255 || h->n_phone < h->n_ciphone || h->n_sen < 1
        || h->n_sen > 0xffff || h->n_ci_sen < 0 || h->n_ci_sen > h->n_sen || h->n_tmat < 1
        || h->n_sseq < 1 || n_tree < 0 || h->n_emit_state > 16) {
        ssw_set_error("%s: implausible header (%d CI phones, %d phones, %d states, %d senones, "
                      "%d tmats, %d senone sequences, %d tree nodes)", path, h->n_ciphone,
                      h->n_phone, h->n_emit_state, h->n_sen, h->n_tmat, h->n_sseq, n_tree);
        goto bad;
    }
    /* CI names: n_ciphone NUL-terminated strings, block padded to 4 bytes */
    names = at = r.at;
    h->sil = -1;
    h->ciname = (char **)calloc((size_t)h->n_ciphone, sizeof(char *));
    if (h->ciname == NULL)
        goto oom;
    for (i = 0; i < h->n_ciphone; ++i) {
        const char *nm = (const char *)r.base + at;
        size_t l = strnlen(nm, r.size - at);
        if (at + l >= r.size)
            goto trunc;
        if (l == 3 && memcmp(nm, "SIL", 3) == 0)
            h->sil = i;
        h->ciname[i] = (char *)malloc(l + 1);
        if (h->ciname[i] == NULL)
            goto oom;
        memcpy(h->ciname[i], nm, l + 1);
        at += l + 1;
    }
    at = names + ((at - names + 3) / 4) * 4;
    if (at + (size_t)n_tree * 8 > r.size)
        goto trunc;
    h->n_cd_tree = n_tree;
    h->cd_tree = (struct ssw_cd_node_s *)malloc(sizeof(*h->cd_tree) * (size_t)(n_tree ? n_tree : 1));
    if (h->cd_tree == NULL)
        goto oom;
    for (i = 0; i < n_tree; ++i) {
        h->cd_tree[i].ctx = (int16_t)peek16(&r, at + 8 * (size_t)i);
        h->cd_tree[i].n_down = (int16_t)peek16(&r, at + 8 * (size_t)i + 2);
        h->cd_tree[i].down_or_pid = (int32_t)peek32(&r, at + 8 * (size_t)i + 4);
    }
    phones = at + (size_t)n_tree * 8;
    at = phones + (size_t)h->n_phone * 12;
    if (at + 4 > r.size)
        goto trunc;
    n_sseq_words = peek32(&r, at);
    at += 4;
    if (at + 2 * n_sseq_words > r.size || n_sseq_words < (size_t)h->n_sseq * h->n_emit_state)
        goto trunc;
    h->sseq = (uint16_t *)malloc(2 * n_sseq_words);
    if (h->sseq == NULL)
        goto oom;
    for (i = 0; (size_t)i < n_sseq_words; ++i)
        h->sseq[i] = peek16(&r, at + 2 * (size_t)i);
    for (i = 0; i < h->n_sseq * h->n_emit_state; ++i)
        if (h->sseq[i] >= h->n_sen) { /* would index past a score row on the device */
            ssw_set_error("%s: senone sequence %d names senone %d of %d", path,
                          i / h->n_emit_state, h->sseq[i], h->n_sen);
            goto bad;
        }

    h->phone_ssid = (int32_t *)malloc(sizeof(int32_t) * (size_t)h->n_phone);
    h->phone_tmat = (int32_t *)malloc(sizeof(int32_t) * (size_t)h->n_phone);
    h->sen2cb = (int16_t *)malloc(sizeof(int16_t) * (size_t)h->n_sen);
    h->ci_filler = (uint8_t *)calloc((size_t)h->n_ciphone, 1);
    if (!h->phone_ssid || !h->phone_tmat || !h->sen2cb || !h->ci_filler)
        goto oom;
    for (i = 0; i < h->n_sen; ++i)
        h->sen2cb[i] = -1;
    for (i = 0; i < h->n_ciphone && i < h->n_phone; ++i)
        h->ci_filler[i] = r.base[phones + (size_t)i * 12 + 8]; /* info.ci.filler */
    for (i = 0; i < h->n_phone; ++i) {
        size_t e = phones + (size_t)i * 12;
        int base = (i < h->n_ciphone) ? i : r.base[e + 9]; /* first context byte = base phone */
        h->phone_ssid[i] = (int32_t)peek32(&r, e);
        h->phone_tmat[i] = (int32_t)peek32(&r, e + 4);
        if (h->phone_ssid[i] < 0 || h->phone_ssid[i] >= h->n_sseq)
            continue;
        for (j = 0; j < h->n_emit_state; ++j) {
            int s = h->sseq[(size_t)h->phone_ssid[i] * h->n_emit_state + j];
            if (s < h->n_sen && h->sen2cb[s] < 0) /* first owner wins, bin_mdef.c:505-506 */
                h->sen2cb[s] = (int16_t)base;
        }
    }
    rd_close(&r);
    return 0;
oom:
    ssw_set_error("%s: out of memory", path);
    goto bad;
trunc:
    ssw_set_error("%s: truncated", path);
bad:
    rd_close(&r);
    return -1;
}

/* ---------------------------------------------------------------------------------- */
/* Mixture weights                                                                     */
/* ---------------------------------------------------------------------------------- */
static int
has_prefix(const uint8_t *s, size_t n, const char *key, int *val)
{
    size_t k = strlen(key);
    if (n > k && memcmp(s, key, k) == 0) {
        *val = atoi((const char *)s + k);
        return 1;
    }
    return 0;
}

static int
load_sendump(ssw_host_model_t *h, const char *path)
{
    rd_t r;
    int32_t len, rows, cols;
    int feats = h->n_feat, dens = h->n_density, sens = h->n_sen, clusters = 0, bits = 8, f, d, s;
    uint8_t codebook[16];
    size_t stride;

    if (rd_open(&r, path) < 0)
        return -1;
    /* title: its length also reveals the byte order */
    if (rd_i32(&r, &len) < 0)
        goto bad;
    if (len < 1 || len > 999) {
        len = (int32_t)flip32((uint32_t)len);
        if (len < 1 || len > 999) {
            ssw_set_error("%s: implausible title length", path);
            goto bad;
        }
        r.swap = 1;
    }
    if (r.at + (size_t)len > r.size || r.base[r.at + len - 1] != 0)
        goto trunc;
    r.at += (size_t)len;
    /* header string */
    if (rd_i32(&r, &len) < 0)
        goto bad;
    if (len < 1 || r.at + (size_t)len > r.size || r.base[r.at + len - 1] != 0)
        goto trunc;
    r.at += (size_t)len;
    /* key/value strings, terminated by a zero length */
    for (;;) {
        if (rd_i32(&r, &len) < 0)
            goto bad;
        if (len == 0)
            break;
        if (len < 0 || r.at + (size_t)len > r.size)
            goto trunc;
        (void)(has_prefix(r.base + r.at, (size_t)len, "feature_count ", &feats)
               || has_prefix(r.base + r.at, (size_t)len, "mixture_count ", &dens)
               || has_prefix(r.base + r.at, (size_t)len, "model_count ", &sens)
               || has_prefix(r.base + r.at, (size_t)len, "cluster_count ", &clusters)
               || has_prefix(r.base + r.at, (size_t)len, "cluster_bits ", &bits));
        r.at += (size_t)len;
    }
    rows = dens;
    cols = sens;
    if (clusters == 0 && (rd_i32(&r, &rows) < 0 || rd_i32(&r, &cols) < 0))
        goto bad;
    if (feats != h->n_feat || dens != h->n_density || sens != h->n_sen || rows != dens) {
        ssw_set_error("%s: sendump is %d x %d x %d, model is %d x %d x %d", path, feats, dens,
                      sens, h->n_feat, h->n_density, h->n_sen);
        goto bad;
    }
    if ((clusters != 0 && clusters != 15 && clusters != 16) || (bits != 8 && bits != 4)) {
        ssw_set_error("%s: unsupported cluster_count %d / cluster_bits %d", path, clusters, bits);
        goto bad;
    }
    memset(codebook, 0, sizeof(codebook));
    if (clusters) {
        int n = clusters == 15 ? 16 : clusters; /* 15 is stored as 16, ptm_mgau.c:575-576 */
        if (r.at + (size_t)n > r.size)
            goto trunc;
        memcpy(codebook, r.base + r.at, (size_t)n);
        r.at += (size_t)n;
    }
    stride = bits == 4 ? ((size_t)cols + 1) / 2 : (size_t)cols;
    if (r.at + stride * (size_t)rows * (size_t)feats > r.size)
        goto trunc;
    h->ptm_mixw = (uint8_t *)malloc((size_t)feats * dens * sens);
    for (f = 0; f < feats; ++f)
        for (d = 0; d < dens; ++d) {
            const uint8_t *src = r.base + r.at + ((size_t)f * rows + d) * stride;
            uint8_t *dst = h->ptm_mixw + ((size_t)f * dens + d) * sens;
            if (!clusters) {
                memcpy(dst, src, (size_t)sens);
                continue;
            }
            /* Clustered dump, expanded once here with the reference's decode rule, which
             * keys the nibble choice on the packed byte's own low bit (ptm_mgau.c:375-378). */
            for (s = 0; s < sens; ++s) {
                int packed = src[s / 2];
                int code = (packed & 1) ? packed >> 4 : packed & 0x0f;
                dst[s] = codebook[code];
            }
        }
    rd_close(&r);
    return 0;
trunc:
    ssw_set_error("%s: truncated", path);
bad:
    rd_close(&r);
    return -1;
}

static int
load_mixw(ssw_host_model_t *h, const lbase_t *lb, const char *path, int fill_ptm)
{
    rd_t r;
    int32_t dims[4], s, f, c;
    float *w = NULL;
    const float wfloor = (float)h->cfg.mixwfloor;

    if (rd_open(&r, path) < 0)
        return -1;
    if (rd_s3_header(&r) < 0 || rd_words(&r, dims, 4) < 0)
        goto bad;
    if (dims[1] != h->n_feat || dims[2] != h->n_density || dims[3] != dims[0] * dims[1] * dims[2]
        || (h->n_sen && dims[0] != h->n_sen)) {
        ssw_set_error("%s: mixture weights are %d x %d x %d, model wants %d x %d x %d", path,
                      dims[0], dims[1], dims[2], h->n_sen, h->n_feat, h->n_density);
        goto bad;
    }
    h->n_sen = dims[0];
    w = (float *)malloc(sizeof(float) * (size_t)dims[2]);
    h->ms_pdf = (uint8_t *)malloc((size_t)dims[3]);
    if (fill_ptm)
        h->ptm_mixw = (uint8_t *)malloc((size_t)dims[3]);
    for (s = 0; s < dims[0]; ++s)
        for (f = 0; f < dims[1]; ++f) {
            if (rd_words(&r, w, (size_t)dims[2]) < 0)
                goto bad;
            renorm(w, dims[2]);
            for (c = 0; c < dims[2]; ++c)
                if (w[c] < (double)wfloor)
                    w[c] = wfloor;
            renorm(w, dims[2]);
            for (c = 0; c < dims[2]; ++c) {
                /* ms: shift-0 log, +511, >>10, saturate 255 (ms_senone.c:176-182) */
                int p = -ilog(lb, 0, w[c]) + ((1 << (SSW_SENSCR_SHIFT - 1)) - 1);
                h->ms_pdf[((size_t)s * dims[1] + f) * dims[2] + c]
                    = (uint8_t)(p < (255 << SSW_SENSCR_SHIFT) ? p >> SSW_SENSCR_SHIFT : 255);
                if (fill_ptm) { /* PTM: shift-10 log, saturate 159 (ptm_mgau.c:678-681) */
                    int q = -ilog(lb, SSW_SENSCR_SHIFT, w[c]);
                    if (q > SSW_MAX_NEG_MIXW || q < 0)
                        q = SSW_MAX_NEG_MIXW;
                    h->ptm_mixw[((size_t)f * dims[2] + c) * dims[0] + s] = (uint8_t)q;
                }
            }
        }
    if (rd_s3_finish(&r) < 0)
        goto bad;
    free(w);
    rd_close(&r);
    return 0;
bad:
    free(w);
    rd_close(&r);
    return -1;
}

/* ---------------------------------------------------------------------------------- */
/* ---------------------------------------------------------------------------------- */
/* device-layout Gaussian records and the quadratic-form scan records                   */
/* ---------------------------------------------------------------------------------- */
static int
cmp_float(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

/* The speculative scan evaluates  det - sum var (x - mean)^2  as a quadratic form in x,
 *   key = c + sum_j (a_j x_j + b_j x_j^2),  a = 2 var mean,  b = -var,
 *   c = (det - d0) - R + bias,  R = sum var mean^2,  d0 = the codebook's median det,
 * with 26 fused multiply-adds.  With u = 2^-24, S = sum var (x - mean)^2 and
 * M = |det - d0| + R + sum |a x| + sum |b| x^2 <= |det - d0| + 6 R + 3 S
 * (Cauchy-Schwarz), the form is within 27 u M of the real number and the reference's
 * fp32 value within 13 u |det| + 17 u S of it (13 subtractions whose partial sums lie
 * between det and the result; (1+u)^4 on every product); with
 * S <= |det - d0| + |value - d0| that is u (125 |det - d0| + 162 R + 13 |det|) -- folded
 * into c as `bias` with a few per cent of slack, so the key is an upper bound -- plus
 * 98 u |value - d0|, which the kernel adds (104 u |key| + 0.001: |value| and |key| differ
 * by at most the bias) to the one key it uses as a bound.  Densities whose bias would
 * exceed 4 score units (floored variances far from the origin) get an inert scan record
 * and go on the codebook's exact-form list: the kernel evaluates them the reference's
 * way after the scan.  tests/test_scan_bound.py replays the kernel's arithmetic on the
 * CPU against these tables. */
/* round-to-nearest-even float -> IEEE binary16, subnormals kept, overflow to infinity: what
 * v_cvt_pk_f16_f32 does with the kernels' default mode (tools/microbench/mfma_f16_scan.hip
 * compares the two on values down to 2^-40) */
static uint16_t
f16_rne(float x)
{
    uint32_t u, sign, mant;
    int e;
    memcpy(&u, &x, 4);
    sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u >= 0x7f800000u) /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | (u > 0x7f800000u ? 0x200u : 0u));
    e = (int)(u >> 23) - 127;
    mant = (u & 0x7fffffu) | 0x800000u; /* 24 bits, value = mant 2^(e - 23) */
    if (e >= 16)
        return (uint16_t)(sign | 0x7c00u);
    if (e >= -14) { /* normal: keep 11 bits */
        uint32_t keep = mant >> 13, rest = mant & 0x1fffu;
        uint32_t h = ((uint32_t)(e + 15) << 10) + (keep & 0x3ffu);
        if (rest > 0x1000u || (rest == 0x1000u && (keep & 1u)))
            ++h; /* may carry into the exponent, up to infinity: correct */
        return (uint16_t)(sign | h);
    }
    if (e < -26) /* below half of the smallest subnormal (2^-25): zero */
        return (uint16_t)sign;
    {
        /* subnormal: value = q 2^-24, q = mant 2^(e + 1) rounded */
        const int shift = -1 - e; /* 14 .. 25 */
        uint32_t q = mant >> shift, rest = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rest > half || (rest == half && (q & 1u)))
            ++q;
        return (uint16_t)(sign | q);
    }
}

static float
f16_to_float(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const int e = (h >> 10) & 0x1f;
    const uint32_t m = h & 0x3ffu;
    float x;
    if (e == 0)
        x = ldexpf((float)m, -24);
    else if (e == 31)
        x = m ? NAN : INFINITY;
    else
        x = ldexpf((float)(m | 0x400u), e - 25);
    return sign ? -x : x;
}

/* x = p[0] + p[1] + r: two binary16 parts of a float, the first residual x - p[0] exact in float
 * arithmetic; |r| <= max(2^-22 |x|, 2^-25) while the parts stay finite */
static void
f16_split2(float x, uint16_t p[2])
{
    p[0] = f16_rne(x);
    p[1] = f16_rne(x - f16_to_float(p[0]));
}

/* Scan records for the matrix-core scan (csrc/ssw_k1a_mfma.inc).  The key of density i for
 * frame x is the same quadratic form c + sum_j (a_j x_j + b_j x_j^2) as in recq.
 *
 * Round 4: W = (a, c, b) and X = (x, 1, x^2) are cut into TWO binary16 parts each (2 x 11 = 22
 * significand bits) and the three part products (1,1) (1,2) (2,1) are accumulated by six
 * v_mfma_f32_32x32x16_f16 per 32 x 32 tile -- rounds 2-3 used three bf16 parts and six part
 * products, twelve MFMAs, for the same error constant.  binary16 has a narrow exponent range, so
 * the rows of a codebook x stream are scaled by one power of two, W' = W 2^-s (its keys come out
 * as key 2^-s: only keys of one codebook x stream are ever compared, and the kernel scales the
 * one key it uses as a bound back), and the constant travels as c' = c 2^-s 2^-ec against
 * X = 2^ec.  The kernel refuses the speculative result of a frame with |x_j| > 255 (x_j^2 would
 * not fit binary16): such a frame takes the exact pass.
 *
 * Error of the evaluated key against the real number, in units of u = 2^-24, with
 * M = sum_k |W_k X_k| (all of it scales with 2^-s, so the analysis is the unscaled one):
 *   (a) x^2 rounded to fp32 before the split:                          1 u M
 *   (b) what the three products leave out: W X - (W1 X1 + W1 X2 + W2 X1) = W2 X2 + (W1 + W2) Xr
 *       + Wr X with Xr, Wr the residuals of the two-part splits.  |W2| <= 2^-11 |W|, |X2| <=
 *       2^-11 |X|: 4 u M; |Xr| <= max(2^-22 |X|, 2^-25) (binary16 subnormals are honoured by the
 *       conversion and by the matrix pipe, tools/microbench/mfma_f16_scan.hip): 4 u M plus the
 *       CONSTANT 2^-25 sum_k |W'_k| 2^s; Wr is known here exactly, element by element: an
 *       element whose |Wr_k| Xmax_k (Xmax = 255, 255^2, 2^ec) stays below 0.02 score units is
 *       charged that constant, the others relatively, rho = max_k |Wr_k| / |W_k| (2^-22 = 4 u
 *       when both parts of the element are normal numbers):                (8.01 u + rho) M
 *   (c) the accumulation error of the six chained MFMAs.  Every product of two binary16 values
 *       is exact in fp32 (22 bits), so MFMA i returns the exact sum of its 16 products and its C
 *       input up to e_i <= eps (m_i + |C_i|), m_i = the sum of the |products| it adds.  The chain
 *       adds the small products first -- (1,2) and (2,1) of both K blocks, m_i <= 2^-11 M_kb
 *       (1 + 2^-11) -- then the (1,1) products of K block 0 and of K block 1 (M_0 + M_1 = M);
 *       |C_i| <= sum_{j<i} m_j (1 + eps)^i, so  sum e_i <= eps sum_j m_j (1 + number of MFMAs
 *       after j) <= eps (2 + 6 * 2 * 2^-11 * 1.001) M <= 2.01 eps M.
 *       eps itself is not documented for v_mfma_f32_32x32x16_f16.  Two models cover what a
 *       17-term fp32 adder can do: (A) the terms are added one after the other, in any order,
 *       each addition rounded OR truncated to fp32: error <= 16 * 2^-23 = 32 u of sum |terms|;
 *       (B) the terms are aligned to the largest exponent, truncated to a 24-bit grid, summed
 *       exactly and rounded once: 16 * 2 u + 2 u = 34 u.  eps <= 34 u is assumed -- 6.5 times
 *       the worst value tools/microbench/mfma_f16_scan.hip measures on cancelling data
 *       (5.2 u):                                                           68.4 u M
 *   (d) a = 2 var mean rounded to fp32 (b = -var is exact; c is rounded UP to a value its two
 *       parts represent exactly, below):                                    1 u M
 * Together K_i u M with K_i = 79 + rho_i / u per density (78.4 rounded up; 83 when every element
 * keeps both parts normal); a density whose K_i would pass SSW_MFMA_K = 96 goes on the
 * exact-form list.  An inert row is a = b = 0 with -65504 in the two spare K slots 13 and 14,
 * which X holds at 32768: key' = -4.3e9.
 * tests/test_gpu_mfma_bound.py checks the resulting bound on the device on stress inputs and on
 * inputs built to cancel (x = 2 mean: every a_j x_j + b_j x_j^2 pair cancels, M is 8 R while the
 * value is det - R), tests/test_scan_bound.py replays (a) and (b) exactly on the CPU (everything
 * but the adder), and tests/test_gpu_scan_agreement.py checks that the matrix-core scan and the
 * vector-unit scan (whose bound IS replayed on the CPU in full) give identical top-N blocks.
 * The reference's own fp32 value is within 13 u |det| + 17 u S of the real number, and
 * M <= |det - d0| + 6 R + 3 S, S <= |det - d0| + |value - d0| as for recq, which gives the
 * per-density constant (4 K_i + 17) |det - d0| + 6 K_i R + 13 |det| (in u) folded into c below
 * with the constants of (b), with slack, and the term (3 K + 17) u |value - d0| the kernel adds
 * (with K = SSW_MFMA_K) to the one key it uses as a bound.  Densities whose constant exceeds
 * SSW_MFMA_MAX_BIAS score units keep an inert record and are evaluated in the exact form
 * (exlistm); so is a density that alone forces a scale s at which others lose their second part
 * (a floored variance of 5e7 next to variances of 2). */
#define SSW_MFMA_K 96.0
#define SSW_MFMA_K0 79.0
#define SSW_MFMA_MAX_BIAS 64.0
#define SSW_MFMA_XMAX 255.0
#define SSW_MFMA_ABS_UNIT 0.02

/* the K index of MFMA operand slot k (record float slot): 0..12 a | 15 c | 16..28 b */
static double
mfma_slot_xmax(int k, double xconst)
{
    if (k < SSW_MAX_VECLEN)
        return SSW_MFMA_XMAX;
    if (k == SSW_REC_DET)
        return xconst;
    return SSW_MFMA_XMAX * SSW_MFMA_XMAX;
}

/* One density under scale 2^-s (and 2^-ec on the constant): its error constant K_i, the additive
 * constant of (b) in score units, and whether every scaled element stays finite in binary16.
 * q = the unscaled record (a, c WITHOUT bias, b). */
static int
mfma_density_terms(const double *q, int s, int ec, double *K_i, double *add)
{
    const double u24 = 1.0 / 16777216.0, scale = ldexp(1.0, s);
    double rho = 0.0, cst = 0.0, sumw = 0.0;
    int k;
    for (k = 0; k < SSW_REC_FLOATS; ++k) {
        uint16_t p[2];
        double w, wr;
        float wf;
        if (q[k] == 0.0)
            continue;
        wf = (float)ldexp(q[k], -(s + (k == SSW_REC_DET ? ec : 0)));
        if (!(fabsf(wf) <= 32768.0f))
            return -1;
        if (k == SSW_REC_DET) /* the constant is rounded up to what its parts represent: no error */
            continue;
        f16_split2(wf, p);
        w = (double)wf;
        wr = fabs(w - ((double)f16_to_float(p[0]) + (double)f16_to_float(p[1])));
        /* (the float rounding of the scaled element itself: q 2^-s is exact unless q 2^-s is a
         * float subnormal, which |q| >= 2^-100 rules out for every sane model) */
        if (wr * mfma_slot_xmax(k, 0.0) * scale <= SSW_MFMA_ABS_UNIT)
            cst += wr * mfma_slot_xmax(k, 0.0) * scale;
        else if (wr / fabs(w) > rho)
            rho = wr / fabs(w);
        sumw += fabs(w); /* X residual below the binary16 grid: 2^-25 per element */
    }
    cst += ldexp(sumw, -25) * scale;
    *K_i = SSW_MFMA_K0 + rho / u24;
    *add = cst;
    return 0;
}

static int
ssw_host_build_mfma_records(ssw_host_model_t *h)
{
    const int ncbf = h->n_cb * h->n_feat;
    const size_t nrec = (size_t)ncbf * h->n_density;
    const double u24 = 1.0 / 16777216.0;
    int cbf, d, j, rb, kb, p, l, e;
    double *qd;       /* [n_density][SSW_REC_FLOATS] unscaled a, c (no bias), b of one cbf */
    double *geo;      /* [n_density][4]: delta, R, det, max |a|,|b| */
    unsigned char *inert;

    h->recqm = NULL;
    h->exlistm = NULL;
    h->wfrag = NULL;
    h->rec28 = NULL;
    h->n_exact_form_m = 0;
    if (h->n_density != 128)
        return 0; /* the MFMA scan is built for 128 densities; other shapes use the FMA scan */
    h->recqm = (float *)calloc(nrec * SSW_REC_FLOATS, sizeof(float));
    h->exlistm = (uint32_t *)calloc((size_t)ncbf * SSW_EXLIST_STRIDE, sizeof(uint32_t));
    h->wfrag = (uint16_t *)calloc((size_t)ncbf * SSW_WFRAG_PER_CBF, sizeof(uint16_t));
    h->rec28 = (float *)calloc(nrec * 28, sizeof(float));
    if (h->rec28 != NULL) {
        size_t i;
        for (i = 0; i < nrec; ++i) {
            const float *r = h->rec + i * SSW_REC_FLOATS;
            float *q28 = h->rec28 + i * 28;
            for (j = 0; j < 13; ++j) {
                q28[j] = r[j];
                q28[14 + j] = r[SSW_REC_VAR + j];
            }
            q28[13] = r[SSW_REC_DET];
        }
    }
    qd = (double *)calloc((size_t)h->n_density * SSW_REC_FLOATS, sizeof(double));
    geo = (double *)calloc((size_t)h->n_density * 4, sizeof(double));
    inert = (unsigned char *)calloc((size_t)h->n_density, 1);
    if (!h->recqm || !h->exlistm || !h->wfrag || !h->rec28 || !qd || !geo || !inert) {
        free(qd);
        free(geo);
        free(inert);
        ssw_set_error("out of memory building the MFMA scan records");
        return -1;
    }
    for (cbf = 0; cbf < ncbf; ++cbf) {
        uint32_t *xl = h->exlistm + (size_t)cbf * SSW_EXLIST_STRIDE;
        float *d0rec = h->recd0 + (size_t)cbf * SSW_REC_FLOATS;
        const float d0 = d0rec[0];
        int s = 0, ec = 0, round;
        memset(inert, 0, (size_t)h->n_density);
        memset(qd, 0, sizeof(double) * (size_t)h->n_density * SSW_REC_FLOATS);
        for (d = 0; d < h->n_density; ++d) {
            const float *r = h->rec + ((size_t)cbf * h->n_density + d) * SSW_REC_FLOATS;
            double *q = qd + (size_t)d * SSW_REC_FLOATS, *g = geo + (size_t)d * 4;
            const double det = r[SSW_REC_DET], delta = det - (double)d0;
            double R = 0.0, mw = 0.0, bias0;
            int finite = isfinite(det);
            for (j = 0; j < SSW_MAX_VECLEN; ++j) {
                double mean = r[j], var = r[SSW_REC_VAR + j];
                R += fabs(var) * mean * mean;
                finite = finite && isfinite(mean) && isfinite(var) && var >= 0.0;
                /* (as floats: these are the values whose parts the matrix cores multiply) */
                q[j] = (double)(float)(2.0 * var * mean);
                q[SSW_REC_VAR + j] = (double)(float)(-var);
                if (fabs(q[j]) > mw)
                    mw = fabs(q[j]);
                if (fabs(q[SSW_REC_VAR + j]) > mw)
                    mw = fabs(q[SSW_REC_VAR + j]);
            }
            g[0] = delta, g[1] = R, g[2] = det, g[3] = mw;
            /* intrinsically too ill-conditioned for the scan whatever the scale */
            bias0 = 1.05 * u24
                * ((4.0 * SSW_MFMA_K0 + 18.0) * fabs(delta) + (6.0 * SSW_MFMA_K0 + 2.0) * R
                   + 14.0 * fabs(det));
            if (!finite || !(bias0 <= SSW_MFMA_MAX_BIAS) || !(mw < 1.0e30))
                inert[d] = 1;
        }
        /* the scale: the largest |a|, |b| of the live densities just fits 2^15; a density that
         * fails at that scale only because of its lost second parts is what the scale-setter
         * costs, and the scale-setter goes on the exact-form list instead */
        for (round = 0; round <= h->n_density; ++round) {
            double mw = 0.0, mc = 0.0;
            int setter = -1, n_fail = 0;
            for (d = 0; d < h->n_density; ++d)
                if (!inert[d]) {
                    const double *g = geo + (size_t)d * 4;
                    if (g[3] > mw) {
                        mw = g[3];
                        setter = d;
                    }
                    /* |c| <= |delta| + R + bias */
                    if (fabs(g[0]) + g[1] + SSW_MFMA_MAX_BIAS > mc)
                        mc = fabs(g[0]) + g[1] + SSW_MFMA_MAX_BIAS;
                }
            s = 0;
            if (mw > 0.0) {
                (void)frexp(mw / 32768.0, &s); /* mw / 32768 = f 2^s, f in [0.5, 1): mw 2^-s <= 32768 */
                if (s < -8)
                    s = -8;
            }
            ec = 0;
            if (mc > 0.0) {
                (void)frexp(ldexp(mc, -s) / 32768.0, &ec);
                if (ec < 0)
                    ec = 0;
                if (ec > 15)
                    ec = 15;
            }
            for (d = 0; d < h->n_density; ++d)
                if (!inert[d]) {
                    const double *g = geo + (size_t)d * 4;
                    double *q = qd + (size_t)d * SSW_REC_FLOATS;
                    double K_i, add, bias;
                    q[SSW_REC_DET] = g[0] - g[1]; /* c without its bias */
                    if (mfma_density_terms(q, s, ec, &K_i, &add) < 0 || K_i > SSW_MFMA_K) {
                        ++n_fail;
                        continue;
                    }
                    bias = 1.05
                        * (u24
                               * ((4.0 * K_i + 18.0) * fabs(g[0]) + (6.0 * K_i + 2.0) * g[1]
                                  + 14.0 * fabs(g[2]))
                           + add);
                    if (!(bias <= SSW_MFMA_MAX_BIAS))
                        ++n_fail;
                }
            if (n_fail == 0 || setter < 0)
                break;
            /* would the others pass without the scale-setter?  Try: drop it and look again; when
             * the scale cannot shrink any more (s at its floor) the failures are their own */
            {
                double next = 0.0;
                int s2 = 0;
                for (d = 0; d < h->n_density; ++d)
                    if (!inert[d] && d != setter && geo[(size_t)d * 4 + 3] > next)
                        next = geo[(size_t)d * 4 + 3];
                if (next > 0.0)
                    (void)frexp(next / 32768.0, &s2);
                if (next > 0.0 && s2 < s && s > -8) {
                    inert[setter] = 1;
                    continue;
                }
            }
            /* the scale is not to blame: the failing densities go on the list themselves */
            for (d = 0; d < h->n_density; ++d)
                if (!inert[d]) {
                    const double *g = geo + (size_t)d * 4;
                    double *q = qd + (size_t)d * SSW_REC_FLOATS;
                    double K_i, add, bias;
                    q[SSW_REC_DET] = g[0] - g[1];
                    if (mfma_density_terms(q, s, ec, &K_i, &add) < 0 || K_i > SSW_MFMA_K) {
                        inert[d] = 1;
                        continue;
                    }
                    bias = 1.05
                        * (u24
                               * ((4.0 * K_i + 18.0) * fabs(g[0]) + (6.0 * K_i + 2.0) * g[1]
                                  + 14.0 * fabs(g[2]))
                           + add);
                    if (!(bias <= SSW_MFMA_MAX_BIAS))
                        inert[d] = 1;
                }
            /* (the scale may now be larger than the survivors need: one more look) */
        }
        d0rec[1] = (float)ldexp(1.0, s);   /* key = key' * this */
        d0rec[2] = (float)ldexp(1.0, -s);  /* key' = key * this (exact-form densities) */
        d0rec[3] = (float)ldexp(1.0, ec);  /* the constant's slot of X */
        for (d = 0; d < h->n_density; ++d) {
            float *q = h->recqm + ((size_t)cbf * h->n_density + d) * SSW_REC_FLOATS;
            const double *g = geo + (size_t)d * 4;
            double *qq = qd + (size_t)d * SSW_REC_FLOATS;
            double K_i = 0.0, add = 0.0, bias, cc;
            float cf;
            if (!inert[d]) {
                qq[SSW_REC_DET] = g[0] - g[1];
                if (mfma_density_terms(qq, s, ec, &K_i, &add) < 0 || K_i > SSW_MFMA_K)
                    inert[d] = 1; /* (cannot happen after the loop above; kept as a guard) */
            }
            if (inert[d]) {
                xl[1 + xl[0]++] = (uint32_t)d;
                q[SSW_REC_DET] = -3.0e38f; /* a = b = 0: the key stays out of the way */
                ++h->n_exact_form_m;
                continue;
            }
            bias = 1.05
                * (u24
                       * ((4.0 * K_i + 18.0) * fabs(g[0]) + (6.0 * K_i + 2.0) * g[1]
                          + 14.0 * fabs(g[2]))
                   + add);
            for (j = 0; j < SSW_MAX_VECLEN; ++j) {
                q[j] = (float)qq[j];
                q[SSW_REC_VAR + j] = (float)qq[SSW_REC_VAR + j];
            }
            cc = g[0] - g[1] + bias;
            /* the constant as the matrix cores will see it: scaled, cut in two parts, rounded
             * UP where the cut loses something */
            {
                uint16_t pp[2];
                float cs = (float)ldexp(cc, -(s + ec));
                double back;
                int tries;
                if ((double)cs < ldexp(cc, -(s + ec)))
                    cs = nextafterf(cs, INFINITY);
                for (tries = 0; tries < 64; ++tries) {
                    f16_split2(cs, pp);
                    back = (double)f16_to_float(pp[0]) + (double)f16_to_float(pp[1]);
                    if (back >= ldexp(cc, -(s + ec)))
                        break;
                    cs = nextafterf(cs, INFINITY);
                }
                if (back < ldexp(cc, -(s + ec))) {
                    /* 64 float steps do not always reach a value the two binary16
                     * parts hold from above -- below 2^-7 the second part is a binary16
                     * subnormal on a 2^-24 grid, hundreds of float ulps wide -- and a key that
                     * is not a proven upper bound is the one failure the exact pass cannot
                     * catch.  Such a density (constant nearly cancelled: none in en-us / fr-fr,
                     * whose smallest |c'| is 0.38) goes on the exact-form list instead. */
                    inert[d] = 1;
                    xl[1 + xl[0]++] = (uint32_t)d;
                    for (j = 0; j < SSW_MAX_VECLEN; ++j)
                        q[j] = q[SSW_REC_VAR + j] = 0.0f;
                    q[SSW_REC_DET] = -3.0e38f;
                    ++h->n_exact_form_m;
                    continue;
                }
                cf = (float)ldexp(back, s + ec);
                if ((double)cf < ldexp(back, s + ec))
                    cf = nextafterf(cf, INFINITY);
                q[SSW_REC_DET] = cf; /* what the two parts add up to, unscaled (for the tests) */
                qq[SSW_REC_DET] = back; /* scaled by 2^-(s + ec): what goes into the fragments */
            }
        }
        /* A fragments: K index k of the MFMA = float slot k of the record (a at 0..12, c at 15
         * against X = 2^ec, b at 16..28 against x^2; the other slots are zero on both sides) */
        for (rb = 0; rb < 4; ++rb)
            for (kb = 0; kb < 2; ++kb)
                for (l = 0; l < 64; ++l)
                    for (e = 0; e < 8; ++e) {
                        const int dens = 32 * rb + (l & 31), k = 16 * kb + 8 * (l >> 5) + e;
                        const double *qq = qd + (size_t)dens * SSW_REC_FLOATS;
                        uint16_t parts[2] = { 0, 0 };
                        if (inert[dens]) {
                            if (k == 13 || k == 14)
                                parts[0] = 0xfbffu; /* -65504, against X = 32768 */
                        } else {
                            const float wf = k == SSW_REC_DET ? (float)qq[k]
                                                               : (float)ldexp(qq[k], -s);
                            f16_split2(wf, parts);
                        }
                        for (p = 0; p < 2; ++p)
                            h->wfrag[(size_t)cbf * SSW_WFRAG_PER_CBF
                                     + ((((size_t)rb * 2 + kb) * 2 + p) * 64 + l) * 8 + e]
                                = parts[p];
                    }
    }
    free(qd);
    free(geo);
    free(inert);
    return 0;
}

int
ssw_host_build_records(ssw_host_model_t *h)
{
    const int ncbf = h->n_cb * h->n_feat;
    const size_t nrec = (size_t)ncbf * h->n_density;
    const double u24 = 1.0 / 16777216.0;
    const float *mp = h->mean, *vp = h->var;
    float *dets;
    int c, f, d, j, cbf;

    if (h->n_density > SSW_EXLIST_STRIDE - 1) {
        ssw_set_error("%d densities per codebook: at most %d are supported", h->n_density,
                      SSW_EXLIST_STRIDE - 1);
        return -1;
    }
    h->rec = (float *)calloc(nrec * SSW_REC_FLOATS, sizeof(float));
    h->recq = (float *)calloc(nrec * SSW_REC_FLOATS, sizeof(float));
    h->recd0 = (float *)calloc((size_t)ncbf * SSW_REC_FLOATS, sizeof(float));
    h->exlist = (uint32_t *)calloc((size_t)ncbf * SSW_EXLIST_STRIDE, sizeof(uint32_t));
    dets = (float *)malloc(sizeof(float) * (size_t)h->n_density);
    if (!h->rec || !h->recq || !h->recd0 || !h->exlist || !dets) {
        free(dets);
        ssw_set_error("out of memory building the Gaussian records");
        return -1;
    }
    for (c = 0; c < h->n_cb; ++c)
        for (f = 0; f < h->n_feat; ++f)
            for (d = 0; d < h->n_density; ++d) {
                float *r = h->rec + (((size_t)c * h->n_feat + f) * h->n_density + d) * SSW_REC_FLOATS;
                for (j = 0; j < h->veclen[f]; ++j) {
                    r[j] = *mp++;
                    r[SSW_REC_VAR + j] = *vp++;
                }
                r[SSW_REC_DET] = h->det[((size_t)c * h->n_feat + f) * h->n_density + d];
            }
    h->n_exact_form = 0;
    for (cbf = 0; cbf < ncbf; ++cbf) {
        uint32_t *xl = h->exlist + (size_t)cbf * SSW_EXLIST_STRIDE;
        float d0;
        for (d = 0; d < h->n_density; ++d)
            dets[d] = h->rec[((size_t)cbf * h->n_density + d) * SSW_REC_FLOATS + SSW_REC_DET];
        qsort(dets, (size_t)h->n_density, sizeof(float), cmp_float);
        d0 = dets[h->n_density / 2];
        h->recd0[(size_t)cbf * SSW_REC_FLOATS] = d0;
        for (d = 0; d < h->n_density; ++d) {
            const float *r = h->rec + ((size_t)cbf * h->n_density + d) * SSW_REC_FLOATS;
            float *q = h->recq + ((size_t)cbf * h->n_density + d) * SSW_REC_FLOATS;
            const double det = r[SSW_REC_DET], delta = det - (double)d0;
            double R = 0.0, bias, cc;
            int finite = isfinite(det);
            float cf;
            for (j = 0; j < SSW_MAX_VECLEN; ++j) {
                double mean = r[j], var = r[SSW_REC_VAR + j];
                R += fabs(var) * mean * mean;
                finite = finite && isfinite(mean) && isfinite(var) && var >= 0.0;
            }
            bias = 1.05 * u24 * (126.0 * fabs(delta) + 164.0 * R + 14.0 * fabs(det));
            if (!finite || !(bias <= 4.0)) {
                xl[1 + xl[0]++] = (uint32_t)d;
                q[SSW_REC_DET] = -3.0e38f; /* a = b = 0: the key stays out of the way */
                ++h->n_exact_form;
                continue;
            }
            for (j = 0; j < SSW_MAX_VECLEN; ++j) {
                q[j] = (float)(2.0 * (double)r[SSW_REC_VAR + j] * (double)r[j]);
                q[SSW_REC_VAR + j] = -r[SSW_REC_VAR + j];
            }
            cc = delta - R + bias;
            cf = (float)cc;
            if ((double)cf < cc)
                cf = nextafterf(cf, INFINITY);
            q[SSW_REC_DET] = cf;
        }
    }
    free(dets);
    return ssw_host_build_mfma_records(h);
}


ssw_host_model_t *
ssw_host_model_load(const char *mdef, const char *means, const char *variances,
                    const char *sendump, const char *mixw, const char *tmat,
                    const ssw_config_t *cfg)
{
    ssw_host_model_t *h = (ssw_host_model_t *)calloc(1, sizeof(*h));
    lbase_t lb;
    int i, n;

    if (h == NULL)
        return NULL;
    if (cfg)
        h->cfg = *cfg;
    else
        ssw_config_defaults(&h->cfg);
    if (!(h->cfg.logbase > 1.0)) {
        ssw_set_error("logbase must be > 1");
        goto bad;
    }
    lb.base = h->cfg.logbase;
    lb.inv_ln_base = 1.0 / log(h->cfg.logbase);
    n = build_logadd8(&lb, SSW_SENSCR_SHIFT, h->logadd8);
    if (n < 0) { /* same refusal as src/ptm_mgau.c:739-743 */
        ssw_set_error("log base %f too small for an 8-bit add table", h->cfg.logbase);
        goto bad;
    }
    h->logadd8_size = n;
    h->zero8 = (int32_t)0x80000000 >> (SSW_SENSCR_SHIFT + 2);

    if (means == NULL || variances == NULL) {
        ssw_set_error("means and variances are required");
        goto bad;
    }
    if (mdef && load_mdef(h, mdef) < 0)
        goto bad;
    if (load_gaussians(h, &lb, means, variances) < 0)
        goto bad;
    if (ssw_host_build_records(h) < 0)
        goto bad;
    if (tmat && load_tmat(h, &lb, tmat) < 0)
        goto bad;
    if (sendump) {
        if (h->n_sen == 0) {
            ssw_set_error("a sendump needs the mdef for its senone count");
            goto bad;
        }
        if (load_sendump(h, sendump) < 0)
            goto bad;
    }
    if (mixw && load_mixw(h, &lb, mixw, sendump == NULL) < 0)
        goto bad;
    if (h->sen2cb == NULL && h->n_sen) {
        ssw_set_error("senone to codebook map needs the mdef");
        goto bad;
    }
    for (i = 0; i < h->n_sen; ++i)
        if (h->sen2cb[i] < 0 || h->sen2cb[i] >= h->n_cb) {
            /* PTM needs one codebook per CI phone (src/ptm_mgau.c:760-764) */
            ssw_set_error("senone %d maps to codebook %d of %d", i, h->sen2cb[i], h->n_cb);
            goto bad;
        }
    if (h->cfg.topn < 1 || h->cfg.topn > h->n_density)
        h->cfg.topn = h->n_density;
    return h;
bad:
    ssw_host_model_free(h);
    return NULL;
}

void
ssw_host_model_free(ssw_host_model_t *h)
{
    if (h == NULL)
        return;
    free(h->mean);
    free(h->var);
    free(h->det);
    free(h->pid_memo);
    free(h->rec);
    free(h->recq);
    free(h->recd0);
    free(h->exlist);
    free(h->recqm);
    free(h->exlistm);
    free(h->wfrag);
    free(h->rec28);
    free(h->sseq);
    free(h->sen2cb);
    free(h->phone_ssid);
    free(h->phone_tmat);
    free(h->cd_tree);
    free(h->ci_filler);
    if (h->ciname) {
        int i;
        for (i = 0; i < h->n_ciphone; ++i)
            free(h->ciname[i]);
        free(h->ciname);
    }
    free(h->tp);
    free(h->ptm_mixw);
    free(h->ms_pdf);
    free(h);
}
