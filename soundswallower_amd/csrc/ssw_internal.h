/* ssw_internal.h -- shared between the host C loaders and the HIP translation unit. */
#ifndef SSW_INTERNAL_H
#define SSW_INTERNAL_H

#include "ssw_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SSW_SENSCR_SHIFT 10
#define SSW_WORST_SCORE ((int32_t)0xE0000000)
#define SSW_MAX_NEG_ASCR 96
#define SSW_MAX_NEG_MIXW 159
#define SSW_MAX_FEAT 8

/* Gaussian record on the device: one density = 32 floats = one 128-byte line.
 *   [0 .. veclen)  mean      [15] det      [16 .. 16+veclen)  precomputed 1/(2 var) scale
 * Unused slots are 0.0f, which makes a padded dimension an exact no-op (d - 0*0*0 = d). */
#define SSW_REC_FLOATS 32
#define SSW_REC_DET 15
#define SSW_REC_VAR 16
#define SSW_MAX_VECLEN 15
/* per codebook-stream: [0] count, [1..] codewords the quadratic scan leaves to the exact form */
#define SSW_EXLIST_STRIDE 132
/* matrix-core scan: binary16 values per (codebook, stream) in wfrag */
#define SSW_WFRAG_PER_CBF (4 * 2 * 2 * 64 * 8)

/* Host-side model: every table derived exactly as the reference derives it. */
typedef struct ssw_host_model_s {
    ssw_config_t cfg;
    /* Gaussians (gauden_t) */
    int32_t n_cb, n_feat, n_density, veclen_total, n_floored;
    int32_t veclen[SSW_MAX_FEAT], featoff[SSW_MAX_FEAT];
    float *mean, *var, *det; /* file order / [cb][feat][density] */
    /* device-layout Gaussian tables (ssw_host_build_records):
     *   rec     [cb*feat][density][32]  exact records (mean, det, scale)
     *   recq    [cb*feat][density][32]  quadratic-form scan records (a, c, b)
     *   recd0   [cb*feat][32]           [0] = the codebook's reference det d0; matrix-core
     *                                   scan: [1] = 2^s, [2] = 2^-s (its keys are key 2^-s),
     *                                   [3] = 2^ec (the constant's slot of X)
     *   exlist  [cb*feat][SSW_EXLIST_STRIDE] */
    float *rec, *recq, *recd0;
    uint32_t *exlist;
    int32_t n_exact_form;
    /* the same scan for the matrix cores (ssw_k1a_mfma.inc): quadratic-form records whose
     * constant carries the error bound of the split-binary16 MFMA evaluation (recqm, exlistm as
     * recq / exlist; recqm's constant is what its two parts add up to, unscaled), and the
     * records, scaled by 2^-s (the constant by 2^-(s + ec)), cut into two binary16 parts in MFMA
     * A-fragment order:
     *   wfrag [cb*feat][4 row blocks][2 K blocks][2 parts][64 lanes][8] binary16
     * (lane l of a fragment: density 32*rb + l%32, K = 16*kb + 8*(l/32) .. +7) */
    float *recqm;
    uint32_t *exlistm;
    uint16_t *wfrag;
    /* the exact records once more, packed for the matrix-core scan's LDS copy:
     * rec28 [cb*feat][density][28] = mean 0..12 | det 13 | scale 14..26 | 0 */
    float *rec28;
    int32_t n_exact_form_m;
    /* mdef */
    int32_t n_ciphone, n_phone, n_emit_state, n_ci_sen, n_sen, n_tmat, n_sseq, sil;
    uint16_t *sseq;
    int16_t *sen2cb; /* bin_mdef sen2cimap */
    int32_t *phone_ssid, *phone_tmat;
    /* context-dependent phone tree (cd_tree_t, bin_mdef.h:103-114), host byte order */
    int32_t n_cd_tree;
    struct ssw_cd_node_s { int16_t ctx, n_down; int32_t down_or_pid; } *cd_tree;
    char **ciname;       /* [n_ciphone] */
    uint8_t *ci_filler;  /* [n_ciphone] */
    /* tmat */
    uint8_t *tp;
    int32_t tp_n_tmat, tp_n_state;
    /* mixture weights */
    uint8_t *ptm_mixw; /* [feat][density][n_sen], 8-bit (4-bit clustered dumps are expanded) */
    uint8_t *ms_pdf;   /* [sen][feat][density] */
    /* log-add tables */
    uint8_t logadd8[256];
    /* memo of bin_mdef_phone_id_nearest, [pos][base][left][right], -2 = not looked up yet
     * (filled lazily by ssw_phone_id_nearest; racing writers store the same value) */
    int32_t *pid_memo;
    int32_t logadd8_size;   /* entries produced by logmath_init (>= 256) */
    int32_t zero8;          /* logmath zero at shift 10 */
} ssw_host_model_t;

ssw_host_model_t *ssw_host_model_load(const char *mdef, const char *means,
                                      const char *variances, const char *sendump,
                                      const char *mixw, const char *tmat,
                                      const ssw_config_t *cfg);
void ssw_host_model_free(ssw_host_model_t *h);
int ssw_host_build_records(ssw_host_model_t *h);
void ssw_set_error(const char *fmt, ...);

/* Pronunciation dictionary (ssw_lexicon.c) */
struct ssw_dict_s {
    int n_words, cap_words;
    char **word;
    int16_t **pron; /* CI phone ids */
    int *pronlen;
    int *slot;      /* open-addressing hash: word index + 1, 0 = empty */
    int n_slots;
    int filler_start; /* first word that came from the filler dictionary (dict_filler_start) */
    int *alt;         /* dict_nextalt: next alternate pronunciation of the same base word, -1 */
    int *base;        /* dict_basewid */
};
int ssw_dict_find(const struct ssw_dict_s *d, const char *w);

/* First-pass graphs of a batch of texts (ssw_fsg.c), flat arrays the kernel reads.
 * Node indices, leaf ordinals and states are local to their utterance. */
typedef struct ssw_fp_graphs_s {
    int32_t n_utts, n_nodes, n_leaves, n_states, n_in;
    int32_t *node_off, *leaf_off, *state_off; /* [n_utts + 1] */
    uint16_t *senid;    /* [n_nodes][4]: three senones, then the transition matrix id */
    int32_t *pen;       /* [n_nodes] log probability added on entry (logs2prob) */
    int32_t *parent;    /* [n_nodes] predecessor in the phone tree, -1 for word-initial nodes */
    uint32_t *info;     /* [n_nodes] bit 0 root, 1 leaf, 2 exit applies to every right context;
                           8..15 the CI phone shown to neighbours; 16..31 FSG state (roots: the
                           state they hang off, leaves that are not roots: unused) */
    uint64_t *ctxt;     /* [n_nodes] roots: left contexts served; leaves: right contexts served */
    int32_t *leaf_ord;  /* [n_nodes] ordinal among the utterance's leaves, -1 */
    int32_t *leaf_wid;  /* [n_leaves] dictionary word the leaf ends */
    int32_t *leaf_to;   /* [n_leaves] FSG state the word leads to */
    int32_t *leaf_node; /* [n_leaves] the leaf's node */
    int32_t *in_off;    /* [n_states + 1] leaves entering each state, into in_leaf */
    int32_t *in_leaf;   /* [n_in] leaf ordinals, by (left-context phone, ordinal) */
    /* alternates pronounced alike (twins): per member a record
     *   [L, n_ancestors, own element index, offset of its 3 x L rank / key buffers,
     *    L node indices: the ancestors root .. predecessor, then the members in chain order]
     * tw_off [n_utts + 1] into tw; twin_ref [n_nodes] = the node's record offset in its
     * utterance's part of tw, or -1; tw_rk [n_utts] = ints of rank buffers the utterance needs */
    int32_t n_tw;
    int32_t *tw, *tw_off, *twin_ref, *tw_rk;
    int32_t beam, pbeam, wbeam;
    uint64_t uid; /* unique per built set of graphs (never 0): the device copy's cache key */
} ssw_fp_graphs_t;
ssw_fp_graphs_t *ssw_fp_graphs_build(const ssw_model_t *m, const struct ssw_dict_s *d,
                                     const ssw_first_pass_config_t *cfg, int32_t n_utts,
                                     const int32_t *word_off, const char *const *words);
void ssw_fp_graphs_free(ssw_fp_graphs_t *g);
int ssw_host_threads(void);
/* fn(arg, k) for k in [0, n_chunks) on a persistent pool of host threads + the caller */
void ssw_parallel_for(int n_chunks, void (*fn)(void *arg, int chunk), void *arg);

#ifdef __cplusplus
}
#endif
#endif
