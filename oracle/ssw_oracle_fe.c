/*
 * ssw_oracle_fe.c -- CPU restatement of the reference's acoustic front end and dynamic-feature
 * computation, whole-utterance mode only.
 *
 * TEST INFRASTRUCTURE ONLY (see ssw_oracle.h).  It exists for one purpose: to turn
 * tests/data/goforward.wav into the feature rows the reference feeds its scorer, so that the
 * reference's recorded alignment of that file can pin the oracle's
 * scoring + Viterbi arithmetic end to end.  The front end is outside the accelerated path.
 *
 * Follows, for the configuration the en-us model uses (16 kHz, 25.625 ms window, 100 frames/s,
 * 512-point FFT, 20 mel filters 130-3700 Hz, unit-area, rounded to DFT points, noise removal,
 * DCT-II, lifter 22, batch CMN, 1s_c_d_dd):
 *   src/fe_interface.c:86-330 (parameters), :560-690 (whole-buffer framing, fe_end)
 *   src/fe_sigproc.c:70-217 (mel filters, DCT basis, lifter), :219-300 (pre-emphasis, Hamming),
 *                    :445-560 (real FFT, power spectrum, mel spectrum), :572-690 (log, DCT-II)
 *   src/fe_noise.c:247-327 (fe_remove_noise and helpers)
 *   src/cmn.c:168-230 (batch CMN), src/feat.c:589-632 (1s_c_d_dd), :977-1008 (utterance padding)
 * frame_t / powspec_t / window_t are float64 and mfcc_t is float32 in the reference
 * (include/soundswallower/fe_type.h:42-44, fe.h:72); the mixed-precision expressions below keep
 * the reference's types operand by operand.
 */
#include "ssw_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct fe_s {
    float sampling_rate, window_length, alpha;
    int frame_rate, frame_shift, frame_size, fft_size, fft_order, ncep, nfilt, lifter_val;
    float lowerf, upperf;
    int unit_area, round_filters, remove_noise, legacy_dct;
    /* tables */
    double *hamming, *ccc, *sss;
    float *filt_coeffs, *mel_cosine, *lifter;
    float sqrt_inv_n, sqrt_inv_2n;
    short *spec_start, *filt_start, *filt_width;
    /* per-utterance state */
    float *spch;
    double *frame, *spec, *mfspec;
    float prior;
    /* noise tracker (struct noise_stats_s, src/fe_noise.c:76-109) */
    double *n_power, *n_noise, *n_floor, *n_peak, *n_signal, *n_gain;
    int n_undefined;
} fe_t;

static float
fe_mel(float x) /* fe_sigproc.c:70-76, neutral warp */
{
    return (float)(2595.0 * log10(1.0 + x / 700.0));
}

static float
fe_melinv(float x) /* fe_sigproc.c:78-83 */
{
    return (float)(700.0 * (pow(10.0, x / 2595.0) - 1.0));
}

static void
filter_freqs(const fe_t *fe, int i, float melbw, float melmin, float fftfreq, float freqs[3])
{
    int j;
    for (j = 0; j < 3; ++j) {
        freqs[j] = fe_melinv((i + j) * melbw + melmin);
        if (fe->round_filters)
            freqs[j] = ((int)(freqs[j] / fftfreq + 0.5)) * fftfreq;
    }
}

/* fe_build_melfilters, fe_sigproc.c:85-182 */
static void
build_melfilters(fe_t *fe)
{
    float melmin = fe_mel(fe->lowerf), melmax = fe_mel(fe->upperf);
    float melbw = (melmax - melmin) / (fe->nfilt + 1);
    float fftfreq = fe->sampling_rate / (float)fe->fft_size;
    int n_coeffs = 0, i, j;

    fe->spec_start = calloc(fe->nfilt, sizeof(short));
    fe->filt_start = calloc(fe->nfilt, sizeof(short));
    fe->filt_width = calloc(fe->nfilt, sizeof(short));
    for (i = 0; i < fe->nfilt; ++i) {
        float freqs[3];
        filter_freqs(fe, i, melbw, melmin, fftfreq, freqs);
        fe->spec_start[i] = -1;
        for (j = 0; j < fe->fft_size / 2 + 1; ++j) {
            float hz = j * fftfreq;
            if (hz < freqs[0])
                continue;
            else if (hz > freqs[2] || j == fe->fft_size / 2) {
                fe->filt_width[i] = (short)(j - fe->spec_start[i]);
                fe->filt_start[i] = (short)n_coeffs;
                n_coeffs += fe->filt_width[i];
                break;
            }
            if (fe->spec_start[i] == -1)
                fe->spec_start[i] = (short)j;
        }
    }
    fe->filt_coeffs = malloc(sizeof(float) * (size_t)(n_coeffs > 0 ? n_coeffs : 1));
    n_coeffs = 0;
    for (i = 0; i < fe->nfilt; ++i) {
        float freqs[3];
        filter_freqs(fe, i, melbw, melmin, fftfreq, freqs);
        for (j = 0; j < fe->filt_width[i]; ++j) {
            float hz = (fe->spec_start[i] + j) * fftfreq;
            float loslope = (hz - freqs[0]) / (freqs[1] - freqs[0]);
            float hislope = (freqs[2] - hz) / (freqs[2] - freqs[1]);
            if (fe->unit_area) {
                loslope *= 2 / (freqs[2] - freqs[0]);
                hislope *= 2 / (freqs[2] - freqs[0]);
            }
            fe->filt_coeffs[n_coeffs++] = loslope < hislope ? loslope : hislope;
        }
    }
}

static fe_t *
fe_new(int nfilt, float lowerf, float upperf, int lifter_val, int remove_noise, int legacy)
{
    fe_t *fe = calloc(1, sizeof(*fe));
    int i, j;
    double freqstep;

    fe->sampling_rate = 16000;
    fe->frame_rate = 100;
    fe->window_length = (float)0.025625;
    fe->alpha = (float)0.97;
    fe->ncep = 13;
    fe->nfilt = nfilt;
    fe->lowerf = lowerf;
    fe->upperf = upperf;
    fe->lifter_val = lifter_val;
    fe->unit_area = 1;
    fe->round_filters = 1;
    fe->remove_noise = remove_noise;
    fe->legacy_dct = legacy;
    /* fe_interface.c:264-265 */
    fe->frame_shift = (int)(fe->sampling_rate / fe->frame_rate + 0.5);
    fe->frame_size = (int)(fe->window_length * fe->sampling_rate + 0.5);
    /* FFT size from the window, fe_interface.c:129-137 */
    {
        int window_samples = (int)(fe->window_length * fe->sampling_rate);
        fe->fft_order = 0;
        fe->fft_size = 1;
        while (fe->fft_size < window_samples) {
            fe->fft_order++;
            fe->fft_size <<= 1;
        }
    }
    fe->hamming = calloc(fe->frame_size / 2, sizeof(double));
    for (i = 0; i < fe->frame_size / 2; ++i) /* fe_create_hamming, fe_sigproc.c:241-252 */
        fe->hamming[i] = 0.54 - 0.46 * cos(2 * M_PI * i / ((double)fe->frame_size - 1.0));
    build_melfilters(fe);
    /* fe_compute_melcosine, fe_sigproc.c:184-217 */
    fe->mel_cosine = calloc((size_t)fe->ncep * fe->nfilt, sizeof(float));
    freqstep = M_PI / fe->nfilt;
    for (i = 0; i < fe->ncep; ++i)
        for (j = 0; j < fe->nfilt; ++j)
            fe->mel_cosine[i * fe->nfilt + j] = (float)cos(freqstep * i * (j + 0.5));
    fe->sqrt_inv_n = (float)sqrt(1.0 / fe->nfilt);
    fe->sqrt_inv_2n = (float)sqrt(2.0 / fe->nfilt);
    if (fe->lifter_val) {
        fe->lifter = calloc(fe->ncep, sizeof(float));
        for (i = 0; i < fe->ncep; ++i)
            fe->lifter[i] = (float)(1 + fe->lifter_val / 2 * sin(i * M_PI / fe->lifter_val));
    }
    fe->ccc = calloc(fe->fft_size / 4, sizeof(double));
    fe->sss = calloc(fe->fft_size / 4, sizeof(double));
    for (i = 0; i < fe->fft_size / 4; ++i) { /* fe_create_twiddle, fe_sigproc.c:447-457 */
        double a = 2 * M_PI * i / fe->fft_size;
        fe->ccc[i] = cos(a);
        fe->sss[i] = sin(a);
    }
    fe->spch = calloc(fe->frame_size, sizeof(float));
    fe->frame = calloc(fe->fft_size, sizeof(double));
    fe->spec = calloc(fe->fft_size, sizeof(double));
    fe->mfspec = calloc(fe->nfilt, sizeof(double));
    fe->n_power = calloc(fe->nfilt, sizeof(double));
    fe->n_noise = calloc(fe->nfilt, sizeof(double));
    fe->n_floor = calloc(fe->nfilt, sizeof(double));
    fe->n_peak = calloc(fe->nfilt, sizeof(double));
    fe->n_signal = calloc(fe->nfilt, sizeof(double));
    fe->n_gain = calloc(fe->nfilt, sizeof(double));
    fe->n_undefined = 1;
    fe->prior = 0;
    return fe;
}

static void
fe_del(fe_t *fe)
{
    free(fe->hamming); free(fe->ccc); free(fe->sss); free(fe->filt_coeffs);
    free(fe->mel_cosine); free(fe->lifter); free(fe->spec_start); free(fe->filt_start);
    free(fe->filt_width); free(fe->spch); free(fe->frame); free(fe->spec); free(fe->mfspec);
    free(fe->n_power); free(fe->n_noise); free(fe->n_floor); free(fe->n_peak);
    free(fe->n_signal); free(fe->n_gain); free(fe);
}

/* fe_spch_to_frame, fe_sigproc.c:276-303: pre-emphasis, zero pad, Hamming */
static void
spch_to_frame(fe_t *fe, int len)
{
    int i;
    fe->frame[0] = (double)fe->spch[0] - (double)fe->prior * fe->alpha;
    for (i = 1; i < len; ++i)
        fe->frame[i] = (double)fe->spch[i] - (double)fe->spch[i - 1] * fe->alpha;
    if (len >= fe->frame_shift)
        fe->prior = fe->spch[fe->frame_shift - 1];
    else
        fe->prior = fe->spch[len - 1];
    memset(fe->frame + len, 0, (size_t)(fe->fft_size - len) * sizeof(double));
    for (i = 0; i < fe->frame_size / 2; ++i) {
        fe->frame[i] = fe->frame[i] * fe->hamming[i];
        fe->frame[fe->frame_size - 1 - i] = fe->frame[fe->frame_size - 1 - i] * fe->hamming[i];
    }
}

/* fe_fft_real, fe_sigproc.c:459-556 */
static void
fft_real(fe_t *fe)
{
    double *x = fe->frame, xt;
    int m = fe->fft_order, n = fe->fft_size, i, j, k;

    j = 0;
    for (i = 0; i < n - 1; ++i) {
        if (i < j) {
            xt = x[j];
            x[j] = x[i];
            x[i] = xt;
        }
        k = n / 2;
        while (k <= j) {
            j -= k;
            k /= 2;
        }
        j += k;
    }
    for (i = 0; i < n; i += 2) {
        xt = x[i];
        x[i] = (xt + x[i + 1]);
        x[i + 1] = (xt - x[i + 1]);
    }
    for (k = 1; k < m; ++k) {
        int n4 = k - 1, n2 = k, n1 = k + 1;
        for (i = 0; i < n; i += (1 << n1)) {
            xt = x[i];
            x[i] = (xt + x[i + (1 << n2)]);
            x[i + (1 << n2)] = (xt - x[i + (1 << n2)]);
            x[i + (1 << n2) + (1 << n4)] = -x[i + (1 << n2) + (1 << n4)];
            for (j = 1; j < (1 << n4); ++j) {
                int i1 = i + j, i2 = i + (1 << n2) - j, i3 = i + (1 << n2) + j,
                    i4 = i + (1 << n2) + (1 << n2) - j;
                double cc = fe->ccc[j << (m - n1)], ss = fe->sss[j << (m - n1)];
                double t1 = x[i3] * cc + x[i4] * ss;
                double t2 = x[i3] * ss - x[i4] * cc;
                x[i4] = (x[i2] - t2);
                x[i3] = (-x[i2] - t2);
                x[i2] = (x[i1] - t1);
                x[i1] = (x[i1] + t1);
            }
        }
    }
}

/* fe_remove_noise, src/fe_noise.c:247-327 with its helpers :111-176 */
static void
remove_noise(fe_t *fe)
{
    const double lambda_power = 0.7, lambda_a = 0.995, lambda_b = 0.5, lambda_t = 0.85,
                 mu_t = 0.2, max_gain = 20, inv_max_gain = 1.0 / 20;
    const double comp_power = 1 - 0.7, comp_a = 1 - 0.995, comp_b = 1 - 0.5;
    double *mf = fe->mfspec;
    int n = fe->nfilt, i, j;

    if (fe->n_undefined) {
        for (i = 0; i < n; ++i) {
            fe->n_power[i] = mf[i];
            fe->n_noise[i] = mf[i] / max_gain;
            fe->n_floor[i] = mf[i] / max_gain;
            fe->n_peak[i] = 0.0;
        }
        fe->n_undefined = 0;
    }
    for (i = 0; i < n; ++i)
        fe->n_power[i] = lambda_power * fe->n_power[i] + comp_power * mf[i];
    for (i = 0; i < n; ++i) { /* fe_lower_envelope(power -> noise) */
        if (fe->n_power[i] >= fe->n_noise[i])
            fe->n_noise[i] = lambda_a * fe->n_noise[i] + comp_a * fe->n_power[i];
        else
            fe->n_noise[i] = lambda_b * fe->n_noise[i] + comp_b * fe->n_power[i];
    }
    for (i = 0; i < n; ++i) {
        fe->n_signal[i] = fe->n_power[i] - fe->n_noise[i];
        if (fe->n_signal[i] < 1.0)
            fe->n_signal[i] = 1.0;
    }
    for (i = 0; i < n; ++i) { /* fe_lower_envelope(signal -> floor) */
        if (fe->n_signal[i] >= fe->n_floor[i])
            fe->n_floor[i] = lambda_a * fe->n_floor[i] + comp_a * fe->n_signal[i];
        else
            fe->n_floor[i] = lambda_b * fe->n_floor[i] + comp_b * fe->n_signal[i];
    }
    for (i = 0; i < n; ++i) { /* fe_temp_masking */
        double cur_in = fe->n_signal[i];
        fe->n_peak[i] *= lambda_t;
        if (fe->n_signal[i] < lambda_t * fe->n_peak[i])
            fe->n_signal[i] = fe->n_peak[i] * mu_t;
        if (cur_in > fe->n_peak[i])
            fe->n_peak[i] = cur_in;
    }
    for (i = 0; i < n; ++i)
        if (fe->n_signal[i] < fe->n_floor[i])
            fe->n_signal[i] = fe->n_floor[i];
    for (i = 0; i < n; ++i) {
        if (fe->n_signal[i] < max_gain * fe->n_power[i])
            fe->n_gain[i] = fe->n_signal[i] / fe->n_power[i];
        else
            fe->n_gain[i] = max_gain;
        if (fe->n_gain[i] < inv_max_gain)
            fe->n_gain[i] = inv_max_gain;
    }
    for (i = 0; i < n; ++i) { /* fe_weight_smooth, SMOOTH_WINDOW 4 */
        int l1 = (i - 4) > 0 ? (i - 4) : 0;
        int l2 = (i + 4) < (n - 1) ? (i + 4) : (n - 1);
        double coef = 0;
        for (j = l1; j <= l2; ++j)
            coef += fe->n_gain[j];
        mf[i] = mf[i] * (coef / (l2 - l1 + 1));
    }
}

/* fe_write_frame, fe_sigproc.c:728-738 */
static void
write_frame(fe_t *fe, float *cep)
{
    int i, j, n = fe->fft_size;
    fft_real(fe);
    fe->spec[0] = fe->frame[0] * fe->frame[0];
    for (j = 1; j <= n / 2; ++j)
        fe->spec[j] = fe->frame[j] * fe->frame[j] + fe->frame[n - j] * fe->frame[n - j];
    for (i = 0; i < fe->nfilt; ++i) { /* fe_mel_spec */
        fe->mfspec[i] = 0;
        for (j = 0; j < fe->filt_width[i]; ++j)
            fe->mfspec[i] += fe->spec[fe->spec_start[i] + j] * fe->filt_coeffs[fe->filt_start[i] + j];
    }
    if (fe->remove_noise)
        remove_noise(fe);
    for (i = 0; i < fe->nfilt; ++i) /* fe_mel_cep, LOG_FLOOR 1e-4 */
        fe->mfspec[i] = log(fe->mfspec[i] + 1e-4);
    if (fe->legacy_dct) { /* fe_spec2cep (transform = legacy), fe_sigproc.c:640-670 */
        cep[0] = (float)(fe->mfspec[0] / 2);
        for (j = 1; j < fe->nfilt; ++j)
            cep[0] = (float)(cep[0] + fe->mfspec[j]);
        cep[0] = (float)(cep[0] / (double)fe->nfilt);
        for (i = 1; i < fe->ncep; ++i) {
            cep[i] = 0;
            for (j = 0; j < fe->nfilt; ++j) {
                int beta = j == 0 ? 1 : 2;
                cep[i] = (float)(cep[i] + fe->mfspec[j] * fe->mel_cosine[i * fe->nfilt + j] * beta);
            }
            cep[i] = (float)(cep[i] / ((double)fe->nfilt * 2));
        }
        return;
    }
    /* fe_dct2 (transform = dct), fe_sigproc.c:672-693: float accumulators */
    cep[0] = (float)fe->mfspec[0];
    for (j = 1; j < fe->nfilt; ++j)
        cep[0] = (float)(cep[0] + fe->mfspec[j]);
    cep[0] = cep[0] * fe->sqrt_inv_n;
    for (i = 1; i < fe->ncep; ++i) {
        cep[i] = 0;
        for (j = 0; j < fe->nfilt; ++j)
            cep[i] = (float)(cep[i] + fe->mfspec[j] * fe->mel_cosine[i * fe->nfilt + j]);
        cep[i] = cep[i] * fe->sqrt_inv_2n;
    }
    if (fe->lifter_val) /* fe_lifter */
        for (i = 0; i < fe->ncep; ++i)
            cep[i] = cep[i] * fe->lifter[i];
}

/* Whole-buffer MFCC as acmod_process_full_raw drives it (src/acmod.c:423-455): fe_start,
 * fe_process_int16 over all samples, fe_end for the trailing partial frame.  Returns the number
 * of frames written to cep[max_frames][13]. */
int
orc_fe_mfcc(const int16_t *pcm, size_t n_samps, int nfilt, double lowerf, double upperf,
            int lifter, int remove_noise_flag, int legacy_transform, float *cep, int max_frames)
{
    fe_t *fe = fe_new(nfilt, (float)lowerf, (float)upperf, lifter, remove_noise_flag,
                      legacy_transform);
    int nfr = 0, frame_count, i, k;
    size_t pos;

    if (n_samps < (size_t)fe->frame_size) {
        /* overflow_append then fe_end: one short frame (fe_interface.c:590-591, 760-776) */
        if (n_samps > 0 && max_frames > 0) {
            for (i = 0; i < (int)n_samps; ++i)
                fe->spch[i] = ((float)pcm[i] / 32768.0F) * 32768.0F;
            spch_to_frame(fe, (int)n_samps);
            write_frame(fe, cep);
            nfr = 1;
        }
        fe_del(fe);
        return nfr;
    }
    frame_count = 1 + (int)((n_samps - fe->frame_size) / fe->frame_shift);
    /* first frame: fe_read_frame_int16 */
    for (i = 0; i < fe->frame_size; ++i)
        fe->spch[i] = pcm[i];
    spch_to_frame(fe, fe->frame_size);
    if (nfr < max_frames)
        write_frame(fe, cep + (size_t)nfr * fe->ncep);
    ++nfr;
    pos = (size_t)fe->frame_size;
    for (k = 1; k < frame_count; ++k) { /* fe_shift_frame_int16 */
        int offset = fe->frame_size - fe->frame_shift;
        memmove(fe->spch, fe->spch + fe->frame_shift, (size_t)offset * sizeof(float));
        for (i = 0; i < fe->frame_shift; ++i)
            fe->spch[i + offset] = pcm[pos + i];
        pos += (size_t)fe->frame_shift;
        spch_to_frame(fe, fe->frame_size);
        if (nfr < max_frames)
            write_frame(fe, cep + (size_t)nfr * fe->ncep);
        ++nfr;
    }
    /* create_overflow_frame + fe_end (fe_interface.c:656-686, 760-776): the samples from the
     * start of the next frame to the end, through float / 32768 * 32768 (exact) */
    {
        size_t remaining = n_samps - pos;
        int n_overflow = remaining < (size_t)fe->frame_shift ? (int)remaining : fe->frame_shift;
        int n_ov = fe->frame_size - fe->frame_shift + n_overflow;
        const int16_t *in = pcm + pos - (size_t)(fe->frame_size - fe->frame_shift);
        if (n_ov > 0) {
            for (i = 0; i < n_ov; ++i)
                fe->spch[i] = ((float)in[i] / 32768.0F) * 32768.0F;
            spch_to_frame(fe, n_ov);
            if (nfr < max_frames)
                write_frame(fe, cep + (size_t)nfr * fe->ncep);
            ++nfr;
        }
    }
    fe_del(fe);
    return nfr;
}

/* feat_s2mfc2feat_block_utt for feat = 1s_c_d_dd, cmn = batch ("current"), no varnorm/agc/lda
 * (src/feat.c:977-1008, :589-632; src/cmn.c:168-200).  cep is modified (CMN in place, as the
 * reference does).  out = [n][39] = c | d | dd, which is also the 0-12/13-25/26-38 stream split. */
int
orc_feat_1s_c_d_dd(float *cep, int n, float *out)
{
    return orc_feat_1s_c_d_dd_ex(cep, n, out, 1);
}

/* cmn = 0: "-cmn none", the configuration of the reference's own tests/test_feat.c:51-55, whose
 * golden table tests/_test_feat.res the oracle is pinned to (tests/test_reference_pins.py) */
int
orc_feat_1s_c_d_dd_ex(float *cep, int n, float *out, int cmn)
{
    const int C = 13, W = 3; /* window = FEAT_DCEP_WIN + 1 */
    float sum[13], mean[13];
    float **row;
    int nframe = 0, f, i;

    if (n <= 0)
        return 0;
    memset(sum, 0, sizeof(sum));
    for (f = 0; f < n && cmn; ++f) {
        const float *c = cep + (size_t)f * C;
        if (c[0] < 0) /* "skip zero energy frames" */
            continue;
        for (i = 0; i < C; ++i)
            sum[i] += c[i];
        ++nframe;
    }
    for (i = 0; i < C; ++i)
        mean[i] = cmn ? sum[i] / nframe : 0.0f;
    for (f = 0; f < n && cmn; ++f)
        for (i = 0; i < C; ++i)
            cep[(size_t)f * C + i] -= mean[i];
    /* pad W frames each side with copies of the first / last frame */
    row = malloc(sizeof(float *) * (size_t)(n + 2 * W));
    for (f = 0; f < W; ++f) {
        row[f] = cep;
        row[n + W + f] = cep + (size_t)(n - 1) * C;
    }
    for (f = 0; f < n; ++f)
        row[W + f] = cep + (size_t)f * C;
    for (f = 0; f < n; ++f) {
        float **m = row + W + f;
        float *o = out + (size_t)f * 3 * C;
        memcpy(o, m[0], sizeof(float) * C);
        for (i = 0; i < C; ++i)
            o[C + i] = m[2][i] - m[-2][i];
        for (i = 0; i < C; ++i) {
            float d1 = m[3][i] - m[-1][i];
            float d2 = m[1][i] - m[-3][i];
            o[2 * C + i] = d1 - d2;
        }
    }
    free(row);
    return n;
}
