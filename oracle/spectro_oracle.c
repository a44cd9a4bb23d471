/*
 * spectro_oracle.c — CPU ORACLE (test infrastructure only; see spectro_oracle.h).
 *
 * Plain-C restatement of the reference CPU algorithm.  Every function cites the
 * reference file:line (under /root/reference/) it follows.  Build with
 * -ffp-contract=off so `a*b + c` is two roundings, as in the Rust source
 * (rustc never contracts); the explicit `mul_add`s of the reference are fma().
 */
#include "spectro_oracle.h"

#include <complex.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static int fail(char *err, size_t errlen, int code, const char *msg) {
    if (err && errlen) snprintf(err, errlen, "%s", msg);
    return code;
}

/* ---- parameter validation ------------------------------------------------ */
/* StftParams::new  src/spectrogram.rs:3479-3506
 * SpectrogramParams::new :4129-4140; MelParams::with_norm :3793-3813;
 * LogParams::new :4071-4077; mel_plan Nyquist check :954-959; n_mels cap :1696-1700;
 * build_mel_filterbank_matrix argument checks :2310-2323 */
int orc_validate(const orc_params *p, char *err, size_t errlen) {
    if (!p) return fail(err, errlen, ORC_INVALID_INPUT, "null params");
    if (p->n_fft == 0) return fail(err, errlen, ORC_INVALID_INPUT, "n_fft must be > 0");
    if (p->hop == 0) return fail(err, errlen, ORC_INVALID_INPUT, "hop_size must be > 0");
    if (p->hop > p->n_fft) return fail(err, errlen, ORC_INVALID_INPUT, "hop_size must be <= n_fft");
    if (p->window_kind < ORC_WIN_RECT || p->window_kind > ORC_WIN_CUSTOM)
        return fail(err, errlen, ORC_INVALID_INPUT, "unknown window kind");
    if (p->window_kind == ORC_WIN_CUSTOM && !p->custom_window)
        return fail(err, errlen, ORC_INVALID_INPUT, "Custom window size must match n_fft");
    if (!(p->sample_rate > 0.0 && isfinite(p->sample_rate)))
        return fail(err, errlen, ORC_INVALID_INPUT, "sample_rate_hz must be finite and > 0");
    if (p->freq_scale == ORC_FREQ_MEL) {
        if (p->n_mels == 0) return fail(err, errlen, ORC_INVALID_INPUT, "n_mels must be > 0");
        if (p->f_min < 0.0) return fail(err, errlen, ORC_INVALID_INPUT, "f_min must be >= 0");
        if (p->f_max <= p->f_min) return fail(err, errlen, ORC_INVALID_INPUT, "f_max must be > f_min");
        if (p->f_max > p->sample_rate * 0.5)
            return fail(err, errlen, ORC_INVALID_INPUT, "mel f_max must be <= Nyquist");
        if (p->n_mels > 10000) return fail(err, errlen, ORC_INVALID_INPUT, "n_mels is unreasonably large");
        if (isinf(p->f_min)) return fail(err, errlen, ORC_INVALID_INPUT, "f_min must be >= 0");
    } else if (p->freq_scale == ORC_FREQ_LOGHZ) { /* LogHzParams::new :3960-3975; new_loghz :1726-1736; build_loghz_matrix :2445-2460 */
        if (p->n_mels == 0) return fail(err, errlen, ORC_INVALID_INPUT, "n_bins must be > 0");
        if (!(p->f_min > 0.0 && isfinite(p->f_min))) return fail(err, errlen, ORC_INVALID_INPUT, "f_min must be finite and > 0");
        if (p->f_max <= p->f_min) return fail(err, errlen, ORC_INVALID_INPUT, "f_max must be > f_min");
        if (p->n_mels > 10000) return fail(err, errlen, ORC_INVALID_INPUT, "n_bins is unreasonably large");
        if (p->f_max > p->sample_rate * 0.5) return fail(err, errlen, ORC_INVALID_INPUT, "f_max must be <= Nyquist");
    } else if (p->freq_scale == ORC_FREQ_ERB) { /* ErbParams::new erb.rs:66-80; erb_plan spectrogram.rs:1014-1022; new_erb :1768-1773 */
        if (p->n_mels < 2) return fail(err, errlen, ORC_INVALID_INPUT, "n_filters must be >= 2 (single filter would cause division by zero)");
        if (p->f_min < 0.0 || isinf(p->f_min)) return fail(err, errlen, ORC_INVALID_INPUT, "f_min must be finite and >= 0");
        if (p->f_max <= p->f_min) return fail(err, errlen, ORC_INVALID_INPUT, "f_max must be > f_min");
        if (p->f_max > p->sample_rate * 0.5) return fail(err, errlen, ORC_INVALID_INPUT, "f_max exceeds Nyquist");
        if (p->n_mels > 10000) return fail(err, errlen, ORC_INVALID_INPUT, "n_filters is unreasonably large");
    } else if (p->freq_scale != ORC_FREQ_LINEAR) {
        return fail(err, errlen, ORC_INVALID_INPUT, "unknown frequency scale");
    }
    if (p->amp_scale < ORC_AMP_POWER || p->amp_scale > ORC_AMP_DECIBELS)
        return fail(err, errlen, ORC_INVALID_INPUT, "unknown amplitude scale");
    if (p->has_db && !isfinite(p->floor_db))
        return fail(err, errlen, ORC_INVALID_INPUT, "floor_db must be finite");
    return ORC_OK;
}

/* StftPlan::frame_count  src/spectrogram.rs:1230-1250 */
size_t orc_frame_count(size_t n_samples, size_t n_fft, size_t hop, int centre) {
    size_t pad = centre ? n_fft / 2 : 0;
    size_t padded_len = n_samples + 2 * pad;
    if (padded_len < n_fft) return 1;
    size_t remaining = padded_len - n_fft;
    return remaining / hop + 1;
}

size_t orc_n_bins(const orc_params *p) {
    /* FrequencyMapping::output_bins src/spectrogram.rs:1809-1820; r2c_output_size fft_backend.rs:16-18 */
    return p->freq_scale != ORC_FREQ_LINEAR ? (size_t)p->n_mels : (size_t)p->n_fft / 2 + 1;
}

/* modified_bessel_i0  src/spectrogram.rs:2237-2259 (Abramowitz-Stegun polynomial, not exact I0) */
static double bessel_i0_poly(double x) {
    double ax = fabs(x);
    if (ax <= 3.75) {
        double t = x / 3.75;
        double t2 = t * t;
        return 1.0 + t2 * (3.5156229 + t2 * (3.0899424 + t2 * (1.2067492 +
                     t2 * (0.2659732 + t2 * (0.0360768 + t2 * 0.0045813)))));
    } else {
        double t = 3.75 / ax;
        double poly = 0.39894228 + t * (0.01328592 + t * (0.00225319 + t * (-0.00157565 +
                      t * (0.00916281 + t * (-0.02057706 + t * (0.02635537 +
                      t * (-0.01647633 + t * 0.00392377)))))));
        return (exp(ax) / (sqrt(ax) * sqrt(2.0 * M_PI))) * poly;
    }
}

/* make_window  src/spectrogram.rs:2159-2235 (coefficients in f64; caller casts to T, :2232) */
int orc_make_window(int kind, double param, const double *custom, size_t n, double *w) {
    if (n == 0 || !w) return ORC_INVALID_INPUT;
    switch (kind) {
    case ORC_WIN_RECT:
        for (size_t i = 0; i < n; i++) w[i] = 1.0;
        break;
    case ORC_WIN_HANNING: {
        double n1 = (double)(n - 1);
        for (size_t i = 0; i < n; i++) w[i] = fma(0.5, -cos(2.0 * M_PI * (double)i / n1), 0.5);
        break;
    }
    case ORC_WIN_HAMMING: {
        double n1 = (double)(n - 1);
        for (size_t i = 0; i < n; i++) w[i] = fma(0.46, -cos(2.0 * M_PI * (double)i / n1), 0.54);
        break;
    }
    case ORC_WIN_BLACKMAN: {
        double n1 = (double)(n - 1);
        for (size_t i = 0; i < n; i++) {
            double a = 2.0 * M_PI * (double)i / n1;
            w[i] = fma(0.08, cos(2.0 * a), fma(0.5, -cos(a), 0.42));
        }
        break;
    }
    case ORC_WIN_KAISER: {
        if (n == 1) { w[0] = 1.0; break; }
        double denom = bessel_i0_poly(param);
        double n_max = (double)(n - 1) / 2.0;
        for (size_t i = 0; i < n; i++) {
            double x = (double)i - n_max;
            double ratio;
            if (n_max == 0.0) ratio = 0.0;
            else { double nr = x / n_max; ratio = fmax(1.0 - nr * nr, 0.0); }
            double arg = param * sqrt(ratio);
            w[i] = (denom == 0.0) ? 0.0 : bessel_i0_poly(arg) / denom;
        }
        break;
    }
    case ORC_WIN_GAUSSIAN: {
        double center = (double)(n - 1) / 2.0;
        for (size_t i = 0; i < n; i++) {
            double q = ((double)i - center) / param;
            double e = -0.5 * (q * q); /* powi(2) */
            w[i] = exp(e);
        }
        break;
    }
    case ORC_WIN_CUSTOM:
        if (!custom) return ORC_INVALID_INPUT;
        memcpy(w, custom, n * sizeof(double));
        break;
    default:
        return ORC_INVALID_INPUT;
    }
    return ORC_OK;
}

/* hz_to_mel / mel_to_hz  src/spectrogram.rs:2268-2300 (Slaney) */
#define MEL_F_SP (200.0 / 3.0)
#define MEL_MIN_LOG_HZ 1000.0
#define MEL_MIN_LOG_MEL (MEL_MIN_LOG_HZ / MEL_F_SP)
#define MEL_LOGSTEP 0.06875177742094923
double orc_hz_to_mel(double hz) {
    if (hz >= MEL_MIN_LOG_HZ) return MEL_MIN_LOG_MEL + log(hz / MEL_MIN_LOG_HZ) / MEL_LOGSTEP;
    return (hz - 0.0) / MEL_F_SP;
}
double orc_mel_to_hz(double mel) {
    if (mel >= MEL_MIN_LOG_MEL) return MEL_MIN_LOG_HZ * exp(MEL_LOGSTEP * (mel - MEL_MIN_LOG_MEL));
    return fma(MEL_F_SP, mel, 0.0);
}

/* build_mel_filterbank_matrix  src/spectrogram.rs:2302-2432, SparseMatrix::set :69-87 */
long orc_mel_filterbank(double sr, size_t n_fft, size_t n_mels, double f_min, double f_max, int norm,
                        size_t *row_ptr, uint32_t *cols, double *vals, size_t cap) {
    if (sr <= 0.0 || !isfinite(sr)) return -ORC_INVALID_INPUT;
    if (f_min < 0.0 || isinf(f_min)) return -ORC_INVALID_INPUT;
    if (f_max <= f_min) return -ORC_INVALID_INPUT;
    if (f_max > sr * 0.5) return -ORC_INVALID_INPUT;
    size_t out_len = n_fft / 2 + 1;
    double df = sr / (double)n_fft;
    double mel_min = orc_hz_to_mel(f_min), mel_max = orc_hz_to_mel(f_max);
    size_t n_points = n_mels + 2;
    double step = (mel_max - mel_min) / (double)(n_points - 1);
    double *mel_points = (double *)malloc(n_points * sizeof(double));
    double *hz_points = (double *)malloc(n_points * sizeof(double));
    if (!mel_points || !hz_points) { free(mel_points); free(hz_points); return -ORC_INTERNAL; }
    for (size_t i = 0; i < n_points; i++) mel_points[i] = fma((double)i, step, mel_min);
    for (size_t i = 0; i < n_points; i++) hz_points[i] = orc_mel_to_hz(mel_points[i]);

    size_t nnz = 0;
    long rc = 0;
    for (size_t m = 0; m < n_mels; m++) {
        row_ptr[m] = nnz;
        double fl = hz_points[m], fc = hz_points[m + 1], fr = hz_points[m + 2];
        double dl = fc - fl, dr = fr - fc;
        if (dl == 0.0 || dr == 0.0) continue; /* degenerate triangle */
        for (size_t k = 0; k < out_len; k++) {
            double bf = (double)k * df;
            double lower = (bf - fl) / dl;
            double upper = (fr - bf) / dr;
            double wgt = fmin(lower, upper);
            wgt = wgt < 0.0 ? 0.0 : (wgt > 1.0 ? 1.0 : wgt);
            if (wgt > 0.0 && fabs(wgt) > 1e-10) { /* set(): only store |v| > 1e-10 (:83) */
                if (nnz >= cap) { rc = -ORC_DIM_MISMATCH; goto done; }
                cols[nnz] = (uint32_t)k;
                vals[nnz] = wgt;
                nnz++;
            }
        }
    }
    row_ptr[n_mels] = nnz;
    for (size_t m = 0; m < n_mels; m++) {
        size_t a = row_ptr[m], b = row_ptr[m + 1];
        if (norm == ORC_MELNORM_SLANEY) {
            double hz_left = orc_mel_to_hz(mel_points[m]);
            double hz_right = orc_mel_to_hz(mel_points[m + 2]);
            double enorm = 2.0 / (hz_right - hz_left);
            for (size_t i = a; i < b; i++) vals[i] *= enorm;
        } else if (norm == ORC_MELNORM_L1) {
            double s = 0.0;
            for (size_t i = a; i < b; i++) s += vals[i];
            if (s > 0.0) { double nz = 1.0 / s; for (size_t i = a; i < b; i++) vals[i] *= nz; }
        } else if (norm == ORC_MELNORM_L2) {
            double s = 0.0;
            for (size_t i = a; i < b; i++) s += vals[i] * vals[i];
            s = sqrt(s);
            if (s > 0.0) { double nz = 1.0 / s; for (size_t i = a; i < b; i++) vals[i] *= nz; }
        }
    }
    rc = (long)nnz;
done:
    free(mel_points);
    free(hz_points);
    return rc;
}

size_t orc_istft_length(size_t n_frames, size_t n_fft, size_t hop, int centre) {
    size_t pad = centre ? n_fft / 2 : 0;
    size_t out_len = (n_frames - 1) * hop + n_fft;
    size_t unpadded = out_len > 2 * pad ? out_len - 2 * pad : 0;
    return (centre && unpadded > 0) ? unpadded : out_len;
}

static size_t sat_usize(double v) { /* Rust `as usize` */
    if (!(v == v) || v <= 0.0) return 0;
    if (v >= 1.8446744073709552e19) return (size_t)-1;
    return (size_t)v;
}

/* build_loghz_matrix  src/spectrogram.rs:2438-2508 */
long orc_loghz_matrix(double sr, size_t n_fft, size_t n_bins, double f_min, double f_max, size_t *row_ptr, uint32_t *cols,
                      double *vals, size_t cap, double *freqs) {
    if (sr <= 0.0 || !isfinite(sr)) return -ORC_INVALID_INPUT;
    if (f_min <= 0.0 || isinf(f_min)) return -ORC_INVALID_INPUT;
    if (f_max <= f_min) return -ORC_INVALID_INPUT;
    if (f_max > sr * 0.5) return -ORC_INVALID_INPUT;
    size_t out_len = n_fft / 2 + 1;
    double df = sr / (double)n_fft;
    double log_f_min = log(f_min), log_f_max = log(f_max);
    double log_step = (log_f_max - log_f_min) / (double)(n_bins - 1);
    size_t nnz = 0;
    for (size_t b = 0; b < n_bins; b++) {
        double target = exp(fma((double)b, log_step, log_f_min));
        if (freqs) freqs[b] = target;
        row_ptr[b] = nnz;
        double exact = target / df;
        size_t lower = sat_usize(floor(exact));
        size_t upper = sat_usize(ceil(exact));
        if (upper > out_len - 1) upper = out_len - 1;
        if (lower >= out_len) continue;
        double w[2]; size_t c[2]; int cnt = 0;
        if (lower == upper) { c[cnt] = lower; w[cnt++] = 1.0; }
        else {
            double frac = exact - (double)lower;
            c[cnt] = lower; w[cnt++] = 1.0 - frac;
            if (upper < out_len) { c[cnt] = upper; w[cnt++] = frac; }
        }
        for (int i = 0; i < cnt; i++)
            if (c[i] < out_len && fabs(w[i]) > 1e-10) { /* SparseMatrix::set :69-87 */
                if (nnz >= cap) return -ORC_DIM_MISMATCH;
                cols[nnz] = (uint32_t)c[i]; vals[nnz] = w[i]; nnz++;
            }
    }
    row_ptr[n_bins] = nnz;
    return (long)nnz;
}

/* build_chroma_filterbank  src/chroma.rs:262-345 (ChromaParams::new :64-90 for the checks) */
int orc_chroma_filterbank(double sr, size_t n_fft, double tuning, double f_min, double f_max, double *fb) {
    if (sr <= 0.0 || !isfinite(sr)) return ORC_INVALID_INPUT;
    if (!(tuning > 0.0 && isfinite(tuning))) return ORC_INVALID_INPUT;
    if (!(f_min > 0.0 && isfinite(f_min))) return ORC_INVALID_INPUT;
    if (f_max <= f_min) return ORC_INVALID_INPUT;
    size_t nb = n_fft / 2 + 1;
    double df = sr / (double)n_fft;
    memset(fb, 0, 12 * nb * sizeof(double));
    for (size_t k = 0; k < nb; k++) {
        double freq = (double)k * df;
        if (freq < f_min || freq > f_max || freq <= 0.0) continue;
        double midi = 69.0 + 12.0 * log(freq / tuning) / 0.6931471805599453; /* std::f64::consts::LN_2 */
        double pc = fmod(midi, 12.0);
        if (pc < 0.0) pc += 12.0; /* rem_euclid */
        for (size_t c = 0; c < 12; c++) {
            double dist = fabs(pc - (double)c);
            double circ = dist < 12.0 - dist ? dist : 12.0 - dist;
            double q = circ / 1.0;
            fb[c * nb + k] = exp(-0.5 * (q * q));
        }
    }
    for (size_t c = 0; c < 12; c++) {
        double row = 0.0;
        for (size_t k = 0; k < nb; k++) row += fb[c * nb + k];
        if (row > 0.0) for (size_t k = 0; k < nb; k++) fb[c * nb + k] /= row;
    }
    return ORC_OK;
}

static void erb_centres(size_t n, double f_min, double f_max, int spacing, double *cf) {
    if (spacing == 1) { /* apple_tr35_center_freqs erb.rs:221-238 */
        double shift = 9.26449 * 24.7, a = -shift, d = f_max + shift;
        double e = (log(f_min + shift) - log(f_max + shift)) / (double)n;
        for (size_t i = 0; i < n; i++) cf[n - 1 - i] = a + exp(((double)i + 1.0) * e) * d;
    } else { /* erb.rs:276-284 with hz_to_erb :208, erb_to_hz :249 */
        double emin = 24.7 * (4.37 * f_min / 1000.0 + 1.0), emax = 24.7 * (4.37 * f_max / 1000.0 + 1.0);
        double step = (emax - emin) / (double)(n - 1);
        for (size_t i = 0; i < n; i++) cf[i] = (fma((double)i, step, emin) / 24.7 - 1.0) * 1000.0 / 4.37;
    }
}

long orc_erb_matrix(double sr, size_t n_fft, size_t nf, double f_min, double f_max, int spacing, size_t *row_ptr,
                    uint32_t *cols, double *vals, size_t cap, double *centres) {
    if (sr <= 0.0) return -ORC_INVALID_INPUT;
    size_t nb = n_fft / 2 + 1;
    if (cap < nf * nb) return -ORC_DIM_MISMATCH;
    double *cf = centres ? centres : (double *)malloc(nf * sizeof(double));
    erb_centres(nf, f_min, f_max, spacing, cf);
    double df = sr / (double)n_fft;
    for (size_t m = 0; m < nf; m++) {
        double bw = 1.019 * (24.7 * (4.37 * cf[m] / 1000.0 + 1.0)); /* erb.rs:296-297 */
        row_ptr[m] = m * nb;
        for (size_t k = 0; k < nb; k++) {
            double freq = (double)k * df;
            double complex den = 1.0 + I * ((freq - cf[m]) / bw); /* :306-309 */
            double complex d2 = den * den, d4 = d2 * d2;
            cols[m * nb + k] = (uint32_t)k;
            vals[m * nb + k] = 1.0 / (creal(d4) * creal(d4) + cimag(d4) * cimag(d4)); /* norm_sqr :312 */
        }
    }
    row_ptr[nf] = nf * nb;
    if (!centres) free(cf);
    return (long)(nf * nb);
}

/* axes: build_time_axis_seconds :2128-2139; frequencies_hz :1909-1931;
 * mel_band_centres_hz :2510-2530 (ignores MelParams f_min/f_max — S10) */
int orc_axes(const orc_params *p, size_t n_frames, double *freqs, double *times) {
    if (!p) return ORC_INVALID_INPUT;
    if (times) {
        double dt = (double)p->hop / p->sample_rate;
        for (size_t i = 0; i < n_frames; i++) times[i] = (double)i * dt;
    }
    if (freqs) {
        if (p->freq_scale == ORC_FREQ_ERB) { /* centre frequencies :1936-1939 */
            erb_centres(p->n_mels, p->f_min, p->f_max, p->mel_norm, freqs);
        } else if (p->freq_scale == ORC_FREQ_LOGHZ) { /* stored log frequencies :1932-1935 */
            double l0 = log(p->f_min), st = (log(p->f_max) - l0) / (double)(p->n_mels - 1);
            for (size_t i = 0; i < p->n_mels; i++) freqs[i] = exp(fma((double)i, st, l0));
        } else if (p->freq_scale == ORC_FREQ_MEL) {
            double nyq = p->sample_rate * 0.5;
            double f_max = fmin(nyq, p->sample_rate * 0.5);
            double mel_min = orc_hz_to_mel(0.0), mel_max = orc_hz_to_mel(f_max);
            double step = (mel_max - mel_min) / (double)(p->n_mels + 1);
            for (size_t i = 0; i < p->n_mels; i++)
                freqs[i] = orc_mel_to_hz(fma((double)i + 1.0, step, mel_min));
        } else {
            double df = p->sample_rate / (double)p->n_fft;
            for (size_t k = 0; k < (size_t)p->n_fft / 2 + 1; k++) freqs[k] = (double)k * df;
        }
    }
    return ORC_OK;
}

/* gaussian_kernel_2d  src/image_ops.rs:188-220 (f64; caller casts v/sum to T) */
int orc_gaussian_kernel_2d(size_t size, double sigma, double *out) {
    if (size == 0 || size % 2 == 0) return ORC_INVALID_INPUT; /* "kernel size must be odd and > 0" */
    if (sigma <= 0.0) return ORC_INVALID_INPUT;
    double center = (double)(size / 2), variance = sigma * sigma;
    double coeff = 1.0 / (2.0 * M_PI * variance), sum = 0.0;
    for (size_t i = 0; i < size; i++)
        for (size_t j = 0; j < size; j++) {
            double x = (double)i - center, y = (double)j - center;
            double e = -(x * x + y * y) / (2.0 * variance);
            out[i * size + j] = coeff * exp(e);
        }
    for (size_t i = 0; i < size * size; i++) sum += out[i]; /* ndarray .sum(): sequential over the standard layout */
    for (size_t i = 0; i < size * size; i++) out[i] = out[i] / sum;
    return ORC_OK;
}

/* create_lowpass_mask  src/image_ops.rs:236-267 — called with the HALF spectrum's dims (quirk S14) */
void orc_lowpass_mask(size_t nrows, size_t ncols, double cutoff, double *mask) {
    double mr = (double)(nrows / 2), mc = (double)(ncols / 2);
    double q = fmin(mr, mc) * cutoff;
    double max_radius = q * q; /* powi(2) */
    for (size_t i = 0; i < nrows; i++)
        for (size_t j = 0; j < ncols; j++) {
            double fr = i <= nrows / 2 ? (double)i : fabs((double)i - (double)nrows);
            double fc = j <= ncols / 2 ? (double)j : fabs((double)j - (double)ncols);
            double d = fma(fc, fc, fr * fr);
            mask[i * ncols + j] = d <= max_radius ? 1.0 : 0.0;
        }
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- long lengths that are not powers of two: chirp-z (Bluestein), f64 ------------------------------------------------
 * The reference plans every length through rustfft (src/fft_backend.rs:372-389), which itself uses Bluestein / Rader for
 * lengths with large prime factors; the oracle's direct O(n^2) sum is the definition (src/fft_backend.rs:16-18,128) but takes
 * minutes at n = 100 003.  From ORC_CZT_MIN points on, non-power-of-two lengths use the published chirp-z identity
 *   X[k] = conj(c_k) sum_j (x_j conj(c_j)) c_{k-j},  c_j = e^{i pi j^2 / n}  (forward; the inverse conjugates in and out),
 * as a circular convolution of length M = 2^ceil(log2(2n-1)) on a plain f64 radix-2 FFT, angles reduced in integers
 * (j^2 mod 2n).  Everything in f64 whatever the caller's type, like the direct sum's accumulation.  Pinned against numpy.fft
 * in tests/test_oracle_golden.py. */
#define ORC_CZT_MIN 2048
typedef struct {
    size_t n, M;
    double *cr, *ci;   /* c_j, j < n */
    double *br, *bi;   /* FFT_M of the wrapped chirp */
    double *wr, *wi;   /* e^{-2 pi i j / M}, j < M/2 */
    double *ar, *ai;   /* work, M */
} orc_czt;

static void orc_czt_free(orc_czt *z) {
    free(z->cr); free(z->ci); free(z->br); free(z->bi); free(z->wr); free(z->wi); free(z->ar); free(z->ai);
    memset(z, 0, sizeof(*z));
}

/* in-place radix-2 DIT on separate re / im arrays of length M (power of two), forward (e^{-}) */
static void orc_czt_fft(const orc_czt *z, double *re, double *im) {
    size_t M = z->M;
    for (size_t i = 1, j = 0; i < M; i++) {
        size_t bit = M >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
    for (size_t h = 1; h < M; h <<= 1) {
        size_t step = M / (2 * h);
        for (size_t b = 0; b < M; b += 2 * h)
            for (size_t j = 0; j < h; j++) {
                double wr = z->wr[j * step], wi = z->wi[j * step];
                double xr = re[b + h + j], xi = im[b + h + j];
                double tr = xr * wr - xi * wi, ti = xr * wi + xi * wr;
                double ur = re[b + j], ui = im[b + j];
                re[b + j] = ur + tr; im[b + j] = ui + ti;
                re[b + h + j] = ur - tr; im[b + h + j] = ui - ti;
            }
    }
}

static int orc_czt_init(orc_czt *z, size_t n) {
    memset(z, 0, sizeof(*z));
    size_t M = 1;
    while (M < 2 * n - 1) M <<= 1;
    z->n = n; z->M = M;
    z->cr = (double *)malloc(n * sizeof(double)); z->ci = (double *)malloc(n * sizeof(double));
    z->br = (double *)calloc(M, sizeof(double)); z->bi = (double *)calloc(M, sizeof(double));
    z->wr = (double *)malloc((M / 2 + 1) * sizeof(double)); z->wi = (double *)malloc((M / 2 + 1) * sizeof(double));
    z->ar = (double *)malloc(M * sizeof(double)); z->ai = (double *)malloc(M * sizeof(double));
    if (!z->cr || !z->ci || !z->br || !z->bi || !z->wr || !z->wi || !z->ar || !z->ai) { orc_czt_free(z); return ORC_INTERNAL; }
    for (size_t j = 0; j < M / 2; j++) {
        double a = -2.0 * M_PI * (double)j / (double)M;
        z->wr[j] = cos(a); z->wi[j] = sin(a);
    }
    for (size_t j = 0; j < n; j++) {
        unsigned long long q = ((unsigned long long)j * (unsigned long long)j) % (2ull * n); /* j < 2^31: no overflow */
        double a = M_PI * (double)q / (double)n;
        z->cr[j] = cos(a); z->ci[j] = sin(a);
    }
    z->br[0] = z->cr[0]; z->bi[0] = z->ci[0];
    for (size_t j = 1; j < n; j++) {
        z->br[j] = z->cr[j]; z->bi[j] = z->ci[j];
        z->br[M - j] = z->cr[j]; z->bi[M - j] = z->ci[j];
    }
    orc_czt_fft(z, z->br, z->bi);
    return ORC_OK;
}

/* (re, im)[0..n) -> its DFT (forward e^{-}, or inverse e^{+}, unnormalised) in place */
static void orc_czt_run(orc_czt *z, double *re, double *im, int inverse) {
    size_t n = z->n, M = z->M;
    double sg = inverse ? -1.0 : 1.0;  /* inverse: conj in, conj out */
    for (size_t j = 0; j < n; j++) {
        double xr = re[j], xi = sg * im[j];
        z->ar[j] = xr * z->cr[j] + xi * z->ci[j];   /* x conj(c) */
        z->ai[j] = xi * z->cr[j] - xr * z->ci[j];
    }
    for (size_t j = n; j < M; j++) { z->ar[j] = 0.0; z->ai[j] = 0.0; }
    orc_czt_fft(z, z->ar, z->ai);
    for (size_t j = 0; j < M; j++) {  /* product, conjugated for the inverse transform by the conj-FFT-conj identity */
        double pr = z->ar[j] * z->br[j] - z->ai[j] * z->bi[j];
        double pi = z->ar[j] * z->bi[j] + z->ai[j] * z->br[j];
        z->ar[j] = pr; z->ai[j] = -pi;
    }
    orc_czt_fft(z, z->ar, z->ai);
    double inv = 1.0 / (double)M;
    for (size_t k = 0; k < n; k++) {
        double yr = z->ar[k] * inv, yi = -z->ai[k] * inv;
        double Xr = yr * z->cr[k] + yi * z->ci[k];    /* y conj(c) */
        double Xi = yi * z->cr[k] - yr * z->ci[k];
        re[k] = Xr; im[k] = sg * Xi;
    }
}

/* ---- typed part (T = f32 / f64), see oracle_typed.inc --------------------- */
#define REAL float
#define SUF(x) x##_f32
#define R_SQRT sqrtf
#define R_LOG10 log10f
#include "oracle_typed.inc"
#undef REAL
#undef SUF
#undef R_SQRT
#undef R_LOG10

#define REAL double
#define SUF(x) x##_f64
#define R_SQRT sqrt
#define R_LOG10 log10
#include "oracle_typed.inc"
#undef REAL
#undef SUF
#undef R_SQRT
#undef R_LOG10
