/*
 * spectro_oracle.h — CPU ORACLE for the STFT / Mel-spectrogram hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * (jmg049/Spectrograms, crate `spectrograms` v2.1.0) CPU algorithm for the hot
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
 * leg may load it — as the checker / the timed CPU baseline, never as the thing
 * shipped.  The product library (spectrograms_amd/csrc → libspectro_hip.so) does
 * not link, include or call anything in this directory.
 *
 * Parity pinning: the 1-D FFT arithmetic of the reference lives in the external
 * crates realfft 3.5.0 / rustfft 6.4.1 (Cargo.lock:920-922,1002-1004), which are
 * not vendored under /root/reference and cannot be built here (no cargo/rustc).
 * The oracle therefore restates the published DFT definition
 * X[k] = sum_n x[n] e^{-2 pi i k n / N} (src/fft_backend.rs:16-18,128) and is
 * pinned (tests/test_oracle_golden.py) against
 *   (i)  golden vectors generated in the build container by importing the
 *        reference's own python/examples/numpy_impls.py (stft / hann_window /
 *        power_spectrogram / magnitude_spectrogram — identical semantics to the
 *        Rust path) — tests/golden/make_golden.py, and
 *   (ii) every numeric known-answer test the reference's test-suite holds for
 *        this path.
 * Bit-level parity with RustFFT output itself is UNPINNED (no golden values
 * exist in the reference); tolerances are stated in the tests.
 */
#ifndef SPECTRO_ORACLE_H
#define SPECTRO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_OK = 0, ORC_INVALID_INPUT = 1, ORC_DIM_MISMATCH = 2, ORC_BACKEND = 3, ORC_INTERNAL = 4 };

enum { ORC_WIN_RECT = 0, ORC_WIN_HANNING = 1, ORC_WIN_HAMMING = 2, ORC_WIN_BLACKMAN = 3,
       ORC_WIN_KAISER = 4, ORC_WIN_GAUSSIAN = 5, ORC_WIN_CUSTOM = 6 };
enum { ORC_FREQ_LINEAR = 0, ORC_FREQ_MEL = 1, ORC_FREQ_LOGHZ = 2 /* n_mels = n_bins */,
       ORC_FREQ_ERB = 3 /* n_mels = n_filters, mel_norm = spacing (0 linear, 1 Apple TR35) */ };
enum { ORC_MELNORM_NONE = 0, ORC_MELNORM_SLANEY = 1, ORC_MELNORM_L1 = 2, ORC_MELNORM_L2 = 3 };
enum { ORC_AMP_POWER = 0, ORC_AMP_MAGNITUDE = 1, ORC_AMP_DECIBELS = 2 };

typedef struct {
    uint32_t n_fft, hop;
    int32_t centre;
    int32_t window_kind;
    double window_param;          /* kaiser beta / gaussian std (samples) */
    const double *custom_window;  /* n_fft coefficients when window_kind == CUSTOM */
    double sample_rate;
    int32_t freq_scale;
    uint32_t n_mels;
    double f_min, f_max;
    int32_t mel_norm;
    int32_t amp_scale;
    int32_t has_db;               /* LogParams supplied? (S6: dB only applied if so) */
    double floor_db;
} orc_params;

/* spectrogram.rs:3479-3506, 4129-4140, 3793-3813, 4071-4077, 944-959 */
int orc_validate(const orc_params *p, char *err, size_t errlen);
/* istft output length (spectrogram.rs:4888-4893, 4933-4941): (n_frames-1)*hop + n_fft, minus 2*pad when centred and > 0 */
size_t orc_istft_length(size_t n_frames, size_t n_fft, size_t hop, int centre);
/* spectrogram.rs:1230-1250 */
size_t orc_frame_count(size_t n_samples, size_t n_fft, size_t hop, int centre);
size_t orc_n_bins(const orc_params *p);
/* spectrogram.rs:2159-2259 */
int orc_make_window(int kind, double param, const double *custom, size_t n, double *out);
/* spectrogram.rs:2268-2432; CSR output. returns nnz (>=0) or -status. */
long orc_mel_filterbank(double sample_rate, size_t n_fft, size_t n_mels, double f_min, double f_max,
                        int norm, size_t *row_ptr /*n_mels+1*/, uint32_t *cols, double *vals,
                        size_t cap);
/* spectrogram.rs:2438-2508; CSR + centre frequencies. returns nnz (>=0) or -status. */
long orc_loghz_matrix(double sample_rate, size_t n_fft, size_t n_bins, double f_min, double f_max, size_t *row_ptr,
                      uint32_t *cols, double *vals, size_t cap, double *freqs);
/* src/erb.rs:266-335 (ErbFilterbank::generate): dense n_filters x (n_fft/2+1) |H|^2 rows as CSR + centre freqs */
long orc_erb_matrix(double sample_rate, size_t n_fft, size_t n_filters, double f_min, double f_max, int spacing,
                    size_t *row_ptr, uint32_t *cols, double *vals, size_t cap, double *centres);
/* build_chroma_filterbank src/chroma.rs:262-345: dense fb[12][n_fft/2+1], rows normalised to unit sum */
int orc_chroma_filterbank(double sample_rate, size_t n_fft, double tuning, double f_min, double f_max, double *fb);
int orc_chromagram_f32(const orc_params *stft_p, double tuning, double f_min, double f_max, int norm, const float *x, size_t n, float *out);
int orc_chromagram_f64(const orc_params *stft_p, double tuning, double f_min, double f_max, int norm, const double *x, size_t n, double *out);
double orc_hz_to_mel(double hz);
double orc_mel_to_hz(double mel);
/* spectrogram.rs:2128-2139,1909-1931,2510-2530 */
int orc_axes(const orc_params *p, size_t n_frames, double *freqs, double *times);

/* forward real DFT, n real -> n/2+1 complex (interleaved re,im); any n >= 1 */
int orc_rfft_f32(const float *in, size_t n, float *out);
int orc_rfft_f64(const double *in, size_t n, double *out);

/* spectrogram.rs:1424-1458: complex STFT, out[(bin*n_frames + frame)*2 + {0,1}] */
int orc_stft_f32(const orc_params *p, const float *x, size_t n, float *out);
int orc_stft_f64(const orc_params *p, const double *x, size_t n, double *out);
/* spectrogram.rs:240-294: out[bin*n_frames + frame] */
int orc_spectrogram_f32(const orc_params *p, const float *x, size_t n, float *out);
int orc_spectrogram_f64(const orc_params *p, const double *x, size_t n, double *out);

/* the reference's batch idiom (src/lib.rs:228-236): one plan, loop over signals;
 * nthreads > 1 = one plan per thread over utterances (OpenMP). */
int orc_spectrogram_batch_f32(const orc_params *p, const float *x, size_t batch, size_t n,
                              size_t stride, float *out, int nthreads);
int orc_spectrogram_batch_f64(const orc_params *p, const double *x, size_t batch, size_t n,
                              size_t stride, double *out, int nthreads);
int orc_stft_batch_f32(const orc_params *p, const float *x, size_t batch, size_t n, size_t stride,
                       float *out, int nthreads);
int orc_stft_batch_f64(const orc_params *p, const double *x, size_t batch, size_t n, size_t stride,
                       double *out, int nthreads);
/* src/mfcc.rs:224-316 mfcc_from_log_mel over the plan's Mel-dB spectrogram (the plan must be Mel + dB);
 * out[(coef * n_frames) + frame], rows = n_mfcc - (include_c0 ? 0 : (n_mfcc > 1)). */
int orc_mfcc_f32(const orc_params *p, uint32_t n_mfcc, int include_c0, uint32_t lifter, const float *x, size_t n, float *out);
int orc_mfcc_f64(const orc_params *p, uint32_t n_mfcc, int include_c0, uint32_t lifter, const double *x, size_t n, double *out);
/* ---- 2-D path: src/fft_backend.rs:653-691 (forward), :744-818 (inverse); src/image_ops.rs:80-152,188-267,301-432 */
int orc_fft2d_f32(const float *img, size_t nrows, size_t ncols, float *spec /*[nrows][ncols/2+1][2]*/);
int orc_fft2d_f64(const double *img, size_t nrows, size_t ncols, double *spec);
/* irfft spectrogram.rs:4789-4811 (C2rPlan::process fft_backend.rs:526-565); istft spectrogram.rs:4860-4946 */
int orc_irfft_f32(const float *spec, size_t n_bins, size_t n_fft, float *out);
int orc_irfft_f64(const double *spec, size_t n_bins, size_t n_fft, double *out);
int orc_istft_f32(const float *stft, size_t n_bins, size_t n_frames, size_t n_fft, size_t hop, int window_kind,
                  double window_param, const double *custom, int centre, float *out);
int orc_istft_f64(const double *stft, size_t n_bins, size_t n_frames, size_t n_fft, size_t hop, int window_kind,
                  double window_param, const double *custom, int centre, double *out);
int orc_ifft2d_f32(const float *spec, size_t nrows, size_t ncols, float *img);
int orc_ifft2d_f64(const double *spec, size_t nrows, size_t ncols, double *img);
int orc_convolve_fft_f32(const float *img, size_t nrows, size_t ncols, const float *ker, size_t kr, size_t kc, float *out);
int orc_convolve_fft_f64(const double *img, size_t nrows, size_t ncols, const double *ker, size_t kr, size_t kc, double *out);
/* kind 0 lowpass(cut_lo), 1 highpass(cut_lo), 2 bandpass(cut_lo, cut_hi) */
int orc_filter2d_f32(const float *img, size_t nrows, size_t ncols, int kind, double cut_lo, double cut_hi, float *out);
int orc_filter2d_f64(const double *img, size_t nrows, size_t ncols, int kind, double cut_lo, double cut_hi, double *out);
int orc_gaussian_kernel_2d(size_t size, double sigma, double *out /*size*size, normalised f64*/);
void orc_lowpass_mask(size_t nrows, size_t ncols, double cutoff, double *mask);
int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
