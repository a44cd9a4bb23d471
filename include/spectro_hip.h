/*
 * spectro_hip.h — C ABI of libspectro_hip.so, the MI355X (gfx950) engine for the
 * batched STFT / Mel-spectrogram hot path of the `spectrograms` crate
 * (jmg049/Spectrograms v2.1.0).  All reference citations are file:line under the
 * reference repository root.
 *
 * This header is the drop-in boundary: a `fft_backend::hip_backend` variant in
 * the reference (the analogue of its FFTW variant, src/fft_backend.rs:1084-1871)
 * binds exactly these symbols — see INTEGRATION.md for the Rust `extern "C"`
 * block and the ctypes binding used by the Python host mirror.
 *
 * Conventions
 *  - Plain C types only; no C++/torch types cross the boundary.
 *  - A plan is the analogue of the reference's `&mut self` plans
 *    (src/fft_backend.rs:21-24: "own scratch, reusable, no heap allocation in
 *    process"): it owns every device table (window, twiddles, filterbank) and all
 *    scratch; it is NOT thread-safe (one caller at a time, like `&mut self`;
 *    Python plan objects are `unsendable`, src/python/planner.rs:674).  Distinct
 *    plans are independent.
 *  - No function aborts or throws across the ABI.  Status codes mirror
 *    `SpectrogramError` (src/error.rs:13-28); the message text is available
 *    from sgx_last_error() / sgx_last_create_error().
 *  - There is no CPU fallback: without a usable HIP device every compute entry
 *    point returns SGX_BACKEND.
 *  - The library keeps no mutable global state and reads no environment
 *    variables; an entry point leaves the caller's current HIP device as it
 *    found it.
 *  - Allocation: everything the per-frame entry points (sgx_r2c / sgx_c2r, the
 *    analogues of R2cPlan / C2rPlan::process) touch is allocated by
 *    sgx_plan_create.  The batched entry points need scratch that depends on the
 *    call's size (host staging, the MFCC Mel tensor, the generic inverse path's
 *    frames): sgx_reserve sizes it ahead; a call that fits what was reserved
 *    allocates nothing, a larger one grows the scratch once.
 */
#ifndef SPECTRO_HIP_H
#define SPECTRO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGX_ABI_VERSION 7

typedef struct sgx_plan sgx_plan; /* opaque */

/* src/error.rs:13-28  InvalidInput / DimensionMismatch / FftBackendError{backend:"hip"} / InternalError */
typedef enum {
    SGX_OK = 0,
    SGX_INVALID_INPUT = 1,
    SGX_DIM_MISMATCH = 2,
    SGX_BACKEND = 3,
    SGX_INTERNAL = 4
} sgx_status;

/* src/window.rs:19-50 WindowType */
enum { SGX_WIN_RECTANGULAR = 0, SGX_WIN_HANNING = 1, SGX_WIN_HAMMING = 2, SGX_WIN_BLACKMAN = 3,
       SGX_WIN_KAISER = 4, SGX_WIN_GAUSSIAN = 5, SGX_WIN_CUSTOM = 6 };
/* src/spectrogram.rs:3374-3442 frequency-scale markers (LinearHz, Mel, LogHz, Erb).  LogHz (LogHzParams :3935-3990,
 * matrix build_loghz_matrix :2438-2508) reuses n_mels / f_min / f_max as n_bins / f_min / f_max.  Erb (ErbParams
 * src/erb.rs:27-92, frequency-domain gammatone bank ErbFilterbank::generate :266-335, applied as a DENSE
 * n_filters x (n_fft/2+1) product with the power spectrum :374-401) reuses them as n_filters / f_min / f_max. */
enum { SGX_FREQ_LINEAR = 0, SGX_FREQ_MEL = 1, SGX_FREQ_LOGHZ = 2, SGX_FREQ_ERB = 3, SGX_FREQ_CHROMA = 4 };
/* Chroma (src/chroma.rs): chromagram() :470-505 = linear MAGNITUDE spectrogram -> dense 12 x (n_fft/2+1) pitch-class bank
 * (build_chroma_filterbank :262-345, bins inside [f_min, f_max], Gaussian over the circular semitone distance, rows
 * normalised to unit sum) applied with sequential accumulation (:378-392) -> per-frame normalisation over the 12 rows
 * (apply_chroma_normalization :403-445).  Requires amp_scale = SGX_AMP_MAGNITUDE and no LogParams; f_min / f_max reused;
 * output has 12 rows.  ChromaNorm :24-31: */
enum { SGX_CHROMA_NORM_NONE = 0, SGX_CHROMA_NORM_L1 = 1, SGX_CHROMA_NORM_L2 = 2, SGX_CHROMA_NORM_MAX = 3 };
/* ErbSpacing src/erb.rs:14-25 */
enum { SGX_ERB_LINEAR = 0, SGX_ERB_APPLE_TR35 = 1 };
/* MelNorm src/spectrogram.rs:2385-2429 */
enum { SGX_MELNORM_NONE = 0, SGX_MELNORM_SLANEY = 1, SGX_MELNORM_L1 = 2, SGX_MELNORM_L2 = 3 };
/* AmpScaleSpec impls src/spectrogram.rs:1986-2037; COMPLEX = StftPlan::compute (:1424-1458) */
enum { SGX_AMP_POWER = 0, SGX_AMP_MAGNITUDE = 1, SGX_AMP_DECIBELS = 2, SGX_AMP_COMPLEX = 3 };
/* Sample impls src/sample.rs:23-86 */
enum { SGX_F32 = 0, SGX_F64 = 1 };
enum { SGX_MEM_HOST = 0, SGX_MEM_DEVICE = 1 };

/* StftParams (src/spectrogram.rs:3452-3506) + SpectrogramParams (:4108-4140) + MelParams (:3744-3813)
 * + LogParams (:4052-4077) + the `T: Sample` choice, flattened. */
typedef struct {
    uint32_t n_fft;
    uint32_t hop_size;
    int32_t centre;               /* zero padding of n_fft/2 on both sides (S1) */
    int32_t window_kind;          /* SGX_WIN_* */
    double window_param;          /* Kaiser beta / Gaussian std in samples */
    const double *custom_window;  /* n_fft coefficients (copied) or NULL */
    uint32_t custom_window_len;   /* must equal n_fft for SGX_WIN_CUSTOM (:3490-3498) */
    double sample_rate_hz;
    int32_t freq_scale;           /* SGX_FREQ_* */
    uint32_t n_mels;
    double f_min, f_max;
    int32_t mel_norm;             /* SGX_MELNORM_* */
    int32_t amp_scale;            /* SGX_AMP_* */
    int32_t has_log_params;       /* Option<&LogParams>: dB applied only when set (S6) */
    double floor_db;
    int32_t dtype;                /* SGX_F32 / SGX_F64 */
    int32_t device;               /* HIP device ordinal; -1 = current device; -2 = host-only plan (no compute) */
    /* MfccParams (src/mfcc.rs:20-90): n_mfcc > 0 turns the plan into mfcc_from_log_mel (:224-273) applied to its
     * Mel-dB output — unnormalised DCT-II with an FMA chain in T (:278-292), sinusoidal lifter (:297-316), optional
     * removal of C0.  Requires freq_scale = MEL and amp_scale = DECIBELS; n_mfcc <= n_mels. */
    uint32_t n_mfcc;
    int32_t mfcc_include_c0;
    uint32_t mfcc_lifter;
    int32_t erb_spacing;          /* SGX_ERB_* (only read when freq_scale = SGX_FREQ_ERB) */
    double chroma_tuning;         /* A4 in Hz (ChromaParams::tuning); only read when freq_scale = SGX_FREQ_CHROMA */
    int32_t chroma_norm;          /* SGX_CHROMA_NORM_* */
} sgx_params;

/* Replaces StftPlan::new (:1204-1228), SpectrogramPlanner::{linear_plan :893-917, mel_plan :944-977}:
 * validates exactly like the reference constructors, builds window / twiddles / CSR filterbank on the
 * host in f64, casts to T and uploads.  On failure *out is NULL and sgx_last_create_error() has the text.
 * Every n_fft up to 2^20 (powers of two: 2^21) has a plan in both types, like the reference's planner (src/fft_backend.rs:372-389);
 * longer frames are SGX_BACKEND ("n_fft too large"). */
sgx_status sgx_plan_create(const sgx_params *params, sgx_plan **out);
void sgx_plan_destroy(sgx_plan *plan);

/* StftPlan::frame_count (:1230-1250) + SpectrogramPlan::output_shape (:512-519) */
sgx_status sgx_output_shape(const sgx_plan *plan, size_t n_samples, size_t *n_bins, size_t *n_frames);

/* The batched fast path replacing the per-signal loop `for s in signals { plan.compute(s) }`
 * (src/lib.rs:228-236) over SpectrogramPlan::compute (:240-294) / StftPlan::compute (:1424-1458).
 *   samples : batch rows of n_samples elements of T, row r at samples + r*sample_stride elements
 *   out     : [batch][n_bins][n_frames] row-major T (frames contiguous, S9); for SGX_AMP_COMPLEX
 *             interleaved (re,im) pairs, out_elems counts T elements (2 per complex value)
 *   out_elems != batch*n_bins*n_frames*(1|2)  ->  SGX_DIM_MISMATCH (compute_into, :423-434)
 *   mem_kind: SGX_MEM_HOST (plan-owned staging + copies, synchronous) or SGX_MEM_DEVICE
 *             (asynchronous on `hip_stream`, a hipStream_t; NULL = the null stream) */
sgx_status sgx_execute(sgx_plan *plan, const void *samples, size_t batch, size_t n_samples,
                       size_t sample_stride, void *out, size_t out_elems, int32_t mem_kind,
                       void *hip_stream);

/* Same as sgx_execute with device pointers, but launches `iters` times back-to-back between two
 * hipEvents recorded on `hip_stream` and returns the mean device time per launch in milliseconds
 * (used by bench.py for the roofline line).  Synchronises the stream. */
sgx_status sgx_execute_timed(sgx_plan *plan, const void *samples, size_t batch, size_t n_samples,
                             size_t sample_stride, void *out, size_t out_elems, void *hip_stream,
                             int32_t iters, float *ms_per_launch);

/* build_time_axis_seconds (:2128-2139), frequencies_hz (:1909-1931), mel_band_centres_hz (:2510-2530) */
sgx_status sgx_axes(const sgx_plan *plan, size_t n_frames, double *freqs /*n_bins*/, double *times /*n_frames*/);

/* Conforming per-call R2cPlan::process (src/fft_backend.rs:25-44, 423-431): host pointers, one frame,
 * in_len must be n_fft and out_len n_fft/2+1 complex values else SGX_DIM_MISMATCH (:264-282). */
sgx_status sgx_r2c(sgx_plan *plan, const void *in, size_t in_len, void *out, size_t out_len);

/* ---- inverse 1-D path ------------------------------------------------------------------------------------------
 * Conforming per-call C2rPlan::process (src/fft_backend.rs:526-565) = irfft (src/spectrogram.rs:4789-4811): host pointers,
 * in_len must be n_fft/2+1 complex values and out_len n_fft else SGX_DIM_MISMATCH; output scaled by 1/n_fft in T.  A
 * non-zero imaginary part in the DC (or, even n_fft, Nyquist) bin is ignored in the arithmetic and reported as
 * SGX_BACKEND after the output has been written — realfft's FftError::InputValues, mapped at :555-557. */
sgx_status sgx_c2r(sgx_plan *plan, const void *in, size_t in_len, void *out, size_t out_len);
/* istft (src/spectrogram.rs:4860-4946), batched over `batch` STFT matrices laid out [batch][n_bins][n_frames] complex T
 * (what sgx_execute writes for SGX_AMP_COMPLEX): per frame C2R, * window, overlap-add in ascending frame order,
 * / sum(w*w) where that exceeds T(1e-10), centre trim.  Uses the plan's n_fft / hop_size / window / centre.
 *   n_bins != n_fft/2+1 or out_elems != batch * sgx_istft_length  ->  SGX_DIM_MISMATCH
 *   mem_kind as sgx_execute; the DC/Nyquist check above is only reported for SGX_MEM_HOST (it needs a synchronisation). */
sgx_status sgx_istft_length(const sgx_plan *plan, size_t n_frames, size_t *n_samples);
sgx_status sgx_istft(sgx_plan *plan, const void *stft, size_t batch, size_t n_bins, size_t n_frames, void *out,
                     size_t out_elems, int32_t mem_kind, void *hip_stream);

/* make_window (:2159-2235): the plan's window coefficients as built (f64, before the cast to T). */
sgx_status sgx_window(const sgx_plan *plan, double *out /*n_fft*/);
/* build_mel_filterbank_matrix (:2302-2432) as CSR; pass NULL arrays to query nnz only. */
sgx_status sgx_mel_weights(const sgx_plan *plan, size_t *nnz, uint32_t *row_ptr /*n_mels+1*/,
                           uint32_t *cols, double *vals);

/* Pre-sizes the plan-owned scratch of the batched entry points for calls of up to `batch` signals of `n_samples` samples:
 * the MFCC plan's Mel-dB tensor always; with `inverse` the frame scratch of sgx_istft (generic path) for spectra of that many
 * frames; with `host_staging` the device staging of the SGX_MEM_HOST paths.  After it, such calls do not allocate
 * ("plans own scratch, no allocation in process", src/fft_backend.rs:21-24). */
sgx_status sgx_reserve(sgx_plan *plan, size_t batch, size_t n_samples, int32_t host_staging, int32_t inverse);

/* The HIP device ordinal the plan is bound to (resolved at creation when the params said -1); -2 for a host-only plan. */
int32_t sgx_plan_device(const sgx_plan *plan);

/* DimensionMismatch { expected, got } (src/error.rs:19-21, raised by validate_fft_io src/fft_backend.rs:264-282 and
 * compute_into src/spectrogram.rs:423-434): the two numbers of the plan's most recent SGX_DIM_MISMATCH. */
sgx_status sgx_last_dim_mismatch(const sgx_plan *plan, size_t *expected, size_t *got);

/* ---- multi-GPU (BASELINE config 4): utterances shard across ranks in contiguous blocks (remainder to the low
 * ranks), one process or thread per GPU, no data-path collective unless the caller asks for the gathered output. */
sgx_status sgx_shard_range(size_t batch, int32_t world_size, int32_t rank, size_t *start, size_t *count);

/* RCCL communicator, resolved at run time (the library does not link RCCL; a host that already carries one gets that copy).
 * sgx_comm_unique_id: rank 0 makes the 128-byte id and hands it to the other ranks by its own means (ncclGetUniqueId);
 * sgx_comm_create: ncclCommInitRank on `device` (-1 = current); collective over all ranks.
 * sgx_comm_adopt: wraps an ncclComm_t the host created itself (not destroyed by sgx_comm_destroy). */
#define SGX_COMM_ID_BYTES 128
typedef struct sgx_comm sgx_comm; /* opaque; one per rank */
sgx_status sgx_comm_unique_id(void *id128);
sgx_status sgx_comm_create(const void *id128, int32_t world_size, int32_t rank, int32_t device, sgx_comm **out);
sgx_status sgx_comm_adopt(void *nccl_comm, int32_t world_size, int32_t rank, int32_t device, sgx_comm **out);
void sgx_comm_destroy(sgx_comm *comm);
const char *sgx_comm_last_error(const sgx_comm *comm); /* NULL: the text of a failed create / adopt / unique_id */

/* All-gather of per-rank shards of `global_batch` items of `elems_per_item` elements (dtype SGX_F32 / SGX_F64; a complex value
 * counts as two): rank r contributes its sgx_shard_range block from `send`, every rank receives all blocks in rank order in
 * `recv` (device pointers; `send` may be the rank's own slice of `recv`).  Asynchronous on `hip_stream`.  Equal shards are one
 * ncclAllGather, ragged ones a group of ncclBroadcast. */
sgx_status sgx_gather(sgx_comm *comm, const void *send, void *recv, size_t global_batch, size_t elems_per_item, int32_t dtype,
                      void *hip_stream);

/* This rank's part of a `global_batch`-signal job: runs the plan on its shard (`shard_samples`: device pointer to the rank's
 * own sgx_shard_range block, rows `sample_stride` apart) into `shard_out` — or, if that is NULL, straight into its slice of
 * `gathered_out` — and, when `gathered_out` is not NULL, gathers all shards into it ([global_batch][n_bins][n_frames] on every
 * rank).  Asynchronous on `hip_stream`.  The plan and the communicator must be on the same device. */
sgx_status sgx_shard_execute(sgx_plan *plan, sgx_comm *comm, const void *shard_samples, size_t global_batch, size_t n_samples,
                             size_t sample_stride, void *shard_out, void *gathered_out, void *hip_stream);

/* The same, with the gather overlapped with the compute inside the call ("chunked and overlapped with compute"):
 * the shard is cut into `chunks` (1..64) runs of signals; while run k + 1 computes on `hip_stream`, run k's pieces of every rank
 * travel on a second stream owned by the communicator (a group of ncclBroadcast per run, each into the piece's own place in
 * `gathered_out`), ordered by events only.  `hip_stream` waits for the last exchange before the call's work counts as complete,
 * so the call is asynchronous on `hip_stream` like sgx_shard_execute.  chunks == 1 or gathered_out == NULL: sgx_shard_execute.
 * The result equals sgx_shard_execute's bit for bit.
 * EXPERIMENTAL: exercised on a stand-in RCCL at 2-3 ranks (tests/c_abi/shard_ranks.c) and on the real RCCL at ONE rank only; no run
 * on two or more real GPUs yet.  On any failure the two streams are re-joined before the call returns (the caller's stream waits
 * for whatever was queued on the exchange stream), but collectives other ranks have already issued for later runs stay pending:
 * after a non-OK status destroy the communicator (sgx_comm_destroy) on every rank. */
sgx_status sgx_shard_execute_chunked(sgx_plan *plan, sgx_comm *comm, const void *shard_samples, size_t global_batch,
                                     size_t n_samples, size_t sample_stride, void *shard_out, void *gathered_out, int32_t chunks,
                                     void *hip_stream);

/* ---- 2-D FFT path (BASELINE config 5): R2cPlan2d / C2rPlan2d (src/fft_backend.rs:169-246, 614-819), fft2d / ifft2d
 * (src/fft2d.rs:77-185), convolve_fft and the radial filters (src/image_ops.rs:80-432).  Batched: `batch` images of
 * nrows x ncols per call.  Real images are [batch][nrows][ncols] T; half spectra [batch][nrows][ncols/2+1] complex T
 * (interleaved).  The plan owns every intermediate buffer. */
typedef struct sgx_fft2d sgx_fft2d; /* opaque; same single-caller rule as sgx_plan */
sgx_status sgx_fft2d_create(size_t nrows, size_t ncols, int32_t dtype, int32_t device, sgx_fft2d **out);
void sgx_fft2d_destroy(sgx_fft2d *plan);
/* fft2d: unnormalised forward (rows R2C, then columns C2C) */
sgx_status sgx_fft2d_forward(sgx_fft2d *plan, const void *images, size_t batch, void *spectrum, int32_t mem_kind,
                             void *hip_stream);
/* ifft2d: columns inverse C2C, DC/Nyquist columns forced real, rows C2R, scale 1/(nrows*ncols) */
sgx_status sgx_fft2d_inverse(sgx_fft2d *plan, const void *spectrum, size_t batch, void *images, int32_t mem_kind,
                             void *hip_stream);
/* convolve_fft(image, kernel): kernel (krows x kcols, host pointer, T) is wrapped so its centre sits at (0,0)
 * (pad_kernel_for_fft, image_ops.rs:123-152), transformed once and multiplied into every image's spectrum.  The plan keeps
 * that spectrum on the device: a later call with the same kernel bytes and shape on the same stream reuses it (the kernel is
 * read from `kernel_host` during the call, never afterwards); sgx_fft2d_filter keeps its mask per (kind, cut-offs, stream).
 * f32 images with 1024 rows (the fused column stage): a kernel that is an outer product u v^T to f32 rounding (gaussian_kernel_2d is,
 * image_ops.rs:188-220) is multiplied in as the two 1-D spectra of u and v, and on 1024 x 1024 images its convolution runs as two
 * separable passes (rows, then columns) — the same linear operators as fft2d . product . ifft2d, within the f32 tolerance of the
 * general path.  Test switches, read at the call: SGX_CONV_RANK1=0 (every kernel through its full 2-D spectrum),
 * SGX_CONV_SEPARABLE=0 (rank-1 kernels through the three passes), SGX_SEP_GROUP=n (images per launch pair of the separable passes). */
sgx_status sgx_fft2d_convolve(sgx_fft2d *plan, const void *images, size_t batch, const void *kernel_host, size_t krows,
                              size_t kcols, void *out, int32_t mem_kind, void *hip_stream);
/* lowpass (kind 0, cut_lo), highpass (1, cut_lo), bandpass (2, cut_lo..cut_hi): binary radial masks built on the HALF
 * spectrum's own dimensions (image_ops.rs:236-267, 301-432 — the reference's quirk S14 is reproduced) */
sgx_status sgx_fft2d_filter(sgx_fft2d *plan, const void *images, size_t batch, int32_t kind, double cut_lo, double cut_hi,
                            void *out, int32_t mem_kind, void *hip_stream);
/* Pre-sizes the plan-owned intermediates (and, with `host_staging`, the device staging of the SGX_MEM_HOST paths) for calls of
 * up to `batch` images, so that those calls do not allocate. */
sgx_status sgx_fft2d_reserve(sgx_fft2d *plan, size_t batch, int32_t host_staging);
int32_t sgx_fft2d_device(const sgx_fft2d *plan); /* the HIP device ordinal the plan is bound to */
const char *sgx_fft2d_last_error(const sgx_fft2d *plan);

/* ---- 1-D complex-to-complex plan: C2cPlan<T> (src/fft_backend.rs:113-137; `Sample::plan_c2c`, src/sample.rs:61-66).  In
 * place, UNNORMALISED in both directions (the caller divides by n after an inverse, as the trait says), host pointers, `len`
 * complex values of T = n else SGX_DIM_MISMATCH.  Everything is allocated at creation. */
typedef struct sgx_c2c sgx_c2c; /* opaque; same single-caller rule as sgx_plan */
sgx_status sgx_c2c_create(size_t n, int32_t dtype, int32_t device, sgx_c2c **out);
void sgx_c2c_destroy(sgx_c2c *plan);
sgx_status sgx_c2c_forward(sgx_c2c *plan, void *buf, size_t len);
sgx_status sgx_c2c_inverse(sgx_c2c *plan, void *buf, size_t len);
const char *sgx_c2c_last_error(const sgx_c2c *plan);

/* ---- measurement utility ("verify the HBM peak on the box with a device memcpy / triad and quote the measured
 * peak next to the nominal"): streams `bytes` (0 = 1 GiB, four times the Infinity Cache) `iters` times with 16-byte accesses
 * from every CU and returns the rate in GB/s.  mode 0: copy (bytes read + bytes written per pass), 1: read only, 2: write only,
 * 3: one buffer read and two written per pass (the read : write mix of the linear-power STFT), counted as 3 x `bytes`.
 * Allocates and frees its own buffers on `device` (-1 = current); no plan involved, nothing on the transform path calls it. */
sgx_status sgx_membench(int32_t device, size_t bytes, int32_t mode, int32_t iters, double *gb_per_s);
/* The shader clock the device holds right now, in MHz: one wave per CU, enqueued on `hip_stream` behind whatever the caller has
 * launched there, counts shader cycles against the constant 100 MHz counter for ~20 us; the call waits for the stream and returns
 * the median over the CUs.  bench.py's `sustained` leg calls it between blocks of back-to-back launches (the clock a caller who
 * streams for seconds gets, next to the short timed region's). */
sgx_status sgx_clock_probe(int32_t device, void *hip_stream, double *mhz);

const char *sgx_last_error(const sgx_plan *plan);
const char *sgx_last_create_error(void);
/* Name of the kernel variant the plan dispatches to: the shape-specific kernels "r32x16_f32", "r32x32_f32", "r64x32_f32", "d512_f64", "d32x16_f64",
 * "d32x32_f64", or "reg_radix", "lds_radix2", "two_factor_dft", "bluestein", "direct_dft", or — frame lengths past the on-chip kernels, every n_fft up
 * to 2^20 (powers of two 2^21) — "big_four_step" / "big_chirpz" (transforms through global memory) (a diagnostic: a call may step down this chain for
 * shapes the plan's kernel does not take). */
const char *sgx_kernel_name(const sgx_plan *plan);
int32_t sgx_abi_version(void);
int32_t sgx_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* SPECTRO_HIP_H */
