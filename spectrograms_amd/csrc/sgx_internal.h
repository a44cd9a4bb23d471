// sgx_internal.h — shared between the host plan code and the HIP kernels of libspectro_hip.so.
// Product code: nothing here includes, links or calls the CPU oracle (oracle/).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <mutex>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "spectro_hip.h"

namespace sgx {

enum OutMode : int { OUT_LINEAR = 0, OUT_MEL = 1, OUT_COMPLEX = 2 };
// AMP_MAG_IN: the mapping consumes sqrt(power) and its output is final (chromagram: bank applied to magnitudes)
enum AmpMode : int { AMP_POWER = 0, AMP_MAGNITUDE = 1, AMP_DB = 2, AMP_MAG_IN = 3 };
enum KernelKind : int { K_DIRECT_DFT = 0, K_LDS_RADIX2 = 1, K_R32X16_F32 = 2, K_TWO_FACTOR = 3, K_REG_RADIX = 4, K_BLUESTEIN = 5, K_R32X32_F32 = 6, K_D32X16_F64 = 7, K_D512_F64 = 8, K_R64X32_F32 = 9, K_D32X32_F64 = 10, K_BIGFFT = 11 };
inline bool kind_is_tuned(KernelKind k) { return k == K_R32X16_F32 || k == K_R32X32_F32 || k == K_D32X16_F64 || k == K_D512_F64 || k == K_R64X32_F32 || k == K_D32X32_F64; }  // the shape-specific kernels at the head of the chain

// Kernel arguments (POD, passed by value).  Layouts in HBM:
//   x      : [batch][sample_stride] T, row b valid for n_samples elements
//   out    : [batch][n_out][n_frames] T  (frames contiguous — reference S9), complex: (re,im) pairs
//   window : [n_fft] T            tw : [n_fft] complex<T>, tw[k] = exp(-2 pi i k / n_fft)
//   mel    : CSR, row_ptr[n_mels+1], cols ascending within a row, vals already cast to T
struct StftArgs {
    const void *x;
    void *out;
    unsigned long long sample_stride;
    unsigned long long n_samples;
    unsigned batch;
    unsigned n_fft, m, log2m, hop, pad;
    unsigned n_frames, nb_fft, n_out;
    unsigned ft, tiles;  // frames per workgroup tile, tiles per signal
    unsigned mel_sub;    // k_reg_radix, filterbank outputs: frames per |X|^2 part (a power of two dividing ft; 0 = ft)
    unsigned staged;     // k_reg_radix: the tile's samples go through LDS (overlapping frames, per-bin outputs: plan_geometry_reg_radix)
    unsigned tw_lds;        // K_LDS_RADIX2: the first n_fft/2 twiddles are copied to LDS behind the tile (set by the geometry)
    unsigned fac_a, fac_b;  // K_TWO_FACTOR: n_fft = fac_a * fac_b, fac_a = largest divisor <= sqrt(n_fft)
    const void *window;
    const void *tw;
    const unsigned *mel_ptr;
    const unsigned *mel_col;
    const void *mel_val;
    // 4-wide padded band table (contiguous bands only, else nullptr): band m covers columns mel_pcol[m] + 4*c + {0..3}
    // for c in [mel_pptr[m], mel_pptr[m+1]) with weights mel_pw[4*c + {0..3}] (zeros outside the true band)
    const unsigned *mel_pptr;
    const unsigned *mel_pcol;
    const void *mel_pw;
    unsigned mel_pchunks;
    // banded / dense bank for the tuned f32 kernel's matrix-core epilogue (Mel with wide bands, ERB).  Block = 16 rows of
    // the bank; its support [lo, lo + 16 n16 + 4 n4) is a multiple-of-4 cover of the rows' non-zero bins (<= 516).
    //   mm_blk[blk]  = {fragment offset, lo, n16, n4 | owner wave << 8}
    //   mm_frag      float4 [fragment][64 lanes]: chunk c < n16: element s = weight[16 blk + (l & 15)][lo + 16 c + 4 (l >> 4) + s];
    //                tail fragment (if n4 > 0): element s < n4 = weight[..][lo + 16 n16 + 4 s + (l >> 4)]
    const void *mm_frag;
    const uint4 *mm_blk;
    unsigned mm_nblk;
    unsigned n_mels;
    unsigned mel_nnz;
    unsigned mel_contig;  // every row of the bank is one run of consecutive columns (Mel, log-Hz, ERB): col[i] = col[ptr] + (i - ptr)
    int out_mode;
    int amp;
    double eps;  // 10^(floor_db/10) in f64; cast to T in the kernel (T::from_f64, spectrogram.rs:2028)
    // tuned-kernel tables (K_R32X16_F32)
    const void *tw1;  // [32][16] complex<f32>: W_512^(k1*n2)
    const void *tw2;  // float4 [16 jobs][17]: the real-split twiddles in each job's consumption order (r32x16_layout.h)
    // filterbank schedule of the tuned kernel (r32x16_layout.h), nullptr: bank not schedulable (matrix cores / CSR path)
    const unsigned *mel_sched;
    unsigned mel_sched_words;
    // tuned kernel, packed tiles (batches of short signals; set by its launcher): batch * n_frames, bytes of the whole sample buffer
    unsigned gframes, x_bytes;
    // fused MFCC epilogue of the tuned f32 kernel (kernels_r32x16.hip mfcc_tile; nullptr: separate launch_mfcc): the DCT-II basis as
    // v_mfma_f32_16x16x4_f32 A-fragments [mtiles][steps][64 lanes] = basis[16 mt + (l & 15)][4 s + (l >> 4)] (0 beyond n_mfcc / n_mels),
    // followed by the lifter weights [n_mfcc] (1.0 without a lifter); `out` then is the MFCC tensor [batch][n_mfcc - mfcc_skip][n_frames]
    const void *mfcc_frag;
    unsigned mfcc_frag_words, mfcc_steps, mfcc_mtiles, n_mfcc, mfcc_skip;
};

// launchers (kernels_generic.hip / kernels_r32x16.hip); return hipSuccess or the launch error
hipError_t launch_direct_dft(const StftArgs &a, int dtype, hipStream_t s);
hipError_t launch_lds_radix2(const StftArgs &a, int dtype, hipStream_t s);
hipError_t launch_r32x16_f32(const StftArgs &a, hipStream_t s);
hipError_t launch_r32x32_f32(const StftArgs &a, hipStream_t s);  // n_fft 2048 (kernels_r32x32.hip)
bool plan_geometry_r32x32_f32(StftArgs &a);
hipError_t launch_d32x16_f64(const StftArgs &a, hipStream_t s);  // f64 n_fft 1024, per-bin and complex outputs (kernels_d32x16.hip)
bool plan_geometry_d32x16_f64(StftArgs &a);
hipError_t launch_d512_f64(const StftArgs &a, hipStream_t s);  // f64 n_fft 512, even hops <= 260, two frames per transform (kernels_d32x16.hip)
bool plan_geometry_d512_f64(StftArgs &a);
hipError_t launch_r64x32_f32(const StftArgs &a, hipStream_t s);  // f32 n_fft 4096, per-bin and complex outputs (kernels_r64x32.hip)
bool plan_geometry_r64x32_f32(StftArgs &a);
hipError_t launch_d32x32_f64(const StftArgs &a, hipStream_t s);  // f64 n_fft 2048, per-bin and complex outputs (kernels_d32x32.hip)
bool plan_geometry_d32x32_f64(StftArgs &a);
// MFCC epilogue over a Mel-dB tensor [batch][n_mels][n_frames] -> [batch][n_out][n_frames]; basis [n_mfcc][n_mels], lifter [n_mfcc]
hipError_t launch_mfcc(const void *mel, void *out, const void *basis, const void *lifter, unsigned batch, unsigned n_mels,
                       unsigned n_frames, unsigned n_mfcc, unsigned skip, int has_lifter, int dtype, hipStream_t s);
// tile geometry chosen per kernel (fills a.ft / a.tiles); returns false if the kernel cannot run the shape
bool plan_geometry_direct_dft(StftArgs &a, int dtype);
bool plan_geometry_two_factor(StftArgs &a, int dtype);
hipError_t launch_two_factor(const StftArgs &a, int dtype, hipStream_t s);
bool plan_geometry_lds_radix2(StftArgs &a, int dtype);
bool plan_geometry_reg_radix(StftArgs &a, int dtype);
hipError_t launch_reg_radix(const StftArgs &a, int dtype, hipStream_t s);
// filterbank rows over a [batch][nb_fft][n_frames] power / magnitude tensor (split filterbank path): CSR bank, amp and eps from `a`
hipError_t launch_bank_rows(const void *pw, void *out, const StftArgs &a, int dtype, hipStream_t s);
bool plan_geometry_r32x16_f32(StftArgs &a);

// chirp-z (Bluestein) forward frames on the power-of-two complex kernels (bluestein.hip): lengths with a large prime factor
struct BsArgs {
    const void *x;
    void *out;  // [batch][nb][n_frames] T, or complex pairs
    unsigned long long sample_stride, n_samples;
    unsigned batch, n_fft, hop, pad, n_frames, nb;
    unsigned M;             // convolution length: the power of two >= 2 n_fft - 1
    const void *wc;         // [n_fft] complex T: window[m] conj(c_m), c_n = e^(+i pi n^2 / n_fft)
    const void *chirp;      // [n_fft] complex T: conj(c_n)
    const void *bhat_fused; // [M] complex T: FFT_M of the wrapped chirp, divided by M, in the order the kernel's product step reads it
    const void *tw_m;       // [M] complex T: e^(-2 pi i k / M)
    int complex_out, amp;
    double eps;
    const unsigned *mel_ptr, *mel_col;  // filterbank outputs: CSR rows of the bank (null: per-bin output), amp applies to its sums
    const void *mel_val;
    unsigned n_mels, n_out;
};
hipError_t launch_bluestein(const BsArgs &a, int dtype, hipStream_t s);
bool bluestein_fused_split(unsigned M, int dtype, unsigned *fa, unsigned *fb, unsigned *fc);
// chirp-z for complex sequences / Hermitian rows at lengths without a pass split (bluestein.hip)
struct BsHostTables {
    unsigned M = 0;
    std::vector<double> chirp, bhp, tw;  // interleaved (re, im): conj(c) [n], FFT_M(b) / M in the kernel's product order [M], W_M [M]
};
struct BsDevTables {
    unsigned M = 0;
    void *chirp = nullptr, *bhp = nullptr, *tw = nullptr;
};
bool bluestein_host_tables(unsigned n, int dtype, BsHostTables &t);
struct C2cArgs;
struct C2rArgs;
hipError_t launch_c2c_bluestein(const C2cArgs &a, const BsDevTables &t, int dtype, hipStream_t s);
hipError_t launch_c2c_bluestein_split(const C2cArgs &a, const BsDevTables &t, void *scratch, int dtype, hipStream_t s);
hipError_t launch_bluestein_half(const BsArgs &a, const BsDevTables &t, const void *window, const void *twn, int dtype, hipStream_t s);
hipError_t launch_c2r_bluestein(const C2rArgs &a, const BsDevTables &t, int dtype, hipStream_t s, bool half = false);

// ---- 2-D FFT path (kernels_fft2d.hip)
struct C2cArgs {
    const void *in;
    void *out;
    unsigned n, log2n;  // log2n == 0: not a power of two (two-factor or direct DFT)
    unsigned n1;        // k_c2c_tile, set by its launcher: n = n1 * n2 two-factor transform through a second LDS buffer (0: direct sum)
    unsigned nseq, batch;
    unsigned long long in_img, out_img;              // elements between images
    unsigned long long in_ss, in_is, out_ss, out_is;  // sequence / index strides (elements)
    unsigned tile, tiles;
    const void *tw;  // e^{-2 pi i k / n}, n entries
    int inverse;
    int in_seq_fast, out_seq_fast;  // which of (sequence, index) is the unit-stride side: drives the thread mapping
    double scale;
    // optional product fused into the store of the register-tiled kernel (convolve_fft / filters): output element k of
    // sequence q is multiplied by mul[k * mul_ks + q] — complex (mul_real == 0) or real mask (mul_real == 1)
    const void *mul;
    unsigned long long mul_ks;
    int mul_real;
    int mul_bcast;  // 1: mul[k * mul_ks] for every sequence (the chirp-z path's transformed chirp)
};
struct C2rArgs {
    const void *in;  // half spectrum, element (row r, col k) at in[b*in_img + k*in_ks + r*in_rs]
    void *out;       // real [batch][nrows][ncols]
    unsigned nrows, ncols, log2c, batch;
    unsigned long long in_img, in_ks, in_rs;
    int k_fast;  // input is [r][k]-major (in_ks == 1): map threads with k fastest
    unsigned n1;  // k_c2r_rows, set by its launcher: ncols = n1 * n2 two-factor transform (0: direct sum)
    unsigned tile, tiles;
    const void *tw;  // e^{-2 pi i k / ncols}, ncols entries
    double scale;
    // ISTFT use (rows = frames): optional synthesis window applied after the scale, and a flag word set when a DC /
    // Nyquist bin carries a non-zero imaginary part (realfft's C2R reports FftError::InputValues for that)
    const void *win;
    unsigned *bad_flag;
    // fused inverse STFT (k_c2r_reg<..., OLA = true>, launch_istft_reg): rows are the frames of one signal, a tile is `tile`
    // consecutive frames of which the first `ov` are halo (recomputed by the previous tile), the windowed frames stay in LDS and
    // are overlap-added into out[batch][out_len] (the padded signal from `start`); nrows = n_frames
    unsigned hop, nbk, ov;
    unsigned long long start, out_len;
};
unsigned fft2d_tile_for(unsigned n, int dtype);
unsigned c2r_tile_for(unsigned n, int dtype, size_t lds_budget);
// overlap-add + normalisation of windowed frames [batch][n_frames][n] into out[batch][out_len] (istft, spectrogram.rs:4911-4930)
hipError_t launch_istft_ola(const void *frames, const void *win, void *out, unsigned n, unsigned hop, unsigned n_frames,
                            unsigned long long start, unsigned long long out_len, unsigned batch, int dtype, hipStream_t s);
hipError_t launch_c2c_tile(const C2cArgs &a, int dtype, hipStream_t s);
hipError_t launch_c2r_rows(const C2rArgs &a, int dtype, hipStream_t s);
// register-tiled versions (kernels_reg2d.hip); they pick their own tile.  hipErrorNotSupported: no pass split for this length
// or layout — use the LDS-tile kernels above.
hipError_t launch_c2c_reg(const C2cArgs &a, int dtype, hipStream_t s);
hipError_t launch_c2r_reg(const C2rArgs &a, int dtype, hipStream_t s);
// fused inverse STFT for every length with a pass split: spec [batch][n/2+1][n_frames] -> out [batch][out_len]; hipErrorNotSupported:
// use launch_c2r_reg / launch_c2r_rows into a frame scratch + launch_istft_ola
hipError_t launch_istft_reg(const void *spec, void *out, const void *win, const void *tw, unsigned n, unsigned n_frames, unsigned hop,
                            unsigned batch, unsigned long long start, unsigned long long out_len, double scale, unsigned *bad_flag,
                            int dtype, hipStream_t s);
// would launch_istft_reg run this shape fused (hipSuccess) — the launcher's own geometry test, for sgx_reserve
bool istft_reg_fuses(const void *win, unsigned n, unsigned n_frames, unsigned hop, unsigned batch, int dtype);
inline hipError_t launch_c2c_any(const C2cArgs &a, int dtype, hipStream_t s) {
    const hipError_t e = launch_c2c_reg(a, dtype, s);
    return e == hipErrorNotSupported ? launch_c2c_tile(a, dtype, s) : e;
}
inline hipError_t launch_c2r_any(const C2rArgs &a, int dtype, hipStream_t s) {
    const hipError_t e = launch_c2r_reg(a, dtype, s);
    return e == hipErrorNotSupported ? launch_c2r_rows(a, dtype, s) : e;
}
// tuned f32 1024-point C2C, 16 sequences per workgroup; tw1c = W_1024^(k1*n2), [32][32] complex f32; output must be
// sequence-contiguous for coalesced stores (a.out_ss == 1)
hipError_t launch_c2c1024(const C2cArgs &a, const void *tw1c, hipStream_t s);
// fused column stage of the 2-D convolution / filters for 1024 rows, f32: forward FFT, product with `mul`, inverse FFT;
// in = [col][row] (a.in_ss = rows, a.in_is = 1), out = [row][col] (a.out_ss = 1, a.out_is = cols).  `mul` by mul_kind:
// MUL_SPECTRUM a complex kernel spectrum, MUL_MASK a real mask (both [row][col] with row stride mul_row), MUL_OUTER the two factors
// of a rank-1 kernel's spectrum K[row][col] = U[row] V[col]: 1024 complex U, then a.nseq complex V (mul_row unused)
// real_io (with MUL_VEC: `mul` = 1024 complex values, one factor of a rank-1 kernel's spectrum): a sequence is the pair of rows (2 q, 2 q + 1)
// of a real [2 nseq][1024] array (a.in_ss = its pitch, a.in_img = floats per image), written to the transposed real array
// out[n][2 q .. 2 q + 1] (a.out_is = its pitch, a.out_img = floats per image)
enum { MUL_SPECTRUM = 0, MUL_MASK = 1, MUL_OUTER = 2, MUL_VEC = 3 };
hipError_t launch_colconv1024(const C2cArgs &a, const void *tw1c, const void *mul, unsigned long long mul_row, int mul_kind,
                              hipStream_t s, bool real_io = false);
// tuned f32 inverse row pass for ncols == 1024 on a [r][k]-major half spectrum (a.in_ks == 1), 16 rows per workgroup
hipError_t launch_c2r1024(const C2rArgs &a, const void *twr, const void *tw1, hipStream_t s);
// fused f32 n_fft = 1024 inverse STFT (C2R + window + overlap-add + normalise + trim); hop >= 64; twr/tw1 as launch_c2r1024
hipError_t launch_istft_d1024(const void *spec, void *out, const void *win, unsigned n_frames, unsigned hop, unsigned batch, unsigned long long start,
                              unsigned long long out_len, double scale, unsigned *bad_flag, const void *twr, const void *tw1,
                              hipStream_t s);  // fused tuned f64 n_fft 1024 inverse, hop >= 64 (kernels_istft_d1024.hip)
hipError_t launch_istft_d512(const void *spec, void *out, const void *win, unsigned n_frames, unsigned hop, unsigned batch, unsigned long long start,
                             unsigned long long out_len, double scale, unsigned *bad_flag, const void *tw1,
                             hipStream_t s);  // fused f64 n_fft 512 inverse, two frames per transform, hop >= 32 (kernels_istft_d1024.hip)
hipError_t launch_istft2048(const void *spec, void *out, const void *win, unsigned n_frames, unsigned hop, unsigned batch, unsigned long long start,
                            unsigned long long out_len, float scale, unsigned *bad_flag, const void *twr, const void *tw1, hipStream_t s);  // kernels_istft2048.hip
hipError_t launch_istft1024(const void *spec, void *out, const void *win, unsigned n_frames, unsigned hop, unsigned batch,
                            unsigned long long start, unsigned long long out_len, float scale, unsigned *bad_flag,
                            const void *twr, const void *tw1, hipStream_t s);
// per-frame normalisation over the 12 chroma rows, in place (apply_chroma_normalization, src/chroma.rs:403-445)
hipError_t launch_chroma_norm(void *data, unsigned batch, unsigned n_frames, int norm, int dtype, hipStream_t s);
hipError_t launch_pointwise(const void *x, const void *y, void *out, unsigned long long n, unsigned long long per, int mode,
                            int dtype, hipStream_t s);

// ---- lengths beyond the on-chip kernels (bigfft.hip): four-step transforms through global memory for powers of two, chirp-z on top
// of them for every other length; n_fft up to 2^20 (powers of two: 2^21), both types
struct BigHost {  // tables as built on the host in f64, interleaved (re, im)
    unsigned n = 0, M = 0, M1 = 0, M2 = 0, l1 = 0, l2 = 0;
    bool chirp = false;
    std::vector<double> wl, thi, tlo, c, bhat;
};
struct BigDev {
    unsigned n = 0, M = 0, M1 = 0, M2 = 0, l1 = 0, l2 = 0;
    bool chirp = false;
    void *wl = nullptr, *thi = nullptr, *tlo = nullptr, *c = nullptr, *bhat = nullptr;
};
constexpr size_t kBigChunkBytes = size_t(256) << 20;  // scratch per buffer: a call is cut into chunks of sequences of at most this size
bool big_supported(unsigned long long n);
bool big_host_tables(unsigned n, BigHost &h);
hipError_t big_upload(const BigHost &h, int dtype, BigDev &d);
void big_free(BigDev &d);
size_t big_scratch_bytes(const BigDev &t, int dtype, size_t nseq);  // for nseq complex sequences (a sequence carries two real frames)
hipError_t launch_big_stft(const BigDev &t, const StftArgs &a, void *scratch, int dtype, hipStream_t s);   // per-bin and complex outputs
hipError_t launch_big_c2r(const BigDev &t, const C2rArgs &c, void *scratch, int dtype, hipStream_t s);     // Hermitian rows -> real rows [batch][nrows][ncols]
hipError_t launch_big_c2c(const BigDev &t, const C2cArgs &c, void *scratch, int dtype, hipStream_t s);     // complex sequences (batch == 1)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device setting: remember which (kernel, device) pairs have been
// configured, so a process that drives several GPUs (sgx_params.device) gets the large-LDS opt-in on each of them.
inline hipError_t set_max_dynamic_lds(const void *fn, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> guard(mu);
    if (done.count({fn, dev})) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert({fn, dev});
    return e;
}

// Compute units of the CURRENT device (the entry points have made the plan's device current), looked up once per device: a
// process may drive GPUs with different CU counts or partition modes, and the persistent grids are sized per device.
// Inverse STFT with the overlap carried in LDS (k_istft1024c, k_istft2048): every signal's `tiles` are cut into R equal runs of
// consecutive tiles, one run per workgroup visit.  A run that starts inside a signal first passes over the tile in front of it to
// build its carry (nothing when the frames do not overlap: ov == 0), so the time is about rounds x (run_len + warm-up) tile periods,
// rounds = ceil(R * batch / workgroups): the R with the smallest product, the smallest such R.
inline void istft_carry_runs(unsigned tiles, unsigned batch, unsigned wgs, unsigned ov, unsigned &R, unsigned &run_len) {
    unsigned long long best = ~0ull;
    R = 1;
    for (unsigned r = 1; r <= tiles; ++r) {
        const unsigned len = (tiles + r - 1u) / r;
        if ((tiles + len - 1u) / len != r) continue;  // (the same cut as a smaller r: no empty runs)
        const unsigned long long rounds = ((unsigned long long)r * batch + wgs - 1u) / wgs;
        const unsigned long long cost = rounds * (len + ((r > 1u && ov) ? 1u : 0u));
        if (cost < best) {
            best = cost;
            R = r;
        }
        if (rounds > 1u && len <= 2u) break;  // shorter runs only add rounds and warm-up tiles from here
    }
    run_len = (tiles + R - 1u) / R;
}

inline unsigned device_cu_count() {
    static std::mutex mu;
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256u;
    std::lock_guard<std::mutex> guard(mu);
    if (cached[dev] <= 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return (unsigned)cached[dev];
}

// Entry points run on the plan's device and leave the caller's current device as they found it.
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    hipError_t enter(int dev) {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) return e;
        if (prev != dev) {
            e = hipSetDevice(dev);
            changed = e == hipSuccess;
        }
        return e;
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
};

}  // namespace sgx

struct sgx_plan {
    sgx_params p{};
    std::vector<double> custom_window;
    int device = -1;       // -2: host-only plan
    bool device_ready = false;
    int dtype = SGX_F32;
    size_t elem = 4;
    unsigned nb_fft = 0, n_out = 0;
    int out_mode = 0, amp = 0;
    double eps = 0.0;
    sgx::KernelKind kind = sgx::K_DIRECT_DFT;

    // host tables (f64, as the reference builds them)
    std::vector<double> window;
    std::vector<uint32_t> mel_ptr, mel_col;
    std::vector<double> mel_val;
    std::vector<double> loghz_freqs;  // LogHz / ERB axis (centre frequencies), empty otherwise

    // device tables
    void *d_window = nullptr, *d_tw = nullptr, *d_tw1 = nullptr, *d_tw2 = nullptr;
    void *d_mel_ptr = nullptr, *d_mel_col = nullptr, *d_mel_val = nullptr, *d_mel_pptr = nullptr, *d_mel_pcol = nullptr, *d_mel_pw = nullptr, *d_mm_frag = nullptr, *d_mm_blk = nullptr, *d_mel_sched = nullptr;
    unsigned mel_sched_words = 0;
    std::vector<uint32_t> h_mel_sched;  // the tuned kernel's band schedule as built on the host (plan.hip build_band_schedule)
    unsigned mm_nblk = 0;
    unsigned mel_pchunks = 0;
    unsigned mel_contig = 0;
    void *d_ones = nullptr;  // rectangular window for sgx_r2c
    // MFCC epilogue: DCT-II basis [n_mfcc][n_mels] and lifter [n_mfcc] in T; Mel-dB scratch (grown on demand)
    void *d_dct = nullptr, *d_lifter = nullptr, *d_melbuf = nullptr;
    void *d_mfcc_frag = nullptr;  // fused MFCC epilogue of the tuned f32 kernel: the basis as matrix-core fragments (null: separate launch)
    unsigned mfcc_frag_words = 0, mfcc_steps = 0, mfcc_mtiles = 0;
    size_t d_melbuf_bytes = 0;
    // split filterbank path (long frames): the per-bin power / magnitude tensor between the two launches (grown on demand)
    void *d_pwbuf = nullptr;
    size_t d_pwbuf_bytes = 0;
    bool split_bank = false;  // decided at plan creation (plan.hip)
    unsigned n_final = 0;  // rows of the final output (n_out, or the MFCC row count)
    void *d_window_half = nullptr, *d_ones_half = nullptr;  // 0.5*window (exact) for the tuned kernel's real split
    // inverse path (sgx_istft / sgx_c2r), created on first use: full twiddle table e^{-2 pi i k/n}, frame scratch, flag
    void *d_itw = nullptr, *d_frames = nullptr, *d_flag = nullptr;
    void *d_itwr = nullptr, *d_itw1 = nullptr;  // tuned f32 n_fft = 1024 inverse: conj(W_1024^k) [32][16], W_512^(k1 n2) [32][16]
    void *d_itwr2 = nullptr, *d_itw12 = nullptr;  // tuned f32 n_fft = 2048 inverse: conj(W_2048^k) [1024], W_1024^(k1 n2) [32][32]
    bool istft_d512 = false;  // f64 n_fft 512, hop >= 32: the fused two-frames-per-transform inverse (d_itw1d holds its W_512^(k1 n2))
    void *d_itwrd = nullptr, *d_itw1d = nullptr;  // tuned f64 n_fft = 1024 inverse: conj(W_1024^k) [512], W_512^(k1 n2) [16][32] (f64)
    // K_BLUESTEIN: chirp, transformed chirp, length-M twiddles (the sequences themselves never leave LDS: no frame scratch)
    void *d_bs_chirp = nullptr, *d_bs_tw = nullptr, *d_bs_wc = nullptr, *d_bs_bhp = nullptr;
    unsigned bs_M = 0;
    bool bs_fwd_half = false;  // K_BLUESTEIN in half-length complex form (even n_fft whose own convolution does not fit LDS)
    sgx::BsDevTables bs_half;  // inverse rows of an even n_fft whose own chirp-z does not fit: tables of length n_fft / 2 (inverse_tables)
    size_t d_frames_bytes = 0;
    // K_BIGFFT: tables of the global-memory transforms and their sequence scratch (grown on demand, pre-sized by sgx_reserve)
    sgx::BigDev big;
    unsigned big_n = 0;  // set at creation when the plan's kind is K_BIGFFT (host-only plans have no tables)
    void *d_big = nullptr;
    size_t d_big_bytes = 0;

    // plan-owned staging for host-pointer execution
    void *d_in = nullptr, *d_out = nullptr;
    size_t d_in_bytes = 0, d_out_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    mutable std::string err;
    mutable size_t dm_expected = 0, dm_got = 0;  // the last DimensionMismatch{expected, got} (src/error.rs:19-21)
};
