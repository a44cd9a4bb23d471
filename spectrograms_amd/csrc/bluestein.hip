// bluestein.hip — forward STFT frames of lengths the register-tiled / two-factor kernels do not take at speed (primes, 2 x prime,
// other lengths with a large prime factor) as a chirp-z transform on top of the power-of-two complex kernels.
//
// The reference plans EVERY length through realfft / RustFFT (src/fft_backend.rs:376-385), which pick mixed radix, Rader or
// Bluestein per length; round 2 ran such frames as an O(n^2) direct sum (n_fft 5003: 12.5 M multiply-adds per frame).
//
//   X[k] = sum_n x[n] W^(n k),  W = e^(-2 pi i / N),  n k = (n^2 + k^2 - (k - n)^2) / 2
//        = conj(c_k) * sum_n (x[n] conj(c_n)) c_(k - n),        c_n = e^(+i pi n^2 / N)
//
// i.e. a circular convolution of a[n] = w[n] x[n] conj(c_n) (zero-padded to M >= 2 N - 1, M a power of two) with the wrapped
// chirp b[n] = c_n (|n| < N):  Y = IFFT_M(FFT_M(a) . FFT_M(b)), X[k] = conj(c_k) Y[k].  FFT_M(b) / M is a plan table (built on the
// host in f64); the chirp's angle is reduced exactly in integers (n^2 mod 2 N) before the f64 sin / cos.
//
// Five launches per chunk of frames over two plan-owned scratch buffers of [frames][M] complex T:
//   k_bs_pre   framing with virtual zero padding (S1), window multiply in T (S4), chirp        x -> A
//   C2C        forward, length M (k_c2c_reg up to 4096, k_c2c_tile above)                       A -> B
//   k_pointwise  B *= FFT_M(b) / M
//   C2C        inverse (unnormalised)                                                           B -> A
//   k_bs_post  conj(c_k) Y[k], k <= N / 2, |.|^2 / sqrt / dB or complex, transposed through LDS into the reference's
//              [signal][bin][frame] layout (S9)
// Filterbank outputs take the plan's split path: per-bin power here, then k_bank_rows.
#include "sgx_internal.h"

namespace sgx {
namespace {

template <typename T>
struct C2 {
    T re, im;
};

template <typename T>
__global__ __launch_bounds__(256) void k_bs_pre(const T *__restrict__ x, const T *__restrict__ win, const C2<T> *__restrict__ chirp,
                                                C2<T> *__restrict__ a, unsigned long long g0, unsigned M, unsigned n, unsigned hop,
                                                unsigned pad, unsigned long long n_samples, unsigned long long stride, unsigned n_frames) {
    const unsigned m = blockIdx.x * 256u + threadIdx.x;
    if (m >= M) return;
    const unsigned long long g = g0 + blockIdx.y;
    const unsigned long long b = g / n_frames;
    const unsigned f = (unsigned)(g - b * n_frames);
    C2<T> v = {T(0), T(0)};
    if (m < n) {
        const long long s = (long long)f * hop + m - pad;  // virtual index into the zero-padded signal (spectrogram.rs:1301-1320)
        if (s >= 0 && (unsigned long long)s < n_samples) {
            const T xv = x[b * stride + (unsigned long long)s] * win[m];  // sample x window in T, then the transform in T (S4)
            const C2<T> c = chirp[m];
            v.re = xv * c.re;
            v.im = xv * c.im;
        }
    }
    a[(unsigned long long)blockIdx.y * M + m] = v;
}

__device__ __forceinline__ float bs_db(float p) { return __builtin_log2f(p) * 3.01029995663981195f; }  // as the other f32 kernels
__device__ __forceinline__ double bs_db(double p) { return 10.0 * log10(p); }

// 32 frames x 32 bins per workgroup: read along bins (contiguous in Y), write along frames (contiguous in the output)
template <typename T>
__global__ __launch_bounds__(256) void k_bs_post(const C2<T> *__restrict__ y, const C2<T> *__restrict__ chirp, T *__restrict__ out,
                                                 unsigned long long g0, unsigned count, unsigned M, unsigned nb, unsigned n_frames,
                                                 int complex_out, int amp, T eps) {
    __shared__ C2<T> tile[32][33];
    const unsigned tx = threadIdx.x & 31u, ty = threadIdx.x >> 5;
    const unsigned k0 = blockIdx.x * 32u, f0 = blockIdx.y * 32u;
#pragma unroll
    for (unsigned r = 0; r < 4; ++r) {
        const unsigned fl = f0 + ty + 8u * r, k = k0 + tx;
        C2<T> X = {T(0), T(0)};
        if (fl < count && k < nb) {
            const C2<T> v = y[(unsigned long long)fl * M + k], c = chirp[k];
            X.re = v.re * c.re - v.im * c.im;
            X.im = v.re * c.im + v.im * c.re;
        }
        tile[ty + 8u * r][tx] = X;
    }
    __syncthreads();
#pragma unroll
    for (unsigned r = 0; r < 4; ++r) {
        const unsigned k = k0 + ty + 8u * r, fl = f0 + tx;
        if (fl >= count || k >= nb) continue;
        const C2<T> X = tile[tx][ty + 8u * r];
        const unsigned long long g = g0 + fl, b = g / n_frames, f = g - b * n_frames;
        const unsigned long long o = (b * nb + k) * n_frames + f;
        if (complex_out) {
            ((C2<T> *)out)[o] = X;
        } else {
            const T p = X.re * X.re + X.im * X.im;  // norm_sqr (spectrogram.rs:1332-1334)
            out[o] = amp == AMP_MAGNITUDE ? sqrt(p) : amp == AMP_DB ? bs_db(p > eps ? p : eps) : p;
        }
    }
}

template <typename T>
hipError_t run_t(const BsArgs &a, int dtype, hipStream_t s) {
    const unsigned long long total = (unsigned long long)a.batch * a.n_frames;
    const unsigned long long chunk = a.chunk_frames < 32768ull ? a.chunk_frames : 32768ull;  // grid.y
    if (chunk == 0) return hipErrorInvalidConfiguration;
    for (unsigned long long g0 = 0; g0 < total; g0 += chunk) {
        const unsigned count = (unsigned)(total - g0 < chunk ? total - g0 : chunk);
        hipLaunchKernelGGL(k_bs_pre<T>, dim3((a.M + 255u) / 256u, count), dim3(256), 0, s, (const T *)a.x, (const T *)a.window,
                           (const C2<T> *)a.chirp, (C2<T> *)a.scratch_a, g0, a.M, a.n_fft, a.hop, a.pad, a.n_samples, a.sample_stride, a.n_frames);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        C2cArgs c{};
        c.in = a.scratch_a; c.out = a.scratch_b;
        c.n = a.M; c.log2n = a.log2M; c.nseq = count; c.batch = 1;
        c.in_img = c.out_img = 0;
        c.in_ss = c.out_ss = a.M; c.in_is = c.out_is = 1;
        c.tile = a.c2c_tile; c.tiles = c.tile ? (count + c.tile - 1) / c.tile : 0;
        c.tw = a.tw_m; c.inverse = 0; c.in_seq_fast = 0; c.out_seq_fast = 0; c.scale = 1.0;
        if ((e = launch_c2c_any(c, dtype, s)) != hipSuccess) return e;
        if ((e = launch_pointwise(a.scratch_b, a.bhat, a.scratch_b, (unsigned long long)count * a.M, a.M, 0, dtype, s)) != hipSuccess) return e;
        c.in = a.scratch_b; c.out = a.scratch_a; c.inverse = 1;
        if ((e = launch_c2c_any(c, dtype, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_bs_post<T>, dim3((a.nb + 31u) / 32u, (count + 31u) / 32u), dim3(256), 0, s, (const C2<T> *)a.scratch_a,
                           (const C2<T> *)a.chirp, (T *)a.out, g0, count, a.M, a.nb, a.n_frames, a.complex_out, a.amp, (T)a.eps);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace

hipError_t launch_bluestein(const BsArgs &a, int dtype, hipStream_t s) {
    return dtype == SGX_F64 ? run_t<double>(a, dtype, s) : run_t<float>(a, dtype, s);
}

}  // namespace sgx
