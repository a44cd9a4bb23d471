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
// The frames are real, so TWO frames ride one complex transform: frames 2p and 2p + 1 of a signal are the real and imaginary part
// of one sequence, a[n] = (x_2p[n] + i x_2p+1[n]) w[n] conj(c_n); with Z[k] = conj(c_k) Y[k] for ALL k < N the two spectra come
// apart by their Hermitian symmetry, X_2p[k] = (Z[k] + conj Z[N - k]) / 2, X_2p+1[k] = -i (Z[k] - conj Z[N - k]) / 2.  Pairs
// never cross a signal (an odd last frame rides alone), so a signal's output does not depend on the batch it is in.
//
// Four launches per chunk of frame pairs over two plan-owned scratch buffers of [pairs][M] complex T (five above M = 4096):
//   k_bs_pre   framing with virtual zero padding (S1), window multiply in T (S4), chirp        x -> A
//   C2C        forward, length M, its store multiplying by FFT_M(b) / M (k_c2c_reg up to M = 4096; above that k_c2c_tile and
//              a k_pointwise launch)                                                            A -> B
//   C2C        inverse (unnormalised)                                                           B -> A
//   k_bs_post  Z[k] = conj(c_k) Y[k], the two-frame split, k <= N / 2, |.|^2 / sqrt / dB or complex, transposed through LDS
//              into the reference's [signal][bin][frame] layout (S9)
// Filterbank outputs take the plan's split path: per-bin power here, then k_bank_rows.
#include "sgx_internal.h"

namespace sgx {
namespace {

template <typename T>
struct C2 {
    T re, im;
};

// sequence q = pair p of signal b (pairs per signal P = ceil(n_frames / 2)): real part frame 2 p, imaginary part frame 2 p + 1
template <typename T>
__global__ __launch_bounds__(256) void k_bs_pre(const T *__restrict__ x, const T *__restrict__ win, const C2<T> *__restrict__ chirp,
                                                C2<T> *__restrict__ a, unsigned long long q0, unsigned M, unsigned n, unsigned hop,
                                                unsigned pad, unsigned long long n_samples, unsigned long long stride, unsigned n_frames,
                                                unsigned pairs) {
    const unsigned m = blockIdx.x * 256u + threadIdx.x;
    if (m >= M) return;
    const unsigned long long q = q0 + blockIdx.y;
    const unsigned long long b = q / pairs;
    const unsigned f = 2u * (unsigned)(q - b * pairs);
    C2<T> v = {T(0), T(0)};
    if (m < n) {
        const T wm = win[m];
        const C2<T> c = chirp[m];
        // virtual index into the zero-padded signal (spectrogram.rs:1301-1320); sample x window in T, then the transform in T (S4)
        const long long s0 = (long long)f * hop + m - pad, s1 = s0 + hop;
        const T xa = (s0 >= 0 && (unsigned long long)s0 < n_samples) ? x[b * stride + (unsigned long long)s0] * wm : T(0);
        const T xb = (f + 1u < n_frames && s1 >= 0 && (unsigned long long)s1 < n_samples) ? x[b * stride + (unsigned long long)s1] * wm : T(0);
        v.re = xa * c.re - xb * c.im;
        v.im = xa * c.im + xb * c.re;
    }
    a[(unsigned long long)blockIdx.y * M + m] = v;
}

__device__ __forceinline__ float bs_db(float p) { return __builtin_log2f(p) * 3.01029995663981195f; }  // as the other f32 kernels
__device__ __forceinline__ double bs_db(double p) { return 10.0 * log10(p); }

// 16 pairs (32 frames) x 32 bins per workgroup: read along bins (contiguous in Y) — bin k and its mirror n - k —, write along
// frames (contiguous in the output)
template <typename T>
__global__ __launch_bounds__(256) void k_bs_post(const C2<T> *__restrict__ y, const C2<T> *__restrict__ chirp, T *__restrict__ out,
                                                 unsigned long long q0, unsigned count, unsigned M, unsigned n, unsigned nb, unsigned n_frames,
                                                 unsigned pairs, int complex_out, int amp, T eps) {
    __shared__ C2<T> tile[32][33];  // [frame within the workgroup][bin]
    const unsigned tx = threadIdx.x & 31u, ty = threadIdx.x >> 5;
    const unsigned k0 = blockIdx.x * 32u, p0 = blockIdx.y * 16u;
#pragma unroll
    for (unsigned r = 0; r < 2; ++r) {
        const unsigned pl = ty + 8u * r, k = k0 + tx;  // pair within the workgroup
        C2<T> Xa = {T(0), T(0)}, Xb = {T(0), T(0)};
        if (p0 + pl < count && k < nb) {
            const C2<T> *row = y + (unsigned long long)(p0 + pl) * M;
            const unsigned km = k == 0 ? 0u : n - k;
            const C2<T> v = row[k], c = chirp[k], vm = row[km], cm = chirp[km];
            const C2<T> Z = {v.re * c.re - v.im * c.im, v.re * c.im + v.im * c.re};
            const C2<T> Zm = {vm.re * cm.re - vm.im * cm.im, -(vm.re * cm.im + vm.im * cm.re)};  // conj Z[n - k]
            Xa.re = T(0.5) * (Z.re + Zm.re);
            Xa.im = T(0.5) * (Z.im + Zm.im);
            Xb.re = T(0.5) * (Z.im - Zm.im);   // -i (Z - conj Z[n - k]) / 2
            Xb.im = T(-0.5) * (Z.re - Zm.re);
        }
        tile[2u * pl][tx] = Xa;
        tile[2u * pl + 1u][tx] = Xb;
    }
    __syncthreads();
#pragma unroll
    for (unsigned r = 0; r < 4; ++r) {
        const unsigned k = k0 + ty + 8u * r, fl = tx;  // frame within the workgroup: pair fl / 2, member fl & 1
        const unsigned long long q = q0 + p0 + (fl >> 1);
        if (p0 + (fl >> 1) >= count || k >= nb) continue;
        const unsigned long long b = q / pairs;
        const unsigned f = 2u * (unsigned)(q - b * pairs) + (fl & 1u);
        if (f >= n_frames) continue;
        const C2<T> X = tile[fl][ty + 8u * r];
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
    const unsigned pairs = (a.n_frames + 1u) / 2u;
    const unsigned long long total = (unsigned long long)a.batch * pairs;
    const unsigned long long chunk = a.chunk_frames < 32768ull ? a.chunk_frames : 32768ull;  // sequences per pass (grid.y)
    if (chunk == 0) return hipErrorInvalidConfiguration;
    for (unsigned long long q0 = 0; q0 < total; q0 += chunk) {
        const unsigned count = (unsigned)(total - q0 < chunk ? total - q0 : chunk);
        hipLaunchKernelGGL(k_bs_pre<T>, dim3((a.M + 255u) / 256u, count), dim3(256), 0, s, (const T *)a.x, (const T *)a.window,
                           (const C2<T> *)a.chirp, (C2<T> *)a.scratch_a, q0, a.M, a.n_fft, a.hop, a.pad, a.n_samples, a.sample_stride, a.n_frames,
                           pairs);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        C2cArgs c{};
        c.in = a.scratch_a; c.out = a.scratch_b;
        c.n = a.M; c.log2n = a.log2M; c.nseq = count; c.batch = 1;
        c.in_img = c.out_img = 0;
        c.in_ss = c.out_ss = a.M; c.in_is = c.out_is = 1;
        c.tile = a.c2c_tile; c.tiles = c.tile ? (count + c.tile - 1) / c.tile : 0;
        c.tw = a.tw_m; c.inverse = 0; c.in_seq_fast = 0; c.out_seq_fast = 0; c.scale = 1.0;
        c.mul = a.bhat; c.mul_ks = 1; c.mul_real = 0; c.mul_bcast = 1;  // the product with the transformed chirp rides the store
        e = launch_c2c_reg(c, dtype, s);
        if (e == hipErrorNotSupported) {  // above the register-tiled range: LDS-tile transform, then the product as its own pass
            c.mul = nullptr;
            if ((e = launch_c2c_tile(c, dtype, s)) != hipSuccess) return e;
            e = launch_pointwise(a.scratch_b, a.bhat, a.scratch_b, (unsigned long long)count * a.M, a.M, 0, dtype, s);
        }
        if (e != hipSuccess) return e;
        c.mul = nullptr; c.mul_bcast = 0;
        c.in = a.scratch_b; c.out = a.scratch_a; c.inverse = 1;
        if ((e = launch_c2c_any(c, dtype, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_bs_post<T>, dim3((a.nb + 31u) / 32u, (count + 15u) / 16u), dim3(256), 0, s, (const C2<T> *)a.scratch_a,
                           (const C2<T> *)a.chirp, (T *)a.out, q0, count, a.M, a.n_fft, a.nb, a.n_frames, pairs, a.complex_out, a.amp, (T)a.eps);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace

hipError_t launch_bluestein(const BsArgs &a, int dtype, hipStream_t s) {
    return dtype == SGX_F64 ? run_t<double>(a, dtype, s) : run_t<float>(a, dtype, s);
}

}  // namespace sgx
