// kernels_fft2d.hip — shape-generic kernels of the 2-D FFT path (SURVEY.md §8 a17 / BASELINE config 5), f32 and f64.
//
// Reference algorithm (src/fft_backend.rs:653-691 forward, :744-818 inverse): row-column.  Forward = R2C along rows,
// then C2C along the nrows-long columns of the (nrows, ncols/2+1) half spectrum.  Inverse = inverse C2C along columns,
// force the DC / Nyquist columns real (:782-793), C2R along rows, scale by 1/(nrows*ncols) (:810-815).
//
// Here the row R2C is the STFT engine itself (n_fft = hop = ncols, rectangular window, no centring, complex output):
// its frame-contiguous output layout [k][r] is exactly the transposed intermediate the column pass wants.  This file
// holds the remaining pieces as LDS-tile kernels with transposing, coalesced loads/stores:
//   k_c2c_tile   `tile` sequences of length n per workgroup in LDS; radix-2 (power-of-two n) or direct DFT; forward or
//                inverse; arbitrary element strides on both sides, the thread mapping follows the unit stride
//   k_c2r_rows   per row: Hermitian extension of the half spectrum into LDS (with the DC/Nyquist fix), inverse complex
//                FFT, real part * scale -> contiguous image rows
//   k_pointwise  spectrum x spectrum (convolve_fft, src/image_ops.rs:109) or spectrum x real mask (:315)
#include "sgx_internal.h"

namespace sgx {

template <typename T>
struct Cx2 {
    T re, im;
};

template <typename T>
__device__ inline Cx2<T> cmul2(Cx2<T> a, Cx2<T> b) {
    return Cx2<T>{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}

// in-LDS radix-2 DIT over `nb` bit-reverse-ordered sequences of length n = 2^log2n, row stride fs; tw[k] = e^{-2 pi i k/n}
template <typename T>
__device__ inline void lds_fft_pow2(Cx2<T> *buf, unsigned nb, unsigned n, unsigned log2n, unsigned fs, const Cx2<T> *tw,
                                    bool inverse) {
    const unsigned half = n >> 1;
    for (unsigned h = 1, lh = 0; h < n; h <<= 1, lh++) {
        const unsigned twstep = n >> (lh + 1);
        for (unsigned idx = threadIdx.x; idx < nb * half; idx += blockDim.x) {
            const unsigned s = idx / half, q = idx - s * half;
            const unsigned j = q & (h - 1), blk = q >> lh;
            const unsigned p0 = s * fs + (blk << (lh + 1)) + j, p1 = p0 + h;
            Cx2<T> w = tw[j * twstep];
            if (inverse) w.im = -w.im;
            const Cx2<T> u = buf[p0], v = cmul2(buf[p1], w);
            buf[p0] = Cx2<T>{u.re + v.re, u.im + v.im};
            buf[p1] = Cx2<T>{u.re - v.re, u.im - v.im};
        }
        __syncthreads();
    }
}

// Lengths without a radix-2 path: n = n1 * n2 (n1 the largest divisor <= sqrt(n)) as a two-factor Cooley-Tukey transform, input index
// i = i1 n2 + i2, output index k = k1 + n1 k2:  X[k1 + n1 k2] = sum_i2 W_n2^(i2 k2) . W_n^(i2 k1) . sum_i1 x[i1 n2 + i2] W_n1^(i1 k1).
// Stage 1 (the inner sums times W_n^(i2 k1)) goes to a second LDS buffer, row k1 at k1 (n2 + 1) so that lanes over k1 spread over the
// banks; stage 2 is evaluated by whoever stores the output.  n (n1 + n2) multiply-adds per sequence instead of n^2 (n = 1000: 65 n
// against 1000 n — a 1000-row image took 29 ms per 134 images in the direct sum, 0.5 ms at 1024 rows); a prime n has n1 = 1 and
// stays with the direct sum.  tw[k] = e^{-2 pi i k / n}: W_n1^j = tw[j n2], W_n2^j = tw[j n1].
template <typename T>
__device__ inline void lds_two_factor_stage1(const Cx2<T> *src, Cx2<T> *dst, unsigned nb, unsigned n, unsigned n1, unsigned fs, unsigned fs2,
                                             const Cx2<T> *tw, bool inverse) {
    const unsigned n2 = n / n1;
    for (unsigned idx = threadIdx.x; idx < nb * n; idx += blockDim.x) {
        const unsigned s = idx / n, q = idx - s * n, k1 = q / n2, i2 = q - k1 * n2;
        const Cx2<T> *x = src + s * fs + i2;
        T sr = T(0), si = T(0);
        const unsigned step = k1 * n2;  // table index of W_n1^(i1 k1) advances by k1 n2 (mod n) per i1
        unsigned tt = 0;
        for (unsigned i1 = 0; i1 < n1; ++i1) {
            Cx2<T> w = tw[tt];
            if (inverse) w.im = -w.im;
            const Cx2<T> v = x[i1 * n2];
            sr += v.re * w.re - v.im * w.im;
            si += v.re * w.im + v.im * w.re;
            tt += step;
            if (tt >= n) tt -= n;
        }
        Cx2<T> w = tw[i2 * k1];  // i2 k1 < n
        if (inverse) w.im = -w.im;
        dst[s * fs2 + k1 * (n2 + 1) + i2] = cmul2(Cx2<T>{sr, si}, w);
    }
}
// output element o = k1 + n1 k2 of one sequence's stage-1 rows y
template <typename T>
__device__ inline Cx2<T> lds_two_factor_stage2(const Cx2<T> *y, unsigned o, unsigned n, unsigned n1, const Cx2<T> *tw, bool inverse) {
    const unsigned n2 = n / n1, k2 = o / n1, k1 = o - k2 * n1;
    const Cx2<T> *row = y + k1 * (n2 + 1);
    T sr = T(0), si = T(0);
    const unsigned step = k2 * n1;
    unsigned tt = 0;
    for (unsigned i2 = 0; i2 < n2; ++i2) {
        Cx2<T> w = tw[tt];
        if (inverse) w.im = -w.im;
        const Cx2<T> v = row[i2];
        sr += v.re * w.re - v.im * w.im;
        si += v.re * w.im + v.im * w.re;
        tt += step;
        if (tt >= n) tt -= n;
    }
    return Cx2<T>{sr, si};
}

template <typename T>
__global__ __launch_bounds__(256) void k_c2c_tile(C2cArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Cx2<T> *buf = (Cx2<T> *)smem;
    const unsigned n = a.n, fs = n + 1;
    const unsigned t = blockIdx.x % a.tiles, b = blockIdx.x / a.tiles;
    const unsigned s0 = t * a.tile;
    const unsigned ns = min(a.tile, a.nseq - s0);
    const Cx2<T> *in = (const Cx2<T> *)a.in + (size_t)b * a.in_img;
    Cx2<T> *out = (Cx2<T> *)a.out + (size_t)b * a.out_img;
    const Cx2<T> *tw = (const Cx2<T> *)a.tw;
    const bool pow2 = a.log2n > 0;
    for (unsigned idx = threadIdx.x; idx < ns * n; idx += 256) {
        unsigned s, i;
        if (a.in_seq_fast) { s = idx % ns; i = idx / ns; } else { i = idx % n; s = idx / n; }
        const Cx2<T> v = in[(size_t)(s0 + s) * a.in_ss + (size_t)i * a.in_is];
        const unsigned pos = pow2 ? (__brev(i) >> (32 - a.log2n)) : i;
        buf[s * fs + pos] = v;
    }
    __syncthreads();
    const T scale = (T)a.scale;
    if (pow2) {
        lds_fft_pow2<T>(buf, ns, n, a.log2n, fs, tw, a.inverse);
        for (unsigned idx = threadIdx.x; idx < ns * n; idx += 256) {
            unsigned s, o;
            if (a.out_seq_fast) { s = idx % ns; o = idx / ns; } else { o = idx % n; s = idx / n; }
            const Cx2<T> v = buf[s * fs + o];
            out[(size_t)(s0 + s) * a.out_ss + (size_t)o * a.out_is] = Cx2<T>{v.re * scale, v.im * scale};
        }
    } else if (a.n1 > 1) {  // two factors, through the second buffer
        Cx2<T> *buf2 = buf + (size_t)a.tile * fs;
        const unsigned fs2 = n + a.n1 + 1;
        lds_two_factor_stage1<T>(buf, buf2, ns, n, a.n1, fs, fs2, tw, a.inverse);
        __syncthreads();
        for (unsigned idx = threadIdx.x; idx < ns * n; idx += 256) {
            unsigned s, o;
            if (a.out_seq_fast) { s = idx % ns; o = idx / ns; } else { o = idx % n; s = idx / n; }
            const Cx2<T> v = lds_two_factor_stage2<T>(buf2 + (size_t)s * fs2, o, n, a.n1, tw, a.inverse);
            out[(size_t)(s0 + s) * a.out_ss + (size_t)o * a.out_is] = Cx2<T>{v.re * scale, v.im * scale};
        }
    } else {  // direct sum (n == 1 lands here too)
        for (unsigned idx = threadIdx.x; idx < ns * n; idx += 256) {
            unsigned s, o;
            if (a.out_seq_fast) { s = idx % ns; o = idx / ns; } else { o = idx % n; s = idx / n; }
            T sr = T(0), si = T(0);
            unsigned tt = 0;
            for (unsigned i = 0; i < n; ++i) {
                Cx2<T> w = tw[tt];
                if (a.inverse) w.im = -w.im;
                const Cx2<T> v = buf[s * fs + i];
                sr += v.re * w.re - v.im * w.im;
                si += v.re * w.im + v.im * w.re;
                tt += o;
                if (tt >= n) tt -= n;
            }
            out[(size_t)(s0 + s) * a.out_ss + (size_t)o * a.out_is] = Cx2<T>{sr * scale, si * scale};
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_c2r_rows(C2rArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Cx2<T> *buf = (Cx2<T> *)smem;
    const unsigned C = a.ncols, Cb = C / 2 + 1, fs = C + 1;
    const unsigned t = blockIdx.x % a.tiles, b = blockIdx.x / a.tiles;
    const unsigned r0 = t * a.tile;
    const unsigned nr = min(a.tile, a.nrows - r0);
    const Cx2<T> *in = (const Cx2<T> *)a.in + (size_t)b * a.in_img;
    T *out = (T *)a.out + (size_t)b * a.nrows * C;
    const Cx2<T> *tw = (const Cx2<T> *)a.tw;
    const bool pow2 = a.log2c > 0;
    // Hermitian extension X[C-k] = conj(X[k]); DC and (even C) Nyquist columns forced real (fft_backend.rs:782-793)
    for (unsigned idx = threadIdx.x; idx < nr * Cb; idx += 256) {
        unsigned r, k;
        if (a.k_fast) { k = idx % Cb; r = idx / Cb; } else { r = idx % nr; k = idx / nr; }  // follow the unit stride
        Cx2<T> v = in[(size_t)k * a.in_ks + (size_t)(r0 + r) * a.in_rs];
        if (k == 0 || (!(C & 1u) && k == Cb - 1)) {
            if (a.bad_flag && v.im != T(0)) atomicOr(a.bad_flag, 1u);
            v.im = T(0);
        }
        const unsigned p = pow2 ? (__brev(k) >> (32 - a.log2c)) : k;
        buf[r * fs + p] = v;
        const unsigned km = C - k;
        if (k != 0 && km != k) {
            const unsigned pm = pow2 ? (__brev(km) >> (32 - a.log2c)) : km;
            buf[r * fs + pm] = Cx2<T>{v.re, -v.im};
        }
    }
    __syncthreads();
    const T scale = (T)a.scale;
    const T *win = (const T *)a.win;
    if (pow2) {
        lds_fft_pow2<T>(buf, nr, C, a.log2c, fs, tw, true);
        for (unsigned idx = threadIdx.x; idx < nr * C; idx += 256) {
            const unsigned c = idx % C, r = idx / C;
            T v = buf[r * fs + c].re * scale;
            if (win) v *= win[c];
            out[(size_t)(r0 + r) * C + c] = v;
        }
    } else if (a.n1 > 1) {  // two factors, through the second buffer (inverse transform, real part)
        Cx2<T> *buf2 = buf + (size_t)a.tile * fs;
        const unsigned fs2 = C + a.n1 + 1;
        lds_two_factor_stage1<T>(buf, buf2, nr, C, a.n1, fs, fs2, tw, true);
        __syncthreads();
        for (unsigned idx = threadIdx.x; idx < nr * C; idx += 256) {
            const unsigned c = idx % C, r = idx / C;
            T v = lds_two_factor_stage2<T>(buf2 + (size_t)r * fs2, c, C, a.n1, tw, true).re * scale;
            if (win) v *= win[c];
            out[(size_t)(r0 + r) * C + c] = v;
        }
    } else {
        for (unsigned idx = threadIdx.x; idx < nr * C; idx += 256) {
            const unsigned c = idx % C, r = idx / C;
            T sr = T(0);
            unsigned tt = 0;
            for (unsigned k = 0; k < C; ++k) {
                const Cx2<T> w = tw[tt], v = buf[r * fs + k];  // e^{+i..} = conj(tw): re part = v.re*w.re + v.im*w.im
                sr += v.re * w.re + v.im * w.im;
                tt += c;
                if (tt >= C) tt -= C;
            }
            T v = sr * scale;
            if (win) v *= win[c];
            out[(size_t)(r0 + r) * C + c] = v;
        }
    }
}

// istft overlap-add (src/spectrogram.rs:4911-4930) as a gather: sample `pos` of the padded signal sums, in ascending frame
// order, the windowed frame values covering it; norm = sum of w*w (each product rounded, then added — no contraction);
// divide where norm > T(1e-10).  One thread per output sample; out[b][i] = padded[start + i].
template <typename T>
__global__ __launch_bounds__(256) void k_istft_ola(const T *frames, const T *win, T *out, unsigned n, unsigned hop,
                                                   unsigned n_frames, unsigned long long start, unsigned long long out_len) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= out_len) return;
    const unsigned b = blockIdx.y;
    const unsigned long long pos = start + i;
    const unsigned long long f_hi = min(pos / hop, (unsigned long long)n_frames - 1);
    const unsigned long long f_lo = pos >= n ? (pos - n) / hop + 1 : 0;
    const T *fr = frames + (size_t)b * n_frames * n;
    T acc = T(0), nrm = T(0);
    for (unsigned long long f = f_lo; f <= f_hi; ++f) {
        const unsigned j = (unsigned)(pos - f * hop);
        acc += fr[f * n + j];
        const T w = win[j];
        if constexpr (sizeof(T) == 4) nrm = __fadd_rn(nrm, __fmul_rn(w, w));
        else nrm = __dadd_rn(nrm, __dmul_rn(w, w));
    }
    if (nrm > T(1e-10)) acc /= nrm;
    out[(size_t)b * out_len + i] = acc;
}

// mode 0: out = a * b[i % per] (complex x complex); mode 1: out = a * m[i % per] (complex x real mask)
template <typename T>
__global__ __launch_bounds__(256) void k_pointwise(const Cx2<T> *x, const void *y, Cx2<T> *out, unsigned long long n,
                                                   unsigned long long per, int mode) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) {
        const Cx2<T> v = x[i];
        if (mode == 0) {
            out[i] = cmul2(v, ((const Cx2<T> *)y)[i % per]);
        } else {
            const T m = ((const T *)y)[i % per];
            out[i] = Cx2<T>{v.re * m, v.im * m};
        }
    }
}

static size_t esz(int dtype) { return dtype == SGX_F64 ? 8 : 4; }

unsigned fft2d_tile_for(unsigned n, int dtype) {
    const size_t per = (size_t)(n + 1) * 2 * esz(dtype);
    // 16 sequences within 64 KiB where they fit; a single long sequence may take the large LDS window (f64 4096: 65 552 B — one
    // element over 64 KiB made plan creation fail although the register-tiled kernels run that length)
    unsigned tile = 16;
    while (tile > 1 && per * tile > 64 * 1024) tile >>= 1;
    return per * tile <= 144 * 1024 ? tile : 0;
}

// largest divisor of n that is <= sqrt(n) (1 for a prime)
static unsigned two_factor_n1(unsigned n) {
    unsigned best = 1;
    for (unsigned d = 2; (unsigned long long)d * d <= n; ++d)
        if (n % d == 0) best = d;
    return best;
}
// sequences per workgroup for the two-factor path (the tile and the stage-1 buffer), 0 if not even one fits
static unsigned two_factor_tile(unsigned tile, unsigned n, unsigned n1, size_t es) {
    const size_t per = ((size_t)(n + 1) + (n + n1 + 1)) * 2 * es;
    while (tile > 1 && per * tile > 144 * 1024) tile >>= 1;
    return per * tile <= 144 * 1024 ? tile : 0;
}

hipError_t launch_c2c_tile(const C2cArgs &a0, int dtype, hipStream_t s) {
    C2cArgs a = a0;
    a.n1 = 0;
    size_t extra = 0;
    if (a.log2n == 0 && a.n >= 4 && a.tile) {
        const unsigned n1 = two_factor_n1(a.n), t2 = n1 > 1 ? two_factor_tile(a.tile, a.n, n1, esz(dtype)) : 0;
        if (t2) {
            a.n1 = n1;
            a.tile = t2;
            a.tiles = (a.nseq + t2 - 1) / t2;
            extra = (size_t)t2 * (a.n + n1 + 1) * 2 * esz(dtype);
        }
    }
    const unsigned long long g = (unsigned long long)a.tiles * a.batch;
    if (g == 0 || g >= 0x7fffffffull || a.tile == 0) return hipErrorInvalidConfiguration;
    const size_t lds = (size_t)a.tile * (a.n + 1) * 2 * esz(dtype) + extra;
    if (lds > 64 * 1024) {
        hipError_t e = set_max_dynamic_lds(dtype == SGX_F64 ? (const void *)k_c2c_tile<double> : (const void *)k_c2c_tile<float>, 144 * 1024);  // (set once)
        if (e != hipSuccess) return e;
    }
    if (dtype == SGX_F64) hipLaunchKernelGGL(k_c2c_tile<double>, dim3((unsigned)g), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(k_c2c_tile<float>, dim3((unsigned)g), dim3(256), lds, s, a);
    return hipGetLastError();
}

unsigned c2r_tile_for(unsigned n, int dtype, size_t lds_budget) {
    const size_t per = (size_t)(n + 1) * 2 * esz(dtype);
    unsigned tile = 16;
    while (tile > 1 && per * tile > lds_budget) tile >>= 1;
    return per * tile <= lds_budget ? tile : 0;
}

hipError_t launch_istft_ola(const void *frames, const void *win, void *out, unsigned n, unsigned hop, unsigned n_frames,
                            unsigned long long start, unsigned long long out_len, unsigned batch, int dtype, hipStream_t s) {
    const unsigned long long gx = (out_len + 255) / 256;
    if (gx == 0 || gx >= 0x7fffffffull || batch == 0 || batch > 65535u) return hipErrorInvalidConfiguration;
    const dim3 grid((unsigned)gx, batch);
    if (dtype == SGX_F64)
        hipLaunchKernelGGL(k_istft_ola<double>, grid, dim3(256), 0, s, (const double *)frames, (const double *)win, (double *)out, n, hop, n_frames, start, out_len);
    else
        hipLaunchKernelGGL(k_istft_ola<float>, grid, dim3(256), 0, s, (const float *)frames, (const float *)win, (float *)out, n, hop, n_frames, start, out_len);
    return hipGetLastError();
}

hipError_t launch_c2r_rows(const C2rArgs &a0, int dtype, hipStream_t s) {
    C2rArgs a = a0;
    a.n1 = 0;
    size_t extra = 0;
    if (a.log2c == 0 && a.ncols >= 4 && a.tile) {
        const unsigned n1 = two_factor_n1(a.ncols), t2 = n1 > 1 ? two_factor_tile(a.tile, a.ncols, n1, esz(dtype)) : 0;
        if (t2) {
            a.n1 = n1;
            a.tile = t2;
            a.tiles = (a.nrows + t2 - 1) / t2;
            extra = (size_t)t2 * (a.ncols + n1 + 1) * 2 * esz(dtype);
        }
    }
    const unsigned long long g = (unsigned long long)a.tiles * a.batch;
    if (g == 0 || g >= 0x7fffffffull || a.tile == 0) return hipErrorInvalidConfiguration;
    const size_t lds = (size_t)a.tile * (a.ncols + 1) * 2 * esz(dtype) + extra;
    if (lds > 64 * 1024) {  // opt in to the large LDS window (160 KiB per CU on gfx950)
        hipError_t e = dtype == SGX_F64 ? hipFuncSetAttribute((const void *)k_c2r_rows<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                        : hipFuncSetAttribute((const void *)k_c2r_rows<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (dtype == SGX_F64) hipLaunchKernelGGL(k_c2r_rows<double>, dim3((unsigned)g), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(k_c2r_rows<float>, dim3((unsigned)g), dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_pointwise(const void *x, const void *y, void *out, unsigned long long n, unsigned long long per, int mode,
                            int dtype, hipStream_t s) {
    unsigned long long blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (blocks == 0) return hipErrorInvalidConfiguration;
    if (dtype == SGX_F64)
        hipLaunchKernelGGL(k_pointwise<double>, dim3((unsigned)blocks), dim3(256), 0, s, (const Cx2<double> *)x, y, (Cx2<double> *)out, n, per, mode);
    else
        hipLaunchKernelGGL(k_pointwise<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const Cx2<float> *)x, y, (Cx2<float> *)out, n, per, mode);
    return hipGetLastError();
}

}  // namespace sgx
