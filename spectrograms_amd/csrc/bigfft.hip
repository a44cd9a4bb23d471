// bigfft.hip — frame lengths beyond the on-chip kernels: transforms through global memory, O(n log n) for EVERY length.
//
// The reference plans any length through rustfft (src/fft_backend.rs:372-389) and its one-shot helpers are routinely called with
// n_fft = the whole signal (src/spectrogram.rs:4490-4643: `fft`, `rfft`, `power_spectrum`).  Up to round 4 this library ran odd lengths
// above 8192 (f64: 4096) as O(n^2) sums and refused n_fft > 32768 (f64: 16384).  Here:
//
//   * powers of two M = M1 * M2 (M1 >= M2 <= 2 M2): the four-step transform.  With n = n1 M2 + n2 and k = k1 + M1 k2,
//       X[k1 + M1 k2] = sum_n2 W_M2^(n2 k2) [ W_M^(n2 k1) sum_n1 a[n1 M2 + n2] W_M1^(n1 k1) ]
//     pass A: for every column n2 a length-M1 transform over n1 (tile = C neighbouring columns x all rows, resident in LDS, radix-2
//     stages in place, two at a time), times W_M^(n2 k1), in place;  pass B: for every row k1 a length-M2 transform over n2 (tile = C neighbouring rows), written
//     in natural order (k1 fastest across the tile's rows: runs of C elements) or left in place as [k1][k2].
//   * every other length n: chirp-z on top of it, X[k] = conj(c_k) sum_j (x_j conj(c_j)) c_(k-j), c_j = e^(i pi j^2 / n), as a circular
//     convolution of length M = 2^ceil(log2(2n-1)): pass A, then ONE row kernel that transforms a row forward, multiplies it by the
//     transformed chirp (stored in that [k1][k2] order, 1/M folded in), transforms it back and applies conj(W_M^(n2 k1)), then the
//     inverse column pass — the spectrum never needs reordering.
//   * real frames ride two to a complex sequence (frames 2p and 2p+1 of one signal: a signal's bits do not depend on its batch), built
//     inside the first column pass straight from the signals (big_frame_elem), and come apart by Hermitian symmetry in the epilogue,
//     which also applies |.|^2 / sqrt / dB and writes the reference's [bins][frames] layout (S9).  Filterbank outputs take the split
//     path (per-bin power, then k_bank_rows).
//   * the inverse (irfft / istft rows) is the same engine behind conj: idft(Z) = conj(dft(conj Z)).
//
// Everything sits in plan-owned scratch; a call is cut into chunks of sequences so that the scratch stays bounded (kBigChunkBytes).
// Bound: the LDS stages next to HBM (each pass reads and writes the sequences once; two to four workgroups per CU overlap the two); this
// is the totality path, not a tuned one — see DESIGN.md §3.6 for its measured rates.
#include <algorithm>
#include <cmath>
#include <type_traits>

#include "db_f64.h"
#include "sgx_internal.h"

namespace sgx {
namespace {

template <typename T> struct C2 { using type = float2; };
template <> struct C2<double> { using type = double2; };
template <typename T> using c2_t = typename C2<T>::type;

template <typename T> __device__ __forceinline__ c2_t<T> mk(T x, T y) { c2_t<T> r; r.x = x; r.y = y; return r; }
template <typename V> __device__ __forceinline__ V cmul(V a, V b) { V r; r.x = a.x * b.x - a.y * b.y; r.y = a.x * b.y + a.y * b.x; return r; }
template <typename V> __device__ __forceinline__ V cmulc(V a, V b) { V r; r.x = a.x * b.x + a.y * b.y; r.y = a.y * b.x - a.x * b.y; return r; }  // a conj(b)
template <typename V> __device__ __forceinline__ V cadd(V a, V b) { V r; r.x = a.x + b.x; r.y = a.y + b.y; return r; }
template <typename V> __device__ __forceinline__ V csub(V a, V b) { V r; r.x = a.x - b.x; r.y = a.y - b.y; return r; }

constexpr unsigned kBigThreads = 512;

// ---- transforms of C sequences of L points resident in LDS: lds[c * (L + 1) + i] ---------------------------------------------------
// DIT: input in bit-reversed places, output in natural order.  DIF: natural in, bit-reversed out.  `inv`: e^(+) twiddles.  Radix-2
// stages taken two at a time (lds_r4_stage), a single one where log2 L is odd.  Work item b of a stage = (lane c = b % C, index b / C):
// neighbouring threads work on different sequences, whose rows start L + 1 elements apart (distinct banks).  tw[j] = W_L^j, j < L / 2, in LDS.
template <typename T>
__device__ __forceinline__ void lds_r2_stage(c2_t<T> *lds, const c2_t<T> *tw, unsigned L, unsigned lL, unsigned C, unsigned s, bool inv, bool dif) {
    using V = c2_t<T>;
    const unsigned total = C * (L >> 1), half = 1u << s;
    for (unsigned b = threadIdx.x; b < total; b += kBigThreads) {
        const unsigned c = b % C, q = b / C;
        const unsigned j = q & (half - 1u);
        const unsigned i0 = ((q >> s) << (s + 1u)) + j, i1 = i0 + half;
        V w = tw[j << (lL - 1u - s)];
        if (inv) w.y = -w.y;
        V *row = lds + c * (L + 1u);
        const V u = row[i0], x1 = row[i1];
        if (dif) {
            row[i0] = cadd(u, x1);
            row[i1] = cmul(csub(u, x1), w);
        } else {
            const V v = cmul(x1, w);
            row[i0] = cadd(u, v);
            row[i1] = csub(u, v);
        }
    }
    __syncthreads();
}
// two radix-2 stages (halves h = 2^s and 2 h) in one pass over the tile: 4 reads, 4 writes and one barrier instead of 8, 8 and two
template <typename T>
__device__ __forceinline__ void lds_r4_stage(c2_t<T> *lds, const c2_t<T> *tw, unsigned L, unsigned lL, unsigned C, unsigned s, bool inv, bool dif) {
    using V = c2_t<T>;
    const unsigned total = C * (L >> 2), h = 1u << s;
    for (unsigned g = threadIdx.x; g < total; g += kBigThreads) {
        const unsigned c = g % C, q = g / C;
        const unsigned j = q & (h - 1u);
        const unsigned i0 = ((q >> s) << (s + 2u)) + j;
        V w1 = tw[j << (lL - 1u - s)], w2 = tw[j << (lL - 2u - s)], w3 = tw[(j + h) << (lL - 2u - s)];  // W_2h^j, W_4h^j, W_4h^(j+h)
        if (inv) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
        V *row = lds + c * (L + 1u) + i0;
        const V x0 = row[0], x1 = row[h], x2 = row[2u * h], x3 = row[3u * h];
        if (dif) {  // half 2 h first, then h
            const V b0 = cadd(x0, x2), b2 = cmul(csub(x0, x2), w2), b1 = cadd(x1, x3), b3 = cmul(csub(x1, x3), w3);
            row[0] = cadd(b0, b1);
            row[h] = cmul(csub(b0, b1), w1);
            row[2u * h] = cadd(b2, b3);
            row[3u * h] = cmul(csub(b2, b3), w1);
        } else {    // half h first, then 2 h
            const V t1 = cmul(x1, w1), t3 = cmul(x3, w1);
            const V a0 = cadd(x0, t1), a1 = csub(x0, t1), a2 = cadd(x2, t3), a3 = csub(x2, t3);
            const V u2 = cmul(a2, w2), u3 = cmul(a3, w3);
            row[0] = cadd(a0, u2);
            row[2u * h] = csub(a0, u2);
            row[h] = cadd(a1, u3);
            row[3u * h] = csub(a1, u3);
        }
    }
    __syncthreads();
}
template <typename T>
__device__ __forceinline__ void lds_fft_dit(c2_t<T> *lds, const c2_t<T> *tw, unsigned L, unsigned lL, unsigned C, bool inv) {
    unsigned s = 0;
    for (; s + 1u < lL; s += 2u) lds_r4_stage<T>(lds, tw, L, lL, C, s, inv, false);
    if (s < lL) lds_r2_stage<T>(lds, tw, L, lL, C, s, inv, false);
}
template <typename T>
__device__ __forceinline__ void lds_fft_dif(c2_t<T> *lds, const c2_t<T> *tw, unsigned L, unsigned lL, unsigned C, bool inv) {
    unsigned r = lL;
    if (r & 1u) { --r; lds_r2_stage<T>(lds, tw, L, lL, C, r, inv, true); }
    for (; r >= 2u; r -= 2u) lds_r4_stage<T>(lds, tw, L, lL, C, r - 2u, inv, true);
}
__device__ __forceinline__ unsigned brev(unsigned i, unsigned bits) { return bits ? __brev(i) >> (32u - bits) : 0u; }

// W_M^p from the two-level table: p = 1024 hi + lo
template <typename T>
__device__ __forceinline__ c2_t<T> tw_big(const c2_t<T> *thi, const c2_t<T> *tlo, unsigned p) {
    return cmul(thi[p >> 10], tlo[p & 1023u]);
}

struct BigPass {
    void *buf;          // [nseq][M] complex T
    void *out;          // pass B, natural-order output [nseq][M] (null: in place, [k1][k2])
    unsigned M, M1, M2, l1, l2;
    unsigned C;         // sequences of the pass per tile
    unsigned tiles;     // tiles per big sequence
    unsigned nseq;
    const void *wl;     // W_M1^j, j < M1 / 2
    const void *thi, *tlo;
    const void *bhat;   // rows kernel of the chirp-z chain: FFT_M(chirp) / M in [k1][k2] order
    int inv;            // column pass: inverse transform (the chain's last pass; no twiddle — the rows kernel applied it)
    // FRAMES (forward STFT): pass A builds its elements from the signals instead of reading `buf` — two windowed frames per sequence
    // (big_frame_elem), so the sequences are written once (by this pass) and the zero half of a chirp-z sequence is never read
    const void *x, *win, *chirp;
    unsigned long long sample_stride, n_samples, q0;
    unsigned n, hop, pad, n_frames, PP;
};

// element m of sequence q (= pair p of signal b): w[m] (xa[m] + i xb[m]), frames 2 p and 2 p + 1; out-of-range samples are the reference's zero
// centre padding (src/spectrogram.rs:1301-1320); chirp-z: times conj(c_m); zero from n on
template <typename T>
__device__ __forceinline__ c2_t<T> big_frame_elem(const BigPass &a, unsigned long long q, unsigned m) {
    using V = c2_t<T>;
    if (m >= a.n) return mk<T>(T(0), T(0));
    const unsigned long long b = q / a.PP;
    const unsigned p = unsigned(q - b * a.PP);
    const T *row = (const T *)a.x + b * a.sample_stride;
    const long long sa = (long long)(2ull * p) * a.hop - (long long)a.pad + m, sb = sa + a.hop;
    const T w = ((const T *)a.win)[m];
    const T xa = (sa >= 0 && (unsigned long long)sa < a.n_samples) ? row[sa] * w : T(0);
    const T xb = (2u * p + 1u < a.n_frames && sb >= 0 && (unsigned long long)sb < a.n_samples) ? row[sb] * w : T(0);
    V v = mk<T>(xa, xb);
    if (a.chirp) v = cmulc(v, ((const V *)a.chirp)[m]);
    return v;
}

// pass A / A': columns.  Tile = columns [n2_0, n2_0 + C) of sequence q; lds[c][n1].
template <typename T, bool FRAMES = false>
__global__ __launch_bounds__(kBigThreads) void k_big_cols(BigPass a) {
    using V = c2_t<T>;
    extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
    V *lds = (V *)big_smem;
    const unsigned L = a.M1, lL = a.l1, C = a.C;
    V *tw = lds + C * (L + 1u);
    for (unsigned j = threadIdx.x; j < (L >> 1); j += kBigThreads) tw[j] = ((const V *)a.wl)[j];
    const unsigned q = blockIdx.x / a.tiles, n20 = (blockIdx.x - q * a.tiles) * C;
    V *g = (V *)a.buf + (size_t)q * a.M + n20;
    for (unsigned e = threadIdx.x; e < C * L; e += kBigThreads) {
        const unsigned c = e % C, n1 = e / C;
        V v;
        if constexpr (FRAMES) v = big_frame_elem<T>(a, a.q0 + q, n1 * a.M2 + n20 + c);
        else v = g[(size_t)n1 * a.M2 + c];
        lds[c * (L + 1u) + (a.inv ? n1 : brev(n1, lL))] = v;
    }
    __syncthreads();
    if (a.inv) lds_fft_dif<T>(lds, tw, L, lL, C, true);
    else lds_fft_dit<T>(lds, tw, L, lL, C, false);
    for (unsigned e = threadIdx.x; e < C * L; e += kBigThreads) {
        const unsigned c = e % C, k1 = e / C;
        V v = lds[c * (L + 1u) + (a.inv ? brev(k1, lL) : k1)];
        if (!a.inv) v = cmul(v, tw_big<T>((const V *)a.thi, (const V *)a.tlo, k1 * (n20 + c)));  // W_M^(n2 k1)
        g[(size_t)k1 * a.M2 + c] = v;
    }
}

// pass B: rows.  Tile = rows [k1_0, k1_0 + C) of sequence q; lds[c][n2].
//   CHAIN = false: forward transform, output in natural order (out[k1 + M1 k2]) or in place
//   CHAIN = true : forward, times bhat[k1][k2], inverse, times conj(W_M^(n2 k1)), in place (the chirp-z convolution's middle)
template <typename T, bool CHAIN>
__global__ __launch_bounds__(kBigThreads) void k_big_rows(BigPass a) {
    using V = c2_t<T>;
    extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
    V *lds = (V *)big_smem;
    const unsigned L = a.M2, lL = a.l2, C = a.C;
    V *tw = lds + C * (L + 1u);
    const unsigned step = a.M1 / a.M2;  // W_M2^j = W_M1^(j step)
    for (unsigned j = threadIdx.x; j < (L >> 1); j += kBigThreads) tw[j] = ((const V *)a.wl)[j * step];
    const unsigned q = blockIdx.x / a.tiles, k10 = (blockIdx.x - q * a.tiles) * C;
    V *g = (V *)a.buf + (size_t)q * a.M + (size_t)k10 * a.M2;
    for (unsigned e = threadIdx.x; e < C * L; e += kBigThreads) {
        const unsigned n2 = e % L, c = e / L;
        lds[c * (L + 1u) + brev(n2, lL)] = g[(size_t)c * a.M2 + n2];
    }
    __syncthreads();
    lds_fft_dit<T>(lds, tw, L, lL, C, false);
    if constexpr (CHAIN) {
        const V *bh = (const V *)a.bhat + (size_t)k10 * a.M2;
        for (unsigned e = threadIdx.x; e < C * L; e += kBigThreads) {
            const unsigned k2 = e % L, c = e / L;
            V *p = lds + c * (L + 1u) + k2;
            *p = cmul(*p, bh[(size_t)c * a.M2 + k2]);
        }
        __syncthreads();
        lds_fft_dif<T>(lds, tw, L, lL, C, true);
        for (unsigned e = threadIdx.x; e < C * L; e += kBigThreads) {
            const unsigned n2 = e % L, c = e / L;
            const V v = lds[c * (L + 1u) + brev(n2, lL)];
            g[(size_t)c * a.M2 + n2] = cmulc(v, tw_big<T>((const V *)a.thi, (const V *)a.tlo, n2 * (k10 + c)));
        }
    } else if (a.out) {
        V *o = (V *)a.out + (size_t)q * a.M + k10;
        for (unsigned e = threadIdx.x; e < C * L; e += kBigThreads) {
            const unsigned c = e % C, k2 = e / C;
            o[(size_t)k2 * a.M1 + c] = lds[c * (L + 1u) + k2];
        }
    } else {
        for (unsigned e = threadIdx.x; e < C * L; e += kBigThreads) {
            const unsigned k2 = e % L, c = e / L;
            g[(size_t)c * a.M2 + k2] = lds[c * (L + 1u) + k2];
        }
    }
}

// ---- epilogue: split the two frames, amplitude scaling, the reference's [bins][frames] layout -----------------------------------
template <typename T>
__device__ __forceinline__ T big_amp(T p, int amp, T eps) {
    if (amp == AMP_MAGNITUDE) return sqrt(p);
    if (amp == AMP_DB) {
        const T v = p > eps ? p : eps;
        if constexpr (sizeof(T) == 8) return db_f64(v);
        else return T(10) * log10(v);
    }
    return p;
}
template <typename T>
__global__ __launch_bounds__(256) void k_big_split(const c2_t<T> *buf, const c2_t<T> *chirp, T *out, unsigned n, unsigned M, unsigned nb,
                                                   unsigned n_frames, unsigned PP, unsigned long long q0, int complex_out, int amp, double eps_d) {
    using V = c2_t<T>;
    const unsigned long long q = q0 + blockIdx.y;
    const unsigned long long b = q / PP;
    const unsigned p = unsigned(q - b * PP);
    const unsigned k = blockIdx.x * 256u + threadIdx.x;
    if (k >= nb) return;
    const V *z = buf + (size_t)blockIdx.y * M;
    const unsigned km = k == 0 ? 0u : n - k;
    V P = z[k], Q = z[km];
    if (chirp) { P = cmulc(P, chirp[k]); Q = cmulc(Q, chirp[km]); }
    // A = (P + conj Q) / 2, B = -i (P - conj Q) / 2
    V A = mk<T>(T(0.5) * (P.x + Q.x), T(0.5) * (P.y - Q.y));
    V B = mk<T>(T(0.5) * (P.y + Q.y), T(0.5) * (Q.x - P.x));
    if (k == 0 || 2u * k == n) { A.y = T(0); B.y = T(0); }  // realfft returns exactly-real DC / Nyquist bins
    const unsigned fa = 2u * p, fb = fa + 1u;
    const size_t base = ((size_t)b * nb + k) * n_frames;
    const T eps = T(eps_d);
    if (complex_out) {
        V *o = (V *)out;
        o[base + fa] = A;
        if (fb < n_frames) o[base + fb] = B;
    } else {
        out[base + fa] = big_amp<T>(A.x * A.x + A.y * A.y, amp, eps);
        if (fb < n_frames) out[base + fb] = big_amp<T>(B.x * B.x + B.y * B.y, amp, eps);
    }
}

// ---- inverse: Hermitian rows of two frames -> conj(Z) (the conj-dft-conj identity), and back to real frames ---------------------
// spec element (frame f, bin k) of signal b at spec[b * img + k * ks + f * fs] (complex).  The imaginary parts of the DC and (even n)
// Nyquist bins are dropped and reported, as realfft's C2R does (src/fft_backend.rs:555-557).
template <typename T>
__global__ __launch_bounds__(256) void k_big_herm(const c2_t<T> *spec, unsigned long long img, unsigned long long ks, unsigned long long fs,
                                                  const c2_t<T> *chirp, c2_t<T> *buf, unsigned n, unsigned M, unsigned nb, unsigned n_frames,
                                                  unsigned PP, unsigned long long q0, unsigned *bad_flag) {
    using V = c2_t<T>;
    const unsigned long long q = q0 + blockIdx.y;
    const unsigned long long b = q / PP;
    const unsigned p = unsigned(q - b * PP);
    const unsigned m = blockIdx.x * 256u + threadIdx.x;
    if (m >= M) return;
    V v = mk<T>(T(0), T(0));
    if (m < n) {
        const unsigned k = m < nb ? m : n - m;  // Z[m] for m > n/2 is the mirror of bin n - m
        const bool mirror = m >= nb;
        const unsigned fa = 2u * p, fb = fa + 1u;
        const V *s = spec + b * img + (size_t)k * ks;
        V A = s[(size_t)fa * fs], B = fb < n_frames ? s[(size_t)fb * fs] : mk<T>(T(0), T(0));
        if (k == 0 || 2u * k == n) {
            if ((A.y != T(0) || B.y != T(0)) && bad_flag) atomicOr(bad_flag, 1u);
            A.y = T(0); B.y = T(0);
        }
        if (mirror) { A.y = -A.y; B.y = -B.y; }
        // Z = A + i B; written conjugated: conj(Z) = (A.x - B.y) - i (A.y + B.x)
        v = mk<T>(A.x - B.y, -(A.y + B.x));
        if (chirp) v = cmulc(v, chirp[m]);
    }
    buf[(size_t)blockIdx.y * M + m] = v;
}
// R = dft(conj Z) (times conj(c) for chirp-z); z = conj(R): frame 2p = Re R, frame 2p+1 = -Im R; times scale, times the window if given.
// frames: [batch][n_frames][n]
template <typename T>
__global__ __launch_bounds__(256) void k_big_unframe(const c2_t<T> *buf, const c2_t<T> *chirp, const T *win, T *frames, unsigned n, unsigned M,
                                                     unsigned n_frames, unsigned PP, unsigned long long q0, double scale_d) {
    using V = c2_t<T>;
    const unsigned long long q = q0 + blockIdx.y;
    const unsigned long long b = q / PP;
    const unsigned p = unsigned(q - b * PP);
    const unsigned m = blockIdx.x * 256u + threadIdx.x;
    if (m >= n) return;
    V R = buf[(size_t)blockIdx.y * M + m];
    if (chirp) R = cmulc(R, chirp[m]);
    const T scale = T(scale_d);
    T va = R.x * scale, vb = -R.y * scale;
    if (win) { va *= win[m]; vb *= win[m]; }
    const unsigned fa = 2u * p, fb = fa + 1u;
    T *o = frames + ((size_t)b * n_frames + fa) * n + m;
    o[0] = va;
    if (fb < n_frames) o[n] = vb;
}

// complex sequences in / out of the scratch (the C2cPlan of long lengths: sgx_c2c_*): in[q * in_ss + m * in_is]; inverse: conj in, conj out
template <typename T>
__global__ __launch_bounds__(256) void k_big_cin(const c2_t<T> *in, unsigned long long in_ss, unsigned long long in_is, const c2_t<T> *chirp,
                                                 c2_t<T> *buf, unsigned n, unsigned M, unsigned long long q0, int inverse) {
    using V = c2_t<T>;
    const unsigned m = blockIdx.x * 256u + threadIdx.x;
    if (m >= M) return;
    V v = mk<T>(T(0), T(0));
    if (m < n) {
        v = in[(q0 + blockIdx.y) * in_ss + (size_t)m * in_is];
        if (inverse) v.y = -v.y;
        if (chirp) v = cmulc(v, chirp[m]);
    }
    buf[(size_t)blockIdx.y * M + m] = v;
}
template <typename T>
__global__ __launch_bounds__(256) void k_big_cout(const c2_t<T> *buf, const c2_t<T> *chirp, c2_t<T> *out, unsigned long long out_ss,
                                                  unsigned long long out_is, unsigned n, unsigned M, unsigned long long q0, int inverse, double scale_d) {
    using V = c2_t<T>;
    const unsigned m = blockIdx.x * 256u + threadIdx.x;
    if (m >= n) return;
    V v = buf[(size_t)blockIdx.y * M + m];
    if (chirp) v = cmulc(v, chirp[m]);
    if (inverse) v.y = -v.y;
    const T s = T(scale_d);
    v.x *= s; v.y *= s;
    out[(q0 + blockIdx.y) * out_ss + (size_t)m * out_is] = v;
}

size_t esz_of(int dtype) { return dtype == SGX_F64 ? 8 : 4; }

// sequences per tile: the smallest of 36 / 72 / 144 KiB (four / two / one workgroup per CU: the loads and stores of one under the stages
// of the others — measured: profiles/bench_r05_bigfft.txt) whose tile (C rows of L + 1 complex elements + the pass's twiddles) still
// gives the column pass row segments of SGX_BIG_MINSEG bytes
#ifndef SGX_BIG_MINSEG
#define SGX_BIG_MINSEG 64
#endif
unsigned tile_lanes(unsigned L, unsigned lanes, int dtype) {
    const size_t cb = 2 * esz_of(dtype);
    auto fit = [&](size_t budget) {
        unsigned C = 1;
        while (2u * C <= lanes && size_t(2u * C) * (L + 1u) * cb + size_t(L / 2) * cb <= budget && 2u * C <= 64u) C *= 2u;
        return C;
    };
    const unsigned seg = unsigned(SGX_BIG_MINSEG / cb);
    for (size_t budget : {size_t(36) << 10, size_t(72) << 10}) {
        const unsigned C = fit(budget);
        if (C >= seg || C >= lanes) return C;
    }
    return fit(size_t(144) << 10);
}

template <typename T>
hipError_t run_chain(const BigDev &t, void *buf, void *nat, unsigned nseq, hipStream_t s, const BigPass *frames = nullptr) {
    // forward transform of `nseq` sequences of length t.n sitting at stride t.M in `buf` (chirp-z: already multiplied by conj(c), zero
    // padded).  Result: powers of two — natural order in `nat`; chirp-z — in `buf`, still to be multiplied by conj(c_k).
    BigPass a{};
    a.buf = buf; a.out = nullptr;
    a.M = t.M; a.M1 = t.M1; a.M2 = t.M2; a.l1 = t.l1; a.l2 = t.l2;
    a.nseq = nseq; a.wl = t.wl; a.thi = t.thi; a.tlo = t.tlo; a.bhat = t.bhat; a.inv = 0;
    const size_t cb = 2 * sizeof(T);
    auto lds_of = [&](unsigned C, unsigned L) { return size_t(C) * (L + 1u) * cb + size_t(L / 2) * cb; };
    hipError_t e;
    // pass A
    a.C = tile_lanes(t.M1, t.M2, sizeof(T) == 8 ? SGX_F64 : SGX_F32);
    a.tiles = t.M2 / a.C;
    const unsigned ca = a.C, ta = a.tiles;
    if ((e = set_max_dynamic_lds((const void *)k_big_cols<T>, 160 * 1024)) != hipSuccess) return e;
    if (frames) {  // pass A reads the signals themselves
        a.x = frames->x; a.win = frames->win; a.chirp = frames->chirp; a.sample_stride = frames->sample_stride; a.n_samples = frames->n_samples;
        a.q0 = frames->q0; a.n = frames->n; a.hop = frames->hop; a.pad = frames->pad; a.n_frames = frames->n_frames; a.PP = frames->PP;
        if ((e = set_max_dynamic_lds((const void *)k_big_cols<T, true>, 160 * 1024)) != hipSuccess) return e;
        hipLaunchKernelGGL((k_big_cols<T, true>), dim3(nseq * a.tiles), dim3(kBigThreads), lds_of(a.C, t.M1), s, a);
    } else {
        hipLaunchKernelGGL(k_big_cols<T>, dim3(nseq * a.tiles), dim3(kBigThreads), lds_of(a.C, t.M1), s, a);
    }
    // pass B
    a.C = tile_lanes(t.M2, t.M1, sizeof(T) == 8 ? SGX_F64 : SGX_F32);
    a.tiles = t.M1 / a.C;
    if (t.chirp) {
        if ((e = set_max_dynamic_lds((const void *)k_big_rows<T, true>, 160 * 1024)) != hipSuccess) return e;
        hipLaunchKernelGGL((k_big_rows<T, true>), dim3(nseq * a.tiles), dim3(kBigThreads), lds_of(a.C, t.M2), s, a);
        a.C = ca; a.tiles = ta; a.inv = 1;
        hipLaunchKernelGGL(k_big_cols<T>, dim3(nseq * a.tiles), dim3(kBigThreads), lds_of(a.C, t.M1), s, a);
    } else {
        a.out = nat;
        if ((e = set_max_dynamic_lds((const void *)k_big_rows<T, false>, 160 * 1024)) != hipSuccess) return e;
        hipLaunchKernelGGL((k_big_rows<T, false>), dim3(nseq * a.tiles), dim3(kBigThreads), lds_of(a.C, t.M2), s, a);
    }
    return hipGetLastError();
}

}  // namespace

// ---- host tables ------------------------------------------------------------------------------------------------------------------
bool big_supported(unsigned long long n) {
    if (n < 64) return false;
    const bool p2 = (n & (n - 1)) == 0;
    return p2 ? n <= (1ull << 21) : n <= (1ull << 20);
}

bool big_host_tables(unsigned n, BigHost &h) {
    if (!big_supported(n)) return false;
    const bool p2 = (n & (n - 1)) == 0;
    unsigned long long M = 1;
    unsigned l = 0;
    while (M < (p2 ? (unsigned long long)n : 2ull * n - 1ull)) { M <<= 1; ++l; }
    h.n = n; h.M = unsigned(M); h.chirp = !p2;
    h.l1 = (l + 1) / 2; h.l2 = l / 2;
    h.M1 = 1u << h.l1; h.M2 = 1u << h.l2;
    const double pi = 3.14159265358979323846264338327950288;
    h.wl.resize(size_t(h.M1));  // W_M1^j, j < M1 / 2, interleaved
    for (unsigned j = 0; j < h.M1 / 2; ++j) {
        const double a = -2.0 * pi * double(j) / double(h.M1);
        h.wl[2 * j] = std::cos(a); h.wl[2 * j + 1] = std::sin(a);
    }
    const unsigned nhi = unsigned((M + 1023) / 1024), nlo = unsigned(std::min<unsigned long long>(M, 1024));
    h.thi.resize(2 * size_t(nhi)); h.tlo.resize(2 * size_t(nlo));
    for (unsigned q = 0; q < nhi; ++q) {
        const double a = -2.0 * pi * double(1024ull * q) / double(M);
        h.thi[2 * q] = std::cos(a); h.thi[2 * q + 1] = std::sin(a);
    }
    for (unsigned r = 0; r < nlo; ++r) {
        const double a = -2.0 * pi * double(r) / double(M);
        h.tlo[2 * r] = std::cos(a); h.tlo[2 * r + 1] = std::sin(a);
    }
    h.c.clear(); h.bhat.clear();
    if (!h.chirp) return true;
    h.c.resize(2 * size_t(n));
    for (unsigned j = 0; j < n; ++j) {  // c_j = e^(+i pi j^2 / n), the angle reduced in integers (j < 2^21: j^2 fits 64 bits)
        const unsigned long long r = ((unsigned long long)j * j) % (2ull * n);
        const double a = pi * double(r) / double(n);
        h.c[2 * j] = std::cos(a); h.c[2 * j + 1] = std::sin(a);
    }
    // bhat = FFT_M(b) / M, b[m] = c_m for |m| < n wrapped — a plain f64 radix-2 transform on the host, once per plan
    std::vector<double> re(M, 0.0), im(M, 0.0);
    re[0] = h.c[0]; im[0] = h.c[1];
    for (unsigned j = 1; j < n; ++j) {
        re[j] = h.c[2 * j]; im[j] = h.c[2 * j + 1];
        re[M - j] = h.c[2 * j]; im[M - j] = h.c[2 * j + 1];
    }
    for (size_t i = 1, j = 0; i < M; ++i) {
        size_t bit = M >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    std::vector<double> wr(M / 2), wi(M / 2);
    for (size_t j = 0; j < M / 2; ++j) {
        const double a = -2.0 * pi * double(j) / double(M);
        wr[j] = std::cos(a); wi[j] = std::sin(a);
    }
    for (size_t hh = 1; hh < M; hh <<= 1) {
        const size_t stp = M / (2 * hh);
        for (size_t b0 = 0; b0 < M; b0 += 2 * hh)
            for (size_t j = 0; j < hh; ++j) {
                const double xr = re[b0 + hh + j], xi = im[b0 + hh + j], cr = wr[j * stp], ci = wi[j * stp];
                const double tr = xr * cr - xi * ci, ti = xr * ci + xi * cr;
                const double ur = re[b0 + j], ui = im[b0 + j];
                re[b0 + j] = ur + tr; im[b0 + j] = ui + ti;
                re[b0 + hh + j] = ur - tr; im[b0 + hh + j] = ui - ti;
            }
    }
    h.bhat.resize(2 * size_t(M));
    const double inv = 1.0 / double(M);
    for (size_t k = 0; k < M; ++k) {  // k = k1 + M1 k2 lives at [k1][k2]
        const size_t k1 = k & (h.M1 - 1), k2 = k >> h.l1, at = k1 * h.M2 + k2;
        h.bhat[2 * at] = re[k] * inv; h.bhat[2 * at + 1] = im[k] * inv;
    }
    return true;
}

size_t big_chunk_seqs(const BigDev &t, int dtype, size_t nseq) {
    const size_t per = size_t(t.M) * 2 * esz_of(dtype);
    size_t c = kBigChunkBytes / per;
    if (c < 1) c = 1;
    if (c > 32768) c = 32768;  // (grid.y of the prologue / epilogue kernels)
    return c < nseq ? c : nseq;
}

hipError_t big_upload(const BigHost &h, int dtype, BigDev &d) {
    d.n = h.n; d.M = h.M; d.M1 = h.M1; d.M2 = h.M2; d.l1 = h.l1; d.l2 = h.l2; d.chirp = h.chirp;
    auto up = [&](void **dst, const std::vector<double> &v) -> hipError_t {
        if (v.empty()) { *dst = nullptr; return hipSuccess; }
        hipError_t e = hipMalloc(dst, v.size() * esz_of(dtype));
        if (e != hipSuccess) return e;
        if (dtype == SGX_F64) return hipMemcpy(*dst, v.data(), v.size() * 8, hipMemcpyHostToDevice);
        std::vector<float> c(v.begin(), v.end());
        return hipMemcpy(*dst, c.data(), c.size() * 4, hipMemcpyHostToDevice);
    };
    hipError_t e;
    if ((e = up(&d.wl, h.wl)) != hipSuccess || (e = up(&d.thi, h.thi)) != hipSuccess || (e = up(&d.tlo, h.tlo)) != hipSuccess ||
        (e = up(&d.c, h.c)) != hipSuccess || (e = up(&d.bhat, h.bhat)) != hipSuccess) {
        big_free(d);
        return e;
    }
    return hipSuccess;
}
void big_free(BigDev &d) {
    for (void **b : {&d.wl, &d.thi, &d.tlo, &d.c, &d.bhat})
        if (*b) { (void)hipFree(*b); *b = nullptr; }
    d.M = 0;
}
size_t big_scratch_bytes(const BigDev &t, int dtype, size_t nseq) {
    return big_chunk_seqs(t, dtype, nseq) * size_t(t.M) * 2 * esz_of(dtype) * (t.chirp ? 1 : 2);
}

template <typename T>
static hipError_t big_stft_t(const BigDev &t, const StftArgs &a, void *scratch, hipStream_t s) {
    using V = c2_t<T>;
    const unsigned PP = (a.n_frames + 1u) / 2u;
    const size_t nseq = size_t(a.batch) * PP, chunk = big_chunk_seqs(t, sizeof(T) == 8 ? SGX_F64 : SGX_F32, nseq);
    V *buf = (V *)scratch, *nat = t.chirp ? buf : buf + chunk * size_t(t.M);
    const V *chirp = (const V *)t.c;
    for (size_t q0 = 0; q0 < nseq; q0 += chunk) {
        const unsigned cn = unsigned(std::min(chunk, nseq - q0));
        BigPass fr{};  // (the prologue inside pass A: the sequences are written once, a chirp-z sequence's zero half is never read)
        fr.x = a.x; fr.win = a.window; fr.chirp = chirp; fr.sample_stride = a.sample_stride; fr.n_samples = a.n_samples; fr.q0 = q0;
        fr.n = a.n_fft; fr.hop = a.hop; fr.pad = a.pad; fr.n_frames = a.n_frames; fr.PP = PP;
        hipError_t e = run_chain<T>(t, buf, nat, cn, s, &fr);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_big_split<T>, dim3((a.nb_fft + 255u) / 256u, cn), dim3(256), 0, s, (const V *)nat, chirp, (T *)a.out, a.n_fft, t.M,
                           a.nb_fft, a.n_frames, PP, (unsigned long long)q0, a.out_mode == OUT_COMPLEX ? 1 : 0, a.amp, a.eps);
    }
    return hipGetLastError();
}
hipError_t launch_big_stft(const BigDev &t, const StftArgs &a, void *scratch, int dtype, hipStream_t s) {
    if (a.out_mode == OUT_MEL) return hipErrorInvalidConfiguration;  // filterbanks: split path
    return dtype == SGX_F64 ? big_stft_t<double>(t, a, scratch, s) : big_stft_t<float>(t, a, scratch, s);
}

template <typename T>
static hipError_t big_c2r_t(const BigDev &t, const C2rArgs &c, void *scratch, hipStream_t s) {
    using V = c2_t<T>;
    const unsigned n = c.ncols, nb = n / 2 + 1, nfr = c.nrows, PP = (nfr + 1u) / 2u;
    const size_t nseq = size_t(c.batch) * PP, chunk = big_chunk_seqs(t, sizeof(T) == 8 ? SGX_F64 : SGX_F32, nseq);
    V *buf = (V *)scratch, *nat = t.chirp ? buf : buf + chunk * size_t(t.M);
    const V *chirp = (const V *)t.c;
    for (size_t q0 = 0; q0 < nseq; q0 += chunk) {
        const unsigned cn = unsigned(std::min(chunk, nseq - q0));
        hipLaunchKernelGGL(k_big_herm<T>, dim3((t.M + 255u) / 256u, cn), dim3(256), 0, s, (const V *)c.in, c.in_img, c.in_ks, c.in_rs, chirp, buf, n,
                           t.M, nb, nfr, PP, (unsigned long long)q0, c.bad_flag);
        hipError_t e = run_chain<T>(t, buf, nat, cn, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_big_unframe<T>, dim3((n + 255u) / 256u, cn), dim3(256), 0, s, (const V *)nat, chirp, (const T *)c.win, (T *)c.out, n, t.M,
                           nfr, PP, (unsigned long long)q0, c.scale);
    }
    return hipGetLastError();
}
hipError_t launch_big_c2r(const BigDev &t, const C2rArgs &c, void *scratch, int dtype, hipStream_t s) {
    return dtype == SGX_F64 ? big_c2r_t<double>(t, c, scratch, s) : big_c2r_t<float>(t, c, scratch, s);
}

template <typename T>
static hipError_t big_c2c_t(const BigDev &t, const C2cArgs &c, void *scratch, hipStream_t s) {
    using V = c2_t<T>;
    const size_t nseq = size_t(c.nseq) * c.batch, chunk = big_chunk_seqs(t, sizeof(T) == 8 ? SGX_F64 : SGX_F32, nseq);
    V *buf = (V *)scratch, *nat = t.chirp ? buf : buf + chunk * size_t(t.M);
    const V *chirp = (const V *)t.c;
    if (c.batch != 1) return hipErrorInvalidConfiguration;  // (sequence strides only: the 1-D C2cPlan)
    for (size_t q0 = 0; q0 < nseq; q0 += chunk) {
        const unsigned cn = unsigned(std::min(chunk, nseq - q0));
        hipLaunchKernelGGL(k_big_cin<T>, dim3((t.M + 255u) / 256u, cn), dim3(256), 0, s, (const V *)c.in, c.in_ss, c.in_is, chirp, buf, c.n, t.M,
                           (unsigned long long)q0, c.inverse);
        hipError_t e = run_chain<T>(t, buf, nat, cn, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_big_cout<T>, dim3((c.n + 255u) / 256u, cn), dim3(256), 0, s, (const V *)nat, chirp, (V *)c.out, c.out_ss, c.out_is, c.n, t.M,
                           (unsigned long long)q0, c.inverse, c.scale);
    }
    return hipGetLastError();
}
hipError_t launch_big_c2c(const BigDev &t, const C2cArgs &c, void *scratch, int dtype, hipStream_t s) {
    return dtype == SGX_F64 ? big_c2c_t<double>(t, c, scratch, s) : big_c2c_t<float>(t, c, scratch, s);
}

}  // namespace sgx
