// kernels_reg2d.hip — register-tiled complex transforms for the 2-D path and the generic inverse row transform, f32 / f64.
//
// Same construction as k_reg_radix (kernels_generic.hip): a length-N transform as two or three in-register passes (lengths
// A, B, C from reg_split_len: powers of two, and the mixed 2-3-5 sizes) around LDS exchanges, every pass after the first
// in place on the elements its work item owns.
//
//   k_c2c_reg   `tile` complex sequences of length N = A B C per workgroup, arbitrary element strides on both sides
//               (the thread mapping of the loads / stores follows the unit stride), forward or inverse (conjugate trick),
//               scale.  Column pass of fft2d / ifft2d (src/fft_backend.rs:674-688, :760-779) for every N with a split.
//   k_c2r_reg   half spectrum (m + 1 bins, m = A B C) -> 2 m real samples per row: Hermitian fold
//               Z'[k] = (X[k] + conj X[m-k]) + i conj(W_2m^k) (X[k] - conj X[m-k]), m-point inverse transform, x[2n] + i x[2n+1];
//               DC / Nyquist bins forced real and reported (fft_backend.rs:782-793), optional synthesis window.  Inverse row
//               pass of ifft2d / convolve_fft and the per-frame C2R of the generic inverse STFT (src/spectrogram.rs:4789-4811).
#include <algorithm>
#include <cstdlib>

#include "reg_radix.h"
#include "rr_layout.h"
#include "xcd_map.h"

namespace sgx {
namespace {

constexpr size_t kR2Budget = 72 * 1024;  // LDS per workgroup: two workgroups per CU
size_t elem_size(int dtype) { return dtype == SGX_F64 ? 8 : 4; }

template <typename T, int A_, int B_, int C_>
__global__ __launch_bounds__(256, (rr_waves<T, A_, B_, C_>())) void k_c2c_reg(C2cArgs a, unsigned ltile) {
    typedef typename PairOf<T>::type V;
    constexpr unsigned A = A_, B = B_, C = C_, BC = B * C, N = A * BC;
    constexpr int LA = ct_log2_ceil(A), LB = ct_log2_ceil(B);
    constexpr bool P2 = ct_is_pow2(N);
    typedef RrLayout<sizeof(V), A_, B_, C_> L;  // where element (k1, hi C + lo) of a sequence lives in LDS (rr_layout.h)
    constexpr unsigned RS = L::RS, FS = L::FS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    V *buf = (V *)smem;  // [tile][FS]
    const unsigned tid = threadIdx.x, tile = 1u << ltile;
    const unsigned lb = xcd_logical_block(a.tiles * a.batch);  // neighbouring tiles share cache lines: keep them in one XCD
    if (lb >= a.tiles * a.batch) return;
    const unsigned t = lb % a.tiles, b = lb / a.tiles;
    const unsigned s0 = t * tile;
    const unsigned ns = min(tile, a.nseq - s0);
    const V *in = (const V *)a.in + (size_t)b * a.in_img;
    V *out = (V *)a.out + (size_t)b * a.out_img;
    const V *tw = (const V *)a.tw;  // W_N^k, N entries
    const T cj = a.inverse ? T(-1) : T(1);  // inverse = conj(forward(conj x))
    auto wrap = [](unsigned e) { return P2 ? (e & (N - 1)) : (e % N); };

    // pass 1, software-pipelined: the loads of work item idx + 256 are in flight while item idx is transformed
    auto item = [&](unsigned idx, unsigned &s, unsigned &r) {
        if (a.in_seq_fast) { s = idx & (tile - 1); r = idx >> ltile; } else { r = idx % BC; s = idx / BC; }
        return idx < tile * BC && s < ns;
    };
    auto fetch = [&](unsigned idx, V (&v)[A]) {
        unsigned s, r;
        if (!item(idx, s, r)) return;
        const V *p = in + (size_t)(s0 + s) * a.in_ss + (size_t)r * a.in_is;
        const size_t step = (size_t)BC * a.in_is;
#pragma unroll
        for (unsigned n1 = 0; n1 < A; ++n1) v[n1] = p[n1 * step];
    };
    V nxt[A];
#pragma unroll
    for (unsigned n1 = 0; n1 < A; ++n1) nxt[n1] = (V){T(0), T(0)};
    fetch(tid, nxt);
    for (unsigned idx = tid; idx < tile * BC; idx += 256) {
        V v[A];
#pragma unroll
        for (unsigned n1 = 0; n1 < A; ++n1) v[n1] = nxt[n1] * (V){T(1), cj};
        fetch(idx + 256, nxt);
        unsigned s, r;
        if (!item(idx, s, r)) continue;
        inreg::MixFft<A, V>::run(v);
        V pw2[LA];
#pragma unroll
        for (int j = 0; j < LA; ++j) pw2[j] = tw[wrap((1u << j) * r)];
        V *dst = buf + (size_t)s * FS;
        const unsigned pp = L::hi_part(r / C) ^ (r % C);
        dst[pp ^ L::k1_mask(0)] = v[0];
#pragma unroll
        for (unsigned k1 = 1; k1 < A; ++k1) (dst + (pp ^ L::k1_mask(k1)))[k1 * RS] = inreg::cmulv(v[k1], rr_twiddle<LA>(pw2, k1));
    }
    __syncthreads();
    for (unsigned idx = tid; idx < ns * A * C; idx += 256) {
        const unsigned s = idx / (A * C), q = idx % (A * C), k1 = q / C, n3 = q % C;
        V *row = buf + (size_t)s * FS + k1 * RS;
        const unsigned lp = n3 ^ L::k1_mask(k1);  // point n2 sits at lp ^ hi_part(n2)
        V x[B];
#pragma unroll
        for (unsigned n2 = 0; n2 < B; ++n2) x[n2] = row[lp ^ L::hi_part(n2)];
        inreg::MixFft<B, V>::run(x);
        row[lp ^ L::hi_part(0)] = x[0];
        if constexpr (C > 1) {  // W_(BC)^(k2 n3) = W_N^(A k2 n3)
            V q2[LB];
#pragma unroll
            for (int j = 0; j < LB; ++j) q2[j] = tw[wrap((A << j) * n3)];
#pragma unroll
            for (unsigned k2 = 1; k2 < B; ++k2) row[lp ^ L::hi_part(k2)] = inreg::cmulv(x[k2], rr_twiddle<LB>(q2, k2));
        } else {
#pragma unroll
            for (unsigned k2 = 1; k2 < B; ++k2) row[lp ^ L::hi_part(k2)] = x[k2];
        }
    }
    __syncthreads();
    if constexpr (C > 1) {
        for (unsigned idx = tid; idx < ns * A * B; idx += 256) {
            const unsigned s = idx / (A * B), q = idx % (A * B), k1 = q / B, k2 = q % B;
            V *row = buf + (size_t)s * FS + k1 * RS;
            const unsigned lp = L::hi_part(k2) ^ L::k1_mask(k1);  // point n3 sits at lp ^ n3
            V x[C];
#pragma unroll
            for (unsigned n3 = 0; n3 < C; ++n3) x[n3] = row[lp ^ n3];
            inreg::MixFft<C, V>::run(x);
#pragma unroll
            for (unsigned k3 = 0; k3 < C; ++k3) row[lp ^ k3] = x[k3];
        }
        __syncthreads();
    }
    // X[k], k = k1 + A (k2 + B k3), sits at row k1, position C k2 + k3
    const T sc = (T)a.scale;
    for (unsigned idx = tid; idx < tile * N; idx += 256) {
        unsigned s, k;
        if (a.out_seq_fast) { s = idx & (tile - 1); k = idx >> ltile; } else { k = idx % N; s = idx / N; }
        if (s >= ns) continue;
        V v = buf[(size_t)s * FS + L::of_output(k)] * (V){sc, cj * sc};
        if (a.mul) {  // fused spectrum product (uniform branch)
            const size_t mi = (size_t)k * a.mul_ks + (a.mul_bcast ? 0 : s0 + s);  // mul_bcast: one table for every sequence
            if (a.mul_real) {
                const T mk = ((const T *)a.mul)[mi];
                v = v * (V){mk, mk};
            } else {
                const V y = ((const V *)a.mul)[mi];  // (a.re b.re - a.im b.im, a.re b.im + a.im b.re), as k_pointwise
                v = (V){v.x * y.x - v.y * y.y, v.x * y.y + v.y * y.x};
            }
        }
        out[(size_t)(s0 + s) * a.out_ss + (size_t)k * a.out_is] = v;
    }
}

// Overlap-add of a tile's frames, read straight from the transform buffer, into one signal's output (src/spectrogram.rs:4906-4930).
// The tile's nbk hop blocks are nbk * hop output positions, walked by all 256 threads (consecutive threads = consecutive
// positions: contiguous stores).  Position pos = (h0 + hb) hop + off receives frames f in [fh - q + 1, fh] (fh = h0 + hb,
// q = ceil((n - off) / hop)) clipped to [0, n_frames), in ascending f as the reference adds them, frame sample j = (fh - f) hop + off;
// norm = sum of w[j]^2 (each product rounded, then added), divide where > 1e-10.  The interior norm of every offset is built
// once per tile into `nrm_tab` (LDS, hop entries).  Requires hop <= n.
// sample(rr, j): sample j of the tile's row rr, scaled and windowed ((x / n) w, each product rounded as the unfused path rounds it)
// (Before: one thread per offset walking all nbk blocks — at hop 128 half the workgroup idled through the longest phase of the
// kernel: f32 n_fft 512 / hop 128 inverse STFT 0.49 ms.)
template <typename T, typename F>
__device__ __forceinline__ void ola_tile(F &&sample, unsigned n, const T *w, T *o, const C2rArgs &a, long long h0, long long fbase, unsigned tid,
                                         T *nrm_tab) {
    const unsigned long long p0 = (unsigned long long)h0 * a.hop;
    const long long last = (long long)a.nrows - 1;
    auto sq_add = [](T acc, T wj) {
        if constexpr (sizeof(T) == 4) return __fadd_rn(acc, __fmul_rn(wj, wj));
        else return __dadd_rn(acc, __dmul_rn(wj, wj));
    };
    // frames overlapping offset off: q_hi below thr, q_hi - 1 from thr on (uniform values, no per-position division)
    const unsigned q_hi = (n + a.hop - 1u) / a.hop, thr = n - (q_hi - 1u) * a.hop;
    for (unsigned off = tid; off < a.hop; off += 256) {
        const unsigned q = off < thr ? q_hi : q_hi - 1u;
        T nrm_full = T(0);
        for (unsigned i = q; i-- > 0;) nrm_full = sq_add(nrm_full, w[i * a.hop + off]);
        nrm_tab[off] = nrm_full;
    }
    __syncthreads();
    const unsigned dq = 256u / a.hop, dr = 256u - dq * a.hop;  // one step of 256 positions = dq blocks + dr offsets
    unsigned hb = tid / a.hop, off = tid - hb * a.hop;
    for (; hb < a.nbk; hb += dq, off += dr) {
        if (off >= a.hop) {
            off -= a.hop;
            if (++hb >= a.nbk) break;
        }
        const unsigned long long pos = p0 + (unsigned long long)hb * a.hop + off;
        if (pos < a.start || pos - a.start >= a.out_len) continue;
        const unsigned q = off < thr ? q_hi : q_hi - 1u;
        const long long fh = h0 + hb;
        const long long f_lo = max(fh - (long long)q + 1, 0ll), f_hi = min(fh, last);
        T acc = T(0), nrm = nrm_tab[off];
        for (long long f = f_lo; f <= f_hi; ++f) acc += sample((unsigned)(f - fbase), (unsigned)(fh - f) * a.hop + off);
        if (f_hi - f_lo + 1 != (long long)q) {  // signal edges: fewer frames
            nrm = T(0);
            for (long long f = f_lo; f <= f_hi; ++f) nrm = sq_add(nrm, w[(unsigned)(fh - f) * a.hop + off]);
        }
        if (nrm > T(1e-10)) acc /= nrm;
        o[pos - a.start] = acc;
    }
}

template <typename T, int A_, int B_, int C_, bool OLA>
__global__ __launch_bounds__(256, (rr_waves<T, A_, B_, C_>())) void k_c2r_reg(C2rArgs a, unsigned ltile) {
    typedef typename PairOf<T>::type V;
    constexpr unsigned A = A_, B = B_, C = C_, BC = B * C, M = A * BC, CN = 2 * M;
    constexpr int LA = ct_log2_ceil(A), LB = ct_log2_ceil(B);
    constexpr bool P2 = ct_is_pow2(M);
    typedef RrLayout<sizeof(V), A_, B_, C_> L;
    constexpr unsigned RS = L::RS, FS = L::FS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x, tile = 1u << ltile;
    V *buf = (V *)smem;                   // [tile][FS]
    const unsigned lb = xcd_logical_block(a.tiles * a.batch);
    if (lb >= a.tiles * a.batch) return;
    const unsigned t = lb % a.tiles, b = lb / a.tiles;
    // OLA: row rr of the tile is frame fbase + rr (the first a.ov rows are the halo); frames outside the signal are zero rows
    const long long h0 = (long long)t * a.nbk, fbase = OLA ? h0 - (long long)a.ov : (long long)t * tile;
    const unsigned r0 = (unsigned)fbase;  // (!OLA)
    const unsigned nr = OLA ? tile : min(tile, a.nrows - r0);
    auto row_ok = [&](unsigned rr) { return OLA ? (fbase + rr >= 0 && fbase + rr < (long long)a.nrows) : rr < nr; };
    const V *in = (const V *)a.in + (size_t)b * a.in_img;
    T *out = (T *)a.out + (size_t)b * a.nrows * CN;
    const V *tw = (const V *)a.tw;  // W_CN^k, CN entries
    auto wrap = [](unsigned e) { return P2 ? (e & (CN - 1)) : (e % CN); };
    // OLA: behind the tile sit the overlap-add's norm table (hop entries) and a copy of the window (CN entries; visible after the
    // barriers of the passes): the walk reads the window four to eight times per output sample
    T *nrm_tab = (T *)(buf + (size_t)tile * FS), *win_lds = nrm_tab + a.hop;
    if constexpr (OLA)
        for (unsigned i = tid; i < CN; i += 256) win_lds[i] = ((const T *)a.win)[i];

    // pass 1: every item loads its A pairs (X[k], X[m - k]) itself — lanes run along the contiguous axis of the input (rows for
    // the [bin][frame] spectra of the inverse STFT, bins for the [row][bin] spectra of the 2-D path), 2 A loads in flight a thread
    for (unsigned idx = tid; idx < tile * BC; idx += 256) {
        unsigned r, rr;
        if (a.k_fast) { r = idx % BC; rr = idx / BC; } else { rr = idx & (tile - 1); r = idx >> ltile; }
        const bool ok = row_ok(rr);  // rows past the image / frames outside the signal transform zeros
        const V *xp = in + (ok ? (size_t)(fbase + rr) * a.in_rs : 0);
        V Xa[A], Yb[A];
#pragma unroll
        for (unsigned n1 = 0; n1 < A; ++n1) {
            const unsigned k = BC * n1 + r;
            Xa[n1] = ok ? xp[(size_t)k * a.in_ks] : (V){T(0), T(0)};
            Yb[n1] = ok ? xp[(size_t)(M - k) * a.in_ks] : (V){T(0), T(0)};
        }
        if (r == 0) {  // k = 0: DC and Nyquist bins forced real; realfft reports a non-zero imaginary part
            if (a.bad_flag && (Xa[0].y != T(0) || Yb[0].y != T(0))) atomicOr(a.bad_flag, 1u);
            Xa[0].y = T(0);
            Yb[0].y = T(0);
        }
        V v[A];
#pragma unroll
        for (unsigned n1 = 0; n1 < A; ++n1) {
            const V w = tw[BC * n1 + r];
            // S = X[k] + conj X[m-k], D = X[k] - conj X[m-k], T = conj(W^k) D; the forward-transform trick wants conj(S + i T)
            const V S = inreg::pfma(Yb[n1], (V){T(1), T(-1)}, Xa[n1]), D = inreg::pfma(Yb[n1], (V){T(-1), T(1)}, Xa[n1]);
            const V Tt = inreg::cmulv(D, (V){w.x, -w.y});
            v[n1] = inreg::pfma(inreg::swp(Tt), (V){T(-1), T(-1)}, S * (V){T(1), T(-1)});
        }
        inreg::MixFft<A, V>::run(v);
        V pw2[LA];
#pragma unroll
        for (int j = 0; j < LA; ++j) pw2[j] = tw[wrap((2u << j) * r)];  // W_m^e = W_CN^(2e)
        V *dst = buf + (size_t)rr * FS;
        const unsigned pp = L::hi_part(r / C) ^ (r % C);
        dst[pp ^ L::k1_mask(0)] = v[0];
#pragma unroll
        for (unsigned k1 = 1; k1 < A; ++k1) (dst + (pp ^ L::k1_mask(k1)))[k1 * RS] = inreg::cmulv(v[k1], rr_twiddle<LA>(pw2, k1));
    }
    __syncthreads();
    for (unsigned idx = tid; idx < tile * A * C; idx += 256) {
        const unsigned rr = idx / (A * C), q = idx % (A * C), k1 = q / C, n3 = q % C;
        V *row = buf + (size_t)rr * FS + k1 * RS;
        const unsigned lp = n3 ^ L::k1_mask(k1);
        V x[B];
#pragma unroll
        for (unsigned n2 = 0; n2 < B; ++n2) x[n2] = row[lp ^ L::hi_part(n2)];
        inreg::MixFft<B, V>::run(x);
        row[lp ^ L::hi_part(0)] = x[0];
        if constexpr (C > 1) {
            V q2[LB];
#pragma unroll
            for (int j = 0; j < LB; ++j) q2[j] = tw[wrap((2u * A << j) * n3)];
#pragma unroll
            for (unsigned k2 = 1; k2 < B; ++k2) row[lp ^ L::hi_part(k2)] = inreg::cmulv(x[k2], rr_twiddle<LB>(q2, k2));
        } else {
#pragma unroll
            for (unsigned k2 = 1; k2 < B; ++k2) row[lp ^ L::hi_part(k2)] = x[k2];
        }
    }
    __syncthreads();
    if constexpr (C > 1) {
        for (unsigned idx = tid; idx < tile * A * B; idx += 256) {
            const unsigned rr = idx / (A * B), q = idx % (A * B), k1 = q / B, k2 = q % B;
            V *row = buf + (size_t)rr * FS + k1 * RS;
            const unsigned lp = L::hi_part(k2) ^ L::k1_mask(k1);
            V x[C];
#pragma unroll
            for (unsigned n3 = 0; n3 < C; ++n3) x[n3] = row[lp ^ n3];
            inreg::MixFft<C, V>::run(x);
#pragma unroll
            for (unsigned k3 = 0; k3 < C; ++k3) row[lp ^ k3] = x[k3];
        }
        __syncthreads();
    }
    // z[n] = conj(result[n]) * scale = (x[2n], x[2n+1]); lanes over n: contiguous row stores
    const T sc = (T)a.scale;
    const V *win = (const V *)a.win;
    if constexpr (OLA) {
        // the frames stay on chip: the overlap-add picks sample j of row rr out of the transform buffer (pair j / 2, real part
        // for even j, minus the imaginary part for odd j), scales and windows it as the unfused path does
        const T *w = win_lds;
        ola_tile<T>([&](unsigned rr, unsigned j) {
            const V z = buf[(size_t)rr * FS + L::of_output(j >> 1)];
            const T x = ((j & 1u) ? -z.y : z.x) * sc;
            if constexpr (sizeof(T) == 4) return __fmul_rn(x, w[j]);
            else return __dmul_rn(x, w[j]);
        }, CN, w, (T *)a.out + (size_t)b * a.out_len, a, h0, fbase, tid, nrm_tab);
    } else {
        for (unsigned idx = tid; idx < nr * M; idx += 256) {
            const unsigned n = idx % M, rr = idx / M;
            V v = buf[(size_t)rr * FS + L::of_output(n)] * (V){sc, -sc};
            if (win) v = v * win[n];
            *(V *)(out + (size_t)(r0 + rr) * CN + 2u * n) = v;
        }
    }
}

template <typename T, int A, int B, int C>
hipError_t launch_c2c_t(const C2cArgs &a, unsigned ltile, size_t lds, hipStream_t s) {
    if (lds > 64 * 1024) {
        hipError_t e = set_max_dynamic_lds((const void *)k_c2c_reg<T, A, B, C>, (int)kR2Budget);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_c2c_reg<T, A, B, C>), dim3(xcd_grid((unsigned long long)a.tiles * a.batch)), dim3(256), lds, s, a, ltile);
    return hipGetLastError();
}

template <typename T, int A, int B, int C>
hipError_t launch_c2r_t(const C2rArgs &a, unsigned ltile, size_t lds, hipStream_t s) {
    const bool ola = a.nbk != 0;
    if (lds > 64 * 1024) {
        hipError_t e = ola ? set_max_dynamic_lds((const void *)k_c2r_reg<T, A, B, C, true>, (int)kR2Budget)
                           : set_max_dynamic_lds((const void *)k_c2r_reg<T, A, B, C, false>, (int)kR2Budget);
        if (e != hipSuccess) return e;
    }
    const dim3 grid(xcd_grid((unsigned long long)a.tiles * a.batch));
    if (ola) hipLaunchKernelGGL((k_c2r_reg<T, A, B, C, true>), grid, dim3(256), lds, s, a, ltile);
    else hipLaunchKernelGGL((k_c2r_reg<T, A, B, C, false>), grid, dim3(256), lds, s, a, ltile);
    return hipGetLastError();
}

constexpr bool kRegOff = false;

}  // namespace

// hipErrorNotSupported: no split for this length / layout — the caller uses the LDS-tile kernels instead
hipError_t launch_c2c_reg(const C2cArgs &a0, int dtype, hipStream_t s) {
    unsigned fa, fb, fc;
    if (kRegOff || !reg_split_len(a0.n, dtype, &fa, &fb, &fc)) return hipErrorNotSupported;
    const size_t es = elem_size(dtype);
    if (((size_t)a0.in | (size_t)a0.out) & (2 * es - 1)) return hipErrorNotSupported;
    const size_t fs = rr_frame_stride(fa, rr_swizzle(2 * (unsigned)es, fa, fb, fc).rs);
    unsigned ltile = 5;  // up to 32 sequences per workgroup; no more than the job has
    // a quarter of a CU's LDS (four workgroups per CU overlap their phases: 2048 x 512^2 fft2d 1.91 -> 1.74 ms, 8192 x 256^2
    // 1.81 -> 1.63 ms) unless that leaves fewer than 8 sequences per tile (128 x 2048^2: 2.32 ms with 4 per tile, 2.77 with 2)
    auto tile_for = [&](size_t budget) {
        unsigned lt = 5;
        while (lt > 0 && ((size_t)(1u << lt) * fs * 2 * es > budget || (1u << (lt - 1)) >= a0.nseq)) --lt;
        return lt;
    };
    ltile = tile_for(40 * 1024);
    if (ltile < 3) ltile = tile_for(kR2Budget);
    const size_t lds = (size_t)(1u << ltile) * fs * 2 * es;
    if (lds > kR2Budget) return hipErrorNotSupported;
    C2cArgs a = a0;
    a.tile = 1u << ltile;
    a.tiles = (a.nseq + a.tile - 1) / a.tile;
    const unsigned long long g = (unsigned long long)a.tiles * a.batch;
    if (g == 0 || g >= 0x7fffffffull) return hipErrorInvalidConfiguration;
#define SGX_C2C_F32(A, B, C) if (fa == A && fb == B && fc == C) return launch_c2c_t<float, A, B, C>(a, ltile, lds, s);
#define SGX_C2C_F64(A, B, C) if (fa == A && fb == B && fc == C) return launch_c2c_t<double, A, B, C>(a, ltile, lds, s);
    if (dtype == SGX_F64) {
        SGX_RR_SPLITS_F64(SGX_C2C_F64)
        SGX_RR_SPLITS_MIXED(SGX_C2C_F64)
    } else {
        SGX_RR_SPLITS_F32(SGX_C2C_F32)
        SGX_RR_SPLITS_MIXED(SGX_C2C_F32)
    }
#undef SGX_C2C_F32
#undef SGX_C2C_F64
    return hipErrorNotSupported;
}

// Geometry of launch_c2r_reg (pass split, rows per workgroup, LDS, halo of the fused inverse STFT) without the launch, so that
// sgx_reserve can tell whether a call will run fused (no frame scratch) with the very test the launcher applies.
static hipError_t c2r_reg_plan(const C2rArgs &a0, int dtype, C2rArgs &a, unsigned &ltile, size_t &lds, unsigned &fa, unsigned &fb, unsigned &fc) {
    if (kRegOff || (a0.ncols & 1u) || !reg_split_len(a0.ncols / 2, dtype, &fa, &fb, &fc)) return hipErrorNotSupported;
    const size_t es = elem_size(dtype);
    if (((size_t)a0.in | (size_t)a0.out | (size_t)a0.win) & (2 * es - 1)) return hipErrorNotSupported;
    const size_t m = a0.ncols / 2;
    const size_t per = (size_t)rr_frame_stride(fa, rr_swizzle(2 * (unsigned)es, fa, fb, fc).rs) * 2 * es;
    (void)m;
    const bool ola = a0.hop != 0;  // fused inverse STFT (launch_istft_reg)
#ifndef SGX_C2R_KB
#define SGX_C2R_KB 40
#endif
    // rows per workgroup: up to 32, within a quarter of a CU's LDS — four workgroups per CU overlap their load, transform and
    // store phases (measured: f32 n_fft 512 inverse STFT 0.65 ms with 32-frame tiles at two per CU, 0.49 ms with 16-frame tiles)
    // (f64: n_fft 400 0.90 vs 1.14 ms with the larger tile; f32 frames of 16 KB and more — n_fft >= 4096 — too, round 5: 40 KB is two frames
    // there, i.e. 16-byte runs of the [bin][frame] spectrum: 64 x 10 s inverse STFT n_fft 4096 / 1024 223 -> 216 us, 8192 / 2048 295 -> 258;
    // 144 KB: 236 / 266; shorter frames lose with the larger tile: 512 / 160 73 -> 97 us, 2048 / 100 0.90 -> 0.96 ms)
    const size_t c2r_budget = dtype == SGX_F64 || per >= 16384 ? kR2Budget : (size_t)SGX_C2R_KB * 1024;
    ltile = 5;
    const size_t ola_tab = ola ? ((size_t)a0.hop + a0.ncols) * es : 0;  // the overlap-add's norm table and window copy behind the tile
    while (ltile > 0 && ((size_t)(1u << ltile) * per + ola_tab > c2r_budget || (!ola && (1u << (ltile - 1)) >= a0.nrows))) --ltile;
    lds = (size_t)(1u << ltile) * per + ola_tab;
    if (lds > kR2Budget) return hipErrorNotSupported;
    a = a0;
    a.tile = 1u << ltile;
    a.tiles = (a.nrows + a.tile - 1) / a.tile;
    a.nbk = 0;
    if (ola) {
        // `ov` halo frames per tile are transformed twice: fused only while at least three quarters of a tile's frames are its
        // own, and in f32 (measured, 256 x 10 s, against rows into a frame scratch + k_istft_ola: f32 n_fft 512 / hop 128 0.90 ->
        // 0.81 ms with the spectrum staged through LDS, 0.49 ms with the direct loads; f64, whose tiles hold half as many frames:
        // 512 / 128 1.03 -> 0.95 ms but 400 / 160 0.89 -> 0.96 ms and 256 / 64 1.00 -> 1.19 ms: not fused)
        a.ov = (a.ncols - 1u) / a.hop;
#ifndef SGX_OLA_HALO
#define SGX_OLA_HALO 2
#endif
        if (a.hop > a.ncols || !a.win || SGX_OLA_HALO * a.ov > a.tile || dtype != SGX_F32) return hipErrorNotSupported;
        a.nbk = a.tile - a.ov;
        const unsigned long long full = (unsigned long long)(a.nrows - 1) * a.hop + a.ncols;
        const unsigned long long blocks = (full + a.hop - 1) / a.hop;
        a.tiles = (unsigned)((blocks + a.nbk - 1) / a.nbk);
    }
    return hipSuccess;
}

hipError_t launch_c2r_reg(const C2rArgs &a0, int dtype, hipStream_t s) {
    unsigned fa, fb, fc, ltile;
    size_t lds;
    C2rArgs a;
    const hipError_t pe = c2r_reg_plan(a0, dtype, a, ltile, lds, fa, fb, fc);
    if (pe != hipSuccess) return pe;
    const unsigned long long g = (unsigned long long)a.tiles * a.batch;
    if (g == 0 || g >= 0x7fffffffull) return hipErrorInvalidConfiguration;
#define SGX_C2R_F32(A, B, C) if (fa == A && fb == B && fc == C) return launch_c2r_t<float, A, B, C>(a, ltile, lds, s);
#define SGX_C2R_F64(A, B, C) if (fa == A && fb == B && fc == C) return launch_c2r_t<double, A, B, C>(a, ltile, lds, s);
    if (dtype == SGX_F64) {
        SGX_RR_SPLITS_F64(SGX_C2R_F64)
        SGX_RR_SPLITS_MIXED(SGX_C2R_F64)
    } else {
        SGX_RR_SPLITS_F32(SGX_C2R_F32)
        SGX_RR_SPLITS_MIXED(SGX_C2R_F32)
    }
#undef SGX_C2R_F32
#undef SGX_C2R_F64
    return hipErrorNotSupported;
}

static C2rArgs istft_reg_args(const void *spec, void *out, const void *win, const void *tw, unsigned n, unsigned n_frames, unsigned hop,
                              unsigned batch, unsigned long long start, unsigned long long out_len, double scale, unsigned *bad_flag) {
    C2rArgs c{};
    c.in = spec; c.out = out;
    c.nrows = n_frames; c.ncols = n; c.batch = batch;
    c.in_img = (unsigned long long)(n / 2 + 1) * n_frames;
    c.in_ks = n_frames; c.in_rs = 1; c.k_fast = 0;  // [bin][frame] (StftResult layout, S9)
    c.tw = tw; c.scale = scale; c.win = win; c.bad_flag = bad_flag;
    c.hop = hop; c.start = start; c.out_len = out_len;
    return c;
}

hipError_t launch_istft_reg(const void *spec, void *out, const void *win, const void *tw, unsigned n, unsigned n_frames, unsigned hop,
                            unsigned batch, unsigned long long start, unsigned long long out_len, double scale, unsigned *bad_flag,
                            int dtype, hipStream_t s) {
    if (hop == 0 || n_frames == 0) return hipErrorNotSupported;
    return launch_c2r_reg(istft_reg_args(spec, out, win, tw, n, n_frames, hop, batch, start, out_len, scale, bad_flag), dtype, s);
}

bool istft_reg_fuses(const void *win, unsigned n, unsigned n_frames, unsigned hop, unsigned batch, int dtype) {
    if (hop == 0 || n_frames == 0 || batch == 0) return false;
    // (the spectrum and output pointers of a real call are hipMalloc'ed or element-aligned: only `win` can fail the alignment test)
    const C2rArgs c = istft_reg_args(nullptr, nullptr, win, nullptr, n, n_frames, hop, batch, 0, 0, 1.0, nullptr);
    unsigned fa, fb, fc, ltile;
    size_t lds;
    C2rArgs a;
    if (c2r_reg_plan(c, dtype, a, ltile, lds, fa, fb, fc) != hipSuccess) return false;
    const unsigned long long g = (unsigned long long)a.tiles * a.batch;
    return g != 0 && g < 0x7fffffffull;
}

}  // namespace sgx
