// kernels_generic.hip — shape-generic STFT kernels (any n_fft, f32 and f64) for gfx950.
//
// These are the correctness/coverage kernels: every (n_fft, hop, window, centre, mapping, amp, dtype)
// the reference accepts runs on the GPU through one of them.  The BASELINE shape (f32, n_fft = 1024)
// dispatches to the tuned kernel in kernels_r32x16.hip instead.
//
//   k_lds_radix2  : power-of-two n_fft >= 4.  One 256-thread workgroup per tile of `ft` frames of one
//                   signal; z[n] = x[2n] + i x[2n+1] (windowed, zero-padded per S1) is loaded bit-reversed
//                   into LDS, log2(n/2) in-LDS radix-2 DIT stages, then the real split, then the epilogue.
//   k_direct_dft  : every other n_fft (the reference accepts arbitrary sizes, e.g. 400 —
//                   tests/mfcc_tests.rs:133).  Windowed frames in LDS, O(n^2) direct sum per bin.
//
// Epilogue (shared): |X|^2 (spectrogram.rs:1332-1334) -> identity or CSR Mel bank with sequential
// accumulation in T in ascending-bin order (:102-117) -> Power / sqrt / 10*log10(max(p, eps))
// (:1986-2036, :2068-2080) -> out[b][bin][frame] with the frame axis contiguous (S9).  Threads are
// mapped (bin, frame) with frame fastest so global stores are contiguous along frames.
#include <algorithm>
#include <numeric>
#include <cstdlib>

#include "buffer_ops.h"
#include "db_f64.h"
#include "reg_radix.h"
#include "rr_layout.h"

#ifndef SGX_RR_LOADPOS
#define SGX_RR_LOADPOS 1  // staged samples: 0 = the next tile is requested at the top of the tile, 1 = after pass 1
#endif

namespace sgx {

#ifndef SGX_RR_STAGGER
#define SGX_RR_STAGGER 0
#endif
#ifndef SGX_RR_STAGE_MEL
#define SGX_RR_STAGE_MEL 0
#endif
#ifndef SGX_RR_TWKEEP
#define SGX_RR_TWKEEP 0
#endif
#ifdef SGX_RR_STAMPS  // diagnostic build only (tools/stamps_generic.py): a wave's cycles per phase of k_reg_radix
__device__ unsigned long long g_rr_stamps[32];
#define RR_STAMP(i)                                                                         \
    do {                                                                                    \
        unsigned long long t_;                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        st_acc[i] += t_ - st_prev;                                                          \
        st_prev = t_;                                                                       \
    } while (0)
#else
#define RR_STAMP(i)
#endif

template <typename T>
struct Cx {
    T re, im;
};

__device__ inline float t_sqrt(float v) { return sqrtf(v); }
__device__ inline double t_sqrt(double v) { return sqrt(v); }
// 10 log10(v).  f32: (10 log10 2) log2(v) — the hardware's log2 and one multiply instead of log10f's extra-precision product
// (< 2e-5 dB off at |dB| <= 100); f64: db_f64.h (frexp + atanh series, branch-free)
__device__ inline float t_db(float v) { return __builtin_log2f(v) * 3.01029995663981195f; }
__device__ inline double t_db(double v) { return db_f64(v); }
__device__ inline float t_max(float a, float b) { return fmaxf(a, b); }
__device__ inline double t_max(double a, double b) { return fmax(a, b); }
// un-fused multiply-add: the reference's `acc += T::from_f64(w) * x` is two roundings (rustc never contracts)
__device__ inline float t_mul_add_unfused(float a, float b, float c) { return __fadd_rn(__fmul_rn(a, b), c); }
__device__ inline double t_mul_add_unfused(double a, double b, double c) { return __dadd_rn(__dmul_rn(a, b), c); }

template <typename T>
__device__ inline T amp_apply(T p, int amp, T eps) {
    if (amp == AMP_MAGNITUDE) return t_sqrt(p);
    if (amp == AMP_DB) return t_db(t_max(p, eps));
    return p;
}

template <typename T>
__device__ inline T load_sample(const T *xb, long long s, unsigned long long n) {
    return (s >= 0 && (unsigned long long)s < n) ? xb[s] : T(0);
}

// One spectrum value X[k] of frame f (tile-local) is ready: route it by output mode.
template <typename T>
__device__ inline void emit_bin(const StftArgs &a, unsigned b, unsigned frame, unsigned f, unsigned k, T re, T im,
                                T *pw, T eps) {
    if (a.out_mode == OUT_COMPLEX) {
        Cx<T> *o = (Cx<T> *)a.out;
        o[((size_t)b * a.n_out + k) * a.n_frames + frame] = Cx<T>{re, im};
    } else {
        T p = re * re + im * im;
        if (a.out_mode == OUT_MEL) {
            pw[(size_t)f * a.nb_fft + k] = a.amp == AMP_MAG_IN ? t_sqrt(p) : p;
        } else {
            T *o = (T *)a.out;
            o[((size_t)b * a.n_out + k) * a.n_frames + frame] = amp_apply(p, a.amp, eps);
        }
    }
}

// Mel stage: pw[f][k] holds the power spectrum of the tile's frames.  `scratch` is LDS the transform no longer needs
// (scratch_bytes of it): when the bank's CSR arrays fit they are staged there once per tile — otherwise every (band, frame)
// thread streams its band's values and columns from global memory, ft times redundantly.
template <typename T>
struct MelCsr {
    const T *val;
    const unsigned *col, *ptr;
};

// CSR arrays of the bank: staged into `scratch` (LDS) when they fit, else left in global memory.  Ends with a barrier
// when it staged (uniform branch).
template <typename T>
__device__ inline MelCsr<T> mel_resolve(const StftArgs &a, unsigned char *scratch, size_t scratch_bytes) {
    MelCsr<T> c{(const T *)a.mel_val, a.mel_col, a.mel_ptr};
    const size_t need = (size_t)a.mel_nnz * (sizeof(T) + 4) + (size_t)(a.n_mels + 1) * 4;
    if (need <= scratch_bytes) {  // uniform
        T *sval = (T *)scratch;
        unsigned *scol = (unsigned *)(sval + a.mel_nnz), *sptr = scol + a.mel_nnz;
        for (unsigned i = threadIdx.x; i < a.mel_nnz; i += blockDim.x) { sval[i] = c.val[i]; scol[i] = c.col[i]; }
        for (unsigned i = threadIdx.x; i <= a.n_mels; i += blockDim.x) sptr[i] = c.ptr[i];
        __syncthreads();
        c.val = sval; c.col = scol; c.ptr = sptr;
    }
    return c;
}

template <typename T>
__device__ inline void mel_apply(const StftArgs &a, const MelCsr<T> &c, unsigned b, unsigned f0, unsigned nf, const T *pw, T eps) {
    T *o = (T *)a.out;
    for (unsigned idx = threadIdx.x; idx < nf * a.n_mels; idx += blockDim.x) {
        unsigned f = idx % nf, mm = idx / nf;
        T acc = T(0);
        unsigned i0 = c.ptr[mm], i1 = c.ptr[mm + 1];
        for (unsigned i = i0; i < i1; i++) acc = t_mul_add_unfused(c.val[i], pw[(size_t)f * a.nb_fft + c.col[i]], acc);
        o[((size_t)b * a.n_out + mm) * a.n_frames + f0 + f] = amp_apply(acc, a.amp, eps);
    }
}

// Mel stage: pw[f][k] holds the power spectrum of the tile's frames.  `scratch` is LDS the transform no longer needs
// (scratch_bytes of it): when the bank's CSR arrays fit they are staged there once per tile — otherwise every (band, frame)
// thread streams its band's values and columns from global memory, ft times redundantly.
template <typename T>
__device__ inline void mel_stage(const StftArgs &a, unsigned b, unsigned f0, unsigned nf, const T *pw, T eps,
                                 unsigned char *scratch, size_t scratch_bytes) {
    const MelCsr<T> c = mel_resolve<T>(a, scratch, scratch_bytes);
    mel_apply<T>(a, c, b, f0, nf, pw, eps);
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_lds_radix2(StftArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned m = a.m, fs = m + 1;
    Cx<T> *buf = (Cx<T> *)smem;           // [ft][m+1]
    T *pw = (T *)(buf + (size_t)a.ft * fs);  // [ft][nb_fft] (Mel only)
    const unsigned tile = blockIdx.x % a.tiles, b = blockIdx.x / a.tiles;
    const unsigned f0 = tile * a.ft;
    const unsigned nf = min(a.ft, a.n_frames - f0);
    const T *xb = (const T *)a.x + (size_t)b * a.sample_stride;
    const T *w = (const T *)a.window;
    const Cx<T> *tw = (const Cx<T> *)a.tw;
    const T eps = (T)a.eps;
    const unsigned tid = threadIdx.x;
    if (a.tw_lds) {  // every stage and the real split index tw[0 .. n_fft/2): serve them from LDS instead of L1 gathers
        Cx<T> *ltw = (Cx<T> *)(smem + a.tw_lds);
        for (unsigned i = tid; i < m; i += 256) ltw[i] = tw[i];
        tw = ltw;  // visible after the barrier that follows the frame load
    }

    // 1. framing + window (S1, S4), bit-reversed placement of z[i] = x[2i] + i x[2i+1]
    for (unsigned idx = tid; idx < nf * m; idx += 256) {
        unsigned f = idx >> a.log2m, i = idx & (m - 1);
        long long s = (long long)(f0 + f) * a.hop + 2ll * i - (long long)a.pad;
        T x0 = load_sample(xb, s, a.n_samples) * w[2 * i];
        T x1 = load_sample(xb, s + 1, a.n_samples) * w[2 * i + 1];
        unsigned r = __brev(i) >> (32 - a.log2m);
        buf[f * fs + r] = Cx<T>{x0, x1};
    }
    __syncthreads();

    // 2. DIT stages on the bit-reversed data, W_{2h}^j = tw[j * n_fft / (2h)].  Two consecutive radix-2 stages are fused
    //    into one pass (radix-2^2: identical arithmetic and ordering, half the LDS round trips and barriers); an odd
    //    stage count starts with one plain radix-2 stage.
    unsigned h = 1, lh = 0;
    if (a.log2m & 1u) {
        const unsigned halfm = m >> 1, lhm = a.log2m - 1;
        for (unsigned idx = tid; idx < nf * halfm; idx += 256) {
            const unsigned f = idx >> lhm, q = idx & (halfm - 1);
            const unsigned p0 = f * fs + (q << 1), p1 = p0 + 1;  // h = 1: twiddle W_2^0 = 1
            const Cx<T> u = buf[p0], v = buf[p1];
            buf[p0] = Cx<T>{u.re + v.re, u.im + v.im};
            buf[p1] = Cx<T>{u.re - v.re, u.im - v.im};
        }
        __syncthreads();
        h = 2;
        lh = 1;
    }
    const unsigned quarter = m >> 2, lq = a.log2m - 2;
    for (; h < m; h <<= 2, lh += 2) {
        const unsigned st2 = a.n_fft >> (lh + 2);  // exponent step of W_{4h}
        for (unsigned idx = tid; idx < nf * quarter; idx += 256) {
            const unsigned f = idx >> lq, q = idx & (quarter - 1);
            const unsigned j = q & (h - 1), blk = q >> lh;
            const unsigned p = f * fs + (blk << (lh + 2)) + j;
            const Cx<T> t0 = tw[j * st2];            // W_{4h}^j
            const Cx<T> w = tw[2 * j * st2];         // W_{2h}^j = W_{4h}^{2j}
            const Cx<T> e0 = buf[p], e1 = buf[p + h], e2 = buf[p + 2 * h], e3 = buf[p + 3 * h];
            // stage h: pairs (e0, e1) and (e2, e3), twiddle w
            const T v1r = e1.re * w.re - e1.im * w.im, v1i = e1.re * w.im + e1.im * w.re;
            const T v3r = e3.re * w.re - e3.im * w.im, v3i = e3.re * w.im + e3.im * w.re;
            const Cx<T> a0{e0.re + v1r, e0.im + v1i}, a1{e0.re - v1r, e0.im - v1i};
            const Cx<T> a2{e2.re + v3r, e2.im + v3i}, a3{e2.re - v3r, e2.im - v3i};
            // stage 2h: pairs (a0, a2) with W_{4h}^j and (a1, a3) with W_{4h}^{j+h} = tw[(j + h) * st2]
            const Cx<T> t1 = tw[(j + h) * st2];
            const T c2r = a2.re * t0.re - a2.im * t0.im, c2i = a2.re * t0.im + a2.im * t0.re;
            const T c3r = a3.re * t1.re - a3.im * t1.im, c3i = a3.re * t1.im + a3.im * t1.re;
            buf[p] = Cx<T>{a0.re + c2r, a0.im + c2i};
            buf[p + 2 * h] = Cx<T>{a0.re - c2r, a0.im - c2i};
            buf[p + h] = Cx<T>{a1.re + c3r, a1.im + c3i};
            buf[p + 3 * h] = Cx<T>{a1.re - c3r, a1.im - c3i};
        }
        __syncthreads();
    }

    // 3. real split X[k] = E[k] + W_n^k O[k], k = 0..m, frame index fastest across threads
    for (unsigned idx = tid; idx < nf * (m + 1); idx += 256) {
        unsigned f = idx % nf, k = idx / nf;
        T re, im;
        if (k == 0 || k == m) {
            Cx<T> z = buf[f * fs];
            re = (k == 0) ? z.re + z.im : z.re - z.im;
            im = T(0);
        } else {
            Cx<T> z = buf[f * fs + k], y = buf[f * fs + m - k];
            const T half = T(0.5);
            T er = (z.re + y.re) * half, ei = (z.im - y.im) * half;
            T orr = (z.im + y.im) * half, oi = (y.re - z.re) * half;
            Cx<T> wv = tw[k];
            re = er + (orr * wv.re - oi * wv.im);
            im = ei + (orr * wv.im + oi * wv.re);
        }
        emit_bin<T>(a, b, f0 + f, f, k, re, im, pw, eps);
    }
    if (a.out_mode == OUT_MEL) {
        __syncthreads();
        mel_stage<T>(a, b, f0, nf, pw, eps, smem, (size_t)a.ft * fs * sizeof(Cx<T>));
    }
}

// ------------------------------------------------------------------------------------------------
// k_reg_radix: power-of-two n_fft from 32 to 8192, f32 and f64 — the register-tiled replacement of k_lds_radix2's
// log2(m) LDS stages.  The m = n_fft/2 = A * B * C point complex transform of z[n] = x[2n] + i x[2n+1] (C = 1 for
// m <= 256) is split as n = (B C) n1 + r, r = C n2 + n3, k = k1 + A (k2 + B k3):
//   pass 1  work item (frame, r): loads the A points n1 (windowed; lanes over r read contiguous samples), an A-point FFT
//           in registers (compile-time twiddles), times W_m^(k1 r), to row k1 / position r of the frame's LDS tile
//   pass 2  work item (frame, k1, n3): the B points n2 of its row, B-point FFT, times W_(BC)^(k2 n3), back in place
//   pass 3  work item (frame, k1, k2): the C points n3, C-point FFT, back in place (Z[k] at row k1, position C k2 + k3)
//           — passes 2 and 3 only touch elements the item owns, so each needs just the barrier that orders the passes
//   split   X[k] = E[k] + W_n^k O[k] from Z[k], Z[m-k]; frame index fastest across lanes (contiguous stores along the
//           frame axis, S9), then the shared epilogue (emit_bin / mel_stage).
// 3-4 barriers per tile instead of log2(m)/2 + 2, and 2-3 LDS round trips per point instead of log2(m)/2 + 1; no
// in-register transform is longer than 16 points, so f32 stays under 128 VGPRs (4 waves per SIMD).
// A, B, C: lengths of the in-register passes (products of 2, 3, 5; C = 1: two passes), m = A B C
template <typename T, int A_, int B_, int C_, bool STAGED_>
__global__ __launch_bounds__(256, (rr_stft_waves<T, A_, B_, C_, STAGED_>())) void k_reg_radix(StftArgs a, unsigned total_tiles, unsigned csr_lds, unsigned band_lds, unsigned rot) {
    typedef typename PairOf<T>::type V;
    constexpr unsigned A = A_, B = B_, C = C_, BC = B * C, M = A * BC;
    constexpr unsigned NI = rr_items<B_, C_>();
    constexpr int LA = ct_log2_ceil(A), LB = ct_log2_ceil(B);
    constexpr bool P2 = ct_is_pow2(M);        // power-of-two n_fft: table indices wrap with a mask instead of a remainder
    // where element (k1, p = hi C + lo) of a frame lives in LDS: rr_layout.h (three-pass splits: XOR swizzle, no bank conflicts
    // in any pass; two passes: rows of B + 1).  index = k1 RS + (lane part ^ instruction-stream part), see the helpers below.
    typedef RrLayout<sizeof(V), A_, B_, C_> L;
    constexpr unsigned RS = L::RS, FS = L::FS;
    auto k1_mask = [](unsigned k1) { return L::k1_mask(k1); };
    auto hi_part = [](unsigned hi) { return L::hi_part(hi); };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    V *buf = (V *)smem;                          // [ft][FS]
    V *stw = buf + (size_t)a.ft * FS;            // [M/2 + 1] split twiddles W_n^k
    V *sw = stw + (M / 2 + 1);                   // [M] window pairs (w[2i], w[2i+1])
    // [sub][pws] (Mel only, 32-byte aligned for the 4-element row reads), then the bank: padded band table or CSR arrays.
    // sub = a.mel_sub frames: the tile's |X|^2 rows are produced and reduced `sub` frames at a time, so that they do not
    // limit the frames per tile (f32 n_fft 400 / Mel: 32 frames per tile instead of 16 — every thread has a pass-1 item)
    const unsigned sub = a.mel_sub ? a.mel_sub : a.ft;
    T *pw = (T *)(smem + ((((size_t)a.ft * FS + M / 2 + 1 + M) * sizeof(V) + 31) & ~size_t(31)));
    const unsigned pws = 4u * (((a.nb_fft + 3u) >> 2) | 1u);  // row stride: 4-element groups, an odd number of them
    const V *tw = (const V *)a.tw;
    const T eps = (T)a.eps;
    const unsigned tid = threadIdx.x;
    // Tables the tile loop needs come from LDS: a global load behind the split's stores would wait for all of them (loads
    // and stores retire through one in-order counter).  The workgroup is persistent: tiles blockIdx.x, + gridDim.x, ...
    for (unsigned i = tid; i <= M / 2; i += 256) stw[i] = tw[i];
    for (unsigned i = tid; i < M; i += 256) sw[i] = ((const V *)a.window)[i] * (V){T(0.5), T(0.5)};  // halved (exact): the real split needs no 1/2
    MelCsr<T> csr{};
    typedef T V4 __attribute__((ext_vector_type(4)));
    const V4 *bw = nullptr;            // band table: 4 weights per group; row mm = groups [bptr[mm], bptr[mm+1]) from column bcol[mm]
    const unsigned *bptr = nullptr, *bcol = nullptr;
    if (a.out_mode == OUT_MEL) {
        unsigned char *bank = (unsigned char *)(pw + (size_t)sub * pws);
        if (band_lds) {  // uniform: rows are runs of consecutive columns and the padded table fits
            V4 *lw = (V4 *)bank;
            unsigned *lp = (unsigned *)(lw + a.mel_pchunks), *lc = lp + a.n_mels + 1;
            for (unsigned i = tid; i < a.mel_pchunks; i += 256) lw[i] = ((const V4 *)a.mel_pw)[i];
            for (unsigned i = tid; i <= a.n_mels; i += 256) lp[i] = a.mel_pptr[i];
            for (unsigned i = tid; i < a.n_mels; i += 256) lc[i] = a.mel_pcol[i];
            // columns nb_fft .. pws-1 of every row meet zero weights only, but must hold finite values
            for (unsigned i = tid; i < sub * (pws - a.nb_fft); i += 256)
                pw[(size_t)(i / (pws - a.nb_fft)) * pws + a.nb_fft + i % (pws - a.nb_fft)] = T(0);
            bw = lw; bptr = lp; bcol = lc;
        } else {
            csr = mel_resolve<T>(a, bank, csr_lds);
        }
    }
    // pass-1 work items of this thread (the geometry keeps ft * BC <= 256 NI): frames p1f + j * (256 / BC) of the tile,
    // residue r — the same for every tile, so the twiddles W_m^(k1 r) = W_n^(2 k1 r) are built once: one table gather per
    // bit of k1 (W^r, W^2r, W^4r, ...), products for the rest (at most LA - 1 roundings on top of the table's)
    const unsigned p1f = tid / BC, r = tid % BC;
    const unsigned p1pos = hi_part(r / C) ^ (r % C);  // position of point r in a row, before the row's own mask
    V pw2[LA];
#pragma unroll
    for (int j = 0; j < LA; ++j) pw2[j] = tw[P2 ? (((2u << j) * r) & (a.n_fft - 1)) : (((2u << j) * r) % a.n_fft)];
    // pass-2 twiddles W_(BC)^(k2 n3) = W_n^(2 A k2 n3), same bit-wise construction.  A thread's n3 is the same for every work
    // item it takes (256 is a multiple of A C for the power-of-two splits), so the gathers are done once here: inside the tile
    // loop they would sit behind the previous tile's stores (one in-order counter) and wait for all of them.
    constexpr bool Q2_FIXED = C > 1 && 256u % (A * C) == 0;
    // instances short of registers (f64; f32 with 16-point passes in three-pass splits) rebuild the twiddle products per tile
#if SGX_RR_TWKEEP  // experiment: let the f64 8-point instances carry their twiddle products through the tile loop
    constexpr bool TW_OPAQUE = !(sizeof(T) == 8 && A <= 8);
#else
    constexpr bool TW_OPAQUE = true;
#endif
    V q2f[LB];
    if constexpr (Q2_FIXED) {
        const unsigned n3 = (tid % (A * C)) % C;
#pragma unroll
        for (int j = 0; j < LB; ++j) q2f[j] = tw[P2 ? (((2u * A << j) * n3) & (a.n_fft - 1)) : (((2u * A << j) * n3) % a.n_fft)];
    }
    // Staged samples (the default): the tile's (ft - 1) hop + n_fft samples are loaded once, as 16-byte chunks (chunk c = tid +
    // 256 i) one tile ahead into registers, written over the (dead) tile buffer at the top of the tile and picked up from there
    // by the pass-1 items — instead of every frame loading its own copy of the overlapping samples (n_fft / hop loads per
    // sample, issued after pass 1 and waited for two passes later: measured 40 of 209 us for f32 n_fft 512).  Buffer loads: the
    // hardware returns 0 outside the signal's row (S1: zero padding).
    constexpr bool STAGED = STAGED_;
    constexpr unsigned EPC = 16 / sizeof(T), R_MAX = rr_stage_rounds(M, sizeof(T));  // elements per chunk, chunk rounds per thread
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const unsigned xs_len = (a.ft - 1) * a.hop + a.n_fft, chunks = (xs_len + EPC - 1) / EPC;  // host: chunks <= 256 R_MAX, fits the tile buffer
    v4u creg[R_MAX];
    auto load_tile = [&](unsigned t) {
        const unsigned tile = t % a.tiles, b = t / a.tiles;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc((const T *)a.x + (size_t)b * a.sample_stride, (unsigned)a.n_samples * (unsigned)sizeof(T));
        const int lo = (int)(tile * a.ft * a.hop) - (int)a.pad;  // host: n_samples * sizeof(T) < 2^31
        const bool interior = lo >= 0 && (unsigned long long)lo + (unsigned long long)chunks * EPC <= a.n_samples;  // uniform
        const int vo = (lo + (int)(EPC * tid)) * (int)sizeof(T);
#pragma unroll
        for (unsigned i = 0; i < R_MAX; ++i) {
#ifdef SGX_ABL_NOGLOAD
            creg[i] = (v4u){t, i, 0u, 0u};
            continue;
#endif
            if (i * 256u + tid >= chunks) continue;
            if (interior) {
                creg[i] = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rx, vo + (int)(i * 4096u), 0, 0));
            } else {  // a chunk may straddle an end of the row: one access per sample, each checked on its own (offsets may be negative)
                T e[EPC];
#pragma unroll
                for (unsigned q = 0; q < EPC; ++q) e[q] = BufLd<T>::template one<true>(rx, vo + (int)(i * 4096u + q * sizeof(T)));
                creg[i] = __builtin_bit_cast(v4u, e);
            }
        }
    };
    // Direct loads (SGX_RR_STAGED=0, kept for A/B): raw (unwindowed) samples of one tile's work items -> registers, one tile
    // ahead; frames past the end of a signal's last tile load (zeros or the row's tail) and are ignored.
    constexpr int kPair = 2 * BC * (int)sizeof(T);  // bytes from point n1 to point n1 + 1
    auto load_raw = [&](unsigned t, V (&raw)[NI][A]) {
#ifdef SGX_ABL_NOGLOAD
        for (unsigned j = 0; j < NI; ++j)
            for (unsigned n1 = 0; n1 < A; ++n1) raw[j][n1] = (V){T(t), T(n1)};
        return;
#endif
        const unsigned tile = t % a.tiles, b = t / a.tiles;
        const unsigned f0 = tile * a.ft;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc((const T *)a.x + (size_t)b * a.sample_stride, (unsigned)a.n_samples * (unsigned)sizeof(T));
        const int lo = (int)(f0 * a.hop) - (int)a.pad;  // host: n_samples * sizeof(T) < 2^31
        // all a.ft frames of the tile (existing or not: every lane loads) lie inside the row — uniform
        const bool interior = lo >= 0 && (unsigned long long)lo + (unsigned long long)(a.ft - 1) * a.hop + a.n_fft <= a.n_samples;
#pragma unroll
        for (unsigned j = 0; j < NI; ++j) {
            const int vo = (lo + (int)((p1f + j * (256u / BC)) * a.hop + 2u * r)) * (int)sizeof(T);
            if (interior) {  // every frame of the tile lies inside the row: one access per point, the step in the unchecked scalar offset
#pragma unroll
                for (unsigned n1 = 0; n1 < A; ++n1) raw[j][n1] = BufLd<T>::pair(rx, vo, (int)n1 * kPair);
            } else {  // a pair may straddle an end of the row: two accesses, each checked on its own (offsets may be negative: one<true>)
#pragma unroll
                for (unsigned n1 = 0; n1 < A; ++n1)
                    raw[j][n1] = (V){BufLd<T>::template one<true>(rx, vo + (int)n1 * kPair), BufLd<T>::template one<true>(rx, vo + (int)n1 * kPair + (int)sizeof(T))};
            }
        }
    };

    V raw[NI][A];
    if constexpr (!STAGED) {
#pragma unroll
        for (unsigned j = 0; j < NI; ++j)
#pragma unroll
            for (unsigned n1 = 0; n1 < A; ++n1) raw[j][n1] = (V){T(0), T(0)};
    } else {
#pragma unroll
        for (unsigned i = 0; i < R_MAX; ++i) creg[i] = (v4u){0u, 0u, 0u, 0u};
    }
    if constexpr (TW_OPAQUE) {  // landed before the loop: inside it nothing but the sample prefetch is ever waited for
#pragma unroll
        for (int j = 0; j < LA; ++j) asm volatile("" : "+v"(pw2[j]));
        if constexpr (Q2_FIXED) {
#pragma unroll
            for (int j = 0; j < LB; ++j) asm volatile("" : "+v"(q2f[j]));
        }
    }
    // Tile order.  Round k covers tiles [k grid, (k + 1) grid); inside a round the workgroups of one XCD (blockIdx.x mod 8: the
    // dispatcher deals workgroups to the XCDs in turn) take one contiguous eighth of it.  Neighbouring tiles of a signal write
    // neighbouring ft-frame pieces of the same output rows — halves or quarters of the same 128-byte lines: in one XCD they
    // meet in its L2 and leave as whole lines, spread over the XCDs every L2 writes its piece on its own (measured on the f64
    // n_fft 1024 tiles of 8 frames = 64 bytes: 58 % of the write requests to memory were 32-byte ones).  Each round the
    // positions move on by `rot`, chosen by the host so that a workgroup meets every tile index of a signal in turn (the first
    // and last tile of a signal are the slower, bounds-checked ones: without that they pile up on a few workgroups).
    // rot = ~0: plain order (grids that are not a multiple of 8: a single round of a small problem).
    const unsigned grid = gridDim.x;
    const unsigned pos0 = rot == ~0u ? blockIdx.x : (blockIdx.x & 7u) * (grid >> 3) + (blockIdx.x >> 3);
    auto tile_of = [&](unsigned k) { return k * grid + (rot == ~0u ? pos0 : (pos0 + k * rot) % grid); };  // >= total_tiles: nothing this round
    // the first tile's samples have landed before the loop: with loads still pending on entry their first use inside the loop
    // gets a vmcnt(0), which from the second tile on waits for the previous tile's stores
    if constexpr (STAGED) {
        load_tile(min(tile_of(0), total_tiles - 1));
#pragma unroll
        for (unsigned i = 0; i < R_MAX; ++i) asm volatile("" : "+v"(creg[i]));
    } else {
        load_raw(min(tile_of(0), total_tiles - 1), raw);
#pragma unroll
        for (unsigned j = 0; j < NI; ++j)
#pragma unroll
            for (unsigned n1 = 0; n1 < A; ++n1) asm volatile("" : "+v"(raw[j][n1]));
    }
    __syncthreads();
#if SGX_RR_STAGGER  // experiment (as k_r32x16's SGX_STAGGER): the workgroups of an XCD start a few hundred cycles apart, so that the
                    // CUs' store bursts and arithmetic phases do not coincide
    if (a.out_mode != OUT_MEL) {
        for (unsigned q = 0; q < (blockIdx.x >> 3) * SGX_RR_STAGGER; ++q) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
    }
#endif
    const unsigned lft = __ffs(a.ft) - 1u;
#ifdef SGX_RR_STAMPS
    unsigned long long st_acc[16] = {0}, st_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
    for (unsigned rnd = 0;; ++rnd) {
        const unsigned t = tile_of(rnd);
        if (t >= total_tiles) break;  // only the last round is partial
        const unsigned t_next = tile_of(rnd + 1);
        const unsigned tile = t % a.tiles, b = t / a.tiles;
        const unsigned f0 = tile * a.ft, nf = min(a.ft, a.n_frames - f0);
        // the twiddle powers are opaque per tile: their products (rr_twiddle) are then rebuilt where they are used instead of
        // being carried through the whole loop in registers
        if constexpr (TW_OPAQUE) {
#pragma unroll
            for (int j = 0; j < LA; ++j) asm volatile("" : "+v"(pw2[j]));
            if constexpr (Q2_FIXED) {
#pragma unroll
                for (int j = 0; j < LB; ++j) asm volatile("" : "+v"(q2f[j]));
            }
        }
        auto pass1_store = [&](unsigned f, V (&v)[A]) {  // times W_m^(k1 r), to row k1 of frame f
            V *dst = buf + (size_t)f * FS;
            unsigned pp = p1pos;
            if constexpr (C > 1) asm volatile("" : "+v"(pp));  // the A swizzled addresses are rebuilt here (one XOR each), not kept across the loop
            dst[pp ^ k1_mask(0)] = v[0];
#pragma unroll
            for (unsigned k1 = 1; k1 < A; ++k1) (dst + (pp ^ k1_mask(k1)))[k1 * RS] = inreg::cmulv(v[k1], rr_twiddle<LA>(pw2, k1));  // k1 RS: an immediate offset
        };
        if constexpr (STAGED) {
            // samples over the tile buffer (free since the previous tile's last barrier), then every item reads its points and
            // transforms them in registers; the results go back to the buffer once every wave has read (r32x16's barrier 2)
#pragma unroll
            for (unsigned i = 0; i < R_MAX; ++i)
                if (i * 256u + tid < chunks) ((v4u *)smem)[i * 256u + tid] = creg[i];
#if SGX_RR_LOADPOS == 0
            if (t_next < total_tiles) load_tile(t_next);  // a whole tile ahead of its use
#endif
            RR_STAMP(0);  // staging writes
            __syncthreads();
            RR_STAMP(1);  // barrier
            V v[NI][A];
            const T *xs = (const T *)smem;
#pragma unroll
            for (unsigned j = 0; j < NI; ++j) {
                const unsigned f = p1f + j * (256u / BC);
                if (f >= nf) continue;
                const T *xp = xs + f * a.hop + 2u * r;
                if (!(a.hop & 1u)) {  // even hop: every frame starts on a pair boundary of the tile
#pragma unroll
                    for (unsigned n1 = 0; n1 < A; ++n1) v[j][n1] = *(const V *)(xp + 2u * BC * n1) * sw[BC * n1 + r];
                } else {
#pragma unroll
                    for (unsigned n1 = 0; n1 < A; ++n1) v[j][n1] = (V){xp[2u * BC * n1], xp[2u * BC * n1 + 1u]} * sw[BC * n1 + r];
                }
                inreg::MixFft<A, V>::run(v[j]);
            }
            RR_STAMP(2);  // pass 1: sample / window reads + transform
            __syncthreads();
            RR_STAMP(3);  // barrier
#pragma unroll
            for (unsigned j = 0; j < NI; ++j) {
                const unsigned f = p1f + j * (256u / BC);
                if (f < nf) pass1_store(f, v[j]);
            }
            RR_STAMP(4);  // pass 1: twiddles + tile writes
#if SGX_RR_LOADPOS != 0
            if (t_next < total_tiles) load_tile(t_next);  // in flight behind passes 2 and 3
#endif
            RR_STAMP(5);  // load issue
        } else {
#pragma unroll
            for (unsigned j = 0; j < NI; ++j) {
                const unsigned f = p1f + j * (256u / BC);
                if (f >= nf) continue;
                V v[A];
#pragma unroll
                for (unsigned n1 = 0; n1 < A; ++n1) v[n1] = raw[j][n1] * sw[BC * n1 + r];
                inreg::MixFft<A, V>::run(v);
                pass1_store(f, v);
            }
            RR_STAMP(2);  // direct: pass 1 complete (window reads, transform, twiddles, tile writes)
            if (t_next < total_tiles) load_raw(t_next, raw);  // in flight behind passes 2 and 3
            RR_STAMP(5);  // load issue
        }
        __syncthreads();
        RR_STAMP(6);  // barrier
        // the thread index is opaque from here on: the element addresses of passes 2, 3 and the split are recomputed per tile
        // (a few integer operations) instead of being carried through the whole loop in registers
        unsigned tl = tid;
        asm volatile("" : "+v"(tl));
        for (unsigned idx = tl; idx < nf * A * C; idx += 256) {
            const unsigned f = idx / (A * C), q = idx % (A * C), k1 = q / C, n3 = q % C;
            V *row = buf + (size_t)f * FS + k1 * RS;
            const unsigned lp = n3 ^ k1_mask(k1);  // lane part; point n2 sits at lp ^ hi_part(n2)
            V x[B];
#pragma unroll
            for (unsigned n2 = 0; n2 < B; ++n2) x[n2] = row[lp ^ hi_part(n2)];
            inreg::MixFft<B, V>::run(x);
            row[lp ^ hi_part(0)] = x[0];
            if constexpr (C > 1) {
                V q2[LB];
#pragma unroll
                for (int j = 0; j < LB; ++j) {
                    if constexpr (Q2_FIXED) q2[j] = q2f[j];
                    else q2[j] = tw[P2 ? (((2u * A << j) * n3) & (a.n_fft - 1)) : (((2u * A << j) * n3) % a.n_fft)];
                }
#pragma unroll
                for (unsigned k2 = 1; k2 < B; ++k2) row[lp ^ hi_part(k2)] = inreg::cmulv(x[k2], rr_twiddle<LB>(q2, k2));
            } else {
#pragma unroll
                for (unsigned k2 = 1; k2 < B; ++k2) row[lp ^ hi_part(k2)] = x[k2];
            }
        }
        RR_STAMP(7);  // pass 2
        __syncthreads();
        RR_STAMP(8);  // barrier
        if constexpr (C > 1) {
            for (unsigned idx = tl; idx < nf * A * B; idx += 256) {
                const unsigned f = idx / (A * B), q = idx % (A * B), k1 = q / B, k2 = q % B;
                V *row = buf + (size_t)f * FS + k1 * RS;
                const unsigned lp = hi_part(k2) ^ k1_mask(k1);  // lane part; point n3 sits at lp ^ n3
                V x[C];
#pragma unroll
                for (unsigned n3 = 0; n3 < C; ++n3) x[n3] = row[lp ^ n3];
                inreg::MixFft<C, V>::run(x);
#pragma unroll
                for (unsigned k3 = 0; k3 < C; ++k3) row[lp ^ k3] = x[k3];
            }
            RR_STAMP(9);  // pass 3
            __syncthreads();
            RR_STAMP(10);  // barrier
        }
        // The next tile's samples have had passes 2 and 3 to arrive: collect them here, before the first store of the split,
        // so that nothing issued after this point ever has to be waited for.
        if constexpr (STAGED) {
#pragma unroll
            for (unsigned i = 0; i < R_MAX; ++i) asm volatile("" : "+v"(creg[i]));
        } else {
#pragma unroll
            for (unsigned j = 0; j < NI; ++j)
#pragma unroll
                for (unsigned n1 = 0; n1 < A; ++n1) asm volatile("" : "+v"(raw[j][n1]));
        }
        RR_STAMP(11);  // wait for the next tile's samples
        // real split, frame index fastest across threads (a.ft is a power of two).  One work item per pair (k, m - k):
        // with E = (Z[k] + conj Z[m-k]) / 2, P = W_n^k (Z[k] - conj Z[m-k]) / (2i):  X[k] = E + P,  X[m-k] = conj(E - P).
        auto at = [](const V *fb, unsigned k) -> V { return fb[L::of_output(k)]; };  // Z[k]: row k mod A, position (hi, lo) with k / A = hi + B lo
        {
            // Packed arithmetic on the pre-halved spectrum (window x 1/2 above): E = Z[k] + conj Z[m-k], D = (z.x - y.x, z.y + y.y),
            // T = W (-i D) = D.y W + D.x (W.y, -W.x);  X[k] = E + T, X[m-k] = conj(E - T).  The output mode and the amplitude
            // scale are uniform: they pick one of five specialised loops instead of being tested per bin.
            // a thread keeps its frame f and walks the bins k0, k0 + kstep, ...
            auto split_frame = [&](unsigned f, unsigned k0, unsigned kstep, auto &&put) {  // put(k, X): X[k] of frame f
                if (f >= nf) return;
                const V *fb = buf + (size_t)f * FS;
                // k = 0 takes the same path with Z[m] := Z[0]: E = (2 z.x, 0), D = (0, 2 z.y), W^0 = (1, 0) give X[0] = (2 (z.x + z.y), 0)
                // and X[m] = (2 (z.x - z.y), 0) with the same roundings as the closed form (the DC and Nyquist bins, exactly real);
                // two bins per trip so that the LDS reads of one overlap the arithmetic of the other
#pragma unroll 2
                for (unsigned k = k0; k <= M / 2; k += kstep) {
                    const V z = at(fb, k), y = at(fb, k ? M - k : 0u), w = stw[k];
                    const V E = inreg::pfma(y, (V){T(1), T(-1)}, z);
                    const V D = inreg::pfma(y, (V){T(-1), T(1)}, z);
                    const V Tt = inreg::pfma(inreg::hi2(D), w, inreg::lo2(D) * (V){w.y, -w.x});
                    V X = E + Tt, Y = E - Tt;
                    if (k == 0) X.y = Y.y = T(0);  // (+0, not the -0 the general path can leave)
                    put(k, X);
                    if (k != M - k) put(M - k, (V){Y.x, -Y.y});
                }
            };
            if (a.out_mode != OUT_MEL) {
                const unsigned f = tl & (a.ft - 1), k0 = tl >> lft, kstep = 256u >> lft;
                const size_t ob = ((size_t)b * a.n_out) * a.n_frames + f0 + f;
                if (a.out_mode == OUT_COMPLEX) {
                    V *o = (V *)a.out + ob;
                    split_frame(f, k0, kstep, [&](unsigned k, V X) { o[(size_t)k * a.n_frames] = X; });
                } else {
                    T *o = (T *)a.out + ob;
                    if (a.amp == AMP_MAGNITUDE) split_frame(f, k0, kstep, [&](unsigned k, V X) { o[(size_t)k * a.n_frames] = t_sqrt(X.x * X.x + X.y * X.y); });
                    else if (a.amp == AMP_DB) split_frame(f, k0, kstep, [&](unsigned k, V X) { o[(size_t)k * a.n_frames] = t_db(t_max(X.x * X.x + X.y * X.y, eps)); });
                    else
#ifdef SGX_ABL_NOSTORE
                        split_frame(f, k0, kstep, [&](unsigned k, V X) { if (X.x == T(1.2345e30)) o[(size_t)k * a.n_frames] = X.x * X.x + X.y * X.y; });
#else
                        split_frame(f, k0, kstep, [&](unsigned k, V X) { o[(size_t)k * a.n_frames] = X.x * X.x + X.y * X.y; });
#endif
                }
            } else {
                // filterbank outputs, `sub` frames at a time: |X|^2 (or |X|) rows to LDS, then the bank rows: thread = (frame, row)
                // with the frame fastest; sequential un-fused accumulation in ascending column order (:102-117).  Rows that are
                // one run of consecutive columns need no column look-up per term, so the LDS reads of a row are independent of
                // each other and pipeline.
                const unsigned lsub = __ffs(sub) - 1u, fl = tl & (sub - 1), q0 = tl >> lsub, qstep = 256u >> lsub;
                T *pf = pw + (size_t)fl * pws;
                for (unsigned part = 0; part < nf; part += sub) {
                    const unsigned f = part + fl;
                    if (a.amp == AMP_MAG_IN) split_frame(f, q0, qstep, [&](unsigned k, V X) { pf[k] = t_sqrt(X.x * X.x + X.y * X.y); });
                    else split_frame(f, q0, qstep, [&](unsigned k, V X) { pf[k] = X.x * X.x + X.y * X.y; });
                    __syncthreads();
                    RR_STAMP(14);  // filterbank outputs: split + |X|^2 rows + barrier (12 is then the bank stage alone)
                    T *o = (T *)a.out + ((size_t)b * a.n_out) * a.n_frames + f0 + f;
                    if (f < nf)
                        for (unsigned mm = q0; mm < a.n_mels; mm += qstep) {
                            T acc = T(0);
                            if (bw) {  // 4 columns per step: one vector read of the weights, one of the row (zero weights pad the run)
                                const unsigned c0 = bptr[mm], c1 = bptr[mm + 1];
                                const V4 *xq = (const V4 *)(pf + bcol[mm]) - c0;
                                for (unsigned c = c0; c < c1; ++c) {
                                    const V4 wq = bw[c], x = xq[c];
                                    acc = t_mul_add_unfused(wq.x, x.x, acc);
                                    acc = t_mul_add_unfused(wq.y, x.y, acc);
                                    acc = t_mul_add_unfused(wq.z, x.z, acc);
                                    acc = t_mul_add_unfused(wq.w, x.w, acc);
                                }
                            } else {
                                const unsigned i0 = csr.ptr[mm], i1 = csr.ptr[mm + 1];
                                for (unsigned i = i0; i < i1; ++i) acc = t_mul_add_unfused(csr.val[i], pf[csr.col[i]], acc);
                            }
                            o[(size_t)mm * a.n_frames] = amp_apply(acc, a.amp, eps);
                        }
                    if (part + sub < nf) __syncthreads();  // the rows are rewritten by the next part
                }
            }
        }
        RR_STAMP(12);  // split + stores (+ filterbank stage)
        __syncthreads();  // the tile buffer (and pw) is free for the next tile
        RR_STAMP(13);  // barrier
#ifdef SGX_RR_STAMPS
        st_acc[15] += 1;
#endif
    }
#ifdef SGX_RR_STAMPS
    if ((tid & 63u) == 0) {
        for (int q = 0; q < 16; ++q) atomicAdd(&g_rr_stamps[q], st_acc[q]);
        atomicAdd(&g_rr_stamps[16], 1ull);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_direct_dft(StftArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned n = a.n_fft;
    T *fr = (T *)smem;                  // [ft][n_fft]
    T *pw = fr + (size_t)a.ft * n;      // [ft][nb_fft] (Mel only)
    const unsigned tile = blockIdx.x % a.tiles, b = blockIdx.x / a.tiles;
    const unsigned f0 = tile * a.ft;
    const unsigned nf = min(a.ft, a.n_frames - f0);
    const T *xb = (const T *)a.x + (size_t)b * a.sample_stride;
    const T *w = (const T *)a.window;
    const Cx<T> *tw = (const Cx<T> *)a.tw;
    const T eps = (T)a.eps;

    for (unsigned idx = threadIdx.x; idx < nf * n; idx += 256) {
        unsigned f = idx / n, i = idx % n;
        long long s = (long long)(f0 + f) * a.hop + (long long)i - (long long)a.pad;
        fr[(size_t)f * n + i] = load_sample(xb, s, a.n_samples) * w[i];
    }
    __syncthreads();
    for (unsigned idx = threadIdx.x; idx < nf * a.nb_fft; idx += 256) {
        unsigned f = idx % nf, k = idx / nf;
        const T *x = fr + (size_t)f * n;
        T sr = T(0), si = T(0);
        unsigned t = 0;
        for (unsigned j = 0; j < n; j++) {
            Cx<T> c = tw[t];
            sr += x[j] * c.re;
            si += x[j] * c.im;
            t += k;
            if (t >= n) t -= n;
        }
        if (k == 0 || (!(n & 1u) && k == a.nb_fft - 1)) si = T(0);  // realfft: DC / Nyquist bins are exactly real
        emit_bin<T>(a, b, f0 + f, f, k, sr, si, pw, eps);
    }
    if (a.out_mode == OUT_MEL) {
        __syncthreads();
        mel_stage<T>(a, b, f0, nf, pw, eps, smem, (size_t)a.ft * n * sizeof(T));
    }
}

// ------------------------------------------------------------------------------------------------
// k_two_factor: composite n_fft that is not a power of two (400 = 20 x 20, the classic 25 ms speech frame; 480, 1000, ...).
// n = A * B, A the largest divisor <= sqrt(n).  With input index i = B n1 + n2 and output index k = k1 + A k2:
//   pass 1  Y[n2][k1] = W_n^(n2 k1) * sum_{n1 < A} x[B n1 + n2] W_A^(n1 k1)          (n outputs of A real x complex MACs)
//   pass 2  X[k1 + A k2] = sum_{n2 < B} Y[n2][k1] W_B^(n2 k2), only for k <= n/2      (n/2+1 outputs of B complex MACs)
// i.e. n (A + B/2)-ish MACs per frame instead of the direct sum's n^2/2 (400: 12 k vs 80 k), every twiddle taken from the
// same n-entry table W_n^m with the exponent reduced mod n incrementally.  Sums are shorter than the direct kernel's, so the
// result is at least as accurate; DC / Nyquist bins are forced real like realfft does.
template <typename T>
__global__ __launch_bounds__(256) void k_two_factor(StftArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned n = a.n_fft, A = a.fac_a, B = a.fac_b;
    T *fr = (T *)smem;                                   // [ft][n] windowed frames
    Cx<T> *Y = (Cx<T> *)(fr + (size_t)a.ft * n);         // [ft][n]  (n2 * A + k1)
    Cx<T> *ltw = Y + (size_t)a.ft * n;                   // [n] twiddles W_n^m: every MAC below gathers one
    T *pw = (T *)(ltw + n);                              // [ft][nb_fft] (Mel only)
    const unsigned tile = blockIdx.x % a.tiles, b = blockIdx.x / a.tiles;
    const unsigned f0 = tile * a.ft;
    const unsigned nf = min(a.ft, a.n_frames - f0);
    const T *xb = (const T *)a.x + (size_t)b * a.sample_stride;
    const T *w = (const T *)a.window;
    const T eps = (T)a.eps;
    // (a frame too long for the LDS copy — per-bin outputs only, a.tw_lds == 0 — gathers the twiddles from the table in memory)
    if (a.tw_lds)
        for (unsigned i = threadIdx.x; i < n; i += 256) ltw[i] = ((const Cx<T> *)a.tw)[i];
    const Cx<T> *tw = a.tw_lds ? ltw : (const Cx<T> *)a.tw;
    for (unsigned idx = threadIdx.x; idx < nf * n; idx += 256) {
        const unsigned f = idx / n, i = idx % n;
        const long long s = (long long)(f0 + f) * a.hop + (long long)i - (long long)a.pad;
        fr[(size_t)f * n + i] = load_sample(xb, s, a.n_samples) * w[i];
    }
    __syncthreads();
    for (unsigned idx = threadIdx.x; idx < nf * n; idx += 256) {
        const unsigned f = idx / n, r = idx % n, n2 = r / A, k1 = r % A;
        const T *x = fr + (size_t)f * n + n2;
        const unsigned step = (unsigned)(((unsigned long long)B * k1) % n);
        T sr = T(0), si = T(0);
        unsigned t = 0;
        for (unsigned n1 = 0; n1 < A; ++n1) {
            const Cx<T> c = tw[t];
            const T v = x[(size_t)B * n1];
            sr += v * c.re;
            si += v * c.im;
            t += step;
            if (t >= n) t -= n;
        }
        const Cx<T> g = tw[(unsigned)(((unsigned long long)n2 * k1) % n)];
        Y[(size_t)f * n + r] = Cx<T>{sr * g.re - si * g.im, sr * g.im + si * g.re};
    }
    __syncthreads();
    for (unsigned idx = threadIdx.x; idx < nf * a.nb_fft; idx += 256) {
        const unsigned f = idx % nf, k = idx / nf, k1 = k % A, k2 = k / A;
        const Cx<T> *y = Y + (size_t)f * n + k1;
        const unsigned step = (unsigned)(((unsigned long long)A * k2) % n);
        T sr = T(0), si = T(0);
        unsigned t = 0;
        for (unsigned n2 = 0; n2 < B; ++n2) {
            const Cx<T> c = tw[t], v = y[(size_t)A * n2];
            sr += v.re * c.re - v.im * c.im;
            si += v.re * c.im + v.im * c.re;
            t += step;
            if (t >= n) t -= n;
        }
        if (k == 0 || (!(n & 1u) && k == a.nb_fft - 1)) si = T(0);  // realfft: DC / Nyquist bins are exactly real
        emit_bin<T>(a, b, f0 + f, f, k, sr, si, pw, eps);
    }
    if (a.out_mode == OUT_MEL) {
        __syncthreads();
        mel_stage<T>(a, b, f0, nf, pw, eps, smem, (size_t)a.ft * n * 3 * sizeof(T));  // (frames and Y; the twiddles are done too)
    }
}

// ------------------------------------------------------------------------------------------------
// MFCC epilogue (src/mfcc.rs:224-316): one thread per (signal, frame); lanes walk frames so every read of the Mel-dB
// tensor and every write is frame-contiguous.  DCT-II as a sequential FMA chain in T over ascending mel index, exactly
// `val.mul_add(basis, acc)` (:286-290); the basis is cos(pi k (i+0.5)/n) evaluated in f64 on the host and cast to T.
template <typename T>
__global__ __launch_bounds__(256) void k_mfcc(const T *mel, T *out, const T *basis, const T *lifter, unsigned batch,
                                              unsigned n_mels, unsigned n_frames, unsigned n_mfcc, unsigned skip, int has_lifter) {
    const unsigned long long gid = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (unsigned long long)batch * n_frames) return;
    const unsigned b = (unsigned)(gid / n_frames), f = (unsigned)(gid - (unsigned long long)b * n_frames);
    const T *m = mel + (size_t)b * n_mels * n_frames + f;
    T *o = out + (size_t)b * (n_mfcc - skip) * n_frames + f;
    for (unsigned k = skip; k < n_mfcc; ++k) {
        T acc = T(0);
        const T *bk = basis + (size_t)k * n_mels;
        for (unsigned i = 0; i < n_mels; ++i) acc = fma(m[(size_t)i * n_frames], bk[i], acc);
        if (has_lifter) acc *= lifter[k];
        o[(size_t)(k - skip) * n_frames] = acc;
    }
}

// Same arithmetic with the loops swapped: a lane keeps NC accumulators (one per coefficient) and walks the mel bands once,
// so every Mel value is read from memory ONCE (coalesced across the lanes' frames) instead of once per coefficient, and the
// basis sits in LDS as [band][coefficient] (16-byte broadcast reads).  Each accumulator still sees its own FMA chain in
// ascending band order, so the results are bit-identical to k_mfcc / the reference.  Measured (256 x 10 s, 13 of 80):
// 300 us -> see DESIGN.md.
template <typename T, int NC>
__global__ __launch_bounds__(256) void k_mfcc_acc(const T *mel, T *out, const T *basis, const T *lifter, unsigned batch,
                                                  unsigned n_mels, unsigned n_frames, unsigned n_mfcc, unsigned skip, int has_lifter) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *sb = (T *)smem;  // [n_mels][NC], zero beyond n_mfcc
    for (unsigned idx = threadIdx.x; idx < n_mels * NC; idx += 256) {
        const unsigned i = idx / NC, k = idx % NC;
        sb[idx] = k < n_mfcc ? basis[(size_t)k * n_mels + i] : T(0);
    }
    __syncthreads();
    const unsigned long long gid = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (unsigned long long)batch * n_frames) return;
    const unsigned b = (unsigned)(gid / n_frames), f = (unsigned)(gid - (unsigned long long)b * n_frames);
    const T *m = mel + (size_t)b * n_mels * n_frames + f;
    T *o = out + (size_t)b * (n_mfcc - skip) * n_frames + f;
    T acc[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) acc[k] = T(0);
    // 16 bands at a time: their Mel values are requested together (one memory round trip per 16 bands instead of one per band),
    // then folded in ascending band order as before
    for (unsigned i0 = 0; i0 < n_mels; i0 += 16) {
        T v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = i0 + u < n_mels ? m[(size_t)(i0 + u) * n_frames] : T(0);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (i0 + u < n_mels) {
                const T *bi = sb + (i0 + u) * NC;
#pragma unroll
                for (int k = 0; k < NC; ++k) acc[k] = fma(v[u], bi[k], acc[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        if ((unsigned)k >= skip && (unsigned)k < n_mfcc) {
            T r = acc[k];
            if (has_lifter) r *= lifter[k];
            o[(size_t)(k - skip) * n_frames] = r;
        }
    }
}

// chroma: out[b][12][n_frames] normalised per frame over the 12 rows, sums / squares accumulated in row order, unfused
template <typename T>
__global__ __launch_bounds__(256) void k_chroma_norm(T *data, unsigned n_frames, unsigned long long total, int norm) {
    const unsigned long long gid = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const unsigned b = (unsigned)(gid / n_frames), f = (unsigned)(gid - (unsigned long long)b * n_frames);
    T *col = data + (size_t)b * 12 * n_frames + f;
    T v[12];
    for (int i = 0; i < 12; ++i) v[i] = col[(size_t)i * n_frames];
    T d = T(0);
    if (norm == 1) {
        for (int i = 0; i < 12; ++i) d = t_mul_add_unfused(T(1), v[i], d);
    } else if (norm == 2) {
        for (int i = 0; i < 12; ++i) d = t_mul_add_unfused(v[i], v[i], d);
        d = t_sqrt(d);
    } else {
        for (int i = 0; i < 12; ++i) d = t_max(d, v[i]);
    }
    if (d > T(0))
        for (int i = 0; i < 12; ++i) col[(size_t)i * n_frames] = v[i] / d;
}

hipError_t launch_chroma_norm(void *data, unsigned batch, unsigned n_frames, int norm, int dtype, hipStream_t s) {
    if (norm == 0) return hipSuccess;
    const unsigned long long total = (unsigned long long)batch * n_frames, blocks = (total + 255) / 256;
    if (blocks == 0 || blocks >= 0x7fffffffull) return hipErrorInvalidConfiguration;
    if (dtype == SGX_F64) hipLaunchKernelGGL(k_chroma_norm<double>, dim3((unsigned)blocks), dim3(256), 0, s, (double *)data, n_frames, total, norm);
    else hipLaunchKernelGGL(k_chroma_norm<float>, dim3((unsigned)blocks), dim3(256), 0, s, (float *)data, n_frames, total, norm);
    return hipGetLastError();
}

// frames per tile are chosen to fill this much LDS.  32 KB (4-5 workgroups per CU) measured 5-20 % faster than 64 KB for
// the sizes in tools/time_generic.py: shorter store segments, but more waves to hide the LDS / barrier latency
static const size_t kLdsBudget = 32 * 1024;

// a single frame may take the large-LDS window (the launchers opt in above 64 KiB): with the 64 KiB default a 6000-point f32
// two-factor tile (120 KB) did not fit and the plan fell through to the O(n^2) direct sum — 320 ms per 64 x 10 s, slower than a CPU
static const size_t kLdsHardLimit = 144 * 1024;
template <typename K>
static hipError_t lds_opt_in(K kernel, size_t lds) {
    return lds > 64 * 1024 ? set_max_dynamic_lds((const void *)kernel, (int)kLdsHardLimit) : hipSuccess;  // (set once per kernel and device)
}

static size_t elem_size(int dtype) { return dtype == SGX_F64 ? 8 : 4; }

// frames per tile: the largest power of two <= 16 whose tile fits the preferred budget; a single frame may use the hard limit
template <typename F>
static bool pick_frames_per_tile(StftArgs &a, F bytes_for) {
    for (unsigned ft = 16; ft >= 1; ft >>= 1)
        if (bytes_for(ft) <= kLdsBudget) {
            a.ft = ft;
            return true;
        }
    if (bytes_for(1) <= kLdsHardLimit) {
        a.ft = 1;
        return true;
    }
    return false;
}

bool plan_geometry_lds_radix2(StftArgs &a, int dtype) {
    if (a.n_fft < 4 || (a.n_fft & (a.n_fft - 1))) return false;
    const size_t es = elem_size(dtype);
    auto tile_bytes = [&](unsigned ft) {
        return (size_t)ft * (a.m + 1) * 2 * es + (a.out_mode == OUT_MEL ? (size_t)ft * a.nb_fft * es : 0);
    };
    const size_t tw_bytes = (size_t)a.m * 2 * es;
    a.tw_lds = 0;
    if (!pick_frames_per_tile(a, tile_bytes)) return false;
    // LDS twiddle copy only where it costs no frames per tile (measured: +15 % for f32 n_fft 512, but slower wherever it
    // shrinks the tile: f32 2048, f64 1024)
    const size_t off = (tile_bytes(a.ft) + 15) & ~size_t(15);
    if (a.ft >= 8 && off + tw_bytes <= kLdsBudget) a.tw_lds = (unsigned)off;  // the per-tile copy needs >= 8 frames to pay off
    return true;
}

bool plan_geometry_direct_dft(StftArgs &a, int dtype) {
    const size_t es = elem_size(dtype);
    return pick_frames_per_tile(a, [&](unsigned ft) {
        return (size_t)ft * a.n_fft * es + (a.out_mode == OUT_MEL ? (size_t)ft * a.nb_fft * es : 0);
    });
}

static size_t two_factor_bytes(const StftArgs &a, unsigned ft, size_t es) {
    return (size_t)ft * a.n_fft * 3 * es + (size_t)a.n_fft * 2 * es + (a.out_mode == OUT_MEL ? (size_t)ft * a.nb_fft * es : 0);
}

bool plan_geometry_two_factor(StftArgs &a, int dtype) {
    unsigned best = 1;
    for (unsigned d = 2; (unsigned long long)d * d <= a.n_fft; ++d)
        if (a.n_fft % d == 0) best = d;
    if (best < 2) return false;  // prime (or tiny) length: the direct sum is all there is
    a.fac_a = best;
    a.fac_b = a.n_fft / best;
    const size_t es = elem_size(dtype);
    a.tw_lds = 1;  // twiddle table copied to LDS
    if (pick_frames_per_tile(a, [&](unsigned ft) { return two_factor_bytes(a, ft, es); })) return true;
    // one frame without the LDS twiddle copy (f64 n_fft 6000: 144 KB instead of 240 KB — the direct sum took 312 ms per 64 x 10 s)
    if (a.out_mode != OUT_MEL && (size_t)a.n_fft * 3 * es <= kLdsHardLimit) {
        a.ft = 1;
        a.tw_lds = 0;
        return true;
    }
    return false;
}

static bool grid_ok(const StftArgs &a, unsigned long long *blocks) {
    unsigned long long g = (unsigned long long)a.tiles * a.batch;
    *blocks = g;
    return g > 0 && g < 0x7fffffffull;
}

// ---- k_reg_radix geometry / launch ---------------------------------------------------------------------------------------
static bool reg_radix_split(const StftArgs &a, int dtype, unsigned *pa, unsigned *pb, unsigned *pc) {
    if (a.n_fft < 32 || (a.n_fft & 1u)) return false;
    return reg_split_len(a.n_fft / 2, dtype, pa, pb, pc);
}

static size_t reg_radix_band_bytes(const StftArgs &a, size_t es) {  // padded band table (rows of consecutive columns only)
    if (a.out_mode != OUT_MEL || !a.mel_pw) return 0;
    return (((size_t)a.mel_pchunks * 4 * es + (size_t)(2 * a.n_mels + 1) * 4) + 15) & ~size_t(15);
}

static size_t reg_radix_csr_bytes(const StftArgs &a, size_t es) {
    return a.out_mode == OUT_MEL ? (((size_t)a.mel_nnz * (es + 4) + (size_t)(a.n_mels + 1) * 4 + 15) & ~size_t(15)) : 0;
}

// LDS bytes of a tile of ft frames with its tables (without the bank)
static size_t reg_radix_bytes(const StftArgs &a, unsigned ft, unsigned sub, unsigned fa, unsigned fb, unsigned fc, size_t es) {
    const size_t fs = rr_frame_stride(fa, rr_swizzle(2 * (unsigned)es, fa, fb, fc).rs);
    const size_t m = (size_t)fa * fb * fc;
    const size_t pws = 4 * ((((size_t)a.nb_fft + 3) >> 2) | 1);
    size_t bytes = (ft * fs + m / 2 + 1 + m) * 2 * es;
    if (a.out_mode == OUT_MEL) bytes = ((bytes + 31) & ~size_t(31)) + (size_t)sub * pws * es;
    return (bytes + 15) & ~size_t(15);
}

// Tile + tables of one of the two persistent workgroups of a CU.  Per-bin outputs may take half of the CU's 160 KiB each: the
// f64 n_fft = 1024 / 512 and f32 2048 tiles then hold 8 frames instead of 4 (64-byte store runs): 758 -> 606, 685 -> 527 and
// 413 -> 363 us per 256 x 10 s.  Filterbank outputs keep the smaller budget (their |X|^2 rows come on top; 4096 / Mel lost 14 %).
static const size_t kRegBudget = 72 * 1024, kRegBudgetBins = 80 * 1024;
static const size_t kRegHardLimit = 144 * 1024;  // a single frame of the largest sizes may take most of the CU

// staged samples: the tile's (ft - 1) hop + n_fft samples must fit the chunk registers (256 threads x R rounds x 16 bytes) and the
// tile buffer they are written over
// rr_stft_waves for the host
static unsigned reg_radix_waves(size_t es, unsigned fa, unsigned fc, bool staged) {
    if (es == 8) return fa >= 16 ? 1 : SGX_RRW64;
    if (fc > 1) return fa <= 8 ? 3 : 2;
    return fa <= 16 ? 4 : (staged && fa <= 30) ? 3 : 2;
}

// the staged variant is the one to use (see rr_can_stage): per-bin outputs, f32 or the long f64 transforms
static bool reg_radix_want_staged(const StftArgs &a, unsigned fa, unsigned fc, size_t es) {
    return SGX_RR_STAGED && (SGX_RR_STAGE_MEL || a.out_mode != OUT_MEL) && (es == 4 || SGX_RR_STAGE_F64 || (fa >= 16 && fc > 1));
}

static bool reg_radix_stage_ok(const StftArgs &a, unsigned ft, unsigned fa, unsigned fb, unsigned fc, size_t es) {
    const unsigned long long len = (unsigned long long)(ft - 1) * a.hop + a.n_fft, epc = 16 / es;
    const unsigned long long chunks = (len + epc - 1) / epc;
    const size_t fs = rr_frame_stride(fa, rr_swizzle(2 * (unsigned)es, fa, fb, fc).rs);
    return chunks <= 256ull * rr_stage_rounds(fa * fb * fc, (unsigned)es) && chunks * 16 <= (unsigned long long)ft * fs * 2 * es;
}

static unsigned reg_radix_ft_max(unsigned fbc) {
    const bool p2 = (fbc & (fbc - 1)) == 0;
    const unsigned ni = (p2 && fbc >= SGX_RR_NI2_MIN) ? 2 : 1;  // rr_items
    unsigned ft = 1;
    while (ft < 32 && 2 * ft * fbc <= 256u * ni) ft *= 2;  // largest power of two with ft * BC <= 256 NI
    return ft;
}

bool plan_geometry_reg_radix(StftArgs &a, int dtype) {
    unsigned fa, fb, fc;
    if (!reg_radix_split(a, dtype, &fa, &fb, &fc)) return false;
    const size_t es = elem_size(dtype);
    if (a.n_samples * es >= (1ull << 31)) return false;  // the kernel addresses a row with signed 32-bit byte offsets
    // instances that run one workgroup per CU anyway (rr_waves == 1: f64 with a 16-point or longer pass) may use most of its LDS
    const size_t base = a.out_mode == OUT_MEL ? kRegBudget : kRegBudgetBins;
    const size_t budget = (dtype == SGX_F64 && fa >= 16) ? std::max(base, kRegHardLimit - 16 * 1024) : base;
    if (a.out_mode == OUT_MEL) {
        // Filterbank outputs write little, so what counts is how many frames a CU keeps in flight: workgroups per CU (registers,
        // LDS with the bank beside the tile) times frames per tile.  The |X|^2 rows may be produced in up to 4 parts of
        // ft / parts frames (two barriers per part) when that buys a larger tile or one more workgroup per CU (f32 n_fft 512:
        // 16-frame tiles in two parts run three workgroups per CU instead of two).  Ties: larger tile, then fewer parts.
        const unsigned by_regs = reg_radix_waves(es, fa, fc, false);
        const size_t bank = std::max(reg_radix_band_bytes(a, es), (size_t)0);
        unsigned best = 0;
        for (unsigned ft = reg_radix_ft_max(fb * fc); ft >= 1; ft >>= 1)
            for (unsigned parts = 1; parts <= 4u && ft / parts >= std::min(ft, 4u); parts *= 2) {
                const size_t lds = reg_radix_bytes(a, ft, ft / parts, fa, fb, fc, es);
                if (lds > budget) continue;
                const unsigned wgs = std::min<unsigned>(by_regs, (unsigned)std::max<size_t>(1, (160 * 1024) / (lds + bank + 512)));
                if (wgs * ft > best) {
                    best = wgs * ft;
                    a.ft = ft;
                    a.mel_sub = ft / parts;
                }
            }
#if SGX_RR_STAGE_MEL  // experiment: staged samples for filterbank outputs too
        if (best) a.staged = reg_radix_want_staged(a, fa, fc, es) && 2 * a.hop <= a.n_fft && reg_radix_stage_ok(a, a.ft, fa, fb, fc, es);
#endif
        if (best) return true;
    } else {
        // staged samples where frames overlap by at least half and the samples of the tile the budget allows fit the chunk
        // registers (hop = n_fft — the 2-D path's row pass — shares nothing between frames and would only lose tile size)
        for (unsigned ft = reg_radix_ft_max(fb * fc); ft >= 1; ft >>= 1)
            if (reg_radix_bytes(a, ft, ft, fa, fb, fc, es) <= budget) {
                a.ft = a.mel_sub = ft;
                a.staged = reg_radix_want_staged(a, fa, fc, es) && 2 * a.hop <= a.n_fft && reg_radix_stage_ok(a, ft, fa, fb, fc, es);
                return true;
            }
    }
    a.ft = a.mel_sub = 1;
    a.staged = 0;
    return reg_radix_bytes(a, 1, 1, fa, fb, fc, es) <= kRegHardLimit;
}

template <typename T, int A, int B, int C>
static hipError_t launch_reg_radix_t(const StftArgs &a, unsigned total, size_t lds, unsigned csr_lds, unsigned band_lds, bool staged, hipStream_t s) {
    if (lds > 64 * 1024) {
        hipError_t e = set_max_dynamic_lds((const void *)k_reg_radix<T, A, B, C, false>, (int)kRegHardLimit);
        if constexpr (rr_can_stage<T, A, C>())
            if (e == hipSuccess) e = set_max_dynamic_lds((const void *)k_reg_radix<T, A, B, C, true>, (int)kRegHardLimit);
        if (e != hipSuccess) return e;
    }
    // persistent workgroups: as many as are resident at once (registers: rr_waves per SIMD = workgroups per CU; LDS: 160 KB
    // per CU), each walking tiles blockIdx.x, + gridDim.x, ...
    const unsigned cus = device_cu_count();  // of the current device = the plan's (DeviceGuard)
    const unsigned by_regs = staged && rr_can_stage<T, A, C>() ? rr_stft_waves<T, A, B, C, true>() : rr_stft_waves<T, A, B, C, false>();
    const unsigned by_lds = (unsigned)std::max<size_t>(1, (160 * 1024) / (lds + 512));
    unsigned grid = std::min(total, cus * std::min(by_regs, by_lds));
    // tile order (see the kernel): XCD-contiguous rounds need a grid that is a multiple of 8; the per-round shift `rot` makes
    // (grid + rot) coprime to the tiles per signal, so that a workgroup walks through all tile indices of a signal instead of
    // staying on one residue class — with 32 tiles per signal on 512 workgroups the two bounds-checked edge tiles of every
    // signal would otherwise all land on 32 of them and the rest would wait (measured: f32 n_fft 400, 192 vs 154 us)
    unsigned rot = ~0u;
    if (total > grid) grid &= ~7u;
    if (grid % 8 == 0) {
        rot = 0;
        if (total > grid)
            while (rot < a.tiles && std::gcd(grid + rot, a.tiles) > 1) ++rot;
    }
    if constexpr (rr_can_stage<T, A, C>()) {
        if (staged) {
            hipLaunchKernelGGL((k_reg_radix<T, A, B, C, true>), dim3(grid), dim3(256), lds, s, a, total, csr_lds, band_lds, rot);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((k_reg_radix<T, A, B, C, false>), dim3(grid), dim3(256), lds, s, a, total, csr_lds, band_lds, rot);
    return hipGetLastError();
}

hipError_t launch_reg_radix(const StftArgs &a, int dtype, hipStream_t s) {
    unsigned long long g;
    unsigned fa, fb, fc;
    if (!grid_ok(a, &g) || !reg_radix_split(a, dtype, &fa, &fb, &fc) || (a.ft & (a.ft - 1))) return hipErrorInvalidConfiguration;
    if (a.ft > reg_radix_ft_max(fb * fc)) return hipErrorInvalidConfiguration;
    const size_t es = elem_size(dtype);
    const unsigned sub = a.mel_sub ? a.mel_sub : a.ft;
    const bool staged = a.staged != 0;
    if (sub > a.ft || (sub & (sub - 1))) return hipErrorInvalidConfiguration;
    if (staged && !(reg_radix_want_staged(a, fa, fc, es) && reg_radix_stage_ok(a, a.ft, fa, fb, fc, es))) return hipErrorInvalidConfiguration;
    size_t lds = reg_radix_bytes(a, a.ft, sub, fa, fb, fc, es);
    if (lds > kRegHardLimit) return hipErrorInvalidConfiguration;
    // the bank stays in LDS for the life of the workgroup when it fits beside the tile: the padded band table if the rows
    // are runs of consecutive columns, else the CSR arrays (else CSR from global memory)
    unsigned csr_lds = 0, band_lds = 0;
    const size_t room = std::min(std::max(kRegBudget, lds) + 16 * 1024, kRegHardLimit);
    const size_t band = reg_radix_band_bytes(a, es), csr = reg_radix_csr_bytes(a, es);
    if (band && lds + band <= room) {
        band_lds = (unsigned)band;
        lds += band;
    } else if (csr && lds + csr <= room) {
        csr_lds = (unsigned)csr;
        lds += csr;
    }
#define SGX_RR_F32(A, B, C) \
    if (fa == A && fb == B && fc == C) return launch_reg_radix_t<float, A, B, C>(a, (unsigned)g, lds, csr_lds, band_lds, staged, s);
#define SGX_RR_F64(A, B, C) \
    if (fa == A && fb == B && fc == C) return launch_reg_radix_t<double, A, B, C>(a, (unsigned)g, lds, csr_lds, band_lds, staged, s);
    if (dtype == SGX_F64) {
        SGX_RR_SPLITS_F64(SGX_RR_F64)
        SGX_RR_SPLITS_MIXED(SGX_RR_F64)
    } else {
        SGX_RR_SPLITS_F32(SGX_RR_F32)
        SGX_RR_SPLITS_MIXED(SGX_RR_F32)
    }
    return hipErrorInvalidConfiguration;
#undef SGX_RR_F32
#undef SGX_RR_F64
}

hipError_t launch_lds_radix2(const StftArgs &a, int dtype, hipStream_t s) {
    unsigned long long g;
    if (!grid_ok(a, &g)) return hipErrorInvalidConfiguration;
    size_t es = elem_size(dtype);
    size_t lds = (size_t)a.ft * (a.m + 1) * 2 * es + (a.out_mode == OUT_MEL ? (size_t)a.ft * a.nb_fft * es : 0);
    if (a.tw_lds) lds = (size_t)a.tw_lds + (size_t)a.m * 2 * es;
    if (hipError_t e = dtype == SGX_F64 ? lds_opt_in(k_lds_radix2<double>, lds) : lds_opt_in(k_lds_radix2<float>, lds); e != hipSuccess) return e;
    if (dtype == SGX_F64)
        hipLaunchKernelGGL(k_lds_radix2<double>, dim3((unsigned)g), dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL(k_lds_radix2<float>, dim3((unsigned)g), dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_direct_dft(const StftArgs &a, int dtype, hipStream_t s) {
    unsigned long long g;
    if (!grid_ok(a, &g)) return hipErrorInvalidConfiguration;
    size_t es = elem_size(dtype);
    size_t lds = (size_t)a.ft * a.n_fft * es + (a.out_mode == OUT_MEL ? (size_t)a.ft * a.nb_fft * es : 0);
    if (hipError_t e = dtype == SGX_F64 ? lds_opt_in(k_direct_dft<double>, lds) : lds_opt_in(k_direct_dft<float>, lds); e != hipSuccess) return e;
    if (dtype == SGX_F64)
        hipLaunchKernelGGL(k_direct_dft<double>, dim3((unsigned)g), dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL(k_direct_dft<float>, dim3((unsigned)g), dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_two_factor(const StftArgs &a, int dtype, hipStream_t s) {
    unsigned long long g;
    if (!grid_ok(a, &g) || a.fac_a < 2 || a.fac_a * a.fac_b != a.n_fft) return hipErrorInvalidConfiguration;
    const size_t lds = two_factor_bytes(a, a.ft, elem_size(dtype)) - (a.tw_lds ? 0 : (size_t)a.n_fft * 2 * elem_size(dtype));
    if (hipError_t e = dtype == SGX_F64 ? lds_opt_in(k_two_factor<double>, lds) : lds_opt_in(k_two_factor<float>, lds); e != hipSuccess) return e;
    if (dtype == SGX_F64)
        hipLaunchKernelGGL(k_two_factor<double>, dim3((unsigned)g), dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL(k_two_factor<float>, dim3((unsigned)g), dim3(256), lds, s, a);
    return hipGetLastError();
}

// Filterbank rows over a [batch][nb][n_frames] power (or magnitude) tensor in HBM -> [batch][n_mels][n_frames]: the second launch
// of the split filterbank path (long frames, see run_device in plan.hip).  One wave = one bank row x 64 consecutive frames: the
// row's CSR terms are wave-uniform (scalar loads), every term reads 64 consecutive frames of one bin (contiguous), and the sum
// runs in ascending column order, un-fused, exactly as SparseMatrix::multiply_vec does (spectrogram.rs:102-117).
template <typename T>
__global__ __launch_bounds__(64) void k_bank_rows(const T *pw, T *out, const unsigned *ptr, const unsigned *col, const T *val, unsigned nb,
                                                  unsigned n_mels, unsigned n_frames, unsigned fblocks, int amp, T eps) {
    const unsigned b = blockIdx.x / fblocks, f = (blockIdx.x - b * fblocks) * 64u + threadIdx.x, mm = blockIdx.y;
    if (f >= n_frames) return;
    const T *p = pw + (size_t)b * nb * n_frames + f;
    const unsigned i0 = ptr[mm], i1 = ptr[mm + 1];
    T acc = T(0);
#pragma unroll 4
    for (unsigned i = i0; i < i1; ++i) acc = t_mul_add_unfused(val[i], p[(size_t)col[i] * n_frames], acc);
    out[((size_t)b * n_mels + mm) * n_frames + f] = amp_apply(acc, amp, eps);
}

hipError_t launch_bank_rows(const void *pw, void *out, const StftArgs &a, int dtype, hipStream_t s) {
    const unsigned fblocks = (a.n_frames + 63u) / 64u;
    const unsigned long long gx = (unsigned long long)fblocks * a.batch;
    if (gx == 0 || gx >= 0x7fffffffull || a.n_mels == 0 || a.n_mels > 65535u) return hipErrorInvalidConfiguration;
    const dim3 grid((unsigned)gx, a.n_mels);
    if (dtype == SGX_F64)
        hipLaunchKernelGGL(k_bank_rows<double>, grid, dim3(64), 0, s, (const double *)pw, (double *)out, a.mel_ptr, a.mel_col,
                           (const double *)a.mel_val, a.nb_fft, a.n_mels, a.n_frames, fblocks, a.amp, (double)a.eps);
    else
        hipLaunchKernelGGL(k_bank_rows<float>, grid, dim3(64), 0, s, (const float *)pw, (float *)out, a.mel_ptr, a.mel_col,
                           (const float *)a.mel_val, a.nb_fft, a.n_mels, a.n_frames, fblocks, a.amp, (float)a.eps);
    return hipGetLastError();
}

hipError_t launch_mfcc(const void *mel, void *out, const void *basis, const void *lifter, unsigned batch, unsigned n_mels,
                       unsigned n_frames, unsigned n_mfcc, unsigned skip, int has_lifter, int dtype, hipStream_t s) {
    const unsigned long long n = (unsigned long long)batch * n_frames;
    const unsigned long long blocks = (n + 255) / 256;
    if (blocks == 0 || blocks >= 0x7fffffffull) return hipErrorInvalidConfiguration;
    const size_t es = dtype == SGX_F64 ? 8 : 4;
#define SGX_MFCC_ACC(T, NC)                                                                                                    \
    hipLaunchKernelGGL((k_mfcc_acc<T, NC>), dim3((unsigned)blocks), dim3(256), (size_t)n_mels * NC * es, s, (const T *)mel,    \
                       (T *)out, (const T *)basis, (const T *)lifter, batch, n_mels, n_frames, n_mfcc, skip, has_lifter)
    const int nc = n_mfcc <= 16 ? 16 : n_mfcc <= 32 ? 32 : n_mfcc <= 64 ? 64 : 0;
    if (nc && (size_t)n_mels * nc * es <= 48 * 1024) {
        if (dtype == SGX_F64) {
            if (nc == 16) SGX_MFCC_ACC(double, 16); else if (nc == 32) SGX_MFCC_ACC(double, 32); else SGX_MFCC_ACC(double, 64);
        } else {
            if (nc == 16) SGX_MFCC_ACC(float, 16); else if (nc == 32) SGX_MFCC_ACC(float, 32); else SGX_MFCC_ACC(float, 64);
        }
        return hipGetLastError();
    }
#undef SGX_MFCC_ACC
    if (dtype == SGX_F64)
        hipLaunchKernelGGL(k_mfcc<double>, dim3((unsigned)blocks), dim3(256), 0, s, (const double *)mel, (double *)out,
                           (const double *)basis, (const double *)lifter, batch, n_mels, n_frames, n_mfcc, skip, has_lifter);
    else
        hipLaunchKernelGGL(k_mfcc<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float *)mel, (float *)out,
                           (const float *)basis, (const float *)lifter, batch, n_mels, n_frames, n_mfcc, skip, has_lifter);
    return hipGetLastError();
}

}  // namespace sgx

#ifdef SGX_RR_STAMPS
extern "C" int sgx_debug_read_rr_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sgx::g_rr_stamps), sizeof(sgx::g_rr_stamps)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(sgx::g_rr_stamps), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif
