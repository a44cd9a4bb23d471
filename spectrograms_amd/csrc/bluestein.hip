// bluestein.hip — transforms of every length from 16 to 8192 (f64: 4096) that is neither a power of two nor one of the register-tiled
// mixed-radix sizes (primes, 2 x prime, 1023, 3000, ...) as a chirp-z convolution resident in LDS: STFT frames (k_bs_fused), complex
// sequences and Hermitian rows -> real (k_bs_c2c: the 2-D path's columns and inverse rows, the 1-D C2C plan, the generic inverse STFT).
//
// The reference plans EVERY length through realfft / RustFFT (src/fft_backend.rs:376-385), which pick mixed radix, Rader or
// Bluestein per length; round 2 ran such lengths as two-factor transforms or O(n^2) direct sums (n_fft 5003: 12.5 M multiply-adds
// per frame).
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
// The whole chain runs in ONE kernel with the sequence resident in LDS (k_bs_fused below): framing with virtual zero padding (S1),
// window x chirp, forward transform, product with FFT_M(b) / M, inverse transform, Z[k] = conj(c_k) Y[k], the two-frame split,
// |.|^2 / sqrt / dB or complex, stores into the reference's [signal][bin][frame] layout (S9).  M up to 16384 in f32 and 8192 in f64,
// i.e. n_fft <= 8192 / 4096; beyond that a plan keeps the two-factor kernel or the direct sum.  (Round 3 first built this as four
// launches per chunk of frame pairs over two [pairs][M] scratch buffers: bound by its six HBM passes — n_fft 1009 616 us against
// 160 us now, 5003 4.3 ms against 0.43 ms.)
// Filterbank outputs: f32 up to M = 1024 apply the bank's rows inside the kernel (the |X|^2 of a tile stay in LDS); longer sequences and
// f64 take the plan's split path — per-bin power here, then k_bank_rows.
#include <algorithm>
#include <cmath>
#include <vector>

#include "reg_radix.h"
#include "rr_layout.h"
#include "db_f64.h"
#include "sgx_internal.h"
#include "xcd_map.h"

// LDS per workgroup of k_bs_fused.  The f32 instances fit 4 waves per SIMD in registers (<= 128 VGPRs up to M = 2048), so four
// workgroups per CU; the f64 ones run at 1-2 waves per SIMD and want the larger tile (>= 256 pass-1 work items) instead.
// Measured, 64 x 10 s: f32 n_fft 1009 218 -> 169 us, 251 204 -> 150 us with 36 KiB; f64 509 316 -> 455 us (kept at 72 KiB).
#ifndef SGX_BS_LDS32
#define SGX_BS_LDS32 (36 * 1024)
#endif
#ifndef SGX_BS_LDS64
#define SGX_BS_LDS64 (72 * 1024)
#endif
#ifndef SGX_BS_OCC
#define SGX_BS_OCC 0  // waves per SIMD the register allocation of k_bs_fused aims at; 0: as k_c2c_reg (rr_waves)
#endif

namespace sgx {
namespace {

// one term of the reference's filterbank sum: the product rounded, then the sum (no fused multiply-add)
__device__ __forceinline__ float bs_mul_add_unfused(float w, float p, float acc) { return __fadd_rn(__fmul_rn(w, p), acc); }
__device__ __forceinline__ double bs_mul_add_unfused(double w, double p, double acc) { return __dadd_rn(__dmul_rn(w, p), acc); }
__device__ __forceinline__ float bs_db(float p) { return __builtin_log2f(p) * 3.01029995663981195f; }  // as the other f32 kernels
__device__ __forceinline__ double bs_db(double p) { return db_f64(p); }

// ---- the fused kernel ----------------------------------------------------------------------------------------------------
// One workgroup carries `tile` sequences (frame pairs) through the whole chain in LDS; the only HBM traffic is the samples in
// and the bins out.
//
// The length-M transform is k_c2c_reg's (kernels_reg2d.hip): M = A B C, passes P1 (A points, over n1 of n = n1 B C + r), T1
// (twiddle W_M^(r k1)), P2 (B points), T2, P3 (C points), each in place on the elements its work item owns, bin
// k = k1 + A (k2 + B k3) ending at row k1, position (k2, k3) of the swizzled tile (rr_layout.h).  The convolution needs no
// reordering at all: the product with FFT_M(b) / M is taken where P3 leaves the bins, and the inverse transform runs the SAME
// passes in the opposite order on the same elements,
//     y = P1^-1 T1^-1 P2^-1 T2^-1 P3^-1 (D X),   Pi^-1 z = conj(Pi conj z),  Ti^-1 = conj Ti
// so with c = conj(D X) it is the forward operators again, conj(y) = P1 T1 P2 T2 P3 c: twiddle BEFORE each transform instead of
// after, nothing else changes.  P3, the product and P3^-1 are one step on a work item's registers (two-pass splits: P2, product,
// P2^-1).  P1^-1 leaves y[n1 B C + r] in the registers of the work item that loaded x[n1 B C + r]; Z[m] = conj(c_m) y[m]
// (m < n only) goes back to that element's own place, and the two-frame split reads Z[k] and Z[n - k] from there.
// Only the first half of a sequence is non-zero going in (n <= M / 2) and only the first half is wanted coming out: the first
// and the last A-point transform have half their inputs / outputs as compile-time zeros / dead values.
//
// Deviation from S4, f32 / f64 alike: the kernel multiplies a sample by wc[m] = w[m] conj(c_m), ONE table rounded from f64, rather
// than by w[m] (rounded) and then by the chirp; the chirp-z result is not bit-comparable with a direct transform either way.
#ifdef SGX_BS_STAMPS  // diagnostic build only (tools/stamps_bs.py): a wave's cycles per stage of k_bs_fused
__device__ unsigned long long g_bs_stamps[16];
#define BS_STAMP(i)                                                                \
    do {                                                                           \
        unsigned long long t_;                                                     \
        __builtin_amdgcn_sched_barrier(0);                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                         \
        st_acc[i] += t_ - st_prev;                                                 \
        st_prev = t_;                                                              \
    } while (0)
#define BS_STAMP_PARAMS , unsigned long long *st_acc, unsigned long long &st_prev
#define BS_STAMP_ARGS , st_acc, st_prev
#else
#define BS_STAMP(i)
#define BS_STAMP_PARAMS
#define BS_STAMP_ARGS
#endif

struct BsFused {
    const void *x;
    void *out;
    unsigned long long sample_stride, n_samples, total;  // total = batch * pairs
    unsigned n, hop, pad, n_frames, nb, pairs, tiles;
    const void *wc, *chirp, *bhp, *tw;  // [n] w conj(c), [n] conj(c), [M] FFT_M(b) / M in the order the product step reads it, [M] W_M^k
    int complex_out, amp;
    double eps;
    // filterbank outputs: the bank's CSR rows over the bins (null: per-bin output); amp is then applied to the bank's sums
    // (AMP_MAG_IN: the bank weighs magnitudes and its sums are final)
    const unsigned *mel_ptr, *mel_col;
    const void *mel_val;
    unsigned n_mels, n_out;
};

// The middle of the convolution, shared by the frame kernel and the complex-sequence kernel: on entry every work item has written
// its P1 + T1 results to the tile; on return the tile holds conj(T1^-1-input of P1^-1), i.e. what P1^-1's work items read.
template <typename T, int A_, int B_, int C_>
__device__ __forceinline__ void bs_middle(typename PairOf<T>::type *buf, unsigned ns, unsigned tid, const typename PairOf<T>::type *tw,
                                          const typename PairOf<T>::type *bhp BS_STAMP_PARAMS) {
    typedef typename PairOf<T>::type V;
    constexpr unsigned A = A_, B = B_, C = C_, N = A * B * C;
    constexpr int LB = ct_log2_ceil(B);
    typedef RrLayout<sizeof(V), A_, B_, C_> L;
    constexpr unsigned RS = L::RS, FS = L::FS;
    auto wrap = [](unsigned e) { return e & (N - 1); };
    __syncthreads();
    BS_STAMP(2);
    // P2 (+ T2); two-pass splits: P2, product, P2^-1
    for (unsigned idx = tid; idx < ns * A * C; idx += 256) {
        const unsigned s = idx / (A * C), q = idx % (A * C), k1 = q / C, n3 = q % C;
        V *row = buf + (size_t)s * FS + k1 * RS;
        const unsigned lp = n3 ^ L::k1_mask(k1);
        V v[B];
#pragma unroll
        for (unsigned n2 = 0; n2 < B; ++n2) v[n2] = row[lp ^ L::hi_part(n2)];
        inreg::MixFft<B, V>::run(v);
        if constexpr (C > 1) {
            V q2[LB];
#pragma unroll
            for (int j = 0; j < LB; ++j) q2[j] = tw[wrap((A << j) * n3)];
            row[lp ^ L::hi_part(0)] = v[0];
#pragma unroll
            for (unsigned k2 = 1; k2 < B; ++k2) row[lp ^ L::hi_part(k2)] = inreg::cmulv(v[k2], rr_twiddle<LB>(q2, k2));
        } else {
#pragma unroll
            for (unsigned k2 = 0; k2 < B; ++k2) {  // bin k1 + A k2; the table is [k2][k1]
                const V y = inreg::cmulv(v[k2], bhp[k2 * A + k1]);
                v[k2] = (V){y.x, -y.y};
            }
            inreg::MixFft<B, V>::run(v);
#pragma unroll
            for (unsigned n2 = 0; n2 < B; ++n2) row[lp ^ L::hi_part(n2)] = v[n2];
        }
    }
    BS_STAMP(3);
    __syncthreads();
    BS_STAMP(4);
    if constexpr (C > 1) {
        // P3, product, P3^-1 (from here on the data is the conjugate of the inverse transform's)
        for (unsigned idx = tid; idx < ns * A * B; idx += 256) {
            const unsigned s = idx / (A * B), q = idx % (A * B), k1 = q / B, k2 = q % B;
            V *row = buf + (size_t)s * FS + k1 * RS;
            const unsigned lp = L::hi_part(k2) ^ L::k1_mask(k1);
            V v[C], h[C];
#pragma unroll
            for (unsigned k3 = 0; k3 < C; ++k3) h[k3] = bhp[k3 * (A * B) + q];  // bin k1 + A (k2 + B k3); the table is [k3][k1][k2]
#pragma unroll
            for (unsigned n3 = 0; n3 < C; ++n3) v[n3] = row[lp ^ n3];
            inreg::MixFft<C, V>::run(v);
#pragma unroll
            for (unsigned k3 = 0; k3 < C; ++k3) {
                const V y = inreg::cmulv(v[k3], h[k3]);
                v[k3] = (V){y.x, -y.y};
            }
            inreg::MixFft<C, V>::run(v);
#pragma unroll
            for (unsigned n3 = 0; n3 < C; ++n3) row[lp ^ n3] = v[n3];
        }
        BS_STAMP(5);
        __syncthreads();
        BS_STAMP(6);
        // T2, P2
        for (unsigned idx = tid; idx < ns * A * C; idx += 256) {
            const unsigned s = idx / (A * C), q = idx % (A * C), k1 = q / C, n3 = q % C;
            V *row = buf + (size_t)s * FS + k1 * RS;
            const unsigned lp = n3 ^ L::k1_mask(k1);
            V q2[LB];
#pragma unroll
            for (int j = 0; j < LB; ++j) q2[j] = tw[wrap((A << j) * n3)];
            V v[B];
            v[0] = row[lp ^ L::hi_part(0)];
#pragma unroll
            for (unsigned k2 = 1; k2 < B; ++k2) v[k2] = inreg::cmulv(row[lp ^ L::hi_part(k2)], rr_twiddle<LB>(q2, k2));
            inreg::MixFft<B, V>::run(v);
#pragma unroll
            for (unsigned n2 = 0; n2 < B; ++n2) row[lp ^ L::hi_part(n2)] = v[n2];
        }
        BS_STAMP(7);
        __syncthreads();
        BS_STAMP(8);
    }
}

template <typename T, int A, int B, int C>
constexpr unsigned bs_waves() {
    if (A > 16) return 1;  // 32-point passes: the 512-register budget
    if (sizeof(T) == 8 && A == 16 && B * C <= 128) return 2;  // f64 16 x 16 x 8 and shorter: two 72 KiB workgroups per CU only if they fit 256 registers
    return SGX_BS_OCC && sizeof(T) == 4 ? SGX_BS_OCC : rr_waves<T, A, B, C>();
}

template <typename T, int A_, int B_, int C_>
__global__ __launch_bounds__(256, (bs_waves<T, A_, B_, C_>())) void k_bs_fused(BsFused a, unsigned ltile) {
    typedef typename PairOf<T>::type V;
    constexpr unsigned A = A_, B = B_, C = C_, BC = B * C, N = A * BC, HA = A / 2;
    constexpr int LA = ct_log2_ceil(A), LB = ct_log2_ceil(B);
    static_assert(ct_is_pow2(N) && A % 2 == 0, "k_bs_fused: power-of-two convolution lengths");
    typedef RrLayout<sizeof(V), A_, B_, C_> L;
    constexpr unsigned RS = L::RS, FS = L::FS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    V *buf = (V *)smem;  // [tile][FS]
    __shared__ unsigned long long sig_of[32];
    __shared__ unsigned frame_of[32], interior_of[32];
    const unsigned tid = threadIdx.x, tile = 1u << ltile;
    const unsigned lb = xcd_logical_block(a.tiles);  // a signal's pairs write neighbouring frames of the same output rows
    if (lb >= a.tiles) return;
#ifdef SGX_BS_STAMPS
    unsigned long long st_acc[14] = {0}, st_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
    const unsigned long long q0 = (unsigned long long)lb * tile;
    const unsigned ns = (unsigned)min((unsigned long long)tile, a.total - q0);
    if (tid < tile) {  // sequence q = pair p of signal b: frames 2 p and 2 p + 1
        const unsigned long long q = q0 + tid, b = q / a.pairs;
        const unsigned f = 2u * (unsigned)(q - b * a.pairs);
        sig_of[tid] = b;
        frame_of[tid] = f;
        // both frames of the pair exist and lie inside the signal: their loads need no range checks
        interior_of[tid] = f + 1u < a.n_frames && (unsigned long long)f * a.hop >= a.pad &&
                           (unsigned long long)(f + 1u) * a.hop + a.n <= a.n_samples + a.pad;
    }
    __syncthreads();
    BS_STAMP(0);
    const T *x = (const T *)a.x;
    const V *wc = (const V *)a.wc, *chirp = (const V *)a.chirp, *bhp = (const V *)a.bhp, *tw = (const V *)a.tw;
    const unsigned n = a.n;
    auto wrap = [](unsigned e) { return e & (N - 1); };
    auto item = [&](unsigned idx, unsigned &s, unsigned &r) {
        r = idx % BC;
        s = idx / BC;
        return idx < tile * BC && s < ns;
    };

    // P1 + T1: framing with virtual zero padding (S1), window x chirp, A-point transform.  Work items are taken NI at a time, all
    // their loads first.  (A software prefetch of the NEXT item's samples in loop-carried registers cost more moves than this stage
    // has butterflies — 435 of its 1000 vector instructions at M = 2048; no overlap at all lost 10 % where a thread has two items.)
    constexpr unsigned NI = A <= 8 ? 2 : 1;  // (the 16- and 32-point instances hold one item's samples at a time: registers = workgroups per CU)
    for (unsigned idx0 = tid; idx0 < tile * BC; idx0 += 256 * NI) {
        V p[NI][HA], w[NI][HA];
#pragma unroll
        for (unsigned q = 0; q < NI; ++q) {
            unsigned s, r;
            if (!item(idx0 + 256 * q, s, r)) continue;
            const T *xs = x + sig_of[s] * a.sample_stride;
            const unsigned f = frame_of[s];
            const long long base = (long long)f * a.hop - (long long)a.pad + r;
            if (interior_of[s]) {
                const T *pa = xs + base, *pb = pa + a.hop;
#pragma unroll
                for (unsigned n1 = 0; n1 < HA; ++n1) {
                    const bool in = n1 * BC + r < n;
                    p[q][n1] = (V){in ? pa[n1 * BC] : T(0), in ? pb[n1 * BC] : T(0)};
                    w[q][n1] = in ? wc[n1 * BC + r] : (V){T(0), T(0)};
                }
            } else {
                const bool two = f + 1u < a.n_frames;
#pragma unroll
                for (unsigned n1 = 0; n1 < HA; ++n1) {
                    const unsigned m = n1 * BC + r;
                    p[q][n1] = (V){T(0), T(0)};
                    w[q][n1] = (V){T(0), T(0)};
                    if (m < n) {
                        const long long sa = base + (long long)(n1 * BC), sb = sa + a.hop;
                        if (sa >= 0 && (unsigned long long)sa < a.n_samples) p[q][n1].x = xs[sa];
                        if (two && sb >= 0 && (unsigned long long)sb < a.n_samples) p[q][n1].y = xs[sb];
                        w[q][n1] = wc[m];
                    }
                }
            }
        }
#pragma unroll
        for (unsigned q = 0; q < NI; ++q) {
            unsigned s, r;
            if (!item(idx0 + 256 * q, s, r)) continue;
            V v[A];
#pragma unroll
            for (unsigned n1 = 0; n1 < HA; ++n1) {
                v[n1] = inreg::cmulv(p[q][n1], w[q][n1]);  // (xa + i xb) wc
                v[n1 + HA] = (V){T(0), T(0)};
            }
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
    }
    BS_STAMP(1);
    bs_middle<T, A_, B_, C_>(buf, ns, tid, tw, bhp BS_STAMP_ARGS);
    // T1, P1, then Z[m] = conj(c_m) y[m] for m < n back to element m's own place
    for (unsigned idx = tid; idx < tile * BC; idx += 256) {
        unsigned s, r;
        if (!item(idx, s, r)) continue;
        V ch[HA];
#pragma unroll
        for (unsigned n1 = 0; n1 < HA; ++n1) ch[n1] = n1 * BC + r < n ? chirp[n1 * BC + r] : (V){T(0), T(0)};
        V pw2[LA];
#pragma unroll
        for (int j = 0; j < LA; ++j) pw2[j] = tw[wrap((1u << j) * r)];
        V *dst = buf + (size_t)s * FS;
        const unsigned pp = L::hi_part(r / C) ^ (r % C);
        V v[A];
        v[0] = dst[pp ^ L::k1_mask(0)];
#pragma unroll
        for (unsigned k1 = 1; k1 < A; ++k1) v[k1] = inreg::cmulv((dst + (pp ^ L::k1_mask(k1)))[k1 * RS], rr_twiddle<LA>(pw2, k1));
        inreg::MixFft<A, V>::run(v);
#pragma unroll
        for (unsigned n1 = 0; n1 < HA; ++n1) {
            const V y = {v[n1].x, -v[n1].y};
            (dst + (pp ^ L::k1_mask(n1)))[n1 * RS] = inreg::cmulv(y, ch[n1]);
        }
    }
    BS_STAMP(9);
    __syncthreads();
    BS_STAMP(10);
    // the two frames of a pair come apart by their Hermitian symmetry; lanes walk the tile's pairs first (neighbouring frames of
    // one output row), a work item writes both frames of its pair.  A thread keeps its pair over all its bins (256 is a multiple of
    // the tile), and the output mode picks one of four specialised loops (uniform branch) — the selects of a single loop made
    // every item pay a sqrt and a log2.
    {
        const unsigned s = tid & (tile - 1u);
        if (s < ns) {
            const unsigned nb = a.nb, f = frame_of[s], kstep = 256u >> ltile;
            const bool two = f + 1u < a.n_frames;
            const V *seq = buf + (size_t)s * FS;
            const unsigned long long o0 = sig_of[s] * nb * a.n_frames + f;
            const T eps = (T)a.eps;
            auto at = [&](unsigned m) {
                const unsigned n1 = m / BC, r = m % BC;
                return seq[n1 * RS + ((L::hi_part(r / C) ^ (r % C)) ^ L::k1_mask(n1))];
            };
            auto walk = [&](auto &&emit) {
                for (unsigned k = tid >> ltile; k < nb; k += kstep) {
                    const V Z = at(k), Zc = at(k == 0 ? 0u : n - k);  // Zm = conj Z[n - k] = (Zc.x, -Zc.y)
                    const V Xa = {T(0.5) * (Z.x + Zc.x), T(0.5) * (Z.y - Zc.y)};   //    (Z + Zm) / 2
                    const V Xb = {T(0.5) * (Z.y + Zc.y), T(-0.5) * (Z.x - Zc.x)};  // -i (Z - Zm) / 2
                    emit(o0 + (unsigned long long)k * a.n_frames, Xa, Xb);
                }
            };
            auto power = [](V X) { return X.x * X.x + X.y * X.y; };  // norm_sqr (spectrogram.rs:1332-1334)
            if (a.mel_ptr) {
                // filterbank outputs: the pair's two powers (magnitudes) take the place of Z[k] — read by this work item only (the
                // mirror Z[n - k] of another bin k' would need k + k' = n, i.e. both n / 2) —, the bank's rows follow below
                V *wseq = buf + (size_t)s * FS;
                for (unsigned k = tid >> ltile; k < nb; k += kstep) {
                    const unsigned n1 = k / BC, r = k % BC, pos = n1 * RS + ((L::hi_part(r / C) ^ (r % C)) ^ L::k1_mask(n1));
                    const V Z = wseq[pos], Zc = at(k == 0 ? 0u : n - k);
                    const V Xa = {T(0.5) * (Z.x + Zc.x), T(0.5) * (Z.y - Zc.y)};
                    const V Xb = {T(0.5) * (Z.y + Zc.y), T(-0.5) * (Z.x - Zc.x)};
                    const T pa = power(Xa), pb = power(Xb);
                    wseq[pos] = a.amp == AMP_MAG_IN ? (V){sqrt(pa), sqrt(pb)} : (V){pa, pb};
                }
            } else if (a.complex_out) {
                walk([&](unsigned long long o, V Xa, V Xb) {
                    ((V *)a.out)[o] = Xa;
                    if (two) ((V *)a.out)[o + 1] = Xb;
                });
            } else if (a.amp == AMP_MAGNITUDE) {
                walk([&](unsigned long long o, V Xa, V Xb) {
                    ((T *)a.out)[o] = sqrt(power(Xa));
                    if (two) ((T *)a.out)[o + 1] = sqrt(power(Xb));
                });
            } else if (a.amp == AMP_DB) {
                walk([&](unsigned long long o, V Xa, V Xb) {
                    const T pa = power(Xa), pb = power(Xb);
                    ((T *)a.out)[o] = bs_db(pa > eps ? pa : eps);
                    if (two) ((T *)a.out)[o + 1] = bs_db(pb > eps ? pb : eps);
                });
            } else {
                walk([&](unsigned long long o, V Xa, V Xb) {
                    ((T *)a.out)[o] = power(Xa);
                    if (two) ((T *)a.out)[o + 1] = power(Xb);
                });
            }
        }
    }
    if (a.mel_ptr) {  // uniform
        // SparseMatrix::multiply_vec (spectrogram.rs:102-117) per frame: sequential, un-fused, in ascending column order — two
        // frames (one pair) per work item, lanes over the tile's pairs first
        __syncthreads();
        const T *val = (const T *)a.mel_val;
        const T eps = (T)a.eps;
        for (unsigned idx = tid; idx < (a.n_mels << ltile); idx += 256) {
            const unsigned s = idx & (tile - 1u), m = idx >> ltile;
            if (s >= ns) continue;
            const V *seq = buf + (size_t)s * FS;
            V acc = {T(0), T(0)};
            for (unsigned i = a.mel_ptr[m], i1 = a.mel_ptr[m + 1]; i < i1; ++i) {
                const unsigned k = a.mel_col[i], n1 = k / BC, r = k % BC;
                const V P = seq[n1 * RS + ((L::hi_part(r / C) ^ (r % C)) ^ L::k1_mask(n1))];
                const T w = val[i];
                acc = (V){bs_mul_add_unfused(w, P.x, acc.x), bs_mul_add_unfused(w, P.y, acc.y)};
            }
            const unsigned f = frame_of[s];
            T *o = (T *)a.out + (sig_of[s] * a.n_out + m) * a.n_frames + f;
            auto fin = [&](T v) { return a.amp == AMP_MAGNITUDE ? sqrt(v) : a.amp == AMP_DB ? bs_db(v > eps ? v : eps) : v; };
            o[0] = fin(acc.x);
            if (f + 1u < a.n_frames) o[1] = fin(acc.y);
        }
    }
    BS_STAMP(11);
#ifdef SGX_BS_STAMPS
    if ((threadIdx.x & 63u) == 0 && (lb & 127u) == 0) {  // a sample of the tiles: the atomics of every wave would be the slowest part of the run
        for (int q = 0; q < 12; ++q) atomicAdd(&g_bs_stamps[q], st_acc[q]);
        atomicAdd(&g_bs_stamps[12], 1ull);
    }
#endif
}

// ---- complex sequences (columns of the 2-D path, the 1-D C2C plan, Hermitian rows -> real) ------------------------------------
// The same chain for `tile` complex sequences of length n with k_c2c_reg's addressing (arbitrary element strides on both sides, the
// thread mapping of the loads / stores follows the unit stride): x[m] conj(c_m) in, conj(c_k) Y[k] out, forward or inverse (conjugate
// on the way in and out).  HERM: the input is the half spectrum of a real row (n / 2 + 1 bins, the rest by Hermitian symmetry; the
// imaginary parts of the DC and, for even n, Nyquist bins are dropped and reported as realfft's C2R does, src/fft_backend.rs:782-793)
// and the output is the row's n real samples, scaled and optionally windowed — the inverse row pass of ifft2d / the per-frame C2R of
// the generic inverse STFT at lengths without a pass split.  TWO rows ride one sequence there too: with Z = X_a + i X_b the inverse
// transform is x_a + i x_b, so sequence s carries rows 2 s and 2 s + 1 of its image (an odd last row rides alone).
// HALF: the same rows for an EVEN length 2 n whose full-length convolution does not fit LDS (f64 above 4096, f32 above 8192): the
// half-length complex form Z'[k] = (X[k] + conj X[n - k]) + i conj(W_2n^k)(X[k] - conj X[n - k]), k < n, inverts to x[2 j] + i x[2 j + 1]
// (k_c2r_reg's fold), one row per sequence, M >= 2 n - 1 = the row length - 1.
// FWDH: the forward direction of the same: STFT frames of an even length 2 n (f64 4098 ... 8192, f32 8194 ... 16384, not powers of two or
// listed sizes) as z[j] = w[2 j] x[2 j] + i w[2 j + 1] x[2 j + 1], Z = DFT_n(z), X[k] = (Z[k] + conj Z[n - k]) / 2 - (i / 2) W_2n^k (Z[k] - conj Z[n - k]).
struct BsC2c {
    const void *in;
    void *out;
    unsigned n, nseq, tiles, total_tiles;  // tiles per image, tiles * batch (HERM: nseq = row PAIRS per image, nrows = rows)
    unsigned nrows;
    unsigned long long in_img, out_img, in_ss, in_is, out_ss, out_is;
    int inverse, in_seq_fast, out_seq_fast;
    double scale;
    const void *chirp, *bhp, *tw;
    const void *win;     // HERM / HALF: optional window (of the real row's length) applied after the scale
    const void *twn;     // HALF: e^(-2 pi i k / (2 n)), k < n: the real row has 2 n samples, the sequence is its half-length complex form
    unsigned *bad_flag;  // HERM: set when a DC / Nyquist bin carries an imaginary part
    // FWDH (forward STFT frames of an even length 2 n in half-length complex form): sequence s of image b = frame s of signal b, read
    // from in[b * in_img + s * hop - pad + j] with virtual zero padding (S1) and the window `win` (S4), bins k <= n written to
    // out[b * out_img + k * out_is + s]; twn as for HALF
    unsigned hop, pad;
    unsigned long long n_samples;
    int complex_out, amp;
    double eps;
};

template <typename T, int A_, int B_, int C_, int RMODE>
__global__ __launch_bounds__(256, (bs_waves<T, A_, B_, C_>())) void k_bs_c2c(BsC2c a, unsigned ltile) {
    constexpr bool HERM = RMODE == 1, HALF = RMODE == 2, FWDH = RMODE == 3;  // 0: complex sequences
    typedef typename PairOf<T>::type V;
    constexpr unsigned A = A_, B = B_, C = C_, BC = B * C, N = A * BC, HA = A / 2;
    constexpr int LA = ct_log2_ceil(A);
    typedef RrLayout<sizeof(V), A_, B_, C_> L;
    constexpr unsigned RS = L::RS, FS = L::FS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    V *buf = (V *)smem;  // [tile][FS]
    const unsigned tid = threadIdx.x, tile = 1u << ltile;
    const unsigned lb = xcd_logical_block(a.total_tiles);
    if (lb >= a.total_tiles) return;
#ifdef SGX_BS_STAMPS
    unsigned long long st_acc[14] = {0}, st_prev = 0;
#endif
    const unsigned t = lb % a.tiles, b = lb / a.tiles;
    const unsigned s0 = t * tile, ns = min(tile, a.nseq - s0);
    const V *in = (const V *)a.in + (size_t)b * a.in_img;
    const V *chirp = (const V *)a.chirp, *bhp = (const V *)a.bhp, *tw = (const V *)a.tw;
    const unsigned n = a.n, half = n / 2;
    const T cj = (RMODE == 0 && a.inverse) ? T(-1) : T(1);  // inverse = conj(forward(conj x)); HERM is an inverse by construction (below)
    auto wrap = [](unsigned e) { return e & (N - 1); };
    auto item = [&](unsigned idx, unsigned &s, unsigned &r) {
        if (a.in_seq_fast) { s = idx & (tile - 1); r = idx >> ltile; } else { r = idx % BC; s = idx / BC; }
        return idx < tile * BC && s < ns;
    };
    // element m of sequence s, conjugated for an inverse transform.  HERM: a row's X[m] = in[m] (m <= n / 2), conj(in[n - m]) above;
    // rows a and b of the pair enter as conj(Z) = conj(X_a) - i conj(X_b), W = DFT(conj Z) = conj(x_a + i x_b): x_a = Re W, x_b = -Im W.
    auto herm = [&](const V *row, unsigned m) {  // conj X[m] of one row
        const bool up = m > half;
        V v = row[(size_t)(up ? n - m : m) * a.in_is];
        if (m == 0 || (2 * m == n)) {
            if (v.y != T(0) && a.bad_flag) *a.bad_flag = 1u;
            v.y = T(0);
        }
        return (V){v.x, up ? v.y : -v.y};
    };
    auto element = [&](const V *seq, unsigned m, bool second) {
        if constexpr (FWDH) {  // seq: the signal's samples (as T), frame start in `fstart`
            return (V){T(0), T(0)};  // (loaded in the stage itself: it needs the frame's start)
        } else if constexpr (HALF) {
            V Xk = seq[(size_t)m * a.in_is], Xm = seq[(size_t)(n - m) * a.in_is];
            if (m == 0) {  // DC and Nyquist bins
                if ((Xk.y != T(0) || Xm.y != T(0)) && a.bad_flag) *a.bad_flag = 1u;
                Xk.y = T(0);
                Xm.y = T(0);
            }
            const V Aa = {Xk.x + Xm.x, Xk.y - Xm.y}, D = {Xk.x - Xm.x, Xk.y + Xm.y};
            const V wn = ((const V *)a.twn)[m];
            const V Tt = inreg::cmulv(D, (V){wn.x, -wn.y});
            return (V){Aa.x - Tt.y, -(Aa.y + Tt.x)};  // conj(A + i T)
        } else if constexpr (HERM) {
            const V ca = herm(seq, m);
            if (!second) return ca;
            const V cb = herm(seq + a.in_ss, m);
            return (V){ca.x + cb.y, ca.y - cb.x};  // conj(X_a) - i conj(X_b)
        } else {
            const V v = seq[(size_t)m * a.in_is];
            return (V){v.x, cj * v.y};
        }
    };
    for (unsigned idx = tid; idx < tile * BC; idx += 256) {
        unsigned s, r;
        if (!item(idx, s, r)) continue;
        const V *seq = in + (size_t)(s0 + s) * (HERM ? 2u : 1u) * a.in_ss;  // (HALF: one row per sequence)
        const bool second = HERM && 2u * (s0 + s) + 1u < a.nrows;
        V v[A];
        if constexpr (FWDH) {
            const T *xs = (const T *)a.in + (size_t)b * a.in_img;
            const long long fstart = (long long)(s0 + s) * a.hop - (long long)a.pad;
            const T *w = (const T *)a.win;
#pragma unroll
            for (unsigned n1 = 0; n1 < HA; ++n1) {
                const unsigned m = n1 * BC + r;
                V z = {T(0), T(0)};
                if (m < n) {
                    const long long p0 = fstart + 2 * (long long)m;
                    if (p0 >= 0 && (unsigned long long)p0 < a.n_samples) z.x = xs[p0] * w[2u * m];
                    if (p0 + 1 >= 0 && (unsigned long long)(p0 + 1) < a.n_samples) z.y = xs[p0 + 1] * w[2u * m + 1u];
                    z = inreg::cmulv(z, chirp[m]);
                }
                v[n1] = z;
                v[n1 + HA] = (V){T(0), T(0)};
            }
        } else {
#pragma unroll
            for (unsigned n1 = 0; n1 < HA; ++n1) {
                const unsigned m = n1 * BC + r;
                v[n1] = m < n ? inreg::cmulv(element(seq, m, second), chirp[m]) : (V){T(0), T(0)};
                v[n1 + HA] = (V){T(0), T(0)};
            }
        }
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
    bs_middle<T, A_, B_, C_>(buf, ns, tid, tw, bhp BS_STAMP_ARGS);
    for (unsigned idx = tid; idx < tile * BC; idx += 256) {
        unsigned s, r;
        if (!item(idx, s, r)) continue;
        V ch[HA];
#pragma unroll
        for (unsigned n1 = 0; n1 < HA; ++n1) ch[n1] = n1 * BC + r < n ? chirp[n1 * BC + r] : (V){T(0), T(0)};
        V pw2[LA];
#pragma unroll
        for (int j = 0; j < LA; ++j) pw2[j] = tw[wrap((1u << j) * r)];
        V *dst = buf + (size_t)s * FS;
        const unsigned pp = L::hi_part(r / C) ^ (r % C);
        V v[A];
        v[0] = dst[pp ^ L::k1_mask(0)];
#pragma unroll
        for (unsigned k1 = 1; k1 < A; ++k1) v[k1] = inreg::cmulv((dst + (pp ^ L::k1_mask(k1)))[k1 * RS], rr_twiddle<LA>(pw2, k1));
        inreg::MixFft<A, V>::run(v);
#pragma unroll
        for (unsigned n1 = 0; n1 < HA; ++n1) {
            const V y = {v[n1].x, -v[n1].y};
            (dst + (pp ^ L::k1_mask(n1)))[n1 * RS] = inreg::cmulv(y, ch[n1]);
        }
    }
    __syncthreads();
    if constexpr (FWDH) {
        // bins 0 ... n of every frame of the tile; lanes walk the frames first (contiguous in the output)
        const T eps = (T)a.eps;
        const V *twn = (const V *)a.twn;
        for (unsigned idx = tid; idx < (n + 1u) << ltile; idx += 256) {
            const unsigned s = idx & (tile - 1u), k = idx >> ltile;
            if (s >= ns) continue;
            const V *seq = buf + (size_t)s * FS;
            auto at = [&](unsigned m) {
                const unsigned n1 = m / BC, r = m % BC;
                return seq[n1 * RS + ((L::hi_part(r / C) ^ (r % C)) ^ L::k1_mask(n1))];
            };
            const V Zk = at(k == n ? 0u : k), Zc = at(k == 0u || k == n ? 0u : n - k);  // Zm = conj Z[n - k]
            const V E = {T(0.5) * (Zk.x + Zc.x), T(0.5) * (Zk.y - Zc.y)};
            const V O = {T(0.5) * (Zk.y + Zc.y), T(-0.5) * (Zk.x - Zc.x)};  // -i (Zk - Zm) / 2
            const V wk = k == n ? (V){T(-1), T(0)} : twn[k];                    // W_2n^k
            const V X = E + inreg::cmulv(O, wk);
            const unsigned long long o = (size_t)b * a.out_img + (size_t)k * a.out_is + (s0 + s);
            if (a.complex_out) {
                ((V *)a.out)[o] = X;
            } else {
                const T pw = X.x * X.x + X.y * X.y;
                ((T *)a.out)[o] = a.amp == AMP_MAGNITUDE ? sqrt(pw) : a.amp == AMP_DB ? bs_db(pw > eps ? pw : eps) : pw;
            }
        }
        return;
    }
    const T sc = (T)a.scale;
    for (unsigned idx = tid; idx < tile * n; idx += 256) {
        unsigned s, k;
        if (a.out_seq_fast) { s = idx & (tile - 1); k = idx >> ltile; } else { k = idx % n; s = idx / n; }
        if (s >= ns) continue;
        const unsigned n1 = k / BC, r = k % BC;
        const V Z = buf[(size_t)s * FS + n1 * RS + ((L::hi_part(r / C) ^ (r % C)) ^ L::k1_mask(n1))];
        if constexpr (HALF) {  // W = DFT(conj Z'): x[2 k] = Re W, x[2 k + 1] = -Im W
            T xa = Z.x * sc, xb = -Z.y * sc;
            if (a.win) {
                xa *= ((const T *)a.win)[2u * k];
                xb *= ((const T *)a.win)[2u * k + 1u];
            }
            T *o = (T *)a.out + (size_t)b * a.out_img + (size_t)(s0 + s) * a.out_ss + 2u * (size_t)k;
            o[0] = xa;
            o[1] = xb;
        } else if constexpr (HERM) {
            T xa = Z.x * sc, xb = -Z.y * sc;
            if (a.win) {
                const T w = ((const T *)a.win)[k];
                xa *= w;
                xb *= w;
            }
            T *o = (T *)a.out + (size_t)b * a.out_img + (size_t)(2u * (s0 + s)) * a.out_ss + (size_t)k * a.out_is;
            o[0] = xa;
            if (2u * (s0 + s) + 1u < a.nrows) o[a.out_ss] = xb;
        } else {
            ((V *)a.out + (size_t)b * a.out_img)[(size_t)(s0 + s) * a.out_ss + (size_t)k * a.out_is] = Z * (V){sc, cj * sc};
        }
    }
}

// M = 8192 / 16384 (n_fft 2049 ... 8192): one sequence per workgroup, 32-point first and last pass; 64 / 128 KiB of LDS (f64: 8192 only)
#define SGX_BS_SPLITS_BIG_F32(X) X(32, 16, 16) X(32, 32, 16)
#define SGX_BS_SPLITS_BIG_F64(X) X(32, 16, 16)
constexpr size_t kBsBigLds = 136 * 1024;

size_t bs_lds_budget(int dtype, unsigned M) { return M > 4096 ? kBsBigLds : dtype == SGX_F64 ? (size_t)SGX_BS_LDS64 : (size_t)SGX_BS_LDS32; }

bool bs_split(unsigned M, int dtype, unsigned &fa, unsigned &fb, unsigned &fc) {
    if (M == 8192) { fa = 32; fb = 16; fc = 16; return true; }
    if (M == 16384 && dtype == SGX_F32) { fa = 32; fb = 32; fc = 16; return true; }
    return M <= 4096 && reg_split_len(M, dtype, &fa, &fb, &fc);
}

bool fused_geometry(unsigned M, int dtype, unsigned &fa, unsigned &fb, unsigned &fc, unsigned &ltile, size_t &lds) {
    if (M & (M - 1)) return false;
    if (!bs_split(M, dtype, fa, fb, fc)) return false;
    const size_t es = dtype == SGX_F64 ? 8 : 4;
    const size_t fs = rr_frame_stride(fa, rr_swizzle(2 * (unsigned)es, fa, fb, fc).rs);
    const size_t budget = bs_lds_budget(dtype, M);
    ltile = 4;  // up to 16 pairs = 32 frames per workgroup
    while (ltile > 0 && (size_t)(1u << ltile) * fs * 2 * es > budget) --ltile;
    lds = (size_t)(1u << ltile) * fs * 2 * es;
    return lds <= budget;
}

template <typename T, int A, int B, int C>
hipError_t launch_fused_t(const BsFused &f, unsigned ltile, size_t lds, hipStream_t s) {
    if (lds > 64 * 1024) {
        hipError_t e = set_max_dynamic_lds((const void *)k_bs_fused<T, A, B, C>, (int)bs_lds_budget(sizeof(T) == 8 ? SGX_F64 : SGX_F32, A * B * C));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_bs_fused<T, A, B, C>), dim3(xcd_grid(f.tiles)), dim3(256), lds, s, f, ltile);
    return hipGetLastError();
}

hipError_t run_fused(const BsArgs &a, int dtype, hipStream_t s) {
    unsigned fa, fb, fc, ltile;
    size_t lds;
    if (!fused_geometry(a.M, dtype, fa, fb, fc, ltile, lds)) return hipErrorNotSupported;
    BsFused f{};
    f.x = a.x; f.out = a.out;
    f.sample_stride = a.sample_stride; f.n_samples = a.n_samples;
    f.n = a.n_fft; f.hop = a.hop; f.pad = a.pad; f.n_frames = a.n_frames; f.nb = a.nb;
    f.pairs = (a.n_frames + 1u) / 2u;
    f.total = (unsigned long long)a.batch * f.pairs;
    const unsigned long long tiles = (f.total + (1ull << ltile) - 1) >> ltile;
    if (tiles == 0 || tiles >= 0x7fffffffull) return hipErrorInvalidConfiguration;
    f.tiles = (unsigned)tiles;
    f.wc = a.wc; f.chirp = a.chirp; f.bhp = a.bhat_fused; f.tw = a.tw_m;
    f.complex_out = a.complex_out; f.amp = a.amp; f.eps = a.eps;
    f.mel_ptr = a.mel_ptr; f.mel_col = a.mel_col; f.mel_val = a.mel_val; f.n_mels = a.n_mels; f.n_out = a.n_out;
#define SGX_BSF_F32(A, B, C) if (fa == A && fb == B && fc == C) return launch_fused_t<float, A, B, C>(f, ltile, lds, s);
#define SGX_BSF_F64(A, B, C) if (fa == A && fb == B && fc == C) return launch_fused_t<double, A, B, C>(f, ltile, lds, s);
    if (dtype == SGX_F64) {
        SGX_RR_SPLITS_F64(SGX_BSF_F64)
        SGX_BS_SPLITS_BIG_F64(SGX_BSF_F64)
    } else {
        SGX_RR_SPLITS_F32(SGX_BSF_F32)
        SGX_BS_SPLITS_BIG_F32(SGX_BSF_F32)
    }
#undef SGX_BSF_F32
#undef SGX_BSF_F64
    return hipErrorNotSupported;
}

template <typename T, int A, int B, int C>
hipError_t launch_bsc_t(const BsC2c &f, int rmode, unsigned ltile, size_t lds, hipStream_t s) {
    const void *fn = rmode == 3 ? (const void *)k_bs_c2c<T, A, B, C, 3> : rmode == 2 ? (const void *)k_bs_c2c<T, A, B, C, 2> : rmode == 1 ? (const void *)k_bs_c2c<T, A, B, C, 1> : (const void *)k_bs_c2c<T, A, B, C, 0>;
    if (lds > 64 * 1024) {
        hipError_t e = set_max_dynamic_lds(fn, (int)bs_lds_budget(sizeof(T) == 8 ? SGX_F64 : SGX_F32, A * B * C));
        if (e != hipSuccess) return e;
    }
    const dim3 grid(xcd_grid(f.total_tiles));
    if (rmode == 3) hipLaunchKernelGGL((k_bs_c2c<T, A, B, C, 3>), grid, dim3(256), lds, s, f, ltile);
    else if (rmode == 2) hipLaunchKernelGGL((k_bs_c2c<T, A, B, C, 2>), grid, dim3(256), lds, s, f, ltile);
    else if (rmode == 1) hipLaunchKernelGGL((k_bs_c2c<T, A, B, C, 1>), grid, dim3(256), lds, s, f, ltile);
    else hipLaunchKernelGGL((k_bs_c2c<T, A, B, C, 0>), grid, dim3(256), lds, s, f, ltile);
    return hipGetLastError();
}

hipError_t run_bsc(BsC2c f, int herm, unsigned M, unsigned batch, int dtype, hipStream_t s) {
    unsigned fa, fb, fc, ltile;
    size_t lds;
    if (!fused_geometry(M, dtype, fa, fb, fc, ltile, lds)) return hipErrorNotSupported;
    ltile = std::min(ltile + 1u, 5u);  // (fused_geometry caps at 16 frame pairs; sequences: up to 32, no more than the job has)
    const size_t es = dtype == SGX_F64 ? 8 : 4;
    const size_t fs = rr_frame_stride(fa, rr_swizzle(2 * (unsigned)es, fa, fb, fc).rs);
    while (ltile > 0 && ((size_t)(1u << ltile) * fs * 2 * es > bs_lds_budget(dtype, M) || (1u << (ltile - 1)) >= f.nseq)) --ltile;
    lds = (size_t)(1u << ltile) * fs * 2 * es;
    f.tiles = (f.nseq + (1u << ltile) - 1) >> ltile;
    const unsigned long long total = (unsigned long long)f.tiles * batch;
    if (total == 0 || total >= 0x7fffffffull) return hipErrorInvalidConfiguration;
    f.total_tiles = (unsigned)total;
#define SGX_BSC_F32(A, B, C) if (fa == A && fb == B && fc == C) return launch_bsc_t<float, A, B, C>(f, herm, ltile, lds, s);
#define SGX_BSC_F64(A, B, C) if (fa == A && fb == B && fc == C) return launch_bsc_t<double, A, B, C>(f, herm, ltile, lds, s);
    if (dtype == SGX_F64) {
        SGX_RR_SPLITS_F64(SGX_BSC_F64)
        SGX_BS_SPLITS_BIG_F64(SGX_BSC_F64)
    } else {
        SGX_RR_SPLITS_F32(SGX_BSC_F32)
        SGX_BS_SPLITS_BIG_F32(SGX_BSC_F32)
    }
#undef SGX_BSC_F32
#undef SGX_BSC_F64
    return hipErrorNotSupported;
}

}  // namespace

// the (A, B, C) split of the kernel at convolution length M, or false: no chirp-z at this length and type
bool bluestein_fused_split(unsigned M, int dtype, unsigned *fa, unsigned *fb, unsigned *fc) {
    unsigned ltile;
    size_t lds;
    return fused_geometry(M, dtype, *fa, *fb, *fc, ltile, lds);
}

hipError_t launch_bluestein(const BsArgs &a, int dtype, hipStream_t s) { return run_fused(a, dtype, s); }

// Host side of the tables for transform length n: M, conj(c) [n], FFT_M(b) / M in the kernel's product order [M], W_M [M] — all as
// interleaved (re, im) doubles for the caller to cast and upload.  false: no chirp-z at this length and type.
bool bluestein_host_tables(unsigned n, int dtype, BsHostTables &t) {
    if (n < 2 || n > 16384u) return false;  // (M <= 32768 below: the 32-bit loop cannot overflow; no fused geometry exists past M = 16384)
    unsigned M = 1, l2 = 0;
    while (M < 2 * n - 1) { M <<= 1; ++l2; }
    unsigned fa, fb, fc;
    if (!bluestein_fused_split(M, dtype, &fa, &fb, &fc)) return false;
    constexpr double kPi = 3.14159265358979323846264338327950288;
    t.M = M;
    t.chirp.assign(2 * size_t(n), 0.0);
    t.bhp.assign(2 * size_t(M), 0.0);
    t.tw.assign(2 * size_t(M), 0.0);
    std::vector<double> bre(M, 0.0), bim(M, 0.0);
    for (unsigned j = 0; j < n; ++j) {
        // c_j = e^(+i pi j^2 / n): the angle is reduced in integers, j^2 mod 2 n, so that a large j loses nothing
        const unsigned long long q = (unsigned long long)j * j % (2ull * n);
        const double ang = kPi * double(q) / double(n), re = std::cos(ang), im = std::sin(ang);
        t.chirp[2 * j] = re;
        t.chirp[2 * j + 1] = -im;  // conj(c_j): multiplies the samples going in and the bins coming out
        bre[j] = re;
        bim[j] = im;
        if (j) { bre[M - j] = re; bim[M - j] = im; }  // b[-j] = c_j
    }
    // FFT_M(b) on the host: iterative radix-2, f64
    for (unsigned i = 1, j = 0; i < M; ++i) {
        unsigned bit = M >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(bre[i], bre[j]); std::swap(bim[i], bim[j]); }
    }
    for (unsigned len = 2; len <= M; len <<= 1) {
        const double ang = -2.0 * kPi / double(len);
        for (unsigned i = 0; i < M; i += len)
            for (unsigned k = 0; k < len / 2; ++k) {
                const double wr = std::cos(ang * k), wi = std::sin(ang * k);
                const double ur = bre[i + k], ui = bim[i + k];
                const double vr = bre[i + k + len / 2] * wr - bim[i + k + len / 2] * wi;
                const double vi = bre[i + k + len / 2] * wi + bim[i + k + len / 2] * wr;
                bre[i + k] = ur + vr; bim[i + k] = ui + vi;
                bre[i + k + len / 2] = ur - vr; bim[i + k + len / 2] = ui - vi;
            }
    }
    for (unsigned k = 0; k < M; ++k) {
        // the inverse transform's 1 / M folded in; [k3][k1][k2] for bin k1 + A (k2 + B k3), [k2][k1] for the two-pass splits
        const unsigned k1 = k % fa, k2 = (k / fa) % fb, k3 = k / (fa * fb);
        const size_t at = fc > 1 ? (size_t(k3) * fa + k1) * fb + k2 : size_t(k2) * fa + k1;
        t.bhp[2 * at] = bre[k] / double(M);
        t.bhp[2 * at + 1] = bim[k] / double(M);
        const double a = -2.0 * kPi * double(k) / double(M);
        t.tw[2 * k] = std::cos(a);
        t.tw[2 * k + 1] = std::sin(a);
    }
    return true;
}

// forward STFT frames of an even n_fft in half-length complex form (t: tables of length n_fft / 2; twn: e^(-2 pi i k / n_fft))
hipError_t launch_bluestein_half(const BsArgs &a, const BsDevTables &t, const void *window, const void *twn, int dtype, hipStream_t s) {
    if ((a.n_fft & 1u) || a.mel_ptr) return hipErrorNotSupported;
    BsC2c f{};
    f.in = a.x; f.out = a.out; f.n = a.n_fft / 2u; f.nseq = a.n_frames; f.nrows = a.n_frames;
    f.in_img = a.sample_stride; f.out_img = (unsigned long long)a.nb * a.n_frames; f.out_is = a.n_frames; f.out_ss = 1;
    f.in_seq_fast = 0; f.out_seq_fast = 1; f.scale = 1.0;
    f.chirp = t.chirp; f.bhp = t.bhp; f.tw = t.tw; f.win = window; f.twn = twn;
    f.hop = a.hop; f.pad = a.pad; f.n_samples = a.n_samples; f.complex_out = a.complex_out; f.amp = a.amp; f.eps = a.eps;
    return run_bsc(f, 3, t.M, a.batch, dtype, s);
}

// complex sequences with C2cArgs' addressing (a.tw / a.tile / a.tiles / a.log2n / a.mul are not used)
hipError_t launch_c2c_bluestein(const C2cArgs &a, const BsDevTables &t, int dtype, hipStream_t s) {
    if (a.mul) return hipErrorNotSupported;
    BsC2c f{};
    f.in = a.in; f.out = a.out; f.n = a.n; f.nseq = a.nseq;
    f.in_img = a.in_img; f.out_img = a.out_img; f.in_ss = a.in_ss; f.in_is = a.in_is; f.out_ss = a.out_ss; f.out_is = a.out_is;
    f.inverse = a.inverse; f.in_seq_fast = a.in_seq_fast; f.out_seq_fast = a.out_seq_fast; f.scale = a.scale;
    f.chirp = t.chirp; f.bhp = t.bhp; f.tw = t.tw;
    return run_bsc(f, 0, t.M, a.batch, dtype, s);
}

// Complex sequences of an EVEN length n = 2 m whose own convolution does not fit LDS (f64 4098 ... 8192, f32 8194 ... 16384): one
// radix-2 step outside — the even and the odd elements as two chirp-z transforms of length m (t: tables of length m) into
// `scratch` ([batch][2][nseq][m] complex), then X[k] = E[k] + W_n^k O[k], X[k + m] = E[k] - W_n^k O[k] (inverse: conj W) with the
// caller's output strides and scale.
template <typename T>
__global__ __launch_bounds__(256) void k_bs_combine(const typename PairOf<T>::type *eo, typename PairOf<T>::type *out, const typename PairOf<T>::type *twn,
                                                    unsigned m, unsigned nseq, unsigned long long total, unsigned long long out_img,
                                                    unsigned long long out_ss, unsigned long long out_is, int seq_fast, int inverse, T scale) {
    typedef typename PairOf<T>::type V;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * 256u) {
        const unsigned long long per = (unsigned long long)nseq * m, b = i / per, r = i - b * per;
        const unsigned q = seq_fast ? (unsigned)(r % nseq) : (unsigned)(r / m), k = seq_fast ? (unsigned)(r / nseq) : (unsigned)(r % m);
        const V E = eo[(b * 2u * nseq + q) * m + k], O = eo[((b * 2u + 1u) * nseq + q) * m + k];
        const V w = twn[k];
        const V t = inreg::cmulv(O, (V){w.x, inverse ? -w.y : w.y});
        V *o = out + b * out_img + q * out_ss + k * out_is;
        o[0] = (E + t) * (V){scale, scale};
        o[(unsigned long long)m * out_is] = (E - t) * (V){scale, scale};
    }
}

hipError_t launch_c2c_bluestein_split(const C2cArgs &a, const BsDevTables &t, void *scratch, int dtype, hipStream_t s) {
    if (a.mul || (a.n & 1u) || !scratch) return hipErrorNotSupported;
    const unsigned m = a.n / 2u;
    const size_t es = dtype == SGX_F64 ? 8 : 4;
    C2cArgs h = a;
    h.n = m; h.in_is = 2ull * a.in_is; h.out_img = 2ull * a.nseq * m; h.out_ss = m; h.out_is = 1; h.out_seq_fast = 0; h.scale = 1.0;
    for (unsigned par = 0; par < 2u; ++par) {
        h.in = (const char *)a.in + (size_t)par * a.in_is * 2 * es;
        h.out = (char *)scratch + (size_t)par * a.nseq * m * 2 * es;
        const hipError_t e = launch_c2c_bluestein(h, t, dtype, s);
        if (e != hipSuccess) return e;
    }
    const unsigned long long total = (unsigned long long)a.batch * a.nseq * m;
    const unsigned blocks = (unsigned)std::min<unsigned long long>((total + 255) / 256, 65536ull);
    if (dtype == SGX_F64)
        hipLaunchKernelGGL(k_bs_combine<double>, dim3(blocks), dim3(256), 0, s, (const inreg::v2d *)scratch, (inreg::v2d *)a.out, (const inreg::v2d *)a.tw, m, a.nseq,
                           total, a.out_img, a.out_ss, a.out_is, a.out_seq_fast, a.inverse, a.scale);
    else
        hipLaunchKernelGGL(k_bs_combine<float>, dim3(blocks), dim3(256), 0, s, (const inreg::v2f *)scratch, (inreg::v2f *)a.out, (const inreg::v2f *)a.tw, m, a.nseq,
                           total, a.out_img, a.out_ss, a.out_is, a.out_seq_fast, a.inverse, (float)a.scale);
    return hipGetLastError();
}

// half spectrum -> real rows with C2rArgs' addressing (rows = sequences)
// `half`: the tables are those of length ncols / 2 (even ncols): the half-length complex form, one row per sequence
hipError_t launch_c2r_bluestein(const C2rArgs &a, const BsDevTables &t, int dtype, hipStream_t s, bool half) {
    if (a.nbk) return hipErrorNotSupported;  // (the fused overlap-add belongs to k_c2r_reg)
    if (half) {
        if (a.ncols & 1u) return hipErrorInvalidValue;
        BsC2c h{};
        h.in = a.in; h.out = a.out; h.n = a.ncols / 2u; h.nseq = a.nrows; h.nrows = a.nrows;
        h.in_img = a.in_img; h.out_img = (unsigned long long)a.nrows * a.ncols; h.in_ss = a.in_rs; h.in_is = a.in_ks; h.out_ss = a.ncols; h.out_is = 1;
        h.inverse = 1; h.in_seq_fast = a.k_fast ? 0 : 1; h.out_seq_fast = 0; h.scale = a.scale;
        h.chirp = t.chirp; h.bhp = t.bhp; h.tw = t.tw; h.win = a.win; h.bad_flag = a.bad_flag; h.twn = a.tw;
        return run_bsc(h, 2, t.M, a.batch, dtype, s);
    }
    BsC2c f{};
    f.in = a.in; f.out = a.out; f.n = a.ncols; f.nseq = (a.nrows + 1u) / 2u; f.nrows = a.nrows;
    f.in_img = a.in_img; f.out_img = (unsigned long long)a.nrows * a.ncols; f.in_ss = a.in_rs; f.in_is = a.in_ks; f.out_ss = a.ncols; f.out_is = 1;
    f.inverse = 1; f.in_seq_fast = a.k_fast ? 0 : 1; f.out_seq_fast = 0; f.scale = a.scale;
    f.chirp = t.chirp; f.bhp = t.bhp; f.tw = t.tw; f.win = a.win; f.bad_flag = a.bad_flag;
    return run_bsc(f, 1, t.M, a.batch, dtype, s);
}


}  // namespace sgx

#ifdef SGX_BS_STAMPS
extern "C" int sgx_debug_read_bs_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sgx::g_bs_stamps), sizeof(sgx::g_bs_stamps)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(sgx::g_bs_stamps), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif
