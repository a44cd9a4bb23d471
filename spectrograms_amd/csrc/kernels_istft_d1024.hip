// kernels_istft_d1024.hip — fused inverse STFT for f64, n_fft = 1024 (round 4): k_istft2048's dataflow at 512 complex f64 points (the same
// bytes per frame), with k_d32x16's lane pairs: one persistent 512-thread workgroup per CU on tiles of 16 new frames
// (src/spectrogram.rs:4860-4946: C2R per frame with 1/n, window, overlap-add in ascending frame order, sum of w^2 normalisation where > 1e-10,
// centre trim).
//
//   A  lane (half, job r, frame): kb = r (half 0) or 32 - r (half 1; row 0: 0 and 16).  The 8 pairs (X[k], X[512 - k]), k = kb + 32 u, folded
//      once: S = P + conj Q, T = conj(W_1024^k)(P - conj Q), v[k] = conj(S + i T), v[512 - k] = S - i T with v = conj(Z').  v[kb + 32 u] are
//      the first 8 EVEN-indexed elements of row r (half 0) / ODD-indexed elements of row 16 - r (half 1) of v[k1 + 16 k2]; the last 8 are the
//      mirrored values the PARTNER lane (l ^ 32) has just folded: traded with v_permlane32_swap_b32 (row 0 keeps its own).  A 16-point
//      transform gives E[r][n] / O[16 - r][n], written to ex[f][k1][E | O][16].
//   B  lane (f, n2 = 0..31): column n2 of the 16 rows: u[k1] = E[k1][n2 mod 16] +- W_32^(n2 mod 16) O[k1][n2 mod 16] (the last radix-2 step of
//      the row transform), twiddle W_512^(k1 n2), 16-point transform over k1: y[n2 + 32 n1] -> (x[2n], x[2n+1]) = conj(y) / 1024, times
//      the window, real frames fr[f][1024] over the dead ex.
//   C  overlap-add with the carry of k_istft1024c (kernels_c2c1024.hip), in f64.
#include <algorithm>

#include "fft_inreg.h"
#include "lane_pair.h"
#include "sgx_internal.h"

namespace sgx {
namespace {

using namespace inreg;

typedef unsigned v2u __attribute__((ext_vector_type(2)));

struct IstDArgs {
    const void *spec;  // [batch][513][n_frames] complex f64
    void *out;         // [batch][out_len] f64
    const void *win;   // [1024] f64
    unsigned n_frames, hop, batch, tiles, ov;
    unsigned long long start, out_len;
    double scale;
    unsigned *bad_flag;
};

constexpr int kIDFS = 8192 + 16;         // bytes per frame of ex[f][16][E 16 | O 16] of 16-byte elements
constexpr int kIDTw = 16 * kIDFS;        // 131 328: conj(W_1024^k), k < 512 (8192 B)
constexpr int kIDWin = kIDTw + 8192;     // the window (8192 B)
constexpr int kIDCarry = kIDWin + 8192;  // the carry, ov * hop <= 1023 doubles
constexpr int kIDLds = kIDCarry + 8192;  // 155 904 B: one workgroup per CU

using lanepair::trade32;  // lanes l and l ^ 32 trade a complex value, each receives the other's as (im, re): lane_pair.h

// (NFFT = frame length, FPT = frames per tile: 1024 / 16 for k_istft_d1024, 512 / 32 for k_istft_d512.)
// Interior tiles at hop NFFT / 4, / 2, / 1: every frame index a compile-time constant.  HOP >= NT: an offset belongs to one thread; HOP < NT:
// NT / HOP threads share an offset and take every (NT / HOP)-th hop block — a carry slot is read and rewritten by the same thread either
// way, so there is no barrier between the carry's reads and writes.
template <unsigned NFFT, unsigned FPT, unsigned HOP, unsigned NT>
__device__ __forceinline__ void olad_fast(const IstDArgs &a, const unsigned char *smem, const double *w, double *carry, unsigned tid, unsigned b,
                                          unsigned F, bool store) {
    constexpr unsigned Q = NFFT / HOP, OV = Q - 1u;
    constexpr unsigned REP = HOP < NT ? NT / HOP : 1u, K = HOP >= NT ? HOP / NT : 1u;
    static_assert((HOP >= NT && HOP % NT == 0) || (HOP < NT && NT % HOP == 0 && FPT % REP == 0), "fast overlap-add: whole owners per offset");
    const double *fr = (const double *)smem;
    double *o = (double *)a.out + (size_t)b * a.out_len + ((unsigned long long)F * HOP - a.start);
    const unsigned g = REP > 1u ? tid / HOP : 0u, off0 = REP > 1u ? tid - g * HOP : tid;
#pragma unroll
    for (unsigned k = 0; k < K; ++k) {
        const unsigned off = off0 + k * NT;
        if (store) {
            double nrm = 0.0;  // ascending frame = descending sample index: the reference's order
#pragma unroll
            for (unsigned i = Q; i-- > 0;) {
                const double wj = w[i * HOP + off];
                nrm = __dadd_rn(nrm, __dmul_rn(wj, wj));
            }
            const bool div = nrm > 1e-10;
#pragma unroll
            for (unsigned hi = 0; hi < FPT / REP; ++hi) {
                const unsigned hb = g + hi * REP;
                double acc = hb < OV ? carry[hb * HOP + off] : 0.0;
#pragma unroll
                for (unsigned d = OV + 1u; d-- > 0;)
                    if (d <= hb) acc += fr[(hb - d) * NFFT + d * HOP + off];  // frames hb - d, ascending
                o[hb * HOP + off] = div ? acc / nrm : acc;
            }
        }
#pragma unroll
        for (unsigned hb2 = 0; hb2 < OV; ++hb2) {
            if (REP > 1u && hb2 % REP != g) continue;
            double acc = 0.0;
#pragma unroll
            for (unsigned d = OV; d > hb2; --d) acc += fr[(FPT + hb2 - d) * NFFT + d * HOP + off];  // rows FPT + hb2 - d < FPT
            carry[hb2 * HOP + off] = acc;
        }
    }
}

// general walk (any hop >= 64, edge tiles): istft_ola_carry of kernels_c2c1024.hip in f64
template <unsigned NFFT, unsigned FPT, unsigned NT>
__device__ __forceinline__ void olad_carry(const IstDArgs &a, const unsigned char *smem, const double *w, double *carry, unsigned tid, unsigned b,
                                           unsigned F, bool store) {
    const double *fr = (const double *)smem;
    double *o = (double *)a.out + (size_t)b * a.out_len;
    const unsigned hop = a.hop, ov = a.ov;
    const unsigned long long p0 = (unsigned long long)F * hop;
    const bool interior = F >= ov && F + (FPT - 1u) < a.n_frames && p0 >= a.start && p0 + (unsigned long long)FPT * hop <= a.start + a.out_len;
    if (interior) {  // (uniform)
        if (hop == NFFT / 4u) return olad_fast<NFFT, FPT, NFFT / 4u, NT>(a, smem, w, carry, tid, b, F, store);
        if (hop == NFFT / 2u) return olad_fast<NFFT, FPT, NFFT / 2u, NT>(a, smem, w, carry, tid, b, F, store);
        if (hop == NFFT) return olad_fast<NFFT, FPT, NFFT, NT>(a, smem, w, carry, tid, b, F, store);
    }
    const bool small = hop < NT;
    const unsigned nrep = small ? NT / hop : 1u, g = small ? tid / hop : 0u, ostep = small ? hop : NT;
    const unsigned off0 = small ? tid - g * hop : tid;
    if (g < nrep && store) {
        for (unsigned off = off0; off < hop; off += ostep) {
            const unsigned q = (NFFT - off + hop - 1u) / hop, back = q - 1u;  // frames h - back .. h overlap this offset
            double nrm_full = 0.0;
            for (unsigned i = q; i-- > 0;) {
                const double wj = w[i * hop + off];
                nrm_full = __dadd_rn(nrm_full, __dmul_rn(wj, wj));
            }
            for (unsigned hb = g; hb < FPT; hb += nrep) {
                const unsigned h = F + hb;
                const unsigned long long pos = p0 + (unsigned long long)hb * hop + off;
                double acc = hb < back ? carry[hb * hop + off] : 0.0;
                const unsigned r_lo = hb < back ? 0u : hb - back;
                const double *src = fr + r_lo * NFFT + (hb - r_lo) * hop + off;
                for (unsigned r = r_lo; r <= hb; ++r) {  // next frame: row + 1, sample index - hop
                    acc += *src;
                    src += (int)NFFT - (int)hop;
                }
                double nrm = nrm_full;
                if (!interior) {
                    if (pos < a.start || pos - a.start >= a.out_len) continue;
                    const long long f_lo = (long long)h - (long long)back < 0 ? 0ll : (long long)h - (long long)back;
                    const long long f_hi = h < a.n_frames ? (long long)h : (long long)a.n_frames - 1;
                    if ((unsigned)(f_hi - f_lo + 1) != q || f_hi < f_lo) {  // signal edges: only the frames that exist count, ascending
                        nrm = 0.0;
                        for (long long f = f_lo; f <= f_hi; ++f) {
                            const double wj = w[(unsigned)((long long)h - f) * hop + off];
                            nrm = __dadd_rn(nrm, __dmul_rn(wj, wj));
                        }
                    }
                }
                if (nrm > 1e-10) acc /= nrm;
                o[pos - a.start] = acc;
            }
        }
    }
    __syncthreads();  // every carry value has been read
    if (g < nrep) {
        for (unsigned off = off0; off < hop; off += ostep) {
            const unsigned q = (NFFT - off + hop - 1u) / hop, back = q - 1u;
            for (unsigned hb2 = g; hb2 < ov; hb2 += nrep) {
                const unsigned hb = FPT + hb2;
                double acc = 0.0;
                if (hb <= FPT - 1u + back) {  // the offset reaches back into this tile: rows hb - back .. FPT - 1
                    const unsigned r_lo = hb - back;
                    const double *src = fr + r_lo * NFFT + (hb - r_lo) * hop + off;
                    for (unsigned r = r_lo; r < FPT; ++r) {
                        acc += *src;
                        src += (int)NFFT - (int)hop;
                    }
                }
                carry[hb2 * hop + off] = acc;
            }
        }
    }
}

__global__ __launch_bounds__(512, 2) void k_istft_d1024(IstDArgs a, const v2d *twr, const v2d *tw1, unsigned per_xcd, unsigned total_runs, unsigned slots,
                                                        unsigned runs_per_signal, unsigned run_len) {
    constexpr unsigned NT = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x;
    v2d *twl = (v2d *)(smem + kIDTw);  // conj(W_1024^k), k < 512
    twl[tid] = twr[tid];
    ((v2d *)(smem + kIDWin))[tid] = ((const v2d *)a.win)[tid];
    double *carry = (double *)(smem + kIDCarry);
    __syncthreads();
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned lo = xcd * per_xcd, hi = min(lo + per_xcd, total_runs);
    const unsigned lane = tid & 63u, half = lane >> 5, fl = lane & 15u;
    const unsigned r = (tid >> 6) + 8u * ((lane >> 4) & 1u);
    const bool j0 = r == 0u;
    const unsigned kb = half ? (j0 ? 16u : 32u - r) : r;     // this lane's pairs: bins kb + 32 u and 512 - kb - 32 u
    const unsigned rowW = half ? ((16u - r) & 15u) : r;      // the half row it transforms: E plane of row r / O plane of row 16 - r
    const unsigned nf16 = a.n_frames * 16u;
    const unsigned st = 32u * nf16;  // byte offsets are stepped by 32 bins
    v2d P[8], Q[8], X256 = {0.0, 0.0};
    auto request = [&](unsigned b, unsigned t) {
        const unsigned f = 16u * t + fl;
        const unsigned fcl = f < a.n_frames ? f : 0u;  // a frame past the signal reads frame 0 and is replaced by zeros in the fold
        const unsigned char *inb = (const unsigned char *)a.spec + (size_t)b * 513u * a.n_frames * 16u;
        unsigned oa = kb * nf16 + fcl * 16u, oy = (512u - kb) * nf16 + fcl * 16u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            P[u] = *(const v2d *)(inb + oa);
            Q[u] = *(const v2d *)(inb + oy);
            oa += st;
            oy -= st;
        }
        if (kb == 0u) X256 = *(const v2d *)(inb + 256u * nf16 + fcl * 16u);  // bin 256 pairs with itself (row 0, half 0)
    };
    auto run_of = [&](unsigned rid, unsigned &b, unsigned &t0, unsigned &t1, unsigned &ts) {
        b = rid / runs_per_signal;
        t0 = (rid - b * runs_per_signal) * run_len;
        t1 = min(a.tiles, t0 + run_len);
        ts = (t0 > 0u && a.ov) ? t0 - 1u : t0;
    };
    unsigned rid = lo + slot, b = 0, t0 = 0, t1 = 0, t = 0;
    if (rid < hi) {
        run_of(rid, b, t0, t1, t);
        if (t0 >= t1) rid = hi;
    }
    bool fresh = true;  // the run has just started: its carry is zero
    if (rid < hi) request(b, t);
    const unsigned f2 = tid >> 5, n2 = tid & 31u, nl = n2 & 15u;  // stage-B identity
    while (rid < hi) {
        const unsigned F = 16u * t;
        unsigned nrid = rid, nb = b, nt0 = t0, nt1 = t1, nt = t + 1u;
        if (nt >= t1) {
            nrid = rid + slots;
            if (nrid < hi) run_of(nrid, nb, nt0, nt1, nt);
        }
        if (fresh) {
#pragma unroll
            for (int q = 0; q < 2; ++q) carry[tid + 512u * q] = 0.0;  // (ordered before its first use by the barriers below)
        }
        {
            const unsigned f = F + fl;
            const bool valid = f < a.n_frames;
            if (!valid) {  // a frame past the signal is zeros by a select: the frame 0 loaded in its place may hold Inf / NaN, which a product with 0 would spread into the tail (the reference poisons only the samples frame 0 covers, spectrogram.rs:4906-4925)
#pragma unroll
                for (int u = 0; u < 8; ++u) P[u] = Q[u] = (v2d){0.0, 0.0};
                X256 = (v2d){0.0, 0.0};
            }
            v2d H[16], QB[8];
            const v2d *tp = twl + kb;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v2d Pp = P[u], Qp = Q[u];
                if (u == 0) {  // kb = 0: bins 0 and 512 — realfft ignores (and reports) their imaginary parts
                    if (kb == 0u) {
                        if (a.bad_flag && valid && (Pp.y != 0.0 || Qp.y != 0.0)) atomicOr(a.bad_flag, 1u);
                        Pp.y = 0.0;
                        Qp.y = 0.0;
                    }
                }
                const v2d cw = tp[32 * u];  // conj(W_1024^k)
                const v2d S = pfma(Qp, (v2d){1.0, -1.0}, Pp), D = pfma(Qp, (v2d){-1.0, 1.0}, Pp);
                const v2d T = cmulv(D, cw);
                H[u] = pfma(swp(T), (v2d){-1.0, -1.0}, S * (v2d){1.0, -1.0});  // conj(S + i T) = v[k]
                QB[u] = pfma(swp(T), (v2d){1.0, -1.0}, S);    // S - i T    = v[512 - k]
            }
            const v2d x256 = X256;
            // the upper 8 elements of the half row are the partner's mirrored values: H[8 + t] = partner's QB[7 - t].  Row 0 mirrors inside
            // itself (v[32 m] in half 0, v[16 + 32 m] in half 1): its lanes fill H from their own QB before the trade overwrites it.
            if (j0) {
                if (half) {
#pragma unroll
                    for (int tq = 0; tq < 8; ++tq) H[8 + tq] = QB[7 - tq];
                } else {
                    H[8] = x256 * (v2d){2.0, 2.0};
#pragma unroll
                    for (int tq = 1; tq < 8; ++tq) H[8 + tq] = QB[8 - tq];
                }
                asm volatile("" ::: "memory");  // keeps this a branch
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) trade32(QB[j]);
            if (!j0) {
#pragma unroll
                for (int tq = 0; tq < 8; ++tq) H[8 + tq] = swp(QB[7 - tq]);
                asm volatile("" ::: "memory");
            }
            Fft<16, false, v2d>::run(H, H);
            v2d *dst = (v2d *)(smem + fl * kIDFS + rowW * 512u + half * 256u);
#pragma unroll
            for (int c = 0; c < 16; ++c) dst[c] = H[c];
            // the next tile's pairs go out now (not right after the fold: 64 more live registers there) and land during stages B and C
            if (nrid < hi) request(nb, nt);
        }
        __syncthreads();  // ex complete
        v2d v[16];
        {
            v2d twa[4], twb[4];  // W_512^(k1 n2) = twa[k1 >> 2] * twb[k1 & 3] (loaded per tile: held across the fold they cost 24 registers)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                twa[q] = tw1[32 * 4 * q + n2];
                twb[q] = tw1[32 * q + n2];
            }
            // the row transform's last radix-2 step: u = E + c O, c = +- W_32^(n2 mod 16) (= W_512^(8 * 2 nl), a table entry)
            const double sgn = n2 < 16u ? 1.0 : -1.0;
            const v2d c32 = tw1[32 * 8 + 2u * nl] * (v2d){sgn, sgn};
            const unsigned char *src = smem + f2 * kIDFS + nl * 16u;
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) {
                const v2d E = *(const v2d *)(src + k1 * 512), O = *(const v2d *)(src + k1 * 512 + 256);
                v2d u = pfma(swp(O), (v2d){-c32.y, c32.y}, pfma(O, lo2(c32), E));  // E + c O
                const int qa = k1 >> 2, qb = k1 & 3;
                if (qb) u = cmulv(u, twb[qb]);
                if (qa) u = cmulv(u, twa[qa]);
                v[k1] = u;
                if ((k1 & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // four rows at a time: all 32 reads in flight are 128 registers
            }
            Fft<16, false, v2d>::run(v, v);
        }
        __syncthreads();  // exchange buffer consumed: overlay the real frames
        {
            const v2d *w2 = (const v2d *)(smem + kIDWin) + n2;
            v2d *fr2 = (v2d *)smem + f2 * 512u + n2;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const v2d ww = w2[32 * n1];
                const v2d sc = v[n1] * (v2d){a.scale, -a.scale};  // conj + 1/n: (x[2n], x[2n+1]), n = n2 + 32 n1
                fr2[32 * n1] = (v2d){__dmul_rn(sc.x, ww.x), __dmul_rn(sc.y, ww.y)};
            }
        }
        __syncthreads();
        olad_carry<1024, 16, NT>(a, smem, (const double *)(smem + kIDWin), carry, tid, b, F, t >= t0);
        __syncthreads();  // the frames are consumed and the carry is complete
        fresh = nrid != rid;
        rid = nrid; b = nb; t0 = nt0; t1 = nt1; t = nt;
    }
}

// ====================================================================================================================================
// k_istft_d512: f64 n_fft = 512 (the reference's speech default 512 / 160 in its default type), hop >= 32.  TWO frames per 512-point complex
// transform, backwards: with A, B the half spectra of frames 2 p and 2 p + 1, Z[k] = A[k] + i B[k] and Z[512 - k] = conj A[k] + i conj B[k] is
// the spectrum of a + i b, so stage A packs instead of folding (no twiddles) — v = conj Z: v[k] = (A.x - B.y, -A.y - B.x), v[512 - k] =
// (A.x + B.y, A.y - B.x) — and stages A and B are k_istft_d1024's on a tile of 32 frames in 16 slots; the transform's real part is frame 2 p,
// minus its imaginary part frame 2 p + 1.  Stage C: the same carried overlap-add at a frame length of 512, 32 frames per tile.
constexpr int kI5Win = 16 * kIDFS;       // 131 328: the window (4096 B)
constexpr int kI5Carry = kI5Win + 4096;  // the carry, ov * hop <= 511 doubles
constexpr int kI5Lds = kI5Carry + 4096;  // 139 520 B

__global__ __launch_bounds__(512, 2) void k_istft_d512(IstDArgs a, const v2d *tw1, unsigned per_xcd, unsigned total_runs, unsigned slots,
                                                       unsigned runs_per_signal, unsigned run_len) {
    constexpr unsigned NT = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x;
    ((double *)(smem + kI5Win))[tid] = ((const double *)a.win)[tid];
    double *carry = (double *)(smem + kI5Carry);
    __syncthreads();
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned lo = xcd * per_xcd, hi = min(lo + per_xcd, total_runs);
    const unsigned lane = tid & 63u, half = lane >> 5, sl = lane & 15u;  // sl: the slot (frame pair) of the tile
    const unsigned r = (tid >> 6) + 8u * ((lane >> 4) & 1u);
    const bool j0 = r == 0u;
    const unsigned kb = half ? (j0 ? 16u : 32u - r) : r;  // this lane's bins kb + 32 u (and the mirrored values it derives from them)
    const unsigned rowW = half ? ((16u - r) & 15u) : r;
    const unsigned nf16 = a.n_frames * 16u;
    const unsigned st = 32u * nf16;
    v2d P[8], Q[8], XA = {0.0, 0.0}, XB = {0.0, 0.0};  // P = A[k] (frame 2 p), Q = B[k] (frame 2 p + 1); XA / XB: bin 256 (row 0, half 0)
    auto request = [&](unsigned b, unsigned t) {
        const unsigned fa = 32u * t + 2u * sl;
        const unsigned fca = fa < a.n_frames ? fa : 0u, fcb = fa + 1u < a.n_frames ? fa + 1u : 0u;  // frames past the signal read frame 0 and are zeroed below
        const unsigned char *inb = (const unsigned char *)a.spec + (size_t)b * 257u * a.n_frames * 16u;
        unsigned oa = kb * nf16 + fca * 16u, ob = kb * nf16 + fcb * 16u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            P[u] = *(const v2d *)(inb + oa);
            Q[u] = *(const v2d *)(inb + ob);
            oa += st;
            ob += st;
        }
        if (kb == 0u) {
            XA = *(const v2d *)(inb + 256u * nf16 + fca * 16u);
            XB = *(const v2d *)(inb + 256u * nf16 + fcb * 16u);
        }
    };
    auto run_of = [&](unsigned rid, unsigned &b, unsigned &t0, unsigned &t1, unsigned &ts) {
        b = rid / runs_per_signal;
        t0 = (rid - b * runs_per_signal) * run_len;
        t1 = min(a.tiles, t0 + run_len);
        ts = (t0 > 0u && a.ov) ? t0 - 1u : t0;
    };
    unsigned rid = lo + slot, b = 0, t0 = 0, t1 = 0, t = 0;
    if (rid < hi) {
        run_of(rid, b, t0, t1, t);
        if (t0 >= t1) rid = hi;
    }
    bool fresh = true;
    if (rid < hi) request(b, t);
    const unsigned f2 = tid >> 5, n2 = tid & 31u, nl = n2 & 15u;  // stage-B identity: slot f2, column n2
    while (rid < hi) {
        const unsigned F = 32u * t;
        unsigned nrid = rid, nb = b, nt0 = t0, nt1 = t1, nt = t + 1u;
        if (nt >= t1) {
            nrid = rid + slots;
            if (nrid < hi) run_of(nrid, nb, nt0, nt1, nt);
        }
        if (fresh) carry[tid] = 0.0;  // (ordered before its first use by the barriers below)
        {
            const unsigned fa = F + 2u * sl;
            const bool va = fa < a.n_frames, vb = fa + 1u < a.n_frames;
            v2d H[16], QB[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v2d A = va ? P[u] : (v2d){0.0, 0.0}, B = vb ? Q[u] : (v2d){0.0, 0.0};  // selects, not products with 0: a non-finite stand-in frame must not reach the tail
                if (u == 0) {  // kb = 0: bin 0 — realfft ignores (and reports) its imaginary part
                    if (kb == 0u) {
                        if (a.bad_flag && ((va && A.y != 0.0) || (vb && B.y != 0.0))) atomicOr(a.bad_flag, 1u);
                        A.y = 0.0;
                        B.y = 0.0;
                    }
                }
                H[u] = (v2d){A.x - B.y, -A.y - B.x};   // v[k]       = conj(A + i B)
                QB[u] = (v2d){A.x + B.y, A.y - B.x};   // v[512 - k] = conj(conj A + i conj B)
            }
            v2d x256 = {0.0, 0.0};
            if (kb == 0u) {  // bin 256 (real in both frames)
                if (a.bad_flag && ((va && XA.y != 0.0) || (vb && XB.y != 0.0))) atomicOr(a.bad_flag, 1u);
                x256 = (v2d){va ? XA.x : 0.0, vb ? -XB.x : 0.0};
            }
            if (j0) {  // row 0 mirrors inside itself: filled from the lane's own values before the trade overwrites them
                if (half) {
#pragma unroll
                    for (int tq = 0; tq < 8; ++tq) H[8 + tq] = QB[7 - tq];
                } else {
                    H[8] = x256;
#pragma unroll
                    for (int tq = 1; tq < 8; ++tq) H[8 + tq] = QB[8 - tq];
                }
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) trade32(QB[j]);
            if (!j0) {
#pragma unroll
                for (int tq = 0; tq < 8; ++tq) H[8 + tq] = swp(QB[7 - tq]);
                asm volatile("" ::: "memory");
            }
            Fft<16, false, v2d>::run(H, H);
            v2d *dst = (v2d *)(smem + sl * kIDFS + rowW * 512u + half * 256u);
#pragma unroll
            for (int c = 0; c < 16; ++c) dst[c] = H[c];
            if (nrid < hi) request(nb, nt);  // the next tile's spectra land during stages B and C
        }
        __syncthreads();  // ex complete
        v2d v[16];
        {
            v2d twa[4], twb[4];  // W_512^(k1 n2) = twa[k1 >> 2] * twb[k1 & 3]
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                twa[q] = tw1[32 * 4 * q + n2];
                twb[q] = tw1[32 * q + n2];
            }
            const double sgn = n2 < 16u ? 1.0 : -1.0;
            const v2d c32 = tw1[32 * 8 + 2u * nl] * (v2d){sgn, sgn};
            const unsigned char *src = smem + f2 * kIDFS + nl * 16u;
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) {
                const v2d E = *(const v2d *)(src + k1 * 512), O = *(const v2d *)(src + k1 * 512 + 256);
                v2d u = pfma(swp(O), (v2d){-c32.y, c32.y}, pfma(O, lo2(c32), E));  // E + c O
                const int qa = k1 >> 2, qb = k1 & 3;
                if (qb) u = cmulv(u, twb[qb]);
                if (qa) u = cmulv(u, twa[qa]);
                v[k1] = u;
                if ((k1 & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            Fft<16, false, v2d>::run(v, v);
        }
        __syncthreads();  // exchange buffer consumed: overlay the real frames fr[32][512]
        {
            const double *w1 = (const double *)(smem + kI5Win) + n2;
            double *fa = (double *)smem + (2u * f2) * 512u + n2, *fb = fa + 512;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {  // z[n] = conj(y[n]) / 512, n = n2 + 32 n1: frame 2 p = Re z, frame 2 p + 1 = Im z
                const double ww = w1[32 * n1];
                fa[32 * n1] = __dmul_rn(v[n1].x * a.scale, ww);
                fb[32 * n1] = __dmul_rn(-v[n1].y * a.scale, ww);
            }
        }
        __syncthreads();
        olad_carry<512, 32, NT>(a, smem, (const double *)(smem + kI5Win), carry, tid, b, F, t >= t0);
        __syncthreads();  // the frames are consumed and the carry is complete
        fresh = nrid != rid;
        rid = nrid; b = nb; t0 = nt0; t1 = nt1; t = nt;
    }
}

}  // namespace

hipError_t launch_istft_d1024(const void *spec, void *out, const void *win, unsigned n_frames, unsigned hop, unsigned batch, unsigned long long start,
                              unsigned long long out_len, double scale, unsigned *bad_flag, const void *twr, const void *tw1, hipStream_t s) {
    if (hop == 0 || hop > 1024) return hipErrorInvalidConfiguration;
    IstDArgs a{};
    a.spec = spec; a.out = out; a.win = win;
    a.n_frames = n_frames; a.hop = hop; a.batch = batch;
    a.ov = 1023u / hop;
    if (a.ov >= 16) return hipErrorInvalidConfiguration;  // hop >= 64
    const unsigned long long full = (unsigned long long)(n_frames - 1) * hop + 1024ull;
    const unsigned long long blocks = (full + hop - 1) / hop;
    a.tiles = (unsigned)((blocks + 15u) / 16u);
    a.start = start; a.out_len = out_len; a.scale = scale; a.bad_flag = bad_flag;
    if ((unsigned long long)a.tiles * batch >= 0x7fffffffull || a.tiles == 0) return hipErrorInvalidConfiguration;
    hipError_t e = set_max_dynamic_lds((const void *)k_istft_d1024, kIDLds);
    if (e != hipSuccess) return e;
    // runs of consecutive tiles, one workgroup per CU (istft_carry_runs, sgx_internal.h)
    const unsigned wgs = device_cu_count();
    unsigned R, run_len;
    istft_carry_runs(a.tiles, batch, wgs, a.ov, R, run_len);
    const unsigned total_runs = R * batch, per_xcd = (total_runs + 7u) / 8u;
    const unsigned slots = std::max(1u, std::min(per_xcd, wgs / 8u));
    hipLaunchKernelGGL(k_istft_d1024, dim3(8u * slots), dim3(512), kIDLds, s, a, (const v2d *)twr, (const v2d *)tw1, per_xcd, total_runs, slots, R, run_len);
    return hipGetLastError();
}


hipError_t launch_istft_d512(const void *spec, void *out, const void *win, unsigned n_frames, unsigned hop, unsigned batch, unsigned long long start,
                             unsigned long long out_len, double scale, unsigned *bad_flag, const void *tw1, hipStream_t s) {
    if (hop < 32 || hop > 512) return hipErrorInvalidConfiguration;
    IstDArgs a{};
    a.spec = spec; a.out = out; a.win = win;
    a.n_frames = n_frames; a.hop = hop; a.batch = batch;
    a.ov = 511u / hop;  // <= 15
    const unsigned long long full = (unsigned long long)(n_frames - 1) * hop + 512ull;
    const unsigned long long blocks = (full + hop - 1) / hop;
    a.tiles = (unsigned)((blocks + 31u) / 32u);
    a.start = start; a.out_len = out_len; a.scale = scale; a.bad_flag = bad_flag;
    if ((unsigned long long)a.tiles * batch >= 0x7fffffffull || a.tiles == 0) return hipErrorInvalidConfiguration;
    hipError_t e = set_max_dynamic_lds((const void *)k_istft_d512, kI5Lds);
    if (e != hipSuccess) return e;
    const unsigned wgs = device_cu_count();
    unsigned R, run_len;
    istft_carry_runs(a.tiles, batch, wgs, a.ov, R, run_len);
    const unsigned total_runs = R * batch, per_xcd = (total_runs + 7u) / 8u;
    const unsigned slots = std::max(1u, std::min(per_xcd, wgs / 8u));
    hipLaunchKernelGGL(k_istft_d512, dim3(8u * slots), dim3(512), kI5Lds, s, a, (const v2d *)tw1, per_xcd, total_runs, slots, R, run_len);
    return hipGetLastError();
}

}  // namespace sgx
