// kernels_d32x32.hip — tuned f64, n_fft = 2048 STFT kernel for gfx950 (round 4): per-bin, complex and (up to hop 585) filterbank
// outputs.  The reference's Criterion suite times this shape in this type (benches/stft_benchmarks.rs:11-50: f64, 2048 / 1024).
// k_r64x32's tile at 1024 complex f64 points — 8 consecutive frames of one signal, one persistent 512-thread workgroup per CU, a 128 KiB
// exchange buffer ex[f][k1][n2] of 16-byte elements — with 32 rows of 32 points:
//
//   pass 1  a column z[32 n1 + n2], n1 = 0..31, is a 32-point transform: 128 data registers in f64, so TWO lanes share it the way pass 2 shares
//           a row: lane (par, f, n2) reads the whole column (window fused) and takes one decimation-in-frequency half — (z[n1] + z[n1 + 16])
//           for the even k1 (par 0, waves 0..3), (z[n1] - z[n1 + 16]) W_32^n1 for the odd k1 (par 1, waves 4..7) — then one 16-point FFT;
//           twiddle W_1024^(k1 n2); one ds_write_b128 per value.
//   pass 2  32 rows of 32 points; the even-indexed outputs of row r pair with the odd-indexed outputs of row 32 - r.  A half row (one DIF step
//           as the row is read, then a 16-point FFT) is one lane's work; the partner halves sit in lanes l and l + 32 of one wave and trade the
//           upper 8 values with v_permlane32_swap_b32.  Each lane then splits 8 pairs = 16 bins: own H[u] = Z[kb + 64 u] with the partner's
//           Z[1024 - kb - 64 u], kb = r (half 0) or 64 - r (half 1).  Row 0's halves pair inside themselves (kb = 0, 32).
//           32 rows x 2 halves x 8 frames = the 512 lanes.
//   store   the 8 lanes of a (row, half) hold one bin of 8 consecutive frames: 64-byte runs (128 for the complex STFT).
//
// Samples: the tile's 7 hop + 2048 samples once, staged in LDS over the idle exchange buffer (hop <= 1024); longer hops load their columns per
// lane.  Reference semantics: spectrogram.rs:1301-1334, :2068-2080.
#include <type_traits>
#include <utility>

#include "buffer_ops.h"
#include "fft_inreg.h"
#include "db_f64.h"
#include "lane_pair.h"
#include "sgx_internal.h"

namespace sgx {
namespace {

using namespace inreg;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

constexpr int kEFS = 16384 + 16;           // LDS bytes per frame of ex[f][32][32] of 16-byte elements
constexpr int kEEx = 8 * kEFS;             // 131200: exchange buffer; also holds the staged samples (<= 73728 B)
constexpr int kEWinOff = 0;                // tables behind it: v2d win[1024] = (w[2n], w[2n+1]) / 2
constexpr int kETw2Off = 16384;            // v2d tw2[64][8]: entry u of lane kind kb = W' = -i W_2048^(kb + 64 u)
constexpr int kELds = kEEx + kETw2Off + 64 * 8 * 16;  // 155776
// filterbank outputs (hop <= 585: 6 staging rounds): the |X|^2 tile (1036 bins x 8 frames of f64: bin k, frame f at k * 8 + f) sits in the upper
// half of the exchange buffer, above the staged samples; the band schedule (plan.hip build_band_schedule: 16 half-waves x 8 slots, 8-byte
// weights) stays in GLOBAL memory — its 20 KB do not fit beside the tables, and every half-wave reads its own rows through L1
constexpr int kEPwOff = kEEx - 1036 * 64;  // 64896 >= 6 * 8192
constexpr int kESegs = 2;
__host__ __device__ constexpr unsigned pwd8_index(unsigned k, unsigned f) { return k * 8u + f; }

template <int AMP>
__device__ __forceinline__ double amp_e(double p, double eps) {
    if constexpr (AMP == AMP_MAGNITUDE) return sqrt(p);
    else if constexpr (AMP == AMP_DB) return db_f64(fmax(p, eps));
    else return p;
}

using lanepair::trade32;  // lanes l and l ^ 32 trade a complex value, each receives the other's as (im, re): lane_pair.h

__device__ __forceinline__ v2d mul_add_unfused_e(double w, v2d p, v2d acc) {  // the reference's `acc += w * x`: two roundings (spectrogram.rs:102-117)
    return (v2d){__dadd_rn(__dmul_rn(w, p.x), acc.x), __dadd_rn(__dmul_rn(w, p.y), acc.y)};
}

// band stage over the tile's 8 frames: 16 half-waves x 8 slots x 4 frame pairs; a lane sums one band for two frames in ascending-bin order
template <int AMP>
__device__ __forceinline__ void mel_tile_sched_e(const StftArgs &a, const double *pw, const unsigned *sched, unsigned b, unsigned f0, unsigned nf,
                                                 double eps, unsigned tid) {
    const unsigned vw = tid >> 5, slot = (tid >> 2) & 7u, fp = tid & 3u;
    constexpr unsigned kDrop = 0x80000000u;  // past the descriptor's range: the hardware drops the store
    const unsigned obytes = (a.n_out * a.n_frames - f0) * 8u;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc((const double *)a.out + (size_t)b * a.n_out * a.n_frames + f0, obytes);
    const unsigned fo0 = 2u * fp < nf ? 16u * fp : kDrop, fo1 = 2u * fp + 1u < nf ? 16u * fp + 8u : kDrop;
    const uint4 *info = (const uint4 *)(sched + 4) + vw * 8u + slot;
#pragma unroll
    for (unsigned seg = 0; seg < (unsigned)kESegs; ++seg) {
        const uint4 cur = info[seg * 128u];
        const unsigned L = cur.x;  // (per half-wave: the two halves of a wave run to the longer one)
        const bool have = cur.w != 0xffffffffu;
        if (!have) continue;
        const v2d *wr = (const v2d *)(sched + cur.y);
        const v2d *pr = (const v2d *)(pw + cur.z * 8u + fp * 2u);
        v2d acc = {0.0, 0.0};
        for (unsigned t = 0; t < L; t += 4u) {  // bins t .. t + 3, frames 2 fp and 2 fp + 1 of each
            const v2d w01 = wr[t >> 1], w23 = wr[(t >> 1) + 1u];
            const v2d q0 = pr[t * 4u], q1 = pr[t * 4u + 4u], q2 = pr[t * 4u + 8u], q3 = pr[t * 4u + 12u];
            acc = mul_add_unfused_e(w01.x, q0, acc);
            acc = mul_add_unfused_e(w01.y, q1, acc);
            acc = mul_add_unfused_e(w23.x, q2, acc);
            acc = mul_add_unfused_e(w23.y, q3, acc);
        }
        const unsigned bo = cur.w * a.n_frames * 8u;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, amp_e<AMP>(acc.x, eps)), ro, (int)(fo0 != kDrop ? bo + fo0 : kDrop), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, amp_e<AMP>(acc.y, eps)), ro, (int)(fo1 != kDrop ? bo + fo1 : kDrop), 0, 0);
    }
}

template <int MODE, int AMP, int ROUNDS, bool ODD = false>  // ODD: odd hops (round 5) — a variant of its own: as a run-time branch it spilled the even path's registers (512 / 64 x 10 s: 79 -> 225 us)
__global__ __launch_bounds__(512, 2) void k_d32x32(StftArgs a, unsigned per_xcd, unsigned total, unsigned slots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x;
    unsigned char *tabs = smem + kEEx;
    ((v4f *)(tabs + kEWinOff))[tid] = ((const v4f *)a.window)[tid];  // 2048 doubles = 1024 x 16 B
    ((v4f *)(tabs + kEWinOff))[tid + 512u] = ((const v4f *)a.window)[tid + 512u];
    ((v4f *)(tabs + kETw2Off))[tid] = ((const v4f *)a.tw2)[tid];     // 8192 B

    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned lo = xcd * per_xcd, hi = min(lo + per_xcd, total);
    unsigned wid = lo + slot;

    // pass-1 identity: par = DIF half of the column transform (wave-uniform), frame, column
    const unsigned par = __builtin_amdgcn_readfirstlane(tid >> 8), p1f = (tid >> 5) & 7u, n2 = tid & 31u;
    const unsigned wave = tid >> 6, lane = tid & 63u, half = lane >> 5, p2f = lane & 7u;
    const unsigned r = wave + 8u * ((lane >> 3) & 3u);  // pass-2 job: row r (half 0: its even outputs) with row 32 - r (half 1: its odd outputs)
    const unsigned row = half ? ((32u - r) & 31u) : r;
    const bool j0 = r == 0u;
    const unsigned kb = half ? (j0 ? 32u : 64u - r) : r;  // own H[u] = Z[kb + 64 u]
    const double eps = a.eps;
    constexpr unsigned ES = MODE == OUT_COMPLEX ? 16u : 8u;
    const unsigned step = 64u * a.n_frames * ES;  // uniform: 64 bins further
    double *pwd = (double *)(smem + kEPwOff);
    const v2d *twj = (const v2d *)(tabs + kETw2Off) + kb * 8u;
    // pass-1 twiddles W_1024^(k1 n2), k1 = 2 m + par: twm[m >> 2] * twl[m & 3] with twm[q] = W^(8 q n2), twl[j] = W^((2 j + par) n2)
    v2d twm[4], twl[4];
    {
        const v2d *t1 = (const v2d *)a.tw1 + n2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            twm[q] = t1[32 * 8 * q];
            twl[q] = t1[32 * (2 * q + par)];
        }
    }
    const double sg = half ? -1.0 : 1.0, hb = half ? 1.0 : 0.0;
    const double sg1 = par ? -1.0 : 1.0;

    constexpr int NCR = ROUNDS > 0 ? ROUNDS : 1;
    v4f creg[NCR];
    const unsigned hop = a.hop;
    const unsigned row_bytes = (unsigned)a.n_samples * 8u;  // host: n_samples < 2^28
    auto load_tile = [&](unsigned w) {
        if constexpr (ROUNDS > 0) {
            const unsigned b = w / a.tiles, f0 = (w - b * a.tiles) * 8u;
            const __amdgpu_buffer_rsrc_t rx = make_rsrc((const double *)a.x + (size_t)b * a.sample_stride, row_bytes);
            const int tile_lo = (int)(f0 * hop) - (int)a.pad;  // (negative in the left padding: out of range as an unsigned offset, reads 0 — S1)
            const int vo = (tile_lo + 2 * (int)tid) * 8;
#pragma unroll
            for (int q = 0; q < ROUNDS; ++q) creg[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rx, vo + q * 8192, 0, 0));
        }
    };
    if (wid < hi) load_tile(wid);
    __syncthreads();  // tables visible

    const unsigned char *xs = smem + p1f * hop * 8u + n2 * 16u;
    const v2d *w2 = (const v2d *)(tabs + kEWinOff) + n2;

    while (wid < hi) {
        const unsigned b = wid / a.tiles, f0 = (wid - b * a.tiles) * 8u;
        const unsigned nf = min(8u, a.n_frames - f0);
        v2d xr[16];
        {
            if constexpr (ROUNDS > 0) {
#pragma unroll
                for (int q = 0; q < ROUNDS; ++q) *(v4f *)(smem + (q * 512u + tid) * 16u) = creg[q];
                __syncthreads();  // barrier 1: the staged samples are complete
            }
            // this lane's half of the column: d[n1] = w z[n1] +- w' z[n1 + 16], the odd half times W_32^n1 (par is wave-uniform)
            const __amdgpu_buffer_rsrc_t rx = make_rsrc((const double *)a.x + (size_t)b * a.sample_stride, row_bytes);  // (per-lane columns only)
            const int vo = ((int)(p1f * hop) + (int)(f0 * hop) - (int)a.pad + 2 * (int)n2) * 8;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                v2d z0, z1;
                if constexpr (ROUNDS > 0) {
                    if constexpr (ODD) {  // (round 5: odd hops): an odd frame's pairs sit at 8-byte-aligned addresses — two 8-byte reads
                        z0 = (v2d){*(const double *)(xs + n1 * 512), *(const double *)(xs + n1 * 512 + 8)};
                        z1 = (v2d){*(const double *)(xs + n1 * 512 + 8192), *(const double *)(xs + n1 * 512 + 8200)};
                    } else {
                    z0 = *(const v2d *)(xs + n1 * 512);
                    z1 = *(const v2d *)(xs + n1 * 512 + 8192);
                    }
                } else {  // per-lane columns (hop > 1024): the tile is not staged and not prefetched
                    int o0 = vo + n1 * 512, o1 = vo + n1 * 512 + 8192;
                    asm("" : "+v"(o0), "+v"(o1));  // the whole offset in the lane register (buffer_ops.h)
                    z0 = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rx, o0, 0, 0));
                    z1 = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rx, o1, 0, 0));
                    if constexpr (ODD) {  // the pair (x[-1], x[0]) of an odd frame starts outside the row: the whole 16-byte access reads 0 — put x[0] back
                        const int s0 = (int)(p1f * hop) + (int)(f0 * hop) - (int)a.pad + 2 * (int)n2 + 64 * n1;
                        if (s0 == -1 || s0 + 1024 == -1) {
                            const double x0 = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rx, 0, 0, 0));
                            if (s0 == -1) z0.y = x0; else z1.y = x0;
                        }
                    }
                }
                const v2d wa = w2[32 * n1], wb = w2[32 * n1 + 512];
                v2d d = pfma(z1 * wb, (v2d){sg1, sg1}, z0 * wa);
                if (par && n1 > 0) {  // (uniform branch)
                    const double c = kCos64[2 * n1], s = -kSin64[2 * n1];  // W_32^n1
                    d = cmulv(d, (v2d){c, s});
                }
                xr[n1] = d;
                if ((n1 & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            Fft<16, false, v2d>::run(xr, xr);
        }
        // barrier 2: every wave has read its columns (and, filterbank outputs, finished the previous tile's band stage): pass 1 may write ex
        if constexpr (ROUNDS > 0 || MODE == OUT_MEL) __syncthreads();
        {
            unsigned char *dst = smem + p1f * kEFS + par * 512u + n2 * 16u;  // rows k1 = 2 m + par
#pragma unroll
            for (int m = 0; m < 16; ++m) {  // twiddle by W_1024^(k1 n2), write row k1 of this lane's column
                const int qa = m >> 2, qb = m & 3;
                v2d v = xr[m];
                if (qb || par) v = cmulv(v, twl[qb]);  // (twl[0] = W^(par n2): 1 for par 0)
                if (qa) v = cmulv(v, twm[qa]);
                *(v2d *)(dst + m * 1024) = v;
            }
        }
        const unsigned next = wid + slots;
        if (next < hi) load_tile(next);  // in flight during pass 2
        __syncthreads();  // barrier 3: ex complete
        // pass 2.  A lane whose frame does not exist (last tile of a signal) mirrors the tile's last frame: same values to the same addresses.
        const unsigned fe = min(p2f, nf - 1u);
        v2d H[16];
        {
            const unsigned char *rp = smem + fe * kEFS + row * 512u;
            double hbl = hb;
            asm volatile("" : "+v"(hbl));  // (not loop-invariant: the 15 lane twiddles below would otherwise be kept in 60 registers across tiles)
#pragma unroll
            for (int n = 0; n < 16; ++n) {  // one decimation-in-frequency step: even outputs x[n] + x[n + 16]; odd: (x[n] - x[n + 16]) W_32^n
                const v2d x0 = *(const v2d *)(rp + n * 16), x1 = *(const v2d *)(rp + n * 16 + 256);
                v2d d = pfma(x1, (v2d){sg, sg}, x0);
                if (n > 0) {
                    const double c = kCos64[2 * n], s = -kSin64[2 * n];
                    const v2d t = {__builtin_fma(hbl, c - 1.0, 1.0), hbl * s};  // half 0: 1; half 1: W_32^n
                    d = cmulv(d, t);
                }
                H[n] = d;
                if ((n & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();  // barrier 4: ex consumed: the next staging may overwrite it
        Fft<16, false, v2d>::run(H, H);
        const v2d h8 = H[8];
        v2d R[8];  // R[j] = the partner's H[8 + j] as (im, re)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            R[j] = H[8 + j];
            trade32(R[j]);
        }
        if (j0) {
            // row 0: both halves pair inside themselves.  Half 0 (E[m] = Z[64 m]): E[u] with E[16 - u] (u = 0: Z[0] with itself gives bins 0 and
            // 1024; E[8] = Z[512] pairs with itself, below).  Half 1 (O[m] = Z[32 + 64 m]): O[u] with O[15 - u].
#pragma unroll
            for (int j = 0; j < 7; ++j) R[j] = swp(half ? H[8 + j] : H[9 + j]);
            R[7] = swp(half ? H[15] : H[0]);
            asm volatile("" ::: "memory");  // keeps this a branch
        }
        const unsigned p2ofs = f0 + fe;
        const __amdgpu_buffer_rsrc_t ro = make_rsrc((unsigned char *)a.out + (size_t)b * 1025u * a.n_frames * ES, 1025u * a.n_frames * ES);
        // bins kb + 64 u upwards; the mirrored bins 1024 - kb - 64 u count down: lane part 7 steps low, scalar part (7 - u) steps
        const unsigned oa = (kb * a.n_frames + p2ofs) * ES, ob = ((1024u - 448u - kb) * a.n_frames + p2ofs) * ES;
        if constexpr (MODE == OUT_MEL) {  // bins 1025..1035 are read with zero weights
            if (tid < 88u) pwd[pwd8_index(1025u + (tid >> 3), tid & 7u)] = 0.0;
        }
        double *pw_a = pwd + pwd8_index(kb, p2f), *pw_b = pwd + pwd8_index(1024u - 448u - kb, p2f);
        constexpr int PSTEP = 64 * 8;  // doubles between bins k and k + 64 in the |X|^2 tile
        auto emit = [&](unsigned voff, unsigned soff, double *pwp, v2d X, bool conj) {
            if constexpr (MODE == OUT_MEL) {
                const double p = __builtin_fma(X.x, X.x, X.y * X.y);
                *pwp = AMP == AMP_MAG_IN ? sqrt(p) : p;  // (a lane without a frame writes its mirror's values into its own slot: never stored)
            } else if constexpr (MODE == OUT_COMPLEX) {
                const v2d V = conj ? (v2d){X.x, -X.y} : X;
                // (16-byte stores: the whole offset in the lane register, soffset = 0 — kernels_d32x16.hip: emit)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, V), ro, (int)(voff + soff), 0, 0);
            } else {
                const double p = __builtin_fma(X.x, X.x, X.y * X.y);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, amp_e<AMP>(p, eps)), ro, (int)voff, (int)soff, 0);
            }
        };
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            // pair (P, Q) = (Z[k], Z[1024 - k]): E = (P.x + Q.x, P.y - Q.y), D = (P.x - Q.x, P.y + Q.y), T = W' D with W' = -i W_2048^k:
            //   X[k] = E + T, X[1024 - k] = conj(E - T)   (window pre-halved: no 1/2)
            const v2d P = H[u], Q = swp(R[7 - u]);
            const v2d E = pfma(Q, (v2d){1.0, -1.0}, P), D = pfma(Q, (v2d){-1.0, 1.0}, P);
            const v2d T = cmulv(D, twj[u]);
            emit(oa, u * step, pw_a + u * PSTEP, E + T, false);
            emit(ob, (7 - u) * step, pw_b + (7 - u) * PSTEP, E - T, true);
        }
        if (j0 && half == 0u) emit((512u * a.n_frames + p2ofs) * ES, 0u, pwd + pwd8_index(512u, p2f), h8 * (v2d){2.0, -2.0}, false);  // X[512] = 2 conj(Z[512])
        if constexpr (MODE == OUT_MEL) {
            __syncthreads();  // |X|^2 tile complete
            mel_tile_sched_e<AMP>(a, pwd, a.mel_sched, b, f0, nf, eps, tid);
        }
        wid = next;
    }
}

template <int MODE, int AMP>
hipError_t launch_variant_e(const StftArgs &a, hipStream_t s) {
    const unsigned total = a.tiles * a.batch;
    const unsigned per_xcd = (total + 7u) / 8u;
    const unsigned cu_slots = std::max(1u, device_cu_count() / 8u);
    const unsigned nslots = per_xcd < cu_slots ? per_xcd : cu_slots;  // one 512-thread workgroup per CU
    const unsigned bytes = (7u * a.hop + 2048u) * 8u;                 // a tile's samples
    auto go = [&](auto kernel) -> hipError_t {
        hipError_t e = set_max_dynamic_lds((const void *)kernel, kELds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3(nslots * 8), dim3(512), kELds, s, a, per_xcd, total, nslots);
        return hipGetLastError();
    };
    if (a.hop & 1u) {
        if (bytes <= 6u * 8192u) return go(k_d32x32<MODE, AMP, 6, true>);
        if constexpr (MODE == OUT_MEL) {
            return hipErrorInvalidConfiguration;
        } else {
            if (bytes <= 9u * 8192u) return go(k_d32x32<MODE, AMP, 9, true>);
            return go(k_d32x32<MODE, AMP, 0, true>);
        }
    }
    if (bytes <= 6u * 8192u) return go(k_d32x32<MODE, AMP, 6>);
    if constexpr (MODE == OUT_MEL) {
        return hipErrorInvalidConfiguration;  // (plan_geometry_d32x32_f64 keeps such hops away)
    } else {
        if (bytes <= 9u * 8192u) return go(k_d32x32<MODE, AMP, 9>);
        return go(k_d32x32<MODE, AMP, 0>);
    }
}

}  // namespace

bool plan_geometry_d32x32_f64(StftArgs &a) {
    if (a.n_fft != 2048) return false;
    // (odd hops since round 5 — while the tile is staged: the per-lane column path with the row-start patch spills and measured 98 us per 64 x 10 s at
    // hop 1025 against ~74 on the register-tiled kernel)
    if ((a.hop & 1u) && (7u * a.hop + 2048u) * 8u > 9u * 8192u) return false;
    // filterbank outputs: fused up to hop 585 (6 staging rounds below the |X|^2 tile) where the bank has a band schedule; else per-bin power + k_bank_rows
    if (a.out_mode == OUT_MEL && (a.mel_sched_words == 0 || (7u * a.hop + 2048u) * 8u > 6u * 8192u)) return false;
    if (a.x != nullptr && a.n_frames < 4u) return false;                                      // batches of very short signals: mostly empty tiles
    if (a.n_samples >= (1ull << 28)) return false;                                        // 32-bit byte offsets into a sample row
    if ((unsigned long long)a.n_frames * 1025ull * 16ull >= 0x7fffffffull) return false;  // and into one output signal
    a.ft = 8;
    return true;
}

hipError_t launch_d32x32_f64(const StftArgs &a, hipStream_t s) {
    const unsigned long long total64 = (unsigned long long)a.tiles * a.batch;
    if (total64 == 0 || total64 >= 0x7ffffff0ull) return hipErrorInvalidConfiguration;
    if (a.out_mode == OUT_COMPLEX) return launch_variant_e<OUT_COMPLEX, AMP_POWER>(a, s);
    if (a.out_mode == OUT_MEL) {
        if (a.amp == AMP_MAGNITUDE) return launch_variant_e<OUT_MEL, AMP_MAGNITUDE>(a, s);
        if (a.amp == AMP_DB) return launch_variant_e<OUT_MEL, AMP_DB>(a, s);
        if (a.amp == AMP_MAG_IN) return launch_variant_e<OUT_MEL, AMP_MAG_IN>(a, s);
        return launch_variant_e<OUT_MEL, AMP_POWER>(a, s);
    }
    if (a.amp == AMP_MAGNITUDE) return launch_variant_e<OUT_LINEAR, AMP_MAGNITUDE>(a, s);
    if (a.amp == AMP_DB) return launch_variant_e<OUT_LINEAR, AMP_DB>(a, s);
    return launch_variant_e<OUT_LINEAR, AMP_POWER>(a, s);
}

}  // namespace sgx
