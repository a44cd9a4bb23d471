// kernels_d32x16.hip — tuned f64, n_fft = 1024 STFT kernel for gfx950 (round 4): BASELINE's shape in the reference's other `Sample`
// type (src/sample.rs:23-86; f64 is the default dtype of the Python API and configs[0]'s type).
//
// k_r32x32's construction (kernels_r32x32.hip) at 512 complex f64 points — the same bytes per frame, so the same tile: 16 consecutive
// frames of one signal, one persistent 512-thread workgroup per CU, a 128 KiB exchange buffer ex[f][k1][n2] of 16-byte elements.
//
//   pass 1  lane (f = 0..15, n2 = 0..31) owns z[32 n1 + n2], n1 = 0..15, of frame f, z[n] = w[2n] x[2n] + i w[2n+1] x[2n+1] (window
//           pre-scaled by 1/2 on the host — exact); window multiply fused into the first butterflies; one 16-point FFT in registers;
//           twiddle W_512^(k1 n2) from two short per-lane register tables; one ds_write_b128 per value.
//   pass 2  16 rows of 32 points.  The real split pairs Z[k] with Z[512 - k] = row 16 - k1, element 31 - k2, and the even-indexed outputs
//           of row r pair with the odd-indexed outputs of row 16 - r.  A half row (one decimation-in-frequency step as the row is read,
//           then a 16-point FFT: 64 data registers in f64) is one lane's work, so the two partner halves sit in lanes l and l + 32 of
//           one wave: lane l (half 0) takes the even outputs of row r, lane l + 32 the odd outputs of row 16 - r, and after the
//           transforms they trade the upper 8 values with v_permlane32_swap_b32 (32 instructions; no LDS).  Each then splits 8 pairs
//           = 16 bins: own H[u] = Z[kb + 32 u] with the partner's Z[512 - kb - 32 u], kb = r (half 0) or 32 - r (half 1).  Row 0's two
//           halves pair inside themselves (kb = 0 and 16): rearranged once under a branch their 32 lanes take.
//           16 rows x 2 halves x 16 frames = the 512 lanes.
//   store   the 16 lanes of a (row, half) hold one bin of 16 consecutive frames: 128-byte runs of the frame-contiguous layout (S9).
//
// Samples: the tile's 15 hop + 1024 samples once, 16-byte buffer loads through the row's descriptor (out-of-range dwords read 0: the
// zero centre padding, spectrogram.rs:1301-1320), one tile ahead, staged in LDS over the idle exchange buffer (hop <= 272); longer hops
// load their columns per lane.  Filterbank outputs: |X|^2 to LDS (pwd_index), reduced per (band, frame pair) in ascending-bin order
// (spectrogram.rs:102-117) along a host-built schedule over the 8 waves.
// Reference semantics: spectrogram.rs:1301-1334, :2068-2080.
#include <type_traits>
#include <utility>

#include "buffer_ops.h"
#include "fft_inreg.h"
#include "d32x16_layout.h"
#include "r32x16_layout.h"
#include "db_f64.h"
#include "lane_pair.h"
#include "sgx_internal.h"

namespace sgx {
namespace {

using namespace inreg;
using namespace d32x16;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

// a bin's 16 frames are 128 bytes = half of the LDS banks: a 16-lane read group is two slots, whose first bins the schedule makes even and
// odd (build_band_schedule), so that the two read different halves at every step
__host__ __device__ constexpr unsigned pwd_index(unsigned k, unsigned f) { return k * 16u + f; }

template <int AMP>
__device__ __forceinline__ double amp_f64(double p, double eps) {
    if constexpr (AMP == AMP_MAGNITUDE) return sqrt(p);
    else if constexpr (AMP == AMP_DB) return db_f64(fmax(p, eps));
    else return p;
}

using lanepair::trade32;  // lanes l and l ^ 32 trade a complex value, each receives the other's as (im, re): lane_pair.h

__device__ __forceinline__ v2d mul_add_unfused_d(double w, v2d p, v2d acc) {  // the reference's `acc += w * x`: two roundings (spectrogram.rs:102-117)
    return (v2d){__dadd_rn(__dmul_rn(w, p.x), acc.x), __dadd_rn(__dmul_rn(w, p.y), acc.y)};
}

// band stage over the tile's 16 frames: 8 waves x 8 slots x 8 frame pairs along the host-built schedule (r32x16_layout.h's format with 8-byte
// weights): a lane sums one band for two frames in ascending-bin order
template <int AMP>
__device__ __forceinline__ void mel_tile_sched_d(const StftArgs &a, const double *pw, const unsigned *sched, unsigned b, unsigned f0, unsigned nf,
                                                 double eps, unsigned tid) {
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u, slot = lane >> 3, fp = lane & 7u;
    constexpr unsigned kDrop = 0x80000000u;  // past the descriptor's range: the hardware drops the store
    const unsigned obytes = (a.n_out * a.n_frames - f0) * 8u;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc((const double *)a.out + (size_t)b * a.n_out * a.n_frames + f0, obytes);
    const unsigned fo0 = 2u * fp < nf ? 16u * fp : kDrop, fo1 = 2u * fp + 1u < nf ? 16u * fp + 8u : kDrop;
    const uint4 *info = (const uint4 *)(sched + 4) + wave * 8u + slot;
    uint4 cur = info[0];
#pragma unroll
    for (unsigned seg = 0; seg < (unsigned)kDSegs; ++seg) {
        const uint4 nxt = seg + 1u < (unsigned)kDSegs ? info[(seg + 1u) * 64u] : cur;
        const unsigned L = __builtin_amdgcn_readfirstlane(cur.x);
        const v2d *wr = (const v2d *)(sched + cur.y);
        const v2d *pr = (const v2d *)(pw + cur.z * 16u + fp * 2u);
        v2d acc = {0.0, 0.0};
        // bins t .. t + 3, frames 2 fp and 2 fp + 1 of each.  (Reading a step ahead of the sums measured the same: 64 x 10 s Mel-80 power 73-75 us
        // both ways.)
        for (unsigned t = 0; t < L; t += 4u) {
            const v2d w01 = wr[t >> 1], w23 = wr[(t >> 1) + 1u];
            const v2d q0 = pr[t * 8u], q1 = pr[t * 8u + 8u], q2 = pr[t * 8u + 16u], q3 = pr[t * 8u + 24u];
            acc = mul_add_unfused_d(w01.x, q0, acc);
            acc = mul_add_unfused_d(w01.y, q1, acc);
            acc = mul_add_unfused_d(w23.x, q2, acc);
            acc = mul_add_unfused_d(w23.y, q3, acc);
        }
        const bool have = cur.w != 0xffffffffu;
        // (a segment the wave has no band in is skipped whole: an f64 log10 is some hundred instructions — 64 x 10 s, Mel-80 dB: 87 -> 81 us)
        if (__builtin_amdgcn_ballot_w64(have) != 0ull) {
            const unsigned bo = cur.w * a.n_frames * 8u;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, amp_f64<AMP>(acc.x, eps)), ro, (int)((have && fo0 != kDrop) ? bo + fo0 : kDrop), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, amp_f64<AMP>(acc.y, eps)), ro, (int)((have && fo1 != kDrop) ? bo + fo1 : kDrop), 0, 0);
        }
        cur = nxt;
    }
}

template <int MODE, int AMP, int ROUNDS, bool ODD = false>  // ODD: odd hops (round 5) — a variant of its own: as a run-time branch it cost the even path spilled registers
__global__ __launch_bounds__(512, 2) void k_d32x16(StftArgs a, unsigned per_xcd, unsigned total, unsigned slots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x;
    unsigned char *tabs = smem + kDEx;
    ((v4f *)(tabs + kDWinOff))[tid] = ((const v4f *)a.window)[tid];  // 1024 doubles
    if (tid < 256u) ((v4f *)(tabs + kDTw2Off))[tid] = ((const v4f *)a.tw2)[tid];
    unsigned *sched = (unsigned *)(tabs + kDSchOff);
    if constexpr (MODE == OUT_MEL)
        for (unsigned i = tid; i < a.mel_sched_words; i += 512u) sched[i] = a.mel_sched[i];

    // XCD x owns the contiguous run of tiles [x per_xcd, (x + 1) per_xcd); its `slots` resident workgroups walk it with that stride
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned lo = xcd * per_xcd, hi = min(lo + per_xcd, total);
    unsigned wid = lo + slot;

    const unsigned p1f = tid >> 5, n2 = tid & 31u;  // pass-1 identity
    const unsigned wave = tid >> 6, lane = tid & 63u, half = lane >> 5, p2f = lane & 15u;
    const unsigned r = wave + 8u * ((lane >> 4) & 1u);  // pass-2 job: row r (half 0: its even outputs) with row 16 - r (half 1: its odd outputs)
    const unsigned row = half ? ((16u - r) & 15u) : r;
    const bool j0 = r == 0u;
    const unsigned kb = half ? (j0 ? 16u : 32u - r) : r;  // own H[u] = Z[kb + 32 u]
    const double eps = a.eps;
    constexpr unsigned ES = MODE == OUT_COMPLEX ? 16u : 8u;
    double *pwd = (double *)(smem + kDPwOff);
    const unsigned step = 32u * a.n_frames * ES;  // uniform: 32 bins further
    const v2d *twj = (const v2d *)(tabs + kDTw2Off) + kb * 8u;
    v2d twa[4], twb[4];  // W_512^(k1 n2) = twa[k1 >> 2] * twb[k1 & 3]
    {
        const v2d *t1 = (const v2d *)a.tw1 + n2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            twa[q] = t1[32 * 4 * q];
            twb[q] = t1[32 * q];
        }
    }
    const double sg = half ? -1.0 : 1.0, hb = half ? 1.0 : 0.0;

    constexpr int NCR = ROUNDS > 0 ? ROUNDS : 1;
    v4f creg[NCR];
    v2d xd[ROUNDS > 0 ? 1 : 16];
    const unsigned hop = a.hop;
    constexpr bool oddhop = ODD;  // (round 5: odd hops run here too — VERDICT r4 item 7)
    const unsigned row_bytes = (unsigned)a.n_samples * 8u;  // host: n_samples < 2^29
    auto load_tile = [&](unsigned w) {
        const unsigned b = w / a.tiles, f0 = (w - b * a.tiles) * 16u;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc((const double *)a.x + (size_t)b * a.sample_stride, row_bytes);
        // first sample of the tile relative to the row: negative in the left padding — as an unsigned byte offset far out of range, so
        // the hardware returns 0 there as it does past the end of the row (S1)
        const int tile_lo = (int)(f0 * hop) - (int)a.pad;
        if constexpr (ROUNDS > 0) {
            const int vo = (tile_lo + 2 * (int)tid) * 8;
#pragma unroll
            for (int q = 0; q < ROUNDS; ++q) creg[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rx, vo + q * 8192, 0, 0));
        } else {
            const int vo = ((int)(p1f * hop) + tile_lo + 2 * (int)n2) * 8;  // (even hop: a pair never straddles the row start)
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                int o = vo + n1 * 512;
                asm("" : "+v"(o));  // the whole offset in the lane register: an immediate part is added without wrapping (buffer_ops.h)
                xd[n1] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rx, o, 0, 0));
            }
            if (oddhop) {  // uniform.  Odd frames sit on odd sample offsets: the pair (x[-1], x[0]) starts outside the row, and a 16-byte access whose first
                           // dwords are out of range returns 0 for all of it: put x[0] back (as k_r32x16 does for its 8-byte pairs)
                const v2i q0 = __builtin_amdgcn_raw_buffer_load_b64(rx, 0, 0, 0);
                const double x0 = __builtin_bit_cast(double, q0);
                const int s0 = (int)(p1f * hop) + tile_lo + 2 * (int)n2;
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1)
                    if (s0 + 64 * n1 == -1) xd[n1].y = x0;
            }
        }
    };
    if (wid < hi) load_tile(wid);
    __syncthreads();  // tables visible

    const unsigned char *xs = smem + p1f * hop * 8u + n2 * 16u;
    const v2d *w2 = (const v2d *)(tabs + kDWinOff) + n2;

    while (wid < hi) {
        const unsigned b = wid / a.tiles, f0 = (wid - b * a.tiles) * 16u;
        const unsigned nf = min(16u, a.n_frames - f0);
        v2d xr[16];
        {
            // even and odd n1 apart (the 16-point transform's first split), so that at most half of the 32 operands are live at once
            v2d e[8], we[8], o[8], wo[8];
            if constexpr (ROUNDS > 0) {
#pragma unroll
                for (int q = 0; q < ROUNDS; ++q) *(v4f *)(smem + (q * 512u + tid) * 16u) = creg[q];
                __syncthreads();  // barrier 1: the staged samples are complete
                if (oddhop) {  // uniform.  Odd frames start on odd samples: their pairs sit at 8-byte-aligned LDS addresses — two 8-byte reads
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        e[k] = (v2d){*(const double *)(xs + k * 1024), *(const double *)(xs + k * 1024 + 8)};
                        we[k] = w2[64 * k];
                    }
                } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    e[k] = *(const v2d *)(xs + k * 1024);
                    we[k] = w2[64 * k];
                }
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    e[k] = xd[2 * k];
                    we[k] = w2[64 * k];
                }
            }
            Fft<8, true, v2d>::run(e, we);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if constexpr (ROUNDS > 0) {
                    if (oddhop) o[k] = (v2d){*(const double *)(xs + k * 1024 + 512), *(const double *)(xs + k * 1024 + 520)};
                    else o[k] = *(const v2d *)(xs + k * 1024 + 512);
                } else o[k] = xd[2 * k + 1];
                wo[k] = w2[64 * k + 32];
            }
            Fft<8, true, v2d>::run(o, wo);
            Comb<16, 0, v2d>::run(xr, e, o);
        }
        // barrier 2: every wave has read its columns (and, filterbank outputs, finished the previous tile's band stage, whose |X|^2 tile the
        // upper half of ex overlays): pass 1 may write ex
        if constexpr (ROUNDS > 0 || MODE == OUT_MEL) __syncthreads();
        {
            unsigned char *dst = smem + p1f * kDFS + n2 * 16u;
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) {  // twiddle by W_512^(k1 n2), write row k1 of this lane's column
                const int qa = k1 >> 2, qb = k1 & 3;
                v2d v = xr[k1];
                if (qb) v = cmulv(v, twb[qb]);
                if (qa) v = cmulv(v, twa[qa]);
                *(v2d *)(dst + k1 * 512) = v;
            }
        }
        const unsigned next = wid + slots;
        if (next < hi) load_tile(next);  // in flight during pass 2
        __syncthreads();  // barrier 3: ex complete
        // pass 2.  A lane whose frame does not exist (last tile of a signal) mirrors the tile's last frame: same values to the same
        // addresses, so every lane stores unconditionally and the compiler counts the stores behind the next tile's loads.
        const unsigned fe = min(p2f, nf - 1u);
        v2d H[16];
        {
            const unsigned char *rp = smem + fe * kDFS + row * 512u;
            double hbl = hb;
            asm volatile("" : "+v"(hbl));  // (not loop-invariant: the 15 lane twiddles below would otherwise be kept in 60 registers across tiles)
            // one decimation-in-frequency step: even outputs x[n] + x[n + 16]; odd: (x[n] - x[n + 16]) W_32^n.  Four points at a time: the
            // scheduler would otherwise put all 32 reads (128 registers) in flight
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const v2d x0 = *(const v2d *)(rp + n * 16), x1 = *(const v2d *)(rp + n * 16 + 256);
                v2d d = pfma(x1, (v2d){sg, sg}, x0);
                if (n > 0) {
                    const double c = kCos64[2 * n], s = -kSin64[2 * n];     // W_32^n
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
            // row 0: both halves pair inside themselves.  Half 0 (E[m] = Z[32 m]): E[u] with E[16 - u] (u = 0: Z[0] with itself gives bins 0 and
            // 512; E[8] = Z[256] pairs with itself, below).  Half 1 (O[m] = Z[16 + 32 m]): O[u] with O[15 - u].
#pragma unroll
            for (int j = 0; j < 7; ++j) R[j] = swp(half ? H[8 + j] : H[9 + j]);
            R[7] = swp(half ? H[15] : H[0]);
            asm volatile("" ::: "memory");  // keeps this a branch
        }
        const unsigned p2ofs = f0 + fe;
        const __amdgpu_buffer_rsrc_t ro = make_rsrc((unsigned char *)a.out + (size_t)b * 513u * a.n_frames * ES, 513u * a.n_frames * ES);
        // bins kb + 32 u upwards; the mirrored bins 512 - kb - 32 u count down: lane part 7 steps low, scalar part (7 - u) steps
        const unsigned oa = (kb * a.n_frames + p2ofs) * ES, ob = ((512u - 224u - kb) * a.n_frames + p2ofs) * ES;
        if constexpr (MODE == OUT_MEL) {  // bins 513..523 are read with zero weights
            if (tid < 176u) pwd[pwd_index(513u + (tid >> 4), tid & 15u)] = 0.0;
        }
        double *pw_a = pwd + pwd_index(kb, p2f), *pw_b = pwd + pwd_index(512u - 224u - kb, p2f);
        constexpr int PSTEP = 16 * 32;  // doubles between bins k and k + 32 in the |X|^2 tile
        auto emit = [&](unsigned voff, unsigned soff, double *pwp, v2d X, bool conj) {
            if constexpr (MODE == OUT_MEL) {
                const double p = __builtin_fma(X.x, X.x, X.y * X.y);
                *pwp = AMP == AMP_MAG_IN ? sqrt(p) : p;  // (a lane without a frame writes its mirror's values into its own slot: never stored)
            } else if constexpr (MODE == OUT_COMPLEX) {
                const v2d V = conj ? (v2d){X.x, -X.y} : X;
                // The whole offset in the lane register, soffset = 0.  With a scalar-register soffset the compiler assumes that a 16-byte store's
                // data registers may be rewritten by the very next instruction (GCNHazardRecognizer: the ">8-byte store data" hazard "only exists
                // without a register soffset") and did so — and on this device lanes 12..15 of every 16-lane row then stored the NEW contents now
                // and then: 0.06 % of the complex STFT wrong at hop >= 274 (64 x 10 s; staged hops happened to schedule apart).  With an
                // immediate soffset it keeps the wait states itself.
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, V), ro, (int)(voff + soff), 0, 0);
            } else {
                const double p = __builtin_fma(X.x, X.x, X.y * X.y);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, amp_f64<AMP>(p, eps)), ro, (int)voff, (int)soff, 0);
            }
        };
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            // pair (P, Q) = (Z[k], Z[512 - k]): E = (P.x + Q.x, P.y - Q.y), D = (P.x - Q.x, P.y + Q.y), T = W' D with W' = -i W_1024^k:
            //   X[k] = E + T, X[512 - k] = conj(E - T)   (window pre-halved: no 1/2)
            const v2d P = H[u], Q = swp(R[7 - u]);
            const v2d E = pfma(Q, (v2d){1.0, -1.0}, P), D = pfma(Q, (v2d){-1.0, 1.0}, P);
            const v2d T = cmulv(D, twj[u]);
            emit(oa, u * step, pw_a + u * PSTEP, E + T, false);
            emit(ob, (7 - u) * step, pw_b + (7 - u) * PSTEP, E - T, true);
        }
        if (j0 && half == 0u) emit((256u * a.n_frames + p2ofs) * ES, 0u, pwd + pwd_index(256u, p2f), h8 * (v2d){2.0, -2.0}, false);  // X[256] = 2 conj(Z[256])
        if constexpr (MODE == OUT_MEL) {
            __syncthreads();  // |X|^2 tile complete
            mel_tile_sched_d<AMP>(a, pwd, sched, b, f0, nf, eps, tid);
        }
        wid = next;
    }
}

template <int MODE, int AMP>
hipError_t launch_variant_d(const StftArgs &a, hipStream_t s) {
    const unsigned total = a.tiles * a.batch;
    const unsigned per_xcd = (total + 7u) / 8u;
    const unsigned cu_slots = std::max(1u, device_cu_count() / 8u);
    const unsigned nslots = per_xcd < cu_slots ? per_xcd : cu_slots;  // one 512-thread workgroup per CU
    const unsigned chunks = (15u * a.hop + 1024u + 1u) >> 1;         // 16-byte chunks of a tile's samples
    const unsigned lds = (unsigned)kDLdsBase + (MODE == OUT_MEL ? ((a.mel_sched_words * 4u + 15u) & ~15u) + 64u : 0u);
    auto go = [&](auto kernel) -> hipError_t {
        hipError_t e = set_max_dynamic_lds((const void *)kernel, kDLdsMax);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3(nslots * 8), dim3(512), lds, s, a, per_xcd, total, nslots);
        return hipGetLastError();
    };
    if (a.hop & 1u) {
        if (chunks <= 5u * 512u) return go(k_d32x16<MODE, AMP, 5, true>);
        return go(k_d32x16<MODE, AMP, 0, true>);
    }
    if (chunks <= 5u * 512u) return go(k_d32x16<MODE, AMP, 5>);
    return go(k_d32x16<MODE, AMP, 0>);
}

// ====================================================================================================================================
// k_d512: f64 n_fft = 512 at even hops up to 260 (the reference's Mel benchmark shape 512 / 256, benches/spectrogram_benchmarks.rs:105-141, in
// its default type).  TWO consecutive frames per 512-point complex transform — z[n] = a[n] + i b[n], a = frame 2 p, b = frame 2 p + 1, both
// times w[n] / 2 — so a tile is 32 frames in 16 slots and passes 1 and 2 are k_d32x16's; the real split becomes the two-sequence split
// (no twiddles): with (P, Q) = (Z[k], Z[512 - k]):  A[k] = P + conj Q,  B[k] = -i (P - conj Q), k = 0 .. 256.  A lane's 8 pairs give 8 bins
// of two neighbouring frames: 16 lanes x 16 bytes = 256-byte runs.
__host__ __device__ constexpr unsigned pwd512_index(unsigned k, unsigned f) { return k * 32u + f; }

template <int AMP>
__device__ __forceinline__ void mel_tile_sched_d512(const StftArgs &a, const double *pw, const unsigned *sched, unsigned b, unsigned f0, unsigned nf,
                                                    double eps, unsigned tid) {
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u, slot = lane >> 3, fp = lane & 7u;
    constexpr unsigned kDrop = 0x80000000u;
    const unsigned obytes = (a.n_out * a.n_frames - f0) * 8u;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc((const double *)a.out + (size_t)b * a.n_out * a.n_frames + f0, obytes);
    const uint4 *info = (const uint4 *)(sched + 4) + wave * 8u + slot;
    uint4 cur = info[0];
#pragma unroll
    for (unsigned seg = 0; seg < (unsigned)d512::kSegs; ++seg) {
        const uint4 nxt = seg + 1u < (unsigned)d512::kSegs ? info[(seg + 1u) * 64u] : cur;
        const unsigned L = __builtin_amdgcn_readfirstlane(cur.x);
        const bool have = cur.w != 0xffffffffu;
        if (__builtin_amdgcn_ballot_w64(have) != 0ull) {
            const v2d *wr = (const v2d *)(sched + cur.y);
            const unsigned bo = cur.w * a.n_frames * 8u;
#pragma unroll
            for (unsigned h = 0; h < 2u; ++h) {  // the tile's two halves of 16 frames; neighbouring slots take them in opposite order (a bin row is
                                                 // 256 bytes = all banks: the two slots of a 16-lane read group then read different halves of it)
                const unsigned fpair = fp + 8u * ((h + slot) & 1u);
                const v2d *pr = (const v2d *)(pw + cur.z * 32u + fpair * 2u);
                v2d acc = {0.0, 0.0};
                for (unsigned t = 0; t < L; t += 4u) {
                    const v2d w01 = wr[t >> 1], w23 = wr[(t >> 1) + 1u];
                    const v2d q0 = pr[t * 16u], q1 = pr[t * 16u + 16u], q2 = pr[t * 16u + 32u], q3 = pr[t * 16u + 48u];
                    acc = mul_add_unfused_d(w01.x, q0, acc);
                    acc = mul_add_unfused_d(w01.y, q1, acc);
                    acc = mul_add_unfused_d(w23.x, q2, acc);
                    acc = mul_add_unfused_d(w23.y, q3, acc);
                }
                const unsigned fo0 = 2u * fpair < nf ? 16u * fpair : kDrop, fo1 = 2u * fpair + 1u < nf ? 16u * fpair + 8u : kDrop;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, amp_f64<AMP>(acc.x, eps)), ro, (int)((have && fo0 != kDrop) ? bo + fo0 : kDrop), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, amp_f64<AMP>(acc.y, eps)), ro, (int)((have && fo1 != kDrop) ? bo + fo1 : kDrop), 0, 0);
            }
        }
        cur = nxt;
    }
}

template <int MODE, int AMP, int ROUNDS>
__global__ __launch_bounds__(512, 2) void k_d512(StftArgs a, unsigned per_xcd, unsigned total, unsigned slots) {
    constexpr bool BIG = MODE == OUT_MEL && ROUNDS == 9;  // the |X|^2 tile sits behind 72 KiB of staged samples: a longer buffer
    constexpr unsigned BUF = BIG ? (unsigned)d512::kBuf9 : (unsigned)d512::kEx, PWOFF = BIG ? (unsigned)d512::kPwOff9 : (unsigned)d512::kPwOff5;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x;
    unsigned char *tabs = smem + BUF;
    if (tid < 256u) ((v4f *)tabs)[tid] = ((const v4f *)a.window)[tid];  // 512 doubles w[n] / 2
    unsigned *sched = (unsigned *)(tabs + d512::kWinBytes);
    if constexpr (MODE == OUT_MEL)
        for (unsigned i = tid; i < a.mel_sched_words; i += 512u) sched[i] = a.mel_sched[i];

    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned lo = xcd * per_xcd, hi = min(lo + per_xcd, total);
    unsigned wid = lo + slot;

    const unsigned p1s = tid >> 5, n2 = tid & 31u;  // pass-1 identity: slot (frame pair) and column
    const unsigned wave = tid >> 6, lane = tid & 63u, half = lane >> 5, p2s = lane & 15u;
    const unsigned r = wave + 8u * ((lane >> 4) & 1u);
    const unsigned row = half ? ((16u - r) & 15u) : r;
    const bool j0 = r == 0u;
    const unsigned kb = half ? (j0 ? 16u : 32u - r) : r;  // own H[u] = Z[kb + 32 u]: bins kb + 32 u of the slot's two frames
    const double eps = a.eps;
    constexpr unsigned ES = MODE == OUT_COMPLEX ? 16u : 8u;
    const unsigned step = 32u * a.n_frames * ES;
    double *pwd = (double *)(smem + PWOFF);
    v2d twa[4], twb[4];  // W_512^(k1 n2) = twa[k1 >> 2] * twb[k1 & 3]
    {
        const v2d *t1 = (const v2d *)a.tw1 + n2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            twa[q] = t1[32 * 4 * q];
            twb[q] = t1[32 * q];
        }
    }
    const double sg = half ? -1.0 : 1.0, hb = half ? 1.0 : 0.0;

    v4f creg[ROUNDS];
    const unsigned hop = a.hop;
    const unsigned row_bytes = (unsigned)a.n_samples * 8u;
    auto load_tile = [&](unsigned w) {
        const unsigned b = w / a.tiles, f0 = (w - b * a.tiles) * 32u;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc((const double *)a.x + (size_t)b * a.sample_stride, row_bytes);
        const int tile_lo = (int)(f0 * hop) - (int)a.pad;  // (negative in the left padding: out of range as an unsigned offset, reads 0 — S1)
        const int vo = (tile_lo + 2 * (int)tid) * 8;
#pragma unroll
        for (int q = 0; q < ROUNDS; ++q) creg[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rx, vo + q * 8192, 0, 0));
    };
    if (wid < hi) load_tile(wid);
    __syncthreads();  // tables visible

    const unsigned char *xa = smem + (2u * p1s * hop + n2) * 8u;  // a[n2] of this slot; b[n] is hop samples further
    const unsigned hop8 = hop * 8u;
    const double *w1 = (const double *)tabs + n2;

    while (wid < hi) {
        const unsigned b = wid / a.tiles, f0 = (wid - b * a.tiles) * 32u;
        const unsigned nf = min(32u, a.n_frames - f0);
        v2d xr[16];
        {
            v2d e[8], we[8], o[8], wo[8];
#pragma unroll
            for (int q = 0; q < ROUNDS; ++q) *(v4f *)(smem + (q * 512u + tid) * 16u) = creg[q];
            __syncthreads();  // barrier 1: the staged samples are complete
#pragma unroll
            for (int k = 0; k < 8; ++k) {  // n = 32 n1 + n2, n1 = 2 k
                const double w = w1[64 * k];
                e[k] = (v2d){*(const double *)(xa + k * 512), *(const double *)(xa + hop8 + k * 512)};
                we[k] = (v2d){w, w};
            }
            Fft<8, true, v2d>::run(e, we);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; ++k) {  // n1 = 2 k + 1
                const double w = w1[64 * k + 32];
                o[k] = (v2d){*(const double *)(xa + k * 512 + 256), *(const double *)(xa + hop8 + k * 512 + 256)};
                wo[k] = (v2d){w, w};
            }
            Fft<8, true, v2d>::run(o, wo);
            Comb<16, 0, v2d>::run(xr, e, o);
        }
        __syncthreads();  // barrier 2: the columns are read (and the previous tile's band stage is done): pass 1 may write ex
        {
            unsigned char *dst = smem + p1s * kDFS + n2 * 16u;
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) {
                const int qa = k1 >> 2, qb = k1 & 3;
                v2d v = xr[k1];
                if (qb) v = cmulv(v, twb[qb]);
                if (qa) v = cmulv(v, twa[qa]);
                *(v2d *)(dst + k1 * 512) = v;
            }
        }
        const unsigned next = wid + slots;
        if (next < hi) load_tile(next);  // in flight during pass 2
        __syncthreads();  // barrier 3: ex complete
        // a slot without frames (last tile of a signal) mirrors the tile's last slot: same values to the same addresses
        const unsigned se = min(p2s, (nf - 1u) >> 1);
        v2d H[16];
        {
            const unsigned char *rp = smem + se * kDFS + row * 512u;
            double hbl = hb;
            asm volatile("" : "+v"(hbl));
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const v2d x0 = *(const v2d *)(rp + n * 16), x1 = *(const v2d *)(rp + n * 16 + 256);
                v2d d = pfma(x1, (v2d){sg, sg}, x0);
                if (n > 0) {
                    const double c = kCos64[2 * n], s = -kSin64[2 * n];
                    const v2d t = {__builtin_fma(hbl, c - 1.0, 1.0), hbl * s};
                    d = cmulv(d, t);
                }
                H[n] = d;
                if ((n & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();  // barrier 4: ex consumed
        Fft<16, false, v2d>::run(H, H);
        const v2d h8 = H[8];
        v2d R[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            R[j] = H[8 + j];
            trade32(R[j]);
        }
        if (j0) {
#pragma unroll
            for (int j = 0; j < 7; ++j) R[j] = swp(half ? H[8 + j] : H[9 + j]);
            R[7] = swp(half ? H[15] : H[0]);
            asm volatile("" ::: "memory");
        }
        constexpr unsigned kDrop = 0x80000000u;
        const unsigned fa = 2u * se;  // the slot's frames fa, fa + 1 of the tile
        const bool vb = fa + 1u < nf;
        const __amdgpu_buffer_rsrc_t ro = make_rsrc((unsigned char *)a.out + (size_t)b * 257u * a.n_frames * ES, 257u * a.n_frames * ES);
        const unsigned oa = (kb * a.n_frames + f0 + fa) * ES;
        if constexpr (MODE == OUT_MEL) {  // bins 257..267 are read with zero weights
            if (tid < 352u) pwd[pwd512_index(257u + (tid >> 5), tid & 31u)] = 0.0;
        }
        double *pw_a = pwd + pwd512_index(kb, fa);
        constexpr int PSTEP = 32 * 32;  // doubles between bins k and k + 32
        auto emit = [&](unsigned voff, unsigned soff, double *pwp, v2d Xa, v2d Xb) {
            if constexpr (MODE == OUT_MEL) {
                const double pa = __builtin_fma(Xa.x, Xa.x, Xa.y * Xa.y), pb = __builtin_fma(Xb.x, Xb.x, Xb.y * Xb.y);
                *(v2d *)pwp = AMP == AMP_MAG_IN ? (v2d){sqrt(pa), sqrt(pb)} : (v2d){pa, pb};
            } else if constexpr (MODE == OUT_COMPLEX) {  // (16-byte stores: the whole offset in the lane register, see k_d32x16)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, Xa), ro, (int)(voff + soff), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, Xb), ro, (int)(vb ? voff + soff + 16u : kDrop), 0, 0);
            } else {
                const double pa = __builtin_fma(Xa.x, Xa.x, Xa.y * Xa.y), pb = __builtin_fma(Xb.x, Xb.x, Xb.y * Xb.y);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, amp_f64<AMP>(pa, eps)), ro, (int)voff, (int)soff, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, amp_f64<AMP>(pb, eps)), ro, (int)(vb ? voff + 8u : kDrop), (int)soff, 0);
            }
        };
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            // (P, Q) = (Z[k], Z[512 - k]): A[k] = P + conj Q = (P.x + Q.x, P.y - Q.y); B[k] = -i (P - conj Q) = (P.y + Q.y, Q.x - P.x)
            const v2d P = H[u], Q = swp(R[7 - u]);
            const v2d Xa = pfma(Q, (v2d){1.0, -1.0}, P);
            const v2d D = pfma(Q, (v2d){-1.0, 1.0}, P);
            const v2d Xb = (v2d){D.y, -D.x};
            emit(oa, u * step, pw_a + u * PSTEP, Xa, Xb);
        }
        if (j0 && half == 0u)  // bin 256: Z[256] pairs with itself
            emit((256u * a.n_frames + f0 + fa) * ES, 0u, pwd + pwd512_index(256u, fa), (v2d){2.0 * h8.x, 0.0}, (v2d){2.0 * h8.y, 0.0});
        if constexpr (MODE == OUT_MEL) {
            __syncthreads();  // |X|^2 tile complete
            mel_tile_sched_d512<AMP>(a, pwd, sched, b, f0, nf, eps, tid);
        }
        wid = next;
    }
}

template <int MODE, int AMP>
hipError_t launch_variant_d512(const StftArgs &a, hipStream_t s) {
    const unsigned total = a.tiles * a.batch;
    const unsigned per_xcd = (total + 7u) / 8u;
    const unsigned cu_slots = std::max(1u, device_cu_count() / 8u);
    const unsigned nslots = per_xcd < cu_slots ? per_xcd : cu_slots;
    const bool r5 = (31u * a.hop + 512u) * 8u <= 5u * 8192u;  // hop <= 148
    const bool big = MODE == OUT_MEL && !r5;
    const unsigned lds = (big ? (unsigned)d512::kBuf9 : (unsigned)d512::kEx) + (unsigned)d512::kWinBytes +
                         (MODE == OUT_MEL ? ((a.mel_sched_words * 4u + 15u) & ~15u) + 64u : 0u);
    auto go = [&](auto kernel) -> hipError_t {
        hipError_t e = set_max_dynamic_lds((const void *)kernel, 163840);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3(nslots * 8), dim3(512), lds, s, a, per_xcd, total, nslots);
        return hipGetLastError();
    };
    if (r5) return go(k_d512<MODE, AMP, 5>);
    return go(k_d512<MODE, AMP, 9>);
}

}  // namespace

bool plan_geometry_d32x16_f64(StftArgs &a) {
    if (a.n_fft != 1024) return false;  // (any hop since round 5: odd ones read their staged pairs as two 8-byte reads / patch the row-start pair)
    // batches of short signals: a tile is 16 frames of ONE signal, so 5-frame signals leave 11/16 of every tile idle and the register-tiled
    // kernel (tiles of 1-2 frames at this length) is faster: 16 384 x 5 frames 351 us here against 254 us (17 frames: 175 against 227)
    if (a.x != nullptr && a.n_frames < 8u) return false;
    // filterbank outputs need the band schedule (built on the host before this is asked; a bank without one takes the register-tiled kernel)
    if (a.out_mode == OUT_MEL && (a.mel_sched_words == 0 || a.mel_sched_words > (unsigned)kDSchMaxWords)) return false;
    if (a.n_samples >= (1ull << 28)) return false;                                        // 32-bit byte offsets into a sample row
    if ((unsigned long long)a.n_frames * 513ull * 16ull >= 0x7fffffffull) return false;  // and into one output signal
    a.ft = 16;
    return true;
}

hipError_t launch_d32x16_f64(const StftArgs &a, hipStream_t s) {
    const unsigned long long total64 = (unsigned long long)a.tiles * a.batch;
    if (total64 == 0 || total64 >= 0x7ffffff0ull) return hipErrorInvalidConfiguration;
    if (a.out_mode == OUT_COMPLEX) return launch_variant_d<OUT_COMPLEX, AMP_POWER>(a, s);
    if (a.out_mode == OUT_MEL) {
        if (a.amp == AMP_MAGNITUDE) return launch_variant_d<OUT_MEL, AMP_MAGNITUDE>(a, s);
        if (a.amp == AMP_DB) return launch_variant_d<OUT_MEL, AMP_DB>(a, s);
        if (a.amp == AMP_MAG_IN) return launch_variant_d<OUT_MEL, AMP_MAG_IN>(a, s);
        return launch_variant_d<OUT_MEL, AMP_POWER>(a, s);
    }
    if (a.amp == AMP_MAGNITUDE) return launch_variant_d<OUT_LINEAR, AMP_MAGNITUDE>(a, s);
    if (a.amp == AMP_DB) return launch_variant_d<OUT_LINEAR, AMP_DB>(a, s);
    return launch_variant_d<OUT_LINEAR, AMP_POWER>(a, s);
}


bool plan_geometry_d512_f64(StftArgs &a) {
    if (a.n_fft != 512 || a.hop > 260u) return false;  // (the whole 32-frame tile is staged: (31 hop + 512) * 8 <= 9 * 8192; odd hops too: the samples are read one by one)
    if (a.x != nullptr && a.n_frames < 16u) return false;  // batches of short signals: mostly empty 32-frame tiles (see plan_geometry_d32x16_f64)
    if (a.out_mode == OUT_MEL && (a.mel_sched_words == 0 || a.mel_sched_words > (unsigned)d512::kSchMaxWords)) return false;
    if (a.n_samples >= (1ull << 28)) return false;
    if ((unsigned long long)a.n_frames * 257ull * 16ull >= 0x7fffffffull) return false;
    a.ft = 32;
    return true;
}

hipError_t launch_d512_f64(const StftArgs &a, hipStream_t s) {
    const unsigned long long total64 = (unsigned long long)a.tiles * a.batch;
    if (total64 == 0 || total64 >= 0x7ffffff0ull) return hipErrorInvalidConfiguration;
    if (a.out_mode == OUT_COMPLEX) return launch_variant_d512<OUT_COMPLEX, AMP_POWER>(a, s);
    if (a.out_mode == OUT_MEL) {
        if (a.amp == AMP_MAGNITUDE) return launch_variant_d512<OUT_MEL, AMP_MAGNITUDE>(a, s);
        if (a.amp == AMP_DB) return launch_variant_d512<OUT_MEL, AMP_DB>(a, s);
        if (a.amp == AMP_MAG_IN) return launch_variant_d512<OUT_MEL, AMP_MAG_IN>(a, s);
        return launch_variant_d512<OUT_MEL, AMP_POWER>(a, s);
    }
    if (a.amp == AMP_MAGNITUDE) return launch_variant_d512<OUT_LINEAR, AMP_MAGNITUDE>(a, s);
    if (a.amp == AMP_DB) return launch_variant_d512<OUT_LINEAR, AMP_DB>(a, s);
    return launch_variant_d512<OUT_LINEAR, AMP_POWER>(a, s);
}

}  // namespace sgx
