// kernels_r64x32.hip — tuned f32, n_fft = 4096 STFT kernel for gfx950 (round 4): per-bin, complex and (up to hop 1170) filterbank outputs
// (longer hops: this kernel's per-bin power, then k_bank_rows).  k_d32x16's construction (kernels_d32x16.hip) at 2048 complex f32 points: a tile = 8
// consecutive frames of one signal, one persistent 512-thread workgroup per CU, a 128 KiB exchange buffer ex[f][k1][n2] of 8-byte elements.
//
//   pass 1  lane (f = 0..7, n2 = 0..63) owns z[64 n1 + n2], n1 = 0..31, of frame f, z[n] = w[2n] x[2n] + i w[2n+1] x[2n+1] (window pre-scaled
//           by 1/2 on the host — exact); one 32-point FFT in registers; twiddle W_2048^(k1 n2); one ds_write_b64 per value.
//   pass 2  32 rows of 64 points.  The real split pairs Z[k] with Z[2048 - k] = row 32 - k1, element 63 - k2: the even-indexed outputs of
//           row r pair with the odd-indexed outputs of row 32 - r.  A half row (one decimation-in-frequency step as the row is read, then a
//           32-point FFT: 64 data registers) is one lane's work; the partner halves sit in lanes l and l + 32 of one wave and trade the
//           upper 16 values with v_permlane32_swap_b32 (32 instructions).  Each lane then splits 16 pairs = 32 bins: own H[u] = Z[kb + 64 u]
//           with the partner's Z[2048 - kb - 64 u], kb = r (half 0) or 64 - r (half 1).  Row 0's halves pair inside themselves (kb = 0, 32).
//           32 rows x 2 halves x 8 frames = the 512 lanes.
//   store   the 8 lanes of a (row, half) hold one bin of 8 consecutive frames: 32-byte runs (64 for the complex STFT).
//
// Samples: the tile's 7 hop + 4096 samples once, 16-byte buffer loads one tile ahead, staged in LDS over the idle exchange buffer (hop <= 2048);
// longer hops load their columns per lane.  Reference semantics: spectrogram.rs:1301-1334, :2068-2080.
#include <type_traits>
#include <utility>

#include "buffer_ops.h"
#include "fft_inreg.h"
#include "lane_pair.h"
#include "sgx_internal.h"

namespace sgx {
namespace {

using namespace inreg;

typedef int v2i __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

constexpr int kQFS = 16384 + 16;           // LDS bytes per frame of ex[f][32][64] (16 B more: the 8 frames of a read group on distinct banks)
constexpr int kQEx = 8 * kQFS;             // 131200: exchange buffer; also holds the staged samples (<= 73728 B)
constexpr int kQWinOff = 0;                // tables behind it: v2f win[2048] = (w[2n], w[2n+1]) / 2
constexpr int kQTw2Off = 16384;            // v2f tw2[64][16]: entry u of lane kind kb = W' = -i W_4096^(kb + 64 u)
constexpr int kQLds = kQEx + kQTw2Off + 64 * 16 * 8;  // 155776
// filterbank outputs (hop <= 1170: 6 staging rounds): the |X|^2 tile (2060 bins x 8 frames of f32: bin k, frame f at k * 8 + f) sits in the upper
// half of the exchange buffer, above the staged samples; the band schedule (plan.hip build_band_schedule: 16 half-waves x 8 slots) stays in GLOBAL
// memory (kernels_d32x32.hip)
constexpr int kQPwOff = kQEx - 2060 * 32;  // 65280 >= 6 * 8192
constexpr int kQSegs = 2;
__host__ __device__ constexpr unsigned pwq_index(unsigned k, unsigned f) { return k * 8u + f; }

template <int AMP>
__device__ __forceinline__ float amp_q(float p, float eps) {
    if constexpr (AMP == AMP_MAGNITUDE) return sqrtf(p);
    else if constexpr (AMP == AMP_DB) return __builtin_log2f(fmaxf(p, eps)) * 3.01029995663981195f;  // as kernels_r32x16.hip
    else return p;
}

using lanepair::trade32;  // lanes l and l ^ 32 trade a complex value, each receives the other's as (im, re): lane_pair.h

__device__ __forceinline__ v2f mul_add_unfused_q(float w, v2f p, v2f acc) {  // the reference's `acc += w * x`: two roundings (spectrogram.rs:102-117)
#pragma clang fp contract(off)
    const v2f m = (v2f){w, w} * p;
    return m + acc;
}

// band stage over the tile's 8 frames: 16 half-waves x 8 slots x 4 frame pairs; a lane sums one band for two frames in ascending-bin order
template <int AMP>
__device__ __forceinline__ void mel_tile_sched_q(const StftArgs &a, const float *pw, const unsigned *sched, unsigned b, unsigned f0, unsigned nf,
                                                 float eps, unsigned tid) {
    const unsigned vw = tid >> 5, slot = (tid >> 2) & 7u, fp = tid & 3u;
    constexpr unsigned kDrop = 0x80000000u;  // past the descriptor's range: the hardware drops the store
    const unsigned obytes = (a.n_out * a.n_frames - f0) * 4u;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc((const float *)a.out + (size_t)b * a.n_out * a.n_frames + f0, obytes);
    const unsigned fo0 = 2u * fp < nf ? 8u * fp : kDrop, fo1 = 2u * fp + 1u < nf ? 8u * fp + 4u : kDrop;
    const uint4 *info = (const uint4 *)(sched + 4) + vw * 8u + slot;
#pragma unroll
    for (unsigned seg = 0; seg < (unsigned)kQSegs; ++seg) {
        const uint4 cur = info[seg * 128u];
        const unsigned L = cur.x;  // (per half-wave: the two halves of a wave run to the longer one)
        const bool have = cur.w != 0xffffffffu;
        if (!have) continue;
        const v4f *wr = (const v4f *)((const float *)sched + cur.y);
        const v2f *pr = (const v2f *)(pw + cur.z * 8u + fp * 2u);
        v2f acc = {0.f, 0.f};
        for (unsigned t = 0; t < L; t += 4u) {  // bins t .. t + 3, frames 2 fp and 2 fp + 1 of each
            const v4f w4 = wr[t >> 2];
            const v2f q0 = pr[t * 4u], q1 = pr[t * 4u + 4u], q2 = pr[t * 4u + 8u], q3 = pr[t * 4u + 12u];
            acc = mul_add_unfused_q(w4.x, q0, acc);
            acc = mul_add_unfused_q(w4.y, q1, acc);
            acc = mul_add_unfused_q(w4.z, q2, acc);
            acc = mul_add_unfused_q(w4.w, q3, acc);
        }
        const unsigned bo = cur.w * a.n_frames * 4u;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, amp_q<AMP>(acc.x, eps)), ro, (int)(fo0 != kDrop ? bo + fo0 : kDrop), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, amp_q<AMP>(acc.y, eps)), ro, (int)(fo1 != kDrop ? bo + fo1 : kDrop), 0, 0);
    }
}

template <int MODE, int AMP, int ROUNDS>
__global__ __launch_bounds__(512, 2) void k_r64x32(StftArgs a, unsigned per_xcd, unsigned total, unsigned slots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x;
    unsigned char *tabs = smem + kQEx;
    ((v4f *)(tabs + kQWinOff))[tid] = ((const v4f *)a.window)[tid];                 // 4096 floats, 1024 x 16 B
    ((v4f *)(tabs + kQWinOff))[tid + 512u] = ((const v4f *)a.window)[tid + 512u];
    ((v4f *)(tabs + kQTw2Off))[tid] = ((const v4f *)a.tw2)[tid];                    // 8192 B

    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned lo = xcd * per_xcd, hi = min(lo + per_xcd, total);
    unsigned wid = lo + slot;

    const unsigned p1f = tid >> 6, n2 = tid & 63u;  // pass-1 identity
    const unsigned wave = tid >> 6, lane = tid & 63u, half = lane >> 5, p2f = lane & 7u;
    const unsigned r = wave + 8u * ((lane >> 3) & 3u);  // pass-2 job: row r (half 0: its even outputs) with row 32 - r (half 1: its odd outputs)
    const unsigned row = half ? ((32u - r) & 31u) : r;
    const bool j0 = r == 0u;
    const unsigned kb = half ? (j0 ? 32u : 64u - r) : r;  // own H[u] = Z[kb + 64 u]
    const float eps = (float)a.eps;
    constexpr unsigned ES = MODE == OUT_COMPLEX ? 8u : 4u;
    const unsigned step = 64u * a.n_frames * ES;  // uniform: 64 bins further
    float *pwq = (float *)(smem + kQPwOff);
    const v2f *twj = (const v2f *)(tabs + kQTw2Off) + kb * 16u;
    v2f twa[4], twb[8];  // W_2048^(k1 n2) = twa[k1 >> 3] * twb[k1 & 7]
    {
        const v2f *t1 = (const v2f *)a.tw1 + n2;
#pragma unroll
        for (int q = 0; q < 4; ++q) twa[q] = t1[64 * 8 * q];
#pragma unroll
        for (int q = 0; q < 8; ++q) twb[q] = t1[64 * q];
    }
    const float sg = half ? -1.f : 1.f, hb = half ? 1.f : 0.f;

    constexpr int NCR = ROUNDS > 0 ? ROUNDS : 1;
    v4f creg[NCR];
    v2f xd[ROUNDS > 0 ? 1 : 32];
    const unsigned hop = a.hop;
    const unsigned row_bytes = (unsigned)a.n_samples * 4u;  // host: n_samples < 2^29
    auto load_tile = [&](unsigned w) {
        const unsigned b = w / a.tiles, f0 = (w - b * a.tiles) * 8u;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc((const float *)a.x + (size_t)b * a.sample_stride, row_bytes);
        const int tile_lo = (int)(f0 * hop) - (int)a.pad;  // (negative in the left padding: out of range as an unsigned offset, reads 0 — S1)
        if constexpr (ROUNDS > 0) {
            const int vo = (tile_lo + 4 * (int)tid) * 4;
#pragma unroll
            for (int q = 0; q < ROUNDS; ++q) creg[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rx, vo + q * 8192, 0, 0));
        } else {
            const int vo = ((int)(p1f * hop) + tile_lo + 2 * (int)n2) * 4;  // (even hop: a pair never straddles the row start)
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) {
                int o = vo + n1 * 512;
                asm("" : "+v"(o));  // the whole offset in the lane register: an immediate part is added without wrapping (buffer_ops.h)
                xd[n1] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rx, o, 0, 0));
            }
            if (hop & 1u) {  // uniform (round 5: odd hops).  The pair (x[-1], x[0]) of an odd frame starts outside the row, and an 8-byte access whose first
                             // dword is out of range returns 0 for both: put x[0] back (as k_r32x16)
                const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, 0, 0, 0));
                const int s0 = (int)(p1f * hop) + tile_lo + 2 * (int)n2;
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1)
                    if (s0 + 128 * n1 == -1) xd[n1].y = x0;
            }
        }
    };
    if (wid < hi) load_tile(wid);
    __syncthreads();  // tables visible

    const unsigned char *xs = smem + p1f * hop * 4u + n2 * 8u;
    const v2f *w2 = (const v2f *)(tabs + kQWinOff) + n2;

    while (wid < hi) {
        const unsigned b = wid / a.tiles, f0 = (wid - b * a.tiles) * 8u;
        const unsigned nf = min(8u, a.n_frames - f0);
        v2f xr[32];
        {
            v2f e[16], we[16], o[16], wo[16];
            if constexpr (ROUNDS > 0) {
#pragma unroll
                for (int q = 0; q < ROUNDS; ++q) *(v4f *)(smem + (q * 512u + tid) * 16u) = creg[q];
                __syncthreads();  // barrier 1: the staged samples are complete
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {  // n1 = 2 k
                if constexpr (ROUNDS > 0) {  // (odd hops: an odd frame's pairs sit at 4-byte-aligned addresses — two 4-byte reads, ds_read2_b32)
                    if (hop & 1u) e[k] = (v2f){*(const float *)(xs + k * 1024), *(const float *)(xs + k * 1024 + 4)};
                    else e[k] = *(const v2f *)(xs + k * 1024);
                } else e[k] = xd[2 * k];
                we[k] = w2[128 * k];
            }
            Fft<16, true>::run(e, we);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 16; ++k) {  // n1 = 2 k + 1
                if constexpr (ROUNDS > 0) {
                    if (hop & 1u) o[k] = (v2f){*(const float *)(xs + k * 1024 + 512), *(const float *)(xs + k * 1024 + 516)};
                    else o[k] = *(const v2f *)(xs + k * 1024 + 512);
                } else o[k] = xd[2 * k + 1];
                wo[k] = w2[128 * k + 64];
            }
            Fft<16, true>::run(o, wo);
            Comb<32, 0, v2f>::run(xr, e, o);
        }
        // barrier 2: every wave has read its columns (and, filterbank outputs, finished the previous tile's band stage): pass 1 may write ex
        if constexpr (ROUNDS > 0 || MODE == OUT_MEL) __syncthreads();
        {
            unsigned char *dst = smem + p1f * kQFS + n2 * 8u;
#pragma unroll
            for (int k1 = 0; k1 < 32; ++k1) {  // twiddle by W_2048^(k1 n2), write row k1 of this lane's column
                const int qa = k1 >> 3, qb = k1 & 7;
                v2f v = xr[k1];
                if (qb) v = cmulv(v, twb[qb]);
                if (qa) v = cmulv(v, twa[qa]);
                *(v2f *)(dst + k1 * 512) = v;
            }
        }
        const unsigned next = wid + slots;
        if (next < hi) load_tile(next);  // in flight during pass 2
        __syncthreads();  // barrier 3: ex complete
        // pass 2.  A lane whose frame does not exist (last tile of a signal) mirrors the tile's last frame: same values to the same addresses.
        const unsigned fe = min(p2f, nf - 1u);
        v2f H[32];
        {
            const unsigned char *rp = smem + fe * kQFS + row * 512u;
            float hbl = hb;
            asm volatile("" : "+v"(hbl));  // (not loop-invariant: the 31 lane twiddles below would otherwise be kept in registers across tiles)
            // one decimation-in-frequency step: even outputs x[n] + x[n + 32]; odd: (x[n] - x[n + 32]) W_64^n
#pragma unroll
            for (int n = 0; n < 32; n += 2) {
                const v4f a0 = *(const v4f *)(rp + n * 8), a1 = *(const v4f *)(rp + n * 8 + 256);
                v2f d0 = pfma((v2f){a1.x, a1.y}, (v2f){sg, sg}, (v2f){a0.x, a0.y});
                v2f d1 = pfma((v2f){a1.z, a1.w}, (v2f){sg, sg}, (v2f){a0.z, a0.w});
                if (n > 0) {
                    const float c = (float)kCos64[n], s = (float)-kSin64[n];  // W_64^n
                    d0 = cmulv(d0, (v2f){__builtin_fmaf(hbl, c - 1.f, 1.f), hbl * s});  // half 0: 1; half 1: W_64^n
                }
                {
                    const float c = (float)kCos64[n + 1], s = (float)-kSin64[n + 1];
                    d1 = cmulv(d1, (v2f){__builtin_fmaf(hbl, c - 1.f, 1.f), hbl * s});
                }
                H[n] = d0;
                H[n + 1] = d1;
                if ((n & 7) == 6) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();  // barrier 4: ex consumed: the next staging may overwrite it
        Fft<32, false>::run(H, H);
        const v2f h16 = H[16];
        v2f R[16];  // R[j] = the partner's H[16 + j] as (im, re)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            R[j] = H[16 + j];
            trade32(R[j]);
        }
        if (j0) {
            // row 0: both halves pair inside themselves.  Half 0 (E[m] = Z[64 m]): E[u] with E[32 - u] (u = 0: Z[0] with itself gives bins 0 and
            // 2048; E[16] = Z[1024] pairs with itself, below).  Half 1 (O[m] = Z[32 + 64 m]): O[u] with O[31 - u].
#pragma unroll
            for (int j = 0; j < 15; ++j) R[j] = swp(half ? H[16 + j] : H[17 + j]);
            R[15] = swp(half ? H[31] : H[0]);
            asm volatile("" ::: "memory");  // keeps this a branch
        }
        const unsigned p2ofs = f0 + fe;
        const __amdgpu_buffer_rsrc_t ro = make_rsrc((unsigned char *)a.out + (size_t)b * 2049u * a.n_frames * ES, 2049u * a.n_frames * ES);
        // bins kb + 64 u upwards; the mirrored bins 2048 - kb - 64 u count down: lane part 15 steps low, scalar part (15 - u) steps
        const unsigned oa = (kb * a.n_frames + p2ofs) * ES, ob = ((2048u - 960u - kb) * a.n_frames + p2ofs) * ES;
        if constexpr (MODE == OUT_MEL) {  // bins 2049..2059 are read with zero weights
            if (tid < 88u) pwq[pwq_index(2049u + (tid >> 3), tid & 7u)] = 0.f;
        }
        float *pw_a = pwq + pwq_index(kb, p2f), *pw_b = pwq + pwq_index(2048u - 960u - kb, p2f);
        constexpr int PSTEP = 64 * 8;  // floats between bins k and k + 64 in the |X|^2 tile
        auto emit = [&](unsigned voff, unsigned soff, float *pwp, v2f X, bool conj) {
            if constexpr (MODE == OUT_MEL) {
                const float p = __builtin_fmaf(X.x, X.x, X.y * X.y);
                *pwp = AMP == AMP_MAG_IN ? sqrtf(p) : p;  // (a lane without a frame writes its mirror's values into its own slot: never stored)
            } else if constexpr (MODE == OUT_COMPLEX) {
                const v2f V = conj ? (v2f){X.x, -X.y} : X;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, V), ro, (int)voff, (int)soff, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, amp_q<AMP>(__builtin_fmaf(X.x, X.x, X.y * X.y), eps)), ro, (int)voff, (int)soff, 0);
            }
        };
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            // pair (P, Q) = (Z[k], Z[2048 - k]): E = (P.x + Q.x, P.y - Q.y), D = (P.x - Q.x, P.y + Q.y), T = W' D with W' = -i W_4096^k:
            //   X[k] = E + T, X[2048 - k] = conj(E - T)   (window pre-halved: no 1/2)
            const v2f P = H[u], Q = swp(R[15 - u]);
            const v2f E = pfma(Q, (v2f){1.f, -1.f}, P), D = pfma(Q, (v2f){-1.f, 1.f}, P);
            const v2f T = cmulv(D, twj[u]);
            emit(oa, u * step, pw_a + u * PSTEP, E + T, false);
            emit(ob, (15 - u) * step, pw_b + (15 - u) * PSTEP, E - T, true);
        }
        if (j0 && half == 0u) emit((1024u * a.n_frames + p2ofs) * ES, 0u, pwq + pwq_index(1024u, p2f), h16 * (v2f){2.f, -2.f}, false);  // X[1024] = 2 conj(Z[1024])
        if constexpr (MODE == OUT_MEL) {
            __syncthreads();  // |X|^2 tile complete
            mel_tile_sched_q<AMP>(a, pwq, a.mel_sched, b, f0, nf, eps, tid);
        }
        wid = next;
    }
}

template <int MODE, int AMP>
hipError_t launch_variant_q(const StftArgs &a, hipStream_t s) {
    const unsigned total = a.tiles * a.batch;
    const unsigned per_xcd = (total + 7u) / 8u;
    const unsigned cu_slots = std::max(1u, device_cu_count() / 8u);
    const unsigned nslots = per_xcd < cu_slots ? per_xcd : cu_slots;  // one 512-thread workgroup per CU
    const unsigned bytes = (7u * a.hop + 4096u) * 4u;                 // a tile's samples
    auto go = [&](auto kernel) -> hipError_t {
        hipError_t e = set_max_dynamic_lds((const void *)kernel, kQLds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3(nslots * 8), dim3(512), kQLds, s, a, per_xcd, total, nslots);
        return hipGetLastError();
    };
    if (bytes <= 6u * 8192u) return go(k_r64x32<MODE, AMP, 6>);
    if constexpr (MODE == OUT_MEL) {
        return hipErrorInvalidConfiguration;  // (plan_geometry_r64x32_f32 keeps such hops away)
    } else {
        if (bytes <= 9u * 8192u) return go(k_r64x32<MODE, AMP, 9>);
        return go(k_r64x32<MODE, AMP, 0>);
    }
}

}  // namespace

bool plan_geometry_r64x32_f32(StftArgs &a) {
    if (a.n_fft != 4096) return false;  // (any hop since round 5)
    // filterbank outputs: fused up to hop 1170 (6 staging rounds below the |X|^2 tile) where the bank has a band schedule; else per-bin power + k_bank_rows
    if (a.out_mode == OUT_MEL && (a.mel_sched_words == 0 || (7u * a.hop + 4096u) * 4u > 6u * 8192u)) return false;
    if (a.x != nullptr && a.n_frames < 4u) return false;                                      // batches of very short signals: mostly empty tiles
    if (a.n_samples >= (1ull << 29)) return false;                                        // 32-bit byte offsets into a sample row
    if ((unsigned long long)a.n_frames * 2049ull * 8ull >= 0x7fffffffull) return false;  // and into one output signal
    a.ft = 8;
    return true;
}

hipError_t launch_r64x32_f32(const StftArgs &a, hipStream_t s) {
    const unsigned long long total64 = (unsigned long long)a.tiles * a.batch;
    if (total64 == 0 || total64 >= 0x7ffffff0ull) return hipErrorInvalidConfiguration;
    if (a.out_mode == OUT_COMPLEX) return launch_variant_q<OUT_COMPLEX, AMP_POWER>(a, s);
    if (a.out_mode == OUT_MEL) {
        if (a.amp == AMP_MAGNITUDE) return launch_variant_q<OUT_MEL, AMP_MAGNITUDE>(a, s);
        if (a.amp == AMP_DB) return launch_variant_q<OUT_MEL, AMP_DB>(a, s);
        if (a.amp == AMP_MAG_IN) return launch_variant_q<OUT_MEL, AMP_MAG_IN>(a, s);
        return launch_variant_q<OUT_MEL, AMP_POWER>(a, s);
    }
    if (a.amp == AMP_MAGNITUDE) return launch_variant_q<OUT_LINEAR, AMP_MAGNITUDE>(a, s);
    if (a.amp == AMP_DB) return launch_variant_q<OUT_LINEAR, AMP_DB>(a, s);
    return launch_variant_q<OUT_LINEAR, AMP_POWER>(a, s);
}

}  // namespace sgx
