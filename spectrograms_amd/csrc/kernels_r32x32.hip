// kernels_r32x32.hip — tuned f32, n_fft = 2048 STFT kernel for gfx950 (round 4): the reference's music default, n_fft 2048 / hop 512
// (src/spectrogram.rs:4243-4248; the first shape of its Criterion suite, benches/stft_benchmarks.rs:27-30).
//
// k_r32x16's construction at 1024 complex points; a tile = 16 consecutive frames of one signal, one persistent 512-thread workgroup
// per CU (the tile's exchange buffer is 128 KiB):
//
//   pass 1  lane (f = 0..15, n2 = 0..31) owns z[32 n1 + n2], n1 = 0..31, of frame f, z[n] = w[2n] x[2n] + i w[2n+1] x[2n+1] (window
//           pre-scaled by 1/2 on the host — exact); window multiply fused into the first butterflies; one 32-point FFT in registers;
//           twiddle W_1024^(k1 n2) from two short per-lane register tables; one ds_write_b64 per value into ex[f][k1][n2].
//   pass 2  a row of the exchange has 32 points, and the real split pairs Z[k] with Z[1024 - k] = row 32 - k1, element 31 - k2: two whole
//           rows are 128 data registers.  So a lane takes HALF of each: the even-indexed outputs of row r and the odd-indexed outputs of
//           row 32 - r — one decimation-in-frequency step (x[n] +- x[n + 16], the odd half times W_32^n) on each row as it is read, then
//           two 16-point FFTs — which are exactly the partners of each other: Z[r + 64 m] pairs with Z[(32 - r) + 32 (2 (15 - m) + 1)].
//           31 such jobs (r = 1..31; r = 16 pairs with itself) and job 0 = both halves of row 0, whose halves pair inside themselves:
//           32 jobs x 16 frames = the 512 lanes.  Every row is read twice (once per half); nothing else is done twice.
//   store   the 16 lanes of a job hold one bin of 16 consecutive frames: 64-byte runs of the reference's frame-contiguous layout (S9).
//           Filterbank outputs: |X|^2 to LDS (transposed, paired: pwt_index), reduced per (band, frame pair) in ascending-bin order
//           (spectrogram.rs:102-117) along a host-built schedule over the 8 waves.
//
// Samples: the tile's 15 hop + 2048 samples once, 16-byte buffer loads through the row's descriptor (out-of-range dwords read 0: the
// zero centre padding, spectrogram.rs:1301-1320), one tile ahead, staged in LDS over the idle exchange buffer (hop <= 544); longer hops
// load their columns per lane.  Reference semantics: spectrogram.rs:1301-1334, :1845-1865, :2068-2080.
#include <type_traits>
#include <utility>

#include "buffer_ops.h"
#include "fft_inreg.h"
#include "r32x16_layout.h"
#include "sgx_internal.h"
#include "xcd_map.h"

namespace sgx {
namespace {

using namespace inreg;
using namespace r32x32;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float db2_f32(float x) { return __builtin_log2f(x) * 3.01029995663981195f; }  // as kernels_r32x16.hip
template <int AMP>
__device__ __forceinline__ float amp2_f32(float p, float eps) {
    if constexpr (AMP == AMP_MAGNITUDE) return sqrtf(p);
    else if constexpr (AMP == AMP_DB) return db2_f32(fmaxf(p, eps));
    else return p;
}
__device__ __forceinline__ float power2_of(v2f x) { return __builtin_fmaf(x.x, x.x, x.y * x.y); }
__device__ __forceinline__ unsigned lds_addr2(const void *p) { return (unsigned)(size_t)p; }

// single-issue 8-byte LDS reads (hipcc fuses neighbours into half-rate ds_read2_b64), collected by an asm s_waitcnt that names them
template <int OFF>
__device__ __forceinline__ void ds_rd64(v2f &d, unsigned addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void tie8x(v2f *d) {
    if constexpr (N >= 0)
        asm volatile("s_waitcnt lgkmcnt(%8)"
                     : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
                     : "n"(N));
    else
        asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
}
template <int N>
__device__ __forceinline__ void tie16x(v2f (&d)[16]) {
    tie8x<N>(&d[0]);
    tie8x<-1>(&d[8]);
}
template <int PAR, int... K>
__device__ __forceinline__ void read_cols2(v2f (&x)[16], v2f (&w)[16], unsigned xaddr, unsigned waddr, std::integer_sequence<int, K...>) {
    ((ds_rd64<(2 * K + PAR) * 256>(x[K], xaddr), ds_rd64<(2 * K + PAR) * 256>(w[K], waddr)), ...);
}

// Odd hops (round 5): odd frames start on an odd sample — their pairs sit at 4-byte-aligned LDS addresses, where ds_read_b64 takes several passes;
// ds_read2_b32 reads the pair as two dwords with no alignment to respect (as k_r32x16's read_cols_odd); one base register per 4 columns keeps
// its 8-bit dword offsets in range (a column = 256 bytes = 64 dwords)
template <int O0, int O1>
__device__ __forceinline__ void ds_rd2x32(v2f &d, unsigned addr) {
    static_assert(O0 >= 0 && O0 < 256 && O1 >= 0 && O1 < 256, "ds_read2_b32: 8-bit dword offsets");
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(d) : "v"(addr), "n"(O0), "n"(O1));
}
template <int PAR, int... K>
__device__ __forceinline__ void read_cols2_odd(v2f (&x)[16], v2f (&w)[16], const unsigned (&base)[8], unsigned waddr, std::integer_sequence<int, K...>) {
    ((ds_rd2x32<((2 * K + PAR) & 3) * 64, ((2 * K + PAR) & 3) * 64 + 1>(x[K], base[(2 * K + PAR) >> 2]), ds_rd64<(2 * K + PAR) * 256>(w[K], waddr)), ...);
}

__host__ __device__ constexpr unsigned pwt2_index(unsigned k, unsigned f) { return (k >> 1) * 32u + (f >> 1) * 4u + (k & 1u) * 2u + (f & 1u); }
__device__ __forceinline__ v2f mul_add_unfused2(float w, v2f p, v2f acc) {
#pragma clang fp contract(off)
    const v2f m = (v2f){w, w} * p;
    return m + acc;
}

// one row of the exchange: 32 points as 16 ds_read_b128
__device__ __forceinline__ void read_row32(const unsigned char *row, v2f (&x)[32]) {
    const v4f *p = (const v4f *)row;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const v4f q = p[c];
        x[2 * c] = (v2f){q.x, q.y};
        x[2 * c + 1] = (v2f){q.z, q.w};
    }
}
// the 16 even-indexed / odd-indexed outputs of a 32-point DFT: one decimation-in-frequency step, then a 16-point transform
template <int... N>
__device__ __forceinline__ void dif_odd(const v2f (&x)[32], v2f (&b)[16], std::integer_sequence<int, N...>) {
    ((b[N] = cmul_wn<32, N, v2f>(x[N] - x[N + 16])), ...);
}
__device__ __forceinline__ void half_even(const v2f (&x)[32], v2f (&z)[16]) {
#pragma unroll
    for (int n = 0; n < 16; ++n) z[n] = x[n] + x[n + 16];
    Fft<16, false>::run(z, z);
}
__device__ __forceinline__ void half_odd(const v2f (&x)[32], v2f (&z)[16]) {
    dif_odd(x, z, std::make_integer_sequence<int, 16>{});
    Fft<16, false>::run(z, z);
}

// band stage over the tile's 16 frames, 8 waves x 8 slots x 8 frame pairs (the r32x16 stage with twice the waves)
template <int AMP>
__device__ __forceinline__ void mel_tile_sched8(const StftArgs &a, const float *pwT, const unsigned *sched, unsigned b, unsigned f0, unsigned nf,
                                                float eps, unsigned tid) {
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u, slot = lane >> 3, fp = lane & 7u;
    constexpr unsigned kDrop = 0x80000000u;  // past the descriptor's range: the hardware drops the store
    const unsigned obytes = (a.n_out * a.n_frames - f0) * 4u;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc((const float *)a.out + (size_t)b * a.n_out * a.n_frames + f0, obytes);
    const unsigned fo0 = 2u * fp < nf ? 8u * fp : kDrop, fo1 = 2u * fp + 1u < nf ? 8u * fp + 4u : kDrop;
    const uint4 *info = (const uint4 *)(sched + r32x16::kSchedHdr) + wave * 8u + slot;
    uint4 cur = info[0];
    // (fixed trip count, unconditional stores: the compiler counts them behind the next tile's sample loads instead of waiting vmcnt(0))
#pragma unroll
    for (unsigned seg = 0; seg < (unsigned)kSegs2; ++seg) {
        const uint4 nxt = seg + 1u < (unsigned)kSegs2 ? info[(seg + 1u) * 64u] : cur;  // fetched ahead (the table holds exactly kSegs2 segments)
        const unsigned L = __builtin_amdgcn_readfirstlane(cur.x);
        const v4f *wr = (const v4f *)((const float *)sched + cur.y);
        const v4f *pr = (const v4f *)(pwT + (cur.z >> 1) * 32u) + fp;  // kstart is even
        v2f acc = {0.0f, 0.0f};
        for (unsigned t = 0; t < L; t += 4u) {  // q0 = (bin t: frames 2 fp, 2 fp + 1; bin t + 1: the same two frames)
            const v4f w4 = wr[t >> 2], q0 = pr[(t >> 1) * 8u], q1 = pr[(t >> 1) * 8u + 8u];
            acc = mul_add_unfused2(w4.x, (v2f){q0.x, q0.y}, acc);
            acc = mul_add_unfused2(w4.y, (v2f){q0.z, q0.w}, acc);
            acc = mul_add_unfused2(w4.z, (v2f){q1.x, q1.y}, acc);
            acc = mul_add_unfused2(w4.w, (v2f){q1.z, q1.w}, acc);
        }
        const bool have = cur.w != 0xffffffffu;
        const unsigned bo = cur.w * a.n_frames * 4u;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, amp2_f32<AMP>(acc.x, eps)), ro, (int)((have && fo0 != kDrop) ? bo + fo0 : kDrop), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, amp2_f32<AMP>(acc.y, eps)), ro, (int)((have && fo1 != kDrop) ? bo + fo1 : kDrop), 0, 0);
        cur = nxt;
    }
}

template <int MODE, int AMP, int ROUNDS>
__global__ __launch_bounds__(512, 2) void k_r32x32(StftArgs a, unsigned per_xcd, unsigned total, unsigned slots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x;
    unsigned char *tabs = smem + kEx2;
    ((v4f *)(tabs + kWin2Off))[tid] = ((const v4f *)a.window)[tid];  // 2048 floats: (w[2n], w[2n+1]) / 2
    for (unsigned i = tid; i < 32u * 17u; i += 512u) ((v4f *)(tabs + kTw22Off))[i] = ((const v4f *)a.tw2)[i];
    unsigned *sched = (unsigned *)(tabs + kSch2Off);
    if constexpr (MODE == OUT_MEL)
        for (unsigned i = tid; i < a.mel_sched_words; i += 512u) sched[i] = a.mel_sched[i];

    // XCD x owns the contiguous run of tiles [x per_xcd, (x + 1) per_xcd); its `slots` resident workgroups walk it with that stride
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned lo = xcd * per_xcd, hi = min(lo + per_xcd, total);
    unsigned wid = lo + slot;

    const unsigned p1f = tid >> 5, n2 = tid & 31u;  // pass-1 identity
    const unsigned wave = tid >> 6, lane = tid & 63u, jq = lane >> 4, p2f = lane & 15u;
    const unsigned J = wave + 8u * jq;  // pass-2 job: even half of row J + odd half of row 32 - J (job 0: both halves of row 0)
    const unsigned rowE = J, rowO = (32u - J) & 31u;
    const bool j0 = J == 0u;
    const unsigned c1 = j0 ? 0u : J, c2 = j0 ? 32u : J + 512u;
    const float eps = (float)a.eps;
    constexpr unsigned ES = MODE == OUT_COMPLEX ? 8u : 4u;
    const unsigned step = 64u * a.n_frames * ES;  // uniform: 64 bins further
    const v4f *twj = (const v4f *)(tabs + kTw22Off) + J * 17u;
    v2f twa[4], twb[8];  // W_1024^(k1 n2) = twa[k1 >> 3] * twb[k1 & 7]
    {
        const v2f *t1 = (const v2f *)a.tw1 + n2;
#pragma unroll
        for (int q = 0; q < 4; ++q) twa[q] = t1[32 * 8 * q];
#pragma unroll
        for (int q = 0; q < 8; ++q) twb[q] = t1[32 * q];
    }

    constexpr int NCR = ROUNDS > 0 ? ROUNDS : 1;
    v4f creg[NCR];
    v2f xd[ROUNDS > 0 ? 1 : 32];
    const unsigned hop = a.hop;
    const unsigned row_bytes = (unsigned)a.n_samples * 4u;  // host: n_samples < 2^29
    auto load_tile = [&](unsigned w) {
        const unsigned b = w / a.tiles, f0 = (w - b * a.tiles) * 16u;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc((const float *)a.x + (size_t)b * a.sample_stride, row_bytes);
        // first sample of the tile relative to the row: negative in the left padding — as an unsigned byte offset far out of range, so
        // the hardware returns 0 there as it does past the end of the row (S1)
        const int tile_lo = (int)(f0 * hop) - (int)a.pad;
        if constexpr (ROUNDS > 0) {
            const int vo = (tile_lo + 4 * (int)tid) * 4;
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) creg[r] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rx, vo + r * 8192, 0, 0));
        } else {
            const int vo = ((int)(p1f * hop) + tile_lo + 2 * (int)n2) * 4;  // (even hop: a pair never straddles the row start)
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) xd[n1] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rx, vo + n1 * 256, 0, 0));
            if (hop & 1u) {  // uniform.  Odd frames sit on odd sample offsets: the pair (x[-1], x[0]) starts outside the row, and an 8-byte access whose
                             // first dword is out of range returns 0 for both: put x[0] back (as k_r32x16)
                const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, 0, 0, 0));
                const int s0 = (int)(p1f * hop) + tile_lo + 2 * (int)n2;
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1)
                    if (s0 + 64 * n1 == -1) xd[n1].y = x0;
            }
        }
    };
    if (wid < hi) load_tile(wid);
    __syncthreads();  // tables visible

    const unsigned xaddr = lds_addr2(smem) + p1f * hop * 4u + n2 * 8u;
    const unsigned waddr = lds_addr2(tabs + kWin2Off) + n2 * 8u;
    float *pwf = (float *)(smem + kPw2Off);

    while (wid < hi) {
        const unsigned b = wid / a.tiles, f0 = (wid - b * a.tiles) * 16u;
        const unsigned nf = min(16u, a.n_frames - f0);
        v2f xr[32];
        {
            v2f e[16], o[16], we[16], wo[16];
            if constexpr (ROUNDS > 0) {
#pragma unroll
                for (int r = 0; r < ROUNDS; ++r) *(v4f *)(smem + (r * 512u + tid) * 16u) = creg[r];
                __syncthreads();  // barrier 1: the staged samples are complete
                if (hop & 1u) {  // uniform
                    const unsigned base[8] = {xaddr, xaddr + 1024u, xaddr + 2048u, xaddr + 3072u, xaddr + 4096u, xaddr + 5120u, xaddr + 6144u, xaddr + 7168u};
                    read_cols2_odd<0>(e, we, base, waddr, std::make_integer_sequence<int, 16>{});
                    read_cols2_odd<1>(o, wo, base, waddr, std::make_integer_sequence<int, 16>{});
                } else {
                read_cols2<0>(e, we, xaddr, waddr, std::make_integer_sequence<int, 16>{});
                read_cols2<1>(o, wo, xaddr, waddr, std::make_integer_sequence<int, 16>{});
                }
                tie16x<15>(e);  // at most 15 of the 64 reads outstanding: the 32 of (e, we) have landed
                tie16x<-1>(we);
            } else {
                const v2f *w2 = (const v2f *)(tabs + kWin2Off) + n2;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    e[k] = xd[2 * k];
                    o[k] = xd[2 * k + 1];
                    we[k] = w2[64 * k];
                    wo[k] = w2[64 * k + 32];
                }
            }
            Fft<16, true>::run(e, we);
            if constexpr (ROUNDS > 0) {
                tie16x<0>(o);
                tie16x<-1>(wo);
            }
            Fft<16, true>::run(o, wo);
            Comb<32, 0, v2f>::run(xr, e, o);
        }
        // barrier 2: every wave has read its columns (and, filterbank outputs, finished the previous tile's band stage, whose |X|^2 tile the
        // upper half of ex overlays): pass 1 may write ex
        if constexpr (ROUNDS > 0 || MODE == OUT_MEL) __syncthreads();
        {
            unsigned char *dst = smem + p1f * kFS2 + n2 * 8u;
#pragma unroll
            for (int k1 = 0; k1 < 32; ++k1) {  // twiddle by W_1024^(k1 n2), write row k1 of this lane's column
                const int qa = k1 >> 3, qb = k1 & 7;
                v2f r = xr[k1];
                if (qb) r = cmulv(r, twb[qb]);
                if (qa) r = cmulv(r, twa[qa]);
                *(v2f *)(dst + k1 * 256) = r;
            }
        }
        const unsigned next = wid + slots;
        if (next < hi) load_tile(next);  // in flight during pass 2
        __syncthreads();  // barrier 3: ex complete
        // pass 2.  A lane whose frame does not exist (last tile of a signal) mirrors the tile's last frame: same values to the same
        // addresses, so every lane stores unconditionally and the compiler counts the stores behind the next tile's loads.
        const unsigned fe = min(p2f, nf - 1u);
        const unsigned char *exf = smem + fe * kFS2;
        v2f A[16], B[16];
        {
            v2f x[32];
            read_row32(exf + rowE * 256u, x);
            half_even(x, A);
            read_row32(exf + rowO * 256u, x);
            __syncthreads();  // barrier 4: ex consumed: the next staging (and the |X|^2 tile) may overwrite it
            half_odd(x, B);
        }
        const v2f a8 = A[8];
        if (j0) {
            // job 0 holds both halves of row 0: E = A (Z[64 m]) pairs with itself (m <-> 16 - m; m = 0: Z[0] with itself gives bins 0
            // and 1024), O = B (Z[32 + 64 m]) with itself (m <-> 15 - m).  Rearranged once, under a branch only its 16 lanes take, into the
            // general pairing below: first loop (A[i], B[15 - i]), second loop (A[8 + t], B[7 - t]).
            v2f nA[16], nB[16];
#pragma unroll
            for (int i = 0; i < 8; ++i) { nA[i] = A[i]; nA[8 + i] = B[i]; }
            nB[15] = A[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) nB[15 - i] = A[16 - i];
#pragma unroll
            for (int t = 0; t < 8; ++t) nB[7 - t] = B[15 - t];
#pragma unroll
            for (int i = 0; i < 16; ++i) { A[i] = nA[i]; B[i] = nB[i]; }
            asm volatile("" ::: "memory");  // keeps this a branch
        }
        const unsigned p2ofs = f0 + fe;
        const __amdgpu_buffer_rsrc_t ro = make_rsrc((unsigned char *)a.out + (size_t)b * 1025u * a.n_frames * ES, 1025u * a.n_frames * ES);
        if constexpr (MODE == OUT_MEL) {  // bins 1025..1035 are read with zero weights
            if (tid < 176u) pwf[pwt2_index(1025u + (tid >> 4), tid & 15u)] = 0.0f;
        }
        // rows c + 64 i upwards; the mirrored rows 1024 - c - 64 i count down: lane part 7 steps low, scalar part (7 - i) steps
        const unsigned oa1 = (c1 * a.n_frames + p2ofs) * ES, ob1 = ((1024u - 448u - c1) * a.n_frames + p2ofs) * ES;
        const unsigned oa2 = (c2 * a.n_frames + p2ofs) * ES, ob2 = ((1024u - 448u - c2) * a.n_frames + p2ofs) * ES;
        float *pw_c1 = pwf + pwt2_index(c1, p2f), *pw_m1 = pwf + pwt2_index(1024u - 448u - c1, p2f);
        float *pw_c2 = pwf + pwt2_index(c2, p2f), *pw_m2 = pwf + pwt2_index(1024u - 448u - c2, p2f);
        constexpr int PSTEP = 32 * 32;  // floats between bins k and k + 64 in the |X|^2 tile
        auto emit = [&](unsigned voff, unsigned soff, float *pwp, v2f X, bool conj) {
            if constexpr (MODE == OUT_COMPLEX) {
                const v2f V = conj ? (v2f){X.x, -X.y} : X;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, V), ro, (int)voff, (int)soff, 0);
            } else if constexpr (MODE == OUT_MEL) {
                *pwp = AMP == AMP_MAG_IN ? sqrtf(power2_of(X)) : power2_of(X);  // (a lane without a frame writes its mirror's values into its own slot: never stored)
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, amp2_f32<AMP>(power2_of(X), eps)), ro, (int)voff, (int)soff, 0);
            }
        };
        // pair (P, Q) = (Z[k], Z[1024 - k]): E = (P.x + Q.x, P.y - Q.y), D = (P.x - Q.x, P.y + Q.y), T = W' D with W' = -i W_2048^k:
        //   T = D.x W' + D.y W'^perp;  X[k] = E + T, X[1024 - k] = conj(E - T)   (window pre-halved: no 1/2)
        auto split = [&](v2f P, v2f Q, v4f w, v2f &X, v2f &Y) {
            const v2f E = pfma(Q, (v2f){1.f, -1.f}, P);
            const v2f D = pfma(Q, (v2f){-1.f, 1.f}, P);
            const v2f T = pfma(hi2(D), (v2f){w.z, w.w}, lo2(D) * (v2f){w.x, w.y});
            X = E + T;
            Y = E - T;
        };
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v2f X, Y;
            split(A[i], B[15 - i], twj[i], X, Y);
            emit(oa1, i * step, pw_c1 + i * PSTEP, X, false);
            emit(ob1, (7 - i) * step, pw_m1 + (7 - i) * PSTEP, Y, true);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            v2f X, Y;
            split(A[8 + t], B[7 - t], twj[8 + t], X, Y);
            emit(oa2, t * step, pw_c2 + t * PSTEP, X, false);
            emit(ob2, (7 - t) * step, pw_m2 + (7 - t) * PSTEP, Y, true);
        }
        if (j0) emit((512u * a.n_frames + p2ofs) * ES, 0u, pwf + pwt2_index(512u, p2f), a8 * (v2f){2.f, -2.f}, false);  // X[512] = 2 conj(Z[512])
        if constexpr (MODE == OUT_MEL) {
            __syncthreads();  // |X|^2 tile complete
            mel_tile_sched8<AMP>(a, pwf, sched, b, f0, nf, eps, tid);
        }
        wid = next;
    }
}

template <int MODE, int AMP>
hipError_t launch_variant2(const StftArgs &a, hipStream_t s) {
    const unsigned total = a.tiles * a.batch;
    const unsigned per_xcd = (total + 7u) / 8u;
    const unsigned cu_slots = std::max(1u, device_cu_count() / 8u);
    const unsigned nslots = per_xcd < cu_slots ? per_xcd : cu_slots;  // one 512-thread workgroup per CU
    const unsigned chunks = (15u * a.hop + 2048u + 3u) >> 2;
    const unsigned lds = (unsigned)kLds2Base + (MODE == OUT_MEL ? ((a.mel_sched_words * 4u + 15u) & ~15u) + 64u : 0u);
    auto go = [&](auto kernel) -> hipError_t {
        hipError_t e = set_max_dynamic_lds((const void *)kernel, kLds2Max);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3(nslots * 8), dim3(512), lds, s, a, per_xcd, total, nslots);
        return hipGetLastError();
    };
    if (chunks <= 5u * 512u) return go(k_r32x32<MODE, AMP, 5>);
    return go(k_r32x32<MODE, AMP, 0>);
}

}  // namespace

bool plan_geometry_r32x32_f32(StftArgs &a) {
    if (a.n_fft != 2048) return false;  // (any hop since round 5: odd ones read their staged pairs with ds_read2_b32 / patch the row-start pair)
    if (a.x != nullptr && a.n_frames < 5u) return false;  // batches of very short signals: mostly empty 16-frame tiles (16 384 x 5 frames: even with k_reg_radix)
    if (a.n_samples >= (1ull << 29)) return false;                                        // 32-bit byte offsets into a sample row
    if ((unsigned long long)a.n_frames * 1025ull * 8ull >= 0x7fffffffull) return false;  // and into one output signal
    // filterbank outputs need the band schedule (built on the host before this is asked; a bank without one takes the register-tiled kernel)
    if (a.out_mode == OUT_MEL && (a.mel_sched_words == 0 || a.mel_sched_words > (unsigned)kSch2MaxWords)) return false;
    a.ft = 16;
    return true;
}

hipError_t launch_r32x32_f32(const StftArgs &a, hipStream_t s) {
    const unsigned long long total64 = (unsigned long long)a.tiles * a.batch;
    if (total64 == 0 || total64 >= 0x7ffffff0ull) return hipErrorInvalidConfiguration;
    if (a.out_mode == OUT_COMPLEX) return launch_variant2<OUT_COMPLEX, AMP_POWER>(a, s);
    if (a.out_mode == OUT_MEL) {
        if (a.amp == AMP_MAGNITUDE) return launch_variant2<OUT_MEL, AMP_MAGNITUDE>(a, s);
        if (a.amp == AMP_DB) return launch_variant2<OUT_MEL, AMP_DB>(a, s);
        if (a.amp == AMP_MAG_IN) return launch_variant2<OUT_MEL, AMP_MAG_IN>(a, s);
        return launch_variant2<OUT_MEL, AMP_POWER>(a, s);
    }
    if (a.amp == AMP_MAGNITUDE) return launch_variant2<OUT_LINEAR, AMP_MAGNITUDE>(a, s);
    if (a.amp == AMP_DB) return launch_variant2<OUT_LINEAR, AMP_DB>(a, s);
    return launch_variant2<OUT_LINEAR, AMP_POWER>(a, s);
}

}  // namespace sgx
