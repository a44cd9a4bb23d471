// kernels_r32x16.hip — tuned f32, n_fft = 1024 STFT kernel for gfx950 (the BASELINE shape).
//
// Transform structure; a tile = 16 consecutive frames of one signal:
//
//   pass 1  lane (f = 0..15, n2 = 0..15) owns z[16*n1 + n2], n1 = 0..31, of frame f, where
//           z[n] = w[2n] x[2n] + i w[2n+1] x[2n+1] is the half-length complex sequence of the real frame
//           (window pre-scaled by 1/2 on the host — exact — so the real split needs no halving).
//           Window multiply fused into the first butterflies; one 32-point FFT entirely in registers; twiddle by
//           W_512^(k1*n2) from two short per-lane register tables; one ds_write_b64 per value.
//   LDS     ex[f][k1][n2] complex f32, frame stride 4096+16 B.  This is the ONLY exchange of the transform.
//   pass 2  lane (jq, f) owns "job" j of frame f: rows k1 = j and 32-j (job 0: rows 0 and 16).  8 + 8
//           ds_read_b128 (conflict-free by the frame-stride / job-to-wave choice), two 16-point FFTs in registers
//           -> Z[j+32*k2], Z[32-j+32*k2], and — because a job holds both members of every (k, 512-k) pair — the
//           real split X[k] = E + W_1024^k O entirely in registers.
//   store   the 16 lanes of a job hold the same bin of 16 consecutive frames, so out[b][k][f0..f0+15] is one
//           contiguous 64-byte segment: the frame-contiguous layout of the reference (S9) needs no LDS
//           transpose.  Filterbank outputs: |X|^2 goes to LDS (overlaying the free ex buffer) and is reduced per
//           (band, frame) in ascending-bin order (spectrogram.rs:102-117), then the dB / sqrt epilogue.
//
// One persistent 512-thread workgroup per CU whose two 256-thread halves each own a tile and an exchange buffer and
// move through the phases in lockstep.  All complex arithmetic is written on 2-float vectors so it compiles to
// packed-f32 VALU (v_pk_add/mul/fma_f32 with op_sel / neg modifiers).
//
// Round-2 changes (each measured on MI355X):
//   * global memory through buffer instructions with a wave-uniform descriptor: the sample rows are bounds-checked by the
//     hardware (out-of-range dwords read as 0 = the reference's zero centre padding, spectrogram.rs:1301-1320), so the edge
//     tiles need no compares and the rows no alignment; the output offsets are a per-lane constant + a scalar offset.
//   * the column / window reads of pass 1 are issued as single ds_read_b64 (inline asm): the compiler fused them into
//     ds_read2_b64, which the LDS serves at half rate (MI355X_MICROARCH.md §LDS).
//   * the real-split twiddles are read as 16-byte (W', W'^perp) pairs in consumption order, so a pair costs six packed
//     instructions and nothing else; the job-0 special case is a rearrangement under a branch only wave 0 takes.
//   * the band reduction works on the transposed |X|^2 tile pw[k][f] with two frames per lane (packed, un-fused) along a
//     host-built schedule that balances the bands over the waves.
//
// Reference semantics implemented: spectrogram.rs:1301-1334 (framing, window, R2C, |.|^2), :1845-1865,
// :2068-2080; replaces the per-frame `R2cPlan::process` call at :1323 (fft_backend.rs:423-431).
#include <type_traits>
#include <utility>

#include "buffer_ops.h"
#include "fft_inreg.h"
#include "r32x16_layout.h"
#include "sgx_internal.h"

namespace sgx {
namespace {

using namespace inreg;
using namespace r32x16;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

// 10 log10(x) as (10 log10 2) log2(x): the hardware's log2 (1 ulp, with the compiler's scaling for subnormal inputs) and one
// multiply — 6 instructions against ~20 for log10f's extra-precision product; ~3 ulp of the logarithm, i.e. < 2e-5 dB at
// |dB| <= 100, far inside the 1e-3 dB the parity tests allow on top of the f32 spectrum's own error
__device__ __forceinline__ float db_f32(float x) { return __builtin_log2f(x) * 3.01029995663981195f; }

template <int AMP>
__device__ __forceinline__ float amp_f32(float p, float eps) {
    if constexpr (AMP == AMP_MAGNITUDE) return sqrtf(p);
    else if constexpr (AMP == AMP_DB) return db_f32(fmaxf(p, eps));
    else return p;
}
__device__ __forceinline__ float power_of(v2f x) { return __builtin_fmaf(x.x, x.x, x.y * x.y); }

__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(size_t)p; }  // low half of a flat LDS address

#ifdef SGX_STAMPS  // diagnostic build only (tools/stamps.py): a wave's cycles per phase
__device__ unsigned long long g_stamps[32];
#define SGX_STAMP(i)                                                                        \
    do {                                                                                    \
        unsigned long long t_;                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        st_acc[i] += t_ - st_prev;                                                          \
        st_prev = t_;                                                                       \
    } while (0)
#define SGX_STAMP_PARAMS , unsigned long long *st_acc, unsigned long long &st_prev
#define SGX_STAMP_ARGS , st_acc, st_prev
#else
#define SGX_STAMP(i)
#define SGX_STAMP_PARAMS
#define SGX_STAMP_ARGS
#endif

#ifndef SGX_BANDSEG
#define SGX_BANDSEG 0  // band stage: all segment records up front, empty segments skipped
#endif
#ifndef SGX_BANDTRIP
#define SGX_BANDTRIP 0  // band stage in trips of 16 steps with their reads requested together
#endif
#ifndef SGX_BANDFMA
#define SGX_BANDFMA 0  // experiment: band sums as four interleaved fused partial sums (not the reference order)
#endif
#ifndef SGX_BANDPF
#define SGX_BANDPF 0  // 1: the band reduction of the n_fft 1024 kernel reads one 8-step group ahead (plan.hip pads L to multiples of 8)
#endif
#ifndef SGX_ODDHOP
#define SGX_ODDHOP 1  // odd hops at n_fft 1024 run here too: sample pairs of odd frames by ds_read2_b32 (staged) / with the row-start pair patched (direct); 0: register-tiled kernel
#endif
#ifndef SGX_DMA
#define SGX_DMA 0  // 1: filterbank outputs at n_fft 1024: the next tile's samples go HBM -> LDS directly (buffer_load ... lds), issued behind barrier 4
#endif
#ifndef SGX_ONEPATH
#define SGX_ONEPATH 1  // 0: the interior / edge split and the per-round chunk predicates of the sample loads (A/B only)
#endif

// ---- single-issue LDS reads -------------------------------------------------------------------------------------------
// hipcc fuses neighbouring 8-byte LDS reads into ds_read2_b64, which the LDS serves at 128 B/clk; a plain ds_read_b64 gets
// 256 B/clk.  These reads are issued from inline asm (the compiler neither fuses nor counts them) and collected by an asm
// s_waitcnt that names the destinations, so no consumer can be scheduled above it.
template <int OFF>
__device__ __forceinline__ void ds_read64(v2f &d, unsigned addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void ds_read128(v4f &d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void tie8(v2f *d) {  // N < 0: no instruction, ordering only ("+v" counts twice towards the 30-operand limit)
    if constexpr (N >= 0)
        asm volatile("s_waitcnt lgkmcnt(%8)"
                     : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
                     : "n"(N));
    else
        asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
}
template <int N>
__device__ __forceinline__ void tie16(v2f (&d)[16]) {
    tie8<N>(&d[0]);
    tie8<-1>(&d[8]);
}
// column n1 of the staged tile sits at byte offset 128 n1 (+ 128 per 8 columns when the staging buffer is padded)
template <bool XSPAD, int N1>
constexpr int col_off() { return N1 * 128 + (XSPAD ? (N1 >> 3) * 128 : 0); }
template <bool XSPAD, int PAR, int... K>
__device__ __forceinline__ void read_cols(v2f (&x)[16], v2f (&w)[16], unsigned xaddr, unsigned waddr, std::integer_sequence<int, K...>) {
    ((ds_read64<col_off<XSPAD, 2 * K + PAR>()>(x[K], xaddr), ds_read64<(2 * K + PAR) * 128>(w[K], waddr)), ...);
}

// P512: z[n] = (a[n], b[n]) with n = 16 n1 + n2 — sample n of frame 2 p and of frame 2 p + 1, one hop (4 HOP bytes) apart — as ONE
// ds_read2_b32 into the register pair.  The staging buffer carries 64 B of padding per slot stride (two hops = 8 HOP bytes), so
// the byte offset of sample s of a slot is 4 s + 64 per slot stride crossed (a slot starts right behind a pad); a base address
// per four n1 keeps both 8-bit dword offsets in range (hop <= 160).
template <int HOP>
constexpr int p512_off_a(int n1) { return 64 * n1 + (64 * n1 / (8 * HOP)) * 64; }
template <int HOP>
constexpr int p512_off_b(int n1) { return 64 * n1 + 4 * HOP + ((64 * n1 + 4 * HOP) / (8 * HOP)) * 64; }
template <int O0, int O1>
__device__ __forceinline__ void ds_read2x32(v2f &d, unsigned addr) {
    static_assert(O0 >= 0 && O0 < 256 && O1 >= 0 && O1 < 256, "ds_read2_b32: 8-bit dword offsets");
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(d) : "v"(addr), "n"(O0), "n"(O1));
}
template <int HOP, int PAR, int... K>
__device__ __forceinline__ void read_cols512(v2f (&x)[16], v2f (&w)[16], const unsigned (&base)[8], unsigned waddr, std::integer_sequence<int, K...>) {
    ((ds_read2x32<(p512_off_a<HOP>(2 * K + PAR) - p512_off_a<HOP>((2 * K + PAR) & ~3)) / 4, (p512_off_b<HOP>(2 * K + PAR) - p512_off_a<HOP>((2 * K + PAR) & ~3)) / 4>(
          x[K], base[(2 * K + PAR) >> 2]),
      ds_read64<(2 * K + PAR) * 128>(w[K], waddr)),
     ...);
}

// Odd hops: odd frames start on an odd sample, i.e. their pairs sit at 4-byte-aligned LDS addresses, where a ds_read_b64 takes
// several passes (measured: the whole kernel at half speed).  ds_read2_b32 reads the same pair as two dwords at half the
// ds_read_b64 rate and has no alignment to respect; one base register per 8 columns keeps its 8-bit dword offsets in range.
template <int PAR, int... K>
__device__ __forceinline__ void read_cols_odd(v2f (&x)[16], v2f (&w)[16], const unsigned (&base)[4], unsigned waddr, std::integer_sequence<int, K...>) {
    ((ds_read2x32<((2 * K + PAR) & 7) * 32, ((2 * K + PAR) & 7) * 32 + 1>(x[K], base[(2 * K + PAR) >> 3]), ds_read64<(2 * K + PAR) * 128>(w[K], waddr)), ...);
}

template <int PAR, int... K>
__device__ __forceinline__ void read_win(v2f (&w)[16], unsigned waddr, std::integer_sequence<int, K...>) {
    (ds_read64<(2 * K + PAR) * 128>(w[K], waddr), ...);
}

// LDS-DMA: 16 bytes per lane from the buffer (range-checked per dword: out-of-range dwords arrive as 0) straight into LDS at
// lds_base + 16 * lane.  Issued from inline asm — for a compiler-visible LDS-DMA hipcc drains vmcnt(0) before the next LDS read
// that may alias it and at every __syncthreads() (cdna_hip_programming.md §5, Pipelining across barriers), which would expose
// the whole HBM latency right behind the request; completion is counted by hand (s_waitcnt vmcnt) where the samples are read.
template <int OFF>
__device__ __forceinline__ void dma16_to_lds(__amdgpu_buffer_rsrc_t r, int voff, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:%4 lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(r), "s"(lds_base), "n"(OFF)
                 : "memory");
}
template <bool XSPAD, int R, int ROUNDS>
__device__ __forceinline__ void dma_rounds(__amdgpu_buffer_rsrc_t r, int voff, unsigned xs_base, unsigned wave) {
    if constexpr (R < ROUNDS) {
        const unsigned c0 = R * 256u + wave * 64u;  // the wave's first chunk of this round
        dma16_to_lds<0>(r, voff + R * 4096, xs_base + c0 * 16u + (XSPAD ? (c0 >> 6) * 128u : 0u));
        dma_rounds<XSPAD, R + 1, ROUNDS>(r, voff, xs_base, wave);
    }
}

// per-lane pass-1 twiddle tables: W_512^(k1*n2) = twa[k1>>3] * twb[k1&7]
__device__ __forceinline__ void load_tw1(const StftArgs &a, unsigned n2, v2f (&twa)[4], v2f (&twb)[8]) {
    const v2f *t1 = (const v2f *)a.tw1 + n2;
#pragma unroll
    for (int q = 0; q < 4; ++q) twa[q] = t1[16 * 8 * q];
#pragma unroll
    for (int q = 0; q < 8; ++q) twb[q] = t1[16 * q];
}

// twiddle by W_512^(k1 n2) and write row k1 of this lane's column to ex
__device__ __forceinline__ void twiddle_store(v2f (&xr)[32], const v2f (&twa)[4], const v2f (&twb)[8], unsigned char *dst) {
#pragma unroll
    for (int k1 = 0; k1 < 32; ++k1) {
        const int qa = k1 >> 3, qb = k1 & 7;
        v2f r = xr[k1];
        if (qb) r = cmulv(r, twb[qb]);
        if (qa) r = cmulv(r, twa[qa]);
#if defined(SGX_ABL_NOEXW)  // timing experiments only (wrong results): the exchange not written at all — the upper bound of any cheaper exchange write
        asm volatile("" ::"v"(r), "v"(dst));
#elif defined(SGX_ABL_ADDTID)  // ... or written as two ds_write_addtid_b32 per value (planar; 128 B/clk against ds_write_b64's ~85), M0 = the wave's 16 KiB
        asm volatile("ds_write_addtid_b32 %0 offset:%2\n\tds_write_addtid_b32 %1 offset:%3" ::"v"(r.x), "v"(r.y), "n"(k1 * 512), "n"(k1 * 512 + 256) : "memory");
#else
        *(v2f *)(dst + k1 * 128) = r;
#endif
    }
}

__device__ __forceinline__ void read_rows(const unsigned char *exf, unsigned ra, unsigned rb, v2f (&A)[16], v2f (&B)[16]) {
    const v4f *pa = (const v4f *)(exf + ra * 128);
    const v4f *pb = (const v4f *)(exf + rb * 128);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const v4f q = pa[c];
        A[2 * c] = (v2f){q.x, q.y};
        A[2 * c + 1] = (v2f){q.z, q.w};
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const v4f q = pb[c];
        B[2 * c] = (v2f){q.x, q.y};
        B[2 * c + 1] = (v2f){q.z, q.w};
    }
}

// Lane-constant output addressing of a job.  A job's 33 bins are k = c1 + 32 i and c2 + 32 t (i, t < 8), their mirrors
// 512 - k, and (job 0) bin 256.  Byte offset of a value = lane part (below, + the tile's frame offset) + a scalar multiple of
// `step` = 32 rows; the mirrored rows count down, so their lane part starts 7 steps low and the scalar part is (7 - i) step.
struct JobOfs {
    unsigned a1, b1, a2, b2, mid;  // rows c1, 512 - c1 - 224, c2, 512 - c2 - 224, 256 — in elements (row * n_frames)
};
__device__ __forceinline__ JobOfs job_offsets(unsigned j, unsigned n_frames) {
    const unsigned c1 = j == 0 ? 16u : j, c2 = j == 0 ? 0u : j + 256u;
    return JobOfs{c1 * n_frames, (512u - 224u - c1) * n_frames, c2 * n_frames, (512u - 224u - c2) * n_frames, 256u * n_frames};
}

// pass 2 arithmetic + real split + output of one lane (job j of one frame).
//   tw   : this job's 16 split twiddles in LDS, 16 bytes each = (W', W'^perp) with W' = -i W_1024^k (see plan.hip)
//   out  : linear / complex: buffer descriptor + this lane's five byte offsets (frame included) + the scalar row step
//   pw   : filterbank modes: this lane's |X|^2 slot bases (see PwAddr)
struct PwAddr {
    float *up, *down;  // up[i * pstep] = bin c + 32 i; down[(7 - i) * pstep] = its mirror (both for c1 and, + c2off, for c2)
};
template <int MODE, int AMP, bool PWT>
__device__ __forceinline__ void pass2_compute(v2f (&A)[16], v2f (&B)[16], bool j0, float eps, const v4f *tw, __amdgpu_buffer_rsrc_t ro,
                                              unsigned oa1, unsigned ob1, unsigned oa2, unsigned ob2, unsigned omid, unsigned step,
                                              float *pw_c1, float *pw_m1, float *pw_c2, float *pw_m2, float *pw_mid SGX_STAMP_PARAMS) {
    // Job 0 owns the two self-paired rows 0 and 16.  Its transformed rows rearranged once, under a branch only its lanes take, it runs the general
    // pairing below: first loop (A[i], B[15-i]), second loop (A[8+t], B[7-t]).
    //   first  loop wants (B[i], B[15-i])            -> A'[i] = B[i], B'[8..15] unchanged
    //   second loop wants (A[t], A[(16-t) & 15])     -> A'[8+t] = A[t], B'[7-t] = A[(16-t) & 15]
    Fft<16, false>::run(A, A);
    Fft<16, false>::run(B, B);
    const v2f a8 = A[8];
    if (j0) {
        v2f nA[16], nB[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { nA[i] = B[i]; nA[8 + i] = A[i]; }
        nB[7] = A[0];
#pragma unroll
        for (int t = 1; t < 8; ++t) nB[7 - t] = A[16 - t];
#pragma unroll
        for (int i = 0; i < 16; ++i) A[i] = nA[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) B[i] = nB[i];
        asm volatile("" ::: "memory");  // keeps this a branch (the compiler turned the selects into 48 v_cndmask per tile)
    }
    SGX_STAMP(9);  // 16-point transforms
    constexpr int PSTEP = PWT ? 32 * 16 : 32;  // floats between bins k and k + 32 in the |X|^2 tile
    auto emit = [&](unsigned voff, unsigned soff, float *pwp, v2f X, bool conj) {
#ifdef SGX_ABL_NOSTORE  // timing experiment only: keep the value alive, drop the store
        asm volatile("" ::"v"(X), "v"(voff));
        return;
#endif
        if constexpr (MODE == OUT_COMPLEX) {
            const v2f V = conj ? (v2f){X.x, -X.y} : X;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, V), ro, (int)voff, (int)soff, 0);
        } else if constexpr (MODE == OUT_MEL) {
#ifdef SGX_ABL_NOPW
            asm volatile("" ::"v"(X), "v"(pwp));
            return;
#endif
            *pwp = AMP == AMP_MAG_IN ? sqrtf(power_of(X)) : power_of(X);
        } else {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, amp_f32<AMP>(power_of(X), eps)), ro, (int)voff, (int)soff, 0);
        }
    };
    // pair (P, Q) = (Z[k], Z[512-k]):  E = (P.x+Q.x, P.y-Q.y), D = (P.x-Q.x, P.y+Q.y), O = -i D, T = W O = W' D with W' = -i W:
    //   T = D.x W' + D.y W'^perp;  X[k] = E + T, X[512-k] = conj(E - T)
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
        split(A[i], B[15 - i], tw[i], X, Y);
        emit(oa1, i * step, pw_c1 + i * PSTEP, X, false);
        emit(ob1, (7 - i) * step, pw_m1 + (7 - i) * PSTEP, Y, true);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        v2f X, Y;
        split(A[8 + t], B[7 - t], tw[8 + t], X, Y);
        emit(oa2, t * step, pw_c2 + t * PSTEP, X, false);
        emit(ob2, (7 - t) * step, pw_m2 + (7 - t) * PSTEP, Y, true);
    }
    if (j0) emit(omid, 0u, pw_mid, a8 * (v2f){2.f, -2.f}, false);  // X[256] = 2 conj(Z[256]) (row 0, k2 = 8)
    SGX_STAMP(10);  // real split (+ direct stores / LDS writes)
}

// n_fft = 512, two frames per transform (P512): the slot's 512-point complex sequence is z[n] = a[n] + i b[n] of two consecutive
// frames a, b (windowed, pre-halved), so the pair (P, Q) = (Z[k], Z[512-k]) gives both frames' bin k without a twiddle:
//   A[k] = P + conj Q = E,   B[k] = -i (P - conj Q) = (D.y, -D.x)
// — one 8-byte (power / magnitude / dB) or 16-byte (complex) store per pair: bin k of frames 2 p and 2 p + 1, which are adjacent
// in memory.  Pairs with k > 256 (the second loop of every job but job 0) are the conjugates of bin 512 - k.  `vfull` is the
// lane's byte offset when both frames exist, `vhalf` when only the first does (the other being out of range: dropped).
// Filterbank outputs: the pair's two powers go to the |X|^2 tile as one 8-byte LDS write (pw1 / pw2 / pwm: this lane's slots of the
// rows the two loops and bin 256 start at; 32 bins further = kP512Step floats, see pwt512_index).
constexpr unsigned kP512Step = 16u * 64u;
// PACK (tiles that continue into the next signal): the slot's two frames need not be neighbours in memory — `vfull` is then the first
// frame's own offset and `vhalf` the second's (either may be out of range: dropped), two element-sized stores per pair.
template <int MODE, int AMP, bool PACK = false>
__device__ __forceinline__ void pass2_pair512(v2f (&A)[16], v2f (&B)[16], bool j0, float eps, __amdgpu_buffer_rsrc_t ro, unsigned vfull1,
                                              unsigned vhalf1, unsigned vfull2, unsigned vhalf2, unsigned vfullm, unsigned vhalfm, unsigned step,
                                              float *pw1, float *pw2, float *pwm) {
    Fft<16, false>::run(A, A);
    Fft<16, false>::run(B, B);
    const v2f a8 = A[8];
    if (j0) {  // as pass2_compute: job 0's two self-paired rows rearranged into the general pairing
        v2f nA[16], nB[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { nA[i] = B[i]; nA[8 + i] = A[i]; }
        nB[7] = A[0];
#pragma unroll
        for (int t = 1; t < 8; ++t) nB[7 - t] = A[16 - t];
#pragma unroll
        for (int i = 0; i < 16; ++i) A[i] = nA[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) B[i] = nB[i];
        asm volatile("" ::: "memory");
    }
    auto emit = [&](unsigned vfull, unsigned vhalf, int soff, v2f Xa, v2f Xb, bool conj, float *pwp) {
        if constexpr (MODE == OUT_MEL) {
            const float pa = power_of(Xa), pb = power_of(Xb);
            *(v2f *)pwp = AMP == AMP_MAG_IN ? (v2f){sqrtf(pa), sqrtf(pb)} : (v2f){pa, pb};
        } else if constexpr (MODE == OUT_COMPLEX) {
            const float sg = conj ? -1.f : 1.f;
            const v4f V = (v4f){Xa.x, sg * Xa.y, Xb.x, sg * Xb.y};
            if constexpr (PACK) {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, (v2f){V.x, V.y}), ro, (int)vfull, soff, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, (v2f){V.z, V.w}), ro, (int)vhalf, soff, 0);
            } else {
                // (16-byte store: the whole offset in the lane register — with a scalar-register soffset the compiler lets the next instructions
                // rewrite the data registers at once, which cost k_d32x16 0.06 % of its complex outputs: kernels_d32x16.hip emit)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, V), ro, (int)vfull + soff, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, (v2f){V.x, V.y}), ro, (int)vhalf, soff, 0);
            }
        } else {
            const v2f V = (v2f){amp_f32<AMP>(power_of(Xa), eps), amp_f32<AMP>(power_of(Xb), eps)};
            if constexpr (PACK) {
                // (scalars first: __builtin_bit_cast of a vector COMPONENT other than .x reads the vector's first element — hipcc 7.2)
                const float va = V.x, vb = V.y;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, va), ro, (int)vfull, soff, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, vb), ro, (int)vhalf, soff, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, V), ro, (int)vfull, soff, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, V.x), ro, (int)vhalf, soff, 0);
            }
        }
    };
    auto pair = [&](v2f P, v2f Q, v2f &Xa, v2f &Xb) {
        Xa = pfma(Q, (v2f){1.f, -1.f}, P);                 // E
        const v2f D = pfma(Q, (v2f){-1.f, 1.f}, P);
        Xb = (v2f){D.y, -D.x};
    };
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // k = c1 + 32 i < 256: rows ascending
        v2f Xa, Xb;
        pair(A[i], B[15 - i], Xa, Xb);
        emit(vfull1, vhalf1, i * (int)step, Xa, Xb, false, pw1 + i * kP512Step);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {  // job 0: k = 32 t (rows ascending); other jobs: k = j + 256 + 32 t -> row 256 - j - 32 t, conjugated
        v2f Xa, Xb;
        pair(A[8 + t], B[7 - t], Xa, Xb);
        if (j0) emit(vfull2 + t * step, vhalf2 + t * step, 0, Xa, Xb, false, pw2 + t * kP512Step);
        else emit(vfull2 + (7 - t) * step, vhalf2 + (7 - t) * step, 0, Xa, Xb, true, pw2 + (7 - t) * kP512Step);
    }
    if (j0) emit(vfullm, vhalfm, 0, (v2f){2.f * a8.x, 0.f}, (v2f){2.f * a8.y, 0.f}, false, pwm);  // bin 256: Z[256] pairs with itself
}

// ---- filterbank stage, generic forms (|X|^2 tile stored pw[f][k], kPS floats per frame) -----------------------------------
// CSR from global memory: banks whose rows are not runs of consecutive bins, or too large for the LDS schedule
template <int AMP>
__device__ __forceinline__ void mel_tile_csr(const StftArgs &a, const float *pwall, unsigned b, unsigned f0, unsigned nf,
                                             float eps, unsigned t, unsigned nthreads) {
    const float *val = (const float *)a.mel_val;
    float *o = (float *)a.out + ((size_t)b * a.n_out) * a.n_frames + f0;
    for (unsigned idx = t; idx < 16u * a.n_mels; idx += nthreads) {
        const unsigned ff = idx & 15u, mm = idx >> 4;
        float acc = 0.0f;
        const unsigned i0 = a.mel_ptr[mm], i1 = a.mel_ptr[mm + 1];
        for (unsigned i = i0; i < i1; ++i) acc = __fadd_rn(__fmul_rn(val[i], pwall[ff * kPS + a.mel_col[i]]), acc);
        if (ff < nf) o[mm * a.n_frames + ff] = amp_f32<AMP>(acc, eps);
    }
}

// Bank rows with wide supports (the dense ERB / gammatone bank of src/erb.rs:374-401, very coarse Mel banks) on the matrix
// cores.  Per tile and 16-row block the product is [16 rows x K] x [K x 16 frames] with K = the block's own bin range;
// v_mfma_f32_16x16x4_f32 is an exact-f32 fmaf chain at the packed VALU rate that does the operand broadcast a per-lane
// loop cannot.  A wave owns whole blocks (host-balanced); lane (i = l & 15, q = l >> 4) feeds A = weight[16 blk + i][k] and
// B = pw[frame i][k] with k = lo + 16 c + 4 q + s for step s of chunk c — the k order inside a chunk is permuted
// identically on both operands, so each lane fetches its 4 steps with ONE 16-byte load (weights: fragment-ordered table,
// 1 KiB per wave-load, L2 resident, prefetched 4 chunks ahead) and ONE ds_read_b128 (pw rows are 516 floats: the 16 lanes
// of a q group cover all 64 banks).  D[4 q + r][frame i] lands frame-contiguous across lanes: 64-byte stores.
// Two accumulator chains hide the 40-cycle dependent latency.  pw[f][513..515] are zero (the cover is a multiple of 4).
typedef float v4acc __attribute__((ext_vector_type(4)));
template <int AMP>
__device__ __forceinline__ void map_tile_mfma(const StftArgs &a, const float *pwall, unsigned b, unsigned f0, unsigned nf,
                                              float eps, unsigned t, unsigned rot) {
    const unsigned wave = ((t >> 6) + rot) & 3u, lane = t & 63u, fi = lane & 15u, q = lane >> 4;
    float *o = (float *)a.out + ((size_t)b * a.n_out) * a.n_frames + f0 + fi;
    for (unsigned blk = 0; blk < a.mm_nblk; ++blk) {
        const uint4 d = a.mm_blk[blk];  // uniform: scalar loads
        if (((d.w >> 8) & 3u) != wave) continue;
        const unsigned lo = d.y, n16 = d.z, n4 = d.w & 3u;
        const v4f *wf = (const v4f *)a.mm_frag + (size_t)d.x * 64u + lane;
        const float *prow = pwall + fi * kPS + lo;
        const v4f *pp = (const v4f *)(prow + 4u * q);
        v4acc acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        // fragment ring, 4 deep.  The table is padded by 4 fragments, so the prefetch never needs a guard: guards would put
        // the loads behind scalar branches and force a full vmcnt(0) wait per chunk.
        v4f wq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) wq[u] = wf[u * 64];
        const unsigned nmain = n16 & ~3u;
        for (unsigned c0 = 0; c0 < nmain; c0 += 4u) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned c = c0 + u;
                const v4f w = wq[u], p = pp[c * 4u];
                wq[u] = wf[(c + 4u) * 64u];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, p.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, p.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, p.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, p.w, acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {  // n16 % 4 remaining chunks: their fragments are already in the ring
            if (nmain + u < n16) {
                const v4f w = wq[u], p = pp[(nmain + u) * 4u];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, p.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, p.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, p.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, p.w, acc1, 0, 0, 0);
            }
        }
        if (n4) {  // trailing 4-wide steps: k = lo + 16 n16 + 4 s + q
            const unsigned u = n16 & 3u;
            const v4f w = u == 0 ? wq[0] : u == 1 ? wq[1] : u == 2 ? wq[2] : wq[3];
            const float *pt = prow + 16u * n16 + q;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, pt[0], acc0, 0, 0, 0);
            if (n4 > 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, pt[4], acc1, 0, 0, 0);
            if (n4 > 2) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, pt[8], acc0, 0, 0, 0);
        }
        const v4acc acc = acc0 + acc1;
        if (fi < nf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned m = 16u * blk + 4u * q + r;
                if (m < a.n_mels) o[(size_t)m * a.n_frames] = amp_f32<AMP>(acc[r], eps);
            }
        }
    }
}

// ---- filterbank stage on the schedule (banks whose rows are runs of consecutive bins: Mel, log-Hz) ------------------------
// |X|^2 tile transposed and paired: bins (2m, 2m+1) x frames (2p, 2p+1) are one 16-byte group, 8 groups = 128 bytes per bin
// pair (pwt_index below).  Lane (slot = lane >> 3, fp = lane & 7) of wave w reduces frames (2 fp, 2 fp + 1) of the band the
// schedule gives (segment, w, slot): one ds_read_b128 serves two bins of both frames (16-byte reads keep the LDS rate at two
// waves per SIMD, 8-byte reads do not), the weights come four bins at a time, and the two running sums are one packed
// multiply and one packed add per bin — un-fused and in ascending-bin order exactly like SparseMatrix::multiply_vec
// (spectrogram.rs:102-117).  Padding steps carry weight +0: 0 * p = +0 added to a non-negative partial sum is exact.
// The host starts slots 0, 1 at a bin = 0 mod 4 and slots 2, 3 at a bin = 2 mod 4, so the slots of a read group fall on different banks.
__host__ __device__ constexpr unsigned pwt_index(unsigned k, unsigned f) { return (k >> 1) * 32u + (f >> 1) * 4u + (k & 1u) * 2u + (f & 1u); }
__device__ __forceinline__ v2f mul_add_unfused(float w, v2f p, v2f acc) {
#pragma clang fp contract(off)
    const v2f m = (v2f){w, w} * p;
    return m + acc;
}
// TOLDS (fused MFCC epilogue; = the row length RL of mfcc_tile): the band's two dB values go to the LDS tile mq[band & 3][frame][band >> 2]
// instead of the output tensor.
template <int AMP, bool PACK = false, int TOLDS = 0>
__device__ __forceinline__ void mel_tile_sched(const StftArgs &a, const float *pwT, const unsigned *sched, unsigned b, unsigned f0,
                                               unsigned nf, float eps, unsigned tid, float *mfl SGX_STAMP_PARAMS) {
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u, slot = lane >> 3, fp = lane & 7u;
    // out[b][band][f0 + 2 fp ..]: one descriptor per tile; a lane without a frame or a slot without a band gets an offset past
    // its range and the hardware drops the store.  The stores are unconditional and the segment loop has a fixed trip count so
    // that the compiler can count them behind the next tile's sample loads (vmcnt(2 kSchedSegs), not vmcnt(0): the loop must
    // never wait for its own stores).
    constexpr unsigned kDrop = 0x80000000u;
    unsigned obytes, fo0, fo1;
    const float *obase;
    if constexpr (PACK) {  // the tile's slots run on into the following signals: a slot's offset counts from signal b
        obase = (const float *)a.out + (size_t)b * a.n_out * a.n_frames;
        obytes = min(17u, a.batch - b) * a.n_out * a.n_frames * 4u;
        const unsigned fl0 = f0 + 2u * fp, fl1 = fl0 + 1u, b0 = fl0 / a.n_frames, b1 = fl1 / a.n_frames;
        fo0 = 2u * fp < nf ? (b0 * a.n_out * a.n_frames + (fl0 - b0 * a.n_frames)) * 4u : kDrop;
        fo1 = 2u * fp + 1u < nf ? (b1 * a.n_out * a.n_frames + (fl1 - b1 * a.n_frames)) * 4u : kDrop;
    } else {
        obase = (const float *)a.out + (size_t)b * a.n_out * a.n_frames + f0;
        obytes = (a.n_out * a.n_frames - f0) * 4u;
        fo0 = 2u * fp < nf ? 8u * fp : kDrop;
        fo1 = 2u * fp + 1u < nf ? 8u * fp + 4u : kDrop;
    }
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(obase, obytes);
    // one 16-byte record per (segment, wave, slot): {L of the wave, word offset of the slot's weight row, first bin, band}
    const uint4 *info = (const uint4 *)(sched + kSchedHdr) + wave * 8u + slot;
#if SGX_BANDSEG
    // All records of the stage requested at once: one LDS round trip instead of one per segment (hipcc sinks a record's read next
    // to its first use).  (Skipping the empty segments under a uniform branch is not an option: with a store behind a branch the
    // compiler no longer counts the stage's stores and waits vmcnt(0) at the top of the next tile.)
    uint4 rec[kSchedSegs];
#pragma unroll
    for (int q = 0; q < kSchedSegs; ++q) rec[q] = info[q * 32];
    __builtin_amdgcn_sched_barrier(0);
    uint4 cur = rec[0];
#else
    uint4 cur = info[0];
#endif
    SGX_STAMP(12);  // mel prologue
#pragma unroll
    for (unsigned seg = 0; seg < (unsigned)kSchedSegs; ++seg) {
#if SGX_BANDSEG
        const uint4 nxt = rec[seg + 1u < (unsigned)kSchedSegs ? seg + 1u : seg];
#else
        const uint4 nxt = info[(seg + 1u) * 32u];  // fetched ahead; the table always holds kSchedSegs + 1 segments
#endif
        const unsigned L = __builtin_amdgcn_readfirstlane(cur.x);
        const v4f *wr = (const v4f *)((const float *)sched + cur.y);
        const v4f *pr = (const v4f *)(pwT + (cur.z >> 1) * 32u) + fp;  // kstart is even
#ifdef SGX_STAMPS2  // finer diagnostic: 12 = record + setup of every segment, 13 = the step loops, 14 = epilogues + stores
        asm volatile("" ::"s"(L), "v"(wr), "v"(pr));
        SGX_STAMP(12);
#endif
        v2f acc = {0.0f, 0.0f};
#if SGX_BANDPF
        // Software pipeline, 8 steps (two weight quads, four bin pairs) per group, L a multiple of 8 (host): the operands of group
        // g + 1 are requested before group g is summed, so a segment exposes the LDS latency once instead of once per 4 steps (the
        // compiler's own loop waits for its three reads every trip unless L >= 32).  The read-ahead past the slot's last group hits the
        // row's zero pad / the next row and bins behind the band: never summed.
        auto ld8 = [&](v4f(&w)[2], v4f(&p)[4], unsigned t) {
            w[0] = wr[t >> 2];
            w[1] = wr[(t >> 2) + 1u];
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u] = pr[(t >> 1) * 8u + 8u * u];
        };
        auto sum8 = [&](const v4f(&w)[2], const v4f(&p)[4]) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                acc = mul_add_unfused(w[u].x, (v2f){p[2 * u].x, p[2 * u].y}, acc);
                acc = mul_add_unfused(w[u].y, (v2f){p[2 * u].z, p[2 * u].w}, acc);
                acc = mul_add_unfused(w[u].z, (v2f){p[2 * u + 1].x, p[2 * u + 1].y}, acc);
                acc = mul_add_unfused(w[u].w, (v2f){p[2 * u + 1].z, p[2 * u + 1].w}, acc);
            }
        };
        // (the scheduling barriers keep the machine scheduler from sinking the requests next to their uses, which it does to save
        // registers — that is the exposed latency this loop exists to remove)
        v4f wa[2], pa[4], wb[2], pb[4];
        ld8(wa, pa, 0u);
        __builtin_amdgcn_sched_barrier(0);
        for (unsigned t = 0; t < L; t += 16u) {
            ld8(wb, pb, t + 8u);
            __builtin_amdgcn_sched_barrier(0);
            sum8(wa, pa);
            if (t + 8u >= L) break;
            ld8(wa, pa, t + 16u);
            __builtin_amdgcn_sched_barrier(0);
            sum8(wb, pb);
        }
#elif SGX_BANDTRIP
        // Trips of 16 steps whose 12 reads are requested together and then summed, and a remainder of 4, 8 or 12 steps handled the same
        // way: one LDS round trip per 16 steps.  (hipcc's own loop requests everything up front only from L = 32 on and otherwise
        // waits for its three reads every 4 steps, so the wave holding the medium-length segments — 24 + 20 + 4 steps for Mel-80 —
        // sets the stage's time.)  Same terms, same order.
        {
            v2f pa[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
            auto trip = [&](auto nq, unsigned t) {
                constexpr int NQ = decltype(nq)::value;
                v4f w[NQ], qa[NQ], qb[NQ];
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    w[i] = wr[(t >> 2) + i];
                    qa[i] = pr[((t >> 1) + 2 * i) * 8u];
                    qb[i] = pr[((t >> 1) + 2 * i) * 8u + 8u];
                }
                __builtin_amdgcn_sched_barrier(0);  // (the scheduler sinks the requests next to their uses otherwise)
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
#if SGX_BANDTRIP == 2  // four interleaved partial sums (bins = 0, 1, 2, 3 mod 4), un-fused: NOT the reference's order
                    pa[0] = mul_add_unfused(w[i].x, (v2f){qa[i].x, qa[i].y}, pa[0]);
                    pa[1] = mul_add_unfused(w[i].y, (v2f){qa[i].z, qa[i].w}, pa[1]);
                    pa[2] = mul_add_unfused(w[i].z, (v2f){qb[i].x, qb[i].y}, pa[2]);
                    pa[3] = mul_add_unfused(w[i].w, (v2f){qb[i].z, qb[i].w}, pa[3]);
#else
                    acc = mul_add_unfused(w[i].x, (v2f){qa[i].x, qa[i].y}, acc);
                    acc = mul_add_unfused(w[i].y, (v2f){qa[i].z, qa[i].w}, acc);
                    acc = mul_add_unfused(w[i].z, (v2f){qb[i].x, qb[i].y}, acc);
                    acc = mul_add_unfused(w[i].w, (v2f){qb[i].z, qb[i].w}, acc);
#endif
                }
            };
            unsigned t = 0;
            for (; t + 16u <= L; t += 16u) trip(std::integral_constant<int, 4>{}, t);
            const unsigned rem = L - t;
            if (rem == 12u) trip(std::integral_constant<int, 3>{}, t);
            else if (rem == 8u) trip(std::integral_constant<int, 2>{}, t);
            else if (rem == 4u) trip(std::integral_constant<int, 1>{}, t);
#if SGX_BANDTRIP == 2
            acc = (pa[0] + pa[1]) + (pa[2] + pa[3]);
#endif
        }
#elif SGX_BANDFMA
        // experiment: four interleaved partial sums with fused multiply-adds (NOT the reference's summation order)
        {
            v2f a0 = {0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
            for (unsigned t = 0; t < L; t += 4u) {
                const v4f w4 = wr[t >> 2], q0 = pr[(t >> 1) * 8u], q1 = pr[(t >> 1) * 8u + 8u];
                a0 = pfma((v2f){w4.x, w4.x}, (v2f){q0.x, q0.y}, a0);
                a1 = pfma((v2f){w4.y, w4.y}, (v2f){q0.z, q0.w}, a1);
                a2 = pfma((v2f){w4.z, w4.z}, (v2f){q1.x, q1.y}, a2);
                a3 = pfma((v2f){w4.w, w4.w}, (v2f){q1.z, q1.w}, a3);
            }
            acc = (a0 + a1) + (a2 + a3);
        }
#else
        for (unsigned t = 0; t < L; t += 4u) {  // q0 = (bin t: frames f, f+1; bin t+1: frames f, f+1)
#if defined(SGX_ABL_BANDW) && defined(SGX_ABL_BANDP)  // timing experiments only (wrong results): no reads at all
            const v4f w4 = {1.f, 0.5f, 0.25f, 2.f}, q0 = {(float)t, 1.f, 2.f, 3.f}, q1 = {4.f, 5.f, (float)t, 7.f};
#elif defined(SGX_ABL_BANDW)  // no weight reads / no |X|^2 reads
            const v4f w4 = {1.f, 0.5f, 0.25f, 2.f}, q0 = pr[(t >> 1) * 8u], q1 = pr[(t >> 1) * 8u + 8u];
#elif defined(SGX_ABL_BANDP)
            const v4f w4 = wr[t >> 2], q0 = {(float)t, 1.f, 2.f, 3.f}, q1 = {4.f, 5.f, (float)t, 7.f};
#else
            const v4f w4 = wr[t >> 2], q0 = pr[(t >> 1) * 8u], q1 = pr[(t >> 1) * 8u + 8u];
#endif
            acc = mul_add_unfused(w4.x, (v2f){q0.x, q0.y}, acc);
            acc = mul_add_unfused(w4.y, (v2f){q0.z, q0.w}, acc);
            acc = mul_add_unfused(w4.z, (v2f){q1.x, q1.y}, acc);
            acc = mul_add_unfused(w4.w, (v2f){q1.z, q1.w}, acc);
        }
#endif
        SGX_STAMP(13);  // mel loops
        const bool have = cur.w != 0xffffffffu;
        if constexpr (TOLDS) {
            if (have) {
                float *o = mfl + ((cur.w & 3u) * 16u + 2u * fp) * (unsigned)TOLDS + (cur.w >> 2);
                o[0] = amp_f32<AMP>(acc.x, eps);
                o[TOLDS] = amp_f32<AMP>(acc.y, eps);
            }
        } else {
        const unsigned bo = cur.w * a.n_frames * 4u;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, amp_f32<AMP>(acc.x, eps)), ro, (int)((have && fo0 != kDrop) ? bo + fo0 : kDrop), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, amp_f32<AMP>(acc.y, eps)), ro, (int)((have && fo1 != kDrop) ? bo + fo1 : kDrop), 0, 0);
        }
        cur = nxt;
#ifdef SGX_STAMPS2
        SGX_STAMP(14);
#endif
    }
}

// ---- fused MFCC epilogue (round 5; SURVEY.md §8 f1, src/mfcc.rs:224-316) ------------------------------------------------------------
// The DCT-II of the reference,  c[k] = fold over bands i ascending of  val_i.mul_add(basis[k][i], acc)   (src/mfcc.rs:278-292),
// is the product [16 coefficients x n_mels] x [n_mels x 16 frames] with ONE fused-multiply-add chain per output in ascending band order —
// what v_mfma_f32_16x16x4_f32 computes (exact f32; step s covers bands 4 s .. 4 s + 3 in order).  Lane (n = l & 15, q = l >> 4) feeds
// A = basis[16 mt + n][4 s + q] and B = Mel-dB[band 4 s + q][frame n]; D[r] = coefficient 16 mt + 4 q + r of frame n, times the lifter
// weight (apply_liftering :297-316; the weights ride behind the fragments, 1.0 without a lifter), stored frame-contiguous (64-byte runs)
// without C0 when the plan drops it (:262-268).  The Mel-dB tensor never reaches HBM.
//
// Both operands sit in LDS with a lane's STEPS values contiguous — the band stage writes the tile's dB values as mq[q][n][s]
// (mel_tile_sched<TOLDS>), the host lays the basis out as frag[mt][lane][s] — so a lane fetches them with STEPS / 2 16-byte reads (rows
// of RL floats, RL / 4 odd: the 16 lanes of a read group on different banks).  Steps past the last band meet a zero weight and whatever
// finite value the exchange left there.  One wave per half and 16 coefficients runs the chain (the f32 matrix instructions share the
// vector pipe: all eight waves running it redundantly, threaded through the next tile's pass 1, cost 12 us per 256 x 10 s for the chain and
// 8 for its operand reads — profiles/experiments_r05/mfcc_fusion.md), at the top of the NEXT tile behind its barrier 1, where the registers are
// still free and the barrier that publishes the band stage's writes is one the loop has anyway; the
// stores are outside the branch so that every wave issues the same vector-memory operations per tile (counted vmcnt at the loop top).
template <int STEPS>
constexpr int mfcc_row() { return (STEPS / 4) % 2 ? STEPS : STEPS + 4; }
template <int STEPS>
__device__ __forceinline__ void mfcc_tile(const StftArgs &a, const float *mq, const float *frag, unsigned mt, unsigned b, unsigned f0, unsigned nf,
                                          unsigned lane) {
    static_assert(STEPS % 4 == 0, "chain lengths are whole 16-byte reads");
    constexpr int RL = mfcc_row<STEPS>();
    const unsigned n = lane & 15u, q = lane >> 4;
    const bool mine = mt < a.mfcc_mtiles;  // wave-uniform
    v4acc acc = {0.f, 0.f, 0.f, 0.f};
    if (mine) {
        const v4f *pa = (const v4f *)(frag + ((size_t)mt * 64u + lane) * RL);
        const v4f *pb = (const v4f *)(mq + (q * 16u + n) * RL);
        v4f wa[STEPS / 4], vb[STEPS / 4];
#pragma unroll
        for (int g = 0; g < STEPS / 4; ++g) { wa[g] = pa[g]; vb[g] = pb[g]; }
#pragma unroll
        for (int g = 0; g < STEPS / 4; ++g) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[g].x, vb[g].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[g].y, vb[g].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[g].z, vb[g].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[g].w, vb[g].w, acc, 0, 0, 0);
        }
    }
    const float *lift = frag + (size_t)a.mfcc_mtiles * 64u * RL;
    constexpr unsigned kDrop = 0x80000000u;
    const unsigned rows = a.n_mfcc - a.mfcc_skip;
    const unsigned obytes = (rows * a.n_frames - f0) * 4u;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc((const float *)a.out + (size_t)b * rows * a.n_frames + f0, obytes);
    const unsigned mtr = mine ? mt : 0u;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned c = 16u * mtr + 4u * q + r;
        const bool ok = mine && c >= a.mfcc_skip && c < a.n_mfcc && n < nf;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc[r] * lift[min(c, a.n_mfcc - 1u)]), ro,
                                              (int)(ok ? ((c - a.mfcc_skip) * a.n_frames + n) * 4u : kDrop), 0, 0);
    }
}

// P512: |X|^2 tile of 32 frames: bin pair x frame pair = four floats, 16 frame pairs per bin pair
__host__ __device__ constexpr unsigned pwt512_index(unsigned k, unsigned f) { return (k >> 1) * 64u + (f >> 1) * 4u + (k & 1u) * 2u + (f & 1u); }
// The same schedule (records per segment, wave, 8 slots) walked by lanes (4 slots x 16 frame pairs): a lane takes slots s and s + 4
// of every segment in turn.  A 16-lane read group is one slot's 16 frame pairs: 256 contiguous bytes.
template <int AMP>
__device__ __forceinline__ void mel_tile_sched512(const StftArgs &a, const float *pwT, const unsigned *sched, unsigned b, unsigned f0,
                                                  unsigned nf, float eps, unsigned tid) {
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u, slot = lane >> 4, fp = lane & 15u;
    const unsigned obytes = (a.n_out * a.n_frames - f0) * 4u;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc((const float *)a.out + (size_t)b * a.n_out * a.n_frames + f0, obytes);
    constexpr unsigned kDrop = 0x80000000u;
    const unsigned fo0 = 2u * fp < nf ? 8u * fp : kDrop, fo1 = 2u * fp + 1u < nf ? 8u * fp + 4u : kDrop;
    const uint4 *info = (const uint4 *)(sched + kSchedHdr) + wave * 8u + slot;
    uint4 ca = info[0], cb = info[4];
#pragma unroll
    for (unsigned seg = 0; seg < (unsigned)kSchedSegs; ++seg) {
        // slots s and s + 4 of the segment side by side: two independent sums per lane, their LDS reads in flight together
        const uint4 na = info[(seg + 1u) * 32u], nb = info[(seg + 1u) * 32u + 4u];  // fetched ahead; the table holds kSchedSegs + 1 segments
        const unsigned L = __builtin_amdgcn_readfirstlane(ca.x);
        const v4f *wa = (const v4f *)((const float *)sched + ca.y), *wb = (const v4f *)((const float *)sched + cb.y);
        const v4f *pa = (const v4f *)(pwT + (ca.z >> 1) * 64u) + fp, *pb = (const v4f *)(pwT + (cb.z >> 1) * 64u) + fp;  // kstart is even
        v2f acca = {0.0f, 0.0f}, accb = {0.0f, 0.0f};
        for (unsigned t = 0; t < L; t += 4u) {
            const v4f w4a = wa[t >> 2], a0 = pa[(t >> 1) * 16u], a1 = pa[(t >> 1) * 16u + 16u];
            const v4f w4b = wb[t >> 2], b0 = pb[(t >> 1) * 16u], b1 = pb[(t >> 1) * 16u + 16u];
            acca = mul_add_unfused(w4a.x, (v2f){a0.x, a0.y}, acca);
            accb = mul_add_unfused(w4b.x, (v2f){b0.x, b0.y}, accb);
            acca = mul_add_unfused(w4a.y, (v2f){a0.z, a0.w}, acca);
            accb = mul_add_unfused(w4b.y, (v2f){b0.z, b0.w}, accb);
            acca = mul_add_unfused(w4a.z, (v2f){a1.x, a1.y}, acca);
            accb = mul_add_unfused(w4b.z, (v2f){b1.x, b1.y}, accb);
            acca = mul_add_unfused(w4a.w, (v2f){a1.z, a1.w}, acca);
            accb = mul_add_unfused(w4b.w, (v2f){b1.z, b1.w}, accb);
        }
        const bool ha = ca.w != 0xffffffffu, hb = cb.w != 0xffffffffu;
        const unsigned boa = ca.w * a.n_frames * 4u, bob = cb.w * a.n_frames * 4u;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, amp_f32<AMP>(acca.x, eps)), ro, (int)((ha && fo0 != kDrop) ? boa + fo0 : kDrop), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, amp_f32<AMP>(acca.y, eps)), ro, (int)((ha && fo1 != kDrop) ? boa + fo1 : kDrop), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, amp_f32<AMP>(accb.x, eps)), ro, (int)((hb && fo0 != kDrop) ? bob + fo0 : kDrop), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, amp_f32<AMP>(accb.y, eps)), ro, (int)((hb && fo1 != kDrop) ? bob + fo1 : kDrop), 0, 0);
        ca = na;
        cb = nb;
    }
}

// ====================================================================================================================
// k_r32x16: one persistent 512-thread workgroup per CU; its two halves each own a tile and an ex buffer and move through the
// phases in lockstep (shared barriers) — measured 6-30 % faster than two independent 256-thread workgroups (round 1).
// ROUNDS > 0: the tile's (15*hop + 1024) samples are fetched ONCE with 16-byte buffer loads (ROUNDS per thread, one tile
// ahead), staged in LDS (xs, overlaying this half's ex) and re-read per frame from there.  ROUNDS == 0 (hop > 272, e.g. the
// row pass of the 2-D path with hop = 1024): per-lane 8-byte loads.
// WIDE (linear / complex outputs): pass 2 runs across the whole workgroup — lane (jq = 0..1, f = 0..31) of wave
// w = 0..7 owns job w + 8 jq of frame f of the PAIR of tiles (frames 16..31 live in the second half's ex buffer, which
// directly follows the first: kExBytes = 16 kFS).  The two halves own neighbouring tiles, so the 32 lanes of a job hold one
// bin of 32 consecutive frames: a store instruction covers 2 rows x 128 bytes instead of 4 rows x 64 bytes.
// XSPAD (hop = 256): the staging buffer carries 128 B of padding per KiB, which keeps the 4 frames of a wave on distinct banks.
// PWT (filterbank outputs): |X|^2 tile transposed + schedule (else pw[f][k] + matrix cores / CSR).
// ====================================================================================================================
// P512 (n_fft = 512, hop = 128, per-bin outputs): a tile is 32 frames, a slot of the exchange buffer holds the 512-point complex
// transform of TWO consecutive frames (real parts: frame 2 p, imaginary parts: frame 2 p + 1); passes 1 and 2 are unchanged (the
// same 32 x 16 transform), the real split becomes the two-sequence split (pass2_pair512).  The staged samples carry 64 B of
// padding per KiB: the four slots of a wave start 1 KiB apart and would otherwise read the same banks.
// PACK (n_fft = 1024, batches of short signals): a tile is 16 consecutive frames of the BATCH — global frame g = 16 tile + slot is
// frame g mod n_frames of signal g / n_frames — so a signal of 4 frames fills a quarter of a tile and the next three signals fill
// the rest (one-signal tiles left 3/4 of every tile empty: 0.44 G frames/s for 65 536 x 4 frames).  Samples come per lane (the
// direct path, ROUNDS = 0) through one descriptor over the whole batch, each pair range-checked against its own row in the lane;
// outputs through one descriptor from the tile's first signal.  The reference's per-frame loop costs the same per frame whatever
// the signal length (src/spectrogram.rs:240-294).
template <int MODE, int AMP, int ROUNDS, bool WIDE, bool XSPAD, bool PWT, int HOP512 = 0, bool PACK = false, int MSTEPS = 0>
__global__ __launch_bounds__(512, 2) void k_r32x16(StftArgs a, unsigned per_xcd, unsigned total, unsigned slots) {
    constexpr bool MFCC = MSTEPS != 0;  // fused MFCC epilogue, MSTEPS = ceil(n_mels / 4) rounded up to the menu (mfcc_tile)
    static_assert(!MFCC || (PWT && AMP == AMP_DB && HOP512 == 0 && !PACK), "fused MFCC epilogue: Mel-dB on the schedule, n_fft 1024, one-signal tiles");
    constexpr bool P512 = HOP512 != 0;           // n_fft 512 at hop HOP512
    static_assert(!PACK || (ROUNDS == 0 && !WIDE && !XSPAD && (HOP512 == 0 ? (MODE != OUT_MEL || PWT) : MODE != OUT_MEL)),
                  "PACK: direct loads, one-half tiles; n_fft 1024: scheduled band stage, n_fft 512: per-bin outputs");
    constexpr unsigned SS512 = 8u * HOP512;      // bytes from one slot's (frame pair's) first sample to the next slot's
    static_assert(!P512 || PACK || (HOP512 % 4 == 0 && ROUNDS * 256 * 4 >= 31 * HOP512 + 512), "P512: 16-byte chunks, the whole tile staged");
    static_assert(!WIDE || MODE != OUT_MEL, "wide pass 2 needs a per-bin output");
    static_assert(!PWT || MODE == OUT_MEL, "PWT is a filterbank layout");
    static_assert(!P512 || (!WIDE && !XSPAD && (ROUNDS > 0 || PACK) && (PWT == (MODE == OUT_MEL))), "P512: staged samples (PACK: per-lane loads), scheduled band stage");
    constexpr bool DMA = SGX_DMA != 0 && PWT && ROUNDS > 0 && !P512;  // samples HBM -> LDS without registers (see dma16_to_lds)
    constexpr unsigned FPT = P512 ? 32u : 16u;   // frames per tile
    constexpr unsigned NB = P512 ? 257u : 513u;  // bins
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const unsigned half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    const unsigned tid = threadIdx.x & 255u;
    constexpr bool H256M = HOP512 == 256 && MODE == OUT_MEL && !PACK;  // larger halves: r32x16_layout.h
    constexpr unsigned EXB = H256M ? (unsigned)kExBytesH256 : (unsigned)kExBytes, POFF = H256M ? (unsigned)kOutOffH256 : (unsigned)kOutOff;
    unsigned char *smem = smem_all + half * EXB;  // this half's ex / xs / pw buffer
    unsigned char *tabs = smem_all + 2 * EXB;     // tables sit behind the two ex buffers
    if (threadIdx.x < 256u) ((v4f *)(tabs + kWinOff))[threadIdx.x] = ((const v4f *)a.window)[threadIdx.x];
    for (unsigned i = threadIdx.x; i < 16u * 17u; i += 512u) ((v4f *)(tabs + kTw2Off))[i] = ((const v4f *)a.tw2)[i];
    unsigned *sched = (unsigned *)(tabs + kMelOff);
    if constexpr (PWT)
        for (unsigned i = threadIdx.x; i < a.mel_sched_words; i += 512u) sched[i] = a.mel_sched[i];
    float *mfrag = (float *)(tabs + kMelOff) + ((a.mel_sched_words + 3u) & ~3u);  // MFCC: the basis fragments behind the schedule
    if constexpr (MFCC)
        for (unsigned i = threadIdx.x; i < a.mfcc_frag_words; i += 512u) mfrag[i] = ((const float *)a.mfcc_frag)[i];

    // XCD-aware work mapping: blocks g and g+8 share an XCD (round-robin dispatch).  XCD x owns the contiguous run of
    // work ids [x*per_xcd, (x+1)*per_xcd); its `slots` resident workgroups walk that run with stride `slots`, so tiles
    // in flight on one XCD are neighbours: they share the 768-sample halo and the output lines they both touch in L2.
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned lo = xcd * per_xcd;
    const unsigned hi = min(lo + per_xcd, total);
    unsigned wid = lo + slot * 2u + half;
    unsigned lead = lo + slot * 2u;  // the first half's tile: uniform loop control for the whole workgroup

    const unsigned p1f = tid >> 4, n2 = tid & 15u;  // pass-1 identity
    // pass-2 identity
    const unsigned lane = tid & 63u;
    const unsigned wv_ = WIDE ? threadIdx.x >> 6 : tid >> 6;
    const unsigned jq = WIDE ? lane >> 5 : lane >> 4, p2f = WIDE ? lane & 31u : lane & 15u;
    const unsigned j = wv_ + (WIDE ? 8u : 4u) * jq;
    const unsigned ra = j, rb = j == 0 ? 16u : 32u - j;
    const float eps = (float)a.eps;
    constexpr unsigned ES = MODE == OUT_COMPLEX ? 8u : 4u;
    const JobOfs jo = job_offsets(j, a.n_frames);
    const unsigned step = 32u * a.n_frames * ES;  // uniform: 32 bins further
    const v4f *twj = (const v4f *)(tabs + kTw2Off) + j * 17u;
    v2f twa[4], twb[8];
    load_tw1(a, n2, twa, twb);

    // staged path: this thread's 16-byte chunks of the tile being prefetched; direct path: its column
    constexpr int NCR = ROUNDS > 0 ? ROUNDS : 1;
    v4f creg[NCR];
    v2f xd[ROUNDS > 0 ? 1 : 32];
    const unsigned chunks = XSPAD ? 1216u : P512 ? (31u * HOP512 + 512u + 3u) >> 2 : (15u * a.hop + 1024u + 3u) >> 2;
    const unsigned hop = XSPAD ? 256u : (P512 && !PACK) ? (unsigned)HOP512 : a.hop;  // (PACK at n_fft 512: HOP512 only marks the mode)
    const unsigned row_bytes = (unsigned)a.n_samples * 4u;  // host: n_samples < 2^29
    auto load_tile = [&](unsigned w) {
        const unsigned b = w / a.tiles, tile = w - b * a.tiles;
        const unsigned f0 = tile * FPT;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc((const float *)a.x + (size_t)b * a.sample_stride, row_bytes);
        // first sample of the tile relative to the row; negative in the left padding: as an unsigned byte offset it is far out
        // of range, so the hardware returns 0 there as it does past the end of the row (S1: zero padding)
        const int tile_lo = (int)(f0 * hop) - (int)a.pad;
        const bool interior = tile_lo >= 0 && (unsigned)tile_lo + (FPT - 1u) * hop + (P512 ? 512u : 1024u) <= (unsigned)a.n_samples;  // wave-uniform
        if constexpr (ROUNDS > 0) {
#ifdef SGX_ABL_NOGLOAD
            for (int r = 0; r < ROUNDS; ++r) creg[r] = (v4f){(float)w, 1.f, 2.f, (float)r};
            return;
#endif
            const int vo = (tile_lo + 4 * (int)tid) * 4;
            if constexpr (DMA) {
                dma_rounds<XSPAD, 0, ROUNDS>(rx, vo, lds_addr(smem), __builtin_amdgcn_readfirstlane(tid >> 6));
                return;
            }
            // One straight-line path for every tile and every chunk round: the range check of a 16-byte buffer load is per dword (the
            // compiler merged the edge tiles' four dword loads into this same instruction anyway) and a chunk never straddles the row
            // start (the tile start and the padding are multiples of 4 samples).  Before: an interior and an edge path over the same
            // registers, and rounds predicated on the tile's chunk count — the merged control flow made the compiler wait for the
            // loads (and with them for the previous tile's stores) right after issuing them at every hop but 256.  Chunks past the
            // tile read the following samples (L2 hits) or zeros and are staged into LDS the transforms never read.
            if (SGX_ONEPATH || interior || P512) {
#pragma unroll
                for (int r = 0; r < ROUNDS; ++r)
                    if (SGX_ONEPATH || XSPAD || P512 || r * 256u + tid < chunks) creg[r] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rx, vo + r * 4096, 0, 0));
            } else {  // edge tile: dword loads, each bounds-checked on its own
#pragma unroll
                for (int r = 0; r < ROUNDS; ++r) {
                    v4i c;
#pragma unroll
                    for (int e = 0; e < 4; ++e) c[e] = __builtin_amdgcn_raw_buffer_load_b32(rx, vo + r * 4096 + 4 * e, 0, 0);
                    creg[r] = __builtin_bit_cast(v4f, c);
                }
            }
        } else if constexpr (PACK && P512) {
            // Slot p1f of packed tile w = slot q of the batch = frames (2 pr, 2 pr + 1) of signal bq (PP = ceil(n_frames / 2) slots per
            // signal): the pairing of a one-signal tile, so a signal's bits do not depend on the batch around it — its last slot's
            // second frame may be the virtual frame n_frames, loaded like any other and never stored.  One dword per sample and frame,
            // range-checked against the row in the lane.
            const __amdgpu_buffer_rsrc_t rall = make_rsrc(a.x, a.x_bytes);
            const unsigned ns = (unsigned)a.n_samples, PP = (a.n_frames + 1u) >> 1;
            const unsigned q = w * 16u + p1f;
            const bool live = q < a.gframes;
            const unsigned bq = live ? q / PP : 0u, pr = q - bq * PP;
            const int sA = (int)(2u * pr * hop) - (int)a.pad + (int)n2, sB = sA + (int)hop;
            const unsigned row = bq * (unsigned)a.sample_stride;
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) {
                const int ia = sA + 16 * n1, ib = sB + 16 * n1;
                const int va = __builtin_amdgcn_raw_buffer_load_b32(rall, (live && (unsigned)ia < ns) ? (int)((row + (unsigned)ia) * 4u) : (int)0xfffffff0u, 0, 0);
                const int vb = __builtin_amdgcn_raw_buffer_load_b32(rall, (live && (unsigned)ib < ns) ? (int)((row + (unsigned)ib) * 4u) : (int)0xfffffff0u, 0, 0);
                xd[n1] = (v2f){__builtin_bit_cast(float, va), __builtin_bit_cast(float, vb)};
            }
        } else if constexpr (PACK) {
            // slot p1f of packed tile w = global frame g = frame fq of signal bq.  One descriptor over the whole batch (host: its bytes
            // fit 32 bits): the row's own range is checked here — a pair is inside, outside (offset past the descriptor: reads 0, the
            // zero padding S1), or, for an odd row length, straddles the row end (second sample cleared).
            const unsigned g = w * 16u + p1f;
            const bool live = g < a.gframes;
            const unsigned bq = live ? g / a.n_frames : 0u, fq = g - bq * a.n_frames;
            const __amdgpu_buffer_rsrc_t rall = make_rsrc(a.x, a.x_bytes);
            const int s0 = (int)(fq * hop) - (int)a.pad + 2 * (int)n2;
            const unsigned rowo = bq * (unsigned)a.sample_stride;
            const unsigned ns = (unsigned)a.n_samples;
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) {
                const int sidx = s0 + 32 * n1;
                const bool in0 = live && (unsigned)sidx < ns;
                v2f v = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rall, in0 ? (int)((rowo + (unsigned)sidx) * 4u) : (int)0xfffffff0u, 0, 0));
                if ((unsigned)(sidx + 1) >= ns) v.y = 0.0f;
                xd[n1] = v;
            }
        } else {
            const int vo = ((int)(p1f * hop) + tile_lo + 2 * (int)n2) * 4;
            if (SGX_ONEPATH || interior) {  // (one path here too: with an even hop a pair never straddles the row start)
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) xd[n1] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rx, vo + n1 * 128, 0, 0));
#if SGX_ODDHOP
                if (a.hop & 1u) {  // uniform.  Odd frames sit on odd sample offsets: the pair (x[-1], x[0]) starts outside the row, and an
                                   // 8-byte access whose first dword is out of range returns 0 for both: put x[0] back
                    const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, 0, 0, 0));
                    const int s0 = (int)(p1f * hop) + tile_lo + 2 * (int)n2;
#pragma unroll
                    for (int n1 = 0; n1 < 32; ++n1)
                        if (s0 + 32 * n1 == -1) xd[n1].y = x0;
                }
#endif
            } else {
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) {
                    v2i c;
                    c.x = __builtin_amdgcn_raw_buffer_load_b32(rx, vo + n1 * 128, 0, 0);
                    c.y = __builtin_amdgcn_raw_buffer_load_b32(rx, vo + n1 * 128 + 4, 0, 0);
                    xd[n1] = __builtin_bit_cast(v2f, c);
                }
            }
        }
    };

    // a second half without a tile of its own (odd run length: the last round of an XCD) repeats the first half's tile — same
    // values to the same addresses — so both halves always run the same number of rounds and barriers
    if (wid >= hi) wid = lead;
    if (lead < hi) load_tile(wid);
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the first tile's samples are in LDS (nothing else is in flight yet)
    __syncthreads();  // tables visible

    const unsigned xaddr = lds_addr(smem) + p1f * hop * 4u + n2 * 8u + (XSPAD ? p1f * 128u : 0u);
    const unsigned waddr = lds_addr(tabs + kWinOff) + n2 * 8u;

#ifdef SGX_SKEW  // experiment (Mel-type outputs: the halves are independent): the second half enters the loop SGX_SKEW barriers late
    if (MODE == OUT_MEL && half == 1u)
        for (unsigned q = 0; q < SGX_SKEW; ++q) __syncthreads();
#endif
    // Per-bin outputs: the workgroups of an XCD start a few hundred cycles apart (slot s waits ~256 s cycles, one round at most).
    // All CUs run the same program on the same amount of work, so without this their store bursts (66 KB per CU per round) hit
    // the memory system together and their arithmetic phases leave it idle together: measured 141 -> 129 us (linear power),
    // 240 -> 221 us (complex); the filterbank outputs write little and lose 3 % (left alone).
#ifndef SGX_STAGGER
#define SGX_STAGGER 4u
#endif
    if constexpr (MODE != OUT_MEL) {
        for (unsigned q = 0; q < (blockIdx.x >> 3) * SGX_STAGGER; ++q) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
    }
#ifdef SGX_PRIO  // experiment: static priority for the younger half (MI355X_MICROARCH.md, two waves per SIMD, item 4)
    if (half == SGX_PRIO - 1u) __builtin_amdgcn_s_setprio(1);
#endif
#ifdef SGX_STAMPS
    unsigned long long st_acc[16] = {0}, st_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
    const unsigned mfcc_mt = (((threadIdx.x & 255u) >> 6) + 4u - half) & 3u;  // (the halves' first blocks sit on different SIMDs: waves w and w + 4 share one)
    // Mel-dB tile of the fused MFCC epilogue: 64 rows of RL floats between the staged samples (<= 22 912 B) and the |X|^2 tile
    float *const mfl = (float *)(smem + kOutOff) - 64 * mfcc_row<MFCC ? MSTEPS : 4>();
    unsigned pvb = 0, pvf0 = 0, pvnf = 0;  // MFCC: this half's previous tile, whose Mel-dB values wait in LDS
    static_assert(!MFCC || kOutOff - 256 * mfcc_row<MFCC ? MSTEPS : 4>() >= 22912, "Mel-dB tile above the staged samples");
    while (lead < hi) {
        // (PACK: b = the tile's first signal, f0 = its first frame's index in that signal, nf = the tile's live slots)
        // (PACK at n_fft 512 packs SLOTS — frame pairs of one signal, PP per signal —: b = the first slot's signal, nf = live slots)
        const unsigned PP = (a.n_frames + 1u) >> 1;
        const unsigned b = PACK ? (P512 ? wid * 16u / PP : wid * 16u / a.n_frames) : wid / a.tiles, tile = wid - b * a.tiles;
        const unsigned f0 = PACK ? wid * 16u - b * a.n_frames : tile * FPT;
        const unsigned nf = PACK ? min(16u, a.gframes - wid * 16u) : min(FPT, a.n_frames - f0);
        v2f xr[32];
        {
            v2f e[16], o[16], we[16], wo[16];
            if constexpr (ROUNDS > 0) {
                // stage: chunk c of the tile -> xs (this half's ex is free: barrier 4 of the previous tile / the prologue)
                if constexpr (DMA) {
                    // this wave's requests have landed: they are older than the 2 kSchedSegs band-stage stores issued behind them (the
                    // first tile's were collected before the loop); barrier 1 then covers the other waves' pieces
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kSchedSegs) : "memory");
                } else {
#pragma unroll
                for (int r = 0; r < ROUNDS; ++r) {
                    const unsigned c = r * 256u + tid;
                    if (SGX_ONEPATH || XSPAD || P512 || c < chunks) *(v4f *)(smem + c * 16u + (XSPAD ? (c >> 6) * 128u : P512 ? (c * 16u / (P512 ? SS512 : 1u)) * 64u : 0u)) = creg[r];
                }
                }
                SGX_STAMP(0);  // wait for the samples + staging writes
                __syncthreads();  // barrier 1: xs complete
                SGX_STAMP(1);
                // MFCC: the PREVIOUS tile's chain runs here, behind barrier 1 (which also covers the band stage's Mel-dB writes: no barrier of its
                // own — 126.6 -> 124.6 us per 256 x 10 s); the first tile's "previous" has nf = 0: every store dropped
                if constexpr (MFCC) mfcc_tile<MFCC ? MSTEPS : 4>(a, mfl, mfrag, mfcc_mt, pvb, pvf0, pvnf, tid & 63u);
                if constexpr (HOP512 == 256) {
                    // hop 256: frame 2 p + 1 starts 1024 (+ pad) bytes behind frame 2 p — out of reach of ds_read2_b32's 8-bit dword
                    // offsets — so the pair is two 4-byte reads with 16-bit immediates off one base
                    const float *pa = (const float *)(smem + p1f * (SS512 + 64u) + n2 * 4u);
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        e[k] = (v2f){pa[p512_off_a<256>(2 * k) / 4], pa[p512_off_b<256>(2 * k) / 4]};
                        o[k] = (v2f){pa[p512_off_a<256>(2 * k + 1) / 4], pa[p512_off_b<256>(2 * k + 1) / 4]};
                    }
                    read_win<0>(we, waddr, std::make_integer_sequence<int, 16>{});
                    read_win<1>(wo, waddr, std::make_integer_sequence<int, 16>{});
                    // (the window reads are asm and not counted by the compiler: everything is collected here)
                    tie16<0>(e);
                    tie16<-1>(o);
                    tie16<-1>(we);
                    tie16<-1>(wo);
                } else if constexpr (P512) {
                    unsigned base[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) base[q] = lds_addr(smem) + p1f * (SS512 + 64u) + n2 * 4u + (unsigned)p512_off_a<P512 ? HOP512 : 128>(4 * q);
                    read_cols512<P512 ? HOP512 : 128, 0>(e, we, base, waddr, std::make_integer_sequence<int, 16>{});
                    read_cols512<P512 ? HOP512 : 128, 1>(o, wo, base, waddr, std::make_integer_sequence<int, 16>{});
                    tie16<15>(e);
                    tie16<-1>(we);
                } else {
                if (SGX_ODDHOP && !XSPAD && (a.hop & 1u)) {  // uniform
                    const unsigned base[4] = {xaddr, xaddr + 1024u, xaddr + 2048u, xaddr + 3072u};
                    read_cols_odd<0>(e, we, base, waddr, std::make_integer_sequence<int, 16>{});
                    read_cols_odd<1>(o, wo, base, waddr, std::make_integer_sequence<int, 16>{});
                } else {
                read_cols<XSPAD, 0>(e, we, xaddr, waddr, std::make_integer_sequence<int, 16>{});
                read_cols<XSPAD, 1>(o, wo, xaddr, waddr, std::make_integer_sequence<int, 16>{});
                }
                tie16<15>(e);  // at most 15 of the 64 reads outstanding: the 32 of (e, we) have landed
                tie16<-1>(we);
                }
            } else {
                const v2f *w2 = (const v2f *)(tabs + kWinOff) + n2;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    e[k] = xd[2 * k];
                    o[k] = xd[2 * k + 1];
                    we[k] = w2[32 * k];
                    wo[k] = w2[32 * k + 16];
                }
            }
            Fft<16, true>::run(e, we);
            if constexpr (ROUNDS > 0) {
                tie16<0>(o);
                tie16<-1>(wo);
            }
            Fft<16, true>::run(o, wo);
            Comb<32, 0, v2f>::run(xr, e, o);
        }
        SGX_STAMP(2);  // column / window reads + 32-point transform
        // barrier 2: every wave has read its columns of xs; pass 1 may overwrite it with ex.  It sits behind the arithmetic:
        // by now the slowest wave's reads landed long ago, so nobody waits here.
        if constexpr (ROUNDS > 0 || MFCC) __syncthreads();  // (MFCC on the direct path: the chain's wave has read the previous tile's Mel-dB values)
        SGX_STAMP(3);  // barrier 2
#ifdef SGX_ABL_ADDTID
        {
            const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_addr(smem) + (tid >> 6) * 16384u);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(m0v) : "memory");
        }
#endif
        twiddle_store(xr, twa, twb, smem + p1f * kFS + n2 * 8);
        SGX_STAMP(4);  // twiddles + ex writes
        unsigned next = lead + slots * 2u + half;
        if (next >= hi) next -= half;  // no tile of its own next round: repeat the first half's
        // requested after pass 1 so the previous tile's store burst has had that long to drain: a vector load issued while
        // the CU's store FIFO is backed up stalls its wave for thousands of cycles
        if constexpr (!DMA)
            if (lead + slots * 2u < hi) load_tile(next);  // in flight during pass 2
        SGX_STAMP(5);  // load issue
        __syncthreads();  // barrier 3: ex complete
        SGX_STAMP(6);
        // Linear / complex outputs: every lane runs pass 2 and stores, so the compiler can count the stores behind the next
        // tile's loads (one in-order counter for loads and stores: it then waits with vmcnt(33), not vmcnt(0)).  A lane whose
        // frame does not exist (last tile of a signal) mirrors the tile's last frame, an idle second half (odd tile count)
        // mirrors the first half's tile: same values to the same addresses.
        constexpr bool ALLSTORE = MODE != OUT_MEL;
        unsigned p2b = b, p2ofs, p2ex;
        if constexpr (WIDE) {
            // this lane's frame belongs to the first half's tile (`lead`) or, for p2f >= 16, to the second half's (`lead + 1`, if it
            // exists — otherwise the lane mirrors the first tile): two uniform decodes and a per-lane select
            const unsigned w1 = lead + 1u < hi ? lead + 1u : lead;  // = the second half's wid
            const unsigned b0 = lead / a.tiles, f00 = (lead - b0 * a.tiles) * 16u;
            const unsigned b1 = w1 / a.tiles, f01 = (w1 - b1 * a.tiles) * 16u;
            const unsigned nf0 = min(16u, a.n_frames - f00), nf1 = min(16u, a.n_frames - f01);
            const bool second = p2f >= 16u && w1 != lead;
            const unsigned fle = min(p2f & 15u, (second ? nf1 : nf0) - 1u);
            p2ex = (second ? 16u : 0u) + fle;
            // one descriptor (signal b0) for the whole workgroup; a second tile in the next signal is one signal further
            // (513 n_frames elements: the host guarantees 2 * 513 * n_frames * 8 < 2^32)
            p2ofs = (second ? f01 : f00) + fle + ((second && b1 != b0) ? 513u * a.n_frames : 0u);
            p2b = b0;
        } else if constexpr (P512) {
            p2ex = p2f;             // slot
            p2ofs = f0 + 2u * p2f;  // its first frame
        } else if constexpr (PACK) {
            // slot -> (signal, frame): the offset is relative to the tile's first signal b, where the descriptor starts
            const unsigned p2f_eff = ALLSTORE ? min(p2f, nf - 1u) : p2f;
            const unsigned fl = f0 + p2f_eff, bq = fl / a.n_frames;  // signals past b
            p2ex = p2f_eff;
            p2ofs = bq * NB * a.n_frames + (fl - bq * a.n_frames);
        } else {
            const unsigned p2f_eff = ALLSTORE ? min(p2f, nf - 1u) : p2f;
            p2ex = p2f_eff;
            p2ofs = f0 + p2f_eff;
        }
        v2f A[16], B[16];
        read_rows((WIDE ? smem_all : smem) + p2ex * kFS, ra, rb, A, B);
        SGX_STAMP(7);  // row reads
        __syncthreads();  // barrier 4: ex consumed: the next staging (or the pw overlay) may overwrite it
        SGX_STAMP(8);
        // DMA: the staging area (below the |X|^2 tile) is free from here on; the next tile's samples land in it during the real split
        // and the band stage (~5 k cycles: more than an HBM round trip)
        if constexpr (DMA)
            if (lead + slots * 2u < hi) load_tile(next);
        float *pwf = (float *)(smem + (PWT ? POFF : 0u));  // PWT: above the staged samples, so the next staging does not wait for it
        if constexpr (MODE == OUT_MEL) {
            if constexpr (P512) {  // bins 257..267 are read with zero weights
                for (unsigned i = tid; i < 11u * 32u; i += 256u) pwf[pwt512_index(257u + (i >> 5), i & 31u)] = 0.0f;
            } else if constexpr (PWT) {  // bins 513..523 are read with zero weights
                if (tid < 176u) pwf[pwt_index(513u + (tid >> 4), tid & 15u)] = 0.0f;
            } else {
                if (tid < 48u) pwf[(tid / 3u) * kPS + 513u + tid % 3u] = 0.0f;
            }
        }
        if constexpr (P512 && PACK) {
            constexpr unsigned kDrop = 0x80000000u;  // out of the descriptor's range: the store is dropped
            const __amdgpu_buffer_rsrc_t ro = make_rsrc((unsigned char *)a.out + (size_t)b * NB * a.n_frames * ES, min(17u, a.batch - b) * NB * a.n_frames * ES);
            const unsigned c1 = j == 0 ? 16u : j, r2 = j == 0 ? 0u : 32u - j;
            // the slot's signal and pair; offsets count from the tile's first signal b
            const unsigned q = wid * 16u + p2f, bq = q / PP, pr = q - bq * PP;
            const unsigned oA = p2f < nf ? ((bq - b) * NB * a.n_frames + 2u * pr) * ES : kDrop;
            const unsigned oB = (p2f < nf && 2u * pr + 1u < a.n_frames) ? oA + ES : kDrop;
            auto at = [&](unsigned base, unsigned row) { return base == kDrop ? kDrop : base + row * a.n_frames * ES; };
            pass2_pair512<MODE, AMP, true>(A, B, j == 0, eps, ro, at(oA, c1), at(oB, c1), at(oA, r2), at(oB, r2), at(oA, 256u), at(oB, 256u), step,
                                           nullptr, nullptr, nullptr);
        } else if constexpr (P512) {
            constexpr unsigned kDrop = 0x80000000u;  // out of the descriptor's range: the store is dropped
            const __amdgpu_buffer_rsrc_t ro = make_rsrc((unsigned char *)a.out + (size_t)b * NB * a.n_frames * ES, NB * a.n_frames * ES);
            // rows: first loop c1 + 32 i; second loop 32 t (job 0) or 256 - j - 32 t = (32 - j) + 32 (7 - t); bin 256
            const unsigned c1 = j == 0 ? 16u : j, r2 = j == 0 ? 0u : 32u - j;
            const unsigned nv = 2u * p2f + 1u < nf ? 2u : 2u * p2f < nf ? 1u : 0u;  // frames of this slot that exist
            const unsigned o1 = (c1 * a.n_frames + p2ofs) * ES, o2 = (r2 * a.n_frames + p2ofs) * ES, om = (256u * a.n_frames + p2ofs) * ES;
            float *pws = pwf + 4u * p2f;  // this slot's frame pair
            pass2_pair512<MODE, AMP>(A, B, j == 0, eps, ro, nv == 2u ? o1 : kDrop, nv == 1u ? o1 : kDrop, nv == 2u ? o2 : kDrop,
                                     nv == 1u ? o2 : kDrop, nv == 2u ? om : kDrop, nv == 1u ? om : kDrop, step, pws + pwt512_index(c1, 0u),
                                     pws + pwt512_index(r2, 0u), pws + pwt512_index(256u, 0u));
        } else if (ALLSTORE || p2f < nf) {
            const unsigned c1 = j == 0 ? 16u : j, c2 = j == 0 ? 0u : j + 256u;
            const unsigned long long obytes = (unsigned long long)min(PACK ? 17u : 2u, a.batch - p2b) * 513ull * a.n_frames * ES;
            const __amdgpu_buffer_rsrc_t ro = make_rsrc((unsigned char *)a.out + (size_t)p2b * 513u * a.n_frames * ES, (unsigned)obytes);
            {
                auto slot_of = [&](unsigned k) { return pwf + (PWT ? pwt_index(k, p2f) : p2f * kPS + k); };  // bins k + 32 i follow at i * PSTEP
                pass2_compute<MODE, AMP, PWT>(A, B, j == 0, eps, twj, ro, (jo.a1 + p2ofs) * ES, (jo.b1 + p2ofs) * ES, (jo.a2 + p2ofs) * ES,
                                                     (jo.b2 + p2ofs) * ES, (jo.mid + p2ofs) * ES, step, slot_of(c1), slot_of(512u - 224u - c1),
                                                     slot_of(c2), slot_of(512u - 224u - c2), slot_of(256u) SGX_STAMP_ARGS);
            }
        }
        if constexpr (MODE == OUT_MEL) {
            __syncthreads();
            SGX_STAMP(11);  // barrier: |X|^2 tile complete
#ifdef SGX_ABL_NOMEL
            if (a.n_mels == 12345u)
#endif
            if constexpr (P512) mel_tile_sched512<AMP>(a, pwf, sched, b, f0, nf, eps, tid);
            else if constexpr (MFCC) {  // Mel-dB tile to LDS, then its DCT (mfcc_tile)
                mel_tile_sched<AMP, false, mfcc_row<MFCC ? MSTEPS : 4>()>(a, pwf, sched, b, f0, nf, eps, tid, mfl SGX_STAMP_ARGS);
                if constexpr (ROUNDS > 0) {  // the chain waits for the next tile's barrier 1 (or the end of the loop)
                    pvb = b; pvf0 = f0; pvnf = nf;
                } else {  // (the direct path has no barrier 1)
                    __syncthreads();
                    mfcc_tile<MFCC ? MSTEPS : 4>(a, mfl, mfrag, mfcc_mt, b, f0, nf, tid & 63u);
                }
            }
            else if constexpr (PWT) mel_tile_sched<AMP, PACK>(a, pwf, sched, b, f0, nf, eps, tid, nullptr SGX_STAMP_ARGS);
            else if (a.mm_frag) map_tile_mfma<AMP>(a, pwf, b, f0, nf, eps, tid, 2u * half);
            else mel_tile_csr<AMP>(a, pwf, b, f0, nf, eps, tid, 256u);
            if constexpr (!PWT) __syncthreads();  // pw consumed before the next staging overwrites it
        }
        SGX_STAMP(14);  // filterbank stage
        wid = next;
        lead += slots * 2u;
#ifdef SGX_STAMPS
        st_acc[15] += 1;
#endif
    }
    if constexpr (MFCC && ROUNDS > 0) {  // the last tile's chain
        __syncthreads();
        mfcc_tile<MFCC ? MSTEPS : 4>(a, mfl, mfrag, mfcc_mt, pvb, pvf0, pvnf, tid & 63u);
    }
#ifdef SGX_SKEW
    if (MODE == OUT_MEL && half == 0u)
        for (unsigned q = 0; q < SGX_SKEW; ++q) __syncthreads();
#endif
#ifdef SGX_STAMPS
    if ((threadIdx.x & 63u) == 0) {
        for (int q = 0; q < 16; ++q) atomicAdd(&g_stamps[q], st_acc[q]);
        atomicAdd(&g_stamps[16], 1ull);
    }
#endif
}

// Packed tiles pay 5 more vector instructions per sample pair and give up the staged loads and the wide stores — measured 1.0-1.1 G
// frames/s (linear power and Mel-80 dB, 4 ... 40 frames per signal) against 1.4-1.5 G x the filled share of one-signal tiles
// (profiles/bench_r03_short_signals.txt): packed once a quarter or more of the one-signal tiles' slots would be empty (626 frames: 2
// of 640 slots, 40 frames: 8 of 48 — not packed; 17 frames: 15 of 32 — packed)
static bool want_pack(const StftArgs &a, bool mel, bool pwt) {
    if (a.mfcc_frag) return false;  // the fused MFCC epilogue works on one-signal tiles
    const bool p512 = a.n_fft == 512u;  // two frames per transform, 32-frame tiles; per-bin outputs only
    // n_fft 512 at a hop without a staged variant (round 5): the packed form's per-lane loads take any even hop, so it runs whatever the
    // slot fill and for one signal too — 256 x 10 s linear power, hop 100 / 320 / 384: 240 -> 195, 79 -> 66, 72 -> 59 us against k_reg_radix
    const bool unlisted512 = p512 && !(a.hop == 64u || a.hop == 128u || a.hop == 160u || a.hop == 256u);
    if ((a.n_fft != 1024u && !p512) || (a.batch < 2u && !unlisted512) || (mel && (!pwt || p512)) || (a.hop & 1u)) return false;
    // (n_fft 512 packs slots = frame pairs of one signal: an odd frame count leaves half a slot empty either way)
    const unsigned long long slots = (unsigned long long)a.tiles * a.ft, g = (unsigned long long)a.batch * a.n_frames;
    const unsigned long long used = p512 ? 2ull * ((a.n_frames + 1ull) / 2ull) : a.n_frames;
    if ((slots - used) * 4ull < slots && !unlisted512) return false;
    if (g >= 0x7fffffffull || (unsigned long long)a.batch * a.sample_stride * 4ull >= 0xfffffff0ull) return false;  // 32-bit offsets
    return (a.ft + 1ull) * (a.n_fft / 2u + 1ull) * a.n_frames * 8ull < 0x7fffffffull;
}

template <int MODE, int AMP>
hipError_t launch_variant(const StftArgs &a0, hipStream_t s) {
    StftArgs a = a0;
    const bool pack = want_pack(a, MODE == OUT_MEL, MODE == OUT_MEL && a.mel_sched != nullptr);
    const bool pack512 = pack && a.n_fft == 512u;
    a.gframes = pack512 ? a.batch * ((a.n_frames + 1u) / 2u) : a.batch * a.n_frames;  // n_fft 512: slots (frame pairs) of the batch
    a.x_bytes = pack ? (unsigned)((unsigned long long)a.batch * a.sample_stride * 4ull) : 0u;
    const unsigned total = pack ? (a.gframes + 15u) / 16u : a.tiles * a.batch;
    const unsigned per_xcd = (total + 7) / 8;
    const unsigned chunks = (15u * a.hop + 1024u + 3u) >> 2;
    const unsigned pairs = (per_xcd + 1) / 2;
#ifndef SGX_SLOTS
#define SGX_SLOTS 0u  // 0: the device's CU count / 8 (MI355X: 32); a number: A/B runs with fewer workgroups
#endif
    const unsigned cu_slots = SGX_SLOTS ? SGX_SLOTS : std::max(1u, device_cu_count() / 8u);
    const unsigned nslots = pairs < cu_slots ? pairs : cu_slots;  // one 512-thread workgroup per CU
    const bool pwt = MODE == OUT_MEL && a.mel_sched != nullptr;
    auto go = [&](auto kernel, unsigned lds = (unsigned)kLdsBytes) -> hipError_t {
        hipError_t e = set_max_dynamic_lds((const void *)kernel, 163840);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3(nslots * 8), dim3(512), lds, s, a, per_xcd, total, nslots);
        return hipGetLastError();
    };
    constexpr bool W = MODE != OUT_MEL;
    if constexpr (MODE != OUT_MEL)
        if (pack && a.n_fft == 512u) return go(k_r32x16<MODE, AMP, 0, false, false, false, 128, true>);  // (128: any hop, the mode's marker)
    if (pack) return go(k_r32x16<MODE, AMP, 0, false, false, MODE == OUT_MEL, 0, true>);
    if constexpr (MODE == OUT_MEL) {
        if (pwt) {
            if (a.n_fft == 512u) {  // two frames per transform
                if (a.hop == 64u) return go(k_r32x16<MODE, AMP, 3, false, false, true, 64>);
                if (a.hop == 128u) return go(k_r32x16<MODE, AMP, 5, false, false, true, 128>);
                if (a.hop == 160u) return go(k_r32x16<MODE, AMP, 6, false, false, true, 160>);
                if (a.hop == 256u)  // the reference's Mel benchmark shape, benches/spectrogram_benchmarks.rs:105-141 (larger halves: r32x16_layout.h)
                    return go(k_r32x16<MODE, AMP, 9, false, false, true, 256>, 2u * kExBytesH256 + kMelOff + ((a.mel_sched_words * 4u + 15u) & ~15u) + 64u);
                return hipErrorInvalidConfiguration;
            }
            if constexpr (AMP == AMP_DB) {
                if (a.mfcc_frag && a.n_fft == 1024u) {  // Mel-dB -> DCT-II + lifter in the same launch (mfcc_tile)
                    const unsigned lds = (unsigned)kLdsBytes - (unsigned)kMelMaxWords * 4u + ((a.mel_sched_words + 3u) & ~3u) * 4u + a.mfcc_frag_words * 4u + 64u;
                    if (lds > 163840u) return hipErrorInvalidConfiguration;
                    const unsigned ldsz = lds < (unsigned)kLdsBytes ? (unsigned)kLdsBytes : lds;
                    auto mf = [&](auto steps) -> hipError_t {
                        constexpr int S = decltype(steps)::value;
                        if (a.hop == 256u) return go(k_r32x16<MODE, AMP, 5, false, true, true, 0, false, S>, ldsz);
                        if (chunks <= 5u * 256u) return go(k_r32x16<MODE, AMP, 5, false, false, true, 0, false, S>, ldsz);
                        return go(k_r32x16<MODE, AMP, 0, false, false, true, 0, false, S>, ldsz);
                    };
                    if (a.mfcc_steps == 12u) return mf(std::integral_constant<int, 12>{});  // (the host pads the fragment table to a menu length:
                    if (a.mfcc_steps == 16u) return mf(std::integral_constant<int, 16>{});  //  zero weights)
                    if (a.mfcc_steps == 20u) return mf(std::integral_constant<int, 20>{});
                    if (a.mfcc_steps == 24u) return mf(std::integral_constant<int, 24>{});
                    return hipErrorInvalidConfiguration;
                }
            }
            if (a.hop == 256u) return go(k_r32x16<MODE, AMP, 5, false, true, true>);
            if (chunks <= 5u * 256u) return go(k_r32x16<MODE, AMP, 5, false, false, true>);
            return go(k_r32x16<MODE, AMP, 0, false, false, true>);
        }
    }
    if constexpr (MODE != OUT_MEL) {
        if (a.n_fft == 512u) {  // two frames per transform
            if (a.hop == 64u) return go(k_r32x16<MODE, AMP, 3, false, false, false, 64>);
            if (a.hop == 128u) return go(k_r32x16<MODE, AMP, 5, false, false, false, 128>);
            if (a.hop == 160u) return go(k_r32x16<MODE, AMP, 6, false, false, false, 160>);
            if (a.hop == 256u) return go(k_r32x16<MODE, AMP, 9, false, false, false, 256>);  // (per-bin outputs only: 37 KB of staged samples)
            return hipErrorInvalidConfiguration;
        }
    }
    if (a.n_fft != 1024u) return hipErrorInvalidConfiguration;
    if (a.hop == 256u) return go(k_r32x16<MODE, AMP, 5, W, true, false>);
    if (chunks <= 5u * 256u) return go(k_r32x16<MODE, AMP, 5, W, false, false>);
    return go(k_r32x16<MODE, AMP, 0, W, false, false>);
}

}  // namespace

#ifdef SGX_STAMPS
extern "C" int sgx_debug_read_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

bool plan_geometry_r32x16_f32(StftArgs &a) {
    // P512: two frames per transform, 32-frame tiles; filterbank outputs need the band schedule (built on the host at plan creation,
    // before this is asked: a bank without one resolves to the register-tiled kernel)
    if (a.n_fft == 512 && a.hop == 256 && a.out_mode == OUT_MEL && a.mel_sched_words > (unsigned)kMelMaxWordsH256) return false;  // (larger halves, shorter schedule)
    if (a.n_fft == 512 && a.out_mode == OUT_MEL && a.mel_sched_words == 0) return false;
    if (a.n_fft == 512 && (a.hop == 64 || a.hop == 128 || a.hop == 160 || a.hop == 256)) {
        if (a.n_samples >= (1ull << 29) || (unsigned long long)a.n_frames * 257ull * 8ull >= 0x7fffffffull) return false;
        a.ft = 32;
        return true;
    }
    if (a.n_fft == 512) {  // any other even hop, per-bin outputs: the packed form (want_pack's limits)
        if (a.out_mode == OUT_MEL || (a.hop & 1u) || a.hop > 512u) return false;
        if (a.n_samples >= (1ull << 29) || (unsigned long long)a.batch * a.n_frames >= 0x7fffffffull ||
            (unsigned long long)a.batch * a.sample_stride * 4ull >= 0xfffffff0ull || 33ull * 257ull * a.n_frames * 8ull >= 0x7fffffffull)
            return false;
        a.ft = 32;
        return true;
    }
    if (a.n_fft != 1024 || (!SGX_ODDHOP && (a.hop & 1u))) return false;
    if (a.n_samples >= (1ull << 29)) return false;                                       // 32-bit byte offsets into a sample row
    if ((unsigned long long)a.n_frames * 513ull * 8ull >= 0x7fffffffull) return false;  // and into a pair of output signals
    a.ft = 16;
    return true;
}

hipError_t launch_r32x16_f32(const StftArgs &a, hipStream_t s) {
    const unsigned long long total64 = (unsigned long long)a.tiles * a.batch;
    if (total64 == 0 || total64 >= 0x7ffffff0ull) return hipErrorInvalidConfiguration;
    if (a.out_mode == OUT_COMPLEX) return launch_variant<OUT_COMPLEX, AMP_POWER>(a, s);
    if (a.out_mode == OUT_MEL) {
        if (a.amp == AMP_MAGNITUDE) return launch_variant<OUT_MEL, AMP_MAGNITUDE>(a, s);
        if (a.amp == AMP_DB) return launch_variant<OUT_MEL, AMP_DB>(a, s);
        if (a.amp == AMP_MAG_IN) return launch_variant<OUT_MEL, AMP_MAG_IN>(a, s);
        return launch_variant<OUT_MEL, AMP_POWER>(a, s);
    }
    if (a.amp == AMP_MAGNITUDE) return launch_variant<OUT_LINEAR, AMP_MAGNITUDE>(a, s);
    if (a.amp == AMP_DB) return launch_variant<OUT_LINEAR, AMP_DB>(a, s);
    return launch_variant<OUT_LINEAR, AMP_POWER>(a, s);
}

}  // namespace sgx
