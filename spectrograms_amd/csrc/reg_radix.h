// reg_radix.h — shared pieces of the register-tiled transforms (k_reg_radix: STFT frames; k_c2c_reg / k_c2r_reg: the 2-D
// path and the generic inverse row transform): pass-length selection, occupancy targets, twiddle construction and the
// list of instantiated (A, B, C) splits.
#pragma once
#include "fft_inreg.h"
#include "sgx_internal.h"

#ifndef SGX_RRW32
#define SGX_RRW32 3  // waves per SIMD the register allocation of the f32 / f64 instances aims at
#endif
#ifndef SGX_RRW64
#define SGX_RRW64 2
#endif
#ifndef SGX_RR_NI2_MIN
#define SGX_RR_NI2_MIN 32  // B C from which a thread takes two pass-1 work items (twice the frames per tile)
#endif

namespace sgx {

template <typename T> struct PairOf;
template <> struct PairOf<float> { typedef inreg::v2f type; };
template <> struct PairOf<double> { typedef inreg::v2d type; };

// product of the table entries selected by the bits of k: p[j] = W^(2^j r)  ->  W^(k r)
template <int L, typename V>
__device__ __forceinline__ V rr_twiddle(const V (&p)[L], unsigned k) {
    V t = p[0];
    bool have = false;
#pragma unroll
    for (int j = 0; j < L; ++j)
        if (k >> j & 1u) {
            t = have ? inreg::cmulv(t, p[j]) : p[j];
            have = true;
        }
    return t;
}

constexpr unsigned ct_log2_ceil(unsigned n) { unsigned l = 0; while ((1u << l) < n) ++l; return l; }
constexpr bool ct_is_pow2(unsigned n) { return n && !(n & (n - 1)); }

// pass-1 work items per thread: the tile holds up to 256 * NI / (B C) frames (wider store segments for the long transforms)
template <int B, int C>
constexpr unsigned rr_items() { return (ct_is_pow2(B * C) && B * C >= SGX_RR_NI2_MIN) ? 2 : 1; }

// Sample staging (k_reg_radix): a tile's samples travel global -> registers (16-byte chunks, chunk c = tid + 256 i, one tile
// ahead) -> LDS, where the pass-1 work items pick their points.  Rounds of 256 chunks a thread holds in registers: 24 KiB of
// samples, 32 KiB for the transforms whose single frame needs that much.
#ifndef SGX_RR_STAGED
#define SGX_RR_STAGED 1
#endif
constexpr unsigned rr_stage_rounds(unsigned m, unsigned elem) { return 2u * m * elem >= 16384u ? 8u : 6u; }
// Instances that have a staged variant (the host picks it for per-bin outputs: measured faster there — f32 n_fft 512 201 -> 180,
// 2048 276 -> 244 us — and slower for filterbank outputs, whose kernels wait for no stores, and for f64 below n_fft 2048,
// where two pass-1 items per thread held across a barrier spill)
#ifndef SGX_RR_STAGE_F64
#define SGX_RR_STAGE_F64 0  // experiment: staged samples for every f64 transform
#endif
template <typename T, int A, int C>
constexpr bool rr_can_stage() { return SGX_RR_STAGED && (sizeof(T) == 4 || SGX_RR_STAGE_F64 || (A >= 16 && C > 1)); }

// k_reg_radix: waves per SIMD (= resident workgroups per CU) the register allocation aims at.  With the twiddle products rebuilt
// per tile the two-pass f32 instances up to 16-point passes stay under 128 registers.
// per tile the two-pass f32 instances up to 16-point passes stay under 128 registers; the longer ones do with staged samples
// (the direct variant carries a frame's A sample pairs per thread across the whole tile)
template <typename T, int A, int B, int C, bool STAGED>
constexpr unsigned rr_stft_waves() {
    if (sizeof(T) == 8) return A >= 16 ? 1 : SGX_RRW64;
    if (C > 1) return A <= 8 ? 3 : 2;
    return A <= 16 ? 4 : (STAGED && A <= 30) ? 3 : 2;
}

// waves per SIMD (= resident workgroups per CU) the register allocation of an instance aims at: the largest transforms
// (16-point and longer passes in f64, the three-pass f32 sizes with a 16-point first pass) would spill at the default
template <typename T, int A, int B, int C>
constexpr unsigned rr_waves() {
    if (sizeof(T) == 8) return A >= 16 ? 1 : SGX_RRW64;
    return ((C > 1 && A >= 16) || A > 16) ? 2 : SGX_RRW32;
}

// Pass lengths for a complex transform of length `len` (the STFT kernels pass len = n_fft / 2).  Powers of two 16 .. 4096:
// in-register transforms up to 16 points in f32 (two passes up to len 256, three above), up to 8 points in f64 where three
// passes reach (len <= 512) — a 16-point f64 pass with its samples and twiddles in flight exceeds 256 registers.  Other
// lengths: the list below (two passes, factors 2, 3, 5): as n_fft = 2 len it covers the usual speech / audio frames —
// 10, 20, 25, 30, 40, 50 ms at 8 / 16 / 32 / 48 kHz and their neighbours.
struct RegSplit { unsigned len, a, b; };
static const RegSplit kMixedSplits[] = {
    {40, 8, 5},     // n_fft 80   (10 ms @ 8 kHz)
    {60, 10, 6},    // 120
    {80, 10, 8},    // 160  (10 ms @ 16 kHz, 20 ms @ 8 kHz)
    {100, 10, 10},  // 200  (25 ms @ 8 kHz)
    {120, 12, 10},  // 240  (30 ms @ 8 kHz, 15 ms @ 16 kHz)
    {160, 16, 10},  // 320  (20 ms @ 16 kHz)
    {200, 25, 8},   // 400  (25 ms @ 16 kHz)
    {240, 16, 15},  // 480  (30 ms @ 16 kHz, 10 ms @ 48 kHz)
    {300, 20, 15},  // 600
    {320, 20, 16},  // 640  (40 ms @ 16 kHz, 20 ms @ 32 kHz)
    {400, 25, 16},  // 800  (50 ms @ 16 kHz, 25 ms @ 32 kHz)
    {480, 24, 20},  // 960  (20 ms @ 48 kHz)
    {500, 25, 20},  // 1000
    {600, 25, 24},  // 1200 (25 ms @ 48 kHz)
    {720, 30, 24},  // 1440 (30 ms @ 48 kHz)
    {800, 32, 25},  // 1600 (50 ms @ 32 kHz)
    {960, 32, 30},  // 1920 (40 ms @ 48 kHz)
    // round numbers as image sides / column lengths of the 2-D path (and n_fft 2000, 2400, 2560): a 40-point second pass
    {640, 16, 40},   // (rows of 1280-wide images: 720p)
    {1000, 25, 40},
    {1080, 30, 36},  // (columns of 1920 x 1080 images)
    {1200, 30, 40},
    {1280, 32, 40},
};

inline bool reg_split_len(unsigned len, int dtype, unsigned *pa, unsigned *pb, unsigned *pc) {
    if (len < 16) return false;
    if ((len & (len - 1)) == 0) {
        unsigned l2 = 0;
        while ((1u << l2) < len) ++l2;
        if (l2 > 12) return false;
        // (f64 two-pass splits up to 32 x 32 at one wave per SIMD were measured in round 2: n_fft 1024 938 vs 628 us — three
        // short passes at two waves per SIMD win)
        const unsigned two_pass_max = dtype == SGX_F64 ? 6 : 8;
        unsigned la, lb, lc;
        if (l2 <= two_pass_max) {
            la = (l2 + 1) / 2; lb = l2 / 2; lc = 0;
        } else {
            la = (l2 + 2) / 3; lb = (l2 + 1) / 3; lc = l2 / 3;
        }
        *pa = 1u << la; *pb = 1u << lb; *pc = 1u << lc;
        return true;
    }
    for (const RegSplit &r : kMixedSplits)
        if (r.len == len) {
            *pa = r.a; *pb = r.b; *pc = 1;
            return true;
        }
    return false;
}

// every (A, B, C) reg_split_len can return, as X-macro lists (one kernel instance per entry and element type)
#define SGX_RR_SPLITS_F32(X) \
    X(4, 4, 1) X(8, 4, 1) X(8, 8, 1) X(16, 8, 1) X(16, 16, 1) X(8, 8, 8) X(16, 8, 8) X(16, 16, 8) X(16, 16, 16)
#define SGX_RR_SPLITS_F64(X) \
    X(4, 4, 1) X(8, 4, 1) X(8, 8, 1) X(8, 4, 4) X(8, 8, 4) X(8, 8, 8) X(16, 8, 8) X(16, 16, 8) X(16, 16, 16)
#define SGX_RR_SPLITS_MIXED(X)                                                                                          \
    X(8, 5, 1) X(10, 6, 1) X(10, 8, 1) X(10, 10, 1) X(12, 10, 1) X(16, 10, 1) X(25, 8, 1) X(16, 15, 1) X(20, 15, 1)       \
    X(20, 16, 1) X(25, 16, 1) X(24, 20, 1) X(25, 20, 1) X(25, 24, 1) X(30, 24, 1) X(32, 25, 1) X(32, 30, 1)       \
    X(25, 40, 1) X(30, 40, 1) X(32, 40, 1) X(16, 40, 1) X(30, 36, 1)

}  // namespace sgx
