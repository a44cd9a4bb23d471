// kernels_c2c1024.hip — tuned f32 1024-point complex FFT for the column passes of the 2-D path (BASELINE config 5:
// 1024 x 1024 images).  Same construction as the STFT kernel: 1024 = 32 x 32, two 32-point FFTs entirely in registers
// (fft_inreg.h, packed-f32 math) around ONE LDS exchange, 16 sequences per 512-thread workgroup.
//
//   pass 1  lane (s, n2) owns z[32*n1 + n2], n1 = 0..31, of sequence s; FFT32 over n1; twiddle W_1024^(k1*n2) from two
//           short per-lane register tables; ds_write_b64 into ex[s][k1][n2] (sequence stride 8192+16 B)
//   pass 2  lane (k1, s) reads row k1 of sequence s with 16 conflict-free ds_read_b128, FFT32 over n2 -> X[k1 + 32*k2];
//           the 16 lanes of a row hold 16 consecutive sequences, so out[(k1+32*k2)][s0..s0+15] is one 128-byte segment
//           when the output is sequence-contiguous ([r][k] layout: both column passes of the 2-D path).
// The load side follows whichever input stride is 1 (IN_SEQ_FAST).  Inverse = conj(FFT(conj(.))).
#include <algorithm>
#include <cstdlib>
#include "buffer_ops.h"
#include "fft_inreg.h"
#include "sgx_internal.h"
#include "xcd_map.h"

namespace sgx {
namespace {

using namespace inreg;
typedef float v4f __attribute__((ext_vector_type(4)));


constexpr int kCFS = 8192 + 16;     // LDS bytes per sequence (odd multiple of 16: conflict-free b128 row reads)
constexpr int kCLds = 16 * kCFS;    // 131328 B -> one workgroup per CU

// NS sequences per workgroup of 32 NS threads: 16 -> 131 KB, one workgroup of 8 waves per CU; 8 -> 66 KB, two independent
// workgroups per CU whose load / transform / store phases could overlap, but with 64-byte instead of 128-byte segments on both
// sides — measured slower (config 5 fft2d 1.78 vs 1.63 ms), so 16 it is
#ifndef SGX_C2C1024_NS
#define SGX_C2C1024_NS 16
#endif
template <bool IN_SEQ_FAST, bool INVERSE, unsigned NS>
__global__ __launch_bounds__(32 * NS, 2) void k_c2c1024(C2cArgs a, const v2f *tw1c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x;
    const unsigned lb = xcd_logical_block(a.tiles * a.batch);
    if (lb >= a.tiles * a.batch) return;
    const unsigned t = lb % a.tiles, b = lb / a.tiles;
    const unsigned s0 = t * NS;
    const v2f *in = (const v2f *)a.in + (size_t)b * a.in_img;
    v2f *out = (v2f *)a.out + (size_t)b * a.out_img;
    // ------------------------------------------------------------------ pass 1
    {
        const unsigned s = IN_SEQ_FAST ? (tid & (NS - 1u)) : (tid >> 5), n2 = IN_SEQ_FAST ? (tid / NS) : (tid & 31u);
        const bool valid = s0 + s < a.nseq;
        v2f v[32];
        const v2f *p = in + (size_t)(s0 + s) * a.in_ss + (size_t)n2 * a.in_is;
#pragma unroll
        for (int n1 = 0; n1 < 32; ++n1) {
            v2f x = valid ? p[(size_t)(32 * n1) * a.in_is] : (v2f){0.f, 0.f};
            if (INVERSE) x.y = -x.y;
            v[n1] = x;
        }
        Fft<32, false>::run(v, v);
        v2f twa[4], twb[8];  // W_1024^(k1*n2) = twa[k1>>3] * twb[k1&7]
#pragma unroll
        for (int q = 0; q < 4; ++q) twa[q] = tw1c[(8 * q) * 32 + n2];
#pragma unroll
        for (int q = 0; q < 8; ++q) twb[q] = tw1c[q * 32 + n2];
        unsigned char *dst = smem + s * kCFS + n2 * 8;
#pragma unroll
        for (int k1 = 0; k1 < 32; ++k1) {
            const int qa = k1 >> 3, qb = k1 & 7;
            v2f r = v[k1];
            if (qb) r = cmulv(r, twb[qb]);
            if (qa) r = cmulv(r, twa[qa]);
            *(v2f *)(dst + k1 * 256) = r;
        }
    }
    __syncthreads();
    // ------------------------------------------------------------------ pass 2
    {
        const unsigned w = tid >> 6, l = tid & 63u, jq = l / NS, s = l & (NS - 1u);
        const unsigned k1 = w * (64u / NS) + jq;
        v2f x[32];
        const v4f *row = (const v4f *)(smem + s * kCFS + k1 * 256);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const v4f q = row[c];
            x[2 * c] = (v2f){q.x, q.y};
            x[2 * c + 1] = (v2f){q.z, q.w};
        }
        Fft<32, false>::run(x, x);
        if (s0 + s < a.nseq) {
            const float sc = (float)a.scale;
            v2f *o = out + (size_t)(s0 + s) * a.out_ss + (size_t)k1 * a.out_is;
#pragma unroll
            for (int k2 = 0; k2 < 32; ++k2) {
                v2f r = x[k2] * (v2f){sc, sc};
                if (INVERSE) r.y = -r.y;
                o[(size_t)(32 * k2) * a.out_is] = r;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_colconv1024: the whole column stage of convolve_fft / the radial filters (src/image_ops.rs:80-115, 301-432) for 1024
// rows in ONE pass over the half spectrum: forward 1024-point FFT of 16 columns, product with the kernel spectrum (or a real
// mask), inverse 1024-point FFT — the spectrum never goes back to HBM between the three steps (3 launches and 25 MB per
// image before; 8.4 MB now).
//   forward   as k_c2c1024: k = k1 + 32 k2 ends up in registers of thread (k1, s), k2 = 0..31
//   product   X[k] *= K[k][col] (lanes walk the 16 columns: 128-byte segments of the [row][col] kernel spectrum / mask)
//             or, for a rank-1 kernel (MUL_OUTER, round 5), K[k][col] = U[k] V[col] from 12 KB of factors: the 4.2 MB spectrum is
//             neither built nor read once per image
//   inverse   y = conj(FFT(conj(X))) with the index split mirrored: FFT32 over k2 IN REGISTERS -> n_b, twiddle
//             W_1024^(k1 n_b), second LDS exchange (same buffer), FFT32 over k1 -> n = 32 n_a + n_b
//   store     out[n][col] (the [row][col] layout k_c2r1024 reads), 128-byte segments; unnormalised (C2R applies 1/(R C))
// Persistent (round 2): one workgroup per CU walks its XCD's contiguous run of tiles; the next tile's 32 column samples per
// thread are requested as soon as this tile's product with the kernel spectrum is done (the registers that held them are free
// since pass 1) and collected at the top of the next tile, the stores of a tile drain under the next tile's first pass.  Loads
// and stores go through buffer descriptors (a column past the image loads zeros and stores nothing), so every memory
// instruction is unconditional and the compiler can count the stores behind the prefetch (`vmcnt(32)`, not `vmcnt(0)`).  The
// twiddle table sits in LDS.  Before: one tile per workgroup, every phase's latency exposed — 19.5 us per tile against 8.4 us
// for its 262 KB at a CU's share of the HBM.
constexpr int kCTwOff = kCLds;              // W_1024^(k1 n2), 32 x 32 complex f32
constexpr int kCLdsP = kCLds + 32 * 32 * 8;  // 139 520 B
template <int MUL, bool REAL_IO = false>
__global__ __launch_bounds__(512, 2) void k_colconv1024(C2cArgs a, const v2f *tw1c, const void *mul, unsigned long long mul_row, unsigned per_xcd,
                                                        unsigned total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x;
    v2f *tws = (v2f *)(smem + kCTwOff);
    tws[tid] = tw1c[tid];
    tws[tid + 512u] = tw1c[tid + 512u];
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const unsigned lo = xcd * per_xcd, hi = min(lo + per_xcd, total);
    // pass-1 identity: lane (s1, n2), element index contiguous in memory; pass-2 identity: row k1 of sequence s
    const unsigned s1 = tid >> 5, n2 = tid & 31u;
    const unsigned w = tid >> 6, l = tid & 63u, jq = l >> 4, s = l & 15u;
    const unsigned k1 = w * 4u + jq;  // forward: row k1; inverse step B: row n_b
    // (REAL_IO: strides and image sizes count f32 elements of real arrays)
    const unsigned in_bytes = (unsigned)a.in_img * (REAL_IO ? 4u : 8u), out_bytes = (unsigned)a.out_img * (REAL_IO ? 4u : 8u);  // host: < 2^31
    const int istep = (int)(32u * (unsigned)a.in_is * 8u);
    constexpr unsigned kOob = 0x80000000u;
    v2f v[32];
    auto load_tile = [&](unsigned wt, bool live) {  // !live (past the run): the same instructions, out of range — no branch around them
        const unsigned t = wt % a.tiles, b = live ? wt / a.tiles : 0u, s0 = t * 16u;
        if constexpr (REAL_IO) {
            // sequence s = the rows 2 s and 2 s + 1 of a real [rows][1024] array (pitch a.in_ss) as ONE complex sequence z = row_a + i row_b:
            // a real kernel convolves both at once (its spectrum is Hermitian: the result's real part is row_a's, the imaginary row_b's)
            // Loaded 8 bytes at a time: the even lane of a pair takes (row_a[c], row_a[c + 1]), the odd lane (row_b[c], row_b[c + 1]), c = its
            // even column; the trade at the top of the tile loop makes lane n2 hold z[n2 + 32 n1].  (The first form — two 4-byte loads per
            // element — gave wrong tiles whenever a workgroup ran more than one tile; four experiments did not cure it: at most 64
            // operations in flight, immediate instead of scalar offsets, `s_waitcnt vmcnt(0)` at the top of every tile, the same tied to
            // the loaded registers.  Cause not found, form not used: DESIGN.md §4.)
            const __amdgpu_buffer_rsrc_t ri = make_rsrc((const float *)a.in + (size_t)b * a.in_img, in_bytes);
            const unsigned va = live && s0 + s1 < a.nseq ? ((2u * (s0 + s1) + (n2 & 1u)) * (unsigned)a.in_ss + (n2 & ~1u)) * 4u : kOob;
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) v[n1] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(ri, (int)va, n1 * 128, 0));
        } else {
        const __amdgpu_buffer_rsrc_t ri = make_rsrc((const v2f *)a.in + (size_t)b * a.in_img, in_bytes);
        const unsigned vo = live && s0 + s1 < a.nseq ? ((s0 + s1) * (unsigned)a.in_ss + n2 * (unsigned)a.in_is) * 8u : kOob;
#pragma unroll
        for (int n1 = 0; n1 < 32; ++n1) v[n1] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(ri, (int)vo, n1 * istep, 0));
        }
    };
    unsigned wt = lo + slot;
    load_tile(wt, wt < hi);
    // the first tile has landed before the loop: with loads still pending on entry their first use inside the loop gets waits
    // that from the second tile on wait for the previous tile's stores
#pragma unroll
    for (int n1 = 0; n1 < 32; ++n1) asm volatile("" : "+v"(v[n1]));
    __syncthreads();  // twiddles visible
    while (wt < hi) {
        const unsigned t = wt % a.tiles, b = wt / a.tiles, s0 = t * 16u;
        if constexpr (REAL_IO) {  // (row_a[c], row_a[c + 1]) | (row_b[c], row_b[c + 1]) in lanes 2 j | 2 j + 1  ->  z[c] | z[c + 1], z = row_a + i row_b
            const bool odd = n2 & 1u;
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) {
                const float send = odd ? v[n1].x : v[n1].y;
                const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, false));  // quad_perm [1, 0, 3, 2]
                v[n1] = odd ? (v2f){recv, v[n1].y} : (v2f){v[n1].x, recv};
            }
        }
        {   // forward pass 1
            Fft<32, false>::run(v, v);
            unsigned char *dst = smem + s1 * kCFS + n2 * 8;
#pragma unroll
            for (int q1 = 0; q1 < 32; ++q1) {
                const int qa = q1 >> 3, qb = q1 & 7;
                v2f r = v[q1];
                if (qb) r = cmulv(r, tws[qb * 32 + n2]);
                if (qa) r = cmulv(r, tws[(8 * qa) * 32 + n2]);
                *(v2f *)(dst + q1 * 256) = r;
            }
        }
        __syncthreads();
        const bool valid = s0 + s < a.nseq;
        v2f x[32];
        {
            const v4f *row = (const v4f *)(smem + s * kCFS + k1 * 256);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const v4f q = row[c];
                x[2 * c] = (v2f){q.x, q.y};
                x[2 * c + 1] = (v2f){q.z, q.w};
            }
        }
        __syncthreads();  // every row has been read: the buffer is free for the second exchange
        Fft<32, false>::run(x, x);  // X[k1 + 32 k2]
        {   // product with the kernel spectrum / mask (a column past the image reads zeros: its lanes hold zeros anyway)
            constexpr bool REAL_MASK = MUL == MUL_MASK;
            constexpr bool VEC = MUL == MUL_OUTER || MUL == MUL_VEC;  // a 1024-entry table indexed by k alone
            const unsigned mrow = VEC ? 8u : (unsigned)mul_row * (REAL_MASK ? 4u : 8u);
            const __amdgpu_buffer_rsrc_t rm = make_rsrc(mul, MUL == MUL_OUTER ? (1024u + a.nseq) * 8u : 1024u * mrow);
            const unsigned mo = VEC ? k1 * 8u : valid ? k1 * mrow + (s0 + s) * (REAL_MASK ? 4u : 8u) : kOob;
            v2f vc = {0.f, 0.f};  // MUL_OUTER: the column's factor V[col], one load per tile; the row factors U[k] come from an 8 KB table
            if constexpr (MUL == MUL_OUTER)
                vc = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rm, (int)(valid ? (1024u + s0 + s) * 8u : kOob), 0, 0));
#pragma unroll
            for (int k2 = 0; k2 < 32; ++k2) {
                if constexpr (MUL == MUL_VEC) {  // one factor of a rank-1 kernel's spectrum, the same for every sequence (the separable passes)
                    const v2f r = cmulv(x[k2], __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rm, (int)mo, k2 * 32 * 8, 0)));
                    x[k2] = (v2f){r.x, -r.y};
                } else if constexpr (MUL == MUL_OUTER) {
                    const v2f u = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rm, (int)mo, k2 * 32 * 8, 0));
                    const v2f r = cmulv(x[k2], cmulv(u, vc));  // K[k][col] = U[k] V[col], rounded to f32 like a stored spectrum
                    x[k2] = (v2f){r.x, -r.y};
                } else if constexpr (REAL_MASK) {
                    const float m = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, (int)mo, k2 * 32 * (int)mrow, 0));
                    x[k2] = x[k2] * (v2f){m, -m};  // product, then conj for the forward-FFT inverse trick
                } else {
                    const v2f r = cmulv(x[k2], __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rm, (int)mo, k2 * 32 * (int)mrow, 0)));
                    x[k2] = (v2f){r.x, -r.y};
                }
            }
        }
        const unsigned next = wt + slots;
        __builtin_amdgcn_sched_barrier(0);  // (the request stays behind the product's loads: they are waited for first, in order)
        load_tile(next, next < hi);  // in flight during the inverse transform and this tile's stores
        __builtin_amdgcn_sched_barrier(0);
        Fft<32, false>::run(x, x);  // over k2 -> n_b
        {
            unsigned char *dst = smem + s * kCFS + k1 * 8;
#pragma unroll
            for (int nb = 0; nb < 32; ++nb) {
                const int qa = nb >> 3, qb = nb & 7;  // W_1024^(k1 n_b) (the table is symmetric in its two indices)
                v2f r = x[nb];
                if (qb) r = cmulv(r, tws[qb * 32 + k1]);
                if (qa) r = cmulv(r, tws[(8 * qa) * 32 + k1]);
                *(v2f *)(dst + nb * 256) = r;
            }
        }
        __syncthreads();
        {
            const v4f *row = (const v4f *)(smem + s * kCFS + k1 * 256);  // row n_b = k1 of this thread: 32 values over the old k1
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const v4f q = row[c];
                x[2 * c] = (v2f){q.x, q.y};
                x[2 * c + 1] = (v2f){q.z, q.w};
            }
        }
        __syncthreads();  // rows read: the next tile's pass 1 may overwrite the buffer while this tile finishes
        Fft<32, false>::run(x, x);  // over k1 -> n_a: Y[32 n_a + n_b]
        {
            // (REAL_IO: element n of the pair (rows 2 q, 2 q + 1) goes to out[n][2 q .. 2 q + 1] — the transposed real array, the pair side
            // by side again; 16 lanes write the 128 bytes of 32 consecutive input rows)
            const __amdgpu_buffer_rsrc_t ro = REAL_IO ? make_rsrc((float *)a.out + (size_t)b * a.out_img, out_bytes)
                                                      : make_rsrc((v2f *)a.out + (size_t)b * a.out_img, out_bytes);
            const unsigned oo = !valid ? kOob : REAL_IO ? (k1 * (unsigned)a.out_is + 2u * (s0 + s)) * 4u
                                                        : ((s0 + s) * (unsigned)a.out_ss + k1 * (unsigned)a.out_is) * 8u;
            const int ostep = (int)(32u * (unsigned)a.out_is * (REAL_IO ? 4u : 8u));
#pragma unroll
            for (int na = 0; na < 32; ++na) {
                typedef unsigned U2 __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(U2, (v2f){x[na].x, -x[na].y}), ro, (int)oo, na * ostep, 0);
            }
        }
        wt = next;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_c2r1024: tuned f32 inverse row pass, half spectrum [r][k] (513 complex, k contiguous) -> 1024 real samples per row.
// With z[n] = x[2n] + i x[2n+1] and Y = X[512-k]:  Z'[k] = (X[k] + conj Y) + i conj(W_1024^k) (X[k] - conj Y)  (= 2 Z[k]),
// 1024 x = IDFT_512(Z') = conj(DFT_512(conj Z')).  The 512-point DFT is the STFT kernel's 32 x 16 split: lane (r, n2)
// owns k = 16*n1 + n2 (both members X[k], X[512-k] of a pair are plain row loads), FFT32 over n1, twiddle W_512^(k1*n2),
// LDS exchange (row stride 128+16 B), FFT16 over n2 -> sample index n = k1 + 32*k2; lanes walk k1, so every store is a
// 256-byte contiguous run of the output row.  DC / Nyquist columns forced real on load (fft_backend.rs:782-793).
constexpr int kRRS = 128 + 16;        // LDS bytes per k1-row (16 complex + pad: conflict-free b128 reads with lanes over k1)
constexpr int kRSeq = 32 * kRRS;      // 4608 B per image row
constexpr int kRLds = 16 * kRSeq;     // 73728 B -> two workgroups per CU

__global__ __launch_bounds__(256, 2) void k_c2r1024(C2rArgs a, const v2f *twr /*[32][16] conj(W_1024^(16 n1 + n2))*/,
                                                     const v2f *tw1 /*[32][16] W_512^(k1 n2)*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x;
    const unsigned t = blockIdx.x % a.tiles, b = blockIdx.x / a.tiles;
    const unsigned r0 = t * 16u;
    const v2f *in = (const v2f *)a.in + (size_t)b * a.in_img;
    float *out = (float *)a.out + (size_t)b * a.nrows * 1024u;
    {
        const unsigned r = tid >> 4, n2 = tid & 15u;
        const bool valid = r0 + r < a.nrows;
        const v2f *row = in + (size_t)(r0 + r) * a.in_rs;  // in_ks == 1
        v2f v[32];
        const v2f wl = twr[n2];  // conj(W_1024^(16 n1 + n2)) = e^{+2 pi i n1 / 64} (constant) * conj(W_1024^n2) (this load)
        // Every element of the row is read twice: as X[k] by lane k and as X[512 - k] by lane 512 - k.  Issued in the order of the fold
        // the two reads of a line are 32 - 2 n1 instructions apart and a third of the second ones reached the fabric again
        // (FETCH_SIZE 5.5 MB per image for 4.2); issued as neighbours — X[16 j + n2] next to X[16 (j + 1) - n2], the same 128 bytes
        // shifted by one element — the second read meets the first in the L1.
        v2f XA[32], XY[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            XA[j] = valid ? row[16u * j + n2] : (v2f){0.f, 0.f};
#ifdef SGX_C2R_FOLD_ORDER  // (A/B: the loads in the order of the fold)
            XY[j] = valid ? row[512u - (16u * j + n2)] : (v2f){0.f, 0.f};
#else
            XY[31 - j] = valid ? row[16u * (j + 1) - n2] : (v2f){0.f, 0.f};  // = row[512 - k] of n1 = 31 - j
#endif
        }
#pragma unroll
        for (int n1 = 0; n1 < 32; ++n1) {
            const unsigned k = 16u * n1 + n2;
            v2f A = XA[n1];
            v2f Y = XY[n1];
            if (k == 0) { A.y = 0.f; Y.y = 0.f; }  // DC (k = 0) and Nyquist (512 - 0) columns are forced real
            const v2f B = (v2f){Y.x, -Y.y};        // conj(X[512-k])
            const v2f S = A + B, D = A - B;
            const v2f cw = n1 == 0 ? wl : cmulv(wl, (v2f){(float)kCos64[n1], (float)kSin64[n1]});
            const v2f T = cmulv(D, cw);                           // conj(W^k) (X[k] - conj Y)
            const v2f Z = pfma(swp(T), (v2f){-1.f, 1.f}, S);      // S + i T
            v[n1] = (v2f){Z.x, -Z.y};                             // conj for the forward-FFT inverse trick
        }
        Fft<32, false>::run(v, v);
        unsigned char *dst = smem + r * kRSeq + n2 * 8;
        *(v2f *)dst = v[0];
        v2f twa[4], twb[8];  // W_512^(k1 n2) = twa[k1 >> 3] * twb[k1 & 7]
#pragma unroll
        for (int q = 0; q < 4; ++q) twa[q] = tw1[16 * 8 * q + n2];
#pragma unroll
        for (int q = 0; q < 8; ++q) twb[q] = tw1[16 * q + n2];
#pragma unroll
        for (int k1 = 1; k1 < 32; ++k1) {
            const int qa = k1 >> 3, qb = k1 & 7;
            v2f r2 = v[k1];
            if (qb) r2 = cmulv(r2, twb[qb]);
            if (qa) r2 = cmulv(r2, twa[qa]);
            *(v2f *)(dst + k1 * kRRS) = r2;
        }
    }
    __syncthreads();
    const float sc = (float)a.scale;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const unsigned r = (tid >> 5) + 8u * it, k1 = tid & 31u;
        v2f x[16];
        const v4f *rowp = (const v4f *)(smem + r * kRSeq + k1 * kRRS);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const v4f q = rowp[c];
            x[2 * c] = (v2f){q.x, q.y};
            x[2 * c + 1] = (v2f){q.z, q.w};
        }
        Fft<16, false>::run(x, x);
        if (r0 + r < a.nrows) {
            v2f *o = (v2f *)(out + (size_t)(r0 + r) * 1024u) + k1;
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) o[32 * k2] = x[k2] * (v2f){sc, -sc};  // conj, scale: (x[2n], x[2n+1])
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// k_istft1024: fused f32 inverse STFT for n_fft = 1024 (src/spectrogram.rs:4860-4946).  A workgroup owns `nbk` consecutive
// hop blocks of one signal's padded output and the NF = nbk + ov frames that touch them (ov = floor(1023/hop) halo
// frames are recomputed by the neighbouring workgroup instead of exchanging partial sums through HBM or atomics).
//   load    [bin][frame] input (frame axis contiguous, S9): lane (r = tid mod NF, n2 = tid / NF) -> 128/256-byte segments
//   C2R     exactly k_c2r1024's construction (32 x 16 split of the 512-point complex transform, one LDS exchange)
//   frames  (x * 1/n) * w written as real rows fr[NF][1024] over the dead exchange buffer
//   OLA     one thread per output sample: ascending-frame sum, norm = sum of w*w (unfused), divide where norm > 1e-10,
//           centre trim by index shift; stores are contiguous runs of the output signal.
struct IstftArgs {
    const void *spec;  // [batch][513][n_frames] complex f32
    void *out;         // [batch][out_len] f32
    const void *win;   // [1024] f32
    unsigned n_frames, hop, batch, tiles, ov, nbk;
    unsigned long long start, out_len;
    float scale;
    unsigned *bad_flag;
};
// Row (frame) stride of the exchange buffer.  In pass 1 the 16 contiguous lanes of a ds_write_b64 group walk r, so the stride
// in dwords must spread them over the 32 write banks: 4624 B = 1156 dwords = 4 (mod 32) gives 8 distinct bank pairs (2-way,
// the best a 16-byte aligned stride allows; 4640 B = 8 (mod 32) was 4-way: SQ_LDS_BANK_CONFLICT 48 % of the LDS cycles);
// 4624 / 16 = 289 = 1 (mod 16) keeps the pass-2 ds_read_b128 (lanes over k1, 144-byte rows) conflict-free.
constexpr int kISeq = kRSeq + 16;  // (kept: the first fused inverse kernel's stride; k_istft1024c uses kBFS below)
// NF frames per workgroup of 16 NF threads: 16 -> 74 240 B, two workgroups per CU; 32 -> 148 480 B, one workgroup of 8 waves
// (the real frames, NF * 4 KiB, overlay the exchange buffer).  32 halves the share of halo frames (3 of 32 instead of 3 of 16
// at hop 256) and doubles the load segments to 256 bytes, but measures slower (see launch_istft1024): 16 is the default.

// ---------------------------------------------------------------------------------------------------------------------
// k_istft1024c (round 4): the fused inverse STFT without halo frames.
//   A  lane (jq, f), job j = wave + 4 jq, owns rows j and 32 - j (job 0: rows 0 and 16) of v[k1 + 32 k2], v = conj(Z'):
//      16 pairs P = X[k], Q = X[512 - k] with S = P + conj Q, T = conj(W^k)(P - conj Q): v[k] = conj(S + i T) and
//      v[512 - k] = S - i T (conj(W^k) from an LDS copy of the plan's table); 16-point transforms of both rows;
//      ds_write_b128 to ex[f][k1][n2].
//   B  lane (f, n2): column n2 of the 32 rows, twiddle W_512^(k1 n2) (two per-lane register tables), 32-point transform:
//      y[n2 + 16 n1] -> (x[2n], x[2n+1]) = conj(y) / 1024, times the window, real frames fr[f][1024] over the dead ex.
//   C  overlap-add with a CARRY: a tile is 16 NEW frames F .. F + 15 and the 16 hop blocks F .. F + 15 of the output; a block h
//      takes frames h - ov .. h, so its first `ov` blocks start from the partial sums the previous tile left in LDS (<= 1023 floats)
//      and the partial sums of blocks F + 16 .. F + 15 + ov are left for the next one — frames are added in ascending order exactly
//      as before (src/spectrogram.rs:4906-4925), the carry is just the sum so far.  Rounds 1-3 recomputed the `ov` frames in front
//      of every tile instead (3 of 16 at hop 256: 19 % of the loads and of the transforms).
// A workgroup walks RUNS of consecutive tiles of one signal (the launcher cuts every signal into equal runs so that all workgroups
// get one); a run that starts inside a signal first passes over the tile in front of it without storing, which builds its carry.
// DFT_512 over m = k1 + 32 k2 -> n = n2 + 16 n1:  W^(mn) = W_16^(k2 n2) W_512^(k1 n2) W_32^(k1 n1).
constexpr int kBFS = 4240;               // bytes per frame of ex[f][32][16] (1060 dwords = 4 mod 32: conflict-free b128 writes)
constexpr int kBTw = 16 * kBFS;          // 67 840: conj(W_1024^k), k < 512 (4096 B)
constexpr int kBWin = kBTw + 4096;       // 71 936: the window (4096 B)
constexpr int kBCarry = kBWin + 4096;    // 76 032: the carry, ov * hop <= 1023 floats
constexpr int kBLds = kBCarry + 4096;    // 80 128 B -> two workgroups per CU (160 256 of 163 840)

#ifdef SGX_IS_STAMPS  // diagnostic build only (tools/stamps_istft.py): a wave's cycles per phase
__device__ unsigned long long g_is_stamps[16];
#define IS_STAMP(i)                                                                \
    do {                                                                           \
        unsigned long long t_;                                                     \
        __builtin_amdgcn_sched_barrier(0);                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                         \
        st_acc[i] += t_ - st_prev;                                                 \
        st_prev = t_;                                                              \
    } while (0)
#else
#define IS_STAMP(i)
#endif

// Overlap-add of a tile's 16 windowed real frames fr[16][1024] (LDS rows = frames F .. F + 15) with the carry.
//   phase 1: output blocks hb = 0 .. 15 (h = F + hb): sum = carry (frames before F, if the offset reaches back that far) + this
//            tile's frames max(F, h - q + 1) .. h in ascending order; norm = sum of w^2 over the frames that exist; store.
//   phase 2: blocks hb = 16 .. 15 + ov: the sums of this tile's frames h - q + 1 .. F + 15 go to the carry.
// A thread owns offsets (hop >= NT: off = tid, tid + NT, ...; hop < NT: off = tid mod hop and every (NT / hop)-th block).
// Interior tiles at hop 256 / 512 / 1024 (every position inside the output, every frame there, q = 1024 / HOP frames over every
// offset): the same sums with every index a compile-time constant, so that a thread's 16 x q frame reads are all in flight together
// (the general walk below exposes an LDS round trip per frame added).
template <unsigned HOP, unsigned NT>
__device__ __forceinline__ void istft_ola_carry_fast(const IstftArgs &a, const unsigned char *smem, const float *w, float *carry, unsigned tid,
                                                     unsigned b, unsigned F, bool store) {
    constexpr unsigned Q = 1024u / HOP, OV = Q - 1u;
    const float *fr = (const float *)smem;
    float *o = (float *)a.out + (size_t)b * a.out_len + ((unsigned long long)F * HOP - a.start);
#pragma unroll
    for (unsigned k = 0; k < HOP / NT; ++k) {
        const unsigned off = tid + k * NT;
        if (store) {
            float nrm = 0.f;  // ascending frame = descending sample index: the reference's order
#pragma unroll
            for (unsigned i = Q; i-- > 0;) {
                const float wj = w[i * HOP + off];
                nrm = __fadd_rn(nrm, __fmul_rn(wj, wj));
            }
            const bool div = nrm > 1e-10f;
#pragma unroll
            for (unsigned hb = 0; hb < 16u; ++hb) {
                float acc = hb < OV ? carry[hb * HOP + off] : 0.f;
#pragma unroll
                for (unsigned d = (hb < OV ? hb : OV) + 1u; d-- > 0;) acc += fr[(hb - d) * 1024u + d * HOP + off];  // frames hb - d, ascending
                o[hb * HOP + off] = div ? acc / nrm : acc;
            }
        }
    }
    // (HOP >= NT: an offset belongs to ONE thread, which has read its carry values above before it overwrites them here — no barrier)
    static_assert(HOP >= NT && HOP % NT == 0, "fast overlap-add: one owner per offset");
#pragma unroll
    for (unsigned k = 0; k < HOP / NT; ++k) {
        const unsigned off = tid + k * NT;
#pragma unroll
        for (unsigned hb2 = 0; hb2 < OV; ++hb2) {
            float acc = 0.f;
#pragma unroll
            for (unsigned d = OV; d > hb2; --d) acc += fr[(16u + hb2 - d) * 1024u + d * HOP + off];  // rows 16 + hb2 - d <= 15
            carry[hb2 * HOP + off] = acc;
        }
    }
}

template <unsigned NT>
__device__ __forceinline__ void istft_ola_carry(const IstftArgs &a, const unsigned char *smem, const float *w, float *carry, unsigned tid, unsigned b,
                                                unsigned F, bool store) {
    const float *fr = (const float *)smem;
    float *o = (float *)a.out + (size_t)b * a.out_len;
    const unsigned hop = a.hop, ov = a.ov;
    const unsigned long long p0 = (unsigned long long)F * hop;
    // interior tile: every frame of the tile and the `ov` before it exist, every position lies inside the output
    const bool interior = F >= ov && F + 15u < a.n_frames && p0 >= a.start && p0 + 16ull * hop <= a.start + a.out_len;
    if (interior) {  // (uniform)
        if (hop == 256u) return istft_ola_carry_fast<256, NT>(a, smem, w, carry, tid, b, F, store);
        if (hop == 512u) return istft_ola_carry_fast<512, NT>(a, smem, w, carry, tid, b, F, store);
        if (hop == 1024u) return istft_ola_carry_fast<1024, NT>(a, smem, w, carry, tid, b, F, store);
    }
    const bool small = hop < NT;
    const unsigned nrep = small ? NT / hop : 1u, g = small ? tid / hop : 0u, ostep = small ? hop : NT;
    const unsigned off0 = small ? tid - g * hop : tid;
    if (g < nrep && store) {  // (a run's first pass over the tile in front of it only builds the carry)
        for (unsigned off = off0; off < hop; off += ostep) {
            const unsigned q = (1024u - off + hop - 1u) / hop;  // frames overlapping this offset: h - q + 1 .. h
            float nrm_full = 0.f;  // sum of w^2 over the q frames, ascending frame = descending sample index: the reference's order
            for (unsigned i = q; i-- > 0;) {
                const float wj = w[i * hop + off];
                nrm_full = __fadd_rn(nrm_full, __fmul_rn(wj, wj));
            }
            for (unsigned hb = g; hb < 16u; hb += nrep) {
                const unsigned h = F + hb;
                const unsigned long long pos = p0 + (unsigned long long)hb * hop + off;
                // frames h - q + 1 .. h; those before F are in the carry
                const unsigned back = q - 1u;  // how far the offset reaches back
                float acc = hb < back ? carry[hb * hop + off] : 0.f;
                const unsigned r_lo = hb < back ? 0u : hb - back;  // first row (frame - F) of this tile taking part
                const float *src = fr + r_lo * 1024u + (hb - r_lo) * hop + off;
                for (unsigned r = r_lo; r <= hb; ++r) {  // next frame: row + 1, sample index - hop
                    acc += *src;
                    src += 1024 - (int)hop;
                }
                float nrm = nrm_full;
                if (!interior) {
                    if (pos < a.start || pos - a.start >= a.out_len) continue;
                    // signal edges: only the frames that exist (0 <= f < n_frames) count, ascending
                    const long long f_lo = (long long)h - (long long)back < 0 ? 0ll : (long long)h - (long long)back;
                    const long long f_hi = h < a.n_frames ? (long long)h : (long long)a.n_frames - 1;
                    if ((unsigned)(f_hi - f_lo + 1) != q || f_hi < f_lo) {
                        nrm = 0.f;
                        for (long long f = f_lo; f <= f_hi; ++f) {
                            const float wj = w[(unsigned)((long long)h - f) * hop + off];
                            nrm = __fadd_rn(nrm, __fmul_rn(wj, wj));
                        }
                    }
                }
                if (nrm > 1e-10f) acc /= nrm;
                o[pos - a.start] = acc;
            }
        }
    }
    __syncthreads();  // every carry value has been read
    if (g < nrep) {
        for (unsigned off = off0; off < hop; off += ostep) {
            const unsigned q = (1024u - off + hop - 1u) / hop;
            for (unsigned hb2 = g; hb2 < ov; hb2 += nrep) {
                const unsigned hb = 16u + hb2, back = q - 1u;
                float acc = 0.f;
                if (hb <= 15u + back) {  // the offset reaches back into this tile: rows hb - back .. 15
                    const unsigned r_lo = hb - back;
                    const float *src = fr + r_lo * 1024u + (hb - r_lo) * hop + off;
                    for (unsigned r = r_lo; r < 16u; ++r) {
                        acc += *src;
                        src += 1024 - (int)hop;
                    }
                }
                carry[hb2 * hop + off] = acc;
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void k_istft1024c(IstftArgs a, const v2f *twr, const v2f *tw1, unsigned per_xcd, unsigned total_runs,
                                                       unsigned slots, unsigned runs_per_signal, unsigned run_len) {
    constexpr unsigned NT = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x;
    v2f *twl = (v2f *)(smem + kBTw);
    twl[tid] = twr[tid];
    twl[tid + 256u] = twr[tid + 256u];
    ((v4f *)(smem + kBWin))[tid] = ((const v4f *)a.win)[tid];
    float *carry = (float *)(smem + kBCarry);
    __syncthreads();
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned lo = xcd * per_xcd, hi = min(lo + per_xcd, total_runs);
    const unsigned lane = tid & 63u, jq = lane >> 4, fl = lane & 15u;
    const unsigned j = (tid >> 6) + 4u * jq;
    const bool j0 = j == 0u;
    const unsigned nf8 = a.n_frames * 8u;
    // pair p: bin kp and its mirror 512 - kp.  General job: kp = j + 32 p.  Job 0: kp = 16 + 32 p (row 16) for p < 8 and
    // kp = 32 (p - 8) (row 0) for p >= 8; bin 256 pairs with itself and is handled apart.
    const unsigned ka = j0 ? 16u : j, kb = j0 ? 0u : j + 256u;
    const unsigned st = 32u * nf8;  // byte offsets are stepped by 32 bins (no per-load multiply)
    v2f P[16], Q[16], X256;
    auto request = [&](unsigned b, unsigned t) {
        const unsigned f = 16u * t + fl;
        const unsigned fcl = f < a.n_frames ? f : 0u;  // a frame past the signal reads frame 0 and is replaced by zeros in the fold
        const unsigned char *inb = (const unsigned char *)a.spec + (size_t)b * 513u * a.n_frames * 8u;
        unsigned oa = ka * nf8 + fcl * 8u, oy = (512u - ka) * nf8 + fcl * 8u;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            if (p == 8) {
                oa = kb * nf8 + fcl * 8u;
                oy = (512u - kb) * nf8 + fcl * 8u;
            }
            P[p] = *(const v2f *)(inb + oa);
            Q[p] = *(const v2f *)(inb + oy);
            oa += st;
            oy -= st;
        }
        X256 = *(const v2f *)(inb + 256u * nf8 + fcl * 8u);
    };
    // run rid = signal rid / R, tiles [t0, t1) of it; a run inside a signal starts one tile early (carry only, nothing stored)
    auto run_of = [&](unsigned rid, unsigned &b, unsigned &t0, unsigned &t1, unsigned &ts) {
        b = rid / runs_per_signal;
        t0 = (rid - b * runs_per_signal) * run_len;
        t1 = min(a.tiles, t0 + run_len);
        ts = (t0 > 0u && a.ov) ? t0 - 1u : t0;
    };
    unsigned rid = lo + slot, b = 0, t0 = 0, t1 = 0, t = 0;
    if (rid < hi) {
        run_of(rid, b, t0, t1, t);
        if (t0 >= t1) rid = hi;  // (an empty run: the launcher makes none, kept for safety)
    }
    bool fresh = true;  // the run has just started: its carry is zero
    if (rid < hi) request(b, t);
#ifdef SGX_IS_STAMPS
    unsigned long long st_acc[12] = {0}, st_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
    while (rid < hi) {
        const unsigned F = 16u * t;
        // where the walk goes next: the next tile of the run, or the first tile of the workgroup's next run
        unsigned nrid = rid, nb = b, nt0 = t0, nt1 = t1, nt = t + 1u;
        if (nt >= t1) {
            nrid = rid + slots;
            if (nrid < hi) run_of(nrid, nb, nt0, nt1, nt);
        }
        if (fresh) {
#pragma unroll
            for (int q = 0; q < 4; ++q) carry[tid + 256u * q] = 0.f;  // (ordered before its first use by the barriers below)
        }
        {
            const unsigned f = F + fl;
            const bool valid = f < a.n_frames;
            if (!valid) {  // a frame past the signal is zeros by a select: the frame 0 loaded in its place may hold Inf / NaN, which a product with 0 would spread into the tail (the reference poisons only the samples frame 0 covers, spectrogram.rs:4906-4925)
#pragma unroll
                for (int p = 0; p < 16; ++p) P[p] = Q[p] = (v2f){0.f, 0.f};
                X256 = (v2f){0.f, 0.f};
            }
            v2f PA[16], QB[16];
            const v2f *tp = twl + ka;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                if (p == 8) tp = twl + kb;
                v2f Pp = P[p], Qp = Q[p];
                if (p == 8) {  // job 0: kp = 0 — DC and Nyquist bins: realfft ignores (and reports) their imaginary parts
                    if (j0) {
                        if (a.bad_flag && valid && (Pp.y != 0.f || Qp.y != 0.f)) atomicOr(a.bad_flag, 1u);
                        Pp.y = 0.f;
                        Qp.y = 0.f;
                    }
                }
                const v2f cw = tp[32 * (p & 7)];  // conj(W_1024^kp)
                const v2f S = pfma(Qp, (v2f){1.f, -1.f}, Pp), D = pfma(Qp, (v2f){-1.f, 1.f}, Pp);
                const v2f T = cmulv(D, cw);
                PA[p] = pfma(swp(T), (v2f){-1.f, -1.f}, S * (v2f){1.f, -1.f});  // conj(S + i T)
                QB[p] = pfma(swp(T), (v2f){1.f, -1.f}, S);    // S - i T
            }
            v2f A[16], B[16];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                A[i] = j0 ? PA[8 + i] : PA[i];        // job 0: v[32 i]
                B[8 + i] = QB[7 - i];                 // both: element 15 - p of the mirror row, p = 7 - i
                B[i] = j0 ? PA[i] : QB[15 - i];       // job 0: v[16 + 32 i]
            }
            // job 0, bin 256: v[256] = conj(Z'[256]) = 2 X[256]
            A[8] = j0 ? X256 * (v2f){2.f, 2.f} : PA[8];
#pragma unroll
            for (int i = 1; i < 8; ++i) A[8 + i] = j0 ? QB[16 - i] : PA[8 + i];  // job 0: v[512 - 32 (8 - i)] = v[32 (8 + i)]
            IS_STAMP(0);  // wait for the pairs + fold
            // the pairs are consumed: the next tile's go out now and land during the rest of this tile
            if (nrid < hi) request(nb, nt);
            IS_STAMP(1);  // request issue
            Fft<16, false>::run(A, A);
            Fft<16, false>::run(B, B);
            const unsigned ra = j, rb = j0 ? 16u : 32u - j;
            v4f *da = (v4f *)(smem + fl * kBFS + ra * 128u), *db = (v4f *)(smem + fl * kBFS + rb * 128u);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                da[c] = (v4f){A[2 * c].x, A[2 * c].y, A[2 * c + 1].x, A[2 * c + 1].y};
                db[c] = (v4f){B[2 * c].x, B[2 * c].y, B[2 * c + 1].x, B[2 * c + 1].y};
            }
        }
        IS_STAMP(2);  // 16-point transforms + exchange writes
        __syncthreads();
        IS_STAMP(3);
        const unsigned f2 = tid >> 4, n2 = tid & 15u;
        v2f v[32];
        {
            const unsigned char *src = smem + f2 * kBFS + n2 * 8u;
#pragma unroll
            for (int k1 = 0; k1 < 32; ++k1) v[k1] = *(const v2f *)(src + k1 * 128);
            v2f twa[4], twb[8];  // W_512^(k1 n2) = twa[k1 >> 3] * twb[k1 & 7]
#pragma unroll
            for (int q = 0; q < 4; ++q) twa[q] = tw1[16 * 8 * q + n2];
#pragma unroll
            for (int q = 0; q < 8; ++q) twb[q] = tw1[16 * q + n2];
#pragma unroll
            for (int k1 = 1; k1 < 32; ++k1) {
                const int qa = k1 >> 3, qb = k1 & 7;
                if (qb) v[k1] = cmulv(v[k1], twb[qb]);
                if (qa) v[k1] = cmulv(v[k1], twa[qa]);
            }
            Fft<32, false>::run(v, v);
        }
        IS_STAMP(4);  // column reads, twiddles, 32-point transform
        __syncthreads();  // exchange buffer consumed: overlay the real frames
        IS_STAMP(5);
        {
            const v2f *w2 = (const v2f *)(smem + kBWin) + n2;
            v2f *fr2 = (v2f *)smem + f2 * 512u + n2;
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) {
                const v2f ww = w2[16 * n1];
                const v2f sc = v[n1] * (v2f){a.scale, -a.scale};  // conj + 1/n: (x[2n], x[2n+1]), n = n2 + 16 n1
                fr2[16 * n1] = (v2f){__fmul_rn(sc.x, ww.x), __fmul_rn(sc.y, ww.y)};
            }
        }
        IS_STAMP(6);  // scale, window, frame writes
        __syncthreads();
        IS_STAMP(7);
        istft_ola_carry<NT>(a, smem, (const float *)(smem + kBWin), carry, tid, b, F, t >= t0);
        IS_STAMP(8);  // overlap-add, normalise, stores, carry
        __syncthreads();  // the frames are consumed and the carry is complete: the next tile may overwrite / read them
        IS_STAMP(9);
#ifdef SGX_IS_STAMPS
        st_acc[10] += 1;
#endif
        fresh = nrid != rid;
        rid = nrid; b = nb; t0 = nt0; t1 = nt1; t = nt;
    }
#ifdef SGX_IS_STAMPS
    if ((threadIdx.x & 63u) == 0) {
        for (int q = 0; q < 11; ++q) atomicAdd(&g_is_stamps[q], st_acc[q]);
        atomicAdd(&g_is_stamps[11], 1ull);
    }
#endif
}

}  // namespace

hipError_t launch_istft1024(const void *spec, void *out, const void *win, unsigned n_frames, unsigned hop, unsigned batch,
                            unsigned long long start, unsigned long long out_len, float scale, unsigned *bad_flag,
                            const void *twr, const void *tw1, hipStream_t s) {
    if (hop == 0 || hop > 1024) return hipErrorInvalidConfiguration;
    IstftArgs a{};
    a.spec = spec; a.out = out; a.win = win;
    a.n_frames = n_frames; a.hop = hop; a.batch = batch;
    a.ov = 1023u / hop;
    if (a.ov >= 16) return hipErrorInvalidConfiguration;
    a.nbk = 16u;
    const unsigned long long full = (unsigned long long)(n_frames - 1) * hop + 1024ull;
    const unsigned long long blocks = (full + hop - 1) / hop;
    a.tiles = (unsigned)((blocks + 15u) / 16u);  // 16 hop blocks (and the 16 frames that start in them) per tile
    a.start = start; a.out_len = out_len; a.scale = scale; a.bad_flag = bad_flag;
    if ((unsigned long long)a.tiles * batch >= 0x7fffffffull || a.tiles == 0) return hipErrorInvalidConfiguration;
    hipError_t e = set_max_dynamic_lds((const void *)k_istft1024c, kBLds);
    if (e != hipSuccess) return e;
    // Runs of consecutive tiles for the 2 workgroups per CU: istft_carry_runs (sgx_internal.h) weighs rounds against warm-up tiles.
    const unsigned wgs = 2u * device_cu_count();
    unsigned R, run_len;
    istft_carry_runs(a.tiles, batch, wgs, a.ov, R, run_len);
    const unsigned total_runs = R * batch, per_xcd = (total_runs + 7u) / 8u;
    const unsigned slots = std::max(1u, std::min(per_xcd, wgs / 8u));
    hipLaunchKernelGGL(k_istft1024c, dim3(8u * slots), dim3(256), kBLds, s, a, (const v2f *)twr, (const v2f *)tw1, per_xcd, total_runs, slots, R,
                       run_len);
    return hipGetLastError();
}

hipError_t launch_colconv1024(const C2cArgs &a, const void *tw1c, const void *mul, unsigned long long mul_row, int mul_kind,
                              hipStream_t s, bool real_io) {
    const unsigned long long g = (unsigned long long)a.tiles * a.batch;
    if (g == 0 || g >= 0x7fffffffull || a.n != 1024) return hipErrorInvalidConfiguration;
    // the kernel addresses an image, and the kernel spectrum / mask, with 32-bit byte offsets
    if (a.in_img * 8ull >= (1ull << 31) || a.out_img * 8ull >= (1ull << 31) || 1024ull * mul_row * 8ull >= (1ull << 31)) return hipErrorInvalidConfiguration;
    {
        hipError_t e;
        if ((e = set_max_dynamic_lds((const void *)k_colconv1024<MUL_SPECTRUM>, kCLdsP)) != hipSuccess) return e;
        if ((e = set_max_dynamic_lds((const void *)k_colconv1024<MUL_MASK>, kCLdsP)) != hipSuccess) return e;
        if ((e = set_max_dynamic_lds((const void *)k_colconv1024<MUL_OUTER>, kCLdsP)) != hipSuccess) return e;
        if ((e = set_max_dynamic_lds((const void *)k_colconv1024<MUL_VEC, true>, kCLdsP)) != hipSuccess) return e;
    }
    const unsigned cus = device_cu_count();  // of the current device = the plan's (DeviceGuard)
    // persistent: one workgroup per CU; XCD x (blockIdx mod 8) walks the contiguous run [x per_xcd, (x + 1) per_xcd) of tiles
    const unsigned total = (unsigned)g, per_xcd = (total + 7u) / 8u;
    const unsigned slots = std::max(1u, std::min(cus / 8u, per_xcd));
    const dim3 grid(slots * 8u);
    if (real_io != (mul_kind == MUL_VEC)) return hipErrorInvalidValue;  // the real-pair form exists for the separable passes only
    if (real_io) hipLaunchKernelGGL((k_colconv1024<MUL_VEC, true>), grid, dim3(512), kCLdsP, s, a, (const v2f *)tw1c, mul, mul_row, per_xcd, total);
    else if (mul_kind == MUL_MASK) hipLaunchKernelGGL((k_colconv1024<MUL_MASK>), grid, dim3(512), kCLdsP, s, a, (const v2f *)tw1c, mul, mul_row, per_xcd, total);
    else if (mul_kind == MUL_OUTER) hipLaunchKernelGGL((k_colconv1024<MUL_OUTER>), grid, dim3(512), kCLdsP, s, a, (const v2f *)tw1c, mul, mul_row, per_xcd, total);
    else if (mul_kind == MUL_SPECTRUM) hipLaunchKernelGGL((k_colconv1024<MUL_SPECTRUM>), grid, dim3(512), kCLdsP, s, a, (const v2f *)tw1c, mul, mul_row, per_xcd, total);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_c2r1024(const C2rArgs &a, const void *twr, const void *tw1, hipStream_t s) {
    const unsigned long long g = (unsigned long long)a.tiles * a.batch;
    if (g == 0 || g >= 0x7fffffffull || a.ncols != 1024 || a.in_ks != 1) return hipErrorInvalidConfiguration;
    {
        hipError_t e = set_max_dynamic_lds((const void *)k_c2r1024, kRLds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_c2r1024, dim3((unsigned)g), dim3(256), kRLds, s, a, (const v2f *)twr, (const v2f *)tw1);
    return hipGetLastError();
}

hipError_t launch_c2c1024(const C2cArgs &a0, const void *tw1c, hipStream_t s) {
    constexpr unsigned NS = SGX_C2C1024_NS;
    constexpr int lds = (int)NS * kCFS;
    C2cArgs a = a0;
    a.tile = NS;
    a.tiles = (a.nseq + NS - 1) / NS;
    const unsigned long long g = (unsigned long long)a.tiles * a.batch;
    if (g == 0 || g >= 0x7fffffffull || a.n != 1024) return hipErrorInvalidConfiguration;
    {
        hipError_t e;
        if ((e = set_max_dynamic_lds((const void *)k_c2c1024<false, false, NS>, lds)) != hipSuccess) return e;
        if ((e = set_max_dynamic_lds((const void *)k_c2c1024<false, true, NS>, lds)) != hipSuccess) return e;
        if ((e = set_max_dynamic_lds((const void *)k_c2c1024<true, false, NS>, lds)) != hipSuccess) return e;
        if ((e = set_max_dynamic_lds((const void *)k_c2c1024<true, true, NS>, lds)) != hipSuccess) return e;
    }
    const v2f *tw = (const v2f *)tw1c;
    const dim3 grid(xcd_grid(g)), block(32 * NS);
    if (a.in_seq_fast) {
        if (a.inverse) hipLaunchKernelGGL((k_c2c1024<true, true, NS>), grid, block, lds, s, a, tw);
        else hipLaunchKernelGGL((k_c2c1024<true, false, NS>), grid, block, lds, s, a, tw);
    } else {
        if (a.inverse) hipLaunchKernelGGL((k_c2c1024<false, true, NS>), grid, block, lds, s, a, tw);
        else hipLaunchKernelGGL((k_c2c1024<false, false, NS>), grid, block, lds, s, a, tw);
    }
    return hipGetLastError();
}

}  // namespace sgx

#ifdef SGX_IS_STAMPS
extern "C" int sgx_debug_read_is_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sgx::g_is_stamps), sizeof(sgx::g_is_stamps)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(sgx::g_is_stamps), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif
