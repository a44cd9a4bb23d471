// kernels_r32x16.hip — tuned f32, n_fft = 1024 STFT kernels for gfx950 (the BASELINE shape).
//
// Transform structure (shared by both kernels below); a tile = 16 consecutive frames of one signal:
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
//           transpose.  Mel: |X|^2 goes to LDS pw[f][k] (overlaying a free ex buffer), then a (mel, frame)-per-lane
//           CSR reduction in ascending-bin order (spectrogram.rs:102-117) and the dB/sqrt epilogue.
//
// All complex arithmetic is written on 2-float vectors so it compiles to packed-f32 VALU (v_pk_add/mul/fma_f32 with
// op_sel / neg modifiers): measured on MI355X a packed op issues at the same cost as a scalar one for a single wave
// (tools/ubench/valu_rate.hip), and the 64 KiB exchange buffer caps occupancy at 2 waves per SIMD.
//
// Two kernels:
//   k_r32x16  (default) persistent workgroups, both passes in every wave.  HALVES = 2: one 512-thread workgroup per CU whose
//             two 256-thread halves each own a tile and an exchange buffer and move through the phases in lockstep (measured
//             6-30 % faster than two independent workgroups).  Linear / complex outputs load samples directly (per-lane
//             float2, one tile ahead) and store unconditionally so the compiler can wait with vmcnt(32) instead of draining
//             every store each tile; Mel-type outputs stage the tile's samples through LDS with coalesced 16-byte loads
//             (ROUNDS = 5) and reduce |X|^2 on the LDS band table or, for dense banks, on the matrix cores.
//   k_ws      (experimental, SGX_KERNEL=ws) wave-specialised pipeline: 4 producer waves (pass 1) + 4 consumer waves (pass 2,
//             stores, global->LDS staging two tiles ahead), ex double-buffered.  Measured slower (204-235 us vs 164 us): the
//             CU's vector-memory queue is in order, so the producers' loads still queue behind the consumers' stores.
//
// Reference semantics implemented: spectrogram.rs:1301-1334 (framing, window, R2C, |.|^2), :1845-1865,
// :2068-2080; replaces the per-frame `R2cPlan::process` call at :1323 (fft_backend.rs:423-431).
#include <cstdlib>

#include "fft_inreg.h"
#include "sgx_internal.h"

namespace sgx {
namespace {

using namespace inreg;

constexpr int kFS = 4096 + 16;        // LDS bytes per frame of ex (odd multiple of 16 -> conflict-free b128 reads)
constexpr int kPS = 516;              // floats per frame of pw (overlays a free ex buffer); multiple of 4: 16-B rows
constexpr int kExBytes = 16 * kFS;    // 65792
constexpr int kTw2Stride = 17;        // row stride of tw2 in float4 (bank spread between the 4 jobs of a wave)
constexpr int kTw2Bytes = 17 * 17 * 16;  // float4 tw2[17][17] = (wr, wi, wi, -wr) of W_1024^(row + 32*idx)
// k_r32x16: ex | win | tw2
constexpr int kWinOff = kExBytes;
constexpr int kTw2Off = kWinOff + 4096;
constexpr int kMelOff = kTw2Off + kTw2Bytes;   // padded Mel bank (Mel modes): w4[chunks] (float4) | pptr[n_mels+1] | pcol[n_mels]
constexpr int kMelMaxRows = 160, kMelMaxChunks = 376;  // 376: kLdsMel = 81824 B, so that two independent workgroups would still fit 160 KiB
constexpr int kMelBytes = kMelMaxChunks * 16 + (2 * kMelMaxRows + 1) * 4 + 12;  // 9488
constexpr int kLds = kTw2Off + kTw2Bytes;      // 74512 B -> two workgroups per CU (160 KiB LDS)
constexpr int kLdsMel = kMelOff + kMelBytes;   // 82208 B -> still two per CU
// k_ws: ex0 | ex1 | xs
constexpr int kWsXsOff = 2 * kExBytes;      // 131584
constexpr int kWsXsBytes = 23040;           // 1280 chunks * 16 B + 128 B per KiB of padding
constexpr int kWsLds = kWsXsOff + kWsXsBytes;  // 154624 B (window and twiddles live in registers)
constexpr int kWsRounds = 5;                // 16-byte chunks per consumer thread per tile

template <int AMP>
__device__ __forceinline__ float amp_f32(float p, float eps) {
    if constexpr (AMP == AMP_MAGNITUDE) return sqrtf(p);
    else if constexpr (AMP == AMP_DB) return 10.0f * log10f(fmaxf(p, eps));
    else return p;
}
__device__ __forceinline__ float power_of(v2f x) { return __builtin_fmaf(x.x, x.x, x.y * x.y); }

// per-lane pass-1 twiddle tables: W_512^(k1*n2) = twa[k1>>3] * twb[k1&7]
__device__ __forceinline__ void load_tw1(const StftArgs &a, unsigned n2, v2f (&twa)[4], v2f (&twb)[8]) {
    const v2f *t1 = (const v2f *)a.tw1 + n2;
#pragma unroll
    for (int q = 0; q < 4; ++q) twa[q] = t1[16 * 8 * q];
#pragma unroll
    for (int q = 0; q < 8; ++q) twb[q] = t1[16 * q];
}

// pass 1 arithmetic of one lane: raw column xr (consumed), window column from LDS, result rows -> ex
__device__ __forceinline__ void pass1_compute(v2f (&xr)[32], const v2f (&wn)[32], const v2f (&twa)[4],
                                              const v2f (&twb)[8], unsigned char *dst) {
#ifndef SGX_ABL_NOFFT32
    Fft<32, true>::run(xr, wn);
#endif
#pragma unroll
    for (int k1 = 0; k1 < 32; ++k1) {
        const int qa = k1 >> 3, qb = k1 & 7;
        v2f r = xr[k1];
#ifndef SGX_ABL_NOP1TW
        if (qb) r = cmulv(r, twb[qb]);
        if (qa) r = cmulv(r, twa[qa]);
#endif
#ifdef SGX_ABL_NOEXW
        asm volatile("" ::"v"(r));
#else
        *(v2f *)(dst + k1 * 128) = r;
#endif
    }
}

__device__ __forceinline__ void read_rows(const unsigned char *exf, unsigned ra, unsigned rb, v2f (&A)[16], v2f (&B)[16]) {
#ifdef SGX_ABL_NOROWS
    for (int c = 0; c < 16; ++c) { A[c] = (v2f){(float)ra, 1.f}; B[c] = (v2f){(float)rb, 2.f}; }
    return;
#endif
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

// Lane-constant part of the output addressing of a job: bins k = c1 + 32 i (i < 8) and c2 + 32 t (t < 8) plus their
// mirrors 512 - k, as element offsets k * n_frames (the frame offset is added per tile).
struct JobOfs {
    unsigned a1, b1, a2, b2;  // c1*nF, (512-c1)*nF, c2*nF, (512-c2)*nF
};
__device__ __forceinline__ JobOfs job_offsets(unsigned j, unsigned n_frames) {
    const unsigned c1 = j == 0 ? 16u : j, c2 = j == 0 ? 0u : j + 256u;
    return JobOfs{c1 * n_frames, (512u - c1) * n_frames, c2 * n_frames, (512u - c2) * n_frames};
}
// twiddles of a job in the order the real split consumes them: tw[i] for pair i of the first loop, tw[8+t] for the second
__device__ __forceinline__ void load_job_twiddles(const v4f *t2, unsigned j, v4f (&tw)[16]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) tw[i] = t2[(j == 0 ? 16u : j) * kTw2Stride + i];
#pragma unroll
    for (int t = 0; t < 8; ++t) tw[8 + t] = t2[j == 0 ? (unsigned)t : j * kTw2Stride + 8u + t];
}

// pass 2 arithmetic + real split + output of one lane (job j of frame f0 + p2f of signal b)
// TW: callable, tw(i) = twiddle (wr, wi, wi, -wr) of pair i (i < 8: first loop, 8 + t: second loop)
template <int MODE, int AMP, typename TW>
__device__ __forceinline__ void pass2_compute(const StftArgs &a, v2f (&A)[16], v2f (&B)[16], unsigned b, unsigned f0,
                                              unsigned p2f, unsigned j, float eps, TW tw, const JobOfs &jo, float *pw) {
    const bool j0 = (j == 0);
#ifndef SGX_ABL_NOFFT16
    Fft<16, false>::run(A, A);
    Fft<16, false>::run(B, B);
#endif
    // wave-uniform base + 32-bit per-lane byte offsets (host guarantees 513*n_frames*8 < 2^31)
    constexpr unsigned ES = MODE == OUT_COMPLEX ? 8u : 4u;
    unsigned char *ob = (unsigned char *)a.out + ((size_t)b * 513u) * a.n_frames * ES;
    const unsigned ofs = f0 + p2f;
    const unsigned step = 32u * a.n_frames * ES;  // uniform: 32 bins further
    auto emit = [&](unsigned off, unsigned k, v2f X, bool conj) {
#ifdef SGX_ABL_NOSTORE  // timing experiment only (tools/ablate.sh): keep the value alive, drop the store
        asm volatile("" ::"v"(X), "v"(off));
        return;
#endif
        if constexpr (MODE == OUT_COMPLEX) {
            *(v2f *)(ob + off) = conj ? (v2f){X.x, -X.y} : X;
        } else if constexpr (MODE == OUT_MEL) {
#ifdef SGX_ABL_NOPW
            asm volatile("" ::"v"(X), "v"(k));
#else
            pw[k] = AMP == AMP_MAG_IN ? sqrtf(power_of(X)) : power_of(X);
#endif
        } else {
            *(float *)(ob + off) = amp_f32<AMP>(power_of(X), eps);
        }
    };
    // pair (P, Q) = (Z[k], Z[512-k]), W = W_1024^k given as (wr, wi, wi, -wr):
    //   E = (P.x+Q.x, P.y-Q.y), D = (P.x-Q.x, P.y+Q.y) = (-O.y, O.x), T = W O, X[k] = E + T, X[512-k] = conj(E - T)
    auto split = [&](unsigned offa, unsigned offb, unsigned k, v2f P, v2f Q, v2f w) {
#ifdef SGX_ABL_NOSPLIT
        asm volatile("" ::"v"(P), "v"(Q));
        return;
#endif
        const v2f E = pfma(Q, (v2f){1.f, -1.f}, P);
        const v2f D = pfma(Q, (v2f){-1.f, 1.f}, P);
        // T = W O with O = (D.y, -D.x): (wr D.y + wi D.x, wi D.y - wr D.x) — two packed ops from the (wr, wi) pair alone
        const v2f T = pfma(D, hi2(w), (v2f){D.y, -D.x} * lo2(w));
        emit(offa, k, E + T, false);
        emit(offb, 512u - k, E - T, true);
    };
    const unsigned oa1 = (jo.a1 + ofs) * ES, ob1 = (jo.b1 + ofs) * ES, oa2 = (jo.a2 + ofs) * ES, ob2 = (jo.b2 + ofs) * ES;
    const unsigned k1 = j0 ? 16u : j, k2 = j0 ? 0u : j + 256u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // general job: (A[i], B[15-i]) at k = j + 32 i;  job 0: (B[i], B[15-i]) at k = 16 + 32 i (row 16)
        const v2f P = j0 ? B[i] : A[i];
        split(oa1 + i * step, ob1 - i * step, k1 + 32u * i, P, B[15 - i], tw(i));
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        // general job: (A[8+t], B[7-t]) at k = j + 32 (8+t);  job 0: (A[t], A[(16-t)%16]) at k = 32 t (row 0)
        const v2f P = j0 ? A[t] : A[8 + t];
        const v2f Q = j0 ? A[(16 - t) & 15] : B[7 - t];
        split(oa2 + t * step, ob2 - t * step, k2 + 32u * t, P, Q, tw(8 + t));
    }
    if (j0) emit((256u * a.n_frames + ofs) * ES, 256u, A[8] * (v2f){2.f, -2.f}, false);  // X[256] = 2 conj(Z[256])
}

// Mel stage over one tile: pwall[f][k] -> out[b][m][f0+f]; `nthreads` cooperating threads, this one is `t`
template <int AMP>
__device__ __forceinline__ void mel_tile(const StftArgs &a, const float *pwall, unsigned b, unsigned f0, unsigned nf,
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

// Same reduction with the 4-wide padded band table resident in LDS: 16-byte reads of weights and powers.  The padding
// weights are +0, and w*p = +0 added to a non-negative partial sum leaves it unchanged, so the result is bit-identical
// to the sequential ascending-bin accumulation of the reference (spectrogram.rs:102-117; unfused multiply-add).
// Lane -> (band, frame) map: ds_read_b128 is served in four fixed 16-lane groups, {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and
// the same + 32 (MI355X LDS), and only lanes of one group conflict.  Each group gets ONE band and all 16 frames: its weight
// read is a broadcast and its power reads fall on 16 different 16-byte slots (pw rows are 516 floats = 129 slots apart)
// whatever the band's first column — with frames on the low lane bits two bands shared a group and collided wherever their
// column offsets differed.
template <int AMP>
__device__ __forceinline__ void mel_tile_lds(const StftArgs &a, const float *pwall, const v4f *lw4, const unsigned *lptr,
                                             const unsigned *lcol, unsigned b, unsigned f0, unsigned nf, float eps,
                                             unsigned t, unsigned nthreads) {
    float *o = (float *)a.out + ((size_t)b * a.n_out) * a.n_frames + f0;
    const unsigned l5 = t & 31u;
    const bool g0 = l5 < 4u || (l5 >= 12u && l5 < 16u) || (l5 >= 20u && l5 < 28u);
    const unsigned ff = g0 ? (l5 < 4u ? l5 : l5 < 16u ? l5 - 8u : l5 - 12u) : (l5 < 12u ? l5 - 4u : l5 < 20u ? l5 - 8u : l5 - 16u);
    const unsigned mloc = (t >> 6) * 4u + ((t >> 5) & 1u) * 2u + (g0 ? 0u : 1u);  // band inside a block of nthreads / 16
    const float *prow = pwall + ff * kPS;
    for (unsigned mm = mloc; mm < a.n_mels; mm += nthreads >> 4) {
        const unsigned c0 = lptr[mm], c1 = lptr[mm + 1];
        const v4f *p4 = (const v4f *)(prow + lcol[mm]);
        float acc = 0.0f;
        for (unsigned c = c0; c < c1; ++c) {
            const v4f w = lw4[c], p = p4[c - c0];
            acc = __fadd_rn(__fmul_rn(w.x, p.x), acc);
            acc = __fadd_rn(__fmul_rn(w.y, p.y), acc);
            acc = __fadd_rn(__fmul_rn(w.z, p.z), acc);
            acc = __fadd_rn(__fmul_rn(w.w, p.w), acc);
        }
        if (ff < nf) o[mm * a.n_frames + ff] = amp_f32<AMP>(acc, eps);
    }
}

// Bank rows with wide supports (Mel bands, the dense ERB / gammatone bank of src/erb.rs:374-401) on the matrix cores.
// Per tile and 16-row block the product is [16 rows x K] x [K x 16 frames] with K = the block's own bin range;
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

#ifdef SGX_STAMPS  // diagnostic build only (tools/stamps.py): share of a wave's cycles per phase, per role
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
#else
#define SGX_STAMP(i)
#endif

// ====================================================================================================================
// k_r32x16: persistent 256-thread workgroups (two per CU), both passes in every wave; per-lane float2 loads issued one
// tile ahead.  Handles any even hop and 8-byte aligned rows.
// ====================================================================================================================
// HALVES = 2: one 512-thread workgroup per CU whose two halves each own a tile and an ex buffer and move through the
// phases in lockstep (shared barriers) — a CU then alternates cleanly between arithmetic and memory phases.
// ROUNDS > 0: the tile's (15*hop + 1024) samples are fetched ONCE with coalesced 16-byte loads (ROUNDS per thread, one tile
// ahead), staged in LDS (xs, overlaying this half's ex) and re-read per frame from there: a per-lane float2 load costs the
// CU's memory pipe ~17 cycles per wave-instruction (4 x 128-byte segments; tools/ubench/vmem_issue.hip) and every line is
// requested ~4 times because frames overlap by 75 %.  ROUNDS == 0 keeps the direct loads (any even hop, 8-byte alignment).
// WIDE (HALVES = 2, linear / complex outputs): pass 2 runs across the whole workgroup — lane (jq = 0..1, f = 0..31) of wave
// w = 0..7 owns job w + 8 jq of frame f of the PAIR of tiles (frames 16..31 live in the second half's ex buffer, which
// directly follows the first: kExBytes = 16 kFS).  The two halves own neighbouring tiles, so the 32 lanes of a job hold one
// bin of 32 consecutive frames: a store instruction covers 2 rows x 128 bytes instead of 4 rows x 64 bytes (the CU's
// address path charges per segment, tools/ubench/vmem_issue.hip) and L2 receives whole-line-sized runs.
// XSPAD (staged loads): the staging buffer carries 128 B of padding per KiB, which keeps the 4 frames of a wave on distinct banks
// when hop is a multiple of 256 samples.  It is a template parameter so that each variant addresses its 32 column reads with
// immediates off ONE base register: as a run-time switch the compiler hoisted both variants' 2 x 24 addresses into VGPRs and
// copied one set per tile.
template <int MODE, int AMP, int HALVES, int ROUNDS, bool WIDE = false, bool XSPAD = true>
__global__ __launch_bounds__(256 * HALVES, 2) void k_r32x16(StftArgs a, unsigned per_xcd, unsigned total, unsigned slots, unsigned skew) {
    static_assert(!WIDE || (HALVES == 2 && MODE != OUT_MEL), "wide pass 2 needs both halves and a per-bin output");
#ifdef SGX_LATEBAR  // experiment (measured equal or 1 % slower): the barrier that frees ex moved behind pass 2
    constexpr bool LATEBAR = MODE != OUT_MEL;  // Mel-type outputs overlay |X|^2 on ex during pass 2: they need the barrier early
#else
    constexpr bool LATEBAR = false;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const unsigned half = HALVES == 2 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0u;
    const unsigned tid = threadIdx.x & 255u;
    unsigned char *smem = smem_all + half * kExBytes;           // this half's ex / pw buffer
    unsigned char *tabs = smem_all + (HALVES - 1) * kExBytes;   // tables sit behind the last ex buffer (kWinOff etc. are relative to ex0 of a 1-half layout)
    if (threadIdx.x < 256u) ((v4f *)(tabs + kWinOff))[threadIdx.x] = ((const v4f *)a.window)[threadIdx.x];
    for (unsigned i = threadIdx.x; i < 17u * 17u; i += 256 * HALVES) {  // LDS copy keeps (wr, wi) only: 8-byte reads in pass 2
        const v4f q = ((const v4f *)a.tw2)[i];
        ((v2f *)(tabs + kTw2Off))[i] = (v2f){q.x, q.y};
    }
    v4f *lw4 = (v4f *)(tabs + kMelOff);
    unsigned *lptr = (unsigned *)(lw4 + a.mel_pchunks);
    unsigned *lcol = lptr + a.n_mels + 1;
    const bool mel_lds = MODE == OUT_MEL && a.mel_pw && a.n_mels <= kMelMaxRows && a.mel_pchunks <= kMelMaxChunks;
    if (mel_lds) {
        for (unsigned i = threadIdx.x; i < a.mel_pchunks; i += 256 * HALVES) lw4[i] = ((const v4f *)a.mel_pw)[i];
        for (unsigned i = threadIdx.x; i <= a.n_mels; i += 256 * HALVES) lptr[i] = a.mel_pptr[i];
        for (unsigned i = threadIdx.x; i < a.n_mels; i += 256 * HALVES) lcol[i] = a.mel_pcol[i];
    }

    // XCD-aware work mapping: blocks g and g+8 share an XCD (round-robin dispatch).  XCD x owns the contiguous run of
    // work ids [x*per_xcd, (x+1)*per_xcd); its `slots` resident workgroups walk that run with stride `slots`, so tiles
    // in flight on one XCD are neighbours: they share the 768-sample halo and the output lines they both touch in L2.
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned lo = xcd * per_xcd;
    const unsigned hi = min(lo + per_xcd, total);
    unsigned wid = lo + slot * HALVES + half;
    unsigned lead = lo + slot * HALVES;  // the first half's tile: uniform loop control for the whole workgroup

    const unsigned p1f = tid >> 4, n2 = tid & 15u;  // pass-1 identity
    // pass-2 identity
    const unsigned lane = tid & 63u;
    const unsigned wv_ = WIDE ? threadIdx.x >> 6 : tid >> 6;
    const unsigned jq = WIDE ? lane >> 5 : lane >> 4, p2f = WIDE ? lane & 31u : lane & 15u;
    const unsigned j = WIDE ? wv_ + 8u * jq : wv_ + 4u * jq;
    const unsigned ra = j, rb = j == 0 ? 16u : 32u - j;
    const float eps = (float)a.eps;
    const JobOfs jo = job_offsets(j, a.n_frames);
    v2f twa[4], twb[8];
    load_tw1(a, n2, twa, twb);

    v2f xr[32];                          // raw samples of this lane's (frame, n2) column
    // staged path: this thread's 16-byte chunks of the tile(s) being prefetched.  AHEAD2 (experiment, -DSGX_AHEAD2): two sets, the
    // tile staged in round t was requested in round t - 2 (into the set that round's staging had just freed) and the loop body is
    // instantiated once per set.  Parity-green, no spills in the linear variants (242-251 VGPRs) — and 1.5 % SLOWER (147 vs 145 us):
    // the cost of the sample loads is not their latency but their place in the CU's in-order vector-memory queue.
#ifdef SGX_AHEAD2
    constexpr bool AHEAD2 = ROUNDS > 0 && MODE != OUT_MEL;  // the Mel variants have no registers to spare (2-8 spilled dwords)
#else
    constexpr bool AHEAD2 = false;
#endif
    constexpr int NCR = ROUNDS > 0 ? ROUNDS : 1;
    v4f cregA[NCR], cregB[AHEAD2 ? NCR : 1];
    const unsigned chunks = (15u * a.hop + 1024u) >> 2;
    constexpr bool xs_pad = XSPAD;  // host: (hop & 255) == 0
    auto load_tile = [&](unsigned w, v4f (&creg)[NCR]) {
        const unsigned b = w / a.tiles, tile = w - b * a.tiles;
        const unsigned f0 = tile * 16u;
#ifdef SGX_ABL_L2LOAD  // timing experiment: every load hits L2 (4 signals = 2.5 MB)
        const float *xb = (const float *)a.x + (size_t)(b & 3u) * a.sample_stride;
#else
        const float *xb = (const float *)a.x + (size_t)b * a.sample_stride;
#endif
        const long long tile_lo = (long long)f0 * a.hop - (long long)a.pad;
        const long long tile_hi = (long long)(f0 + 15u) * a.hop - (long long)a.pad + 1024;
        const bool interior = tile_lo >= 0 && tile_hi <= (long long)a.n_samples;  // wave-uniform
        if constexpr (ROUNDS > 0) {
#ifdef SGX_ABL_NOGLOAD
            for (int r = 0; r < ROUNDS; ++r) creg[r] = (v4f){(float)w, 1.f, 2.f, (float)r};
            return;
#endif
            if (interior) {
                const v4f *xp = (const v4f *)(xb + tile_lo) + tid;
#pragma unroll
                for (int r = 0; r < ROUNDS; ++r)
                    if (r * 256u + tid < chunks) creg[r] = xp[r * 256];
            } else {  // edge tile: zero padding (S1) by predication
                const long long n = (long long)a.n_samples;
#pragma unroll
                for (int r = 0; r < ROUNDS; ++r) {
                    const long long sx = tile_lo + 4ll * (r * 256u + tid);
                    v4f c;
                    c.x = (sx >= 0 && sx < n) ? xb[sx] : 0.0f;
                    c.y = (sx + 1 >= 0 && sx + 1 < n) ? xb[sx + 1] : 0.0f;
                    c.z = (sx + 2 >= 0 && sx + 2 < n) ? xb[sx + 2] : 0.0f;
                    c.w = (sx + 3 >= 0 && sx + 3 < n) ? xb[sx + 3] : 0.0f;
                    creg[r] = c;
                }
            }
        } else {
            const long long s0 = (long long)(f0 + p1f) * a.hop - (long long)a.pad + 2 * n2;
#ifdef SGX_ABL_NOLOAD
            if (true) {
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) xr[n1] = (v2f){(float)(s0 + n1), 1.0f};
            } else
#endif
            if (!interior) {  // edge tile: zero padding (S1) by predication
                const long long n = (long long)a.n_samples;
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) {
                    const long long sx = s0 + 32 * n1;
                    xr[n1].x = (sx >= 0 && sx < n) ? xb[sx] : 0.0f;
                    xr[n1].y = (sx + 1 >= 0 && sx + 1 < n) ? xb[sx + 1] : 0.0f;
                }
                if constexpr (MODE != OUT_MEL) {
                    // consume the edge tile's samples here (2 tiles in 40): with these predicated loads still pending at the
                    // join, the compiler protects the interior path's loads into the same registers with s_waitcnt vmcnt(0),
                    // i.e. a full drain of the previous tile's stores on EVERY tile
#pragma unroll
                    for (int n1 = 0; n1 < 32; ++n1) asm volatile("" : "+v"(xr[n1]));
                }
            } else {
                const v2f *xp = (const v2f *)(xb + s0);
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) xr[n1] = xp[16 * n1];
            }
        }
    };

    // (HALVES = 2) a second half without a tile of its own (odd run length: the last round of an XCD) repeats the first half's
    // tile — same values to the same addresses — so both halves always run the same number of rounds and barriers
    if (HALVES == 2 && wid >= hi) wid = lead;
    if (wid < hi) load_tile(wid, cregA);
    if constexpr (AHEAD2) {  // second round's tile
        const unsigned lead1 = lead + slots * HALVES;
        unsigned w1 = lead1 + half;
        if (HALVES == 2 && w1 >= hi) w1 = lead1;
        if (lead1 < hi) load_tile(w1, cregB);
    }
    if constexpr (ROUNDS == 0 && MODE != OUT_MEL) {
        // "use" the first tile's samples here: the compiler then waits for these loads in the prologue, and inside the
        // loop every sample load is followed by this half's >= 32 unconditional output stores — which lets it emit
        // s_waitcnt vmcnt(32) at the top of the loop instead of vmcnt(0).  gfx9-family hardware counts loads and stores
        // with ONE in-order counter: with vmcnt(0) every wave waited for all of its stores to be acknowledged by a
        // write-saturated memory system before it began the next tile.
#pragma unroll
        for (int n1 = 0; n1 < 32; ++n1) asm volatile("" : "+v"(xr[n1]));
    }
    __syncthreads();  // tables visible
    // Phase skew (experiment, SGX_SKEW=n, default 0; HALVES = 2 with the per-half pass 2): the second half enters the loop `skew`
    // barriers late and the first half leaves it `skew` barriers late.  s_barrier counts arrivals, not call sites, so from then
    // on the halves meet at every barrier one or more phases apart: while one half is in an LDS-bound phase (staging, column /
    // window reads, row reads) the other is in an arithmetic one on the same SIMDs.  Every hazard the barriers protect is
    // internal to a half (own ex buffer).  Measured SLOWER at every skew (linear 139 -> 164 / 202 / 209 us for 1 / 2 / 3): each
    // interval then lasts as long as the longer of two different phases, and a phase run by one wave per SIMD is barely
    // shorter than the same phase run by two in lockstep — the phases are latency-bound per wave, not throughput-bound.
    if (HALVES == 2 && half == 1u)
        for (unsigned q = 0; q < skew; ++q) __syncthreads();
#ifdef SGX_STAMPS
    unsigned long long st_acc[8] = {0}, st_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif

    auto round = [&](v4f (&creg)[NCR]) {
        const unsigned b = wid / a.tiles, tile = wid - b * a.tiles;
        const unsigned f0 = tile * 16u;
        const unsigned nf = min(16u, a.n_frames - f0);
        // (LATEBAR) the barrier that frees ex sits here, AFTER the previous tile's pass 2, not between its row reads and its
        // arithmetic: a wave starts its 16-point transforms as soon as its own rows have arrived instead of meeting the other
        // seven at a barrier first, and the row-read latency overlaps the first transform
        if constexpr (LATEBAR) __syncthreads();
        v2f wn[32];
        auto read_window = [&]() {
            const v2f *w2 = (const v2f *)(tabs + kWinOff) + n2;
#pragma unroll
#ifdef SGX_ABL_NOWIN
            for (int n1 = 0; n1 < 32; ++n1) wn[n1] = (v2f){0.5f, 0.25f};
#else
            for (int n1 = 0; n1 < 32; ++n1) wn[n1] = w2[16 * n1];
#endif
        };
#ifdef SGX_ABL_NOXS
        for (int n1 = 0; n1 < 32; ++n1) xr[n1] = (v2f){creg[n1 % ROUNDS].x + n1, creg[n1 % ROUNDS].y};
        if constexpr (false) {
#else
        if constexpr (ROUNDS > 0) {
#endif
            // stage: chunk c of the tile -> xs (this half's ex is free: barrier 2 of the previous tile / the prologue)
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const unsigned c = r * 256u + tid;
                if (c < chunks) *(v4f *)(smem + c * 16u + (xs_pad ? (c >> 6) * 128u : 0u)) = creg[r];
            }
            SGX_STAMP(7);  // wait for the tile's samples + stage writes
            if constexpr (LATEBAR) read_window();  // table reads fill the short interval between the two barriers
            __syncthreads();
            const unsigned o = p1f * a.hop + 2u * n2;  // float offset of this lane's column inside the tile
            if (xs_pad) {
                const unsigned char *src = smem + o * 4u + p1f * (a.hop >> 8) * 128u;
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) xr[n1] = *(const v2f *)(src + n1 * 128 + (n1 >> 3) * 128);
            } else {
                const unsigned char *src = smem + o * 4u;
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) xr[n1] = *(const v2f *)(src + n1 * 128);
            }
        }
        {
            if constexpr (!(LATEBAR && ROUNDS > 0)) read_window();
#ifndef SGX_ABL_NOXS
            if constexpr (ROUNDS > 0) __syncthreads();
#endif  // every wave has read xs: pass 1 may overwrite it with ex
            SGX_STAMP(0);
            pass1_compute(xr, wn, twa, twb, smem + p1f * kFS + n2 * 8);
            SGX_STAMP(1);
        }
        unsigned next = lead + slots * HALVES + half;
        if (HALVES == 2 && next >= hi) next -= half;  // no tile of its own next round: repeat the first half's
        // requested after pass 1 so the previous tile's store burst has had that long to drain: a vector load issued while
        // the CU's store FIFO is backed up stalls its wave for thousands of cycles
        if constexpr (AHEAD2) {  // the set staged above is free: request the tile of the round after next
            const unsigned lead2 = lead + 2u * slots * HALVES;
            unsigned w2 = lead2 + half;
            if (HALVES == 2 && w2 >= hi) w2 = lead2;
            if (lead2 < hi) load_tile(w2, creg);
        } else {
            if (next < hi) load_tile(next, creg);  // in flight during pass 2
        }
        SGX_STAMP(2);
        __syncthreads();
        SGX_STAMP(3);
        // Linear / complex outputs: every lane runs pass 2 and stores, so the compiler can count the stores behind the next
        // tile's loads (see the prologue).  A lane whose frame does not exist (last tile of a signal) mirrors the tile's last
        // frame, an idle second half (odd tile count) mirrors the first half's tile: same values to the same addresses.
        constexpr bool ALLSTORE = MODE != OUT_MEL;
        // WIDE: this lane's frame belongs to the first half's tile (`lead`) or, for p2f >= 16, to the second half's (`lead + 1`,
        // if it exists — otherwise the lane mirrors the first tile).  All of it is two uniform decodes and a per-lane select.
        unsigned p2b = b, p2ofs, p2ex;
        if constexpr (WIDE) {
            const unsigned w1 = lead + 1u < hi ? lead + 1u : lead;  // = the second half's wid
            const unsigned b0 = lead / a.tiles, f00 = (lead - b0 * a.tiles) * 16u;
            const unsigned b1 = w1 / a.tiles, f01 = (w1 - b1 * a.tiles) * 16u;
            const unsigned nf0 = min(16u, a.n_frames - f00), nf1 = min(16u, a.n_frames - f01);
            const bool second = p2f >= 16u && w1 != lead;
            const unsigned fle = min(p2f & 15u, (second ? nf1 : nf0) - 1u);
            p2ex = (second ? 16u : 0u) + fle;
            // one uniform base (signal b0) for the whole workgroup; a second tile in the next signal is one signal further
            // (513 n_frames elements: the host guarantees 2 * 513 * n_frames * 8 < 2^32)
            p2ofs = (second ? f01 : f00) + fle + ((second && b1 != b0) ? 513u * a.n_frames : 0u);
            p2b = b0;
        } else {
            const unsigned p2f_eff = ALLSTORE ? min(p2f, nf - 1u) : p2f;
            p2ex = p2f_eff;
            p2ofs = f0 + p2f_eff;
        }
        const unsigned char *ex_src = WIDE ? smem_all : smem;
        v2f A[16], B[16];
        read_rows(ex_src + p2ex * kFS, ra, rb, A, B);
        SGX_STAMP(4);
        if constexpr (!LATEBAR) __syncthreads();  // ex consumed: the next pass 1 (or the pw overlay) may overwrite it
        SGX_STAMP(5);
        if constexpr (MODE == OUT_MEL) {  // pw rows are 516 floats wide: bins 513..515 are read with zero weights
            if (tid < 48u) ((float *)smem)[(tid / 3u) * kPS + 513u + tid % 3u] = 0.0f;
        }
        if (ALLSTORE || p2f < nf) {
            const v2f *t2 = (const v2f *)(tabs + kTw2Off);
            auto tw = [&](int i) {  // read from LDS where consumed (this kernel has no registers to keep them)
#ifdef SGX_ABL_NOTW2
                return (v2f){1.f, 0.5f};
#endif
                return i < 8 ? t2[(j == 0 ? 16u : j) * kTw2Stride + i] : t2[j == 0 ? (unsigned)(i - 8) : j * kTw2Stride + i];
            };
            pass2_compute<MODE, AMP>(a, A, B, p2b, p2ofs, 0u, j, eps, tw, jo, (float *)smem + p2f * kPS);
        }
        if constexpr (MODE == OUT_MEL) {
            __syncthreads();
#ifndef SGX_ABL_NOMELTILE
            {
                if (a.mm_frag) map_tile_mfma<AMP>(a, (const float *)smem, b, f0, nf, eps, tid, 2u * half);
                else if (mel_lds) mel_tile_lds<AMP>(a, (const float *)smem, lw4, lptr, lcol, b, f0, nf, eps, tid, 256u);
                else mel_tile<AMP>(a, (const float *)smem, b, f0, nf, eps, tid, 256u);
            }
#endif
            __syncthreads();  // pw consumed before the next pass 1 overwrites ex
        }
        SGX_STAMP(6);
        wid = next;
        lead += slots * HALVES;
    };
    while (lead < hi) {
        round(cregA);
        if constexpr (AHEAD2) {
            if (!(lead < hi)) break;
            round(cregB);
        }
    }
    if (HALVES == 2 && half == 0u)
        for (unsigned q = 0; q < skew; ++q) __syncthreads();
#ifdef SGX_STAMPS
    if ((threadIdx.x & 63u) == 0) {
        for (int q = 0; q < 7; ++q) atomicAdd(&g_stamps[q], st_acc[q]);
        atomicAdd(&g_stamps[7], 1ull);
        atomicAdd(&g_stamps[16], st_acc[7]);
    }
#endif
}

// ====================================================================================================================
// k_ws: wave-specialised pipeline.  512 threads: waves 0-3 = producers (pass 1), waves 4-7 = consumers (pass 2, real
// split, stores, and the global->LDS staging of the samples two tiles ahead).  One persistent workgroup per CU.
// A workgroup's k-th tile is processed by the producers in tick k and by the consumers in tick k+1:
//
//   tick t      producers                                   consumers
//   ---------   -----------------------------------------   ----------------------------------------------------
//   phase R     read column of tile t from xs, window       read rows of tile t-1 from ex[(t-1)&1]
//   barrier M   (xs and ex[(t-1)&1] are now free)
//   phase C     write xs <- chunks of tile t+1 (registers),     FFT16 x2, real split, store tile t-1
//               request chunks of tile t+2 from HBM,
//               FFT32, twiddle, write ex[t&1]
//   [Mel: barrier X, consumers reduce pw (in ex[(t-1)&1]) to Mel bands]
//   barrier E   (ex[t&1] and xs(t+1) complete)
// ====================================================================================================================
template <int MODE, int AMP>
__global__ __launch_bounds__(512, 2) void k_ws(StftArgs a, unsigned per_xcd, unsigned total, unsigned slots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x;
    const bool producer = __builtin_amdgcn_readfirstlane(tid) < 256u;  // wave-uniform, in an SGPR
    const unsigned rt = tid & 255u;                                    // thread index inside the role

    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned lo = xcd * per_xcd;
    const unsigned hi = min(lo + per_xcd, total);
    const unsigned first = lo + slot;
    const unsigned ntiles = first < hi ? (hi - first + slots - 1) / slots : 0u;  // tiles of this workgroup

    const unsigned p1f = rt >> 4, n2 = rt & 15u;                                       // producer identity
    const unsigned lane = rt & 63u, wv_ = rt >> 6, jq = lane >> 4, p2f = lane & 15u;  // consumer identity
    const unsigned j = wv_ + 4u * jq;
    const unsigned ra = j, rb = j == 0 ? 16u : 32u - j;
    const float eps = (float)a.eps;
    const unsigned chunks = (15u * a.hop + 1024u) >> 2;
    const bool xs_pad = (a.hop & 255u) == 0;  // +128 B per KiB keeps the 4 frames of a wave on distinct banks

    // The roles run two separate loops with the same barrier sequence (s_barrier counts waves, not call sites), so
    // the register allocator sees each role's live ranges on their own.
#ifdef SGX_STAMPS
    unsigned long long st_acc[8] = {0}, st_prev = 0;
#endif
    if (producer) {
        // ============================================================ producers: pass 1 of tile t in tick t
        v2f twa[4], twb[8], wn[32];  // lane-constant tables stay in registers for the whole kernel
        load_tw1(a, n2, twa, twb);
        {
            const v2f *w2 = (const v2f *)a.window + n2;  // pre-scaled by 1/2
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) wn[n1] = w2[16 * n1];
        }
        v4f creg[kWsRounds];  // this thread's 16-byte chunks of the tile being prefetched
        auto fetch_chunks = [&](unsigned k) {  // request the k-th tile of this workgroup from HBM
            const unsigned w = first + k * slots;
            const unsigned b = w / a.tiles, tile = w - b * a.tiles;
            const unsigned f0 = tile * 16u;
            const float *xb = (const float *)a.x + (size_t)b * a.sample_stride;
            const long long tile_lo = (long long)f0 * a.hop - (long long)a.pad;
            const long long tile_hi = (long long)(f0 + 15u) * a.hop - (long long)a.pad + 1024;
            if (tile_lo >= 0 && tile_hi <= (long long)a.n_samples) {  // interior tile (wave-uniform)
                const v4f *xp = (const v4f *)(xb + tile_lo) + rt;
#pragma unroll
                for (int r = 0; r < kWsRounds; ++r)
                    if (r * 256u + rt < chunks) creg[r] = xp[r * 256];
            } else {  // edge tile: zero padding (S1) by predication
                const long long n = (long long)a.n_samples;
#pragma unroll
                for (int r = 0; r < kWsRounds; ++r) {
                    const long long sx = tile_lo + 4ll * (r * 256u + rt);
                    v4f c;
                    c.x = (sx >= 0 && sx < n) ? xb[sx] : 0.0f;
                    c.y = (sx + 1 >= 0 && sx + 1 < n) ? xb[sx + 1] : 0.0f;
                    c.z = (sx + 2 >= 0 && sx + 2 < n) ? xb[sx + 2] : 0.0f;
                    c.w = (sx + 3 >= 0 && sx + 3 < n) ? xb[sx + 3] : 0.0f;
                    creg[r] = c;
                }
            }
        };
        auto stage_chunks = [&]() {  // registers -> xs
#pragma unroll
            for (int r = 0; r < kWsRounds; ++r) {
                const unsigned c = r * 256u + rt;
                if (c < chunks) *(v4f *)(smem + kWsXsOff + c * 16u + (xs_pad ? (c >> 6) * 128u : 0u)) = creg[r];
            }
        };
        // prologue: xs <- tile 0, registers <- tile 1
        if (ntiles > 0) {
            fetch_chunks(0);
            stage_chunks();
        }
        if (ntiles > 1) fetch_chunks(1);
        __syncthreads();  // prologue barrier: xs(0) visible
#ifdef SGX_STAMPS
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
        for (unsigned t = 0; t <= ntiles; ++t) {
            v2f xr[32];
            if (t < ntiles) {
                const unsigned o = p1f * a.hop + 2u * n2;  // float offset of this lane's column inside the tile
                if (xs_pad) {
                    const unsigned char *src = smem + kWsXsOff + o * 4u + p1f * (a.hop >> 8) * 128u;
#pragma unroll
                    for (int n1 = 0; n1 < 32; ++n1) xr[n1] = *(const v2f *)(src + n1 * 128 + (n1 >> 3) * 128);
                } else {
                    const unsigned char *src = smem + kWsXsOff + o * 4u;
#pragma unroll
                    for (int n1 = 0; n1 < 32; ++n1) xr[n1] = *(const v2f *)(src + n1 * 128);
                }
            }
            SGX_STAMP(0);
            __syncthreads();  // barrier M: xs consumed
            SGX_STAMP(1);
            if (t + 1 < ntiles) stage_chunks();      // xs <- tile t+1 (requested from HBM one tick ago)
            SGX_STAMP(4);
            if (t + 2 < ntiles) fetch_chunks(t + 2);  // in flight for a whole tick
            SGX_STAMP(5);
            if (t < ntiles) pass1_compute(xr, wn, twa, twb, smem + (t & 1u) * kExBytes + p1f * kFS + n2 * 8);
            if constexpr (MODE == OUT_MEL) __syncthreads();  // barrier X
            SGX_STAMP(2);
            __syncthreads();  // barrier E: ex[t&1] complete
            SGX_STAMP(3);
        }
    } else {
        // ============================================================ consumers: pass 2 + stores of tile t-1 in tick t
        v4f tw[16];  // lane-constant real-split twiddles stay in registers for the whole kernel
        load_job_twiddles((const v4f *)a.tw2, j, tw);
        const JobOfs jo = job_offsets(j, a.n_frames);
        __syncthreads();  // prologue barrier
#ifdef SGX_STAMPS
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
        for (unsigned t = 0; t <= ntiles; ++t) {
            unsigned char *ex_r = smem + ((t & 1u) ^ 1u) * kExBytes;
            v2f A[16], B[16];
            if (t >= 1) read_rows(ex_r + p2f * kFS, ra, rb, A, B);
            SGX_STAMP(0);
            __syncthreads();  // barrier M: ex_r consumed, xs free
            SGX_STAMP(1);
            unsigned cb = 0, cf0 = 0, cnf = 0;
            if (t >= 1) {
                const unsigned w = first + (t - 1) * slots;
                cb = w / a.tiles;
                cf0 = (w - cb * a.tiles) * 16u;
                cnf = min(16u, a.n_frames - cf0);
                if (p2f < cnf)
                    pass2_compute<MODE, AMP>(a, A, B, cb, cf0, p2f, j, eps, [&](int i) { return (v2f){tw[i].x, tw[i].y}; }, jo,
                                             (float *)ex_r + p2f * kPS);
            }
            if constexpr (MODE == OUT_MEL) {
                __syncthreads();  // barrier X: pw (in ex_r) complete
                if (t >= 1) mel_tile<AMP>(a, (const float *)ex_r, cb, cf0, cnf, eps, rt, 256u);
            }
            SGX_STAMP(2);
            __syncthreads();  // barrier E: xs(t+1) complete
            SGX_STAMP(3);
        }
    }
#ifdef SGX_STAMPS
    if ((tid & 63u) == 0) {
        const int base = producer ? 0 : 8;
        for (int q = 0; q < 7; ++q) atomicAdd(&g_stamps[base + q], st_acc[q]);
        atomicAdd(&g_stamps[base + 7], 1ull);
    }
#endif
}

template <typename K>
hipError_t set_lds_once(K kernel, int bytes, bool &) { return set_max_dynamic_lds((const void *)kernel, bytes); }

template <int MODE, int AMP>
hipError_t launch_variant(const StftArgs &a, hipStream_t s) {
    const unsigned total = a.tiles * a.batch;
    const unsigned per_xcd = (total + 7) / 8;
    // the wave-specialised kernel needs 16-byte aligned rows (x base and row stride), hop % 4 == 0 and a tile of
    // at most 5 x 256 chunks (hop <= 272)
    const bool aligned16 = (reinterpret_cast<uintptr_t>(a.x) % 16 == 0) && (a.sample_stride % 4 == 0) && (a.hop % 4 == 0);
    const unsigned chunks = (15u * a.hop + 1024u) >> 2;
    hipError_t e;
    // Measured on MI355X (profiles/r01_*): the wave-specialised pipeline is not yet faster than the two-pass kernel —
    // the CU's vector-memory FIFO is in order, so the producers' sample loads queue behind the consumers' HBM-bound
    // stores.  It stays selectable for tuning (SGX_KERNEL=ws) until that is solved.
    static const bool want_ws = [] {
        const char *v = getenv("SGX_KERNEL");
        return v && v[0] == 'w' && v[1] == 's';
    }();
    if (want_ws && aligned16 && chunks <= (unsigned)kWsRounds * 256u) {
        static bool done = false;
        if ((e = set_lds_once(k_ws<MODE, AMP>, kWsLds, done)) != hipSuccess) return e;
        const unsigned slots = per_xcd < 32u ? per_xcd : 32u;  // one workgroup per CU: 32 per XCD
        hipLaunchKernelGGL((k_ws<MODE, AMP>), dim3(slots * 8), dim3(512), kWsLds, s, a, per_xcd, total, slots);
    } else {
        constexpr int lds = MODE == OUT_MEL ? kLdsMel : kLds;
        static const bool want_single = [] {
            const char *v = getenv("SGX_KERNEL");
            return v && v[0] == 's';  // "single": two independent 256-thread workgroups per CU (round-1 first design)
        }();
        // Sample loads: "direct" = per-lane float2 loads (4 x 128-byte segments per wave-instruction, every line requested ~4
        // times because frames overlap by 75 %); "staged" = the tile's samples fetched once with coalesced 16-byte loads,
        // staged in LDS and re-read per frame from there (5 load instructions per thread instead of 32 through the CU's
        // in-order vector-memory pipe).  Measured on MI355X (256 x 10 s): Mel-dB 162 us staged vs 186 us direct; linear power
        // 147 us staged vs 171 us direct — once the loop no longer drained its stores every tile (vmcnt(32), see the kernel);
        // before that fix staged was the slower one (189 us).  Complex output: 277 us staged vs 258 us direct at that time, 226 vs
        // 240 us since the staging pad became a template parameter (40 VGPRs freed).  SGX_LOADS=staged|direct overrides.
        static const int loads_mode = [] {
            const char *v = getenv("SGX_LOADS");
            return !v ? 0 : v[0] == 's' ? 1 : v[0] == 'd' ? 2 : 0;
        }();
        // (late round 1: with the staging pad a template parameter the staged complex variant no longer runs out of registers and
        // wins too — 226 us vs 240 us — so every output stages its samples by default)
        const bool want_staged = loads_mode != 2;
        const bool stage5 = want_staged && aligned16 && chunks <= 5u * 256u;
        if (want_single) {
            static bool done = false;
            const unsigned slots = per_xcd < 64u ? per_xcd : 64u;  // two workgroups per CU (LDS-limited)
            if (stage5 && (a.hop & 255u) == 0) {
                if ((e = set_lds_once(k_r32x16<MODE, AMP, 1, 5>, lds, done)) != hipSuccess) return e;
                hipLaunchKernelGGL((k_r32x16<MODE, AMP, 1, 5>), dim3(slots * 8), dim3(256), lds, s, a, per_xcd, total, slots, 0u);
            } else {
                if ((e = set_lds_once(k_r32x16<MODE, AMP, 1, 0>, lds, done)) != hipSuccess) return e;
                hipLaunchKernelGGL((k_r32x16<MODE, AMP, 1, 0>), dim3(slots * 8), dim3(256), lds, s, a, per_xcd, total, slots, 0u);
            }
        } else {
            const unsigned pairs = (per_xcd + 1) / 2;
            const unsigned slots = pairs < 32u ? pairs : 32u;  // one 512-thread workgroup per CU
            // pass 2 across both halves (32-frame rows per store instruction) for the per-bin outputs; SGX_WIDE=0 keeps the
            // per-half mapping for A/B runs
            static const unsigned skew = [] {
                const char *v = getenv("SGX_SKEW");
                return v ? (unsigned)atoi(v) % 8u : 0u;
            }();
            static const bool want_wide = [] {
                const char *v = getenv("SGX_WIDE");
                return !(v && v[0] == '0') && skew == 0u;  // the wide pass 2 reads both halves' ex buffers: lockstep only
            }();
            constexpr bool CAN_WIDE = MODE != OUT_MEL;
            auto go = [&](auto kernel) -> hipError_t {
                static bool done = false;
                hipError_t e2 = set_lds_once(kernel, lds + kExBytes, done);
                if (e2 != hipSuccess) return e2;
                hipLaunchKernelGGL(kernel, dim3(slots * 8), dim3(512), lds + kExBytes, s, a, per_xcd, total, slots, skew);
                return hipSuccess;
            };
            if (stage5 && (a.hop & 255u) == 0) {
                if (CAN_WIDE && want_wide) e = go(k_r32x16<MODE, AMP, 2, 5, CAN_WIDE, true>);
                else e = go(k_r32x16<MODE, AMP, 2, 5, false, true>);
            } else if (stage5) {
                if (CAN_WIDE && want_wide) e = go(k_r32x16<MODE, AMP, 2, 5, CAN_WIDE, false>);
                else e = go(k_r32x16<MODE, AMP, 2, 5, false, false>);
            } else {
                if (CAN_WIDE && want_wide) e = go(k_r32x16<MODE, AMP, 2, 0, CAN_WIDE>);
                else e = go(k_r32x16<MODE, AMP, 2, 0, false>);
            }
            if (e != hipSuccess) return e;
        }
    }
    return hipGetLastError();
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
    if (a.n_fft != 1024 || (a.hop & 1u)) return false;
    if (a.n_samples >= (1ull << 40)) return false;
    if ((unsigned long long)a.n_frames * 513ull * 8ull >= 0x7fffffffull) return false;  // 32-bit byte offsets
    a.ft = 16;
    return true;
}

hipError_t launch_r32x16_f32(const StftArgs &a, hipStream_t s) {
    const unsigned long long total64 = (unsigned long long)a.tiles * a.batch;
    if (total64 == 0 || total64 >= 0x7ffffff0ull) return hipErrorInvalidConfiguration;
    static const bool want_q = [] {
        const char *v = getenv("SGX_KERNEL");
        return v && v[0] == 'q';  // experimental 16-values-per-lane kernel (kernels_q16x32.hip)
    }();
    if (want_q && q16x32_takes(a)) return launch_q16x32_f32(a, s);
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
