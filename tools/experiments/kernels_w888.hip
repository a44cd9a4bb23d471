// kernels_w888.hip — wave-per-frame STFT kernel for Mel-type outputs (f32, n_fft = 1024): Mel / log-Hz banks on the LDS band
// table, i.e. BASELINE configs 2-4.
//
// Why a second design: k_r32x16 keeps 16 frames of one bin in 16 adjacent lanes so that a [bin][frame] row is stored in
// 64-byte runs — that is what a 513-row linear output needs, and it costs workgroup-wide barriers and an exchange buffer that
// caps the CU at 2 waves per SIMD, both in lockstep.  A Mel output has 80 rows, not 513: the coalescing constraint is gone,
// so here ONE WAVE owns a frame end to end and nothing ever synchronises across waves:
//
//   load      lane t reads z[t + 64 a], a = 0..7 (z[n] = x[2n] + i x[2n+1]): eight 512-byte coalesced loads per frame, the
//             next frame's eight are in flight while this one is transformed
//   FFT       512 = 8 x 8 x 8.  n = 64 n2 + 8 n1 + n0, k = k2 + 8 k1 + 64 k0:
//               pass 1  FFT8 over n2 (window fused), twiddle W_64^(n1 k2)        lane = 8 n1 + n0
//               pass 2  FFT8 over n1, twiddle W_512^(n0 (k2 + 8 k1))              lane = 8 k2 + n0
//               pass 3  FFT8 over n0 -> Z[lane + 64 k0]                           lane = 8 k1 + k2
//             between the passes the wave transposes 8 x 8 x 8 through its PRIVATE 5 KB of LDS (rows padded to 80 bytes:
//             the four ds_read_b128 of a lane's row are conflict-free); a wave's LDS operations execute in order, so the
//             exchanges need no barrier
//   split     lane t already holds Z[t + 64 j]; the partners Z[512 - k] (k = t + 64 j, j < 4) come from lane 64 - t through
//             LDS (upper half of Z only), X[k] and X[512 - k] in registers, |X|^2 -> pw[0..512] over the dead exchange data
//   bank      lane m reduces band m (and m + 64) from the padded 4-wide band table with the reference's sequential, unfused
//             accumulation (spectrogram.rs:102-117), keeps 8 frames of results in registers and stores them as two 16-byte
//             runs per band row
//
// Occupancy is set by registers alone (~100 VGPRs -> 4-5 waves per SIMD) and every wave is at a different point of its frame,
// so LDS latency, the dependent FFT chains and the loads hide behind each other without any software pipelining.
#include <cstdlib>

#include "fft_inreg.h"
#include "sgx_internal.h"

namespace sgx {
namespace {

using namespace inreg;
typedef float v4f __attribute__((ext_vector_type(4)));
struct Cx2f {
    float re, im;
};

#ifndef SGX_W888_WAVES
#define SGX_W888_WAVES 4
#endif
#ifndef SGX_W888_MINW
#define SGX_W888_MINW 3
#endif
constexpr int kWaves = SGX_W888_WAVES;  // waves per workgroup (independent of each other; they only share the band table)
constexpr int kMinWavesPerSimd = SGX_W888_MINW;  // register budget: 512 / kMinWavesPerSimd VGPRs per lane
constexpr int kRow = 80;                // bytes per exchange row: 8 complex + 16 B pad
constexpr int kXBytes = 64 * kRow;      // 5120 B per wave
constexpr int kFT = 8;                  // frames per wave-tile (its outputs are stored as 32-byte runs per band row)
constexpr int kMaxRows = 128;           // bank rows a wave can reduce (2 per lane)
constexpr int kMaxChunks = 512;         // 4-wide chunks of the padded band table held in LDS
constexpr int kTabBytes = kMaxChunks * 16 + (2 * kMaxRows + 1) * 4 + 12;
constexpr int kTwBytes = 19 * 64 * 8;    // per-lane twiddles shared by the workgroup's waves: [7 t1 | 8 t2 | 4 ws][64 lanes]

template <int AMP>
__device__ __forceinline__ float amp_w(float p, float eps) {
    if constexpr (AMP == AMP_MAGNITUDE) return sqrtf(p);
    else if constexpr (AMP == AMP_DB) return 10.0f * log10f(fmaxf(p, eps));
    else return p;
}

template <int AMP>
__global__ __launch_bounds__(64 * kWaves, kMinWavesPerSimd) void k_w888(StftArgs a, unsigned per_xcd, unsigned total, unsigned slots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // shared, read-only: padded band table  w4[chunks] | pptr[n_mels + 1] | pcol[n_mels]
    v4f *lw4 = (v4f *)smem;
    unsigned *lptr = (unsigned *)(lw4 + a.mel_pchunks);
    unsigned *lcol = lptr + a.n_mels + 1;
    for (unsigned i = threadIdx.x; i < a.mel_pchunks; i += 64 * kWaves) lw4[i] = ((const v4f *)a.mel_pw)[i];
    for (unsigned i = threadIdx.x; i <= a.n_mels; i += 64 * kWaves) lptr[i] = a.mel_pptr[i];
    for (unsigned i = threadIdx.x; i < a.n_mels; i += 64 * kWaves) lcol[i] = a.mel_pcol[i];
    v2f *ltw = (v2f *)(smem + kTabBytes);                                  // [19][64] twiddles by lane
    unsigned char *xb_lds = smem + kTabBytes + kTwBytes + wave * kXBytes;  // this wave's private exchange buffer

    // per-lane constants
    const Cx2f *twn = (const Cx2f *)a.tw;  // e^{-2 pi i k / 1024}, 1024 entries (W_512^e = twn[2 e])
    const unsigned hi3 = lane >> 3, lo3 = lane & 7u;
    v2f win[8];   // 0.5 * window at samples 2 (lane + 64 a), +1 (the 1/2 makes the real split a plain sum)
    {
        const v2f *w2 = (const v2f *)a.window + lane;
#pragma unroll
        for (int q = 0; q < 8; ++q) win[q] = w2[64 * q];
    }
    if (wave == 0) {  // twiddles by lane, read back just in time every frame: registers are what limits occupancy here
#pragma unroll
        for (int k2 = 1; k2 < 8; ++k2) {  // pass 1: lane = 8 n1 + n0 -> W_64^(n1 k2) = W_1024^(16 n1 k2)
            const Cx2f c = twn[(16u * hi3 * k2) & 1023u];
            ltw[(k2 - 1) * 64 + lane] = (v2f){c.re, c.im};
        }
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) {  // pass 2: lane = 8 k2 + n0 -> W_512^(n0 (k2 + 8 k1)) = W_1024^(2 n0 (k2 + 8 k1))
            const Cx2f c = twn[(2u * lo3 * (hi3 + 8u * k1)) & 1023u];
            ltw[(7 + k1) * 64 + lane] = (v2f){c.re, c.im};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // real split: W_1024^k, k = lane + 64 j
            const Cx2f c = twn[lane + 64u * j];
            ltw[(15 + j) * 64 + lane] = (v2f){c.re, c.im};
        }
    }
    __syncthreads();  // the band table is in place; from here on the waves never meet again

    const float eps = (float)a.eps;
    const unsigned xcd = blockIdx.x & 7u, slot = (blockIdx.x >> 3) * kWaves + wave;
    const unsigned lo = xcd * per_xcd, hi = min(lo + per_xcd, total);
    const long long n = (long long)a.n_samples;

    for (unsigned wid = lo + slot; wid < hi; wid += slots) {
        const unsigned b = wid / a.tiles, tile = wid - b * a.tiles;
        const unsigned f0 = tile * kFT;
        const unsigned nf = min((unsigned)kFT, a.n_frames - f0);
        const float *xb = (const float *)a.x + (size_t)b * a.sample_stride;
        v2f xr[8];
        auto load_frame = [&](unsigned f) {
            const long long s0 = (long long)f * a.hop - (long long)a.pad;  // first sample of the frame
            if (s0 >= 0 && s0 + 1024 <= n) {                                 // wave-uniform
                const v2f *xp = (const v2f *)(xb + s0) + lane;
#pragma unroll
                for (int q = 0; q < 8; ++q) xr[q] = xp[64 * q];
            } else {  // zero padding (S1)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const long long sx = s0 + 2ll * (lane + 64 * q);
                    xr[q].x = (sx >= 0 && sx < n) ? xb[sx] : 0.0f;
                    xr[q].y = (sx + 1 >= 0 && sx + 1 < n) ? xb[sx + 1] : 0.0f;
                }
            }
        };
        load_frame(f0);
#pragma unroll 1
        for (unsigned fi = 0; fi < nf; ++fi) {
            {
                v2f v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = xr[q];
                if (fi + 1 < nf) load_frame(f0 + fi + 1);  // in flight during this frame's transform
                // ---- pass 1: FFT8 over n2 (window fused), twiddle, transpose [k2][n0][n1]
                Fft<8, true>::run(v, win);
                {
                    unsigned char *dst = xb_lds + lo3 * kRow + hi3 * 8;  // row = 8 k2 + n0, column n1
                    *(v2f *)dst = v[0];
#pragma unroll
                    for (int k2 = 1; k2 < 8; ++k2) *(v2f *)(dst + k2 * 8 * kRow) = cmulv(v[k2], ltw[(k2 - 1) * 64 + lane]);
                }
                {
                    const v4f *row = (const v4f *)(xb_lds + lane * kRow);  // lane = 8 k2 + n0: its 8 values over n1
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const v4f q = row[c];
                        v[2 * c] = (v2f){q.x, q.y};
                        v[2 * c + 1] = (v2f){q.z, q.w};
                    }
                }
                // ---- pass 2: FFT8 over n1, twiddle, transpose [k1][k2][n0]
                Fft<8, false>::run(v, v);
                {
                    unsigned char *dst = xb_lds + hi3 * kRow + lo3 * 8;  // row = 8 k1 + k2, column n0
#pragma unroll
                    for (int k1 = 0; k1 < 8; ++k1) *(v2f *)(dst + k1 * 8 * kRow) = cmulv(v[k1], ltw[(7 + k1) * 64 + lane]);
                }
                {
                    const v4f *row = (const v4f *)(xb_lds + lane * kRow);  // lane = 8 k1 + k2: its 8 values over n0
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const v4f q = row[c];
                        v[2 * c] = (v2f){q.x, q.y};
                        v[2 * c + 1] = (v2f){q.z, q.w};
                    }
                }
                // ---- pass 3: FFT8 over n0 -> v[k0] = Z[lane + 64 k0]
                Fft<8, false>::run(v, v);
                // ---- real split.  Upper half of Z to LDS in natural order; partner of k = lane + 64 j is Z[512 - k]
                v2f *zl = (v2f *)xb_lds;  // zl[i] = Z[256 + i]
#pragma unroll
                for (int q = 4; q < 8; ++q) zl[lane + 64 * (q - 4)] = v[q];
                v2f Q[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned qi = (256u - lane - 64u * j) & 255u;  // lane 0: own registers, the read is a dummy
                    Q[j] = zl[qi];
                }
                if (lane == 0) {  // 512 - 64 j = 64 (8 - j): Z[0] for j = 0, else register 8 - j of this lane
                    Q[0] = v[0];
                    Q[1] = v[7];
                    Q[2] = v[6];
                    Q[3] = v[5];
                }
                float *pw = (float *)xb_lds;  // overlays zl: every read above is complete before these writes (in-order LDS)
                float pk[4], pq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const v2f P = v[j], Qv = Q[j], w = ltw[(15 + j) * 64 + lane];
                    const v2f E = pfma(Qv, (v2f){1.f, -1.f}, P);
                    const v2f D = pfma(Qv, (v2f){-1.f, 1.f}, P);
                    const v2f T = pfma(D, hi2(w), (v2f){D.y, -D.x} * lo2(w));  // W O, O = (D.y, -D.x)
                    const v2f X0 = E + T, X1 = E - T;
                    pk[j] = __builtin_fmaf(X0.x, X0.x, X0.y * X0.y);
                    pq[j] = __builtin_fmaf(X1.x, X1.x, X1.y * X1.y);
                }
                const float p256 = 4.0f * __builtin_fmaf(v[4].x, v[4].x, v[4].y * v[4].y);  // X[256] = 2 conj(Z[256]) (lane 0)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pw[lane + 64 * j] = pk[j];
                    pw[512u - lane - 64u * j] = pq[j];
                }
                if (lane == 0) pw[256] = p256;
                if (lane < 3) pw[513 + lane] = 0.0f;  // the 4-wide band chunks may reach bins 513..515 with zero weights
                // ---- bank: lane m reduces rows m and m + 64
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const unsigned mm = lane + 64u * s;
                    float acc = 0.0f;
                    if (mm < a.n_mels) {
                        const unsigned c0 = lptr[mm], c1 = lptr[mm + 1];
                        const v4f *p4 = (const v4f *)(pw + lcol[mm]);
                        for (unsigned c = c0; c < c1; ++c) {
                            const v4f w = lw4[c], p = p4[c - c0];
                            acc = __fadd_rn(__fmul_rn(w.x, p.x), acc);
                            acc = __fadd_rn(__fmul_rn(w.y, p.y), acc);
                            acc = __fadd_rn(__fmul_rn(w.z, p.z), acc);
                            acc = __fadd_rn(__fmul_rn(w.w, p.w), acc);
                        }
                    }
                    if (mm < a.n_mels) ((float *)a.out)[((size_t)b * a.n_out + mm) * a.n_frames + f0 + fi] = amp_w<AMP>(acc, eps);
                }
            }
        }
    }
}

}  // namespace

bool w888_supported(const StftArgs &a) {
    return a.n_fft == 1024 && !(a.hop & 1u) && a.out_mode == OUT_MEL && a.amp != AMP_MAG_IN && a.mel_pw && !a.mm_frag &&
           a.n_mels <= (unsigned)kMaxRows && a.mel_pchunks <= (unsigned)kMaxChunks;
}

hipError_t launch_w888(const StftArgs &a0, hipStream_t s) {
    StftArgs a = a0;
    a.tiles = (a.n_frames + kFT - 1) / kFT;
    const unsigned long long total64 = (unsigned long long)a.tiles * a.batch;
    if (total64 == 0 || total64 >= 0x7ffffff0ull) return hipErrorInvalidConfiguration;
    const unsigned total = (unsigned)total64, per_xcd = (total + 7) / 8;
    const int lds = kTabBytes + kTwBytes + kWaves * kXBytes;
    // persistent waves: 32 CUs per XCD x 2 workgroups x kWaves waves each walk the XCD's contiguous run of tiles
    const unsigned wgs_per_xcd = 32u * (4u * kMinWavesPerSimd / kWaves);  // 32 CUs x resident workgroups per CU
    const unsigned need = (per_xcd + kWaves - 1) / kWaves;
    const unsigned wgs = need < wgs_per_xcd ? (need ? need : 1) : wgs_per_xcd;
    const unsigned slots = wgs * kWaves;
#define SGX_W888(AMPV)                                                                                             \
    do {                                                                                                           \
        hipError_t e = set_max_dynamic_lds((const void *)k_w888<AMPV>, lds);                                      \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL((k_w888<AMPV>), dim3(wgs * 8), dim3(64 * kWaves), lds, s, a, per_xcd, total, slots);    \
    } while (0)
    if (a.amp == AMP_MAGNITUDE) SGX_W888(AMP_MAGNITUDE);
    else if (a.amp == AMP_DB) SGX_W888(AMP_DB);
    else SGX_W888(AMP_POWER);
#undef SGX_W888
    return hipGetLastError();
}

}  // namespace sgx
