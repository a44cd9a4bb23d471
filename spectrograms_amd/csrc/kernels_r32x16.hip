// kernels_r32x16.hip — tuned f32, n_fft = 1024 STFT kernel for gfx950 (the BASELINE shape).
//
// Structure (one 256-thread workgroup = 4 wave64; persistent, loops over tiles of 16 consecutive frames of one signal):
//
//   pass 1  lane (f = tid/16, n2 = tid%16) owns z[16*n1 + n2], n1 = 0..31, of frame f, where
//           z[n] = w[2n] x[2n] + i w[2n+1] x[2n+1] is the half-length complex sequence of the real frame
//           (window pre-scaled by 1/2 on the host — exact — so the real split needs no halving).
//           float2 global loads (16 lanes = 128 contiguous bytes; overlapping frames are served by L1/L2), issued one
//           tile ahead so HBM latency hides under pass 2; window multiply fused into the first butterflies; one
//           32-point FFT entirely in registers; twiddle by W_512^(k1*n2); one ds_write_b64 per value.
//   LDS     ex[f][k1][n2] complex f32, frame stride 4096+16 B.  This is the ONLY exchange of the transform.
//           Window and twiddle tables also live in LDS (loaded once per persistent workgroup).
//   pass 2  lane (jq = lane/16, f = lane%16) of wave w owns "job" j = w + 4*jq of frame f: rows k1 = j and
//           32-j (job 0: rows 0 and 16).  8 + 8 ds_read_b128 (conflict-free by the frame-stride / job-to-wave
//           choice), two 16-point FFTs in registers -> Z[j+32*k2], Z[32-j+32*k2], and — because a job holds both
//           members of every (k, 512-k) pair — the real split X[k] = E + W_1024^k O entirely in registers.
//   store   the 16 lanes of a job hold the same bin of 16 consecutive frames, so out[b][k][f0..f0+15] is one
//           contiguous 64-byte segment: the frame-contiguous layout of the reference (S9) needs no LDS
//           transpose.  Mel: |X|^2 goes to LDS pw[f][k] (overlaying ex), then a (mel, frame)-per-lane CSR
//           reduction in ascending-bin order (spectrogram.rs:102-117) and the dB/sqrt epilogue.
//
// All complex arithmetic is written on 2-float vectors so it compiles to packed-f32 VALU (v_pk_add/mul/fma_f32 with
// op_sel / neg modifiers): measured on MI355X a packed op issues at the same cost as a scalar one for a single wave,
// which matters because the 64 KiB exchange buffer caps occupancy at 2 waves per SIMD (tools/ubench/valu_rate.hip).
//
// Reference semantics implemented: spectrogram.rs:1301-1334 (framing, window, R2C, |.|^2), :1845-1865,
// :2068-2080; replaces the per-frame `R2cPlan::process` call at :1323 (fft_backend.rs:423-431).
#include "sgx_internal.h"

namespace sgx {
namespace {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr double kCos64[64] = {1.0, 0.9951847266721969, 0.9807852804032304, 0.9569403357322088, 0.9238795325112867, 0.881921264348355, 0.8314696123025452, 0.773010453362737, 0.7071067811865476, 0.6343932841636455, 0.5555702330196023, 0.4713967368259978, 0.38268343236508984, 0.29028467725446233, 0.19509032201612833, 0.09801714032956077, 6.123233995736766e-17, -0.09801714032956065, -0.1950903220161282, -0.29028467725446216, -0.3826834323650897, -0.4713967368259977, -0.555570233019602, -0.6343932841636454, -0.7071067811865475, -0.773010453362737, -0.8314696123025453, -0.8819212643483549, -0.9238795325112867, -0.9569403357322088, -0.9807852804032304, -0.9951847266721968, -1.0, -0.9951847266721969, -0.9807852804032304, -0.9569403357322089, -0.9238795325112868, -0.881921264348355, -0.8314696123025455, -0.7730104533627371, -0.7071067811865477, -0.6343932841636459, -0.5555702330196022, -0.47139673682599786, -0.38268343236509034, -0.29028467725446244, -0.19509032201612866, -0.09801714032956045, -1.8369701987210297e-16, 0.09801714032956009, 0.1950903220161283, 0.29028467725446205, 0.38268343236509, 0.4713967368259976, 0.5555702330196018, 0.6343932841636456, 0.7071067811865474, 0.7730104533627367, 0.8314696123025452, 0.8819212643483548, 0.9238795325112865, 0.9569403357322088, 0.9807852804032303, 0.9951847266721969};
constexpr double kSin64[64] = {0.0, 0.0980171403295606, 0.19509032201612825, 0.29028467725446233, 0.3826834323650898, 0.47139673682599764, 0.5555702330196022, 0.6343932841636455, 0.7071067811865475, 0.773010453362737, 0.8314696123025452, 0.8819212643483549, 0.9238795325112867, 0.9569403357322089, 0.9807852804032304, 0.9951847266721968, 1.0, 0.9951847266721969, 0.9807852804032304, 0.9569403357322089, 0.9238795325112867, 0.881921264348355, 0.8314696123025455, 0.7730104533627371, 0.7071067811865476, 0.6343932841636455, 0.5555702330196022, 0.47139673682599786, 0.3826834323650899, 0.2902846772544624, 0.1950903220161286, 0.09801714032956083, 1.2246467991473532e-16, -0.09801714032956059, -0.19509032201612836, -0.2902846772544621, -0.38268343236508967, -0.47139673682599764, -0.555570233019602, -0.6343932841636453, -0.7071067811865475, -0.7730104533627367, -0.8314696123025452, -0.8819212643483549, -0.9238795325112865, -0.9569403357322088, -0.9807852804032303, -0.9951847266721969, -1.0, -0.9951847266721969, -0.9807852804032304, -0.9569403357322089, -0.9238795325112866, -0.881921264348355, -0.8314696123025455, -0.7730104533627369, -0.7071067811865477, -0.6343932841636459, -0.5555702330196022, -0.4713967368259979, -0.3826834323650904, -0.2902846772544625, -0.19509032201612872, -0.0980171403295605};

constexpr int kFS = 4096 + 16;        // LDS bytes per frame of ex (odd multiple of 16 -> conflict-free b128 reads)
constexpr int kPS = 513;              // floats per frame of pw (overlays ex)
constexpr int kExBytes = 16 * kFS;    // 65792
constexpr int kWinOff = kExBytes;     // float  win[1024]        (0.5 * window)
constexpr int kTw2Off = kWinOff + 4096;  // float4 tw2[17][17]   (wr, wi, wi, -wr) of W_1024^(row + 32*idx)
constexpr int kTw2Stride = 17;           // row stride in float4 (bank spread between the 4 jobs of a wave)
constexpr int kTw2Bytes = 17 * 17 * 16;  // 4624
constexpr int kLds = kTw2Off + kTw2Bytes;  // 78608 B -> two workgroups per CU (160 KiB LDS)

__device__ __forceinline__ v2f swp(v2f a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ v2f lo2(v2f a) { return __builtin_shufflevector(a, a, 0, 0); }
__device__ __forceinline__ v2f hi2(v2f a) { return __builtin_shufflevector(a, a, 1, 1); }
__device__ __forceinline__ v2f pfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// x0 = e + W o, x1 = e - W o, W = W_N^K a compile-time constant; every case is 2-4 packed instructions
template <int N, int K>
__device__ __forceinline__ void bfly(v2f e, v2f o, v2f &x0, v2f &x1) {
    constexpr int idx = K * (64 / N);
    if constexpr (idx == 0) {
        x0 = e + o;
        x1 = e - o;
    } else if constexpr (idx == 16) {  // W = -i : W o = (o.y, -o.x)
        const v2f so = swp(o);
        x0 = pfma(so, (v2f){1.f, -1.f}, e);
        x1 = pfma(so, (v2f){-1.f, 1.f}, e);
    } else if constexpr (idx == 8) {  // W = c(1 - i): W o = c (o.x + o.y, o.y - o.x)
        constexpr float c = 0.70710678118654752440f;
        const v2f s = pfma(swp(o), (v2f){1.f, -1.f}, o);
        x0 = pfma(s, (v2f){c, c}, e);
        x1 = pfma(s, (v2f){-c, -c}, e);
    } else if constexpr (idx == 24) {  // W = c(-1 - i): W o = c (o.y - o.x, -o.x - o.y)
        constexpr float c = 0.70710678118654752440f;
        const v2f s = pfma(swp(o), (v2f){1.f, -1.f}, -o);
        x0 = pfma(s, (v2f){c, c}, e);
        x1 = pfma(s, (v2f){-c, -c}, e);
    } else {
        constexpr float wr = (float)kCos64[idx], wi = (float)(-kSin64[idx]);
        const v2f so = swp(o);
        x0 = pfma(so, (v2f){-wi, wi}, pfma(o, (v2f){wr, wr}, e));
        x1 = pfma(so, (v2f){wi, -wi}, pfma(o, (v2f){-wr, -wr}, e));
    }
}
template <int N, int K>
struct Comb {
    static __device__ __forceinline__ void run(v2f (&x)[N], const v2f (&e)[N / 2], const v2f (&o)[N / 2]) {
        bfly<N, K>(e[K], o[K], x[K], x[K + N / 2]);
        if constexpr (K + 1 < N / 2) Comb<N, K + 1>::run(x, e, o);
    }
};
// in-register radix-2 DIT, natural order in and out; all indices and twiddles are compile-time.
// WIN: x holds raw samples and w the window; the multiply is fused into the first butterfly.
template <int N, bool WIN>
struct Fft {
    static __device__ __forceinline__ void run(v2f (&x)[N], const v2f (&w)[N]) {
        if constexpr (N == 2) {
            if constexpr (WIN) {
                const v2f t = x[0] * w[0];
                const v2f u = x[1];
                x[0] = pfma(u, w[1], t);
                x[1] = pfma(-u, w[1], t);
            } else {
                const v2f a = x[0], b = x[1];
                x[0] = a + b;
                x[1] = a - b;
            }
        } else {
            v2f e[N / 2], o[N / 2], we[N / 2], wo[N / 2];
#pragma unroll
            for (int k = 0; k < N / 2; ++k) {
                e[k] = x[2 * k];
                o[k] = x[2 * k + 1];
                we[k] = w[2 * k];
                wo[k] = w[2 * k + 1];
            }
            Fft<N / 2, WIN>::run(e, we);
            Fft<N / 2, WIN>::run(o, wo);
            Comb<N, 0>::run(x, e, o);
        }
    }
};

#ifdef SGX_STAMPS  // diagnostic build only (tools/stamps.py): where does a wave spend its cycles, per tile phase
__device__ unsigned long long g_stamps[16];
#define SGX_STAMP(i)                                                                         \
    do {                                                                                     \
        unsigned long long t_;                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        st_acc[i] += t_ - st_prev;                                                           \
        st_prev = t_;                                                                        \
    } while (0)
#else
#define SGX_STAMP(i)
#endif

template <int AMP>
__device__ __forceinline__ float amp_f32(float p, float eps) {
    if constexpr (AMP == AMP_MAGNITUDE) return sqrtf(p);
    else if constexpr (AMP == AMP_DB) return 10.0f * log10f(fmaxf(p, eps));
    else return p;
}

__device__ __forceinline__ float power_of(v2f x) { return __builtin_fmaf(x.x, x.x, x.y * x.y); }

// Persistent workgroups: each loops over its tiles; the next tile's samples are requested from HBM right after
// pass 1 has consumed the current ones, so their latency hides under pass 2 (2 waves/SIMD cannot hide it by
// occupancy alone: the 64 KiB exchange buffer limits a CU to two workgroups).
// ROUNDS > 0: the tile's (15*hop + 1024) samples are fetched ONCE with coalesced 16-byte loads (ROUNDS per thread,
// issued one tile ahead), staged in LDS (xs, overlaying ex) and re-read per frame from there: per-lane float2 loads
// straight from global re-request every line ~4 times (frames overlap by 75 %) and their issue stalls on the L1 miss
// queue (tools/stamps.py).  ROUNDS == 0 keeps the direct per-lane loads (any even hop).
template <int MODE, int AMP, int ROUNDS>
__global__ __launch_bounds__(256, 2) void k_r32x16(StftArgs a, unsigned per_xcd, unsigned total, unsigned slots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x;

    // ---- one-time: tables -> LDS
    {
        ((v4f *)(smem + kWinOff))[tid] = ((const v4f *)a.window)[tid];
        for (unsigned i = tid; i < kTw2Bytes / 16; i += 256) ((v4f *)(smem + kTw2Off))[i] = ((const v4f *)a.tw2)[i];
    }

    // XCD-aware work mapping: blocks g and g+8 share an XCD (round-robin dispatch).  XCD x owns the contiguous run of
    // work ids [x*per_xcd, (x+1)*per_xcd); its `slots` resident workgroups walk that run with stride `slots`, so tiles
    // in flight on one XCD are neighbours: they share the 768-sample halo and the output lines they both touch in L2.
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned lo = xcd * per_xcd;
    const unsigned hi = min(lo + per_xcd, total);
    unsigned wid = lo + slot;

    const unsigned p1f = tid >> 4, n2 = tid & 15u;  // pass-1 identity
    const unsigned lane = tid & 63u, wv_ = tid >> 6, jq = lane >> 4, p2f = lane & 15u;  // pass-2 identity
    const unsigned j = wv_ + 4u * jq;
    const bool j0 = (j == 0);
    const unsigned ra = j, rb = j0 ? 16u : 32u - j;
    const float eps = (float)a.eps;

    // pass-1 twiddles W_512^(k1*n2), k1 = 8a + b, kept in registers as two short per-lane tables (10 values) instead
    // of 31 LDS reads per tile: W(k1) = Wa[a] * Wb[b]
    v2f twa[4], twb[8];
    {
        const v2f *t1 = (const v2f *)a.tw1 + n2;
#pragma unroll
        for (int q = 0; q < 4; ++q) twa[q] = t1[16 * 8 * q];
#pragma unroll
        for (int q = 0; q < 8; ++q) twb[q] = t1[16 * q];
    }

    v2f xr[32];                          // raw samples of this lane's (frame, n2) column
    v4f creg[ROUNDS > 0 ? ROUNDS : 1];   // staged path: this thread's 16-byte chunks of the tile being prefetched
    const unsigned chunks = (15u * a.hop + 1024u) >> 2;
    const bool xs_pad = (a.hop & 255u) == 0;  // +128 B per KiB keeps the 4 frames of a wave on distinct banks
    auto load_tile = [&](unsigned w) {
        const unsigned b = w / a.tiles, tile = w - b * a.tiles;
        const unsigned f0 = tile * 16u;
        const float *xb = (const float *)a.x + (size_t)b * a.sample_stride;
        const long long tile_lo = (long long)f0 * a.hop - (long long)a.pad;
        const long long tile_hi = (long long)(f0 + 15u) * a.hop - (long long)a.pad + 1024;
        const bool interior = tile_lo >= 0 && tile_hi <= (long long)a.n_samples;  // wave-uniform
        if constexpr (ROUNDS > 0) {
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
            if (interior) {
                const v2f *xp = (const v2f *)(xb + s0);
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) xr[n1] = xp[16 * n1];
            } else {
                const long long n = (long long)a.n_samples;
#pragma unroll
                for (int n1 = 0; n1 < 32; ++n1) {
                    const long long sx = s0 + 32 * n1;
                    xr[n1].x = (sx >= 0 && sx < n) ? xb[sx] : 0.0f;
                    xr[n1].y = (sx + 1 >= 0 && sx + 1 < n) ? xb[sx + 1] : 0.0f;
                }
            }
        }
    };

    if (wid < hi) load_tile(wid);
    __syncthreads();  // tables visible
#ifdef SGX_STAGGER  // experiment: offset the two co-resident workgroups of a CU by about half a tile period
    if (slot & 1u) {
        for (int q = 0; q < SGX_STAGGER; ++q) __builtin_amdgcn_s_sleep(127);
    }
#endif
#ifdef SGX_STAMPS
    unsigned long long st_acc[12] = {0}, st_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif

    while (wid < hi) {
        SGX_STAMP(0);
        const unsigned b = wid / a.tiles, tile = wid - b * a.tiles;
        const unsigned f0 = tile * 16u;
        const unsigned nf = min(16u, a.n_frames - f0);

        // ------------------------------------------------------------------ pass 1
        if constexpr (ROUNDS > 0) {
            // stage: chunk c of the tile -> xs (ex is free here: barrier 2 of the previous tile / the prologue barrier)
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const unsigned c = r * 256u + tid;
                if (c < chunks) *(v4f *)(smem + c * 16u + (xs_pad ? (c >> 6) * 128u : 0u)) = creg[r];
            }
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
            v2f v[32], wn[32];
            const v2f *w2 = (const v2f *)(smem + kWinOff) + n2;
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) wn[n1] = w2[16 * n1];
            if constexpr (ROUNDS > 0) {
                __syncthreads();  // every wave has read xs: pass 1 may overwrite it with ex
                if (wid + slots < hi) load_tile(wid + slots);  // next tile's chunks: in flight for the whole tile
            }
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) v[n1] = xr[n1];
#ifdef SGX_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            SGX_STAMP(1);
#endif
#ifndef SGX_ABL_NOFFT32  // timing experiment only
            Fft<32, true>::run(v, wn);
#endif
#ifdef SGX_STAMPS
#pragma unroll
            for (int q = 0; q < 32; ++q) asm volatile("" : "+v"(v[q]));
            SGX_STAMP(2);
#endif
            unsigned char *dst = smem + p1f * kFS + n2 * 8;
            auto cm = [](v2f x, v2f t) { return pfma(swp(x), (v2f){-t.y, t.y}, x * lo2(t)); };
#pragma unroll
            for (int k1 = 0; k1 < 32; ++k1) {
                const int qa = k1 >> 3, qb = k1 & 7;
                v2f r = v[k1];
#ifndef SGX_ABL_NOTW1
                if (qb) r = cm(r, twb[qb]);
                if (qa) r = cm(r, twa[qa]);
#endif
                *(v2f *)(dst + k1 * 128) = r;
            }
        }
        SGX_STAMP(3);
        const unsigned next = wid + slots;
        if constexpr (ROUNDS == 0) {
            if (next < hi) load_tile(next);  // in flight during pass 2
        }
        SGX_STAMP(4);
#ifndef SGX_ABL_NOBARRIER
        __syncthreads();
#endif

        SGX_STAMP(5);
        // ------------------------------------------------------------------ pass 2 + real split + epilogue
        v2f A[16], B[16];
        {
            const v4f *pa = (const v4f *)(smem + p2f * kFS + ra * 128);
            const v4f *pb = (const v4f *)(smem + p2f * kFS + rb * 128);
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
        SGX_STAMP(6);
#ifndef SGX_ABL_NOBARRIER
        __syncthreads();  // ex consumed: the next pass 1 (or the pw overlay) may overwrite it
#endif
        SGX_STAMP(7);

        if (p2f < nf) {
#ifndef SGX_ABL_NOFFT16  // timing experiment only
            Fft<16, false>::run(A, A);
            Fft<16, false>::run(B, B);
#endif
#ifdef SGX_STAMPS
#pragma unroll
            for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(A[q]), "+v"(B[q]));
            SGX_STAMP(8);
#endif

            const v4f *t2 = (const v4f *)(smem + kTw2Off);
            float *pw = (float *)smem + p2f * kPS;
            // 32-bit element offsets from a wave-uniform base (host guarantees 513*n_frames*2 < 2^31)
            float *ob = (float *)a.out + ((size_t)b * 513u) * a.n_frames * (MODE == OUT_COMPLEX ? 2 : 1);
            const unsigned ofs = f0 + p2f;

            auto emit = [&](unsigned k, v2f X, bool conj) {
#ifdef SGX_ABL_NOSTORE  // timing experiment only (tools/ablate.sh): keep the value alive, drop the store
                asm volatile("" ::"v"(X), "v"(k));
                return;
#endif
                if constexpr (MODE == OUT_COMPLEX) {
                    ((v2f *)ob)[k * a.n_frames + ofs] = conj ? (v2f){X.x, -X.y} : X;
                } else if constexpr (MODE == OUT_MEL) {
                    pw[k] = power_of(X);
                } else {
                    ob[k * a.n_frames + ofs] = amp_f32<AMP>(power_of(X), eps);
                }
            };
            // pair (P, Q) = (Z[k], Z[512-k]), W = W_1024^k given as (wr, wi, wi, -wr):
            //   E = (P.x+Q.x, P.y-Q.y), D = (P.x-Q.x, P.y+Q.y) = (-O.y, O.x), T = W O, X[k] = E + T, X[512-k] = conj(E - T)
            auto split = [&](unsigned k, v2f P, v2f Q, v4f w4) {
                const v2f E = pfma(Q, (v2f){1.f, -1.f}, P);
                const v2f D = pfma(Q, (v2f){-1.f, 1.f}, P);
                const v2f T = pfma(lo2(D), (v2f){w4.z, w4.w}, hi2(D) * (v2f){w4.x, w4.y});
                emit(k, E + T, false);
                emit(512u - k, E - T, true);
            };
#ifdef SGX_ABL_NOSPLIT
            emit(j, A[0] + B[1] + A[5] + B[7] + A[15] + B[12], false);
#else
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // general job: (A[i], B[15-i]) at k = j + 32 i;  job 0: (B[i], B[15-i]) at k = 16 + 32 i (row 16)
                const v2f P = j0 ? B[i] : A[i];
                const unsigned k = (j0 ? 16u : j) + 32u * i;
                const v4f w4 = t2[(j0 ? 16u : j) * kTw2Stride + i];
                split(k, P, B[15 - i], w4);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                // general job: (A[8+t], B[7-t]) at k = j + 32 (8+t);  job 0: (A[t], A[(16-t)%16]) at k = 32 t (row 0)
                const v2f P = j0 ? A[t] : A[8 + t];
                const v2f Q = j0 ? A[(16 - t) & 15] : B[7 - t];
                const unsigned k = j0 ? 32u * t : j + 32u * (8 + t);
                const v4f w4 = t2[j0 ? (unsigned)t : j * kTw2Stride + 8u + t];
                split(k, P, Q, w4);
            }
            if (j0) emit(256u, A[8] * (v2f){2.f, -2.f}, false);  // bin 256 pairs with itself: X[256] = 2 conj(Z[256])
#endif
        }

        if constexpr (MODE == OUT_MEL) {
            __syncthreads();
            const float *val = (const float *)a.mel_val;
            const float *pwall = (const float *)smem;
            float *o = (float *)a.out + ((size_t)b * a.n_out) * a.n_frames + f0;
            for (unsigned idx = tid; idx < 16u * a.n_mels; idx += 256u) {
                const unsigned ff = idx & 15u, mm = idx >> 4;
                float acc = 0.0f;
                const unsigned i0 = a.mel_ptr[mm], i1 = a.mel_ptr[mm + 1];
                for (unsigned i = i0; i < i1; ++i)
                    acc = __fadd_rn(__fmul_rn(val[i], pwall[ff * kPS + a.mel_col[i]]), acc);
                if (ff < nf) o[mm * a.n_frames + ff] = amp_f32<AMP>(acc, eps);
            }
            __syncthreads();  // pw consumed before the next pass 1 overwrites ex
        }
        SGX_STAMP(9);
        wid = next;
    }
#ifdef SGX_STAMPS
    if ((tid & 63u) == 0) {
        for (int q = 0; q < 10; ++q) atomicAdd(&g_stamps[q], st_acc[q]);
        atomicAdd(&g_stamps[15], 1ull);
    }
#endif
}

template <int MODE, int AMP, int ROUNDS>
hipError_t launch_variant3(const StftArgs &a, hipStream_t s, unsigned per_xcd, unsigned total, unsigned slots) {
    static bool attr_set = false;  // per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)k_r32x16<MODE, AMP, ROUNDS>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((k_r32x16<MODE, AMP, ROUNDS>), dim3(slots * 8), dim3(256), kLds, s, a, per_xcd, total, slots);
    return hipGetLastError();
}

template <int MODE, int AMP>
hipError_t launch_variant(const StftArgs &a, hipStream_t s, unsigned per_xcd, unsigned total, unsigned slots) {
    // staged loads need 16-byte aligned rows (x base and row stride) and hop % 4 == 0; ROUNDS = chunks per thread
    const bool aligned16 = (reinterpret_cast<uintptr_t>(a.x) % 16 == 0) && (a.sample_stride % 4 == 0) && (a.hop % 4 == 0);
    const unsigned chunks = (15u * a.hop + 1024u) >> 2;
    if (aligned16 && chunks <= 5u * 256u) return launch_variant3<MODE, AMP, 5>(a, s, per_xcd, total, slots);
    if (aligned16 && chunks <= 9u * 256u) return launch_variant3<MODE, AMP, 9>(a, s, per_xcd, total, slots);
    return launch_variant3<MODE, AMP, 0>(a, s, per_xcd, total, slots);
}

}  // namespace

#ifdef SGX_STAMPS
extern "C" int sgx_debug_read_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

bool plan_geometry_r32x16_f32(StftArgs &a) {
    if (a.n_fft != 1024 || (a.hop & 1u)) return false;
    if (a.n_samples >= (1ull << 40)) return false;
    if ((unsigned long long)a.n_frames * 513ull * 2ull >= 0x7fffffffull) return false;  // 32-bit store offsets
    a.ft = 16;
    return true;
}

hipError_t launch_r32x16_f32(const StftArgs &a, hipStream_t s) {
    const unsigned long long total64 = (unsigned long long)a.tiles * a.batch;
    if (total64 == 0 || total64 >= 0x7ffffff0ull) return hipErrorInvalidConfiguration;
    const unsigned total = (unsigned)total64;
    const unsigned per_xcd = (total + 7) / 8;
    // persistent grid: 2 workgroups per CU (LDS-limited) on 256 CUs = 64 slots per XCD
    const unsigned slots = per_xcd < 64u ? per_xcd : 64u;
    if (a.out_mode == OUT_COMPLEX) return launch_variant<OUT_COMPLEX, AMP_POWER>(a, s, per_xcd, total, slots);
    if (a.out_mode == OUT_MEL) {
        if (a.amp == AMP_MAGNITUDE) return launch_variant<OUT_MEL, AMP_MAGNITUDE>(a, s, per_xcd, total, slots);
        if (a.amp == AMP_DB) return launch_variant<OUT_MEL, AMP_DB>(a, s, per_xcd, total, slots);
        return launch_variant<OUT_MEL, AMP_POWER>(a, s, per_xcd, total, slots);
    }
    if (a.amp == AMP_MAGNITUDE) return launch_variant<OUT_LINEAR, AMP_MAGNITUDE>(a, s, per_xcd, total, slots);
    if (a.amp == AMP_DB) return launch_variant<OUT_LINEAR, AMP_DB>(a, s, per_xcd, total, slots);
    return launch_variant<OUT_LINEAR, AMP_POWER>(a, s, per_xcd, total, slots);
}

}  // namespace sgx
