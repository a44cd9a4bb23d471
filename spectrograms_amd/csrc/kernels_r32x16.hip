// kernels_r32x16.hip — tuned f32, n_fft = 1024 STFT kernel for gfx950 (the BASELINE shape).
//
// Structure (one 256-thread workgroup = 4 wave64 = one tile of 16 consecutive frames of one signal):
//
//   pass 1  lane (f = tid/16, n2 = tid%16) owns z[16*n1 + n2], n1 = 0..31, of frame f, where
//           z[n] = w[2n] x[2n] + i w[2n+1] x[2n+1] is the half-length complex sequence of the real frame
//           (window pre-scaled by 1/2 on the host — exact — so the real split needs no halving).
//           float2 global loads (16 lanes = 128 contiguous bytes; overlapping frames are served by L1/L2),
//           one 32-point FFT entirely in registers, twiddle by W_512^(k1*n2), one ds_write_b64 per value.
//   LDS     ex[f][k1][n2] complex f32, frame stride 4096+16 B.  This is the ONLY exchange of the transform.
//   pass 2  lane (jq = lane/16, f = lane%16) of wave w owns "job" j = w + 4*jq of frame f: rows k1 = j and
//           32-j (job 0: rows 0 and 16).  8 + 8 ds_read_b128 (conflict-free by the frame-stride / job-to-wave
//           choice), two 16-point FFTs in registers -> Z[j+32*k2], Z[32-j+32*k2], and — because a job holds both
//           members of every (k, 512-k) pair — the real split X[k] = E + W_1024^k O entirely in registers.
//   store   the 16 lanes of a job hold the same bin of 16 consecutive frames, so out[b][k][f0..f0+15] is one
//           contiguous 64-byte segment: the frame-contiguous layout of the reference (S9) needs no LDS
//           transpose.  Mel: |X|^2 goes to LDS pw[f][k] (overlaying ex), then a (mel, frame)-per-lane CSR
//           reduction in ascending-bin order (spectrogram.rs:102-117) and the dB/sqrt epilogue.
//
// Reference semantics implemented: spectrogram.rs:1301-1334 (framing, window, R2C, |.|^2), :1845-1865,
// :2068-2080; replaces the per-frame `R2cPlan::process` call at :1323 (fft_backend.rs:423-431).
#include "sgx_internal.h"

namespace sgx {
namespace {

constexpr double kCos64[64] = {1.0, 0.9951847266721969, 0.9807852804032304, 0.9569403357322088, 0.9238795325112867, 0.881921264348355, 0.8314696123025452, 0.773010453362737, 0.7071067811865476, 0.6343932841636455, 0.5555702330196023, 0.4713967368259978, 0.38268343236508984, 0.29028467725446233, 0.19509032201612833, 0.09801714032956077, 6.123233995736766e-17, -0.09801714032956065, -0.1950903220161282, -0.29028467725446216, -0.3826834323650897, -0.4713967368259977, -0.555570233019602, -0.6343932841636454, -0.7071067811865475, -0.773010453362737, -0.8314696123025453, -0.8819212643483549, -0.9238795325112867, -0.9569403357322088, -0.9807852804032304, -0.9951847266721968, -1.0, -0.9951847266721969, -0.9807852804032304, -0.9569403357322089, -0.9238795325112868, -0.881921264348355, -0.8314696123025455, -0.7730104533627371, -0.7071067811865477, -0.6343932841636459, -0.5555702330196022, -0.47139673682599786, -0.38268343236509034, -0.29028467725446244, -0.19509032201612866, -0.09801714032956045, -1.8369701987210297e-16, 0.09801714032956009, 0.1950903220161283, 0.29028467725446205, 0.38268343236509, 0.4713967368259976, 0.5555702330196018, 0.6343932841636456, 0.7071067811865474, 0.7730104533627367, 0.8314696123025452, 0.8819212643483548, 0.9238795325112865, 0.9569403357322088, 0.9807852804032303, 0.9951847266721969};
constexpr double kSin64[64] = {0.0, 0.0980171403295606, 0.19509032201612825, 0.29028467725446233, 0.3826834323650898, 0.47139673682599764, 0.5555702330196022, 0.6343932841636455, 0.7071067811865475, 0.773010453362737, 0.8314696123025452, 0.8819212643483549, 0.9238795325112867, 0.9569403357322089, 0.9807852804032304, 0.9951847266721968, 1.0, 0.9951847266721969, 0.9807852804032304, 0.9569403357322089, 0.9238795325112867, 0.881921264348355, 0.8314696123025455, 0.7730104533627371, 0.7071067811865476, 0.6343932841636455, 0.5555702330196022, 0.47139673682599786, 0.3826834323650899, 0.2902846772544624, 0.1950903220161286, 0.09801714032956083, 1.2246467991473532e-16, -0.09801714032956059, -0.19509032201612836, -0.2902846772544621, -0.38268343236508967, -0.47139673682599764, -0.555570233019602, -0.6343932841636453, -0.7071067811865475, -0.7730104533627367, -0.8314696123025452, -0.8819212643483549, -0.9238795325112865, -0.9569403357322088, -0.9807852804032303, -0.9951847266721969, -1.0, -0.9951847266721969, -0.9807852804032304, -0.9569403357322089, -0.9238795325112866, -0.881921264348355, -0.8314696123025455, -0.7730104533627369, -0.7071067811865477, -0.6343932841636459, -0.5555702330196022, -0.4713967368259979, -0.3826834323650904, -0.2902846772544625, -0.19509032201612872, -0.0980171403295605};

constexpr int kFS = 4096 + 16;   // LDS bytes per frame of ex (odd multiple of 16 -> conflict-free b128 reads)
constexpr int kPS = 513;         // floats per frame of pw
constexpr int kLds = 16 * kFS;   // 65792 B

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 w) {
    return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
}

// v * W_N^K with the twiddle a compile-time constant (trivial ones cost no multiplies)
template <int N, int K>
__device__ __forceinline__ float2 mul_tw(float2 v) {
    constexpr int idx = K * (64 / N);
    if constexpr (idx == 0) {
        return v;
    } else if constexpr (idx == 16) {
        return make_float2(v.y, -v.x);
    } else if constexpr (idx == 8) {
        constexpr float c = 0.70710678118654752440f;
        return make_float2((v.x + v.y) * c, (v.y - v.x) * c);
    } else if constexpr (idx == 24) {
        constexpr float c = 0.70710678118654752440f;
        return make_float2((v.y - v.x) * c, -(v.x + v.y) * c);
    } else {
        constexpr float wr = (float)kCos64[idx], wi = (float)(-kSin64[idx]);
        return make_float2(v.x * wr - v.y * wi, v.x * wi + v.y * wr);
    }
}

template <int N, int K>
struct Combine {
    static __device__ __forceinline__ void run(float2 (&x)[N], const float2 (&e)[N / 2], const float2 (&o)[N / 2]) {
        const float2 t = mul_tw<N, K>(o[K]);
        x[K] = cadd(e[K], t);
        x[K + N / 2] = csub(e[K], t);
        if constexpr (K + 1 < N / 2) Combine<N, K + 1>::run(x, e, o);
    }
};

// in-register radix-2 DIT, natural order in and out; all indices and twiddles are compile-time
template <int N>
struct Fft {
    static __device__ __forceinline__ void run(float2 (&x)[N]) {
        float2 e[N / 2], o[N / 2];
#pragma unroll
        for (int k = 0; k < N / 2; ++k) {
            e[k] = x[2 * k];
            o[k] = x[2 * k + 1];
        }
        Fft<N / 2>::run(e);
        Fft<N / 2>::run(o);
        Combine<N, 0>::run(x, e, o);
    }
};
template <>
struct Fft<1> {
    static __device__ __forceinline__ void run(float2 (&)[1]) {}
};

__device__ __forceinline__ float2 csel(bool c, float2 a, float2 b) { return make_float2(c ? a.x : b.x, c ? a.y : b.y); }

__device__ __forceinline__ float amp_f32(float p, int amp, float eps) {
    if (amp == AMP_MAGNITUDE) return sqrtf(p);
    if (amp == AMP_DB) return 10.0f * log10f(fmaxf(p, eps));
    return p;
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_r32x16(StftArgs a, unsigned per_xcd, unsigned total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x;
    // XCD-aware work mapping: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a contiguous
    // run of tiles — neighbouring tiles share the 768-sample halo and the output lines they both touch in L2.
    const unsigned wid = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (wid >= total) return;
    const unsigned b = wid / a.tiles, tile = wid - b * a.tiles;
    const unsigned f0 = tile * 16u;
    const unsigned nf = min(16u, a.n_frames - f0);
    const float *xb = (const float *)a.x + (size_t)b * a.sample_stride;

    // ------------------------------------------------------------------ pass 1
    {
        const unsigned f = tid >> 4, n2 = tid & 15u;
        float2 v[32];
        const float2 *w2 = (const float2 *)a.window + n2;  // pre-scaled by 1/2
        const long long s0 = (long long)(f0 + f) * a.hop - (long long)a.pad + 2 * n2;
        const long long tile_lo = (long long)f0 * a.hop - (long long)a.pad;
        const long long tile_hi = (long long)(f0 + 15u) * a.hop - (long long)a.pad + 1024;
        if (tile_lo >= 0 && tile_hi <= (long long)a.n_samples) {  // interior tile (wave-uniform): no bounds checks
            const float2 *xp = (const float2 *)(xb + s0);
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) {
                const float2 xv = xp[16 * n1];
                const float2 wv = w2[16 * n1];
                v[n1] = make_float2(xv.x * wv.x, xv.y * wv.y);
            }
        } else {  // edge tile: zero padding (S1) by predication
            const long long n = (long long)a.n_samples;
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) {
                const long long s = s0 + 32 * n1;
                const float x0 = (s >= 0 && s < n) ? xb[s] : 0.0f;
                const float x1 = (s + 1 >= 0 && s + 1 < n) ? xb[s + 1] : 0.0f;
                const float2 wv = w2[16 * n1];
                v[n1] = make_float2(x0 * wv.x, x1 * wv.y);
            }
        }
        Fft<32>::run(v);
        const float2 *t1 = (const float2 *)a.tw1 + n2;
        unsigned char *dst = smem + f * kFS + n2 * 8;
        *(float2 *)dst = v[0];
#pragma unroll
        for (int k1 = 1; k1 < 32; ++k1) *(float2 *)(dst + k1 * 128) = cmul(v[k1], t1[16 * k1]);
    }
    __syncthreads();

    // ------------------------------------------------------------------ pass 2 + real split + epilogue
    {
        const unsigned lane = tid & 63u, w = tid >> 6, jq = lane >> 4, f = lane & 15u;
        const unsigned j = w + 4u * jq;
        const bool j0 = (j == 0);
        const unsigned ra = j, rb = j0 ? 16u : 32u - j;
        float2 A[16], B[16];
        {
            const float4 *pa = (const float4 *)(smem + f * kFS + ra * 128);
            const float4 *pb = (const float4 *)(smem + f * kFS + rb * 128);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 q = pa[c];
                A[2 * c] = make_float2(q.x, q.y);
                A[2 * c + 1] = make_float2(q.z, q.w);
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 q = pb[c];
                B[2 * c] = make_float2(q.x, q.y);
                B[2 * c + 1] = make_float2(q.z, q.w);
            }
        }
        if constexpr (MODE == OUT_MEL) __syncthreads();  // ex fully consumed before pw overlays it
        Fft<16>::run(A);
        Fft<16>::run(B);

        const float eps = (float)a.eps;
        const float2 *t2 = (const float2 *)a.tw2;
        const bool live = f < nf;
        float *pw = (float *)smem + f * kPS;
        float *ol = (float *)a.out + ((size_t)b * 513u) * a.n_frames + f0 + f;
        float2 *oc = (float2 *)a.out + ((size_t)b * 513u) * a.n_frames + f0 + f;

        auto emit = [&](unsigned k, float re, float im) {
            if constexpr (MODE == OUT_COMPLEX) {
                if (live) oc[(size_t)k * a.n_frames] = make_float2(re, im);
            } else if constexpr (MODE == OUT_MEL) {
                pw[k] = re * re + im * im;
            } else {
                if (live) ol[(size_t)k * a.n_frames] = amp_f32(re * re + im * im, a.amp, eps);
            }
        };
        // pair (P, Q) = (Z[k], Z[512-k]), twiddle W_1024^k:  X[k] = E + W O,  X[512-k] = conj(E - W O)
        auto split = [&](unsigned k, float2 P, float2 Q, float2 wv) {
            const float er = P.x + Q.x, ei = P.y - Q.y, orr = P.y + Q.y, oi = Q.x - P.x;
            const float tr = orr * wv.x - oi * wv.y, ti = orr * wv.y + oi * wv.x;
            emit(k, er + tr, ei + ti);
            emit(512u - k, er - tr, ti - ei);
        };
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // general job: (A[i], B[15-i]) at k = j + 32 i;  job 0: (B[i], B[15-i]) at k = 16 + 32 i (row 16)
            const float2 P = csel(j0, B[i], A[i]);
            const unsigned k = j0 ? 16u + 32u * i : j + 32u * i;
            const float2 wv = t2[j0 ? 16u * 16u + i : j * 16u + i];
            split(k, P, B[15 - i], wv);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            // general job: (A[8+t], B[7-t]) at k = j + 32 (8+t);  job 0: (A[t], A[(16-t)%16]) at k = 32 t (row 0)
            const float2 P = csel(j0, A[t], A[8 + t]);
            const float2 Q = csel(j0, A[(16 - t) & 15], B[7 - t]);
            const unsigned k = j0 ? 32u * t : j + 32u * (8 + t);
            const float2 wv = t2[j0 ? (unsigned)t : j * 16u + 8u + t];
            split(k, P, Q, wv);
        }
        if (j0) {  // bin 256 pairs with itself: X[256] = 2 conj(Z[256]) (Z at half scale)
            emit(256u, 2.0f * A[8].x, -2.0f * A[8].y);
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
                if (ff < nf) o[(size_t)mm * a.n_frames + ff] = amp_f32(acc, a.amp, eps);
            }
        }
    }
}

}  // namespace

bool plan_geometry_r32x16_f32(StftArgs &a) {
    if (a.n_fft != 1024 || (a.hop & 1u)) return false;
    if (a.n_samples >= (1ull << 40)) return false;
    a.ft = 16;
    return true;
}

hipError_t launch_r32x16_f32(const StftArgs &a, hipStream_t s) {
    const unsigned long long total = (unsigned long long)a.tiles * a.batch;
    if (total == 0 || total >= 0x7ffffff0ull) return hipErrorInvalidConfiguration;
    const unsigned per_xcd = (unsigned)((total + 7) / 8);
    const unsigned grid = per_xcd * 8;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e;
        if ((e = hipFuncSetAttribute((const void *)k_r32x16<OUT_LINEAR>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute((const void *)k_r32x16<OUT_MEL>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute((const void *)k_r32x16<OUT_COMPLEX>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds)) != hipSuccess) return e;
        attr_set = true;
    }
    switch (a.out_mode) {
    case OUT_MEL: hipLaunchKernelGGL(k_r32x16<OUT_MEL>, dim3(grid), dim3(256), kLds, s, a, per_xcd, (unsigned)total); break;
    case OUT_COMPLEX: hipLaunchKernelGGL(k_r32x16<OUT_COMPLEX>, dim3(grid), dim3(256), kLds, s, a, per_xcd, (unsigned)total); break;
    default: hipLaunchKernelGGL(k_r32x16<OUT_LINEAR>, dim3(grid), dim3(256), kLds, s, a, per_xcd, (unsigned)total); break;
    }
    return hipGetLastError();
}

}  // namespace sgx
