// kernels_q16x32.hip — f32, n_fft = 1024 STFT kernel with 16 values per lane (EXPERIMENT, SGX_KERNEL=q; parity-green, slower
// than k_r32x16: 174-184 us vs 136-140 us per 256 x 10 s, see DESIGN.md §4 "Second look").  The same transform as k_r32x16
// (kernels_r32x16.hip) laid out for FOUR waves per SIMD instead of two.
//
// k_r32x16 keeps 32 complex values per lane (a 32-point and two 16-point transforms in registers), which costs ~250 VGPRs
// and pins the CU at 2 waves per SIMD.  Here the 512-point complex transform of z[n] = w[2n] x[2n] + i w[2n+1] x[2n+1] is
// split 16 x 32 with the 32 done as 2 x 16 across a lane pair, so no lane ever holds more than 16 values (104-118 VGPRs):
//
//   tile    16 consecutive frames of one signal, one 512-thread workgroup; two workgroups per CU (16 waves).
//   front   lane (f = 0..15, n2 = 0..31) owns z[32 n1 + n2], n1 = 0..15: window fused into a 16-point FFT over n1, twiddle
//           W_512^(k1 n2) (two 3-entry per-lane tables), ds_write_b64 to ex[f][k1][n2]; then wave q = 0..7 reads the row pair
//           (q, 16 - q) (wave 0: rows 0 and 8) of all 16 frames: lane = 4 f + m, quad member m = (row, parity p), 16 values
//           n2 = 2 i + p by 8 conflict-free ds_read_b128.
//   back    16-point FFT over i, the odd lane multiplies by W_32^k2 (compile-time constants), and one DPP exchange with the
//           neighbour (quad_perm [1,0,3,2]) gives Z[k1 + 16 k2]: k2 = 0..15 in the even lane, 16..31 in the odd one.
//           X[k] = E + W_1024^k O needs Z[512 - k], which is register 15 - i of the lane diagonally opposite in the quad
//           (quad_perm [3,2,1,0]): every lane produces its OWN 16 bins from DPP reads, no second exchange.  Wave 0 (rows 0
//           and 8 pair with themselves) uses the neighbour lane and its own index maps; it also stores bin 512.
//   pipe    the samples of tile t + 1 are staged and the loads of tile t + 2 issued before the back half (and the stores) of
//           tile t.
//
// What the measurement says (MI355X): without loads and stores the kernel takes 115 us against k_r32x16's 92-96 us although
// it runs 4 waves per SIMD and its VALU pipe demand is within 8 % — one workgroup per CU (2 waves per SIMD) already reaches
// 130 us, i.e. doubling the occupancy buys 12 %.  Both kernels fit  time = VALU pipe cycles + LDS pipe cycles  per CU round
// (q: 8.4 k + 5.8 k, r32x16: 7.8 k + 5.1 k at the measured issue rates), so occupancy is not what limits them, and q has
// more of both (64 DPP moves per lane, twice the lanes reading window / twiddle tables).  Its stores cost 49 us (r32x16:
// 11-16 us): a quad must sit in 4 consecutive lanes for DPP, so a 16-lane quarter wave holds 4 bins x 4 frames = four
// 16-byte pieces per store instruction instead of one 64-byte run.
//
// Reference semantics: spectrogram.rs:1301-1334 (framing, window, R2C, |.|^2), :2068-2080 (amplitude scaling).
#include <cstdlib>

#include "fft_inreg.h"
#include "sgx_internal.h"

namespace sgx {
namespace {

using namespace inreg;

// ex[f][k1]: a row is 256 B = [the 16 even n2][the 16 odd n2]; rows 8..15 sit 64 B further and a frame is 4368 B
// (273 16-byte slots, = 1 mod 16).  A ds_read_b128 lane group holds 4 frames {0,3,5,6} / {1,2,4,7} (+8) x the 4 quad members,
// whose slots are f + {0, 8} (parity) + {0, 4} (row < 8 / row >= 8: a row pair (q, 16 - q) or (0, 8) always has one of
// each): 16 different slots, so the pass-2 reads are conflict-free; pass-1 lanes are ordered evens-then-odds so each
// 16-lane group of a ds_write_b64 writes 128 contiguous bytes.
constexpr int kQFS = 16 * 256 + 64 + 208;  // 4368 B per frame
constexpr int kQEx = 16 * kQFS;            // 69 888
__device__ __forceinline__ unsigned q_row_off(unsigned r) { return r * 256u + (r >= 8u ? 64u : 0u); }
constexpr int kQWin = kQEx;               // window (pre-scaled by 1/2), 4096 B
constexpr int kQTw = kQWin + 4096;        // W_1024^k, k = 0..512 as (re, im): 4104 B (+ pad)
constexpr int kQLds = kQTw + 4112;        // 78 096 B -> two workgroups per CU
constexpr int kQRounds = 3;               // 16-byte sample chunks per thread per tile (hop <= 341)

__device__ __forceinline__ float dppf(float v, int ctrl_is_x3) {
    // quad_perm [1,0,3,2] = 0xB1 (lane ^ 1), quad_perm [3,2,1,0] = 0x1B (lane ^ 3)
    return ctrl_is_x3 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x1B, 0xF, 0xF, true))
                      : __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
template <int X3> __device__ __forceinline__ v2f dpp2(v2f v) { return (v2f){dppf(v.x, X3), dppf(v.y, X3)}; }

template <int AMP>
__device__ __forceinline__ float q_amp(float p, float eps) {
    if constexpr (AMP == AMP_MAGNITUDE) return sqrtf(p);
    else if constexpr (AMP == AMP_DB) return 10.0f * log10f(fmaxf(p, eps));
    else return p;
}

// odd-parity lanes: R[k2] *= W_32^k2 (compile-time constants)
template <int K>
__device__ __forceinline__ void mul_w32(v2f (&R)[16]) {
    constexpr float wr = (float)kCos64[2 * K], wi = -(float)kSin64[2 * K];  // W = (cos, -sin)
    if constexpr (K > 0) R[K] = pfma(swp(R[K]), (v2f){-wi, wi}, R[K] * (v2f){wr, wr});
    if constexpr (K + 1 < 16) mul_w32<K + 1>(R);
}

template <int MODE, int AMP>
__global__ __launch_bounds__(512, 2) void k_q16x32(StftArgs a, unsigned per_xcd, unsigned total, unsigned slots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x;
    // ---- tables: window (as uploaded: pre-scaled by 1/2) and the split twiddles W_1024^k
    if (tid < 256u) ((v4f *)(smem + kQWin))[tid] = ((const v4f *)a.window)[tid];
    for (unsigned k = tid; k <= 512u; k += 512u) {
        float sn, cs;
        sincospif((float)k * (1.0f / 512.0f), &sn, &cs);  // e^{-2 pi i k / 1024}
        ((v2f *)(smem + kQTw))[k] = (v2f){cs, -sn};
    }
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned lo = xcd * per_xcd, hi = min(lo + per_xcd, total);
    unsigned wid = lo + slot;

    // ---- pass-1 identity and tables
    const unsigned p1f = tid >> 5, n2 = 2u * (tid & 15u) + ((tid >> 4) & 1u);  // lanes 0-15: even n2, 16-31: odd n2
    v2f twa[4], twb[4];  // W_512^(k1 n2) = twa[k1 >> 2] * twb[k1 & 3]
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        float sn, cs;
        sincospif((float)((4 * q * n2) & 511u) * (1.0f / 256.0f), &sn, &cs);
        twa[q] = (v2f){cs, -sn};
        sincospif((float)(q * n2) * (1.0f / 256.0f), &sn, &cs);
        twb[q] = (v2f){cs, -sn};
    }
    // ---- pass-2 identity: wave = row pair, lane = 4 f + m
    const unsigned q = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    const unsigned p2f = lane >> 2, m = lane & 3u, par = m & 1u;
    const unsigned row = q == 0 ? (m < 2u ? 0u : 8u) : (m < 2u ? q : 16u - q);
    const unsigned kbase = row + 256u * par;  // this lane's bins: kbase + 16 i
    const float eps = (float)a.eps;

    v4f creg[kQRounds];
    const unsigned chunks = (15u * a.hop + 1024u) >> 2;
    auto load_tile = [&](unsigned w) {
        const unsigned b = w / a.tiles, tile = w - b * a.tiles;
        const unsigned f0 = tile * 16u;
        const float *xb = (const float *)a.x + (size_t)b * a.sample_stride;
        const long long tile_lo = (long long)f0 * a.hop - (long long)a.pad;
        const long long tile_hi = (long long)(f0 + 15u) * a.hop - (long long)a.pad + 1024;
#ifdef SGX_ABL_NOGLOAD
        for (int r = 0; r < kQRounds; ++r) creg[r] = (v4f){(float)w, 1.f, 2.f, (float)r};
        return;
#endif
        if (tile_lo >= 0 && tile_hi <= (long long)a.n_samples) {  // interior tile (uniform)
            const v4f *xp = (const v4f *)(xb + tile_lo) + tid;
#pragma unroll
            for (int r = 0; r < kQRounds; ++r)
                if (r * 512u + tid < chunks) creg[r] = xp[r * 512];
        } else {  // edge tile: zero padding (S1) by predication
            const long long n = (long long)a.n_samples;
#pragma unroll
            for (int r = 0; r < kQRounds; ++r) {
                const long long sx = tile_lo + 4ll * (r * 512u + tid);
                v4f c;
                c.x = (sx >= 0 && sx < n) ? xb[sx] : 0.0f;
                c.y = (sx + 1 >= 0 && sx + 1 < n) ? xb[sx + 1] : 0.0f;
                c.z = (sx + 2 >= 0 && sx + 2 < n) ? xb[sx + 2] : 0.0f;
                c.w = (sx + 3 >= 0 && sx + 3 < n) ? xb[sx + 3] : 0.0f;
                creg[r] = c;
            }
            // consume the edge tile's samples here (2 tiles in 40): with predicated loads still pending at the join the compiler
            // guards the interior path with s_waitcnt vmcnt(0), i.e. a full drain of the previous tile's stores on EVERY tile
#pragma unroll
            for (int r = 0; r < kQRounds; ++r) asm volatile("" : "+v"(creg[r]));
        }
    };
    auto stage_tile = [&]() {  // registers -> xs (overlays ex, which is free behind the barrier that follows the row reads)
#pragma unroll
        for (int r = 0; r < kQRounds; ++r) {
            const unsigned c = r * 512u + tid;
            if (c < chunks) *(v4f *)(smem + c * 16u) = creg[r];
        }
    };
    v2f R[16];  // pass-2 rows of the tile `cur`
    // front half of a tile: staged samples -> pass 1 -> ex -> this lane's row in R (three barriers; the last one frees ex)
    auto front = [&](unsigned w) {
        const unsigned b = w / a.tiles, tile = w - b * a.tiles;
        const unsigned nf = min(16u, a.n_frames - tile * 16u);
        v2f xr[16], wn[16];
        {
            const unsigned char *src = smem + (p1f * a.hop + 2u * n2) * 4u;
            const v2f *w2 = (const v2f *)(smem + kQWin) + n2;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) xr[n1] = *(const v2f *)(src + n1 * 256);
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) wn[n1] = w2[32 * n1];
        }
        __syncthreads();  // every wave has read xs: pass 1 may overwrite it with ex
        Fft<16, true>::run(xr, wn);
        {
            unsigned char *dst = smem + p1f * kQFS + (n2 & 1u) * 128u + (n2 >> 1) * 8u;
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) {
                const int qa = k1 >> 2, qb = k1 & 3;
                v2f r = xr[k1];
                if (qb) r = cmulv(r, twb[qb]);
                if (qa) r = cmulv(r, twa[qa]);
                *(v2f *)(dst + (k1 * 256 + (k1 >= 8 ? 64 : 0))) = r;
            }
        }
        __syncthreads();
        const unsigned fe = min(p2f, nf - 1u);  // a lane whose frame does not exist mirrors the tile's last frame
        const v4f *src = (const v4f *)(smem + fe * kQFS + q_row_off(row) + par * 128u);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const v4f v = src[c];
            R[2 * c] = (v2f){v.x, v.y};
            R[2 * c + 1] = (v2f){v.z, v.w};
        }
        __syncthreads();  // ex consumed: the next tile's staging may overwrite it
    };
    // back half: 16-point FFT of the row, radix-2 and real split across the quad by DPP, output
    auto back = [&](unsigned w) {
        const unsigned b = w / a.tiles, tile = w - b * a.tiles;
        const unsigned f0 = tile * 16u;
        const unsigned nf = min(16u, a.n_frames - f0);
        const unsigned fe = min(p2f, nf - 1u);
        const v2f *tw = (const v2f *)(smem + kQTw) + kbase;  // split twiddles of this lane's bins: 4 addresses per wave
        Fft<16, false>::run(R, R);
        if (par) mul_w32<0>(R);
        {   // even lane: E + W O (k2 = i), odd lane: E - W O (k2 = 16 + i): other + sgn * mine
            const v2f sg = par ? (v2f){-1.f, -1.f} : (v2f){1.f, 1.f};
#pragma unroll
            for (int i = 0; i < 16; ++i) R[i] = pfma(R[i], sg, dpp2<0>(R[i]));  // Z[row + 16 (i + 16 par)]
        }
        constexpr unsigned ES = MODE == OUT_COMPLEX ? 8u : 4u;
        unsigned char *ob = (unsigned char *)a.out + ((size_t)b * 513u) * a.n_frames * ES;
        const unsigned step = 16u * a.n_frames * ES;
        const unsigned off0 = (kbase * a.n_frames + f0 + fe) * ES;
        auto emit = [&](unsigned off, v2f X) {
#ifdef SGX_ABL_NOSTORE  // timing experiment only (tools/mkvariant.sh)
            asm volatile("" ::"v"(X), "v"(off));
            return;
#endif
            if constexpr (MODE == OUT_COMPLEX) *(v2f *)(ob + off) = X;
            else *(float *)(ob + off) = q_amp<AMP>(__builtin_fmaf(X.x, X.x, X.y * X.y), eps);
        };
        // X[k] = E + W^k O with P = Z[k], Q = Z[512 - k]: E = (P.x + Q.x, P.y - Q.y), D = (P.x - Q.x, P.y + Q.y),
        // O = (D.y, -D.x) (the window carries the 1/2)
        auto xk = [&](v2f P, v2f Qv, v2f w) {
            const v2f E = pfma(Qv, (v2f){1.f, -1.f}, P);
            const v2f D = pfma(Qv, (v2f){-1.f, 1.f}, P);
            const v2f T = pfma(D, hi2(w), (v2f){D.y, -D.x} * lo2(w));
            return E + T;
        };
        if (q != 0) {  // rows (q, 16 - q): the partner of register i is register 15 - i of the diagonal lane
#pragma unroll
            for (int i = 0; i < 16; ++i) emit(off0 + i * step, xk(R[i], dpp2<1>(R[15 - i]), tw[16 * i]));
        } else {
            // rows 0 and 8 pair with themselves: the partner lives in the neighbour lane (parity swapped).
            //   row 8:  bin 8 + 16 k2  <->  8 + 16 (31 - k2): register 15 - i of the neighbour
            //   row 0:  bin 16 k2      <->  16 (32 - k2): register 16 - i of the neighbour for i >= 1; register 0 (bins 0 and
            //           256) pairs with itself
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const v2f n15 = dpp2<0>(R[15 - i]);
                const v2f n16 = i == 0 ? R[0] : dpp2<0>(R[16 - i]);
                emit(off0 + i * step, xk(R[i], m < 2u ? n16 : n15, tw[16 * i]));
            }
            // bin 512 = X[512] = conj(E - T) of the pair (Z[0], Z[0]) with W^0 = 1: real, held by the (row 0, even) lane
            if (m == 0u) {
                const v2f P = R[0];
                emit((512u * a.n_frames + f0 + fe) * ES, (v2f){2.0f * (P.x - P.y), 0.0f});
            }
        }
    };

    // Software pipeline over the workgroup's tiles.  The samples of tile t + 1 are staged into LDS and the loads of tile t + 2
    // issued BEFORE the back half (and the output stores) of tile t: a sample load never queues behind this CU's own stores in
    // the in-order vector-memory path, has a whole tile of time to arrive, and the wait in front of the staging writes only
    // meets stores that were issued most of a tile earlier.
    unsigned cur = wid, nxt = wid + slots;
    if (cur < hi) {
        load_tile(cur);
        stage_tile();
    }
    if (nxt < hi) load_tile(nxt);
    __syncthreads();  // xs and tables visible
    if (cur < hi) front(cur);
    while (cur < hi) {
        const unsigned nn = nxt + slots;
        if (nxt < hi) {
            stage_tile();
            if (nn < hi) load_tile(nn);
        }
        back(cur);
        if (nxt >= hi) break;
        __syncthreads();  // xs visible
        front(nxt);
        cur = nxt;
        nxt = nn;
    }
}

template <int MODE, int AMP>
hipError_t launch_q(const StftArgs &a, hipStream_t s) {
    const unsigned total = a.tiles * a.batch;
    const unsigned per_xcd = (total + 7) / 8;
    static const unsigned max_slots = [] {
        const char *v = getenv("SGX_Q_SLOTS");  // experiment: workgroups per XCD (64 = two per CU)
        return v ? (unsigned)atoi(v) : 64u;
    }();
    const unsigned slots = per_xcd < max_slots ? per_xcd : max_slots;  // two workgroups per CU
    hipError_t e = set_max_dynamic_lds((const void *)k_q16x32<MODE, AMP>, kQLds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_q16x32<MODE, AMP>), dim3(slots * 8), dim3(512), kQLds, s, a, per_xcd, total, slots);
    return hipGetLastError();
}

}  // namespace

// f32, n_fft = 1024, linear-frequency outputs (power / magnitude / dB / complex), 16-byte aligned rows, hop % 4 == 0 and a
// tile of at most 3 x 512 chunks (hop <= 341)
bool q16x32_takes(const StftArgs &a) {
    if (a.n_fft != 1024 || a.out_mode == OUT_MEL) return false;
    const bool aligned16 = (reinterpret_cast<uintptr_t>(a.x) % 16 == 0) && (a.sample_stride % 4 == 0) && (a.hop % 4 == 0);
    const unsigned chunks = (15u * a.hop + 1024u) >> 2;
    return aligned16 && chunks <= (unsigned)kQRounds * 512u;
}

hipError_t launch_q16x32_f32(const StftArgs &a, hipStream_t s) {
    if (a.out_mode == OUT_COMPLEX) return launch_q<OUT_COMPLEX, AMP_POWER>(a, s);
    if (a.amp == AMP_MAGNITUDE) return launch_q<OUT_LINEAR, AMP_MAGNITUDE>(a, s);
    if (a.amp == AMP_DB) return launch_q<OUT_LINEAR, AMP_DB>(a, s);
    return launch_q<OUT_LINEAR, AMP_POWER>(a, s);
}

}  // namespace sgx
