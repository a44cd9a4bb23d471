// istft_read_pattern.hip — how fast can the inverse STFT's input be read in its own access pattern, and does the load width matter?
// spec[b][513][n_frames] complex f32 (frame contiguous, the reference's layout S9); a tile = 16 frames (13 new + 3 halo) x 513 bins.
//   MODE 0: 8 bytes per lane, lane = (job = wave + 4 (lane >> 4), frame = lane & 15): 33 loads per lane, each wave-instruction
//           touches 4 rows x 128 B — k_istft1024b today
//   MODE 1: 16 bytes per lane, lane = (job = wave + 4 (lane >> 3) [8 jobs per wave], frame pair = lane & 7), 128 threads per tile:
//           the same bytes with half the wave-instructions, each touching 8 rows x 128 B
//   MODE 2: 16 bytes per lane, 32-frame tiles (26 new + 6 halo?) is NOT the kernel's tiling: instead two adjacent tiles per workgroup of
//           256 threads, lane = (job, frame pair) as MODE 1, second half of the workgroup takes the next tile
// No arithmetic: every lane sums what it loaded and stores one float.  One workgroup per tile (MODE 2: per two tiles), XCD-contiguous order.
// build: hipcc -O3 --offload-arch=gfx950 -o istft_read_pattern istft_read_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const float *spec, float *sink, unsigned n_frames, unsigned tiles, unsigned total) {
    const unsigned per = (total + 7u) >> 3;
    unsigned lb = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (lb >= total) return;
    const unsigned tid = threadIdx.x;
    float acc = 0.f;
    if (MODE == 0) {
        const unsigned t = lb % tiles, b = lb / tiles;
        const unsigned lane = tid & 63u, jq = lane >> 4, fl = lane & 15u, j = (tid >> 6) + 4u * jq;
        long long f = (long long)t * 13 - 3 + fl;
        if (f < 0 || f >= (long long)n_frames) f = 0;
        const v2f *in = (const v2f *)spec + (size_t)b * 513u * n_frames + (size_t)f;
        const unsigned ka = j == 0 ? 16u : j, kb = j == 0 ? 0u : j + 256u;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const unsigned k0 = (p < 8 ? ka : kb) + 32u * (p & 7);
            const v2f P = in[(size_t)k0 * n_frames], Q = in[(size_t)(512u - k0) * n_frames];
            acc += P.x + P.y + Q.x + Q.y;
        }
        acc += in[(size_t)256u * n_frames].x;
    } else {
        // a tile per 128 threads: 2 waves x 8 jobs, frame pair per lane
        const unsigned half = tid >> 7, t2 = tid & 127u;
        const unsigned w = MODE == 2 ? 2u * lb + half : lb;
        if (MODE == 1 && half) return;
        if (w >= total * (MODE == 2 ? 2u : 1u)) return;
        const unsigned t = w % tiles, b = w / tiles;
        const unsigned lane = t2 & 63u, jq = lane >> 3, fp = lane & 7u, j = (t2 >> 6) + 2u * jq;
        long long f = (long long)t * 13 - 3 + 2 * fp;
        if (f < 0 || f + 1 >= (long long)n_frames) f = 0;
        const float *in = spec + ((size_t)b * 513u * n_frames + (size_t)f) * 2u;
        const unsigned ka = j == 0 ? 16u : j, kb = j == 0 ? 0u : j + 256u;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const unsigned k0 = (p < 8 ? ka : kb) + 32u * (p & 7);
            v4f P, Q;
            __builtin_memcpy(&P, in + (size_t)k0 * n_frames * 2u, 16);
            __builtin_memcpy(&Q, in + (size_t)(512u - k0) * n_frames * 2u, 16);
            acc += P.x + P.y + P.z + P.w + Q.x + Q.y + Q.z + Q.w;
        }
    }
    if (acc == 123456.789f) sink[blockIdx.x] = acc;
}

int main() {
    const unsigned B = 256, nf = 626, tiles = (nf + 3 + 12) / 13;  // 49 tiles of 13 new frames
    const size_t elems = (size_t)B * 513 * nf * 2;
    float *spec, *sink;
    hipMalloc(&spec, elems * 4);
    hipMalloc(&sink, 1 << 20);
    hipMemset(spec, 0, elems * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const unsigned total = tiles * B;
    for (int mode = 0; mode < 3; ++mode) {
        const unsigned wgs = mode == 2 ? (total + 1) / 2 : total, grid = ((wgs + 7) / 8) * 8;
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < 5; ++i) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, spec, sink, nf, tiles, total);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, spec, sink, nf, tiles, total);
                else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, spec, sink, nf, tiles, wgs);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms / 5 < best) best = ms / 5;
        }
        const double bytes = (double)total * 513 * 16 * 8;  // what the tiles read (halo included)
        printf("mode %d: %.1f us per pass, %.2f TB/s of tile bytes (%.2f TB/s of the spectrum)\n", mode, best * 1e3, bytes / best / 1e9, elems * 4.0 / best / 1e9);
    }
    return 0;
}
