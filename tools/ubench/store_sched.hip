// store_sched.hip — which store shape / issue pattern overlaps best with arithmetic?  Models the tuned STFT kernel's output
// stream: persistent 512-thread workgroups (one per CU), each round a pair of 16-frame tiles = 32 consecutive frames x 513
// rows of out[b][513][n_frames] (frame contiguous), next to a pure-VALU phase of adjustable length.
//   MODE 0: dword stores, 33 per lane, in one burst after the arithmetic (2 rows x 128 B per wave-instruction) — the kernel today
//   MODE 1: dwordx4 stores, ~8 per lane, burst (lane = 4 consecutive frames of one row: 8 rows x 128 B per wave-instruction)
//   MODE 2: dword stores spread evenly through the arithmetic
//   MODE 3: dwordx4 stores spread evenly through the arithmetic
//   MODE 4: dwordx2 stores (2 frames per lane), burst
//   MODE 5: dwordx4 stores spread over the first 40 % of the arithmetic (the window between two uses of the exchange buffer)
//   MODE 6: dwordx2 stores spread evenly
// build: hipcc -O3 --offload-arch=gfx950 -o store_sched store_sched.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void work_chunk(v2f (&acc)[8], v2f c, int iters) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = __builtin_elementwise_fma(acc[u], c, (v2f){1.0f, 0.5f});
    }
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float *out, int n_frames, int pairs_per_sig, int total, int per_xcd, int slots, int work) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int lo = xcd * per_xcd, hi = min(lo + per_xcd, total);
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    v2f acc[8];
    for (int u = 0; u < 8; ++u) acc[u] = (v2f){(float)threadIdx.x, (float)u};
    const v2f c = {0.999f, 1.001f};
    for (int p = lo + slot; p < hi; p += slots) {
        const int b = p / pairs_per_sig, f0 = (p - b * pairs_per_sig) * 32;
        float *ob = out + (size_t)b * 513 * n_frames + f0;
        if constexpr (MODE == 0 || MODE == 2) {
            const int jq = lane >> 5, f = lane & 31, j = wv + 8 * jq;
            const bool ok = f0 + f < n_frames;
            for (int i = 0; i < 33; ++i) {
                if (MODE == 2) work_chunk(acc, c, work / 33);
                else if (i == 0) work_chunk(acc, c, work);
                const int row = i < 32 ? j + 16 * i : 512;
                if (ok && (i < 32 || j == 0)) ob[(size_t)row * n_frames + f] = acc[i & 7].x;
            }
        } else if constexpr (MODE == 1 || MODE == 3 || MODE == 5) {
            const int jq = lane >> 5, q = (lane & 31) >> 2, i4 = lane & 3, j = wv + 8 * jq;
            const bool ok = f0 + 4 * q + 3 < n_frames;
            for (int g = 0; g < 8; ++g) {
                if (MODE == 3) work_chunk(acc, c, work / 8);
                else if (MODE == 5) work_chunk(acc, c, work / 20);
                else if (g == 0) work_chunk(acc, c, work);
                const int row = j + 16 * (4 * g + i4);
                float *pp = ob + (size_t)row * n_frames + 4 * q;
                const v4f v = {acc[g].x, acc[g].y, acc[(g + 1) & 7].x, acc[(g + 1) & 7].y};
                if (ok) *(v4f *)pp = v;
            }
            if (j == 0 && f0 + (lane & 31) < n_frames) ob[(size_t)512 * n_frames + (lane & 31)] = acc[0].x;
            if (MODE == 5) work_chunk(acc, c, work - 8 * (work / 20));
        } else {
            const int jq = lane >> 5, q = (lane & 31) >> 1, i2 = lane & 1, j = wv + 8 * jq;
            const bool ok = f0 + 2 * q + 1 < n_frames;
            for (int g = 0; g < 16; ++g) {
                if (MODE == 6) work_chunk(acc, c, work / 16);
                else if (g == 0) work_chunk(acc, c, work);
                const int row = j + 16 * (2 * g + i2);
                float *pp = ob + (size_t)row * n_frames + 2 * q;
                if (ok) *(v2f *)pp = acc[g & 7];
            }
            if (j == 0 && f0 + (lane & 31) < n_frames) ob[(size_t)512 * n_frames + (lane & 31)] = acc[0].x;
        }
    }
    if (acc[0].x == 123.456f) out[0] = acc[1].y + acc[2].x + acc[3].x + acc[4].x + acc[5].x + acc[6].x + acc[7].x;
}

template <int MODE>
float run(float *d, int n_frames, int batch, int work) {
    const int pps = (n_frames + 31) / 32, total = pps * batch, per_xcd = (total + 7) / 8, slots = 32;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, d, n_frames, pps, total, per_xcd, slots, work);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, d, n_frames, pps, total, per_xcd, slots, work);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 10 * 1e3f;
}

int main() {
    const int batch = 256;
    for (int n_frames : {626, 640}) {
        float *d;
        (void)hipMalloc(&d, (size_t)batch * 513 * n_frames * 4 + 4096);
        printf("n_frames = %d (%.0f MB)\n", n_frames, batch * 513.0 * n_frames * 4 / 1e6);
        for (int work : {0, 100, 140}) {
            printf("  work=%3d: dword burst %.1f us | x4 burst %.1f | dword spread %.1f | x4 spread %.1f | x2 burst %.1f | x4 first-40%% %.1f | x2 spread %.1f\n", work,
                   run<0>(d, n_frames, batch, work), run<1>(d, n_frames, batch, work), run<2>(d, n_frames, batch, work),
                   run<3>(d, n_frames, batch, work), run<4>(d, n_frames, batch, work), run<5>(d, n_frames, batch, work), run<6>(d, n_frames, batch, work));
        }
        (void)hipFree(d);
    }
    return 0;
}
