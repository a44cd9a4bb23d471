// store_pattern.hip — how fast can MI355X write the reference's [B][513][n_frames] frame-contiguous layout when a
// workgroup owns a tile of F consecutive frames (segments of F*4 bytes per bin row)?  Decides the tile width.
#include <hip/hip_runtime.h>
#include <cstdio>

// Each 256-thread workgroup owns tile (b, f0..f0+F) and writes all 513 rows; a wave-instruction covers 64/F rows.
template <int F, int VEC>
__global__ void k(float *out, int n_frames, int tiles, int total) {
    const int per_xcd = (total + 7) / 8;
    for (int wid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3); wid < min(((int)(blockIdx.x & 7) + 1) * per_xcd, total); wid += gridDim.x >> 3) {
        const int b = wid / tiles, tile = wid - b * tiles, f0 = tile * F;
        float *ob = out + (size_t)b * 513 * n_frames + f0;
        constexpr int LPR = F / VEC;          // lanes per row
        constexpr int RPI = 256 / LPR;        // rows per workgroup-instruction
        const int lf = (threadIdx.x % LPR) * VEC, r0 = threadIdx.x / LPR;
        const float v = (float)wid;
        for (int r = r0; r < 513; r += RPI) {
            float *p = ob + (size_t)r * n_frames + lf;
            if (f0 + lf + VEC <= n_frames) {
                if constexpr (VEC == 1) p[0] = v;
                else if constexpr (VEC == 2) { *(float2*)p = make_float2(v, v); }
                else { p[0] = v; p[1] = v; p[2] = v; p[3] = v; }
            }
        }
    }
}

template <int F, int VEC>
void run(float *d, int n_frames, int batch) {
    const int tiles = (n_frames + F - 1) / F, total = tiles * batch;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int grid : {256, 512, 1024}) {
        hipLaunchKernelGGL((k<F, VEC>), dim3(grid), dim3(256), 0, 0, d, n_frames, tiles, total);
        (void)hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<F, VEC>), dim3(grid), dim3(256), 0, 0, d, n_frames, tiles, total);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        const double bytes = (double)batch * 513 * n_frames * 4;
        printf("F=%2d VEC=%d grid=%4d  %.3f ms  %.2f TB/s\n", F, VEC, grid, ms, bytes / (ms * 1e-3) / 1e12);
    }
}

int main() {
    const int n_frames = 626, batch = 256;
    float *d; (void)hipMalloc(&d, (size_t)batch * 513 * n_frames * 4 + 4096);
    run<16, 1>(d, n_frames, batch);
    run<16, 2>(d, n_frames, batch);
    run<32, 1>(d, n_frames, batch);
    run<32, 2>(d, n_frames, batch);
    run<64, 1>(d, n_frames, batch);
    // aligned variant: 640 frames per row (rows 16-B aligned, tiles never straddle)
    return 0;
    printf("-- n_frames = 640 (aligned rows)\n");
    float *d2; (void)hipMalloc(&d2, (size_t)batch * 513 * 640 * 4 + 4096);
    run<16, 1>(d2, 640, batch);
    run<32, 1>(d2, 640, batch);
    run<64, 1>(d2, 640, batch);
    return 0;
}
