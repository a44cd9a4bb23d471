// vmem_issue.hip — cost per vector-memory wave-instruction on MI355X for the access shapes of the STFT kernel.
// Everything is L2/MALL resident (small footprint) so this isolates the CU-side (TA/TCP) cost per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// KIND 0: per-lane float2, lane (f=lane/16, n2=lane%16): 4 segments of 128 B, 1 KiB apart (the direct-load shape)
// KIND 1: coalesced float4: 64 lanes x 16 B = 1 KiB contiguous
// KIND 2: coalesced float (dword): 256 B contiguous
// KIND 3: per-lane float, 4 segments of 64 B in 4 rows 2504 B apart (the store shape) — as loads
template <int KIND>
__global__ void kload(const float *x, float *out, int iters) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const float *base = x + (size_t)(wave & 255) * 8192;
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int o = ((i * 16 + u) & 31);
            if (KIND == 0) { v2f v = *(const v2f *)(base + (lane >> 4) * 256 + (lane & 15) * 2 + o * 32); acc += v.x + v.y; }
            if (KIND == 1) { v4f v = *(const v4f *)(base + lane * 4 + (o & 7) * 256); acc += v.x + v.w; }
            if (KIND == 2) { acc += base[lane + o * 64]; }
            if (KIND == 3) { acc += base[(lane >> 4) * 626 + (lane & 15) + o * 16]; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int KIND>
__global__ void kstore(float *x, int iters) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float *base = x + (size_t)wave * 16384;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int o = ((i * 16 + u) & 31);
            if (KIND == 2) base[lane + o * 64] = (float)i;
            if (KIND == 3) base[(lane >> 4) * 626 + (lane & 15) + o * 16 + (o >> 3) * 2504] = (float)i;
            if (KIND == 1) *(v4f *)(base + lane * 4 + (o & 7) * 256) = (v4f){(float)i, 0, 0, 0};
        }
    }
}
template <typename F>
void timeit(const char *name, F launch, int waves_per_cu, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(10);
    (void)hipEventRecord(e0); launch(iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_cu = (double)iters * 16 * waves_per_cu;
    printf("%-44s waves/CU=%2d  %.3f ms  %.1f ns per wave-instr per CU (%.0f cyc @2.3GHz)\n", name, waves_per_cu, ms,
           ms * 1e6 / insts_per_cu, ms * 1e6 / insts_per_cu * 2.3);
}
int main() {
    float *x, *out; (void)hipMalloc(&x, (size_t)2048 * 16 * 16384 * 4 / 8); (void)hipMalloc(&out, 1 << 24);
    (void)hipMemset(x, 0, (size_t)2048 * 16 * 16384 * 4 / 8);
    const int iters = 2000;
    for (int wpc : {4, 8, 16}) {
        const int blocks = 256 * wpc / 4;
        timeit("load  per-lane float2, 4 x 128 B segments", [&](int it) { hipLaunchKernelGGL(kload<0>, dim3(blocks), dim3(256), 0, 0, x, out, it); }, wpc, iters);
        timeit("load  coalesced float4 (1 KiB)", [&](int it) { hipLaunchKernelGGL(kload<1>, dim3(blocks), dim3(256), 0, 0, x, out, it); }, wpc, iters);
        timeit("load  coalesced dword (256 B)", [&](int it) { hipLaunchKernelGGL(kload<2>, dim3(blocks), dim3(256), 0, 0, x, out, it); }, wpc, iters);
        timeit("load  dword, 4 x 64 B segments (rows)", [&](int it) { hipLaunchKernelGGL(kload<3>, dim3(blocks), dim3(256), 0, 0, x, out, it); }, wpc, iters);
        timeit("store coalesced dword (256 B)", [&](int it) { hipLaunchKernelGGL(kstore<2>, dim3(blocks), dim3(256), 0, 0, x, it); }, wpc, iters);
        timeit("store dword, 4 x 64 B segments (rows)", [&](int it) { hipLaunchKernelGGL(kstore<3>, dim3(blocks), dim3(256), 0, 0, x, it); }, wpc, iters);
        timeit("store coalesced float4 (1 KiB)", [&](int it) { hipLaunchKernelGGL(kstore<1>, dim3(blocks), dim3(256), 0, 0, x, it); }, wpc, iters);
    }
    return 0;
}
