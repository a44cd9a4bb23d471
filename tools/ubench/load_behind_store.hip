// load_behind_store.hip — do vector loads of one wave queue behind the (HBM-bound) stores of OTHER waves of the CU?
// 8 waves per CU: waves 0-3 ("storers") write row-shaped 64-byte segments continuously; waves 4-7 ("loaders")
// alternate 5 coalesced 16-byte loads with ~3k cycles of FMA work.  Reports the loaders' cycles spent issuing the loads
// and waiting for their data, with the storers active vs idle, for several load flavours.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int FLAVOUR>  // 0 plain global_load_dwordx4, 1 nontemporal, 2 scalar (s_load) broadcast, 3 LDS-DMA
__global__ __launch_bounds__(512) void k(float *out, const float *in, unsigned long long *stamps, int ticks, int store_on) {
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool storer = __builtin_amdgcn_readfirstlane(wave) < 4;
    if (storer) {
        if (!store_on) return;
        float *ob = out + ((size_t)blockIdx.x * 4 + wave) * (size_t)ticks * 33 * 16 * 626 / 64;  // private region
        for (int t = 0; t < ticks; ++t)
            for (int i = 0; i < 33; ++i)
                ob[((size_t)t * 33 + i) * 2504 + (lane >> 4) * 626 + (lane & 15)] = (float)t;
    } else {
        const float *ib = in + ((size_t)blockIdx.x * 4 + (wave - 4)) * (size_t)ticks * 5 * 256;
        unsigned long long issue = 0, wait = 0;
        float acc = 0.f;
        for (int t = 0; t < ticks; ++t) {
            unsigned long long t0 = __builtin_amdgcn_s_memtime();
            v4f c[5];
            if (FLAVOUR == 0) {
#pragma unroll
                for (int r = 0; r < 5; ++r) c[r] = *(const v4f *)(ib + ((size_t)t * 5 + r) * 256 + lane * 4);
            } else if (FLAVOUR == 1) {
#pragma unroll
                for (int r = 0; r < 5; ++r) c[r] = __builtin_nontemporal_load((const v4f *)(ib + ((size_t)t * 5 + r) * 256 + lane * 4));
            } else if (FLAVOUR == 2) {
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    const float *p = ib + ((size_t)t * 5 + r) * 256;  // wave-uniform address -> s_load
                    c[r] = (v4f){p[0], p[1], p[2], p[3]};
                }
            } else {
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    __builtin_amdgcn_global_load_lds(ib + ((size_t)t * 5 + r) * 256 + lane * 4, lds + (wave - 4) * 1024 + r * 64 * 4 /*per-wave region, dwordx4: 1 KiB*/ , 16, 0, 0);
                    c[r] = (v4f){0, 0, 0, 0};
                }
            }
            asm volatile("" ::: "memory");
            unsigned long long t1 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (FLAVOUR == 3) { c[0].x = lds[(wave - 4) * 1024 + lane]; }
            unsigned long long t2 = __builtin_amdgcn_s_memtime();
            issue += t1 - t0;
            wait += t2 - t1;
            for (int r = 0; r < 5; ++r) acc += c[r].x + c[r].y + c[r].z + c[r].w;
#pragma unroll 1
            for (int q = 0; q < 600; ++q) acc = acc * 1.0001f + 0.5f;  // ~3k cycles of dependent VALU
        }
        if (lane == 0) {
            atomicAdd(&stamps[0], issue);
            atomicAdd(&stamps[1], wait);
            atomicAdd(&stamps[2], 1ull);
        }
        out[(size_t)1 << 28 | (blockIdx.x * 512 + threadIdx.x)] = acc;
    }
}

template <int FLAVOUR>
void run(const char *name, float *out, const float *in, unsigned long long *st, int store_on) {
    const int ticks = 40;
    unsigned long long h[3] = {0, 0, 0};
    (void)hipMemcpy(st, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<FLAVOUR>, dim3(256), dim3(512), 0, 0, out, in, st, ticks, store_on);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-28s stores %-3s  kernel %.3f ms  loader: issue %6.0f cyc/tick, data wait %6.0f cyc/tick\n", name, store_on ? "ON" : "off", ms,
           (double)h[0] / h[2] / ticks, (double)h[1] / h[2] / ticks);
}

int main() {
    float *out, *in; unsigned long long *st;
    (void)hipMalloc(&out, ((size_t)1 << 30) + (1 << 22) * 4);  // 1 GiB+: stores stream to HBM
    (void)hipMalloc(&in, (size_t)256 * 4 * 40 * 5 * 256 * 4 + 4096);
    (void)hipMalloc(&st, 64);
    (void)hipMemset(in, 0, (size_t)256 * 4 * 40 * 5 * 256 * 4);
    for (int on : {0, 1}) {
        run<0>("global_load_dwordx4", out, in, st, on);
        run<1>("nontemporal load", out, in, st, on);
        run<2>("scalar load (uniform)", out, in, st, on);
        run<3>("global_load_lds (LDS-DMA)", out, in, st, on);
    }
    return 0;
}
