// membench.hip — sgx_membench: the HBM rate this device really delivers to plain streaming kernels, measured in-process so that
// bench.py can quote it next to the nominal 8 TB/s ("verify on the box with a device memcpy / triad and quote
// the measured peak next to the nominal").  Measurement utility of the C ABI: it allocates its own buffers, runs, frees them; no
// plan is involved and nothing on the transform path calls it.
//
// Four kernels, all 16 bytes per lane, grid-stride, 4 / 8 / 16 workgroups of 256 threads per CU or one pass per workgroup (the best
// of the four is reported): copy (read + write), read (sum kept
// alive, nothing written), write (fill), and the linear-power STFT's own mix (one buffer read, two written).  Buffers default to 1 GiB each — four times the 256 MiB Infinity Cache — so the rate is
// the memory's, not the cache's (MI355X_MICROARCH.md §Infinity Cache).
#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "sgx_internal.h"

namespace sgx {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_mb_copy(const v4f *__restrict__ src, v4f *__restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256u * 4u;
    for (size_t i = (size_t)blockIdx.x * 256u * 4u + threadIdx.x; i < n; i += stride) {
        v4f r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = i + u * 256u < n ? src[i + u * 256u] : (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * 256u < n) dst[i + u * 256u] = r[u];
    }
}

__global__ __launch_bounds__(256) void k_mb_read(const v4f *__restrict__ src, float *__restrict__ sink, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256u * 4u;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256u * 4u + threadIdx.x; i < n; i += stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * 256u < n) acc += src[i + u * 256u];
    }
    const float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123456.789f) sink[0] = s;  // never true for the zero-filled source: keeps the loads alive without a write stream
}

__global__ __launch_bounds__(256) void k_mb_write(v4f *__restrict__ dst, size_t n, float v) {
    const size_t stride = (size_t)gridDim.x * 256u * 4u;
    const v4f val = {v, v + 1.f, v + 2.f, v + 3.f};
    for (size_t i = (size_t)blockIdx.x * 256u * 4u + threadIdx.x; i < n; i += stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * 256u < n) dst[i + u * 256u] = val;
    }
}

// the linear-power STFT's traffic mix: every 16 bytes read go with 32 bytes written (1022 B read, 2052 B written per frame)
__global__ __launch_bounds__(256) void k_mb_mix(const v4f *__restrict__ src, v4f *__restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256u * 4u;
    for (size_t i = (size_t)blockIdx.x * 256u * 4u + threadIdx.x; i < n; i += stride) {
        v4f r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = i + u * 256u < n ? src[i + u * 256u] : (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * 256u < n) {
                dst[i + u * 256u] = r[u];
                dst[n + i + u * 256u] = r[u] + r[u];
            }
    }
}

// sgx_clock_probe: one wave per CU stamps the shader clock counter (s_memtime) against the constant 100 MHz counter
// (s_memrealtime) over ~20 us (MI355X_MICROARCH.md, DVFS give-back item 6).  Every wave leaves after a bounded number of polls.
__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long *out, unsigned ticks) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    for (unsigned spin = 0; spin < (1u << 16) && r1 - r0 < ticks; ++spin) {
        __builtin_amdgcn_s_sleep(8);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t1 - t0;
        out[2 * blockIdx.x + 1] = r1 - r0;
    }
}

}  // namespace
}  // namespace sgx

using namespace sgx;

extern "C" sgx_status sgx_clock_probe(int32_t device, void *hip_stream, double *mhz) {
    if (!mhz) return SGX_INVALID_INPUT;
    *mhz = 0.0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SGX_BACKEND;
    int dev = device;
    if (dev == -1 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev < 0 || dev >= ndev) return SGX_INVALID_INPUT;
    DeviceGuard dg;
    if (dg.enter(dev) != hipSuccess) return SGX_BACKEND;
    const unsigned nwg = std::max(1u, device_cu_count());
    unsigned long long *h = nullptr;
    if (hipHostMalloc((void **)&h, 2 * sizeof(unsigned long long) * nwg, hipHostMallocDefault) != hipSuccess) return SGX_BACKEND;
    for (unsigned i = 0; i < 2 * nwg; ++i) h[i] = 0;
    hipStream_t s = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(k_clock_probe, dim3(nwg), dim3(64), 0, s, h, 2000u);
    sgx_status st = SGX_BACKEND;
    if (hipGetLastError() == hipSuccess && hipStreamSynchronize(s) == hipSuccess) {
        std::vector<double> r;
        for (unsigned i = 0; i < nwg; ++i)
            if (h[2 * i + 1] > 0) r.push_back(double(h[2 * i]) / double(h[2 * i + 1]) * 100.0);
        if (!r.empty()) {
            std::nth_element(r.begin(), r.begin() + r.size() / 2, r.end());
            *mhz = r[r.size() / 2];
            st = SGX_OK;
        }
    }
    (void)hipHostFree(h);
    return st;
}

extern "C" sgx_status sgx_membench(int32_t device, size_t bytes, int32_t mode, int32_t iters, double *gb_per_s) {
    if (!gb_per_s || mode < 0 || mode > 3 || iters <= 0) return SGX_INVALID_INPUT;
    *gb_per_s = 0.0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SGX_BACKEND;
    int dev = device;
    if (dev == -1 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev < 0 || dev >= ndev) return SGX_INVALID_INPUT;
    DeviceGuard dg;
    if (dg.enter(dev) != hipSuccess) return SGX_BACKEND;
    if (bytes == 0) bytes = size_t(1) << 30;
    bytes &= ~size_t(4095);
    if (bytes < 4096) return SGX_INVALID_INPUT;
    void *src = nullptr, *dst = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    sgx_status st = SGX_BACKEND;
    float ms = 0.f;
    const size_t n = bytes / 16;
    // the rate depends on how many workgroups stream at once: try a few grids and report the best (a "peak" is the most the
    // device delivers to a plain kernel, not what one launch geometry happens to get)
    const size_t full = (n + 1023) / 1024;
    const size_t cus = device_cu_count();
    const unsigned grids[4] = {(unsigned)std::min(full, cus * 4u), (unsigned)std::min(full, cus * 8u), (unsigned)std::min(full, cus * 16u),
                               (unsigned)std::min<size_t>(full, 0x7fffffffu)};
    auto run = [&](unsigned grid) {
        if (mode == 0) hipLaunchKernelGGL(k_mb_copy, dim3(grid), dim3(256), 0, nullptr, (const v4f *)src, (v4f *)dst, n);
        else if (mode == 1) hipLaunchKernelGGL(k_mb_read, dim3(grid), dim3(256), 0, nullptr, (const v4f *)src, (float *)dst, n);
        else if (mode == 3) hipLaunchKernelGGL(k_mb_mix, dim3(grid), dim3(256), 0, nullptr, (const v4f *)src, (v4f *)dst, n);
        else hipLaunchKernelGGL(k_mb_write, dim3(grid), dim3(256), 0, nullptr, (v4f *)dst, n, 1.0f);
    };
    do {
        if (mode != 2 && hipMalloc(&src, bytes) != hipSuccess) break;
        if (hipMalloc(&dst, mode == 1 ? 4096 : mode == 3 ? 2 * bytes : bytes) != hipSuccess) break;
        if (src && hipMemsetAsync(src, 0, bytes, nullptr) != hipSuccess) break;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) break;
        bool ok = true;
        for (unsigned grid : grids) {
            for (int w = 0; w < 2; ++w) run(grid);
            if (hipEventRecord(e0, nullptr) != hipSuccess) { ok = false; break; }
            for (int i = 0; i < iters; ++i) run(grid);
            if (hipEventRecord(e1, nullptr) != hipSuccess) { ok = false; break; }
            if (hipEventSynchronize(e1) != hipSuccess || hipGetLastError() != hipSuccess) { ok = false; break; }
            if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f) { ok = false; break; }
            const double moved = double(bytes) * (mode == 0 ? 2.0 : mode == 3 ? 3.0 : 1.0) * double(iters);
            *gb_per_s = std::max(*gb_per_s, moved / (double(ms) * 1e-3) / 1e9);
        }
        if (ok) st = SGX_OK;
    } while (false);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (src) (void)hipFree(src);
    if (dst) (void)hipFree(dst);
    return st;
}
