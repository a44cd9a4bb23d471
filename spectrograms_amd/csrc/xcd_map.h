// xcd_map.h — workgroup -> tile order for one-tile-per-workgroup kernels.
// Workgroups are dispatched round-robin over the 8 XCDs, each with its own L2.  Neighbouring tiles share cache lines (a
// 16-sequence store segment is 128 bytes at a 4104-byte row pitch, an inverse-STFT tile re-reads 3 halo frames), so XCD x
// takes the contiguous range [x * per, (x + 1) * per) of the logical tile space; the grid is rounded up to a multiple of 8
// and a workgroup whose logical index is past the end returns at once.
#pragma once
#include <hip/hip_runtime.h>

namespace sgx {

__device__ __forceinline__ unsigned xcd_logical_block(unsigned nblocks) {
    const unsigned per = (nblocks + 7u) >> 3;
    return (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
}
static inline unsigned xcd_grid(unsigned long long nblocks) { return (unsigned)(((nblocks + 7ull) >> 3) << 3); }

}  // namespace sgx
