// lane_pair.h — the exchange of the lane-pair kernels (k_d32x16 / k_d512, k_d32x32, k_r64x32, k_istft_d1024 / k_istft_d512): a row of the
// exchange buffer is split between lanes l and l ^ 32 of one wave, which trade half of their transformed values with v_permlane32_swap_b32
// (DESIGN.md §3.5).  One definition for all of them (round 5; VERDICT r4: a fix here is one edit).
#pragma once
#include <hip/hip_runtime.h>

namespace sgx {
namespace lanepair {

typedef float lp_v2f __attribute__((ext_vector_type(2)));
typedef double lp_v2d __attribute__((ext_vector_type(2)));
typedef unsigned lp_v2u __attribute__((ext_vector_type(2)));

// lanes l and l ^ 32 trade a complex value; each receives the other's as (im, re)
// (scalars first: __builtin_bit_cast of a vector-element lvalue other than .x reads element 0 with this clang — hipcc 7.2)
__device__ __forceinline__ void trade32(lp_v2d &v) {
    const double vx = v.x, vy = v.y;
    lp_v2u re = __builtin_bit_cast(lp_v2u, vx), im = __builtin_bit_cast(lp_v2u, vy);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        // vdst's lanes 32..63 <-> src0's lanes 0..31.  First: re = {a.re | a.im}, im = {b.re | b.im}; second (im, re): im = {b.re | a.re}, re = {b.im | a.im}
        const lp_v2u s1 = __builtin_amdgcn_permlane32_swap(re[c], im[c], false, false);
        const lp_v2u s2 = __builtin_amdgcn_permlane32_swap(s1.y, s1.x, false, false);
        im[c] = s2.x;
        re[c] = s2.y;
    }
    v.x = __builtin_bit_cast(double, re);
    v.y = __builtin_bit_cast(double, im);
}
__device__ __forceinline__ void trade32(lp_v2f &v) {
    const float vx = v.x, vy = v.y;
    const unsigned re = __builtin_bit_cast(unsigned, vx), im = __builtin_bit_cast(unsigned, vy);
    const lp_v2u s1 = __builtin_amdgcn_permlane32_swap(re, im, false, false);     // re = {a.re | a.im}, im = {b.re | b.im}
    const lp_v2u s2 = __builtin_amdgcn_permlane32_swap(s1.y, s1.x, false, false);  // im = {b.re | a.re}, re = {b.im | a.im}
    const unsigned nim = s2.x, nre = s2.y;
    v.x = __builtin_bit_cast(float, nre);
    v.y = __builtin_bit_cast(float, nim);
}

}  // namespace lanepair
}  // namespace sgx
