// Buffer-descriptor addressing (gfx950): a 128-bit descriptor (base, byte count) in SGPRs plus a 32-bit byte offset per lane.
// The hardware checks every access against the byte count: a load outside returns 0, a store outside is dropped — which is
// exactly the zero padding of the framing (S1) and needs no branches, no 64-bit lane addresses and no exec masking.
#pragma once
#include <hip/hip_runtime.h>

namespace sgx {

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

// One sample / one pair of consecutive samples at byte offset `vo` (+ a wave-uniform `so`, which is NOT range-checked: pass it
// only where the whole access is known to lie inside the row).  A pair is one access: use it only where both samples are
// inside, or both outside, the row.  one<true> keeps the complete offset in the lane register: a negative offset (left
// padding) is out of range as an unsigned number, but if the compiler moves a constant part of the sum into the instruction's
// immediate field the hardware adds that without wrapping, and samples just right of the row start come back 0 (seen in the
// parity tests).  Where the offset cannot be negative, one<false> lets the constants fold (and neighbouring loads merge).
template <typename T>
struct BufLd;
template <>
struct BufLd<float> {
    typedef float V2 __attribute__((ext_vector_type(2)));
    template <bool MAY_BE_NEGATIVE>
    static __device__ __forceinline__ float one(__amdgpu_buffer_rsrc_t r, int vo) {
        if constexpr (MAY_BE_NEGATIVE) asm("" : "+v"(vo));
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, 0, 0));
    }
    static __device__ __forceinline__ V2 pair(__amdgpu_buffer_rsrc_t r, int vo, int so) {
        return __builtin_bit_cast(V2, __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0));
    }
};
template <>
struct BufLd<double> {
    typedef double V2 __attribute__((ext_vector_type(2)));
    template <bool MAY_BE_NEGATIVE>
    static __device__ __forceinline__ double one(__amdgpu_buffer_rsrc_t r, int vo) {
        if constexpr (MAY_BE_NEGATIVE) asm("" : "+v"(vo));
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, vo, 0, 0));
    }
    static __device__ __forceinline__ V2 pair(__amdgpu_buffer_rsrc_t r, int vo, int so) {
        return __builtin_bit_cast(V2, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0));
    }
};

}  // namespace sgx
