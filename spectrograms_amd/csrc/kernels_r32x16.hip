// kernels_r32x16.hip — tuned f32 n_fft = 1024 kernel (placeholder until the tuned kernel lands).
#include "sgx_internal.h"
namespace sgx {
bool plan_geometry_r32x16_f32(StftArgs &) { return false; }
hipError_t launch_r32x16_f32(const StftArgs &, hipStream_t) { return hipErrorNotSupported; }
}  // namespace sgx
