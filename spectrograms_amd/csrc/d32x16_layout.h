// d32x16_layout.h — LDS layout of the tuned f64 n_fft = 1024 kernel, shared by the kernel (kernels_d32x16.hip) and the host code that
// builds its band schedule (plan.hip).
#pragma once

namespace sgx {
// k_r32x32's tile at 512 complex f64 points
namespace d32x16 {
constexpr int kDFS = 8192 + 16;       // LDS bytes per frame of ex[f][16][32] (odd multiple of 16: conflict-free b128 row reads over frames)
constexpr int kDEx = 16 * kDFS;       // 131328: exchange buffer; also holds the staged samples (<= 40960 B) and the |X|^2 tile
constexpr int kDPwOff = kDEx - 262 * 256;  // 64256: |X|^2 tile, 524 bins x 16 frames of f64 (pwd_index), above the staged samples
constexpr int kDWinOff = 0;           // tables behind the exchange buffer: v2d win[512] = (w[2n], w[2n+1]) / 2
constexpr int kDTw2Off = 8192;        // v2d tw2[32][8]: entry u of lane kind kb = W' = -i W_1024^(kb + 32 u)
constexpr int kDSchOff = kDTw2Off + 32 * 8 * 16;  // 12288: band schedule, 8 waves x 8 slots (r32x16's format with 8-byte weights)
constexpr int kDLdsBase = kDEx + kDSchOff;        // 143616
constexpr int kDLdsMax = 163840;
constexpr int kDSchMaxWords = (kDLdsMax - kDLdsBase - 64) / 4;  // 5040
constexpr int kDSegs = 3;             // groups of 8 bands dealt to 8 waves by length: up to 3 per wave at 128 bands
}  // namespace d32x16
}  // namespace sgx
