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

// k_d512 (kernels_d32x16.hip): f64 n_fft = 512, two frames per 512-point complex transform, 32-frame tiles (16 slots of a frame pair)
namespace d512 {
constexpr int kEx = 16 * (8192 + 16);          // 131328: exchange buffer ex[slot][16][32]
constexpr int kPwBytes = 268 * 256;            // |X|^2 tile: 268 bins x 32 frames of f64 (bin k, frame f at k * 32 + f)
constexpr int kPwOff5 = kEx - kPwBytes;        // 62720: staged samples of 5 rounds (40 960 B, hop <= 148) below it
constexpr int kPwOff9 = 9 * 8192;              // 73728: 9 rounds (hop <= 260) — the tile then ends at 142 336, behind the exchange buffer
constexpr int kBuf9 = kPwOff9 + kPwBytes;      // 142336
constexpr int kWinBytes = 4096;                // tables behind the buffer: 512 doubles w[n] / 2, then the band schedule
constexpr int kSchMaxWords = (163840 - kBuf9 - kWinBytes - 64) / 4;  // 4336
constexpr int kSegs = 3;
}  // namespace d512
}  // namespace sgx
