// r32x16_layout.h — LDS layout and table formats of the tuned f32 n_fft = 1024 kernel, shared by the kernel
// (kernels_r32x16.hip) and the host code that builds its tables (plan.hip).
#pragma once

namespace sgx {
namespace r32x16 {

constexpr int kFS = 4096 + 16;      // LDS bytes per frame of ex (odd multiple of 16 -> conflict-free b128 row reads)
constexpr int kPS = 516;            // floats per frame of pw[f][k] (matrix-core / CSR filterbank stage); multiple of 4
constexpr int kExBytes = 16 * kFS;  // 65792: one tile's exchange buffer; also holds xs (staged samples) and the |X|^2 tile
// filterbank outputs: the |X|^2 tile sits in the upper part of the exchange buffer, above the staged samples (<= 22896 B), so
// the next tile's staging does not have to wait for the band reduction
constexpr int kOutOff = kExBytes - 16 * 17 * 128;  // 30976
// tables behind the two exchange buffers
constexpr int kWinOff = 0;                      // float2 win[512]: (w[2n], w[2n+1]) / 2
constexpr int kTw2Off = 4096;                   // float4 tw2[16 jobs][17]: entry i of job j = (W', W'^perp) of the i-th pair it splits
constexpr int kTw2Bytes = 16 * 17 * 16;         // row stride 272 B: jobs j and j + 4 (one read group) sit on different banks
constexpr int kMelOff = kTw2Off + kTw2Bytes;    // filterbank schedule (below)
constexpr int kMelMaxWords = 4096;
constexpr int kLdsBytes = 2 * kExBytes + kMelOff + kMelMaxWords * 4;  // 156416 of the CU's 163840

// Filterbank schedule (32-bit words; floats where noted), for the 4 waves x 8 slots of a half:
//   [0] nseg  [1] total words  [2..3] 0
//   rec [seg][wave][slot], seg <= kSchedSegs (segments past nseg are empty; the kernel fetches one record ahead)
//       = {L (steps of the wave in this segment, multiple of 4), word offset of the slot's weight row, kstart, band (0xffffffff: none)}
//   weight rows (floats): lpad floats per slot, lpad / 4 odd, the 8 slots of a (seg, wave) consecutive; step t weighs bin kstart + t
constexpr int kSchedHdr = 4;
constexpr int kSchedSegs = 4;  // the kernel always runs this many segments (empty ones have L = 0, band = none): n_mels <= 128

}  // namespace r32x16
}  // namespace sgx
