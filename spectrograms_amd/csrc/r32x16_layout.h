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
// n_fft 512 at hop 256 with filterbank outputs: the tile's staged samples (36 864 B + 64 B of padding per 2 KiB = 38 016 B) and the |X|^2 tile
// (34 816 B) do not fit one 65 792-byte half, so this variant's halves are larger and its schedule shorter (a 257-bin bank needs ~1500 words)
constexpr int kExBytesH256 = 72960;                       // 285 x 256
constexpr int kOutOffH256 = kExBytesH256 - 16 * 17 * 128;  // 38144 >= 38016
constexpr int kMelMaxWordsH256 = (163840 - 2 * kExBytesH256 - kMelOff) / 4 - 16;  // 2352

// Filterbank schedule (32-bit words; floats where noted), for the 4 waves x 8 slots of a half:
//   [0] nseg  [1] total words  [2..3] 0
//   rec [seg][wave][slot], seg <= kSchedSegs (segments past nseg are empty; the kernel fetches one record ahead)
//       = {L (steps of the wave in this segment, multiple of 4), word offset of the slot's weight row, kstart, band (0xffffffff: none)}
//   weight rows (floats): lpad floats per slot, lpad / 4 odd, the 8 slots of a (seg, wave) consecutive; step t weighs bin kstart + t
constexpr int kSchedHdr = 4;
constexpr int kSchedSegs = 4;  // the kernel always runs this many segments (empty ones have L = 0, band = none): n_mels <= 128

}  // namespace r32x16

// ---- k_r32x32: the tuned f32 n_fft = 2048 kernel (kernels_r32x32.hip), one 16-frame tile per 512-thread workgroup ----------
namespace r32x32 {
constexpr int kFS2 = 8192 + 16;        // LDS bytes per frame of ex[f][32][32] (odd multiple of 16: conflict-free b128 row reads)
constexpr int kEx2 = 16 * kFS2;        // 131328: exchange buffer; also holds the staged samples (<= 40960 B) and the |X|^2 tile
constexpr int kPw2Off = kEx2 - 518 * 128;  // 65024: |X|^2 tile, 1036 bins x 16 frames (pwt2_index), above the staged samples
constexpr int kWin2Off = 0;            // tables behind the exchange buffer: float2 win[1024] = (w[2n], w[2n+1]) / 2
constexpr int kTw22Off = 8192;         // float4 tw2[32 jobs][17]: entry i of job J = (W', W'^perp) of the i-th pair it splits
constexpr int kSch2Off = kTw22Off + 32 * 17 * 16;  // 16896: band schedule, 8 waves x 8 slots (format of r32x16's, record stride 64 per segment)
constexpr int kLds2Base = kEx2 + kSch2Off;         // 148224
constexpr int kLds2Max = 163840;
constexpr int kSch2MaxWords = (kLds2Max - kLds2Base - 64) / 4;  // 3888
constexpr int kSegs2 = 3;              // 16 groups of 8 bands dealt to 8 waves by length: up to 3 per wave at n_mels = 128
}  // namespace r32x32
}  // namespace sgx
