// kernels_istft2048.hip — fused inverse STFT for f32, n_fft = 2048 (round 4): k_istft1024c's dataflow at 1024 complex points, one
// persistent 512-thread workgroup per CU on tiles of 16 new frames (src/spectrogram.rs:4860-4946: C2R per frame with 1/n, window, overlap-add
// in ascending frame order, sum of w^2 normalisation where > 1e-10, centre trim).
//
//   A  lane (jq, f), job J = wave + 8 jq: the 16 pairs (X[k], X[1024 - k]) of k = J + 64 p (p < 8) and J + 512 + 64 (p - 8) (job 0: 64 p and
//      32 + 64 (p - 8), + bin 512), folded once: S = P + conj Q, T = conj(W_2048^k)(P - conj Q), v[k] = conj(S + i T), v[1024 - k] = S - i T
//      with v = conj(Z').  These are the EVEN-indexed elements of row J and the ODD-indexed elements of row 32 - J of v[k1 + 32 k2] (job 0: both
//      of row 0) — k_r32x32's half-row jobs backwards: a 16-point transform of each gives E[J][n] and O[32 - J][n], written to
//      ex[f][k1][E | O][16].
//   B  lane (f, n2 = 0..31): column n2 of the 32 rows: u[k1] = E[k1][n2 mod 16] +- W_32^(n2 mod 16) O[k1][n2 mod 16] (the last radix-2 step of
//      the row transform; each E / O value is read by two lanes), twiddle W_1024^(k1 n2), 32-point transform over k1: y[n2 + 32 n1] ->
//      (x[2n], x[2n+1]) = conj(y) / 2048, times the window, real frames fr[f][2048] over the dead ex.
//   C  overlap-add with the carry of k_istft1024c: a tile is 16 NEW frames F .. F + 15 and the 16 hop blocks they start in; the first
//      ov = floor(2047 / hop) blocks start from the partial sums the previous tile left in LDS, the partial sums past the tile are left
//      for the next one.  A workgroup walks runs of consecutive tiles of one signal; a run inside a signal first passes over the tile in
//      front of it without storing.
#include <algorithm>

#include "fft_inreg.h"
#include "sgx_internal.h"

namespace sgx {
namespace {

using namespace inreg;

struct Ist2Args {
    const void *spec;  // [batch][1025][n_frames] complex f32
    void *out;         // [batch][out_len] f32
    const void *win;   // [2048] f32
    unsigned n_frames, hop, batch, tiles, ov;
    unsigned long long start, out_len;
    float scale;
    unsigned *bad_flag;
};

constexpr int kI2FS = 8192 + 16;         // bytes per frame of ex[f][32][E 16 | O 16] (2052 dwords = 4 mod 32: conflict-free b128 writes)
constexpr int kI2Tw = 16 * kI2FS;        // 131 328: conj(W_2048^k), k < 1024 (8192 B)
constexpr int kI2Win = kI2Tw + 8192;     // the window (8192 B)
constexpr int kI2Carry = kI2Win + 8192;  // the carry, ov * hop <= 2047 floats
constexpr int kI2Lds = kI2Carry + 8192;  // 155 904 B: one workgroup per CU

// Interior tiles at hop 512 / 1024 / 2048: every index a compile-time constant, an offset owned by one thread (no barrier between the
// carry's reads and writes)
template <unsigned HOP, unsigned NT>
__device__ __forceinline__ void ola2_fast(const Ist2Args &a, const unsigned char *smem, const float *w, float *carry, unsigned tid, unsigned b,
                                          unsigned F, bool store) {
    constexpr unsigned Q = 2048u / HOP, OV = Q - 1u;
    static_assert(HOP >= NT && HOP % NT == 0, "fast overlap-add: one owner per offset");
    const float *fr = (const float *)smem;
    float *o = (float *)a.out + (size_t)b * a.out_len + ((unsigned long long)F * HOP - a.start);
#pragma unroll
    for (unsigned k = 0; k < HOP / NT; ++k) {
        const unsigned off = tid + k * NT;
        if (store) {
            float nrm = 0.f;  // ascending frame = descending sample index: the reference's order
#pragma unroll
            for (unsigned i = Q; i-- > 0;) {
                const float wj = w[i * HOP + off];
                nrm = __fadd_rn(nrm, __fmul_rn(wj, wj));
            }
            const bool div = nrm > 1e-10f;
#pragma unroll
            for (unsigned hb = 0; hb < 16u; ++hb) {
                float acc = hb < OV ? carry[hb * HOP + off] : 0.f;
#pragma unroll
                for (unsigned d = (hb < OV ? hb : OV) + 1u; d-- > 0;) acc += fr[(hb - d) * 2048u + d * HOP + off];  // frames hb - d, ascending
                o[hb * HOP + off] = div ? acc / nrm : acc;
            }
        }
#pragma unroll
        for (unsigned hb2 = 0; hb2 < OV; ++hb2) {
            float acc = 0.f;
#pragma unroll
            for (unsigned d = OV; d > hb2; --d) acc += fr[(16u + hb2 - d) * 2048u + d * HOP + off];  // rows 16 + hb2 - d <= 15
            carry[hb2 * HOP + off] = acc;
        }
    }
}

// general walk (any hop >= 128, edge tiles): see istft_ola_carry in kernels_c2c1024.hip — the same sums at a frame length of 2048
template <unsigned NT>
__device__ __forceinline__ void ola2_carry(const Ist2Args &a, const unsigned char *smem, const float *w, float *carry, unsigned tid, unsigned b,
                                           unsigned F, bool store) {
    const float *fr = (const float *)smem;
    float *o = (float *)a.out + (size_t)b * a.out_len;
    const unsigned hop = a.hop, ov = a.ov;
    const unsigned long long p0 = (unsigned long long)F * hop;
    const bool interior = F >= ov && F + 15u < a.n_frames && p0 >= a.start && p0 + 16ull * hop <= a.start + a.out_len;
    if (interior) {  // (uniform)
        if (hop == 512u) return ola2_fast<512, NT>(a, smem, w, carry, tid, b, F, store);
        if (hop == 1024u) return ola2_fast<1024, NT>(a, smem, w, carry, tid, b, F, store);
        if (hop == 2048u) return ola2_fast<2048, NT>(a, smem, w, carry, tid, b, F, store);
    }
    const bool small = hop < NT;
    const unsigned nrep = small ? NT / hop : 1u, g = small ? tid / hop : 0u, ostep = small ? hop : NT;
    const unsigned off0 = small ? tid - g * hop : tid;
    if (g < nrep && store) {
        for (unsigned off = off0; off < hop; off += ostep) {
            const unsigned q = (2048u - off + hop - 1u) / hop, back = q - 1u;  // frames h - back .. h overlap this offset
            float nrm_full = 0.f;
            for (unsigned i = q; i-- > 0;) {
                const float wj = w[i * hop + off];
                nrm_full = __fadd_rn(nrm_full, __fmul_rn(wj, wj));
            }
            for (unsigned hb = g; hb < 16u; hb += nrep) {
                const unsigned h = F + hb;
                const unsigned long long pos = p0 + (unsigned long long)hb * hop + off;
                float acc = hb < back ? carry[hb * hop + off] : 0.f;
                const unsigned r_lo = hb < back ? 0u : hb - back;
                const float *src = fr + r_lo * 2048u + (hb - r_lo) * hop + off;
                for (unsigned r = r_lo; r <= hb; ++r) {  // next frame: row + 1, sample index - hop
                    acc += *src;
                    src += 2048 - (int)hop;
                }
                float nrm = nrm_full;
                if (!interior) {
                    if (pos < a.start || pos - a.start >= a.out_len) continue;
                    const long long f_lo = (long long)h - (long long)back < 0 ? 0ll : (long long)h - (long long)back;
                    const long long f_hi = h < a.n_frames ? (long long)h : (long long)a.n_frames - 1;
                    if ((unsigned)(f_hi - f_lo + 1) != q || f_hi < f_lo) {  // signal edges: only the frames that exist count, ascending
                        nrm = 0.f;
                        for (long long f = f_lo; f <= f_hi; ++f) {
                            const float wj = w[(unsigned)((long long)h - f) * hop + off];
                            nrm = __fadd_rn(nrm, __fmul_rn(wj, wj));
                        }
                    }
                }
                if (nrm > 1e-10f) acc /= nrm;
                o[pos - a.start] = acc;
            }
        }
    }
    __syncthreads();  // every carry value has been read
    if (g < nrep) {
        for (unsigned off = off0; off < hop; off += ostep) {
            const unsigned q = (2048u - off + hop - 1u) / hop, back = q - 1u;
            for (unsigned hb2 = g; hb2 < ov; hb2 += nrep) {
                const unsigned hb = 16u + hb2;
                float acc = 0.f;
                if (hb <= 15u + back) {  // the offset reaches back into this tile: rows hb - back .. 15
                    const unsigned r_lo = hb - back;
                    const float *src = fr + r_lo * 2048u + (hb - r_lo) * hop + off;
                    for (unsigned r = r_lo; r < 16u; ++r) {
                        acc += *src;
                        src += 2048 - (int)hop;
                    }
                }
                carry[hb2 * hop + off] = acc;
            }
        }
    }
}

__global__ __launch_bounds__(512, 2) void k_istft2048(Ist2Args a, const v2f *twr, const v2f *tw1, unsigned per_xcd, unsigned total_runs, unsigned slots,
                                                      unsigned runs_per_signal, unsigned run_len) {
    constexpr unsigned NT = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x;
    v2f *twl = (v2f *)(smem + kI2Tw);  // conj(W_2048^k), k < 1024
    twl[tid] = twr[tid];
    twl[tid + 512u] = twr[tid + 512u];
    ((v4f *)(smem + kI2Win))[tid] = ((const v4f *)a.win)[tid];
    float *carry = (float *)(smem + kI2Carry);
    __syncthreads();
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned lo = xcd * per_xcd, hi = min(lo + per_xcd, total_runs);
    const unsigned lane = tid & 63u, jq = lane >> 4, fl = lane & 15u;
    const unsigned J = (tid >> 6) + 8u * jq;
    const bool j0 = J == 0u;
    const unsigned nf8 = a.n_frames * 8u;
    // pair p: bin kp and its mirror 1024 - kp.  General job: kp = J + 64 p (p < 8), J + 512 + 64 (p - 8).  Job 0: kp = 64 p and 32 + 64 (p - 8);
    // bin 512 pairs with itself and is handled apart.
    const unsigned ka = j0 ? 0u : J, kb = j0 ? 32u : J + 512u;
    const unsigned st = 64u * nf8;  // byte offsets are stepped by 64 bins
    v2f P[16], Q[16], X512;
    auto request = [&](unsigned b, unsigned t) {
        const unsigned f = 16u * t + fl;
        const unsigned fcl = f < a.n_frames ? f : 0u;  // a frame past the signal reads frame 0 and is replaced by zeros in the fold
        const unsigned char *inb = (const unsigned char *)a.spec + (size_t)b * 1025u * a.n_frames * 8u;
        unsigned oa = ka * nf8 + fcl * 8u, oy = (1024u - ka) * nf8 + fcl * 8u;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            if (p == 8) {
                oa = kb * nf8 + fcl * 8u;
                oy = (1024u - kb) * nf8 + fcl * 8u;
            }
            P[p] = *(const v2f *)(inb + oa);
            Q[p] = *(const v2f *)(inb + oy);
            oa += st;
            oy -= st;
        }
        X512 = *(const v2f *)(inb + 512u * nf8 + fcl * 8u);
    };
    auto run_of = [&](unsigned rid, unsigned &b, unsigned &t0, unsigned &t1, unsigned &ts) {
        b = rid / runs_per_signal;
        t0 = (rid - b * runs_per_signal) * run_len;
        t1 = min(a.tiles, t0 + run_len);
        ts = (t0 > 0u && a.ov) ? t0 - 1u : t0;
    };
    unsigned rid = lo + slot, b = 0, t0 = 0, t1 = 0, t = 0;
    if (rid < hi) {
        run_of(rid, b, t0, t1, t);
        if (t0 >= t1) rid = hi;
    }
    bool fresh = true;  // the run has just started: its carry is zero
    if (rid < hi) request(b, t);
    // stage-B identity and constants
    const unsigned f2 = tid >> 5, n2 = tid & 31u, nl = n2 & 15u;
    while (rid < hi) {
        const unsigned F = 16u * t;
        unsigned nrid = rid, nb = b, nt0 = t0, nt1 = t1, nt = t + 1u;
        if (nt >= t1) {
            nrid = rid + slots;
            if (nrid < hi) run_of(nrid, nb, nt0, nt1, nt);
        }
        if (fresh) {
#pragma unroll
            for (int q = 0; q < 4; ++q) carry[tid + 512u * q] = 0.f;  // (ordered before its first use by the barriers below)
        }
        {
            const unsigned f = F + fl;
            const bool valid = f < a.n_frames;
            if (!valid) {  // a frame past the signal is zeros by a select: the frame 0 loaded in its place may hold Inf / NaN, which a product with 0 would spread into the tail (the reference poisons only the samples frame 0 covers, spectrogram.rs:4906-4925)
#pragma unroll
                for (int p = 0; p < 16; ++p) P[p] = Q[p] = (v2f){0.f, 0.f};
                X512 = (v2f){0.f, 0.f};
            }
            v2f PA[16], QB[16];
            const v2f *tp = twl + ka;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                if (p == 8) tp = twl + kb;
                v2f Pp = P[p], Qp = Q[p];
                if (p == 0) {  // job 0: kp = 0 — DC and Nyquist bins: realfft ignores (and reports) their imaginary parts
                    if (j0) {
                        if (a.bad_flag && valid && (Pp.y != 0.f || Qp.y != 0.f)) atomicOr(a.bad_flag, 1u);
                        Pp.y = 0.f;
                        Qp.y = 0.f;
                    }
                }
                const v2f cw = tp[64 * (p & 7)];  // conj(W_2048^kp)
                const v2f S = pfma(Qp, (v2f){1.f, -1.f}, Pp), D = pfma(Qp, (v2f){-1.f, 1.f}, Pp);
                const v2f T = cmulv(D, cw);
                PA[p] = pfma(swp(T), (v2f){-1.f, -1.f}, S * (v2f){1.f, -1.f});  // conj(S + i T) = v[kp]
                QB[p] = pfma(swp(T), (v2f){1.f, -1.f}, S);    // S - i T    = v[1024 - kp]
            }
            // A = even-indexed elements of row J: A[m] = v[J + 64 m]; B = odd-indexed elements of row 32 - J: B[15 - p] = v[1024 - kp].
            // Job 0: A = v[64 m] (v[512] = 2 X[512], v[64 (8 + i)] = the mirror of kp = 64 (8 - i)), B = v[32 + 64 m].
            v2f A[16], B[16];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                A[i] = PA[i];
                B[i] = j0 ? PA[8 + i] : QB[15 - i];
                B[15 - i] = j0 ? QB[8 + i] : QB[i];
            }
            A[8] = j0 ? X512 * (v2f){2.f, 2.f} : PA[8];
#pragma unroll
            for (int i = 1; i < 8; ++i) A[8 + i] = j0 ? QB[8 - i] : PA[8 + i];
            // the pairs are consumed: the next tile's go out now and land during the rest of this tile
            if (nrid < hi) request(nb, nt);
            Fft<16, false>::run(A, A);  // E[n] of row J
            Fft<16, false>::run(B, B);  // O[n] of row 32 - J
            const unsigned rowE = J, rowO = (32u - J) & 31u;
            v4f *da = (v4f *)(smem + fl * kI2FS + rowE * 256u), *db = (v4f *)(smem + fl * kI2FS + rowO * 256u + 128u);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                da[c] = (v4f){A[2 * c].x, A[2 * c].y, A[2 * c + 1].x, A[2 * c + 1].y};
                db[c] = (v4f){B[2 * c].x, B[2 * c].y, B[2 * c + 1].x, B[2 * c + 1].y};
            }
        }
        __syncthreads();  // ex complete
        v2f v[32];
        {
            // (loaded per tile, like k_istft1024c: held across the fold they cost 26 registers next to the prefetched pairs)
            v2f twa[4], twb[8];  // W_1024^(k1 n2) = twa[k1 >> 3] * twb[k1 & 7]
#pragma unroll
            for (int q = 0; q < 4; ++q) twa[q] = tw1[32 * 8 * q + n2];
#pragma unroll
            for (int q = 0; q < 8; ++q) twb[q] = tw1[32 * q + n2];
            // the row transform's last radix-2 step: u = E + c O, c = +- W_32^(n2 mod 16) (= W_1024^(16 * 2 nl), a table entry)
            const v2f c32 = tw1[32 * 16 + 2u * nl] * (n2 < 16u ? (v2f){1.f, 1.f} : (v2f){-1.f, -1.f});
            const unsigned char *src = smem + f2 * kI2FS + nl * 8u;
#pragma unroll
            for (int k1 = 0; k1 < 32; ++k1) {
                const v2f E = *(const v2f *)(src + k1 * 256), O = *(const v2f *)(src + k1 * 256 + 128);
                v2f u = pfma(swp(O), (v2f){-c32.y, c32.y}, pfma(O, lo2(c32), E));  // E + c O
                const int qa = k1 >> 3, qb = k1 & 7;
                if (qb) u = cmulv(u, twb[qb]);
                if (qa) u = cmulv(u, twa[qa]);
                v[k1] = u;
            }
            Fft<32, false>::run(v, v);
        }
        __syncthreads();  // exchange buffer consumed: overlay the real frames
        {
            const v2f *w2 = (const v2f *)(smem + kI2Win) + n2;
            v2f *fr2 = (v2f *)smem + f2 * 1024u + n2;
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) {
                const v2f ww = w2[32 * n1];
                const v2f sc = v[n1] * (v2f){a.scale, -a.scale};  // conj + 1/n: (x[2n], x[2n+1]), n = n2 + 32 n1
                fr2[32 * n1] = (v2f){__fmul_rn(sc.x, ww.x), __fmul_rn(sc.y, ww.y)};
            }
        }
        __syncthreads();
        ola2_carry<NT>(a, smem, (const float *)(smem + kI2Win), carry, tid, b, F, t >= t0);
        __syncthreads();  // the frames are consumed and the carry is complete
        fresh = nrid != rid;
        rid = nrid; b = nb; t0 = nt0; t1 = nt1; t = nt;
    }
}

}  // namespace

hipError_t launch_istft2048(const void *spec, void *out, const void *win, unsigned n_frames, unsigned hop, unsigned batch, unsigned long long start,
                            unsigned long long out_len, float scale, unsigned *bad_flag, const void *twr, const void *tw1, hipStream_t s) {
    if (hop == 0 || hop > 2048) return hipErrorInvalidConfiguration;
    Ist2Args a{};
    a.spec = spec; a.out = out; a.win = win;
    a.n_frames = n_frames; a.hop = hop; a.batch = batch;
    a.ov = 2047u / hop;
    if (a.ov >= 16) return hipErrorInvalidConfiguration;  // hop >= 128
    const unsigned long long full = (unsigned long long)(n_frames - 1) * hop + 2048ull;
    const unsigned long long blocks = (full + hop - 1) / hop;
    a.tiles = (unsigned)((blocks + 15u) / 16u);
    a.start = start; a.out_len = out_len; a.scale = scale; a.bad_flag = bad_flag;
    if ((unsigned long long)a.tiles * batch >= 0x7fffffffull || a.tiles == 0) return hipErrorInvalidConfiguration;
    hipError_t e = set_max_dynamic_lds((const void *)k_istft2048, kI2Lds);
    if (e != hipSuccess) return e;
    // runs of consecutive tiles, one workgroup per CU (istft_carry_runs, sgx_internal.h)
    const unsigned wgs = device_cu_count();
    unsigned R, run_len;
    istft_carry_runs(a.tiles, batch, wgs, a.ov, R, run_len);
    const unsigned total_runs = R * batch, per_xcd = (total_runs + 7u) / 8u;
    const unsigned slots = std::max(1u, std::min(per_xcd, wgs / 8u));
    hipLaunchKernelGGL(k_istft2048, dim3(8u * slots), dim3(512), kI2Lds, s, a, (const v2f *)twr, (const v2f *)tw1, per_xcd, total_runs, slots, R, run_len);
    return hipGetLastError();
}

}  // namespace sgx
