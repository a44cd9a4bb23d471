// fft2d.hip — host side of the 2-D FFT path of libspectro_hip.so (C ABI: sgx_fft2d_* in include/spectro_hip.h).
// Row R2C = the STFT engine (an internal sgx_plan with n_fft = hop = ncols, rectangular window, complex output), whose
// frame-contiguous output is the transposed intermediate [k][r]; columns / inverse rows / pointwise = kernels_fft2d.hip.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "reg_radix.h"
#include "sgx_internal.h"

using namespace sgx;

struct sgx_fft2d {
    size_t nrows = 0, ncols = 0, cb = 0;
    int dtype = SGX_F32, device = -1;
    size_t elem = 4;
    sgx_plan *rows = nullptr;              // row R2C through the STFT engine
    void *d_tw_r = nullptr, *d_tw_c = nullptr;  // e^{-2 pi i k/nrows}, e^{-2 pi i k/ncols}
    void *d_tw1c = nullptr;                     // tuned column pass (f32, nrows == 1024): W_1024^(k1*n2), [32][32]
    void *d_twr = nullptr, *d_tw1r = nullptr;   // tuned inverse row pass (f32, ncols == 1024): conj(W_1024^k) [32][16], W_512^(k1 n2) [32][16]
    void *d_inter = nullptr, *d_spec = nullptr, *d_kspec = nullptr, *d_mask = nullptr, *d_in = nullptr, *d_out = nullptr, *d_kimg = nullptr;
    size_t inter_bytes = 0, spec_bytes = 0, kspec_bytes = 0, mask_bytes = 0, in_bytes = 0, out_bytes = 0, kimg_bytes = 0;
    unsigned log2r = 0, log2c = 0, tile_r = 0, tile_c = 0;
    // chirp-z tables for a dimension that is neither a power of two nor a listed size (bluestein.hip): columns (length nrows), inverse rows (ncols)
    BsDevTables bs_r, bs_c;
    BsDevTables bs_rh;  // columns of an even length whose own chirp-z does not fit LDS: tables of length nrows / 2 (radix-2 step outside)
    void *d_half = nullptr;  // its [batch][2][cb][nrows / 2] scratch
    size_t half_bytes = 0;
    BsDevTables bs_ch;  // inverse rows of an even ncols whose own chirp-z does not fit LDS: tables of length ncols / 2 (half-length complex form)
    // what d_kspec / d_mask currently hold, and the stream they were produced on: a plan that convolves or filters batch after
    // batch with the same kernel / cut-offs (on the same stream, so the order is the stream's) prepares them once, not per call
    std::vector<unsigned char> kspec_of;
    size_t kspec_rows = 0, kspec_cols = 0;
    hipStream_t kspec_stream = nullptr;
    bool kspec_valid = false, mask_valid = false;
    // a rank-1 kernel (an outer product, e.g. gaussian_kernel_2d: image_ops.rs:188-220) on the fused f32 path: d_kouter holds the two 1-D
    // factors of its spectrum (1024 + cb complex values, 12 KB) and d_kspec is not built — k_colconv1024<MUL_OUTER> multiplies them in
    bool kspec_outer = false, kspec_outer_allowed = true;
    // ... and on 1024 x 1024 images the whole convolution runs as two passes of k_colconv1024 over pairs of REAL rows (rows, then columns:
    // fused_separable_chunk); d_kouter then also holds U / 1024 and the full-length V / 1024 behind the first two tables
    bool kspec_separable = false, kspec_separable_allowed = true;
    void *d_kouter = nullptr;
    size_t kouter_bytes = 0;
    int mask_kind = -1;
    double mask_lo = 0.0, mask_hi = 0.0;
    hipStream_t mask_stream = nullptr;
    // second stream of the chunked convolve / filter schedule (fused_product_dev), forked from and joined to the caller's
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    mutable std::string err;
};

namespace {

thread_local std::string g_err2d;
constexpr double kPi2 = 3.14159265358979323846264338327950288;

sgx_status fail(const sgx_fft2d *p, sgx_status st, const std::string &m) {
    if (p) p->err = m; else g_err2d = m;
    return st;
}
#define F2_HIP(plan, call)                                                                                            \
    do {                                                                                                              \
        hipError_t e_ = (call);                                                                                       \
        if (e_ != hipSuccess)                                                                                         \
            return fail(plan, SGX_BACKEND, std::string("hip -- FFT backend error: ") + #call + ": " + hipGetErrorString(e_)); \
    } while (0)

sgx_status grow2(sgx_fft2d *p, void **buf, size_t *have, size_t need) {
    if (*have >= need) return SGX_OK;
    if (*buf) F2_HIP(p, hipFree(*buf));
    *buf = nullptr;
    *have = 0;
    F2_HIP(p, hipMalloc(buf, need));
    *have = need;
    return SGX_OK;
}

unsigned ilog2_pow2(size_t n) {  // 0 unless n is a power of two >= 2
    if (n < 2 || (n & (n - 1))) return 0;
    unsigned l = 0;
    while ((size_t(1) << l) < n) ++l;
    return l;
}

template <typename T>
sgx_status upload_tw(sgx_fft2d *p, void **dst, size_t n) {
    std::vector<T> tw(2 * n);
    for (size_t k = 0; k < n; ++k) {
        const double a = -2.0 * kPi2 * double(k) / double(n);
        tw[2 * k] = T(std::cos(a));
        tw[2 * k + 1] = T(std::sin(a));
    }
    F2_HIP(p, hipMalloc(dst, tw.size() * sizeof(T)));
    F2_HIP(p, hipMemcpy(*dst, tw.data(), tw.size() * sizeof(T), hipMemcpyHostToDevice));
    return SGX_OK;
}

template <typename T>
sgx_status upload_bs(sgx_fft2d *p, BsDevTables &d, const BsHostTables &h) {
    auto up = [&](void **dst, const std::vector<double> &v) -> sgx_status {
        std::vector<T> c(v.begin(), v.end());
        F2_HIP(p, hipMalloc(dst, c.size() * sizeof(T)));
        F2_HIP(p, hipMemcpy(*dst, c.data(), c.size() * sizeof(T), hipMemcpyHostToDevice));
        return SGX_OK;
    };
    sgx_status st;
    if ((st = up(&d.chirp, h.chirp)) != SGX_OK || (st = up(&d.bhp, h.bhp)) != SGX_OK || (st = up(&d.tw, h.tw)) != SGX_OK) return st;
    d.M = h.M;
    return SGX_OK;
}

// would a length take the chirp-z kernels: not a power of two (those have their own radix-2 tile kernel at any size that fits), no
// register-tiled pass split, and a convolution length that fits LDS
bool wants_bluestein(size_t n, bool has_split, int dtype, BsHostTables &h) {
#ifdef SGX_NO_BS_C2C  // A/B builds only: the LDS-tile kernels (two-factor / direct sums) as before
    return false;
#endif
    return n >= 16 && (n & (n - 1)) != 0 && !has_split && bluestein_host_tables(unsigned(n), dtype, h);
}

// complex sequences: register-tiled passes, else chirp-z, else the LDS-tile kernel (radix-2 / two-factor / direct sum)
hipError_t c2c_dispatch(const sgx_fft2d *p, const C2cArgs &a, const BsDevTables &bs, hipStream_t s) {
    hipError_t e = launch_c2c_reg(a, p->dtype, s);
    if (e == hipErrorNotSupported && bs.M && !a.mul) e = launch_c2c_bluestein(a, bs, p->dtype, s);
    if (e == hipErrorNotSupported && p->bs_rh.M && !a.mul && p->half_bytes >= (size_t)a.batch * a.nseq * a.n * 2 * p->elem)
        e = launch_c2c_bluestein_split(a, p->bs_rh, p->d_half, p->dtype, s);
    return e == hipErrorNotSupported ? launch_c2c_tile(a, p->dtype, s) : e;
}
hipError_t c2r_dispatch(const sgx_fft2d *p, const C2rArgs &c, hipStream_t s) {
    hipError_t e = launch_c2r_reg(c, p->dtype, s);
    if (e == hipErrorNotSupported && p->bs_c.M) e = launch_c2r_bluestein(c, p->bs_c, p->dtype, s);
    if (e == hipErrorNotSupported && p->bs_ch.M) e = launch_c2r_bluestein(c, p->bs_ch, p->dtype, s, true);
    return e == hipErrorNotSupported ? launch_c2r_rows(c, p->dtype, s) : e;
}

// the radix-2-outside column pass (bs_rh) works through a scratch of the spectrum's size
sgx_status grow_half(sgx_fft2d *p, size_t batch) {
    if (!p->bs_rh.M) return SGX_OK;
    return grow2(p, &p->d_half, &p->half_bytes, batch * p->cb * p->nrows * 2 * p->elem);
}

// device pointers in, device pointers out
// `mul` (optional): kernel spectrum / real mask [R][Cb] multiplied into the result (convolve_fft, filters) — fused into the
// column kernel's store where the register-tiled kernel runs, a k_pointwise launch otherwise
sgx_status forward_dev(sgx_fft2d *p, const void *img, size_t batch, void *spec, hipStream_t s, const void *mul = nullptr,
                       int mul_real = 0) {
    const size_t R = p->nrows, C = p->ncols, Cb = p->cb;
    sgx_status st = grow2(p, &p->d_inter, &p->inter_bytes, batch * Cb * R * 2 * p->elem);
    if (st != SGX_OK) return st;
    if ((st = grow_half(p, batch)) != SGX_OK) return st;
    // rows: every image is one "signal" of R*C samples, frames = rows -> inter[b][k][r]
    st = sgx_execute(p->rows, img, batch, R * C, R * C, p->d_inter, batch * Cb * R * 2, SGX_MEM_DEVICE, s);
    if (st != SGX_OK) return fail(p, st, sgx_last_error(p->rows));
    C2cArgs a{};
    a.in = p->d_inter; a.out = spec;
    a.n = unsigned(R); a.log2n = p->log2r; a.nseq = unsigned(Cb); a.batch = unsigned(batch);
    a.in_img = Cb * R; a.out_img = R * Cb;
    a.in_ss = R; a.in_is = 1; a.out_ss = 1; a.out_is = Cb;
    a.tile = p->tile_r; a.tiles = unsigned((Cb + a.tile - 1) / a.tile);
    a.tw = p->d_tw_r; a.inverse = 0; a.in_seq_fast = 0; a.out_seq_fast = 1; a.scale = 1.0;
    bool fused_mul = false;
    if (p->d_tw1c) {
        a.tile = 16; a.tiles = unsigned((Cb + 15) / 16);
        F2_HIP(p, launch_c2c1024(a, p->d_tw1c, s));
    } else {
        a.mul = mul; a.mul_ks = Cb; a.mul_real = mul_real;
        const hipError_t e = launch_c2c_reg(a, p->dtype, s);
        if (e == hipErrorNotSupported) {
            a.mul = nullptr;
            F2_HIP(p, c2c_dispatch(p, a, p->bs_r, s));
        } else {
            F2_HIP(p, e);
            fused_mul = true;
        }
    }
    if (mul && !fused_mul) F2_HIP(p, launch_pointwise(spec, mul, spec, batch * R * Cb, R * Cb, mul_real, p->dtype, s));
    return SGX_OK;
}

sgx_status inverse_dev(sgx_fft2d *p, const void *spec, size_t batch, void *img, hipStream_t s) {
    const size_t R = p->nrows, C = p->ncols, Cb = p->cb;
    sgx_status st = grow2(p, &p->d_inter, &p->inter_bytes, batch * Cb * R * 2 * p->elem);
    if (st != SGX_OK) return st;
    if ((st = grow_half(p, batch)) != SGX_OK) return st;
    C2cArgs a{};
    a.in = spec; a.out = p->d_inter;
    a.n = unsigned(R); a.log2n = p->log2r; a.nseq = unsigned(Cb); a.batch = unsigned(batch);
    a.in_img = R * Cb; a.out_img = Cb * R;
    a.in_ss = 1; a.in_is = Cb; a.out_ss = R; a.out_is = 1;
    a.tile = p->tile_r; a.tiles = unsigned((Cb + a.tile - 1) / a.tile);
    a.tw = p->d_tw_r; a.inverse = 1; a.in_seq_fast = 1; a.out_seq_fast = 0; a.scale = 1.0;
    C2rArgs c{};
    c.in = p->d_inter; c.out = img;
    c.nrows = unsigned(R); c.ncols = unsigned(C); c.log2c = p->log2c; c.batch = unsigned(batch);
    c.in_img = Cb * R;
    if (p->d_tw1c) {  // tuned columns write [r][k] (sequence-contiguous stores); the row pass then reads rows contiguously
        a.out_ss = 1; a.out_is = Cb; a.out_seq_fast = 1;
        a.tile = 16; a.tiles = unsigned((Cb + 15) / 16);
        F2_HIP(p, launch_c2c1024(a, p->d_tw1c, s));
        c.in_ks = 1; c.in_rs = Cb; c.k_fast = 1;
    } else {
        F2_HIP(p, c2c_dispatch(p, a, p->bs_r, s));
        c.in_ks = R; c.in_rs = 1; c.k_fast = 0;
    }
    c.tile = p->tile_c; c.tiles = unsigned((R + c.tile - 1) / c.tile);
    c.tw = p->d_tw_c; c.scale = 1.0 / (double(R) * double(C));
    if (p->d_twr && c.in_ks == 1) {
        c.tile = 16; c.tiles = unsigned((R + 15) / 16);
        F2_HIP(p, launch_c2r1024(c, p->d_twr, p->d_tw1r, s));
    } else {
        F2_HIP(p, c2r_dispatch(p, c, s));
    }
    return SGX_OK;
}

// convolve_fft / filters with the fused column stage (f32, 1024 rows): rows R2C -> k_colconv1024 -> rows C2R.  `mul` is the
// kernel's half spectrum (complex) or a real mask, both [row][col].
//
// The three passes of one image group are bound by different things (the row passes by HBM, the column pass by LDS and VALU), so a
// large batch runs as chunks of kConvChunk images alternating between the caller's stream and a plan-owned second stream: the row
// passes of one chunk overlap the column pass of the other (measured on 512 x 1024^2: 2.83 -> 2.70 ms; chunks of 32: 2.72, of 128:
// 2.75; profiles/experiments_r03/convolve_two_stream_in_library_abv.txt).  The second stream forks from and joins the caller's stream through
// events, so the call keeps plain stream semantics (and stays capturable into a hipGraph); every image's arithmetic is the same
// in either schedule, so the result does not depend on the batch it came in.
#ifndef SGX_CONV_CHUNK
#define SGX_CONV_CHUNK 64
#endif
constexpr size_t kConvChunk = SGX_CONV_CHUNK;

// The spectrum between the column kernel and the inverse row pass is the plan's own scratch, so its row pitch is ours to choose: with the
// tuned inverse row pass (1024 columns: 513 bins) the rows are padded to 528 bins = 33 whole 128-byte lines.  k_colconv1024's 128-byte
// store segments (16 columns of one row) then are whole lines; at 513 bins (4104 B) every segment straddled two, and the partial-line
// writes cost 5.1 MB of WRITE_SIZE per image for 4.2 MB of data (round-5 PMC pass; the complex STFT's pitch experiment).
#ifdef SGX_SPEC_PITCH_OFF  // (A/B: the unpadded rows)
size_t fused_spec_pitch(const sgx_fft2d *p) { return p->cb; }
#else
size_t fused_spec_pitch(const sgx_fft2d *p) { return p->d_twr ? (p->cb + 15) / 16 * 16 : p->cb; }
#endif

// A rank-1 kernel on a 1024 x 1024 image: the 2-D transform pair factors into (row FFT . V . row IFFT) and (column FFT . U . column IFFT) —
// the same linear operators as fft2d . (U V^T) . ifft2d in another order — and each factor is k_colconv1024 on pairs of real rows packed
// as one complex sequence (a real kernel's spectrum is Hermitian: real part = the first row's convolution, imaginary = the second's).
// The kernel's transposing store puts the pair back side by side, so pass 1 (rows of the image) leaves the transposed image in `inter`
// and pass 2 (its rows = the image's columns) the result in place: two passes of 4 MiB read + 4 MiB written per image instead of
// three of 8.4 MB.  Each table carries 1 / 1024 (exact), so nothing is left to normalise.
sgx_status fused_separable_chunk(sgx_fft2d *p, const void *img, size_t batch, const void *tables, void *out, void *inter, hipStream_t s) {
    const size_t R = p->nrows, C = p->ncols;
    const char *t = static_cast<const char *>(tables);
    const void *vfull = t + (R + p->cb + R) * 2 * sizeof(float), *uscaled = t + (R + p->cb) * 2 * sizeof(float);
    C2cArgs a{};
    a.n = 1024; a.log2n = 10; a.batch = unsigned(batch);
    a.in_img = R * C; a.out_img = R * C;
    a.tile = 16;
    a.in = img; a.out = inter; a.nseq = unsigned(R / 2); a.in_ss = C; a.out_is = R; a.tiles = unsigned((a.nseq + 15) / 16);
    F2_HIP(p, launch_colconv1024(a, p->d_tw1c, vfull, 0, MUL_VEC, s, true));   // rows: inter[col][row]
    a.in = inter; a.out = out; a.nseq = unsigned(C / 2); a.in_ss = R; a.out_is = C; a.tiles = unsigned((a.nseq + 15) / 16);
    F2_HIP(p, launch_colconv1024(a, p->d_tw1c, uscaled, 0, MUL_VEC, s, true));  // columns: out[row][col]
    return SGX_OK;
}

sgx_status fused_product_chunk(sgx_fft2d *p, const void *img, size_t batch, const void *mul, int mul_kind, void *out, void *inter,
                               void *spec, hipStream_t s) {
    const size_t R = p->nrows, C = p->ncols, Cb = p->cb, Cp = fused_spec_pitch(p);
    sgx_status st = sgx_execute(p->rows, img, batch, R * C, R * C, inter, batch * Cb * R * 2, SGX_MEM_DEVICE, s);
    if (st != SGX_OK) return fail(p, st, sgx_last_error(p->rows));
    C2cArgs a{};
    a.in = inter; a.out = spec;
    a.n = unsigned(R); a.log2n = p->log2r; a.nseq = unsigned(Cb); a.batch = unsigned(batch);
    a.in_img = Cb * R; a.out_img = R * Cp;
    a.in_ss = R; a.in_is = 1; a.out_ss = 1; a.out_is = Cp;
    a.tile = 16; a.tiles = unsigned((Cb + 15) / 16);
    F2_HIP(p, launch_colconv1024(a, p->d_tw1c, mul, Cb, mul_kind, s));
    C2rArgs c{};
    c.in = spec; c.out = out;
    c.nrows = unsigned(R); c.ncols = unsigned(C); c.log2c = p->log2c; c.batch = unsigned(batch);
    c.in_img = Cp * R; c.in_ks = 1; c.in_rs = Cp; c.k_fast = 1;
    c.tw = p->d_tw_c; c.scale = 1.0 / (double(R) * double(C));
    if (p->d_twr) {
        c.tile = 16; c.tiles = unsigned((R + 15) / 16);
        F2_HIP(p, launch_c2r1024(c, p->d_twr, p->d_tw1r, s));
    } else {
        c.tile = p->tile_c; c.tiles = unsigned((R + c.tile - 1) / c.tile);
        F2_HIP(p, c2r_dispatch(p, c, s));
    }
    return SGX_OK;
}

// chunked only where all three passes are the tuned kernels (1024 x 1024 f32): those use no plan-owned scratch besides the two
// buffers split below, so two chunks can be in flight at once (a Bluestein row plan, say, owns scratch that cannot be shared)
bool fused_chunked(const sgx_fft2d *p, size_t batch) { return p->d_twr != nullptr && batch >= 2 * kConvChunk; }
size_t fused_scratch_images(const sgx_fft2d *p, size_t batch) { return fused_chunked(p, batch) ? 2 * kConvChunk : batch; }

sgx_status fused_streams(sgx_fft2d *p) {
    if (p->aux_stream) return SGX_OK;
    F2_HIP(p, hipStreamCreateWithFlags(&p->aux_stream, hipStreamNonBlocking));
    F2_HIP(p, hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
    F2_HIP(p, hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
    return SGX_OK;
}

sgx_status fused_product_dev(sgx_fft2d *p, const void *img, size_t batch, const void *mul, int mul_kind, void *out, hipStream_t s) {
    const size_t R = p->nrows, C = p->ncols, Cb = p->cb;
    const size_t slice = Cb * R * 2 * p->elem;  // one image's intermediate / spectrum
    sgx_status st;
    if (mul_kind == MUL_VEC) {
        // The separable passes are one kind of kernel: a second stream has nothing of another kind to overlap, and every launch boundary
        // is a tail of idle CUs.  512 images: chunks of 16 / 32 / 64 / 128 on two streams 2.09 / 1.95 / 1.83 / 1.79 ms, groups of 128 one
        // after the other 1.90, one launch pair 1.78.  So: groups of up to kSepGroup images on the caller's stream, the transposed
        // intermediate (4 MiB per image) in d_inter.
        size_t kSepGroup = 512;
        if (const char *g = std::getenv("SGX_SEP_GROUP")) kSepGroup = std::max<size_t>(1, std::strtoull(g, nullptr, 10));  // (tests: several groups in a small batch)
        const size_t group = std::min(batch, kSepGroup), img_bytes = R * C * p->elem;
        if ((st = grow2(p, &p->d_inter, &p->inter_bytes, group * img_bytes)) != SGX_OK) return st;
        for (size_t b0 = 0; b0 < batch; b0 += group)
            if ((st = fused_separable_chunk(p, static_cast<const char *>(img) + b0 * img_bytes, std::min(group, batch - b0), mul,
                                            static_cast<char *>(out) + b0 * img_bytes, p->d_inter, s)) != SGX_OK)
                return st;
        return SGX_OK;
    }
    if ((st = grow2(p, &p->d_inter, &p->inter_bytes, fused_scratch_images(p, batch) * slice)) != SGX_OK) return st;
    const size_t pslice = fused_spec_pitch(p) * R * 2 * p->elem;  // one image's spectrum at the padded pitch
    if ((st = grow2(p, &p->d_spec, &p->spec_bytes, fused_scratch_images(p, batch) * pslice)) != SGX_OK) return st;
    if (!fused_chunked(p, batch)) return fused_product_chunk(p, img, batch, mul, mul_kind, out, p->d_inter, p->d_spec, s);
    if ((st = fused_streams(p)) != SGX_OK) return st;
    F2_HIP(p, hipEventRecord(p->ev_fork, s));
    F2_HIP(p, hipStreamWaitEvent(p->aux_stream, p->ev_fork, 0));
    const size_t img_bytes = R * C * p->elem;
    size_t idx = 0;
    for (size_t b0 = 0; b0 < batch; b0 += kConvChunk, ++idx) {
        const size_t nb = std::min(kConvChunk, batch - b0), half = idx & 1;  // each stream owns one half of the scratch
        st = fused_product_chunk(p, static_cast<const char *>(img) + b0 * img_bytes, nb, mul, mul_kind,
                                 static_cast<char *>(out) + b0 * img_bytes, static_cast<char *>(p->d_inter) + half * kConvChunk * slice,
                                 static_cast<char *>(p->d_spec) + half * kConvChunk * pslice, half ? p->aux_stream : s);
        if (st != SGX_OK) break;
    }
    // join even after a failed launch: the caller's stream must not run ahead of work already queued on the second one
    const hipError_t e1 = hipEventRecord(p->ev_join, p->aux_stream);
    const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(s, p->ev_join, 0) : e1;
    if (st != SGX_OK) return st;
    F2_HIP(p, e2);
    return SGX_OK;
}

bool use_fused(const sgx_fft2d *p) { return p->d_tw1c != nullptr; }

// Is the kernel an outer product u v^T to f32 rounding?  Pivot on the largest |K|: u = its column, v = its row / the pivot, in f64; every
// element within 2^-22 max|K| of u_i v_j (gaussian_kernel_2d, computed in f64 and rounded once to f32, is within 1.8e-7).  The kernel's
// padded image (pad_kernel_for_fft, image_ops.rs:123-152: centre to (0,0), wrapped) is then the outer product of the two padded vectors
// and its 2-D spectrum the outer product of their 1-D spectra, built here in f64 from the definition (angles reduced in integers):
// U[k] = sum_i u_i e^{-2 pi i k t(i) / R}, t(i) = (i - krows / 2) mod R; V[c] likewise over the columns, c = 0 .. C / 2.
bool outer_product_spectrum(const float *K, size_t krows, size_t kcols, size_t R, size_t C, size_t Cb, std::vector<float> &uv) {
    size_t pi = 0, pj = 0;
    double big = 0.0;
    for (size_t i = 0; i < krows; ++i)
        for (size_t j = 0; j < kcols; ++j) {
            const double m = std::fabs(double(K[i * kcols + j]));
            if (!(m <= 3.0e38)) return false;  // non-finite: the general path carries it as the reference does
            if (m > big) { big = m; pi = i; pj = j; }
        }
    if (big == 0.0) return false;
    std::vector<double> u(krows), v(kcols);
    const double piv = double(K[pi * kcols + pj]);
    for (size_t i = 0; i < krows; ++i) u[i] = double(K[i * kcols + pj]);
    for (size_t j = 0; j < kcols; ++j) v[j] = double(K[pi * kcols + j]) / piv;
    const double tol = big * 0x1p-22;
    for (size_t i = 0; i < krows; ++i)
        for (size_t j = 0; j < kcols; ++j)
            if (std::fabs(double(K[i * kcols + j]) - u[i] * v[j]) > tol) return false;
    uv.assign((R + Cb + R + C) * 2, 0.f);  // U, V (half), then for the separable passes U / R and the full-length V / C
    auto spectrum = [&](const std::vector<double> &w, size_t n, size_t bins, float *dst, double scale) {
        std::vector<double> cs(n), sn(n);
        for (size_t q = 0; q < n; ++q) {
            const double a = -2.0 * kPi2 * double(q) / double(n);
            cs[q] = std::cos(a);
            sn[q] = std::sin(a);
        }
        const size_t centre = w.size() / 2;
        for (size_t k = 0; k < bins; ++k) {
            double re = 0.0, im = 0.0;
            for (size_t i = 0; i < w.size(); ++i) {
                const size_t t = (i + n - centre % n) % n, q = (k * t) % n;
                re += w[i] * cs[q];
                im += w[i] * sn[q];
            }
            dst[2 * k] = float(re * scale);
            dst[2 * k + 1] = float(im * scale);
        }
    };
    spectrum(u, R, R, uv.data(), 1.0);
    spectrum(v, C, Cb, uv.data() + 2 * R, 1.0);
    spectrum(u, R, R, uv.data() + 2 * (R + Cb), 1.0 / double(R));
    spectrum(v, C, C, uv.data() + 2 * (R + Cb + R), 1.0 / double(C));
    return true;
}

// create_lowpass_mask (image_ops.rs:236-267) on the half spectrum's own dims (quirk S14), f64 logic
void lowpass_mask(size_t nrows, size_t ncols, double cutoff, std::vector<double> &m) {
    m.assign(nrows * ncols, 0.0);
    const double mr = double(nrows / 2), mc = double(ncols / 2);
    const double q = std::min(mr, mc) * cutoff;
    const double max_radius = q * q;
    for (size_t i = 0; i < nrows; ++i)
        for (size_t j = 0; j < ncols; ++j) {
            const double fr = i <= nrows / 2 ? double(i) : std::fabs(double(i) - double(nrows));
            const double fc = j <= ncols / 2 ? double(j) : std::fabs(double(j) - double(ncols));
            if (std::fma(fc, fc, fr * fr) <= max_radius) m[i * ncols + j] = 1.0;
        }
}

template <typename F>
sgx_status with_staging(sgx_fft2d *p, const void *in, size_t in_bytes, void *out, size_t out_bytes, int mem_kind,
                        hipStream_t s, F body) {
    DeviceGuard dg;
    F2_HIP(p, dg.enter(p->device));
    if (mem_kind == SGX_MEM_DEVICE) return body(in, out);
    if (mem_kind != SGX_MEM_HOST) return fail(p, SGX_INVALID_INPUT, "Invalid input: unknown mem_kind");
    sgx_status st;
    if ((st = grow2(p, &p->d_in, &p->in_bytes, in_bytes)) != SGX_OK) return st;
    if ((st = grow2(p, &p->d_out, &p->out_bytes, out_bytes)) != SGX_OK) return st;
    F2_HIP(p, hipMemcpyAsync(p->d_in, in, in_bytes, hipMemcpyHostToDevice, s));
    if ((st = body(p->d_in, p->d_out)) != SGX_OK) return st;
    F2_HIP(p, hipMemcpyAsync(out, p->d_out, out_bytes, hipMemcpyDeviceToHost, s));
    F2_HIP(p, hipStreamSynchronize(s));
    return SGX_OK;
}

sgx_status check(sgx_fft2d *p, const void *a, const void *b, size_t batch) {
    if (!p) return SGX_INVALID_INPUT;
    if (!a || !b) return fail(p, SGX_INVALID_INPUT, "Invalid input: null buffer");
    if (batch == 0) return fail(p, SGX_INVALID_INPUT, "Invalid input: batch must be > 0");
    if (!p->rows) return fail(p, SGX_BACKEND, "hip -- FFT backend error: plan has no HIP device (host-only plan)");
    return SGX_OK;
}

}  // namespace

extern "C" {

int32_t sgx_fft2d_device(const sgx_fft2d *p) { return p ? p->device : -2; }

sgx_status sgx_fft2d_reserve(sgx_fft2d *p, size_t batch, int32_t host_staging) {
    if (!p || batch == 0) return SGX_INVALID_INPUT;
    DeviceGuard dg;
    F2_HIP(p, dg.enter(p->device));
    const size_t R = p->nrows, C = p->ncols, Cb = p->cb;
    sgx_status st;
    if ((st = grow2(p, &p->d_inter, &p->inter_bytes, batch * Cb * R * 2 * p->elem)) != SGX_OK) return st;
    if ((st = grow2(p, &p->d_spec, &p->spec_bytes, batch * R * fused_spec_pitch(p) * 2 * p->elem)) != SGX_OK) return st;  // (the fused path pads its rows)
    if (fused_chunked(p, batch) && (st = fused_streams(p)) != SGX_OK) return st;
    if ((st = grow_half(p, batch)) != SGX_OK) return st;  // nothing left to create inside a graph capture
    if (host_staging) {
        const size_t big = batch * R * Cb * 2 * p->elem;  // a half spectrum is the larger of (image, spectrum)
        if ((st = grow2(p, &p->d_in, &p->in_bytes, big)) != SGX_OK) return st;
        if ((st = grow2(p, &p->d_out, &p->out_bytes, big)) != SGX_OK) return st;
    }
    return SGX_OK;
}

const char *sgx_fft2d_last_error(const sgx_fft2d *plan) { return plan ? plan->err.c_str() : g_err2d.c_str(); }

sgx_status sgx_fft2d_create(size_t nrows, size_t ncols, int32_t dtype, int32_t device, sgx_fft2d **out) {
    if (out) *out = nullptr;
    if (!out) return fail(nullptr, SGX_INVALID_INPUT, "Invalid input: null argument");
    if (nrows == 0 || ncols == 0) return fail(nullptr, SGX_INVALID_INPUT, "Invalid input: array dimensions must be > 0");  // fft2d.rs:80-84
    if (dtype != SGX_F32 && dtype != SGX_F64) return fail(nullptr, SGX_INVALID_INPUT, "Invalid input: dtype must be f32 or f64");
    if (nrows > 0x7fffffffull || ncols > 0x7fffffffull) return fail(nullptr, SGX_INVALID_INPUT, "Invalid input: dimensions too large");
    sgx_fft2d *p = new (std::nothrow) sgx_fft2d();
    if (!p) return fail(nullptr, SGX_INTERNAL, "Internal error: out of memory");
    p->nrows = nrows; p->ncols = ncols; p->cb = ncols / 2 + 1;
    p->dtype = dtype; p->elem = dtype == SGX_F64 ? 8 : 4; p->device = device;
    p->log2r = ilog2_pow2(nrows); p->log2c = ilog2_pow2(ncols);
    p->tile_r = fft2d_tile_for(unsigned(nrows), dtype);
    p->tile_c = fft2d_tile_for(unsigned(ncols), dtype);
    if (p->tile_r == 0 || p->tile_c == 0) {
        delete p;
        return fail(nullptr, SGX_BACKEND, "hip -- FFT backend error: image dimension too large for the on-chip tile");
    }
    if (device == -2) { *out = p; return SGX_OK; }  // host-only: shapes / validation only
    sgx_params sp{};
    sp.n_fft = uint32_t(ncols); sp.hop_size = uint32_t(ncols); sp.centre = 0;
    sp.window_kind = SGX_WIN_RECTANGULAR; sp.sample_rate_hz = 1.0;
    sp.freq_scale = SGX_FREQ_LINEAR; sp.amp_scale = SGX_AMP_COMPLEX; sp.dtype = dtype; sp.device = device;
    sgx_status st = sgx_plan_create(&sp, &p->rows);
    if (st != SGX_OK) {
        g_err2d = sgx_last_create_error();
        delete p;
        return st;
    }
    p->device = p->rows->device;
    auto tables = [&]() -> sgx_status {
        DeviceGuard dg;
    F2_HIP(p, dg.enter(p->device));
        sgx_status s1 = dtype == SGX_F64 ? upload_tw<double>(p, &p->d_tw_r, nrows) : upload_tw<float>(p, &p->d_tw_r, nrows);
        if (s1 != SGX_OK) return s1;
        s1 = dtype == SGX_F64 ? upload_tw<double>(p, &p->d_tw_c, ncols) : upload_tw<float>(p, &p->d_tw_c, ncols);
        if (s1 != SGX_OK) return s1;
        {
            unsigned fa, fb, fc;
            BsHostTables h;
            if (wants_bluestein(nrows, reg_split_len(unsigned(nrows), dtype, &fa, &fb, &fc), dtype, h) &&
                (s1 = dtype == SGX_F64 ? upload_bs<double>(p, p->bs_r, h) : upload_bs<float>(p, p->bs_r, h)) != SGX_OK)
                return s1;
            // even column lengths with neither: one radix-2 step outside two half-length chirp-z transforms
            if (!p->bs_r.M && nrows % 2 == 0 && nrows >= 32 && (nrows & (nrows - 1)) != 0 && !reg_split_len(unsigned(nrows), dtype, &fa, &fb, &fc) &&
                wants_bluestein(nrows / 2, false, dtype, h) &&
                (s1 = dtype == SGX_F64 ? upload_bs<double>(p, p->bs_rh, h) : upload_bs<float>(p, p->bs_rh, h)) != SGX_OK)
                return s1;
            // (the inverse row pass is register-tiled for even ncols whose half has a split)
            if (wants_bluestein(ncols, ncols % 2 == 0 && reg_split_len(unsigned(ncols / 2), dtype, &fa, &fb, &fc), dtype, h) &&
                (s1 = dtype == SGX_F64 ? upload_bs<double>(p, p->bs_c, h) : upload_bs<float>(p, p->bs_c, h)) != SGX_OK)
                return s1;
            // even ncols without either: the half-length complex form, if THAT convolution fits (f64 4098 ... 8192, f32 8194 ... 16384)
            if (!p->bs_c.M && ncols % 2 == 0 && ncols >= 32 && (ncols & (ncols - 1)) != 0 && !reg_split_len(unsigned(ncols / 2), dtype, &fa, &fb, &fc) &&
                bluestein_host_tables(unsigned(ncols / 2), dtype, h) &&
                (s1 = dtype == SGX_F64 ? upload_bs<double>(p, p->bs_ch, h) : upload_bs<float>(p, p->bs_ch, h)) != SGX_OK)
                return s1;
        }
        if (dtype == SGX_F32 && nrows == 1024) {
            std::vector<float> t(2 * 32 * 32);
            for (unsigned k1 = 0; k1 < 32; ++k1)
                for (unsigned n2 = 0; n2 < 32; ++n2) {
                    const double a = -2.0 * kPi2 * double(k1 * n2) / 1024.0;
                    t[2 * (k1 * 32 + n2)] = float(std::cos(a));
                    t[2 * (k1 * 32 + n2) + 1] = float(std::sin(a));
                }
            F2_HIP(p, hipMalloc(&p->d_tw1c, t.size() * sizeof(float)));
            F2_HIP(p, hipMemcpy(p->d_tw1c, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        if (dtype == SGX_F32 && ncols == 1024 && p->d_tw1c) {  // the tuned row pass reads the [r][k] layout of the tuned columns
            std::vector<float> tr(2 * 32 * 16), t1(2 * 32 * 16);
            for (unsigned n1 = 0; n1 < 32; ++n1)
                for (unsigned n2 = 0; n2 < 16; ++n2) {
                    const double a = 2.0 * kPi2 * double(16 * n1 + n2) / 1024.0;  // conj(W_1024^k) = e^{+2 pi i k/1024}
                    tr[2 * (n1 * 16 + n2)] = float(std::cos(a));
                    tr[2 * (n1 * 16 + n2) + 1] = float(std::sin(a));
                    const double b2 = -2.0 * kPi2 * double(n1 * n2) / 512.0;
                    t1[2 * (n1 * 16 + n2)] = float(std::cos(b2));
                    t1[2 * (n1 * 16 + n2) + 1] = float(std::sin(b2));
                }
            F2_HIP(p, hipMalloc(&p->d_twr, tr.size() * sizeof(float)));
            F2_HIP(p, hipMemcpy(p->d_twr, tr.data(), tr.size() * sizeof(float), hipMemcpyHostToDevice));
            F2_HIP(p, hipMalloc(&p->d_tw1r, t1.size() * sizeof(float)));
            F2_HIP(p, hipMemcpy(p->d_tw1r, t1.data(), t1.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        return SGX_OK;
    };
    st = tables();
    if (st != SGX_OK) {
        g_err2d = p->err;
        sgx_fft2d_destroy(p);
        return st;
    }
    *out = p;
    return SGX_OK;
}

void sgx_fft2d_destroy(sgx_fft2d *p) {
    if (!p) return;
    if (p->rows) {
        DeviceGuard dg;
        (void)dg.enter(p->device);
        void *bufs[] = {p->d_tw_r, p->d_tw_c, p->d_tw1c, p->d_twr, p->d_tw1r, p->d_inter, p->d_spec, p->d_kspec, p->d_kouter, p->d_mask, p->d_in, p->d_out, p->d_kimg,
                        p->bs_r.chirp, p->bs_r.bhp, p->bs_r.tw, p->bs_c.chirp, p->bs_c.bhp, p->bs_c.tw, p->bs_ch.chirp, p->bs_ch.bhp, p->bs_ch.tw, p->bs_rh.chirp, p->bs_rh.bhp, p->bs_rh.tw, p->d_half};
        for (void *b : bufs)
            if (b) (void)hipFree(b);
        if (p->aux_stream) (void)hipStreamDestroy(p->aux_stream);
        if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
        if (p->ev_join) (void)hipEventDestroy(p->ev_join);
        sgx_plan_destroy(p->rows);
    }
    delete p;
}

sgx_status sgx_fft2d_forward(sgx_fft2d *p, const void *images, size_t batch, void *spectrum, int32_t mem_kind, void *stream) {
    sgx_status st = check(p, images, spectrum, batch);
    if (st != SGX_OK) return st;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t inb = batch * p->nrows * p->ncols * p->elem, outb = batch * p->nrows * p->cb * 2 * p->elem;
    return with_staging(p, images, inb, spectrum, outb, mem_kind, s,
                        [&](const void *i, void *o) { return forward_dev(p, i, batch, o, s); });
}

sgx_status sgx_fft2d_inverse(sgx_fft2d *p, const void *spectrum, size_t batch, void *images, int32_t mem_kind, void *stream) {
    sgx_status st = check(p, spectrum, images, batch);
    if (st != SGX_OK) return st;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t inb = batch * p->nrows * p->cb * 2 * p->elem, outb = batch * p->nrows * p->ncols * p->elem;
    return with_staging(p, spectrum, inb, images, outb, mem_kind, s,
                        [&](const void *i, void *o) { return inverse_dev(p, i, batch, o, s); });
}

sgx_status sgx_fft2d_convolve(sgx_fft2d *p, const void *images, size_t batch, const void *kernel_host, size_t krows,
                              size_t kcols, void *out, int32_t mem_kind, void *stream) {
    sgx_status st = check(p, images, out, batch);
    if (st != SGX_OK) return st;
    if (!kernel_host) return fail(p, SGX_INVALID_INPUT, "Invalid input: null buffer");
    if (krows > p->nrows || kcols > p->ncols)  // image_ops.rs:87-91
        return fail(p, SGX_INVALID_INPUT, "Invalid input: kernel dimensions must not exceed image dimensions");
    if (krows == 0 || kcols == 0) return fail(p, SGX_INVALID_INPUT, "Invalid input: kernel dimensions must be > 0");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t R = p->nrows, C = p->ncols, Cb = p->cb;
    DeviceGuard dg;
    F2_HIP(p, dg.enter(p->device));
    const size_t kbytes = krows * kcols * p->elem;
    // (SGX_CONV_RANK1=0: every kernel through its full 2-D spectrum — the A/B and parity switch of tests/test_fft2d.py)
    const char *r1 = std::getenv("SGX_CONV_RANK1");
    const bool outer_allowed = !(r1 && r1[0] == '0');
    const char *r2 = std::getenv("SGX_CONV_SEPARABLE");  // (=0: a rank-1 kernel through the three passes with the outer-product multiplier)
    const bool separable_allowed = !(r2 && r2[0] == '0');
    const bool same_kernel = p->kspec_valid && p->kspec_rows == krows && p->kspec_cols == kcols && p->kspec_stream == s &&
                             p->kspec_outer_allowed == outer_allowed && p->kspec_separable_allowed == separable_allowed && p->kspec_of.size() == kbytes &&
                             std::memcmp(p->kspec_of.data(), kernel_host, kbytes) == 0;
    std::vector<float> uv;
    if (!same_kernel && outer_allowed && use_fused(p) && p->elem == 4 &&
        outer_product_spectrum(static_cast<const float *>(kernel_host), krows, kcols, R, C, Cb, uv)) {
        p->kspec_valid = false;
        if ((st = grow2(p, &p->d_kouter, &p->kouter_bytes, uv.size() * sizeof(float))) != SGX_OK) return st;
        F2_HIP(p, hipMemcpyAsync(p->d_kouter, uv.data(), uv.size() * sizeof(float), hipMemcpyHostToDevice, s));
        F2_HIP(p, hipStreamSynchronize(s));  // `uv` goes out of scope
        p->kspec_outer = true;
        p->kspec_separable = separable_allowed && R == 1024 && C == 1024;
    } else if (!same_kernel) {
        p->kspec_valid = false;
        p->kspec_outer = false;
        // pad_kernel_for_fft (image_ops.rs:123-152): kernel centre -> (0,0), wrapped
        std::vector<unsigned char> padded(R * C * p->elem, 0);
        const long cr = long(krows / 2), cc = long(kcols / 2);
        for (size_t i = 0; i < krows; ++i)
            for (size_t j = 0; j < kcols; ++j) {
                const long tr = ((long(i) - cr) % long(R) + long(R)) % long(R), tc = ((long(j) - cc) % long(C) + long(C)) % long(C);
                std::memcpy(&padded[(size_t(tr) * C + size_t(tc)) * p->elem], (const unsigned char *)kernel_host + (i * kcols + j) * p->elem, p->elem);
            }
        if ((st = grow2(p, &p->d_kimg, &p->kimg_bytes, padded.size())) != SGX_OK) return st;
        if ((st = grow2(p, &p->d_kspec, &p->kspec_bytes, R * Cb * 2 * p->elem)) != SGX_OK) return st;
        F2_HIP(p, hipMemcpyAsync(p->d_kimg, padded.data(), padded.size(), hipMemcpyHostToDevice, s));
        F2_HIP(p, hipStreamSynchronize(s));  // `padded` goes out of scope
        if ((st = forward_dev(p, p->d_kimg, 1, p->d_kspec, s)) != SGX_OK) return st;
    }
    if (!same_kernel) {
        p->kspec_outer_allowed = outer_allowed;
        p->kspec_separable_allowed = separable_allowed;
        p->kspec_of.assign((const unsigned char *)kernel_host, (const unsigned char *)kernel_host + kbytes);
        p->kspec_rows = krows;
        p->kspec_cols = kcols;
        p->kspec_stream = s;
        p->kspec_valid = true;
    }
    const size_t imgb = batch * R * C * p->elem;
    return with_staging(p, images, imgb, out, imgb, mem_kind, s, [&](const void *i, void *o) -> sgx_status {
        if (use_fused(p) && p->kspec_outer) return fused_product_dev(p, i, batch, p->d_kouter, p->kspec_separable ? MUL_VEC : MUL_OUTER, o, s);
        if (use_fused(p)) return fused_product_dev(p, i, batch, p->d_kspec, MUL_SPECTRUM, o, s);
        sgx_status s2 = grow2(p, &p->d_spec, &p->spec_bytes, batch * R * Cb * 2 * p->elem);
        if (s2 != SGX_OK) return s2;
        if ((s2 = forward_dev(p, i, batch, p->d_spec, s, p->d_kspec, 0)) != SGX_OK) return s2;
        return inverse_dev(p, p->d_spec, batch, o, s);
    });
}

sgx_status sgx_fft2d_filter(sgx_fft2d *p, const void *images, size_t batch, int32_t kind, double cut_lo, double cut_hi,
                            void *out, int32_t mem_kind, void *stream) {
    sgx_status st = check(p, images, out, batch);
    if (st != SGX_OK) return st;
    auto in01 = [](double v) { return v >= 0.0 && v <= 1.0; };
    if (kind == 0 || kind == 1) {
        if (!in01(cut_lo)) return fail(p, SGX_INVALID_INPUT, "Invalid input: cutoff_fraction must be between 0.0 and 1.0");
    } else if (kind == 2) {
        if (!in01(cut_lo) || !in01(cut_hi)) return fail(p, SGX_INVALID_INPUT, "Invalid input: cutoff fractions must be between 0.0 and 1.0");
        if (cut_lo >= cut_hi) return fail(p, SGX_INVALID_INPUT, "Invalid input: high_cutoff must be greater than low_cutoff");
    } else {
        return fail(p, SGX_INVALID_INPUT, "Invalid input: unknown filter kind");
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t R = p->nrows, C = p->ncols, Cb = p->cb;
    DeviceGuard dg;
    F2_HIP(p, dg.enter(p->device));
    if (!(p->mask_valid && p->mask_kind == kind && p->mask_lo == cut_lo && p->mask_hi == cut_hi && p->mask_stream == s)) {
        p->mask_valid = false;
        std::vector<double> m, m2;
        lowpass_mask(R, Cb, cut_lo, m);  // spectrum.dim() = (nrows, ncols/2+1): S14
        if (kind == 1) for (double &v : m) v = 1.0 - v;
        if (kind == 2) {
            lowpass_mask(R, Cb, cut_hi, m2);
            for (size_t i = 0; i < m.size(); ++i) m[i] = m2[i] - m[i];
        }
        std::vector<unsigned char> mt(m.size() * p->elem);
        for (size_t i = 0; i < m.size(); ++i) {
            if (p->dtype == SGX_F64) ((double *)mt.data())[i] = m[i]; else ((float *)mt.data())[i] = float(m[i]);
        }
        if ((st = grow2(p, &p->d_mask, &p->mask_bytes, mt.size())) != SGX_OK) return st;
        F2_HIP(p, hipMemcpyAsync(p->d_mask, mt.data(), mt.size(), hipMemcpyHostToDevice, s));
        F2_HIP(p, hipStreamSynchronize(s));
        p->mask_kind = kind;
        p->mask_lo = cut_lo;
        p->mask_hi = cut_hi;
        p->mask_stream = s;
        p->mask_valid = true;
    }
    const size_t imgb = batch * R * C * p->elem;
    return with_staging(p, images, imgb, out, imgb, mem_kind, s, [&](const void *i, void *o) -> sgx_status {
        if (use_fused(p)) return fused_product_dev(p, i, batch, p->d_mask, MUL_MASK, o, s);
        sgx_status s2 = grow2(p, &p->d_spec, &p->spec_bytes, batch * R * Cb * 2 * p->elem);
        if (s2 != SGX_OK) return s2;
        if ((s2 = forward_dev(p, i, batch, p->d_spec, s, p->d_mask, 1)) != SGX_OK) return s2;
        return inverse_dev(p, p->d_spec, batch, o, s);
    });
}

}  // extern "C"

// ---- 1-D complex-to-complex plan: C2cPlan<T> (src/fft_backend.rs:113-137), the third plan type `Sample` names --------------
// In place, unnormalised in both directions, host pointers, one sequence of n per call: the column kernels of the 2-D path with
// one sequence.  Off the batched hot path (the reference's callers are single transforms); everything is allocated at creation.
struct sgx_c2c {
    size_t n = 0;
    int dtype = SGX_F32, device = -1;
    size_t elem = 4;
    unsigned log2n = 0, tile = 0;
    void *d_tw = nullptr, *d_buf = nullptr, *d_out = nullptr;
    BsDevTables bs;  // chirp-z tables (lengths without a pass split that are not powers of two)
    BigDev big;      // lengths past every on-chip kernel: the global-memory transforms (bigfft.hip) and their scratch
    void *d_big = nullptr;
    mutable std::string err;
};

namespace {
sgx_status fail1(const sgx_c2c *p, sgx_status st, const std::string &m) {
    if (p) p->err = m; else g_err2d = m;
    return st;
}
sgx_status c2c_run(sgx_c2c *p, void *buf, size_t len, int inverse) {
    if (!p) return SGX_INVALID_INPUT;
    if (!buf) return fail1(p, SGX_INVALID_INPUT, "Invalid input: null buffer");
    if (len != p->n)  // dimension_mismatch(n_fft, buf.len())
        return fail1(p, SGX_DIM_MISMATCH, "Dimension mismatch: expected " + std::to_string(p->n) + ", got " + std::to_string(len));
    if (p->device < 0) return fail1(p, SGX_BACKEND, "hip -- FFT backend error: plan has no HIP device (host-only plan)");
    DeviceGuard dg;
    if (dg.enter(p->device) != hipSuccess) return fail1(p, SGX_BACKEND, "hip -- FFT backend error: hipSetDevice failed");
    const size_t bytes = 2 * p->n * p->elem;
    if (hipMemcpy(p->d_buf, buf, bytes, hipMemcpyHostToDevice) != hipSuccess) return fail1(p, SGX_BACKEND, "hip -- FFT backend error: copy in");
    C2cArgs a{};
    a.in = p->d_buf; a.out = p->d_out;
    a.n = unsigned(p->n); a.log2n = p->log2n; a.nseq = 1; a.batch = 1;
    a.in_img = p->n; a.out_img = p->n;
    a.in_ss = p->n; a.in_is = 1; a.out_ss = p->n; a.out_is = 1;
    a.tile = p->tile; a.tiles = 1;
    a.tw = p->d_tw; a.inverse = inverse; a.in_seq_fast = 0; a.out_seq_fast = 0; a.scale = 1.0;
    hipError_t e = p->big.M ? launch_big_c2c(p->big, a, p->d_big, p->dtype, nullptr) : launch_c2c_reg(a, p->dtype, nullptr);  // (c2c_reg picks its own tile)
    if (e == hipErrorNotSupported && p->bs.M) e = launch_c2c_bluestein(a, p->bs, p->dtype, nullptr);
    if (e == hipErrorNotSupported) {
        if (p->tile == 0) return fail1(p, SGX_BACKEND, "hip -- FFT backend error: length too large for the on-chip tile");
        e = launch_c2c_tile(a, p->dtype, nullptr);
    }
    if (e != hipSuccess) return fail1(p, SGX_BACKEND, std::string("hip -- FFT backend error: ") + hipGetErrorString(e));
    if (hipMemcpy(buf, p->d_out, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail1(p, SGX_BACKEND, "hip -- FFT backend error: copy out");
    return SGX_OK;
}
}  // namespace

extern "C" {
sgx_status sgx_c2c_create(size_t n, int32_t dtype, int32_t device, sgx_c2c **out) {
    if (out) *out = nullptr;
    if (!out || n == 0) return fail1(nullptr, SGX_INVALID_INPUT, "Invalid input: n must be > 0");
    if (dtype != SGX_F32 && dtype != SGX_F64) return fail1(nullptr, SGX_INVALID_INPUT, "Invalid input: dtype must be f32 or f64");
    if (n > 0x7fffffffull) return fail1(nullptr, SGX_INVALID_INPUT, "Invalid input: n too large");
    sgx_c2c *p = new (std::nothrow) sgx_c2c();
    if (!p) return fail1(nullptr, SGX_INTERNAL, "Internal error: out of memory");
    p->n = n; p->dtype = dtype; p->elem = dtype == SGX_F64 ? 8 : 4; p->device = device; p->log2n = ilog2_pow2(n);
    if (device == -2) { *out = p; return SGX_OK; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { delete p; return fail1(nullptr, SGX_BACKEND, "hip -- FFT backend error: no HIP device available"); }
    int dev = device;
    if (dev == -1 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev < 0 || dev >= ndev) { delete p; return fail1(nullptr, SGX_INVALID_INPUT, "Invalid input: device ordinal out of range"); }
    p->device = dev;
    p->tile = fft2d_tile_for(unsigned(n), dtype);
    DeviceGuard dg;
    std::vector<double> tw(2 * n);
    for (size_t k = 0; k < n; ++k) {
        const double a = -2.0 * kPi2 * double(k) / double(n);
        tw[2 * k] = std::cos(a);
        tw[2 * k + 1] = std::sin(a);
    }
    bool ok = dg.enter(dev) == hipSuccess && hipMalloc(&p->d_tw, 2 * n * p->elem) == hipSuccess &&
              hipMalloc(&p->d_buf, 2 * n * p->elem) == hipSuccess && hipMalloc(&p->d_out, 2 * n * p->elem) == hipSuccess;
    if (ok) {
        if (dtype == SGX_F64) {
            ok = hipMemcpy(p->d_tw, tw.data(), 2 * n * 8, hipMemcpyHostToDevice) == hipSuccess;
        } else {
            std::vector<float> t32(tw.begin(), tw.end());
            ok = hipMemcpy(p->d_tw, t32.data(), 2 * n * 4, hipMemcpyHostToDevice) == hipSuccess;
        }
    }
    if (ok) {
        unsigned fa, fb, fc;
        BsHostTables h;
        if (wants_bluestein(n, reg_split_len(unsigned(n), dtype, &fa, &fb, &fc), dtype, h)) {
            auto up = [&](void **dst, const std::vector<double> &v) {
                if (hipMalloc(dst, v.size() * p->elem) != hipSuccess) return false;
                if (dtype == SGX_F64) return hipMemcpy(*dst, v.data(), v.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
                std::vector<float> c(v.begin(), v.end());
                return hipMemcpy(*dst, c.data(), c.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
            };
            ok = up(&p->bs.chirp, h.chirp) && up(&p->bs.bhp, h.bhp) && up(&p->bs.tw, h.tw);
            p->bs.M = h.M;
        }
    }
    if (ok) {
        // no register-tiled split, no chirp-z in LDS, and more than 2048 points (the LDS-tile kernel would run a two-factor or direct
        // sum, or has no tile at all): through global memory, O(n log n) at every length (src/fft_backend.rs:372-389 plans any length)
        unsigned fa, fb, fc;
        BigHost bh;
        if (!p->bs.M && n > 2048 && !reg_split_len(unsigned(n), dtype, &fa, &fb, &fc) && (p->tile == 0 || p->log2n == 0) && big_host_tables(unsigned(n), bh)) {
            ok = big_upload(bh, dtype, p->big) == hipSuccess && hipMalloc(&p->d_big, big_scratch_bytes(p->big, dtype, 1)) == hipSuccess;
        }
    }
    if (!ok) {
        big_free(p->big);
        for (void *b : {p->d_tw, p->d_buf, p->d_out, p->bs.chirp, p->bs.bhp, p->bs.tw, p->d_big}) if (b) (void)hipFree(b);
        delete p;
        return fail1(nullptr, SGX_BACKEND, "hip -- FFT backend error: could not set up the C2C plan (allocation failed)");
    }
    *out = p;
    return SGX_OK;
}
void sgx_c2c_destroy(sgx_c2c *p) {
    if (!p) return;
    if (p->device >= 0) {
        DeviceGuard dg;
        (void)dg.enter(p->device);
        for (void *b : {p->d_tw, p->d_buf, p->d_out, p->bs.chirp, p->bs.bhp, p->bs.tw, p->d_big}) if (b) (void)hipFree(b);
        big_free(p->big);
    }
    delete p;
}
sgx_status sgx_c2c_forward(sgx_c2c *p, void *buf, size_t len) { return c2c_run(p, buf, len, 0); }
sgx_status sgx_c2c_inverse(sgx_c2c *p, void *buf, size_t len) { return c2c_run(p, buf, len, 1); }
const char *sgx_c2c_last_error(const sgx_c2c *p) { return p ? p->err.c_str() : g_err2d.c_str(); }
}  // extern "C"
