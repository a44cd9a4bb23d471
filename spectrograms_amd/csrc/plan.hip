// plan.hip — host side of libspectro_hip.so: the C ABI of include/spectro_hip.h.
//
// Plan construction mirrors the reference constructors (validation order and error texts follow
// src/spectrogram.rs), builds window / twiddle / filterbank tables on the host in f64, casts to the
// plan's scalar type T and uploads them.  Execution dispatches one batched kernel launch per call.
// There is no CPU compute path in this library: without a HIP device, compute entry points return
// SGX_BACKEND.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <type_traits>

#include "d32x16_layout.h"
#include "r32x16_layout.h"
#include "reg_radix.h"
#include "sgx_internal.h"

// chirp-z vs the direct / two-factor sums: multiply-adds per sample the latter may cost, per log2(M) M / n.  Measured crossovers
// (64 x 10 s, hop n / 4, profiles/bench_r03_chirpz_crossover.txt): per-bin outputs win from n_fft 17-18 on (0.25); filterbank outputs
// pay the split path's second launch and win from n_fft ~46 in f32 (0.65), from 18 in f64 (0.25), where the direct kernels' own
// filterbank stage is the slow part.
#ifndef SGX_BS_COST
#define SGX_BS_COST 0.25
#endif
#ifndef SGX_BS_COST_BANK32
#define SGX_BS_COST_BANK32 0.65
#endif

using namespace sgx;

constexpr unsigned kBigMin = 2048;  // frame lengths above this never run an O(n^2) kernel: bigfft.hip takes what has no O(n log n) kernel on chip

#ifndef SGX_BANDPF
#define SGX_BANDPF 0  // must match kernels_r32x16.hip
#endif
#ifndef SGX_ODDHOP
#define SGX_ODDHOP 1  // must match kernels_r32x16.hip
#endif
namespace {

thread_local std::string g_create_err;

constexpr double kPi = 3.14159265358979323846264338327950288;

sgx_status set_err(const sgx_plan *p, sgx_status st, const std::string &msg) {
    if (p) p->err = msg;
    return st;
}

sgx_status dim_err(const sgx_plan *p, size_t expected, size_t got) {  // DimensionMismatch{expected, got} (src/error.rs:19-21)
    if (p) { p->dm_expected = expected; p->dm_got = got; }
    return set_err(p, SGX_DIM_MISMATCH, "Dimension mismatch: expected " + std::to_string(expected) + ", got " + std::to_string(got));
}

sgx_status create_fail(sgx_status st, const std::string &msg) {
    g_create_err = msg;
    return st;
}

#define SGX_HIP(plan, call)                                                                        \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return set_err(plan, SGX_BACKEND, std::string("hip -- FFT backend error: ") + #call +  \
                                                  ": " + hipGetErrorString(e_));                   \
    } while (0)

// ---- window coefficients (make_window, src/spectrogram.rs:2159-2235), f64 ------------------------
double i0_polynomial(double x) {  // modified_bessel_i0 :2237-2259 — the reference's A&S polynomial, incl. its
    const double ax = std::fabs(x);  // extra 1/sqrt(2 pi) in the large-argument branch (kept for parity)
    if (ax <= 3.75) {
        const double t = x / 3.75, t2 = t * t;
        static const double c[] = {3.5156229, 3.0899424, 1.2067492, 0.2659732, 0.0360768, 0.0045813};
        double acc = c[5];
        for (int i = 4; i >= 0; --i) acc = c[i] + t2 * acc;
        return 1.0 + t2 * acc;
    }
    const double t = 3.75 / ax;
    static const double d[] = {0.39894228, 0.01328592, 0.00225319, -0.00157565, 0.00916281,
                               -0.02057706, 0.02635537, -0.01647633, 0.00392377};
    double acc = d[8];
    for (int i = 7; i >= 0; --i) acc = d[i] + t * acc;
    return (std::exp(ax) / (std::sqrt(ax) * std::sqrt(2.0 * kPi))) * acc;
}

void build_window(const sgx_params &p, const std::vector<double> &custom, std::vector<double> &w) {
    const size_t n = p.n_fft;
    w.assign(n, 0.0);
    const double n1 = double(n - 1);
    switch (p.window_kind) {
    case SGX_WIN_RECTANGULAR:
        std::fill(w.begin(), w.end(), 1.0);
        break;
    case SGX_WIN_HANNING:
        for (size_t i = 0; i < n; ++i) w[i] = std::fma(0.5, -std::cos(2.0 * kPi * double(i) / n1), 0.5);
        break;
    case SGX_WIN_HAMMING:
        for (size_t i = 0; i < n; ++i) w[i] = std::fma(0.46, -std::cos(2.0 * kPi * double(i) / n1), 0.54);
        break;
    case SGX_WIN_BLACKMAN:
        for (size_t i = 0; i < n; ++i) {
            const double a = 2.0 * kPi * double(i) / n1;
            w[i] = std::fma(0.08, std::cos(2.0 * a), std::fma(0.5, -std::cos(a), 0.42));
        }
        break;
    case SGX_WIN_KAISER: {
        if (n == 1) { w[0] = 1.0; break; }
        const double beta = p.window_param, denom = i0_polynomial(beta), half = n1 / 2.0;
        for (size_t i = 0; i < n; ++i) {
            const double u = (double(i) - half) / half;
            const double ratio = std::max(1.0 - u * u, 0.0);
            w[i] = denom == 0.0 ? 0.0 : i0_polynomial(beta * std::sqrt(ratio)) / denom;
        }
        break;
    }
    case SGX_WIN_GAUSSIAN: {
        const double centre = n1 / 2.0;
        for (size_t i = 0; i < n; ++i) {
            const double q = (double(i) - centre) / p.window_param;
            w[i] = std::exp(-0.5 * (q * q));
        }
        break;
    }
    case SGX_WIN_CUSTOM:
        w = custom;
        break;
    }
}

// ---- Slaney mel scale + Hz-space triangular bank (src/spectrogram.rs:2268-2432) as CSR -------------
constexpr double kFsp = 200.0 / 3.0, kMinLogHz = 1000.0, kMinLogMel = kMinLogHz / kFsp;
constexpr double kLogStep = 0.06875177742094923;  // ln(6.4)/27
double hz2mel(double hz) { return hz >= kMinLogHz ? kMinLogMel + std::log(hz / kMinLogHz) / kLogStep : hz / kFsp; }
double mel2hz(double mel) {
    return mel >= kMinLogMel ? kMinLogHz * std::exp(kLogStep * (mel - kMinLogMel)) : std::fma(kFsp, mel, 0.0);
}

void build_mel_csr(const sgx_params &p, std::vector<uint32_t> &ptr, std::vector<uint32_t> &col,
                   std::vector<double> &val) {
    const size_t n_mels = p.n_mels, nb = p.n_fft / 2 + 1;
    const double df = p.sample_rate_hz / double(p.n_fft);
    const double mel_lo = hz2mel(p.f_min), mel_hi = hz2mel(p.f_max);
    const double step = (mel_hi - mel_lo) / double(n_mels + 1);
    std::vector<double> mel_pts(n_mels + 2), edge(n_mels + 2);
    for (size_t i = 0; i < n_mels + 2; ++i) {
        mel_pts[i] = std::fma(double(i), step, mel_lo);
        edge[i] = mel2hz(mel_pts[i]);
    }
    ptr.assign(n_mels + 1, 0);
    col.clear();
    val.clear();
    for (size_t m = 0; m < n_mels; ++m) {
        ptr[m] = uint32_t(col.size());
        const double lo = edge[m], mid = edge[m + 1], hi = edge[m + 2];
        const double rise = mid - lo, fall = hi - mid;
        if (rise == 0.0 || fall == 0.0) continue;  // degenerate triangle (:2360-2363)
        for (size_t k = 0; k < nb; ++k) {
            const double f = double(k) * df;
            double wgt = std::min((f - lo) / rise, (hi - f) / fall);
            wgt = std::min(std::max(wgt, 0.0), 1.0);
            if (wgt > 0.0 && std::fabs(wgt) > 1e-10) {  // SparseMatrix::set drops |v| <= 1e-10 (:83)
                col.push_back(uint32_t(k));
                val.push_back(wgt);
            }
        }
    }
    ptr[n_mels] = uint32_t(col.size());
    for (size_t m = 0; m < n_mels; ++m) {
        double *v = val.data() + ptr[m];
        const size_t cnt = ptr[m + 1] - ptr[m];
        double scale = 1.0;
        bool apply = false;
        if (p.mel_norm == SGX_MELNORM_SLANEY) {
            scale = 2.0 / (mel2hz(mel_pts[m + 2]) - mel2hz(mel_pts[m]));
            apply = true;
        } else if (p.mel_norm == SGX_MELNORM_L1) {
            double s = 0.0;
            for (size_t i = 0; i < cnt; ++i) s += v[i];
            if (s > 0.0) { scale = 1.0 / s; apply = true; }
        } else if (p.mel_norm == SGX_MELNORM_L2) {
            double s = 0.0;
            for (size_t i = 0; i < cnt; ++i) s += v[i] * v[i];
            s = std::sqrt(s);
            if (s > 0.0) { scale = 1.0 / s; apply = true; }
        }
        if (apply)
            for (size_t i = 0; i < cnt; ++i) v[i] *= scale;
    }
}

// ---- log-frequency interpolation matrix (build_loghz_matrix, src/spectrogram.rs:2438-2508) as CSR -----------------
size_t sat_usize(double v) {  // Rust `as usize`: NaN -> 0, negative -> 0, saturating
    if (!(v == v) || v <= 0.0) return 0;
    if (v >= 1.8446744073709552e19) return ~size_t(0);
    return size_t(v);
}

void build_loghz_csr(const sgx_params &p, std::vector<uint32_t> &ptr, std::vector<uint32_t> &col, std::vector<double> &val,
                     std::vector<double> &freqs) {
    const size_t nb = p.n_mels, out_len = p.n_fft / 2 + 1;
    const double df = p.sample_rate_hz / double(p.n_fft);
    const double lo = std::log(p.f_min), hi = std::log(p.f_max);
    const double step = (hi - lo) / double(nb - 1);
    freqs.resize(nb);
    for (size_t i = 0; i < nb; ++i) freqs[i] = std::exp(std::fma(double(i), step, lo));
    ptr.assign(nb + 1, 0);
    col.clear();
    val.clear();
    auto set = [&](size_t c, double v) {  // SparseMatrix::set (:69-87): bounds + |v| > 1e-10
        if (c >= out_len) return;
        if (std::fabs(v) > 1e-10) { col.push_back(uint32_t(c)); val.push_back(v); }
    };
    for (size_t b = 0; b < nb; ++b) {
        ptr[b] = uint32_t(col.size());
        const double exact = freqs[b] / df;
        const size_t lower = sat_usize(std::floor(exact));
        const size_t upper = std::min(sat_usize(std::ceil(exact)), out_len - 1);
        if (lower >= out_len) continue;
        if (lower == upper) {
            set(lower, 1.0);
        } else {
            const double frac = exact - double(lower);
            set(lower, 1.0 - frac);
            if (upper < out_len) set(upper, frac);
        }
    }
    ptr[nb] = uint32_t(col.size());
}

// ---- ERB / gammatone bank (ErbFilterbank::generate, src/erb.rs:266-335): DENSE rows of |H(f)|^2, no threshold --------
void build_erb_dense(const sgx_params &p, std::vector<uint32_t> &ptr, std::vector<uint32_t> &col, std::vector<double> &val,
                     std::vector<double> &centres) {
    const size_t nf = p.n_mels, nb = p.n_fft / 2 + 1;
    centres.resize(nf);
    if (p.erb_spacing == SGX_ERB_APPLE_TR35) {  // apple_tr35_center_freqs :221-238 (computed high->low, stored low->high)
        const double shift = 9.26449 * 24.7;
        const double d = p.f_max + shift;
        const double e = (std::log(p.f_min + shift) - std::log(p.f_max + shift)) / double(nf);
        for (size_t i = 0; i < nf; ++i) centres[nf - 1 - i] = -shift + std::exp((double(i) + 1.0) * e) * d;
    } else {  // uniform on hz_to_erb :208-210, back through erb_to_hz :249-251
        const double lo = 24.7 * (4.37 * p.f_min / 1000.0 + 1.0), hi = 24.7 * (4.37 * p.f_max / 1000.0 + 1.0);
        const double step = (hi - lo) / double(nf - 1);
        for (size_t i = 0; i < nf; ++i) centres[i] = (std::fma(double(i), step, lo) / 24.7 - 1.0) * 1000.0 / 4.37;
    }
    const double df = p.sample_rate_hz / double(p.n_fft);
    ptr.resize(nf + 1);
    col.resize(nf * nb);
    val.resize(nf * nb);
    for (size_t m = 0; m < nf; ++m) {
        ptr[m] = uint32_t(m * nb);
        const double bw = 1.019 * (24.7 * (4.37 * centres[m] / 1000.0 + 1.0));
        for (size_t k = 0; k < nb; ++k) {
            // |1/(1 + j x)^4|^2 through the same complex products as the reference (:306-312)
            const double x = (double(k) * df - centres[m]) / bw;
            const double r2 = 1.0 * 1.0 - x * x, i2 = 1.0 * x + x * 1.0;  // (1 + jx)^2
            const double r4 = r2 * r2 - i2 * i2, i4 = r2 * i2 + i2 * r2;  // ^4
            col[m * nb + k] = uint32_t(k);
            val[m * nb + k] = 1.0 / (r4 * r4 + i4 * i4);
        }
    }
    ptr[nf] = uint32_t(nf * nb);
}

// ---- chroma pitch-class bank (build_chroma_filterbank, src/chroma.rs:262-345): 12 rows, non-zero inside [f_min, f_max] -----
void build_chroma_csr(const sgx_params &p, std::vector<uint32_t> &ptr, std::vector<uint32_t> &col, std::vector<double> &val) {
    const size_t nb = p.n_fft / 2 + 1;
    const double df = p.sample_rate_hz / double(p.n_fft);
    std::vector<double> fb(12 * nb, 0.0);
    for (size_t k = 0; k < nb; ++k) {
        const double freq = double(k) * df;
        if (freq < p.f_min || freq > p.f_max || freq <= 0.0) continue;
        const double midi = 69.0 + 12.0 * std::log(freq / p.chroma_tuning) / 0.6931471805599453;  // LN_2
        double pc = std::fmod(midi, 12.0);  // rem_euclid(12.0)
        if (pc < 0.0) pc += 12.0;
        for (size_t c = 0; c < 12; ++c) {
            const double dist = std::fabs(pc - double(c));
            const double circ = std::min(dist, 12.0 - dist);
            const double q = circ / 1.0;
            fb[c * nb + k] = std::exp(-0.5 * (q * q));
        }
    }
    ptr.assign(13, 0);
    col.clear();
    val.clear();
    for (size_t c = 0; c < 12; ++c) {
        double row_sum = 0.0;
        for (size_t k = 0; k < nb; ++k) row_sum += fb[c * nb + k];
        ptr[c] = uint32_t(col.size());
        for (size_t k = 0; k < nb; ++k) {
            double w = fb[c * nb + k];
            if (row_sum > 0.0) w /= row_sum;
            if (w != 0.0) {  // the reference multiplies every bin; an exactly-zero weight adds +0
                col.push_back(uint32_t(k));
                val.push_back(w);
            }
        }
    }
    ptr[12] = uint32_t(col.size());
}

// ---- validation (same conditions and texts as the reference constructors) ------------------------
sgx_status validate(const sgx_params &p, std::string &msg) {
    auto bad = [&](const char *m) { msg = std::string("Invalid input: ") + m; return SGX_INVALID_INPUT; };
    if (p.n_fft == 0) return bad("n_fft must be > 0");
    if (p.hop_size == 0) return bad("hop_size must be > 0");
    if (p.hop_size > p.n_fft) return bad("hop_size must be <= n_fft");  // spectrogram.rs:3485-3487
    if (p.window_kind < SGX_WIN_RECTANGULAR || p.window_kind > SGX_WIN_CUSTOM) return bad("unknown window type");
    if (p.window_kind == SGX_WIN_CUSTOM && (!p.custom_window || p.custom_window_len != p.n_fft)) {
        msg = "Invalid input: Custom window size (" + std::to_string(p.custom_window ? p.custom_window_len : 0) +
              ") must match n_fft (" + std::to_string(p.n_fft) + ")";  // :3490-3498
        return SGX_INVALID_INPUT;
    }
    if (!(p.sample_rate_hz > 0.0 && std::isfinite(p.sample_rate_hz)))
        return bad("sample_rate_hz must be finite and > 0");  // :4130-4134
    if (p.freq_scale == SGX_FREQ_MEL) {
        if (p.n_mels == 0) return bad("n_mels must be > 0");
        if (p.f_min < 0.0) return bad("f_min must be >= 0");         // :3799-3801
        if (p.f_max <= p.f_min) return bad("f_max must be > f_min");  // :3803-3805
        if (p.f_max > p.sample_rate_hz * 0.5) return bad("mel f_max must be <= Nyquist");  // :954-959
        if (p.n_mels > 10000) return bad("n_mels is unreasonably large");                  // :1696-1700
        if (std::isinf(p.f_min)) return bad("f_min must be >= 0");                         // :2315
    } else if (p.freq_scale == SGX_FREQ_LOGHZ) {
        if (p.n_mels == 0) return bad("n_bins must be > 0");
        if (!(p.f_min > 0.0 && std::isfinite(p.f_min))) return bad("f_min must be finite and > 0");  // :3961-3965
        if (p.f_max <= p.f_min) return bad("f_max must be > f_min");                                  // :3967-3969
        if (p.n_mels > 10000) return bad("n_bins is unreasonably large");                             // :1732-1736
        if (p.f_max > p.sample_rate_hz * 0.5) return bad("f_max must be <= Nyquist");                 // :2458-2460
    } else if (p.freq_scale == SGX_FREQ_ERB) {
        // ErbParams::new src/erb.rs:66-80; erb_plan src/spectrogram.rs:1014-1022; new_erb :1768-1773
        if (p.n_mels < 2) return bad("n_filters must be >= 2 (single filter would cause division by zero)");
        if (p.f_min < 0.0 || !std::isfinite(p.f_min)) return bad("f_min must be finite and >= 0");
        if (p.f_max <= p.f_min) return bad("f_max must be > f_min");
        if (p.f_max > p.sample_rate_hz * 0.5) {
            char b[128];
            snprintf(b, sizeof b, "f_max=%g exceeds Nyquist=%g", p.f_max, p.sample_rate_hz * 0.5);
            return bad(b);
        }
        if (p.n_mels > 10000) return bad("n_filters is unreasonably large");
        if (p.erb_spacing != SGX_ERB_LINEAR && p.erb_spacing != SGX_ERB_APPLE_TR35) return bad("unknown ERB spacing");
    } else if (p.freq_scale == SGX_FREQ_CHROMA) {  // ChromaParams::new src/chroma.rs:64-90
        if (!(p.chroma_tuning > 0.0 && std::isfinite(p.chroma_tuning))) return bad("tuning must be finite and > 0");
        if (!(p.f_min > 0.0 && std::isfinite(p.f_min))) return bad("f_min must be finite and > 0");
        if (p.f_max <= p.f_min) return bad("f_max must be > f_min");
        if (p.chroma_norm < SGX_CHROMA_NORM_NONE || p.chroma_norm > SGX_CHROMA_NORM_MAX) return bad("unknown chroma normalisation");
        if (p.amp_scale != SGX_AMP_MAGNITUDE || p.has_log_params || p.n_mfcc > 0)
            return bad("chromagram is computed from the magnitude spectrogram (amp_scale = magnitude, no LogParams, no MFCC)");
    } else if (p.freq_scale != SGX_FREQ_LINEAR) {
        return bad("unknown frequency scale");
    }
    if (p.amp_scale < SGX_AMP_POWER || p.amp_scale > SGX_AMP_COMPLEX) return bad("unknown amplitude scale");
    if (p.amp_scale == SGX_AMP_COMPLEX && p.freq_scale != SGX_FREQ_LINEAR)
        return bad("complex STFT output requires the linear frequency scale");
    if (p.has_log_params && !std::isfinite(p.floor_db)) return bad("floor_db must be finite");  // :4072-4074
    if (p.dtype != SGX_F32 && p.dtype != SGX_F64) return bad("dtype must be f32 or f64");
    if (p.n_mfcc > 0) {
        if (p.freq_scale != SGX_FREQ_MEL || p.amp_scale != SGX_AMP_DECIBELS)
            return bad("MFCC requires a Mel / Decibels plan");
        if (p.n_mfcc > p.n_mels) return bad("n_mfcc must be <= n_mels");  // src/mfcc.rs:231-233
    }
    return SGX_OK;
}

size_t frame_count(const sgx_params &p, size_t n_samples) {  // StftPlan::frame_count :1230-1250
    const size_t pad = p.centre ? p.n_fft / 2 : 0;
    const size_t padded = n_samples + 2 * pad;
    if (padded < p.n_fft) return 1;
    return (padded - p.n_fft) / p.hop_size + 1;
}

template <typename T>
sgx_status upload(sgx_plan *pl, void **dst, const std::vector<T> &src) {
    if (src.empty()) { *dst = nullptr; return SGX_OK; }
    SGX_HIP(pl, hipMalloc(dst, src.size() * sizeof(T)));
    SGX_HIP(pl, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return SGX_OK;
}

template <typename T>
sgx_status upload_cast(sgx_plan *pl, void **dst, const std::vector<double> &src) {
    std::vector<T> tmp(src.size());
    for (size_t i = 0; i < src.size(); ++i) tmp[i] = T(src[i]);  // T::from_f64
    return upload<T>(pl, dst, tmp);
}

sgx_status grow(sgx_plan *pl, void **buf, size_t *have, size_t need);

// Band schedule of the tuned f32 kernel (n_fft 1024, and 512 in its two-frames-per-transform mode), built on the HOST at plan
// creation — before the kernel kind is resolved, so that a bank without a schedule (rows that are not runs of bins, too many
// words) resolves to the kernel that will really run it, and the split-filterbank decision is made for that kernel.
// A padding step has weight +0: it adds +0 to the running sum as long as |X|^2 of that bin is finite.  A non-finite |X|^2
// (possible only with non-finite or > 1e19 samples, which poison the whole frame's spectrum anyway) within the padded cover
// of a band — up to 11 bins past its last — makes that band NaN where the reference's CSR sum would not look at the bin.
// (NW waves; `ahead` = 1: one more, empty, segment of records behind the last — k_r32x16 fetches a record ahead unconditionally)
void build_band_schedule(sgx_plan *pl, unsigned NW = 4, unsigned kSegs = r32x16::kSchedSegs, unsigned max_words = r32x16::kMelMaxWords, unsigned ahead = 1,
                         unsigned wwords = 1 /* 32-bit words per weight: 1 = f32, 2 = f64 (k_d32x16) */) {
    pl->h_mel_sched.clear();
    pl->mel_sched_words = 0;
    if (pl->out_mode != OUT_MEL || pl->mel_ptr.size() != size_t(pl->p.n_mels) + 1) return;
    for (size_t m = 0; m < pl->p.n_mels; ++m)
        for (uint32_t i = pl->mel_ptr[m]; i + 1 < pl->mel_ptr[m + 1]; ++i)
            if (pl->mel_col[i + 1] != pl->mel_col[i] + 1) return;  // rows must be runs of consecutive bins
    if (wwords == 1 && pl->p.n_fft == 1024 && pl->mel_val.size() >= size_t(48) * pl->p.n_mels) return;  // f32, wide rows: matrix-core epilogue
    {
        const unsigned nm = pl->p.n_mels;
        std::vector<unsigned> order(nm);
        for (unsigned m = 0; m < nm; ++m) order[m] = m;
        auto len = [&](unsigned m) { return pl->mel_ptr[m + 1] - pl->mel_ptr[m]; };
        std::stable_sort(order.begin(), order.end(), [&](unsigned x, unsigned y) { return len(x) > len(y); });
        const unsigned ngroups = (nm + 7) / 8;
        struct Slot { unsigned band, ks, steps; };
        struct Group { unsigned L; Slot s[8]; };
        std::vector<Group> groups(ngroups);
        for (unsigned g = 0; g < ngroups; ++g) {
            Group &G = groups[g];
            G.L = 0;
            for (unsigned q = 0; q < 8; ++q) {
                Slot &S = G.s[q];
                S = Slot{0xffffffffu, q, 0};
                if (8 * g + q >= nm) continue;
                S.band = order[8 * g + q];
                if (len(S.band) == 0) continue;  // degenerate triangle: empty sum, still written
                const unsigned c0 = pl->mel_col[pl->mel_ptr[S.band]], c1 = pl->mel_col[pl->mel_ptr[S.band + 1] - 1];
                if (wwords == 2) {  // k_d32x16: a bin row of the |X|^2 tile is half of the banks; neighbouring slots start on even / odd bins
                    const unsigned back = (c0 + 2u - (q & 1u)) & 1u;
                    S.ks = c0 >= back ? c0 - back : c0;
                } else {
                    const unsigned want = (q & 2u) ? 2u : 0u;    // slots 0, 1 start at 0 mod 4, slots 2, 3 at 2 mod 4 (bank halves)
                    const unsigned back = (c0 + 4u - want) & 3u;  // (c0 - want) mod 4
                    S.ks = c0 >= back ? c0 - back : (c0 & ~1u);   // always even: the kernel reads bin pairs
                }
                S.steps = c1 - S.ks + 1;
#if SGX_BANDPF
                G.L = std::max(G.L, (S.steps + 7u) & ~7u);  // the kernel takes 8 steps per group (software pipeline)
#else
                G.L = std::max(G.L, (S.steps + 3u) & ~3u);  // the kernel takes 4 steps per trip
#endif
            }
        }
        std::vector<std::vector<unsigned>> per_wave(NW);
        std::vector<unsigned> load(NW, 0u);
        for (unsigned g = 0; g < ngroups; ++g) {  // groups are already in descending length order
            unsigned best = 0;
            for (unsigned w = 1; w < NW; ++w) if (load[w] < load[best]) best = w;
            per_wave[best].push_back(g);
            load[best] += groups[g].L + 8;  // + the per-segment epilogue
        }
        unsigned nseg = 0;
        for (auto &v : per_wave) nseg = std::max<unsigned>(nseg, unsigned(v.size()));
        std::vector<uint32_t> words(r32x16::kSchedHdr + (kSegs + ahead) * NW * 8 * 4, 0);
        words[0] = nseg;
        bool ok = nseg <= kSegs;
        for (unsigned seg = 0; ok && seg < kSegs + ahead; ++seg)
            for (unsigned w = 0; w < NW; ++w) {
                const size_t ro = r32x16::kSchedHdr + ((seg * NW + w) * 8) * 4;
                const bool have = seg < per_wave[w].size();
                const Group *G = have ? &groups[per_wave[w][seg]] : nullptr;
                const unsigned L = have ? G->L : 0, lpad = L == 0 ? 0u : ((L / 4) & 1u) ? L : L + 4;  // lpad / 4 odd: the 8 slots' rows start on different banks (an empty segment reads no weights)
                const unsigned woff = unsigned((words.size() + 3) & ~size_t(3));
                words.resize(woff + 8 * size_t(lpad) * wwords, 0);
                for (unsigned q = 0; q < 8; ++q) {
                    Slot S = have ? G->s[q] : Slot{0xffffffffu, q, 0};
                    if (S.band != 0xffffffffu && S.steps > 0) {
                        const unsigned c1 = S.ks + S.steps - 1;
                        const unsigned zlast = pl->nb_fft + 10u;  // the kernel zeroes the 11 rows behind the last bin (513..523; n_fft 512: 257..267)
                        if (c1 + (L - S.steps) > zlast) {  // would read past the zeroed rows: pad in front instead
                            const unsigned d = (c1 + (L - S.steps) - zlast + 3u) / 4u;  // (keeps kstart = slot mod 4)
                            if (S.ks < 4 * d) { ok = false; break; }
                            S.ks -= 4 * d;
                            S.steps += 4 * d;
                        }
                        const uint32_t p0 = pl->mel_ptr[S.band], c0 = pl->mel_col[p0];
                        for (uint32_t i = p0; i < pl->mel_ptr[S.band + 1]; ++i) {
                            const size_t at = woff + (size_t(q) * lpad + (c0 - S.ks) + (i - p0)) * wwords;
                            if (wwords == 2) {
                                const double wv = double(pl->mel_val[i]);
                                std::memcpy(&words[at], &wv, 8);
                            } else {
                                const float wv = float(pl->mel_val[i]);
                                std::memcpy(&words[at], &wv, 4);
                            }
                        }
                    }
                    uint32_t *r = &words[ro + 4 * q];
                    r[0] = L; r[1] = woff + q * lpad * wwords; r[2] = S.ks; r[3] = S.band;
                }
            }
        words.resize(words.size() + 4, 0);
        words[1] = uint32_t(words.size());
        if (ok && words.size() + 16 <= size_t(max_words)) {  // (+16: the kernel's read-ahead stays inside the LDS table)
            pl->mel_sched_words = unsigned(words.size());
            pl->h_mel_sched = std::move(words);
        }
    }
}

template <typename T>
sgx_status build_device_tables(sgx_plan *pl) {
    const unsigned n = pl->p.n_fft;
    sgx_status st;
    if ((st = upload_cast<T>(pl, &pl->d_window, pl->window)) != SGX_OK) return st;
    std::vector<T> tw(2 * size_t(n));
    for (unsigned k = 0; k < n; ++k) {  // tw[k] = exp(-2 pi i k / n), evaluated in f64
        const double a = -2.0 * kPi * double(k) / double(n);
        tw[2 * k] = T(std::cos(a));
        tw[2 * k + 1] = T(std::sin(a));
    }
    if ((st = upload<T>(pl, &pl->d_tw, tw)) != SGX_OK) return st;
    if (pl->out_mode == OUT_MEL) {
        if ((st = upload<uint32_t>(pl, &pl->d_mel_ptr, pl->mel_ptr)) != SGX_OK) return st;
        if ((st = upload<uint32_t>(pl, &pl->d_mel_col, pl->mel_col)) != SGX_OK) return st;
        if ((st = upload_cast<T>(pl, &pl->d_mel_val, pl->mel_val)) != SGX_OK) return st;
        // triangular bands are contiguous column runs; build a 4-wide padded copy (4-element
        // aligned column groups, zero weights outside the true band) so the reduction reads LDS 16 bytes at a time
        bool contig = true;
        for (size_t m = 0; m < pl->p.n_mels && contig; ++m)
            for (uint32_t i = pl->mel_ptr[m]; i + 1 < pl->mel_ptr[m + 1]; ++i)
                contig = contig && (pl->mel_col[i + 1] == pl->mel_col[i] + 1);
        pl->mel_contig = contig ? 1u : 0u;
        if (contig) {
            std::vector<uint32_t> pptr(pl->p.n_mels + 1, 0), pcol(pl->p.n_mels, 0);
            std::vector<T> pw;
            for (size_t m = 0; m < pl->p.n_mels; ++m) {
                pptr[m] = uint32_t(pw.size() / 4);
                const uint32_t a = pl->mel_ptr[m], b = pl->mel_ptr[m + 1];
                if (b == a) continue;
                const uint32_t c0 = pl->mel_col[a], c1 = pl->mel_col[b - 1];
                const uint32_t s0 = c0 & ~3u, s1 = (c1 | 3u) + 1u;  // [s0, s1) multiple-of-4 cover
                pcol[m] = s0;
                for (uint32_t c = s0; c < s1; ++c) pw.push_back(c >= c0 && c <= c1 ? T(pl->mel_val[a + (c - c0)]) : T(0));
            }
            pptr[pl->p.n_mels] = uint32_t(pw.size() / 4);
            pl->mel_pchunks = uint32_t(pw.size() / 4);
            if ((st = upload<uint32_t>(pl, &pl->d_mel_pptr, pptr)) != SGX_OK) return st;
            if ((st = upload<uint32_t>(pl, &pl->d_mel_pcol, pcol)) != SGX_OK) return st;
            if ((st = upload<T>(pl, &pl->d_mel_pw, pw)) != SGX_OK) return st;
        }
        // Matrix-core epilogue of the tuned kernel (f32, n_fft = 1024): banks with wide rows (the dense ERB bank, very coarse
        // Mel banks) are applied as [16 rows x K] x [K x 16 frames] products on v_mfma_f32_16x16x4_f32, K restricted to the
        // block's own bin range.  Measured on MI355X (256 x 10 s): ERB-64 266 us vs 4.3 ms on the CSR loop; Mel-80 (12.5
        // non-zeros per row) 183 us on the matrix cores vs 169 us on the LDS band table — so ordinary Mel banks and log-Hz
        // (<= 2 non-zeros) stay on the scheduled band reduction below.
        const bool want_mm = pl->mel_val.size() >= size_t(48) * pl->p.n_mels;
        if (std::is_same<T, float>::value && pl->p.n_fft == 1024 && want_mm) {
            const unsigned nm = pl->p.n_mels, nblk = (nm + 15) / 16, nb = 513;
            std::vector<uint32_t> blk(size_t(nblk) * 4, 0);
            std::vector<float> frag;
            std::vector<std::pair<unsigned, unsigned>> load;  // (chunk-equivalents, block) for the wave assignment
            for (unsigned bk = 0; bk < nblk; ++bk) {
                unsigned lo = nb, hi = 0;
                for (unsigned m = 16 * bk; m < std::min(nm, 16 * bk + 16); ++m)
                    for (uint32_t i = pl->mel_ptr[m]; i < pl->mel_ptr[m + 1]; ++i) {
                        lo = std::min(lo, pl->mel_col[i]);
                        hi = std::max(hi, pl->mel_col[i] + 1);
                    }
                if (hi <= lo) { lo = 0; hi = 0; }
                lo &= ~3u;
                const unsigned len4 = (hi - lo + 3) / 4, n16 = len4 / 4, n4 = len4 % 4;
                // dense lookup of this block's weights
                std::vector<float> w(size_t(16) * 516, 0.0f);
                for (unsigned m = 16 * bk; m < std::min(nm, 16 * bk + 16); ++m)
                    for (uint32_t i = pl->mel_ptr[m]; i < pl->mel_ptr[m + 1]; ++i)
                        w[size_t(m - 16 * bk) * 516 + pl->mel_col[i]] = float(pl->mel_val[i]);
                blk[4 * bk + 0] = uint32_t(frag.size() / 256);
                blk[4 * bk + 1] = lo;
                blk[4 * bk + 2] = n16;
                blk[4 * bk + 3] = n4;
                for (unsigned c = 0; c < n16; ++c)
                    for (unsigned l = 0; l < 64; ++l)
                        for (unsigned e = 0; e < 4; ++e) frag.push_back(w[size_t(l & 15u) * 516 + lo + 16 * c + 4 * (l >> 4) + e]);
                if (n4)
                    for (unsigned l = 0; l < 64; ++l)
                        for (unsigned e = 0; e < 4; ++e)
                            frag.push_back(e < n4 ? w[size_t(l & 15u) * 516 + lo + 16 * n16 + 4 * e + (l >> 4)] : 0.0f);
                load.push_back({4 * n16 + n4, bk});
            }
            // longest-first assignment of blocks to the 4 waves of a half (the second half rotates the owners by 2 so the two
            // waves sharing a SIMD do not both hold the widest block)
            std::sort(load.begin(), load.end(), [](auto &x, auto &y) { return x.first > y.first; });
            unsigned sum[4] = {0, 0, 0, 0};
            for (auto &lb : load) {
                unsigned best = 0;
                for (unsigned w = 1; w < 4; ++w) if (sum[w] < sum[best]) best = w;
                sum[best] += lb.first;
                blk[4 * lb.second + 3] |= best << 8;
            }
            frag.resize(frag.size() + 4 * 256, 0.0f);  // the kernel's 4-deep fragment ring prefetches unguarded
            pl->mm_nblk = nblk;
            if ((st = upload<float>(pl, &pl->d_mm_frag, frag)) != SGX_OK) return st;
            if ((st = upload<uint32_t>(pl, &pl->d_mm_blk, blk)) != SGX_OK) return st;
        }
        // Band schedule of the tuned kernel: built on the host at plan creation (build_band_schedule), uploaded here
        if (((std::is_same<T, float>::value && (pl->kind == K_R32X16_F32 || pl->kind == K_R32X32_F32 || pl->kind == K_R64X32_F32)) || (std::is_same<T, double>::value && (pl->kind == K_D32X16_F64 || pl->kind == K_D512_F64 || pl->kind == K_D32X32_F64))) &&
            !pl->h_mel_sched.empty() && !pl->d_mm_frag) {
            if ((st = upload<uint32_t>(pl, &pl->d_mel_sched, pl->h_mel_sched)) != SGX_OK) return st;
        }
    }
    if (pl->p.n_mfcc > 0) {
        const unsigned nm = pl->p.n_mels, nc = pl->p.n_mfcc;
        std::vector<double> basis(size_t(nc) * nm), lift(nc, 1.0);
        for (unsigned k = 0; k < nc; ++k)
            for (unsigned i = 0; i < nm; ++i) basis[size_t(k) * nm + i] = std::cos(kPi * double(k) * (double(i) + 0.5) / double(nm));
        if (pl->p.mfcc_lifter > 0)
            for (unsigned i = 0; i < nc; ++i)
                lift[i] = std::fma(double(pl->p.mfcc_lifter) / 2.0, std::sin(kPi * double(i) / double(pl->p.mfcc_lifter)), 1.0);
        if ((st = upload_cast<T>(pl, &pl->d_dct, basis)) != SGX_OK) return st;
        if ((st = upload_cast<T>(pl, &pl->d_lifter, lift)) != SGX_OK) return st;
        // Fused epilogue of the tuned f32 kernel at n_fft 1024 (kernels_r32x16.hip mfcc_tile): the basis as matrix-core A-operands with a
        // lane's steps contiguous, frag[mt][lane][s] = basis[16 mt + (l & 15)][4 s + (l >> 4)] in rows of RL floats (zero beyond n_mfcc /
        // n_mels), then the lifter weights.  Needs the band schedule (Mel-dB tile in LDS), at most 96 bands (the tile sits between the staged
        // samples and the |X|^2 tile) and 64 coefficients (one wave per 16).
        if (std::is_same<T, float>::value && pl->kind == K_R32X16_F32 && pl->p.n_fft == 1024 && pl->d_mel_sched && pl->amp == AMP_DB && nm <= 96 && nc <= 64) {
            const unsigned need = (nm + 3) / 4, mtiles = (nc + 15) / 16;
            const unsigned steps = need <= 12 ? 12 : need <= 16 ? 16 : need <= 20 ? 20 : 24;  // the kernel's menu of chain lengths (48 / 64 / 80 / 96 bands)
            const unsigned RL = (steps / 4) % 2 ? steps : steps + 4;                          // = mfcc_row<STEPS>()
            std::vector<float> frag(size_t(mtiles) * 64 * RL + ((nc + 3) & ~3u), 0.0f);
            for (unsigned mt = 0; mt < mtiles; ++mt)
                for (unsigned l = 0; l < 64; ++l)
                    for (unsigned s2 = 0; s2 < steps; ++s2) {
                        const unsigned c = 16 * mt + (l & 15), band = 4 * s2 + (l >> 4);
                        if (c < nc && band < nm) frag[(size_t(mt) * 64 + l) * RL + s2] = float(basis[size_t(c) * nm + band]);
                    }
            for (unsigned c = 0; c < ((nc + 3) & ~3u); ++c) frag[size_t(mtiles) * 64 * RL + c] = c < nc ? float(lift[c]) : 1.0f;
            const size_t lds = size_t(r32x16::kLdsBytes) - size_t(r32x16::kMelMaxWords) * 4 + ((pl->mel_sched_words + 3u) & ~3u) * 4 + frag.size() * 4 + 64;
            if (lds <= 163840) {
                if ((st = upload<float>(pl, &pl->d_mfcc_frag, frag)) != SGX_OK) return st;
                pl->mfcc_frag_words = unsigned(frag.size());
                pl->mfcc_steps = steps;
                pl->mfcc_mtiles = mtiles;
            }
        }
    }
    if (pl->kind == K_R32X16_F32) {
        // tw1[k1][n2] = W_512^(k1*n2) (pass-1 twiddles), tw2[j][k2] = W_1024^(j + 32*k2) (real-split twiddles)
        // tw1[k1][n2] = W_512^(k1*n2); tw2[row][idx] = (wr, wi, wi, -wr) of W_1024^(row + 32*idx), row stride 17 float4
        std::vector<float> t1(2 * 32 * 16), t2(16 * 17 * 4, 0.0f);
        for (unsigned k1 = 0; k1 < 32; ++k1)
            for (unsigned n2 = 0; n2 < 16; ++n2) {
                const double a = -2.0 * kPi * double(k1 * n2) / 512.0;
                t1[2 * (k1 * 16 + n2)] = float(std::cos(a));
                t1[2 * (k1 * 16 + n2) + 1] = float(std::sin(a));
            }
        // tw2[job][i], i < 16: the pair the job splits i-th is (Z[k], Z[512 - k]) with k = c1 + 32 i (i < 8) or c2 + 32 (i - 8),
        // c1 = j, c2 = j + 256 (job 0: 16 and 0).  With W = W_1024^k = (wr, wi) the kernel needs W' = -i W = (wi, -wr) and
        // W'^perp = (-W'.y, W'.x) = (wr, wi):  T = D.x W' + D.y W'^perp.
        for (unsigned j = 0; j < 16; ++j)
            for (unsigned i = 0; i < 16; ++i) {
                const unsigned c1 = j == 0 ? 16u : j, c2 = j == 0 ? 0u : j + 256u;
                const unsigned k = i < 8 ? c1 + 32 * i : c2 + 32 * (i - 8);
                const double a = -2.0 * kPi * double(k) / 1024.0;
                const float wr = float(std::cos(a)), wi = float(std::sin(a));
                float *q = &t2[4 * (j * 17 + i)];
                q[0] = wi; q[1] = -wr; q[2] = wr; q[3] = wi;
            }
        if ((st = upload<float>(pl, &pl->d_tw1, t1)) != SGX_OK) return st;
        if ((st = upload<float>(pl, &pl->d_tw2, t2)) != SGX_OK) return st;
        std::vector<float> wh(1024), oh(1024, 0.5f);
        if (n == 1024) {
            for (unsigned i = 0; i < n; ++i) wh[i] = 0.5f * float(pl->window[i]);  // exact scaling of the f32 window
        } else {  // n_fft 512, two frames per transform: the pair (w[i], w[i]) multiplies z[i] = a[i] + i b[i]
            for (unsigned i = 0; i < 512; ++i) wh[2 * i] = wh[2 * i + 1] = 0.5f * float(pl->window[i]);
        }
        if ((st = upload<float>(pl, &pl->d_window_half, wh)) != SGX_OK) return st;
        if ((st = upload<float>(pl, &pl->d_ones_half, oh)) != SGX_OK) return st;
    }
    if (pl->kind == K_R32X32_F32) {
        // tw1[k1][n2] = W_1024^(k1 n2), 32 x 32 (pass-1 twiddles); tw2[J][i], i < 16: the pair job J splits i-th is (Z[k], Z[1024 - k]) with
        // k = c1 + 64 i (i < 8) or c2 + 64 (i - 8), c1 = J, c2 = J + 512 (job 0: 0 and 32); entry = (W', W'^perp), W' = -i W_2048^k
        std::vector<float> t1(2 * 32 * 32), t2(32 * 17 * 4, 0.0f);
        for (unsigned k1 = 0; k1 < 32; ++k1)
            for (unsigned n2 = 0; n2 < 32; ++n2) {
                const double a = -2.0 * kPi * double(k1 * n2) / 1024.0;
                t1[2 * (k1 * 32 + n2)] = float(std::cos(a));
                t1[2 * (k1 * 32 + n2) + 1] = float(std::sin(a));
            }
        for (unsigned J = 0; J < 32; ++J)
            for (unsigned i = 0; i < 16; ++i) {
                const unsigned c1 = J == 0 ? 0u : J, c2 = J == 0 ? 32u : J + 512u;
                const unsigned k = i < 8 ? c1 + 64 * i : c2 + 64 * (i - 8);
                const double a = -2.0 * kPi * double(k) / 2048.0;
                const float wr = float(std::cos(a)), wi = float(std::sin(a));
                float *q = &t2[4 * (J * 17 + i)];
                q[0] = wi; q[1] = -wr; q[2] = wr; q[3] = wi;
            }
        if ((st = upload<float>(pl, &pl->d_tw1, t1)) != SGX_OK) return st;
        if ((st = upload<float>(pl, &pl->d_tw2, t2)) != SGX_OK) return st;
        std::vector<float> wh(2048), oh(2048, 0.5f);
        for (unsigned i = 0; i < 2048; ++i) wh[i] = 0.5f * float(pl->window[i]);  // exact scaling of the f32 window
        if ((st = upload<float>(pl, &pl->d_window_half, wh)) != SGX_OK) return st;
        if ((st = upload<float>(pl, &pl->d_ones_half, oh)) != SGX_OK) return st;
    }
    if (pl->kind == K_D32X16_F64) {
        // tw1[k1][n2] = W_512^(k1 n2), 16 x 32 (pass-1 twiddles); tw2[kb][u] = W' = -i W_1024^(kb + 32 u): the pair (Z[k], Z[512 - k]) a lane of
        // kind kb splits u-th (kernels_d32x16.hip)
        std::vector<double> t1(2 * 16 * 32), t2(32 * 8 * 2);
        for (unsigned k1 = 0; k1 < 16; ++k1)
            for (unsigned n2 = 0; n2 < 32; ++n2) {
                const double a = -2.0 * kPi * double(k1 * n2) / 512.0;
                t1[2 * (k1 * 32 + n2)] = std::cos(a);
                t1[2 * (k1 * 32 + n2) + 1] = std::sin(a);
            }
        for (unsigned kb = 0; kb < 32; ++kb)
            for (unsigned u = 0; u < 8; ++u) {
                const double a = -2.0 * kPi * double(kb + 32 * u) / 1024.0;
                t2[2 * (kb * 8 + u)] = std::sin(a);       // W' = -i (wr + i wi) = (wi, -wr)
                t2[2 * (kb * 8 + u) + 1] = -std::cos(a);
            }
        if ((st = upload<double>(pl, &pl->d_tw1, t1)) != SGX_OK) return st;
        if ((st = upload<double>(pl, &pl->d_tw2, t2)) != SGX_OK) return st;
        std::vector<double> wh(1024), oh(1024, 0.5);
        for (unsigned i = 0; i < 1024; ++i) wh[i] = 0.5 * double(pl->window[i]);  // exact scaling
        if ((st = upload<double>(pl, &pl->d_window_half, wh)) != SGX_OK) return st;
        if ((st = upload<double>(pl, &pl->d_ones_half, oh)) != SGX_OK) return st;
    }
    if (pl->kind == K_R64X32_F32) {
        // tw1[k1][n2] = W_2048^(k1 n2), 32 x 64 (pass-1 twiddles); tw2[kb][u] = W' = -i W_4096^(kb + 64 u) (kernels_r64x32.hip); window w / 2
        std::vector<float> t1(2 * 32 * 64), t2(2 * 64 * 16);
        for (unsigned k1 = 0; k1 < 32; ++k1)
            for (unsigned n2 = 0; n2 < 64; ++n2) {
                const double a = -2.0 * kPi * double(k1 * n2) / 2048.0;
                t1[2 * (k1 * 64 + n2)] = float(std::cos(a));
                t1[2 * (k1 * 64 + n2) + 1] = float(std::sin(a));
            }
        for (unsigned kb = 0; kb < 64; ++kb)
            for (unsigned u = 0; u < 16; ++u) {
                const double a = -2.0 * kPi * double(kb + 64 * u) / 4096.0;
                t2[2 * (kb * 16 + u)] = float(std::sin(a));       // W' = -i (wr + i wi) = (wi, -wr)
                t2[2 * (kb * 16 + u) + 1] = float(-std::cos(a));
            }
        if ((st = upload<float>(pl, &pl->d_tw1, t1)) != SGX_OK) return st;
        if ((st = upload<float>(pl, &pl->d_tw2, t2)) != SGX_OK) return st;
        std::vector<float> wh(4096), oh(4096, 0.5f);
        for (unsigned i = 0; i < 4096; ++i) wh[i] = 0.5f * float(pl->window[i]);  // exact scaling of the f32 window
        if ((st = upload<float>(pl, &pl->d_window_half, wh)) != SGX_OK) return st;
        if ((st = upload<float>(pl, &pl->d_ones_half, oh)) != SGX_OK) return st;
    }
    if (pl->kind == K_D32X32_F64) {
        // tw1[k1][n2] = W_1024^(k1 n2), 32 x 32 (pass-1 twiddles); tw2[kb][u] = W' = -i W_2048^(kb + 64 u) (kernels_d32x32.hip); window w / 2
        std::vector<double> t1(2 * 32 * 32), t2(2 * 64 * 8);
        for (unsigned k1 = 0; k1 < 32; ++k1)
            for (unsigned n2 = 0; n2 < 32; ++n2) {
                const double a = -2.0 * kPi * double(k1 * n2) / 1024.0;
                t1[2 * (k1 * 32 + n2)] = std::cos(a);
                t1[2 * (k1 * 32 + n2) + 1] = std::sin(a);
            }
        for (unsigned kb = 0; kb < 64; ++kb)
            for (unsigned u = 0; u < 8; ++u) {
                const double a = -2.0 * kPi * double(kb + 64 * u) / 2048.0;
                t2[2 * (kb * 8 + u)] = std::sin(a);       // W' = -i (wr + i wi) = (wi, -wr)
                t2[2 * (kb * 8 + u) + 1] = -std::cos(a);
            }
        if ((st = upload<double>(pl, &pl->d_tw1, t1)) != SGX_OK) return st;
        if ((st = upload<double>(pl, &pl->d_tw2, t2)) != SGX_OK) return st;
        std::vector<double> wh(2048), oh(2048, 0.5);
        for (unsigned i = 0; i < 2048; ++i) wh[i] = 0.5 * double(pl->window[i]);
        if ((st = upload<double>(pl, &pl->d_window_half, wh)) != SGX_OK) return st;
        if ((st = upload<double>(pl, &pl->d_ones_half, oh)) != SGX_OK) return st;
    }
    if (pl->kind == K_D512_F64) {
        // tw1[k1][n2] = W_512^(k1 n2), 16 x 32 (the 512-point complex transform of a frame pair: kernels_d32x16.hip k_d512); window w[n] / 2
        std::vector<double> t1(2 * 16 * 32);
        for (unsigned k1 = 0; k1 < 16; ++k1)
            for (unsigned n2 = 0; n2 < 32; ++n2) {
                const double a = -2.0 * kPi * double(k1 * n2) / 512.0;
                t1[2 * (k1 * 32 + n2)] = std::cos(a);
                t1[2 * (k1 * 32 + n2) + 1] = std::sin(a);
            }
        if ((st = upload<double>(pl, &pl->d_tw1, t1)) != SGX_OK) return st;
        std::vector<double> wh(512), oh(512, 0.5);
        for (unsigned i = 0; i < 512; ++i) wh[i] = 0.5 * double(pl->window[i]);
        if ((st = upload<double>(pl, &pl->d_window_half, wh)) != SGX_OK) return st;
        if ((st = upload<double>(pl, &pl->d_ones_half, oh)) != SGX_OK) return st;
    }
    if (pl->kind == K_BIGFFT) {  // global-memory transforms (bigfft.hip): stage twiddles, the two-level W_M table, chirp + transformed chirp
        BigHost h;
        if (!big_host_tables(n, h)) return set_err(pl, SGX_INTERNAL, "Internal error: K_BIGFFT plan at an unsupported length");
        SGX_HIP(pl, big_upload(h, pl->dtype, pl->big));
    }
    if (pl->kind == K_BLUESTEIN && pl->bs_fwd_half) {  // half-length complex form: tables of length n / 2 (shared with the inverse rows)
        BsHostTables h;
        if (!bluestein_host_tables(n / 2, pl->dtype, h)) return set_err(pl, SGX_INTERNAL, "Internal error: chirp-z plan without a pass split");
        if ((st = upload_cast<T>(pl, &pl->bs_half.chirp, h.chirp)) != SGX_OK) return st;
        if ((st = upload_cast<T>(pl, &pl->bs_half.bhp, h.bhp)) != SGX_OK) return st;
        if ((st = upload_cast<T>(pl, &pl->bs_half.tw, h.tw)) != SGX_OK) return st;
        pl->bs_half.M = h.M;
    } else if (pl->kind == K_BLUESTEIN) {  // chirp-z tables (bluestein.hip), evaluated in f64
        const unsigned M = pl->bs_M;
        // c_j = e^(+i pi j^2 / n): the angle is reduced in integers, j^2 mod 2 n, so that a large j loses nothing
        auto chirp = [&](unsigned j, double &re, double &im) {
            const unsigned long long q = (unsigned long long)j * j % (2ull * n);
            const double ang = kPi * double(q) / double(n);
            re = std::cos(ang);
            im = std::sin(ang);
        };
        std::vector<double> cc(2 * size_t(n)), bre(M, 0.0), bim(M, 0.0), twm(2 * size_t(M));
        for (unsigned j = 0; j < n; ++j) {
            double re, im;
            chirp(j, re, im);
            cc[2 * j] = re;
            cc[2 * j + 1] = -im;  // conj(c_j): multiplies the samples going in and the bins coming out
            bre[j] = re;
            bim[j] = im;
            if (j) { bre[M - j] = re; bim[M - j] = im; }  // b[-j] = c_j
        }
        // FFT_M(b) on the host: iterative radix-2, f64
        for (unsigned i = 1, j = 0; i < M; ++i) {
            unsigned bit = M >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) { std::swap(bre[i], bre[j]); std::swap(bim[i], bim[j]); }
        }
        for (unsigned len = 2; len <= M; len <<= 1) {
            const double ang = -2.0 * kPi / double(len);
            for (unsigned i = 0; i < M; i += len)
                for (unsigned k = 0; k < len / 2; ++k) {
                    const double wr = std::cos(ang * k), wi = std::sin(ang * k);
                    const double ur = bre[i + k], ui = bim[i + k];
                    const double vr = bre[i + k + len / 2] * wr - bim[i + k + len / 2] * wi;
                    const double vi = bre[i + k + len / 2] * wi + bim[i + k + len / 2] * wr;
                    bre[i + k] = ur + vr; bim[i + k] = ui + vi;
                    bre[i + k + len / 2] = ur - vr; bim[i + k + len / 2] = ui - vi;
                }
        }
        std::vector<double> bh(2 * size_t(M));
        for (unsigned k = 0; k < M; ++k) {
            bh[2 * k] = bre[k] / double(M);  // the inverse transform's 1 / M folded in
            bh[2 * k + 1] = bim[k] / double(M);
            const double a = -2.0 * kPi * double(k) / double(M);
            twm[2 * k] = std::cos(a);
            twm[2 * k + 1] = std::sin(a);
        }
        unsigned fa = 0, fb = 0, fc = 0;
        if (!bluestein_fused_split(M, pl->dtype, &fa, &fb, &fc)) return set_err(pl, SGX_INTERNAL, "Internal error: chirp-z plan without a pass split");
        // window and chirp as one table; FFT_M(b) / M in the order the kernel's product step reads it — [k3][k1][k2] for bin
        // k1 + A (k2 + B k3), [k2][k1] for the two-pass splits
        std::vector<double> wc(2 * size_t(n)), bp(2 * size_t(M));
        for (unsigned j = 0; j < n; ++j) {
            wc[2 * j] = pl->window[j] * cc[2 * j];
            wc[2 * j + 1] = pl->window[j] * cc[2 * j + 1];
        }
        for (unsigned k = 0; k < M; ++k) {
            const unsigned k1 = k % fa, k2 = (k / fa) % fb, k3 = k / (fa * fb);
            const size_t at = fc > 1 ? (size_t(k3) * fa + k1) * fb + k2 : size_t(k2) * fa + k1;
            bp[2 * at] = bh[2 * k];
            bp[2 * at + 1] = bh[2 * k + 1];
        }
        if ((st = upload_cast<T>(pl, &pl->d_bs_chirp, cc)) != SGX_OK) return st;
        if ((st = upload_cast<T>(pl, &pl->d_bs_tw, twm)) != SGX_OK) return st;
        if ((st = upload_cast<T>(pl, &pl->d_bs_wc, wc)) != SGX_OK) return st;
        if ((st = upload_cast<T>(pl, &pl->d_bs_bhp, bp)) != SGX_OK) return st;
    }
    return SGX_OK;
}

void fill_args(const sgx_plan *pl, StftArgs &a, const void *x, void *out, size_t batch, size_t n_samples,
               size_t stride, size_t n_frames) {
    std::memset(&a, 0, sizeof(a));
    const sgx_params &p = pl->p;
    a.x = x;
    a.out = out;
    a.sample_stride = stride;
    a.n_samples = n_samples;
    a.batch = unsigned(batch);
    a.n_fft = p.n_fft;
    a.m = p.n_fft / 2;
    unsigned l = 0;
    while ((1u << l) < a.m) ++l;
    a.log2m = l;
    a.hop = p.hop_size;
    a.pad = p.centre ? p.n_fft / 2 : 0;
    a.n_frames = unsigned(n_frames);
    a.nb_fft = pl->nb_fft;
    a.n_out = pl->n_out;
    a.window = pl->d_window;
    a.tw = pl->d_tw;
    a.tw1 = pl->d_tw1;
    a.tw2 = pl->d_tw2;
    a.mel_ptr = (const unsigned *)pl->d_mel_ptr;
    a.mel_col = (const unsigned *)pl->d_mel_col;
    a.mel_val = pl->d_mel_val;
    a.mel_pptr = (const unsigned *)pl->d_mel_pptr;
    a.mel_pcol = (const unsigned *)pl->d_mel_pcol;
    a.mel_pw = pl->d_mel_pw;
    a.mel_pchunks = pl->mel_pchunks;
    a.mel_contig = pl->mel_contig;
    a.mm_frag = pl->d_mm_frag;
    a.mm_blk = (const uint4 *)pl->d_mm_blk;
    a.mm_nblk = pl->mm_nblk;
    a.mel_sched = (const unsigned *)pl->d_mel_sched;
    a.mel_sched_words = pl->mel_sched_words;
    a.n_mels = p.n_mels;
    a.mel_nnz = unsigned(pl->mel_col.size());
    a.out_mode = pl->out_mode;
    a.amp = pl->amp;
    a.eps = pl->eps;
}

bool set_geometry(const sgx_plan *pl, StftArgs &a, KernelKind kind) {
    bool ok = false;
    switch (kind) {
    case K_R32X16_F32: ok = plan_geometry_r32x16_f32(a); break;
    case K_R32X32_F32: ok = plan_geometry_r32x32_f32(a); break;
    case K_D32X16_F64: ok = plan_geometry_d32x16_f64(a); break;
    case K_D512_F64: ok = plan_geometry_d512_f64(a); break;
    case K_R64X32_F32: ok = plan_geometry_r64x32_f32(a); break;
    case K_D32X32_F64: ok = plan_geometry_d32x32_f64(a); break;
    case K_LDS_RADIX2: ok = plan_geometry_lds_radix2(a, pl->dtype); break;
    case K_DIRECT_DFT: ok = plan_geometry_direct_dft(a, pl->dtype); break;
    case K_TWO_FACTOR: ok = plan_geometry_two_factor(a, pl->dtype); break;
    case K_REG_RADIX: ok = plan_geometry_reg_radix(a, pl->dtype); break;
    case K_BLUESTEIN: a.ft = 1; ok = pl->bs_M != 0 && (a.out_mode != OUT_MEL || a.mel_ptr != nullptr); break;
    case K_BIGFFT: a.ft = 2; ok = pl->big_n != 0 && a.out_mode != OUT_MEL; break;  // two frames per complex sequence; filterbanks: split path
    }
    if (ok) a.tiles = (a.n_frames + a.ft - 1) / a.ft;
    return ok;
}

sgx_status grow(sgx_plan *pl, void **buf, size_t *have, size_t need);

hipError_t launch_bluestein_plan(sgx_plan *pl, const StftArgs &a, hipStream_t s) {
    BsArgs b{};
    const bool half = pl->bs_fwd_half;
    b.x = a.x; b.out = a.out;
    b.sample_stride = a.sample_stride; b.n_samples = a.n_samples;
    b.batch = a.batch; b.n_fft = a.n_fft; b.hop = a.hop; b.pad = a.pad; b.n_frames = a.n_frames; b.nb = a.nb_fft;
    b.M = pl->bs_M;
    // (the per-frame entry point sgx_r2c transforms unwindowed frames: ones x conj(c) is the chirp table itself)
    b.wc = a.window == pl->d_ones ? pl->d_bs_chirp : pl->d_bs_wc;
    b.chirp = pl->d_bs_chirp; b.bhat_fused = pl->d_bs_bhp; b.tw_m = pl->d_bs_tw;
    b.complex_out = a.out_mode == OUT_COMPLEX; b.amp = a.amp; b.eps = a.eps;
    if (a.out_mode == OUT_MEL) {  // (f32, M <= 1024) the bank's rows in the same launch: the |X|^2 of a tile never leave LDS
        b.mel_ptr = a.mel_ptr; b.mel_col = a.mel_col; b.mel_val = a.mel_val; b.n_mels = a.n_mels; b.n_out = a.n_out;
    }
    if (half) return launch_bluestein_half(b, pl->bs_half, a.window, pl->d_tw, pl->dtype, s);
    return launch_bluestein(b, pl->dtype, s);
}

hipError_t launch(sgx_plan *pl, const StftArgs &a, KernelKind kind, hipStream_t s) {
    switch (kind) {
    case K_BIGFFT: {  // sequence scratch: sized by sgx_reserve, else grown here
        const size_t need = big_scratch_bytes(pl->big, pl->dtype, size_t(a.batch) * ((a.n_frames + 1u) / 2u));
        if (grow(pl, &pl->d_big, &pl->d_big_bytes, need) != SGX_OK) return hipErrorOutOfMemory;
        return launch_big_stft(pl->big, a, pl->d_big, pl->dtype, s);
    }
    case K_BLUESTEIN: return launch_bluestein_plan(pl, a, s);
    case K_R32X16_F32: return launch_r32x16_f32(a, s);
    case K_R32X32_F32: return launch_r32x32_f32(a, s);
    case K_D32X16_F64: return launch_d32x16_f64(a, s);
    case K_D512_F64: return launch_d512_f64(a, s);
    case K_R64X32_F32: return launch_r64x32_f32(a, s);
    case K_D32X32_F64: return launch_d32x32_f64(a, s);
    case K_LDS_RADIX2: return launch_lds_radix2(a, pl->dtype, s);
    case K_TWO_FACTOR: return launch_two_factor(a, pl->dtype, s);
    case K_REG_RADIX: return launch_reg_radix(a, pl->dtype, s);
    default: return launch_direct_dft(a, pl->dtype, s);
    }
}

// The tuned kernel reads the samples through bounds-checked buffer loads, which only need the element's own alignment: any
// base address and any row stride run on it (round 1 fell back to the register-tiled kernel for odd strides).
KernelKind pick_kernel(const sgx_plan *pl, const void *, size_t) { return pl->kind; }

sgx_status check_call(sgx_plan *pl, const void *samples, size_t batch, size_t n_samples, size_t stride,
                      void *out, size_t out_elems, size_t *n_frames_out) {
    if (!pl) return SGX_INVALID_INPUT;
    if (!samples || !out) return set_err(pl, SGX_INVALID_INPUT, "Invalid input: null buffer");
    if (batch == 0 || n_samples == 0)
        return set_err(pl, SGX_INVALID_INPUT, "Invalid input: samples must be non-empty");  // NonEmptySlice
    if (stride < n_samples) return set_err(pl, SGX_INVALID_INPUT, "Invalid input: sample_stride < n_samples");
    if (batch > 0xffffffffull) return set_err(pl, SGX_INVALID_INPUT, "Invalid input: batch too large");
    const size_t nf = frame_count(pl->p, n_samples);
    if (nf > 0x7fffffffull) return set_err(pl, SGX_INVALID_INPUT, "Invalid input: too many frames");
    const size_t expect = batch * size_t(pl->n_final) * nf * (pl->out_mode == OUT_COMPLEX ? 2 : 1);
    if (out_elems != expect)  // compute_into: DimensionMismatch{expected, got} (:423-434)
        return dim_err(pl, expect, out_elems);
    if (!pl->device_ready)
        return set_err(pl, SGX_BACKEND, "hip -- FFT backend error: plan has no HIP device (host-only plan)");
    *n_frames_out = nf;
    return SGX_OK;
}

sgx_status grow(sgx_plan *pl, void **buf, size_t *have, size_t need);

#ifndef SGX_SPLIT_BANK_BYTES
// n_fft * sizeof(T) from which the filterbank runs as a second launch (f64 from n_fft 1600, f32 from 4096): measured per 64 x 10 s,
// fused vs split, f64 1024 (8 KiB) 126 vs 129 us, f64 1600 (12.5 KiB) 456 vs 348 us, f64 1920 707 vs 395 us, f64 2048 690 vs 364 us
#define SGX_SPLIT_BANK_BYTES 12000
#endif
// the first launch of the split filterbank path: per-bin power (magnitude for the magnitude-domain banks) instead of the bank's rows
void per_bin_args(const sgx_plan *pl, StftArgs &a) {
    a.amp = a.amp == AMP_MAG_IN ? AMP_MAGNITUDE : AMP_POWER;
    a.out_mode = OUT_LINEAR;
    a.n_out = pl->nb_fft;
}
// geometry for `kind`, else down the chain: tuned -> register-tiled -> LDS radix-2 / two factors -> direct sum
bool resolve_geometry(const sgx_plan *pl, StftArgs &a, KernelKind &kind) {
    if (set_geometry(pl, a, kind)) return true;
    const bool p2 = (a.n_fft & (a.n_fft - 1)) == 0;
    bool ok = false;
    if (kind_is_tuned(kind)) ok = set_geometry(pl, a, kind = K_REG_RADIX);
    if (!ok && p2 && kind != K_LDS_RADIX2) ok = set_geometry(pl, a, kind = K_LDS_RADIX2);
    if (!ok && !p2 && kind == K_REG_RADIX) ok = set_geometry(pl, a, kind = K_TWO_FACTOR);
    if (!ok) ok = set_geometry(pl, a, kind = K_DIRECT_DFT);
    return ok;
}
int chain_pos(KernelKind k) { return kind_is_tuned(k) ? 0 : k == K_REG_RADIX ? 1 : k == K_DIRECT_DFT ? 3 : 2; }  // (K_BLUESTEIN is chosen after the chain, at creation)

sgx_status run_device(sgx_plan *pl, const void *x, size_t batch, size_t n_samples, size_t stride, void *out,
                      size_t n_frames, hipStream_t s, int iters, float *ms) {
    StftArgs a;
    void *stage_out = out;
    bool mfcc = pl->p.n_mfcc > 0;
    KernelKind kind = pick_kernel(pl, x, stride);
    // MFCC: fused into the tuned kernel's launch where the plan carries the basis fragments AND this call stays on that kernel (a call may
    // step down the chain, e.g. signals of fewer frames than half a tile); else the Mel-dB tensor goes to plan-owned scratch
    // (sgx_reserve sizes it ahead) and the DCT / lifter epilogue launch writes the caller's buffer
    bool mfcc_fused = false;
    if (mfcc && pl->d_mfcc_frag && !pl->split_bank) {
        StftArgs probe;
        fill_args(pl, probe, x, out, batch, n_samples, stride, n_frames);
        KernelKind k2 = kind;
        mfcc_fused = resolve_geometry(pl, probe, k2) && k2 == K_R32X16_F32;
    }
    if (mfcc && !mfcc_fused) {
        sgx_status st = grow(pl, &pl->d_melbuf, &pl->d_melbuf_bytes, batch * size_t(pl->n_out) * n_frames * pl->elem);
        if (st != SGX_OK) return st;
        stage_out = pl->d_melbuf;
    }
    fill_args(pl, a, x, stage_out, batch, n_samples, stride, n_frames);
    const unsigned skip = pl->p.n_mfcc - pl->n_final * (mfcc ? 1u : 0u);
    if (mfcc_fused) {
        a.mfcc_frag = pl->d_mfcc_frag; a.mfcc_frag_words = pl->mfcc_frag_words; a.mfcc_steps = pl->mfcc_steps; a.mfcc_mtiles = pl->mfcc_mtiles;
        a.n_mfcc = pl->p.n_mfcc; a.mfcc_skip = skip;
        mfcc = false;  // no epilogue launch
    }
    // Split filterbank path (decided at plan creation, `split_bank`): the per-bin power goes to a plan-owned tensor and a second
    // launch reduces it with one wave per (row, 64 frames) — the same terms in the same order, so the same bits.  Taken (i) for long
    // frames on the register-tiled kernel, where a tile holds one or two frames, the fused bank stage has 80-160 (frame, row) items
    // for 256 threads and its longest rows set the time (64 x 10 s, f64 n_fft 4096 Mel-80 dB: 1.93 ms against 0.32 ms for the
    // per-bin output), and (ii) where the tile with its |X|^2 rows no longer fits the kernel the per-bin output runs on (n_fft 3000
    // Mel fell from the two-factor kernel to the direct sum: 130 ms against 5.3 ms).
    StftArgs a1 = a;  // the first (or only) launch
    const bool split_bank = pl->split_bank;
    if (split_bank) {
        sgx_status st = grow(pl, &pl->d_pwbuf, &pl->d_pwbuf_bytes, batch * size_t(pl->nb_fft) * n_frames * pl->elem);
        if (st != SGX_OK) return st;
        per_bin_args(pl, a1);
        a1.out = pl->d_pwbuf;
    }
    if (!resolve_geometry(pl, a1, kind))
        return set_err(pl, SGX_BACKEND, "hip -- FFT backend error: n_fft too large for the on-chip frame tile");
    if (kind_is_tuned(kind)) a1.window = pl->d_window_half;
    if (ms) SGX_HIP(pl, hipEventRecord(pl->ev0, s));
    for (int i = 0; i < iters; ++i) {
        SGX_HIP(pl, launch(pl, a1, kind, s));
        if (split_bank) SGX_HIP(pl, launch_bank_rows(pl->d_pwbuf, stage_out, a, pl->dtype, s));
        if (pl->p.freq_scale == SGX_FREQ_CHROMA)
            SGX_HIP(pl, launch_chroma_norm(out, unsigned(batch), unsigned(n_frames), pl->p.chroma_norm, pl->dtype, s));
        if (mfcc)
            SGX_HIP(pl, launch_mfcc(pl->d_melbuf, out, pl->d_dct, pl->d_lifter, unsigned(batch), pl->p.n_mels, unsigned(n_frames),
                                    pl->p.n_mfcc, skip, pl->p.mfcc_lifter > 0, pl->dtype, s));
    }
    if (ms) {
        SGX_HIP(pl, hipEventRecord(pl->ev1, s));
        SGX_HIP(pl, hipEventSynchronize(pl->ev1));
        float t = 0.f;
        SGX_HIP(pl, hipEventElapsedTime(&t, pl->ev0, pl->ev1));
        *ms = t / float(iters);
    }
    return SGX_OK;
}

sgx_status grow(sgx_plan *pl, void **buf, size_t *have, size_t need) {
    if (*have >= need) return SGX_OK;
    if (*buf) SGX_HIP(pl, hipFree(*buf));
    *buf = nullptr;
    *have = 0;
    SGX_HIP(pl, hipMalloc(buf, need));
    *have = need;
    return SGX_OK;
}

void free_device(sgx_plan *pl) {
    void **bufs[] = {&pl->d_window, &pl->d_tw, &pl->d_tw1, &pl->d_tw2, &pl->d_mel_ptr, &pl->d_mel_col,
                     &pl->d_mel_val, &pl->d_dct, &pl->d_lifter, &pl->d_mfcc_frag, &pl->d_melbuf, &pl->d_pwbuf, &pl->d_mel_pptr, &pl->d_mel_pcol, &pl->d_mel_pw, &pl->d_mm_frag, &pl->d_mm_blk, &pl->d_mel_sched, &pl->d_itw, &pl->d_itwr, &pl->d_itw1, &pl->d_itwr2, &pl->d_itw12, &pl->d_itwrd, &pl->d_itw1d, &pl->d_frames, &pl->d_flag, &pl->d_ones, &pl->d_in, &pl->d_out, &pl->d_window_half, &pl->d_ones_half, &pl->d_bs_chirp, &pl->d_bs_tw, &pl->d_bs_wc, &pl->d_bs_bhp, &pl->bs_half.chirp, &pl->bs_half.bhp, &pl->bs_half.tw};
    for (void **b : bufs)
        if (*b) { (void)hipFree(*b); *b = nullptr; }
    big_free(pl->big);
    if (pl->d_big) { (void)hipFree(pl->d_big); pl->d_big = nullptr; pl->d_big_bytes = 0; }
    if (pl->ev0) (void)hipEventDestroy(pl->ev0);
    if (pl->ev1) (void)hipEventDestroy(pl->ev1);
    pl->ev0 = pl->ev1 = nullptr;
}

}  // namespace

// ---- inverse path: C2rPlan::process / irfft / istft ------------------------------------------------------------------
namespace {

size_t istft_length(const sgx_params &p, size_t n_frames) {  // spectrogram.rs:4888-4893, 4933-4941
    const size_t pad = p.centre ? p.n_fft / 2 : 0;
    const size_t out_len = (n_frames - 1) * size_t(p.hop_size) + p.n_fft;
    const size_t unpadded = out_len > 2 * pad ? out_len - 2 * pad : 0;
    return (p.centre && unpadded > 0) ? unpadded : out_len;
}

template <typename T>
sgx_status inverse_tables(sgx_plan *pl) {
    if (pl->d_itw || pl->d_flag) return SGX_OK;
    const size_t n = pl->p.n_fft;
    if (pl->kind == K_BIGFFT) {  // the inverse runs on the forward tables (bigfft.hip): only the DC / Nyquist flag word
        SGX_HIP(pl, hipMalloc(&pl->d_flag, sizeof(unsigned)));
        return SGX_OK;
    }
    std::vector<T> tw(2 * n);
    for (size_t k = 0; k < n; ++k) {
        const double a = -2.0 * kPi * double(k) / double(n);
        tw[2 * k] = T(std::cos(a));
        tw[2 * k + 1] = T(std::sin(a));
    }
    sgx_status st = upload<T>(pl, &pl->d_itw, tw);
    if (st != SGX_OK) return st;
    // Even lengths whose rows have neither a register-tiled split (n / 2) nor a chirp-z convolution of their own in LDS (f64 above
    // 4096, f32 above 8192) invert through the chirp-z kernel in half-length complex form: tables of length n / 2.  (Before: the
    // direct sum — f64 n_fft 6000, 64 x 10 s: 1.27 s.)
    {
        unsigned fa, fb, fc;
        BsHostTables h;
        // (the stated condition, checked: the length's own convolution M >= 2 n - 1 has no geometry in LDS.  Without it, short even
        // lengths the forward cost model leaves on the direct / two-factor kernels — n_fft 34 — paid for these tables at every plan
        // creation and inverted through k_bs_c2c<HALF> instead of launch_c2r_rows)
        unsigned long long Mfull = 1;
        while (Mfull < 2ull * n - 1ull) Mfull <<= 1;
        const bool own_fits = Mfull <= 16384ull && bluestein_fused_split(unsigned(Mfull), pl->dtype, &fa, &fb, &fc);
        if (!pl->bs_half.M && !own_fits && n >= 32 && n <= 32768 && n % 2 == 0 && (n & (n - 1)) != 0 && !pl->d_bs_bhp &&
            !reg_split_len(unsigned(n / 2), pl->dtype, &fa, &fb, &fc) && bluestein_host_tables(unsigned(n / 2), pl->dtype, h)) {
            if ((st = upload_cast<T>(pl, &pl->bs_half.chirp, h.chirp)) != SGX_OK) return st;
            if ((st = upload_cast<T>(pl, &pl->bs_half.bhp, h.bhp)) != SGX_OK) return st;
            if ((st = upload_cast<T>(pl, &pl->bs_half.tw, h.tw)) != SGX_OK) return st;
            pl->bs_half.M = h.M;
        }
    }
    // The fused tuned kernel carries the overlap from tile to tile (k_istft1024c: no halo frames), which needs ov = floor(1023 / hop) < 16
    // hop blocks of carry: hop >= 64.  (Rounds 1-3 recomputed the halo frames and left hops below 94 to the register-tiled rows +
    // overlap-add through a frame scratch; with the carry, 256 x 10 s: hop 90 1.77 -> 0.86 ms, hop 80 2.03 -> 0.91, hop 64 2.44 -> 1.08.)
#ifndef SGX_ISTFT1024_MIN_HOP
#define SGX_ISTFT1024_MIN_HOP 64
#endif
    if (std::is_same<T, float>::value && n == 1024 && pl->p.hop_size >= SGX_ISTFT1024_MIN_HOP) {  // tables of the fused tuned kernel
        std::vector<float> tr(2 * 32 * 16), t1(2 * 32 * 16);
        for (unsigned n1 = 0; n1 < 32; ++n1)
            for (unsigned n2 = 0; n2 < 16; ++n2) {
                const double a = 2.0 * kPi * double(16 * n1 + n2) / 1024.0;  // conj(W_1024^k)
                tr[2 * (n1 * 16 + n2)] = float(std::cos(a));
                tr[2 * (n1 * 16 + n2) + 1] = float(std::sin(a));
                const double b2 = -2.0 * kPi * double(n1 * n2) / 512.0;     // W_512^(k1 n2)
                t1[2 * (n1 * 16 + n2)] = float(std::cos(b2));
                t1[2 * (n1 * 16 + n2) + 1] = float(std::sin(b2));
            }
        if ((st = upload<float>(pl, &pl->d_itwr, tr)) != SGX_OK) return st;
        if ((st = upload<float>(pl, &pl->d_itw1, t1)) != SGX_OK) return st;
    }
    if (std::is_same<T, float>::value && n == 2048 && pl->p.hop_size >= 128) {  // tables of the fused tuned n_fft 2048 kernel (ov = 2047 / hop < 16)
        std::vector<float> tr(2 * 1024), t1(2 * 32 * 32);
        for (unsigned k = 0; k < 1024; ++k) {
            const double a = 2.0 * kPi * double(k) / 2048.0;  // conj(W_2048^k)
            tr[2 * k] = float(std::cos(a));
            tr[2 * k + 1] = float(std::sin(a));
        }
        for (unsigned k1 = 0; k1 < 32; ++k1)
            for (unsigned n2 = 0; n2 < 32; ++n2) {
                const double b2 = -2.0 * kPi * double(k1 * n2) / 1024.0;  // W_1024^(k1 n2)
                t1[2 * (k1 * 32 + n2)] = float(std::cos(b2));
                t1[2 * (k1 * 32 + n2) + 1] = float(std::sin(b2));
            }
        if ((st = upload<float>(pl, &pl->d_itwr2, tr)) != SGX_OK) return st;
        if ((st = upload<float>(pl, &pl->d_itw12, t1)) != SGX_OK) return st;
    }
    if (std::is_same<T, double>::value && n == 512 && pl->p.hop_size >= 32) {  // table of the fused f64 n_fft 512 kernel (two frames per transform): W_512^(k1 n2)
        std::vector<double> t1(2 * 16 * 32);
        for (unsigned k1 = 0; k1 < 16; ++k1)
            for (unsigned n2 = 0; n2 < 32; ++n2) {
                const double b2 = -2.0 * kPi * double(k1 * n2) / 512.0;
                t1[2 * (k1 * 32 + n2)] = std::cos(b2);
                t1[2 * (k1 * 32 + n2) + 1] = std::sin(b2);
            }
        if ((st = upload<double>(pl, &pl->d_itw1d, t1)) != SGX_OK) return st;
        pl->istft_d512 = true;
    }
    if (std::is_same<T, double>::value && n == 1024 && pl->p.hop_size >= 64) {  // tables of the fused tuned f64 n_fft 1024 kernel (ov = 1023 / hop < 16)
        std::vector<double> tr(2 * 512), t1(2 * 16 * 32);
        for (unsigned k = 0; k < 512; ++k) {
            const double a = 2.0 * kPi * double(k) / 1024.0;  // conj(W_1024^k)
            tr[2 * k] = std::cos(a);
            tr[2 * k + 1] = std::sin(a);
        }
        for (unsigned k1 = 0; k1 < 16; ++k1)
            for (unsigned n2 = 0; n2 < 32; ++n2) {
                const double b2 = -2.0 * kPi * double(k1 * n2) / 512.0;  // W_512^(k1 n2)
                t1[2 * (k1 * 32 + n2)] = std::cos(b2);
                t1[2 * (k1 * 32 + n2) + 1] = std::sin(b2);
            }
        if ((st = upload<double>(pl, &pl->d_itwrd, tr)) != SGX_OK) return st;
        if ((st = upload<double>(pl, &pl->d_itw1d, t1)) != SGX_OK) return st;
    }
    SGX_HIP(pl, hipMalloc(&pl->d_flag, sizeof(unsigned)));
    return SGX_OK;
}

// frames (rows) of `spec` -> real rows of n_fft samples: inverse real FFT, * 1/n in T (fft_backend.rs:559-563), optional window
sgx_status launch_c2r_frames(sgx_plan *pl, const void *spec, void *frames, size_t batch, size_t n_frames, bool frame_fast,
                             const void *win, hipStream_t s) {
    const unsigned n = pl->p.n_fft;
    C2rArgs c{};
    c.in = spec; c.out = frames;
    c.nrows = unsigned(n_frames); c.ncols = n; c.batch = unsigned(batch);
    c.log2c = 0;
    if (n >= 2 && !(n & (n - 1))) while ((1u << c.log2c) < n) ++c.log2c;
    c.in_img = (unsigned long long)pl->nb_fft * n_frames;
    if (frame_fast) { c.in_ks = n_frames; c.in_rs = 1; c.k_fast = 0; }  // [bin][frame] (StftResult layout, S9)
    else { c.in_ks = 1; c.in_rs = pl->nb_fft; c.k_fast = 1; }
    if (pl->kind == K_BIGFFT) {  // rows through global memory (two frames per complex sequence, the forward engine behind conj)
        c.scale = pl->dtype == SGX_F64 ? 1.0 / double(n) : double(1.0f / float(n));
        c.win = win;
        c.bad_flag = (unsigned *)pl->d_flag;
        sgx_status st = grow(pl, &pl->d_big, &pl->d_big_bytes, big_scratch_bytes(pl->big, pl->dtype, batch * ((n_frames + 1) / 2)));
        if (st != SGX_OK) return st;
        SGX_HIP(pl, launch_big_c2r(pl->big, c, pl->d_big, pl->dtype, s));
        return SGX_OK;
    }
    c.tile = c2r_tile_for(n, pl->dtype, 144 * 1024);
    if (c.tile == 0) return set_err(pl, SGX_BACKEND, "hip -- FFT backend error: n_fft too large for the on-chip frame tile");
    c.tiles = unsigned((n_frames + c.tile - 1) / c.tile);
    c.tw = pl->d_itw;
    c.scale = pl->dtype == SGX_F64 ? 1.0 / double(n) : double(1.0f / float(n));  // T::one() / T::from_usize(n_fft)
    c.win = win;
    c.bad_flag = (unsigned *)pl->d_flag;
    // register-tiled passes; at lengths without a split the chirp-z rows (a plan of such a length carries the tables); else the
    // LDS-tile rows (radix-2 / two-factor / direct sum)
    hipError_t e = launch_c2r_reg(c, pl->dtype, s);
#ifndef SGX_NO_BS_C2C
    if (e == hipErrorNotSupported && pl->d_bs_bhp && (n & (n - 1)) != 0) {
        BsDevTables t;
        t.M = pl->bs_M; t.chirp = pl->d_bs_chirp; t.bhp = pl->d_bs_bhp; t.tw = pl->d_bs_tw;
        e = launch_c2r_bluestein(c, t, pl->dtype, s);
    }
#endif
    if (e == hipErrorNotSupported && pl->bs_half.M) e = launch_c2r_bluestein(c, pl->bs_half, pl->dtype, s, true);
    if (e == hipErrorNotSupported) e = launch_c2r_rows(c, pl->dtype, s);
    SGX_HIP(pl, e);
    return SGX_OK;
}

sgx_status run_istft(sgx_plan *pl, const void *spec, size_t batch, size_t n_frames, void *out, size_t out_len, hipStream_t s) {
    sgx_status st;
    const size_t n = pl->p.n_fft;
    SGX_HIP(pl, hipMemsetAsync(pl->d_flag, 0, sizeof(unsigned), s));
    // fused tuned kernel: no frame scratch in HBM; it addresses one signal's spectrum with 32-bit byte offsets
    if (pl->d_itwr && n_frames * 513ull * 8ull < 0x7fffffffull) {
        const size_t pad0 = pl->p.centre ? n / 2 : 0;
        const size_t full0 = (n_frames - 1) * size_t(pl->p.hop_size) + n;
        SGX_HIP(pl, launch_istft1024(spec, out, pl->d_window, unsigned(n_frames), pl->p.hop_size, unsigned(batch),
                                     out_len == full0 ? 0 : pad0, out_len, 1.0f / 1024.0f, (unsigned *)pl->d_flag, pl->d_itwr,
                                     pl->d_itw1, s));
        return SGX_OK;
    }
    const size_t pad = pl->p.centre ? n / 2 : 0;
    const size_t full = (n_frames - 1) * size_t(pl->p.hop_size) + n;
    const size_t start = out_len == full ? 0 : pad;  // untrimmed when the centred signal would be empty (:4933)
    if (pl->d_itwr2 && n_frames * 1025ull * 8ull < 0x7fffffffull) {  // fused tuned kernel at n_fft 2048 (kernels_istft2048.hip)
        SGX_HIP(pl, launch_istft2048(spec, out, pl->d_window, unsigned(n_frames), pl->p.hop_size, unsigned(batch), start, out_len, 1.0f / 2048.0f,
                                     (unsigned *)pl->d_flag, pl->d_itwr2, pl->d_itw12, s));
        return SGX_OK;
    }
    if (pl->istft_d512 && n_frames * 257ull * 16ull < 0x7fffffffull) {  // fused f64 kernel at n_fft 512 (kernels_istft_d1024.hip: k_istft_d512)
        SGX_HIP(pl, launch_istft_d512(spec, out, pl->d_window, unsigned(n_frames), pl->p.hop_size, unsigned(batch), start, out_len, 1.0 / 512.0,
                                      (unsigned *)pl->d_flag, pl->d_itw1d, s));
        return SGX_OK;
    }
    if (pl->d_itwrd && n_frames * 513ull * 16ull < 0x7fffffffull) {  // fused tuned f64 kernel at n_fft 1024 (kernels_istft_d1024.hip)
        SGX_HIP(pl, launch_istft_d1024(spec, out, pl->d_window, unsigned(n_frames), pl->p.hop_size, unsigned(batch), start, out_len, 1.0 / 1024.0,
                                       (unsigned *)pl->d_flag, pl->d_itwrd, pl->d_itw1d, s));
        return SGX_OK;
    }
    // fused register-tiled kernel (every length with a pass split, hop <= n_fft, at most half a tile of halo frames): the windowed
    // frames never leave the chip
    if (pl->kind != K_BIGFFT && n_frames <= 0xffffffffull && batch <= 0xffffffffull) {
        const hipError_t e = launch_istft_reg(spec, out, pl->d_window, pl->d_itw, unsigned(n), unsigned(n_frames), pl->p.hop_size, unsigned(batch),
                                              start, out_len, pl->dtype == SGX_F64 ? 1.0 / double(n) : double(1.0f / float(n)),
                                              (unsigned *)pl->d_flag, pl->dtype, s);
        if (e == hipSuccess) return SGX_OK;
        if (e != hipErrorNotSupported) SGX_HIP(pl, e);
    }
    if ((st = grow(pl, &pl->d_frames, &pl->d_frames_bytes, batch * n_frames * n * pl->elem)) != SGX_OK) return st;
    if ((st = launch_c2r_frames(pl, spec, pl->d_frames, batch, n_frames, true, pl->d_window, s)) != SGX_OK) return st;
    SGX_HIP(pl, launch_istft_ola(pl->d_frames, pl->d_window, out, unsigned(n), pl->p.hop_size, unsigned(n_frames), start, out_len,
                                 unsigned(batch), pl->dtype, s));
    return SGX_OK;
}

sgx_status check_flag(sgx_plan *pl, hipStream_t s) {
    unsigned flag = 0;
    SGX_HIP(pl, hipMemcpyAsync(&flag, pl->d_flag, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    SGX_HIP(pl, hipStreamSynchronize(s));
    if (flag)  // realfft: FftError::InputValues, mapped at fft_backend.rs:555-557
        return set_err(pl, SGX_BACKEND, "hip -- FFT backend error: imaginary part of the DC or Nyquist bin is non-zero");
    return SGX_OK;
}

}  // namespace

extern "C" {

int32_t sgx_abi_version(void) { return SGX_ABI_VERSION; }

int32_t sgx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *sgx_last_create_error(void) { return g_create_err.c_str(); }
const char *sgx_last_error(const sgx_plan *plan) { return plan ? plan->err.c_str() : g_create_err.c_str(); }

const char *sgx_kernel_name(const sgx_plan *plan) {
    if (!plan) return "";
    switch (plan->kind) {
    case K_R32X16_F32: return "r32x16_f32";
    case K_R32X32_F32: return "r32x32_f32";
    case K_D32X16_F64: return "d32x16_f64";
    case K_D512_F64: return "d512_f64";
    case K_R64X32_F32: return "r64x32_f32";
    case K_D32X32_F64: return "d32x32_f64";
    case K_LDS_RADIX2: return "lds_radix2";
    case K_TWO_FACTOR: return "two_factor_dft";
    case K_REG_RADIX: return "reg_radix";
    case K_BLUESTEIN: return "bluestein";
    case K_BIGFFT: return plan->big_n & (plan->big_n - 1) ? "big_chirpz" : "big_four_step";
    default: return "direct_dft";
    }
}

sgx_status sgx_plan_create(const sgx_params *params, sgx_plan **out) {
    if (out) *out = nullptr;
    if (!params || !out) return create_fail(SGX_INVALID_INPUT, "Invalid input: null argument");
    std::string msg;
    sgx_status st = validate(*params, msg);
    if (st != SGX_OK) return create_fail(st, msg);
    // Frames past the global-memory transforms' range (bigfft.hip: 2^20, powers of two 2^21) fail here, before gigabytes of host tables
    // are built for them (n_fft = 2^30 + 2: 49 s and 8.7 GB to reach the same answer further down)
    if (params->n_fft > (1u << 15) && !big_supported(params->n_fft))
        return create_fail(SGX_BACKEND, "hip -- FFT backend error: n_fft too large (the global-memory transforms take n_fft up to 2^20, powers of two up to 2^21)");
    sgx_plan *pl = new (std::nothrow) sgx_plan();
    if (!pl) return create_fail(SGX_INTERNAL, "Internal error: out of memory");
    pl->p = *params;
    if (params->window_kind == SGX_WIN_CUSTOM) {
        pl->custom_window.assign(params->custom_window, params->custom_window + params->n_fft);
    }
    pl->p.custom_window = nullptr;
    pl->dtype = params->dtype;
    pl->elem = params->dtype == SGX_F64 ? 8 : 4;
    pl->nb_fft = params->n_fft / 2 + 1;
    if (params->freq_scale == SGX_FREQ_CHROMA) pl->p.n_mels = 12;  // N_CHROMA rows go through the mapping slot
    // Mel and LogHz are both sparse row mappings of the power spectrum (MappingKind::{Mel, LogHz}, :1845-1865): one path
    pl->out_mode = params->amp_scale == SGX_AMP_COMPLEX ? OUT_COMPLEX
                   : params->freq_scale != SGX_FREQ_LINEAR ? OUT_MEL : OUT_LINEAR;
    pl->n_out = pl->out_mode == OUT_MEL ? pl->p.n_mels : pl->nb_fft;
    const unsigned mfcc_skip = (params->n_mfcc > 1 && !params->mfcc_include_c0) ? 1u : 0u;  // src/mfcc.rs:262-268
    pl->n_final = params->n_mfcc > 0 ? params->n_mfcc - mfcc_skip : pl->n_out;
    // S6: dB is applied only when LogParams were supplied; Decibels without them returns power
    pl->amp = params->amp_scale == SGX_AMP_MAGNITUDE ? AMP_MAGNITUDE
              : (params->amp_scale == SGX_AMP_DECIBELS && params->has_log_params) ? AMP_DB : AMP_POWER;
    if (params->freq_scale == SGX_FREQ_CHROMA) pl->amp = AMP_MAG_IN;  // the bank weighs magnitudes; nothing after it
    pl->eps = pl->amp == AMP_DB ? std::pow(10.0, params->floor_db / 10.0) : 0.0;
    build_window(pl->p, pl->custom_window, pl->window);
    if (params->freq_scale == SGX_FREQ_MEL) build_mel_csr(pl->p, pl->mel_ptr, pl->mel_col, pl->mel_val);
    if (params->freq_scale == SGX_FREQ_LOGHZ) build_loghz_csr(pl->p, pl->mel_ptr, pl->mel_col, pl->mel_val, pl->loghz_freqs);
    if (params->freq_scale == SGX_FREQ_CHROMA) build_chroma_csr(pl->p, pl->mel_ptr, pl->mel_col, pl->mel_val);
    if (params->freq_scale == SGX_FREQ_ERB) build_erb_dense(pl->p, pl->mel_ptr, pl->mel_col, pl->mel_val, pl->loghz_freqs);

    const bool pow2 = params->n_fft >= 4 && (params->n_fft & (params->n_fft - 1)) == 0;
    // powers of two (32..8192) and the listed even composite sizes: register-tiled kernel; other powers of two: LDS radix-2;
    // other composite lengths: two-factor DFT; primes fall through to the direct sum
    pl->kind = (pow2 || params->n_fft % 2 == 0) ? K_REG_RADIX : K_TWO_FACTOR;  // (even sizes outside the register-tiled list fall through)
    if (params->dtype == SGX_F32 && params->n_fft == 1024 && (SGX_ODDHOP || params->hop_size % 2 == 0)) pl->kind = K_R32X16_F32;
    // n_fft 512 (two frames per transform): staged variants at hops 64 / 128 / 160 / 256, the packed form's per-lane loads at every other even hop (per-bin outputs)
    if (params->dtype == SGX_F32 && params->n_fft == 512 && params->hop_size % 2 == 0 && params->hop_size <= 512) pl->kind = K_R32X16_F32;  // per-bin outputs (else falls back)
    if (params->dtype == SGX_F32 && params->n_fft == 2048) pl->kind = K_R32X32_F32;  // (odd hops: register-tiled kernel)
    if (params->dtype == SGX_F64 && params->n_fft == 1024) pl->kind = K_D32X16_F64;  // per-bin and complex outputs (filterbanks, odd hops: register-tiled kernel)
    if (params->dtype == SGX_F64 && params->n_fft == 512 && params->hop_size <= 260) pl->kind = K_D512_F64;  // two frames per transform
    if (params->dtype == SGX_F32 && params->n_fft == 4096) pl->kind = K_R64X32_F32;  // per-bin and complex outputs; filterbanks: split path
    if (params->dtype == SGX_F64 && params->n_fft == 2048) pl->kind = K_D32X32_F64;  // per-bin and complex outputs
    if (pl->kind == K_R32X16_F32) build_band_schedule(pl);  // before the kind is resolved: plan_geometry_r32x16_f32 asks for it
    if (pl->kind == K_R32X32_F32) build_band_schedule(pl, 8, r32x32::kSegs2, r32x32::kSch2MaxWords, 0);
    if (pl->kind == K_D32X16_F64) build_band_schedule(pl, 8, d32x16::kDSegs, d32x16::kDSchMaxWords, 0, 2);
    if (pl->kind == K_D512_F64) build_band_schedule(pl, 8, d512::kSegs, d512::kSchMaxWords, 0, 2);
    if (pl->kind == K_R64X32_F32) build_band_schedule(pl, 16, 2, 1u << 20, 0, 1);  // (as k_d32x32's, 4-byte weights)
    if (pl->kind == K_D32X32_F64) build_band_schedule(pl, 16, 2, 1u << 20, 0, 2);  // 16 half-waves x 8 slots; the table stays in global memory (kernels_d32x32.hip)
    {
        StftArgs probe;
        fill_args(pl, probe, nullptr, nullptr, 1, params->n_fft, params->n_fft, 1);
        KernelKind kind = pl->kind;
        bool ok = resolve_geometry(pl, probe, kind);
        if (pl->out_mode == OUT_MEL) {  // would the per-bin output run further up the chain?  (see run_device)
            StftArgs lin;
            fill_args(pl, lin, nullptr, nullptr, 1, params->n_fft, params->n_fft, 1);
            per_bin_args(pl, lin);
            KernelKind kind_lin = pl->kind;
            const bool ok_lin = resolve_geometry(pl, lin, kind_lin);
            const bool long_frames = ok && !kind_is_tuned(kind) && size_t(params->n_fft) * pl->elem >= SGX_SPLIT_BANK_BYTES;
            // f64 at the composite sizes (25-, 30-, 32-point first passes at one wave per SIMD): the fused stage never wins there —
            // 64 x 10 s, fused vs split: 400 198 vs 163 us, 800 193 vs 148, 1440 240 vs 173, ties at 240 / 480 / 960 / 1000 (f32: fused wins)
            const bool f64_mixed = ok && kind == K_REG_RADIX && pl->dtype == SGX_F64 && !pow2;
            const bool further_up = ok_lin && !kind_is_tuned(kind_lin) && (!ok || chain_pos(kind_lin) < chain_pos(kind));
            if (ok_lin && (long_frames || f64_mixed || further_up)) {
                pl->split_bank = true;
                kind = kind_lin;
                ok = true;
            }
        }
        // Lengths the chain above runs as a direct sum (primes: n / 2 multiply-adds per sample) or as a two-factor transform
        // (n = a b: a + b / 2 per sample) go through the chirp-z transform instead (bluestein.hip: one kernel, the length-M
        // convolution resident in LDS, M >= 2 n - 1) wherever that is cheaper — the reference's RustFFT plans such lengths with
        // Rader / Bluestein too (src/fft_backend.rs:376-385).
        {
            const unsigned n = params->n_fft;
            // (64-bit, and only for lengths a convolution can exist for — M <= 16384 fused, half-length form up to n = 16384: for n_fft
            // above 2^30 a 32-bit M shifts to 0 and never reaches 2 n - 1, above 2^31 `2 * n` wraps and a tiny M would pass)
            const bool bs_range = n <= 16384u;
            unsigned long long M64 = 1;
            unsigned l2 = 0;
            while (bs_range && M64 < 2ull * n - 1ull) { M64 <<= 1; ++l2; }
            const unsigned M = (unsigned)M64;
            double per_sample = 0.0;
            if (kind == K_DIRECT_DFT) per_sample = double(n) / 2.0;
            if (kind == K_TWO_FACTOR) {
                unsigned fa = 1;
                for (unsigned d = 2; (unsigned long long)d * d <= n; ++d)
                    if (n % d == 0) fa = d;
                per_sample = double(fa) + double(n / fa) / 2.0;
            }
            unsigned fa3 = 0, fb3 = 0, fc3 = 0;
            const bool can = bs_range && n >= 16 && bluestein_fused_split(M, pl->dtype, &fa3, &fb3, &fc3);
            const double cost = pl->out_mode == OUT_MEL && pl->dtype == SGX_F32 ? SGX_BS_COST_BANK32 : SGX_BS_COST;
            // even lengths whose own convolution does not fit LDS (f64 4098 ... 8192, f32 8194 ... 16384): half-length complex form,
            // one frame per sequence of n / 2 points (k_bs_c2c, RMODE 3), M >= n - 1
            unsigned Mh = 1, l2h = 0;
            while (bs_range && Mh < n - 1) { Mh <<= 1; ++l2h; }
            if (bs_range && !can && n >= 32 && n % 2 == 0 && (n & (n - 1)) != 0 && bluestein_fused_split(Mh, pl->dtype, &fa3, &fb3, &fc3) &&
                (!ok || per_sample > cost * double(l2h) * double(Mh) / double(n / 2))) {
                pl->bs_M = Mh;
                pl->bs_fwd_half = true;
                kind = K_BLUESTEIN;
                pl->split_bank = pl->out_mode == OUT_MEL;
                ok = true;
            } else if (can && (!ok || per_sample > cost * double(l2) * double(M) / double(n))) {
                pl->bs_M = M;
                kind = K_BLUESTEIN;
                // filterbank outputs: up to M = 1024 the bank's rows run inside the kernel (a tile holds >= 4 frame pairs: 320+ (band,
                // pair) work items; n_fft 251 Mel-80 dB 195 -> 181 us); longer sequences leave too few items with too long rows
                // (n_fft 2003: 323 us against 247) and take the split path: per-bin power, then one wave per (band, 64 frames)
                // (f64: the split path is faster at every length — n_fft 509 305 us against 322)
                pl->split_bank = pl->out_mode == OUT_MEL && (M > 1024 || pl->dtype == SGX_F64);
                ok = true;
            }
        }
        // What is left on the O(n^2) kernels above kBigMin points — odd lengths past the LDS chirp-z (8192; f64: 4096), even ones past
        // twice that — and every length no on-chip tile holds (f32 above 32768, f64 above 16384) goes through global memory
        // (bigfft.hip): four-step transforms for powers of two, chirp-z on top of them for the rest, O(n log n) for every n_fft
        // up to 2^20 like the reference's planner (src/fft_backend.rs:372-389).
        if (big_supported(params->n_fft) && (!ok || ((kind == K_DIRECT_DFT || kind == K_TWO_FACTOR) && params->n_fft > kBigMin))) {
            kind = K_BIGFFT;
            pl->big_n = params->n_fft;
            pl->split_bank = pl->out_mode == OUT_MEL;
            pl->bs_M = 0;
            pl->bs_fwd_half = false;
            ok = true;
        }
        if (!ok) {
            delete pl;
            return create_fail(SGX_BACKEND, "hip -- FFT backend error: n_fft too large (the global-memory transforms take n_fft up to 2^20, powers of two up to 2^21)");
        }
        pl->kind = kind;
    }

    pl->device = params->device;
    if (params->device != -2) {
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0) {
            delete pl;
            return create_fail(SGX_BACKEND, std::string("hip -- FFT backend error: no HIP device available (") +
                                                hipGetErrorString(e) + ")");
        }
        int dev = params->device;
        if (dev == -1 && hipGetDevice(&dev) != hipSuccess) dev = 0;
        if (dev < 0 || dev >= ndev) {
            delete pl;
            return create_fail(SGX_INVALID_INPUT, "Invalid input: device ordinal out of range");
        }
        pl->device = dev;
        DeviceGuard dg;
        auto dev_init = [&]() -> sgx_status {
            SGX_HIP(pl, dg.enter(dev));
            SGX_HIP(pl, hipEventCreate(&pl->ev0));
            SGX_HIP(pl, hipEventCreate(&pl->ev1));
            sgx_status s2 = pl->dtype == SGX_F64 ? build_device_tables<double>(pl) : build_device_tables<float>(pl);
            if (s2 != SGX_OK) return s2;
            // Everything the per-frame entry points (sgx_r2c / sgx_c2r = R2cPlan / C2rPlan::process) need is allocated here, so
            // that `process` never allocates (src/fft_backend.rs:21-24): one frame of staging each way, the rectangular window,
            // the inverse tables and the DC/Nyquist flag.  Batched calls size their scratch through sgx_reserve.
            const size_t frame_bytes = 2 * size_t(pl->nb_fft) * pl->elem;
            if (pl->kind == K_BIGFFT && (s2 = grow(pl, &pl->d_big, &pl->d_big_bytes, big_scratch_bytes(pl->big, pl->dtype, 1))) != SGX_OK) return s2;  // one sequence: the per-frame entry points
            if ((s2 = grow(pl, &pl->d_in, &pl->d_in_bytes, frame_bytes)) != SGX_OK) return s2;
            if ((s2 = grow(pl, &pl->d_out, &pl->d_out_bytes, frame_bytes)) != SGX_OK) return s2;
            std::vector<double> ones(pl->p.n_fft, 1.0);
            s2 = pl->dtype == SGX_F64 ? upload_cast<double>(pl, &pl->d_ones, ones) : upload_cast<float>(pl, &pl->d_ones, ones);
            if (s2 != SGX_OK) return s2;
            return pl->dtype == SGX_F64 ? inverse_tables<double>(pl) : inverse_tables<float>(pl);
        };
        st = dev_init();
        if (st != SGX_OK) {
            g_create_err = pl->err;
            free_device(pl);
            delete pl;
            return st;
        }
        pl->device_ready = true;
    }
    *out = pl;
    return SGX_OK;
}

void sgx_plan_destroy(sgx_plan *plan) {
    if (!plan) return;
    if (plan->device_ready) {
        DeviceGuard dg;
        (void)dg.enter(plan->device);
        free_device(plan);
    }
    delete plan;
}

sgx_status sgx_output_shape(const sgx_plan *plan, size_t n_samples, size_t *n_bins, size_t *n_frames) {
    if (!plan) return SGX_INVALID_INPUT;
    if (n_samples == 0) return set_err(plan, SGX_INVALID_INPUT, "Invalid input: signal length must be non-zero");
    if (n_bins) *n_bins = plan->n_final;
    if (n_frames) *n_frames = frame_count(plan->p, n_samples);
    return SGX_OK;
}

sgx_status sgx_execute(sgx_plan *plan, const void *samples, size_t batch, size_t n_samples, size_t sample_stride,
                       void *out, size_t out_elems, int32_t mem_kind, void *hip_stream) {
    size_t nf = 0;
    sgx_status st = check_call(plan, samples, batch, n_samples, sample_stride, out, out_elems, &nf);
    if (st != SGX_OK) return st;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if (mem_kind != SGX_MEM_HOST && mem_kind != SGX_MEM_DEVICE) return set_err(plan, SGX_INVALID_INPUT, "Invalid input: unknown mem_kind");
    DeviceGuard dg;
    SGX_HIP(plan, dg.enter(plan->device));
    if (mem_kind == SGX_MEM_DEVICE) return run_device(plan, samples, batch, n_samples, sample_stride, out, nf, s, 1, nullptr);
    // host pointers: plan-owned staging (sized by sgx_reserve, else grown on demand; reused across calls), synchronous
    const size_t in_bytes = ((batch - 1) * sample_stride + n_samples) * plan->elem;
    const size_t out_bytes = out_elems * plan->elem;
    if ((st = grow(plan, &plan->d_in, &plan->d_in_bytes, in_bytes)) != SGX_OK) return st;
    if ((st = grow(plan, &plan->d_out, &plan->d_out_bytes, out_bytes)) != SGX_OK) return st;
    SGX_HIP(plan, hipMemcpyAsync(plan->d_in, samples, in_bytes, hipMemcpyHostToDevice, s));
    if ((st = run_device(plan, plan->d_in, batch, n_samples, sample_stride, plan->d_out, nf, s, 1, nullptr)) != SGX_OK)
        return st;
    SGX_HIP(plan, hipMemcpyAsync(out, plan->d_out, out_bytes, hipMemcpyDeviceToHost, s));
    SGX_HIP(plan, hipStreamSynchronize(s));
    return SGX_OK;
}

sgx_status sgx_execute_timed(sgx_plan *plan, const void *samples, size_t batch, size_t n_samples,
                             size_t sample_stride, void *out, size_t out_elems, void *hip_stream, int32_t iters,
                             float *ms_per_launch) {
    size_t nf = 0;
    sgx_status st = check_call(plan, samples, batch, n_samples, sample_stride, out, out_elems, &nf);
    if (st != SGX_OK) return st;
    if (iters < 1 || !ms_per_launch) return set_err(plan, SGX_INVALID_INPUT, "Invalid input: iters must be >= 1");
    DeviceGuard dg;
    SGX_HIP(plan, dg.enter(plan->device));
    return run_device(plan, samples, batch, n_samples, sample_stride, out, nf, static_cast<hipStream_t>(hip_stream),
                      iters, ms_per_launch);
}

sgx_status sgx_axes(const sgx_plan *plan, size_t n_frames, double *freqs, double *times) {
    if (!plan) return SGX_INVALID_INPUT;
    const sgx_params &p = plan->p;
    if (times) {  // build_time_axis_seconds :2128-2139 — no centre offset (S10)
        const double dt = double(p.hop_size) / p.sample_rate_hz;
        for (size_t i = 0; i < n_frames; ++i) times[i] = double(i) * dt;
    }
    if (freqs && plan->p.n_mfcc > 0) {  // Mfcc carries no frequency axis (src/mfcc.rs:130-133): report coefficient indices
        const unsigned skip = plan->p.n_mfcc - plan->n_final;
        for (unsigned i = 0; i < plan->n_final; ++i) freqs[i] = double(i + skip);
    } else if (freqs && p.freq_scale == SGX_FREQ_CHROMA) {  // pitch classes C..B carry no Hz axis: report their indices
        for (unsigned i = 0; i < 12; ++i) freqs[i] = double(i);
    } else if (freqs) {
        if (p.freq_scale == SGX_FREQ_LOGHZ || p.freq_scale == SGX_FREQ_ERB) {  // frequencies stored with the mapping (:1932-1939)
            for (size_t i = 0; i < plan->loghz_freqs.size(); ++i) freqs[i] = plan->loghz_freqs[i];
        } else if (plan->out_mode == OUT_MEL) {  // mel_band_centres_hz :2510-2530 — 0..Nyquist, ignores f_min/f_max
            const double lo = hz2mel(0.0), hi = hz2mel(p.sample_rate_hz * 0.5);
            const double step = (hi - lo) / double(p.n_mels + 1);
            for (size_t i = 0; i < p.n_mels; ++i) freqs[i] = mel2hz(std::fma(double(i) + 1.0, step, lo));
        } else {  // :1911-1923 / :1446-1448
            const double df = p.sample_rate_hz / double(p.n_fft);
            for (size_t k = 0; k < plan->nb_fft; ++k) freqs[k] = double(k) * df;
        }
    }
    return SGX_OK;
}

sgx_status sgx_r2c(sgx_plan *plan, const void *in, size_t in_len, void *out, size_t out_len) {
    if (!plan) return SGX_INVALID_INPUT;
    if (!in || !out) return set_err(plan, SGX_INVALID_INPUT, "Invalid input: null buffer");
    const size_t n = plan->p.n_fft, nb = plan->nb_fft;
    if (in_len != n)  // validate_fft_io src/fft_backend.rs:264-282
        return dim_err(plan, n, in_len);
    if (out_len != nb)
        return dim_err(plan, nb, out_len);
    if (!plan->device_ready)
        return set_err(plan, SGX_BACKEND, "hip -- FFT backend error: plan has no HIP device (host-only plan)");
    DeviceGuard dg;
    SGX_HIP(plan, dg.enter(plan->device));  // staging, window of ones: allocated at plan creation
    SGX_HIP(plan, hipMemcpy(plan->d_in, in, n * plan->elem, hipMemcpyHostToDevice));
    StftArgs a;
    fill_args(plan, a, plan->d_in, plan->d_out, 1, n, n, 1);
    a.window = plan->d_ones;
    a.pad = 0;
    a.out_mode = OUT_COMPLEX;
    a.n_out = plan->nb_fft;
    a.amp = AMP_POWER;
    KernelKind kind = plan->kind;
    if (!resolve_geometry(plan, a, kind))  // (down the chain: a single frame leaves the tuned kernels' multi-frame tiles for the register-tiled kernel)
        return set_err(plan, SGX_BACKEND, "hip -- FFT backend error: n_fft too large");
    if (kind_is_tuned(kind)) a.window = plan->d_ones_half;
    SGX_HIP(plan, launch(plan, a, kind, nullptr));
    SGX_HIP(plan, hipMemcpy(out, plan->d_out, 2 * nb * plan->elem, hipMemcpyDeviceToHost));
    return SGX_OK;
}

// ---- inverse path entry points (helpers above the extern "C" block)
sgx_status sgx_istft_length(const sgx_plan *plan, size_t n_frames, size_t *n_samples) {
    if (!plan || !n_samples) return SGX_INVALID_INPUT;
    if (n_frames == 0) return set_err(const_cast<sgx_plan *>(plan), SGX_INVALID_INPUT, "Invalid input: stft matrix must be non-empty");
    *n_samples = istft_length(plan->p, n_frames);
    return SGX_OK;
}

sgx_status sgx_istft(sgx_plan *plan, const void *stft, size_t batch, size_t n_bins, size_t n_frames, void *out,
                     size_t out_elems, int32_t mem_kind, void *hip_stream) {
    if (!plan) return SGX_INVALID_INPUT;
    if (!stft || !out) return set_err(plan, SGX_INVALID_INPUT, "Invalid input: null buffer");
    if (batch == 0 || n_frames == 0) return set_err(plan, SGX_INVALID_INPUT, "Invalid input: stft matrix must be non-empty");
    if (n_bins != plan->nb_fft)  // :4876-4879
        return dim_err(plan, plan->nb_fft, n_bins);
    if (batch > 65535 || n_frames > 0x7fffffffull) return set_err(plan, SGX_INVALID_INPUT, "Invalid input: batch or frame count too large");
    const size_t len = istft_length(plan->p, n_frames);
    if (out_elems != batch * len)
        return dim_err(plan, batch * len, out_elems);
    if (!plan->device_ready)
        return set_err(plan, SGX_BACKEND, "hip -- FFT backend error: plan has no HIP device (host-only plan)");
    DeviceGuard dg;
    SGX_HIP(plan, dg.enter(plan->device));
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if (mem_kind == SGX_MEM_DEVICE) return run_istft(plan, stft, batch, n_frames, out, len, s);
    if (mem_kind != SGX_MEM_HOST) return set_err(plan, SGX_INVALID_INPUT, "Invalid input: unknown mem_kind");
    const size_t in_bytes = batch * n_bins * n_frames * 2 * plan->elem, out_bytes = out_elems * plan->elem;
    sgx_status st;
    if ((st = grow(plan, &plan->d_in, &plan->d_in_bytes, in_bytes)) != SGX_OK) return st;
    if ((st = grow(plan, &plan->d_out, &plan->d_out_bytes, out_bytes)) != SGX_OK) return st;
    SGX_HIP(plan, hipMemcpyAsync(plan->d_in, stft, in_bytes, hipMemcpyHostToDevice, s));
    if ((st = run_istft(plan, plan->d_in, batch, n_frames, plan->d_out, len, s)) != SGX_OK) return st;
    SGX_HIP(plan, hipMemcpyAsync(out, plan->d_out, out_bytes, hipMemcpyDeviceToHost, s));
    return check_flag(plan, s);  // synchronises
}

sgx_status sgx_c2r(sgx_plan *plan, const void *in, size_t in_len, void *out, size_t out_len) {
    if (!plan) return SGX_INVALID_INPUT;
    if (!in || !out) return set_err(plan, SGX_INVALID_INPUT, "Invalid input: null buffer");
    const size_t n = plan->p.n_fft, nb = plan->nb_fft;
    if (in_len != nb)  // fft_backend.rs:538-544
        return dim_err(plan, nb, in_len);
    if (out_len != n)  // :545-550
        return dim_err(plan, n, out_len);
    if (!plan->device_ready)
        return set_err(plan, SGX_BACKEND, "hip -- FFT backend error: plan has no HIP device (host-only plan)");
    DeviceGuard dg;
    SGX_HIP(plan, dg.enter(plan->device));  // staging, inverse tables, flag: allocated at plan creation
    sgx_status st;
    SGX_HIP(plan, hipMemcpy(plan->d_in, in, 2 * nb * plan->elem, hipMemcpyHostToDevice));
    SGX_HIP(plan, hipMemsetAsync(plan->d_flag, 0, sizeof(unsigned), nullptr));
    if ((st = launch_c2r_frames(plan, plan->d_in, plan->d_out, 1, 1, false, nullptr, nullptr)) != SGX_OK) return st;
    SGX_HIP(plan, hipMemcpy(out, plan->d_out, n * plan->elem, hipMemcpyDeviceToHost));
    return check_flag(plan, nullptr);
}

int32_t sgx_plan_device(const sgx_plan *plan) { return plan ? plan->device : -2; }

sgx_status sgx_last_dim_mismatch(const sgx_plan *plan, size_t *expected, size_t *got) {
    if (!plan || !expected || !got) return SGX_INVALID_INPUT;
    *expected = plan->dm_expected;
    *got = plan->dm_got;
    return SGX_OK;
}

sgx_status sgx_reserve(sgx_plan *plan, size_t batch, size_t n_samples, int32_t host_staging, int32_t inverse) {
    if (!plan) return SGX_INVALID_INPUT;
    if (batch == 0 || n_samples == 0) return set_err(plan, SGX_INVALID_INPUT, "Invalid input: samples must be non-empty");
    if (!plan->device_ready)
        return set_err(plan, SGX_BACKEND, "hip -- FFT backend error: plan has no HIP device (host-only plan)");
    DeviceGuard dg;
    SGX_HIP(plan, dg.enter(plan->device));
    const size_t nf = frame_count(plan->p, n_samples);
    const size_t spec_elems = batch * size_t(plan->n_final) * nf * (plan->out_mode == OUT_COMPLEX ? 2 : 1);
    sgx_status st;
    if (plan->p.n_mfcc > 0 &&
        (st = grow(plan, &plan->d_melbuf, &plan->d_melbuf_bytes, batch * size_t(plan->n_out) * nf * plan->elem)) != SGX_OK)
        return st;
    if (!inverse && plan->split_bank &&
        (st = grow(plan, &plan->d_pwbuf, &plan->d_pwbuf_bytes, batch * size_t(plan->nb_fft) * nf * plan->elem)) != SGX_OK)
        return st;
    if (plan->kind == K_BIGFFT &&
        (st = grow(plan, &plan->d_big, &plan->d_big_bytes, big_scratch_bytes(plan->big, plan->dtype, batch * ((nf + 1) / 2)))) != SGX_OK)
        return st;
    if (inverse) {  // sgx_istft of `batch` spectra whose frame count is that of n_samples-long signals
        // the same tests run_istft applies: tuned n_fft = 1024 kernel, else the fused register-tiled kernel; only the unfused
        // fallback (rows + overlap-add) touches the frame scratch
        const bool fused = (plan->d_itwr && nf * 513ull * 8ull < 0x7fffffffull) || (plan->d_itwr2 && nf * 1025ull * 8ull < 0x7fffffffull) || (plan->d_itwrd && nf * 513ull * 16ull < 0x7fffffffull) || (plan->istft_d512 && nf * 257ull * 16ull < 0x7fffffffull) ||
                           (plan->kind != K_BIGFFT && nf <= 0xffffffffull && batch <= 0xffffffffull &&
                            istft_reg_fuses(plan->d_window, plan->p.n_fft, unsigned(nf), plan->p.hop_size, unsigned(batch), plan->dtype));
        if (!fused && (st = grow(plan, &plan->d_frames, &plan->d_frames_bytes, batch * nf * plan->p.n_fft * plan->elem)) != SGX_OK) return st;
    }
    if (host_staging) {
        const size_t sig_bytes = batch * n_samples * plan->elem, spec_bytes = std::max(spec_elems, batch * size_t(plan->nb_fft) * nf * 2) * plan->elem;
        if ((st = grow(plan, &plan->d_in, &plan->d_in_bytes, inverse ? spec_bytes : sig_bytes)) != SGX_OK) return st;
        if ((st = grow(plan, &plan->d_out, &plan->d_out_bytes, inverse ? sig_bytes + batch * plan->p.n_fft * plan->elem : spec_bytes)) != SGX_OK) return st;
    }
    return SGX_OK;
}

sgx_status sgx_window(const sgx_plan *plan, double *out) {
    if (!plan || !out) return SGX_INVALID_INPUT;
    std::copy(plan->window.begin(), plan->window.end(), out);
    return SGX_OK;
}

sgx_status sgx_mel_weights(const sgx_plan *plan, size_t *nnz, uint32_t *row_ptr, uint32_t *cols, double *vals) {
    if (!plan) return SGX_INVALID_INPUT;
    if (plan->out_mode != OUT_MEL) return set_err(plan, SGX_INVALID_INPUT, "Invalid input: plan has no Mel filterbank");
    if (nnz) *nnz = plan->mel_col.size();
    if (row_ptr) std::copy(plan->mel_ptr.begin(), plan->mel_ptr.end(), row_ptr);
    if (cols) std::copy(plan->mel_col.begin(), plan->mel_col.end(), cols);
    if (vals) std::copy(plan->mel_val.begin(), plan->mel_val.end(), vals);
    return SGX_OK;
}

sgx_status sgx_shard_range(size_t batch, int32_t world_size, int32_t rank, size_t *start, size_t *count) {
    if (world_size <= 0 || rank < 0 || rank >= world_size || !start || !count) return SGX_INVALID_INPUT;
    const size_t base = batch / size_t(world_size), rem = batch % size_t(world_size);
    const size_t r = size_t(rank);
    *count = base + (r < rem ? 1 : 0);
    *start = r * base + std::min(r, rem);
    return SGX_OK;
}

}  // extern "C"
