// db_f64.h — 10 log10(x) in f64 without libm's log10 (round 5).
//
// The reference's dB scaling is `10 * max(p, eps).log10()` in T (src/spectrogram.rs:2068-2080).  In f32 the kernels use the hardware's log2;
// in f64 every dB output went through libm's log10 — some hundred instructions with branches — and a per-bin dB output took twice the
// time of the power it is computed from (256 x 10 s, n_fft 1024: 528 against 289 us, profiles/bench_r04_f64.txt).  Here, branch-free:
//   x = m 2^e, m in [sqrt(1/2), sqrt(2));  s = (m - 1) / (m + 1), |s| <= 0.1716;  ln m = 2 s (1 + z/3 + z^2/5 + ... + z^9/19), z = s^2
// (truncation z^10 / 21 < 2.4e-17), the quotient by the hardware reciprocal and two Newton steps, and
//   10 log10 x = e (10 log10 2) + (20 / ln 10) s P(z).
// Measured against the f64 oracle (libm): |error| < 4e-15 dB + 3e-16 |dB| over 1e-300 .. 1e300 (tests/test_gpu_parity.py::test_f64_db_epilogue);
// the parity tests allow 1e-8 dB.  0 -> -inf, +inf -> +inf, NaN and negative arguments -> NaN, subnormals through v_frexp.
#pragma once
#include <hip/hip_runtime.h>

namespace sgx {

__device__ __forceinline__ double db_f64(double x) {
    int e;
    double m = frexp(x, &e);  // m in [0.5, 1): v_frexp_mant_f64 / v_frexp_exp_i32_f64
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double num = m - 1.0, den = m + 1.0;
    double r = __builtin_amdgcn_rcp(den);   // ~ 1e-8 relative; two Newton steps -> f64
    r = fma(fma(-den, r, 1.0), r, r);
    r = fma(fma(-den, r, 1.0), r, r);
    double s = num * r;
    s = fma(fma(-den, s, num), r, s);       // one correction of the quotient itself
    const double z = s * s;
    double p = 1.0 / 19.0;
    p = fma(p, z, 1.0 / 17.0);
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    p = fma(p, z, 1.0);
    const double kA = 8.6858896380650365530;    // 20 / ln 10
    const double kB = 3.0102999566398119521;    // 10 log10 2
    double y = fma((double)e, kB, kA * (s * p));
    y = x == 0.0 ? -__builtin_huge_val() : y;
    y = x == __builtin_huge_val() ? x : y;
    y = x < 0.0 ? __builtin_nan("") : y;
    return y;
}

}  // namespace sgx
