// fft_inreg.h — in-register power-of-two FFTs on packed f32 pairs (complex = 2-float vector), shared by the tuned
// gfx950 kernels.  All indices and twiddles are compile-time; every butterfly is 2-4 packed-f32 VALU instructions
// (v_pk_add/mul/fma_f32 with op_sel / neg modifiers — checked in the ISA).
#pragma once
#include <hip/hip_runtime.h>

#ifndef SGX_BFLY3
#define SGX_BFLY3 1  // twiddled butterflies as three packed instructions (x1 = 2 e - x0); 0: four.  Measured on the tuned kernel: Mel-80 power
                     // 115.8 -> 111.2 us, linear power 115.4 -> 113.0 us per 256 x 10 s (profiles/experiments_r03/bfly3_bandpf_abv.txt)
#endif

namespace sgx {
namespace inreg {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr double kCos64[64] = {1.0, 0.9951847266721969, 0.9807852804032304, 0.9569403357322088, 0.9238795325112867, 0.881921264348355, 0.8314696123025452, 0.773010453362737, 0.7071067811865476, 0.6343932841636455, 0.5555702330196023, 0.4713967368259978, 0.38268343236508984, 0.29028467725446233, 0.19509032201612833, 0.09801714032956077, 6.123233995736766e-17, -0.09801714032956065, -0.1950903220161282, -0.29028467725446216, -0.3826834323650897, -0.4713967368259977, -0.555570233019602, -0.6343932841636454, -0.7071067811865475, -0.773010453362737, -0.8314696123025453, -0.8819212643483549, -0.9238795325112867, -0.9569403357322088, -0.9807852804032304, -0.9951847266721968, -1.0, -0.9951847266721969, -0.9807852804032304, -0.9569403357322089, -0.9238795325112868, -0.881921264348355, -0.8314696123025455, -0.7730104533627371, -0.7071067811865477, -0.6343932841636459, -0.5555702330196022, -0.47139673682599786, -0.38268343236509034, -0.29028467725446244, -0.19509032201612866, -0.09801714032956045, -1.8369701987210297e-16, 0.09801714032956009, 0.1950903220161283, 0.29028467725446205, 0.38268343236509, 0.4713967368259976, 0.5555702330196018, 0.6343932841636456, 0.7071067811865474, 0.7730104533627367, 0.8314696123025452, 0.8819212643483548, 0.9238795325112865, 0.9569403357322088, 0.9807852804032303, 0.9951847266721969};
constexpr double kSin64[64] = {0.0, 0.0980171403295606, 0.19509032201612825, 0.29028467725446233, 0.3826834323650898, 0.47139673682599764, 0.5555702330196022, 0.6343932841636455, 0.7071067811865475, 0.773010453362737, 0.8314696123025452, 0.8819212643483549, 0.9238795325112867, 0.9569403357322089, 0.9807852804032304, 0.9951847266721968, 1.0, 0.9951847266721969, 0.9807852804032304, 0.9569403357322089, 0.9238795325112867, 0.881921264348355, 0.8314696123025455, 0.7730104533627371, 0.7071067811865476, 0.6343932841636455, 0.5555702330196022, 0.47139673682599786, 0.3826834323650899, 0.2902846772544624, 0.1950903220161286, 0.09801714032956083, 1.2246467991473532e-16, -0.09801714032956059, -0.19509032201612836, -0.2902846772544621, -0.38268343236508967, -0.47139673682599764, -0.555570233019602, -0.6343932841636453, -0.7071067811865475, -0.7730104533627367, -0.8314696123025452, -0.8819212643483549, -0.9238795325112865, -0.9569403357322088, -0.9807852804032303, -0.9951847266721969, -1.0, -0.9951847266721969, -0.9807852804032304, -0.9569403357322089, -0.9238795325112866, -0.881921264348355, -0.8314696123025455, -0.7730104533627369, -0.7071067811865477, -0.6343932841636459, -0.5555702330196022, -0.4713967368259979, -0.3826834323650904, -0.2902846772544625, -0.19509032201612872, -0.0980171403295605};

// The helpers below are generic in the 2-element vector type V (v2f: packed-f32 instructions; v2d: pairs of f64 ops).
typedef double v2d __attribute__((ext_vector_type(2)));
template <typename V> struct ElemOf;
template <> struct ElemOf<v2f> { typedef float type; };
template <> struct ElemOf<v2d> { typedef double type; };

template <typename V> __device__ __forceinline__ V swp(V a) { return __builtin_shufflevector(a, a, 1, 0); }
template <typename V> __device__ __forceinline__ V lo2(V a) { return __builtin_shufflevector(a, a, 0, 0); }
template <typename V> __device__ __forceinline__ V hi2(V a) { return __builtin_shufflevector(a, a, 1, 1); }
template <typename V> __device__ __forceinline__ V pfma(V a, V b, V c) { return __builtin_elementwise_fma(a, b, c); }
template <typename V> __device__ __forceinline__ V cmulv(V x, V t) { return pfma(swp(x), (V){-t.y, t.y}, x * lo2(t)); }

// x0 = e + W o, x1 = e - W o, W = W_N^K a compile-time constant; every case is 2-4 packed instructions
template <int N, int K, typename V>
__device__ __forceinline__ void bfly(V e, V o, V &x0, V &x1) {
    typedef typename ElemOf<V>::type E;
    constexpr int idx = K * (64 / N);
    if constexpr (idx == 0) {
        x0 = e + o;
        x1 = e - o;
    } else if constexpr (idx == 16) {  // W = -i : W o = (o.y, -o.x)
        const V so = swp(o);
        x0 = pfma(so, (V){E(1), E(-1)}, e);
        x1 = pfma(so, (V){E(-1), E(1)}, e);
    } else if constexpr (idx == 8) {  // W = c(1 - i): W o = c (o.x + o.y, o.y - o.x)
        constexpr E c = E(0.70710678118654752440);
        const V s = pfma(swp(o), (V){E(1), E(-1)}, o);
        x0 = pfma(s, (V){c, c}, e);
        x1 = pfma(s, (V){-c, -c}, e);
    } else if constexpr (idx == 24) {  // W = c(-1 - i): W o = c (o.y - o.x, -o.x - o.y)
        constexpr E c = E(0.70710678118654752440);
        const V s = pfma(swp(o), (V){E(1), E(-1)}, -o);
        x0 = pfma(s, (V){c, c}, e);
        x1 = pfma(s, (V){-c, -c}, e);
    } else {
        constexpr E wr = (E)kCos64[idx], wi = (E)(-kSin64[idx]);
        const V so = swp(o);
        x0 = pfma(so, (V){-wi, wi}, pfma(o, (V){wr, wr}, e));
#if SGX_BFLY3  // x1 = 2 e - x0: three packed instructions per twiddled butterfly instead of four (one more rounding of x0 carried into x1)
        x1 = pfma(e, (V){E(2), E(2)}, -x0);
#else
        x1 = pfma(so, (V){wi, -wi}, pfma(o, (V){-wr, -wr}, e));
#endif
    }
}
template <int N, int K, typename V>
struct Comb {
    static __device__ __forceinline__ void run(V (&x)[N], const V (&e)[N / 2], const V (&o)[N / 2]) {
        bfly<N, K, V>(e[K], o[K], x[K], x[K + N / 2]);
        if constexpr (K + 1 < N / 2) Comb<N, K + 1, V>::run(x, e, o);
    }
};
// in-register radix-2 DIT, natural order in and out; all indices and twiddles are compile-time.
// WIN: x holds raw samples and w the window; the multiply is fused into the first butterfly.
template <int N, bool WIN, typename V = v2f>
struct Fft {
    static __device__ __forceinline__ void run(V (&x)[N], const V (&w)[N]) {
        if constexpr (N == 1) {
            if constexpr (WIN) x[0] = x[0] * w[0];
        } else if constexpr (N == 2) {
            if constexpr (WIN) {
                const V t = x[0] * w[0];
                const V u = x[1];
                x[0] = pfma(u, w[1], t);
                x[1] = pfma(-u, w[1], t);
            } else {
                const V a = x[0], b = x[1];
                x[0] = a + b;
                x[1] = a - b;
            }
        } else {
            V e[N / 2], o[N / 2], we[N / 2], wo[N / 2];
#pragma unroll
            for (int k = 0; k < N / 2; ++k) {
                e[k] = x[2 * k];
                o[k] = x[2 * k + 1];
                we[k] = w[2 * k];
                wo[k] = w[2 * k + 1];
            }
            Fft<N / 2, WIN, V>::run(e, we);
            Fft<N / 2, WIN, V>::run(o, wo);
            Comb<N, 0, V>::run(x, e, o);
        }
    }
};

// ---- mixed radix (factors 2, 3, 5) -----------------------------------------------------------------------------------------
// compile-time cos / sin (argument reduced to [-pi/2, pi/2], Taylor series in double: error ~1e-16)
constexpr double kPiC = 3.14159265358979323846264338327950288;
constexpr double ct_sin_reduced(double x) {  // |x| <= pi/2
    double term = x, sum = x;
    for (int i = 1; i < 16; ++i) {
        term *= -x * x / double((2 * i) * (2 * i + 1));
        sum += term;
    }
    return sum;
}
constexpr double ct_sin(double x) {  // any |x| <= 4 pi
    while (x > kPiC) x -= 2.0 * kPiC;
    while (x < -kPiC) x += 2.0 * kPiC;
    if (x > 0.5 * kPiC) x = kPiC - x;
    if (x < -0.5 * kPiC) x = -kPiC - x;
    return ct_sin_reduced(x);
}
constexpr double ct_cos(double x) { return ct_sin(x + 0.5 * kPiC); }

// x * W_N^E, E a compile-time exponent (forward transform: W = e^{-2 pi i / N})
template <int N, int E, typename V>
__device__ __forceinline__ V cmul_wn(V x) {
    typedef typename ElemOf<V>::type T;
    constexpr int e = E % N;
    if constexpr (e == 0) {
        return x;
    } else {
        constexpr T wr = (T)ct_cos(2.0 * kPiC * double(e) / double(N)), wi = (T)(-ct_sin(2.0 * kPiC * double(e) / double(N)));
        return pfma(swp(x), (V){-wi, wi}, x * (V){wr, wr});
    }
}

template <typename V>
__device__ __forceinline__ void dft3(V (&c)[3]) {
    typedef typename ElemOf<V>::type T;
    constexpr T h = (T)0.86602540378443864676;  // sin(2 pi / 3)
    const V s = c[1] + c[2], d = c[1] - c[2];
    const V t = pfma(s, (V){T(-0.5), T(-0.5)}, c[0]);
    const V u = swp(d) * (V){h, -h};  // -i h d
    c[0] = c[0] + s;
    c[1] = t + u;
    c[2] = t - u;
}

template <typename V>
__device__ __forceinline__ void dft5(V (&c)[5]) {
    typedef typename ElemOf<V>::type T;
    constexpr T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410;  // cos(2 pi / 5), cos(4 pi / 5)
    constexpr T s1 = (T)0.95105651629515357212, s2 = (T)0.58778525229247312917;   // sin(2 pi / 5), sin(4 pi / 5)
    const V a1 = c[1] + c[4], a2 = c[2] + c[3], b1 = c[1] - c[4], b2 = c[2] - c[3];
    const V e1 = pfma(a2, (V){c2, c2}, pfma(a1, (V){c1, c1}, c[0]));
    const V e2 = pfma(a2, (V){c1, c1}, pfma(a1, (V){c2, c2}, c[0]));
    const V d1 = pfma(b2, (V){s2, s2}, b1 * (V){s1, s1});
    const V d2 = pfma(b2, (V){-s1, -s1}, b1 * (V){s2, s2});
    const V j1 = swp(d1) * (V){T(1), T(-1)}, j2 = swp(d2) * (V){T(1), T(-1)};  // -i d
    c[0] = c[0] + a1 + a2;
    c[1] = e1 + j1;
    c[4] = e1 - j1;
    c[2] = e2 + j2;
    c[3] = e2 - j2;
}

// in-register N-point forward DFT, N = 2^a 3^b 5^c, natural order in and out (decimation in time over the odd factor)
template <int N, typename V>
struct MixFft {
    static __device__ __forceinline__ void run(V (&x)[N]) {
        if constexpr ((N & (N - 1)) == 0) {
            Fft<N, false, V>::run(x, x);
        } else {
            constexpr int P = (N % 5 == 0) ? 5 : 3;
            static_assert(N % P == 0, "MixFft: factors 2, 3 and 5 only");
            constexpr int Q = N / P;
            V t[P][Q];
            step1<0>(x, t);
#pragma unroll
            for (int k1 = 0; k1 < P; ++k1) MixFft<Q, V>::run(t[k1]);
#pragma unroll
            for (int k1 = 0; k1 < P; ++k1)
#pragma unroll
                for (int k2 = 0; k2 < Q; ++k2) x[k1 + P * k2] = t[k1][k2];
        }
    }

private:
    // column n2: P-point DFT over n1 of x[Q n1 + n2], then the twiddle W_N^(k1 n2) (compile-time, hence the recursion)
    template <int N2, int P, int Q>
    static __device__ __forceinline__ void step1(const V (&x)[N], V (&t)[P][Q]) {
        V c[P];
#pragma unroll
        for (int n1 = 0; n1 < P; ++n1) c[n1] = x[Q * n1 + N2];
        if constexpr (P == 5) dft5(c); else dft3(c);
        twid<N2, 0>(c, t);
        if constexpr (N2 + 1 < Q) step1<N2 + 1>(x, t);
    }
    template <int N2, int K1, int P, int Q>
    static __device__ __forceinline__ void twid(const V (&c)[P], V (&t)[P][Q]) {
        t[K1][N2] = cmul_wn<N, K1 * N2, V>(c[K1]);
        if constexpr (K1 + 1 < P) twid<N2, K1 + 1>(c, t);
    }
};

}  // namespace inreg
}  // namespace sgx
