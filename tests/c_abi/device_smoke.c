/* Stand-alone consumer of libspectro_hip.so with DEVICE pointers: no Python, no torch — hipMalloc / hipMemcpy from the HIP
 * runtime and the C ABI of include/spectro_hip.h only.  Checks a batched linear-power STFT and its Mel-dB variant against a
 * direct O(n^2) DFT of a few frames computed here, then a forward -> inverse round trip.  Built (hipcc, C mode) and run by
 * tests/test_c_abi.py::test_device_smoke on the GPU box. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spectro_hip.h"

#define CHECK(c)                                                          \
    do {                                                                  \
        if (!(c)) {                                                       \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                     \
        }                                                                 \
    } while (0)

static const double kPi = 3.14159265358979323846;

int main(void) {
    const size_t B = 3, N = 20000, n_fft = 1024, hop = 256, pad = 512;
    float *x = (float *)malloc(B * N * sizeof(float));
    unsigned s = 12345u;
    for (size_t i = 0; i < B * N; ++i) {
        s = s * 1664525u + 1013904223u;
        x[i] = (float)((double)(s >> 8) / 16777216.0 - 0.5) + 0.5f * (float)sin(2.0 * kPi * 440.0 * (double)(i % N) / 16000.0);
    }
    sgx_params p;
    memset(&p, 0, sizeof p);
    p.n_fft = (uint32_t)n_fft; p.hop_size = (uint32_t)hop; p.centre = 1; p.window_kind = SGX_WIN_HANNING; p.sample_rate_hz = 16000.0;
    p.freq_scale = SGX_FREQ_LINEAR; p.amp_scale = SGX_AMP_POWER; p.dtype = SGX_F32; p.device = -1;
    sgx_plan *plan = NULL;
    CHECK(sgx_plan_create(&p, &plan) == SGX_OK);
    size_t nb = 0, nf = 0;
    CHECK(sgx_output_shape(plan, N, &nb, &nf) == SGX_OK && nb == 513 && nf == (N + 2 * pad - n_fft) / hop + 1);
    float *dx = NULL, *dout = NULL;
    CHECK(hipMalloc((void **)&dx, B * N * sizeof(float)) == hipSuccess);
    CHECK(hipMalloc((void **)&dout, B * nb * nf * sizeof(float)) == hipSuccess);
    CHECK(hipMemcpy(dx, x, B * N * sizeof(float), hipMemcpyHostToDevice) == hipSuccess);
    CHECK(sgx_execute(plan, dx, B, N, N, dout, B * nb * nf, SGX_MEM_DEVICE, NULL) == SGX_OK);
    CHECK(hipDeviceSynchronize() == hipSuccess);
    float *out = (float *)malloc(B * nb * nf * sizeof(float));
    CHECK(hipMemcpy(out, dout, B * nb * nf * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess);
    /* direct DFT of frames 0 (zero-padded edge), 7 and the last one of signal 1, symmetric Hann (S3) */
    const size_t frames[3] = {0, 7, 0};
    double worst = 0.0, peak = 0.0;
    for (int q = 0; q < 3; ++q) {
        const size_t f = q == 2 ? nf - 1 : frames[q], b = 1;
        for (size_t k = 0; k < nb; k += 37) {
            double re = 0.0, im = 0.0;
            for (size_t i = 0; i < n_fft; ++i) {
                const long long sidx = (long long)(f * hop + i) - (long long)pad;
                const double v = (sidx >= 0 && sidx < (long long)N) ? (double)x[b * N + (size_t)sidx] : 0.0;
                const double w = (double)(float)(0.5 - 0.5 * cos(2.0 * kPi * (double)i / (double)(n_fft - 1)));
                const double a = -2.0 * kPi * (double)((i * k) % n_fft) / (double)n_fft;
                re += v * w * cos(a);
                im += v * w * sin(a);
            }
            const double ref = re * re + im * im, got = (double)out[(b * nb + k) * nf + f];
            if (fabs(got - ref) > worst) worst = fabs(got - ref);
            if (ref > peak) peak = ref;
        }
    }
    printf("linear power: max abs err %.3e (peak %.3e)\n", worst, peak);
    CHECK(worst <= 1e-4 * peak);
    CHECK(strcmp(sgx_kernel_name(plan), "r32x16_f32") == 0);
    CHECK(sgx_plan_device(plan) >= 0);
    CHECK(sgx_reserve(plan, B, N, 1, 0) == SGX_OK); /* host staging sized ahead: the host-pointer call below does not allocate */
    {
        float *hout = (float *)malloc(B * nb * nf * sizeof(float));
        CHECK(sgx_execute(plan, x, B, N, N, hout, B * nb * nf, SGX_MEM_HOST, NULL) == SGX_OK);
        CHECK(memcmp(hout, out, B * nb * nf * sizeof(float)) == 0); /* host path == device path, bit for bit */
        free(hout);
    }
    /* multi-GPU entry points with a world of one rank: unique id, communicator, sharded execute + RCCL gather */
    {
        unsigned char id[SGX_COMM_ID_BYTES];
        sgx_comm *comm = NULL;
        CHECK(sgx_comm_unique_id(id) == SGX_OK);
        CHECK(sgx_comm_create(id, 1, 0, -1, &comm) == SGX_OK && comm != NULL);
        float *dg = NULL;
        CHECK(hipMalloc((void **)&dg, B * nb * nf * sizeof(float)) == hipSuccess);
        CHECK(hipMemset(dg, 0, B * nb * nf * sizeof(float)) == hipSuccess);
        CHECK(sgx_shard_execute(plan, comm, dx, B, N, N, NULL, dg, NULL) == SGX_OK);
        CHECK(hipDeviceSynchronize() == hipSuccess);
        float *g = (float *)malloc(B * nb * nf * sizeof(float));
        CHECK(hipMemcpy(g, dg, B * nb * nf * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess);
        CHECK(memcmp(g, out, B * nb * nf * sizeof(float)) == 0);
        /* a separate shard buffer gathered into the full one */
        CHECK(hipMemset(dg, 0, B * nb * nf * sizeof(float)) == hipSuccess);
        CHECK(sgx_shard_execute(plan, comm, dx, B, N, N, dout, dg, NULL) == SGX_OK);
        CHECK(hipDeviceSynchronize() == hipSuccess);
        CHECK(hipMemcpy(g, dg, B * nb * nf * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess);
        CHECK(memcmp(g, out, B * nb * nf * sizeof(float)) == 0);
        /* the chunked form (ABI 7): compute on the caller's stream, each chunk's exchange on the communicator's own stream, ordered by
         * events — here through the real RCCL with one rank, in place and through the shard buffer */
        for (int chunks = 1; chunks <= 4; chunks += 3) {
            CHECK(hipMemset(dg, 0, B * nb * nf * sizeof(float)) == hipSuccess);
            CHECK(sgx_shard_execute_chunked(plan, comm, dx, B, N, N, chunks == 4 ? dout : NULL, dg, chunks, NULL) == SGX_OK);
            CHECK(hipDeviceSynchronize() == hipSuccess);
            CHECK(hipMemcpy(g, dg, B * nb * nf * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess);
            CHECK(memcmp(g, out, B * nb * nf * sizeof(float)) == 0);
        }
        CHECK(sgx_shard_execute_chunked(plan, comm, dx, B, N, N, NULL, dg, 65, NULL) == SGX_INVALID_INPUT);
        free(g);
        (void)hipFree(dg);
        sgx_comm_destroy(comm);
    }
    /* wrong output size -> DimensionMismatch, and the message says so */
    CHECK(sgx_execute(plan, dx, B, N, N, dout, B * nb * nf - 1, SGX_MEM_DEVICE, NULL) == SGX_DIM_MISMATCH);
    CHECK(strstr(sgx_last_error(plan), "Dimension mismatch") != NULL);
    sgx_plan_destroy(plan);

    /* the other Sample type through the same entry points (src/sample.rs:23-86): f64, host pointers, the tuned f64 kernel */
    {
        sgx_params pd = p;
        pd.dtype = SGX_F64;
        sgx_plan *pl64 = NULL;
        CHECK(sgx_plan_create(&pd, &pl64) == SGX_OK);
        CHECK(strcmp(sgx_kernel_name(pl64), "d32x16_f64") == 0);
        double *xd = (double *)malloc(B * N * sizeof(double)), *od = (double *)malloc(B * nb * nf * sizeof(double));
        for (size_t i = 0; i < B * N; ++i) xd[i] = (double)x[i];
        CHECK(sgx_execute(pl64, xd, B, N, N, od, B * nb * nf, SGX_MEM_HOST, NULL) == SGX_OK);
        double w64 = 0.0, p64 = 0.0;
        for (int q = 0; q < 3; ++q) {
            const size_t f = q == 2 ? nf - 1 : frames[q], b = 2;
            for (size_t k = 0; k < nb; k += 37) {
                double re = 0.0, im = 0.0;
                for (size_t i = 0; i < n_fft; ++i) {
                    const long long si = (long long)(f * hop + i) - (long long)pad;
                    const double v = (si < 0 || si >= (long long)N) ? 0.0 : xd[b * N + (size_t)si];
                    const double w = 0.5 - 0.5 * cos(2.0 * kPi * (double)i / (double)(n_fft - 1));
                    const double a = -2.0 * kPi * (double)((i * k) % n_fft) / (double)n_fft;
                    re += v * w * cos(a);
                    im += v * w * sin(a);
                }
                const double ref = re * re + im * im, got = od[(b * nb + k) * nf + f];
                if (fabs(got - ref) > w64) w64 = fabs(got - ref);
                if (ref > p64) p64 = ref;
            }
        }
        printf("f64 linear power: max abs err %.3e (peak %.3e)\n", w64, p64);
        CHECK(w64 <= 1e-10 * p64);
        free(xd);
        free(od);
        sgx_plan_destroy(pl64);
    }

    /* complex STFT -> inverse STFT round trip, all on the device */
    p.amp_scale = SGX_AMP_COMPLEX;
    CHECK(sgx_plan_create(&p, &plan) == SGX_OK);
    float *dspec = NULL, *dy = NULL;
    size_t ny = 0;
    CHECK(sgx_istft_length(plan, nf, &ny) == SGX_OK && ny <= N && ny + hop > N);
    CHECK(hipMalloc((void **)&dspec, B * nb * nf * 2 * sizeof(float)) == hipSuccess);
    CHECK(hipMalloc((void **)&dy, B * ny * sizeof(float)) == hipSuccess);
    CHECK(sgx_execute(plan, dx, B, N, N, dspec, B * nb * nf * 2, SGX_MEM_DEVICE, NULL) == SGX_OK);
    CHECK(sgx_istft(plan, dspec, B, nb, nf, dy, B * ny, SGX_MEM_DEVICE, NULL) == SGX_OK);
    CHECK(hipDeviceSynchronize() == hipSuccess);
    float *y = (float *)malloc(B * ny * sizeof(float));
    CHECK(hipMemcpy(y, dy, B * ny * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess);
    double rt = 0.0;
    for (size_t b = 0; b < B; ++b)
        for (size_t i = n_fft; i + n_fft < ny; ++i) {
            const double d = fabs((double)y[b * ny + i] - (double)x[b * N + i]);
            if (d > rt) rt = d;
        }
    printf("stft -> istft round trip: max abs err %.3e\n", rt);
    CHECK(rt < 5e-6);
    sgx_plan_destroy(plan);
    (void)hipFree(dout); (void)hipFree(dspec); (void)hipFree(dy);
    free(out); free(y);
    /* a prime frame length (the reference plans every length, src/fft_backend.rs:376-385): the chirp-z kernel through the same
     * entry points — complex STFT against a direct DFT, then stft -> istft back to the samples */
    {
        const size_t pn = 251, phop = 63, ppad = pn / 2;
        sgx_params q = p;
        q.n_fft = (uint32_t)pn; q.hop_size = (uint32_t)phop; q.amp_scale = SGX_AMP_COMPLEX;
        sgx_plan *pp = NULL;
        CHECK(sgx_plan_create(&q, &pp) == SGX_OK);
        CHECK(strcmp(sgx_kernel_name(pp), "bluestein") == 0);
        size_t pb = 0, pf = 0;
        CHECK(sgx_output_shape(pp, N, &pb, &pf) == SGX_OK && pb == pn / 2 + 1);
        float *dsp = NULL, *dyy = NULL;
        CHECK(hipMalloc((void **)&dsp, B * pb * pf * 2 * sizeof(float)) == hipSuccess);
        CHECK(sgx_execute(pp, dx, B, N, N, dsp, B * pb * pf * 2, SGX_MEM_DEVICE, NULL) == SGX_OK);
        float *sp = (float *)malloc(B * pb * pf * 2 * sizeof(float));
        CHECK(hipMemcpy(sp, dsp, B * pb * pf * 2 * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess);
        double werr = 0.0, wpeak = 0.0;
        const size_t fr[3] = {0, 11, pf - 1};
        for (int c = 0; c < 3; ++c)
            for (size_t k = 0; k < pb; k += 13) {
                double re = 0.0, im = 0.0;
                for (size_t i = 0; i < pn; ++i) {
                    const long long sidx = (long long)(fr[c] * phop + i) - (long long)ppad;
                    const double v = (sidx >= 0 && sidx < (long long)N) ? (double)x[2 * N + (size_t)sidx] : 0.0;
                    const double w = (double)(float)(0.5 - 0.5 * cos(2.0 * kPi * (double)i / (double)(pn - 1)));
                    const double a = -2.0 * kPi * (double)((i * k) % pn) / (double)pn;
                    re += v * w * cos(a);
                    im += v * w * sin(a);
                }
                const float *g = sp + ((2 * pb + k) * pf + fr[c]) * 2;
                const double d = hypot((double)g[0] - re, (double)g[1] - im), m = hypot(re, im);
                if (d > werr) werr = d;
                if (m > wpeak) wpeak = m;
            }
        printf("n_fft 251 (chirp-z) complex STFT: max abs err %.3e (peak %.3e)\n", werr, wpeak);
        CHECK(werr <= 1e-4 * wpeak);
        size_t pny = 0;
        CHECK(sgx_istft_length(pp, pf, &pny) == SGX_OK);
        CHECK(hipMalloc((void **)&dyy, B * pny * sizeof(float)) == hipSuccess);
        CHECK(sgx_istft(pp, dsp, B, pb, pf, dyy, B * pny, SGX_MEM_DEVICE, NULL) == SGX_OK);
        float *yy = (float *)malloc(B * pny * sizeof(float));
        CHECK(hipMemcpy(yy, dyy, B * pny * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess);
        double prt = 0.0;
        for (size_t b = 0; b < B; ++b)
            for (size_t i = pn; i + pn < pny && i + pn < N; ++i) {
                const double d = fabs((double)yy[b * pny + i] - (double)x[b * N + i]);
                if (d > prt) prt = d;
            }
        printf("n_fft 251 stft -> istft round trip: max abs err %.3e\n", prt);
        CHECK(prt < 5e-6);
        sgx_plan_destroy(pp);
        (void)hipFree(dsp); (void)hipFree(dyy);
        free(sp); free(yy);
    }
    (void)hipFree(dx);
    free(x);
    printf("c_abi device smoke passed\n");
    return 0;
}
