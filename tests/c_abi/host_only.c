/* Plain-C consumer of include/spectro_hip.h: proves the boundary is a C ABI (no C++ types, compiles as C99 with -Wall
 * -Werror) and exercises the host-only entry points (no GPU needed): validation texts, shapes, axes, window, Mel CSR,
 * sharding, inverse-length, error behaviour of compute calls on a host-only plan.  Built and run by tests/test_c_abi.py. */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "spectro_hip.h"

#define CHECK(c)                                                     \
    do {                                                             \
        if (!(c)) {                                                  \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                \
        }                                                            \
    } while (0)

int main(void) {
    CHECK(sgx_abi_version() == SGX_ABI_VERSION);
    sgx_params p;
    memset(&p, 0, sizeof p);
    p.n_fft = 1024; p.hop_size = 256; p.centre = 1; p.window_kind = SGX_WIN_HANNING; p.sample_rate_hz = 16000.0;
    p.freq_scale = SGX_FREQ_MEL; p.n_mels = 80; p.f_min = 0.0; p.f_max = 8000.0; p.mel_norm = SGX_MELNORM_NONE;
    p.amp_scale = SGX_AMP_DECIBELS; p.has_log_params = 1; p.floor_db = -80.0; p.dtype = SGX_F32; p.device = -2;
    sgx_plan *plan = NULL;
    CHECK(sgx_plan_create(&p, &plan) == SGX_OK && plan != NULL);
    size_t nb = 0, nf = 0;
    CHECK(sgx_output_shape(plan, 160000, &nb, &nf) == SGX_OK && nb == 80 && nf == 626);
    double w[1024], freqs[80], times[4];
    CHECK(sgx_window(plan, w) == SGX_OK && w[0] == 0.0 && fabs(w[511] - 1.0) < 1e-5);
    CHECK(sgx_axes(plan, 4, freqs, times) == SGX_OK && times[1] == 256.0 / 16000.0 && freqs[0] > 0.0);
    size_t nnz = 0;
    CHECK(sgx_mel_weights(plan, &nnz, NULL, NULL, NULL) == SGX_OK && nnz == 1001); /* SURVEY.md §8 a8 */
    size_t n_inv = 0;
    CHECK(sgx_istft_length(plan, 626, &n_inv) == SGX_OK && n_inv == 160000);
    /* no CPU fallback: compute on a host-only plan is a backend error with a message */
    float x[8] = {0}, y[8];
    CHECK(sgx_execute(plan, x, 1, 8, 8, y, 8, SGX_MEM_HOST, NULL) == SGX_DIM_MISMATCH);
    size_t want = 0, got = 0; /* DimensionMismatch { expected, got } (src/error.rs:19-21) as numbers, not only as text */
    CHECK(sgx_last_dim_mismatch(plan, &want, &got) == SGX_OK && want == 80 && got == 8);
    CHECK(sgx_plan_device(plan) == -2);
    CHECK(sgx_reserve(plan, 4, 16000, 1, 0) == SGX_BACKEND); /* host-only plan: nothing to reserve on */
    float big[80];
    CHECK(sgx_execute(plan, x, 1, 8, 8, big, 80, SGX_MEM_HOST, NULL) == SGX_BACKEND);
    CHECK(strstr(sgx_last_error(plan), "hip") != NULL);
    sgx_plan_destroy(plan);
    /* validation mirrors the reference's constructors */
    p.hop_size = 2048;
    CHECK(sgx_plan_create(&p, &plan) == SGX_INVALID_INPUT && plan == NULL);
    CHECK(strstr(sgx_last_create_error(), "hop_size must be <= n_fft") != NULL);
    size_t start = 0, count = 0;
    CHECK(sgx_shard_range(8192, 8, 3, &start, &count) == SGX_OK && start == 3072 && count == 1024);
    CHECK(sgx_shard_range(10, 4, 1, &start, &count) == SGX_OK && start == 3 && count == 3);
    /* the multi-GPU entry points fail cleanly without a device / communicator */
    sgx_comm *comm = NULL;
    unsigned char id[SGX_COMM_ID_BYTES];
    memset(id, 0, sizeof id);
    CHECK(sgx_comm_create(id, 0, 0, -1, &comm) == SGX_INVALID_INPUT && comm == NULL);
    CHECK(sgx_comm_adopt(NULL, 2, 0, -1, &comm) == SGX_INVALID_INPUT && comm == NULL);
    CHECK(strlen(sgx_comm_last_error(NULL)) > 0);
    CHECK(sgx_gather(NULL, x, y, 8, 1, SGX_F32, NULL) == SGX_INVALID_INPUT);
    CHECK(sgx_shard_execute(NULL, NULL, x, 8, 8, 8, y, NULL, NULL) == SGX_INVALID_INPUT);
    sgx_comm_destroy(NULL);
    sgx_fft2d *f2 = NULL;
    CHECK(sgx_fft2d_create(0, 8, SGX_F32, -2, &f2) == SGX_INVALID_INPUT && f2 == NULL);
    CHECK(sgx_fft2d_create(16, 16, SGX_F32, -2, &f2) == SGX_OK && f2 != NULL);
    CHECK(sgx_fft2d_device(f2) == -2);
    sgx_fft2d_destroy(f2);
    printf("c_abi host-only checks passed\n");
    return 0;
}
