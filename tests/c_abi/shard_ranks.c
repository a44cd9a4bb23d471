/* shard_ranks.c — the multi-rank paths of the C ABI (sgx_comm_create, sgx_shard_execute, sgx_gather) executed with 2 and 3 ranks on
 * ONE GPU: each rank is a host thread with its own plan, communicator and stream; the collectives are served by fake_rccl.c
 * (linked into this executable and exported with -rdynamic, so that the library's dlsym lookup finds it as "the host's RCCL").
 * Every case compares what every rank gathered with a single launch over the whole batch, bit for bit:
 *   worlds 2 and 3; batches that divide evenly (one ncclAllGather) and ragged ones (a group of ncclBroadcast, incl. a batch smaller
 *   than the world, i.e. ranks with an empty shard); shard computed in place (straight into the rank's slice) and through a separate
 *   shard buffer; per-bin (linear power), filterbank (Mel-80 dB) and complex outputs; sgx_shard_execute_chunked with K = 1, 4 and 7
 *   chunks (the exchange of chunk k on the communicator's own stream while chunk k + 1 computes).
 * Built (hipcc, C mode) and run by tests/test_c_abi.py::test_shard_execute_multi_rank on the GPU box. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spectro_hip.h"

extern int fake_rccl_allgathers, fake_rccl_broadcasts, fake_rccl_groups;

#define CHECK(c)                                                          \
    do {                                                                  \
        if (!(c)) {                                                       \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                     \
        }                                                                 \
    } while (0)
#define TCHECK(c)                                                                          \
    do {                                                                                   \
        if (!(c)) {                                                                        \
            fprintf(stderr, "FAILED rank %d %s:%d: %s\n", a->rank, __FILE__, __LINE__, #c); \
            a->failed = 1;                                                                 \
            return NULL;                                                                   \
        }                                                                                  \
    } while (0)

enum { N = 9000 };

typedef struct {
    int world, rank, failed;
    size_t batch;
    int separate; /* 1: compute into a shard buffer, then gather; 0: straight into the rank's slice of the gathered buffer */
    int chunks;   /* 0: sgx_shard_execute; K >= 1: sgx_shard_execute_chunked with K chunks (gathers on the communicator's own stream) */
    const sgx_params *params;
    const unsigned char *id;
    const float *host_x;  /* [batch][N] */
    const float *ref;     /* single-launch output, [batch][per_item] */
    size_t per_item;      /* elements of T per signal (2 per complex value) */
    pthread_barrier_t *bar;
} rank_arg;

static void *rank_main(void *vp) {
    rank_arg *a = (rank_arg *)vp;
    TCHECK(hipSetDevice(0) == hipSuccess);
    hipStream_t s = NULL;
    TCHECK(hipStreamCreate(&s) == hipSuccess);
    sgx_plan *plan = NULL;
    TCHECK(sgx_plan_create(a->params, &plan) == SGX_OK);
    sgx_comm *comm = NULL;
    TCHECK(sgx_comm_create(a->id, a->world, a->rank, 0, &comm) == SGX_OK && comm != NULL);
    size_t start = 0, count = 0;
    TCHECK(sgx_shard_range(a->batch, a->world, a->rank, &start, &count) == SGX_OK);
    float *dx = NULL, *dshard = NULL, *dg = NULL;
    const size_t gbytes = a->batch * a->per_item * sizeof(float);
    TCHECK(hipMalloc((void **)&dx, (count ? count : 1) * N * sizeof(float)) == hipSuccess);
    TCHECK(hipMalloc((void **)&dg, gbytes) == hipSuccess);
    TCHECK(hipMemsetAsync(dg, 0xff, gbytes, s) == hipSuccess); /* poison: a slice nobody wrote shows up as NaN */
    if (count) TCHECK(hipMemcpyAsync(dx, a->host_x + start * N, count * N * sizeof(float), hipMemcpyHostToDevice, s) == hipSuccess);
    if (a->separate) {
        TCHECK(hipMalloc((void **)&dshard, (count ? count : 1) * a->per_item * sizeof(float)) == hipSuccess);
    }
    /* a rank with an empty shard still takes part in the collective */
    const sgx_status st = a->chunks
                              ? sgx_shard_execute_chunked(plan, comm, count ? dx : NULL, a->batch, N, N, a->separate ? dshard : NULL, dg, a->chunks, s)
                              : sgx_shard_execute(plan, comm, count ? dx : NULL, a->batch, N, N, a->separate ? dshard : NULL, dg, s);
    if (st != SGX_OK) fprintf(stderr, "rank %d: sgx_shard_execute -> %d: %s\n", a->rank, (int)st, sgx_comm_last_error(comm));
    TCHECK(st == SGX_OK);
    TCHECK(hipStreamSynchronize(s) == hipSuccess);
    float *g = (float *)malloc(gbytes);
    TCHECK(hipMemcpy(g, dg, gbytes, hipMemcpyDeviceToHost) == hipSuccess);
    if (memcmp(g, a->ref, gbytes) != 0) {
        size_t bad = 0;
        while (bad < a->batch * a->per_item && memcmp(&g[bad], &a->ref[bad], sizeof(float)) == 0) ++bad;
        fprintf(stderr, "rank %d of %d (batch %zu, separate %d): gathered output differs from the single launch at element %zu (signal %zu)\n",
                a->rank, a->world, a->batch, a->separate, bad, bad / a->per_item);
        a->failed = 1;
    }
    free(g);
    pthread_barrier_wait(a->bar); /* no rank tears down while another is still inside a collective */
    sgx_comm_destroy(comm);
    sgx_plan_destroy(plan);
    (void)hipFree(dx);
    (void)hipFree(dg);
    if (dshard) (void)hipFree(dshard);
    (void)hipStreamDestroy(s);
    return NULL;
}

static int run_case_k(const sgx_params *p, const float *x, size_t batch, int world, int separate, int chunks, const char *what) {
    sgx_plan *plan = NULL;
    CHECK(sgx_plan_create(p, &plan) == SGX_OK);
    size_t nb = 0, nf = 0;
    CHECK(sgx_output_shape(plan, N, &nb, &nf) == SGX_OK);
    const size_t per_item = nb * nf * (p->amp_scale == SGX_AMP_COMPLEX ? 2 : 1);
    float *dx = NULL, *dout = NULL;
    CHECK(hipMalloc((void **)&dx, batch * N * sizeof(float)) == hipSuccess);
    CHECK(hipMalloc((void **)&dout, batch * per_item * sizeof(float)) == hipSuccess);
    CHECK(hipMemcpy(dx, x, batch * N * sizeof(float), hipMemcpyHostToDevice) == hipSuccess);
    CHECK(sgx_execute(plan, dx, batch, N, N, dout, batch * per_item, SGX_MEM_DEVICE, NULL) == SGX_OK);
    CHECK(hipDeviceSynchronize() == hipSuccess);
    float *ref = (float *)malloc(batch * per_item * sizeof(float));
    CHECK(hipMemcpy(ref, dout, batch * per_item * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess);
    sgx_plan_destroy(plan);
    (void)hipFree(dx);
    (void)hipFree(dout);

    unsigned char id[SGX_COMM_ID_BYTES];
    CHECK(sgx_comm_unique_id(id) == SGX_OK);
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)world);
    pthread_t th[8];
    rank_arg args[8];
    const int ag0 = fake_rccl_allgathers, bc0 = fake_rccl_broadcasts, gr0 = fake_rccl_groups;
    for (int r = 0; r < world; ++r) {
        rank_arg a = {world, r, 0, batch, separate, chunks, p, id, x, ref, per_item, &bar};
        args[r] = a;
        CHECK(pthread_create(&th[r], NULL, rank_main, &args[r]) == 0);
    }
    int failed = 0;
    for (int r = 0; r < world; ++r) {
        pthread_join(th[r], NULL);
        failed |= args[r].failed;
    }
    pthread_barrier_destroy(&bar);
    free(ref);
    const int ag = fake_rccl_allgathers - ag0, bc = fake_rccl_broadcasts - bc0, gr = fake_rccl_groups - gr0;
    printf("%-14s world %d batch %zu %-9s chunks %d: %s  (ncclAllGather x%d, ncclBroadcast x%d in %d groups)\n", what, world, batch,
           separate ? "separate" : "in-place", chunks, failed ? "FAILED" : "ok", ag, bc, gr);
    CHECK(!failed);
    if (chunks > 1) {
        /* one group per chunk and rank; in it one broadcast per rank that holds a piece [k n / K, (k + 1) n / K) of that chunk */
        int pieces = 0;
        for (int r = 0; r < world; ++r) {
            size_t rs = 0, rc = 0;
            CHECK(sgx_shard_range(batch, world, r, &rs, &rc) == SGX_OK);
            for (int k = 0; k < chunks; ++k) pieces += rc * (size_t)(k + 1) / (size_t)chunks > rc * (size_t)k / (size_t)chunks;
        }
        CHECK(ag == 0 && gr == chunks * world && bc == pieces * world);
    } else if (batch % (size_t)world == 0) {
        CHECK(ag == world && bc == 0); /* equal shards: one all-gather per rank */
    } else {
        size_t roots = batch < (size_t)world ? batch : (size_t)world; /* ranks with a non-empty shard */
        CHECK(ag == 0 && gr == world && bc == (int)(roots * (size_t)world));
    }
    return 0;
}

static int run_case(const sgx_params *p, const float *x, size_t batch, int world, int separate, const char *what) {
    return run_case_k(p, x, batch, world, separate, 0, what);
}

int main(void) {
    CHECK(hipSetDevice(0) == hipSuccess);
    const size_t maxb = 7;
    float *x = (float *)malloc(maxb * N * sizeof(float));
    unsigned s = 2468u;
    for (size_t i = 0; i < maxb * N; ++i) {
        s = s * 1664525u + 1013904223u;
        x[i] = (float)((double)(s >> 8) / 16777216.0 - 0.5) + 0.4f * (float)sin(2.0 * 3.14159265358979323846 * (200.0 + 97.0 * (double)(i / N)) * (double)(i % N) / 16000.0);
    }
    sgx_params lin;
    memset(&lin, 0, sizeof lin);
    lin.n_fft = 1024; lin.hop_size = 256; lin.centre = 1; lin.window_kind = SGX_WIN_HANNING; lin.sample_rate_hz = 16000.0;
    lin.freq_scale = SGX_FREQ_LINEAR; lin.amp_scale = SGX_AMP_POWER; lin.dtype = SGX_F32; lin.device = 0;
    sgx_params mel = lin;
    mel.freq_scale = SGX_FREQ_MEL; mel.n_mels = 80; mel.f_min = 0.0; mel.f_max = 8000.0; mel.amp_scale = SGX_AMP_DECIBELS;
    mel.has_log_params = 1; mel.floor_db = -80.0;
    sgx_params cpx = lin;
    cpx.amp_scale = SGX_AMP_COMPLEX;
    for (int world = 2; world <= 3; ++world)
        for (int separate = 0; separate <= 1; ++separate) {
            if (run_case(&lin, x, 6, world, separate, "linear power")) return 1; /* equal shards */
            if (run_case(&lin, x, 7, world, separate, "linear power")) return 1; /* ragged */
            if (run_case(&mel, x, 7, world, separate, "Mel-80 dB")) return 1;
            if (run_case(&cpx, x, 5, world, separate, "complex STFT")) return 1;
        }
    /* sgx_shard_execute_chunked: K = 1 (the plain path) and K = 4 (compute on the caller's stream, each chunk's exchange on the
     * communicator's stream behind an event), equal and ragged shards, chunks with no signals (7 signals over 3 ranks x 4 chunks), ranks
     * with an empty shard — bit for bit against one launch */
    for (int world = 2; world <= 3; ++world)
        for (int separate = 0; separate <= 1; ++separate) {
            if (run_case_k(&lin, x, 6, world, separate, 1, "linear power")) return 1;
            if (run_case_k(&lin, x, 6, world, separate, 4, "linear power")) return 1;
            if (run_case_k(&mel, x, 7, world, separate, 4, "Mel-80 dB")) return 1;
            if (run_case_k(&cpx, x, 5, world, separate, 4, "complex STFT")) return 1;
        }
    if (run_case_k(&mel, x, 2, 3, 0, 4, "Mel-80 dB")) return 1;
    if (run_case_k(&mel, x, 7, 2, 0, 7, "Mel-80 dB")) return 1;
    if (run_case(&mel, x, 2, 3, 0, "Mel-80 dB")) return 1; /* batch < world: rank 2 has nothing to compute and still gathers */
    if (run_case(&lin, x, 1, 2, 1, "linear power")) return 1;
    free(x);
    printf("c_abi multi-rank shard test passed\n");
    return 0;
}
