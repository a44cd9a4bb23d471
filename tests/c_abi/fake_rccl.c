/* fake_rccl.c — TEST-ONLY stand-in for the eight RCCL entry points libspectro_hip.so resolves at run time (shard.hip looks
 * them up with dlsym on the process first, so a test executable that exports them — linked with -rdynamic — is "the host's own
 * RCCL").  The "ranks" of a communicator are host threads of ONE process that share one GPU; a collective is a rendezvous of
 * those threads plus device-to-device copies, which is enough to execute every multi-rank code path of sgx_gather /
 * sgx_shard_execute (equal shards: ncclAllGather; ragged shards: a group of ncclBroadcast with per-root slices, the root's send
 * aliasing its own slice; ranks with an empty shard) on a one-GPU box and to compare the gathered result with a single launch,
 * bit for bit.  It is stricter than RCCL in one way — a collective completes before the call returns — and makes no claim about
 * RCCL's transport.  Types and prototypes come from the real <rccl/rccl.h>, so a signature drift fails to compile. */
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define FAKE_MAX_RANKS 8
#define FAKE_MAX_GROUP 32

typedef struct fake_world {
    char id[sizeof(ncclUniqueId)];
    int nranks, joined;
    pthread_barrier_t bar;
    const void *send[FAKE_MAX_RANKS]; /* published by each rank for the collective in flight */
    struct fake_world *next;
} fake_world;

struct ncclComm {
    fake_world *w;
    int rank;
};

typedef struct {
    int kind; /* 0 all-gather, 1 broadcast */
    const void *send;
    void *recv;
    size_t bytes;
    int root;
    ncclComm_t comm;
    hipStream_t stream;
} fake_op;

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_cv = PTHREAD_COND_INITIALIZER;
static fake_world *g_worlds = NULL;
static unsigned g_next_id = 1;
static __thread int t_group = 0, t_nops = 0;
static __thread fake_op t_ops[FAKE_MAX_GROUP];
/* counters the test reads: how often each entry point really ran */
int fake_rccl_allgathers = 0, fake_rccl_broadcasts = 0, fake_rccl_groups = 0;

static size_t dtype_bytes(ncclDataType_t t) {
    switch (t) {
    case ncclFloat32: return 4;
    case ncclFloat64: return 8;
    default: return 0;
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    pthread_mutex_lock(&g_mu);
    snprintf(id->internal, sizeof id->internal, "fake-rccl-%u", g_next_id++);
    pthread_mutex_unlock(&g_mu);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks <= 0 || nranks > FAKE_MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    pthread_mutex_lock(&g_mu);
    fake_world *w = g_worlds;
    while (w && memcmp(w->id, id.internal, sizeof w->id) != 0) w = w->next;
    if (!w) {
        w = (fake_world *)calloc(1, sizeof *w);
        memcpy(w->id, id.internal, sizeof w->id);
        w->nranks = nranks;
        pthread_barrier_init(&w->bar, NULL, (unsigned)nranks);
        w->next = g_worlds;
        g_worlds = w;
    }
    if (w->nranks != nranks) {
        pthread_mutex_unlock(&g_mu);
        return ncclInvalidArgument;
    }
    w->joined++;
    pthread_cond_broadcast(&g_cv);
    while (w->joined < nranks) pthread_cond_wait(&g_cv, &g_mu); /* collective, like the real one */
    pthread_mutex_unlock(&g_mu);
    struct ncclComm *c = (struct ncclComm *)calloc(1, sizeof *c);
    c->w = w;
    c->rank = rank;
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    free(comm); /* worlds stay until exit: other ranks may still be inside their last barrier */
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake rccl error"; }

static ncclResult_t run_op(const fake_op *op) {
    fake_world *w = op->comm->w;
    const int me = op->comm->rank;
    if (hipStreamSynchronize(op->stream) != hipSuccess) return ncclUnhandledCudaError; /* this rank's producer kernel is done */
    w->send[me] = op->send;
    pthread_barrier_wait(&w->bar); /* every rank has published its send buffer */
    ncclResult_t res = ncclSuccess;
    if (op->kind == 0) {
        for (int r = 0; r < w->nranks; ++r) {
            char *dst = (char *)op->recv + (size_t)r * op->bytes;
            if ((const void *)dst == w->send[r]) continue; /* in place: this rank's own slice */
            if (hipMemcpy(dst, w->send[r], op->bytes, hipMemcpyDeviceToDevice) != hipSuccess) res = ncclUnhandledCudaError;
        }
    } else {
        const void *src = w->send[op->root];
        if (src != (const void *)op->recv && hipMemcpy(op->recv, src, op->bytes, hipMemcpyDeviceToDevice) != hipSuccess)
            res = ncclUnhandledCudaError;
    }
    pthread_barrier_wait(&w->bar); /* nobody reuses a send buffer before every rank has read it */
    return res;
}

static ncclResult_t submit(fake_op op) {
    if (t_group > 0) {
        if (t_nops >= FAKE_MAX_GROUP) return ncclInternalError;
        t_ops[t_nops++] = op;
        return ncclSuccess;
    }
    return run_op(&op);
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
    const size_t es = dtype_bytes(datatype);
    if (!sendbuff || !recvbuff || !comm || es == 0) return ncclInvalidArgument;
    __sync_fetch_and_add(&fake_rccl_allgathers, 1);
    fake_op op = {0, sendbuff, recvbuff, sendcount * es, 0, comm, stream};
    return submit(op);
}

ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm,
                           hipStream_t stream) {
    const size_t es = dtype_bytes(datatype);
    if (!sendbuff || !recvbuff || !comm || es == 0 || root < 0 || root >= comm->w->nranks) return ncclInvalidArgument;
    __sync_fetch_and_add(&fake_rccl_broadcasts, 1);
    fake_op op = {1, sendbuff, recvbuff, count * es, root, comm, stream};
    return submit(op);
}

ncclResult_t ncclGroupStart(void) {
    ++t_group;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd(void) {
    if (t_group <= 0) return ncclInvalidUsage;
    if (--t_group > 0) return ncclSuccess;
    __sync_fetch_and_add(&fake_rccl_groups, 1);
    ncclResult_t res = ncclSuccess;
    /* every rank queued the same operations in the same order (sgx_gather walks the roots in ascending order) */
    for (int i = 0; i < t_nops; ++i) {
        const ncclResult_t e = run_op(&t_ops[i]);
        if (e != ncclSuccess) res = e;
    }
    t_nops = 0;
    return res;
}
