// shard.hip — multi-GPU entry points of the C ABI: utterance sharding with an optional RCCL all-gather of the output shards
// (BASELINE config 4).  One process (or thread) per GPU; each rank runs the single-GPU plan on its contiguous
// block of utterances (sgx_shard_range) and, if asked, every rank receives the whole [batch][n_bins][n_frames] output.
//
// RCCL is resolved at run time (dlsym on the process first, then dlopen of librccl.so.1): the library has no link-time
// dependency on it, a host that already carries an RCCL (a Rust binary linked against librccl, PyTorch's bundled copy) gets
// THAT copy — which is what makes sgx_comm_adopt of the host's own ncclComm_t valid — and single-GPU users never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "sgx_internal.h"

using namespace sgx;

struct sgx_comm {
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0;
    int device = -1;
    bool owned = false;  // created by sgx_comm_create (destroyed with it) vs adopted from the host
    std::string err;
    // sgx_shard_execute_chunked: the stream its gathers travel on and the events that order it against the caller's stream
    hipStream_t gstream = nullptr;
    std::vector<hipEvent_t> chunk_done;  // chunk k computed (recorded on the caller's stream)
    hipEvent_t gathered = nullptr;       // last gather issued (recorded on gstream)
};

namespace {

struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why;
};

const Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = RTLD_DEFAULT;
#ifdef SGX_NO_RCCL  // test build only (tests/test_abi.py): a host without any RCCL — every candidate name fails to load
        constexpr bool probe_process = false;
        const char *const names[] = {"librccl-absent-for-this-test.so.1"};
#else
        constexpr bool probe_process = true;
        const char *const names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
#endif
        if (!probe_process || !dlsym(RTLD_DEFAULT, "ncclAllGather")) {
            h = nullptr;
            for (const char *name : names)
                if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
            if (!h) {
                const char *de = dlerror();  // one call: dlerror() clears the message it returns
                r.why = std::string("RCCL not found: ") + (de ? de : "dlopen failed");
                return;
            }
        }
        auto sym = [&](const char *n) { return dlsym(h, n); };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.Broadcast && r.GroupStart && r.GroupEnd;
        if (!r.ok) r.why = "RCCL library lacks a required symbol";
    });
    return r;
}

thread_local std::string g_comm_err;

sgx_status comm_fail(sgx_comm *c, sgx_status st, const std::string &msg) {
    (c ? c->err : g_comm_err) = msg;
    return st;
}

std::string nccl_text(ncclResult_t e) {
    const Rccl &r = rccl();
    return std::string("hip -- FFT backend error: RCCL: ") + (r.GetErrorString ? r.GetErrorString(e) : "error ") + " (" + std::to_string(int(e)) + ")";
}

#define SGX_NCCL(c, call)                                                \
    do {                                                                 \
        ncclResult_t e_ = (call);                                        \
        if (e_ != ncclSuccess) return comm_fail(c, SGX_BACKEND, nccl_text(e_)); \
    } while (0)

}  // namespace

extern "C" {

const char *sgx_comm_last_error(const sgx_comm *c) { return c ? c->err.c_str() : g_comm_err.c_str(); }

sgx_status sgx_comm_unique_id(void *id128) {
    if (!id128) return comm_fail(nullptr, SGX_INVALID_INPUT, "Invalid input: null argument");
    const Rccl &r = rccl();
    if (!r.ok) return comm_fail(nullptr, SGX_BACKEND, "hip -- FFT backend error: " + r.why);
    static_assert(sizeof(ncclUniqueId) == SGX_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    SGX_NCCL(nullptr, r.GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof id);
    return SGX_OK;
}

sgx_status sgx_comm_create(const void *id128, int32_t world_size, int32_t rank, int32_t device, sgx_comm **out) {
    if (out) *out = nullptr;
    if (!id128 || !out || world_size <= 0 || rank < 0 || rank >= world_size)
        return comm_fail(nullptr, SGX_INVALID_INPUT, "Invalid input: bad communicator arguments");
    const Rccl &r = rccl();
    if (!r.ok) return comm_fail(nullptr, SGX_BACKEND, "hip -- FFT backend error: " + r.why);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return comm_fail(nullptr, SGX_BACKEND, "hip -- FFT backend error: no HIP device available");
    int dev = device;
    if (dev == -1 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev < 0 || dev >= ndev) return comm_fail(nullptr, SGX_INVALID_INPUT, "Invalid input: device ordinal out of range");
    sgx_comm *c = new (std::nothrow) sgx_comm();
    if (!c) return comm_fail(nullptr, SGX_INTERNAL, "Internal error: out of memory");
    c->world = world_size; c->rank = rank; c->device = dev; c->owned = true;
    DeviceGuard dg;
    if (dg.enter(dev) != hipSuccess) { delete c; return comm_fail(nullptr, SGX_BACKEND, "hip -- FFT backend error: hipSetDevice failed"); }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    ncclResult_t e = r.CommInitRank(&c->comm, world_size, id, rank);
    if (e != ncclSuccess) { delete c; return comm_fail(nullptr, SGX_BACKEND, nccl_text(e)); }
    *out = c;
    return SGX_OK;
}

sgx_status sgx_comm_adopt(void *nccl_comm, int32_t world_size, int32_t rank, int32_t device, sgx_comm **out) {
    if (out) *out = nullptr;
    if (!nccl_comm || !out || world_size <= 0 || rank < 0 || rank >= world_size)
        return comm_fail(nullptr, SGX_INVALID_INPUT, "Invalid input: bad communicator arguments");
    const Rccl &r = rccl();
    if (!r.ok) return comm_fail(nullptr, SGX_BACKEND, "hip -- FFT backend error: " + r.why);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return comm_fail(nullptr, SGX_BACKEND, "hip -- FFT backend error: no HIP device available");
    int dev = device;
    if (dev == -1 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev < 0 || dev >= ndev) return comm_fail(nullptr, SGX_INVALID_INPUT, "Invalid input: device ordinal out of range");
    sgx_comm *c = new (std::nothrow) sgx_comm();
    if (!c) return comm_fail(nullptr, SGX_INTERNAL, "Internal error: out of memory");
    c->comm = static_cast<ncclComm_t>(nccl_comm);
    c->world = world_size; c->rank = rank; c->device = dev; c->owned = false;
    *out = c;
    return SGX_OK;
}

void sgx_comm_destroy(sgx_comm *c) {
    if (!c) return;
    {
        DeviceGuard dg;
        (void)dg.enter(c->device);
        if (c->gstream) (void)hipStreamSynchronize(c->gstream);
        for (hipEvent_t e : c->chunk_done) (void)hipEventDestroy(e);
        if (c->gathered) (void)hipEventDestroy(c->gathered);
        if (c->gstream) (void)hipStreamDestroy(c->gstream);
        if (c->owned && c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    }
    delete c;
}

// Every rank contributes counts[rank] elements (its shard) and receives all shards back to back, rank order, in `recv`.
// Equal shards: one ncclAllGather.  Ragged shards (batch % world != 0): one grouped ncclBroadcast per rank, each into its slice.
sgx_status sgx_gather(sgx_comm *c, const void *send, void *recv, size_t global_batch, size_t elems_per_item, int32_t dtype,
                      void *hip_stream) {
    if (!c) return SGX_INVALID_INPUT;
    if (!send || !recv || global_batch == 0 || elems_per_item == 0) return comm_fail(c, SGX_INVALID_INPUT, "Invalid input: null or empty buffer");
    if (dtype != SGX_F32 && dtype != SGX_F64) return comm_fail(c, SGX_INVALID_INPUT, "Invalid input: dtype must be f32 or f64");
    const Rccl &r = rccl();
    if (!r.ok) return comm_fail(c, SGX_BACKEND, "hip -- FFT backend error: " + r.why);
    if (!c->comm) return comm_fail(c, SGX_INVALID_INPUT, "Invalid input: communicator has no RCCL handle");
    const ncclDataType_t nt = dtype == SGX_F64 ? ncclFloat64 : ncclFloat32;
    const size_t elem = dtype == SGX_F64 ? 8 : 4;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    DeviceGuard dg;
    if (dg.enter(c->device) != hipSuccess) return comm_fail(c, SGX_BACKEND, "hip -- FFT backend error: hipSetDevice failed");
    size_t start = 0, count = 0;
    (void)sgx_shard_range(global_batch, c->world, c->rank, &start, &count);
    if (global_batch % size_t(c->world) == 0) {
        SGX_NCCL(c, r.AllGather(send, recv, count * elems_per_item, nt, c->comm, s));
        return SGX_OK;
    }
    SGX_NCCL(c, r.GroupStart());
    for (int root = 0; root < c->world; ++root) {
        size_t rs = 0, rc = 0;
        (void)sgx_shard_range(global_batch, c->world, root, &rs, &rc);
        if (rc == 0) continue;
        void *slice = static_cast<char *>(recv) + rs * elems_per_item * elem;
        ncclResult_t e = r.Broadcast(root == c->rank ? send : slice, slice, rc * elems_per_item, nt, root, c->comm, s);
        if (e != ncclSuccess) { (void)r.GroupEnd(); return comm_fail(c, SGX_BACKEND, nccl_text(e)); }
    }
    SGX_NCCL(c, r.GroupEnd());
    return SGX_OK;
}

sgx_status sgx_shard_execute(sgx_plan *plan, sgx_comm *c, const void *shard_samples, size_t global_batch, size_t n_samples,
                             size_t sample_stride, void *shard_out, void *gathered_out, void *hip_stream) {
    if (!plan || !c) return SGX_INVALID_INPUT;
    size_t start = 0, count = 0;
    if (sgx_shard_range(global_batch, c->world, c->rank, &start, &count) != SGX_OK) return comm_fail(c, SGX_INVALID_INPUT, "Invalid input: bad shard");
    size_t nb = 0, nf = 0;
    sgx_status st = sgx_output_shape(plan, n_samples, &nb, &nf);
    if (st != SGX_OK) return comm_fail(c, st, sgx_last_error(plan));
    const size_t per_item = nb * nf * (plan->out_mode == OUT_COMPLEX ? 2 : 1);
    if (sgx_plan_device(plan) != c->device) return comm_fail(c, SGX_INVALID_INPUT, "Invalid input: plan and communicator are on different devices");
    void *dst = shard_out;
    if (!dst && gathered_out) dst = static_cast<char *>(gathered_out) + start * per_item * plan->elem;  // compute straight into this rank's slice
    if (count > 0) {
        if (!dst || !shard_samples) return comm_fail(c, SGX_INVALID_INPUT, "Invalid input: null buffer");
        st = sgx_execute(plan, shard_samples, count, n_samples, sample_stride, dst, count * per_item, SGX_MEM_DEVICE, hip_stream);
        if (st != SGX_OK) return comm_fail(c, st, sgx_last_error(plan));
    }
    if (!gathered_out) return SGX_OK;  // compute-only sharding: no data-path collective
    return sgx_gather(c, dst ? dst : gathered_out, gathered_out, global_batch, per_item, plan->dtype, hip_stream);
}

// The same job with the gather pipelined behind the compute ("chunked and overlapped with compute"): the
// rank's shard is cut into `chunks` runs of signals; chunk k is computed on the caller's stream, and its exchange — one grouped
// ncclBroadcast per rank holding a piece of chunk k, each into that piece's own place in `gathered_out` — is issued on the
// communicator's second stream behind an event, so it travels over xGMI while chunk k + 1 computes.  (ncclAllGather cannot express
// it: its receive layout is rank-major per call, the result's is rank-major per SHARD.)  No host synchronisation; the caller's
// stream waits for the last exchange, so "asynchronous on hip_stream" holds as for sgx_shard_execute.  A signal's bits do not
// depend on the launch it is computed in, so the gathered result equals the unchunked one bit for bit (tests/c_abi/shard_ranks.c).
sgx_status sgx_shard_execute_chunked(sgx_plan *plan, sgx_comm *c, const void *shard_samples, size_t global_batch, size_t n_samples,
                                     size_t sample_stride, void *shard_out, void *gathered_out, int32_t chunks, void *hip_stream) {
    if (!plan || !c) return SGX_INVALID_INPUT;
    if (chunks < 1 || chunks > 64) return comm_fail(c, SGX_INVALID_INPUT, "Invalid input: chunks must be in 1..64");
    if (chunks == 1 || !gathered_out)
        return sgx_shard_execute(plan, c, shard_samples, global_batch, n_samples, sample_stride, shard_out, gathered_out, hip_stream);
    const Rccl &r = rccl();
    if (!r.ok) return comm_fail(c, SGX_BACKEND, "hip -- FFT backend error: " + r.why);
    if (!c->comm) return comm_fail(c, SGX_INVALID_INPUT, "Invalid input: communicator has no RCCL handle");
    size_t start = 0, count = 0;
    if (sgx_shard_range(global_batch, c->world, c->rank, &start, &count) != SGX_OK) return comm_fail(c, SGX_INVALID_INPUT, "Invalid input: bad shard");
    size_t nb = 0, nf = 0;
    sgx_status st = sgx_output_shape(plan, n_samples, &nb, &nf);
    if (st != SGX_OK) return comm_fail(c, st, sgx_last_error(plan));
    const size_t per_item = nb * nf * (plan->out_mode == OUT_COMPLEX ? 2 : 1);
    if (sgx_plan_device(plan) != c->device) return comm_fail(c, SGX_INVALID_INPUT, "Invalid input: plan and communicator are on different devices");
    if (count > 0 && !shard_samples) return comm_fail(c, SGX_INVALID_INPUT, "Invalid input: null buffer");
    const ncclDataType_t nt = plan->dtype == SGX_F64 ? ncclFloat64 : ncclFloat32;
    const size_t elem = plan->elem;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    DeviceGuard dg;
    if (dg.enter(c->device) != hipSuccess) return comm_fail(c, SGX_BACKEND, "hip -- FFT backend error: hipSetDevice failed");
    auto hip_ok = [&](hipError_t e) { return e == hipSuccess; };
    if (!c->gstream && !hip_ok(hipStreamCreateWithFlags(&c->gstream, hipStreamNonBlocking)))
        return comm_fail(c, SGX_BACKEND, "hip -- FFT backend error: hipStreamCreate failed");
    if (!c->gathered && !hip_ok(hipEventCreateWithFlags(&c->gathered, hipEventDisableTiming)))
        return comm_fail(c, SGX_BACKEND, "hip -- FFT backend error: hipEventCreate failed");
    while (c->chunk_done.size() < size_t(chunks)) {
        hipEvent_t e = nullptr;
        if (!hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming))) return comm_fail(c, SGX_BACKEND, "hip -- FFT backend error: hipEventCreate failed");
        c->chunk_done.push_back(e);
    }
    // piece k of a shard of n signals: [k n / K, (k + 1) n / K)
    auto piece = [&](size_t n, int k, size_t *lo, size_t *hi) {
        *lo = n * size_t(k) / size_t(chunks);
        *hi = n * size_t(k + 1) / size_t(chunks);
    };
    char *gout = static_cast<char *>(gathered_out);
    char *own = shard_out ? static_cast<char *>(shard_out) : gout + start * per_item * elem;  // where this rank's shard is computed
    // the exchange stream starts behind whatever the caller has queued (e.g. the last consumer of gathered_out)
    if (!hip_ok(hipEventRecord(c->gathered, s)) || !hip_ok(hipStreamWaitEvent(c->gstream, c->gathered, 0)))
        return comm_fail(c, SGX_BACKEND, "hip -- FFT backend error: stream ordering failed");
    // A failure inside the loop (a launch that fails, an RCCL call that fails) must not leave the two streams half-joined: whatever
    // happened, `gathered` is recorded on the exchange stream behind everything queued there and the caller's stream waits for it, so
    // the next call — and the caller's own work on `s` — start from a defined order.  An RCCL group that was opened is
    // always closed.  (What other ranks have already issued for later chunks cannot be withdrawn from here: a failed call means the
    // communicator should be destroyed, which the header says.)
    sgx_status result = SGX_OK;
    std::string why;
    auto fail = [&](sgx_status code, const std::string &msg) {
        if (result == SGX_OK) { result = code; why = msg; }
    };
    for (int k = 0; k < chunks && result == SGX_OK; ++k) {
        size_t lo = 0, hi = 0;
        piece(count, k, &lo, &hi);
        if (hi > lo) {
            const char *xin = static_cast<const char *>(shard_samples) + lo * sample_stride * elem;
            st = sgx_execute(plan, xin, hi - lo, n_samples, sample_stride, own + lo * per_item * elem, (hi - lo) * per_item, SGX_MEM_DEVICE, hip_stream);
            if (st != SGX_OK) { fail(st, sgx_last_error(plan)); break; }
        }
        if (!hip_ok(hipEventRecord(c->chunk_done[k], s)) || !hip_ok(hipStreamWaitEvent(c->gstream, c->chunk_done[k], 0))) {
            fail(SGX_BACKEND, "hip -- FFT backend error: stream ordering failed");
            break;
        }
        ncclResult_t ge = r.GroupStart();
        if (ge != ncclSuccess) { fail(SGX_BACKEND, nccl_text(ge)); break; }
        for (int root = 0; root < c->world; ++root) {
            size_t rs = 0, rc = 0, plo = 0, phi = 0;
            (void)sgx_shard_range(global_batch, c->world, root, &rs, &rc);
            piece(rc, k, &plo, &phi);
            if (phi == plo) continue;  // (every rank skips the same roots)
            char *slice = gout + (rs + plo) * per_item * elem;
            const void *src = root == c->rank ? static_cast<const void *>(own + plo * per_item * elem) : static_cast<const void *>(slice);
            ncclResult_t e = r.Broadcast(src, slice, (phi - plo) * per_item, nt, root, c->comm, c->gstream);
            if (e != ncclSuccess) { fail(SGX_BACKEND, nccl_text(e)); break; }
        }
        ge = r.GroupEnd();  // closes the group on the error path too
        if (ge != ncclSuccess) fail(SGX_BACKEND, nccl_text(ge));
    }
    const bool joined = hip_ok(hipEventRecord(c->gathered, c->gstream)) && hip_ok(hipStreamWaitEvent(s, c->gathered, 0));
    if (result != SGX_OK) return comm_fail(c, result, why);
    if (!joined) return comm_fail(c, SGX_BACKEND, "hip -- FFT backend error: stream ordering failed");
    return SGX_OK;
}

}  // extern "C"
