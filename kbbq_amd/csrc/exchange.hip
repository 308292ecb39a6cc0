// exchange.hip -- the multi-GPU exchange steps behind include/kbbq_exchange.h, written on top of the engine's public
// ABI (kbbq_engine.h) and two transports: RCCL (dlopen-ed: ncclSend/ncclRecv groups, ncclAllGather, ncclAllReduce,
// ncclBroadcast on the engine's HIP stream) and the ranks of one process (device-to-device copies + barriers).
// See the header for the protocol; kbbq_amd/dist.py (torch.distributed) is the same protocol from Python.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/kbbq_exchange.h"
#include "abi_internal.h"

#define fail kbbq_fail
#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return fail(_e == hipErrorOutOfMemory ? KBBQ_ENOMEM : KBBQ_EIO, "%s: %s (%s:%d)", #expr,   \
                        hipGetErrorString(_e), __FILE__, __LINE__);                                    \
    } while (0)

namespace {

// ---- RCCL, loaded on first use (rccl.h's declarations, restated: the library is found at run time) -------------------
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[KBBQ_RCCL_ID_BYTES]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclUint8 = 1, ncclInt32 = 2, ncclUint64 = 5 };      // ncclDataType_t (rccl.h:460-464)
enum { ncclSum = 0 };                                        // ncclRedOp_t
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return KBBQ_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return fail(KBBQ_ENODEV, "RCCL is not available: %s", dlerror());
#define SYM(field, name)                                                                 \
    do {                                                                                 \
        *(void **)(&g_rccl.field) = dlsym(h, name);                                      \
        if (!g_rccl.field) { dlclose(h); return fail(KBBQ_ENODEV, "RCCL has no %s", name); } \
    } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
    SYM(AllGather, "ncclAllGather"); SYM(AllReduce, "ncclAllReduce"); SYM(Broadcast, "ncclBroadcast"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.lib = h;
    return KBBQ_OK;
}
#define RCCL_TRY(call)                                                                              \
    do {                                                                                            \
        const int _r = (call);                                                                      \
        if (_r != ncclSuccess) return fail(KBBQ_EIO, "%s: %s", #call, g_rccl.GetErrorString(_r));   \
    } while (0)

// ---- the ranks of one process ----------------------------------------------------------------------------------------
struct LocalShared {
    int n = 0, refs = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<const void *> ptr;
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t gen = generation;
        if (++arrived == n) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
    }
};

}  // namespace

struct kbbq_group {
    int rank = 0, n = 1;
    ncclComm_t comm = nullptr;      // rccl
    bool own_comm = false;
    LocalShared *sh = nullptr;      // local
    double ms[4] = {0, 0, 0, 0};

    // piece j of send goes to rank j; piece j of recv comes from rank j
    int all_to_all(const uint64_t *send, uint64_t *recv, size_t piece, hipStream_t st) {
        if (comm) {
            RCCL_TRY(g_rccl.GroupStart());
            for (int j = 0; j < n; ++j) {
                RCCL_TRY(g_rccl.Send(send + (size_t)j * piece, piece, ncclUint64, j, comm, st));
                RCCL_TRY(g_rccl.Recv(recv + (size_t)j * piece, piece, ncclUint64, j, comm, st));
            }
            RCCL_TRY(g_rccl.GroupEnd());
            return KBBQ_OK;
        }
        HIP_TRY(hipStreamSynchronize(st));      // what this rank sends is complete
        sh->ptr[rank] = send;
        sh->barrier();
        for (int j = 0; j < n; ++j)
            HIP_TRY(hipMemcpyAsync(recv + (size_t)j * piece, (const uint64_t *)sh->ptr[j] + (size_t)rank * piece, piece * 8, hipMemcpyDefault, st));
        HIP_TRY(hipStreamSynchronize(st));
        sh->barrier();                          // everybody has read: the senders' buffers are theirs again
        return KBBQ_OK;
    }
    int all_gather(const uint64_t *mine, uint64_t *all, size_t piece, hipStream_t st) {
        if (comm) {
            RCCL_TRY(g_rccl.AllGather(mine, all, piece, ncclUint64, comm, st));
            return KBBQ_OK;
        }
        HIP_TRY(hipStreamSynchronize(st));
        sh->ptr[rank] = mine;
        sh->barrier();
        for (int j = 0; j < n; ++j)
            HIP_TRY(hipMemcpyAsync(all + (size_t)j * piece, sh->ptr[j], piece * 8, hipMemcpyDefault, st));
        HIP_TRY(hipStreamSynchronize(st));
        sh->barrier();
        return KBBQ_OK;
    }
    int all_reduce_sum(uint64_t *buf, size_t words, hipStream_t st) {
        if (comm) {
            RCCL_TRY(g_rccl.AllReduce(buf, buf, words, ncclUint64, ncclSum, comm, st));
            return KBBQ_OK;
        }
        HIP_TRY(hipStreamSynchronize(st));
        sh->ptr[rank] = buf;
        sh->barrier();
        std::vector<uint64_t> sum(words, 0), part(words);
        // (every copy on this rank's stream and waited for there: a plain hipMemcpy between device buffers need not be
        // complete on return, and nothing orders the null stream against the engine's non-blocking one)
        for (int j = 0; j < n; ++j) {
            HIP_TRY(hipMemcpyAsync(part.data(), sh->ptr[j], words * 8, hipMemcpyDefault, st));
            HIP_TRY(hipStreamSynchronize(st));
            for (size_t i = 0; i < words; ++i) sum[i] += part[i];
        }
        sh->barrier();                          // everybody has read every buffer
        HIP_TRY(hipMemcpyAsync(buf, sum.data(), words * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        sh->barrier();
        return KBBQ_OK;
    }
    int broadcast(void *buf, size_t bytes, int root, hipStream_t st) {
        if (comm) {
            RCCL_TRY(g_rccl.Broadcast(buf, buf, bytes, ncclUint8, root, comm, st));
            return KBBQ_OK;
        }
        HIP_TRY(hipStreamSynchronize(st));
        sh->ptr[rank] = buf;
        sh->barrier();
        if (rank != root) {
            HIP_TRY(hipMemcpyAsync(buf, sh->ptr[root], bytes, hipMemcpyDefault, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        sh->barrier();
        return KBBQ_OK;
    }
};

namespace {

struct DeviceOf {      // the device an engine's memory lives on, made current for the call
    int prev = -1, dev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceOf(const void *device_ptr) {
        hipPointerAttribute_t a;
        err = hipPointerGetAttributes(&a, device_ptr);
        if (err != hipSuccess) return;
        dev = a.device;
        (void)hipGetDevice(&prev);
        if (prev != dev) err = hipSetDevice(dev); else prev = -1;
    }
    ~DeviceOf() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

extern "C" {

int kbbq_group_rccl_unique_id(uint8_t id[KBBQ_RCCL_ID_BYTES]) {
    if (!id) return fail(KBBQ_EINVAL, "null argument");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId u;
    RCCL_TRY(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, KBBQ_RCCL_ID_BYTES);
    return KBBQ_OK;
}

int kbbq_group_rccl_create(const uint8_t id[KBBQ_RCCL_ID_BYTES], int32_t rank, int32_t n_ranks, int32_t device, kbbq_group **out) {
    if (!id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(KBBQ_EINVAL, "bad argument");
    int rc = load_rccl();
    if (rc) return rc;
    KbbqDeviceGuard guard(device);
    HIP_TRY(guard.err);
    ncclUniqueId u;
    memcpy(u.internal, id, KBBQ_RCCL_ID_BYTES);
    ncclComm_t comm = nullptr;
    RCCL_TRY(g_rccl.CommInitRank(&comm, n_ranks, u, rank));
    kbbq_group *g = new kbbq_group;
    g->rank = rank; g->n = n_ranks; g->comm = comm; g->own_comm = true;
    *out = g;
    return KBBQ_OK;
}

int kbbq_group_from_nccl_comm(void *nccl_comm, int32_t rank, int32_t n_ranks, kbbq_group **out) {
    if (!nccl_comm || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(KBBQ_EINVAL, "bad argument");
    int rc = load_rccl();
    if (rc) return rc;
    kbbq_group *g = new kbbq_group;
    g->rank = rank; g->n = n_ranks; g->comm = (ncclComm_t)nccl_comm; g->own_comm = false;
    *out = g;
    return KBBQ_OK;
}

int kbbq_group_local_create(int32_t n_ranks, kbbq_group **out) {
    if (!out || n_ranks < 1 || n_ranks > 64) return fail(KBBQ_EINVAL, "bad argument");
    LocalShared *sh = new LocalShared;
    sh->n = n_ranks; sh->refs = n_ranks;
    sh->ptr.assign((size_t)n_ranks, nullptr);
    for (int r = 0; r < n_ranks; ++r) {
        kbbq_group *g = new kbbq_group;
        g->rank = r; g->n = n_ranks; g->sh = sh;
        out[r] = g;
    }
    return KBBQ_OK;
}

void kbbq_group_destroy(kbbq_group *g) {
    if (!g) return;
    if (g->comm && g->own_comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(g->comm);
    if (g->sh) {
        bool last;
        { std::lock_guard<std::mutex> lk(g->sh->mu); last = --g->sh->refs == 0; }
        if (last) delete g->sh;
    }
    delete g;
}

int kbbq_group_rank(const kbbq_group *g, int32_t *rank, int32_t *n_ranks) {
    if (!g) return fail(KBBQ_EINVAL, "null argument");
    if (rank) *rank = g->rank;
    if (n_ranks) *n_ranks = g->n;
    return KBBQ_OK;
}

int kbbq_exchange_ms(const kbbq_group *g, double out[4]) {
    if (!g || !out) return fail(KBBQ_EINVAL, "null argument");
    for (int i = 0; i < 4; ++i) out[i] = g->ms[i];
    return KBBQ_OK;
}

int kbbq_exchange_filter(kbbq_engine *e, int which, kbbq_group *g, uint64_t slab_words, uint64_t *inserted_total) {
    if (!e || !g || which < 0 || which > 1) return fail(KBBQ_EINVAL, "bad argument");
    int rc = kbbq_engine_sync(e);      // deferred inserts reach the filter, every kernel of the pass is through
    if (rc) return rc;
    const double t0 = now_ms();        // (the step's own time: what the pass still had queued is the pass's)
    kbbq_filter_info info;
    if ((rc = kbbq_filter_info_get(e, which, &info))) return rc;
    uint64_t *table = (uint64_t *)kbbq_filter_device_table(e, which);
    if (!table) return fail(KBBQ_ESTATE, "the filter has no device table");
    DeviceOf dev(table);
    HIP_TRY(dev.err);
    hipStream_t st = (hipStream_t)kbbq_engine_stream(e);
    const uint64_t total = info.table_bytes / 8;
    const int n = g->n, rank = g->rank;
    if (!slab_words) slab_words = 1ull << 26;
    uint64_t piece = std::max<uint64_t>(2, (std::min(slab_words, total) + n - 1) / n);
    piece += piece & 1;                    // pieces stay 16-byte aligned (kbbq_device_or_pieces)
    const uint64_t slab = piece * n, n_slabs = (total + slab - 1) / slab, tail = total - (n_slabs - 1) * slab;
    DevBuf recv, mine, pad, cnt;
    HIP_TRY(hipMalloc(&recv.p, slab * 8));
    HIP_TRY(hipMalloc(&mine.p, piece * 8));
    HIP_TRY(hipMalloc(&cnt.p, 64));
    if (tail != slab) {                    // the ragged last slab travels through a zero-padded copy
        HIP_TRY(hipMalloc(&pad.p, slab * 8));
        HIP_TRY(hipMemsetAsync(pad.p, 0, slab * 8, st));
        HIP_TRY(hipMemcpyAsync(pad.p, table + (total - tail), tail * 8, hipMemcpyDeviceToDevice, st));
    }
    for (uint64_t s = 0; s < n_slabs; ++s) {
        uint64_t *view = (s == n_slabs - 1 && pad.p) ? (uint64_t *)pad.p : table + s * slab;
        if ((rc = g->all_to_all(view, (uint64_t *)recv.p, piece, st))) return rc;
        HIP_TRY(hipMemcpyAsync(mine.p, (uint64_t *)recv.p + (uint64_t)rank * piece, piece * 8, hipMemcpyDeviceToDevice, st));
        if (n > 1 && (rc = kbbq_device_or_pieces(e, mine.p, recv.p, piece, n, rank))) return rc;      // (on the engine's stream)
        if ((rc = g->all_gather((const uint64_t *)mine.p, view, piece, st))) return rc;
    }
    if (pad.p) HIP_TRY(hipMemcpyAsync(table + (total - tail), pad.p, tail * 8, hipMemcpyDeviceToDevice, st));
    // the insert counter (inserted_element_count_ counts duplicates: a sum, not a population count)
    uint64_t count = info.inserted;
    HIP_TRY(hipMemcpyAsync(cnt.p, &count, 8, hipMemcpyHostToDevice, st));
    if ((rc = g->all_reduce_sum((uint64_t *)cnt.p, 1, st))) return rc;
    HIP_TRY(hipMemcpyAsync(&count, cnt.p, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if ((rc = kbbq_filter_set_inserted(e, which, count))) return rc;
    if (inserted_total) *inserted_total = count;
    g->ms[which] = now_ms() - t0;
    return KBBQ_OK;
}

int kbbq_exchange_histograms(kbbq_engine *e, kbbq_group *g) {
    if (!e || !g) return fail(KBBQ_EINVAL, "bad argument");
    int rc = kbbq_engine_sync(e);
    if (rc) return rc;
    const double t0 = now_ms();
    uint64_t n_words = 0;
    uint64_t *h = (uint64_t *)kbbq_covariates_device(e, &n_words);
    if (!h || !n_words) return fail(KBBQ_ESTATE, "the engine has no histograms");
    DeviceOf dev(h);
    HIP_TRY(dev.err);
    hipStream_t st = (hipStream_t)kbbq_engine_stream(e);
    if ((rc = g->all_reduce_sum(h, n_words, st))) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    g->ms[2] = now_ms() - t0;
    return KBBQ_OK;
}

int kbbq_exchange_dq(kbbq_engine *e, kbbq_group *g) {
    if (!e || !g) return fail(KBBQ_EINVAL, "bad argument");
    const double t0 = now_ms();
    uint64_t n_rg = 0, n_cycle = 0;
    int rc = kbbq_engine_dims(e, &n_rg, &n_cycle);
    if (rc) return rc;
    const size_t sz[5] = {(size_t)n_rg, (size_t)n_rg, (size_t)n_rg * KBBQ_NQ, (size_t)n_rg * KBBQ_NQ * 2 * n_cycle, (size_t)n_rg * KBBQ_NQ * 16};
    size_t words = 0;
    for (size_t s : sz) words += s;
    std::vector<int32_t> flat(words, 0);
    kbbq_dq dq;
    dq.n_rg = n_rg; dq.n_cycle = n_cycle;
    dq.meanq = flat.data(); dq.rgdq = dq.meanq + sz[0]; dq.qdq = dq.rgdq + sz[1]; dq.cycledq = dq.qdq + sz[2]; dq.dinucdq = dq.cycledq + sz[3];
    if (g->rank == 0) {
        // get_dqs on rank 0 alone (covariateutils.cc:204-230: x87 long double on the host), as the north star's "the final
        // delta-Q table is broadcast"
        if ((rc = kbbq_train(e))) return rc;
        if ((rc = kbbq_dq_get(e, &dq))) return rc;
    }
    uint64_t n_words = 0;
    void *h = kbbq_covariates_device(e, &n_words);      // (any device pointer of the engine: which device it lives on)
    DeviceOf dev(h);
    HIP_TRY(dev.err);
    hipStream_t st = (hipStream_t)kbbq_engine_stream(e);
    DevBuf buf;
    HIP_TRY(hipMalloc(&buf.p, words * 4 + 64));
    HIP_TRY(hipMemcpyAsync(buf.p, flat.data(), words * 4, hipMemcpyHostToDevice, st));
    if ((rc = g->broadcast(buf.p, words * 4, 0, st))) return rc;
    HIP_TRY(hipMemcpyAsync(flat.data(), buf.p, words * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (g->rank != 0 && (rc = kbbq_set_dq(e, &dq))) return rc;
    g->ms[3] = now_ms() - t0;
    return KBBQ_OK;
}

}  // extern "C"
