// bgzf_device.hip -- the BGZF writer object behind include/kbbq_bgzf.h: buffers, streams, two submissions in flight,
// and the launches of bgzf_device.h's kernels (MI355X, gfx950).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/kbbq_bgzf.h"
#include "../../include/kbbq_engine.h"
#include "abi_internal.h"
#include "bgzf_device.h"

using namespace kbbq::dfl;

#define fail kbbq_fail
#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return fail(_e == hipErrorOutOfMemory ? KBBQ_ENOMEM : KBBQ_EIO, "%s: %s (%s:%d)", #expr,   \
                        hipGetErrorString(_e), __FILE__, __LINE__);                                    \
    } while (0)

namespace {

// a device (or page-locked host) buffer that only ever grows
struct Buf {
    void *p = nullptr;
    size_t bytes = 0;
    bool host = false;
    int reserve(size_t need) {
        if (bytes >= need) return KBBQ_OK;
        if (p) { if (host) (void)hipHostFree(p); else (void)hipFree(p); p = nullptr; bytes = 0; }
        const size_t want = need + need / 8 + 4096;
        HIP_TRY(host ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want));
        bytes = want;
        return KBBQ_OK;
    }
    void release() {
        if (p) { if (host) (void)hipHostFree(p); else (void)hipFree(p); }
        p = nullptr; bytes = 0;
    }
};

struct Submission {
    Buf payload, slots, sizes, offsets, out, h_out, h_meta;      // h_*: page-locked host memory
    Buf blob, lens, blob_off, text_off, h_off;                  // FASTQ pieces (device) and the host staging of the offsets
    hipEvent_t ev_meta = nullptr, ev_done = nullptr;            // total size known; blocks gathered
    hipEvent_t t0 = nullptr, t1 = nullptr, t2 = nullptr, t3 = nullptr;      // kernel timing: format | deflate | gather
    uint64_t n = 0;
    uint32_t n_blocks = 0;
    bool busy = false, formatted = false;
};

}  // namespace

struct kbbq_bgzf {
    int device = 0;
    hipStream_t st = nullptr, copy = nullptr;
    hipEvent_t ev_after = nullptr;
    Submission sub[2];
    int head = 0, tail = 0, in_flight = 0;
    Buf tokens;
    int grid = 0;
    double ms_format = 0, ms_deflate = 0, ms_gather = 0;
};

namespace {

int launch_deflate(kbbq_bgzf *z, Submission &s) {
    s.n_blocks = (uint32_t)((s.n + BGZF_PAYLOAD - 1) / BGZF_PAYLOAD);
    int rc;
    if ((rc = s.slots.reserve((size_t)s.n_blocks * SLOT_BYTES))) return rc;
    if ((rc = s.sizes.reserve((size_t)s.n_blocks * 4))) return rc;
    if ((rc = s.offsets.reserve(((size_t)s.n_blocks + 1) * 8))) return rc;
    const size_t bound = (size_t)kbbq_bgzf_bound(s.n);
    if ((rc = s.out.reserve(bound))) return rc;
    if ((rc = s.h_out.reserve(bound))) return rc;
    if ((rc = s.h_meta.reserve(64))) return rc;
    // one wavefront per block in flight; the chip holds 8 of these workgroups per CU (18 KB of LDS each)
    if (!z->grid) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, z->device));
        z->grid = std::max(1, prop.multiProcessorCount) * 8;
    }
    const int grid = (int)std::min<uint32_t>(s.n_blocks, (uint32_t)z->grid);
    if ((rc = z->tokens.reserve((size_t)z->grid * DFL_WAVES * TOKENS_PER_WAVE * 4))) return rc;
    HIP_TRY(hipMemsetAsync(s.slots.p, 0, (size_t)s.n_blocks * SLOT_BYTES, z->st));
    HIP_TRY(hipEventRecord(s.t1, z->st));
    DeflateArgs A;
    A.payload = (const uint8_t *)s.payload.p;
    A.n = s.n;
    A.n_blocks = s.n_blocks;
    A.slots = (uint8_t *)s.slots.p;
    A.sizes = (uint32_t *)s.sizes.p;
    A.tokens = (uint32_t *)z->tokens.p;
    hipLaunchKernelGGL(k_deflate, dim3(grid), dim3(64 * DFL_WAVES), 0, z->st, A);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s.t2, z->st));
    hipLaunchKernelGGL(k_block_offsets, dim3(1), dim3(1024), 0, z->st, (const uint32_t *)s.sizes.p, s.n_blocks, (uint64_t *)s.offsets.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(s.h_meta.p, (const uint64_t *)s.offsets.p + s.n_blocks, 8, hipMemcpyDeviceToHost, z->st));
    HIP_TRY(hipEventRecord(s.ev_meta, z->st));
    hipLaunchKernelGGL(k_gather, dim3(s.n_blocks), dim3(256), 0, z->st, (const uint8_t *)s.slots.p, (const uint32_t *)s.sizes.p,
                       (const uint64_t *)s.offsets.p, (uint8_t *)s.out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s.t3, z->st));
    HIP_TRY(hipEventRecord(s.ev_done, z->st));
    s.busy = true;
    z->head ^= 1;
    z->in_flight += 1;
    return KBBQ_OK;
}

int begin_submission(kbbq_bgzf *z, void *after_stream, Submission **out) {
    if (z->in_flight >= 2) return fail(KBBQ_ESTATE, "two submissions are in flight: collect one first");
    Submission &s = z->sub[z->head];
    if (s.busy) return fail(KBBQ_ESTATE, "the writer's slot is still in flight");
    if (after_stream) {
        HIP_TRY(hipEventRecord(z->ev_after, (hipStream_t)after_stream));
        HIP_TRY(hipStreamWaitEvent(z->st, z->ev_after, 0));
    }
    *out = &s;
    return KBBQ_OK;
}

}  // namespace

extern "C" {

int kbbq_bgzf_create(int32_t device, kbbq_bgzf **out) {
    if (!out) return fail(KBBQ_EINVAL, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(KBBQ_ENODEV, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(KBBQ_ENODEV, "device %d of %d", device, ndev);
    KbbqDeviceGuard guard(device);
    HIP_TRY(guard.err);
    kbbq_bgzf *z = new kbbq_bgzf;
    z->device = device;
    hipError_t he = hipStreamCreateWithFlags(&z->st, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&z->copy, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&z->ev_after, hipEventDisableTiming);
    for (int i = 0; i < 2 && he == hipSuccess; ++i) {
        Submission &s = z->sub[i];
        s.h_out.host = s.h_meta.host = s.h_off.host = true;
        he = hipEventCreateWithFlags(&s.ev_meta, hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventCreate(&s.t0);
        if (he == hipSuccess) he = hipEventCreate(&s.t1);
        if (he == hipSuccess) he = hipEventCreate(&s.t2);
        if (he == hipSuccess) he = hipEventCreate(&s.t3);
    }
    if (he != hipSuccess) {
        kbbq_bgzf_destroy(z);
        return fail(KBBQ_EIO, "creating the writer's streams: %s", hipGetErrorString(he));
    }
    *out = z;
    return KBBQ_OK;
}

void kbbq_bgzf_destroy(kbbq_bgzf *z) {
    if (!z) return;
    KbbqDeviceGuard guard(z->device);
    if (z->st) (void)hipStreamSynchronize(z->st);
    if (z->copy) (void)hipStreamSynchronize(z->copy);
    for (int i = 0; i < 2; ++i) {
        Submission &s = z->sub[i];
        Buf *all[] = {&s.payload, &s.slots, &s.sizes, &s.offsets, &s.out, &s.h_out, &s.h_meta, &s.blob, &s.lens, &s.blob_off, &s.text_off, &s.h_off};
        for (Buf *b : all) b->release();
        hipEvent_t evs[] = {s.ev_meta, s.ev_done, s.t0, s.t1, s.t2, s.t3};
        for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    }
    z->tokens.release();
    if (z->ev_after) (void)hipEventDestroy(z->ev_after);
    if (z->copy) (void)hipStreamDestroy(z->copy);
    if (z->st) (void)hipStreamDestroy(z->st);
    delete z;
}

int kbbq_bgzf_submit(kbbq_bgzf *z, const void *payload, uint64_t n, int32_t payload_on_device, void *after_stream) {
    if (!z || !payload || !n) return fail(KBBQ_EINVAL, "bad argument");
    if ((n + BGZF_PAYLOAD - 1) / BGZF_PAYLOAD > 0xFFFFFFFFull) return fail(KBBQ_ERANGE, "more than 2^32 blocks in one submission");
    KbbqDeviceGuard guard(z->device);
    HIP_TRY(guard.err);
    Submission *sp;
    int rc = begin_submission(z, after_stream, &sp);
    if (rc) return rc;
    Submission &s = *sp;
    s.n = n;
    s.formatted = false;
    if ((rc = s.payload.reserve(n + 16))) return rc;
    // the encoder reads whole 8-byte words: zeros behind the last byte
    HIP_TRY(hipMemsetAsync((char *)s.payload.p + n, 0, 16, z->st));
    HIP_TRY(hipMemcpyAsync(s.payload.p, payload, n, payload_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, z->st));
    HIP_TRY(hipEventRecord(s.t0, z->st));
    if ((rc = launch_deflate(z, s))) return rc;
    // a host payload is the caller's again on return (page-locked memory: the copy above is asynchronous)
    if (!payload_on_device) HIP_TRY(hipEventSynchronize(s.t0));
    return KBBQ_OK;
}

int kbbq_bgzf_submit_fastq(kbbq_bgzf *z, const char *blob, const uint32_t *lens, uint64_t n_records, const uint8_t *d_qual,
                           const uint64_t *d_qual_offsets, uint32_t uniform_len, void *after_stream) {
    if (!z || !blob || !lens || !n_records || !d_qual) return fail(KBBQ_EINVAL, "bad argument");
    if (!d_qual_offsets && !uniform_len) return fail(KBBQ_EINVAL, "neither quality offsets nor a uniform read length");
    KbbqDeviceGuard guard(z->device);
    HIP_TRY(guard.err);
    Submission *sp;
    int rc = begin_submission(z, after_stream, &sp);
    if (rc) return rc;
    Submission &s = *sp;
    // where every record's pieces and text start: two running sums over the lengths, on the host (it holds them)
    if ((rc = s.h_off.reserve((n_records + 1) * 16))) return rc;
    uint64_t *h_blob_off = (uint64_t *)s.h_off.p, *h_text_off = h_blob_off + n_records + 1;
    uint64_t b = 0, t = 0;
    for (uint64_t r = 0; r < n_records; ++r) {
        h_blob_off[r] = b;
        h_text_off[r] = t;
        const uint64_t nl = lens[3 * r], cl = lens[3 * r + 1], sl = lens[3 * r + 2];
        if (d_qual_offsets == nullptr && sl != uniform_len) return fail(KBBQ_EINVAL, "record %llu has %llu bases, not the uniform %u", (unsigned long long)r, (unsigned long long)sl, uniform_len);
        b += nl + cl + sl;
        t += nl + cl + 2 * sl + 6;
    }
    h_blob_off[n_records] = b;
    h_text_off[n_records] = t;
    s.n = t;
    s.formatted = true;
    if ((rc = s.blob.reserve(b + 16))) return rc;
    if ((rc = s.lens.reserve(n_records * 12))) return rc;
    if ((rc = s.blob_off.reserve((n_records + 1) * 8))) return rc;
    if ((rc = s.text_off.reserve((n_records + 1) * 8))) return rc;
    if ((rc = s.payload.reserve(t + 16))) return rc;
    HIP_TRY(hipMemcpyAsync(s.blob.p, blob, b, hipMemcpyHostToDevice, z->st));
    HIP_TRY(hipMemcpyAsync(s.lens.p, lens, n_records * 12, hipMemcpyHostToDevice, z->st));
    HIP_TRY(hipMemcpyAsync(s.blob_off.p, h_blob_off, (n_records + 1) * 8, hipMemcpyHostToDevice, z->st));
    HIP_TRY(hipMemcpyAsync(s.text_off.p, h_text_off, (n_records + 1) * 8, hipMemcpyHostToDevice, z->st));
    HIP_TRY(hipMemsetAsync((char *)s.payload.p + t, 0, 16, z->st));
    HIP_TRY(hipEventRecord(s.t0, z->st));
    FastqArgs F;
    F.blob = (const uint8_t *)s.blob.p;
    F.lens = (const uint32_t *)s.lens.p;
    F.blob_off = (const uint64_t *)s.blob_off.p;
    F.text_off = (const uint64_t *)s.text_off.p;
    F.qual = d_qual;
    F.qual_off = d_qual_offsets;
    F.uniform_len = uniform_len;
    F.n_records = n_records;
    F.text = (uint8_t *)s.payload.p;
    const unsigned grid = (unsigned)std::min<uint64_t>((n_records + 3) / 4, 256 * 32);
    hipLaunchKernelGGL(k_fastq_text, dim3(grid), dim3(256), 0, z->st, F);
    HIP_TRY(hipGetLastError());
    if ((rc = launch_deflate(z, s))) return rc;
    // blob and lens are the caller's again on return
    HIP_TRY(hipEventSynchronize(s.t0));
    return KBBQ_OK;
}

int kbbq_bgzf_collect(kbbq_bgzf *z, const uint8_t **blocks, uint64_t *n_bytes, uint64_t *payload_bytes) {
    if (!z || !blocks || !n_bytes) return fail(KBBQ_EINVAL, "null argument");
    if (!z->in_flight) return fail(KBBQ_ESTATE, "nothing was submitted");
    KbbqDeviceGuard guard(z->device);
    HIP_TRY(guard.err);
    Submission &s = z->sub[z->tail];
    HIP_TRY(hipEventSynchronize(s.ev_meta));
    const uint64_t total = *(const uint64_t *)s.h_meta.p;
    if (total > s.h_out.bytes) return fail(KBBQ_ESTATE, "compressed size %llu exceeds its bound", (unsigned long long)total);
    // the copy back runs on its own stream: the kernels of the next submission are not held up behind it
    HIP_TRY(hipStreamWaitEvent(z->copy, s.ev_done, 0));
    HIP_TRY(hipMemcpyAsync(s.h_out.p, s.out.p, total, hipMemcpyDeviceToHost, z->copy));
    HIP_TRY(hipStreamSynchronize(z->copy));
    float a = 0, b = 0, c = 0;
    if (hipEventElapsedTime(&a, s.t0, s.t1) == hipSuccess && s.formatted) z->ms_format += a;
    if (hipEventElapsedTime(&b, s.t1, s.t2) == hipSuccess) z->ms_deflate += b;
    if (hipEventElapsedTime(&c, s.t2, s.t3) == hipSuccess) z->ms_gather += c;
    *blocks = (const uint8_t *)s.h_out.p;
    *n_bytes = total;
    if (payload_bytes) *payload_bytes = s.n;
    s.busy = false;
    z->tail ^= 1;
    z->in_flight -= 1;
    return KBBQ_OK;
}

const uint8_t *kbbq_bgzf_eof_block(void) {
    static const uint8_t eof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43,
                                    0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return eof;
}

int kbbq_bgzf_kernel_ms(kbbq_bgzf *z, double *format_ms, double *deflate_ms, double *gather_ms) {
    if (!z) return fail(KBBQ_EINVAL, "null argument");
    if (format_ms) *format_ms = z->ms_format;
    if (deflate_ms) *deflate_ms = z->ms_deflate;
    if (gather_ms) *gather_ms = z->ms_gather;
    return KBBQ_OK;
}

}  // extern "C"
