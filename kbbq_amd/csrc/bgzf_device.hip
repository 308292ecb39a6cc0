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
#include "bgzf_inflate.h"
#include "bam_device.h"

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
    bool exact = false;      // no room to grow into: the buffer is filled once and kept
    int reserve(size_t need) {
        if (bytes >= need) return KBBQ_OK;
        if (p) { if (host) (void)hipHostFree(p); else (void)hipFree(p); p = nullptr; bytes = 0; }
        const size_t want = exact ? need : need + need / 8 + 4096;
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
    Buf payload, slots, sizes, offsets, out, h_meta;             // h_*: page-locked host memory
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
    Buf h_out[3];               // page-locked: the blocks of the last three collected submissions (kbbq_bgzf_collect)
    int h_next = 0;
    Buf tokens;
#ifdef KBBQ_DFL_PROFILE
    void *prof = nullptr;
#endif
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
    if ((rc = s.h_meta.reserve(64))) return rc;
    // one wavefront per block in flight, as many as stay resident (11 KB of LDS, 152 registers: 12 per CU)
    if (!z->grid) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, z->device));
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_deflate, 64 * DFL_WAVES, 0));
        z->grid = std::max(1, prop.multiProcessorCount) * std::max(1, per_cu);
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
#ifdef KBBQ_DFL_PROFILE
    if (!z->prof && hipMalloc(&z->prof, 16 * 8) == hipSuccess) (void)hipMemset(z->prof, 0, 16 * 8);
    A.prof = (unsigned long long *)z->prof;
#endif
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
        s.h_meta.host = s.h_off.host = true;
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
        Buf *all[] = {&s.payload, &s.slots, &s.sizes, &s.offsets, &s.out, &s.h_meta, &s.blob, &s.lens, &s.blob_off, &s.text_off, &s.h_off};
        for (Buf *b : all) b->release();
        hipEvent_t evs[] = {s.ev_meta, s.ev_done, s.t0, s.t1, s.t2, s.t3};
        for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    }
    z->tokens.release();
    for (Buf &b : z->h_out) b.release();
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
    if (total > s.out.bytes) return fail(KBBQ_ESTATE, "compressed size %llu exceeds its bound", (unsigned long long)total);
    // three host buffers in turn: what a collect returns stays put while the next two are collected (a caller writes one
    // submission's blocks out while it waits for the next)
    Buf &h_out = z->h_out[z->h_next];
    z->h_next = (z->h_next + 1) % 3;
    h_out.host = true;
    int rc;
    if ((rc = h_out.reserve((size_t)total + 64))) return rc;
    // the copy back runs on its own stream: the kernels of the next submission are not held up behind it
    HIP_TRY(hipStreamWaitEvent(z->copy, s.ev_done, 0));
    HIP_TRY(hipMemcpyAsync(h_out.p, s.out.p, total, hipMemcpyDeviceToHost, z->copy));
    HIP_TRY(hipStreamSynchronize(z->copy));
    float a = 0, b = 0, c = 0;
    if (hipEventElapsedTime(&a, s.t0, s.t1) == hipSuccess && s.formatted) z->ms_format += a;
    if (hipEventElapsedTime(&b, s.t1, s.t2) == hipSuccess) z->ms_deflate += b;
    if (hipEventElapsedTime(&c, s.t2, s.t3) == hipSuccess) z->ms_gather += c;
    *blocks = (const uint8_t *)h_out.p;
    *n_bytes = total;
    if (payload_bytes) *payload_bytes = s.n;
    s.busy = false;
    z->tail ^= 1;
    z->in_flight -= 1;
    return KBBQ_OK;
}

#ifdef KBBQ_DFL_PROFILE
// (a build of its own for tools/deflate_probe.py: cycles per phase of k_deflate, summed over the wavefronts)
int kbbq_bgzf_debug_profile(kbbq_bgzf *z, uint64_t *out) {
    if (!z || !out || !z->prof) return fail(KBBQ_EINVAL, "no profile");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, z->prof, 16 * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(z->prof, 0, 16 * 8));
    return KBBQ_OK;
}
#endif

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

// ============================================================ the input side: BGZF FASTQ read on the device

// A piece of the file copied to the device AHEAD of the chunk call that will take it (kbbq_*_reader_preload): the caller's I/O
// thread starts the copy the moment a piece has been read, on a copy stream of the reader's own, so the host link moves piece
// i + 1 while the kernels of piece i run -- without it a chunk's 256 MB cross the link in front of its own inflation, 1.1 s
// of a 30x run.  Two slots; a slot holds the host range [host, host + n) at dev + front: the bytes a chunk call carries over
// from the piece before (less than a BGZF block) go in front of them.
struct Preload {
    hipStream_t copy = nullptr;
    Buf dev[2];
    hipEvent_t done[2] = {nullptr, nullptr};
    const uint8_t *host[2] = {nullptr, nullptr};
    uint64_t n[2] = {0, 0};
    uint64_t front = 0;
    int next = 0;
    int start(int device, const uint8_t *bytes, uint64_t n_bytes, uint64_t front_room) {
        if (!copy) {
            HIP_TRY(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
            for (int i = 0; i < 2; ++i) HIP_TRY(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
        }
        const int i = next;
        next ^= 1;
        host[i] = nullptr;
        front = front_room;
        int rc = dev[i].reserve(front_room + n_bytes + 4096);
        if (rc) { (void)hipGetLastError(); return KBBQ_OK; }      // no room: the chunk call copies as before
        HIP_TRY(hipMemcpyAsync((char *)dev[i].p + front_room, bytes, n_bytes, hipMemcpyHostToDevice, copy));
        HIP_TRY(hipMemsetAsync((char *)dev[i].p + front_room + n_bytes, 0, 4096, copy));
        HIP_TRY(hipEventRecord(done[i], copy));
        host[i] = bytes;
        n[i] = n_bytes;
        return KBBQ_OK;
    }
    // the device address of file_bytes[0, n_bytes) if it ends a preloaded piece and starts at most `front` bytes before it
    // (those first bytes are copied here, on st); st then waits for the piece's copy.  nullptr: not preloaded.
    void *take(const uint8_t *file_bytes, uint64_t n_bytes, hipStream_t st) {
        for (int i = 0; i < 2; ++i) {
            if (!host[i] || file_bytes > host[i] || file_bytes + n_bytes != host[i] + n[i]) continue;
            const uint64_t prefix = (uint64_t)(host[i] - file_bytes);
            if (prefix > front) continue;
            char *at = (char *)dev[i].p + front - prefix;
            if (prefix && hipMemcpyAsync(at, file_bytes, prefix, hipMemcpyHostToDevice, st) != hipSuccess) return nullptr;
            if (hipStreamWaitEvent(st, done[i], 0) != hipSuccess) return nullptr;
            host[i] = nullptr;      // (the slot is written again only by a later preload: the caller's buffer protocol orders that)
            return at;
        }
        return nullptr;
    }
    void release() {
        for (int i = 0; i < 2; ++i) { dev[i].release(); if (done[i]) (void)hipEventDestroy(done[i]); done[i] = nullptr; host[i] = nullptr; }
        if (copy) (void)hipStreamDestroy(copy);
        copy = nullptr;
    }
};

// How many k_inflate wavefronts the device keeps resident (registers and LDS decide): the grid of every inflate launch -- blocks
// are handed out round-robin, a second round of workgroups would only queue behind the first.  KBBQ_DEBUG_CODEC=1 prints it.
static int inflate_resident_waves(int device, unsigned *out) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_inflate, 64 * INF_WAVES, 0));
    *out = (unsigned)std::max(1, prop.multiProcessorCount) * (unsigned)std::max(1, per_cu);
    if (getenv("KBBQ_DEBUG_CODEC")) fprintf(stderr, "k_inflate: %d wavefronts per CU on %d CUs\n", per_cu, prop.multiProcessorCount);
    return KBBQ_OK;
}

struct kbbq_fastq_reader {
    Preload pre;
    int device = 0;
    hipStream_t st = nullptr;
    hipEvent_t t0 = nullptr, t1 = nullptr, t2 = nullptr;
    Buf comp, text, status;                 // compressed chunk, inflated text (carry first), per-block status
    Buf blk_meta, h_meta;                   // per block: c_off, o_off (u64), c_len, o_len (u32) -- device and page-locked host copies
    Buf tile_counts, tile_sums, nl_pos;     // newline index
    Buf idx_u32, idx_second, base_sz, text_sz, flags;      // record index (FastqIndex)
    Buf carry;                              // text of the record the previous chunk's end cut (device)
    Buf h_small;                            // page-locked scratch for small read-backs
    Buf seq_text, counter;                  // scratch of kbbq_fastq_reader_batch (the chunk's sequence lines back to back)
    unsigned inflate_grid = 0;
    uint64_t carry_bytes = 0;
    // the current chunk
    uint64_t text_bytes = 0, n_records = 0, n_bases = 0;
    uint64_t out_text_bytes = 0;            // bytes of the chunk's records written out as FASTQ text
    uint32_t longest = 0, shortest = 0;
    bool have_chunk = false;
    double ms_inflate = 0, ms_index = 0;
    // chunks of the first scan that stay in device memory (kbbq_fastq_reader_keep): their text and record index
    // Two forms: the whole text with its record index, or -- when the chunk's batch was built and its sequence lines hold
    // nothing but ACGTN / acgt, so that the packed batch gives them back exactly -- only names and comments (a seventh of the
    // memory: what is not allocated need not be cleared by the driver either, 20-36 ms per GB of re-used memory).
    struct Kept {
        Buf text, idx_u32, base_sz, text_sz;
        Buf names, lens;            // the short form: names + comments back to back, (name, comment) lengths
        bool short_form = false;
        uint64_t text_bytes = 0, n_records = 0, n_bases = 0, out_text_bytes = 0;
        uint32_t longest = 0, shortest = 0;
    };
    bool packed_is_exact = false;              // the current chunk's batch gives its sequence text back (set by batch())
    const uint64_t *att_bases = nullptr, *att_nmask = nullptr, *att_offcase = nullptr;      // kbbq_fastq_reader_attach
    std::vector<Kept> kept;
    // the short form's arrays are carved from slabs of 2 GB (four hipMalloc per chunk were four trips to the driver)
    std::vector<Buf> slabs;
    size_t slab_used = 0;
    bool keeping = false;
    int64_t selected = -1;      // the kept chunk that is the current one (pass 4), or -1: the live buffers
    uint64_t kept_bytes = 0;
};

namespace {

// exclusive scan of d[0, n) in place, the total to *d_total (device); tile_sums: the caller's scratch
int device_scan_on(Buf &tile_sums, hipStream_t st, uint64_t *d, uint64_t n, uint64_t *d_total /* device */) {
    const uint64_t n_tiles = (n + DSCAN_TILE - 1) / DSCAN_TILE;
    int rc = tile_sums.reserve((n_tiles + 1) * 8);
    if (rc) return rc;
    uint64_t *ts = (uint64_t *)tile_sums.p;
    hipLaunchKernelGGL(k_dscan_tiles, dim3((unsigned)n_tiles), dim3(256), 0, st, d, n, ts);
    hipLaunchKernelGGL(k_dscan_sums, dim3(1), dim3(1024), 0, st, ts, n_tiles, d_total);
    hipLaunchKernelGGL(k_dscan_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d, n, (const uint64_t *)ts);
    HIP_TRY(hipGetLastError());
    return KBBQ_OK;
}
int device_scan(kbbq_fastq_reader *r, uint64_t *d, uint64_t n, uint64_t *d_total /* device */) {
    return device_scan_on(r->tile_sums, r->st, d, n, d_total);
}

FastqIndex index_from(void *idx_u32, void *second, void *base_sz, void *text_sz, void *flags, uint64_t cap) {
    FastqIndex X;
    uint32_t *u = (uint32_t *)idx_u32;
    X.name_off = u; X.name_len = u + cap; X.com_off = u + 2 * cap; X.com_len = u + 3 * cap;
    X.seq_off = u + 4 * cap; X.seq_len = u + 5 * cap; X.qual_off = u + 6 * cap;
    X.second = (uint8_t *)second;
    X.base_sz = (uint64_t *)base_sz;
    X.text_sz = (uint64_t *)text_sz;
    X.flags = (uint32_t *)flags;
    return X;
}
FastqIndex index_of(kbbq_fastq_reader *r, uint64_t cap) {
    return index_from(r->idx_u32.p, r->idx_second.p, r->base_sz.p, r->text_sz.p, r->flags.p, cap);
}

void release_kept(kbbq_fastq_reader *r) {
    for (auto &k : r->kept) {
        if (k.short_form) continue;      // (its arrays are pieces of the slabs)
        k.text.release(); k.idx_u32.release(); k.base_sz.release(); k.text_sz.release();
    }
    for (auto &b : r->slabs) b.release();
    r->slabs.clear();
    r->slab_used = 0;
    r->kept.clear();
    r->kept_bytes = 0;
    r->selected = -1;
}

// a piece of a slab (256-byte aligned); null when the device is full
void *slab_piece(kbbq_fastq_reader *r, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (r->slabs.empty() || r->slab_used + bytes > r->slabs.back().bytes) {
        // slabs grow from 64 MB to 2 GB (a small file keeps little)
        const size_t next = r->slabs.empty() ? ((size_t)64 << 20) : std::min<size_t>(r->slabs.back().bytes * 2, (size_t)2 << 30);
        Buf b;
        b.exact = true;
        if (b.reserve(std::max(next, bytes))) { (void)hipGetLastError(); return nullptr; }
        r->slabs.push_back(b);
        r->slab_used = 0;
    }
    void *p = (char *)r->slabs.back().p + r->slab_used;
    r->slab_used += bytes;
    r->kept_bytes += bytes;
    return p;
}

// The live chunk moves into the kept list (its buffers with it: the next chunk allocates its own).
void stash_current(kbbq_fastq_reader *r) {
    if (!r->keeping || r->selected >= 0 || !r->have_chunk || !r->n_records) return;
    kbbq_fastq_reader::Kept k;
    k.text_bytes = r->text_bytes; k.n_records = r->n_records; k.n_bases = r->n_bases;
    k.longest = r->longest; k.shortest = r->shortest;
    k.out_text_bytes = r->out_text_bytes;
    const uint64_t n = r->n_records;
    const uint64_t names_bytes = r->out_text_bytes - 2 * r->n_bases - 6 * n;      // sum of name + comment lengths
    bool short_form = r->packed_is_exact;
    if (short_form) {
        // names, lengths and the two scans into pieces of exactly their size; the working buffers stay the reader's
        k.names.p = slab_piece(r, names_bytes + 64);
        k.lens.p = k.names.p ? slab_piece(r, n * 8) : nullptr;
        k.base_sz.p = k.lens.p ? slab_piece(r, (n + 1) * 8) : nullptr;
        k.text_sz.p = k.base_sz.p ? slab_piece(r, (n + 1) * 8) : nullptr;
        if (!k.text_sz.p) {
            k.names.p = k.lens.p = k.base_sz.p = k.text_sz.p = nullptr;
            short_form = false;      // (the long form below takes the buffers that exist already)
        } else {
            const FastqIndex X = index_of(r, n);
            hipLaunchKernelGGL(k_fastq_keep_names, dim3((unsigned)std::min<uint64_t>((n + 3) / 4, 256 * 32)), dim3(256), 0, r->st, (const uint8_t *)r->text.p, X,
                               (const uint64_t *)X.text_sz, (const uint64_t *)X.base_sz, n, (uint8_t *)k.names.p, (uint32_t *)k.lens.p);
            (void)hipMemcpyAsync(k.base_sz.p, X.base_sz, (n + 1) * 8, hipMemcpyDeviceToDevice, r->st);
            (void)hipMemcpyAsync(k.text_sz.p, X.text_sz, (n + 1) * 8, hipMemcpyDeviceToDevice, r->st);
            // (the reader's next chunk is queued on the same stream: it overwrites the working buffers behind these)
            k.short_form = true;
        }
    }
    if (!short_form) {
        k.text = r->text; k.idx_u32 = r->idx_u32; k.base_sz = r->base_sz; k.text_sz = r->text_sz;
        r->text = Buf(); r->idx_u32 = Buf(); r->base_sz = Buf(); r->text_sz = Buf();
        r->kept_bytes += k.text.bytes + k.idx_u32.bytes + k.base_sz.bytes + k.text_sz.bytes;
    }
    r->kept.push_back(k);
    r->have_chunk = false;
    r->packed_is_exact = false;
}

}  // namespace

extern "C" {

int kbbq_fastq_reader_create(int32_t device, kbbq_fastq_reader **out) {
    if (!out) return fail(KBBQ_EINVAL, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(KBBQ_ENODEV, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(KBBQ_ENODEV, "device %d of %d", device, ndev);
    KbbqDeviceGuard guard(device);
    HIP_TRY(guard.err);
    kbbq_fastq_reader *r = new kbbq_fastq_reader;
    r->device = device;
    r->h_meta.host = r->h_small.host = true;
    hipError_t he = hipStreamCreateWithFlags(&r->st, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipEventCreate(&r->t0);
    if (he == hipSuccess) he = hipEventCreate(&r->t1);
    if (he == hipSuccess) he = hipEventCreate(&r->t2);
    if (he != hipSuccess) {
        kbbq_fastq_reader_destroy(r);
        return fail(KBBQ_EIO, "creating the reader's stream: %s", hipGetErrorString(he));
    }
    *out = r;
    return KBBQ_OK;
}

void kbbq_fastq_reader_destroy(kbbq_fastq_reader *r) {
    if (!r) return;
    KbbqDeviceGuard guard(r->device);
    if (r->st) (void)hipStreamSynchronize(r->st);
    Buf *all[] = {&r->comp, &r->text, &r->status, &r->blk_meta, &r->h_meta, &r->tile_counts, &r->tile_sums, &r->nl_pos, &r->idx_u32,
                  &r->idx_second, &r->base_sz, &r->text_sz, &r->flags, &r->carry, &r->h_small, &r->seq_text, &r->counter};
    for (Buf *b : all) b->release();
    r->pre.release();
    release_kept(r);
    hipEvent_t evs[] = {r->t0, r->t1, r->t2};
    for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    if (r->st) (void)hipStreamDestroy(r->st);
    delete r;
}

int kbbq_fastq_reader_rewind(kbbq_fastq_reader *r) {
    if (!r) return fail(KBBQ_EINVAL, "null argument");
    KbbqDeviceGuard guard(r->device);      // stash_current launches kernels and may allocate
    HIP_TRY(guard.err);
    stash_current(r);
    r->keeping = false;      // what was kept stays; a second scan keeps nothing more
    r->selected = -1;
    r->carry_bytes = 0;
    r->have_chunk = false;
    return KBBQ_OK;
}

int kbbq_fastq_reader_keep(kbbq_fastq_reader *r, int32_t on) {
    if (!r) return fail(KBBQ_EINVAL, "null argument");
    KbbqDeviceGuard guard(r->device);
    HIP_TRY(guard.err);
    if (on) {
        if (r->have_chunk || !r->kept.empty()) return fail(KBBQ_ESTATE, "keeping starts before the first chunk of a scan");
        r->keeping = true;
    } else {
        HIP_TRY(hipStreamSynchronize(r->st));
        release_kept(r);
        r->keeping = false;
    }
    return KBBQ_OK;
}

int kbbq_fastq_reader_kept(kbbq_fastq_reader *r, uint64_t *n_chunks, uint64_t *n_bytes) {
    if (!r) return fail(KBBQ_EINVAL, "null argument");
    if (n_chunks) *n_chunks = r->kept.size() + ((r->keeping && r->selected < 0 && r->have_chunk && r->n_records) ? 1 : 0);
    if (n_bytes) *n_bytes = r->kept_bytes + ((r->keeping && r->selected < 0 && r->have_chunk && r->n_records)
                                                 ? r->text.bytes + r->idx_u32.bytes + r->base_sz.bytes + r->text_sz.bytes : 0);
    return KBBQ_OK;
}

int kbbq_fastq_reader_select(kbbq_fastq_reader *r, uint64_t i, kbbq_fastq_chunk *info) {
    if (!r) return fail(KBBQ_EINVAL, "null argument");
    KbbqDeviceGuard guard(r->device);
    HIP_TRY(guard.err);
    stash_current(r);
    if (i >= r->kept.size()) return fail(KBBQ_EINVAL, "kept chunk %llu of %llu", (unsigned long long)i, (unsigned long long)r->kept.size());
    const kbbq_fastq_reader::Kept &k = r->kept[i];
    r->selected = (int64_t)i;
    r->have_chunk = true;
    r->att_bases = r->att_nmask = r->att_offcase = nullptr;
    r->text_bytes = k.text_bytes; r->n_records = k.n_records; r->n_bases = k.n_bases;
    r->longest = k.longest; r->shortest = k.shortest;
    r->out_text_bytes = k.out_text_bytes;
    if (info) {
        memset(info, 0, sizeof *info);
        info->n_records = k.n_records; info->n_bases = k.n_bases; info->longest = k.longest; info->shortest = k.shortest;
        info->text_bytes = k.text_bytes;
    }
    return KBBQ_OK;
}

int kbbq_fastq_reader_chunk(kbbq_fastq_reader *r, const uint8_t *file_bytes, uint64_t n_bytes, int32_t last, kbbq_fastq_chunk *info) {
    if (!r || !info || (!file_bytes && n_bytes)) return fail(KBBQ_EINVAL, "bad argument");
    KbbqDeviceGuard guard(r->device);
    HIP_TRY(guard.err);
    memset(info, 0, sizeof *info);
    stash_current(r);
    r->selected = -1;
    r->have_chunk = false;
    r->packed_is_exact = false;
    // ---- the block boundaries: hop from header to header (RFC 1952 member with the 'BC' extra subfield, SAM spec 4.1)
    std::vector<uint64_t> c_off, o_off;
    std::vector<uint32_t> c_len, o_len;
    uint64_t at = 0, text = r->carry_bytes;
    const uint64_t text_cap = 3500000000ull;      // record offsets travel in 32 bits
    while (at + 18 <= n_bytes) {
        const uint8_t *h = file_bytes + at;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) { info->flags |= 1; break; }      // not BGZF
        const uint32_t xlen = h[10] | (h[11] << 8);
        if (at + 12 + xlen > n_bytes) break;
        uint32_t bsize = 0;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            const uint8_t *sf = h + 12 + x;
            const uint32_t slen = sf[2] | (sf[3] << 8);
            if (sf[0] == 66 && sf[1] == 67 && slen == 2 && x + 6 <= xlen) bsize = (sf[4] | (sf[5] << 8)) + 1u;
            x += 4 + slen;
        }
        if (!bsize || bsize < 12 + xlen + 8) { info->flags |= 1; break; }
        if (at + bsize > n_bytes) break;      // the chunk ends inside this block
        const uint8_t *tail = h + bsize - 8;
        const uint32_t isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
        if (isize > 65536) { info->flags |= 1; break; }
        if (text + isize > text_cap) break;
        if (isize) {
            c_off.push_back(at + 12 + xlen);
            c_len.push_back(bsize - (12 + xlen) - 8);
            o_off.push_back(text);
            o_len.push_back(isize);
            text += isize;
        }
        at += bsize;
    }
    info->consumed = at;
    info->n_blocks = (uint32_t)c_off.size();
    if (info->flags & 1) return KBBQ_OK;
    if (at == 0 && n_bytes && !last && c_off.empty()) return fail(KBBQ_EINVAL, "the chunk holds no complete BGZF block");
    const uint32_t nb = (uint32_t)c_off.size();
    int rc;
    // (while chunks are kept, a buffer that no longer fits gives up the kept ones: pass 4 then inflates the file again)
    auto reserve_or_drop = [&](Buf &b, size_t need) -> int {
        int rc2 = b.reserve(need);
        if (rc2 == KBBQ_ENOMEM && (r->keeping || !r->kept.empty())) {
            (void)hipGetLastError();
            release_kept(r);
            r->keeping = false;
            rc2 = b.reserve(need);
        }
        return rc2;
    };
    // ---- compressed bytes and block table to the device, inflate
    void *d_comp = r->pre.take(file_bytes, n_bytes, r->st);      // copied ahead by the caller's I/O thread?
    if (!d_comp && (rc = r->comp.reserve(at + 4096))) return rc;
    if ((rc = reserve_or_drop(r->text, text + 4096))) return rc;
    if ((rc = r->status.reserve((size_t)nb * 4 + 64))) return rc;
    const size_t meta_bytes = (size_t)nb * 24 + 64;
    if ((rc = r->blk_meta.reserve(meta_bytes))) return rc;
    if ((rc = r->h_meta.reserve(meta_bytes))) return rc;
    if ((rc = r->h_small.reserve(4096))) return rc;
    uint64_t *hm = (uint64_t *)r->h_meta.p;
    if (nb) {
        memcpy(hm, c_off.data(), (size_t)nb * 8);
        memcpy(hm + nb, o_off.data(), (size_t)nb * 8);
        memcpy((uint32_t *)(hm + 2 * (size_t)nb), c_len.data(), (size_t)nb * 4);
        memcpy((uint32_t *)(hm + 2 * (size_t)nb) + nb, o_len.data(), (size_t)nb * 4);
    }
    if (r->carry_bytes) HIP_TRY(hipMemcpyAsync(r->text.p, r->carry.p, r->carry_bytes, hipMemcpyDeviceToDevice, r->st));
    if (nb) {
        if (!d_comp) {
            d_comp = r->comp.p;
            HIP_TRY(hipMemcpyAsync(d_comp, file_bytes, at, hipMemcpyHostToDevice, r->st));
            HIP_TRY(hipMemsetAsync((char *)d_comp + at, 0, 4096, r->st));
        }
        HIP_TRY(hipMemcpyAsync(r->blk_meta.p, hm, (size_t)nb * 24, hipMemcpyHostToDevice, r->st));
    }
    HIP_TRY(hipEventRecord(r->t0, r->st));      // (the kernel alone: the upload of the compressed bytes is not in its time)
    if (nb) {
        InflateArgs A;
        A.comp = (const uint8_t *)d_comp;
        A.c_off = (const uint64_t *)r->blk_meta.p;
        A.o_off = A.c_off + nb;
        A.c_len = (const uint32_t *)(A.c_off + 2 * (size_t)nb);
        A.o_len = A.c_len + nb;
        A.out = (uint8_t *)r->text.p;
        A.n_blocks = nb;
        A.status = (uint32_t *)r->status.p;
        // as many wavefronts as stay resident (5.6 KB of LDS each: 2 KB ring + 9-bit table): blocks are handed out round-robin, a second round of
        // workgroups would only queue behind the first
        if (!r->inflate_grid && (rc = inflate_resident_waves(r->device, &r->inflate_grid))) return rc;
        const unsigned grid = std::min<unsigned>(nb, r->inflate_grid);
        hipLaunchKernelGGL(k_inflate, dim3(grid), dim3(64 * INF_WAVES), 0, r->st, A);
        HIP_TRY(hipGetLastError());
        // the blocks' checksums, as bgzf_read verifies them
        hipLaunchKernelGGL(k_block_crc, dim3(std::min<unsigned>((nb + 3) / 4, 256 * 16)), dim3(256), 0, r->st, A);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemsetAsync((char *)r->text.p + text, 0, 64, r->st));
    HIP_TRY(hipEventRecord(r->t1, r->st));
    // ---- lines
    const uint64_t n_tiles = (text + NL_TILE - 1) / NL_TILE;
    uint64_t n_lines = 0;
    if (text) {
        if ((rc = r->tile_counts.reserve((n_tiles + 2) * 8))) return rc;
        uint64_t *tc = (uint64_t *)r->tile_counts.p;
        hipLaunchKernelGGL(k_count_newlines, dim3((unsigned)n_tiles), dim3(256), 0, r->st, (const uint8_t *)r->text.p, text, tc);
        HIP_TRY(hipGetLastError());
        if ((rc = device_scan(r, tc, n_tiles, tc + n_tiles))) return rc;
        HIP_TRY(hipMemcpyAsync(r->h_small.p, tc + n_tiles, 8, hipMemcpyDeviceToHost, r->st));
        // (and whether every block inflated: one word per block, OR-ed on the host -- a few thousand words)
        HIP_TRY(hipStreamSynchronize(r->st));
        n_lines = *(const uint64_t *)r->h_small.p;
        if (nb) {
            std::vector<uint32_t> stt(nb);
            HIP_TRY(hipMemcpy(stt.data(), r->status.p, (size_t)nb * 4, hipMemcpyDeviceToHost));
            for (uint32_t b = 0; b < nb; ++b) {
                if (stt[b] == INF_BAD_CRC) return fail(KBBQ_EIO, "BGZF block %u of the chunk: CRC32 checksum mismatch", b);
                if (stt[b] != INF_OK) return fail(KBBQ_EIO, "BGZF block %u of the chunk does not inflate (code %u)", b, stt[b]);
            }
        }
    }
    const uint64_t n_rec = n_lines / 4;
    info->text_bytes = text - r->carry_bytes;
    uint64_t rec_end = 0;      // first byte behind the last complete record
    if (n_rec) {
        if ((rc = r->nl_pos.reserve((n_lines + 4) * 4))) return rc;
        hipLaunchKernelGGL(k_newline_positions, dim3((unsigned)n_tiles), dim3(256), 0, r->st, (const uint8_t *)r->text.p, text,
                           (const uint64_t *)r->tile_counts.p, (uint32_t *)r->nl_pos.p, n_lines);
        HIP_TRY(hipGetLastError());
        // ---- records
        if ((rc = reserve_or_drop(r->idx_u32, n_rec * 7 * 4))) return rc;
        if ((rc = r->idx_second.reserve(n_rec))) return rc;
        if ((rc = reserve_or_drop(r->base_sz, (n_rec + 2) * 8))) return rc;
        if ((rc = reserve_or_drop(r->text_sz, (n_rec + 2) * 8))) return rc;
        if ((rc = r->flags.reserve(64))) return rc;
        const uint32_t init_flags[4] = {0, 0, 0xFFFFFFFFu, 0};
        HIP_TRY(hipMemcpyAsync(r->flags.p, init_flags, 16, hipMemcpyHostToDevice, r->st));
        const FastqIndex X = index_of(r, n_rec);
        hipLaunchKernelGGL(k_fastq_records, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, r->st, (const uint8_t *)r->text.p,
                           (const uint32_t *)r->nl_pos.p, n_rec, X);
        HIP_TRY(hipGetLastError());
        if ((rc = device_scan(r, X.base_sz, n_rec, X.base_sz + n_rec))) return rc;
        if ((rc = device_scan(r, X.text_sz, n_rec, X.text_sz + n_rec))) return rc;
        uint64_t *hs = (uint64_t *)r->h_small.p;
        HIP_TRY(hipMemcpyAsync(hs, X.base_sz + n_rec, 8, hipMemcpyDeviceToHost, r->st));
        HIP_TRY(hipMemcpyAsync(hs + 1, X.flags, 16, hipMemcpyDeviceToHost, r->st));
        HIP_TRY(hipMemcpyAsync(hs + 4, (const uint32_t *)r->nl_pos.p + (4 * n_rec - 1), 4, hipMemcpyDeviceToHost, r->st));
        HIP_TRY(hipMemcpyAsync(hs + 5, X.text_sz + n_rec, 8, hipMemcpyDeviceToHost, r->st));
        HIP_TRY(hipStreamSynchronize(r->st));
        r->n_bases = hs[0];
        r->out_text_bytes = hs[5];
        const uint32_t *fl = (const uint32_t *)(hs + 1);
        info->flags |= fl[0];
        r->longest = fl[1];
        r->shortest = fl[2];
        rec_end = (uint64_t)(*(const uint32_t *)(hs + 4)) + 1;
    }
    HIP_TRY(hipEventRecord(r->t2, r->st));
    // ---- what the chunk's end cut: kept for the next chunk
    const uint64_t left = text - rec_end;
    if (left) {
        if (last) info->flags |= 4;      // the file ends inside a record (or without a final newline): the serial reader's case
        if ((rc = r->carry.reserve(left + 64))) return rc;
        HIP_TRY(hipMemcpyAsync(r->carry.p, (const char *)r->text.p + rec_end, left, hipMemcpyDeviceToDevice, r->st));
    }
    HIP_TRY(hipStreamSynchronize(r->st));
    r->carry_bytes = left;
    r->text_bytes = text;
    r->n_records = n_rec;
    r->have_chunk = true;
    info->n_records = n_rec;
    info->n_bases = n_rec ? r->n_bases : 0;
    info->longest = n_rec ? r->longest : 0;
    info->shortest = n_rec ? r->shortest : 0;
    float a = 0, b = 0;
    if (hipEventElapsedTime(&a, r->t0, r->t1) == hipSuccess) r->ms_inflate += a;
    if (hipEventElapsedTime(&b, r->t1, r->t2) == hipSuccess) r->ms_index += b;
    return KBBQ_OK;
}

int kbbq_fastq_reader_inflate(kbbq_fastq_reader *r, const uint8_t *file_bytes, uint64_t n_bytes, uint8_t *host_out, uint64_t capacity,
                              uint64_t *consumed, uint64_t *produced) {
    if (!r || !host_out || !consumed || !produced || (!file_bytes && n_bytes)) return fail(KBBQ_EINVAL, "bad argument");
    KbbqDeviceGuard guard(r->device);
    HIP_TRY(guard.err);
    stash_current(r);
    r->selected = -1;
    r->have_chunk = false;
    *consumed = *produced = 0;
    // whole blocks while their inflated bytes fit
    std::vector<uint64_t> c_off, o_off;
    std::vector<uint32_t> c_len, o_len;
    uint64_t at = 0, text = 0;
    while (at + 18 <= n_bytes) {
        const uint8_t *h = file_bytes + at;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return fail(KBBQ_EIO, "not a BGZF block at byte %llu of the piece", (unsigned long long)at);
        const uint32_t xlen = h[10] | (h[11] << 8);
        if (at + 12 + xlen > n_bytes) break;
        uint32_t bsize = 0;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            const uint8_t *sf = h + 12 + x;
            const uint32_t slen = sf[2] | (sf[3] << 8);
            if (sf[0] == 66 && sf[1] == 67 && slen == 2 && x + 6 <= xlen) bsize = (sf[4] | (sf[5] << 8)) + 1u;
            x += 4 + slen;
        }
        if (!bsize || bsize < 12 + xlen + 8) return fail(KBBQ_EIO, "a BGZF header without its BC field at byte %llu of the piece", (unsigned long long)at);
        if (at + bsize > n_bytes) break;
        const uint8_t *tail = h + bsize - 8;
        const uint32_t isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
        if (isize > 65536) return fail(KBBQ_EIO, "a BGZF block of %u bytes", isize);
        if (text + isize > capacity) break;
        if (isize) {
            c_off.push_back(at + 12 + xlen);
            c_len.push_back(bsize - (12 + xlen) - 8);
            o_off.push_back(text);
            o_len.push_back(isize);
            text += isize;
        }
        at += bsize;
    }
    *consumed = at;
    *produced = text;
    const uint32_t nb = (uint32_t)c_off.size();
    if (!nb) return KBBQ_OK;
    int rc;
    if ((rc = r->comp.reserve(at + 4096))) return rc;
    if ((rc = r->text.reserve(text + 4096))) return rc;
    if ((rc = r->status.reserve((size_t)nb * 4 + 64))) return rc;
    const size_t meta_bytes = (size_t)nb * 24 + 64;
    if ((rc = r->blk_meta.reserve(meta_bytes))) return rc;
    if ((rc = r->h_meta.reserve(std::max<size_t>(meta_bytes, (size_t)nb * 4 + 64)))) return rc;
    uint64_t *hm = (uint64_t *)r->h_meta.p;
    memcpy(hm, c_off.data(), (size_t)nb * 8);
    memcpy(hm + nb, o_off.data(), (size_t)nb * 8);
    memcpy((uint32_t *)(hm + 2 * (size_t)nb), c_len.data(), (size_t)nb * 4);
    memcpy((uint32_t *)(hm + 2 * (size_t)nb) + nb, o_len.data(), (size_t)nb * 4);
    HIP_TRY(hipMemcpyAsync(r->comp.p, file_bytes, at, hipMemcpyHostToDevice, r->st));
    HIP_TRY(hipMemsetAsync((char *)r->comp.p + at, 0, 4096, r->st));
    HIP_TRY(hipMemcpyAsync(r->blk_meta.p, hm, (size_t)nb * 24, hipMemcpyHostToDevice, r->st));
    HIP_TRY(hipEventRecord(r->t0, r->st));
    InflateArgs A;
    A.comp = (const uint8_t *)r->comp.p;
    A.c_off = (const uint64_t *)r->blk_meta.p;
    A.o_off = A.c_off + nb;
    A.c_len = (const uint32_t *)(A.c_off + 2 * (size_t)nb);
    A.o_len = A.c_len + nb;
    A.out = (uint8_t *)r->text.p;
    A.n_blocks = nb;
    A.status = (uint32_t *)r->status.p;
    if (!r->inflate_grid && (rc = inflate_resident_waves(r->device, &r->inflate_grid))) return rc;
    hipLaunchKernelGGL(k_inflate, dim3(std::min<unsigned>(nb, r->inflate_grid)), dim3(64 * INF_WAVES), 0, r->st, A);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_block_crc, dim3(std::min<unsigned>((nb + 3) / 4, 256 * 16)), dim3(256), 0, r->st, A);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(r->t1, r->st));
    HIP_TRY(hipMemcpyAsync(host_out, r->text.p, text, hipMemcpyDeviceToHost, r->st));
    // (the status words travel through the page-locked meta buffer, which the block table no longer needs)
    HIP_TRY(hipMemcpyAsync(r->h_meta.p, r->status.p, (size_t)nb * 4, hipMemcpyDeviceToHost, r->st));
    HIP_TRY(hipStreamSynchronize(r->st));
    const uint32_t *stt = (const uint32_t *)r->h_meta.p;
    for (uint32_t b = 0; b < nb; ++b) {
        if (stt[b] == INF_BAD_CRC) return fail(KBBQ_EIO, "BGZF block %u of the piece: CRC32 checksum mismatch", b);
        if (stt[b] != INF_OK) return fail(KBBQ_EIO, "BGZF block %u of the piece does not inflate (code %u)", b, stt[b]);
    }
    float a = 0;
    if (hipEventElapsedTime(&a, r->t0, r->t1) == hipSuccess) r->ms_inflate += a;
    return KBBQ_OK;
}

int kbbq_fastq_reader_batch(kbbq_fastq_reader *r, kbbq_reads *dev) {
    if (!r || !dev) return fail(KBBQ_EINVAL, "null argument");
    if (!r->have_chunk || !r->n_records || r->selected >= 0) return fail(KBBQ_ESTATE, "no records in the current chunk");
    KbbqDeviceGuard guard(r->device);
    HIP_TRY(guard.err);
    const uint64_t n = r->n_records, nbases = r->n_bases;
    const FastqIndex X = index_of(r, n);
    memset(dev, 0, sizeof *dev);
    dev->n_reads = n;
    dev->n_bases = nbases;
    dev->on_device = 1;
    void *b = nullptr, *m = nullptr, *q = nullptr, *oc = nullptr, *off = nullptr, *fl = nullptr;
    auto release = [&]() { void *all[] = {b, m, q, oc, off, fl}; for (void *x : all) (void)hipFree(x); };
    int rc0;
    const uint64_t words = nbases / 64 + 1;
    if ((rc0 = r->seq_text.reserve(nbases + 64))) return rc0;
    // [0..1] the two counts of k_pack_text, behind them the off-case words: nearly every chunk has none, and an array
    // allocated and freed again per chunk was a hipMalloc (which clears) and a hipFree (which waits for the device) for nothing
    if ((rc0 = r->counter.reserve((words + 4) * 8 + 64))) return rc0;
    void *seq_text = r->seq_text.p, *cnt = r->counter.p;
    void *oc_scratch = (char *)r->counter.p + 16;
#define RB_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { release(); return fail(_e == hipErrorOutOfMemory ? KBBQ_ENOMEM : KBBQ_EIO, "%s: %s", #expr, hipGetErrorString(_e)); } } while (0)
    RB_TRY(hipMalloc(&b, (2 * words + 2) * 8));
    RB_TRY(hipMalloc(&m, (words + 2) * 8));
    RB_TRY(hipMalloc(&q, nbases + 16));
    RB_TRY(hipMalloc(&fl, n));
    const bool uniform = r->longest == r->shortest;
    if (!uniform) RB_TRY(hipMalloc(&off, (n + 1) * 8));
    RB_TRY(hipMemsetAsync(cnt, 0, 16, r->st));
    RB_TRY(hipMemsetAsync((char *)b + 2 * words * 8, 0, 16, r->st));
    RB_TRY(hipMemsetAsync((char *)m + words * 8, 0, 16, r->st));
    RB_TRY(hipMemsetAsync((char *)oc_scratch + words * 8, 0, 16, r->st));
    RB_TRY(hipMemsetAsync((char *)q + nbases, 0, 16, r->st));
    hipLaunchKernelGGL(k_fastq_gather, dim3((unsigned)std::min<uint64_t>((n + 3) / 4, 256 * 32)), dim3(256), 0, r->st, (const uint8_t *)r->text.p, X,
                       (const uint64_t *)X.base_sz, n, (uint8_t *)seq_text, (uint8_t *)q);
    hipLaunchKernelGGL(k_pack_text, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, r->st, (const uint8_t *)seq_text, nbases, (uint64_t *)b,
                       (uint64_t *)m, (uint64_t *)oc_scratch, (unsigned long long *)cnt);
    RB_TRY(hipGetLastError());
    RB_TRY(hipMemcpyAsync(fl, X.second, n, hipMemcpyDeviceToDevice, r->st));
    if (!uniform) RB_TRY(hipMemcpyAsync(off, X.base_sz, (n + 1) * 8, hipMemcpyDeviceToDevice, r->st));
    unsigned long long counts[2] = {0, 0};      // off-case bases; characters the packed form cannot give back
    RB_TRY(hipMemcpyAsync(counts, cnt, 16, hipMemcpyDeviceToHost, r->st));
    RB_TRY(hipStreamSynchronize(r->st));
    const unsigned long long n_off = counts[0];
    if (n_off) {      // soft-masked text: the batch gets its off-case bits
        RB_TRY(hipMalloc(&oc, (words + 2) * 8));
        RB_TRY(hipMemcpyAsync(oc, oc_scratch, (words + 2) * 8, hipMemcpyDeviceToDevice, r->st));
        RB_TRY(hipStreamSynchronize(r->st));
    }
#undef RB_TRY
    r->packed_is_exact = counts[1] == 0;
    dev->bases = (const uint64_t *)b;
    dev->nmask = (const uint64_t *)m;
    dev->qual = (const uint8_t *)q;
    dev->offsets = (const uint64_t *)off;
    dev->flags = (const uint8_t *)fl;
    dev->rg = nullptr;
    dev->read_len = uniform ? r->longest : 0;
    dev->offcase = (const uint64_t *)oc;
    return KBBQ_OK;
}

int kbbq_fastq_reader_write(kbbq_fastq_reader *r, kbbq_bgzf *z, const uint8_t *d_qual, void *after_stream) {
    if (!r || !z || !d_qual) return fail(KBBQ_EINVAL, "null argument");
    if (!r->have_chunk || !r->n_records) return fail(KBBQ_ESTATE, "no records in the current chunk");
    if (r->device != z->device) return fail(KBBQ_EINVAL, "reader and writer are on different devices");
    KbbqDeviceGuard guard(z->device);
    HIP_TRY(guard.err);
    if (r->selected >= 0 && r->kept[(size_t)r->selected].short_form && !r->att_bases)
        return fail(KBBQ_ESTATE, "the chunk was kept without its sequence text: attach its batch first (kbbq_fastq_reader_attach)");
    Submission *sp;
    int rc = begin_submission(z, after_stream, &sp);
    if (rc) return rc;
    Submission &s = *sp;
    const uint64_t n = r->n_records;
    const bool from_kept = r->selected >= 0;
    const kbbq_fastq_reader::Kept *k = from_kept ? &r->kept[(size_t)r->selected] : nullptr;
    const FastqIndex X = from_kept ? index_from(k->idx_u32.p, nullptr, k->base_sz.p, k->text_sz.p, nullptr, n) : index_of(r, n);
    const uint8_t *text = (const uint8_t *)(from_kept ? k->text.p : r->text.p);
    const uint64_t t = r->out_text_bytes;      // (the scanned sizes' total, read back with the chunk's other counts)
    s.n = t;
    s.formatted = true;
    if ((rc = s.payload.reserve(t + 16))) return rc;
    HIP_TRY(hipMemsetAsync((char *)s.payload.p + t, 0, 16, z->st));
    HIP_TRY(hipEventRecord(s.t0, z->st));
    if (from_kept && k->short_form)
        hipLaunchKernelGGL(k_fastq_text_packed, dim3((unsigned)std::min<uint64_t>((n + 3) / 4, 256 * 32)), dim3(256), 0, z->st, (const uint8_t *)k->names.p,
                           (const uint32_t *)k->lens.p, (const uint64_t *)k->text_sz.p, (const uint64_t *)k->base_sz.p, r->att_bases, r->att_nmask,
                           r->att_offcase, d_qual, n, (uint8_t *)s.payload.p);
    else
        hipLaunchKernelGGL(k_fastq_text_indexed, dim3((unsigned)std::min<uint64_t>((n + 3) / 4, 256 * 32)), dim3(256), 0, z->st, text, X,
                           (const uint64_t *)X.text_sz, (const uint64_t *)X.base_sz, d_qual, n, (uint8_t *)s.payload.p);
    HIP_TRY(hipGetLastError());
    if ((rc = launch_deflate(z, s))) return rc;
    // the reader's live text and index are read by the kernel just queued: the next kbbq_fastq_reader_chunk must not
    // overwrite them before it has run (a kept chunk's buffers stay as they are)
    if (!from_kept) HIP_TRY(hipEventSynchronize(s.t1));
    return KBBQ_OK;
}

int kbbq_fastq_reader_attach(kbbq_fastq_reader *r, const kbbq_reads *batch) {
    if (!r || !batch) return fail(KBBQ_EINVAL, "null argument");
    if (!batch->on_device || !batch->bases || !batch->nmask) return fail(KBBQ_EINVAL, "not a device batch");
    if (r->selected < 0) return fail(KBBQ_ESTATE, "no kept chunk is selected");
    const kbbq_fastq_reader::Kept &k = r->kept[(size_t)r->selected];
    if (batch->n_reads != k.n_records || batch->n_bases != k.n_bases) return fail(KBBQ_EINVAL, "the batch is not this chunk's");
    r->att_bases = batch->bases;
    r->att_nmask = batch->nmask;
    r->att_offcase = batch->offcase;
    return KBBQ_OK;
}

// A host batch made resident with its bases packed ON the device: the sequence characters travel as they are (1 byte per
// base) and k_pack_text makes the 2-bit words, the non-ACGT mask and the off-case bits there -- kbbq_pack_bases_case's
// table, 40x its rate -- so the host thread that assembles the batches does not spend a third of its time packing.
int kbbq_reads_upload_text(kbbq_engine *e, const kbbq_reads *host, const uint8_t *seq_text, kbbq_reads *dev) {
    if (!host || !seq_text || !dev) return fail(KBBQ_EINVAL, "null argument");
    if (host->on_device) return fail(KBBQ_EINVAL, "batch is already on the device");
    kbbq_reads h2 = *host;
    h2.bases = nullptr; h2.nmask = nullptr; h2.offcase = nullptr;
    int rc = kbbq_reads_upload(e, &h2, dev);      // qualities, offsets, flags, read groups
    if (rc) return rc;
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    const uint64_t nbases = host->n_bases, words = nbases / 64 + 1;
    void *b = nullptr, *m = nullptr, *oc = nullptr, *text = nullptr, *cnt = nullptr;
    auto give_up = [&](hipError_t he, const char *what) {
        void *all[] = {b, m, oc, text, cnt};
        for (void *x : all) (void)hipFree(x);
        kbbq_reads_free(e, dev);
        return fail(he == hipErrorOutOfMemory ? KBBQ_ENOMEM : KBBQ_EIO, "%s: %s", what, hipGetErrorString(he));
    };
    hipError_t he;
    if ((he = hipMalloc(&b, (2 * words + 2) * 8)) != hipSuccess) return give_up(he, "bases");
    if ((he = hipMalloc(&m, (words + 2) * 8)) != hipSuccess) return give_up(he, "mask");
    if ((he = hipMalloc(&oc, (words + 2) * 8)) != hipSuccess) return give_up(he, "off-case bits");
    if ((he = hipMalloc(&text, nbases + 64)) != hipSuccess) return give_up(he, "sequence text");
    if ((he = hipMalloc(&cnt, 16)) != hipSuccess) return give_up(he, "counter");
    hipStream_t st = nullptr;      // the null stream: ordered behind kbbq_reads_upload's copies, which it waited for
    if ((he = hipMemsetAsync(cnt, 0, 16, st)) != hipSuccess) return give_up(he, "memset");
    (void)hipMemsetAsync((char *)b + 2 * words * 8, 0, 16, st);
    (void)hipMemsetAsync((char *)m + words * 8, 0, 16, st);
    (void)hipMemsetAsync((char *)oc + words * 8, 0, 16, st);
    if ((he = hipMemcpyAsync(text, seq_text, nbases, hipMemcpyHostToDevice, st)) != hipSuccess) return give_up(he, "upload of the sequence text");
    hipLaunchKernelGGL(k_pack_text, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, (const uint8_t *)text, nbases, (uint64_t *)b, (uint64_t *)m,
                       (uint64_t *)oc, (unsigned long long *)cnt);
    if ((he = hipGetLastError()) != hipSuccess) return give_up(he, "k_pack_text");
    unsigned long long counts[2] = {0, 0};
    if ((he = hipMemcpyAsync(counts, cnt, 16, hipMemcpyDeviceToHost, st)) != hipSuccess) return give_up(he, "counts");
    if ((he = hipStreamSynchronize(st)) != hipSuccess) return give_up(he, "packing");
    (void)hipFree(text);
    (void)hipFree(cnt);
    if (!counts[0]) { (void)hipFree(oc); oc = nullptr; }
    dev->bases = (const uint64_t *)b;
    dev->nmask = (const uint64_t *)m;
    dev->offcase = (const uint64_t *)oc;
    return KBBQ_OK;
}

int kbbq_fastq_reader_preload(kbbq_fastq_reader *r, const uint8_t *file_bytes, uint64_t n_bytes, uint64_t front_room) {
    if (!r || !file_bytes || !n_bytes) return fail(KBBQ_EINVAL, "bad argument");
    KbbqDeviceGuard guard(r->device);
    HIP_TRY(guard.err);
    return r->pre.start(r->device, file_bytes, n_bytes, front_room);
}

int kbbq_fastq_reader_kernel_ms(kbbq_fastq_reader *r, double *inflate_ms, double *index_ms) {
    if (!r) return fail(KBBQ_EINVAL, "null argument");
    if (inflate_ms) *inflate_ms = r->ms_inflate;
    if (index_ms) *index_ms = r->ms_index;
    return KBBQ_OK;
}

}  // extern "C"

// ============================================================ the input side: BAM read on the device
#include "bam_reader.h"
