// engine.hip -- the engine object and the C ABI of include/kbbq_engine.h: filters, scratch, streams, one
// entry point per pass that launches the kernels of kernels.h (MI355X, gfx950).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kbbq_engine.h"
#include "abi_internal.h"
#include "correct.h"
#include "correct_wave.h"
#include "device_common.h"
#include "host_model.h"

using namespace kbbq;

// ============================================================ error plumbing
static thread_local char g_err[512] = "";
int kbbq_fail(int code, const char *fmt, ...) {      // (abi_internal.h: shared with bgzf_device.hip)
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define fail kbbq_fail
#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return fail(_e == hipErrorOutOfMemory ? KBBQ_ENOMEM : KBBQ_EIO, "%s: %s (%s:%d)", #expr,   \
                        hipGetErrorString(_e), __FILE__, __LINE__);                                    \
    } while (0)

#include "kernels.h"
#include "bucket.h"
#include "long_reads.h"

// ============================================================ engine object

struct FilterHost {
    FilterSpec spec;
    uint64_t *d_table = nullptr;      // n_blocks x 2 words: the engine's 128-bit blocks
    uint64_t *d_patterns = nullptr;   // 65536 x 2 words
    unsigned long long *d_inserted = nullptr;
    uint64_t table_bytes() const { return spec.n_blocks * kEngineBlockBytes; }
    FiltDev dev() const {
        FiltDev f;
        f.table = reinterpret_cast<ulonglong2 *>(d_table);
        f.patterns = reinterpret_cast<const ulonglong2 *>(d_patterns);
        f.n_blocks = spec.n_blocks;
        const ModMagic mm = make_mod_magic(spec.n_blocks);
        f.mod_magic = mm.m64;
        f.mod_m32 = mm.m32;
        f.salt0 = spec.salt[0];
        f.salt1 = spec.salt[1];
        return f;
    }
};

struct ProfileSlot {
    std::string name;
    uint64_t launches = 0;
    double ms = 0;
};
struct PendingEvent {
    int slot;
    hipEvent_t a, b;
};

// Behavioural switches, fixed when the engine is created: a KBBQ_F_* flag of kbbq_params.flags, or -- where no flag
// says otherwise -- the environment variable of the same meaning (README.md) as it stands at kbbq_engine_create.
// None changes a result.
struct Options {
    bool no_overlap = false;          // KBBQ_F_NO_OVERLAP / KBBQ_NO_OVERLAP: every kernel in order on one stream
    bool no_fastpath = false;         // KBBQ_F_NO_FASTPATH / KBBQ_NO_FASTPATH: every read with untrusted k-mers takes the walk
    bool lane_walk = false;           // KBBQ_F_LANE_WALK / KBBQ_CORRECT=lane: the one-read-per-lane form of the walk
    bool no_pass4_pipeline = false;   // KBBQ_F_NO_PASS4_PIPELINE / KBBQ_NO_PASS4_PIPELINE: pass 4 of a host batch in one piece
    int pass2_side = 2;               // KBBQ_F_PASS2_INORDER / KBBQ_PASS2_SIDE=0: pass 2 in order; 1: its insert side (emit, split, apply) on the
                                      // side stream beside k_infer; 2: only the emits there, a flush on the engine's stream between two k_infer.
                                      // profiles/r04_ab_pass2_side_modes.json: pass 2 1868 / 1836 / 1816 ms -- k_apply beside k_infer takes 87 ms
                                      // instead of 16.7 (both wait for the L2's request path), the ALU-bound emit costs k_infer half its own time
    bool tally_general = false;       // KBBQ_TALLY_GENERAL: the general tally kernel for every batch shape (A/B)
    bool infer_subset = true;         // KBBQ_INFER_SUBSET=0 / kbbq_engine_tune("infer_subset", 0): k_infer makes every lookup at once (round 3's form)
    bool debug_bucket = false;        // KBBQ_DEBUG_BUCKET: one stderr line per flush of the bucketed inserts
    int bucket = -1;                  // KBBQ_F_BUCKET_ON / _OFF, KBBQ_BUCKET=1/0: bucketed / direct inserts; -1: by filter size
    uint64_t bucket_records = 0;      // KBBQ_BUCKET_RECORDS: records gathered per flush (0: a share of the free HBM)
    uint64_t pass4_piece = 0;         // KBBQ_PASS4_PIECE: piece size of pass 4's pipeline in bases (0: about a quarter of a batch)
    // Workgroups per CU of the persistent kernels while two streams are in use (0: as many as fit).  k_scan_trusted, k_infer
    // and k_correct_wave take their reads from a global counter, so a grid of any size finishes the batch; a kernel that
    // fills every wave slot keeps the other stream's kernel out until its own last read is done.
    // Measured on the 30x workload (profiles/r04_ab_pass3_grid_caps.json, r04_ab_grid_caps_b.json): pass 3 with four scan and two
    // walk workgroups per CU 1415 ms against 1568 ms uncapped; k_infer capped loses (its insert side needs more room than a cap
    // that it tolerates leaves).  -1 = that setting for batches of equally long reads of up to 192 bases (no offsets array: where it
    // was measured), none otherwise.  It only pays when the two streams really run side by side: a host process whose streams
    // outnumber the runtime's hardware queues (four by default) may find both of the engine's on one queue, and capped kernels
    // in order are slower than uncapped ones -- GPU_MAX_HW_QUEUES; profiles/r04_e2e_3e10_hw_queues.txt.
    int scan_blocks = -1;             // KBBQ_SCAN_BLOCKS / kbbq_engine_tune("scan_blocks", n)
    int walk_blocks = -1;             // KBBQ_WALK_BLOCKS / "walk_blocks"
    int infer_blocks = 0;             // KBBQ_INFER_BLOCKS / "infer_blocks"
};

struct kbbq_engine {
    kbbq_params p;
    Options opt;
    KParams K;
    hipStream_t stream = nullptr;
    // Pass 3 alternates device-resident batches between two streams: the latency-bound correction walk and the
    // ALU-bound tally of batch i run beside the memory-bound Bloom scan of batch i+1.  Every scratch array of
    // the pass and its counters exist twice; an event per side says when a side's buffers are free again.
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_main = nullptr, ev_side[2] = {nullptr, nullptr};
    // pass 1: the draw of batch i+1 (ALU only, side stream) runs beside the insert of batch i (main stream); two mask buffers
    hipEvent_t ev_draw = nullptr, ev_ins[2] = {nullptr, nullptr};
    bool ins_pending[2] = {false, false};
    int draw_turn = 0;
    bool side_busy[2] = {false, false};     // a batch of pass 3 used that side since the totals were last read
    bool pass3_shared = false;              // the batch being submitted shares the chip with its neighbours' kernels (grid caps)
    unsigned long long *d_totals = nullptr; // pass 3: [0] reads sent to the correction kernels, [1] Bloom queries there (k_add_counters)
    int side_turn = 0;
    uint32_t *d_qpresent = nullptr;     // quality values seen by pass 2 (256 bits), read at kbbq_trusted_finish
    uint32_t qpresent[8] = {};          // ... as read then; pass 3 on its own (--fixed mode) adds each batch's values (k_qpresence)
    bool qpresent_known = false;        // pass 2 has finished since the last reset: its set covers every batch of pass 3
    uint8_t *d_dq_qslot = nullptr;      // apply kernel: quality -> LDS table slot (upload_dq)
    int dq_slots = 0;
    uint32_t *d_rg_present[2] = {nullptr, nullptr};     // which read groups a batch contains (run_tally), per stream
    hipStream_t cur = nullptr;              // stream and counter pair the pass-3 launch helpers use
    unsigned long long *cur_cnt = nullptr;
    FilterHost filt[2];
    Xoshiro256 seed_state;
    uint64_t draw_threshold = 0;
    bool draw_always = false;
    bool thresholds_set = false;
    std::vector<int32_t> thresholds;
    // histograms and delta-Q tables
    unsigned long long *d_hist = nullptr;   // cycle then dinuc, contiguous
    uint64_t hist_cycle_words = 0, hist_dinuc_words = 0;
    DqTables dq;
    bool dq_set = false;
    int16_t *d_dq_base = nullptr;
    int8_t *d_dq_cycle = nullptr;
    int8_t *d_dq_dinuc = nullptr;
    // scratch
    void *scratch[24] = {};
    size_t scratch_bytes[24] = {};
    unsigned long long *d_counters = nullptr;   // [0] work-list length, [1] correction queries, [2] scan total
    unsigned int *d_tickets = nullptr;          // chunk counters of the kernels that hand their reads out dynamically (ReadChunks): [0] k_infer, [1 + side] k_scan_trusted, [3 + side] k_correct_wave
    // Host batches: a ring of device staging slots owned by the engine and a copy stream.  A host batch is copied
    // into the next slot with hipMemcpyAsync on the copy stream (DMA straight from the caller's memory when that is
    // page-locked), the pass's kernels wait for the copy by event, and the entry point returns as soon as the COPY
    // has finished: the caller's memory is its own again while the kernels still run, so the copy of batch i+1
    // overlaps the kernels of batch i.  A slot is reused once the kernels that read it have finished (events).
    struct StageSlot {
        char *dev = nullptr;
        size_t bytes = 0;
        hipEvent_t h2d = nullptr, done[2] = {nullptr, nullptr};     // done[0]: main stream, done[1]: side stream
        hipEvent_t d2h = nullptr;        // a result of this batch has landed in the caller's host memory (pass 4, asynchronous form)
        bool busy[2] = {false, false};
        bool d2h_pending = false;
        uint32_t gen = 0;                // how often the slot has been handed out: part of a batch's ticket
    } slot[3];
    int slot_turn = 0, cur_slot = -1;
    bool cur_slot_side = false;      // the batch in cur_slot is also read on the side stream (pass 3)
    bool cur_h2d_recorded = false;   // the slot's h2d event stands for THIS batch's copies (HostBatchDone)
    // the asynchronous form of the batch entry points (kbbq_*_batch_submit + kbbq_batch_wait): the entry point does not
    // wait for its batch's copy; the ticket names the staging slot and its generation
    bool async_call = false;
    uint64_t last_ticket = 0;
    hipStream_t copy = nullptr;
    uint64_t stats[4] = {0, 0, 0, 0};
    // profiling
    std::vector<ProfileSlot> prof;
    std::vector<PendingEvent> pending;
    uint32_t *d_qcum = nullptr, *d_errthr = nullptr;
    uint32_t qcum_len = 0;
    // slice-bucketed inserts (bucket.h): record buffers shared by the two filters (a pass inserts into one)
    struct Buckets {
        int mode[2] = {-1, -1};          // -1 undecided, 0 direct inserts (k_insert_marked), 1 bucketed
        bool allocated = false;
        uint64_t capacity = 0;           // records gathered between two flushes
        void *l1 = nullptr, *l2 = nullptr;
        uint32_t *l1_cnt = nullptr, *l2_cnt = nullptr, *tickets = nullptr;
        unsigned long long *direct = nullptr;
        bool pending[2] = {false, false};
        double pending_est[2] = {0, 0};  // estimated records since the last flush
        double frac_trusted = 0.75;      // trusted inserts per base, learnt at every flush
        double frac_used = 0.75;         // ... as used for the estimate of the batch being reserved
        uint64_t bases_since[2] = {0, 0}, inserted_at_flush[2] = {0, 0};
        uint64_t flushes[2] = {0, 0};
        // the stream filter w's emits and flushes are queued on: the engine's, or -- pass 2 -- the side stream, where the
        // whole insert side of the pass (emit, split, apply) runs beside the next batches' k_infer
        hipStream_t stream[2] = {nullptr, nullptr};
        hipEvent_t ev_flush = nullptr;       // a flush on the side stream has finished (pass boundaries wait for it)
        // the learnt estimate is read back without stopping either stream: copy into page-locked memory + event
        unsigned long long *h_inserted = nullptr;
        hipEvent_t ev_est = nullptr;
        bool est_pending = false, est_known = false;
        uint64_t est_bases = 0, est_prev = 0, cur_bases = 0;
    } bk;
    // pass 2: insert decisions of k_infer for batches that bring no hint array: two buffers, so that the side stream
    // can still read batch i's while k_infer writes batch i+1's
    hipEvent_t ev_infer = nullptr, ev_take[2] = {nullptr, nullptr};
    bool take_pending[2] = {false, false};
    int take_turn = 0;
    // hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: what this engine has raised on ITS device
    size_t attr_lds_tally = 0, attr_lds_recal = 0;
    std::map<const void *, size_t> attr_lds_correct;
};

namespace {

int ensure_scratch(kbbq_engine *e, int idx, size_t bytes) {
    if (e->scratch_bytes[idx] >= bytes) return KBBQ_OK;
    if (e->scratch[idx]) {
        HIP_TRY(hipStreamSynchronize(e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream2));
        HIP_TRY(hipFree(e->scratch[idx]));
        e->scratch[idx] = nullptr;
        e->scratch_bytes[idx] = 0;
    }
    const size_t want = bytes + bytes / 8 + 256;
    HIP_TRY(hipMalloc(&e->scratch[idx], want));
    e->scratch_bytes[idx] = want;
    return KBBQ_OK;
}

// quality -> slot map of the tally from the presence bits (256: a quality is any uint8_t).  At most `max_slots`
// values get a slot (what the LDS tables hold, and never more than 255: 255 means "none"); the rest -- more distinct
// qualities than any real instrument writes -- are counted through global atomics by the kernels.
static void plan_tally_slots(TallyPlan &P, const uint32_t mask[8], int max_slots, int n_rg) {
    memset(P.qslot, 255, sizeof P.qslot);
    memset(P.qof, 0, sizeof P.qof);
    P.n_slots = 0;
    max_slots = std::max(1, std::min(max_slots, 255));
    int top = -1, distinct = 0;
    for (int q = 0; q < KBBQ_NQ; ++q)
        if ((mask[q >> 5] >> (q & 31)) & 1) { top = q; ++distinct; }
    // With several read groups every unused slot is LDS that another read group's tables could have had (binned
    // qualities {2,11,25,37}: 38 identity slots against 4), and fewer read groups per launch means more passes over
    // the batch: the identity layout only when it wastes less than half of its slots, or there is one read group.
    if (top >= 0 && top + 1 <= max_slots && (n_rg <= 1 || 2 * distinct >= top + 1)) {
        // few, small values (the usual FASTQ range): slot = quality, no lookup in the kernels
        for (int q = 0; q <= top; ++q) { P.qof[q] = (uint8_t)q; P.qslot[q] = (uint8_t)q; }
        P.n_slots = top + 1;
        P.identity = 1;
    } else {
        for (int q = 0; q < KBBQ_NQ && P.n_slots < max_slots; ++q)
            if ((mask[q >> 5] >> (q & 31)) & 1) { P.qof[P.n_slots] = (uint8_t)q; P.qslot[q] = (uint8_t)P.n_slots++; }
        P.identity = 0;
    }
    if (P.n_slots == 0) { P.n_slots = 1; P.qof[0] = 0; P.qslot[0] = 0; P.identity = 1; }      // (an empty set: one unused slot)
    P.rg_base = 0; P.n_rgs = 1; P.cbase = 0; P.direct_cycles = 0;
}

struct Timed {
    kbbq_engine *e;
    int slot = -1;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t on = nullptr;
    Timed(kbbq_engine *e_, const char *name, hipStream_t stream = nullptr) : e(e_), on(stream ? stream : e_->stream) {
        if (!(e->p.flags & KBBQ_F_PROFILE)) return;
        for (size_t i = 0; i < e->prof.size(); ++i)
            if (e->prof[i].name == name) slot = (int)i;
        if (slot < 0) {
            ProfileSlot s;
            s.name = name;
            e->prof.push_back(s);
            slot = (int)e->prof.size() - 1;
        }
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipEventRecord(a, on);
    }
    ~Timed() {
        if (slot < 0) return;
        hipEventRecord(b, on);
        PendingEvent pe = {slot, a, b};
        e->pending.push_back(pe);
    }
};

void drain_profile(kbbq_engine *e) {
    for (size_t i = 0; i < e->pending.size(); ++i) {
        PendingEvent &pe = e->pending[i];
        float ms = 0;
        if (hipEventSynchronize(pe.b) == hipSuccess && hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess) {
            e->prof[pe.slot].ms += ms;
            e->prof[pe.slot].launches += 1;
        }
        hipEventDestroy(pe.a);
        hipEventDestroy(pe.b);
    }
    e->pending.clear();
}

// Pass 3 counts on the device: every batch's work-list length and Bloom-query count are added to running totals by a
// one-lane kernel behind its walk (k_add_counters), so submitting a batch never waits for the batch that used its side's
// scratch before (stream waits order that); the host reads the totals when it synchronises anyway.
__global__ void k_add_counters(const unsigned long long *side, const unsigned long long *offcase, unsigned long long *totals) {
    totals[0] += side[0] + offcase[0];      // reads sent to the correction kernels
    totals[1] += side[1];                   // Bloom queries issued there
}

int collect_totals(kbbq_engine *e) {      // (both streams are idle)
    unsigned long long t[2] = {0, 0};
    HIP_TRY(hipMemcpy(t, e->d_totals, 16, hipMemcpyDeviceToHost));
    e->stats[0] = t[0];
    e->stats[1] = t[1];
    e->side_busy[0] = e->side_busy[1] = false;
    return KBBQ_OK;
}

int sync_engine(kbbq_engine *e) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream2));
    if (e->side_busy[0] || e->side_busy[1]) {
        int rc = collect_totals(e);
        if (rc) return rc;
    }
    drain_profile(e);
    return KBBQ_OK;
}

// A host batch is the caller's again when the entry point returns, on EVERY path (error returns included): the
// guard waits for the batch's copy into its staging slot -- not for the kernels -- and notes when the slot is free.
struct HostBatchDone {
    kbbq_engine *e;
    HostBatchDone(kbbq_engine *e_, const kbbq_reads *) : e(e_) { e->cur_slot = -1; e->cur_slot_side = false; e->cur_h2d_recorded = false; e->last_ticket = 0; }
    ~HostBatchDone() {
        if (e->cur_slot < 0) return;
        kbbq_engine::StageSlot &s = e->slot[e->cur_slot];
        if (hipEventRecord(s.done[0], e->stream) == hipSuccess) s.busy[0] = true;
        if (e->cur_slot_side && hipEventRecord(s.done[1], e->stream2) == hipSuccess) s.busy[1] = true;
        if (e->async_call && e->cur_h2d_recorded) {
            // asynchronous form: the caller waits with kbbq_batch_wait when it wants the batch's memory back; the next
            // batch's copy is queued straight behind this one's
            e->last_ticket = ((uint64_t)s.gen << 8) | (uint64_t)(e->cur_slot + 1);
        } else if (e->cur_h2d_recorded) {
            // the event of THIS batch's copies
            hipEventSynchronize(s.h2d);
        } else {
            // an error came before the event could be recorded (it still carries the batch before): everything
            // queued on the copy stream
            hipStreamSynchronize(e->copy);
        }
        e->cur_slot = -1;
    }
};

// (KbbqDeviceGuard, abi_internal.h: the engine's device for the call, the caller's current device restored after it)
typedef KbbqDeviceGuard DeviceGuard;
#define ENGINE_DEVICE(e)                                                                              \
    if (!(e)) return fail(KBBQ_EINVAL, "null engine");                                                \
    DeviceGuard engine_device_guard((e)->p.device);                                                   \
    if (engine_device_guard.err != hipSuccess)                                                        \
        return fail(KBBQ_EIO, "hipSetDevice(%d): %s", (e)->p.device, hipGetErrorString(engine_device_guard.err))

// one non-blocking copy stream per device for uploads made before any engine exists (kept for the life of the process)
hipStream_t shared_copy_stream(int device) {
    static std::mutex m;
    static std::map<int, hipStream_t> streams;
    std::lock_guard<std::mutex> lock(m);
    std::map<int, hipStream_t>::iterator it = streams.find(device);
    if (it != streams.end()) return it->second;
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return nullptr;
    streams[device] = st;
    return st;
}

// next staging slot, at least `need` bytes, not read by any kernel any more
int acquire_slot(kbbq_engine *e, size_t need, kbbq_engine::StageSlot **out) {
    kbbq_engine::StageSlot &s = e->slot[e->slot_turn];
    for (int t = 0; t < 2; ++t)
        if (s.busy[t]) { HIP_TRY(hipEventSynchronize(s.done[t])); s.busy[t] = false; }
    if (s.d2h_pending) { HIP_TRY(hipEventSynchronize(s.d2h)); s.d2h_pending = false; }
    s.gen += 1;
    if (s.bytes < need) {
        if (s.dev) HIP_TRY(hipFree(s.dev));
        s.dev = nullptr; s.bytes = 0;
        const size_t want = need + need / 8 + 4096;
        HIP_TRY(hipMalloc(&s.dev, want));
        s.bytes = want;
    }
    e->cur_slot = e->slot_turn;
    e->slot_turn = (e->slot_turn + 1) % 3;
    *out = &s;
    return KBBQ_OK;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// (pass 4 on a host batch: the caller copies bases, N mask and qualities itself, piece by piece -- recalibrate_impl)
struct DeferredCopy {
    char *d_bases = nullptr, *d_nmask = nullptr, *d_qual = nullptr;
    kbbq_engine::StageSlot *slot = nullptr;
};
// a host batch whose result goes back to host memory without the call waiting for it: the device copy of the result
// lives in the batch's staging slot (two such batches may be in flight)
struct SlotOutput {
    size_t bytes = 0;            // in: bytes wanted
    char *dev = nullptr;         // out
    kbbq_engine::StageSlot *slot = nullptr;
};

// device view of a batch; host batches are copied (freed at the next sync)
// need_qual = false (pass 1: k-mers only): a host batch's qualities are not copied
// defer (host batches): the three big arrays are NOT copied and the engine's stream does not wait: the caller does both
int device_view(kbbq_engine *e, const kbbq_reads *in, ReadsDev *out, int *max_len, bool need_qual = true, DeferredCopy *defer = nullptr,
                SlotOutput *slot_out = nullptr) {
    if (!in) return fail(KBBQ_EINVAL, "null batch");
    if (in->n_reads == 0) return fail(KBBQ_EINVAL, "empty batch");
    if (!in->offsets && in->read_len == 0) return fail(KBBQ_EINVAL, "batch has neither offsets nor read_len");
    if (!in->offsets && in->n_bases != in->n_reads * (uint64_t)in->read_len)
        return fail(KBBQ_EINVAL, "uniform batch: n_bases != n_reads * read_len");
    if (in->n_reads > 0xFFFFFFFFULL) return fail(KBBQ_ERANGE, "more than 2^32-1 reads in one batch");
    ReadsDev R;
    R.n_reads = in->n_reads;
    R.n_bases = in->n_bases;
    R.read_len = in->read_len;
    R.hint_sampled = nullptr;
    R.hint_trusted = nullptr;
    R.offcase = nullptr;
    if (in->on_device) {
        R.bases = in->bases; R.nmask = in->nmask; R.qual = in->qual;
        R.offsets = in->offsets; R.flags = in->flags; R.rg = in->rg;
        R.hint_sampled = (uint32_t *)in->hint_sampled;
        R.hint_trusted = (uint32_t *)in->hint_trusted;
        R.offcase = in->offcase;
    }
    uint64_t host_longest = 0;
    if (!in->on_device && in->offsets) {
        if (in->offsets[0] != 0 || in->offsets[in->n_reads] != in->n_bases) return fail(KBBQ_EINVAL, "offsets do not span n_bases");
        for (uint64_t r = 0; r < in->n_reads; ++r) {
            if (in->offsets[r + 1] < in->offsets[r]) return fail(KBBQ_EINVAL, "offsets are not monotone");
            const uint64_t l = in->offsets[r + 1] - in->offsets[r];
            if (l > (uint64_t)e->p.max_read_len)
                return fail(KBBQ_ERANGE, "read %llu is longer than params.max_read_len %d", (unsigned long long)r, e->p.max_read_len);
            host_longest = std::max(host_longest, l);
        }
    }
    if (!in->on_device) {
        if (!in->bases || !in->nmask || !in->qual) return fail(KBBQ_EINVAL, "batch without bases, nmask or qual");
        // one staging slot holds the batch: bases, N mask, qualities (+ offsets, flags, read groups), each padded as
        // the kernels expect (one spare zero word behind the packed arrays, 16 zero bytes behind the qualities)
        const size_t nb_b = (in->n_bases / 32 + 1) * 8, nb_m = (in->n_bases / 64 + 1) * 8, nb_q = in->n_bases;
        const size_t nb_o = in->offsets ? (in->n_reads + 1) * 8 : 0, nb_f = in->flags ? in->n_reads : 0, nb_g = in->rg ? in->n_reads * 2 : 0;
        const size_t nb_c = in->offcase ? nb_m : 0;
        const size_t o_b = 0, o_m = align256(o_b + nb_b + 8), o_q = align256(o_m + nb_m + 8), o_o = align256(o_q + nb_q + 16),
                     o_f = align256(o_o + nb_o), o_g = align256(o_f + nb_f), o_c = align256(o_g + nb_g), o_r = align256(o_c + nb_c + 8),
                     total = align256(o_r + (slot_out ? slot_out->bytes : 0));
        kbbq_engine::StageSlot *sl;
        int rc = acquire_slot(e, total, &sl);
        if (rc) return rc;
        char *d = sl->dev;
        if (slot_out) { slot_out->dev = d + o_r; slot_out->slot = sl; }
        HIP_TRY(hipMemsetAsync(d + o_b + nb_b, 0, 8, e->copy));
        HIP_TRY(hipMemsetAsync(d + o_m + nb_m, 0, 8, e->copy));
        HIP_TRY(hipMemsetAsync(d + o_q + nb_q, 0, 16, e->copy));
        if (defer) {
            defer->d_bases = d + o_b; defer->d_nmask = d + o_m; defer->d_qual = d + o_q; defer->slot = sl;
        } else {
            HIP_TRY(hipMemcpyAsync(d + o_b, in->bases, nb_b, hipMemcpyHostToDevice, e->copy));
            HIP_TRY(hipMemcpyAsync(d + o_m, in->nmask, nb_m, hipMemcpyHostToDevice, e->copy));
            if (need_qual) HIP_TRY(hipMemcpyAsync(d + o_q, in->qual, nb_q, hipMemcpyHostToDevice, e->copy));
        }
        if (nb_o) HIP_TRY(hipMemcpyAsync(d + o_o, in->offsets, nb_o, hipMemcpyHostToDevice, e->copy));
        if (nb_f) HIP_TRY(hipMemcpyAsync(d + o_f, in->flags, nb_f, hipMemcpyHostToDevice, e->copy));
        if (nb_g) HIP_TRY(hipMemcpyAsync(d + o_g, in->rg, nb_g, hipMemcpyHostToDevice, e->copy));
        if (nb_c) {
            HIP_TRY(hipMemsetAsync(d + o_c + nb_c, 0, 8, e->copy));
            HIP_TRY(hipMemcpyAsync(d + o_c, in->offcase, nb_c, hipMemcpyHostToDevice, e->copy));
        }
        HIP_TRY(hipEventRecord(sl->h2d, e->copy));
        e->cur_h2d_recorded = true;
        HIP_TRY(hipStreamWaitEvent(e->stream, sl->h2d, 0));      // (the side stream only ever follows the main one; with `defer`: the small arrays)
        R.bases = (const uint64_t *)(d + o_b); R.nmask = (const uint64_t *)(d + o_m); R.qual = (const uint8_t *)(d + o_q);
        R.offsets = nb_o ? (const uint64_t *)(d + o_o) : nullptr;
        R.flags = nb_f ? (const uint8_t *)(d + o_f) : nullptr;
        R.rg = nb_g ? (const uint16_t *)(d + o_g) : nullptr;
        R.offcase = nb_c ? (const uint64_t *)(d + o_c) : nullptr;
    }
    *out = R;
    // longest read (selects the kernel variants): uniform => read_len; ragged host batch => measured;
    // ragged device batch => the engine's declared maximum
    *max_len = !in->offsets ? (int)in->read_len : in->on_device ? e->p.max_read_len : std::max<int>(1, (int)host_longest);
    if (*max_len > KBBQ_MAX_READ_LEN) return fail(KBBQ_ERANGE, "read length %d > %d", *max_len, KBBQ_MAX_READ_LEN);
    if (*max_len > e->p.max_read_len) return fail(KBBQ_ERANGE, "read length %d > params.max_read_len %d", *max_len, e->p.max_read_len);
    return KBBQ_OK;
}

// grid of a kernel that handles one read per wavefront, four wavefronts per workgroup; per_cu: Options::*_blocks
inline int wave_grid(uint64_t n_reads, int per_cu = 0) {
    const uint64_t blocks = (n_reads + 3) / 4;
    return (int)std::min<uint64_t>(blocks, 256 * (uint64_t)(per_cu > 0 ? per_cu : 16));
}
// ... the cap only while the other stream has work of the same pass (Options)
inline int shared_cap(const kbbq_engine *e, int per_cu, int auto_value = 0, int max_len = 0, bool uniform = true) {
    if (e->opt.no_overlap) return 0;
    return per_cu >= 0 ? per_cu : max_len <= 192 && uniform ? auto_value : 0;
}

// exclusive k-mer-position prefix for ragged batches (scratch slot 1); null for uniform ones
int kmer_prefix(kbbq_engine *e, const ReadsDev &R, const uint64_t **kofs, uint64_t *total) {
    if (!R.offsets) {
        *kofs = nullptr;
        const uint64_t nk = R.read_len >= (uint32_t)e->p.k ? R.read_len - e->p.k + 1 : 0;
        *total = nk * R.n_reads;
        return KBBQ_OK;
    }
    const uint64_t n_tiles = (R.n_reads + SCAN_TILE - 1) / SCAN_TILE;
    int rc = ensure_scratch(e, 1, (R.n_reads + 1 + n_tiles) * 8);
    if (rc) return rc;
    uint64_t *d = (uint64_t *)e->scratch[1], *tiles = d + R.n_reads + 1;
    hipLaunchKernelGGL(k_kmer_counts, dim3((unsigned)((R.n_reads + 255) / 256)), dim3(256), 0, e->stream, R, e->p.k, d);
    hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)n_tiles), dim3(256), 0, e->stream, d, R.n_reads, tiles);
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(1024), 0, e->stream, tiles, n_tiles, (uint64_t *)&e->d_counters[2]);
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)((R.n_reads + 255) / 256)), dim3(256), 0, e->stream, d, R.n_reads, (const uint64_t *)tiles);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(total, &e->d_counters[2], 8, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    *kofs = d;
    return KBBQ_OK;
}

int upload_dq(kbbq_engine *e) {
    const DqTables &d = e->dq;
    const size_t nb = d.n_rg * kNQ, nc = d.n_rg * kNQ * 2 * d.n_cycle, nd = d.n_rg * kNQ * 16;
    std::vector<int16_t> base(nb);
    std::vector<int8_t> cyc(nc), di(nd);
    for (uint64_t r = 0; r < d.n_rg; ++r)
        for (int q = 0; q < kNQ; ++q) base[r * kNQ + q] = (int16_t)(d.meanq[r] + d.rgdq[r] + d.qdq[r * kNQ + q]);
    for (size_t i = 0; i < nc; ++i) cyc[i] = (int8_t)d.cycledq[i];
    for (size_t i = 0; i < nd; ++i) di[i] = (int8_t)d.dinucdq[i];
    HIP_TRY(hipMemcpyAsync(e->d_dq_base, base.data(), nb * 2, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->d_dq_cycle, cyc.data(), nc, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->d_dq_dinuc, di.data(), nd, hipMemcpyHostToDevice, e->stream));
    // quality values that carry a cycle or dinucleotide delta in some read group get a slot in the apply
    // kernel's LDS tables; the others only need their base value
    uint8_t slot[KBBQ_NQ];
    memset(slot, 255, sizeof slot);
    int n_slots = 0;
    for (int q = 0; q < kNQ; ++q) {
        bool any = false;
        for (uint64_t r = 0; r < d.n_rg && !any; ++r) {
            const size_t c0 = (r * kNQ + q) * 2 * d.n_cycle, d0 = (r * kNQ + q) * 16;
            for (size_t i = 0; i < 2 * d.n_cycle && !any; ++i) any = cyc[c0 + i] != 0;
            for (size_t i = 0; i < 16 && !any; ++i) any = di[d0 + i] != 0;
        }
        if (any && n_slots < 255) slot[q] = (uint8_t)n_slots;      // (255 means "none"; dq_slots > 255 sends the apply kernel to its global tables)
        if (any) ++n_slots;
    }
    e->dq_slots = n_slots;
    HIP_TRY(hipMemcpyAsync(e->d_dq_qslot, slot, KBBQ_NQ, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->dq_set = true;
    return KBBQ_OK;
}

// ---- slice-bucketed inserts, host side (bucket.h) ----------------------------------------------------------
uint64_t env_u64(const char *name, uint64_t dflt) {
    const char *v = getenv(name);
    return v && *v ? strtoull(v, nullptr, 10) : dflt;
}

// level-1 regions belong to the emitting workgroup (bucket.h; per-XCD shared regions were measured in round 2 and lost)
constexpr bool bucket_private() { return true; }

void bucket_shape(uint64_t n_blocks, uint64_t capacity, uint32_t *n_sub, int *nb1, uint32_t *cap1, uint32_t *cap2) {
    *n_sub = (uint32_t)((n_blocks + SUB_BLOCKS - 1) >> SUB_BITS);
    *nb1 = (int)((*n_sub + NB2 - 1) >> NB2_BITS);
    // Regions are sized for evenly spread hashes: a full level-1 bucket receives 2^21 / n_blocks of the records (one
    // n_src-th of that per source region), a full subslice 2^12 / n_blocks; plus 15 % (the sources' shares of the work
    // differ a little) and 25 % (a few thousand records per subslice).  What does not fit is inserted directly.
    const double f1 = std::min(1.0, (double)(1ULL << L1_SHIFT) / (double)n_blocks), f2 = std::min(1.0, (double)SUB_BLOCKS / (double)n_blocks);
    const int n_src = bucket_private() ? EMIT_GRID : N_XCD;
    *cap1 = (uint32_t)std::min(4.0e9, (double)capacity * 1.15 * f1 / n_src) + 64;
    *cap2 = (uint32_t)std::min(4.0e9, (double)capacity * 1.25 * f2) + 32;
}

BucketDev bucket_dev(kbbq_engine *e, int w) {
    BucketDev B;
    bucket_shape(e->filt[w].spec.n_blocks, e->bk.capacity, &B.n_sub, &B.nb1, &B.cap1, &B.cap2);
    B.l1 = (unsigned long long *)e->bk.l1;
    B.l2 = (uint32_t *)e->bk.l2;
    B.l1_cnt = e->bk.l1_cnt;
    B.l2_cnt = e->bk.l2_cnt;
    B.tickets = e->bk.tickets;
    B.direct = e->bk.direct;
    B.n_src = bucket_private() ? EMIT_GRID : N_XCD;
    B.cnt_stride = bucket_private() ? 1 : CNT_STRIDE;
    return B;
}

const size_t kL1CntBytes = (size_t)EMIT_GRID * MAX_NB1 * 4, kL2CntBytes = (size_t)MAX_NB1 * NB2 * 4, kTicketBytes = (size_t)N_XCD * CNT_STRIDE * 4;

// Decide (once per filter) whether its inserts are bucketed, and allocate the record buffers at the first use:
// by then the caller's resident batches are in HBM, so "a share of what is free" is a safe size.
bool bucket_on(kbbq_engine *e, int w) {
    kbbq_engine::Buckets &b = e->bk;
    if (b.mode[w] >= 0) return b.mode[w] == 1;
    const uint64_t n_blocks = e->filt[w].spec.n_blocks;
    bool want = e->opt.bucket >= 0 ? e->opt.bucket != 0 : e->filt[w].table_bytes() >= (256u << 20);
    if (n_blocks > ((uint64_t)MAX_NB1 << L1_SHIFT)) want = false;
    if (want && !b.allocated) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        const uint64_t big = std::max(e->filt[0].spec.n_blocks, e->filt[1].spec.n_blocks);
        uint64_t cap = (uint64_t)((double)free_b * 0.45 / 14.5);
        cap = std::min<uint64_t>(cap, 4000000000ULL);
        cap = std::min<uint64_t>(cap, std::max<uint64_t>(1u << 20, 32 * big));
        if (e->opt.bucket_records) cap = e->opt.bucket_records;
        cap = std::max<uint64_t>(cap, 4096);
        size_t l1_bytes = 0, l2_bytes = 0;
        for (int f = 0; f < 2; ++f) {
            uint32_t n_sub, cap1, cap2; int nb1;
            bucket_shape(e->filt[f].spec.n_blocks, cap, &n_sub, &nb1, &cap1, &cap2);
            l1_bytes = std::max(l1_bytes, (size_t)(bucket_private() ? EMIT_GRID : N_XCD) * nb1 * cap1 * 8);
            l2_bytes = std::max(l2_bytes, (size_t)n_sub * cap2 * 4);
        }
        hipError_t he = hipMalloc(&b.l1, l1_bytes);
        if (he == hipSuccess) he = hipMalloc(&b.l2, l2_bytes);
        if (he == hipSuccess) he = hipMalloc(&b.l1_cnt, kL1CntBytes);
        if (he == hipSuccess) he = hipMalloc(&b.l2_cnt, kL2CntBytes);
        if (he == hipSuccess) he = hipMalloc(&b.tickets, kTicketBytes);
        if (he == hipSuccess) he = hipMalloc(&b.direct, 8);
        if (he == hipSuccess) he = hipMemsetAsync(b.l1_cnt, 0, kL1CntBytes, e->stream);
        if (he == hipSuccess) he = hipMemsetAsync(b.l2_cnt, 0, kL2CntBytes, e->stream);
        if (he == hipSuccess) he = hipMemsetAsync(b.direct, 0, 8, e->stream);
        if (he != hipSuccess) {      // no room: the direct path needs none
            (void)hipGetLastError();
            hipFree(b.l1); hipFree(b.l2); hipFree(b.l1_cnt); hipFree(b.l2_cnt); hipFree(b.tickets); hipFree(b.direct);
            b.l1 = b.l2 = nullptr; b.l1_cnt = b.l2_cnt = b.tickets = nullptr; b.direct = nullptr;
            want = false;
            b.mode[0] = b.mode[1] = 0;
        } else {
            b.allocated = true;
            b.capacity = cap;
        }
    }
    if (b.mode[w] < 0) b.mode[w] = want ? 1 : 0;
    return b.mode[w] == 1;
}

// Partition the gathered records of filter w by subslice and OR them into the filter (k_split, k_apply).
hipStream_t bucket_stream(kbbq_engine *e, int w) { return e->bk.stream[w] ? e->bk.stream[w] : e->stream; }

// the learnt trusted-inserts-per-base figure, once its copy has landed
void bucket_poll_estimate(kbbq_engine *e) {
    kbbq_engine::Buckets &b = e->bk;
    if (!b.est_pending || hipEventQuery(b.ev_est) != hipSuccess) return;
    const unsigned long long now = *b.h_inserted;
    if (b.est_bases > 0 && now >= b.est_prev)
        b.frac_trusted = std::min(1.0, 1.03 * (double)(now - b.est_prev) / (double)b.est_bases + 0.005);
    b.inserted_at_flush[1] = now;
    b.est_pending = false;
    if (e->opt.debug_bucket) fprintf(stderr, "[bucket] inserted %llu (before %llu) over %llu bases -> %.4f trusted inserts per base\n", now,
                       (unsigned long long)b.est_prev, (unsigned long long)b.est_bases, b.frac_trusted);
}

// Partition the gathered records of filter w by subslice and OR them into the filter (k_split, k_apply), on the stream
// the filter's emits run on.  barrier: the engine's main stream waits for it (pass boundaries: the next pass reads the
// filter there); a flush in the middle of pass 2 leaves the main stream alone.
int bucket_flush(kbbq_engine *e, int w, bool barrier = true) {
    kbbq_engine::Buckets &b = e->bk;
    if (!b.pending[w]) return KBBQ_OK;
    const BucketDev B = bucket_dev(e, w);
    const FiltDev F = e->filt[w].dev();
    const hipStream_t emit_st = bucket_stream(e, w);
    hipStream_t st = emit_st;
    if (w == 1 && e->opt.pass2_side == 2 && emit_st != e->stream) {
        // the flush alone on the engine's stream, behind the emits that filled the buffers; the emits after it wait for it
        HIP_TRY(hipEventRecord(b.ev_flush, emit_st));
        HIP_TRY(hipStreamWaitEvent(e->stream, b.ev_flush, 0));
        st = e->stream;
    }
    HIP_TRY(hipMemsetAsync(b.tickets, 0, kTicketBytes, st));
    {
        Timed t(e, w ? "k_split_trusted" : "k_split_sampled", st);
        const uint32_t chunks = (B.cap1 + SPLIT_TILE - 1) / SPLIT_TILE;
        hipLaunchKernelGGL(k_split, dim3(256 * 3), dim3(BK_THREADS), 0, st, F, B, chunks);
        HIP_TRY(hipGetLastError());
    }
    {
        Timed t(e, w ? "k_apply_trusted" : "k_apply_sampled", st);
        hipLaunchKernelGGL(k_apply, dim3(B.n_sub), dim3(APPLY_THREADS), 0, st, F, B);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemsetAsync(b.l1_cnt, 0, kL1CntBytes, st));
    HIP_TRY(hipMemsetAsync(b.l2_cnt, 0, (size_t)B.nb1 * NB2 * 4, st));
    if (st != emit_st) {
        HIP_TRY(hipEventRecord(b.ev_flush, st));
        HIP_TRY(hipStreamWaitEvent(emit_st, b.ev_flush, 0));
    }
    if (st != e->stream && barrier) {
        HIP_TRY(hipEventRecord(b.ev_flush, st));
        HIP_TRY(hipStreamWaitEvent(e->stream, b.ev_flush, 0));
    }
    // how many records that was (k_infer counts every insert it decides, on the main stream): the estimate that times
    // the next flush, fetched without stopping a stream
    if (w == 1 && !barrier) {
        bucket_poll_estimate(e);
        if (!b.est_pending && b.h_inserted) {
            // (k_infer of the batch that triggered this flush has been queued already: its bases count)
            HIP_TRY(hipMemcpyAsync(b.h_inserted, e->filt[1].d_inserted, 8, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipEventRecord(b.ev_est, e->stream));
            b.est_pending = true;
            b.est_bases = b.bases_since[1] + b.cur_bases;
            b.est_prev = b.inserted_at_flush[1];
            if (!b.est_known) {
                // the first flush of a run: the host has queued the whole pass ahead of the GPU, so a poll would never
                // see this copy in time -- wait for it once (the main stream has just been drained up to here; the side
                // stream keeps working); later flushes refine the figure when their copy happens to have landed
                HIP_TRY(hipEventSynchronize(b.ev_est));
                bucket_poll_estimate(e);
                b.est_known = true;
            }
        }
    }
    {
        if (e->opt.debug_bucket) fprintf(stderr, "[bucket] flush %d of filter %d: estimate %.3g records over %llu bases (%.4f per base), barrier %d\n",
                           (int)b.flushes[w], w, b.pending_est[w], (unsigned long long)b.bases_since[w], w ? b.frac_trusted : 0.0, (int)barrier);
    }
    b.bases_since[w] = 0;
    b.pending[w] = false;
    b.pending_est[w] = 0;
    b.flushes[w] += 1;
    return KBBQ_OK;
}

// before filter w is read, sized up, exchanged or reset
int bucket_flush_all(kbbq_engine *e) {
    for (int w = 0; w < 2; ++w) {
        int rc = bucket_flush(e, w);
        if (rc) return rc;
    }
    return KBBQ_OK;
}

// room for `est` more records of filter w?  (the other filter's records share the buffers: they go first)
int bucket_reserve(kbbq_engine *e, int w, double est, uint64_t bases) {
    kbbq_engine::Buckets &b = e->bk;
    int rc = bucket_flush(e, 1 - w);
    if (rc) return rc;
    b.cur_bases = bases;
    if (b.pending[w] && b.pending_est[w] + est > 0.93 * (double)b.capacity) {
        if ((rc = bucket_flush(e, w, false))) return rc;
        if (w == 1) est = (est - 4096.0) / std::max(1e-9, b.frac_used) * b.frac_trusted + 4096.0;      // the figure may just have been learnt
    }
    b.cur_bases = 0;
    b.pending[w] = true;
    b.pending_est[w] += est;
    b.bases_since[w] += bases;
    return KBBQ_OK;
}

}  // namespace

// ============================================================ C ABI

extern "C" {

const char *kbbq_last_error(void) { return g_err; }

int kbbq_engine_create(const kbbq_params *params, kbbq_engine **out) {
    if (!params || !out) return fail(KBBQ_EINVAL, "null argument");
    if (params->k < 1 || params->k > KBBQ_MAX_KMER) return fail(KBBQ_ERANGE, "k must be <= %d and > 0", KBBQ_MAX_KMER);
    if (params->n_rg < 1 || params->n_rg > 65535) return fail(KBBQ_EINVAL, "n_rg must be in 1..65535");
    if (params->max_read_len < 1 || params->max_read_len > KBBQ_MAX_READ_LEN)
        return fail(KBBQ_ERANGE, "max_read_len must be in 1..%d", KBBQ_MAX_READ_LEN);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(KBBQ_ENODEV, "no HIP device visible");
    if (params->device < 0 || params->device >= ndev) return fail(KBBQ_ENODEV, "device %d of %d", params->device, ndev);
    DeviceGuard create_guard(params->device);      // (the caller's current device is restored on return)
    HIP_TRY(create_guard.err);
    kbbq_engine *e = new kbbq_engine;
    e->p = *params;
    {
        Options &o = e->opt;
        auto env_set = [](const char *name) { const char *v = getenv(name); return v && *v; };
        auto env_int = [](const char *name, int dflt) { const char *v = getenv(name); return v && *v ? atoi(v) : dflt; };
        const int fl = params->flags;
        o.no_overlap = (fl & KBBQ_F_NO_OVERLAP) || env_set("KBBQ_NO_OVERLAP");
        o.no_fastpath = (fl & KBBQ_F_NO_FASTPATH) || env_set("KBBQ_NO_FASTPATH");
        o.lane_walk = (fl & KBBQ_F_LANE_WALK) || (getenv("KBBQ_CORRECT") && !strcmp(getenv("KBBQ_CORRECT"), "lane"));
        o.no_pass4_pipeline = (fl & KBBQ_F_NO_PASS4_PIPELINE) || env_set("KBBQ_NO_PASS4_PIPELINE");      // (no_overlap implies it, where it is used)
        o.pass2_side = (fl & KBBQ_F_PASS2_INORDER) ? 0 : env_int("KBBQ_PASS2_SIDE", o.pass2_side);                // (no_overlap switches it off, where it is used)
        o.tally_general = env_set("KBBQ_TALLY_GENERAL");
        o.infer_subset = env_int("KBBQ_INFER_SUBSET", 1) != 0;
        o.debug_bucket = env_set("KBBQ_DEBUG_BUCKET");
        o.bucket = (fl & KBBQ_F_BUCKET_ON) ? 1 : (fl & KBBQ_F_BUCKET_OFF) ? 0 : env_set("KBBQ_BUCKET") ? (env_int("KBBQ_BUCKET", 0) != 0 ? 1 : 0) : -1;
        o.bucket_records = env_u64("KBBQ_BUCKET_RECORDS", 0);
        o.pass4_piece = env_u64("KBBQ_PASS4_PIECE", 0);
        o.scan_blocks = env_int("KBBQ_SCAN_BLOCKS", o.scan_blocks);
        o.walk_blocks = env_int("KBBQ_WALK_BLOCKS", o.walk_blocks);
        o.infer_blocks = env_int("KBBQ_INFER_BLOCKS", o.infer_blocks);
    }
    e->K.k = params->k;
    e->K.shift = 2u * (unsigned)(params->k - 1);
    e->K.mask = params->k < 32 ? ((1ULL << (2 * params->k)) - 1) : ~0ULL;
    e->K.nmask_bits = params->k < 32 ? ((1u << params->k) - 1) : 0xFFFFFFFFu;
    const double fprs[2] = {params->fpr_sampled, params->fpr_trusted};
    for (int w = 0; w < 2; ++w) {
        if (!make_filter_spec(params->approx_kmers, fprs[w], params->bloom_seed, e->filt[w].spec)) {
            delete e;
            // bloom.cc:18-21 throws std::invalid_argument
            return fail(KBBQ_EINVAL, "Error: Invalid bloom filter parameters. Adjust parameters and try again.");
        }
    }
    hipError_t he = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_main, hipEventDisableTiming);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_draw, hipEventDisableTiming);
    for (int t = 0; t < 2 && he == hipSuccess; ++t) he = hipEventCreateWithFlags(&e->ev_ins[t], hipEventDisableTiming);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_side[0], hipEventDisableTiming);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_side[1], hipEventDisableTiming);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&e->copy, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&e->bk.ev_flush, hipEventDisableTiming);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&e->bk.ev_est, hipEventDisableTiming);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_infer, hipEventDisableTiming);
    for (int t = 0; t < 2 && he == hipSuccess; ++t) he = hipEventCreateWithFlags(&e->ev_take[t], hipEventDisableTiming);
    if (he == hipSuccess) he = hipHostMalloc((void **)&e->bk.h_inserted, 64, hipHostMallocDefault);
    for (int i = 0; i < 3 && he == hipSuccess; ++i) {
        he = hipEventCreateWithFlags(&e->slot[i].h2d, hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&e->slot[i].d2h, hipEventDisableTiming);
        for (int t = 0; t < 2 && he == hipSuccess; ++t) he = hipEventCreateWithFlags(&e->slot[i].done[t], hipEventDisableTiming);
    }
    e->cur = e->stream;
    if (he != hipSuccess) { kbbq_engine_destroy(e); return fail(KBBQ_EIO, "hipStreamCreate: %s", hipGetErrorString(he)); }
#define CREATE_TRY(expr)                                                                           \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            int code = _e == hipErrorOutOfMemory ? KBBQ_ENOMEM : KBBQ_EIO;                         \
            fail(code, "%s: %s", #expr, hipGetErrorString(_e));                                    \
            kbbq_engine_destroy(e);                                                                \
            return code;                                                                           \
        }                                                                                          \
    } while (0)
    CREATE_TRY(hipMalloc(&e->d_counters, 64));
    CREATE_TRY(hipMemset(e->d_counters, 0, 64));
    CREATE_TRY(hipMalloc(&e->d_tickets, 64));
    CREATE_TRY(hipMemset(e->d_tickets, 0, 64));
    CREATE_TRY(hipMalloc(&e->d_totals, 32));
    CREATE_TRY(hipMemset(e->d_totals, 0, 32));      // ([2]: the running byte sum of kbbq_digest_add)
    e->cur_cnt = e->d_counters;
    CREATE_TRY(hipMalloc(&e->d_dq_qslot, KBBQ_NQ));
    CREATE_TRY(hipMalloc(&e->d_qpresent, 32));
    for (int i = 0; i < 2; ++i) CREATE_TRY(hipMalloc(&e->d_rg_present[i], (((size_t)params->n_rg + 31) / 32) * 4 + 4));
    for (int w = 0; w < 2; ++w) {
        FilterHost &f = e->filt[w];
        CREATE_TRY(hipMalloc(&f.d_table, f.table_bytes()));
        CREATE_TRY(hipMalloc(&f.d_patterns, kNumPatterns * kEngineBlockBytes));
        CREATE_TRY(hipMalloc(&f.d_inserted, 8));
        std::vector<uint64_t> packed(kNumPatterns * 2);
        for (uint64_t i = 0; i < kNumPatterns; ++i) {
            if (!squeeze_block(&f.spec.patterns[i * 8], &packed[i * 2])) {
                kbbq_engine_destroy(e);
                return fail(KBBQ_ESTATE, "pattern %llu uses a bit outside the 128 the generator can reach", (unsigned long long)i);
            }
        }
        CREATE_TRY(hipMemcpy(f.d_patterns, packed.data(), kNumPatterns * kEngineBlockBytes, hipMemcpyHostToDevice));
    }
    e->hist_cycle_words = (uint64_t)params->n_rg * kNQ * 2 * params->max_read_len * 2;
    e->hist_dinuc_words = (uint64_t)params->n_rg * kNQ * 16 * 2;
    CREATE_TRY(hipMalloc(&e->d_hist, (e->hist_cycle_words + e->hist_dinuc_words) * 8));
    CREATE_TRY(hipMalloc(&e->d_dq_base, (size_t)params->n_rg * kNQ * 2));
    CREATE_TRY(hipMalloc(&e->d_dq_cycle, (size_t)params->n_rg * kNQ * 2 * params->max_read_len));
    CREATE_TRY(hipMalloc(&e->d_dq_dinuc, (size_t)params->n_rg * kNQ * 16));
    CREATE_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_jump), xoshiro_jump_table(), 64 * 4 * 8));
    e->seed_state.seed32(params->seed);   // KmerSubsampler ctor: rng.Seed(seed), htsiter.hh:143
    e->draw_threshold = bernoulli_threshold(params->alpha, &e->draw_always);
    int rc = kbbq_engine_reset(e);
    if (rc) { kbbq_engine_destroy(e); return rc; }
    CREATE_TRY(hipDeviceSynchronize());
    *out = e;
    return KBBQ_OK;
}

void kbbq_engine_destroy(kbbq_engine *e) {
    if (!e) return;
    DeviceGuard destroy_guard(e->p.device);
    if (e->stream) { hipStreamSynchronize(e->stream); }
    if (e->stream2) hipStreamSynchronize(e->stream2);
    if (e->copy) hipStreamSynchronize(e->copy);
    drain_profile(e);
    for (int i = 0; i < 3; ++i) {
        hipFree(e->slot[i].dev);
        if (e->slot[i].h2d) hipEventDestroy(e->slot[i].h2d);
        if (e->slot[i].d2h) hipEventDestroy(e->slot[i].d2h);
        for (int t = 0; t < 2; ++t) if (e->slot[i].done[t]) hipEventDestroy(e->slot[i].done[t]);
    }
    if (e->copy) hipStreamDestroy(e->copy);
    if (e->bk.ev_flush) hipEventDestroy(e->bk.ev_flush);
    if (e->bk.ev_est) hipEventDestroy(e->bk.ev_est);
    if (e->ev_infer) hipEventDestroy(e->ev_infer);
    for (int t = 0; t < 2; ++t) if (e->ev_take[t]) hipEventDestroy(e->ev_take[t]);
    if (e->bk.h_inserted) hipHostFree(e->bk.h_inserted);
    for (int w = 0; w < 2; ++w) {
        hipFree(e->filt[w].d_table);
        hipFree(e->filt[w].d_patterns);
        hipFree(e->filt[w].d_inserted);
    }
    hipFree(e->d_hist);
    hipFree(e->d_dq_base);
    hipFree(e->d_dq_cycle);
    hipFree(e->d_dq_dinuc);
    hipFree(e->d_counters);
    hipFree(e->d_tickets);
    hipFree(e->d_totals);
    hipFree(e->d_dq_qslot);
    hipFree(e->d_qpresent);
    hipFree(e->d_rg_present[0]);
    hipFree(e->d_rg_present[1]);
    hipFree(e->d_qcum);
    hipFree(e->d_errthr);
    hipFree(e->bk.l1); hipFree(e->bk.l2); hipFree(e->bk.l1_cnt); hipFree(e->bk.l2_cnt); hipFree(e->bk.tickets); hipFree(e->bk.direct);
    for (int i = 0; i < 24; ++i) hipFree(e->scratch[i]);
    if (e->stream2) { hipStreamSynchronize(e->stream2); hipStreamDestroy(e->stream2); }
    if (e->ev_main) hipEventDestroy(e->ev_main);
    if (e->ev_draw) hipEventDestroy(e->ev_draw);
    for (int t = 0; t < 2; ++t) if (e->ev_ins[t]) hipEventDestroy(e->ev_ins[t]);
    for (int i = 0; i < 2; ++i) if (e->ev_side[i]) hipEventDestroy(e->ev_side[i]);
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
}

int kbbq_engine_reset(kbbq_engine *e) {
    ENGINE_DEVICE(e);
    if (!e) return fail(KBBQ_EINVAL, "null engine");
    int rc0 = sync_engine(e);     // a tally may still be adding to the histograms on the side stream
    if (rc0) return rc0;
    for (int w = 0; w < 2; ++w) {
        HIP_TRY(hipMemsetAsync(e->filt[w].d_table, 0, e->filt[w].table_bytes(), e->stream));
        HIP_TRY(hipMemsetAsync(e->filt[w].d_inserted, 0, 8, e->stream));
    }
    HIP_TRY(hipMemsetAsync(e->d_hist, 0, (e->hist_cycle_words + e->hist_dinuc_words) * 8, e->stream));
    HIP_TRY(hipMemsetAsync(e->d_counters, 0, 64, e->stream));
    HIP_TRY(hipMemsetAsync(e->d_totals, 0, 16, e->stream));
    HIP_TRY(hipMemsetAsync(e->d_qpresent, 0, 32, e->stream));
    memset(e->qpresent, 0, sizeof e->qpresent);
    e->qpresent_known = false;
    if (e->bk.allocated) {      // records gathered for the old filters are dropped
        HIP_TRY(hipMemsetAsync(e->bk.l1_cnt, 0, kL1CntBytes, e->stream));
        HIP_TRY(hipMemsetAsync(e->bk.l2_cnt, 0, kL2CntBytes, e->stream));
        HIP_TRY(hipMemsetAsync(e->bk.direct, 0, 8, e->stream));
    }
    e->bk.est_pending = false;
    e->bk.est_known = false;
    e->bk.frac_trusted = 0.75;
    e->bk.stream[0] = e->bk.stream[1] = nullptr;
    for (int w = 0; w < 2; ++w) { e->bk.pending[w] = false; e->bk.pending_est[w] = 0; e->bk.bases_since[w] = 0; e->bk.inserted_at_flush[w] = 0; e->bk.flushes[w] = 0; }
    e->thresholds_set = false;
    e->dq_set = false;
    memset(e->stats, 0, sizeof e->stats);
    return sync_engine(e);
}

int kbbq_engine_sync(kbbq_engine *e) {
    ENGINE_DEVICE(e);
    { int frc = bucket_flush_all(e); if (frc) return frc; }      // deferred inserts reach the filters first (bucket.h)
    if (!e) return fail(KBBQ_EINVAL, "null engine");
    return sync_engine(e);
}

void *kbbq_engine_stream(kbbq_engine *e) { return e ? (void *)e->stream : nullptr; }

int kbbq_engine_dims(kbbq_engine *e, uint64_t *n_rg, uint64_t *n_cycle) {
    if (!e) return fail(KBBQ_EINVAL, "null engine");
    if (n_rg) *n_rg = (uint64_t)e->p.n_rg;
    if (n_cycle) *n_cycle = (uint64_t)e->p.max_read_len;
    return KBBQ_OK;
}

int kbbq_engine_tune(kbbq_engine *e, const char *name, uint64_t value) {
    if (!e || !name) return fail(KBBQ_EINVAL, "null argument");
    if (!strcmp(name, "bucket_records")) {
        if (e->bk.allocated) return fail(KBBQ_ESTATE, "the record buffers are allocated already");
        e->opt.bucket_records = value;
    } else if (!strcmp(name, "pass4_piece")) {
        e->opt.pass4_piece = value;
    } else if (!strcmp(name, "scan_blocks") || !strcmp(name, "walk_blocks") || !strcmp(name, "infer_blocks")) {
        // (17 = the default: the measured setting for short reads, none otherwise)
        if (value > 17) return fail(KBBQ_EINVAL, "%s: 0 (as many as fit) to 16 workgroups per CU, 17 = default", name);
        (name[0] == 's' ? e->opt.scan_blocks : name[0] == 'w' ? e->opt.walk_blocks : e->opt.infer_blocks) = value == 17 ? (name[0] == 'i' ? 0 : -1) : (int)value;
    } else if (!strcmp(name, "infer_subset")) {
        e->opt.infer_subset = value != 0;      // (same results either way: an A/B switch between two runs)
    } else if (!strcmp(name, "pass2_side")) {
        ENGINE_DEVICE(e);
        { int frc = bucket_flush_all(e); if (frc) return frc; }
        int rc = sync_engine(e);
        if (rc) return rc;
        if (value > 2) return fail(KBBQ_EINVAL, "pass2_side: 0, 1 or 2");
        e->opt.pass2_side = (int)value;
        e->bk.stream[0] = e->bk.stream[1] = nullptr;
    } else if (!strcmp(name, "no_overlap")) {
        // between two runs: every kernel in order on one stream (exclusive kernel durations for a profile) or back
        ENGINE_DEVICE(e);
        { int frc = bucket_flush_all(e); if (frc) return frc; }
        int rc = sync_engine(e);
        if (rc) return rc;
        e->opt.no_overlap = value != 0;
        e->bk.stream[0] = e->bk.stream[1] = nullptr;
    } else {
        return fail(KBBQ_EINVAL, "unknown knob '%s'", name);
    }
    return KBBQ_OK;
}

int kbbq_filter_info_get(kbbq_engine *e, int which, kbbq_filter_info *out) {
    ENGINE_DEVICE(e);
    { int frc = bucket_flush_all(e); if (frc) return frc; }      // deferred inserts reach the filters first (bucket.h)
    if (!e || !out || which < 0 || which > 1) return fail(KBBQ_EINVAL, "bad argument");
    const FilterSpec &s = e->filt[which].spec;
    memset(out, 0, sizeof *out);
    out->bits = s.bits;
    out->bits_unblocked = s.bits_unblocked;
    out->n_blocks = s.n_blocks;
    out->table_bytes = e->filt[which].table_bytes();
    out->random_seed = s.random_seed;
    out->n_hash = s.n_hash;
    out->n_salt = s.n_salt;
    for (uint32_t i = 0; i < s.n_salt; ++i) out->salt[i] = s.salt[i];
    int rc = sync_engine(e);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(&out->inserted, e->filt[which].d_inserted, 8, hipMemcpyDeviceToHost));
    return KBBQ_OK;
}

void *kbbq_filter_device_table(kbbq_engine *e, int which) {
    if (!e || which < 0 || which > 1) return nullptr;
    DeviceGuard guard(e->p.device);
    if (guard.err != hipSuccess || bucket_flush_all(e) != KBBQ_OK) return nullptr;      // the array is about to be read
    return e->filt[which].d_table;
}
void *kbbq_filter_device_counter(kbbq_engine *e, int which) { return e && which >= 0 && which < 2 ? e->filt[which].d_inserted : nullptr; }

int kbbq_filter_download(kbbq_engine *e, int which, uint64_t *host_words, uint64_t n_words) {
    ENGINE_DEVICE(e);
    { int frc = bucket_flush_all(e); if (frc) return frc; }      // deferred inserts reach the filters first (bucket.h)
    if (!e || !host_words || which < 0 || which > 1) return fail(KBBQ_EINVAL, "bad argument");
    const uint64_t n_blocks = e->filt[which].spec.n_blocks;
    if (n_words != n_blocks * 8) return fail(KBBQ_EINVAL, "filter has %llu words", (unsigned long long)n_blocks * 8);
    int rc = sync_engine(e);
    if (rc) return rc;
    // the device holds 128-bit blocks; the caller gets the reference's 512-bit ones
    const uint64_t chunk = 1 << 20;
    std::vector<uint64_t> packed(chunk * 2);
    for (uint64_t b0 = 0; b0 < n_blocks; b0 += chunk) {
        const uint64_t nb = std::min(chunk, n_blocks - b0);
        HIP_TRY(hipMemcpy(packed.data(), e->filt[which].d_table + b0 * 2, nb * kEngineBlockBytes, hipMemcpyDeviceToHost));
        for (uint64_t b = 0; b < nb; ++b) expand_block(&packed[b * 2], host_words + (b0 + b) * 8);
    }
    return KBBQ_OK;
}

int kbbq_filter_patterns_download(kbbq_engine *e, int which, uint64_t *host_words) {
    ENGINE_DEVICE(e);
    if (!e || !host_words || which < 0 || which > 1) return fail(KBBQ_EINVAL, "bad argument");
    std::vector<uint64_t> packed(kNumPatterns * 2);
    HIP_TRY(hipMemcpy(packed.data(), e->filt[which].d_patterns, kNumPatterns * kEngineBlockBytes, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < kNumPatterns; ++i) expand_block(&packed[i * 2], host_words + i * 8);
    return KBBQ_OK;
}

int kbbq_filter_or_from(kbbq_engine *e, int which, const void *src_device, uint64_t word_offset, uint64_t n_words) {
    ENGINE_DEVICE(e);
    { int frc = bucket_flush_all(e); if (frc) return frc; }      // deferred inserts reach the filters first (bucket.h)
    if (!e || !src_device || which < 0 || which > 1) return fail(KBBQ_EINVAL, "bad argument");
    const uint64_t total = e->filt[which].spec.n_blocks * 2;   // words of the engine's 128-bit blocks
    if (word_offset > total || n_words > total - word_offset || (word_offset & 1)) return fail(KBBQ_EINVAL, "range outside the filter");
    if (!n_words) return KBBQ_OK;
    Timed t(e, "k_or_words");
    hipLaunchKernelGGL(k_or_words, dim3((unsigned)((n_words / 2 + 256) / 256)), dim3(256), 0, e->stream,
                       e->filt[which].d_table + word_offset, (const uint64_t *)src_device, n_words);
    HIP_TRY(hipGetLastError());
    return KBBQ_OK;
}

int kbbq_device_or(kbbq_engine *e, void *dst_device, const void *src_device, uint64_t n_words) {
    ENGINE_DEVICE(e);
    if (!e || !dst_device || !src_device) return fail(KBBQ_EINVAL, "bad argument");
    if (((uintptr_t)dst_device | (uintptr_t)src_device) & 15) return fail(KBBQ_EINVAL, "buffers must be 16-byte aligned");
    if (!n_words) return KBBQ_OK;
    Timed t(e, "k_or_words");
    hipLaunchKernelGGL(k_or_words, dim3((unsigned)((n_words / 2 + 256) / 256)), dim3(256), 0, e->stream,
                       (uint64_t *)dst_device, (const uint64_t *)src_device, n_words);
    HIP_TRY(hipGetLastError());
    return KBBQ_OK;
}

// sum of n bytes, added to *total (one 64-bit atomic per wavefront)
__global__ void __launch_bounds__(256) k_sum_bytes(const uint8_t *d, uint64_t n, unsigned long long *total) {
    unsigned long long mine = 0;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += (uint64_t)gridDim.x * blockDim.x * 16) {
        if (i + 16 <= n && !(((uintptr_t)d + i) & 15)) {
            const uint4 v = *reinterpret_cast<const uint4 *>(d + i);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            for (int j = 0; j < 4; ++j) mine += (w[j] & 0xFF) + ((w[j] >> 8) & 0xFF) + ((w[j] >> 16) & 0xFF) + (w[j] >> 24);
        } else {
            for (uint64_t j = i; j < n && j < i + 16; ++j) mine += d[j];
        }
    }
    for (int o = 32; o; o >>= 1) mine += __shfl_down(mine, o);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(total, mine);
}

int kbbq_digest_add(kbbq_engine *e, const uint8_t *device_bytes, uint64_t n) {
    ENGINE_DEVICE(e);
    if (!e || (!device_bytes && n)) return fail(KBBQ_EINVAL, "bad argument");
    if (!n) return KBBQ_OK;
    hipLaunchKernelGGL(k_sum_bytes, dim3((unsigned)std::min<uint64_t>((n / 16 + 256) / 256, 4096)), dim3(256), 0, e->stream, device_bytes, n, e->d_totals + 2);
    HIP_TRY(hipGetLastError());
    return KBBQ_OK;
}

int kbbq_digest_get(kbbq_engine *e, uint64_t *sum, int32_t reset) {
    ENGINE_DEVICE(e);
    if (!e || !sum) return fail(KBBQ_EINVAL, "bad argument");
    HIP_TRY(hipStreamSynchronize(e->stream));
    unsigned long long v = 0;
    HIP_TRY(hipMemcpy(&v, e->d_totals + 2, 8, hipMemcpyDeviceToHost));
    *sum = v;
    if (reset) HIP_TRY(hipMemset(e->d_totals + 2, 0, 8));
    return KBBQ_OK;
}

int kbbq_device_or_pieces(kbbq_engine *e, void *dst_device, const void *src_device, uint64_t piece_words,
                          int32_t n_pieces, int32_t skip) {
    ENGINE_DEVICE(e);
    if (!e || !dst_device || !src_device || n_pieces < 1) return fail(KBBQ_EINVAL, "bad argument");
    if ((((uintptr_t)dst_device | (uintptr_t)src_device) & 15) || (piece_words & 1))
        return fail(KBBQ_EINVAL, "buffers must be 16-byte aligned and pieces an even number of words");
    if (!piece_words) return KBBQ_OK;
    Timed t(e, "k_or_pieces");
    hipLaunchKernelGGL(k_or_pieces, dim3((unsigned)((piece_words / 2 + 255) / 256)), dim3(256), 0, e->stream,
                       (uint64_t *)dst_device, (const uint64_t *)src_device, piece_words, n_pieces, skip);
    HIP_TRY(hipGetLastError());
    return KBBQ_OK;
}

int kbbq_filter_set_inserted(kbbq_engine *e, int which, uint64_t inserted) {
    ENGINE_DEVICE(e);
    { int frc = bucket_flush_all(e); if (frc) return frc; }      // deferred inserts reach the filters first (bucket.h)
    if (!e || which < 0 || which > 1) return fail(KBBQ_EINVAL, "bad argument");
    HIP_TRY(hipMemcpyAsync(e->filt[which].d_inserted, &inserted, 8, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return KBBQ_OK;
}

// ---- staging
// One pass over the text, 64 bases per step with the three bit streams accumulated in registers; large batches are
// packed by a few threads (the ranges are independent 64-base words).
static void pack_words(const uint8_t *seq, uint64_t n_bases, uint64_t w_lo, uint64_t w_hi, uint64_t *bases, uint64_t *nmask,
                       uint64_t *offcase, uint64_t *n_off) {
    // seq_nt16_int[seq_nt16_table[ch]] (bloom.hh:351): A/a=0 C/c=1 G/g=2 T/t=3 and '0'..'3'; all else non-ACGT (bit 2);
    // bit 3: an ACGT base whose raw character is not the upper-case letter (kbbq_reads.offcase)
    uint8_t lut[256];
    memset(lut, 4, sizeof lut);
    lut['A'] = 0; lut['C'] = 1; lut['G'] = 2; lut['T'] = 3;
    lut['a'] = 8; lut['c'] = 9; lut['g'] = 10; lut['t'] = 11;
    lut['0'] = 8; lut['1'] = 9; lut['2'] = 10; lut['3'] = 11;
    uint64_t odd_total = 0;
    for (uint64_t w = w_lo; w < w_hi; ++w) {
        const uint64_t first = w * 64;
        const int n = (int)std::min<uint64_t>(64, n_bases > first ? n_bases - first : 0);
        uint64_t b0 = 0, b1 = 0, nm = 0, oc = 0;
        for (int j = 0; j < n; ++j) {
            const uint8_t c = lut[seq[first + j]];
            const uint64_t code = (c & 4) ? 0 : (uint64_t)(c & 3);
            if (j < 32) b0 |= code << (2 * j); else b1 |= code << (2 * (j - 32));
            nm |= (uint64_t)((c >> 2) & 1) << j;
            oc |= (uint64_t)((c >> 3) & 1) << j;
        }
        bases[2 * w] = b0;
        bases[2 * w + 1] = b1;
        nmask[w] = nm;
        if (offcase) offcase[w] = oc;
        odd_total += (uint64_t)__builtin_popcountll(oc);
    }
    if (n_off) *n_off = odd_total;
}

static int pack_impl(const uint8_t *seq, uint64_t n_bases, uint64_t *bases_out, uint64_t *nmask_out, uint64_t *offcase_out, uint64_t *n_offcase) {
    // arrays hold n/32+2 and n/64+2 words: everything up to the spare words is written
    const uint64_t words = n_bases / 64 + 1;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned n_threads = n_bases < (1u << 22) ? 1u : std::min(8u, hw);
    std::vector<uint64_t> odd(n_threads, 0);
    if (n_threads == 1) {
        pack_words(seq, n_bases, 0, words, bases_out, nmask_out, offcase_out, &odd[0]);
    } else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_threads; ++t) {
            const uint64_t lo = words * t / n_threads, hi = words * (t + 1) / n_threads;
            pool.emplace_back(pack_words, seq, n_bases, lo, hi, bases_out, nmask_out, offcase_out, &odd[t]);
        }
        for (auto &th : pool) th.join();
    }
    // the spare words behind the data (the kernels read unaligned 64-bit windows)
    for (uint64_t i = 2 * words; i < n_bases / 32 + 2; ++i) bases_out[i] = 0;
    nmask_out[words] = 0;
    if (offcase_out) offcase_out[words] = 0;
    uint64_t total = 0;
    for (uint64_t x : odd) total += x;
    if (n_offcase) *n_offcase = total;
    return KBBQ_OK;
}

int kbbq_pack_bases(const uint8_t *seq, uint64_t n_bases, uint64_t *bases_out, uint64_t *nmask_out) {
    if (!seq || !bases_out || !nmask_out) return fail(KBBQ_EINVAL, "null argument");
    return pack_impl(seq, n_bases, bases_out, nmask_out, nullptr, nullptr);
}

int kbbq_pack_bases_case(const uint8_t *seq, uint64_t n_bases, uint64_t *bases_out, uint64_t *nmask_out, uint64_t *offcase_out,
                         uint64_t *n_offcase) {
    if (!seq || !bases_out || !nmask_out || !offcase_out) return fail(KBBQ_EINVAL, "null argument");
    return pack_impl(seq, n_bases, bases_out, nmask_out, offcase_out, n_offcase);
}

int kbbq_reads_upload(kbbq_engine *e, const kbbq_reads *host, kbbq_reads *dev) {
    // e may be NULL: batches can be made resident before the engine (whose size depends on them) exists
    if (!host || !dev) return fail(KBBQ_EINVAL, "null argument");
    int cur_dev = 0;
    HIP_TRY(hipGetDevice(&cur_dev));
    DeviceGuard guard(e ? e->p.device : cur_dev);
    HIP_TRY(guard.err);
    if (host->on_device) return fail(KBBQ_EINVAL, "batch is already on the device");
    *dev = *host;
    dev->on_device = 1;
    dev->hint_sampled = nullptr;
    dev->hint_trusted = nullptr;
    dev->bases = nullptr; dev->nmask = nullptr; dev->qual = nullptr; dev->offsets = nullptr; dev->flags = nullptr; dev->rg = nullptr;
    dev->offcase = nullptr;
    // all arrays travel on ONE copy stream (the engine's, or the device's shared one when there is no engine yet) as
    // asynchronous copies -- DMA straight from the caller's memory when it is page-locked (kbbq_host_alloc) -- and the
    // call returns when the last one has landed: kernels of earlier batches keep running underneath
    hipStream_t cs = e ? e->copy : nullptr;
    if (!cs) {
        int cur = 0;
        HIP_TRY(hipGetDevice(&cur));
        if (!(cs = shared_copy_stream(cur))) return fail(KBBQ_EIO, "no copy stream for device %d", cur);
    }
#define UP(field, type, count, pad)                                                               \
    if (host->field) {                                                                            \
        void *d = nullptr;                                                                        \
        hipError_t he = hipMalloc(&d, ((count) + (pad)) * sizeof(type));                          \
        if (he == hipSuccess) {                                                                   \
            dev->field = (const type *)d;                                                         \
            if (pad) he = hipMemsetAsync((char *)d + (count) * sizeof(type), 0, (pad) * sizeof(type), cs); \
        }                                                                                         \
        if (he == hipSuccess) he = hipMemcpyAsync(d, host->field, (count) * sizeof(type), hipMemcpyHostToDevice, cs); \
        if (he != hipSuccess) {                                                                   \
            hipStreamSynchronize(cs);                                                             \
            kbbq_reads_free(nullptr, dev);   /* what was allocated so far */                      \
            return fail(he == hipErrorOutOfMemory ? KBBQ_ENOMEM : KBBQ_EIO, "upload of %s: %s", #field, hipGetErrorString(he)); \
        }                                                                                         \
    }
    UP(bases, uint64_t, host->n_bases / 32 + 1, 1)
    UP(nmask, uint64_t, host->n_bases / 64 + 1, 1)
    UP(qual, uint8_t, host->n_bases, 16)
    UP(offsets, uint64_t, host->n_reads + 1, 0)
    UP(flags, uint8_t, host->n_reads, 0)
    UP(rg, uint16_t, host->n_reads, 0)
    UP(offcase, uint64_t, host->n_bases / 64 + 1, 1)
#undef UP
    HIP_TRY(hipStreamSynchronize(cs));
    return KBBQ_OK;
}

// A device batch copied to another device (or the same one): what a single-process multi-device caller hands the engines of
// the other devices (their shards of a data set that was made resident on the first).  with_hints: zeroed hint arrays too.
int kbbq_reads_clone(const kbbq_reads *src, int32_t device, int32_t with_hints, kbbq_reads *out) {
    if (!src || !out || !src->on_device) return fail(KBBQ_EINVAL, "not a device batch");
    DeviceGuard guard(device);
    HIP_TRY(guard.err);
    hipStream_t cs = shared_copy_stream(device);
    if (!cs) return fail(KBBQ_EIO, "no copy stream for device %d", device);
    *out = *src;
    out->hint_sampled = nullptr; out->hint_trusted = nullptr;
    out->bases = nullptr; out->nmask = nullptr; out->qual = nullptr; out->offsets = nullptr; out->flags = nullptr; out->rg = nullptr; out->offcase = nullptr;
#define CL(field, type, count, pad)                                                               \
    if (src->field) {                                                                             \
        void *d = nullptr;                                                                        \
        hipError_t he = hipMalloc(&d, ((count) + (pad)) * sizeof(type));                          \
        if (he == hipSuccess) {                                                                   \
            out->field = (const type *)d;                                                         \
            if (pad) he = hipMemsetAsync((char *)d + (count) * sizeof(type), 0, (pad) * sizeof(type), cs); \
        }                                                                                         \
        if (he == hipSuccess) he = hipMemcpyAsync(d, src->field, (count) * sizeof(type), hipMemcpyDefault, cs); \
        if (he != hipSuccess) {                                                                   \
            hipStreamSynchronize(cs);                                                             \
            kbbq_reads_free(nullptr, out);                                                        \
            return fail(he == hipErrorOutOfMemory ? KBBQ_ENOMEM : KBBQ_EIO, "copy of %s: %s", #field, hipGetErrorString(he)); \
        }                                                                                         \
    }
    CL(bases, uint64_t, src->n_bases / 32 + 1, 1)
    CL(nmask, uint64_t, src->n_bases / 64 + 1, 1)
    CL(qual, uint8_t, src->n_bases, 16)
    CL(offsets, uint64_t, src->n_reads + 1, 0)
    CL(flags, uint8_t, src->n_reads, 0)
    CL(rg, uint16_t, src->n_reads, 0)
    CL(offcase, uint64_t, src->n_bases / 64 + 1, 1)
#undef CL
    HIP_TRY(hipStreamSynchronize(cs));
    if (with_hints) {
        const int rc = kbbq_reads_alloc_hints(out);      // (on the current device: the guard's)
        if (rc) { kbbq_reads_free(nullptr, out); return rc; }
    }
    return KBBQ_OK;
}

// ---- page-locked host memory for batches, and what the host link delivers
int kbbq_host_alloc(size_t bytes, void **out) {
    if (!out || !bytes) return fail(KBBQ_EINVAL, "bad argument");
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return KBBQ_OK;
}

int kbbq_host_free(void *p) {
    if (p) HIP_TRY(hipHostFree(p));
    return KBBQ_OK;
}

int kbbq_device_alloc(kbbq_engine *e, size_t bytes, void **out) {
    ENGINE_DEVICE(e);
    if (!out || !bytes) return fail(KBBQ_EINVAL, "bad argument");
    *out = nullptr;
    HIP_TRY(hipMalloc(out, bytes));
    return KBBQ_OK;
}

int kbbq_device_free(kbbq_engine *e, void *p) {
    ENGINE_DEVICE(e);
    if (p) {
        int rc = sync_engine(e);      // nothing queued may still use it
        if (rc) return rc;
        HIP_TRY(hipFree(p));
    }
    return KBBQ_OK;
}

int kbbq_measure_host_link(int32_t device, uint64_t bytes, double *h2d_gbps, double *d2h_gbps) {
    if (!bytes) return fail(KBBQ_EINVAL, "bad argument");
    int cur_dev = 0;
    HIP_TRY(hipGetDevice(&cur_dev));
    DeviceGuard guard(device >= 0 ? device : cur_dev);
    HIP_TRY(guard.err);
    void *h = nullptr, *d = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st = nullptr;
    int rc = KBBQ_OK;
    float ms_up = 0, ms_dn = 0;
#define LINK_TRY(x) do { hipError_t _e = (x); if (_e != hipSuccess && rc == KBBQ_OK) rc = fail(KBBQ_EIO, "%s: %s", #x, hipGetErrorString(_e)); } while (0)
    LINK_TRY(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    LINK_TRY(hipMalloc(&d, bytes));
    LINK_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    LINK_TRY(hipEventCreate(&a));
    LINK_TRY(hipEventCreate(&b));
    if (rc == KBBQ_OK) {
        memset(h, 1, bytes);
        LINK_TRY(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st));      // warm-up
        LINK_TRY(hipEventRecord(a, st));
        LINK_TRY(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st));
        LINK_TRY(hipEventRecord(b, st));
        LINK_TRY(hipEventSynchronize(b));
        LINK_TRY(hipEventElapsedTime(&ms_up, a, b));
        LINK_TRY(hipEventRecord(a, st));
        LINK_TRY(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st));
        LINK_TRY(hipEventRecord(b, st));
        LINK_TRY(hipEventSynchronize(b));
        LINK_TRY(hipEventElapsedTime(&ms_dn, a, b));
    }
#undef LINK_TRY
    if (a) hipEventDestroy(a);
    if (b) hipEventDestroy(b);
    if (st) hipStreamDestroy(st);
    hipFree(d);
    if (h) hipHostFree(h);
    if (rc) return rc;
    if (h2d_gbps) *h2d_gbps = ms_up > 0 ? (double)bytes / ms_up / 1e6 : 0;
    if (d2h_gbps) *d2h_gbps = ms_dn > 0 ? (double)bytes / ms_dn / 1e6 : 0;
    return KBBQ_OK;
}

// both directions at once (two streams): what pass 4 of the host-batch mode can hope for
int kbbq_measure_host_link_duplex(int32_t device, uint64_t bytes, double *h2d_gbps, double *d2h_gbps) {
    if (!bytes) return fail(KBBQ_EINVAL, "bad argument");
    int cur_dev = 0;
    HIP_TRY(hipGetDevice(&cur_dev));
    DeviceGuard guard(device >= 0 ? device : cur_dev);
    HIP_TRY(guard.err);
    void *h_up = nullptr, *h_dn = nullptr, *d_up = nullptr, *d_dn = nullptr;
    hipEvent_t a0 = nullptr, a1 = nullptr, b0 = nullptr, b1 = nullptr;
    hipStream_t s_up = nullptr, s_dn = nullptr;
    int rc = KBBQ_OK;
    float ms_up = 0, ms_dn = 0;
#define LINK_TRY(x) do { hipError_t _e = (x); if (_e != hipSuccess && rc == KBBQ_OK) rc = fail(KBBQ_EIO, "%s: %s", #x, hipGetErrorString(_e)); } while (0)
    LINK_TRY(hipHostMalloc(&h_up, bytes, hipHostMallocDefault));
    LINK_TRY(hipHostMalloc(&h_dn, bytes, hipHostMallocDefault));
    LINK_TRY(hipMalloc(&d_up, bytes));
    LINK_TRY(hipMalloc(&d_dn, bytes));
    LINK_TRY(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking));
    LINK_TRY(hipStreamCreateWithFlags(&s_dn, hipStreamNonBlocking));
    LINK_TRY(hipEventCreate(&a0)); LINK_TRY(hipEventCreate(&a1)); LINK_TRY(hipEventCreate(&b0)); LINK_TRY(hipEventCreate(&b1));
    if (rc == KBBQ_OK) {
        memset(h_up, 1, bytes);
        memset(h_dn, 0, bytes);
        LINK_TRY(hipMemcpyAsync(d_up, h_up, bytes, hipMemcpyHostToDevice, s_up));      // warm-up, both ways
        LINK_TRY(hipMemcpyAsync(h_dn, d_dn, bytes, hipMemcpyDeviceToHost, s_dn));
        LINK_TRY(hipStreamSynchronize(s_up));
        LINK_TRY(hipStreamSynchronize(s_dn));
        LINK_TRY(hipEventRecord(a0, s_up));
        LINK_TRY(hipEventRecord(b0, s_dn));
        for (int i = 0; i < 3; ++i) {
            LINK_TRY(hipMemcpyAsync(d_up, h_up, bytes, hipMemcpyHostToDevice, s_up));
            LINK_TRY(hipMemcpyAsync(h_dn, d_dn, bytes, hipMemcpyDeviceToHost, s_dn));
        }
        LINK_TRY(hipEventRecord(a1, s_up));
        LINK_TRY(hipEventRecord(b1, s_dn));
        LINK_TRY(hipEventSynchronize(a1));
        LINK_TRY(hipEventSynchronize(b1));
        LINK_TRY(hipEventElapsedTime(&ms_up, a0, a1));
        LINK_TRY(hipEventElapsedTime(&ms_dn, b0, b1));
    }
#undef LINK_TRY
    hipEvent_t evs[] = {a0, a1, b0, b1};
    for (hipEvent_t ev : evs) if (ev) hipEventDestroy(ev);
    if (s_up) hipStreamDestroy(s_up);
    if (s_dn) hipStreamDestroy(s_dn);
    hipFree(d_up); hipFree(d_dn);
    if (h_up) hipHostFree(h_up);
    if (h_dn) hipHostFree(h_dn);
    if (rc) return rc;
    if (h2d_gbps) *h2d_gbps = ms_up > 0 ? 3.0 * (double)bytes / ms_up / 1e6 : 0;
    if (d2h_gbps) *d2h_gbps = ms_dn > 0 ? 3.0 * (double)bytes / ms_dn / 1e6 : 0;
    return KBBQ_OK;
}

int kbbq_reads_free(kbbq_engine *e, kbbq_reads *dev) {
    if (!dev) return fail(KBBQ_EINVAL, "null argument");
    if (!dev->on_device) return fail(KBBQ_EINVAL, "not a device batch");
    int cur_dev = 0;
    HIP_TRY(hipGetDevice(&cur_dev));
    DeviceGuard guard(e ? e->p.device : cur_dev);
    HIP_TRY(guard.err);
    if (e) {
        int rc = sync_engine(e);
        if (rc) return rc;
    } else {
        HIP_TRY(hipDeviceSynchronize());
    }
    hipFree((void *)dev->bases); hipFree((void *)dev->nmask); hipFree((void *)dev->qual);
    hipFree((void *)dev->offsets); hipFree((void *)dev->flags); hipFree((void *)dev->rg); hipFree((void *)dev->offcase);
    memset(dev, 0, sizeof *dev);
    return KBBQ_OK;
}

int kbbq_reads_alloc_hints(kbbq_reads *dev) {
    if (!dev || !dev->on_device) return fail(KBBQ_EINVAL, "not a device batch");
    if (dev->hint_sampled || dev->hint_trusted) return fail(KBBQ_ESTATE, "batch already has hint arrays");
    const size_t bytes = (dev->n_bases / 64 + 2) * 8;
    void *a = nullptr, *b = nullptr;
    hipError_t he = hipMalloc(&a, bytes);
    if (he == hipSuccess) he = hipMalloc(&b, bytes);
    if (he == hipSuccess) he = hipMemset(a, 0, bytes);
    if (he == hipSuccess) he = hipMemset(b, 0, bytes);
    if (he != hipSuccess) {
        hipFree(a); hipFree(b);
        return fail(he == hipErrorOutOfMemory ? KBBQ_ENOMEM : KBBQ_EIO, "hint arrays: %s", hipGetErrorString(he));
    }
    dev->hint_sampled = (uint64_t *)a;
    dev->hint_trusted = (uint64_t *)b;
    return KBBQ_OK;
}

int kbbq_reads_free_hints(kbbq_reads *dev) {
    if (!dev || !dev->on_device) return fail(KBBQ_EINVAL, "not a device batch");
    HIP_TRY(hipDeviceSynchronize());
    hipFree(dev->hint_sampled); hipFree(dev->hint_trusted);
    dev->hint_sampled = nullptr;
    dev->hint_trusted = nullptr;
    return KBBQ_OK;
}

int kbbq_device_memory(int32_t device, uint64_t *free_bytes, uint64_t *total_bytes) {
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    if (device >= 0 && device != cur) HIP_TRY(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (device >= 0 && device != cur) HIP_TRY(hipSetDevice(cur));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return KBBQ_OK;
}

// ---- pass 1
int kbbq_count_kmer_positions(kbbq_engine *e, const kbbq_reads *reads, uint64_t *out) {
    ENGINE_DEVICE(e);
    if (!e || !reads || !out) return fail(KBBQ_EINVAL, "null argument");
    HostBatchDone host_done(e, reads);
    ReadsDev R; int max_len;
    int rc = device_view(e, reads, &R, &max_len, false);
    if (rc) return rc;
    const uint64_t *kofs;
    return kmer_prefix(e, R, &kofs, out);      // (a ragged batch: the prefix scan has finished; a uniform one: no device work)
}

}  // extern "C"

constexpr int kStagedMax = 512;      // longest read the staged kernels take (Stage<8>); longer: long_reads.h
constexpr int kTallyMaxWindows = 24;  // windows of 192 cycles k_tally takes as launches of their own (reads up to 4608 bases); longer
                                      // reads: one launch, cycle counters straight to the histograms (run_tally)
template <template <int> class Launcher, typename... Args>
static int dispatch_nw(int max_len, Args... args) {
    if (max_len <= 192) return Launcher<3>::go(args...);
    if (max_len <= 320) return Launcher<5>::go(args...);
    return Launcher<8>::go(args...);      // (reads longer than 512 bases: the callers branch to long_reads.h first)
}

template <int NW> struct LaunchSample {
    static int go(kbbq_engine *e, ReadsDev R, const uint64_t *mask, uint64_t mask_words, const uint64_t *kofs) {
        Timed t(e, "k_insert_sampled");
        hipLaunchKernelGGL((k_insert_marked<NW, false>), dim3(wave_grid(R.n_reads)), dim3(256), 0, e->stream, R, e->K,
                           e->filt[0].dev(), mask, mask_words, kofs, e->filt[0].d_inserted);
        HIP_TRY(hipGetLastError());
        return KBBQ_OK;
    }
};

// the same reads -> (block, pattern) records for the slice-bucketed insert (bucket.h)
template <int NW, int CH, int RPW>
static int launch_emit(kbbq_engine *e, int w, ReadsDev R, const uint64_t *mask, uint64_t mask_words, const uint64_t *kofs,
                       unsigned long long *inserted) {
    const BucketDev B = bucket_dev(e, w);
    const uint64_t n_tiles = (R.n_reads + 8 * RPW - 1) / (8 * RPW);
    const int grid = (int)std::min<uint64_t>(n_tiles, EMIT_GRID);
    const size_t lds = (size_t)8 * RPW * CH * 64 * 8;
    hipStream_t st = bucket_stream(e, w);
    Timed t(e, w ? "k_emit_trusted" : "k_emit_sampled", st);
    // (level-1 regions per emitting workgroup, only the marked k-mers hashed: the round-2 winners of bucket.h's variants)
#define KBBQ_EMIT(BYB)                                                                                                        \
    do {                                                                                                                      \
        const void *fn = (const void *)k_emit_marked<NW, CH, BYB, RPW, true, true>;                                            \
        size_t &raised = e->attr_lds_correct[fn];                                                                             \
        if (lds > 48 * 1024 && lds > raised) {                                                                                \
            HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                            \
            raised = lds;                                                                                                     \
        }                                                                                                                     \
        hipLaunchKernelGGL((k_emit_marked<NW, CH, BYB, RPW, true, true>), dim3(grid), dim3(BK_THREADS), lds, st, R, e->K,      \
                           e->filt[w].dev(), B, mask, mask_words, kofs, inserted);                                            \
    } while (0)
    if (w == 0) KBBQ_EMIT(false); else KBBQ_EMIT(true);
#undef KBBQ_EMIT
    HIP_TRY(hipGetLastError());
    return KBBQ_OK;
}

// NW by the longest read, CH by the most k-mer starts a read can have
static int dispatch_emit(kbbq_engine *e, int w, int max_len, ReadsDev R, const uint64_t *mask, uint64_t mask_words, const uint64_t *kofs,
                         unsigned long long *inserted) {
    const int max_nk = std::max(1, max_len - e->p.k + 1);
    if (max_len <= 192) {      // four reads per emitting wave (two and eight were measured in round 2: no better)
        if (max_nk <= 128) return launch_emit<3, 2, 4>(e, w, R, mask, mask_words, kofs, inserted);
        return launch_emit<3, 3, 4>(e, w, R, mask, mask_words, kofs, inserted);
    }
    if (max_len <= 320) return launch_emit<5, 5, 2>(e, w, R, mask, mask_words, kofs, inserted);
    return launch_emit<8, 8, 1>(e, w, R, mask, mask_words, kofs, inserted);
}

extern "C" {

int kbbq_sample_batch(kbbq_engine *e, const kbbq_reads *reads, uint64_t first_kmer_ordinal) {
    ENGINE_DEVICE(e);
    if (!e) return fail(KBBQ_EINVAL, "null engine");
    HostBatchDone host_done(e, reads);
    ReadsDev R; int max_len;
    int rc = device_view(e, reads, &R, &max_len, false);      // pass 1 reads bases only
    if (rc) return rc;
    const uint64_t *kofs; uint64_t n_draws;
    if ((rc = kmer_prefix(e, R, &kofs, &n_draws))) return rc;
    if (n_draws == 0) return KBBQ_OK;
    const bool overlap = reads->on_device && !e->opt.no_overlap;
    const int turn = overlap ? e->draw_turn : 0;
    const int slot = turn ? 13 : 0;
    if ((rc = ensure_scratch(e, slot, (n_draws / 64 + 2) * 8))) return rc;
    uint64_t *mask = (uint64_t *)e->scratch[slot];
    hipStream_t ds = overlap ? e->stream2 : e->stream;
    if (overlap) {
        e->draw_turn ^= 1;
        if (e->ins_pending[turn]) HIP_TRY(hipStreamWaitEvent(ds, e->ev_ins[turn], 0));   // the insert that last read this mask
    }
    {
        Timed t(e, "k_draw_mask", ds);
        const uint64_t lanes = (n_draws + DRAWS_PER_LANE - 1) / DRAWS_PER_LANE;
        // the host jumps to the batch's first draw (a few thousand xoshiro steps); the lanes then only jump by
        // their offset inside the batch, which has far fewer set bits than the file-wide ordinal
        uint64_t st[4];
        xoshiro_state_at(e->p.seed, first_kmer_ordinal, st);
        hipLaunchKernelGGL(k_draw_mask, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, ds,
                           st[0], st[1], st[2], st[3],
                           (uint64_t)0, n_draws, e->draw_threshold, e->draw_always ? 1 : 0, mask);
        HIP_TRY(hipGetLastError());
    }
    if (overlap) {
        HIP_TRY(hipEventRecord(e->ev_draw, ds));
        HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_draw, 0));
    }
    if (max_len > kStagedMax) {
        Timed t(e, "k_insert_sampled_long");
        hipLaunchKernelGGL(k_insert_marked_long<false>, dim3(wave_grid(R.n_reads)), dim3(256), 0, e->stream, R, e->K, e->filt[0].dev(),
                           (const uint64_t *)mask, n_draws / 64 + 2, kofs, e->filt[0].d_inserted);
        HIP_TRY(hipGetLastError());
        rc = KBBQ_OK;
    } else if (bucket_on(e, 0)) {
        // deferred: the k-mers become records now and reach the filter at the next flush (bucket.h)
        if ((rc = bucket_reserve(e, 0, (double)n_draws * std::min(1.0, e->p.alpha) * 1.01 + 4096.0, R.n_bases))) return rc;
        rc = dispatch_emit(e, 0, max_len, R, (const uint64_t *)mask, n_draws / 64 + 2, kofs, e->filt[0].d_inserted);
    } else {
        rc = dispatch_nw<LaunchSample>(max_len, e, R, (const uint64_t *)mask, n_draws / 64 + 2, kofs);
    }
    if (!rc) {     // (also for an in-order batch: a later overlapped draw into the same buffer must wait for this insert)
        HIP_TRY(hipEventRecord(e->ev_ins[turn], e->stream));
        e->ins_pending[turn] = true;
    }
    // (a host batch is the caller's again when the call returns: host_done waits for its copy, not for the kernels)
    return rc;
}

int kbbq_sample_finish(kbbq_engine *e, uint64_t *inserted) {
    ENGINE_DEVICE(e);
    { int frc = bucket_flush_all(e); if (frc) return frc; }      // deferred inserts reach the filters first (bucket.h)
    if (!e) return fail(KBBQ_EINVAL, "null engine");
    int rc = sync_engine(e);
    if (rc) return rc;
    if (inserted) HIP_TRY(hipMemcpy(inserted, e->filt[0].d_inserted, 8, hipMemcpyDeviceToHost));
    return KBBQ_OK;
}

// ---- between passes
int kbbq_compute_thresholds(kbbq_engine *e, const char *alpha_text, int32_t *thresholds_out, double *fpr_out,
                            char *p_text_out, size_t p_text_len) {
    ENGINE_DEVICE(e);
    { int frc = bucket_flush_all(e); if (frc) return frc; }      // deferred inserts reach the filters first (bucket.h)
    if (!e || !alpha_text) return fail(KBBQ_EINVAL, "null argument");
    int rc = sync_engine(e);
    if (rc) return rc;
    uint64_t inserted = 0;
    HIP_TRY(hipMemcpy(&inserted, e->filt[0].d_inserted, 8, hipMemcpyDeviceToHost));
    if (inserted == 0) return fail(KBBQ_ESTATE, "no k-mers were sampled");   // the reference divides by zero (bloom.hh:319)
    double fpr = 0;
    std::string p_text;
    e->thresholds = thresholds_from_counts(e->p.k, e->filt[0].spec.bits, inserted, e->filt[0].spec.n_salt, alpha_text, &fpr, &p_text);
    if (fpr_out) *fpr_out = fpr;
    if (p_text_out && p_text_len) snprintf(p_text_out, p_text_len, "%s", p_text.c_str());
    if (thresholds_out) memcpy(thresholds_out, e->thresholds.data(), (e->p.k + 1) * 4);
    rc = kbbq_set_thresholds(e, e->thresholds.data(), e->p.k + 1);
    if (rc) return rc;
    return fpr > .15 ? 1 : 0;   // kbbq.cc:306
}

int kbbq_set_thresholds(kbbq_engine *e, const int32_t *thresholds, int32_t n) {
    ENGINE_DEVICE(e);
    if (!e || !thresholds) return fail(KBBQ_EINVAL, "null argument");
    if (n != e->p.k + 1) return fail(KBBQ_EINVAL, "expected k+1 = %d thresholds", e->p.k + 1);
    if (thresholds != e->thresholds.data()) e->thresholds.assign(thresholds, thresholds + n);
    e->thresholds_set = true;
    return KBBQ_OK;
}

// ---- pass 2
}  // extern "C"


// k_infer<SUB>: every how many k-mer starts phase 1 may skip a lookup (4, 8 or 16) and how far from the read ends it
// starts doing so (kernels.h); false when the thresholds leave no room for it.  A window of n starts whose k-mers are all
// present must still be decided without its skipped lookups: skipped <= n - thr[n] - 1.  Inside the read a window of k
// starts holds at most ceil(k / period) skipped ones; at the ends the windows are the prefixes [0, i] (and, mirrored, the
// suffixes), which hold the skipped starts from `edge` on -- counted as the worst case over the residues.  Only speed
// depends on this choice; every base still gets its exact decision (phase 2).
static bool infer_subset_plan(const std::vector<int> &thr, int k, int *edge_out, int *pmask_out) {
    if (k < 8 || (int)thr.size() < k + 1) return false;
    for (int period = 4; period <= 16; period *= 2) {
        if ((k + period - 1) / period > k - thr[k] - 1) continue;
        for (int edge = 0; edge < k; ++edge) {
            bool ok = true;
            for (int n = edge + 1; n <= k && ok; ++n)      // the window of the first (last) n starts
                ok = (n - edge + period - 1) / period <= n - thr[n] - 1;
            if (ok) { *edge_out = edge; *pmask_out = period - 1; return true; }
        }
    }
    return false;
}

template <int NW> struct LaunchTrusted {
    static int go(kbbq_engine *e, ReadsDev R, uint32_t *take_bits, uint32_t *err_out, int max_len) {
        Thresholds thr;
        memset(&thr, 0, sizeof thr);
        for (size_t i = 0; i < e->thresholds.size() && i <= KBBQ_MAX_KMER; ++i) thr.v[i] = e->thresholds[i];
        HIP_TRY(hipMemsetAsync(e->d_tickets, 0, 4, e->stream));      // the kernel's chunk counter (ReadChunks)
        {
            Timed t(e, "k_infer");
            int edge = -1, pmask = 3;
            if (e->opt.infer_subset && !infer_subset_plan(e->thresholds, e->p.k, &edge, &pmask)) edge = -1;
            if (edge >= 0 && getenv("KBBQ_SUBSET_EDGE")) edge = atoi(getenv("KBBQ_SUBSET_EDGE"));        // (diagnostic overrides: results
            if (edge >= 0 && getenv("KBBQ_SUBSET_PMASK")) pmask = atoi(getenv("KBBQ_SUBSET_PMASK"));     //  never depend on either)
            // NK: chunks of 64 lanes that can hold a k-mer start (150-base reads, k = 32: 119 starts, two of the three chunks)
            const bool short_nk = std::max(1, max_len - e->p.k + 1) <= (NW - 1) * 64;
#define KBBQ_LAUNCH_INFER(NK_, SUB_)                                                                                              \
    hipLaunchKernelGGL((k_infer<NW, NK_, 1, SUB_>), dim3(wave_grid(R.n_reads, shared_cap(e, e->opt.infer_blocks))), dim3(256), 0, e->stream, R, e->K, e->filt[0].dev(), thr, \
                       take_bits, e->filt[1].d_inserted, err_out, e->d_qpresent, e->d_counters + 3, e->d_tickets, edge, pmask)
            if (edge >= 0) { if (short_nk) KBBQ_LAUNCH_INFER(NW - 1, true); else KBBQ_LAUNCH_INFER(NW, true); }
            else { if (short_nk) KBBQ_LAUNCH_INFER(NW - 1, false); else KBBQ_LAUNCH_INFER(NW, false); }
#undef KBBQ_LAUNCH_INFER
            HIP_TRY(hipGetLastError());
        }
        if (bucket_on(e, 1)) {
            // The emits of pass 2 run on the side stream: ALU-bound, beside the next batch's k_infer, which is bound by random
            // HBM lines (the emit costs it about half of its own 1.8 ms).  A flush -- split + apply, whenever the record buffers
            // fill -- runs where Options::pass2_side says: alone on the engine's stream between two k_infer (2, the default:
            // k_apply and k_infer both wait for the L2's request path and only slow each other down), or on the side stream
            // too (1, round 3's form).  (Overflow records are inserted directly by the emit kernel; emit and k_apply never
            // touch the trusted filter at the same time in either mode: same stream, or ordered by events in bucket_flush.
            // The direct inserts of long reads are ordered against the side stream by events, kbbq_trusted_batch.)
            // The exclusive duration of k_infer -- the kernel the roofline is quoted for -- comes from the in-order run
            // (KBBQ_F_NO_OVERLAP).
            const bool side = e->opt.pass2_side != 0 && !e->opt.no_overlap;
            e->bk.stream[1] = side ? e->stream2 : e->stream;
            if (side) {
                HIP_TRY(hipEventRecord(e->ev_infer, e->stream));
                HIP_TRY(hipStreamWaitEvent(e->stream2, e->ev_infer, 0));
                e->cur_slot_side = true;      // (a host batch: its staging slot is read on the side stream as well)
            }
            bucket_poll_estimate(e);
            e->bk.frac_used = e->bk.frac_trusted;
            int rc = bucket_reserve(e, 1, (double)R.n_bases * e->bk.frac_trusted + 4096.0, R.n_bases);
            if (rc) return rc;
            return dispatch_emit(e, 1, max_len, R, (const uint64_t *)take_bits, R.n_bases / 64 + 2, (const uint64_t *)nullptr,
                                 (unsigned long long *)nullptr);
        }
        {
            Timed t(e, "k_insert_trusted");
            hipLaunchKernelGGL((k_insert_marked<NW, true>), dim3(wave_grid(R.n_reads)), dim3(256), 0, e->stream, R, e->K,
                               e->filt[1].dev(), (const uint64_t *)take_bits, R.n_bases / 64 + 2,
                               (const uint64_t *)nullptr, (unsigned long long *)nullptr);
            HIP_TRY(hipGetLastError());
        }
        return KBBQ_OK;
    }
};
extern "C" {

static int bit_out_begin(kbbq_engine *e, const kbbq_reads *reads, uint64_t *user, int slot, uint32_t **dev) {
    *dev = nullptr;
    if (!user) return KBBQ_OK;
    const size_t bytes = (reads->n_bases / 64 + 2) * 8;
    if (reads->on_device) {
        *dev = (uint32_t *)user;
    } else {
        int rc = ensure_scratch(e, slot, bytes);
        if (rc) return rc;
        *dev = (uint32_t *)e->scratch[slot];
    }
    HIP_TRY(hipMemsetAsync(*dev, 0, (reads->n_bases / 64 + 1) * 8, e->stream));
    return KBBQ_OK;
}
static int bit_out_end(kbbq_engine *e, const kbbq_reads *reads, uint64_t *user, uint32_t *dev) {
    if (!user || reads->on_device) return KBBQ_OK;
    HIP_TRY(hipMemcpyAsync(user, dev, (reads->n_bases / 64 + 1) * 8, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return KBBQ_OK;
}

int kbbq_trusted_batch(kbbq_engine *e, const kbbq_reads *reads, uint64_t *infer_errors_out) {
    ENGINE_DEVICE(e);
    { int frc = bucket_flush(e, 0); if (frc) return frc; }      // pass 2 reads the sampled filter: its deferred inserts go in first
    if (!e) return fail(KBBQ_EINVAL, "null engine");
    if (!e->thresholds_set) return fail(KBBQ_ESTATE, "thresholds are not set");
    HostBatchDone host_done(e, reads);
    ReadsDev R; int max_len;
    int rc = device_view(e, reads, &R, &max_len);
    if (rc) return rc;
    uint32_t *d_err;
    if ((rc = bit_out_begin(e, reads, infer_errors_out, 2, &d_err))) return rc;
    // the insert decisions of k_infer: the caller's hint array when there is one (pass 3 then reuses
    // them), a scratch bit array otherwise
    uint32_t *take_bits = R.hint_trusted;
    int take_slot = -1;
    if (!take_bits) {
        // two scratch arrays in turn: the side stream may still be reading the one before last (ev_take)
        take_slot = e->take_turn;
        e->take_turn ^= 1;
        const int sl = take_slot ? 22 : 8;
        if (e->take_pending[take_slot]) {
            HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_take[take_slot], 0));
            e->take_pending[take_slot] = false;
        }
        if ((rc = ensure_scratch(e, sl, (R.n_bases / 64 + 2) * 8))) return rc;
        take_bits = (uint32_t *)e->scratch[sl];
        HIP_TRY(hipMemsetAsync(take_bits, 0, (R.n_bases / 64 + 2) * 8, e->stream));
    }
    if (max_len > kStagedMax) {
        Thresholds thr;
        memset(&thr, 0, sizeof thr);
        for (size_t i = 0; i < e->thresholds.size() && i <= KBBQ_MAX_KMER; ++i) thr.v[i] = e->thresholds[i];
        {
            Timed t(e, "k_infer_long");
            hipLaunchKernelGGL(k_infer_long, dim3(wave_grid(R.n_reads)), dim3(256), 0, e->stream, R, e->K, e->filt[0].dev(), thr, take_bits,
                               e->filt[1].d_inserted, d_err, e->d_qpresent, e->d_counters + 3);
            HIP_TRY(hipGetLastError());
        }
        // These inserts are atomic ORs straight into the trusted filter.  Batches of short reads of the same pass may have
        // records pending, or a flush (k_apply: load a slice, OR, store it back -- no atomics) running, on the side
        // stream: the direct inserts wait for what is queued there, and what is queued there later waits for them.
        const bool side_busy = e->bk.stream[1] && e->bk.stream[1] != e->stream;
        if (side_busy) {
            HIP_TRY(hipEventRecord(e->bk.ev_flush, e->bk.stream[1]));
            HIP_TRY(hipStreamWaitEvent(e->stream, e->bk.ev_flush, 0));
        }
        {
            Timed t(e, "k_insert_trusted_long");
            hipLaunchKernelGGL(k_insert_marked_long<true>, dim3(wave_grid(R.n_reads)), dim3(256), 0, e->stream, R, e->K, e->filt[1].dev(),
                               (const uint64_t *)take_bits, R.n_bases / 64 + 2, (const uint64_t *)nullptr, (unsigned long long *)nullptr);
            HIP_TRY(hipGetLastError());
        }
        if (side_busy) {
            HIP_TRY(hipEventRecord(e->ev_infer, e->stream));
            HIP_TRY(hipStreamWaitEvent(e->bk.stream[1], e->ev_infer, 0));
        }
    } else if ((rc = dispatch_nw<LaunchTrusted>(max_len, e, R, take_bits, d_err, max_len))) return rc;
    if (take_slot >= 0 && bucket_stream(e, 1) != e->stream && e->bk.mode[1] == 1 && max_len <= kStagedMax) {
        HIP_TRY(hipEventRecord(e->ev_take[take_slot], bucket_stream(e, 1)));      // the emit has read this scratch array
        e->take_pending[take_slot] = true;
    }
    if ((rc = bit_out_end(e, reads, infer_errors_out, d_err))) return rc;
    return KBBQ_OK;      // device batches are queued; a host batch has been copied (host_done), its kernels are queued too
}

int kbbq_trusted_finish(kbbq_engine *e, uint64_t *inserted) {
    ENGINE_DEVICE(e);
    { int frc = bucket_flush_all(e); if (frc) return frc; }      // deferred inserts reach the filters first (bucket.h)
    if (!e) return fail(KBBQ_EINVAL, "null engine");
    int rc = sync_engine(e);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(e->qpresent, e->d_qpresent, 32, hipMemcpyDeviceToHost));      // the tally sizes its LDS tables by these (run_tally)
    e->qpresent_known = true;
    if (inserted) HIP_TRY(hipMemcpy(inserted, e->filt[1].d_inserted, 8, hipMemcpyDeviceToHost));
    return KBBQ_OK;
}

// ---- pass 3
}  // extern "C"

template <int NW> struct LaunchScan {
    static int go(kbbq_engine *e, ReadsDev R, uint64_t *tmask, uint8_t *dirty, uint32_t *err_bits, int fast, int max_len) {
        unsigned int *ticket = e->d_tickets + 1 + (e->cur_cnt != e->d_counters ? 1 : 0);      // per side of pass 3
        HIP_TRY(hipMemsetAsync(ticket, 0, 4, e->cur));
        Timed t(e, "k_scan_trusted", e->cur);
        if (std::max(1, max_len - e->p.k + 1) <= (NW - 1) * 64)      // (NK: see k_infer)
            hipLaunchKernelGGL((k_scan_trusted<NW, NW - 1>), dim3(wave_grid(R.n_reads, e->pass3_shared ? shared_cap(e, e->opt.scan_blocks, 4, max_len, !R.offsets) : 0)), dim3(256), 0, e->cur, R, e->K,
                               e->filt[1].dev(), tmask, dirty, err_bits, e->cur_cnt, fast, ticket);
        else
            hipLaunchKernelGGL((k_scan_trusted<NW, NW>), dim3(wave_grid(R.n_reads, e->pass3_shared ? shared_cap(e, e->opt.scan_blocks, 4, max_len, !R.offsets) : 0)), dim3(256), 0, e->cur, R, e->K,
                               e->filt[1].dev(), tmask, dirty, err_bits, e->cur_cnt, fast, ticket);
        HIP_TRY(hipGetLastError());
        return KBBQ_OK;
    }
};

template <int MAXL, int BLOCK>
static int launch_correct(kbbq_engine *e, ReadsDev R, const uint32_t *list, const unsigned long long *n_list, const uint64_t *tmask, int tw,
                          uint32_t *err_bits, uint32_t *patch, int max_len = 0, int side = 0) {
    typedef Corrector<MAXL> C;
    if (MAXL == 0) {
        // reads longer than 512 bases: the lane's working words (C::words_for(len) of them) live in a global scratch
        // array, one slice per lane of the grid; at most 1 GiB of it
        const int dyn_len = ((max_len + 31) / 32) * 32;
        const size_t words = (size_t)C::words_for(dyn_len);
        const uint64_t fit = std::max<uint64_t>(1, ((uint64_t)1 << 30) / (words * 4 * BLOCK));
        const int blocks = (int)std::min<uint64_t>(std::min<uint64_t>((R.n_reads + BLOCK - 1) / BLOCK, 256 * 4), fit);
        const int slot = side ? 21 : 20;
        int rc = ensure_scratch(e, slot, (size_t)blocks * BLOCK * words * 4);
        if (rc) return rc;
        Timed t(e, "k_correct_long", e->cur);
        hipLaunchKernelGGL((k_correct<MAXL, BLOCK>), dim3(blocks), dim3(BLOCK), 0, e->cur, R, e->K, e->filt[1].dev(), list,
                           n_list, tmask, tw, err_bits, patch, e->cur_cnt, dyn_len, (uint32_t *)e->scratch[slot]);
        HIP_TRY(hipGetLastError());
        return KBBQ_OK;
    }
    const size_t lds = (size_t)C::WORDS * BLOCK * 4;
    size_t &raised = e->attr_lds_correct[(const void *)k_correct<MAXL, BLOCK>];
    if (lds > raised) {
        HIP_TRY(hipFuncSetAttribute((const void *)k_correct<MAXL, BLOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        raised = lds;
    }
    Timed t(e, "k_correct", e->cur);
    // the work-list length is only known on the device: size the grid for the batch and let lanes stride
    const int blocks = (int)std::min<uint64_t>((R.n_reads + BLOCK - 1) / BLOCK, 256 * 8);
    hipLaunchKernelGGL((k_correct<MAXL, BLOCK>), dim3(blocks), dim3(BLOCK), lds, e->cur, R, e->K, e->filt[1].dev(), list,
                       n_list, tmask, tw, err_bits, patch, e->cur_cnt, 0, (uint32_t *)nullptr);
    HIP_TRY(hipGetLastError());
    return KBBQ_OK;
}

template <int NB, int NN>
static int launch_correct_wave(kbbq_engine *e, ReadsDev R, const uint32_t *list, const uint64_t *tmask, int tw,
                               uint32_t *err_bits, uint32_t *patch) {
    unsigned int *ticket = e->d_tickets + 3 + (e->cur_cnt != e->d_counters ? 1 : 0);      // per side of pass 3
    HIP_TRY(hipMemsetAsync(ticket, 0, 4, e->cur));
    Timed t(e, "k_correct_wave", e->cur);
    const int blocks = wave_grid(R.n_reads, e->pass3_shared ? shared_cap(e, e->opt.walk_blocks, 2, NB <= 5 ? 160 : 512, !R.offsets) : 0);
    hipLaunchKernelGGL((k_correct_wave<NB, NN>), dim3(blocks), dim3(256), 0, e->cur, R, e->K, e->filt[1].dev(), list,
                       (const unsigned long long *)e->cur_cnt, tmask, tw, err_bits, patch, e->cur_cnt, ticket);
    HIP_TRY(hipGetLastError());
    return KBBQ_OK;
}
extern "C" {

// coarse base -> read index of a ragged batch (kernels.h: k_read_index), in scratch slot `slot`; null for uniform batches
static int build_read_index(kbbq_engine *e, const ReadsDev &R, int slot, hipStream_t stream, const uint32_t **index) {
    *index = nullptr;
    if (!R.offsets || !R.n_reads) return KBBQ_OK;
    const size_t entries = R.n_bases / READ_INDEX_STEP + 2;
    int rc = ensure_scratch(e, slot, entries * 4);
    if (rc) return rc;
    hipLaunchKernelGGL(k_read_index, dim3((unsigned)((R.n_reads + 255) / 256)), dim3(256), 0, stream, R.offsets, R.n_reads, (uint32_t *)e->scratch[slot]);
    HIP_TRY(hipGetLastError());
    *index = (const uint32_t *)e->scratch[slot];
    return KBBQ_OK;
}

static int run_tally(kbbq_engine *e, const ReadsDev &R, const uint32_t *err_bits, const uint32_t *patch, int max_len,
                     hipStream_t stream = nullptr) {
    if (!stream) stream = e->cur;
    HistDev H;
    H.cycle = e->d_hist;
    H.dinuc = e->d_hist + e->hist_cycle_words;
    H.n_rg = e->p.n_rg;
    H.n_cycle = e->p.max_read_len;
    // LDS tables: ccap cycles x the quality values the batches hold x as many read groups as fit (one launch per set of
    // read groups and per window of ccap cycles; a launch whose read groups do not occur in the batch returns at once)
    const int n_rg = R.rg ? e->p.n_rg : 1;
    const int ccap = std::min(((max_len + 31) / 32) * 32, 192);
    if (!e->qpresent_known) {
        // pass 3 on its own (--fixed mode, tests): no pass 2 has said which quality values occur -- this batch's are
        // added to the set of the batches before it (a wait per batch; not the path of a normal run)
        hipLaunchKernelGGL(k_qpresence, dim3(1024), dim3(256), 0, stream, R.qual, R.n_bases, e->d_qpresent);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(e->qpresent, e->d_qpresent, 32, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    // Reads of many thousand bases: a launch per window of ccap cycles would read the batch max_len / ccap times over, and a
    // long read's cycle counters are as many addresses as it has bases anyway -- one launch, cycle counts straight to the
    // histograms, only the dinucleotide counters in LDS (TallyPlan::direct_cycles).
    const int n_windows = (max_len + ccap - 1) / ccap;
    const bool direct_cycles = n_windows > kTallyMaxWindows;
    // per quality slot: totals and errors of 2 x ccap cycles as 16-bit counters + 2 x 16 dinucleotide words
    const size_t per_slot = (direct_cycles ? 0 : (size_t)8 * ccap) + 128;
    TallyPlan P;
    plan_tally_slots(P, e->qpresent, (int)((152 * 1024 - 512) / per_slot), n_rg);
    P.direct_cycles = direct_cycles ? 1 : 0;
    const size_t per_rg = (size_t)P.n_slots * per_slot;
    // (half of the LDS if everything fits in it: two blocks per CU)
    const size_t budget = (size_t)n_rg * per_rg <= 70 * 1024 ? 70 * 1024 : 140 * 1024;
    const int per_launch = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_rg, budget / per_rg));
    const size_t lds = (size_t)per_launch * per_rg + 4 + 256 + 4;
    if (lds > e->attr_lds_tally) {
        HIP_TRY(hipFuncSetAttribute((const void *)k_tally, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void *)k_tally_uniform<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void *)k_tally_uniform<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        e->attr_lds_tally = lds;
    }
    // 16 wavefronts share one set of LDS tables: one 1024-lane block per CU, two when the tables leave room
    const uint64_t groups = (R.n_bases + 15) / 16;
    const int blocks = (int)std::min<uint64_t>((groups + 1023) / 1024, lds <= 76 * 1024 ? 512 : 256);
    const int vec_ok = ((uintptr_t)R.qual & 15) == 0;
    const uint32_t *read_index;
    int rc = build_read_index(e, R, stream == e->stream2 ? 15 : 14, stream, &read_index);
    if (rc) return rc;
    const uint32_t *present = nullptr;
    if (n_rg > per_launch) {     // several launches: mark the read groups that occur, so that the others cost nothing
        uint32_t *pr = e->d_rg_present[stream == e->stream2 ? 1 : 0];
        HIP_TRY(hipMemsetAsync(pr, 0, (((size_t)e->p.n_rg + 31) / 32) * 4, stream));
        hipLaunchKernelGGL(k_rg_presence, dim3((unsigned)((R.n_reads + 255) / 256)), dim3(256), 0, stream, R.rg, R.n_reads, (uint32_t)e->p.n_rg, pr);
        present = pr;
    }
    // the common shape -- equally long reads, one read group, every cycle in one table -- has a kernel of its own
    if (!e->opt.tally_general && !R.offsets && n_rg == 1 && n_windows == 1 && R.read_len >= 16 && (int)R.read_len <= ccap && R.n_bases < (1ULL << 32) && vec_ok &&
        e->p.n_rg == 1) {
        const unsigned long long inv_len = ~0ULL / R.read_len + 1;      // ceil(2^64 / read_len): read_len is no power of two times... exact below
        Timed t(e, "k_tally", stream);
        if (P.identity) hipLaunchKernelGGL(k_tally_uniform<false>, dim3(blocks), dim3(1024), lds, stream, R, H, err_bits, patch, ccap, 6, inv_len, P);
        else hipLaunchKernelGGL(k_tally_uniform<true>, dim3(blocks), dim3(1024), lds, stream, R, H, err_bits, patch, ccap, 6, inv_len, P);
        HIP_TRY(hipGetLastError());
        return KBBQ_OK;
    }
    Timed t(e, "k_tally", stream);
    for (int g = 0; g < n_rg; g += per_launch)
        for (int w = 0; w < (direct_cycles ? 1 : n_windows); ++w) {
            P.rg_base = g;
            P.n_rgs = std::min(per_launch, n_rg - g);
            P.cbase = w * ccap;
            hipLaunchKernelGGL(k_tally, dim3(blocks), dim3(1024), lds, stream, R, H, err_bits, patch, ccap, 6, vec_ok, P, present, read_index);
        }
    HIP_TRY(hipGetLastError());
    return KBBQ_OK;
}

int kbbq_errors_batch(kbbq_engine *e, const kbbq_reads *reads, uint64_t *errors_out) {
    ENGINE_DEVICE(e);
    { int frc = bucket_flush_all(e); if (frc) return frc; }      // deferred inserts reach the filters first (bucket.h)
    if (!e || !reads) return fail(KBBQ_EINVAL, "null argument");
    const bool own_err = !(errors_out && reads->on_device);
    // A batch whose flags stay inside the engine may still be in flight while the next one is submitted: such
    // batches alternate between the two sides (a device-resident batch stays put by contract, a host batch sits in
    // a staging slot that is not reused before its kernels have finished).  A call that returns the flags runs
    // alone, in order: the caller's array is the caller's again on return.
    const bool overlap = own_err && !(errors_out && !reads->on_device) && !e->opt.no_overlap;
    int side = 0, rc;
    if (overlap) {
        side = e->side_turn;
        e->side_turn ^= 1;
        // the batch that last used this side's scratch may still be in its walk on the side stream: this batch's scan
        // (engine's stream) overwrites that scratch, so the stream waits -- the host does not
        if (e->side_busy[side]) HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_side[side], 0));
    } else {
        if ((rc = sync_engine(e))) return rc;              // both sides' scratch is free
    }
    e->pass3_shared = overlap;
    HostBatchDone host_done(e, reads);
    ReadsDev R; int max_len;
    if ((rc = device_view(e, reads, &R, &max_len))) return rc;
    const bool long_reads = max_len > kStagedMax;
    // words of trusted mask per read: the staged kernels' NW; long reads: 8 per window of 512 k-mer starts
    const int NW = max_len <= 192 ? 3 : max_len <= 320 ? 5 : !long_reads ? 8 : 8 * ((std::max(1, max_len - e->p.k + 1) + 511) / 512);
    // scratch per side: trusted masks, dirty flags, work list, error bits, seq patches
    const int s_tmask = side ? 10 : 3, s_dirty = side ? 11 : 4, s_list = side ? 12 : 5, s_err = side ? 8 : 6, s_patch = side ? 9 : 7;
    if ((rc = ensure_scratch(e, s_tmask, R.n_reads * NW * 8))) return rc;
    if ((rc = ensure_scratch(e, s_dirty, R.n_reads))) return rc;
    if ((rc = ensure_scratch(e, s_list, R.n_reads * 4))) return rc;
    if ((rc = ensure_scratch(e, s_patch, R.n_reads * 4))) return rc;
    uint32_t *d_err;
    if (own_err) {
        if ((rc = ensure_scratch(e, s_err, (R.n_bases / 64 + 2) * 8))) return rc;
        d_err = (uint32_t *)e->scratch[s_err];
    } else {
        d_err = (uint32_t *)errors_out;
    }
    e->cur = e->stream;
    e->cur_cnt = e->d_counters + 4 * side;
    struct Restore { kbbq_engine *e; ~Restore() { e->cur = e->stream; e->cur_cnt = e->d_counters; } } restore = {e};
    HIP_TRY(hipMemsetAsync(d_err, 0, (R.n_bases / 64 + 1) * 8, e->cur));
    HIP_TRY(hipMemsetAsync(e->scratch[s_patch], 0, R.n_reads * 4, e->cur));
    HIP_TRY(hipMemsetAsync(e->cur_cnt, 0, 16, e->cur));
    uint64_t *tmask = (uint64_t *)e->scratch[s_tmask];
    uint8_t *dirty = (uint8_t *)e->scratch[s_dirty];
    uint32_t *list = (uint32_t *)e->scratch[s_list];
    uint32_t *patch = (uint32_t *)e->scratch[s_patch];
    // isolated single errors are settled inside the scan (fast_path); the walk gets what is left (dirty == 1)
    const bool no_fast = e->opt.no_fastpath;
    if (long_reads) {
        Timed t(e, "k_scan_trusted_long", e->cur);
        hipLaunchKernelGGL(k_scan_trusted_long, dim3(wave_grid(R.n_reads)), dim3(256), 0, e->cur, R, e->K, e->filt[1].dev(), tmask, NW, dirty);
        HIP_TRY(hipGetLastError());
    } else if ((rc = dispatch_nw<LaunchScan>(max_len, e, R, tmask, dirty, d_err, (!no_fast && e->p.k >= 3) ? 1 : 0, max_len))) return rc;
    {
        Timed t(e, "k_compact", e->cur);
        // (long reads: every read that is not clean takes the run-time-sized lane-form walk, off-case or not)
        hipLaunchKernelGGL(k_compact, dim3((unsigned)((R.n_reads + 1023) / 1024)), dim3(1024), 0, e->cur, dirty, R.n_reads, list, e->cur_cnt,
                           long_reads ? 0 : 1);
        HIP_TRY(hipGetLastError());
    }
    // The scan and the fast path of every batch run on the engine's stream; the walk and the tally of a
    // device-resident batch move to the side stream, where they overlap the next batch's scan.
    if (overlap) {
        HIP_TRY(hipEventRecord(e->ev_main, e->stream));
        HIP_TRY(hipStreamWaitEvent(e->stream2, e->ev_main, 0));
        e->cur = e->stream2;
        e->cur_slot_side = true;      // (a host batch: its staging slot is read on the side stream as well)
    }
    // one read per wavefront (correct_wave.h); the one-read-per-lane form (correct.h) serves k < 3
    // and KBBQ_CORRECT=lane (A/B checks)
    const bool lane_form = e->opt.lane_walk;
    if (long_reads) {
        rc = launch_correct<0, 64>(e, R, list, e->cur_cnt, tmask, NW, d_err, patch, max_len, side);
    } else if (lane_form || e->p.k < 3) {
        if (max_len <= 160) rc = launch_correct<160, 256>(e, R, list, e->cur_cnt, tmask, NW, d_err, patch);
        else if (max_len <= 320) rc = launch_correct<320, 128>(e, R, list, e->cur_cnt, tmask, NW, d_err, patch);
        else rc = launch_correct<512, 64>(e, R, list, e->cur_cnt, tmask, NW, d_err, patch);
    } else {
        if (max_len <= 160) rc = launch_correct_wave<5, 3>(e, R, list, tmask, NW, d_err, patch);
        else if (max_len <= 320) rc = launch_correct_wave<10, 5>(e, R, list, tmask, NW, d_err, patch);
        else rc = launch_correct_wave<16, 8>(e, R, list, tmask, NW, d_err, patch);
    }
    if (rc) return rc;
    if (R.offcase && !long_reads) {
        // reads with off-case bases (scan state 3) follow the reference's raw-character comparisons: the one-read-per-lane
        // walk carries the case bits (correct.h); a second, usually empty, work list
        if ((rc = ensure_scratch(e, side ? 18 : 17, R.n_reads * 4))) return rc;
        uint32_t *list3 = (uint32_t *)e->scratch[side ? 18 : 17];
        unsigned long long *cnt3 = e->d_counters + 6 + side;
        HIP_TRY(hipMemsetAsync(cnt3, 0, 8, e->cur));
        hipLaunchKernelGGL(k_compact, dim3((unsigned)((R.n_reads + 1023) / 1024)), dim3(1024), 0, e->cur, dirty, R.n_reads, list3, cnt3, 3);
        HIP_TRY(hipGetLastError());
        if (max_len <= 160) rc = launch_correct<160, 256>(e, R, list3, cnt3, tmask, NW, d_err, patch);
        else if (max_len <= 320) rc = launch_correct<320, 128>(e, R, list3, cnt3, tmask, NW, d_err, patch);
        else rc = launch_correct<512, 64>(e, R, list3, cnt3, tmask, NW, d_err, patch);
        if (rc) return rc;
    }
    if ((rc = run_tally(e, R, d_err, patch, max_len, e->cur))) return rc;
    if (!(R.offcase && !long_reads)) HIP_TRY(hipMemsetAsync(e->d_counters + 6 + side, 0, 8, e->cur));
    hipLaunchKernelGGL(k_add_counters, dim3(1), dim3(1), 0, e->cur, (const unsigned long long *)(e->d_counters + 4 * side),
                       (const unsigned long long *)(e->d_counters + 6 + side), e->d_totals);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e->ev_side[side], e->cur));
    e->side_busy[side] = true;
    e->stats[2] += R.n_reads;
    if (!overlap) {
        if ((rc = sync_engine(e))) return rc;
        if (errors_out && !reads->on_device)
            HIP_TRY(hipMemcpy(errors_out, d_err, (R.n_bases / 64 + 1) * 8, hipMemcpyDeviceToHost));
    }
    return KBBQ_OK;
}

int kbbq_tally_batch(kbbq_engine *e, const kbbq_reads *reads, const uint64_t *errors) {
    ENGINE_DEVICE(e);
    { int frc = bucket_flush_all(e); if (frc) return frc; }      // deferred inserts reach the filters first (bucket.h)
    if (!e || !errors) return fail(KBBQ_EINVAL, "null argument");
    HostBatchDone host_done(e, reads);
    ReadsDev R; int max_len;
    int rc = device_view(e, reads, &R, &max_len);
    if (rc) return rc;
    const uint32_t *d_err = (const uint32_t *)errors;
    if (!reads->on_device) {      // --fixed mode with host batches: the caller's flags go through a scratch array
        const size_t bytes = (reads->n_bases / 64 + 1) * 8;
        if ((rc = ensure_scratch(e, 16, bytes + 8))) return rc;
        HIP_TRY(hipMemsetAsync((char *)e->scratch[16] + bytes, 0, 8, e->stream));
        HIP_TRY(hipMemcpyAsync(e->scratch[16], errors, bytes, hipMemcpyHostToDevice, e->stream));
        d_err = (const uint32_t *)e->scratch[16];
    }
    if ((rc = run_tally(e, R, d_err, nullptr, max_len))) return rc;
    return reads->on_device ? KBBQ_OK : sync_engine(e);      // the caller's flag array is free again on return
}

void *kbbq_covariates_device(kbbq_engine *e, uint64_t *n_words) {
    if (!e) return nullptr;
    if (n_words) *n_words = e->hist_cycle_words + e->hist_dinuc_words;
    return e->d_hist;
}

static int fetch_hist(kbbq_engine *e, std::vector<uint64_t> &cyc, std::vector<uint64_t> &di) {
    int rc = sync_engine(e);
    if (rc) return rc;
    cyc.resize(e->hist_cycle_words);
    di.resize(e->hist_dinuc_words);
    HIP_TRY(hipMemcpy(cyc.data(), e->d_hist, e->hist_cycle_words * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(di.data(), e->d_hist + e->hist_cycle_words, e->hist_dinuc_words * 8, hipMemcpyDeviceToHost));
    return KBBQ_OK;
}

int kbbq_covariates_get(kbbq_engine *e, kbbq_covariates *out) {
    ENGINE_DEVICE(e);
    if (!e || !out) return fail(KBBQ_EINVAL, "null argument");
    std::vector<uint64_t> cyc, di, q, rg;
    int rc = fetch_hist(e, cyc, di);
    if (rc) return rc;
    derive_q_rg(e->p.n_rg, e->p.max_read_len, cyc.data(), q, rg);
    out->n_rg = e->p.n_rg;
    out->n_cycle = e->p.max_read_len;
    if (out->rg) memcpy(out->rg, rg.data(), rg.size() * 8);
    if (out->q) memcpy(out->q, q.data(), q.size() * 8);
    if (out->cycle) memcpy(out->cycle, cyc.data(), cyc.size() * 8);
    if (out->dinuc) memcpy(out->dinuc, di.data(), di.size() * 8);
    return KBBQ_OK;
}

int kbbq_train(kbbq_engine *e) {
    ENGINE_DEVICE(e);
    if (!e) return fail(KBBQ_EINVAL, "null engine");
    std::vector<uint64_t> cyc, di, q, rg;
    int rc = fetch_hist(e, cyc, di);
    if (rc) return rc;
    derive_q_rg(e->p.n_rg, e->p.max_read_len, cyc.data(), q, rg);
    e->dq = train_model(e->p.n_rg, e->p.max_read_len, rg.data(), q.data(), cyc.data(), di.data());
    return upload_dq(e);
}

int kbbq_dq_get(kbbq_engine *e, kbbq_dq *out) {
    ENGINE_DEVICE(e);
    if (!e || !out) return fail(KBBQ_EINVAL, "null argument");
    if (!e->dq_set) return fail(KBBQ_ESTATE, "no delta-Q tables yet");
    const DqTables &d = e->dq;
    out->n_rg = d.n_rg;
    out->n_cycle = d.n_cycle;
    if (out->meanq) memcpy(out->meanq, d.meanq.data(), d.meanq.size() * 4);
    if (out->rgdq) memcpy(out->rgdq, d.rgdq.data(), d.rgdq.size() * 4);
    if (out->qdq) memcpy(out->qdq, d.qdq.data(), d.qdq.size() * 4);
    if (out->cycledq) memcpy(out->cycledq, d.cycledq.data(), d.cycledq.size() * 4);
    if (out->dinucdq) memcpy(out->dinucdq, d.dinucdq.data(), d.dinucdq.size() * 4);
    return KBBQ_OK;
}

int kbbq_set_dq(kbbq_engine *e, const kbbq_dq *in) {
    ENGINE_DEVICE(e);
    if (!e || !in || !in->meanq || !in->rgdq || !in->qdq || !in->cycledq || !in->dinucdq) return fail(KBBQ_EINVAL, "null argument");
    if (in->n_rg != (uint64_t)e->p.n_rg || in->n_cycle != (uint64_t)e->p.max_read_len)
        return fail(KBBQ_EINVAL, "delta-Q tables must be [%d rg][%d cycles]", e->p.n_rg, e->p.max_read_len);
    DqTables &d = e->dq;
    d.n_rg = in->n_rg;
    d.n_cycle = in->n_cycle;
    d.meanq.assign(in->meanq, in->meanq + d.n_rg);
    d.rgdq.assign(in->rgdq, in->rgdq + d.n_rg);
    d.qdq.assign(in->qdq, in->qdq + d.n_rg * kNQ);
    d.cycledq.assign(in->cycledq, in->cycledq + d.n_rg * kNQ * 2 * d.n_cycle);
    d.dinucdq.assign(in->dinucdq, in->dinucdq + d.n_rg * kNQ * 16);
    return upload_dq(e);
}

// ---- pass 4
static int recalibrate_impl(kbbq_engine *e, const kbbq_reads *reads, uint8_t *qual_out, bool out_on_host);

int kbbq_recalibrate_batch(kbbq_engine *e, const kbbq_reads *reads, uint8_t *qual_out) {
    ENGINE_DEVICE(e);
    if (!reads) return fail(KBBQ_EINVAL, "null argument");
    return recalibrate_impl(e, reads, qual_out, !reads->on_device);
}

int kbbq_recalibrate_batch_host(kbbq_engine *e, const kbbq_reads *reads, uint8_t *host_qual_out) {
    ENGINE_DEVICE(e);
    return recalibrate_impl(e, reads, host_qual_out, true);
}

static int recalibrate_impl(kbbq_engine *e, const kbbq_reads *reads, uint8_t *qual_out, bool out_on_host) {
    ENGINE_DEVICE(e);
    if (!e || !qual_out || !reads) return fail(KBBQ_EINVAL, "null argument");
    if (!e->dq_set) return fail(KBBQ_ESTATE, "no delta-Q tables yet");
    HostBatchDone host_done(e, reads);
    ReadsDev R; int max_len;
    // a large host batch with its result in host memory goes through in pieces (below): 64 Ki-base multiples, at least 2^23
    // bases each, about four per batch
    const bool no_pipe = e->opt.no_pass4_pipeline || e->opt.no_overlap;
    const uint64_t piece_env = e->opt.pass4_piece;      // tests: pieces of that many bases (rounded up to 64)
    const uint64_t piece = piece_env ? ((piece_env + 63) >> 6) << 6 : std::max<uint64_t>(1ull << 23, ((reads->n_bases / 4 + 65535) >> 16) << 16);
    const bool pipelined = !reads->on_device && out_on_host && !no_pipe && reads->n_bases >= 2 * piece;
    DeferredCopy dc;
    // the asynchronous form (kbbq_recalibrate_batch_submit) of a host batch keeps its result in the batch's staging slot
    // until the copy back has landed: two such batches may be in flight
    const bool deferred_out = e->async_call && !reads->on_device && out_on_host;
    SlotOutput so;
    so.bytes = reads->n_bases + 16;
    int rc = device_view(e, reads, &R, &max_len, true, pipelined ? &dc : nullptr, deferred_out ? &so : nullptr);
    if (rc) return rc;
    uint8_t *d_out = qual_out;
    if (deferred_out) {
        d_out = (uint8_t *)so.dev;
    } else if (out_on_host) {
        if ((rc = ensure_scratch(e, 2, R.n_bases + 16))) return rc;
        d_out = (uint8_t *)e->scratch[2];
    }
    DqDev D;
    D.base = e->d_dq_base; D.cycle = e->d_dq_cycle; D.dinuc = e->d_dq_dinuc;
    D.qslot = e->d_dq_qslot;
    D.n_rg = e->p.n_rg; D.n_cycle = e->p.max_read_len; D.n_slots = e->dq_slots;
    const uint32_t *read_index;
    if ((rc = build_read_index(e, R, 14, e->stream, &read_index))) return rc;
    const int per_rg = (D.n_slots * (4 * D.n_cycle + 16) + KBBQ_NQ * 2 + 3) & ~3;
    // tables of as many read groups as fit (kernels.h: compacted over the quality axis): two 1024-lane blocks
    // per CU share the 160 KB of LDS when one group takes at most 64 KB; otherwise one block per CU and up to 152 KB.
    // (More than 255 quality values with a delta of their own -- no real data -- leave the slot map: global tables.)
    const int budget = per_rg + 256 <= 64 * 1024 ? 64 * 1024 - 256 : 152 * 1024 - 256;
    const int lds_rgs = D.n_slots > 255 ? 0 : std::max(0, std::min(D.n_rg, budget / per_rg));
    const size_t lds = 256 + (size_t)lds_rgs * per_rg;
    if (lds > e->attr_lds_recal) {
        HIP_TRY(hipFuncSetAttribute((const void *)k_recalibrate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        e->attr_lds_recal = lds;
    }
    const int vec_ok = (((uintptr_t)R.qual | (uintptr_t)d_out) & 15) == 0;
    auto launch = [&](uint64_t base0, uint64_t base1) -> int {      // the bases [base0, base1), base0 a multiple of 16
        Timed t(e, "k_recalibrate");
        const uint64_t lanes = (base1 - base0 + 15) / 16;
        const unsigned blocks = (unsigned)std::min<uint64_t>((lanes + 1023) / 1024, lds > 76 * 1024 ? 256 : 256 * 2);
        hipLaunchKernelGGL(k_recalibrate, dim3(blocks), dim3(1024), lds, e->stream,
                           R, D, d_out, 6, vec_ok, lds_rgs, read_index, base0, base1);
        HIP_TRY(hipGetLastError());
        return KBBQ_OK;
    };
    if (pipelined) {
        // piece i's copy in, piece i-1's kernel and piece i-2's copy out at the same time: host -> device on the copy
        // stream, the kernel on the engine's, device -> host on the side stream (the link carries both directions at
        // once); the kernel reads the base before its first one, which an earlier piece brought
        const uint64_t nb = R.n_bases;
        struct SideDone {      // the caller's output buffer is its own again on every return path (synchronous form)
            hipStream_t st;
            bool wait;
            ~SideDone() { if (wait) (void)hipStreamSynchronize(st); }
        } side_done{e->stream2, !deferred_out};
        for (uint64_t b0 = 0; b0 < nb; b0 += piece) {
            const uint64_t b1 = std::min(nb, b0 + piece);
            const uint64_t w0 = b0 / 32, w1 = b1 == nb ? nb / 32 + 1 : b1 / 32, m0 = b0 / 64, m1 = b1 == nb ? nb / 64 + 1 : b1 / 64;
            HIP_TRY(hipMemcpyAsync(dc.d_bases + w0 * 8, reads->bases + w0, (w1 - w0) * 8, hipMemcpyHostToDevice, e->copy));
            HIP_TRY(hipMemcpyAsync(dc.d_nmask + m0 * 8, reads->nmask + m0, (m1 - m0) * 8, hipMemcpyHostToDevice, e->copy));
            HIP_TRY(hipMemcpyAsync(dc.d_qual + b0, reads->qual + b0, b1 - b0, hipMemcpyHostToDevice, e->copy));
            HIP_TRY(hipEventRecord(dc.slot->h2d, e->copy));      // (the guard waits for the last of them: the whole batch has left the caller's memory)
            HIP_TRY(hipStreamWaitEvent(e->stream, dc.slot->h2d, 0));
            if ((rc = launch(b0, b1))) return rc;
            HIP_TRY(hipEventRecord(e->ev_main, e->stream));
            HIP_TRY(hipStreamWaitEvent(e->stream2, e->ev_main, 0));
            HIP_TRY(hipMemcpyAsync(qual_out + b0, d_out + b0, b1 - b0, hipMemcpyDeviceToHost, e->stream2));
        }
        if (deferred_out) {
            // kbbq_batch_wait waits for this; the slot is not handed out again before it either (acquire_slot)
            HIP_TRY(hipEventRecord(so.slot->d2h, e->stream2));
            so.slot->d2h_pending = true;
            e->cur_slot_side = true;
            return KBBQ_OK;
        }
        HIP_TRY(hipStreamSynchronize(e->stream2));      // (reports a failed copy; the guard's wait is then a no-op)
        return KBBQ_OK;
    }
    if ((rc = launch(0, R.n_bases))) return rc;
    if (out_on_host) {
        HIP_TRY(hipMemcpyAsync(qual_out, d_out, R.n_bases, hipMemcpyDeviceToHost, e->stream));
        if (deferred_out) {
            HIP_TRY(hipEventRecord(so.slot->d2h, e->stream));
            so.slot->d2h_pending = true;
            return KBBQ_OK;
        }
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    return KBBQ_OK;
}

// ---- the asynchronous form of the batch entry points ------------------------------------------------------------------
// The synchronous calls return when their host batch has left the caller's memory, so the next batch's copy is queued
// only after the previous one has landed and every pass has a bubble per batch (0.64-0.77 of the link's bound in round
// 2).  Here the call only queues -- copy, kernels, copy back -- and the caller, with two or three page-locked batches in
// flight, asks for a batch's memory back when it needs it.
namespace {
struct AsyncCall {
    kbbq_engine *e;
    explicit AsyncCall(kbbq_engine *e_) : e(e_) { e->async_call = true; e->last_ticket = 0; }
    ~AsyncCall() { e->async_call = false; }
    // What the caller gets: the ticket of THIS call, or 0 when the call failed -- a failure before HostBatchDone's
    // constructor would otherwise hand back the previous call's ticket.  A failed call may have queued copies out of
    // or into the caller's memory (pass 4 in pieces: copies back without their event): they are drained here, so that
    // after an error nothing of the engine touches the caller's batch any more.
    uint64_t finish(int rc) {
        if (rc == KBBQ_OK) return e->last_ticket;
        (void)hipStreamSynchronize(e->copy);
        (void)hipStreamSynchronize(e->stream);
        (void)hipStreamSynchronize(e->stream2);
        e->last_ticket = 0;
        return 0;
    }
};
}  // namespace

int kbbq_sample_batch_submit(kbbq_engine *e, const kbbq_reads *reads, uint64_t first_kmer_ordinal, kbbq_ticket *ticket) {
    if (!e || !ticket) return fail(KBBQ_EINVAL, "null argument");
    AsyncCall a(e);
    const int rc = kbbq_sample_batch(e, reads, first_kmer_ordinal);
    *ticket = a.finish(rc);
    return rc;
}

int kbbq_trusted_batch_submit(kbbq_engine *e, const kbbq_reads *reads, kbbq_ticket *ticket) {
    if (!e || !ticket) return fail(KBBQ_EINVAL, "null argument");
    AsyncCall a(e);
    const int rc = kbbq_trusted_batch(e, reads, nullptr);
    *ticket = a.finish(rc);
    return rc;
}

int kbbq_errors_batch_submit(kbbq_engine *e, const kbbq_reads *reads, kbbq_ticket *ticket) {
    if (!e || !ticket) return fail(KBBQ_EINVAL, "null argument");
    AsyncCall a(e);
    const int rc = kbbq_errors_batch(e, reads, nullptr);
    *ticket = a.finish(rc);
    return rc;
}

int kbbq_recalibrate_batch_submit(kbbq_engine *e, const kbbq_reads *reads, uint8_t *qual_out, kbbq_ticket *ticket) {
    if (!e || !ticket) return fail(KBBQ_EINVAL, "null argument");
    AsyncCall a(e);
    const int rc = kbbq_recalibrate_batch(e, reads, qual_out);
    *ticket = a.finish(rc);
    return rc;
}

int kbbq_batch_wait(kbbq_engine *e, kbbq_ticket ticket) {
    ENGINE_DEVICE(e);
    if (!ticket) return KBBQ_OK;      // a device batch, or a call that had waited itself
    const int idx = (int)(ticket & 0xFF) - 1;
    if (idx < 0 || idx > 2) return fail(KBBQ_EINVAL, "not a ticket of this engine");
    kbbq_engine::StageSlot &s = e->slot[idx];
    if (s.gen != (uint32_t)(ticket >> 8)) return KBBQ_OK;      // the slot has been handed out again since: that waited for all of it
    HIP_TRY(hipEventSynchronize(s.h2d));
    if (s.d2h_pending) { HIP_TRY(hipEventSynchronize(s.d2h)); s.d2h_pending = false; }
    return KBBQ_OK;
}

// ---- synthetic data
int kbbq_synth_tables(const kbbq_synth_params *sp, uint32_t *qcum /* [read_len][4] */, uint32_t *errthr /* [94] */) {
    if (!sp || !qcum || !errthr) return fail(KBBQ_EINVAL, "null argument");
    // Quality profile: a few discrete values whose low-quality share grows
    // quadratically along the read; substitution probability 10^(-q/10).
    const uint32_t L = sp->read_len;
    for (uint32_t c = 0; c < L; ++c) {
        const double f = L > 1 ? (double)c / (double)(L - 1) : 0.0;
        const double w2 = 0.002 + 0.010 * f * f, w12 = 0.010 + 0.080 * f * f, w22 = 0.030 + 0.120 * f * f,
                     w32 = 0.100 + 0.100 * f;
        const double cum[4] = {w2, w2 + w12, w2 + w12 + w22, w2 + w12 + w22 + w32};
        for (int j = 0; j < 4; ++j) qcum[4 * c + j] = (uint32_t)(cum[j] * 4294967296.0);
    }
    for (int q = 0; q < 94; ++q) {
        const double p = pow(10.0, -q / 10.0);
        errthr[q] = p >= 1.0 ? 0xFFFFFFFFu : (uint32_t)(p * 4294967296.0);
    }
    return KBBQ_OK;
}

int kbbq_synth_reads(kbbq_engine *e, const kbbq_synth_params *sp, uint64_t first_read, uint64_t n, kbbq_reads *dev) {
    ENGINE_DEVICE(e);
    if (!e || !sp || !dev || n == 0) return fail(KBBQ_EINVAL, "bad argument");
    if (sp->read_len == 0 || sp->genome_len < sp->read_len || sp->n_rg == 0) return fail(KBBQ_EINVAL, "bad synthetic parameters");
    if (e->qcum_len != sp->read_len) {
        std::vector<uint32_t> qc(4 * (size_t)sp->read_len), et(94);
        kbbq_synth_tables(sp, qc.data(), et.data());
        hipFree(e->d_qcum); hipFree(e->d_errthr);
        e->d_qcum = e->d_errthr = nullptr;
        HIP_TRY(hipMalloc(&e->d_qcum, qc.size() * 4));
        HIP_TRY(hipMalloc(&e->d_errthr, 94 * 4));
        HIP_TRY(hipMemcpy(e->d_qcum, qc.data(), qc.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->d_errthr, et.data(), 94 * 4, hipMemcpyHostToDevice));
        e->qcum_len = sp->read_len;
    }
    const uint64_t nb = n * sp->read_len;
    memset(dev, 0, sizeof *dev);
    dev->n_reads = n; dev->n_bases = nb; dev->read_len = sp->read_len; dev->on_device = 1;
    void *b = nullptr, *m = nullptr, *q = nullptr, *f = nullptr, *g = nullptr;
    HIP_TRY(hipMalloc(&b, (nb / 32 + 2) * 8));
    HIP_TRY(hipMalloc(&m, (nb / 64 + 2) * 8));
    HIP_TRY(hipMalloc(&q, nb + 16));
    HIP_TRY(hipMalloc(&f, n));
    HIP_TRY(hipMalloc(&g, n * 2));
    HIP_TRY(hipMemsetAsync(b, 0, (nb / 32 + 2) * 8, e->stream));
    HIP_TRY(hipMemsetAsync(m, 0, (nb / 64 + 2) * 8, e->stream));
    HIP_TRY(hipMemsetAsync((char *)q + nb, 0, 16, e->stream));
    dev->bases = (const uint64_t *)b; dev->nmask = (const uint64_t *)m; dev->qual = (const uint8_t *)q;
    dev->flags = (const uint8_t *)f; dev->rg = (const uint16_t *)g;
    SynthDev S;
    S.seed = sp->seed; S.genome_len = sp->genome_len; S.first_read = first_read; S.n_reads = n;
    S.read_len = sp->read_len; S.n_rg = sp->n_rg; S.paired = sp->paired;
    S.n_thr = (uint32_t)(((uint64_t)sp->n_per_million << 20) / 1000000ULL);
    S.qcum = e->d_qcum; S.errthr = e->d_errthr;
    const uint64_t words = (nb + 31) / 32;
    hipLaunchKernelGGL(k_synth, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, e->stream, S, (uint64_t *)b,
                       (uint64_t *)m, (uint8_t *)q, (uint8_t *)f, (uint16_t *)g);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
    return KBBQ_OK;
}

// ---- measurement
int kbbq_profile_get(kbbq_engine *e, kbbq_profile_entry *out, int32_t max_entries, int32_t *n_out) {
    ENGINE_DEVICE(e);
    if (!e || !n_out) return fail(KBBQ_EINVAL, "null argument");
    int rc = sync_engine(e);
    if (rc) return rc;
    int n = 0;
    for (size_t i = 0; i < e->prof.size() && n < max_entries; ++i, ++n) {
        if (!out) continue;
        snprintf(out[n].name, sizeof out[n].name, "%s", e->prof[i].name.c_str());
        out[n].launches = e->prof[i].launches;
        out[n].total_ms = e->prof[i].ms;
    }
    *n_out = (int)e->prof.size();
    return KBBQ_OK;
}

int kbbq_profile_reset(kbbq_engine *e) {
    ENGINE_DEVICE(e);
    if (!e) return fail(KBBQ_EINVAL, "null engine");
    int rc = sync_engine(e);
    if (rc) return rc;
    for (size_t i = 0; i < e->prof.size(); ++i) { e->prof[i].launches = 0; e->prof[i].ms = 0; }
    return KBBQ_OK;
}

int kbbq_stats_get(kbbq_engine *e, uint64_t *out, int32_t n) {
    ENGINE_DEVICE(e);
    if (!e || !out) return fail(KBBQ_EINVAL, "null argument");
    int rc = sync_engine(e);      // batches of pass 3 may still be in flight; their counters are collected here
    if (rc) return rc;
    HIP_TRY(hipMemcpy(&e->stats[3], e->d_counters + 3, 8, hipMemcpyDeviceToHost));      // [3] Bloom blocks fetched by k_infer
    for (int i = 0; i < n && i < 4; ++i) out[i] = e->stats[i];
    // [4],[5] flushes of the bucketed inserts per filter, [6] records inserted directly because a region was full,
    // [7] records gathered per flush (0: the filters take direct inserts)
    unsigned long long direct = 0;
    if (e->bk.allocated && n > 6) HIP_TRY(hipMemcpy(&direct, e->bk.direct, 8, hipMemcpyDeviceToHost));
    const uint64_t extra[4] = {e->bk.flushes[0], e->bk.flushes[1], direct, e->bk.allocated ? e->bk.capacity : 0};
    for (int i = 4; i < n && i < 8; ++i) out[i] = extra[i - 4];
    if (n > 8) out[8] = 0;      // (rounds 1-2: "a quality above 93 was left out of the model"; every quality is modelled now)
    return KBBQ_OK;
}

// ---- host-only entry points (no GPU touched): the scalar parts of the path ----------
int kbbq_host_filter_spec(uint64_t approx_kmers, double fpr, uint64_t bloom_seed, kbbq_filter_info *info,
                          uint64_t *patterns_out) {
    if (!info) return fail(KBBQ_EINVAL, "null argument");
    FilterSpec s;
    if (!make_filter_spec(approx_kmers, fpr, bloom_seed, s))
        return fail(KBBQ_EINVAL, "Error: Invalid bloom filter parameters. Adjust parameters and try again.");
    memset(info, 0, sizeof *info);
    info->bits = s.bits; info->bits_unblocked = s.bits_unblocked; info->n_blocks = s.n_blocks;
    info->table_bytes = s.n_blocks * kEngineBlockBytes;
    info->random_seed = s.random_seed; info->n_hash = s.n_hash; info->n_salt = s.n_salt;
    for (uint32_t i = 0; i < s.n_salt; ++i) info->salt[i] = s.salt[i];
    if (patterns_out) memcpy(patterns_out, s.patterns.data(), kNumPatterns * 64);
    return KBBQ_OK;
}

uint32_t kbbq_host_block_index(uint32_t hash, uint64_t n_blocks) {
    if (!n_blocks) return 0;
    const ModMagic mm = make_mod_magic(n_blocks);
    return mod_hash(hash, mm.d, mm.m64, mm.m32);
}

int kbbq_host_blocks_squeeze(const uint64_t *reference_words, uint64_t n_blocks, uint64_t *engine_words) {
    if (!reference_words || !engine_words) return fail(KBBQ_EINVAL, "null argument");
    for (uint64_t b = 0; b < n_blocks; ++b)
        if (!squeeze_block(reference_words + b * 8, engine_words + b * 2))
            return fail(KBBQ_ERANGE, "block %llu has a bit no pattern can set", (unsigned long long)b);
    return KBBQ_OK;
}

int kbbq_host_blocks_expand(const uint64_t *engine_words, uint64_t n_blocks, uint64_t *reference_words) {
    if (!reference_words || !engine_words) return fail(KBBQ_EINVAL, "null argument");
    for (uint64_t b = 0; b < n_blocks; ++b) expand_block(engine_words + b * 2, reference_words + b * 8);
    return KBBQ_OK;
}

int kbbq_host_thresholds(int32_t k, uint64_t filter_bits, uint64_t inserted, uint32_t n_salt, const char *alpha_text,
                         int32_t *thresholds_out, double *fpr_out, char *p_text_out, size_t p_text_len) {
    if (!alpha_text || !thresholds_out) return fail(KBBQ_EINVAL, "null argument");
    if (k < 1 || k > KBBQ_MAX_KMER) return fail(KBBQ_ERANGE, "k must be <= %d and > 0", KBBQ_MAX_KMER);
    if (inserted == 0) return fail(KBBQ_ESTATE, "no k-mers were sampled");
    double fpr = 0;
    std::string p_text;
    std::vector<int32_t> t = thresholds_from_counts(k, filter_bits, inserted, n_salt, alpha_text, &fpr, &p_text);
    memcpy(thresholds_out, t.data(), (k + 1) * 4);
    if (fpr_out) *fpr_out = fpr;
    if (p_text_out && p_text_len) snprintf(p_text_out, p_text_len, "%s", p_text.c_str());
    return fpr > .15 ? 1 : 0;
}

int kbbq_host_train(const kbbq_covariates *cov, kbbq_dq *out) {
    if (!cov || !out || !cov->cycle || !cov->dinuc) return fail(KBBQ_EINVAL, "null argument");
    std::vector<uint64_t> q, rg;
    derive_q_rg(cov->n_rg, cov->n_cycle, cov->cycle, q, rg);
    DqTables d = train_model(cov->n_rg, cov->n_cycle, rg.data(), q.data(), cov->cycle, cov->dinuc);
    out->n_rg = d.n_rg; out->n_cycle = d.n_cycle;
    if (out->meanq) memcpy(out->meanq, d.meanq.data(), d.meanq.size() * 4);
    if (out->rgdq) memcpy(out->rgdq, d.rgdq.data(), d.rgdq.size() * 4);
    if (out->qdq) memcpy(out->qdq, d.qdq.data(), d.qdq.size() * 4);
    if (out->cycledq) memcpy(out->cycledq, d.cycledq.data(), d.cycledq.size() * 4);
    if (out->dinucdq) memcpy(out->dinucdq, d.dinucdq.data(), d.dinucdq.size() * 4);
    return KBBQ_OK;
}

uint64_t kbbq_host_bernoulli_threshold(double p, int32_t *always) {
    bool a = false;
    const uint64_t t = bernoulli_threshold(p, &a);
    if (always) *always = a ? 1 : 0;
    return t;
}

int kbbq_rng_state_at(uint32_t seed, uint64_t ordinal, uint64_t state_out[4]) {
    if (!state_out) return fail(KBBQ_EINVAL, "null argument");
    xoshiro_state_at(seed, ordinal, state_out);
    return KBBQ_OK;
}

}  // extern "C"
