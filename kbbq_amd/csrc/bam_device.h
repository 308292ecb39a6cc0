// bam_device.h -- the BAM side of the input and output on the device (gfx950): the alignment records of an inflated BAM
// stream found, decoded into the engine's read layout and -- pass 4 -- rewritten around the new qualities, all in HBM.
// Included once by bgzf_device.hip, behind bgzf_inflate.h.
//
// What it replaces: sam_read1 (BamFile::next, htsiter.cc:5), the BAM constructor of
// CReadData (readutils.cc:13-61; bam_seq_str, readutils.hh:30-42), BamFile::recalibrate (htsiter.cc:11-32) and
// sam_write1 (htsiter.cc:45) -- which the reference runs once per record and pass on the host.  bam_io.cc (BamReader,
// decode_bam_read, BamChunkParser) stays the definition: every shape these kernels do not take is flagged and the
// caller starts over with the host parsers, and tests/test_bam_gpu.py compares the two word for word.
//
//   k_bam_seg_guess      A BAM stream is a chain: [block_size u32][block], the next record starts block_size + 4 bytes on.
//                        The stream is cut into 32 KB segments; one wavefront per segment looks for the first offset in it
//                        where a record could start (64 candidate offsets per step, one per lane: block_size, refID,
//                        l_read_name + the name's NUL, n_cigar_op, l_seq consistent; then the same for the two records
//                        chained behind it).  A guess, nothing more.
//   k_bam_seg_walk       one lane per segment follows the chain from the segment's start to the segment's end: record
//                        offsets into the segment's slots, their count, and where the chain LANDS in a later segment.
//   k_bam_seg_check / k_bam_seg_repair
//                        Segment 0 starts at a known offset (behind the header, or at the record the previous chunk's
//                        end cut).  If every segment's start equals the landing point of the segment before it, the
//                        walks together ARE the chain from that offset, by induction -- no property of the guess is
//                        relied on.  A segment whose start is not that landing point (a wrong guess, no guess, a record
//                        longer than a segment) is walked again from the landing point by one lane.
//   k_bam_records        one lane per record: the fixed fields, where sequence and qualities lie, the RG:Z and OQ:Z tags
//                        with bam_aux_get's rules (first match wins, a malformed tag in front of it is an error), the read
//                        group looked up in the header's @RG table.  What the host parser would report -- a missing or
//                        corrupt tag, an OQ of another length, a read group the header does not name -- raises a flag.
//   k_bam_gather         one wavefront per record: 4-bit codes to sequence text (reverse-strand records complemented and
//                        reversed, every non-ACGT code 'N' there, readutils.hh:35-36), qualities or OQ - 33 (reversed for
//                        reverse-strand records, readutils.cc:36-39) into the batch's arrays; k_pack_text packs the text.
//   k_bam_out_sizes / k_bam_rewrite
//                        pass 4: every record again with the new qualities in its quality field (reversed back,
//                        htsiter.cc:27-31) and -- --set-oq -- the old ones as OQ:Z, replaced in place or appended as
//                        bam_aux_update_str does (htsiter.cc:13-26), block_size patched; the payload goes to k_deflate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kbbq {
namespace dfl {

constexpr uint32_t BAM_SEG = 1u << 15;                  // bytes of stream per segment
constexpr uint32_t BAM_SEG_SLOTS = BAM_SEG / 37 + 2;    // a record is at least 4 + 32 + 1 bytes (l_read_name >= 1)
constexpr uint32_t BAM_NONE = 0xFFFFFFFFu;
constexpr int BAM_CHAIN = 3;                            // records a guess must chain through

// flags of a chunk (kbbq_bam_chunk.flags)
enum : uint32_t {
    BAMF_FALLBACK = 1,        // a shape for the host parsers: a malformed block, a tag the host path reports, too many repairs
    BAMF_TRUNCATED = 4,       // the stream ended inside a record
    BAMF_OQ_UNWRITABLE = 8,   // a record whose OQ tag bam_aux_update_str could not update (--set-oq must take the host path)
};

struct BamSegs {
    uint32_t *start;      // per segment: offset of the first record that starts in it (BAM_NONE: none found)
    uint32_t *land;       //              where its chain ends: the first record start at or behind the segment's end, or the
                          //              start of the record the stream's end cuts
    uint32_t *count;      //              records that start in it
    uint32_t *bad;        //              a malformed block_size met on the way (only meaningful once the start is verified)
    uint32_t *slots;      //              BAM_SEG_SLOTS record offsets
    uint32_t n_segs;
};

// per record (structure of arrays, n_records long)
struct BamIndex {
    uint32_t *rec_off;      // offset of the record's block_size field in the stream
    uint32_t *seq_off;      // of its 4-bit sequence
    uint32_t *qual_off;     // of its quality field
    uint32_t *qsrc_off;     // of the qualities the passes read: the quality field, or the OQ value (text) with use_oq
    uint32_t *l_seq;
    uint32_t *oq_at;        // offset of the OQ tag's type byte (0: no OQ tag)
    uint32_t *oq_vlen;      // length of its value without the NUL
    uint16_t *flag;         // the record's FLAG
    uint16_t *rg;           // index into the header's @RG table
    uint64_t *base_sz;      // l_seq, then -- scanned -- the batch's base offsets (n_records + 1)
    uint64_t *out_sz;       // pass 4: bytes of the rewritten record with its block_size field, then scanned
};

struct BamRgTable {
    const uint8_t *ids;         // the ids back to back
    const uint32_t *id_off;     // n_ids + 1 offsets
    uint32_t n_ids;
    // more than a handful of @RG lines (merged cohorts carry hundreds): an open-addressing table over the ids' FNV-1a
    // hashes, so that a record compares its RG value with one or two ids instead of all of them
    const uint16_t *hash_slots; // hash_mask + 1 entries: id index, 0xFFFF = empty
    uint32_t hash_mask;         // 0: no table (few ids: compared one by one)
};
__host__ __device__ __forceinline__ uint32_t bam_fnv1a(uint32_t h, uint8_t c) { return (h ^ c) * 16777619u; }

// unaligned little-endian loads from the stream (the buffer is readable 4 KB behind its end)
__device__ __forceinline__ uint32_t bam_ld32(const uint8_t *t, uint64_t p) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(t + (p & ~3ull));
    const uint32_t sh = (uint32_t)(p & 3) * 8;
    const uint32_t a = w[0];
    if (!sh) return a;
    return (a >> sh) | (w[1] << (32 - sh));
}
__device__ __forceinline__ uint32_t bam_ld16(const uint8_t *t, uint64_t p) { return (uint32_t)t[p] | ((uint32_t)t[p + 1] << 8); }

// Could a record start at offset p of a stream of n bytes?  Fixed fields in range and consistent with block_size, the read
// name NUL-terminated where l_read_name says.  (The chain decides; this only has to be true for real records.)
__device__ __forceinline__ bool bam_plausible(const uint8_t *t, uint64_t p, uint64_t n, int32_t n_ref, uint32_t *block_size) {
    if (p + 36 > n) return false;
    const uint32_t bs = bam_ld32(t, p);
    if (bs < 33 || bs > (1u << 24)) return false;
    const int32_t refid = (int32_t)bam_ld32(t, p + 4), pos = (int32_t)bam_ld32(t, p + 8);
    if (refid < -1 || refid >= n_ref || pos < -1) return false;
    const uint32_t w = bam_ld32(t, p + 12), w2 = bam_ld32(t, p + 16);
    const uint32_t l_name = w & 0xFF, n_cigar = w2 & 0xFFFF;
    const int32_t l_seq = (int32_t)bam_ld32(t, p + 20), nref = (int32_t)bam_ld32(t, p + 24), npos = (int32_t)bam_ld32(t, p + 28);
    if (l_name < 1 || l_seq < 0 || nref < -1 || nref >= n_ref || npos < -1) return false;
    const uint64_t fixed = 32ull + l_name + 4ull * n_cigar + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq;
    if (fixed > bs) return false;
    const uint64_t name_end = p + 36 + l_name - 1;
    if (name_end < n) {
        if (t[name_end] != 0) return false;
        if (l_name >= 2 && t[name_end - 1] == 0) return false;
    }
    *block_size = bs;
    return true;
}

__global__ void __launch_bounds__(256) k_bam_seg_guess(const uint8_t *text, uint64_t n, int32_t n_ref, BamSegs G) {
    const int lane = threadIdx.x & 63;
    const uint32_t seg = blockIdx.x * 4 + (threadIdx.x >> 6) + 1;      // segment 0 starts where the caller says
    if (seg >= G.n_segs) return;
    const uint64_t lo = (uint64_t)seg * BAM_SEG, hi = lo + BAM_SEG < n ? lo + BAM_SEG : n;
    uint32_t found = BAM_NONE;
    for (uint64_t base = lo; base < hi && found == BAM_NONE; base += 64) {
        const uint64_t p = base + lane;
        uint32_t bs = 0;
        unsigned long long m = __ballot(p < hi && bam_plausible(text, p, n, n_ref, &bs));
        while (m) {
            const int f = __ffsll(m) - 1;
            m &= m - 1;
            // the candidate's chain (every lane follows it: uniform)
            uint64_t q = base + f;
            bool good = true;
            for (int j = 0; j < BAM_CHAIN && good; ++j) {
                if (q + 36 > n) break;      // the stream ends: nothing more to check
                uint32_t b2 = 0;
                good = bam_plausible(text, q, n, n_ref, &b2);
                q += 4ull + b2;
            }
            if (good) { found = (uint32_t)(base + f); break; }
        }
    }
    if (lane == 0) G.start[seg] = found;
}

// one lane: the chain from p to the first record start at or behind `end` (or the stream's cut); returns the landing point
__device__ __forceinline__ uint32_t bam_walk(const uint8_t *text, uint64_t n, uint64_t p, uint64_t end, uint32_t *slots, uint32_t *count,
                                             uint32_t *bad) {
    uint32_t c = 0, b = 0;
    while (p < end) {
        if (p + 4 > n) break;                                   // the stream's end cuts the size field
        const uint32_t bs = bam_ld32(text, p);
        if (bs < 33 || bs > (1u << 29)) { b = 1; break; }       // BamReader::next returns -2 / well_formed fails: the host's case
        if (p + 4 + (uint64_t)bs > n) break;                    // ... or the record
        if (c < BAM_SEG_SLOTS) slots[c] = (uint32_t)p;
        ++c;
        p += 4ull + bs;
    }
    *count = c;
    *bad = b;
    return (uint32_t)p;
}

__global__ void __launch_bounds__(256) k_bam_seg_walk(const uint8_t *text, uint64_t n, BamSegs G) {
    const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
    if (seg >= G.n_segs) return;
    const uint32_t s = G.start[seg];
    uint32_t c = 0, b = 0, l = BAM_NONE;
    if (s != BAM_NONE) {
        const uint64_t end = (uint64_t)(seg + 1) * BAM_SEG;
        l = bam_walk(text, n, s, end < n ? end : n, G.slots + (size_t)seg * BAM_SEG_SLOTS, &c, &b);
    }
    G.count[seg] = c;
    G.bad[seg] = b;
    G.land[seg] = l;
}

// out[0] = 1 when some segment does not start where the one before it lands
__global__ void __launch_bounds__(256) k_bam_seg_check(BamSegs G, uint32_t *out) {
    const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (seg >= G.n_segs) return;
    const uint32_t s = G.start[seg], prev = G.land[seg - 1];
    if (s == BAM_NONE || prev == BAM_NONE || s != prev) atomicOr(out, 1u);
}

// One wavefront puts right what does not follow from the segment before: its lanes look at 64 segments at a time and
// skip the runs that are consistent; the first segment that is not -- a wrong guess, no guess, a record longer than a
// segment, the tail behind the record the stream's end cuts -- is walked again from the landing point of the segment
// before it by lane 0, and the scan goes on behind it with the new landing point.  out[1] counts the segments walked
// again; more than `budget` of them gives the chunk to the host parsers (out[0] |= 2).
__global__ void __launch_bounds__(64) k_bam_seg_repair(const uint8_t *text, uint64_t n, BamSegs G, uint32_t *out, uint32_t budget) {
    if (blockIdx.x || !(out[0] & 1)) return;
    const int lane = threadIdx.x;
    uint32_t prev = G.land[0], walked = 0, seg = 1;
    while (seg < G.n_segs) {
        const uint32_t mine = seg + lane;
        bool ok = true;
        if (mine < G.n_segs) {
            const uint32_t before = lane == 0 ? prev : G.land[mine - 1];
            const uint32_t st = G.start[mine];
            ok = st != BAM_NONE && before != BAM_NONE && st == before;
        }
        const unsigned long long m = __ballot(!ok);
        if (!m) {
            const uint32_t last = seg + 63 < G.n_segs ? seg + 63 : G.n_segs - 1;
            prev = G.land[last];
            seg = last + 1;
            continue;
        }
        const uint32_t f = (uint32_t)__ffsll(m) - 1;
        if (f) prev = G.land[seg + f - 1];
        const uint32_t fix = seg + f;
        const uint64_t lo = (uint64_t)fix * BAM_SEG, end = lo + BAM_SEG < n ? lo + BAM_SEG : n;
        uint32_t nl = prev, stop = 0;
        if (lane == 0) {
            if (prev >= end || prev == BAM_NONE) {               // a record covers the whole segment: nothing starts here
                G.start[fix] = BAM_NONE; G.count[fix] = 0; G.bad[fix] = 0; G.land[fix] = prev;
            } else {
                uint32_t c = 0, b = 0;
                G.start[fix] = prev;
                nl = bam_walk(text, n, prev, end, G.slots + (size_t)fix * BAM_SEG_SLOTS, &c, &b);
                G.land[fix] = nl; G.count[fix] = c; G.bad[fix] = b;
                stop = b | 0x100;                                // (a malformed block: the chunk is the host's anyway)
            }
        }
        stop = (uint32_t)__builtin_amdgcn_readfirstlane((int)stop);
        prev = (uint32_t)__builtin_amdgcn_readfirstlane((int)nl);
        if (stop & 0x100) ++walked;
        if ((stop & 1) || walked > budget) { if (lane == 0 && walked > budget) atomicOr(out, 2u); break; }
        __threadfence();
        seg = fix + 1;
    }
    if (lane == 0) out[1] = walked;
}

// counts -> u64 for the scan; any verified bad block -> out[0] |= 4
__global__ void __launch_bounds__(256) k_bam_seg_counts(BamSegs G, uint64_t *counts, uint32_t *out) {
    const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
    if (seg >= G.n_segs) return;
    counts[seg] = G.count[seg];
    if (G.bad[seg] || G.count[seg] > BAM_SEG_SLOTS) atomicOr(out, 4u);
}

// (bias: the offset of the stream's part the segments were cut from)
__global__ void __launch_bounds__(256) k_bam_rec_offsets(BamSegs G, const uint64_t *first, uint32_t bias, uint32_t *rec_off) {
    const uint32_t seg = blockIdx.x;
    const uint32_t c = G.count[seg];
    const uint64_t at = first[seg];
    const uint32_t *s = G.slots + (size_t)seg * BAM_SEG_SLOTS;
    for (uint32_t j = threadIdx.x; j < c; j += blockDim.x) rec_off[at + j] = s[j] + bias;
}

// bytes of the aux value whose type byte is at a (skip_aux of htslib / aux_value_size of bam_io.cc); 0: malformed
__device__ __forceinline__ uint32_t bam_aux_size(const uint8_t *t, uint64_t a, uint64_t end) {
    if (a >= end) return 0;
    uint32_t fixed = 0;
    switch (t[a]) {
        case 'A': case 'c': case 'C': fixed = 1; break;
        case 's': case 'S': fixed = 2; break;
        case 'i': case 'I': case 'f': fixed = 4; break;
        case 'd': fixed = 8; break;
        case 'Z': case 'H': {
            for (uint64_t q = a + 1; q < end; q += 4) {      // memchr(a + 1, 0, end - (a + 1)), four bytes a step
                const uint32_t w = bam_ld32(t, q);
                const uint32_t z = (w - 0x01010101u) & ~w & 0x80808080u;
                if (z) {
                    const uint64_t at = q + ((__ffs((int)z) - 1) >> 3);
                    return at < end ? (uint32_t)(at - a) + 1 : 0;
                }
            }
            return 0;
        }
        case 'B': {
            if (end - a < 6) return 0;
            uint32_t each;
            switch (t[a + 1]) {
                case 'c': case 'C': each = 1; break;
                case 's': case 'S': each = 2; break;
                case 'i': case 'I': case 'f': each = 4; break;
                default: return 0;
            }
            const uint64_t total = 6 + (uint64_t)each * bam_ld32(t, a + 2);
            return total <= end - a ? (uint32_t)total : 0;
        }
        default: return 0;
    }
    return 1 + fixed <= end - a ? 1 + fixed : 0;
}

// out: [0] flags, [1] longest, [2] shortest; first_seen[id] = smallest record ordinal (of the chunk) that carries it
__global__ void __launch_bounds__(256) k_bam_records(const uint8_t *text, uint64_t n_records, int use_oq, BamRgTable T, BamIndex X,
                                                      uint32_t *out, unsigned long long *first_seen) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_records) return;
    const uint64_t p = X.rec_off[r];
    const uint32_t bs = bam_ld32(text, p);
    const uint64_t d = p + 4;                                  // the block (BamRecord::data)
    const uint32_t l_name = text[d + 8], n_cigar = bam_ld16(text, d + 12), flag = bam_ld16(text, d + 14), l_seq = bam_ld32(text, d + 16);
    const uint64_t seq_at = 32ull + l_name + 4ull * n_cigar, qual_at = seq_at + ((uint64_t)l_seq + 1) / 2, aux_at = qual_at + l_seq;
    uint32_t fl = 0;
    // BamRecord::well_formed
    if (l_name < 1 || aux_at > bs || (int32_t)l_seq < 0) {
        atomicOr(&out[0], (uint32_t)BAMF_FALLBACK);
        X.seq_off[r] = X.qual_off[r] = X.qsrc_off[r] = (uint32_t)d; X.l_seq[r] = 0; X.oq_at[r] = X.oq_vlen[r] = 0; X.flag[r] = 0; X.rg[r] = 0;
        X.base_sz[r] = 0;
        return;
    }
    // the tags: bam_aux_get walks them in order and stops at its tag; a malformed one in front of it is EINVAL
    uint64_t a = d + aux_at;
    const uint64_t end = d + bs;
    uint64_t rg_at = 0, oq_at = 0;
    bool corrupt = false;
    while (end - a >= 3) {
        const uint8_t t0 = text[a], t1 = text[a + 1];
        a += 2;
        const uint32_t sz = bam_aux_size(text, a, end);
        if (!sz) { corrupt = true; break; }
        if (t0 == 'R' && t1 == 'G' && !rg_at) rg_at = a;
        if (t0 == 'O' && t1 == 'Q' && !oq_at) oq_at = a;
        a += sz;
    }
    // RG (readutils.cc:41-58): there, type Z or H (bam_aux2Z), named by the header
    uint32_t rg = 0xFFFF;
    if (!rg_at || (text[rg_at] != 'Z' && text[rg_at] != 'H')) {
        fl |= BAMF_FALLBACK;
    } else {
        const uint64_t v = rg_at + 1;
        auto same_as = [&](uint32_t i) -> bool {
            const uint32_t o = T.id_off[i], len = T.id_off[i + 1] - o;
            bool same = v + len < end && text[v + len] == 0;      // (the value's NUL lies inside the record)
            for (uint32_t j = 0; j < len && same; ++j) same = text[v + j] == T.ids[o + j];
            return same;
        };
        if (T.hash_mask) {
            uint32_t h = 2166136261u;
            for (uint64_t j = v; j < end && text[j]; ++j) h = bam_fnv1a(h, text[j]);
            for (uint32_t probe = 0; probe <= T.hash_mask && rg == 0xFFFF; ++probe) {
                const uint32_t i = T.hash_slots[(h + probe) & T.hash_mask];
                if (i == 0xFFFF) break;
                if (same_as(i)) rg = i;
            }
        } else {
            for (uint32_t i = 0; i < T.n_ids && rg == 0xFFFF; ++i)
                if (same_as(i)) rg = i;
        }
        if (rg == 0xFFFF) fl |= BAMF_FALLBACK;      // a read group without an @RG line: the host path's dictionary handles it
        // first appearance: a look first -- after the first wavefronts of a chunk nearly every record finds a smaller
        // ordinal there already (with an atomic per record, 2.5e6 of them on one address were 25 of the kernel's 28 ms)
        else if (__hip_atomic_load(&first_seen[rg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (unsigned long long)r)
            atomicMin(&first_seen[rg], (unsigned long long)r);
    }
    // OQ (readutils.cc:16-31; htsiter.cc:13-26)
    uint32_t oq_vlen = 0;
    if (oq_at) {
        const uint32_t sz = bam_aux_size(text, oq_at, end);      // (> 0: the walk got past it)
        const uint8_t ty = text[oq_at];
        oq_vlen = (ty == 'Z' || ty == 'H') ? sz - 2 : 0;
        if (ty != 'Z') fl |= BAMF_OQ_UNWRITABLE;                 // bam_aux_update_str: EINVAL
        if (use_oq && ((ty != 'Z' && ty != 'H') || oq_vlen != l_seq)) fl |= BAMF_FALLBACK;
    } else {
        if (corrupt) fl |= BAMF_OQ_UNWRITABLE;
        if (use_oq) fl |= BAMF_FALLBACK;                         // "--use-oq was specified but unable to read OQ tag"
    }
    X.seq_off[r] = (uint32_t)(d + seq_at);
    X.qual_off[r] = (uint32_t)(d + qual_at);
    X.qsrc_off[r] = use_oq && oq_at ? (uint32_t)(oq_at + 1) : (uint32_t)(d + qual_at);
    X.l_seq[r] = l_seq;
    X.oq_at[r] = (uint32_t)oq_at;
    X.oq_vlen[r] = oq_vlen;
    X.flag[r] = (uint16_t)flag;
    X.rg[r] = (uint16_t)rg;
    X.base_sz[r] = l_seq;
    if (fl) atomicOr(&out[0], fl);
    if (__hip_atomic_load(&out[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < l_seq) atomicMax(&out[1], l_seq);
    if (__hip_atomic_load(&out[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > l_seq) atomicMin(&out[2], l_seq);
}

// second-in-pair flags (readutils.cc:59) and the dense read-group index of every record
__global__ void __launch_bounds__(256) k_bam_read_meta(BamIndex X, uint64_t n_records, const uint16_t *dense, uint8_t *second, uint16_t *rg) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_records) return;
    second[r] = (X.flag[r] & 0x80) ? 1 : 0;
    rg[r] = dense[X.rg[r]];
}

__global__ void __launch_bounds__(256) k_bam_gather(const uint8_t *text, BamIndex X, const uint64_t *base_off, uint64_t n_records, int use_oq,
                                                     uint8_t *seq_text, uint8_t *qual) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
    for (uint64_t r = wave; r < n_records; r += n_waves) {
        const uint32_t n = X.l_seq[r];
        const uint8_t *s = text + X.seq_off[r], *q = text + X.qsrc_off[r];
        const bool rev = X.flag[r] & 16;      // bam_is_rev
        const uint64_t at = base_off[r];
        for (uint32_t i = lane; i < n; i += 64) {
            const uint32_t code = (s[i >> 1] >> ((~i & 1) << 2)) & 15;      // bam_seqi
            const uint8_t qv = use_oq ? (uint8_t)(q[i] - 33) : q[i];
            if (!rev) {
                seq_text[at + i] = (uint8_t)"=ACMGRSVTWYHKDBN"[code];       // seq_nt16_str
                qual[at + i] = qv;
            } else {
                // readutils.hh:35-36: the complement of A/C/G/T, 'N' for every other code; then reversed (with the qualities)
                const uint8_t c = code == 1 ? 'T' : code == 2 ? 'G' : code == 4 ? 'C' : code == 8 ? 'A' : 'N';
                seq_text[at + (n - 1 - i)] = c;
                qual[at + (n - 1 - i)] = qv;
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_bam_out_sizes(const uint8_t *text, BamIndex X, uint64_t n_records, int set_oq) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_records) return;
    const uint64_t bs = bam_ld32(text, X.rec_off[r]);
    uint64_t sz = 4 + bs;
    if (set_oq) sz = X.oq_at[r] ? sz + X.l_seq[r] - X.oq_vlen[r] : sz + 4 + X.l_seq[r];      // replaced in place / "OQ" 'Z' value NUL appended
    X.out_sz[r] = sz;
}

// BamFile::recalibrate + sam_write1 (htsiter.cc:11-45) of every record; new_qual: the batch's new qualities in the batch's
// base order (sequencing orientation)
__global__ void __launch_bounds__(256) k_bam_rewrite(const uint8_t *text, BamIndex X, const uint64_t *base_off, const uint64_t *out_off,
                                                      uint64_t n_records, int set_oq, const uint8_t *new_qual, uint8_t *payload) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
    for (uint64_t r = wave; r < n_records; r += n_waves) {
        const uint64_t p = X.rec_off[r];
        const uint8_t *src = text + p;
        uint8_t *dst = payload + out_off[r];
        const uint32_t out_len = (uint32_t)(out_off[r + 1] - out_off[r]);
        const uint32_t old_len = 4 + bam_ld32(text, p);
        const uint32_t n = X.l_seq[r];
        const uint32_t q0 = (uint32_t)(X.qual_off[r] - p), q1 = q0 + n;          // the quality field, relative to the record
        const bool rev = X.flag[r] & 16;
        const uint8_t *nq = new_qual + base_off[r];
        const uint32_t oq = X.oq_at[r] ? (uint32_t)(X.oq_at[r] - p) : 0;         // the OQ tag's type byte
        // with --set-oq: the bytes up to v0 are the record's own (new qualities in [q0, q1)); [v0, v0 + n] is the new OQ
        // value and its NUL; what follows comes from `tail` on in the old record.  A missing tag is appended behind the
        // old record with its three header bytes.
        uint32_t v0 = 0xFFFFFFFFu, tail = 0;
        if (set_oq) {
            if (oq) { v0 = oq + 1; tail = oq + 1 + X.oq_vlen[r] + 1; }
            else { v0 = old_len + 3; tail = old_len; }
        }
        for (uint32_t j = lane; j < out_len; j += 64) {
            uint8_t b;
            if (j < 4) {
                b = (uint8_t)((out_len - 4) >> (8 * j));
            } else if (j >= q0 && j < q1) {
                const uint32_t i = j - q0;
                b = nq[rev ? n - 1 - i : i];                                     // htsiter.cc:27-31
            } else if (j < v0) {
                if (!oq && set_oq && j >= old_len) b = (uint8_t)"OQZ"[j - old_len];
                else b = src[j];
            } else if (j < v0 + n) {
                b = (uint8_t)(src[q0 + (j - v0)] + 33);                          // htsiter.cc:15-17: the quality field as it was
            } else if (j == v0 + n) {
                b = 0;
            } else {
                b = src[tail + (j - (v0 + n + 1))];
            }
            dst[j] = b;
        }
    }
}

// ---- the synthetic data set as files (tools: `kbbq --io-test synth-fastq / synth-bam`): the bench's own reads
// (kbbq_synth_reads, engine.hip: k_synth) formatted on the device, so that a true 30x WGS-size FASTQ or BAM can be fed
// to the command line and its digest compared with bench.py's.  Fixed-width names: every record has the same size.
struct SynthBatch {
    const uint64_t *bases, *nmask;
    const uint8_t *qual;
    uint64_t first, n;      // global index of the batch's first read, reads
    uint32_t read_len;
};
__device__ __forceinline__ void synth_base(const SynthBatch &B, uint64_t g, uint32_t &code, bool &is_n) {
    code = (uint32_t)(B.bases[g >> 5] >> (2 * (g & 31))) & 3;
    is_n = (B.nmask[g >> 6] >> (g & 63)) & 1;
}
__device__ __forceinline__ void synth_name(uint8_t *dst, uint64_t idx) {      // 'r' + 10 decimal digits
    dst[0] = 'r';
    for (int j = 10; j >= 1; --j) { dst[j] = (uint8_t)('0' + idx % 10); idx /= 10; }
}
constexpr uint32_t synth_fastq_record(uint32_t L) { return 2 * L + 17; }      // "@" name "\n" seq "\n+\n" qual "\n"
constexpr uint32_t synth_bam_record(uint32_t L, bool oq) { return 4 + 32 + 12 + (L + 1) / 2 + L + 8 + (oq ? L + 4 : 0); }

__global__ void __launch_bounds__(256) k_synth_fastq(SynthBatch B, uint8_t *out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
    const uint32_t L = B.read_len, W = synth_fastq_record(L);
    for (uint64_t r = wave; r < B.n; r += n_waves) {
        uint8_t *d = out + r * W;
        if (lane == 0) {
            d[0] = '@';
            synth_name(d + 1, B.first + r);
            d[12] = '\n';
            d[13 + L] = '\n'; d[14 + L] = '+'; d[15 + L] = '\n';
            d[16 + 2 * L] = '\n';
        }
        for (uint32_t i = lane; i < L; i += 64) {
            uint32_t c; bool nn;
            synth_base(B, r * L + i, c, nn);
            d[13 + i] = nn ? 'N' : (uint8_t)"ACGT"[c];
            d[16 + L + i] = (uint8_t)(B.qual[r * L + i] + 33);
        }
    }
}

// unaligned BAM records: RG:Z:grp0, FLAG 4 (+16 for about half: those store the reverse complement with the qualities
// reversed, as an aligner would); oq: the true qualities travel in OQ:Z, the quality field holds 11s (configs[3])
__global__ void __launch_bounds__(256) k_synth_bam(SynthBatch B, int oq, uint8_t *out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
    const uint32_t L = B.read_len, W = synth_bam_record(L, oq != 0), half = (L + 1) / 2;
    for (uint64_t r = wave; r < B.n; r += n_waves) {
        uint8_t *d = out + r * W;
        const uint64_t idx = B.first + r;
        const bool rev = ((idx * 2654435761ull) >> 7) & 1;
        if (lane < 36) {
            const uint32_t bs = W - 4, flag = 4u | (rev ? 16u : 0u);
            uint8_t v = 0;
            if (lane < 4) v = (uint8_t)(bs >> (8 * lane));
            else if (lane < 12) v = 0xFF;                                    // refID, pos = -1
            else if (lane == 12) v = 12;                                     // l_read_name
            else if (lane == 14) v = 4680 & 0xFF;                            // bin
            else if (lane == 15) v = 4680 >> 8;
            else if (lane == 18) v = (uint8_t)flag;
            else if (lane >= 20 && lane < 24) v = (uint8_t)(L >> (8 * (lane - 20)));
            else if (lane >= 24 && lane < 32) v = 0xFF;                      // next refID, next pos = -1
            d[lane] = v;
        }
        if (lane == 36) { synth_name(d + 36, idx); d[47] = 0; }
        uint8_t *sq = d + 48, *ql = sq + half, *aux = ql + L;
        if (lane == 37) { aux[0] = 'R'; aux[1] = 'G'; aux[2] = 'Z'; aux[3] = 'g'; aux[4] = 'r'; aux[5] = 'p'; aux[6] = '0'; aux[7] = 0; }
        if (oq && lane == 38) { aux[8] = 'O'; aux[9] = 'Q'; aux[10] = 'Z'; aux[11 + L] = 0; }
        for (uint32_t j = lane; j < half; j += 64) {                        // two stored bases per byte
            uint8_t byte = 0;
            for (uint32_t h = 0; h < 2; ++h) {
                const uint32_t i = 2 * j + h;
                uint32_t code = 0;
                if (i < L) {
                    uint32_t c; bool nn;
                    synth_base(B, r * L + (rev ? L - 1 - i : i), c, nn);
                    if (rev) c = 3 - c;
                    code = nn ? 15u : (1u << c);
                }
                byte |= (uint8_t)(code << (h ? 0 : 4));
            }
            sq[j] = byte;
        }
        for (uint32_t i = lane; i < L; i += 64) {
            const uint8_t q = B.qual[r * L + (rev ? L - 1 - i : i)];
            ql[i] = oq ? (uint8_t)11 : q;
            if (oq) aux[11 + i] = (uint8_t)(q + 33);
        }
    }
}

}  // namespace dfl
}  // namespace kbbq
