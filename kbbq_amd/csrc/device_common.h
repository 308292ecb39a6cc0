// device_common.h -- device-side building blocks shared by the kernels in
// kernels.h: packed-read access, k-mer extraction, hash_ap, the blocked Bloom
// filter (128-bit blocks, one lane per lookup), per-wave LDS staging of reads.  gfx950 only:
// wavefront = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "modmath.h"

namespace kbbq {

constexpr int WAVE = 64;

struct ReadsDev {
    const uint64_t *bases;
    const uint64_t *nmask;
    const uint8_t *qual;
    const uint64_t *offsets;   // null => uniform
    const uint8_t *flags;      // null => 0
    const uint16_t *rg;        // null => 0
    const uint64_t *offcase;   // null => none: 1 bit per base, ACGT bases whose raw character is not the upper-case letter
    uint32_t *hint_sampled;    // optional: 1 bit per base, set where this read itself inserted the k-mer starting there
    uint32_t *hint_trusted;    //           into the sampled (pass 1) / trusted (pass 2) filter
    uint64_t n_reads;
    uint64_t n_bases;
    uint32_t read_len;
};

__device__ __forceinline__ void read_span(const ReadsDev &R, uint64_t r, uint64_t &off, uint32_t &len) {
    if (R.offsets) {
        off = R.offsets[r];
        len = (uint32_t)(R.offsets[r + 1] - off);
    } else {
        off = r * (uint64_t)R.read_len;
        len = R.read_len;
    }
}

// 64 bits of a packed bit stream starting at bit position `bitpos` (arrays carry
// one spare word, see kbbq_engine.h).
__device__ __forceinline__ uint64_t window64(const uint64_t *words, uint64_t bitpos) {
    const uint64_t w = bitpos >> 6;
    const unsigned s = (unsigned)(bitpos & 63);
    const uint64_t lo = words[w];
    if (s == 0) return lo;
    const uint64_t hi = words[w + 1];
    return (lo >> s) | (hi << (64 - s));
}

// reverse the order of the 32 two-bit groups of x
__device__ __forceinline__ uint64_t rev2(uint64_t x) {
    x = __brevll(x);
    return ((x & 0x5555555555555555ULL) << 1) | ((x >> 1) & 0x5555555555555555ULL);
}

struct KParams {
    int k;
    unsigned shift;      // 2*(k-1)
    uint64_t mask;       // low 2k bits
    uint32_t nmask_bits; // low k bits
};

// Canonical k-mer of the k bases starting at base offset b of the batch
// (bloom::Kmer after k pushes, bloom.hh:350-360): the forward word holds the
// first base in its top two bits, the reverse-complement word holds 3-c of the
// first base in its low two bits; key = min of the two.  valid <=> no non-ACGT
// base in the window (Kmer::valid, bloom.hh:368).
__device__ __forceinline__ uint64_t kmer_at(const ReadsDev &R, const KParams &K, uint64_t b, bool &valid) {
    const uint64_t w = window64(R.bases, 2 * b);
    const uint64_t rc = (~w) & K.mask;
    const uint64_t fw = rev2(w) >> (64 - 2 * K.k);
    const uint32_t nm = (uint32_t)window64(R.nmask, b) & K.nmask_bits;
    valid = nm == 0;
    return fw < rc ? fw : rc;
}

// bloom_filter::hash_ap on an 8-byte key (bloom_filter.hpp:551-571): one round
// of the >= 8-byte loop.
__device__ __forceinline__ uint32_t hash_ap8(uint64_t key, uint32_t h) {
    const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
    h ^= (h << 7) ^ (lo * (h >> 3)) ^ (~((h << 11) + (hi ^ (h >> 5))));
    return h;
}

// The filter in the engine's layout.  The reference's pattern generator sends sampled bit number b
// (0..511) to 64-bit word (b>>8)*4 + ((b>>3)&3) and to bit b&63 of that word (get_vector_unit,
// bloom.hh:110-113,228): bits 3-4 of the bit index always equal the word's unit number, so every
// 64-bit word of a pattern -- and therefore of the table, which only ever receives ORs of patterns --
// uses just 16 of its bits, 128 of the 512 in a block.  The engine stores exactly those: a block and a
// pattern are 128 bits (two u64; word w of the reference is the 16-bit field w&3 of u64 w>>2, field bit
// c = reference bit ((c>>3)<<5) | ((w&3)<<3) | (c&7)).  One lane fetches a whole block with one 16-byte
// load, the pattern table is 1 MiB and stays in L2, the filters are a quarter of the reference's size.
// kbbq_filter_download expands to the reference layout (host_model.h: expand_block).
struct FiltDev {
    ulonglong2 *table;           // n_blocks x 16 bytes
    const ulonglong2 *patterns;  // 65536 x 16 bytes
    uint64_t n_blocks;
    uint64_t mod_magic;          // modmath.h: ModMagic.m64
    uint32_t mod_m32;            //            ModMagic.m32
    uint32_t salt0, salt1;
};

// get_block (bloom.hh:99-105): hash % num_blocks, exact (modmath.h)
__device__ __forceinline__ uint32_t block_of(const FiltDev &f, uint64_t key) {
    return mod_hash(hash_ap8(key, f.salt0), f.n_blocks, f.mod_magic, f.mod_m32);
}
// get_pattern (bloom.hh:250-253)
__device__ __forceinline__ uint32_t pattern_of(const FiltDev &f, uint64_t key) {
    return hash_ap8(key, f.salt1) & 0xFFFFu;
}

// One lane, one k-mer.  tools/probe_hbm: 64 lanes x one random 16-byte block each reach the chip's
// random-access ceiling (about 50 G blocks/s) with a single load in flight per lane.
// pattern_blocked_bf::contains, bloom.hh:276-292
__device__ __forceinline__ bool bloom_has(const FiltDev &f, uint32_t blk, uint32_t pat) {
    const ulonglong2 t = f.table[blk];
    const ulonglong2 p = f.patterns[pat];
    return ((p.x & ~t.x) | (p.y & ~t.y)) == 0;
}
__device__ __forceinline__ bool bloom_has(const FiltDev &f, uint64_t key) {
    return bloom_has(f, block_of(f, key), pattern_of(f, key));
}
// pattern_blocked_bf::insert, bloom.hh:255-267; returns whether the pattern was already there.  Only the
// missing bits are OR-ed in, so a block that already holds the pattern is not written at all (legal
// because bits are only ever set).
__device__ __forceinline__ bool bloom_put(const FiltDev &f, uint32_t blk, uint32_t pat) {
    const ulonglong2 t = f.table[blk];
    const ulonglong2 p = f.patterns[pat];
    const uint64_t mx = p.x & ~t.x, my = p.y & ~t.y;
    unsigned long long *w = reinterpret_cast<unsigned long long *>(f.table + blk);
    if (mx) atomicOr(w, (unsigned long long)mx);
    if (my) atomicOr(w + 1, (unsigned long long)my);
    return (mx | my) == 0;
}

// ---- per-wave staging of one read's packed words in LDS -------------------------------------------
// The streaming kernels handle one read per wavefront.  Instead of every lane fetching its own
// unaligned 64-bit windows from global memory (four dependent round trips per 64 k-mers), the wave
// loads the read's words once -- one u64 per lane, next read's words in flight while this one is
// processed -- parks them in its LDS slice and every lane cuts its windows out of LDS with
// v_alignbit.  Streams: 2-bit bases, N mask, the pass's hint bits, one extra 1-bit stream.
template <int NW>
struct Stage {
    static constexpr int NB = 2 * NW + 2;   // u64 words: NW*64 bases at any alignment + the window's spill word
    static constexpr int NM = NW + 2;       // same for a 1-bit-per-base stream
    static constexpr int B = 0, M = NB, H = NB + NM, X = NB + 2 * NM;
    static constexpr int WORDS = NB + 3 * NM;          // 48 for NW = 8: one word per lane
    static constexpr int RES = 2 * NW + 4;             // u32: zero, 2*NW result dwords, zero, zero (+1 to stay even)
    static constexpr int LDS_U32 = 2 * WORDS + 2 * RES;
    static_assert(WORDS <= 64, "one staged word per lane");
};

template <int NW>
__device__ __forceinline__ uint64_t stage_fetch(const ReadsDev &R, const uint64_t *hint, const uint64_t *extra,
                                                uint64_t extra_pos, uint64_t extra_max, uint64_t off, int lane) {
    using S = Stage<NW>;
    const uint64_t *p = nullptr;
    uint64_t idx = 0, mx = 0;
    if (lane < S::M) { p = R.bases; idx = (off >> 5) + lane; mx = R.n_bases / 32 + 1; }
    else if (lane < S::H) { p = R.nmask; idx = (off >> 6) + (lane - S::M); mx = R.n_bases / 64 + 1; }
    else if (lane < S::X) { p = hint; idx = (off >> 6) + (lane - S::H); mx = R.n_bases / 64 + 1; }
    else if (lane < S::WORDS) { p = extra; idx = (extra_pos >> 6) + (lane - S::X); mx = extra_max; }
    return p ? p[idx < mx ? idx : mx] : 0;
}

// the LDS slice is dwords throughout (windows are cut with 32-bit funnel shifts)
__device__ __forceinline__ void stage_store(uint32_t *slice, int lane, uint64_t word) {
    reinterpret_cast<uint2 *>(slice)[lane] = make_uint2((uint32_t)word, (uint32_t)(word >> 32));
}

// 64 / 32 bits of an LDS bit stream (viewed as dwords) starting at bit `pos`
__device__ __forceinline__ uint64_t lds_window64(const uint32_t *d, int pos) {
    const int i = pos >> 5, sh = pos & 31;
    const uint32_t a = d[i], b = d[i + 1], c = d[i + 2];
    return (uint64_t)__builtin_amdgcn_alignbit(b, a, sh) | ((uint64_t)__builtin_amdgcn_alignbit(c, b, sh) << 32);
}
__device__ __forceinline__ uint32_t lds_window32(const uint32_t *d, int pos) {
    const int i = pos >> 5, sh = pos & 31;
    return __builtin_amdgcn_alignbit(d[i + 1], d[i], sh);
}
__device__ __forceinline__ uint32_t lds_bit(const uint32_t *d, int pos) { return (d[pos >> 5] >> (pos & 31)) & 1u; }

// canonical key of the k bases in the low 2k bits of w (bloom::Kmer, bloom.hh:350-360; see kmer_at)
__device__ __forceinline__ uint64_t canon_key(uint64_t w, const KParams &K) {
    const uint64_t rc = (~w) & K.mask;
    const uint64_t fw = rev2(w) >> (64 - 2 * K.k);
    return fw < rc ? fw : rc;
}

// OR the 64 flags of one chunk (bit l = position `first` + l of the batch) into a shared bit array
__device__ __forceinline__ void or_bits64(uint32_t *bits, uint64_t first, uint64_t word, int lane) {
    if (lane < 2) {
        const uint32_t v = (uint32_t)(word >> (32 * lane));
        const uint64_t g = first + 32 * lane;
        if (v) {
            atomicOr(&bits[g >> 5], v << (g & 31));
            if (g & 31) atomicOr(&bits[(g >> 5) + 1], v >> (32 - (g & 31)));
        }
    }
}

// Which quality values a kernel has met: a 256-bit set per workgroup in LDS (the reference's covariate tables grow
// with the largest quality seen, covariateutils.cc:65-76,102-116; a quality is a uint8_t), OR-ed into the engine's
// 8-word array when the block ends.  A look first, an LDS atomic only the first time a value shows up.
__device__ __forceinline__ void qseen_note(uint32_t *lds_mask, uint32_t q) {
    const uint32_t bit = 1u << (q & 31);
    if (!(lds_mask[q >> 5] & bit)) atomicOr(&lds_mask[q >> 5], bit);
}

// ---- dynamic distribution of a batch's reads over the wavefronts of a launch --------------------------------------
// A grid-stride loop gives every wavefront the same number of reads, so the launch runs in rounds of "as many
// workgroups as the chip holds" and its last round leaves CUs idle (4096 workgroups over 1536 resident ones: 2.67
// rounds of work in the time of 3).  Here the reads are handed out in chunks of CH consecutive ones by a global
// counter (zeroed before the launch): a wavefront that starts late, or got cheap reads, simply takes fewer chunks.
// The next chunk is requested one chunk ahead, so the counter's round trip is hidden, and the caller can still
// prefetch the read after the current one (peek()).
template <int CH>
struct ReadChunks {
    unsigned int *counter;
    uint64_t n_reads, cur, end;
    uint32_t next_chunk;
    __device__ __forceinline__ uint32_t grab(int lane) {
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(counter, 1u);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
    }
    // first read of this wavefront (>= n_reads: none)
    __device__ __forceinline__ uint64_t begin(unsigned int *counter_, uint64_t n_reads_, int lane) {
        counter = counter_;
        n_reads = n_reads_;
        const uint32_t c = grab(lane);
        next_chunk = grab(lane);
        cur = (uint64_t)c * CH;
        end = cur + CH < n_reads ? cur + CH : n_reads;
        return cur;
    }
    // the read after `cur` (>= n_reads: none); call once per read, before processing it, then advance with the value
    __device__ __forceinline__ uint64_t peek(int lane) {
        if (cur + 1 < end) return cur + 1;
        const uint64_t nb = (uint64_t)next_chunk * CH;
        return nb;
    }
    __device__ __forceinline__ void advance(int lane) {
        if (cur + 1 < end) { ++cur; return; }
        cur = (uint64_t)next_chunk * CH;
        end = cur + CH < n_reads ? cur + CH : n_reads;
        if (cur < n_reads) next_chunk = grab(lane);
    }
};

// select W[idx] from a small wave-uniform array without dynamic indexing
template <int NW>
__device__ __forceinline__ uint64_t sel_word(const uint64_t (&W)[NW], int idx) {
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) v = (idx == i) ? W[i] : v;
    return v;
}

// popcount of bits lo..hi (inclusive, hi-lo < 64) of a multi-word bit set
template <int NW>
__device__ __forceinline__ int range_popc(const uint64_t (&W)[NW], int lo, int hi) {
    const int a = lo >> 6, b = hi >> 6;
    const uint64_t wa = sel_word<NW>(W, a);
    if (a == b) {
        const int n = hi - lo + 1;
        const uint64_t m = (n >= 64 ? ~0ULL : ((1ULL << n) - 1)) << (lo & 63);
        return __popcll(wa & m);
    }
    const uint64_t wb = sel_word<NW>(W, b);
    const int nb = (hi & 63) + 1;
    return __popcll(wa >> (lo & 63)) + __popcll(wb & (nb >= 64 ? ~0ULL : ((1ULL << nb) - 1)));
}

// splitmix64 finalizer: the counter-based generator of the synthetic data set
__device__ __host__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

}  // namespace kbbq
