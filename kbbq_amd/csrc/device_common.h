// device_common.h -- device-side building blocks shared by the kernels in
// engine.hip: packed-read access, k-mer extraction, hash_ap, the blocked Bloom
// filter (8-lane cooperative form and single-lane form).  gfx950 only:
// wavefront = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kbbq {

constexpr int WAVE = 64;

struct ReadsDev {
    const uint64_t *bases;
    const uint64_t *nmask;
    const uint8_t *qual;
    const uint64_t *offsets;   // null => uniform
    const uint8_t *flags;      // null => 0
    const uint16_t *rg;        // null => 0
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

struct FiltDev {
    uint64_t *table;           // n_blocks x 8 words
    const uint64_t *patterns;  // 65536 x 8 words
    uint64_t n_blocks;
    uint64_t mod_magic;        // 2^64 / n_blocks + 1 (fastmod), 0 when n_blocks >= 2^32
    uint32_t salt0, salt1;
};

// get_block (bloom.hh:99-105): hash % num_blocks, exact via Lemire's fastmod
__device__ __forceinline__ uint32_t block_of(const FiltDev &f, uint64_t key) {
    const uint32_t h = hash_ap8(key, f.salt0);
    if (f.n_blocks > 0xFFFFFFFFULL) return h;
    const uint64_t low = f.mod_magic * h;
    return (uint32_t)__umul64hi(low, f.n_blocks);
}
// get_pattern (bloom.hh:250-253)
__device__ __forceinline__ uint32_t pattern_of(const FiltDev &f, uint64_t key) {
    return hash_ap8(key, f.salt1) & 0xFFFFu;
}

// The cooperative Bloom access.  A wavefront holds one k-mer per lane (active, blk, pat); the 64-byte
// block and the 64-byte pattern of every active k-mer are fetched by a GROUP of lanes so that each
// block is exactly one fully used 64-byte request (tools/probe_hbm: only such shapes reach the chip's
// random-64-byte ceiling):
//   LANES = 8: eight lanes x 8 bytes, 8 rounds of 8 k-mers.  One atomic request per inserted block:
//              used by the kernels that insert.
//   LANES = 4: four lanes x 16 bytes, 4 rounds of 16 k-mers: half the shuffles and ballots, shorter
//              dependent chain: used by the latency-bound correction walk.
// Returns, in the OWNING lane, whether the filter contained the pattern before this call
// (pattern_blocked_bf::contains, bloom.hh:276-292).  With INSERT the missing bits are OR-ed in
// atomically (pattern_blocked_bf::insert, bloom.hh:255-267); a block that already holds the pattern is
// not written at all, which is legal because bits are only ever set.
// ALL 64 LANES MUST CALL THIS TOGETHER (never behind a short-circuit || or &&).
template <bool INSERT, int LANES = 8>
__device__ __forceinline__ bool bloom_coop(const FiltDev &f, bool active, uint32_t blk, uint32_t pat) {
    const int lane = threadIdx.x & 63;
    const unsigned long long live = __ballot(active);
    const uint32_t pa = pat | ((uint32_t)active << 16);   // pattern index and the active flag travel in one shuffle
    bool contained = false;
    if constexpr (LANES == 8) {
        const int sub = lane & 7, grp = lane >> 3;
        uint64_t tv[8], pv[8];
        uint32_t bb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            tv[j] = 0;
            pv[j] = 0;
            bb[j] = 0;
            if (((live >> (8 * j)) & 0xFFull) == 0) continue;   // wave-uniform: nobody owns a k-mer in this round
            const int src = j * 8 + grp;
            bb[j] = __shfl(blk, src);
            const uint32_t q = __shfl(pa, src);
            const uint32_t p = q & 0xFFFFu;
            if (q >> 16) {
                pv[j] = f.patterns[(uint64_t)p * 8 + sub];
                tv[j] = f.table[(uint64_t)bb[j] * 8 + sub];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint64_t miss = pv[j] & ~tv[j];
            if (INSERT) {
                if (miss) atomicOr((unsigned long long *)&f.table[(uint64_t)bb[j] * 8 + sub], (unsigned long long)miss);
            }
            const unsigned long long bal = __ballot(miss != 0);
            if (grp == j) contained = ((bal >> (8 * sub)) & 0xFFull) == 0;
        }
    } else {
        const int sub = lane & 3, grp = lane >> 2;
        ulonglong2 tv[4], pv[4];
        uint32_t bb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tv[j] = make_ulonglong2(0, 0);
            pv[j] = make_ulonglong2(0, 0);
            bb[j] = 0;
            if (((live >> (16 * j)) & 0xFFFFull) == 0) continue;
            const int src = j * 16 + grp;
            bb[j] = __shfl(blk, src);
            const uint32_t q = __shfl(pa, src);
            const uint32_t p = q & 0xFFFFu;
            if (q >> 16) {
                pv[j] = *reinterpret_cast<const ulonglong2 *>(f.patterns + (uint64_t)p * 8 + sub * 2);
                tv[j] = *reinterpret_cast<const ulonglong2 *>(f.table + (uint64_t)bb[j] * 8 + sub * 2);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t mx = pv[j].x & ~tv[j].x, my = pv[j].y & ~tv[j].y;
            if (INSERT) {
                if (mx) atomicOr((unsigned long long *)&f.table[(uint64_t)bb[j] * 8 + sub * 2], (unsigned long long)mx);
                if (my) atomicOr((unsigned long long *)&f.table[(uint64_t)bb[j] * 8 + sub * 2 + 1], (unsigned long long)my);
            }
            const unsigned long long bal = __ballot((mx | my) != 0);
            if ((lane >> 4) == j) contained = ((bal >> (4 * (lane & 15))) & 0xFull) == 0;
        }
    }
    return contained;
}

// One lane, one k-mer: 64-byte block + 64-byte pattern as four 16-byte loads each.
__device__ __forceinline__ bool bloom_query1(const FiltDev &f, uint64_t key) {
    const uint32_t blk = block_of(f, key), pat = pattern_of(f, key);
    const ulonglong2 *t = reinterpret_cast<const ulonglong2 *>(f.table + (uint64_t)blk * 8);
    const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(f.patterns + (uint64_t)pat * 8);
    const ulonglong2 t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
    const ulonglong2 p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3];
    const uint64_t miss = (p0.x & ~t0.x) | (p0.y & ~t0.y) | (p1.x & ~t1.x) | (p1.y & ~t1.y) | (p2.x & ~t2.x) |
                          (p2.y & ~t2.y) | (p3.x & ~t3.x) | (p3.y & ~t3.y);
    return miss == 0;
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
