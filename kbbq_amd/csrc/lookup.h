// lookup.h -- slice-bucketed Bloom LOOKUPS for pass 2 (gfx950), the experiment VERDICT r01 #3 asks for after the
// bucketed inserts: the membership tests of infer_read_errors (overlapping_kmers_in_bf, bloom.cc:28-67) without one
// random 128-byte HBM line per k-mer.
//
// k_infer (kernels.h) fetches, for every k-mer a read did not sample itself, the 16-byte block its hash names: a
// random 128-byte line of the 6 GB sampled filter.  The sampled filter is final when pass 2 starts and a lookup's
// answer is one bit that is only needed when the read's flags are decided, so the lookups of a batch can be deferred
// like the inserts of bucket.h, with a way back for the answers:
//
//   k_emit_lookup    read -> one record (block, pattern index, position id) per k-mer to look up, scattered into the
//                    level-1 buckets of the filter (LDS counting sort per tile, workgroup-private regions)
//   k_split_lookup   level-1 bucket -> its 512 subslices, records become 8 bytes
//   k_apply_lookup   one workgroup per 64 KB subslice: slice -> LDS, every record tested against it, the ids of the
//                    ABSENT k-mers (one in seven) appended to a list per XCD
//   k_scatter_ids    absent ids -> bins of 2^22 positions (LDS counting sort per tile)
//   k_ids_to_bits    one workgroup per bin: ids -> a bit mask in LDS -> the batch's absent-bit array
//   k_infer<.., true>  the flags and insert decisions as before, the answer of a lookup = a bit of that array
//
// A record that does not fit its region is looked up on the spot and, if absent, sets its bit with a global atomic:
// results never depend on the capacities.  Position id = offset of the k-mer's first base in the batch (< 2^31).
//
// STATUS: measurement harness only (KBBQ_LOOKUP_PROBE=1 runs the chain beside the product kernels and discards its
// results; profiles/README.md has the numbers and the decision).
#pragma once
#include "bucket.h"

namespace kbbq {

struct LookupDev {
    unsigned long long *l1;   // [EMIT_GRID][nb1][cap1]  block << 16 | pattern
    uint32_t *l1id;           // parallel: position id
    uint32_t *l1_cnt;         // [EMIT_GRID][nb1]
    unsigned long long *l2;   // [n_sub][cap2]  block-in-subslice << 48 | pattern << 32 | id
    uint32_t *l2_cnt;         // [nb1 * NB2]
    uint32_t *tickets;        // [N_XCD] x CNT_STRIDE
    uint32_t *abs_list;       // [N_XCD][cap_abs] ids of absent k-mers
    uint32_t *abs_cnt;        // [N_XCD] x CNT_STRIDE
    uint32_t *absent_bits;    // 1 bit per position of the batch
    unsigned long long *direct;
    uint32_t cap1, cap2, cap_abs;
    int nb1;
    uint32_t n_sub;
};

__device__ __forceinline__ void absent_set(const LookupDev &B, uint32_t id) { atomicOr(&B.absent_bits[id >> 5], 1u << (id & 31)); }

// ---- level 1 -------------------------------------------------------------------------------------------------
template <int NW, int CH, int RPW>
__global__ void __launch_bounds__(BK_THREADS) k_emit_lookup(ReadsDev R, KParams K, FiltDev F, LookupDev B, uint32_t gbase,
                                                             unsigned long long *lookups) {
    using S = Stage<NW>;
    constexpr int SLOTS = RPW * CH;
    constexpr int TCAP = 8 * SLOTS * 64;
    extern __shared__ unsigned long long sorted[];      // TCAP records, then TCAP ids
    uint32_t *sorted_id = reinterpret_cast<uint32_t *>(sorted + TCAP);
    __shared__ uint32_t stage[8][2 * S::WORDS];
    __shared__ uint32_t hist[MAX_NB1], ofs_l[MAX_NB1], gbase_l[MAX_NB1], fill_l[MAX_NB1], wave_tot[8];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *L32 = stage[w];
    const int k = K.k;
    const int src = (int)blockIdx.x;
    const uint64_t *hint = reinterpret_cast<const uint64_t *>(R.hint_sampled);
    const uint64_t reads_per_tile = 8 * RPW;
    const uint64_t n_tiles = (R.n_reads + reads_per_tile - 1) / reads_per_tile;
    unsigned long long looked = 0, direct = 0;
    hist[threadIdx.x] = 0;
    fill_l[threadIdx.x] = (int)threadIdx.x < B.nb1 ? B.l1_cnt[(size_t)src * B.nb1 + threadIdx.x] : 0u;
    __syncthreads();
    uint64_t off[RPW], word[RPW];
    uint32_t len[RPW];
    auto fetch_tile = [&](uint64_t tile) {
        const uint64_t r0 = (tile * 8 + w) * RPW;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            off[rr] = 0; word[rr] = 0; len[rr] = 0;
            if (r0 + rr < R.n_reads) {
                read_span(R, r0 + rr, off[rr], len[rr]);
                word[rr] = stage_fetch<NW>(R, hint, nullptr, 0, 0, off[rr], lane);
            }
        }
    };
    if (blockIdx.x < n_tiles) fetch_tile(blockIdx.x);
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        unsigned long long rec[SLOTS];
        uint32_t rid[SLOTS];
        int rk[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) { rec[s] = 0; rid[s] = 0; rk[s] = -1; }
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int nk = (int)len[rr] - k + 1;
            if (nk > 0) {
                __builtin_amdgcn_wave_barrier();
                if (lane < S::WORDS) stage_store(L32, lane, word[rr]);
                __builtin_amdgcn_wave_barrier();
                const int o31 = (int)(off[rr] & 31), o63 = (int)(off[rr] & 63);
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    if (c * 64 < nk) {
                        const int s = c * 64 + lane;
                        const bool valid = (lds_window32(L32 + 2 * S::M, o63 + s) & K.nmask_bits) == 0;
                        const bool known = hint && lds_bit(L32 + 2 * S::H, o63 + s);
                        const bool ask = s < nk && valid && !known;
                        const uint64_t key = canon_key(lds_window64(L32 + 2 * S::B, 2 * (o31 + s)), K);
                        const uint32_t blk = block_of(F, key);
                        rec[rr * CH + c] = ((unsigned long long)blk << 16) | pattern_of(F, key);
                        rid[rr * CH + c] = gbase + (uint32_t)off[rr] + (uint32_t)s;
                        if (ask) rk[rr * CH + c] = (int)atomicAdd(&hist[blk >> L1_SHIFT], 1u);
                        looked += __popcll(__ballot(ask));
                    }
                }
            }
        }
        if (tile + gridDim.x < n_tiles) fetch_tile(tile + gridDim.x);
        __syncthreads();
        const uint32_t cnt = hist[threadIdx.x];
        hist[threadIdx.x] = 0;
        const uint32_t g = fill_l[threadIdx.x];
        fill_l[threadIdx.x] = g + cnt;
        uint32_t total;
        const uint32_t ex = block_scan512(cnt, wave_tot, &total);
        ofs_l[threadIdx.x] = ex;
        gbase_l[threadIdx.x] = g;
        __syncthreads();
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (rk[s] >= 0) {
                const uint32_t at = ofs_l[(uint32_t)(rec[s] >> (16 + L1_SHIFT))] + (uint32_t)rk[s];
                sorted[at] = rec[s];
                sorted_id[at] = rid[s];
            }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < total; i += BK_THREADS) {
            const unsigned long long v = sorted[i];
            const uint32_t id = sorted_id[i];
            const uint32_t b = (uint32_t)(v >> (16 + L1_SHIFT));
            const uint32_t pos = gbase_l[b] + (i - ofs_l[b]);
            if (pos < B.cap1) {
                const size_t at = ((size_t)src * B.nb1 + b) * B.cap1 + pos;
                B.l1[at] = v;
                B.l1id[at] = id;
            } else {
                if (!bloom_has(F, (uint32_t)(v >> 16), (uint32_t)(v & 0xFFFFu))) absent_set(B, id);
                ++direct;
            }
        }
    }
    if ((int)threadIdx.x < B.nb1) B.l1_cnt[(size_t)src * B.nb1 + threadIdx.x] = fill_l[threadIdx.x];
    if (lane == 0 && looked) atomicAdd(lookups, looked);      // (wave-uniform: every lane added the ballots' counts)
    if (direct) atomicAdd(B.direct, direct);
}

// ---- level 2 -------------------------------------------------------------------------------------------------
constexpr int LSPLIT_PER_THREAD = 8;
constexpr int LSPLIT_TILE = BK_THREADS * LSPLIT_PER_THREAD;

__global__ void __launch_bounds__(BK_THREADS) k_split_lookup(FiltDev F, LookupDev B, uint32_t chunks_per_region) {
    __shared__ uint32_t hist[NB2], ofs_l[NB2], gbase_l[NB2], wave_tot[8], unit_l;
    __shared__ unsigned long long sorted[LSPLIT_TILE];
    __shared__ uint16_t sorted_b[LSPLIT_TILE];
    const int home = xcc_id();
    unsigned long long direct = 0;
    hist[threadIdx.x] = 0;
    for (int qi = 0; qi < N_XCD; ++qi) {
        const int q = (home + qi) & (N_XCD - 1);
        const uint32_t n_buckets = B.nb1 > q ? (uint32_t)(B.nb1 - q + N_XCD - 1) / N_XCD : 0;
        const uint32_t n_units = n_buckets * (uint32_t)EMIT_GRID * chunks_per_region;
        for (;;) {
            __syncthreads();
            if (threadIdx.x == 0) unit_l = atomicAdd(&B.tickets[q * CNT_STRIDE], 1u);
            __syncthreads();
            const uint32_t unit = unit_l;
            if (unit >= n_units) break;
            const uint32_t chunk = unit % chunks_per_region, rest = unit / chunks_per_region;
            const int x = (int)(rest % (uint32_t)EMIT_GRID), b1 = q + (int)(rest / (uint32_t)EMIT_GRID) * N_XCD;
            const size_t region = (size_t)x * B.nb1 + b1;
            const uint32_t n = min(B.l1_cnt[region], B.cap1);
            const uint32_t first = chunk * (uint32_t)LSPLIT_TILE;
            if (first >= n) continue;
            const uint32_t m = min((uint32_t)LSPLIT_TILE, n - first);
            const unsigned long long *src = B.l1 + region * B.cap1 + first;
            const uint32_t *src_id = B.l1id + region * B.cap1 + first;
            unsigned long long rec[LSPLIT_PER_THREAD];
            int rk[LSPLIT_PER_THREAD];
#pragma unroll
            for (int j = 0; j < LSPLIT_PER_THREAD; ++j) {
                const uint32_t i = j * BK_THREADS + threadIdx.x;
                rk[j] = -1;
                rec[j] = 0;
                if (i < m) {
                    const unsigned long long v = src[i];
                    const uint32_t b2 = (uint32_t)(v >> (16 + SUB_BITS)) & (NB2 - 1);
                    rec[j] = ((v >> 16) & (unsigned long long)(SUB_BLOCKS - 1)) << 48 | (v & 0xFFFFull) << 32 | src_id[i];
                    rk[j] = (int)atomicAdd(&hist[b2], 1u) | (int)(b2 << 16);
                }
            }
            __syncthreads();
            const uint32_t cnt = hist[threadIdx.x];
            hist[threadIdx.x] = 0;
            uint32_t g = 0;
            if (cnt) g = atomicAdd(&B.l2_cnt[(size_t)b1 * NB2 + threadIdx.x], cnt);
            uint32_t total;
            const uint32_t ex = block_scan512(cnt, wave_tot, &total);
            ofs_l[threadIdx.x] = ex;
            gbase_l[threadIdx.x] = g;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < LSPLIT_PER_THREAD; ++j)
                if (rk[j] >= 0) {
                    const uint32_t b2 = (uint32_t)rk[j] >> 16, at = ofs_l[b2] + ((uint32_t)rk[j] & 0xFFFFu);
                    sorted[at] = rec[j];
                    sorted_b[at] = (uint16_t)b2;
                }
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < total; i += BK_THREADS) {
                const unsigned long long v = sorted[i];
                const uint32_t b2 = sorted_b[i];
                const uint32_t pos = gbase_l[b2] + (i - ofs_l[b2]);
                const size_t sub = (size_t)b1 * NB2 + b2;
                if (pos < B.cap2) {
                    B.l2[sub * B.cap2 + pos] = v;
                } else {
                    if (!bloom_has(F, (uint32_t)(sub << SUB_BITS) | (uint32_t)(v >> 48), (uint32_t)(v >> 32) & 0xFFFFu)) absent_set(B, (uint32_t)v);
                    ++direct;
                }
            }
        }
    }
    if (direct) atomicAdd(B.direct, direct);
}

// ---- level 3: the lookups ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(APPLY_THREADS) k_apply_lookup(FiltDev F, LookupDev B) {
    __shared__ ulonglong2 blk_l[SUB_BLOCKS];
    __shared__ uint32_t n_abs, base_abs;
    const uint32_t sub = blockIdx.x;
    const uint32_t n = min(B.l2_cnt[sub], B.cap2);
    if (n == 0) return;
    const uint64_t b0 = (uint64_t)sub << SUB_BITS;
    const uint32_t nblk = (uint32_t)min((uint64_t)SUB_BLOCKS, F.n_blocks - b0);
    for (uint32_t i = threadIdx.x; i < nblk; i += APPLY_THREADS) blk_l[i] = F.table[b0 + i];
    if (threadIdx.x == 0) n_abs = 0;
    __syncthreads();
    unsigned long long *region = B.l2 + (size_t)sub * B.cap2;
    uint32_t *out = reinterpret_cast<uint32_t *>(region);      // absent ids, compacted over the records already read
    const int lane = threadIdx.x & 63;
    for (uint32_t i0 = 0; i0 < n; i0 += APPLY_THREADS) {
        const uint32_t i = i0 + threadIdx.x;
        const unsigned long long v = i < n ? region[i] : 0ull;
        __syncthreads();      // every record of this round is in registers before an id overwrites its place
        bool absent = false;
        if (i < n) {
            const ulonglong2 p = F.patterns[(uint32_t)(v >> 32) & 0xFFFFu];
            const ulonglong2 t = blk_l[(uint32_t)(v >> 48)];
            absent = ((p.x & ~t.x) | (p.y & ~t.y)) != 0;
        }
        const unsigned long long bal = __ballot(absent);
        uint32_t wbase = 0;
        if (lane == 0 && bal) wbase = atomicAdd(&n_abs, (uint32_t)__popcll(bal));
        wbase = __shfl(wbase, 0);
        if (absent) out[wbase + __popcll(bal & ((1ULL << lane) - 1))] = (uint32_t)v;
    }
    __syncthreads();
    const uint32_t a = n_abs;
    if (a == 0) return;
    const int x = xcc_id();
    if (threadIdx.x == 0) base_abs = atomicAdd(&B.abs_cnt[x * CNT_STRIDE], a);
    __syncthreads();
    const uint32_t base = base_abs;
    for (uint32_t i = threadIdx.x; i < a; i += APPLY_THREADS) {
        const uint32_t id = out[i];
        if (base + i < B.cap_abs) B.abs_list[(size_t)x * B.cap_abs + base + i] = id;
        else absent_set(B, id);
    }
}

// ---- the way back: absent ids -> bits -----------------------------------------------------------------------------
constexpr int IDS_BIN_BITS = 22;                 // 4 Mi positions per bin = 512 KB of bits, four windows of 128 KB in LDS
constexpr int IDS_PER_THREAD = 8;
constexpr int IDS_TILE = BK_THREADS * IDS_PER_THREAD;
constexpr int IDS_GRID = 256;

struct IdsDev {
    uint32_t *regions;        // [IDS_GRID][n_bins][cap]
    uint32_t *cnt;            // [IDS_GRID][n_bins]
    uint32_t cap;
    int n_bins;               // <= 512
};

__global__ void __launch_bounds__(BK_THREADS) k_scatter_ids(LookupDev B, IdsDev D) {
    __shared__ uint32_t hist[512], ofs_l[512], gbase_l[512], fill_l[512], wave_tot[8];
    __shared__ uint32_t sorted[IDS_TILE];
    hist[threadIdx.x] = 0;
    fill_l[threadIdx.x] = 0;
    __syncthreads();
    uint32_t n_x[N_XCD], tiles_before[N_XCD + 1];      // the eight lists as one sequence of tiles
    tiles_before[0] = 0;
#pragma unroll
    for (int x = 0; x < N_XCD; ++x) {
        n_x[x] = min(B.abs_cnt[x * CNT_STRIDE], B.cap_abs);
        tiles_before[x + 1] = tiles_before[x] + (n_x[x] + IDS_TILE - 1) / IDS_TILE;
    }
    for (uint32_t tile = blockIdx.x; tile < tiles_before[N_XCD]; tile += gridDim.x) {
        int x = 0;
#pragma unroll
        for (int y = 1; y < N_XCD; ++y) x = tile >= tiles_before[y] ? y : x;
        const uint32_t first = (tile - tiles_before[x]) * IDS_TILE;
        const uint32_t m = min((uint32_t)IDS_TILE, n_x[x] - first);
        const uint32_t *src = B.abs_list + (size_t)x * B.cap_abs + first;
        uint32_t id[IDS_PER_THREAD];
        int rk[IDS_PER_THREAD];
#pragma unroll
        for (int j = 0; j < IDS_PER_THREAD; ++j) {
            const uint32_t i = j * BK_THREADS + threadIdx.x;
            rk[j] = -1;
            id[j] = 0;
            if (i < m) {
                id[j] = src[i];
                rk[j] = (int)atomicAdd(&hist[id[j] >> IDS_BIN_BITS], 1u);
            }
        }
        __syncthreads();
        const uint32_t cnt = hist[threadIdx.x];
        hist[threadIdx.x] = 0;
        const uint32_t g = fill_l[threadIdx.x];
        fill_l[threadIdx.x] = g + cnt;
        uint32_t total;
        const uint32_t ex = block_scan512(cnt, wave_tot, &total);
        ofs_l[threadIdx.x] = ex;
        gbase_l[threadIdx.x] = g;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < IDS_PER_THREAD; ++j)
            if (rk[j] >= 0) sorted[ofs_l[id[j] >> IDS_BIN_BITS] + (uint32_t)rk[j]] = id[j];
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < total; i += BK_THREADS) {
            const uint32_t v = sorted[i], b = v >> IDS_BIN_BITS;
            const uint32_t pos = gbase_l[b] + (i - ofs_l[b]);
            if (pos < D.cap) D.regions[((size_t)blockIdx.x * D.n_bins + b) * D.cap + pos] = v;
            else absent_set(B, v);
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < D.n_bins) D.cnt[(size_t)blockIdx.x * D.n_bins + threadIdx.x] = fill_l[threadIdx.x];
}

// one workgroup per window of 2^20 positions (128 KB of bits in LDS; four windows per bin): the bin's ids that fall into
// the window set their bits, the window is OR-ed into the batch's array (the overflow paths above may have set bits
// there already)
__global__ void __launch_bounds__(1024) k_ids_to_bits(LookupDev B, IdsDev D, uint32_t n_positions) {
    extern __shared__ uint32_t bits[];      // 32768 words
    constexpr uint32_t WIN = 1u << 20;
    const uint32_t bin = blockIdx.x >> 2;
    const uint32_t p0 = (bin << IDS_BIN_BITS) + (blockIdx.x & 3u) * WIN;
    if (p0 >= n_positions) return;
    for (uint32_t i = threadIdx.x; i < WIN / 32; i += 1024) bits[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int g = wv; g < IDS_GRID; g += 16) {      // one source region per wavefront at a time
        const uint32_t n = min(D.cnt[(size_t)g * D.n_bins + bin], D.cap);
        const uint32_t *src = D.regions + ((size_t)g * D.n_bins + bin) * D.cap;
        for (uint32_t i = lane; i < n; i += 64) {
            const uint32_t v = src[i] - p0;
            if (v < WIN) atomicOr(&bits[v >> 5], 1u << (v & 31));
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < WIN / 32; i += 1024) {
        const uint32_t v = bits[i];
        if (v) atomicOr(&B.absent_bits[(p0 >> 5) + i], v);
    }
}

}  // namespace kbbq
