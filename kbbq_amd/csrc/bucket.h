// bucket.h -- slice-bucketed Bloom inserts (gfx950): the inserts of passes 1 and 2 without one random
// 128-byte HBM line per k-mer.
//
// A Bloom insert is `table[block] |= pattern` at a uniformly random block of a multi-GB array: done directly
// (kernels.h: k_insert_marked) every insert fetches a whole 128-byte line of the L2 for its 16-byte block and,
// when the k-mer is new, sends two memory-side atomics after it.  Inserts have no result to return and only have
// to be complete before the NEXT pass reads the filter (pass 1 -> sampled filter -> pass 2; pass 2 -> trusted
// filter -> pass 3), so the engine defers them: the pass emits one 8-byte record (block, pattern) per insert,
// the records of many batches are partitioned by filter slice in two levels, and each 64 KB slice of the filter
// is then loaded into LDS ONCE, receives all its records through LDS atomics, and is written back once:
//
//   k_emit_marked   read -> (block, pattern) records, scattered into level-1 buckets (2^21 blocks = 32 MB of
//                   filter each; <= 512 of them) with an LDS counting sort per tile of a few thousand records
//   k_split         level-1 bucket -> its 512 subslices (2^12 blocks = 64 KB each), records shrink to 4 bytes
//   k_apply         subslice -> LDS, OR the patterns in (ds_or_b64), store the subslice
//
// HBM traffic per insert: 8 (emit) + 8 + 4 (split) + 4 (apply) bytes of streamed records + the filter itself
// read and written once per flush (a few bytes per insert once several 10^9 records are gathered), against
// 128 bytes of random line + atomics before.  Streaming writes of short runs are made whole lines by the L2:
// every XCD (each has its own L2) appends to regions of its own -- level 1: one region per (XCD, bucket);
// level 2: a bucket is split by workgroups of ONE XCD (work queues per XCD, drained by the others when a queue's
// own XCD is idle) -- so the partly written tail line of a region stays in one L2 until it is complete.
//
// Capacity: regions have fixed capacities sized for uniformly spread hashes; a record that does not fit (skewed
// input: the same k-mer millions of times) is inserted directly with the look-first atomic path, so results
// never depend on the capacities.  Results are bitwise ORs: order never matters.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.h"

namespace kbbq {

constexpr int SUB_BITS = 12;                     // blocks per subslice: 4096 x 16 B = 64 KB (two workgroups per CU)
constexpr int NB2_BITS = 9;                      // subslices per level-1 bucket
constexpr int SUB_BLOCKS = 1 << SUB_BITS;
constexpr int NB2 = 1 << NB2_BITS;
constexpr int L1_SHIFT = SUB_BITS + NB2_BITS;    // 21
constexpr int MAX_NB1 = 512;                     // filters up to 2^30 blocks (16 GiB in the engine's layout)
constexpr int N_XCD = 8;
constexpr int CNT_STRIDE = 32;                   // shared level-1 counters sit on 128-byte lines of their own
constexpr int BK_THREADS = 512;
constexpr int EMIT_GRID = 512;                   // workgroups of k_emit_marked = level-1 source regions per bucket (private mode)

struct BucketDev {
    unsigned long long *l1;   // [n_src][nb1][cap1] records: block << 16 | pattern
    uint32_t *l1_cnt;         // [n_src][nb1] x cnt_stride
    uint32_t *l2;             // [n_sub][cap2] records: block-in-subslice << 16 | pattern
    uint32_t *l2_cnt;         // [nb1 * NB2]
    uint32_t *tickets;        // [N_XCD] x CNT_STRIDE: work queues of k_split
    unsigned long long *direct;   // records that did not fit a region and were inserted directly (statistics)
    uint32_t cap1, cap2;
    int nb1;
    int n_src;                // source regions per level-1 bucket: EMIT_GRID (one per emitting workgroup) or N_XCD (one per XCD)
    int cnt_stride;           // 1 or CNT_STRIDE
    uint32_t n_sub;           // subslices that exist: ceil(n_blocks / SUB_BLOCKS)
};

__device__ __forceinline__ int xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    return (int)(x & (N_XCD - 1));
}

// exclusive scan of one value per thread over a 512-thread block (8 wavefronts); `wave_tot` is 8 words of LDS.
// Contains one barrier; returns the exclusive prefix, *total gets the block total.
__device__ __forceinline__ uint32_t block_scan512(uint32_t v, uint32_t *wave_tot, uint32_t *total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o); if (lane >= o) inc += y; }
    if (lane == 63) wave_tot[w] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const uint32_t t = wave_tot[i]; before += i < w ? t : 0; all += t; }
    *total = all;
    return before + inc - v;
}

// ---- level 1: emit -----------------------------------------------------------------------------------
// The read loop of k_insert_marked (one read per wavefront, staged in the wave's LDS slice, one lane per k-mer
// start), but a marked k-mer becomes a record instead of a read-modify-write of the table.  A tile = RPW reads
// per wavefront x 8 wavefronts (the packed words of all RPW reads are in flight at once, and the next tile's
// travel during this tile's sort); its records are counted per bucket while they are produced (LDS atomics give
// each its rank), every bucket's run gets its place in the bucket's region, the records are put in bucket order
// in LDS and copied out in runs.  PRIV: every workgroup appends to regions of its own (region = blockIdx.x), whose
// fill counters live in its LDS for the length of the launch -- no global atomic anywhere, at the price of 512
// partly written tail lines per bucket instead of eight (they are completed in the L2 / the Infinity Cache before
// they reach HBM).  !PRIV: one region per (XCD, bucket) and one global atomic per bucket and tile.
// CH = chunks of 64 k-mer starts a read can have (<= NW).
template <int NW, int CH, bool BY_BASE, int RPW, bool PRIV, bool DENSE = true>
__global__ void __launch_bounds__(BK_THREADS) k_emit_marked(ReadsDev R, KParams K, FiltDev F, BucketDev B, const uint64_t *mask,
                                                             uint64_t mask_words, const uint64_t *kofs,
                                                             unsigned long long *inserted) {
    using S = Stage<NW>;
    constexpr int SLOTS = RPW * CH;
    extern __shared__ unsigned long long sorted[];      // 8 * SLOTS * 64 records
    __shared__ uint32_t stage[8][DENSE ? RPW : 1][2 * S::WORDS];      // DENSE: all RPW reads of the wave staged at once
    __shared__ uint32_t hist[MAX_NB1], ofs_l[MAX_NB1], gbase_l[MAX_NB1], fill_l[MAX_NB1], wave_tot[8];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *L32 = stage[w][0];
    const int k = K.k;
    const int src = PRIV ? (int)blockIdx.x : xcc_id();
    const uint64_t reads_per_tile = 8 * RPW;
    const uint64_t n_tiles = (R.n_reads + reads_per_tile - 1) / reads_per_tile;
    unsigned long long mine = 0, direct = 0;
    hist[threadIdx.x] = 0;
    if (PRIV) fill_l[threadIdx.x] = (int)threadIdx.x < B.nb1 ? B.l1_cnt[(size_t)src * B.nb1 + threadIdx.x] : 0u;
    __syncthreads();
    uint64_t off[RPW], kb[RPW], word[RPW];
    uint32_t len[RPW];
    auto fetch_tile = [&](uint64_t tile) {
        const uint64_t r0 = (tile * 8 + w) * RPW;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            off[rr] = 0; kb[rr] = 0; word[rr] = 0; len[rr] = 0;
            if (r0 + rr < R.n_reads) {     // wave-uniform
                read_span(R, r0 + rr, off[rr], len[rr]);
                kb[rr] = BY_BASE ? off[rr] : kmer_base(kofs, r0 + rr, R.read_len, k);
                word[rr] = stage_fetch<NW>(R, nullptr, mask, kb[rr], mask_words - 1, off[rr], lane);
            }
        }
    };
    if (blockIdx.x < n_tiles) fetch_tile(blockIdx.x);
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        unsigned long long rec[SLOTS];
        int rk[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) { rec[s] = 0; rk[s] = -1; }
        // ---- produce and count
        if (DENSE) {
            // Only a part of the k-mer starts is marked (23 % in pass 1, 58 % in pass 2) and the hashes are most of this
            // kernel's instructions: first gather the 64-bit windows of the marked starts of all RPW reads in a dense
            // list (the wave's share of sorted[], which is idle until the tile's barrier), then hash full wavefronts.
            unsigned long long *wlist = sorted + (size_t)w * SLOTS * 64;
            int n_dense = 0;      // wave-uniform
            __builtin_amdgcn_wave_barrier();
            if (lane < S::WORDS) {
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr) stage_store(stage[w][rr], lane, word[rr]);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int nk = (int)len[rr] - k + 1;
                if (nk > 0) {
                    const uint32_t *L32 = stage[w][rr];      // (no barrier between the reads: their chunks are independent instruction streams)
                    const int o31 = (int)(off[rr] & 31), o63 = (int)(off[rr] & 63), x63 = (int)(kb[rr] & 63);
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        if (c * 64 < nk) {
                            const int s = c * 64 + lane;
                            const bool marked = lds_bit(L32 + 2 * S::X, x63 + s) != 0;
                            const bool valid = (lds_window32(L32 + 2 * S::M, o63 + s) & K.nmask_bits) == 0;
                            const bool take = s < nk && marked && (BY_BASE || valid);
                            const unsigned long long bal = __ballot(take);
                            if (take) wlist[n_dense + __popcll(bal & ((1ULL << lane) - 1))] = lds_window64(L32 + 2 * S::B, 2 * (o31 + s));
                            n_dense += (int)__popcll(bal);
                            if (!BY_BASE && R.hint_sampled) or_bits64(R.hint_sampled, off[rr] + (uint64_t)c * 64, bal, lane);
                            mine += __popcll(bal);
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < SLOTS; ++i) {
                if (i * 64 < n_dense && i * 64 + lane < n_dense) {
                    const uint64_t key = canon_key(wlist[i * 64 + lane], K);
                    const uint32_t blk = block_of(F, key);
                    rec[i] = ((unsigned long long)blk << 16) | pattern_of(F, key);
                    rk[i] = (int)atomicAdd(&hist[blk >> L1_SHIFT], 1u);
                }
            }
        } else {
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int nk = (int)len[rr] - k + 1;
            if (nk > 0) {     // wave-uniform (a read past the end of the batch has len 0)
                __builtin_amdgcn_wave_barrier();
                if (lane < S::WORDS) stage_store(L32, lane, word[rr]);
                __builtin_amdgcn_wave_barrier();
                const int o31 = (int)(off[rr] & 31), o63 = (int)(off[rr] & 63), x63 = (int)(kb[rr] & 63);
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    if (c * 64 < nk) {
                        // straight-line for every lane (the staged words cover the whole chunk): only the counter's
                        // atomic is predicated -- a lane's hash costs nothing when its neighbours need theirs anyway
                        const int s = c * 64 + lane;
                        const bool marked = lds_bit(L32 + 2 * S::X, x63 + s) != 0;
                        const bool valid = (lds_window32(L32 + 2 * S::M, o63 + s) & K.nmask_bits) == 0;
                        const bool take = s < nk && marked && (BY_BASE || valid);
                        const uint64_t key = canon_key(lds_window64(L32 + 2 * S::B, 2 * (o31 + s)), K);
                        const uint32_t blk = block_of(F, key);
                        rec[rr * CH + c] = ((unsigned long long)blk << 16) | pattern_of(F, key);
                        if (take) rk[rr * CH + c] = (int)atomicAdd(&hist[blk >> L1_SHIFT], 1u);
                        const unsigned long long bal = __ballot(take);
                        if (!BY_BASE && R.hint_sampled) or_bits64(R.hint_sampled, off[rr] + (uint64_t)c * 64, bal, lane);
                        mine += __popcll(bal);
                    }
                }
            }
        }
        }
        if (tile + gridDim.x < n_tiles) fetch_tile(tile + gridDim.x);      // travels while this tile is sorted and written
        __syncthreads();
        // ---- one run per bucket: its place in the bucket's region and in the tile
        const uint32_t cnt = hist[threadIdx.x];
        hist[threadIdx.x] = 0;
        uint32_t g = 0;
        if (PRIV) {
            g = fill_l[threadIdx.x];
            fill_l[threadIdx.x] = g + cnt;
        } else if (cnt) {
            g = atomicAdd(&B.l1_cnt[((size_t)src * B.nb1 + threadIdx.x) * CNT_STRIDE], cnt);
        }
        uint32_t total;
        const uint32_t ex = block_scan512(cnt, wave_tot, &total);
        ofs_l[threadIdx.x] = ex;
        gbase_l[threadIdx.x] = g;
        __syncthreads();
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (rk[s] >= 0) sorted[ofs_l[(uint32_t)(rec[s] >> (16 + L1_SHIFT))] + (uint32_t)rk[s]] = rec[s];
        __syncthreads();
        // ---- copy the runs out; what does not fit its region is inserted on the spot
        for (uint32_t i = threadIdx.x; i < total; i += BK_THREADS) {
            const unsigned long long v = sorted[i];
            const uint32_t b = (uint32_t)(v >> (16 + L1_SHIFT));
            const uint32_t pos = gbase_l[b] + (i - ofs_l[b]);
            if (pos < B.cap1) {
                B.l1[((size_t)src * B.nb1 + b) * B.cap1 + pos] = v;
            } else {
                bloom_put(F, (uint32_t)(v >> 16), (uint32_t)(v & 0xFFFFu));
                ++direct;
            }
        }
        // (the next tile's counting only touches hist[]; ofs_l/gbase_l/sorted are rewritten behind its barriers --
        // except the dense form's window lists, which live in sorted[])
        if (DENSE) __syncthreads();
    }
    if (PRIV && (int)threadIdx.x < B.nb1) B.l1_cnt[(size_t)src * B.nb1 + threadIdx.x] = fill_l[threadIdx.x];
    if (inserted && lane == 0 && mine) atomicAdd(inserted, mine);
    if (direct) atomicAdd(B.direct, direct);
}

// ---- level 2: split a level-1 bucket into its subslices ------------------------------------------------
// Work unit = (bucket b1, source region x of it, chunk of SPLIT_TILE records).  Bucket b1 belongs to the queue of
// XCD b1 % 8; a workgroup drains the queue of the XCD it runs on first and then the others', so every unit is
// processed whatever the placement of workgroups, and by one XCD when every XCD has workgroups (the usual case).
constexpr int SPLIT_PER_THREAD = 12;
constexpr int SPLIT_TILE = BK_THREADS * SPLIT_PER_THREAD;

__global__ void __launch_bounds__(BK_THREADS) k_split(FiltDev F, BucketDev B, uint32_t chunks_per_region) {
    __shared__ uint32_t hist[NB2], ofs_l[NB2], gbase_l[NB2], wave_tot[8], unit_l;
    __shared__ uint32_t sorted[SPLIT_TILE];
    __shared__ uint16_t sorted_b[SPLIT_TILE];
    const int home = xcc_id();
    unsigned long long direct = 0;
    hist[threadIdx.x] = 0;      // NB2 = 512 = one subslice counter per thread
    for (int qi = 0; qi < N_XCD; ++qi) {
        const int q = (home + qi) & (N_XCD - 1);
        const uint32_t n_buckets = B.nb1 > q ? (uint32_t)(B.nb1 - q + N_XCD - 1) / N_XCD : 0;     // b1 = q, q+8, ...
        const uint32_t n_units = n_buckets * (uint32_t)B.n_src * chunks_per_region;
        for (;;) {
            __syncthreads();      // (also: everyone is done with unit_l, sorted[] and the tables of the previous unit)
            if (threadIdx.x == 0) unit_l = atomicAdd(&B.tickets[q * CNT_STRIDE], 1u);
            __syncthreads();
            const uint32_t unit = unit_l;
            if (unit >= n_units) break;      // block-uniform: every wave of the block leaves together
            const uint32_t chunk = unit % chunks_per_region, rest = unit / chunks_per_region;
            const int x = (int)(rest % (uint32_t)B.n_src), b1 = q + (int)(rest / (uint32_t)B.n_src) * N_XCD;
            const size_t region = (size_t)x * B.nb1 + b1;
            const uint32_t n = min(B.l1_cnt[region * B.cnt_stride], B.cap1);
            const uint32_t first = chunk * (uint32_t)SPLIT_TILE;
            if (first >= n) continue;
            const uint32_t m = min((uint32_t)SPLIT_TILE, n - first);
            const unsigned long long *src = B.l1 + region * B.cap1 + first;
            uint32_t rec[SPLIT_PER_THREAD];
            int rk[SPLIT_PER_THREAD];
#pragma unroll
            for (int j = 0; j < SPLIT_PER_THREAD; ++j) {
                const uint32_t i = j * BK_THREADS + threadIdx.x;
                rk[j] = -1;
                rec[j] = 0;
                if (i < m) {
                    const unsigned long long v = src[i];
                    rec[j] = (uint32_t)v;      // low 32 bits: 16 of block (12 in the subslice + 4 of the subslice number), 16 of pattern
                    rk[j] = (int)atomicAdd(&hist[(uint32_t)(v >> (16 + SUB_BITS)) & (NB2 - 1)], 1u);
                    rk[j] |= (int)(((uint32_t)(v >> (16 + SUB_BITS)) & (NB2 - 1)) << 16);      // rank < SPLIT_TILE < 2^16
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
            for (int j = 0; j < SPLIT_PER_THREAD; ++j)
                if (rk[j] >= 0) {
                    const uint32_t b2 = (uint32_t)rk[j] >> 16, at = ofs_l[b2] + ((uint32_t)rk[j] & 0xFFFFu);
                    sorted[at] = rec[j] & 0x0FFFFFFFu;      // block-in-subslice << 16 | pattern
                    sorted_b[at] = (uint16_t)b2;
                }
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < total; i += BK_THREADS) {
                const uint32_t v = sorted[i], b2 = sorted_b[i];
                const uint32_t pos = gbase_l[b2] + (i - ofs_l[b2]);
                const size_t sub = (size_t)b1 * NB2 + b2;
                if (pos < B.cap2) {
                    B.l2[sub * B.cap2 + pos] = v;
                } else {
                    bloom_put(F, (uint32_t)(sub << SUB_BITS) | (v >> 16), v & 0xFFFFu);
                    ++direct;
                }
            }
        }
    }
    if (direct) atomicAdd(B.direct, direct);
}

// ---- level 3: apply ------------------------------------------------------------------------------------
// One workgroup per subslice: the 64 KB of the filter go to LDS with coalesced 16-byte loads, every record ORs
// its pattern in (look first: nine inserts in ten repeat an earlier one), the subslice is stored back.  The
// pattern table (1 MiB) is read from L2.  A subslice without records is neither loaded nor stored.
constexpr int APPLY_THREADS = 1024;

__global__ void __launch_bounds__(APPLY_THREADS) k_apply(FiltDev F, BucketDev B) {
    __shared__ ulonglong2 blk_l[SUB_BLOCKS];
    const uint32_t sub = blockIdx.x;
    const uint32_t n = min(B.l2_cnt[sub], B.cap2);
    if (n == 0) return;
    const uint64_t b0 = (uint64_t)sub << SUB_BITS;
    const uint32_t nblk = (uint32_t)min((uint64_t)SUB_BLOCKS, F.n_blocks - b0);
    const uint32_t *src = B.l2 + (size_t)sub * B.cap2;
    // Record -> pattern (L2) -> LDS is a chain of two loads per record, and a subslice has only ~20 records per lane: the
    // chain is kept two rounds deep -- the records of round r + 2 and the patterns of round r + 1 travel while round r is
    // applied; the first of them go out before the subslice itself is loaded.  The loads are unconditional (clamped
    // index: a lane past the end fetches the last record again and applies nothing), so that nothing waits for them early.
    const uint32_t last = n - 1;
    // three register sets take turns (round r uses set r mod 3; no register is copied: a copy of a loaded register waits
    // for the load): behind round r its set receives the records of round r + 3; in round r the set of round r + 1,
    // whose records were sent for two rounds ago, sends for its patterns
    uint32_t v[3][4];
    ulonglong2 p[3][4];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[s][j] = src[min(threadIdx.x + (4 * s + j) * APPLY_THREADS, last)];
#pragma unroll
    for (int j = 0; j < 4; ++j) p[0][j] = F.patterns[v[0][j] & 0xFFFFu];
    for (uint32_t i = threadIdx.x; i < nblk; i += APPLY_THREADS) blk_l[i] = F.table[b0 + i];
    __syncthreads();
    unsigned long long *w = reinterpret_cast<unsigned long long *>(blk_l);
    for (uint32_t i1 = threadIdx.x; i1 < n; i1 += 12 * APPLY_THREADS) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const uint32_t i0 = i1 + 4 * s * APPLY_THREADS;      // (rounds past the end apply nothing)
            const int s2 = (s + 1) % 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) p[s2][j] = F.patterns[v[s2][j] & 0xFFFFu];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (i0 + j * APPLY_THREADS < n) {
                    const uint32_t bi = v[s][j] >> 16;
                    const ulonglong2 t = blk_l[bi];
                    const unsigned long long mx = p[s][j].x & ~t.x, my = p[s][j].y & ~t.y;
                    if (mx) atomicOr(&w[2 * bi], mx);
                    if (my) atomicOr(&w[2 * bi + 1], my);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[s][j] = src[min(i0 + (12 + j) * APPLY_THREADS, last)];
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nblk; i += APPLY_THREADS) F.table[b0 + i] = blk_l[i];
}

}  // namespace kbbq
