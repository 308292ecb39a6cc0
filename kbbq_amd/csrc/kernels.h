// kernels.h -- every __global__ kernel of the engine (gfx950), included once by engine.hip.
//
//   pass 1  k_draw_mask        one xoshiro256** draw per k-mer position (htsiter.cc:113-129)
//           k_insert_marked    sampled & valid k-mers -> sampled filter   (recalibrateutils.cc:7-13)
//   pass 2  k_infer            infer_read_errors, which k-mers are trusted (recalibrateutils.cc:15-40)
//           k_insert_marked    those k-mers -> trusted filter
//   pass 3  k_scan_trusted     trusted mask of every k-mer; clean reads finish here, reads with up to four isolated
//                              errors in its fast path (fast_path)
//           k_compact          work list of reads that need more
//           k_correct_wave     get_errors, one read per wavefront (correct_wave.h); k_correct: one read per lane (correct.h)
//           k_tally            covariate histograms                      (covariateutils.cc:30-164,193-202)
//   pass 4  k_recalibrate      delta-Q apply                             (readutils.cc:572-595)
//   helpers k_kmer_counts, k_scan_tiles / k_scan_tile_sums / k_scan_add, k_read_index, k_rg_presence, k_or_words,
//           k_or_pieces, k_synth
// Integer / hash / bit work throughout: no MFMA.  What bounds passes 1-3 is HBM bandwidth at the L2's line
// granularity: every Bloom lookup fetches one random 128-byte line for its 16-byte block.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/kbbq_engine.h"
#include "correct.h"
#include "correct_wave.h"
#include "device_common.h"

using namespace kbbq;

// ============================================================ kernels

// ---- pass 1a: the sampler's draw stream ------------------------------------
// The reference draws serially (one std::bernoulli_distribution call per k-mer
// position, htsiter.cc:113-129).  xoshiro256 is linear over GF(2), so the state
// after n draws is (x^n mod P)(M) applied to the seed state; each lane jumps to
// its own chunk of DRAWS_PER_LANE consecutive draws and emits a bit mask.
constexpr int DRAWS_PER_LANE = 4096;   // multiple of 64: each lane owns whole mask words

__constant__ uint64_t c_jump[64][4];

__device__ __forceinline__ uint64_t xo_next(uint64_t (&s)[4]) {
    uint64_t x = s[1] * 5;
    x = (x << 7) | (x >> 57);
    x *= 9;
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = (s[3] << 45) | (s[3] >> 19);
    return x;
}

__global__ void __launch_bounds__(256) k_draw_mask(uint64_t s0, uint64_t s1, uint64_t s2, uint64_t s3,
                                                    uint64_t first_ordinal, uint64_t n_draws, uint64_t threshold,
                                                    int always, uint64_t *mask) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t begin = t * DRAWS_PER_LANE;
    if (begin >= n_draws) return;
    uint64_t s[4] = {s0, s1, s2, s3};
    const uint64_t ordinal = first_ordinal + begin;
    for (int b = 0; b < 64; ++b) {
        if (!((ordinal >> b) & 1)) continue;
        uint64_t acc[4] = {0, 0, 0, 0};
        for (int w = 0; w < 4; ++w) {
            const uint64_t poly = c_jump[b][w];
            for (int i = 0; i < 64; ++i) {
                const uint64_t m = 0 - ((poly >> i) & 1);
                acc[0] ^= s[0] & m; acc[1] ^= s[1] & m; acc[2] ^= s[2] & m; acc[3] ^= s[3] & m;
                xo_next(s);
            }
        }
        s[0] = acc[0]; s[1] = acc[1]; s[2] = acc[2]; s[3] = acc[3];
    }
    const uint64_t end = min(begin + (uint64_t)DRAWS_PER_LANE, n_draws);
    uint64_t *out = mask + begin / 64;
    for (uint64_t o = begin; o < end; o += 64) {
        uint64_t bits = 0;
        const int cnt = (int)min((uint64_t)64, end - o);
        for (int i = 0; i < cnt; ++i) {
            const uint64_t u = xo_next(s);
            bits |= (uint64_t)((always || u < threshold) ? 1 : 0) << i;
        }
        *out++ = bits;
    }
}

// exclusive prefix of max(0, len-k+1) over the reads of a ragged batch
__global__ void k_kmer_counts(ReadsDev R, int k, uint64_t *counts) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R.n_reads) return;
    uint64_t off; uint32_t len;
    read_span(R, r, off, len);
    counts[r] = len >= (uint32_t)k ? len - k + 1 : 0;
}
// Exclusive scan in three launches (ragged batches: every command-line batch has offsets): tiles of 2048 counts
// are scanned in LDS and leave their total; one block scans the tile totals; the tile offsets are added back.
constexpr int SCAN_TILE = 2048;
__global__ void __launch_bounds__(256) k_scan_tiles(uint64_t *data, uint64_t n, uint64_t *tile_sums) {
    __shared__ uint64_t wave_tot[4];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * 8;
    uint64_t v[8], run = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = base + j < n ? data[base + j] : 0; const uint64_t x = v[j]; v[j] = run; run += x; }
    // exclusive scan of the 256 thread totals: inside the wave by shuffles, across the four waves through LDS
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t inc = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint64_t y = __shfl_up(inc, o); if (lane >= o) inc += y; }
    if (lane == 63) wave_tot[w] = inc;
    __syncthreads();
    uint64_t before = inc - run;
    for (int i = 0; i < w; ++i) before += wave_tot[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) if (base + j < n) data[base + j] = v[j] + before;
    if (threadIdx.x == 255) tile_sums[blockIdx.x] = before + run;
}
__global__ void __launch_bounds__(1024) k_scan_tile_sums(uint64_t *tile_sums, uint64_t n_tiles, uint64_t *total) {
    __shared__ uint64_t part[1024];
    const int tid = threadIdx.x;
    const uint64_t per = (n_tiles + 1023) / 1024;
    const uint64_t b = min(n_tiles, per * tid), e = min(n_tiles, b + per);
    uint64_t s = 0;
    for (uint64_t i = b; i < e; ++i) s += tile_sums[i];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        uint64_t run = 0;
        for (int i = 0; i < 1024; ++i) { const uint64_t v = part[i]; part[i] = run; run += v; }
        *total = run;
    }
    __syncthreads();
    uint64_t run = part[tid];
    for (uint64_t i = b; i < e; ++i) { const uint64_t v = tile_sums[i]; tile_sums[i] = run; run += v; }
}
__global__ void __launch_bounds__(256) k_scan_add(uint64_t *data, uint64_t n, const uint64_t *tile_sums) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) data[i] += tile_sums[i / SCAN_TILE];
}

// Which read holds base m*1024 (ragged batches): entry m of a coarse index, so that the kernels that walk the
// batch by base (tally, apply) start their search one load away from the answer instead of bisecting n_reads offsets.
constexpr int READ_INDEX_STEP = 1024;
__global__ void k_read_index(const uint64_t *offsets, uint64_t n_reads, uint32_t *index) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t a = offsets[r], b = offsets[r + 1];
    for (uint64_t m = (a + READ_INDEX_STEP - 1) / READ_INDEX_STEP; m * READ_INDEX_STEP < b; ++m) index[m] = (uint32_t)r;
}
// the read that contains base g0 (skipping empty reads), from the coarse index
__device__ __forceinline__ void find_read(const ReadsDev &R, const uint32_t *index, uint64_t g0, uint64_t &r, uint64_t &start, uint64_t &end) {
    if (index) {
        r = index[g0 / READ_INDEX_STEP];
    } else {
        uint64_t lo = 0, hi = R.n_reads;   // last r with offsets[r] <= g0
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (R.offsets[mid] <= g0) lo = mid; else hi = mid;
        }
        r = lo;
    }
    start = R.offsets[r];
    end = R.offsets[r + 1];
    while (end <= g0) { ++r; start = end; end = R.offsets[r + 1]; }
}

__device__ __forceinline__ uint64_t kmer_base(const uint64_t *kofs, uint64_t r, uint32_t read_len, int k) {
    if (kofs) return kofs[r];
    return r * (uint64_t)(read_len >= (uint32_t)k ? read_len - k + 1 : 0);
}

// ---- passes 1b and 2b: insert the marked k-mers of every read into a filter --------------
// One wavefront per read, one lane per k-mer start, one 128-bit block per lane.  BY_BASE = false:
// `mask` is the sampler's draw mask, one bit per k-mer position in file order (pass 1: insert where
// drawn and valid, and record that in hint_sampled).  BY_BASE = true: `mask` has one bit per base of
// the batch and already means "insert the k-mer starting here" (pass 2: the decisions of k_infer).
template <int NW, bool BY_BASE>
__global__ void __launch_bounds__(256) k_insert_marked(ReadsDev R, KParams K, FiltDev F, const uint64_t *mask,
                                                        uint64_t mask_words, const uint64_t *kofs,
                                                        unsigned long long *inserted) {
    using S = Stage<NW>;
    __shared__ uint32_t lds[4][2 * S::WORDS];
    const int lane = threadIdx.x & 63;
    uint32_t *L32 = lds[threadIdx.x >> 6];
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const int k = K.k;
    unsigned long long mine = 0;
    uint64_t off = 0, kb = 0, word = 0;
    uint32_t len = 0;
    if (wave < R.n_reads) {
        read_span(R, wave, off, len);
        kb = BY_BASE ? off : kmer_base(kofs, wave, R.read_len, k);
        word = stage_fetch<NW>(R, nullptr, mask, kb, mask_words - 1, off, lane);
    }
    for (uint64_t r = wave; r < R.n_reads; r += n_waves) {
        __builtin_amdgcn_wave_barrier();
        if (lane < S::WORDS) stage_store(L32, lane, word);
        __builtin_amdgcn_wave_barrier();
        const uint64_t cur = off;
        const int o31 = (int)(off & 31), o63 = (int)(off & 63), x63 = (int)(kb & 63);
        const int nk = (int)len - k + 1;
        if (r + n_waves < R.n_reads) {   // the next read's words travel while this one is processed
            read_span(R, r + n_waves, off, len);
            kb = BY_BASE ? off : kmer_base(kofs, r + n_waves, R.read_len, k);
            word = stage_fetch<NW>(R, nullptr, mask, kb, mask_words - 1, off, lane);
        }
        if (nk <= 0) continue;
        // all block and pattern loads of the read are issued before the first is used
        bool take[NW];
        uint32_t blk[NW];
        ulonglong2 t[NW], p[NW];
#pragma unroll
        for (int c = 0; c < NW; ++c) {
            take[c] = false;
            blk[c] = 0;
            const int s = c * 64 + lane;
            if (c * 64 < nk) {
                if (s < nk && lds_bit(L32 + 2 * S::X, x63 + s)) {
                    const bool valid = (lds_window32(L32 + 2 * S::M, o63 + s) & K.nmask_bits) == 0;
                    take[c] = BY_BASE || valid;
                }
                if (take[c]) {
                    const uint64_t key = canon_key(lds_window64(L32 + 2 * S::B, 2 * (o31 + s)), K);
                    blk[c] = block_of(F, key);
                    t[c] = F.table[blk[c]];
                    p[c] = F.patterns[pattern_of(F, key)];
                }
                const unsigned long long bal = __ballot(take[c]);
                if (!BY_BASE && R.hint_sampled) or_bits64(R.hint_sampled, cur + (uint64_t)c * 64, bal, lane);
                mine += __popcll(bal);
            }
        }
#pragma unroll
        for (int c = 0; c < NW; ++c)
            if (take[c]) {
                const uint64_t mx = p[c].x & ~t[c].x, my = p[c].y & ~t[c].y;
                unsigned long long *w = reinterpret_cast<unsigned long long *>(F.table + blk[c]);
                if (mx) atomicOr(w, (unsigned long long)mx);
                if (my) atomicOr(w + 1, (unsigned long long)my);
            }
    }
    if (inserted && lane == 0 && mine) atomicAdd(inserted, mine);
}

// ---- pass 2 ------------------------------------------------------------------
// overlapping_kmers_in_bf (bloom.cc:28-67) + infer_read_errors (readutils.cc:173-193)
// + the trusted-insert loop of find_trusted_kmers (recalibrateutils.cc:26-38).
// Wave per read.  present[] and err[] live as wave-uniform bit words produced
// by __ballot; the sliding counters in[i]/possible[i] and the "k clean bases"
// window become range pop-counts on those words.
struct Thresholds { int v[KBBQ_MAX_KMER + 1]; };

// MINW: waves per SIMD the register allocation must leave room for (the kernel waits on random HBM lines three
// quarters of the time: more resident waves keep more lookups in flight)
// NK: 64-lane chunks that can hold k-mer starts (the batch's longest read has at most NK * 64 of them; NK <= NW): their
// loads are unconditional, so that they sit in one basic block and all go out together.
//
// SUB (round 4): decide from a SUBSET of the lookups first.  err[i] only asks whether in[i] exceeds thr[possible[i]], and
// the k-mers known to be present -- hint bits plus the lookups made so far -- bound in[i] from below, the skipped ones
// from above: a base with lower bound > thr is clean, one with lower + skipped <= thr is flagged, whatever the skipped
// lookups would say.  Phase 1 skips every fourth k-mer start ((s & pmask) == pmask; every 8th or 16th when the
// thresholds sit closer to k) at least `edge` starts away from both read ends (near the ends thr[possible] leaves no
// room: the host derives period and `edge` from the thresholds, infer_subset_plan);
// phase 2 makes the skipped lookups whose k bases hold an undecided base -- around real errors, where in[i] crosses
// the threshold -- and only if the read has any.  Every base ends up with the exact decision: for a decided base the
// final count moves inside the bounds that decided it, for an undecided one every start in its window has been looked
// up.  take_bits, err_out, qpresent and the insert count are bit-identical to the plain form (tests/test_parity_gpu.py
// runs both); `lookups` counts the lines really fetched.
template <int NW, int NK, int MINW = 1, bool SUB = false>
__global__ void __launch_bounds__(256, MINW) k_infer(ReadsDev R, KParams K, FiltDev S, Thresholds thr, uint32_t *take_bits,
                                                unsigned long long *inserted, uint32_t *err_out, uint32_t *qpresent,
                                                unsigned long long *lookups, unsigned int *ticket, int edge, int pmask) {
    using St = Stage<NW>;
    __shared__ uint32_t lds[4][St::LDS_U32];
    __shared__ uint32_t lds_sub[SUB ? 4 : 1][SUB ? 2 * St::RES : 1];      // skipped starts, undecided bases (same shape as PW)
    __shared__ int thr_lds[KBBQ_MAX_KMER + 1];
    __shared__ uint32_t qseen[8];                  // quality values this block has met (256 bits), flushed once at the end
    const int lane = threadIdx.x & 63;
    if (threadIdx.x <= KBBQ_MAX_KMER) thr_lds[threadIdx.x] = thr.v[threadIdx.x];
    if (threadIdx.x < 8) qseen[threadIdx.x] = 0;
    __syncthreads();
    uint32_t *L32 = lds[threadIdx.x >> 6];
    uint32_t *PW = L32 + 2 * St::WORDS;            // present bits: dword 0 = 0, dwords 1..2NW, then zeros
    uint32_t *EW = PW + St::RES;                   // error bits, same shape
    const uint64_t *hint = reinterpret_cast<const uint64_t *>(R.hint_sampled);
    const int k = K.k;
    unsigned long long mine = 0, looked = 0;
    if (lane < St::RES) { PW[lane] = 0; EW[lane] = 0; }
    uint32_t *SW = lds_sub[SUB ? (threadIdx.x >> 6) : 0];      // SUB: the starts phase 1 skipped
    uint32_t *UW = SW + (SUB ? St::RES : 0);                    //      the bases phase 1 left undecided
    if (SUB && lane < St::RES) { SW[lane] = 0; UW[lane] = 0; }
    uint64_t off = 0, word = 0;
    uint32_t len = 0;
    ReadChunks<32> Q;      // reads in chunks of 32 from a global counter: no idle last round (device_common.h)
    uint64_t r = Q.begin(ticket, R.n_reads, lane);
    if (r < R.n_reads) {
        read_span(R, r, off, len);
        word = stage_fetch<NW>(R, hint, nullptr, 0, 0, off, lane);
    }
    for (; r < R.n_reads; Q.advance(lane), r = Q.cur) {
        __builtin_amdgcn_wave_barrier();
        if (lane < St::WORDS) stage_store(L32, lane, word);
        __builtin_amdgcn_wave_barrier();
        const uint64_t cur = off;
        const int o31 = (int)(off & 31), o63 = (int)(off & 63);
        const int Lr = (int)len, nk = Lr - k + 1;
        const uint64_t r_next = Q.peek(lane);
        if (r_next < R.n_reads) {   // the next read's words travel while this one is processed
            read_span(R, r_next, off, len);
            word = stage_fetch<NW>(R, hint, nullptr, 0, 0, off, lane);
        }
        if (nk <= 0) {
            // engine-defined: the reference underflows size_t here.  The read is still tallied and recalibrated, so its
            // quality values belong to the set pass 3 sizes its tables by
            for (int i = lane; i < Lr; i += 64) qseen_note(qseen, R.qual[cur + i]);
            continue;
        }
        // Step A (ALU and LDS only): which block and pattern every lane wants.  A lane with nothing to look up -- past the
        // last k-mer, a k-mer with a non-ACGT base, one this read put into the sampled filter itself in pass 1 (hint
        // bit) -- points at block 0 / pattern 0: one shared, cache-resident line per instruction, no branch around the load.
        bool valid[NK], known[NK], skip[NK];
        uint32_t blk[NK], pat[NK];
        uint32_t blk2[SUB ? NK : 1], pat2[SUB ? NK : 1];      // SUB: where the skipped starts would look (phase 2)
#pragma unroll
        for (int c = 0; c < NK; ++c) {
            valid[c] = false;
            known[c] = false;
            skip[c] = false;
            blk[c] = 0;
            pat[c] = 0;
            if (SUB) { blk2[c] = 0; pat2[c] = 0; }
            const int s = c * 64 + lane;
            if (s < nk) {
                valid[c] = (lds_window32(L32 + 2 * St::M, o63 + s) & K.nmask_bits) == 0;
                known[c] = hint && lds_bit(L32 + 2 * St::H, o63 + s);
                const uint64_t key = canon_key(lds_window64(L32 + 2 * St::B, 2 * (o31 + s)), K);
                if (valid[c] && !known[c]) {
                    const uint32_t b = block_of(S, key), pt = pattern_of(S, key);
                    // (selects, not branches: with `if (skip) .. else ..` hipcc 7.2 left blk[c] unassigned on the lanes
                    // whose first two terms hold and whose last does not -- found by the parity test, seen in the ISA)
                    const bool sk = SUB & ((s & pmask) == pmask) & (s >= edge) & (s < nk - edge);
                    skip[c] = sk;
                    blk[c] = sk ? 0u : b;
                    pat[c] = sk ? 0u : pt;
                    if (SUB) { blk2[c] = sk ? b : 0u; pat2[c] = sk ? pt : 0u; }
                }
            }
        }
        // Step B: every load of the read goes out back to back -- quality bytes (streamed), blocks (one random HBM line
        // each), patterns (1 MiB table, L2) -- before the first result is looked at: NK independent line fetches in
        // flight per lane instead of one.  (Until round 3 the compiler sank the first use of a block right behind its
        // load, and the quality byte fed a dependent global load + atomic in front of it: one fetch in flight per wave.)
        uint8_t q[NW];
        ulonglong2 t[NK], p[NK];
#pragma unroll
        for (int c = 0; c < NW; ++c) {
            const int s = c * 64 + lane;
            q[c] = (c * 64 < Lr && s < Lr) ? R.qual[cur + s] : (uint8_t)0;
        }
#pragma unroll
        for (int c = 0; c < NK; ++c) t[c] = S.table[blk[c]];
#pragma unroll
        for (int c = 0; c < NK; ++c) p[c] = S.patterns[pat[c]];
        __builtin_amdgcn_sched_barrier(0);
        uint64_t V[NK], P[NK];
        uint64_t any_skip = 0;
#pragma unroll
        for (int c = 0; c < NK; ++c) {
            const bool need = valid[c] && !known[c] && !skip[c];
            const bool present = known[c] || (need && ((p[c].x & ~t[c].x) | (p[c].y & ~t[c].y)) == 0);
            P[c] = __ballot(present);
            V[c] = __ballot(valid[c]);
            looked += __popcll(__ballot(need));      // blocks actually wanted (reported, not used)
            if (lane < 2) PW[1 + 2 * c + lane] = (uint32_t)(P[c] >> (32 * lane));
            if (SUB) {
                const uint64_t SKb = __ballot(skip[c]);
                any_skip |= SKb;
                if (lane < 2) SW[1 + 2 * c + lane] = (uint32_t)(SKb >> (32 * lane));
            }
        }
#pragma unroll
        for (int c = NK; c < NW; ++c)
            if (lane < 2) { PW[1 + 2 * c + lane] = 0; if (SUB) SW[1 + 2 * c + lane] = 0; }
        // which quality values occur at all (the tally of pass 3 sizes its LDS tables by them): a look at the block's
        // 256-bit set in LDS, an LDS atomic the first time
#pragma unroll
        for (int c = 0; c < NW; ++c)
            if (c * 64 < Lr && c * 64 + lane < Lr) qseen_note(qseen, q[c]);
        __builtin_amdgcn_wave_barrier();
        if (SUB && any_skip) {
            // Phase 1's verdict per base: lower bound (PW) and skipped starts (SW) of its window against the threshold
            uint64_t any_und = 0;
#pragma unroll
            for (int c = 0; c < NW; ++c) {
                if (c * 64 < Lr) {
                    const int i = c * 64 + lane;
                    bool und = false;
                    if (i < Lr && q[c] > 2) {
                        const int possible = min(i, nk - 1) - max(0, i - k + 1) + 1;
                        const int lb = __popc(lds_window32(PW, i - k + 1 + 32) & K.nmask_bits);
                        const int nsk = __popc(lds_window32(SW, i - k + 1 + 32) & K.nmask_bits);
                        const int th = thr_lds[possible];
                        und = lb <= th && lb + nsk > th;
                    }
                    const uint64_t U = __ballot(und);
                    any_und |= U;
                    if (lane < 2) UW[1 + 2 * c + lane] = (uint32_t)(U >> (32 * lane));
                } else if (lane < 2) {
                    UW[1 + 2 * c + lane] = 0;
                }
            }
            if (any_und) {
                // Phase 2: the skipped starts whose k bases hold an undecided base (a second round of line fetches, only
                // around the places where in[i] crosses its threshold)
                __builtin_amdgcn_wave_barrier();
                bool need2[NK];
#pragma unroll
                for (int c = 0; c < NK; ++c) {
                    const int s = c * 64 + lane;
                    need2[c] = skip[c] && (lds_window32(UW, s + 32) & K.nmask_bits) != 0;
                    if (!need2[c]) { blk2[c] = 0; pat2[c] = 0; }
                }
#pragma unroll
                for (int c = 0; c < NK; ++c) t[c] = S.table[blk2[c]];
#pragma unroll
                for (int c = 0; c < NK; ++c) p[c] = S.patterns[pat2[c]];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < NK; ++c) {
                    const bool present2 = need2[c] && ((p[c].x & ~t[c].x) | (p[c].y & ~t[c].y)) == 0;
                    P[c] |= __ballot(present2);
                    looked += __popcll(__ballot(need2[c]));
                    if (lane < 2) PW[1 + 2 * c + lane] = (uint32_t)(P[c] >> (32 * lane));
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        // in[i] = present k-mers among the starts max(0,i-k+1)..min(i,nk-1): the k bits of the present
        // stream that end at bit i (bits before 0 and from nk on are zero)
#pragma unroll
        for (int c = 0; c < NW; ++c) {
            if (c * 64 < Lr) {
                const int i = c * 64 + lane;
                bool err = false;
                if (i < Lr) {
                    const int possible = min(i, nk - 1) - max(0, i - k + 1) + 1;
                    const int in = __popc(lds_window32(PW, i - k + 1 + 32) & K.nmask_bits);
                    err = in <= thr_lds[possible] || q[c] <= 2;
                }
                const uint64_t E = __ballot(err);
                if (lane < 2) EW[1 + 2 * c + lane] = (uint32_t)(E >> (32 * lane));
                if (err_out) or_bits64(err_out, cur + (uint64_t)c * 64, E, lane);
            } else if (lane < 2) {
                EW[1 + 2 * c + lane] = 0;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // the k-mer ending at i goes into the trusted filter iff it is valid and its k bases are all
        // unflagged (recalibrateutils.cc:26-38); the inserts themselves are k_insert_marked's
#pragma unroll
        for (int c = 0; c < NK; ++c) {
            if (c * 64 < nk) {
                const int s = c * 64 + lane;
                const bool take = s < nk && ((V[c] >> lane) & 1) && (lds_window32(EW, s + 32) & K.nmask_bits) == 0;
                const unsigned long long bal = __ballot(take);
                or_bits64(take_bits, cur + (uint64_t)c * 64, bal, lane);
                mine += __popcll(bal);
            }
        }
    }
    if (lane == 0 && mine) atomicAdd(inserted, mine);
    if (lane == 0 && looked) atomicAdd(lookups, looked);
    __syncthreads();
    if (threadIdx.x < 8 && qseen[threadIdx.x]) atomicOr(&qpresent[threadIdx.x], qseen[threadIdx.x]);
}

// ---- pass 3a': the isolated-error fast path (inside k_scan_trusted) -------------------
// Most reads that need work carry a few ISOLATED errors: their trusted mask has up to four separate
// runs of untrusted k-mers, each run being exactly the k-mers that cover one base p.  For such a
// read get_errors (readutils.cc:238-570) reduces to: the anchor is the longest trusted run; walking
// away from it, find_longest_fix meets the errors one at a time; if exactly one alternative base at
// p makes every k-mer covering p trusted, that alternative alone has the longest walk, it is
// applied, and the walk continues through the trusted k-mers to the next run (or the read end).  At
// most four flags are set, so the over-correction window cannot fire and nothing is left for the
// recursion: errors = {p1..pm}.  The scan kernel decides that with two small rounds of lookups per
// read, while the read is still staged in its LDS slice, and marks the read done (dirty = 2); if any
// run has no or several full alternatives, or the mask has any other shape, the read goes to k_correct_wave.
// Z = the read's untrusted k-mer starts (complement of the trusted mask inside [0, nk)), `zeros` their number;
// the read's words are staged in the wave's LDS slice L32 (Stage<NW>).  Returns true when the read is
// settled (its flags are then OR-ed into err_bits); q_total counts the lookups.
template <int NW>
__device__ __forceinline__ bool fast_path(const uint32_t *L32, const KParams &K, const FiltDev &T, const uint64_t (&Z)[NW], int zeros,
                                          uint64_t off, int nk, int o31, int o63, uint32_t *err_bits, int lane,
                                          unsigned long long &q_total) {
    using S = Stage<NW>;
    constexpr int MAXRUN = 4;
    const int k = K.k;
    auto next_bit = [&](int from, bool one) -> int {      // first index >= from with Z bit == one, nk if none
        while (from < nk) {
            uint64_t x = sel_word<NW>(Z, from >> 6);
            if (!one) x = ~x;
            x >>= (from & 63);
            if (x) { const int q = from + __ffsll((unsigned long long)x) - 1; return q < nk ? q : nk; }
            from = ((from >> 6) + 1) << 6;
        }
        return nk;
    };
    int m = 0, z0s[MAXRUN], z1s[MAXRUN], ps[MAXRUN];
    bool eligible = zeros > 0 && zeros < nk;
    for (int pos = next_bit(0, true); eligible && pos < nk;) {
        const int z0 = pos, z1 = next_bit(pos, false) - 1;
        if (m == MAXRUN || z1 - z0 + 1 > k) { eligible = false; break; }
        const int p = z0 > 0 ? z0 + k - 1 : z1;
        // the run must be exactly the in-range starts that cover p
        if (!(z0 == max(0, p - k + 1) && z1 == min(p, nk - 1))) { eligible = false; break; }
#pragma unroll
        for (int q = 0; q < MAXRUN; ++q) if (q == m) { z0s[q] = z0; z1s[q] = z1; ps[q] = p; }
        ++m;
        pos = next_bit(z1 + 1, true);
    }
    int singles = 0;   // runs with exactly one full alternative
    if (eligible) {
        auto pick = [&](const int (&a)[MAXRUN], int idx) -> int {
            int v = a[0];
#pragma unroll
            for (int q = 1; q < MAXRUN; ++q) v = (idx == q) ? a[q] : v;
            return v;
        };
        // k-mer starting at st with base p := y
        auto key_of = [&](int st, int p, int y, bool &valid) -> uint64_t {
            uint64_t w = lds_window64(L32 + 2 * S::B, 2 * (o31 + st));
            uint32_t nm = lds_window32(L32 + 2 * S::M, o63 + st) & K.nmask_bits;
            const int j = p - st;
            w = (w & ~(3ULL << (2 * j))) | ((uint64_t)y << (2 * j));
            nm &= ~(1u << j);
            valid = nm == 0;
            return canon_key(w, K);
        };
        // round 1: one covering k-mer per (run, alternative): lane 4*run + y
        int alive;
        {
            const int run = lane >> 2, y = lane & 3;
            bool act = lane < 4 * m;
            int p = 0, st = 0;
            if (act) {
                p = pick(ps, run);
                st = pick(z0s, run);
                const int cur = lds_bit(L32 + 2 * S::M, o63 + p) ? 4 : (int)((L32[2 * S::B + ((o31 + p) >> 4)] >> (((o31 + p) & 15) * 2)) & 3);
                act = y != cur;
            }
            bool valid = false;
            const uint64_t key = key_of(st, p, y, valid);
            const bool go = act && valid;
            q_total += __popcll(__ballot(go));
            const bool t = go && bloom_has(T, key);
            alive = (int)(__ballot(t) & 0xFFFF);
        }
        // round 2: every covering k-mer of the survivors, two (run, alternative) pairs per lookup
        int full = 0;   // 3 bits per run: number of alternatives whose covering k-mers are all trusted
        while (alive) {
            const int a = __ffs(alive) - 1;
            alive &= alive - 1;
            int b = -1;
            if (alive) { b = __ffs(alive) - 1; alive &= alive - 1; }
            const int mine = (lane >> 5) ? b : a;
            const int run = mine >= 0 ? mine >> 2 : 0, y = mine >= 0 ? (mine & 3) : 0;
            const int z0 = pick(z0s, run), z1 = pick(z1s, run), p = pick(ps, run);
            const int st = z0 + (lane & 31);
            const bool act = mine >= 0 && st <= z1;
            bool valid = false;
            const uint64_t key = key_of(act ? st : z0, p, y, valid);
            const bool go = act && valid;
            q_total += __popcll(__ballot(go));
            const bool t = go && bloom_has(T, key);
            const unsigned long long bal = __ballot(t);
            const int ra = a >> 2;
            if (__popcll(bal & 0xFFFFFFFFULL) == pick(z1s, ra) - pick(z0s, ra) + 1) full += 1 << (3 * ra);
            if (b >= 0) {
                const int rb = b >> 2;
                if (__popcll(bal >> 32) == pick(z1s, rb) - pick(z0s, rb) + 1) full += 1 << (3 * rb);
            }
        }
        for (int q = 0; q < m; ++q) singles += ((full >> (3 * q)) & 7) == 1 ? 1 : 0;
    }
    const bool fixed = eligible && singles == m;
    if (fixed && lane < m) {
        const uint64_t g = off + (lane == 0 ? ps[0] : lane == 1 ? ps[1] : lane == 2 ? ps[2] : ps[3]);
        atomicOr(&err_bits[g >> 5], 1u << (g & 31));
    }
    return fixed;
}

// ---- pass 3a: trusted mask of every k-mer ------------------------------------
// Wave per read.  A read whose k-mers are all trusted has no errors
// (readutils.cc:263-265) and needs nothing more than the tally; the others get
// their mask stored and a flag for the correction kernel.
template <int NW, int NK, int MINW = 1>
__global__ void __launch_bounds__(256, MINW) k_scan_trusted(ReadsDev R, KParams K, FiltDev T, uint64_t *tmask,
                                                       uint8_t *dirty, uint32_t *err_bits, unsigned long long *stats,
                                                       int fast, unsigned int *ticket) {
    using S = Stage<NW>;
    __shared__ uint32_t lds[4][2 * S::WORDS];
    const int lane = threadIdx.x & 63;
    uint32_t *L32 = lds[threadIdx.x >> 6];
    const uint64_t *hint = reinterpret_cast<const uint64_t *>(R.hint_trusted);
    const int k = K.k;
    unsigned long long q_total = 0;
    uint64_t off = 0, word = 0;
    uint32_t len = 0;
    const uint64_t oc_max = R.n_bases / 64 + 1;
    ReadChunks<32> Q;      // reads in chunks of 32 from a global counter (device_common.h)
    uint64_t r = Q.begin(ticket, R.n_reads, lane);
    if (r < R.n_reads) {
        read_span(R, r, off, len);
        word = stage_fetch<NW>(R, hint, R.offcase, off, oc_max, off, lane);
    }
    for (; r < R.n_reads; Q.advance(lane), r = Q.cur) {
        __builtin_amdgcn_wave_barrier();
        if (lane < S::WORDS) stage_store(L32, lane, word);
        __builtin_amdgcn_wave_barrier();
        const uint64_t cur = off;
        const int o31 = (int)(off & 31), o63 = (int)(off & 63);
        const int Lr = (int)len, nk = Lr - k + 1;
        const uint64_t r_next = Q.peek(lane);
        if (r_next < R.n_reads) {   // the next read's words travel while this one is processed
            read_span(R, r_next, off, len);
            word = stage_fetch<NW>(R, hint, R.offcase, off, oc_max, off, lane);
        }
        if (nk <= 0) { if (lane == 0) dirty[r] = 0; continue; }
        // a read with off-case bases (soft-masked FASTQ) follows the reference's raw-character comparisons: it takes
        // the one-read-per-lane walk (correct.h), which carries the case bits (state 3)
        bool odd = false;
        if (R.offcase) {
#pragma unroll
            for (int c = 0; c < NW; ++c)
                if (c * 64 < Lr) odd = odd || __ballot(c * 64 + lane < Lr && lds_bit(L32 + 2 * S::X, o63 + c * 64 + lane)) != 0;
        }
        uint64_t M[NW];
        int trusted = 0;
        // addresses first, then every block (one random HBM line each) and pattern (L2) load of the read back to back,
        // then the tests: NW line fetches in flight per lane (k_infer, steps A and B); a lane with nothing to look up
        // reads block 0 / pattern 0
        bool valid[NK], known[NK];
        uint32_t blk[NK], pat[NK];
        ulonglong2 t[NK], p[NK];
#pragma unroll
        for (int c = 0; c < NW; ++c) M[c] = 0;
#pragma unroll
        for (int c = 0; c < NK; ++c) {
            valid[c] = false;
            known[c] = false;   // this read put the k-mer into the trusted filter itself (pass 2)
            blk[c] = 0;
            pat[c] = 0;
            const int s = c * 64 + lane;
            if (s < nk) {
                const uint64_t key = canon_key(lds_window64(L32 + 2 * S::B, 2 * (o31 + s)), K);
                valid[c] = (lds_window32(L32 + 2 * S::M, o63 + s) & K.nmask_bits) == 0;
                known[c] = hint && lds_bit(L32 + 2 * S::H, o63 + s);
                if (valid[c] && !known[c]) { blk[c] = block_of(T, key); pat[c] = pattern_of(T, key); }
            }
        }
#pragma unroll
        for (int c = 0; c < NK; ++c) t[c] = T.table[blk[c]];
#pragma unroll
        for (int c = 0; c < NK; ++c) p[c] = T.patterns[pat[c]];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NK; ++c) {
            const bool ok = known[c] || (valid[c] && ((p[c].x & ~t[c].x) | (p[c].y & ~t[c].y)) == 0);
            M[c] = __ballot(ok);
            trusted += __popcll(M[c]);
        }
        // 0 = every k-mer trusted, nothing to do; 2 = settled by the fast path right here (the read is still staged:
        // waves in their lookup rounds and waves streaming through clean reads share the CU); 1 = needs the walk
        int state = trusted != nk ? (odd ? 3 : 1) : 0;
        if (state == 1 && fast) {
            uint64_t Z[NW];
#pragma unroll
            for (int c = 0; c < NW; ++c) {
                const int rem = nk - c * 64;
                Z[c] = rem <= 0 ? 0ULL : (~M[c] & (rem >= 64 ? ~0ULL : ((1ULL << rem) - 1)));
            }
            if (fast_path<NW>(L32, K, T, Z, nk - trusted, cur, nk, o31, o63, err_bits, lane, q_total)) state = 2;
        }
        if (lane == 0) dirty[r] = (uint8_t)state;
        if ((state & 1) && lane < NW) tmask[r * NW + lane] = sel_word<NW>(M, lane);
    }
    if (lane == 0 && q_total) atomicAdd(&stats[1], q_total);
}

// match = 0: every read with a non-zero flag; otherwise the reads whose flag equals `match`
__global__ void __launch_bounds__(1024) k_compact(const uint8_t *dirty, uint64_t n, uint32_t *list,
                                                   unsigned long long *count, int match) {
    // one global atomic per 1024-lane block: wave counts -> LDS scan -> block base
    __shared__ unsigned int wave_cnt[16];
    __shared__ unsigned long long block_base;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool d = i < n && (match ? dirty[i] == match : dirty[i] != 0);
    const unsigned long long bal = __ballot(d);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[w] = (unsigned int)__popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int tot = 0;
        for (int k = 0; k < 16; ++k) { const unsigned int c = wave_cnt[k]; wave_cnt[k] = tot; tot += c; }
        block_base = tot ? atomicAdd(count, (unsigned long long)tot) : 0ULL;
    }
    __syncthreads();
    if (d) list[block_base + wave_cnt[w] + __popcll(bal & ((1ULL << lane) - 1))] = (uint32_t)i;
}

// ---- pass 3b: the correction walk, one read per lane ---------------------------
template <int MAXL, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_correct(ReadsDev R, KParams K, FiltDev T, const uint32_t *list,
                                                    const unsigned long long *n_list, const uint64_t *tmask,
                                                    int tmask_words, uint32_t *err_bits, uint32_t *patch,
                                                    unsigned long long *stats,        // stats[1] += Bloom queries
                                                    int dyn_len, uint32_t *dyn_words) {
    typedef Corrector<MAXL> C;
    extern __shared__ uint32_t lds[];
    const uint64_t n = *n_list;
    unsigned long long q_total = 0;
    for (uint64_t slot = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; slot < n; slot += (uint64_t)gridDim.x * BLOCK) {
        const uint64_t r = list[slot];
        uint64_t off; uint32_t len32;
        read_span(R, r, off, len32);
        const int len = (int)len32;
        C cx;
        cx.dyn_len = dyn_len;
        // MAXL = 0: the lane's words live in global scratch (one slice per lane of the grid), not in LDS
        cx.L = MAXL ? lds + threadIdx.x : dyn_words + ((size_t)blockIdx.x * BLOCK + threadIdx.x);
        cx.stride = MAXL ? BLOCK : (int)(gridDim.x * BLOCK);
        cx.f = T;
        cx.K = K;
        cx.qual = R.qual + off;
        cx.t_ok = true;
        cx.queries = 0;
        // stage the read: 2-bit bases, non-ACGT mask, trusted mask; clear flags
        for (int w = 0; w < cx.NWB(); ++w)
            cx.word(cx.OFF_W(), w) = w * 16 < len ? (uint32_t)window64(R.bases, 2 * (off + (uint64_t)w * 16)) : 0u;
        for (int w = 0; w < cx.NWN(); ++w) {
            cx.word(cx.OFF_NM(), w) = w * 32 < len ? (uint32_t)window64(R.nmask, off + (uint64_t)w * 32) : 0u;
            cx.word(cx.OFF_LC(), w) = (R.offcase && w * 32 < len) ? (uint32_t)window64(R.offcase, off + (uint64_t)w * 32) : 0u;
            cx.word(cx.OFF_E(), w) = 0;
            const int tw = w >> 1;
            cx.word(cx.OFF_T(), w) = tw < tmask_words ? (uint32_t)(tmask[r * tmask_words + tw] >> (32 * (w & 1))) : 0u;
        }
        // bits past the read end must not look like bases
        if (len & 31) { cx.word(cx.OFF_NM(), len >> 5) &= (1u << (len & 31)) - 1; cx.word(cx.OFF_LC(), len >> 5) &= (1u << (len & 31)) - 1; }
        if (len & 15) cx.word(cx.OFF_W(), len >> 4) &= (1u << ((len & 15) * 2)) - 1;

        const int k = K.k;
        const CallResult top = cx.run_call(0, len, true, 6);
        if (top.patch_pos >= 0) patch[r] = 0x80000000u | ((uint32_t)top.patch_pos << 8) | (uint32_t)top.patch_base;
        // readutils.cc:547-563
        if (top.bad_prefix > 0 && (top.bad_prefix >= len / 2 || top.bad_prefix >= 2 * k))
            cx.run_call(0, top.bad_prefix + 1, false, 6);
        if (top.bad_suffix >= 0 && top.bad_suffix < len &&
            (len - top.bad_suffix > len / 2 || len - top.bad_suffix > 2 * k))
            cx.run_call(top.bad_suffix, len - top.bad_suffix, false, 6);
        // publish the flags into the batch-wide bit array
        for (int w = 0; w * 32 < len; ++w) {
            const uint32_t v = cx.word(cx.OFF_E(), w);
            if (!v) continue;
            const uint64_t g = off + (uint64_t)w * 32;
            atomicOr(&err_bits[g >> 5], v << (g & 31));
            if (g & 31) atomicOr(&err_bits[(g >> 5) + 1], v >> (32 - (g & 31)));
        }
        q_total += cx.queries;
    }
    // per-wave reduction of the query counter
    for (int o = 32; o > 0; o >>= 1) q_total += __shfl_down(q_total, o);
    if ((threadIdx.x & 63) == 0 && q_total) atomicAdd(&stats[1], q_total);
}


// ---- pass 3b (default): the correction walk, one read per WAVEFRONT (correct_wave.h) ----
template <int NB, int NN>
__global__ void __launch_bounds__(256, 5) k_correct_wave(ReadsDev R, KParams K, FiltDev T, const uint32_t *list,
                                                       const unsigned long long *n_list, const uint64_t *tmask,
                                                       int tmask_words, uint32_t *err_bits, uint32_t *patch,
                                                       unsigned long long *stats, unsigned int *ticket) {
    typedef WaveCorrector<NB, NN> C;
    __shared__ uint64_t lds_words[4 * C::WORDS];
    const int lane = threadIdx.x & 63;
    const uint64_t n = *n_list;
    unsigned long long q_total = 0;
    // the work list goes out in chunks of four entries from a global counter (device_common.h: ReadChunks): a walk costs
    // anything from a dozen to a few thousand lookups, and an even split leaves the last round of workgroups half empty
    ReadChunks<4> Q;
    for (uint64_t slot = Q.begin(ticket, n, lane); slot < n; Q.advance(lane), slot = Q.cur) {
        const uint64_t r = C::uni((uint64_t)list[slot]);
        uint64_t off; uint32_t len32;
        read_span(R, r, off, len32);
        off = C::uni(off);
        const int len = C::uni((int)len32);
        C cx;
        cx.S = lds_words + (threadIdx.x >> 6) * C::WORDS;
        cx.f = T;
        cx.K = K;
        cx.qual = R.qual + off;
        cx.lane = lane;
        cx.queries = 0;
        // stage the read into the wavefront's LDS slice (lane w writes word w)
        for (int slot_w = lane; slot_w < C::WORDS; slot_w += 64) {
            uint64_t v = 0;
            if (slot_w < NB) {
                const int w = slot_w;
                if (w * 32 < len) v = window64(R.bases, 2 * (off + (uint64_t)w * 32));
                const int rem = len - w * 32;
                if (rem < 32) v &= rem > 0 ? ((1ULL << (2 * rem)) - 1) : 0ULL;
            } else if (slot_w >= C::NM && slot_w < C::NM + NN) {
                const int c = slot_w - C::NM;
                if (c * 64 < len) v = window64(R.nmask, off + (uint64_t)c * 64);
                const int rem = len - c * 64;
                if (rem < 64) v &= rem > 0 ? ((1ULL << rem) - 1) : 0ULL;
            } else if (slot_w >= C::Tc && slot_w < C::Tc + NN) {
                const int c = slot_w - C::Tc;
                if (c < tmask_words) v = tmask[r * tmask_words + c];
            }
            cx.S[slot_w] = v;
        }
        const int k = K.k;
        // activation 0 is the read; 1 and 2 are the one-level recursion on a long unfixed prefix /
        // suffix (readutils.cc:547-563).  One call site keeps a single inlined copy of the walk.
        int pre_n = 0, suf_lo = 0, suf_n = 0;
#pragma unroll 1
        for (int act = 0; act < 3; ++act) {
            int lo = 0, nn = len;
            if (act == 1) { if (!pre_n) continue; nn = pre_n; }
            if (act == 2) { if (!suf_n) continue; lo = suf_lo; nn = suf_n; }
            const typename C::CallOut out = cx.run_call(lo, nn, 6);
            if (act == 0) {
                if (out.patch_pos >= 0 && lane == 0)
                    patch[r] = 0x80000000u | ((uint32_t)out.patch_pos << 8) | (uint32_t)out.patch_base;
                if (out.bad_prefix > 0 && (out.bad_prefix >= len / 2 || out.bad_prefix >= 2 * k)) pre_n = out.bad_prefix + 1;
                if (out.bad_suffix >= 0 && out.bad_suffix < len &&
                    (len - out.bad_suffix > len / 2 || len - out.bad_suffix > 2 * k)) {
                    suf_lo = out.bad_suffix;
                    suf_n = len - out.bad_suffix;
                }
            }
        }
        // publish the flags: lane l owns the 32-bit piece l of the read's flag words
        if (lane < 2 * NN && lane * 32 < len) {
            const uint32_t v = (uint32_t)(cx.ldw(C::E, lane >> 1) >> (32 * (lane & 1)));
            if (v) {
                const uint64_t g = off + (uint64_t)lane * 32;
                atomicOr(&err_bits[g >> 5], v << (g & 31));
                if (g & 31) atomicOr(&err_bits[(g >> 5) + 1], v >> (32 - (g & 31)));
            }
        }
        q_total += cx.queries;
    }
    if (lane == 0 && q_total) atomicAdd(&stats[1], q_total);
}

// ---- pass 3c: covariate tally ---------------------------------------------------
// CCovariateData::consume_read (covariateutils.cc:193-202).  Only the cycle and
// dinucleotide tables are tallied: qcov[rg][q] and rgcov[rg] are exact sums of
// the cycle table (every base is counted in all three, :30-42, :65-76, :102-116)
// and are formed once at the end.  Totals go through an LDS-private table
// (16-bit counters, packed two per word, flushed before they can wrap) for the
// read group of the block's first read; error counts and anything outside the
// LDS table go straight to 64-bit global atomics (rare).
struct HistDev {
    unsigned long long *cycle;   // [n_rg][256][2][n_cycle][2]
    unsigned long long *dinuc;   // [n_rg][256][16][2]
    int n_rg, n_cycle;
};

// A read's patch word (k_correct / k_correct_wave): bit 31 = the walk returned early with one base replaced
// (readutils.cc:263-265), bits 8-30 = its position in the read (KBBQ_MAX_READ_LEN is what 23 bits hold), bits 0-1 = the base.
__device__ __forceinline__ int patch_at(uint32_t pt) { return (int)((pt >> 8) & 0x7FFFFFu); }

__device__ __forceinline__ uint64_t cyc_index(const HistDev &H, int rg, int q, int s, int c) {
    return ((((uint64_t)rg * KBBQ_NQ + q) * 2 + s) * H.n_cycle + c) * 2;
}

// One lane per 16 consecutive bases of the batch (16-byte quality load, 4-byte base load), a block of
// 1024 lanes per 16 Ki bases; the block's LDS table is flushed before any 16-bit counter could wrap
// (a read adds at most 1 to a counter, so the block counts the reads it has touched).
// One launch tallies the cycles [cbase, cbase + ccap) (reads longer than the LDS tables' 192 cycles take several
// launches; cycles beyond the tables used to go through global atomics: 100 ms per launch for 250-base reads) of
// the reads of ONE read group (lds_rg; its tables are the ones in LDS) and ignores the rest: a
// batch with several read groups gets one launch per group that occurs in it (`present`, a bit per group from
// k_rg_presence; a launch for an absent group returns at once).  Counting the other groups through global atomics
// in the same launch serialises on a few hundred hot addresses -- 400 ms instead of 1 ms for four groups.
__global__ void k_rg_presence(const uint16_t *rg, uint64_t n_reads, uint32_t n_rg, uint32_t *present) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_reads) {
        const uint32_t g = rg[i] < n_rg ? rg[i] : 0;      // reads of an undeclared group are not tallied anyway
        // a few bits, millions of reads: look before the atomic (the words sit in cache), or every read of the
        // batch queues up on the same address
        const uint32_t bit = 1u << (g & 31);
        if (!(present[g >> 5] & bit)) atomicOr(&present[g >> 5], bit);
    }
}

struct TallyPlan {
    uint8_t qslot[256];    // slot of a quality value in the LDS tables, 255 = none (counted through global atomics)
    uint8_t qof[256];      // quality value of a slot
    int n_slots;           // slots in use (at most 255)
    int identity;          // slot == quality for every quality below n_slots (few, small values: no lookup needed)
    int rg_base, n_rgs;    // this launch tallies the read groups [rg_base, rg_base + n_rgs)
    int cbase;             // ... and the cycles [cbase, cbase + ccap)
    int direct_cycles;     // k_tally only: the cycle counters have no LDS table (reads of many thousand bases: a window of ccap
                           // cycles per launch would re-read the batch once per window) -- every cycle in one launch, straight to
                           // the histograms, where a long read's counters are as many addresses as it has bases; the few, hot
                           // dinucleotide counters stay in LDS
};

// The quality axis of the LDS tables is compacted to the values the batches hold (TallyPlan; a quality is any uint8_t,
// as in the reference, whose tables grow with the largest one seen).
__global__ void __launch_bounds__(1024) k_tally(ReadsDev R, HistDev H, const uint32_t *err_bits, const uint32_t *patch,
                                                 int ccap, int minscore, int vec_ok, TallyPlan P, const uint32_t *present,
                                                 const uint32_t *read_index) {
    extern __shared__ uint32_t lds[];
    const int ns = P.n_slots, cbase = P.cbase;
    if (present) {      // none of this launch's read groups occurs in the batch
        bool any = false;
        for (int g = P.rg_base; g < P.rg_base + P.n_rgs; ++g) any = any || ((present[g >> 5] >> (g & 31)) & 1);
        if (!any) return;
    }
    // layout per read group: cycle totals [2][ns][ccap] u16 (packed, cycle slots permuted), cycle errors likewise,
    // dinuc totals [ns][16] u32, dinuc errors [ns][16] u32 (few, hot addresses: global atomics on them serialise
    // at the memory side); then the quality -> slot map and the reads-touched counter
    const bool direct = P.direct_cycles != 0;
    const int cyc_words = direct ? 0 : (2 * ccap * ns + 1) / 2;
    const int per_rg = 2 * cyc_words + 2 * ns * 16;
    uint32_t *l_reads = lds + P.n_rgs * per_rg;
    uint8_t *l_qslot = reinterpret_cast<uint8_t *>(l_reads + 1);
    const int lds_words = P.n_rgs * per_rg + 1;
    for (int i = threadIdx.x; i < lds_words; i += blockDim.x) lds[i] = 0;
    if (threadIdx.x < 256) l_qslot[threadIdx.x] = P.qslot[threadIdx.x];
    __syncthreads();
    // one base: cycle and dinucleotide counters of (read group slot lr, quality q, second, cycle cyc)
    auto count = [&](int lr, int rg, int q, int second, int cyc, int er, bool dinuc_ok, int d) {
        const int cr = cyc - cbase;      // inside this launch's cycle window?
        if (!direct && (unsigned)cr >= (unsigned)ccap) return;
        const int sl = (int)l_qslot[q];
        if (direct) {
            atomicAdd(&H.cycle[cyc_index(H, rg, q, second, cyc) + 1], 1ULL);
            if (er) atomicAdd(&H.cycle[cyc_index(H, rg, q, second, cyc)], 1ULL);
            if (dinuc_ok) {
                if (sl != 255) {
                    uint32_t *t = lds + lr * per_rg;
                    atomicAdd(&t[sl * 16 + d], 1u);
                    if (er) atomicAdd(&t[ns * 16 + sl * 16 + d], 1u);
                } else {
                    atomicAdd(&H.dinuc[(((uint64_t)rg * KBBQ_NQ + q) * 16 + d) * 2 + 1], 1ULL);
                    if (er) atomicAdd(&H.dinuc[(((uint64_t)rg * KBBQ_NQ + q) * 16 + d) * 2], 1ULL);
                }
            }
        } else if (sl != 255) {
            uint32_t *t = lds + lr * per_rg;
            // lanes of one instruction are 16 cycles apart: store cycle c at slot (c%16)*(ccap/16) + c/16 so that
            // they land in neighbouring words, not in two banks
            const int idx = (second * ns + sl) * ccap + (cr & 15) * (ccap >> 4) + (cr >> 4);
            const uint32_t one = 1u << (16 * (idx & 1));
            atomicAdd(&t[idx >> 1], one);
            if (er) atomicAdd(&t[cyc_words + (idx >> 1)], one);
            if (dinuc_ok) {
                atomicAdd(&t[2 * cyc_words + sl * 16 + d], 1u);
                if (er) atomicAdd(&t[2 * cyc_words + ns * 16 + sl * 16 + d], 1u);
            }
        } else {      // a quality value without a slot (not announced, or more distinct values than fit): straight to the histograms
            atomicAdd(&H.cycle[cyc_index(H, rg, q, second, cyc) + 1], 1ULL);
            if (er) atomicAdd(&H.cycle[cyc_index(H, rg, q, second, cyc)], 1ULL);
            if (dinuc_ok) {
                atomicAdd(&H.dinuc[(((uint64_t)rg * KBBQ_NQ + q) * 16 + d) * 2 + 1], 1ULL);
                if (er) atomicAdd(&H.dinuc[(((uint64_t)rg * KBBQ_NQ + q) * 16 + d) * 2], 1ULL);
            }
        }
    };
    const uint64_t n_groups = (R.n_bases + 15) / 16;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t iters = (n_groups + stride - 1) / stride;
    for (uint64_t it = 0; it < iters; ++it) {
        const uint64_t grp = it * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const uint64_t g0 = grp * 16;
        int starts = 0;
        if (g0 < R.n_bases) {
            uint64_t r, start, end;
            if (R.offsets) {
                find_read(R, read_index, g0, r, start, end);
            } else {
                r = g0 / R.read_len;
                start = r * R.read_len;
                end = start + R.read_len;
            }
            const int n = (int)min((uint64_t)16, R.n_bases - g0);
            uint8_t qv[16];
            if (n == 16 && vec_ok) {
                const uint4 v = *reinterpret_cast<const uint4 *>(R.qual + g0);
                memcpy(qv, &v, 16);
            } else {
                for (int i = 0; i < 16; ++i) qv[i] = i < n ? R.qual[g0 + i] : 0;
            }
            const uint32_t bw = (uint32_t)(R.bases[g0 >> 5] >> ((g0 & 31) * 2));
            const uint32_t nw = (uint32_t)(R.nmask[g0 >> 6] >> (g0 & 63)) & 0xFFFFu;
            const uint32_t ew = (err_bits[g0 >> 5] >> (g0 & 31)) & 0xFFFFu;
            int prev_b = 0, prev_n = 1;
            if (g0 > 0) {
                const uint64_t gp = g0 - 1;
                prev_b = (int)((R.bases[gp >> 5] >> ((gp & 31) * 2)) & 3);
                prev_n = (int)((R.nmask[gp >> 6] >> (gp & 63)) & 1);
            }
            int rg = R.rg ? (int)R.rg[r] : 0;
            int second = R.flags ? (R.flags[r] & 1) : 0;
            uint32_t pt = patch ? patch[r] : 0u;
            starts = g0 == start ? 1 : 0;
            // Same shape as the apply kernel: a group lies in one read or straddles one boundary, and which read a
            // base belongs to is a select on its index (also for the read's patch, if it has one).  Anything else
            // -- a second boundary inside the group, the batch's last group -- takes the general loop.
            const int bpos = end - g0 < 16 ? (int)(end - g0) : 16;
            int rg2 = rg, second2 = second;
            uint32_t pt2 = 0;
            uint64_t end2 = end;
            if (bpos < 16 && r + 1 < R.n_reads) {
                rg2 = R.rg ? (int)R.rg[r + 1] : 0;
                second2 = R.flags ? (R.flags[r + 1] & 1) : 0;
                pt2 = patch ? patch[r + 1] : 0u;
                end2 = R.offsets ? R.offsets[r + 2] : end + R.read_len;
            }
            const int c0 = (int)(g0 - start);
            // read groups of this launch are [rg_base, rg_base + n_rgs); everything else belongs to another launch
            const int lr1 = rg - P.rg_base, lr2 = rg2 - P.rg_base;
            const bool mine1 = (unsigned)lr1 < (unsigned)P.n_rgs && rg < H.n_rg;
            const bool mine2 = (unsigned)lr2 < (unsigned)P.n_rgs && rg2 < H.n_rg;
            const bool one_boundary = bpos == 16 || (r + 1 < R.n_reads && end2 >= g0 + 16);
            if (n == 16 && one_boundary && !mine1 && (bpos == 16 || !mine2)) {
                starts += bpos < 16 ? 1 : 0;       // a group of other launches' read groups
            } else if (n == 16 && one_boundary && c0 + bpos <= H.n_cycle && 16 - bpos <= H.n_cycle) {
                const int pp1 = (pt >> 31) ? patch_at(pt) : -2, pp2 = (pt2 >> 31) ? patch_at(pt2) : -2;
                if (pp1 >= 0 && pp1 == c0 - 1) { prev_b = (int)(pt & 3); prev_n = 0; }
                starts += bpos < 16 ? 1 : 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const bool in2 = i >= bpos;
                    const int cyc = in2 ? i - bpos : c0 + i;
                    int b = (int)((bw >> (2 * i)) & 3), nn = (int)((nw >> i) & 1);
                    if (cyc == (in2 ? pp2 : pp1)) { b = (int)((in2 ? pt2 : pt) & 3); nn = 0; }
                    const int q = qv[i];
                    if (in2 ? mine2 : mine1)
                        count(in2 ? lr2 : lr1, in2 ? rg2 : rg, q, in2 ? second2 : second, cyc, (int)((ew >> i) & 1u),
                              cyc >= 1 && q >= minscore && !(nn | prev_n), (prev_b << 2) | b);
                    prev_b = b;
                    prev_n = nn;
                }
            } else {
            // the read this group starts in may carry a patch on the base just before the group
            if ((pt >> 31) && g0 > start && patch_at(pt) == (int)(g0 - 1 - start)) { prev_b = (int)(pt & 3); prev_n = 0; }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint64_t g = g0 + i;
                if (i < n) {
                    while (g >= end) {
                        ++r; start = end;
                        end = R.offsets ? R.offsets[r + 1] : end + R.read_len;
                        rg = R.rg ? (int)R.rg[r] : 0;
                        second = R.flags ? (R.flags[r] & 1) : 0;
                        pt = patch ? patch[r] : 0u;
                        if (end > start) ++starts;
                    }
                    const int cyc = (int)(g - start);
                    int b = (int)((bw >> (2 * i)) & 3), nn = (int)((nw >> i) & 1);
                    if ((pt >> 31) && patch_at(pt) == cyc) { b = (int)(pt & 3); nn = 0; }
                    const int q = qv[i];
                    const int lr = rg - P.rg_base;
                    if ((unsigned)lr < (unsigned)P.n_rgs && rg < H.n_rg && cyc < H.n_cycle)
                        count(lr, rg, q, second, cyc, (int)((ew >> i) & 1), cyc >= 1 && q >= minscore && !nn && !prev_n, (prev_b << 2) | b);
                    prev_b = b;
                    prev_n = nn;
                }
            }
            }
            // a read that began before this group and continues into it also counts once for this block
            starts += g0 != start ? 1 : 0;
        }
        {
            // reads touched by this wavefront (a lane touches at most 17): bit-sliced ballot sum, one LDS add per wave
            const int s = g0 < R.n_bases ? starts : 0;
            unsigned int tot = 0;
#pragma unroll
            for (int b = 0; b < 5; ++b) tot += (unsigned int)__popcll(__ballot((s >> b) & 1)) << b;
            if ((threadIdx.x & 63) == 0 && tot) atomicAdd(l_reads, tot);
        }
        __syncthreads();
        // a block-uniform decision: everyone reads the counter before anyone may add to it again (next iteration)
        // (direct_cycles: only the 32-bit dinucleotide counters are in LDS, and a block adds at most 2^14 per iteration to one)
        const bool flush = (direct ? (it & 0xFFFF) == 0xFFFF : *l_reads >= 40000u) || it + 1 == iters;
        __syncthreads();
        if (flush) {
            for (int lr = 0; lr < P.n_rgs; ++lr) {
                uint32_t *t = lds + lr * per_rg;
                const int rg = P.rg_base + lr;
                for (int w = threadIdx.x; w < cyc_words; w += blockDim.x) {
                    const uint32_t v = t[w], ve = t[cyc_words + w];
                    if (!v) continue;      // no total, no error
                    t[w] = 0;
                    t[cyc_words + w] = 0;
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t cnt = (v >> (16 * h)) & 0xFFFFu, cne = (ve >> (16 * h)) & 0xFFFFu;
                        if (!cnt) continue;
                        const int idx = 2 * w + h;
                        const int slot = idx % ccap, rest = idx / ccap;
                        const int q = P.qof[rest % ns], s = rest / ns;
                        const int c = cbase + (slot % (ccap >> 4)) * 16 + slot / (ccap >> 4);
                        if (c < H.n_cycle) {
                            atomicAdd(&H.cycle[cyc_index(H, rg, q, s, c) + 1], (unsigned long long)cnt);
                            if (cne) atomicAdd(&H.cycle[cyc_index(H, rg, q, s, c)], (unsigned long long)cne);
                        }
                    }
                }
                for (int w = threadIdx.x; w < ns * 16; w += blockDim.x) {
                    const uint32_t v = t[2 * cyc_words + w], ve = t[2 * cyc_words + ns * 16 + w];
                    const uint64_t cell = ((uint64_t)rg * KBBQ_NQ + P.qof[w >> 4]) * 16 + (w & 15);
                    if (v) { t[2 * cyc_words + w] = 0; atomicAdd(&H.dinuc[cell * 2 + 1], (unsigned long long)v); }
                    if (ve) { t[2 * cyc_words + ns * 16 + w] = 0; atomicAdd(&H.dinuc[cell * 2], (unsigned long long)ve); }
                }
            }
            if (threadIdx.x == 0) *l_reads = 0;
            __syncthreads();
        }
    }
}

// ---- pass 3c, the common shape: uniform read length, one read group, one cycle window ------------------------
// k_tally's general loop spends 1 700 VALU + 1 700 SALU instructions per 16 bases on bookkeeping that a batch of
// equally long reads of one read group does not need (read boundaries from an offsets array, read-group slots,
// cycle windows).  Here: the read of a 16-base group is one 64-bit multiply-high (inv_len = ceil(2^64 / read_len),
// exact for the 32-bit base offsets of a batch), a group lies in one read or straddles one boundary (read_len >= 16)
// and every cycle is in the table.  Same LDS tables (packed 16-bit counters, cycle slots permuted against bank
// conflicts) and same result as k_tally.  MAP = false: the plan is the identity (slot = quality for the values below
// n_slots: the usual FASTQ range), no slot lookup; MAP = true: slot from the plan's table in LDS.  A quality without a
// slot goes to the histograms directly.
template <bool MAP>
__global__ void __launch_bounds__(1024) k_tally_uniform(ReadsDev R, HistDev H, const uint32_t *err_bits, const uint32_t *patch,
                                                         int ccap, int minscore, unsigned long long inv_len, TallyPlan P) {
    extern __shared__ uint32_t lds[];
    const int ns = P.n_slots;
    const int L = (int)R.read_len;
    const int cyc_words = (2 * ccap * ns + 1) / 2, cstep = ccap >> 4;
    uint32_t *t_err = lds + cyc_words, *t_di = lds + 2 * cyc_words, *t_die = t_di + ns * 16;
    const int lds_words = 2 * cyc_words + 2 * ns * 16;
    uint8_t *l_qslot = reinterpret_cast<uint8_t *>(lds + lds_words);
    for (int i = threadIdx.x; i < lds_words; i += blockDim.x) lds[i] = 0;
    if (MAP && threadIdx.x < 256) l_qslot[threadIdx.x] = P.qslot[threadIdx.x];
    __syncthreads();
    auto count = [&](int second, int q, int cyc, int er, bool dinuc_ok, int d) {
        const int sl = MAP ? (int)l_qslot[q] : q;
        if (MAP ? sl != 255 : sl < ns) {
            const int idx = (second * ns + sl) * ccap + (cyc & 15) * cstep + (cyc >> 4);
            const uint32_t one = 1u << (16 * (idx & 1));
            atomicAdd(&lds[idx >> 1], one);
            if (er) atomicAdd(&t_err[idx >> 1], one);
            if (dinuc_ok) {
                atomicAdd(&t_di[sl * 16 + d], 1u);
                if (er) atomicAdd(&t_die[sl * 16 + d], 1u);
            }
        } else {
            atomicAdd(&H.cycle[cyc_index(H, 0, q, second, cyc) + 1], 1ULL);
            if (er) atomicAdd(&H.cycle[cyc_index(H, 0, q, second, cyc)], 1ULL);
            if (dinuc_ok) {
                atomicAdd(&H.dinuc[((uint64_t)q * 16 + d) * 2 + 1], 1ULL);
                if (er) atomicAdd(&H.dinuc[((uint64_t)q * 16 + d) * 2], 1ULL);
            }
        }
    };
    auto flush = [&]() {
        for (int w = threadIdx.x; w < cyc_words; w += blockDim.x) {
            const uint32_t v = lds[w], ve = t_err[w];
            if (!v) continue;
            lds[w] = 0;
            t_err[w] = 0;
            for (int h = 0; h < 2; ++h) {
                const uint32_t cnt = (v >> (16 * h)) & 0xFFFFu, cne = (ve >> (16 * h)) & 0xFFFFu;
                if (!cnt) continue;
                const int idx = 2 * w + h;
                const int slot = idx % ccap, rest = idx / ccap;
                const int q = MAP ? (int)P.qof[rest % ns] : rest % ns, sec = rest / ns;
                const int c = (slot % cstep) * 16 + slot / cstep;
                if (c < H.n_cycle) {
                    atomicAdd(&H.cycle[cyc_index(H, 0, q, sec, c) + 1], (unsigned long long)cnt);
                    if (cne) atomicAdd(&H.cycle[cyc_index(H, 0, q, sec, c)], (unsigned long long)cne);
                }
            }
        }
        for (int w = threadIdx.x; w < ns * 16; w += blockDim.x) {
            const uint32_t v = t_di[w], ve = t_die[w];
            const uint64_t cell = (uint64_t)(MAP ? (int)P.qof[w >> 4] : (w >> 4)) * 16 + (w & 15);
            if (v) { t_di[w] = 0; atomicAdd(&H.dinuc[cell * 2 + 1], (unsigned long long)v); }
            if (ve) { t_die[w] = 0; atomicAdd(&H.dinuc[cell * 2], (unsigned long long)ve); }
        }
    };
    const uint64_t n_groups = (R.n_bases + 15) / 16;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t iters = (n_groups + stride - 1) / stride;
    // a read adds at most one to a 16-bit counter; a block touches at most 16384 / L + 2 reads per iteration
    const uint64_t flush_every = max((uint64_t)1, (uint64_t)40000 / ((uint64_t)16384 / (uint64_t)L + 2));
    for (uint64_t it = 0; it < iters; ++it) {
        const uint64_t g0 = (it * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
        if (g0 + 16 <= R.n_bases) {
            const uint32_t r = (uint32_t)__umul64hi((unsigned long long)g0, inv_len);
            const int c0 = (int)(g0 - (uint64_t)r * (uint64_t)L);
            const int bpos = L - c0 < 16 ? L - c0 : 16;      // first base of the next read inside the group
            const uint4 qv4 = *reinterpret_cast<const uint4 *>(R.qual + g0);
            uint8_t qv[16];
            memcpy(qv, &qv4, 16);
            const uint32_t bw = (uint32_t)(R.bases[g0 >> 5] >> ((g0 & 31) * 2));
            const uint32_t nw = (uint32_t)(R.nmask[g0 >> 6] >> (g0 & 63)) & 0xFFFFu;
            const uint32_t ew = (err_bits[g0 >> 5] >> (g0 & 31)) & 0xFFFFu;
            int prev_b = 0, prev_n = 1;
            if (g0 > 0) {
                const uint64_t gp = g0 - 1;
                prev_b = (int)((R.bases[gp >> 5] >> ((gp & 31) * 2)) & 3);
                prev_n = (int)((R.nmask[gp >> 6] >> (gp & 63)) & 1);
            }
            const bool two = bpos < 16;
            const bool mine1 = !R.rg || R.rg[r] == 0, mine2 = !two || !R.rg || R.rg[r + 1] == 0;
            const int sec1 = R.flags ? (R.flags[r] & 1) : 0;
            const int sec2 = two && R.flags ? (R.flags[r + 1] & 1) : 0;
            const uint32_t pt1 = patch ? patch[r] : 0u, pt2 = two && patch ? patch[r + 1] : 0u;
            const int pp1 = (pt1 >> 31) ? patch_at(pt1) : -2, pp2 = (pt2 >> 31) ? patch_at(pt2) : -2;
            if (pp1 >= 0 && pp1 == c0 - 1) { prev_b = (int)(pt1 & 3); prev_n = 0; }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool in2 = i >= bpos;
                const int cyc = c0 + i - (in2 ? L : 0);
                int b = (int)((bw >> (2 * i)) & 3), nn = (int)((nw >> i) & 1);
                if (cyc == (in2 ? pp2 : pp1)) { b = (int)((in2 ? pt2 : pt1) & 3); nn = 0; }
                const int q = qv[i];
                if (in2 ? mine2 : mine1)
                    count(in2 ? sec2 : sec1, q, cyc, (int)((ew >> i) & 1u), cyc >= 1 && q >= minscore && !(nn | prev_n), (prev_b << 2) | b);
                prev_b = b;
                prev_n = nn;
            }
        } else if (g0 < R.n_bases) {
            // the batch's last, partial group: base by base, everything looked up afresh
            for (uint64_t g = g0; g < R.n_bases; ++g) {
                const uint64_t r = g / (uint64_t)L;
                const int cyc = (int)(g - r * (uint64_t)L);
                const uint32_t pt = patch ? patch[r] : 0u;
                const int pp = (pt >> 31) ? patch_at(pt) : -2;
                auto base_at = [&](uint64_t x, int cx, int &bb, int &nb) {
                    bb = (int)((R.bases[x >> 5] >> ((x & 31) * 2)) & 3);
                    nb = (int)((R.nmask[x >> 6] >> (x & 63)) & 1);
                    if (cx == pp) { bb = (int)(pt & 3); nb = 0; }
                };
                int b, nn, pb = 0, pn = 1;
                base_at(g, cyc, b, nn);
                if (cyc >= 1) base_at(g - 1, cyc - 1, pb, pn);
                const int q = R.qual[g];
                const int er = (int)((err_bits[g >> 5] >> (g & 31)) & 1u);
                if (!R.rg || R.rg[r] == 0)
                    count(R.flags ? (R.flags[r] & 1) : 0, q, cyc, er, cyc >= 1 && q >= minscore && !(nn | pn), (pb << 2) | b);
            }
        }
        if ((it + 1) % flush_every == 0 || it + 1 == iters) {      // block-uniform
            __syncthreads();
            flush();
            __syncthreads();
        }
    }
}

// quality values of a batch that was not seen by pass 2 (--fixed mode, pass 3 on its own): 256 bits, the same set k_infer leaves
__global__ void __launch_bounds__(256) k_qpresence(const uint8_t *qual, uint64_t n_bases, uint32_t *qpresent) {
    __shared__ uint32_t qseen[8];
    if (threadIdx.x < 8) qseen[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_bases; i += (uint64_t)gridDim.x * blockDim.x)
        qseen_note(qseen, qual[i]);
    __syncthreads();
    if (threadIdx.x < 8 && qseen[threadIdx.x]) atomicOr(&qpresent[threadIdx.x], qseen[threadIdx.x]);
}

// ---- pass 4: delta-Q apply -------------------------------------------------------
// CReadData::recalibrate (readutils.cc:572-595).  One lane per 16 consecutive
// bases of the batch: 16-byte quality load and store, 4-byte base load.
struct DqDev {
    const int16_t *base;   // [n_rg][256]  meanq + rgdq + qscoredq
    const int8_t *cycle;   // [n_rg][256][2][n_cycle]
    const int8_t *dinuc;   // [n_rg][256][16]
    const uint8_t *qslot;  // [256] slot of a quality in the LDS tables (255 = it has no cycle / dinucleotide delta anywhere)
    int n_rg, n_cycle, n_slots;
};

__global__ void __launch_bounds__(1024) k_recalibrate(ReadsDev R, DqDev D, uint8_t *out, int minqual, int vec_ok, int lds_rgs,
                                                       const uint32_t *read_index, uint64_t base0, uint64_t base1) {
    // The delta-Q tables of the first `lds_rgs` read groups sit in LDS, compacted over the quality axis: only the
    // D.n_slots quality values that have a non-zero cycle or dinucleotide delta anywhere get a slot (D.qslot; a
    // handful for binned qualities, so a dozen read groups fit), holding per (slot, second, cycle) one int16 with
    // meanq + rg + q delta-Q + cycle delta-Q already summed and the int8 dinucleotide delta-Q; every other
    // quality only needs its base value.  Two or three dependent LDS reads per base instead of three global ones.
    extern __shared__ uint8_t l_tab[];
    uint8_t *l_qslot = l_tab;                                   // [256]: slot of a quality, 255 = none
    const int ns = D.n_slots;
    const int cyc_cells = ns * 2 * D.n_cycle, di_bytes = ns * 16, base_bytes = KBBQ_NQ * 2;
    const int cyc_bytes = 2 * cyc_cells;
    const int per_rg = (cyc_bytes + di_bytes + base_bytes + 3) & ~3;
    uint8_t *l_rg = l_tab + 256;
    if (threadIdx.x < 256) l_qslot[threadIdx.x] = D.qslot[threadIdx.x];
    __syncthreads();
    for (int i = threadIdx.x; i < lds_rgs * KBBQ_NQ; i += blockDim.x) {
        const int rg = i / KBBQ_NQ, q = i % KBBQ_NQ;
        reinterpret_cast<int16_t *>(l_rg + (size_t)rg * per_rg + cyc_bytes + di_bytes)[q] = D.base[rg * KBBQ_NQ + q];
        const int sl = l_qslot[q];
        if (sl != 255) {
            int16_t *tc = reinterpret_cast<int16_t *>(l_rg + (size_t)rg * per_rg) + sl * 2 * D.n_cycle;
            const int8_t *src = D.cycle + ((size_t)rg * KBBQ_NQ + q) * 2 * D.n_cycle;
            for (int c = 0; c < 2 * D.n_cycle; ++c) tc[c] = (int16_t)(D.base[rg * KBBQ_NQ + q] + src[c]);
            for (int x = 0; x < 16; ++x) l_rg[(size_t)rg * per_rg + cyc_bytes + sl * 16 + x] = (uint8_t)D.dinuc[((size_t)rg * KBBQ_NQ + q) * 16 + x];
        }
    }
    __syncthreads();
    // persistent blocks: the table load above is paid once per block, not once per 4 KB of qualities
    // (the launch covers the bases [base0, base1) of the batch: a host batch may arrive in pieces)
    for (uint64_t g0 = base0 + ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; g0 < base1;
         g0 += (uint64_t)gridDim.x * blockDim.x * 16) {
    // read containing g0
    uint64_t r, start, end;
    if (R.offsets) {
        find_read(R, read_index, g0, r, start, end);
    } else {
        r = g0 / R.read_len;
        start = r * R.read_len;
        end = start + R.read_len;
    }
    const int n = (int)min((uint64_t)16, R.n_bases - g0);
    uint8_t qv[16];
    if (n == 16 && vec_ok) {
        const uint4 v = *reinterpret_cast<const uint4 *>(R.qual + g0);
        memcpy(qv, &v, 16);
    } else {
        for (int i = 0; i < 16; ++i) qv[i] = i < n ? R.qual[g0 + i] : 0;
    }
    const uint32_t bw = (uint32_t)(R.bases[g0 >> 5] >> ((g0 & 31) * 2));
    const uint32_t nw = (uint32_t)(R.nmask[g0 >> 6] >> (g0 & 63)) & 0xFFFFu;
    int prev_b = 0, prev_n = 1;
    if (g0 > 0) {
        const uint64_t gp = g0 - 1;
        prev_b = (int)((R.bases[gp >> 5] >> ((gp & 31) * 2)) & 3);
        prev_n = (int)((R.nmask[gp >> 6] >> (gp & 63)) & 1);
    }
    int rg = R.rg ? (int)R.rg[r] : 0;
    int second = R.flags ? (R.flags[r] & 1) : 0;
    uint8_t res[16];
    // A group of 16 bases lies in one read or straddles one boundary (reads of 16 bases or more): handled
    // without a branch per base -- the read a base belongs to is a select on its index.  Everything else (a
    // second boundary inside the group, a read group whose tables are not in LDS, the batch's last group)
    // takes the general loop below.
    const int bpos = end - g0 < 16 ? (int)(end - g0) : 16;      // first base of the next read inside the group
    int rg2 = rg, second2 = second;
    uint64_t end2 = end;
    if (bpos < 16 && r + 1 < R.n_reads) {
        rg2 = R.rg ? (int)R.rg[r + 1] : 0;
        second2 = R.flags ? (R.flags[r + 1] & 1) : 0;
        end2 = R.offsets ? R.offsets[r + 2] : end + R.read_len;
    }
    const int c0 = (int)(g0 - start);
    const bool plain = n == 16 && rg < lds_rgs && rg2 < lds_rgs && (bpos == 16 || (r + 1 < R.n_reads && end2 >= g0 + 16)) &&
                       c0 + bpos <= D.n_cycle && 16 - bpos <= D.n_cycle;
    if (plain) {
        const int16_t *ta = reinterpret_cast<const int16_t *>(l_rg + rg * per_rg) + second * D.n_cycle + c0;
        const int16_t *tb = reinterpret_cast<const int16_t *>(l_rg + rg2 * per_rg) + second2 * D.n_cycle - bpos;
        const int8_t *da = reinterpret_cast<const int8_t *>(l_rg + rg * per_rg + cyc_bytes);
        const int8_t *db = reinterpret_cast<const int8_t *>(l_rg + rg2 * per_rg + cyc_bytes);
        const int16_t *ba = reinterpret_cast<const int16_t *>(l_rg + rg * per_rg + cyc_bytes + di_bytes);
        const int16_t *bb = reinterpret_cast<const int16_t *>(l_rg + rg2 * per_rg + cyc_bytes + di_bytes);
        const int qstride = 2 * D.n_cycle;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool in2 = i >= bpos;
            const int b = (int)((bw >> (2 * i)) & 3), nn = (int)((nw >> i) & 1);
            const int q = qv[i];
            int v = q;
            if (q >= minqual) {
                const int sl = l_qslot[q];
                if (sl != 255) {
                    v = (in2 ? tb : ta)[sl * qstride + i];
                    const bool first = in2 ? i == bpos : c0 + i == 0;        // cycle 0 has no dinucleotide context
                    if (!first && !(nn | prev_n)) v += (in2 ? db : da)[sl * 16 + ((prev_b << 2) | b)];
                } else {
                    v = (in2 ? bb : ba)[q];      // no cycle or dinucleotide delta anywhere for this quality
                }
            }
            res[i] = (uint8_t)(v < 0 ? 0 : (v > KBBQ_MAXQ ? KBBQ_MAXQ : v));
            prev_b = b;
            prev_n = nn;
        }
    } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint64_t g = g0 + i;
        if (i < n) {
            while (g >= end) {
                ++r; start = end;
                end = R.offsets ? R.offsets[r + 1] : end + R.read_len;
                rg = R.rg ? (int)R.rg[r] : 0;
                second = R.flags ? (R.flags[r] & 1) : 0;
            }
        }
        const int cyc = (int)(g - start);
        const int b = (int)((bw >> (2 * i)) & 3), nn = (int)((nw >> i) & 1);
        const int q = qv[i];
        int v = q;
        if (i < n && q >= minqual && rg < D.n_rg && cyc < D.n_cycle) {
            const int cell = rg * KBBQ_NQ + q;
            const bool use_di = cyc > 0 && !nn && !prev_n;
            if (rg < lds_rgs) {
                const uint8_t *t = l_rg + rg * per_rg;
                const int sl = l_qslot[q];
                if (sl != 255) {
                    v = reinterpret_cast<const int16_t *>(t)[(sl * 2 + second) * D.n_cycle + cyc];
                    if (use_di) v += (int8_t)t[cyc_bytes + sl * 16 + ((prev_b << 2) | b)];
                } else {
                    v = reinterpret_cast<const int16_t *>(t + cyc_bytes + di_bytes)[q];
                }
            } else {
                v = D.base[cell] + D.cycle[((uint64_t)cell * 2 + second) * D.n_cycle + cyc];
                if (use_di) v += D.dinuc[cell * 16 + ((prev_b << 2) | b)];
            }
        }
        res[i] = (uint8_t)(v < 0 ? 0 : (v > KBBQ_MAXQ ? KBBQ_MAXQ : v));
        prev_b = b;
        prev_n = nn;
    }
    }
    if (n == 16 && vec_ok) {
        uint4 v;
        memcpy(&v, res, 16);
        *reinterpret_cast<uint4 *>(out + g0) = v;
    } else {
        for (int i = 0; i < n; ++i) out[g0 + i] = res[i];
    }
    }
}

// ---- helpers ----------------------------------------------------------------------
__global__ void k_or_words(uint64_t *dst, const uint64_t *src, uint64_t n) {
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 < n) {
        ulonglong2 a = *reinterpret_cast<ulonglong2 *>(dst + i);
        const ulonglong2 b = *reinterpret_cast<const ulonglong2 *>(src + i);
        a.x |= b.x; a.y |= b.y;
        *reinterpret_cast<ulonglong2 *>(dst + i) = a;
    } else if (i < n) {
        dst[i] |= src[i];
    }
}

// dst |= OR of n_pieces consecutive pieces of `src` (each piece_words long), skipping piece `skip`:
// the reduce step of the OR all-reduce in one launch
__global__ void k_or_pieces(uint64_t *dst, const uint64_t *src, uint64_t piece_words, int n_pieces, int skip) {
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= piece_words) return;
    if (i + 1 < piece_words) {
        ulonglong2 a = *reinterpret_cast<ulonglong2 *>(dst + i);
        for (int p = 0; p < n_pieces; ++p) {
            if (p == skip) continue;
            const ulonglong2 b = *reinterpret_cast<const ulonglong2 *>(src + (uint64_t)p * piece_words + i);
            a.x |= b.x; a.y |= b.y;
        }
        *reinterpret_cast<ulonglong2 *>(dst + i) = a;
    } else {
        uint64_t a = dst[i];
        for (int p = 0; p < n_pieces; ++p)
            if (p != skip) a |= src[(uint64_t)p * piece_words + i];
        dst[i] = a;
    }
}

// synthetic data set (kbbq_amd/synth.py is the host twin, bit for bit)
struct SynthDev {
    uint64_t seed, genome_len, first_read, n_reads;
    uint32_t read_len, n_rg, paired, n_thr;   // n_thr: N threshold on 20 bits
    const uint32_t *qcum;     // [read_len][4] cumulative 32-bit thresholds for Q2,Q12,Q22,Q32 (else Q37)
    const uint32_t *errthr;   // [94] 32-bit error thresholds
};
__device__ __host__ __forceinline__ uint64_t synth_hash(uint64_t seed, uint64_t stream, uint64_t idx) {
    return mix64(mix64(seed + stream * 0x9e3779b97f4a7c15ULL) + idx * 0xD1342543DE82EF95ULL);
}
__device__ __forceinline__ int synth_genome(uint64_t seed, uint64_t pos) { return (int)(synth_hash(seed, 0, pos) >> 62); }

__global__ void __launch_bounds__(256) k_synth(SynthDev S, uint64_t *bases, uint64_t *nmask, uint8_t *qual,
                                                uint8_t *flags, uint16_t *rg) {
    // one lane per 32 consecutive bases of the batch (one 2-bit word)
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t total = S.n_reads * S.read_len;
    const uint64_t g0 = w * 32;
    if (g0 >= total) return;
    uint64_t bword = 0;
    uint32_t nbits = 0;
    for (int i = 0; i < 32; ++i) {
        const uint64_t g = g0 + i;
        if (g >= total) break;
        const uint64_t lr = g / S.read_len;
        const uint32_t c = (uint32_t)(g - lr * S.read_len);
        const uint64_t r = S.first_read + lr;
        const uint64_t h1 = synth_hash(S.seed, 1, r);
        const uint64_t start = (h1 >> 1) % (S.genome_len - S.read_len + 1);
        const int strand = (int)(h1 & 1);
        int b = strand ? 3 - synth_genome(S.seed, start + S.read_len - 1 - c) : synth_genome(S.seed, start + c);
        const uint64_t hb = synth_hash(S.seed, 3, r * S.read_len + c);
        const uint32_t uq = (uint32_t)hb, ue = (uint32_t)(hb >> 32);
        const uint32_t *qc = S.qcum + 4 * c;
        int q = uq < qc[0] ? 2 : uq < qc[1] ? 12 : uq < qc[2] ? 22 : uq < qc[3] ? 32 : 37;
        if (ue < S.errthr[q]) b = (b + 1 + (int)(synth_hash(S.seed, 4, r * S.read_len + c) % 3)) & 3;
        const bool isn = (synth_hash(S.seed, 5, r * S.read_len + c) & 0xFFFFF) < S.n_thr;
        if (isn) { b = 0; q = 2; nbits |= 1u << i; }
        bword |= (uint64_t)b << (2 * i);
        qual[g] = (uint8_t)q;
        if (c == 0) {
            flags[lr] = S.paired ? (uint8_t)(r & 1) : 0;
            rg[lr] = (uint16_t)((synth_hash(S.seed, 6, r) >> 32) % S.n_rg);
        }
    }
    bases[w] = bword;
    // two lanes share one mask word
    uint32_t *nm32 = reinterpret_cast<uint32_t *>(nmask);
    nm32[w] = nbits;
}

