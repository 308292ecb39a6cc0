// bgzf_device.h -- the BGZF writer's kernels (gfx950): FASTQ record assembly, one DEFLATE block + CRC-32 per
// wavefront, and the gather of the finished blocks into file order.  Included once by bgzf_device.hip.
//
//   k_fastq_text   "@name\nseq\n+comment\nqual\n" per record (FastqFile::write, htsiter.cc:75-86), one wavefront per
//                  record, the quality line straight from the recalibrated qualities in HBM (+33, htsiter.cc:61-65)
//   k_deflate      one BGZF block of <= 0xff00 payload bytes per wavefront: LZ77 matches over a hash table in LDS,
//                  64 positions per step; dynamic Huffman codes from the block's own counts; the bits of 64 tokens per step
//                  placed by a wave prefix sum; CRC-32 of 64 slices chained by their x^(8 n) factors
//   k_block_offsets / k_gather   exclusive sum of the block sizes, blocks copied back to back
//
// Integer / byte work, no MFMA.  The encoder is built for HBM-resident text that never visits the host uncompressed:
// what leaves the GPU is the compressed stream (about a quarter of the text).  A wavefront works through its block at its
// own pace -- the chip holds a couple of thousand of them -- so the per-block algorithms are simple and mostly serial
// within the wave; the scalar pieces (Huffman lengths, header, token bits) are shared with the host twin
// (deflate_common.h) and run on lane 0.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "deflate_common.h"

namespace kbbq {
namespace dfl {

// The match finder's table: HASH_SIZE buckets of HASH_WAYS positions (u16), the most recent position of a hash in way 0 and
// -- with two ways -- the one it displaced in way 1: every position then has two candidates, the nearer one preferred on
// equal length.  One way of 4096 (the default): 8 KB of LDS (8192: 0.5 % smaller output, 15 % slower; 2048 with five waves
// per SIMD: 7 % faster, 0.5 % larger: profiles/r03_deflate_variants.txt).  Two ways were built and measured in round 4
// (profiles/r04_deflate_variants.txt, the same 503 MB of FASTQ text, zlib level 6 = 181 MB): one way 194 MB at 40 GB/s, two
// ways of 4096 191 MB at 34.6 GB/s -- 1.5 % of size for 14 % of the rate; on that text zlib itself goes from 1.098x (level 1,
// chains of 4) over 1.067x (level 2, chains of 8) to 1.037x (level 3, chains of 32): the gap to level 6 is chain depth, and
// every way costs a sweep-2 compare per position.  -DKBBQ_DFL_WAYS=2 / -DKBBQ_DFL_HASH_BITS build the variants.
#ifndef KBBQ_DFL_WAYS
#define KBBQ_DFL_WAYS 1
#endif
#ifndef KBBQ_DFL_HASH_BITS
#define KBBQ_DFL_HASH_BITS 12
#endif
constexpr int HASH_WAYS = KBBQ_DFL_WAYS;
constexpr int HASH_BITS = KBBQ_DFL_HASH_BITS;
constexpr int HASH_SIZE = 1 << HASH_BITS;
constexpr int DFL_WAVES = 1;                        // wavefronts per workgroup: each works alone (no workgroup barrier anywhere), 18 KB of LDS
constexpr int MIN_TAKE = 4;                         // shortest match the 4-byte hash can find
constexpr int SWEEP2_BYTES = 16;                    // how far the match-length sweep looks; longer matches are followed by the parse
constexpr uint32_t SLOT_BYTES = 65536 + 64;         // one finished block per slot; its bytes start at SLOT_SHIFT so that
constexpr uint32_t SLOT_SHIFT = 6;                  // the DEFLATE stream (behind the 18-byte header) is 8-byte aligned
constexpr uint32_t TOKENS_PER_WAVE = 65536;
constexpr int PF = 4;                               // steps of 64 positions whose loads are in flight ahead of the one worked on
constexpr int PF2 = 2;                              // steps per round of the match-length sweep (two rounds in flight: registers)

struct DeflateArgs {
    const uint8_t *payload;     // n bytes (+ 16 readable bytes behind them)
    uint64_t n;
    uint32_t n_blocks;
    uint8_t *slots;             // n_blocks x SLOT_BYTES, zeroed
    uint32_t *sizes;            // n_blocks: bytes of every finished block
    uint32_t *tokens;           // per wavefront of the grid: TOKENS_PER_WAVE words
#ifdef KBBQ_DFL_PROFILE
    unsigned long long *prof;   // (tools/deflate_probe.py, a build of its own) cycles per phase, summed over the wavefronts
#endif
};
#ifdef KBBQ_DFL_PROFILE
#define DFL_MARK(i) do { const unsigned long long _t = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(&A.prof[i], _t - prof_t); prof_t = _t; } while (0)
#else
#define DFL_MARK(i) do { } while (0)
#endif

__device__ __forceinline__ uint64_t load8(const uint8_t *p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ uint32_t hash4(uint32_t v) { return (v * 2654435761u) >> (32 - HASH_BITS); }

// CRC-32 (gzip polynomial, reflected) of len bytes by one wavefront: 256 pieces of q bytes, four per lane (four
// independent table walks keep the LDS pipe busy), every piece's register started at 0; the pieces are joined by the rule
// for running B behind A, r_AB = r_A * x^(8|B|) + r_B in GF(2)[x] mod P (crc_chain), an associative rule: a lane folds its
// four, the wave reduces in six steps.  tab: crc_table_entry(0..255) in LDS; xq_for / xq / xq4: the caller's cache of
// x^(8 q), x^(32 q) for the piece length last seen (blocks are nearly all of one length).  Returns the value of the trailer.
__device__ __forceinline__ uint32_t wave_crc32(const uint8_t *in, int len, const uint32_t *tab, int lane, int &xq_for, uint32_t &xq, uint32_t &xq4) {
    const int q = (len + 255) / 256;
    uint32_t crc_r = 0, crc_x = 0x80000000u;      // the lane's four pieces as one: register from 0, x^(8 * its bytes)
    if (q != xq_for) { xq = crc_xpow8((uint64_t)q); xq4 = crc_mulmod(xq, xq); xq4 = crc_mulmod(xq4, xq4); xq_for = q; }
    int a[4], n[4];
    uint32_t c[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = min(len, (4 * lane + j) * q); n[j] = min(len, a[j] + q) - a[j]; }
    int i = 0;
    for (; i + 8 <= n[3]; i += 8) {      // (n[0] >= n[1] >= n[2] >= n[3])
        uint64_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = load8(in + a[j] + i);
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) { c[j] = tab[(c[j] ^ (uint32_t)(v[j] >> (8 * k))) & 0xFFu] ^ (c[j] >> 8); }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        for (int t = i; t < n[j]; ++t) c[j] = tab[(c[j] ^ in[a[j] + t]) & 0xFFu] ^ (c[j] >> 8);
    // the lane's fold; only the block's last piece is shorter than q (and the ones behind it empty)
    if (__ballot(n[3] != q) == 0) {
        crc_r = crc_chain(crc_chain(crc_chain(c[0], c[1], xq), c[2], xq), c[3], xq);
        crc_x = xq4;
    } else {
        crc_r = c[0];
        crc_x = n[0] == q ? xq : crc_xpow8((uint64_t)n[0]);
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            const uint32_t xj = n[j] == q ? xq : crc_xpow8((uint64_t)n[j]);
            crc_r = crc_chain(crc_r, c[j], xj);
            crc_x = crc_mulmod(crc_x, xj);
        }
    }
    // lanes 2o apart join their runs of o lanes: (r, x) <- (r * x' + r', x * x'); lane 0 ends up with the whole block
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t r2 = __shfl_down(crc_r, o), x2 = __shfl_down(crc_x, o);
        crc_r = crc_chain(crc_r, r2, x2);
        crc_x = crc_mulmod(crc_x, x2);
    }
    return crc_chain(0xFFFFFFFFu, (uint32_t)__builtin_amdgcn_readfirstlane((int)crc_r), (uint32_t)__builtin_amdgcn_readfirstlane((int)crc_x)) ^ 0xFFFFFFFFu;
}

// LDS of one wavefront.  The Huffman scratch of the second phase lies over the hash table of the first.
struct WaveLds {
    union {
        uint16_t htab[HASH_WAYS * HASH_SIZE];      // way 0, then way 1
        struct {
            uint8_t head[HEAD_BYTES];
            uint16_t order[N_LL];
            uint32_t w[N_LL];
            uint16_t runs[N_LL + N_D];
            uint8_t all[N_LL + N_D];
            alignas(8) unsigned long long stage[64];      // the bits of one step's 64 tokens (at most 48 each) before they leave
        } hs;
    } u;
    uint32_t ll_freq[N_LL];
    uint32_t d_freq[N_D];
    BlockCodes codes;
    uint32_t crc_tab[256];      // CRC-32 of one byte (reflected polynomial 0xEDB88320), filled once per launch
};

__global__ void __launch_bounds__(64 * DFL_WAVES) k_deflate(DeflateArgs A) {
    __shared__ WaveLds lds_all[DFL_WAVES];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveLds &S = lds_all[wv];
    const uint32_t wave = blockIdx.x * DFL_WAVES + wv, n_waves = gridDim.x * DFL_WAVES;
    uint32_t *tokens = A.tokens + (size_t)wave * TOKENS_PER_WAVE;
    const uint64_t lane_lt = (1ull << lane) - 1;
    for (int i = lane; i < 256; i += 64) S.crc_tab[i] = crc_table_entry((uint32_t)i);
    int xq_for = -1;            // the piece length the two factors below belong to
    uint32_t xq = 0, xq4 = 0;   // x^(8 q), x^(32 q) mod P
    __builtin_amdgcn_wave_barrier();
    for (uint32_t blk = wave; blk < A.n_blocks; blk += n_waves) {
        const uint8_t *in = A.payload + (uint64_t)blk * BGZF_PAYLOAD;
        const int len = (int)min((uint64_t)BGZF_PAYLOAD, A.n - (uint64_t)blk * BGZF_PAYLOAD);
        uint8_t *slot = A.slots + (size_t)blk * SLOT_BYTES;
        uint8_t *block = slot + SLOT_SHIFT;                    // BGZF header here, the DEFLATE stream 18 bytes on
        unsigned long long *body64 = reinterpret_cast<unsigned long long *>(block + BGZF_HEAD);

#ifdef KBBQ_DFL_PROFILE
        unsigned long long prof_t = __builtin_readcyclecounter();
#endif
        // ---- reset
        for (int i = lane; i < HASH_WAYS * HASH_SIZE; i += 64) S.u.htab[i] = 0xFFFFu;
        for (int i = lane; i < N_LL; i += 64) S.ll_freq[i] = 0;
        if (lane < N_D) S.d_freq[lane] = 0;
        __builtin_amdgcn_wave_barrier();

        DFL_MARK(0);
        // ---- CRC-32 of the payload
        const uint32_t crc = wave_crc32(in, len, S.crc_tab, lane, xq_for, xq, xq4);
        DFL_MARK(1);
        // ---- LZ77 in three sweeps over the block, 64 positions per step.  A sweep never waits for a load that depends on
        // another load of the same step's chain more than once, and what it needs next is already on its way: the text is
        // read from HBM/L2, and a wave that followed hash table -> candidate -> text in one loop spent its time in those
        // round trips (round 3: 1.2 ms per block before).  The sweeps hand their results on in the
        // wave's token array (one word per position, overwritten in place).
        // Sweep 1: the hash table.  tokens[p] = the most recent earlier position with p's hash (0xFFFF: none).  The text of
        // four steps ahead is on its way while a step is worked on (a load from HBM takes longer than a step); the loads
        // are unconditional (clamped addresses, steps past the end run with every lane idle) so that a loaded register is
        // not touched before its step comes: a conditional load ends in a copy right behind it, and the wait with it.
        {
            uint64_t ahead[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) ahead[u] = load8(in + min(64 * u + lane, len));
            for (int b1 = 0; b1 < len; b1 += 64 * PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int p = b1 + 64 * u + lane;
                    const uint64_t cur = ahead[u];
                    ahead[u] = load8(in + min(p + 64 * PF, len));
                    const bool hashed = p + 4 <= len;
                    const uint32_t h = hash4((uint32_t)cur);
                    uint32_t cand = 0xFFFFu, cand2 = 0xFFFFu;
                    if (hashed) {
                        // (every lane of the step reads before any writes: LDS instructions of a wave run in order)
                        cand = S.u.htab[h];
                        if (HASH_WAYS > 1) { cand2 = S.u.htab[HASH_SIZE + h]; S.u.htab[HASH_SIZE + h] = (uint16_t)cand; }
                        S.u.htab[h] = (uint16_t)p;
                    }
                    __builtin_amdgcn_wave_barrier();
                    // of several lanes with one hash the highest position stays
                    for (;;) {
                        const bool lose = hashed && S.u.htab[h] < (uint16_t)p;
                        if (!__ballot(lose)) break;
                        if (lose) S.u.htab[h] = (uint16_t)p;
                        __builtin_amdgcn_wave_barrier();
                    }
                    if (p < len) tokens[p] = cand | (cand2 << 16);
                }
            }
        }
        DFL_MARK(2);
        // Sweep 2: how far the candidate matches, two steps at a time, up to 16 bytes by two compares (a FASTQ match is 6
        // bytes on average, a read name's 9-17).  A match that may be longer is followed to its end by sweep 3, and only if
        // the parse takes it: inside the 150-byte runs and repeats of BAM records every position has such a match, and a
        // lane that compared its 258 bytes alone held up its step (BAM: 52 % of the kernel in this sweep, round 4).  Nothing
        // a round needs is asked for inside it: the candidates come two rounds ahead, and with them -- one round ahead --
        // the text at the positions and at their candidates.  tokens[p] = length << 16 | distance, 0: no match.
        {
            uint64_t cur_n[PF2], cur2_n[PF2], old_n[HASH_WAYS][PF2], old2_n[HASH_WAYS][PF2];
            uint32_t cand_nn[PF2], prev_n[PF2];
            int cand_n[HASH_WAYS][PF2];
            auto clean = [&](int p, uint32_t c) -> int { return (p + 4 <= len && c != 0xFFFFu && p - (int)c <= MAX_DIST) ? (int)c : -1; };
#pragma unroll
            for (int u = 0; u < PF2; ++u) {      // round 0: everything; round 1: the candidates
                const int p = min(64 * u + lane, len), pn = min(64 * (u + PF2) + lane, len);
                const uint32_t raw = tokens[min(p, (int)TOKENS_PER_WAVE - 1)];
                cand_nn[u] = tokens[min(pn, (int)TOKENS_PER_WAVE - 1)];
                cur_n[u] = load8(in + p);
                cur2_n[u] = load8(in + p + 8);
                prev_n[u] = in[max(p, 1) - 1];
#pragma unroll
                for (int w = 0; w < HASH_WAYS; ++w) {
                    cand_n[w][u] = clean(64 * u + lane, w ? raw >> 16 : raw & 0xFFFFu);
                    old_n[w][u] = load8(in + max(cand_n[w][u], 0));
                    old2_n[w][u] = load8(in + max(cand_n[w][u], 0) + 8);
                }
            }
            for (int b0 = 0; b0 < len; b0 += 64 * PF2) {
                int cand[HASH_WAYS][PF2];
                uint64_t cur[PF2], cur2[PF2], old[HASH_WAYS][PF2], old2[HASH_WAYS][PF2];
                uint32_t prev[PF2];
#pragma unroll
                for (int u = 0; u < PF2; ++u) {
                    cur[u] = cur_n[u]; cur2[u] = cur2_n[u]; prev[u] = prev_n[u];
#pragma unroll
                    for (int w = 0; w < HASH_WAYS; ++w) { old[w][u] = old_n[w][u]; old2[w][u] = old2_n[w][u]; cand[w][u] = cand_n[w][u]; }
                }
#pragma unroll
                for (int u = 0; u < PF2; ++u) {
                    const int q1 = b0 + 64 * (u + PF2) + lane, p1 = min(q1, len), p2 = min(q1 + 64 * PF2, len);
                    const uint32_t raw = cand_nn[u];
                    cand_nn[u] = tokens[min(p2, (int)TOKENS_PER_WAVE - 1)];
                    cur_n[u] = load8(in + p1);
                    cur2_n[u] = load8(in + p1 + 8);
                    prev_n[u] = in[p1 - 1];
#pragma unroll
                    for (int w = 0; w < HASH_WAYS; ++w) {
                        cand_n[w][u] = clean(q1, w ? raw >> 16 : raw & 0xFFFFu);
                        old_n[w][u] = load8(in + max(cand_n[w][u], 0));
                        old2_n[w][u] = load8(in + max(cand_n[w][u], 0) + 8);
                    }
                }
#pragma unroll
                for (int u = 0; u < PF2; ++u) {
                    const int p = b0 + 64 * u + lane;
                    const bool hashed = p + 4 <= len;
                    int L = 0, D = 0;
                    const int lim = min((int)MAX_MATCH, len - p);
#pragma unroll
                    for (int w = 0; w < HASH_WAYS; ++w) {      // way 0 is the nearer candidate: it wins a tie
                        if (cand[w][u] >= 0) {
                            const uint64_t x = cur[u] ^ old[w][u], x2 = cur2[u] ^ old2[w][u];
                            int n = x ? (int)(__builtin_ctzll(x) >> 3) : x2 ? 8 + (int)(__builtin_ctzll(x2) >> 3) : SWEEP2_BYTES;
                            n = min(n, lim);
                            if (n >= MIN_TAKE && n > L) { L = n; D = p - cand[w][u]; }
                        }
                    }
                    // a run of one byte (the candidate the table cannot hold: the position just before, inside this step)
                    if (hashed && p > 0 && (uint8_t)prev[u] == (uint8_t)cur[u]) {
                        const uint64_t rep = (uint64_t)(uint8_t)cur[u] * 0x0101010101010101ull;
                        const uint64_t x = cur[u] ^ rep, x2 = cur2[u] ^ rep;
                        int n = x ? (int)(__builtin_ctzll(x) >> 3) : x2 ? 8 + (int)(__builtin_ctzll(x2) >> 3) : SWEEP2_BYTES;
                        n = min(n, lim);
                        if (n >= MIN_TAKE && n > L) { L = n; D = 1; }
                    }
                    if (p < len) tokens[p] = ((uint32_t)L << 16) | (uint32_t)D;
                }
            }
        }
        DFL_MARK(3);
        // Sweep 3: greedy parse with one step of lazy evaluation (the positions where a token starts, and which are
        // matches); the tokens are packed to the front of the same array, behind the positions already read.
        int next_free = 0;       // first position not covered by a token yet
        uint32_t n_tok = 0;
#ifdef KBBQ_DFL_PROFILE
        unsigned long long prof_parse = 0, prof_emit = 0;
#endif
        {
            uint32_t ld_ahead[PF], byte_ahead[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int p = min(64 * u + lane, len);
                ld_ahead[u] = tokens[min(p, (int)TOKENS_PER_WAVE - 1)];
                byte_ahead[u] = in[p];
            }
            for (int b1 = 0; b1 < len; b1 += 64 * PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int b0 = b1 + 64 * u;
                    const int p = b0 + lane;
                    const bool in_block = p < len;
                    const uint32_t ld = ld_ahead[u];
                    const uint8_t byte = (uint8_t)byte_ahead[u];
                    // (the tokens of this step are written below the positions read so far: never where a step ahead was or
                    // will be read; the loads are unconditional, see sweep 1)
                    {
                        const int pn = min(p + 64 * PF, len);
                        ld_ahead[u] = tokens[min(pn, (int)TOKENS_PER_WAVE - 1)];
                        byte_ahead[u] = in[pn];
                    }
                    int L = in_block ? (int)(ld >> 16) : 0;
                    const int D = (int)(ld & 0xFFFFu);      // (past the end: whatever the clamped load found)
#ifdef KBBQ_DFL_PROFILE
                    const unsigned long long pt0 = __builtin_readcyclecounter();
#endif
                    // Every lane first says where the next token would start if one started at its position: behind its match,
                    // or one byte on when it has none or the position after it has a longer one (one step of lazy evaluation;
                    // lane 63 cannot see the next step's first position and takes what it has).  The walk from the first free
                    // position is then one lane read per token.
                    const uint64_t valid = __ballot(in_block);
                    const int L1 = __shfl_down(L, 1);
                    const bool take = L >= MIN_TAKE && !(lane < 63 && L1 > L);
                    const int nxt = lane + (take ? L : 1);
                    const uint64_t takes = __ballot(take);
                    const uint64_t longer = __ballot(take && L >= SWEEP2_BYTES);      // (as far as sweep 2 looked)
                    uint64_t starts = 0;
                    int rel = next_free - b0;
                    // Where matches are dense (FASTQ text: most positions lie in one) the walk reads one lane per token.  Where
                    // they are sparse (the sequence and quality bytes of BAM records) it goes from match to match: from a
                    // position without a taken match every position up to the next taken match starts a literal, bits set in
                    // one go (50 against 46 GB/s on BAM records; on FASTQ text the longer turn costs 14 %, hence the two forms).
                    if (!longer && __popcll(takes) >= 32) {
                        while (rel < 64) {
                            starts |= 1ull << rel;
                            rel = __builtin_amdgcn_readlane(nxt, rel);
                        }
                    } else if (!longer) {
                        while (rel < 64) {
                            const uint64_t ahead = takes >> rel;
                            if (!ahead) { starts |= ~0ull << rel; rel = 64; break; }
                            const int t = rel + (int)__builtin_ctzll(ahead);
                            starts |= ((2ull << t) - 1) & (~0ull << rel);
                            rel = __builtin_amdgcn_readlane(nxt, t);
                        }
                    } else {
                        // a token that is such a match: its bytes from SWEEP2_BYTES on, eight per lane, against the candidate's
                        while (rel < 64) {
                            const uint64_t ahead = takes >> rel;
                            if (!ahead) { starts |= ~0ull << rel; rel = 64; break; }
                            const int t = rel + (int)__builtin_ctzll(ahead);
                            starts |= ((2ull << t) - 1) & (~0ull << rel);
                            rel = t;
                            if ((longer >> rel) & 1) {
                                const int pr = b0 + rel, dr = __builtin_amdgcn_readlane(D, rel);
                                const int limr = min((int)MAX_MATCH, len - pr);
                                const int off = SWEEP2_BYTES + 8 * lane;
                                const bool on = off < limr;
                                uint64_t x = 0;
                                if (on) x = load8(in + pr + off) ^ load8(in + pr - dr + off);      // (reads stay inside [0, len + 8))
                                const uint64_t differ = __ballot(on && x != 0);
                                int lx;
                                if (differ) {
                                    const int f = (int)__builtin_ctzll(differ);
                                    const uint64_t xf = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, f) |
                                                        ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), f) << 32);
                                    lx = SWEEP2_BYTES + 8 * f + (int)(__builtin_ctzll(xf) >> 3);
                                } else {
                                    lx = limr;
                                }
                                lx = min(lx, limr);
                                if (lane == rel) L = lx;
                                rel += lx;
                            } else {
                                rel = __builtin_amdgcn_readlane(nxt, rel);
                            }
                        }
                    }
                    starts &= valid;
                    const uint64_t taken = starts & takes;
                    next_free = b0 + rel;
#ifdef KBBQ_DFL_PROFILE
                    const unsigned long long pt1 = __builtin_readcyclecounter();
                    prof_parse += pt1 - pt0;
#endif
                    const bool start = (starts >> lane) & 1, is_match = (taken >> lane) & 1;
                    if (start) {
                        uint32_t t;
                        if (is_match) {
                            t = token_match(L, D);
                            int ls, e, v, ds;
                            length_symbol(L, ls, e, v);
                            distance_symbol(D, ds, e, v);
                            atomicAdd(&S.ll_freq[ls], 1u);
                            atomicAdd(&S.d_freq[ds], 1u);
                        } else {
                            t = token_literal(byte);
                            atomicAdd(&S.ll_freq[byte], 1u);
                        }
                        tokens[n_tok + (uint32_t)__popcll(starts & lane_lt)] = t;
                    }
                    n_tok += (uint32_t)__popcll(starts);
#ifdef KBBQ_DFL_PROFILE
                    prof_emit += __builtin_readcyclecounter() - pt1;
#endif
                }
            }
        }
#ifdef KBBQ_DFL_PROFILE
        if (lane == 0) { atomicAdd(&A.prof[8], prof_parse); atomicAdd(&A.prof[9], prof_emit); }
#endif
        __builtin_amdgcn_wave_barrier();

        DFL_MARK(4);
        // ---- codes and header (lane 0; the scratch lies over the hash table, which is done with)
        for (int i = lane; i < HEAD_BYTES; i += 64) S.u.hs.head[i] = 0;
        __builtin_amdgcn_wave_barrier();
        // The literal/length alphabet (up to 286 used symbols in a BAM block) is sorted by (count, symbol) by all lanes: the
        // rank of a symbol is the number of used symbols with a smaller key, 286 broadcast reads against five keys per lane.
        // (Lane 0's Shell sort over arrays in LDS was 1.4 of the 2.3 M cycles this phase took per BAM block.)
        bool ll_done = false;
        {
            if (lane == 0) S.ll_freq[256] += 1;      // end of block
            __builtin_amdgcn_wave_barrier();
            uint32_t key[5], cnt[5], rank[5];
            uint32_t m = 0;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int sym = lane + 64 * j;
                cnt[j] = sym < N_LL ? S.ll_freq[sym] : 0u;
                key[j] = cnt[j] ? (cnt[j] << 9) | (uint32_t)sym : 0xFFFFFFFFu;
                rank[j] = 0;
                if (sym < N_LL) { S.u.hs.w[sym] = key[j]; S.codes.ll_len[sym] = 0; }
                m += (uint32_t)__popcll(__ballot(cnt[j] != 0));
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll 8
            for (int t = 0; t < N_LL; ++t) {
                const uint32_t kt = S.u.hs.w[t];
#pragma unroll
                for (int j = 0; j < 5; ++j) rank[j] += kt < key[j] ? 1u : 0u;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 5; ++j)
                if (cnt[j]) { S.u.hs.order[rank[j]] = (uint16_t)(lane + 64 * j); S.u.hs.w[rank[j]] = cnt[j]; }
            __builtin_amdgcn_wave_barrier();
            int ok = 0;
            if (lane == 0) ok = (m >= 2 && lengths_from_sorted((int)m, S.u.hs.order, S.u.hs.w, MAX_BITS, S.codes.ll_len)) ? 1 : 0;
            ll_done = __builtin_amdgcn_readfirstlane(ok) != 0;      // (a code longer than 15 bits: the serial form raises the rare counts)
        }
        if (lane == 0) build_block_codes(S.ll_freq, S.d_freq, S.codes, S.u.hs.head, S.u.hs.order, S.u.hs.w, S.u.hs.runs, S.u.hs.all, ll_done);
        __builtin_amdgcn_wave_barrier();
        const uint32_t head_bits = S.codes.head_bits;

        DFL_MARK(5);
        // ---- how long is the dynamic form?  counts x code lengths (+ extra bits) over the two alphabets; the end-of-block
        // symbol is in the counts
        uint64_t my_bits = 0;
        for (int s = lane; s < N_LL; s += 64) {
            const int eb = (s >= 265 && s < 285) ? (s - 261) >> 2 : 0;
            my_bits += (uint64_t)S.ll_freq[s] * (uint64_t)(S.codes.ll_len[s] + eb);
        }
        if (lane < N_D) my_bits += (uint64_t)S.d_freq[lane] * (uint64_t)(S.codes.d_len[lane] + (lane >= 4 ? (lane >> 1) - 1 : 0));
        for (int o = 32; o > 0; o >>= 1) my_bits += __shfl_xor(my_bits, o);
        const uint64_t total_bits = (uint64_t)head_bits + my_bits;
        const uint32_t dyn_bytes = (uint32_t)((total_bits + 7) >> 3), stored_bytes = (uint32_t)len + 5;
        uint32_t body_bytes;
        if (dyn_bytes < stored_bytes && dyn_bytes + BGZF_HEAD + BGZF_TAIL <= (uint32_t)BGZF_MAX_BLOCK) {
            // Header bits, then the tokens 64 at a time: an exclusive prefix sum of their lengths places them; the bits of a
            // step are OR-ed together in LDS and leave as whole 8-byte words, 64 lanes side by side -- the word a step ends
            // in travels on as `carry`.  (Round 3 OR-ed every token into the zeroed slot with one or two 64-bit atomics in
            // HBM: 128 of them per step, a fifth of the kernel.)
            unsigned long long carry = 0;
            for (uint32_t i0 = 0; i0 * 64 < head_bits; i0 += 64) {      // 8-byte word i holds the header's bits [64 i, 64 i + 64)
                const uint32_t i = i0 + lane;
                uint64_t v = 0;
                if (i * 64 < head_bits) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v |= (uint64_t)S.u.hs.head[i * 8 + k] << (8 * k);
                    if ((i + 1) * 64 <= head_bits) body64[i] = (unsigned long long)v;
                }
                const uint32_t part = head_bits >> 6;      // the word the header ends in (if it ends inside one)
                if ((head_bits & 63) && part >= i0 && part < i0 + 64)
                    carry = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)(part - i0)) |
                            ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)(part - i0)) << 32);
            }
            __builtin_amdgcn_wave_barrier();
            uint64_t base = head_bits;
            uint32_t tok_ahead[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) tok_ahead[u] = tokens[min(64u * u + lane, TOKENS_PER_WAVE - 1)];
            for (uint32_t i1 = 0; i1 < n_tok; i1 += 64 * PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    if (i1 + 64 * u >= n_tok) break;
                    const uint32_t i = i1 + 64 * u + lane;
                    const uint32_t tok = tok_ahead[u];
                    tok_ahead[u] = tokens[min(i + 64 * PF, TOKENS_PER_WAVE - 1)];
                    uint64_t v = 0;
                    int nb = 0;
                    if (i < n_tok) token_bits(tok, S.codes, v, nb);
                    int incl = nb;
                    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o); if (lane >= o) incl += y; }
                    const uint32_t w0 = (uint32_t)(base >> 6);                    // the step's first word: `carry` so far
                    const uint32_t rel = (uint32_t)(base & 63) + (uint32_t)(incl - nb);      // this token's first bit, from word w0's
                    S.u.hs.stage[lane] = lane == 0 ? carry : 0ull;
                    __builtin_amdgcn_wave_barrier();
                    if (nb) {
                        const uint32_t word = rel >> 6, sh = rel & 63;
                        atomicOr(&S.u.hs.stage[word], (unsigned long long)(v << sh));
                        if (sh + (uint32_t)nb > 64) atomicOr(&S.u.hs.stage[word + 1], (unsigned long long)(v >> (64 - sh)));
                    }
                    __builtin_amdgcn_wave_barrier();
                    base += (uint64_t)__shfl(incl, 63);
                    const uint32_t full = (uint32_t)(base >> 6) - w0;               // whole words finished by this step (< 50)
                    const unsigned long long mine = S.u.hs.stage[lane];
                    if ((uint32_t)lane < full) body64[w0 + lane] = mine;
                    carry = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, (int)full) |
                            ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine >> 32), (int)full) << 32);
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if (lane == 0) {
                // the end-of-block code, then the word(s) still on the way (before the trailer, which may share their bytes)
                const uint64_t v = S.codes.ll_code[256];
                const uint32_t word = (uint32_t)(base >> 6), sh = (uint32_t)(base & 63), nb = S.codes.ll_len[256];
                carry |= (unsigned long long)(v << sh);
                body64[word] = carry;
                if (sh + nb > 64) body64[word + 1] = (unsigned long long)(v >> (64 - sh));
            }
            body_bytes = dyn_bytes;
        } else {
            // stored block: BFINAL = 1, BTYPE = 00, LEN, ~LEN, the bytes as they are
            uint8_t *body = block + BGZF_HEAD;
            if (lane == 0) {
                body[0] = 1;
                body[1] = (uint8_t)(len & 0xFF); body[2] = (uint8_t)(len >> 8);
                body[3] = (uint8_t)(~len & 0xFF); body[4] = (uint8_t)((~len >> 8) & 0xFF);
            }
            for (int i = lane; i < len; i += 64) body[5 + i] = in[i];
            body_bytes = stored_bytes;
        }
        DFL_MARK(6);
        // ---- CRC chain, framing
        // (atomic / plain stores of this wave to its own slot: complete before the kernel ends, nobody else reads them earlier)
        if (lane == 0) {
            const uint32_t total = BGZF_HEAD + body_bytes + BGZF_TAIL;
            bgzf_header(block, total);
            A.sizes[blk] = total;
        }
        // (the trailer may share an 8-byte word with the last bits of the stream: a byte store and an atomic OR of zeros
        // into those bytes give the same word in either order)
        if (lane == 0) bgzf_trailer(block + BGZF_HEAD + body_bytes, crc, (uint32_t)len);
        DFL_MARK(7);
    }
}

// exclusive sum of the block sizes (one workgroup; a batch has a few thousand blocks), total in offsets[n]
__global__ void __launch_bounds__(1024) k_block_offsets(const uint32_t *sizes, uint32_t n, uint64_t *offsets) {
    __shared__ uint64_t part[1024];
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t a = min(n, per * threadIdx.x), b = min(n, a + per);
    uint64_t s = 0;
    for (uint32_t i = a; i < b; ++i) s += sizes[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (int i = 0; i < 1024; ++i) { const uint64_t v = part[i]; part[i] = run; run += v; }
        offsets[n] = run;
    }
    __syncthreads();
    uint64_t run = part[threadIdx.x];
    for (uint32_t i = a; i < b; ++i) { offsets[i] = run; run += sizes[i]; }
}

// one workgroup per block: slot -> its place in the output
__global__ void __launch_bounds__(256) k_gather(const uint8_t *slots, const uint32_t *sizes, const uint64_t *offsets, uint8_t *out) {
    const uint32_t blk = blockIdx.x;
    const uint8_t *src = slots + (size_t)blk * SLOT_BYTES + SLOT_SHIFT;
    uint8_t *dst = out + offsets[blk];
    const uint32_t n = sizes[blk];
    // (source 2 bytes off an 8-byte boundary, destination anywhere: bytes up to the destination's first 8-byte boundary,
    // 8-byte stores from unaligned 8-byte loads, bytes for the tail)
    const uint32_t lead = min(n, (uint32_t)((8 - ((uintptr_t)dst & 7)) & 7));
    for (uint32_t i = threadIdx.x; i < lead; i += blockDim.x) dst[i] = src[i];
    const uint32_t words = (n - lead) >> 3;
    for (uint32_t w = threadIdx.x; w < words; w += blockDim.x)
        *reinterpret_cast<uint64_t *>(dst + lead + (size_t)w * 8) = load8(src + lead + (size_t)w * 8);
    for (uint32_t i = lead + words * 8 + threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

// ---- FASTQ record text ------------------------------------------------------------------------------------------------
struct FastqArgs {
    const uint8_t *blob;          // name | comment | sequence of every record, back to back
    const uint32_t *lens;         // 3 per record
    const uint64_t *blob_off;     // n + 1: where a record's pieces start in blob
    const uint64_t *text_off;     // n + 1: where its text starts
    const uint8_t *qual;          // new qualities (phred) of the batch
    const uint64_t *qual_off;     // n + 1, or null: record r at r * uniform_len
    uint32_t uniform_len;
    uint64_t n_records;
    uint8_t *text;
};

// per-record sizes of blob and text (the scans turn them into offsets)
__global__ void k_fastq_sizes(const uint32_t *lens, uint64_t n, uint64_t *blob_sz, uint64_t *text_sz) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint64_t nl = lens[3 * r], cl = lens[3 * r + 1], sl = lens[3 * r + 2];
    blob_sz[r] = nl + cl + sl;
    text_sz[r] = nl + cl + 2 * sl + 6;      // '@' '\n' '\n' '+' '\n' '\n'
}

// "@" name "\n" seq "\n+" comment "\n" qual "\n" of one record, written by one wavefront: lane l takes the bytes l, l + 64, ...;
// four of them per round, their source bytes loaded before any is stored (the record's pieces are a few hundred bytes
// spread over four places: what bounds this is the latency of those loads, so they travel together)
// The sequence line of a record either as text or rebuilt from the engine's packed batch (2 bit per base, the non-ACGT mask
// and the off-case bits: exact for text over ACGTN and acgt, which is what a chunk kept without its text was checked for).
struct SeqSource {
    const uint8_t *text;           // the sequence characters, or null: from the packed arrays at base index `first`
    const uint64_t *bases, *nmask, *offcase;
    uint64_t first;
    __device__ __forceinline__ uint8_t at(uint32_t j) const {
        if (text) return text[j];
        const uint64_t g = first + j;
        const uint32_t b = (uint32_t)(bases[g >> 5] >> ((g & 31) * 2)) & 3u;
        const bool n = (nmask[g >> 6] >> (g & 63)) & 1, low = offcase && ((offcase[g >> 6] >> (g & 63)) & 1);
        const uint8_t c = n ? (uint8_t)'N' : (uint8_t)((0x54474341u >> (8 * b)) & 0xFFu);      // "ACGT"
        return low ? (uint8_t)(c | 0x20) : c;
    }
};
__device__ __forceinline__ void emit_fastq_record(int lane, const uint8_t *name, uint32_t nl, const uint8_t *comment, uint32_t cl, const SeqSource &seq,
                                                  uint32_t sl, const uint8_t *q, uint8_t *out) {
    const uint32_t a_seq = 1 + nl + 1, a_plus = a_seq + sl, a_com = a_plus + 2, a_q = a_com + cl + 1, total = a_q + sl + 1;
    for (uint32_t i0 = 0; i0 < total; i0 += 256) {
        uint8_t c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + 64 * (uint32_t)u + (uint32_t)lane;
            const uint8_t *sp = nullptr;
            uint8_t k = '\n', add = 0;
            if (i == 0) k = '@';
            else if (i < 1 + nl) sp = name + (i - 1);
            else if (i < a_seq) k = '\n';
            else if (i < a_plus) { if (seq.text) sp = seq.text + (i - a_seq); else k = seq.at(i - a_seq); }
            else if (i == a_plus) k = '\n';
            else if (i == a_plus + 1) k = '+';
            else if (i < a_com + cl) sp = comment + (i - a_com);
            else if (i < a_q) k = '\n';
            else if (i < a_q + sl) { sp = q + (i - a_q); add = 33; }
            c[u] = (sp && i < total) ? (uint8_t)(*sp + add) : k;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + 64 * (uint32_t)u + (uint32_t)lane;
            if (i < total) out[i] = c[u];
        }
    }
}

__global__ void __launch_bounds__(256) k_fastq_text(FastqArgs F) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
    for (uint64_t r = wave; r < F.n_records; r += n_waves) {
        const uint32_t nl = F.lens[3 * r], cl = F.lens[3 * r + 1], sl = F.lens[3 * r + 2];
        const uint8_t *name = F.blob + F.blob_off[r], *comment = name + nl, *seq = comment + cl;
        const uint8_t *q = F.qual + (F.qual_off ? F.qual_off[r] : r * (uint64_t)F.uniform_len);
        const SeqSource from = {seq, nullptr, nullptr, nullptr, 0};
        emit_fastq_record(lane, name, nl, comment, cl, from, sl, q, F.text + F.text_off[r]);
    }
}

}  // namespace dfl
}  // namespace kbbq
