// bgzf_inflate.h -- the input side of the BGZF codec on the device (gfx950): every BGZF block of a file inflated by one
// wavefront, then the four-line FASTQ records of the inflated text indexed and packed into the engine's read layout.
// Included once by bgzf_device.hip.
//
// What it replaces: the reference reads its input through htslib -- bgzf_read under kseq_read (htsiter.hh:101-126,
// htsiter.cc:49-60) -- on the host, once per pass.  Here the compressed file goes to HBM as it is and stays there;
// the host only finds the block boundaries (18-byte headers) and reads the file.
//
//   k_inflate            RFC 1951 decoder, one BGZF block per wavefront.  The decoder is a serial machine, so a wave runs
//                        it in uniform control flow and uses its lanes as storage and for the copies: 256 bytes of the
//                        compressed stream per vector register, a 512-entry literal/length table (up to three literals
//                        per entry) and a 256-entry distance table in LDS, the canonical-code bounds in vector registers for
//                        the codes the tables leave out (lane l holds the bound of the codes of length l: one compare +
//                        ballot finds a code's length, v_readlane fetches its symbol), and the most recent 2 KB of output
//                        in an LDS ring laid out at (HBM address) mod 2 KB: a match inside the ring is an LDS-to-LDS copy
//                        by up to 64 lanes, a finished 256-byte line leaves for HBM as 64 coalesced dword stores, and a
//                        match that reaches further back (DEFLATE allows 32 KB) is read back from HBM.  What the serial
//                        chain computes is split between the scalar unit (counters, branches) and the vector ALU (the
//                        stream's bits, table indices, lengths and distances): in_vgpr below.
//   k_count_newlines / k_newline_positions   where the lines of the inflated text start
//   k_fastq_records      per record (four lines): name / comment / sequence / quality fields, kseq's rules
//                        (htsiter.cc:52-59, kseq.h) and the read-name rules of readutils.cc:74-97 that need no dictionary
//   k_fastq_gather       sequence text and qualities (- 33) of the records into the batch's contiguous arrays
//   k_pack_text          2 bit per base, N mask, off-case bits from the sequence text (kbbq_pack_bases_case on the device)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bgzf_device.h"
#include "deflate_common.h"

namespace kbbq {
namespace dfl {

// Ring, table and register budget were chosen by measurement (tools/inflate_probe.py over builds with other values,
// profiles/r03_inflate_variants.txt): the decoder is bound by the latency of its own serial chain, so what counts is how many
// wavefronts a CU holds -- 8 KB ring + 10-bit table, 12 per CU: 17.8 GB/s of zlib-6 FASTQ text; 4 KB + 10 bits, 16: 28.7;
// 2 KB + 9 bits at six waves per SIMD (80 registers), 24: 32.7 -- although most matches of a 32 KB window then come back
// from HBM.  Round 4 (profiles/r04_inflate_variants.txt): 6 KB of LDS and 71 registers, 26 per CU, 40.8 GB/s of that text.
constexpr int INF_RING = 2048;               // the recent output of one wavefront in LDS: matches that reach no further back are
                                             // LDS-to-LDS copies; the rest of DEFLATE's 32 KB window is read back from HBM
constexpr int INF_FLUSH = 256;               // bytes that leave the ring for HBM at a time: one aligned 256-byte line of the output
constexpr int INF_NEAR = INF_RING - 512;     // a match at most this far back still lies in the ring (the 512: lanes write up to 63
                                             // bytes ahead of the output position, see the literal and match copies)
constexpr int INF_TBITS = 9;                 // the literal/length table is indexed by the next 9 bits of the stream
#ifndef KBBQ_INF_EXPERIMENT_NOFAR
#define KBBQ_INF_EXPERIMENT_NOFAR 0      // 1: far matches copy nothing (wrong output: a timing experiment, tools/inflate_probe.py)
#endif
#ifndef KBBQ_INF_DBITS
#define KBBQ_INF_DBITS 8
#endif
constexpr int INF_DBITS = KBBQ_INF_DBITS;    // the distance table by the next 8
constexpr int INF_WAVES = 1;                 // one wavefront per workgroup

// status codes of a block
enum : uint32_t { INF_OK = 0, INF_BAD_BLOCK_TYPE = 1, INF_BAD_STORED = 2, INF_BAD_CODE = 3, INF_OVERRUN_IN = 4, INF_OVERRUN_OUT = 5,
                  INF_BAD_DISTANCE = 6, INF_BAD_LENGTHS = 7, INF_SIZE_MISMATCH = 8 };

struct InflateArgs {
    const uint8_t *comp;        // the compressed bytes of the chunk (+ 4 KB readable behind them)
    const uint64_t *c_off;      // per block: where its DEFLATE stream starts in comp
    const uint32_t *c_len;      //            bytes of DEFLATE stream
    const uint64_t *o_off;      //            where its bytes go in out
    const uint32_t *o_len;      //            ISIZE
    uint8_t *out;               // (+ 4 KB writable behind the last block: a block that overruns its size is caught a line late)
    uint32_t n_blocks;
    uint32_t *status;           // per block.  ANY status other than INF_OK invalidates the whole chunk, not only that block's
                                // bytes: a block whose stream overruns its ISIZE stores one full 256-byte line behind its
                                // region -- the head of its neighbour's -- before INF_OVERRUN_OUT is raised, and the first
                                // line of a block is shared with the block before it.  Every caller rejects the chunk
                                // (kbbq_fastq_reader_chunk / _inflate return an error); never keep "the good blocks".
};

// The canonical decoder of one alphabet: lane l (1..15) of `lim` holds the exclusive upper bound of the codes of length
// <= l, left-aligned to 15 bits (0 in every other lane); lane l of `offs` holds (index of the first symbol of length l) -
// (first code of length l).  The symbols in (length, value) order lie in LDS (literal/length alphabet) or, for the two small
// alphabets, in the lanes of one more register.
struct CodeBounds {
    uint32_t lim, offs;
};

// Entries of the literal/length table (u32), found by the next INF_TBITS bits of the stream as they lie in it (LSB first):
//   literals   [3:0] bits of the codes together  [5:4] how many literals (1..3)  [15:8] [23:16] [31:24] the bytes
//   otherwise  [5:4] = 0 and [7:6]: 0 a length code: [3:0] code bits, [12:8] extra bits, [24:16] base length, [28:25] code +
//                                     extra bits
//                                   1 end of block:  [3:0] code bits
//                                   2 a code longer than INF_TBITS bits (or no code at all): decoded by the bounds
//                                   3 a symbol the alphabet does not have (286, 287)
enum : uint32_t { INF_E_LENGTH = 0u << 6, INF_E_EOB = 1u << 6, INF_E_LONG = 2u << 6, INF_E_BAD = 3u << 6 };
// Entries of the distance table (u32), found by the next INF_DBITS bits:
//   [3:0] code bits  [7:4] extra bits  [12:8] both together  [31:16] base distance
//   [13] instead: a code longer than INF_DBITS bits, no code at all, or a symbol the alphabet does not have (30, 31) -- decoded
//        (or refused) by the bounds
enum : uint32_t { INF_D_SLOW = 1u << 13 };

struct InflateLds {
    alignas(16) uint8_t ring[INF_RING];
    uint32_t ll[1 << INF_TBITS];
    union {
        uint32_t dd[1 << INF_DBITS];     // (built last, when the code lengths have served)
        uint8_t lens[320];               // code lengths of a dynamic block (literal/length then distance)
    };
    uint16_t sorted_ll[320];     // literal/length symbols in (length, value) order
    uint16_t sorted_small[64];   // the same of the code-length alphabet, then of the distance alphabet
    alignas(16) uint8_t head[INF_FLUSH];      // the block's first line while it is incomplete in HBM (it starts inside it)
};

// Bounds and sorted symbols of one alphabet from lens[0..n) (LDS).  Uniform control flow; returns false for an
// over-subscribed set of lengths.  An incomplete code is accepted (RFC 1951 allows a single distance code); bit patterns
// it does not define decode as an error later.  *n_coded: how many symbols have a code.
template <int NREG>
__device__ __forceinline__ bool build_code(const uint8_t *lens, int n, uint16_t *sorted, CodeBounds &C, uint32_t *n_coded, int lane) {
    uint32_t my_len[NREG];
#pragma unroll
    for (int j = 0; j < NREG; ++j) { const int s = 64 * j + lane; my_len[j] = s < n ? lens[s] : 0u; }
    uint32_t lim = 0, offs = 0;
    uint32_t code = 0, index = 0;
    int left = 1;
    bool ok = true;
    for (int l = 1; l <= MAX_BITS; ++l) {
        uint64_t b[NREG];
        uint32_t cnt = 0;
#pragma unroll
        for (int j = 0; j < NREG; ++j) { b[j] = __ballot(my_len[j] == (uint32_t)l); cnt += (uint32_t)__popcll(b[j]); }
        left = (left << 1) - (int)cnt;
        if (left < 0) ok = false;
        // lane l: bound of the codes of length <= l, and the offset that turns a code of length l into its symbol's index
        const uint32_t bound = (code + cnt) << (MAX_BITS - l);
        const uint32_t off = index - code;
        if (lane == l) { lim = bound; offs = off; }
        if (cnt) {
#pragma unroll
            for (int j = 0; j < NREG; ++j) {
                if (my_len[j] == (uint32_t)l) sorted[index + (uint32_t)__popcll(b[j] & ((1ull << lane) - 1))] = (uint16_t)(64 * j + lane);
                index += (uint32_t)__popcll(b[j]);
            }
        }
        code = (code + cnt) << 1;
    }
    __builtin_amdgcn_wave_barrier();
    C.lim = lim;
    C.offs = offs;
    *n_coded = index;
    return ok;
}

// A wave-uniform value the compiler takes for a per-lane one: what is computed from it is computed by the vector ALU.
// The decoder's loop is one serial chain per wavefront in uniform control flow, and left to itself the compiler puts all of
// that chain on the scalar unit -- one scalar instruction per cycle and CU, shared by the two dozen wavefronts a CU holds,
// which is what the kernel then waits for (0.8 scalar instructions per cycle and CU, profiles/r04_sq_inflate.txt), while the
// vector ALUs idle.  With the stream's bits and the table entries in vector registers the table index, the extra bits of
// lengths and distances and the byte extraction are computed there; the scalar unit keeps the counters and the branches.
__device__ __forceinline__ uint32_t in_vgpr(uint32_t x) {
    uint32_t r;
    asm("v_mov_b32 %0, %1" : "=v"(r) : "s"(x));
    return r;
}
__device__ __forceinline__ uint32_t in_sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// The bit reader: 256 bytes of the stream per vector register (lane i = dword i), two registers ahead; from them a window of
// two consecutive dwords, the same in every lane (in_vgpr above), and the scalar count of the window's bits already used.
// The next 32 bits of the stream are one v_alignbit away; using bits up moves nothing but the count.  It never reads further
// than two 256-byte pieces behind the end of the stream it was given (a damaged stream cannot walk out of the buffer);
// consumed() against the stream's length tells an overrun.
struct BitReader {
    const uint32_t *base;     // dword-aligned start
    uint32_t w0, w1;          // current and next 64 dwords (per lane)
    uint32_t idx;             // the dword of w0 that enters the window next (0..63)
    uint32_t chunk;           // index of the 256-byte chunk in w0
    uint32_t last_chunk;      // the last chunk that may be loaded
    uint32_t wa, wb;          // the window: dwords D and D + 1 of the stream
    uint32_t used;            // bits of the window behind the reader (< 32 after refill())
    uint32_t skip;            // bits of the first dword that lie before the stream
    __device__ __forceinline__ uint32_t next_dword() {
        const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)w0, (int)idx);
        if (++idx == 64) {
            idx = 0;
            ++chunk;
            w0 = w1;
            const uint32_t next = chunk + 1 < last_chunk ? chunk + 1 : last_chunk;
            w1 = base[(size_t)next * 64 + (threadIdx.x & 63)];
        }
        return d;
    }
    __device__ __forceinline__ void init(const uint8_t *p, uint32_t n_bytes, int lane) {
        const uintptr_t a = (uintptr_t)p;
        base = reinterpret_cast<const uint32_t *>(p - (a & 3));      // (pointer arithmetic: the loads stay global ones)
        last_chunk = (n_bytes + 3 + 255) / 256 + 1;
        w0 = base[lane];
        w1 = base[64 + lane];
        idx = 0;
        chunk = 0;
        wa = in_vgpr(next_dword());
        wb = in_vgpr(next_dword());
        skip = (uint32_t)(a & 3) * 8;
        used = skip;
    }
    __device__ __forceinline__ void refill() {      // 32 valid bits afterwards
        if (used >= 32) {
            const uint32_t d = next_dword();
            // (both registers updated in place: written as two assignments the window travels through copies on every turn
            // of the decoder's loop, also the turns that do not come by here)
            asm("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2" : "+v"(wa), "+v"(wb) : "s"(d));
            used -= 32;
        }
    }
    __device__ __forceinline__ uint64_t consumed() const { return ((uint64_t)chunk * 64 + idx - 2) * 32 + used - skip; }
    __device__ __forceinline__ uint32_t vbits() const { return __builtin_amdgcn_alignbit(wb, wa, used); }      // in a vector register
    __device__ __forceinline__ uint32_t bits() const { return in_sgpr(vbits()); }                             // in a scalar one
    __device__ __forceinline__ uint32_t peek(int n) const { return bits() & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(int n) { used += (uint32_t)n; }      // (at most 32 between two refill()s)
    __device__ __forceinline__ void to_byte_boundary() { drop((int)((0u - (used - skip)) & 7u)); }
    __device__ __forceinline__ uint32_t get(int n) {      // n <= 16
        refill();
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
};

// index (in (length, value) order) of the symbol whose code starts the reader's bits, by the bounds; -1: a bit pattern the
// code does not define.  *len: bits of the code.  The reader holds at least 15 bits.
__device__ __forceinline__ int decode_index(uint32_t bits, const CodeBounds &C, int *len) {
    // the next 15 bits, first bit of the code in the highest place (Huffman codes are packed starting with their MSB)
    const uint32_t v = __brev(bits) >> (32 - MAX_BITS);
    const uint64_t fits = __ballot(v < C.lim);
    if (!fits) return -1;
    const int l = (int)__builtin_ctzll(fits);
    *len = l;
    return (int)((uint32_t)__builtin_amdgcn_readlane((int)C.offs, l) + (v >> (MAX_BITS - l)));
}

// length symbol 257..285 -> base length and extra bits (RFC 1951, 3.2.5)
__device__ __forceinline__ void length_base(int li, int *base, int *eb) {
    if (li < 8) { *base = 3 + li; *eb = 0; }
    else if (li == 28) { *base = 258; *eb = 0; }
    else { *eb = (li >> 2) - 1; *base = 3 + ((4 + (li & 3)) << *eb); }
}

// The literal/length table of a block from its bounds and sorted symbols: every lane decodes its share of the 2^INF_TBITS bit
// patterns the slow way, then looks whether one or two more literals fit behind a literal in the same bits.
__device__ __forceinline__ void build_ll_table(InflateLds &S, const CodeBounds &LL, uint32_t n_coded, int lane) {
    constexpr int PER = (1 << INF_TBITS) / 64;
    uint32_t e[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const uint32_t x = (uint32_t)(64 * j + lane);
        const uint32_t v = (__brev(x) >> (32 - INF_TBITS)) << (MAX_BITS - INF_TBITS);      // the pattern's bits, first in the highest place
        uint32_t len = 0, off = 0;
        for (int l = INF_TBITS; l >= 1; --l) {
            const uint32_t lim_l = (uint32_t)__builtin_amdgcn_readlane((int)LL.lim, l);
            const uint32_t off_l = (uint32_t)__builtin_amdgcn_readlane((int)LL.offs, l);
            if (v < lim_l) { len = (uint32_t)l; off = off_l; }      // the bounds grow with l: the smallest l that fits is written last
        }
        uint32_t ent = INF_E_LONG;
        if (len) {
            const uint32_t index = off + (v >> (MAX_BITS - len));
            const uint32_t sym = index < n_coded ? S.sorted_ll[index] : 287u;
            if (sym < 256) ent = len | (1u << 4) | (sym << 8);
            else if (sym == 256) ent = len | INF_E_EOB;
            else if (sym <= 285) {
                int base, eb;
                length_base((int)sym - 257, &base, &eb);
                ent = len | INF_E_LENGTH | ((uint32_t)eb << 8) | ((uint32_t)base << 16) | ((len + (uint32_t)eb) << 25);
            } else ent = len | INF_E_BAD;
        }
        e[j] = ent;
        S.ll[x] = ent;
    }
    __builtin_amdgcn_wave_barrier();
    // more literals behind a literal: the pattern's remaining bits (zeros above them) find the next code's entry; it counts
    // when that code is a literal and no longer than the bits that are left
    uint32_t packed[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const uint32_t x = (uint32_t)(64 * j + lane);
        uint32_t ent = e[j];
        if ((ent & 0x30u) == 0x10u) {
            const uint32_t l1 = ent & 15u;
            if (l1 < (uint32_t)INF_TBITS) {
                const uint32_t e2 = S.ll[x >> l1];
                const uint32_t l2 = e2 & 15u;
                if ((e2 & 0x30u) == 0x10u && l1 + l2 <= (uint32_t)INF_TBITS) {
                    ent = (l1 + l2) | (2u << 4) | (ent & 0xFF00u) | ((e2 & 0xFF00u) << 8);
                    if (l1 + l2 < (uint32_t)INF_TBITS) {
                        const uint32_t e3 = S.ll[x >> (l1 + l2)];
                        const uint32_t l3 = e3 & 15u;
                        if ((e3 & 0x30u) == 0x10u && l1 + l2 + l3 <= (uint32_t)INF_TBITS)
                            ent = (l1 + l2 + l3) | (3u << 4) | (ent & 0xFFFF00u) | ((e3 & 0xFF00u) << 16);
                    }
                }
            }
        }
        packed[j] = ent;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < PER; ++j) S.ll[64 * j + lane] = packed[j];
    __builtin_amdgcn_wave_barrier();
}

// distance symbol -> base distance | extra bits << 16 (RFC 1951, 3.2.5)
__device__ __forceinline__ uint32_t distance_info(uint32_t ds) {
    if (ds < 4) return 1u + ds;
    const uint32_t de = (ds >> 1) - 1;
    return (1u + ((2u + (ds & 1u)) << de)) | (de << 16);
}

// The distance table of a block, as the literal/length one: every lane decodes its share of the bit patterns by the bounds.
__device__ __forceinline__ void build_dd_table(InflateLds &S, const CodeBounds &DD, uint32_t n_coded, int lane) {
    constexpr int PER = (1 << INF_DBITS) / 64;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const uint32_t x = (uint32_t)(64 * j + lane);
        const uint32_t v = (__brev(x) >> (32 - INF_DBITS)) << (MAX_BITS - INF_DBITS);
        uint32_t len = 0, off = 0;
        for (int l = INF_DBITS; l >= 1; --l) {
            const uint32_t lim_l = (uint32_t)__builtin_amdgcn_readlane((int)DD.lim, l);
            const uint32_t off_l = (uint32_t)__builtin_amdgcn_readlane((int)DD.offs, l);
            if (v < lim_l) { len = (uint32_t)l; off = off_l; }
        }
        uint32_t ent = INF_D_SLOW;
        if (len) {
            const uint32_t index = off + (v >> (MAX_BITS - len));
            const uint32_t ds = index < n_coded ? S.sorted_small[index] : 31u;
            if (ds < 30) {
                const uint32_t info = distance_info(ds);
                ent = len | ((info >> 16) << 4) | ((len + (info >> 16)) << 8) | ((info & 0xFFFFu) << 16);
            }
        }
        S.dd[x] = ent;
    }
    __builtin_amdgcn_wave_barrier();
}

__global__ void __launch_bounds__(64 * INF_WAVES, 6) k_inflate(InflateArgs A) {
    __shared__ InflateLds lds_all[INF_WAVES];
    const int lane = threadIdx.x & 63;
    InflateLds &S = lds_all[0];
    const uint32_t wave = blockIdx.x, n_waves = gridDim.x;
    constexpr uint32_t MASK = INF_RING - 1;
    const uint32_t lit_shift = (8u + 8u * (uint32_t)lane) & 31u;      // lane i < 3 takes byte i of a literal entry
    for (uint32_t blk = wave; blk < A.n_blocks; blk += n_waves) {
        const uint8_t *src = A.comp + A.c_off[blk];
        const uint32_t src_bytes = A.c_len[blk];
        const uint64_t src_bits = (uint64_t)src_bytes * 8;
        uint8_t *dst = A.out + A.o_off[blk];
        const uint32_t want = A.o_len[blk];
        // The ring holds the block's latest INF_RING bytes at the index (address in HBM) mod INF_RING, so that an aligned
        // 256-byte line of the output is an aligned 256-byte piece of the ring: it leaves as 64 coalesced dword stores as
        // soon as it is complete (the block's first and last lines, which it shares with its neighbours, byte by byte).
        // Positions below are `sp` = skew + (bytes produced): the ring index is sp & MASK, the address dst_base + sp.
        const uint32_t skew = (uint32_t)((uintptr_t)dst & MASK);
        uint8_t *dst_base = dst - skew;
        const uint32_t sp_end = skew + want;
        uint32_t sp = skew, fl = skew;                   // produced / flushed
        uint32_t next_line = (skew | (INF_FLUSH - 1)) + 1;      // where the line that holds `fl` ends
        uint32_t err = INF_OK;
        uint64_t bits_before = 0;                        // bits consumed before the reader was last aimed
        BitReader B;
        B.init(src, src_bytes, lane);
        // Whole lines up to sp leave the ring.  Everything inside the symbol loop is wave-uniform control flow (no lane
        // ever branches on its own: the compiler then keeps the loop's branches as the plain scalar jumps they are), so
        // the block's first line -- which starts inside the line, the bytes before belong to the block in front -- is not
        // written byte by byte here: all 256 bytes of the ring's line are parked in `head` and the block's own part of
        // them goes out at the block's end.
        const uint32_t head_end = next_line;             // the first line is [skew, head_end)
        const bool head_whole = (skew & (INF_FLUSH - 1)) == 0;
        auto flush_lines = [&]() {
            while (sp >= next_line && err == INF_OK) {
                if (next_line > sp_end + INF_FLUSH) { err = INF_OVERRUN_OUT; break; }
                const uint32_t line = next_line - INF_FLUSH;      // aligned start of the line that holds fl
                const uint32_t v = *reinterpret_cast<const uint32_t *>(&S.ring[(line & MASK) + 4 * lane]);
                if (fl == line) *reinterpret_cast<uint32_t *>(dst_base + line + 4 * lane) = v;
                else *reinterpret_cast<uint32_t *>(&S.head[4 * lane]) = v;
                fl = next_line;
                next_line += INF_FLUSH;
            }
        };
        bool last = false;
        while (!last && err == INF_OK) {
            last = B.get(1) != 0;
            const uint32_t type = B.get(2);
            if (type == 0) {
                // stored: to the next byte boundary, LEN, ~LEN, the bytes
                B.refill();
                B.to_byte_boundary();
                const uint32_t len = B.get(16), nlen = B.get(16);
                const uint64_t at = bits_before + B.consumed();
                if ((len ^ nlen) != 0xFFFFu) { err = INF_BAD_STORED; break; }
                if (at + (uint64_t)len * 8 > src_bits) { err = INF_OVERRUN_IN; break; }
                if (sp + len > sp_end) { err = INF_OVERRUN_OUT; break; }
                // the reader is at a byte boundary: the bytes straight from the stream, through the ring (later matches may
                // want them there)
                const uint8_t *from = src + (at >> 3);
                for (uint32_t done = 0; done < len;) {
                    const uint32_t n = min(len - done, (uint32_t)1024);
#pragma clang loop vectorize(disable) unroll(disable)
                    for (uint32_t i = lane; i < n; i += 64) S.ring[(sp + i) & MASK] = from[done + i];
                    done += n;
                    sp += n;
                    __builtin_amdgcn_wave_barrier();
                    flush_lines();
                }
                // re-aim the reader behind the stored bytes
                bits_before = at + (uint64_t)len * 8;
                B.init(src + (bits_before >> 3), src_bytes - (uint32_t)(bits_before >> 3), lane);
                continue;
            }
            if (type == 3) { err = INF_BAD_BLOCK_TYPE; break; }
            CodeBounds LL, DD;
            uint32_t n_ll = 0, n_dd = 0;
            if (type == 1) {
                // fixed codes (RFC 1951, 3.2.6)
                for (int s = lane; s < 288; s += 64) S.lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
                __builtin_amdgcn_wave_barrier();
                build_code<5>(S.lens, 288, S.sorted_ll, LL, &n_ll, lane);
                if (lane < 32) S.lens[lane] = 5;
                __builtin_amdgcn_wave_barrier();
                build_code<1>(S.lens, 30, S.sorted_small, DD, &n_dd, lane);
            } else {
                const int hlit = (int)B.get(5) + 257, hdist = (int)B.get(5) + 1, hclen = (int)B.get(4) + 4;
                if (hlit > 286 || hdist > 30) { err = INF_BAD_LENGTHS; break; }
                if (lane < N_CL) S.lens[lane] = 0;
                __builtin_amdgcn_wave_barrier();
                // the code-length code's own lengths arrive in the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15 (RFC 1951,
                // 3.2.7), four bits of a constant per place
                const uint64_t order_tail = 0xF1E2D3C4B5A69780ull;      // places 3..18: 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
                for (int i = 0; i < hclen; ++i) {
                    const uint32_t v = B.get(3);
                    const uint32_t where = i < 3 ? 16u + (uint32_t)i : (uint32_t)((order_tail >> (4 * (i - 3))) & 15);
                    if (lane == 0) S.lens[where] = (uint8_t)v;
                }
                __builtin_amdgcn_wave_barrier();
                CodeBounds CL;
                uint32_t n_cl = 0;
                if (!build_code<1>(S.lens, N_CL, S.sorted_small, CL, &n_cl, lane)) { err = INF_BAD_LENGTHS; break; }
                const uint32_t cl_sym = S.sorted_small[lane];      // lane i: the i-th symbol of the code-length alphabet
                __builtin_amdgcn_wave_barrier();
                // the code lengths of both alphabets, run-length coded
                int i = 0, prev = 0;
                const int total = hlit + hdist;
                while (i < total) {
                    B.refill();
                    int cl = 0;
                    const int ci = decode_index(B.bits(), CL, &cl);
                    if (ci < 0 || (uint32_t)ci >= n_cl) { err = INF_BAD_CODE; break; }
                    B.drop(cl);
                    const int sym = __builtin_amdgcn_readlane((int)cl_sym, ci);
                    int rep = 1, val = sym;
                    if (sym == 16) { if (i == 0) { err = INF_BAD_LENGTHS; break; } rep = 3 + (int)B.get(2); val = prev; }
                    else if (sym == 17) { rep = 3 + (int)B.get(3); val = 0; }
                    else if (sym == 18) { rep = 11 + (int)B.get(7); val = 0; }
                    if (i + rep > total) { err = INF_BAD_LENGTHS; break; }
                    if (lane < rep) S.lens[i + lane] = (uint8_t)val;
                    if (rep > 64 && lane + 64 < rep) S.lens[i + 64 + lane] = (uint8_t)val;
                    if (rep > 128 && lane + 128 < rep) S.lens[i + 128 + lane] = (uint8_t)val;
                    i += rep;
                    prev = val;
                }
                if (err != INF_OK) break;
                if (bits_before + B.consumed() > src_bits) { err = INF_OVERRUN_IN; break; }
                __builtin_amdgcn_wave_barrier();
                // distance lengths follow the literal/length ones: move them to the front of a second array
                uint8_t dl = 0;
                if (lane < hdist) dl = S.lens[hlit + lane];
                __builtin_amdgcn_wave_barrier();
                for (int s = hlit + lane; s < 320; s += 64) S.lens[s] = 0;
                __builtin_amdgcn_wave_barrier();
                if (S.lens[256] == 0) { err = INF_BAD_LENGTHS; break; }      // no end-of-block code
                if (!build_code<5>(S.lens, hlit, S.sorted_ll, LL, &n_ll, lane)) { err = INF_BAD_LENGTHS; break; }
                __builtin_amdgcn_wave_barrier();
                if (lane < 32) S.lens[lane] = lane < hdist ? dl : 0;
                __builtin_amdgcn_wave_barrier();
                if (!build_code<1>(S.lens, hdist, S.sorted_small, DD, &n_dd, lane)) { err = INF_BAD_LENGTHS; break; }
            }
            build_ll_table(S, LL, n_ll, lane);
            build_dd_table(S, DD, n_dd, lane);
            // lane i: base distance and extra bits of the i-th distance symbol in (length, value) order (codes the table leaves out)
            const uint32_t dd_info = (uint32_t)lane < n_dd ? distance_info(S.sorted_small[lane]) : 0u;
            // ---- the block's symbols.  One loop with two ways out (the end-of-block code, a failed line flush): a code or a
            // distance the stream cannot mean is noted in `bad` and replaced by something harmless, and `bad` is looked at with
            // every line and at the block's end -- fewer ways out of the loop are fewer scalar moves on every way round it.
            // The lengths and distances are worked out from the vector copies of the bits and of the table entries.
            uint32_t bad = 0;
            for (;;) {
                // Runs of literals stay in a loop of their own (lookup, store, advance): one to three literals per turn, lane i
                // writes byte i -- the lanes behind them write bytes that the following symbols overwrite (what lies up to 63
                // bytes ahead of the output position is nobody's yet).
                uint32_t vb, vent, ent;
                for (;;) {
                    B.refill();
                    vb = B.vbits();
                    vent = S.ll[vb & ((1u << INF_TBITS) - 1)];      // (every lane the same entry)
                    ent = in_sgpr(vent);
                    if (!(ent & 0x30u)) break;
                    S.ring[(sp + (uint32_t)lane) & MASK] = (uint8_t)(vent >> lit_shift);
                    B.drop((int)(ent & 15u));
                    sp += (ent >> 4) & 3u;
                    if (sp >= next_line) break;
                }
                if (sp >= next_line) {      // (the one place where lines leave; a symbol looked up but not taken is looked up again)
                    __builtin_amdgcn_wave_barrier();
                    flush_lines();
                    if (bad) err = INF_BAD_CODE;
                    if (err != INF_OK) break;
                    continue;
                }
                if (ent & 0xC0u) {      // not a length code of the table
                    if (ent & 0x80u) {
                        // A code longer than the table's index, or none: decoded by the bounds into the entry a table wide
                        // enough would have held (what the stream cannot mean becomes a literal 0 of one bit, and is noted).
                        int cl = 1;
                        int ci = (ent & 0xC0u) == INF_E_LONG ? decode_index(in_sgpr(vb), LL, &cl) : -1;
                        uint32_t sym = 0;
                        if (ci < 0 || (uint32_t)ci >= n_ll) bad = 1;
                        else sym = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.sorted_ll[ci]);
                        if (sym > 285) { bad = 1; sym = 0; }
                        if (sym < 256) {
                            S.ring[sp & MASK] = (uint8_t)sym;      // (every lane the same byte to the same place)
                            B.drop(cl);
                            ++sp;
                            continue;
                        }
                        if (sym == 256) ent = (uint32_t)cl | INF_E_EOB;
                        else {
                            int base, eb;
                            length_base((int)sym - 257, &base, &eb);
                            ent = (uint32_t)cl | INF_E_LENGTH | ((uint32_t)eb << 8) | ((uint32_t)base << 16) | ((uint32_t)(cl + eb) << 25);
                        }
                        vent = in_vgpr(ent);
                    }
                    if (ent & 0x40u) {      // the end-of-block code
                        B.drop((int)(ent & 15u));
                        break;
                    }
                }
                // a length: base + extra bits
                const uint32_t vlen = ((vent >> 16) & 511u) + __builtin_amdgcn_ubfe(vb, vent & 15u, (vent >> 8) & 31u);
                B.drop((int)((ent >> 25) & 31u));
                // the distance
                B.refill();
                const uint32_t vb2 = B.vbits();
                uint32_t vde = S.dd[vb2 & ((1u << INF_DBITS) - 1)];
                uint32_t de = in_sgpr(vde);
                if (de & INF_D_SLOW) {
                    int dl = 1;
                    const int di = decode_index(in_sgpr(vb2), DD, &dl);
                    if (di < 0 || (uint32_t)di >= n_dd) { bad = 1; de = 1u | (1u << 8) | (1u << 16); }      // (one bit, distance 1)
                    else {
                        const uint32_t dinfo = (uint32_t)__builtin_amdgcn_readlane((int)dd_info, di);
                        de = (uint32_t)dl | ((dinfo >> 16) << 4) | (((uint32_t)dl + (dinfo >> 16)) << 8) | ((dinfo & 0xFFFFu) << 16);
                    }
                    vde = in_vgpr(de);
                }
                const uint32_t vdist = (vde >> 16) + __builtin_amdgcn_ubfe(vb2, vde & 15u, (vde >> 4) & 15u);
                B.drop((int)((de >> 8) & 31u));
                const uint32_t len = in_sgpr(vlen);
                uint32_t dist = in_sgpr(vdist);
                if (dist > sp - skew) { bad = 1; dist = 1; }      // (then a copy inside the ring, whatever lies there)
                // The copy: byte i of the match is byte (i mod dist) of the dist bytes before it.  A source inside the ring is an
                // LDS-to-LDS copy; a source further back has left the ring long ago (at least 1 KB of output lies between it and
                // the flush frontier) and is read back from HBM -- behind this wave's own stores of those lines, which the
                // memory pipeline of a CU keeps in order.  Like the literals, the lanes behind the match's length copy bytes
                // nobody owns yet.
                __builtin_amdgcn_wave_barrier();
                if (dist <= (uint32_t)INF_NEAR) {
                    if (len <= min(dist, 64u)) {
                        // source and destination apart, one round: all reads, then all writes
                        const uint8_t v = S.ring[(sp - dist + (uint32_t)lane) & MASK];
                        S.ring[(sp + (uint32_t)lane) & MASK] = v;
                    } else {
                        // rounds of up to 64 bytes whose source is finished: of a match that runs into itself the first `dist`
                        // bytes, then twice as many, ... (the stride stays a multiple of dist)
                        uint32_t c = dist;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                        for (uint32_t done = 0; done < len;) {
                            const uint32_t n = min(c, 64u);
                            const uint8_t v = S.ring[(sp + done - c + (uint32_t)lane) & MASK];
                            __builtin_amdgcn_wave_barrier();
                            S.ring[(sp + done + (uint32_t)lane) & MASK] = v;      // (lanes from n on: bytes nobody owns yet)
                            __builtin_amdgcn_wave_barrier();
                            done += n;
                            if (c < 64u) c *= 2;
                        }
                    }
                } else {
                    // (lanes behind the match's length read bytes of this block's own output that may not have arrived yet:
                    // they land ahead of the output position like every other such byte)
                    // The block's first line is not in HBM before the block's end: its bytes come from `head`.
                    const uint32_t head_stop = head_whole ? 0u : head_end;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                    for (uint32_t i0 = 0; i0 < (KBBQ_INF_EXPERIMENT_NOFAR ? 0u : len); i0 += 64) {
                        const uint32_t from = sp - dist + i0 + (uint32_t)lane;
                        const uint8_t g = dst_base[from], h = S.head[from & (INF_FLUSH - 1)];
                        S.ring[(sp + i0 + (uint32_t)lane) & MASK] = from < head_stop ? h : g;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                sp += len;
            }
            if (err == INF_OK && sp >= next_line) { __builtin_amdgcn_wave_barrier(); flush_lines(); }      // (behind the last symbol)
            if (bad && err == INF_OK) err = INF_BAD_CODE;
            if (err == INF_OK && bits_before + B.consumed() > src_bits) err = INF_OVERRUN_IN;
        }
        if (err == INF_OK && sp != sp_end) err = sp > sp_end ? INF_OVERRUN_OUT : INF_SIZE_MISMATCH;
        // what is left in the ring
        __builtin_amdgcn_wave_barrier();
        if (err == INF_OK) {
            // the first line's own bytes: parked in `head` if the line was completed, otherwise still in the ring
            if (!head_whole) {
                const uint32_t stop = min(head_end, sp);
                const bool parked = fl >= head_end;
#pragma clang loop vectorize(disable) unroll(disable)
                for (uint32_t o = skew + lane; o < stop; o += 64) dst_base[o] = parked ? S.head[o & (INF_FLUSH - 1)] : S.ring[o & MASK];
            }
#pragma clang loop vectorize(disable) unroll(disable)
            for (uint32_t o = max(fl, head_whole ? skew : head_end) + lane; o < sp; o += 64) dst_base[o] = S.ring[o & MASK];
        }
        if (lane == 0) A.status[blk] = err;
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- the blocks' CRC-32 (what bgzf_read checks per block: "CRC32 checksum mismatch"): one wavefront per block over the
// inflated bytes, against the value in the block's trailer; a block whose inflation already failed keeps that status
enum : uint32_t { INF_BAD_CRC = 9 };
__global__ void __launch_bounds__(256) k_block_crc(InflateArgs A) {
    __shared__ uint32_t tab[256];
    tab[threadIdx.x] = crc_table_entry(threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int xq_for = -1;
    uint32_t xq = 0, xq4 = 0;
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    for (uint32_t blk = wave; blk < A.n_blocks; blk += n_waves) {
        if (A.status[blk] != INF_OK) continue;
        const uint8_t *t = A.comp + A.c_off[blk] + A.c_len[blk];      // CRC32 then ISIZE, little-endian (RFC 1952)
        const uint32_t want = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        const uint32_t got = wave_crc32(A.out + A.o_off[blk], (int)A.o_len[blk], tab, lane, xq_for, xq, xq4);
        if (lane == 0 && got != want && !KBBQ_INF_EXPERIMENT_NOFAR) A.status[blk] = INF_BAD_CRC;
    }
}

// ---- exclusive scan of u64 values in place, three launches (tiles of 2048, their sums by one workgroup, the offsets back)
constexpr int DSCAN_TILE = 2048;
__global__ void __launch_bounds__(256) k_dscan_tiles(uint64_t *data, uint64_t n, uint64_t *tile_sums) {
    __shared__ uint64_t wave_tot[4];
    const uint64_t base = (uint64_t)blockIdx.x * DSCAN_TILE + (uint64_t)threadIdx.x * 8;
    uint64_t v[8], run = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = base + j < n ? data[base + j] : 0; const uint64_t x = v[j]; v[j] = run; run += x; }
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
__global__ void __launch_bounds__(1024) k_dscan_sums(uint64_t *tile_sums, uint64_t n_tiles, uint64_t *total) {
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
__global__ void __launch_bounds__(256) k_dscan_add(uint64_t *data, uint64_t n, const uint64_t *tile_sums) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) data[i] += tile_sums[i / DSCAN_TILE];
}

// ---- where the lines start ------------------------------------------------------------------------------------------------
// newlines per tile of 16 KB (256 lanes x 64 bytes), then -- behind the scan of the tile counts -- their positions
constexpr int NL_TILE = 16384;
__device__ __forceinline__ uint64_t newline_bits(const uint8_t *p, uint64_t avail) {      // bit i: p[i] == '\n', i < min(64, avail)
    uint64_t m = 0;
    if (avail >= 64) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint4 v = *reinterpret_cast<const uint4 *>(p + 16 * w);
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (((d[k] >> (8 * b)) & 0xFF) == 10u) m |= 1ull << (16 * w + 4 * k + b);
        }
    } else {
        for (uint64_t i = 0; i < avail; ++i) if (p[i] == 10) m |= 1ull << i;
    }
    return m;
}
__global__ void __launch_bounds__(256) k_count_newlines(const uint8_t *text, uint64_t n, uint64_t *tile_counts) {
    __shared__ uint32_t wave_cnt[4];
    const uint64_t at = (uint64_t)blockIdx.x * NL_TILE + (uint64_t)threadIdx.x * 64;
    const uint32_t c = at < n ? (uint32_t)__popcll(newline_bits(text + at, n - at)) : 0u;
    uint32_t s = c;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = (uint64_t)wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}
__global__ void __launch_bounds__(256) k_newline_positions(const uint8_t *text, uint64_t n, const uint64_t *tile_first, uint32_t *nl_pos,
                                                            uint64_t nl_capacity) {
    __shared__ uint32_t wave_cnt[4];
    const uint64_t at = (uint64_t)blockIdx.x * NL_TILE + (uint64_t)threadIdx.x * 64;
    uint64_t m = at < n ? newline_bits(text + at, n - at) : 0ull;
    const uint32_t c = (uint32_t)__popcll(m);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = c;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o); if (lane >= o) inc += y; }
    if (lane == 63) wave_cnt[w] = inc;
    __syncthreads();
    uint64_t idx = tile_first[blockIdx.x] + (inc - c);
    for (int i = 0; i < w; ++i) idx += wave_cnt[i];
    while (m) {
        const int b = __builtin_ctzll(m);
        m &= m - 1;
        if (idx < nl_capacity) nl_pos[idx] = (uint32_t)(at + (uint64_t)b);
        ++idx;
    }
}

// ---- records ------------------------------------------------------------------------------------------------------------------
// Record r is lines 4r .. 4r+3 of the text.  kseq's reading of a four-line record (htsiter.cc:52-59): the name is the
// header line behind '@' up to the first white-space character, the comment what follows that character; the third line
// only has to start with '+'; sequence and quality lines are equally long.  The read-name rules of the FASTQ constructor
// (readutils.cc:74-97): the part before the first '_' names the read, "/2" at its end makes it second-in-pair; a later
// field "RG:..." would name a read group -- that needs the dictionary of the host path, as do all shapes other than
// this one (flagged, and the caller falls back to the serial reader, which stays the definition).
struct FastqIndex {
    uint32_t *name_off, *name_len, *com_off, *com_len, *seq_off, *seq_len, *qual_off;      // per record, offsets into the text
    uint8_t *second;
    uint64_t *base_sz;        // per record: seq_len (u64, scanned into base offsets)
    uint64_t *text_sz;        // per record: bytes of its output text (scanned into text offsets)
    uint32_t *flags;          // [0] bit 0: a shape the device path does not take; bit 1: a read name shorter than 2 characters
                              // [1] longest read  [2] shortest read
};
__device__ __forceinline__ bool is_space(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }

__global__ void __launch_bounds__(256) k_fastq_records(const uint8_t *text, const uint32_t *nl_pos, uint64_t n_records, FastqIndex X) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_records) return;
    const uint32_t l0 = r ? nl_pos[4 * r - 1] + 1 : 0u;
    const uint32_t e0 = nl_pos[4 * r], e1 = nl_pos[4 * r + 1], e2 = nl_pos[4 * r + 2], e3 = nl_pos[4 * r + 3];
    const uint32_t l1 = e0 + 1, l2 = e1 + 1, l3 = e2 + 1;
    uint32_t bad = 0;
    if (e0 == l0 || text[l0] != '@') bad |= 1;
    if (e2 == l2 || text[l2] != '+') bad |= 1;
    const uint32_t sl = e1 - l1, ql = e3 - l3;
    if (sl != ql || sl == 0) bad |= 1;
    // a carriage return before any of the four newlines: kseq would strip it; not this path
    if ((e0 > l0 && text[e0 - 1] == 13) || (e1 > l1 && text[e1 - 1] == 13) || (e2 > l2 && text[e2 - 1] == 13) || (e3 > l3 && text[e3 - 1] == 13)) bad |= 1;
    // name and comment
    uint32_t p = l0 + 1;
    while (p < e0 && !is_space(text[p])) ++p;
    const uint32_t nl = p > l0 ? p - (l0 + 1) : 0u;
    const uint32_t c0 = p < e0 ? p + 1 : e0, cl = e0 - c0;
    if (nl == 0) bad |= 1;
    // the read-name rules
    uint32_t first_len = nl;
    for (uint32_t i = 0; i < nl; ++i)
        if (text[l0 + 1 + i] == '_') {
            if (first_len == nl) first_len = i;
            if (i + 3 < nl && text[l0 + 2 + i] == 'R' && text[l0 + 3 + i] == 'G' && text[l0 + 4 + i] == ':') bad |= 1;      // a read-group field
        }
    if (first_len < 2) bad |= 2;
    const bool second = first_len >= 2 && text[l0 + 1 + first_len - 2] == '/' && text[l0 + 1 + first_len - 1] == '2';
    X.name_off[r] = l0 + 1; X.name_len[r] = nl; X.com_off[r] = c0; X.com_len[r] = cl;
    X.seq_off[r] = l1; X.seq_len[r] = sl; X.qual_off[r] = l3;
    X.second[r] = second ? 1 : 0;
    X.base_sz[r] = sl;
    X.text_sz[r] = (uint64_t)nl + cl + 2 * (uint64_t)sl + 6;
    if (bad) atomicOr(&X.flags[0], bad);
    atomicMax(&X.flags[1], sl);
    atomicMin(&X.flags[2], sl);
}

// sequence text and qualities of every record into the batch's contiguous arrays (one wavefront per record)
__global__ void __launch_bounds__(256) k_fastq_gather(const uint8_t *text, FastqIndex X, const uint64_t *base_off, uint64_t n_records,
                                                       uint8_t *seq_text, uint8_t *qual) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
    for (uint64_t r = wave; r < n_records; r += n_waves) {
        const uint32_t sl = X.seq_len[r];
        const uint8_t *s = text + X.seq_off[r], *q = text + X.qual_off[r];
        const uint64_t at = base_off[r];
        for (uint32_t i = lane; i < sl; i += 64) {
            seq_text[at + i] = s[i];
            qual[at + i] = (uint8_t)(q[i] - 33);      // readutils.cc:70-71
        }
    }
}

// 64 bases per lane: the 2-bit words, the non-ACGT mask and the off-case bits (kbbq_pack_bases_case, engine.hip: same table)
// n_offcase[0] counts the off-case bases, n_offcase[1] the characters that the packed form cannot give back (anything but
// ACGTN and acgt: digits, IUPAC codes, a lower-case n)
__global__ void __launch_bounds__(256) k_pack_text(const uint8_t *seq_text, uint64_t n_bases, uint64_t *bases, uint64_t *nmask, uint64_t *offcase,
                                                    unsigned long long *n_offcase) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t words = n_bases / 64 + 1;
    if (w >= words) return;
    const uint64_t first = w * 64;
    const int n = (int)min((uint64_t)64, n_bases > first ? n_bases - first : 0);
    uint64_t b0 = 0, b1 = 0, nm = 0, oc = 0;
    uint32_t exotic = 0;
    for (int j = 0; j < n; ++j) {
        const uint8_t ch = seq_text[first + j];
        exotic += !(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T' || ch == 'N' || ch == 'a' || ch == 'c' || ch == 'g' || ch == 't');
        // seq_nt16_int[seq_nt16_table[ch]] (bloom.hh:351): A/a/0 = 0, C/c/1 = 1, G/g/2 = 2, T/t/3 = 3, everything else non-ACGT
        uint32_t code = 4, odd = 0;
        switch (ch) {
            case 'A': code = 0; break; case 'C': code = 1; break; case 'G': code = 2; break; case 'T': code = 3; break;
            case 'a': case '0': code = 0; odd = 1; break; case 'c': case '1': code = 1; odd = 1; break;
            case 'g': case '2': code = 2; odd = 1; break; case 't': case '3': code = 3; odd = 1; break;
            default: break;
        }
        const uint64_t c2 = code & 3 & (code < 4 ? 3u : 0u);
        if (j < 32) b0 |= c2 << (2 * j); else b1 |= c2 << (2 * (j - 32));
        nm |= (uint64_t)(code >> 2) << j;
        oc |= (uint64_t)odd << j;
    }
    bases[2 * w] = b0;
    bases[2 * w + 1] = b1;
    nmask[w] = nm;
    offcase[w] = oc;
    if (oc) atomicAdd(n_offcase, (unsigned long long)__popcll(oc));
    if (exotic) atomicAdd(n_offcase + 1, (unsigned long long)exotic);
}

// the output text of a batch assembled from the DEVICE copy of the input text (pass 4 of the device path): the same
// "@name\nseq\n+comment\nqual\n" as k_fastq_text, the pieces found by the record index
__global__ void __launch_bounds__(256) k_fastq_text_indexed(const uint8_t *text, FastqIndex X, const uint64_t *text_off, const uint64_t *base_off,
                                                             const uint8_t *new_qual, uint64_t n_records, uint8_t *out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
    for (uint64_t r = wave; r < n_records; r += n_waves) {
        const uint32_t nl = X.name_len[r], cl = X.com_len[r], sl = X.seq_len[r];
        const uint8_t *name = text + X.name_off[r], *comment = text + X.com_off[r], *seq = text + X.seq_off[r];
        const SeqSource from = {seq, nullptr, nullptr, nullptr, 0};
        emit_fastq_record(lane, name, nl, comment, cl, from, sl, new_qual + base_off[r], out + text_off[r]);
    }
}

// A chunk kept without its text (kbbq_fastq_reader_keep): the names and comments of its records back to back -- record r's at
// text_off[r] - 2 base_off[r] - 6 r, the sum of the name and comment lengths before it (a record's output text is name +
// comment + 2 x sequence + 6 bytes) -- and the two lengths; the sequence line comes back from the packed batch.
__global__ void __launch_bounds__(256) k_fastq_keep_names(const uint8_t *text, FastqIndex X, const uint64_t *text_off, const uint64_t *base_off,
                                                           uint64_t n_records, uint8_t *names, uint32_t *lens) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
    for (uint64_t r = wave; r < n_records; r += n_waves) {
        const uint32_t nl = X.name_len[r], cl = X.com_len[r];
        uint8_t *to = names + (text_off[r] - 2 * base_off[r] - 6 * r);
        const uint8_t *name = text + X.name_off[r], *comment = text + X.com_off[r];
        for (uint32_t i = lane; i < nl + cl; i += 64) to[i] = i < nl ? name[i] : comment[i - nl];
        if (lane == 0) { lens[2 * r] = nl; lens[2 * r + 1] = cl; }
    }
}
__global__ void __launch_bounds__(256) k_fastq_text_packed(const uint8_t *names, const uint32_t *lens, const uint64_t *text_off, const uint64_t *base_off,
                                                            const uint64_t *bases, const uint64_t *nmask, const uint64_t *offcase,
                                                            const uint8_t *new_qual, uint64_t n_records, uint8_t *out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
    for (uint64_t r = wave; r < n_records; r += n_waves) {
        const uint32_t nl = lens[2 * r], cl = lens[2 * r + 1];
        const uint64_t b0 = base_off[r];
        const uint32_t sl = (uint32_t)(base_off[r + 1] - b0);
        const uint8_t *name = names + (text_off[r] - 2 * b0 - 6 * r);
        const SeqSource from = {nullptr, bases, nmask, offcase, b0};
        emit_fastq_record(lane, name, nl, name + nl, cl, from, sl, new_qual + b0, out + text_off[r]);
    }
}

}  // namespace dfl
}  // namespace kbbq
