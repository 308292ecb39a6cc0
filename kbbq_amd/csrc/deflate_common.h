// deflate_common.h -- the scalar parts of a DEFLATE (RFC 1951) encoder, shared by the device kernels of
// bgzf_device.h and by the host-only entry points that the CPU tests drive: symbol tables, Huffman code lengths,
// canonical codes, the code-length header, CRC-32 combination.  No I/O, no allocation; everything works on plain
// arrays handed in by the caller, so the same functions run on one lane of a wavefront and on the host.
//
// This is the output side of the reference's writer: FastqFile::write / BamFile::write hand their bytes to htslib's
// BGZF layer (htsiter.cc:45,75-86), which deflates them block by block.  The decompressed stream is what the
// reference defines; the compressed bytes are the encoder's own choice.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DFL_HD __host__ __device__ __forceinline__
#else
#define DFL_HD inline
#endif

namespace kbbq {
namespace dfl {

constexpr int N_LL = 286;        // literal / length symbols 0..285 (256 = end of block)
constexpr int N_D = 30;          // distance symbols
constexpr int N_CL = 19;         // code-length symbols
constexpr int MAX_BITS = 15;     // longest literal / length and distance code
constexpr int MAX_CL_BITS = 7;   // longest code-length code
constexpr int MIN_MATCH = 3, MAX_MATCH = 258, MAX_DIST = 32768;

DFL_HD int floor_log2(uint32_t x) { return 31 - __builtin_clz(x); }

// match length 3..258 -> length symbol 257..285, number and value of its extra bits (RFC 1951, 3.2.5)
DFL_HD void length_symbol(int len, int &sym, int &ebits, int &eval) {
    const int l = len - 3;
    if (l < 8) { sym = 257 + l; ebits = 0; eval = 0; return; }
    if (l == 255) { sym = 285; ebits = 0; eval = 0; return; }
    const int e = floor_log2((uint32_t)l) - 2;
    sym = 261 + 4 * e + ((l >> e) & 3);
    ebits = e;
    eval = l & ((1 << e) - 1);
}
// match distance 1..32768 -> distance symbol 0..29 and its extra bits
DFL_HD void distance_symbol(int dist, int &sym, int &ebits, int &eval) {
    const int d = dist - 1;
    if (d < 4) { sym = d; ebits = 0; eval = 0; return; }
    const int hb = floor_log2((uint32_t)d), e = hb - 1;
    sym = 2 * hb + ((d >> e) & 1);
    ebits = e;
    eval = d & ((1 << e) - 1);
}

DFL_HD uint32_t reverse_bits(uint32_t v, int n) {
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
}

// ---- Huffman code lengths -----------------------------------------------------------------------------------------
// Minimum-redundancy code lengths of the symbols with a non-zero count, limited to max_bits.
//   freq[n]        symbol counts
//   len[n]         out: code length per symbol (0 = unused)
//   order[n], w[n] scratch (uint16_t / uint32_t)
// At least two symbols get a code (a lone symbol is paired with a dummy: inflate implementations want a complete
// code).  Lengths come from the in-place algorithm of Moffat and Katajainen on the counts sorted upwards; when the
// longest code exceeds max_bits the counts below a floor are raised to it and the floor doubles until it fits.
// The code lengths of m >= 2 symbols order[0..m) whose counts w[0..m) are sorted upwards (overwritten): len[order[i]] is set
// when the longest code fits in max_bits, otherwise nothing is written and false returned.
DFL_HD bool lengths_from_sorted(int m, const uint16_t *order, uint32_t *w, int max_bits, uint8_t *len) {
    if (m == 2) {
        w[0] = w[1] = 1;
    } else {
        // phase 1: pair the two smallest items, leaves or internal nodes, left to right
        w[0] += w[1];
        int root = 0, leaf = 2;
        for (int next = 1; next < m - 1; ++next) {
            if (leaf >= m || w[root] < w[leaf]) { w[next] = w[root]; w[root++] = (uint32_t)next; }
            else w[next] = w[leaf++];
            if (leaf >= m || (root < next && w[root] < w[leaf])) { w[next] += w[root]; w[root++] = (uint32_t)next; }
            else w[next] += w[leaf++];
        }
        // phase 2: parent pointers -> depths of the internal nodes
        w[m - 2] = 0;
        for (int next = m - 3; next >= 0; --next) w[next] = w[w[next]] + 1;
        // phase 3: depths of the internal nodes -> depths of the leaves
        int avail = 1, used = 0, depth = 0, rootp = m - 2, next = m - 1;
        while (avail > 0) {
            while (rootp >= 0 && (int)w[rootp] == depth) { ++used; --rootp; }
            while (avail > used) { w[next--] = (uint32_t)depth; --avail; }
            avail = 2 * used;
            ++depth;
            used = 0;
        }
    }
    // (the rarest symbol comes first: the longest code)
    if ((int)w[0] > max_bits) return false;
    for (int i = 0; i < m; ++i) len[order[i]] = (uint8_t)w[i];
    return true;
}

DFL_HD void huffman_lengths(const uint32_t *freq, int n, int max_bits, uint8_t *len, uint16_t *order, uint32_t *w) {
    for (uint32_t floor_count = 1;; floor_count <<= 1) {
        int m = 0;
        for (int s = 0; s < n; ++s) {
            len[s] = 0;
            if (freq[s]) order[m++] = (uint16_t)s;
        }
        // fewer than two used symbols: add the lowest unused ones with a count of one
        for (int s = 0; m < 2 && s < n; ++s) {
            bool used = false;
            for (int i = 0; i < m; ++i) used = used || order[i] == s;
            if (!used) order[m++] = (uint16_t)s;
        }
        auto count_of = [&](int s) -> uint32_t { const uint32_t f = freq[s] ? freq[s] : 1u; return f < floor_count ? floor_count : f; };
        // Shell sort by (count, symbol): at most 286 entries, a few dozen for text
        const int gaps[6] = {132, 57, 23, 10, 4, 1};
        for (int g = 0; g < 6; ++g) {
            const int gap = gaps[g];
            for (int i = gap; i < m; ++i) {
                const uint16_t s = order[i];
                const uint32_t c = count_of(s);
                int j = i - gap;
                while (j >= 0 && (count_of(order[j]) > c || (count_of(order[j]) == c && order[j] > s))) { order[j + gap] = order[j]; j -= gap; }
                order[j + gap] = s;
            }
        }
        for (int i = 0; i < m; ++i) w[i] = count_of(order[i]);
        if (lengths_from_sorted(m, order, w, max_bits, len)) return;
    }
}

// canonical codes of the given lengths, already bit-reversed for DEFLATE's LSB-first packing
DFL_HD void canonical_codes(const uint8_t *len, int n, int max_bits, uint16_t *code) {
    uint16_t count[MAX_BITS + 1], next[MAX_BITS + 2];
    for (int b = 0; b <= max_bits; ++b) count[b] = 0;
    for (int s = 0; s < n; ++s) ++count[len[s]];
    count[0] = 0;
    uint32_t c = 0;
    next[0] = 0;
    for (int b = 1; b <= max_bits; ++b) { c = (c + count[b - 1]) << 1; next[b] = (uint16_t)c; }
    for (int s = 0; s < n; ++s) code[s] = len[s] ? (uint16_t)reverse_bits(next[len[s]]++, len[s]) : (uint16_t)0;
}

// ---- the code-length header (RFC 1951, 3.2.7) -----------------------------------------------------------------------
// Run-length form of `n` code lengths: entries (symbol | extra_value << 8) with symbol 0..18.  Returns the number of
// entries (at most n).
DFL_HD int code_length_runs(const uint8_t *lens, int n, uint16_t *out) {
    int m = 0;
    for (int i = 0; i < n;) {
        const int v = lens[i];
        int run = 1;
        while (i + run < n && lens[i + run] == v) ++run;
        i += run;
        if (v == 0) {
            while (run >= 11) { const int r = run < 138 ? run : 138; out[m++] = (uint16_t)(18 | ((r - 11) << 8)); run -= r; }
            if (run >= 3) { out[m++] = (uint16_t)(17 | ((run - 3) << 8)); run = 0; }
            while (run-- > 0) out[m++] = 0;
        } else {
            out[m++] = (uint16_t)v;
            --run;
            while (run >= 3) { const int r = run < 6 ? run : 6; out[m++] = (uint16_t)(16 | ((r - 3) << 8)); run -= r; }
            while (run-- > 0) out[m++] = (uint16_t)v;
        }
    }
    return m;
}
DFL_HD int cl_extra_bits(int sym) { return sym == 16 ? 2 : sym == 17 ? 3 : sym == 18 ? 7 : 0; }

// order in which the code lengths of the code-length alphabet are stored
DFL_HD int cl_order(int i) {
    const uint8_t o[N_CL] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    return o[i];
}

// A little-endian bit writer over a zeroed byte buffer (one lane, or the host)
struct BitSink {
    uint8_t *p;
    uint64_t bitpos;      // bits put so far (whole bytes of them are in p, the rest in acc)
    uint64_t acc;
    int n_acc;            // < 8 between calls
    DFL_HD void put(uint64_t value, int nbits) {      // nbits <= 56; bits of value above nbits are zero
        acc |= value << n_acc;
        n_acc += nbits;
        bitpos += (uint64_t)nbits;
        while (n_acc >= 8) { p[(bitpos - (uint64_t)n_acc) >> 3] = (uint8_t)acc; acc >>= 8; n_acc -= 8; }
    }
    DFL_HD void finish() { if (n_acc) p[(bitpos - (uint64_t)n_acc) >> 3] = (uint8_t)acc; }      // the last, partial byte (zero bits above)
};

// Everything one dynamic block needs beside its tokens: code lengths and codes of the two alphabets and the header
// bits in front of the first token.  `head` must be zeroed and hold 320 bytes (3 + 14 + 19 * 3 + 316 * 14 bits at most
// = 568 bytes is the theoretical bound for pathological lengths; the run-length form keeps real headers under 200).
struct BlockCodes {
    uint8_t ll_len[N_LL], d_len[N_D];
    uint16_t ll_code[N_LL], d_code[N_D];
    uint32_t head_bits;
};
constexpr int HEAD_BYTES = 640;

// scratch: order[N_LL], w[N_LL], runs[N_LL + N_D], all[N_LL + N_D] (on the device all of it in LDS: an array of the
// function's own would live in scratch memory, a trip to HBM per access)
// ll_lengths_done: B.ll_len holds the literal/length code lengths already (the device sorts that alphabet with all its lanes)
DFL_HD void build_block_codes(const uint32_t *ll_freq, const uint32_t *d_freq, BlockCodes &B, uint8_t *head, uint16_t *order, uint32_t *w,
                              uint16_t *runs, uint8_t *all, bool ll_lengths_done = false) {
    if (!ll_lengths_done) huffman_lengths(ll_freq, N_LL, MAX_BITS, B.ll_len, order, w);
    huffman_lengths(d_freq, N_D, MAX_BITS, B.d_len, order, w);
    canonical_codes(B.ll_len, N_LL, MAX_BITS, B.ll_code);
    canonical_codes(B.d_len, N_D, MAX_BITS, B.d_code);
    int n_ll = N_LL, n_d = N_D;
    while (n_ll > 257 && B.ll_len[n_ll - 1] == 0) --n_ll;
    while (n_d > 1 && B.d_len[n_d - 1] == 0) --n_d;
    for (int i = 0; i < n_ll; ++i) all[i] = B.ll_len[i];
    for (int i = 0; i < n_d; ++i) all[n_ll + i] = B.d_len[i];
    const int n_runs = code_length_runs(all, n_ll + n_d, runs);
    uint32_t cl_freq[N_CL];
    uint8_t cl_len[N_CL];
    uint16_t cl_code[N_CL];
    for (int i = 0; i < N_CL; ++i) cl_freq[i] = 0;
    for (int i = 0; i < n_runs; ++i) ++cl_freq[runs[i] & 0xFF];
    huffman_lengths(cl_freq, N_CL, MAX_CL_BITS, cl_len, order, w);
    canonical_codes(cl_len, N_CL, MAX_CL_BITS, cl_code);
    int n_cl = N_CL;
    while (n_cl > 4 && cl_len[cl_order(n_cl - 1)] == 0) --n_cl;
    BitSink s = {head, 0, 0, 0};
    s.put(1, 1);      // BFINAL: a BGZF block is one complete DEFLATE stream
    s.put(2, 2);      // BTYPE = dynamic Huffman
    s.put((uint32_t)(n_ll - 257), 5);
    s.put((uint32_t)(n_d - 1), 5);
    s.put((uint32_t)(n_cl - 4), 4);
    for (int i = 0; i < n_cl; ++i) s.put(cl_len[cl_order(i)], 3);
    for (int i = 0; i < n_runs; ++i) {
        const int sym = runs[i] & 0xFF;
        s.put(cl_code[sym], cl_len[sym]);
        const int eb = cl_extra_bits(sym);
        if (eb) s.put((uint32_t)(runs[i] >> 8), eb);
    }
    s.finish();
    B.head_bits = (uint32_t)s.bitpos;
}

// A token: a literal byte, or a match.  Packed in 32 bits: bit 31 = match, bits 8..22 = distance - 1, bits 0..7 =
// length - 3 (match) or the byte (literal).
DFL_HD uint32_t token_literal(uint8_t b) { return b; }
DFL_HD uint32_t token_match(int len, int dist) { return 0x80000000u | ((uint32_t)(dist - 1) << 8) | (uint32_t)(len - 3); }
DFL_HD bool token_is_match(uint32_t t) { return (t >> 31) != 0; }
DFL_HD int token_len(uint32_t t) { return (int)(t & 0xFF) + 3; }
DFL_HD int token_dist(uint32_t t) { return (int)((t >> 8) & 0x7FFF) + 1; }

// the bits of one token under the block's codes: value (LSB first, up to 48 bits) and length
DFL_HD void token_bits(uint32_t t, const BlockCodes &B, uint64_t &value, int &nbits) {
    if (!token_is_match(t)) {
        value = B.ll_code[t & 0xFF];
        nbits = B.ll_len[t & 0xFF];
        return;
    }
    int ls, leb, lev, ds, deb, dev;
    length_symbol(token_len(t), ls, leb, lev);
    distance_symbol(token_dist(t), ds, deb, dev);
    uint64_t v = B.ll_code[ls];
    int nb = B.ll_len[ls];
    v |= (uint64_t)lev << nb; nb += leb;
    v |= (uint64_t)B.d_code[ds] << nb; nb += B.d_len[ds];
    v |= (uint64_t)dev << nb; nb += deb;
    value = v;
    nbits = nb;
}

// ---- CRC-32 (the gzip one: polynomial 0xEDB88320 reflected) ----------------------------------------------------------
// Register update over bytes, from register `s` (no pre- or post-inversion here).
DFL_HD uint32_t crc_table_entry(uint32_t i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
    return c;
}
// a * b mod P in the reflected representation (bit 31 = x^0)
DFL_HD uint32_t crc_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; ++i) {
        if (a & (0x80000000u >> i)) p ^= b;
        b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}
// x^(8 * n_bytes) mod P
DFL_HD uint32_t crc_xpow8(uint64_t n_bytes) {
    uint32_t result = 0x80000000u;           // x^0
    uint32_t sq = 0x00800000u;               // x^8
    for (uint64_t n = n_bytes; n; n >>= 1) {
        if (n & 1) result = crc_mulmod(result, sq);
        sq = crc_mulmod(sq, sq);
    }
    return result;
}
// register after the bytes of B, given the register `s` before them and r = the register that B alone produces from 0:
// the update is affine in the register, its linear part is the multiplication by x^(8 |B|)
DFL_HD uint32_t crc_chain(uint32_t s, uint32_t r_of_b, uint32_t xpow_len_b) { return crc_mulmod(s, xpow_len_b) ^ r_of_b; }

// ---- BGZF framing (SAM specification, section 4.1) --------------------------------------------------------------------
constexpr int BGZF_HEAD = 18, BGZF_TAIL = 8, BGZF_MAX_BLOCK = 65536;
constexpr int BGZF_PAYLOAD = 0xff00;      // htslib's BGZF_BLOCK_SIZE: payload bytes per block
DFL_HD void bgzf_header(uint8_t *p, uint32_t total_block_bytes) {
    const uint8_t h[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0};
    for (int i = 0; i < 16; ++i) p[i] = h[i];
    p[16] = (uint8_t)((total_block_bytes - 1) & 0xFF);
    p[17] = (uint8_t)((total_block_bytes - 1) >> 8);
}
DFL_HD void bgzf_trailer(uint8_t *p, uint32_t crc, uint32_t isize) {
    for (int i = 0; i < 4; ++i) { p[i] = (uint8_t)(crc >> (8 * i)); p[4 + i] = (uint8_t)(isize >> (8 * i)); }
}

}  // namespace dfl
}  // namespace kbbq
