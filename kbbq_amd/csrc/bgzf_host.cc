// bgzf_host.cc -- host-only twin of the device BGZF encoder (bgzf_device.hip): the same scalar pieces
// (deflate_common.h: Huffman lengths, canonical codes, code-length header, token bits, CRC chaining, framing) around a
// plain serial match finder.  It exists so that those shared pieces are exercised without a GPU (tests/test_bgzf_cpu.py
// inflates its output with zlib); the command line never calls it -- its output path is the device encoder.
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/kbbq_bgzf.h"
#include "deflate_common.h"

using namespace kbbq::dfl;

namespace {

uint32_t crc_bytes(uint32_t s, const uint8_t *p, size_t n) {      // register update, no inversion
    for (size_t i = 0; i < n; ++i) {
        s ^= p[i];
        for (int k = 0; k < 8; ++k) s = (s & 1) ? (s >> 1) ^ 0xEDB88320u : s >> 1;
    }
    return s;
}

// one BGZF block from in[0..len), len <= BGZF_PAYLOAD; returns its size
size_t encode_block(const uint8_t *in, size_t len, uint8_t *out) {
    // greedy matches over a table of the most recent position of every 4-byte hash
    std::vector<uint32_t> tokens;
    tokens.reserve(len + 1);
    std::vector<int32_t> table(1 << 13, -1);
    std::vector<uint32_t> ll_freq(N_LL, 0), d_freq(N_D, 0);
    auto hash4 = [&](size_t p) { uint32_t v; memcpy(&v, in + p, 4); return (v * 2654435761u) >> (32 - 13); };
    for (size_t p = 0; p < len;) {
        int best = 0, dist = 0;
        if (p + 4 <= len) {
            const uint32_t h = hash4(p);
            const int32_t c = table[h];
            table[h] = (int32_t)p;
            if (c >= 0 && p - (size_t)c <= (size_t)MAX_DIST) {
                const size_t lim = std::min<size_t>(MAX_MATCH, len - p);
                size_t n = 0;
                while (n < lim && in[c + n] == in[p + n]) ++n;
                if (n >= 4) { best = (int)n; dist = (int)(p - (size_t)c); }
            }
        }
        if (best) {
            tokens.push_back(token_match(best, dist));
            int ls, e, v, ds;
            length_symbol(best, ls, e, v);
            distance_symbol(dist, ds, e, v);
            ++ll_freq[ls];
            ++d_freq[ds];
            for (size_t q = p + 1; q < p + (size_t)best && q + 4 <= len; ++q) table[hash4(q)] = (int32_t)q;
            p += (size_t)best;
        } else {
            tokens.push_back(token_literal(in[p]));
            ++ll_freq[in[p]];
            ++p;
        }
    }
    ++ll_freq[256];
    BlockCodes B;
    uint8_t head[HEAD_BYTES];
    memset(head, 0, sizeof head);
    uint16_t order[N_LL], runs[N_LL + N_D];
    uint32_t w[N_LL];
    uint8_t all[N_LL + N_D];
    build_block_codes(ll_freq.data(), d_freq.data(), B, head, order, w, runs, all);
    uint64_t bits = B.head_bits;
    for (uint32_t t : tokens) { uint64_t v; int nb; token_bits(t, B, v, nb); bits += (uint64_t)nb; }
    bits += B.ll_len[256];
    const size_t dyn_bytes = (size_t)((bits + 7) / 8), stored_bytes = len + 5;
    uint8_t *body = out + BGZF_HEAD;
    size_t body_bytes;
    if (dyn_bytes < stored_bytes && dyn_bytes + BGZF_HEAD + BGZF_TAIL <= (size_t)BGZF_MAX_BLOCK) {
        memset(body, 0, dyn_bytes);
        memcpy(body, head, (B.head_bits + 7) / 8);
        BitSink s = {body, B.head_bits, body[B.head_bits >> 3], (int)(B.head_bits & 7)};      // goes on inside the header's last byte
        for (uint32_t t : tokens) { uint64_t v; int nb; token_bits(t, B, v, nb); s.put(v, nb); }
        s.put(B.ll_code[256], B.ll_len[256]);
        s.finish();
        body_bytes = dyn_bytes;
    } else {      // stored: BFINAL = 1, BTYPE = 00, LEN, ~LEN, the bytes
        body[0] = 1;
        body[1] = (uint8_t)(len & 0xFF); body[2] = (uint8_t)(len >> 8);
        body[3] = (uint8_t)(~len & 0xFF); body[4] = (uint8_t)((~len >> 8) & 0xFF);
        memcpy(body + 5, in, len);
        body_bytes = stored_bytes;
    }
    const size_t total = BGZF_HEAD + body_bytes + BGZF_TAIL;
    bgzf_header(out, (uint32_t)total);
    // the CRC in two halves through the chaining rule the device uses across its 64 lanes
    const size_t half = len / 2;
    const uint32_t r1 = crc_bytes(0, in, half), r2 = crc_bytes(0, in + half, len - half);
    uint32_t s = 0xFFFFFFFFu;
    s = crc_chain(s, r1, crc_xpow8(half));
    s = crc_chain(s, r2, crc_xpow8(len - half));
    bgzf_trailer(out + BGZF_HEAD + body_bytes, s ^ 0xFFFFFFFFu, (uint32_t)len);
    return total;
}

}  // namespace

extern "C" {

int kbbq_host_bgzf_compress(const uint8_t *payload, uint64_t n, uint8_t *out, uint64_t out_capacity, uint64_t *out_bytes) {
    if ((!payload && n) || !out || !out_bytes) return -22;
    uint64_t at = 0;
    uint8_t block[BGZF_MAX_BLOCK];
    for (uint64_t p = 0; p < n; p += BGZF_PAYLOAD) {
        const size_t len = (size_t)std::min<uint64_t>(BGZF_PAYLOAD, n - p);
        const size_t sz = encode_block(payload + p, len, block);
        if (at + sz > out_capacity) return -34;
        memcpy(out + at, block, sz);
        at += sz;
    }
    *out_bytes = at;
    return 0;
}

uint64_t kbbq_bgzf_bound(uint64_t n) {
    const uint64_t blocks = (n + BGZF_PAYLOAD - 1) / BGZF_PAYLOAD;
    return n + blocks * (BGZF_HEAD + BGZF_TAIL + 5) + 28;
}

}  // extern "C"
