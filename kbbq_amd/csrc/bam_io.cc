// bam_io.cc -- see bam_io.h.
#include "bam_io.h"

#include <algorithm>
#include <cstring>

namespace kbbq {

// ------------------------------------------------------------------ record ----
void BamRecord::sequence(std::string &out) const {
    static const char letters[] = "=ACMGRSVTWYHKDBN";                      // seq_nt16_str
    static const int8_t two_bit[16] = {4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4};   // seq_nt16_int
    const uint32_t n = l_seq();
    const uint8_t *s = data.data() + seq_at();
    out.resize(n);
    const bool rev = reverse();
    for (uint32_t i = 0; i < n; ++i) {
        const int code = (s[i >> 1] >> ((~i & 1) << 2)) & 15;              // bam_seqi
        if (!rev) {
            out[i] = letters[code];
        } else {
            const int b = two_bit[code];
            out[n - 1 - i] = b < 4 ? "TGCA"[b] : 'N';                       // readutils.hh:35-36, then the reversal
        }
    }
}

// length of the value that starts with the type byte at s (skip_aux of htslib), or 0 if it does not fit
static size_t aux_value_size(const uint8_t *s, const uint8_t *end) {
    if (s >= end) return 0;
    size_t fixed = 0;
    switch (*s) {
        case 'A': case 'c': case 'C': fixed = 1; break;
        case 's': case 'S': fixed = 2; break;
        case 'i': case 'I': case 'f': fixed = 4; break;
        case 'd': fixed = 8; break;
        case 'Z': case 'H': {
            const void *nul = memchr(s + 1, 0, (size_t)(end - (s + 1)));
            return nul ? (size_t)((const uint8_t *)nul - s) + 1 : 0;
        }
        case 'B': {
            if (end - s < 6) return 0;
            size_t each;
            switch (s[1]) {
                case 'c': case 'C': each = 1; break;
                case 's': case 'S': each = 2; break;
                case 'i': case 'I': case 'f': each = 4; break;
                default: return 0;
            }
            const uint64_t count = (uint64_t)s[2] | (uint64_t)s[3] << 8 | (uint64_t)s[4] << 16 | (uint64_t)s[5] << 24;
            const uint64_t total = 6 + each * count;
            return total <= (uint64_t)(end - s) ? (size_t)total : 0;
        }
        default: return 0;
    }
    return 1 + fixed <= (size_t)(end - s) ? 1 + fixed : 0;
}

size_t BamRecord::aux_find(const char tag[2], int &status) const {
    const uint8_t *base = data.data(), *end = base + data.size();
    const uint8_t *s = base + aux_at();
    while (end - s >= 3) {
        const bool hit = s[0] == (uint8_t)tag[0] && s[1] == (uint8_t)tag[1];
        s += 2;
        const size_t sz = aux_value_size(s, end);
        if (!sz) { status = BAM_AUX_CORRUPT; return 0; }
        if (hit) { status = BAM_AUX_OK; return (size_t)(s - base); }
        s += sz;
    }
    status = BAM_AUX_MISSING;
    return 0;
}

bool BamRecord::aux_string(const char tag[2], std::string &out, int &status) const {
    const size_t at = aux_find(tag, status);
    if (!at) return false;
    if (data[at] != 'Z' && data[at] != 'H') { status = BAM_AUX_CORRUPT; return false; }   // bam_aux2Z returns NULL, EINVAL
    out.assign((const char *)data.data() + at + 1);
    return true;
}

bool BamRecord::aux_update_string(const char tag[2], const std::string &text, int &status) {
    const size_t at = aux_find(tag, status);
    if (at) {
        if (data[at] != 'Z') { status = BAM_AUX_CORRUPT; return false; }
        const size_t old_len = strlen((const char *)data.data() + at + 1) + 1;
        std::vector<uint8_t> value(text.begin(), text.end());
        value.push_back(0);
        data.erase(data.begin() + at + 1, data.begin() + at + 1 + old_len);
        data.insert(data.begin() + at + 1, value.begin(), value.end());
        return true;
    }
    if (status != BAM_AUX_MISSING) return false;
    data.push_back((uint8_t)tag[0]);
    data.push_back((uint8_t)tag[1]);
    data.push_back('Z');
    data.insert(data.end(), text.begin(), text.end());
    data.push_back(0);
    status = BAM_AUX_OK;
    return true;
}

// ------------------------------------------------------------------ reader ----
BamReader::BamReader(const std::string &path, int threads) {
    fh_ = open_bytes(path, threads);      // BGZF blocks inflated by a pool; or gzread, which concatenates the members
    if (!fh_) return;
    unsigned char magic[4];
    uint32_t l_text = 0, n_ref = 0;
    if (!read_exact(magic, 4) || memcmp(magic, "BAM\1", 4) != 0) return;
    if (!read_exact(&l_text, 4)) return;
    header_.text.resize(l_text);
    if (l_text && !read_exact(&header_.text[0], l_text)) return;
    if (!read_exact(&n_ref, 4)) return;
    for (uint32_t i = 0; i < n_ref; ++i) {
        uint32_t l_name = 0, l_ref = 0;
        if (!read_exact(&l_name, 4) || l_name == 0 || l_name > (1u << 20)) return;
        std::string name(l_name, '\0');
        if (!read_exact(&name[0], l_name) || !read_exact(&l_ref, 4)) return;
        name.resize(l_name - 1);
        header_.refs.emplace_back(name, l_ref);
    }
    ok_ = true;
}

BamReader::~BamReader() {}

bool BamReader::read_exact(void *dst, size_t n) {
    unsigned char *p = (unsigned char *)dst;
    while (n) {
        const int got = fh_->read(p, (unsigned)std::min<size_t>(n, 1u << 30));
        if (got <= 0) return false;
        p += got;
        n -= (size_t)got;
    }
    return true;
}

int BamReader::next(BamRecord &rec) {
    if (!ok_) return -2;
    unsigned char len[4];
    const int got = fh_->read(len, 4);
    if (got == 0) return -1;
    if (got != 4) return -2;
    const uint32_t block = (uint32_t)len[0] | (uint32_t)len[1] << 8 | (uint32_t)len[2] << 16 | (uint32_t)len[3] << 24;
    if (block < 32 || block > (1u << 29)) return -2;
    rec.data.resize(block);
    if (!read_exact(rec.data.data(), block)) return -2;
    if (!rec.well_formed()) return -2;
    return (int)block;
}

// ------------------------------------------------------------ read of a record ----
bool decode_bam_read(const BamRecord &b, bool use_oq, std::string &seq, std::vector<uint8_t> &qual, std::string &rg, bool &second,
                     std::string &err) {
    b.sequence(seq);
    int status = 0;
    if (use_oq) {
        std::string oq;
        if (!b.aux_string("OQ", oq, status)) {
            err = "Error: --use-oq was specified but unable to read OQ tag on read " + b.name() + "\n";
            err += status == BAM_AUX_MISSING ? "OQ not found. Try again without the --use-oq option.\n" : "Tag data is corrupt. Repair the tags and try again.\n";
            return false;
        }
        if (oq.size() != b.l_seq()) {   // the reference indexes past the shorter of the two
            err = "Error: OQ tag of read " + b.name() + " has " + std::to_string(oq.size()) + " values for " + std::to_string(b.l_seq()) + " bases.\n";
            return false;
        }
        qual.resize(oq.size());
        for (size_t i = 0; i < oq.size(); ++i) qual[i] = (uint8_t)(oq[i] - 33);
    } else {
        qual.assign(b.qual(), b.qual() + b.l_seq());
    }
    if (b.reverse()) std::reverse(qual.begin(), qual.end());   // readutils.cc:36-39
    if (!b.aux_string("RG", rg, status)) {
        err = "Error: Unable to read RG tag on read " + b.name() + "\n";
        err += status == BAM_AUX_MISSING ? "RG not found. Every read in the BAM must have an RG tag; add tags with samtools addreplacerg and try again.\n"
                                         : "Tag data is corrupt. Repair the tags and try again.\n";
        return false;
    }
    second = b.second();
    return true;
}

// ------------------------------------------------------------ chunk parser ----
BamChunkParser::BamChunkParser(const std::string &path, bool use_oq, int io_threads, int parse_threads, bool keep_records)
    : ChunkPipeline(parse_threads), use_oq_(use_oq), keep_records_(keep_records) {
    BamReader head(path, io_threads);
    if (!head.ok()) return;
    header_ = head.header();
    src_ = head.release();
    if (src_) start();
}

void BamChunkParser::produce() {
    constexpr size_t kChunk = 32u << 20, kPiece = 2u << 20;
    std::vector<unsigned char> carry;
    bool eof = false, bad = false, failed = false;
    while (!eof && !bad && !stopping()) {
        auto chunk = chunk_buffer(carry.size() + kChunk);
        if (!carry.empty()) memcpy(chunk->data(), carry.data(), carry.size());
        size_t have = carry.size();
        carry.clear();
        while (have < chunk->size()) {
            const int got = src_->read(chunk->data() + have, (unsigned)std::min<size_t>(chunk->size() - have, 1u << 30));
            if (got < 0) failed = true;      // (a read error -- bad checksum, file cut inside a block -- is never a clean end)
            if (got <= 0) { eof = true; break; }
            have += (size_t)got;
        }
        if (have == 0 && !failed) break;      // (the buffer keeps its size: `have` says how much of it is the stream's)
        const unsigned char *b = chunk->data();
        // whole records: [block_size u32][block]; a partial one waits for the next chunk, a malformed size or -- at the
        // end of the stream -- a partial record ends the stream after the piece before it
        size_t begin = 0, p = 0;
        auto flush = [&](size_t end, bool last) {
            auto job = std::make_shared<Job>();
            job->chunk = chunk;
            job->begin = begin;
            job->end = end;
            job->last = last;      // here: a bad record follows
            submit(job);
            begin = end;
        };
        for (;;) {
            if (have - p < 4) break;
            const uint32_t block = (uint32_t)b[p] | (uint32_t)b[p + 1] << 8 | (uint32_t)b[p + 2] << 16 | (uint32_t)b[p + 3] << 24;
            if (block < 32 || block > (1u << 29)) { bad = true; break; }
            if (have - p - 4 < block) break;
            p += 4 + (size_t)block;
            if (p - begin >= kPiece) flush(p, false);
        }
        const bool truncated = eof && (p < have || failed);      // bytes that are no whole record at the end of the stream
        if (bad || truncated) {
            flush(p, true);      // (possibly an empty piece: it carries the end-of-stream mark)
        } else {
            if (p > begin) flush(p, false);
            carry.assign(b + p, b + have);
        }
    }
}

void BamChunkParser::parse(Job &job) {
    auto piece = std::make_shared<ReadPiece>();
    job.piece = piece;
    ReadPiece &P = *piece;
    P.off.assign(1, 0);
    P.blob_off.assign(1, 0);
    P.end_of_stream = job.last;
    const unsigned char *b = job.chunk->data();
    const size_t bytes = job.end - job.begin;
    P.seq.reserve(bytes / 2);
    P.qual.reserve(bytes / 2);
    if (keep_records_) P.blob.reserve(bytes);
    BamRecord rec;
    std::string seq, rg, last_rg, err;
    std::vector<uint8_t> qual;
    uint32_t last_rg_index = 0;
    bool have_last = false;
    for (size_t p = job.begin; p < job.end;) {
        const uint32_t block = (uint32_t)b[p] | (uint32_t)b[p + 1] << 8 | (uint32_t)b[p + 2] << 16 | (uint32_t)b[p + 3] << 24;
        rec.data.assign(b + p + 4, b + p + 4 + block);
        p += 4 + (size_t)block;
        if (!rec.well_formed()) {      // sam_read1 < -1: the passes' loops end here
            P.end_of_stream = true;
            return;
        }
        bool second = false;
        if (!decode_bam_read(rec, use_oq_, seq, qual, rg, second, err)) {
            P.fatal_at = (long)P.n();
            P.fatal_msg = err;
            return;
        }
        if (!have_last || rg != last_rg) {
            size_t i = 0;
            while (i < P.rg_names.size() && P.rg_names[i] != rg) ++i;
            if (i == P.rg_names.size()) P.rg_names.push_back(rg);
            last_rg = rg;
            last_rg_index = (uint32_t)i;
            have_last = true;
        }
        P.rg.push_back(last_rg_index);
        P.second.push_back(second ? 1 : 0);
        P.seq.insert(P.seq.end(), seq.begin(), seq.end());
        P.qual.insert(P.qual.end(), qual.begin(), qual.end());
        P.qual.resize(P.seq.size(), 0);
        P.off.push_back(P.seq.size());
        P.longest = std::max(P.longest, seq.size());
        if (keep_records_) {
            P.blob.append((const char *)rec.data.data(), rec.data.size());
            P.lens.push_back((uint32_t)rec.data.size());
            P.blob_off.push_back(P.blob.size());
        }
    }
}

// ------------------------------------------------------------------ writer ----
static void put32(std::string &s, uint32_t v) {
    for (int i = 0; i < 4; ++i) s.push_back((char)(v >> (8 * i)));
}

bool BamWriter::write_header(const BamHeader &h) {
    std::string s("BAM\1", 4);
    put32(s, (uint32_t)h.text.size());
    s += h.text;
    put32(s, (uint32_t)h.refs.size());
    for (auto &r : h.refs) {
        put32(s, (uint32_t)r.first.size() + 1);
        s += r.first;
        s.push_back('\0');
        put32(s, r.second);
    }
    // htslib flushes the BGZF block after the header, so records start in a block of their own; the
    // decompressed stream is the same either way
    return out_.write(s.data(), s.size());
}

bool BamWriter::write(const BamRecord &rec) {
    unsigned char len[4];
    const uint32_t n = (uint32_t)rec.data.size();
    for (int i = 0; i < 4; ++i) len[i] = (unsigned char)(n >> (8 * i));
    return out_.write((const char *)len, 4) && out_.write((const char *)rec.data.data(), rec.data.size());
}

}  // namespace kbbq
