// fastq_io.cc -- see fastq_io.h.
#include "fastq_io.h"

#include <cstring>

namespace kbbq {

// ------------------------------------------------------------------ reader ----
FastqReader::FastqReader(const std::string &path) : buf_(1 << 18) {
    fh_ = path == "-" ? gzdopen(0, "rb") : gzopen(path.c_str(), "rb");
    if (fh_) gzbuffer(fh_, 1 << 20);
}
FastqReader::~FastqReader() {
    if (fh_) gzclose(fh_);
}

int FastqReader::getc_() {
    if (pos_ >= end_) {
        if (eof_) return -1;
        const int n = gzread(fh_, buf_.data(), (unsigned)buf_.size());
        if (n <= 0) { eof_ = true; return -1; }
        pos_ = 0;
        end_ = (size_t)n;
    }
    return buf_[pos_++];
}

bool FastqReader::getline_(std::string &out, bool append) {
    if (!append) out.clear();
    bool any = false;
    for (;;) {
        if (pos_ >= end_) {
            if (eof_) return any;
            const int n = gzread(fh_, buf_.data(), (unsigned)buf_.size());
            if (n <= 0) { eof_ = true; return any; }
            pos_ = 0;
            end_ = (size_t)n;
        }
        const unsigned char *p = buf_.data() + pos_;
        const void *nl = memchr(p, '\n', end_ - pos_);
        const size_t take = nl ? (size_t)((const unsigned char *)nl - p) : end_ - pos_;
        out.append((const char *)p, take);
        any = true;
        pos_ += take;
        if (nl) {
            ++pos_;
            if (!out.empty() && out.back() == '\r') out.pop_back();
            return true;
        }
    }
}

// kseq_read (htslib kseq.h): skip to the next '@' (or '>'), name up to the first blank, comment = rest
// of that line, sequence lines until a line starting with '+', '>' or '@'; after '+': quality lines
// until the quality is at least as long as the sequence.
int FastqReader::next(FastqRecord &rec) {
    if (!fh_) return -1;
    int c;
    if (last_char_ == 0) {
        while ((c = getc_()) != -1 && c != '>' && c != '@') {}
        if (c == -1) return -1;
        last_char_ = c;
    }
    rec.name.clear(); rec.comment.clear(); rec.seq.clear(); rec.qual.clear();
    std::string header;
    if (!getline_(header, false)) return -1;
    const size_t blank = header.find_first_of(" \t");
    if (blank == std::string::npos) {
        rec.name = header;
    } else {
        rec.name = header.substr(0, blank);
        rec.comment = header.substr(blank + 1);
    }
    // sequence lines
    std::string line;
    for (;;) {
        c = getc_();
        if (c == -1) break;
        if (c == '>' || c == '+' || c == '@') break;
        if (c == '\n') continue;
        rec.seq.push_back((char)c);
        getline_(rec.seq, true);
    }
    if (c == '>' || c == '@') last_char_ = c; else last_char_ = 0;
    if (c != '+') return (int)rec.seq.size();      // FASTA record
    getline_(line, false);                          // rest of the '+' line
    while (rec.qual.size() < rec.seq.size()) {
        if (!getline_(rec.qual, true)) break;
    }
    last_char_ = 0;
    if (rec.qual.size() != rec.seq.size()) return -2;
    return (int)rec.seq.size();
}

// -------------------------------------------------------------- read names ----
bool parse_read_name(const std::string &name, std::string &rg, bool &second, std::string &first_name) {
    std::string fullname(name);
    const std::string delim("_");
    size_t current_pos = fullname.find(delim);
    first_name = fullname.substr(0, current_pos);
    rg.clear();
    while (rg.empty() && current_pos != std::string::npos) {
        fullname = fullname.substr(current_pos + 1);
        current_pos = fullname.find(delim);
        if (fullname.substr(0, 3) == "RG:") {
            const size_t last_colon = fullname.find_last_of(":", current_pos);
            // readutils.cc:84 passes the POSITION current_pos as the COUNT argument of substr
            rg = fullname.substr(last_colon + 1, current_pos);
        }
    }
    if (first_name.length() < 2) return false;     // readutils.cc:90: substr(length() - 2) throws
    const std::string tail = first_name.substr(first_name.length() - 2);
    second = tail == "/2";
    if (second || tail == "/1") first_name = first_name.substr(0, first_name.length() - 2);
    return true;
}

// ------------------------------------------------------------------ writer ----
namespace {
const unsigned char kEofBlock[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43,
                                     0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
}

// one BGZF block from data[0..len) appended to `out`; halves the input if the compressed form does not fit 64 KiB
static bool deflate_block(const unsigned char *data, size_t len, std::vector<unsigned char> &out) {
    if (!len) return true;
    unsigned char block[0x10000];
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    zs.next_in = const_cast<unsigned char *>(data);
    zs.avail_in = (uInt)len;
    zs.next_out = block + 18;
    zs.avail_out = sizeof block - 18 - 8;
    const int rc = deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) {
        if (len < 2) return false;
        return deflate_block(data, len / 2, out) && deflate_block(data + len / 2, len - len / 2, out);
    }
    const size_t total = clen + 18 + 8;
    // gzip header with the BGZF 'BC' extra field carrying the block size - 1
    const unsigned char head[18] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0,
                                    (unsigned char)((total - 1) & 0xff), (unsigned char)((total - 1) >> 8)};
    memcpy(block, head, 18);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)len);
    const uint32_t isize = (uint32_t)len;
    unsigned char *tail = block + 18 + clen;
    for (int i = 0; i < 4; ++i) { tail[i] = (unsigned char)(crc >> (8 * i)); tail[4 + i] = (unsigned char)(isize >> (8 * i)); }
    out.insert(out.end(), block, block + total);
    return true;
}

BgzfWriter::BgzfWriter(FILE *out, int threads) : out_(out) {
    pending_.reserve(kBlock);
    for (int i = 0; i < threads && threads > 1; ++i) pool_.emplace_back(&BgzfWriter::worker, this);
}

BgzfWriter::~BgzfWriter() {
    close();
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
    }
    cv_todo_.notify_all();
    for (auto &t : pool_) t.join();
}

void BgzfWriter::worker() {
    for (;;) {
        std::shared_ptr<Job> job;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_todo_.wait(lk, [&] { return stop_ || !todo_.empty(); });
            if (todo_.empty()) return;
            job = todo_.front();
            todo_.pop_front();
        }
        const bool ok = deflate_block(job->in.data(), job->in.size(), job->out);
        {
            std::lock_guard<std::mutex> lk(mu_);
            job->ok = ok;
            job->done = true;
        }
        cv_done_.notify_all();
    }
}

bool BgzfWriter::drain(size_t keep) {
    while (order_.size() > keep) {
        std::shared_ptr<Job> job = order_.front();
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_done_.wait(lk, [&] { return job->done; });
        }
        order_.pop_front();
        if (!job->ok || fwrite(job->out.data(), 1, job->out.size(), out_) != job->out.size()) failed_ = true;
    }
    return !failed_;
}

bool BgzfWriter::submit() {
    if (pending_.empty()) return !failed_;
    if (pool_.empty()) {
        std::vector<unsigned char> out;
        const bool ok = deflate_block(pending_.data(), pending_.size(), out) && fwrite(out.data(), 1, out.size(), out_) == out.size();
        pending_.clear();
        if (!ok) failed_ = true;
        return ok;
    }
    auto job = std::make_shared<Job>();
    job->in.swap(pending_);
    pending_.reserve(kBlock);
    order_.push_back(job);
    {
        std::lock_guard<std::mutex> lk(mu_);
        todo_.push_back(job);
    }
    cv_todo_.notify_one();
    return drain(pool_.size() * 8);    // bounded memory: at most 8 blocks per worker in flight
}

bool BgzfWriter::write(const char *data, size_t n) {
    while (n) {
        const size_t room = kBlock - pending_.size();
        const size_t take = n < room ? n : room;
        pending_.insert(pending_.end(), (const unsigned char *)data, (const unsigned char *)data + take);
        data += take;
        n -= take;
        if (pending_.size() == kBlock && !submit()) return false;
    }
    return !failed_;
}

bool BgzfWriter::close() {
    if (closed_) return !failed_;
    closed_ = true;
    if (!submit() || !drain(0)) return false;
    if (fwrite(kEofBlock, 1, sizeof kEofBlock, out_) != sizeof kEofBlock) return false;
    return fflush(out_) == 0;
}

}  // namespace kbbq
