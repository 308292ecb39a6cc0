// fastq_io.cc -- see fastq_io.h.
#include "fastq_io.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace kbbq {

// ------------------------------------------------------------ byte sources ----
namespace {

class GzSource : public ByteSource {
public:
    explicit GzSource(gzFile f) : f_(f) { gzbuffer(f_, 1 << 20); }
    ~GzSource() override { gzclose(f_); }
    int read(void *dst, unsigned n) override { return gzread(f_, dst, n); }

private:
    gzFile f_;
};

// size of the BGZF block that starts with this gzip header, or 0 if it is not one
size_t bgzf_block_size(const unsigned char *h, size_t have) {
    if (have < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return 0;
    const size_t xlen = h[10] | (size_t)h[11] << 8;
    if (have < 12 + xlen) return 0;
    for (size_t at = 12; at + 4 <= 12 + xlen;) {
        const size_t slen = h[at + 2] | (size_t)h[at + 3] << 8;
        if (h[at] == 'B' && h[at + 1] == 'C' && slen == 2 && at + 6 <= 12 + xlen) return (size_t)(h[at + 4] | (size_t)h[at + 5] << 8) + 1;
        at += 4 + slen;
    }
    return 0;
}

class BgzfSource : public ByteSource {
public:
    BgzfSource(FILE *f, int threads) : f_(f) {
        for (int i = 0; i < threads; ++i) pool_.emplace_back(&BgzfSource::worker, this);
        depth_ = (size_t)threads * 8;
    }
    ~BgzfSource() override {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_todo_.notify_all();
        for (auto &t : pool_) t.join();
        fclose(f_);
    }
    int read(void *dst, unsigned n) override {
        unsigned char *out = (unsigned char *)dst;
        unsigned done = 0;
        while (done < n) {
            if (cur_ && pos_ < cur_->out.size()) {
                const size_t take = std::min<size_t>(n - done, cur_->out.size() - pos_);
                memcpy(out + done, cur_->out.data() + pos_, take);
                pos_ += take;
                done += (unsigned)take;
                continue;
            }
            if (!next_block()) return failed_ ? -1 : (int)done;
        }
        return (int)done;
    }

private:
    struct Job {
        std::vector<unsigned char> in, out;
        size_t data_at = 0;
        bool done = false, ok = true;
    };
    void worker() {
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_todo_.wait(lk, [&] { return stop_ || !todo_.empty(); });
                if (todo_.empty()) return;
                job = todo_.front();
                todo_.pop_front();
            }
            bool ok = true;
            const size_t total = job->in.size();
            const unsigned char *tail = job->in.data() + total - 8;
            const uint32_t crc = tail[0] | (uint32_t)tail[1] << 8 | (uint32_t)tail[2] << 16 | (uint32_t)tail[3] << 24;
            const uint32_t isize = tail[4] | (uint32_t)tail[5] << 8 | (uint32_t)tail[6] << 16 | (uint32_t)tail[7] << 24;
            job->out.resize(isize);
            if (isize) {
                z_stream zs;
                memset(&zs, 0, sizeof zs);
                ok = inflateInit2(&zs, -15) == Z_OK;
                if (ok) {
                    zs.next_in = job->in.data() + job->data_at;
                    zs.avail_in = (uInt)(total - 8 - job->data_at);
                    zs.next_out = job->out.data();
                    zs.avail_out = isize;
                    ok = inflate(&zs, Z_FINISH) == Z_STREAM_END && zs.total_out == isize;
                    inflateEnd(&zs);
                }
                ok = ok && (uint32_t)crc32(crc32(0L, Z_NULL, 0), job->out.data(), isize) == crc;
            }
            {
                std::lock_guard<std::mutex> lk(mu_);
                job->ok = ok;
                job->done = true;
            }
            cv_done_.notify_all();
        }
    }
    // read compressed blocks ahead until `depth_` are outstanding
    void feed() {
        while (!eof_ && order_.size() < depth_) {
            unsigned char head[18];
            const size_t got = fread(head, 1, sizeof head, f_);
            if (got == 0) { eof_ = true; break; }
            size_t total = bgzf_block_size(head, got);
            // the size sits in the first subfield in every BGZF writer; anything else ends the stream as an error
            if (!total || total < 26) { eof_ = true; failed_ = true; break; }
            auto job = std::make_shared<Job>();
            job->in.resize(total);
            memcpy(job->in.data(), head, sizeof head);
            if (fread(job->in.data() + sizeof head, 1, total - sizeof head, f_) != total - sizeof head) { eof_ = true; failed_ = true; break; }
            job->data_at = 12 + (job->in[10] | (size_t)job->in[11] << 8);
            if (job->data_at + 8 > total) { eof_ = true; failed_ = true; break; }
            order_.push_back(job);
            {
                std::lock_guard<std::mutex> lk(mu_);
                todo_.push_back(job);
            }
            cv_todo_.notify_one();
        }
    }
    bool next_block() {
        feed();
        if (order_.empty()) return false;
        std::shared_ptr<Job> job = order_.front();
        order_.pop_front();
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_done_.wait(lk, [&] { return job->done; });
        }
        if (!job->ok) { failed_ = true; return false; }
        cur_ = job;
        pos_ = 0;
        return true;
    }
    FILE *f_;
    std::vector<std::thread> pool_;
    std::deque<std::shared_ptr<Job>> order_, todo_;
    std::shared_ptr<Job> cur_;
    size_t pos_ = 0, depth_ = 8;
    std::mutex mu_;
    std::condition_variable cv_todo_, cv_done_;
    bool stop_ = false, eof_ = false, failed_ = false;
};

// A plain gzip stream cannot be inflated in parallel, but it can be inflated AHEAD: one thread runs zlib and
// fills a short queue of 4 MiB pieces while the caller parses the previous ones.
class AheadSource : public ByteSource {
public:
    explicit AheadSource(std::unique_ptr<ByteSource> inner) : inner_(std::move(inner)), th_(&AheadSource::run, this) {}
    ~AheadSource() override {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_space_.notify_all();
        th_.join();
    }
    int read(void *dst, unsigned n) override {
        unsigned char *out = (unsigned char *)dst;
        unsigned done = 0;
        while (done < n) {
            if (pos_ < cur_.size()) {
                const size_t take = std::min<size_t>(n - done, cur_.size() - pos_);
                memcpy(out + done, cur_.data() + pos_, take);
                pos_ += take;
                done += (unsigned)take;
                continue;
            }
            std::unique_lock<std::mutex> lk(mu_);
            cv_data_.wait(lk, [&] { return !ready_.empty() || finished_; });
            if (ready_.empty()) return error_ && !done ? -1 : (int)done;
            cur_.swap(ready_.front());
            ready_.pop_front();
            pos_ = 0;
            lk.unlock();
            cv_space_.notify_one();
        }
        return (int)done;
    }

private:
    void run() {
        for (;;) {
            std::vector<unsigned char> piece(4 << 20);
            const int got = inner_->read(piece.data(), (unsigned)piece.size());
            std::unique_lock<std::mutex> lk(mu_);
            if (got <= 0) {
                error_ = got < 0;
                finished_ = true;
                lk.unlock();
                cv_data_.notify_all();
                return;
            }
            piece.resize((size_t)got);
            cv_space_.wait(lk, [&] { return stop_ || ready_.size() < 4; });
            if (stop_) return;
            ready_.push_back(std::move(piece));
            lk.unlock();
            cv_data_.notify_one();
        }
    }
    std::unique_ptr<ByteSource> inner_;
    std::deque<std::vector<unsigned char>> ready_;
    std::vector<unsigned char> cur_;
    size_t pos_ = 0;
    std::mutex mu_;
    std::condition_variable cv_data_, cv_space_;
    bool stop_ = false, finished_ = false, error_ = false;
    std::thread th_;      // last member: starts when everything above exists
};

}  // namespace

static BgzfSourceFactory g_bgzf_factory = nullptr;
void set_bgzf_source_factory(BgzfSourceFactory f) { g_bgzf_factory = f; }

std::unique_ptr<ByteSource> open_bytes(const std::string &path, int threads) {
    if (threads > 1 && path != "-") {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) return nullptr;
        unsigned char head[18];
        const size_t got = fread(head, 1, sizeof head, f);
        if (bgzf_block_size(head, got)) {
            if (g_bgzf_factory) {
                std::unique_ptr<ByteSource> dev = g_bgzf_factory(path);
                if (dev) { fclose(f); return dev; }
            }
            rewind(f);
            return std::unique_ptr<ByteSource>(new BgzfSource(f, threads));
        }
        fclose(f);
    }
    gzFile g = path == "-" ? gzdopen(0, "rb") : gzopen(path.c_str(), "rb");
    if (!g) return nullptr;
    std::unique_ptr<ByteSource> src(new GzSource(g));
    if (threads > 1) return std::unique_ptr<ByteSource>(new AheadSource(std::move(src)));
    return src;
}

// ------------------------------------------------------------------ reader ----
FastqReader::FastqReader(const std::string &path, int threads) : fh_(open_bytes(path, threads)), buf_(1 << 18) {}
FastqReader::~FastqReader() {}

int FastqReader::getc_() {
    if (pos_ >= end_) {
        if (eof_) return -1;
        const int n = fh_->read(buf_.data(), (unsigned)buf_.size());
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
            const int n = fh_->read(buf_.data(), (unsigned)buf_.size());
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

// ------------------------------------------------------------ chunk parser ----
namespace {
constexpr size_t kChunkBytes = 32u << 20, kPieceBytes = 2u << 20;

// is `p` (a position right after a newline, or 0) the start of a record as far as two lines can tell?
// 1 yes, 0 no, -1 the lines are not all inside [p, end)
int looks_like_record_start(const unsigned char *b, size_t p, size_t end) {
    if (p >= end) return -1;
    if (b[p] != '@') return 0;
    const void *n0 = memchr(b + p, '\n', end - p);
    if (!n0) return -1;
    const size_t s = (size_t)((const unsigned char *)n0 - b) + 1;
    if (s >= end) return -1;
    const void *n1 = memchr(b + s, '\n', end - s);
    if (!n1) return -1;
    const size_t plus = (size_t)((const unsigned char *)n1 - b) + 1;
    if (plus >= end) return -1;
    return b[plus] == '+' ? 1 : 0;
}
}  // namespace

// ---- the pipeline: reader thread -> pool -> consumer, in order ----
ChunkPipeline::~ChunkPipeline() { stop(); }

std::shared_ptr<std::vector<unsigned char>> ChunkPipeline::chunk_buffer(size_t bytes) {
    std::vector<unsigned char> v;
    {
        std::lock_guard<std::mutex> lk(buffers_->mu);
        if (!buffers_->free.empty()) { v = std::move(buffers_->free.back()); buffers_->free.pop_back(); }
    }
    if (v.size() < bytes) v.resize(bytes);
    std::shared_ptr<BufferPool> pool = buffers_;      // (outlives the pipeline if a piece is still held somewhere)
    auto *raw = new std::vector<unsigned char>(std::move(v));
    return std::shared_ptr<std::vector<unsigned char>>(raw, [pool](std::vector<unsigned char> *p) {
        {
            std::lock_guard<std::mutex> lk(pool->mu);
            if (pool->free.size() < 8) pool->free.push_back(std::move(*p));
        }
        delete p;
    });
}

void ChunkPipeline::start() {
    max_in_flight_ = (size_t)parse_threads_ * 6 + 16;
    for (int i = 0; i < parse_threads_; ++i) pool_.emplace_back(&ChunkPipeline::worker, this);
    reader_ = std::thread(&ChunkPipeline::run_reader, this);
    started_ = true;
}

void ChunkPipeline::stop() {
    if (!started_) return;
    started_ = false;
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
    }
    cv_todo_.notify_all();
    cv_room_.notify_all();
    if (reader_.joinable()) reader_.join();
    for (auto &t : pool_) t.join();
    pool_.clear();
}

bool ChunkPipeline::stopping() {
    std::lock_guard<std::mutex> lk(mu_);
    return stop_;
}

void ChunkPipeline::submit(std::shared_ptr<Job> job) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_room_.wait(lk, [&] { return stop_ || in_flight_ < max_in_flight_; });
    if (stop_) return;
    ++in_flight_;
    order_.push_back(job);
    todo_.push_back(job);
    lk.unlock();
    cv_todo_.notify_one();
    cv_done_.notify_all();      // (the consumer may be waiting for a first piece)
}

void ChunkPipeline::run_reader() {
    produce();
    {
        std::lock_guard<std::mutex> lk(mu_);
        finished_ = true;
    }
    cv_done_.notify_all();
}

void ChunkPipeline::worker() {
    for (;;) {
        std::shared_ptr<Job> job;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_todo_.wait(lk, [&] { return stop_ || !todo_.empty(); });
            if (stop_) return;
            job = todo_.front();
            todo_.pop_front();
        }
        parse(*job);
        {
            std::lock_guard<std::mutex> lk(mu_);
            job->done = true;
        }
        cv_done_.notify_all();
    }
}

std::shared_ptr<ReadPiece> ChunkPipeline::next() {
    if (!started_ || ended_) return nullptr;
    std::shared_ptr<Job> job;
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return !order_.empty() || finished_; });
        if (order_.empty()) { ended_ = true; return nullptr; }
        job = order_.front();
        cv_done_.wait(lk, [&] { return job->done; });
        order_.pop_front();
        --in_flight_;
    }
    cv_room_.notify_one();
    if (job->piece->complex || job->piece->fatal_at >= 0 || job->piece->end_of_stream) ended_ = true;
    return job->piece;
}

FastqChunkParser::FastqChunkParser(const std::string &path, int io_threads, int parse_threads, bool keep_records)
    : ChunkPipeline(parse_threads), src_(open_bytes(path, io_threads)), keep_records_(keep_records) {
    if (src_) start();
}

void FastqChunkParser::produce() {
    std::vector<unsigned char> carry;
    bool eof = false;
    while (!eof && !stopping()) {
        auto chunk = chunk_buffer(carry.size() + kChunkBytes);
        if (!carry.empty()) memcpy(chunk->data(), carry.data(), carry.size());
        size_t have = carry.size();
        carry.clear();
        bool failed = false;
        while (have < chunk->size()) {
            const int got = src_->read(chunk->data() + have, (unsigned)std::min<size_t>(chunk->size() - have, 1u << 30));
            if (got < 0) failed = true;
            if (got <= 0) { eof = true; break; }
            have += (size_t)got;
        }
        if (have == 0) break;      // (the buffer keeps its size: `have` says how much of it is the stream's)
        const unsigned char *b = chunk->data();
        size_t cut = have;
        bool whole_complex = failed;      // (a read error: FastqReader decides what the passes see of such a file)
        if (!eof) {
            // the last record start whose first three lines are inside the chunk; what follows it waits for the next chunk
            cut = 0;
            for (size_t p = have; p > 1;) {
                const void *nl = memrchr(b, '\n', p - 1);
                if (!nl) break;
                const size_t cand = (size_t)((const unsigned char *)nl - b) + 1;
                if (looks_like_record_start(b, cand, have) == 1) { cut = cand; break; }
                p = cand;      // continue before this newline
                if (have - cand > (8u << 20)) break;      // (no record start in the last 8 MB: not the shape this parser is for)
            }
            if (cut == 0) whole_complex = true;
            else carry.assign(b + cut, b + have);
        }
        if (whole_complex) {
            auto job = std::make_shared<Job>();
            job->force_complex = true;
            submit(job);
            break;
        }
        // pieces of about kPieceBytes, cut at guessed record starts
        size_t begin = 0;
        while (begin < cut) {
            size_t end = cut;
            if (cut - begin > kPieceBytes + kPieceBytes / 2) {
                size_t p = begin + kPieceBytes;
                for (;;) {
                    const void *nl = memchr(b + p, '\n', cut - p);
                    if (!nl) break;
                    const size_t cand = (size_t)((const unsigned char *)nl - b) + 1;
                    if (cand >= cut) break;
                    const int v = looks_like_record_start(b, cand, cut);
                    if (v == 1) { end = cand; break; }
                    if (v < 0) break;
                    p = cand;
                }
            }
            auto job = std::make_shared<Job>();
            job->chunk = chunk;
            job->begin = begin;
            job->end = end;
            job->last = eof && end == cut;
            submit(job);
            begin = end;
        }
    }
}

void FastqChunkParser::parse(Job &job) {
    const bool keep_records = keep_records_;
    auto piece = std::make_shared<FastqPiece>();
    job.piece = piece;
    FastqPiece &P = *piece;
    P.off.assign(1, 0);
    P.blob_off.assign(1, 0);
    if (job.force_complex) { P.complex = true; return; }
    const unsigned char *b = job.chunk->data();
    const size_t end = job.end;
    const size_t bytes = end - job.begin;
    P.seq.reserve(bytes / 2);
    P.qual.reserve(bytes / 2);
    if (keep_records) P.blob.reserve(bytes * 2 / 3);
    std::string name, rg, first, last_rg;
    uint32_t last_rg_index = 0;
    bool have_last = false;
    size_t p = job.begin;
    while (p < end) {
        if (b[p] != '@') { P.complex = true; return; }
        const unsigned char *n0 = (const unsigned char *)memchr(b + p, '\n', end - p);
        if (!n0) { P.complex = true; return; }
        const size_t h0 = p + 1, h1 = (size_t)(n0 - b);                     // header without '@'
        const size_t s0 = h1 + 1;
        if (s0 >= end) { P.complex = true; return; }
        const unsigned char *n1 = (const unsigned char *)memchr(b + s0, '\n', end - s0);
        if (!n1) { P.complex = true; return; }
        const size_t s1 = (size_t)(n1 - b);                                  // sequence [s0, s1)
        const size_t plus = s1 + 1;
        if (plus >= end || b[plus] != '+') { P.complex = true; return; }
        const unsigned char *n2 = (const unsigned char *)memchr(b + plus, '\n', end - plus);
        if (!n2) { P.complex = true; return; }
        const size_t q0 = (size_t)(n2 - b) + 1;
        const unsigned char *n3 = q0 < end ? (const unsigned char *)memchr(b + q0, '\n', end - q0) : nullptr;
        if (!n3 && !job.last) { P.complex = true; return; }
        const size_t q1 = n3 ? (size_t)(n3 - b) : end;                       // quality [q0, q1)
        const size_t sl = s1 - s0;
        if (sl == 0 || q1 - q0 != sl || q0 > end) { P.complex = true; return; }
        const unsigned char c0 = b[s0];
        if (c0 == '+' || c0 == '@' || c0 == '>') { P.complex = true; return; }
        if ((h1 > h0 && b[h1 - 1] == '\r') || b[s1 - 1] == '\r' || b[q1 - 1] == '\r' || b[q0 - 2] == '\r') { P.complex = true; return; }
        // name = up to the first blank or tab, comment = the rest of the line (FastqReader::next)
        size_t blank = h0;
        while (blank < h1 && b[blank] != ' ' && b[blank] != '\t') ++blank;
        name.assign((const char *)b + h0, blank - h0);
        bool second = false;
        if (!parse_read_name(name, rg, second, first)) {
            P.fatal_at = (long)P.n();
            P.fatal_name = name;
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
        P.seq.insert(P.seq.end(), b + s0, b + s1);
        const size_t qa = P.qual.size();
        P.qual.resize(qa + sl);
        for (size_t i = 0; i < sl; ++i) P.qual[qa + i] = (uint8_t)(b[q0 + i] - 33);
        P.off.push_back(P.seq.size());
        P.longest = std::max(P.longest, sl);
        if (keep_records) {
            const size_t cl = blank < h1 ? h1 - blank - 1 : 0;
            P.blob.append((const char *)b + h0, blank - h0);
            if (cl) P.blob.append((const char *)b + blank + 1, cl);
            P.blob.append((const char *)b + s0, sl);
            P.lens.push_back((uint32_t)(blank - h0));
            P.lens.push_back((uint32_t)cl);
            P.lens.push_back((uint32_t)sl);
            P.blob_off.push_back(P.blob.size());
        }
        p = n3 ? q1 + 1 : end;
    }
    // (p == end: the piece ended exactly where the next one starts, which proves that cut)
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
// compression level: zlib's default, which is what bgzf_open(fn, "w") of the reference gives (htsiter.cc:73);
// KBBQ_BGZF_LEVEL=0..9 trades size for speed (the decompressed stream is the same)
static int bgzf_level() {
    static const int level = [] {
        const char *s = getenv("KBBQ_BGZF_LEVEL");
        if (!s || !*s) return Z_DEFAULT_COMPRESSION;
        return s[0] >= '0' && s[0] <= '9' && !s[1] ? s[0] - '0' : Z_DEFAULT_COMPRESSION;
    }();
    return level;
}

static bool deflate_block(const unsigned char *data, size_t len, std::vector<unsigned char> &out) {
    if (!len) return true;
    unsigned char block[0x10000];
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, bgzf_level(), Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
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
