// kbbq_cli.cc -- the `kbbq` command line over the MI355X engine.
//
// Mirrors main() of the reference (kbbq.cc:81-459): same flags, same defaults, same stderr lines with
// the same "[%F %T %Z]" stamps, recalibrated FASTQ through BGZF on stdout.  Every pass re-opens the
// input like the reference does (kbbq.cc:232,258,336,365,456) and hands batches of reads to the engine
// through the C ABI (include/kbbq_engine.h); nothing is computed on the host except what the reference
// also computes there (coverage, alpha, thresholds, the delta-Q model -- inside the library).
//
// Differences, all deliberate:
//   * BAM goes through this tool's own codec (bam_io.*) because htslib is not in this image; CRAM and SAM
//     text are refused (SURVEY risk R1);
//   * one extra read-only scan of the input sizes the histograms (read groups, longest read) before
//     the engine is created; the reference grows its tables on the fly;
//   * the sampler seed can be fixed with KBBQ_SEED=<u32> (the reference always draws it from time+pid,
//     kbbq.cc:268-270, so two of its own runs differ: SURVEY hazard H1);
//   * --threads sizes the BGZF output pool (it sizes htslib's pool in the reference, kbbq.cc:159-168);
//     0 = up to 16 threads instead of none.  The compressed stream does not depend on the thread count;
//   * the packed reads stay resident in GPU memory between the passes when they fit (KBBQ_RESIDENT=0
//     turns that off): same results, four decodes of the input fewer;
//   * where the reference prints an error and then crashes or throws (missing --genomelen on FASTQ,
//     kbbq.cc:218; missing RG / OQ tags, readutils.cc:20-30,42-53) this prints the same text and exits 1.
#include <fcntl.h>
#include <getopt.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iomanip>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kbbq_bgzf.h"
#include "../../include/kbbq_exchange.h"
#include "../../include/kbbq_engine.h"
#include "bam_io.h"
#include "fastq_io.h"
#include "host_model.h"

using namespace kbbq;

static std::ostream &put_now(std::ostream &os) {   // kbbq.cc:49-53
    std::time_t t = std::time(nullptr);
    std::tm tm = *std::localtime(&t);
    return os << std::put_time(&tm, "[%F %T %Z]");
}

static struct option long_options[] = {   // kbbq.cc:66-79
    {"ksize", required_argument, 0, 'k'},    {"use-oq", no_argument, 0, 'u'},     {"set-oq", no_argument, 0, 's'},
    {"genomelen", required_argument, 0, 'g'}, {"coverage", required_argument, 0, 'c'}, {"fixed", required_argument, 0, 'f'},
    {"alpha", required_argument, 0, 'a'},     {"threads", required_argument, 0, 't'},  {0, 0, 0, 0}};

// minion::create_seed_seq().GenerateOne() (minion.hpp:320-345, 377-408; the chrono/random_device branch
// is disabled there by the __cpluscplus typo, so the inputs are time(nullptr), getpid() and two constants)
static uint32_t time_pid_seed() {
    auto splitmix = [](uint64_t &st) {
        st += 0x9e3779b97f4a7c15ULL;
        uint64_t z = st;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    };
    const uint64_t u = (uint64_t)time(nullptr);
    const uint32_t seq[5] = {(uint32_t)u, (uint32_t)(u >> 32), (uint32_t)getpid(), 0xC8F978DBu, 0x0B32F62Eu};
    uint64_t s = 0xFD57D105u;
    uint64_t sum = splitmix(s);
    for (uint32_t v : seq) sum += splitmix(s) * v;
    sum += splitmix(s) * 1;
    return (uint32_t)(sum >> 32);
}

enum class Format { fastq, bam, cram, unknown };
static Format sniff(const std::string &path) {   // hts_detect_format, as far as this tool needs it
    gzFile f = path == "-" ? nullptr : gzopen(path.c_str(), "rb");
    if (!f) return Format::unknown;
    unsigned char b[4] = {0, 0, 0, 0};
    const int n = gzread(f, b, 4);
    gzclose(f);
    if (n >= 4 && b[0] == 'B' && b[1] == 'A' && b[2] == 'M' && b[3] == 1) return Format::bam;
    if (n >= 4 && b[0] == 'C' && b[1] == 'R' && b[2] == 'A' && b[3] == 'M') return Format::cram;
    if (n >= 1 && b[0] == '@') return Format::fastq;
    return Format::unknown;
}

// One read as the passes see it (what HTSFile::get() puts into CReadData), plus the record it came from.
struct Item {
    std::string seq;              // sequencing orientation
    std::vector<uint8_t> qual;    // numeric, sequencing orientation
    std::string rg;
    bool second = false;
    FastqRecord fq;
    BamRecord bam;
};

enum { SRC_FATAL = -100 };        // an error the reference throws on; the message is already on stderr

// The record source of the passes: htsiter::HTSFile (htsiter.hh:40-49) for this tool.
class Source {
public:
    virtual ~Source() {}
    virtual bool ok() const = 0;
    virtual int next(Item &it) = 0;   // >= 0 ok, -1 end of file, < -1 error (the reference's loops just end), SRC_FATAL
};

class FastqSource : public Source {   // FastqFile + CReadData(kseq_t*), htsiter.cc:49-59, readutils.cc:64-104
public:
    FastqSource(const std::string &path, int threads) : in_(path, threads) {}
    bool ok() const override { return in_.ok(); }
    int next(Item &it) override {
        const int rc = in_.next(it.fq);
        if (rc < 0) return rc;
        std::string first;
        if (!parse_read_name(it.fq.name, it.rg, it.second, first)) {
            std::cerr << put_now << " Error: read name '" << it.fq.name << "' is shorter than 2 characters before the first '_'." << std::endl;
            return SRC_FATAL;   // std::out_of_range in the reference (readutils.cc:90)
        }
        it.seq = it.fq.seq;
        it.qual.resize(it.fq.qual.size());
        for (size_t i = 0; i < it.qual.size(); ++i) it.qual[i] = (uint8_t)(it.fq.qual[i] - 33);   // readutils.cc:70-71
        return rc;
    }

private:
    FastqReader in_;
};

class BamSource : public Source {     // BamFile + CReadData(bam1_t*, use_oq), htsiter.cc:5-9, readutils.cc:13-61
public:
    BamSource(const std::string &path, bool use_oq, int threads) : in_(path, threads), use_oq_(use_oq) {}
    bool ok() const override { return in_.ok(); }
    const BamHeader &header() const { return in_.header(); }
    int next(Item &it) override {
        const int rc = in_.next(it.bam);
        if (rc < 0) return rc;
        std::string err;
        if (!decode_bam_read(it.bam, use_oq_, it.seq, it.qual, it.rg, it.second, err)) {
            std::cerr << err << std::flush;
            return SRC_FATAL;
        }
        return rc;
    }

private:
    BamReader in_;
    bool use_oq_;
};

// KBBQ_TIMING=1: wall-clock per phase on stderr at the end ("[timing] scan 12.3 s ..."): where an end-to-end run goes
struct PhaseClock {
    std::vector<std::pair<std::string, double>> phases;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    bool on = getenv("KBBQ_TIMING") != nullptr;
    void mark(const char *name) {
        const auto t1 = std::chrono::steady_clock::now();
        phases.emplace_back(name, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    }
    ~PhaseClock() { report(); }
    void report() {
        if (!on) return;
        on = false;
        std::cerr << "[timing]";
        for (auto &p : phases) std::cerr << " " << p.first << " " << p.second << " s";
        struct rusage ru;
        if (getrusage(RUSAGE_SELF, &ru) == 0) std::cerr << " peak_host_rss_MB " << ru.ru_maxrss / 1024;
        std::cerr << std::endl;
    }
};

static int g_io_threads = 1;   // inflate pool of BGZF inputs (hts_set_thread_pool on the input handle, htsiter.hh:64-66,110-112)
static std::unique_ptr<Source> open_source(const std::string &path, bool is_bam, bool use_oq) {   // open_file, kbbq.cc:55-64
    if (is_bam) return std::unique_ptr<Source>(new BamSource(path, use_oq, g_io_threads));
    return std::unique_ptr<Source>(new FastqSource(path, g_io_threads));
}

// BGZF output through the encoder on the GPU (include/kbbq_bgzf.h): what the reference leaves to htslib's bgzf_write /
// sam_write1 (htsiter.cc:45,75-86).  Bytes written here are gathered in page-locked memory, a chunk of 2048 whole
// blocks at a time, copied to the device and deflated there; up to two chunks are in flight, so the kernels of one
// overlap the write(2) of the one before.  fastq_batch() hands over a batch whose record text the device assembles
// itself around the recalibrated qualities it already holds.  The decompressed stream is the host writer's, byte for byte.
class DeviceBgzfWriter : public ByteSink {
public:
    DeviceBgzfWriter(FILE *out, int device) : out_(out) {
        if (kbbq_bgzf_create(device, &z_) < 0) { z_ = nullptr; failed_ = true; return; }
        // Output that lands in a regular file (kbbq ... > out.fq.gz) is written by several threads at once, each its own
        // range with pwrite: one thread fills the page cache at ~5 GB/s, which at BASELINE size is longer than the GPU
        // needs for the blocks.  A pipe, a terminal or an append-mode file gets the plain sequential writes.
        // KBBQ_WRITE_THREADS=1: sequential always.
        fflush(out_);
        const int fd = fileno(out_);
        struct stat st;
        const int fl = fd >= 0 ? fcntl(fd, F_GETFL) : -1;
        int want = getenv("KBBQ_WRITE_THREADS") ? atoi(getenv("KBBQ_WRITE_THREADS")) : 4;
#ifdef F_SETPIPE_SZ
        // a pipe: as much buffer as the system grants (64 KB by default, 1 MB usually allowed): fewer hand-overs to the reader
        if (fd >= 0 && fstat(fd, &st) == 0 && S_ISFIFO(st.st_mode)) (void)fcntl(fd, F_SETPIPE_SZ, 1 << 20);
#endif
        if (fd >= 0 && want > 1 && fl >= 0 && !(fl & O_APPEND) && fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
            const off_t at = lseek(fd, 0, SEEK_CUR);
            if (at >= 0) { fd_ = fd; file_at_ = (uint64_t)at; write_threads_ = std::min(want, 16); }
        }
        void *p = nullptr;
        if (kbbq_host_alloc(kChunk, &p) < 0) { failed_ = true; return; }
        buf_ = (char *)p;
        // Everything else (a pipe, a terminal): the blocks are written in order by a thread of their own, so that the main
        // thread is back at the device -- the next chunk's inflation, the next submission -- while write(2) waits for the
        // reader of the pipe.  The encoder keeps a collected submission's blocks in place for the next two collects.
        if (fd_ < 0) sink_ = std::thread([this] { sink_loop(); });
    }
    ~DeviceBgzfWriter() override {
        close();
        if (sink_.joinable()) {
            { std::lock_guard<std::mutex> lk(sink_mu_); sink_stop_ = true; }
            sink_cv_.notify_all();
            sink_.join();
        }
        if (buf_) kbbq_host_free(buf_);
        if (z_) kbbq_bgzf_destroy(z_);
    }
    bool ok() const { return !failed_; }
    bool write(const char *data, size_t n) override {
        while (n && !failed_) {
            const size_t take = std::min(n, kChunk - fill_);
            memcpy(buf_ + fill_, data, take);
            fill_ += take; data += take; n -= take;
            if (fill_ == kChunk && !flush_host()) return false;
        }
        return !failed_;
    }
    // One batch of FASTQ records (RecordStore layout) whose quality lines come from device memory.
    bool fastq_batch(const char *blob, const uint32_t *lens, uint64_t n_records, const uint8_t *d_qual, const uint64_t *d_qual_offsets,
                     uint32_t uniform_len, void *after_stream) {
        if (!flush_host()) return false;      // bytes written before this batch come first
        if (!make_room()) return false;
        if (kbbq_bgzf_submit_fastq(z_, blob, lens, n_records, d_qual, d_qual_offsets, uniform_len, after_stream) < 0) return fail_here();
        ++in_flight_;
        return true;
    }
    // Bytes the caller has formatted itself in page-locked memory (a whole batch of BAM records): submitted as they are
    // (the caller's buffer is free again on return); what write() gathered before them goes first.
    bool submit_buffer(const char *pinned, size_t n) {
        if (!flush_host()) return false;
        if (!n) return true;
        if (!make_room()) return false;
        if (kbbq_bgzf_submit(z_, pinned, n, 0, nullptr) < 0) return fail_here();
        ++in_flight_;
        return true;
    }
    // The current chunk of a device reader with new qualities (kbbq_fastq_reader_write): text assembled from the device's
    // own copy of the input.
    bool reader_chunk(kbbq_fastq_reader *reader, const uint8_t *d_qual, void *after_stream) {
        if (!flush_host()) return false;
        if (!make_room()) return false;
        if (kbbq_fastq_reader_write(reader, z_, d_qual, after_stream) < 0) return fail_here();
        ++in_flight_;
        return true;
    }
    // The current chunk of a device BAM reader with new qualities (kbbq_bam_reader_write): BamFile::recalibrate + write of
    // every record on the device.
    bool bam_chunk(kbbq_bam_reader *reader, const uint8_t *d_qual, bool set_oq, void *after_stream) {
        if (!flush_host()) return false;
        if (!make_room()) return false;
        if (kbbq_bam_reader_write(reader, z_, d_qual, set_oq ? 1 : 0, after_stream) < 0) return fail_here();
        ++in_flight_;
        return true;
    }
    // reads of the synthetic data set formatted on the device (kbbq_bgzf_submit_synth)
    bool synth_batch(kbbq_engine *e, const kbbq_synth_params *sp, uint64_t first, uint64_t n, int format) {
        if (!flush_host()) return false;
        if (!make_room()) return false;
        if (kbbq_bgzf_submit_synth(z_, e, sp, first, n, format, nullptr) < 0) return fail_here();
        ++in_flight_;
        return true;
    }
    // every submission so far has left the device (a caller may then reuse device memory the submissions read)
    bool drain() { return drain_to(0); }
    // all but the newest `keep` submissions
    bool drain_to(int keep) {
        while (in_flight_ > keep) if (!collect_one()) return false;
        return !failed_;
    }
    bool close() override {
        if (closed_) return !failed_;
        closed_ = true;
        if (failed_ || !flush_host() || !drain()) return false;
        if (!put(kbbq_bgzf_eof_block(), 28) || !sink_wait(0)) return false;
        if (fd_ >= 0 && lseek(fd_, (off_t)file_at_, SEEK_SET) < 0) return false;      // whoever writes next continues behind the blocks
        return fflush(out_) == 0;
    }
    int in_flight() const { return in_flight_; }
    uint64_t payload_bytes = 0, compressed_bytes = 0;
    void kernel_ms(double &format, double &deflate, double &gather) const { format = deflate = gather = 0; if (z_) kbbq_bgzf_kernel_ms(z_, &format, &deflate, &gather); }

private:
    static constexpr size_t kChunk = (size_t)2048 * KBBQ_BGZF_PAYLOAD;      // whole blocks: no short block inside the stream
    bool fail_here() {
        if (!failed_) std::cerr << "BGZF writer: " << kbbq_last_error() << std::endl;
        failed_ = true;
        return false;
    }
    bool collect_one() {
        const uint8_t *blocks = nullptr;
        uint64_t n = 0, raw = 0;
        if (!sink_wait(2)) return false;      // (the buffer this collect fills was handed out three collects ago)
        if (kbbq_bgzf_collect(z_, &blocks, &n, &raw) < 0) return fail_here();
        --in_flight_;
        payload_bytes += raw;
        compressed_bytes += n;
        return put(blocks, n);
    }
    // n finished bytes to the output
    bool put(const uint8_t *data, uint64_t n) {
        if (fd_ < 0) {
            if (!sink_.joinable()) {
                if (fwrite(data, 1, n, out_) != n) { failed_ = true; return false; }
                return true;
            }
            {
                std::lock_guard<std::mutex> lk(sink_mu_);
                if (sink_failed_) { failed_ = true; return false; }
                sink_q_.emplace_back(data, n);
            }
            sink_cv_.notify_all();
            return true;
        }
        const uint64_t at = file_at_;
        auto range = [&](uint64_t b, uint64_t e) -> bool {
            while (b < e) {
                const ssize_t w = pwrite(fd_, data + b, (size_t)std::min<uint64_t>(e - b, 8u << 20), (off_t)(at + b));
                if (w <= 0) return false;
                b += (uint64_t)w;
            }
            return true;
        };
        const int nt = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)write_threads_, n >> 16));      // at least 64 KB each
        bool ok_all = true;
        if (nt == 1) ok_all = range(0, n);
        else {
            std::vector<std::thread> th;
            std::vector<char> ok((size_t)nt, 1);
            const uint64_t per = ((n + nt - 1) / nt + 4095) & ~(uint64_t)4095;
            for (int t = 0; t < nt; ++t)
                th.emplace_back([&, t] { const uint64_t b = std::min(n, per * t), e = std::min(n, b + per); if (!range(b, e)) ok[(size_t)t] = 0; });
            for (auto &x : th) x.join();
            for (char c : ok) ok_all = ok_all && c;
        }
        if (!ok_all) { failed_ = true; return false; }
        file_at_ += n;
        return true;
    }
    bool make_room() {
        while (in_flight_ >= 2) if (!collect_one()) return false;
        return !failed_;
    }
    // the writing thread of a sequential sink
    void sink_loop() {
        std::unique_lock<std::mutex> lk(sink_mu_);
        for (;;) {
            sink_cv_.wait(lk, [&] { return sink_stop_ || !sink_q_.empty(); });
            if (sink_q_.empty()) return;
            const std::pair<const uint8_t *, uint64_t> job = sink_q_.front();
            lk.unlock();
            const bool ok = sink_failed_ || fwrite(job.first, 1, job.second, out_) == job.second;
            lk.lock();
            if (!ok) sink_failed_ = true;
            sink_q_.pop_front();      // (behind the write: a job counts as pending while its bytes are being read)
            sink_cv_.notify_all();
        }
    }
    // until at most `pending` handed-over pieces are unwritten
    bool sink_wait(size_t pending) {
        if (!sink_.joinable()) return !failed_;
        std::unique_lock<std::mutex> lk(sink_mu_);
        sink_cv_.wait(lk, [&] { return sink_q_.size() <= pending; });
        if (sink_failed_) failed_ = true;
        return !failed_;
    }
    bool flush_host() {
        if (failed_) return false;
        if (!fill_) return true;
        if (!make_room()) return false;
        if (kbbq_bgzf_submit(z_, buf_, fill_, 0, nullptr) < 0) return fail_here();      // (returns when the chunk has left buf_)
        ++in_flight_;
        fill_ = 0;
        return true;
    }
    FILE *out_;
    int fd_ = -1;                   // >= 0: a regular file, written with pwrite at file_at_
    uint64_t file_at_ = 0;
    int write_threads_ = 1;
    kbbq_bgzf *z_ = nullptr;
    char *buf_ = nullptr;
    size_t fill_ = 0;
    int in_flight_ = 0;
    bool failed_ = false, closed_ = false;
    std::thread sink_;
    std::mutex sink_mu_;
    std::condition_variable sink_cv_;
    std::deque<std::pair<const uint8_t *, uint64_t>> sink_q_;
    bool sink_stop_ = false, sink_failed_ = false;
};

// BGZF input for the host parsers (BAM; FASTQ that the device's record kernels do not take) inflated on the GPU
// (kbbq_fastq_reader_inflate): a thread reads the file in 32 MB pieces, the device inflates and checks their blocks and
// copies the bytes into one of two page-locked buffers, read() hands them on.  What bgzf_mt's thread pool does for the
// reference (htsiter.hh:64-66,110-112) -- the pool's cores go to the parsers instead.  Installed as fastq_io's BGZF source;
// KBBQ_DEVICE_INFLATE=0 (or KBBQ_HOST_DEFLATE=1) leaves the thread pool in place.
class DeviceBgzfSource : public ByteSource {
public:
    static std::unique_ptr<ByteSource> open(const std::string &path) {
        std::unique_ptr<DeviceBgzfSource> s(new DeviceBgzfSource);
        if (!s->init(path)) return nullptr;
        return std::unique_ptr<ByteSource>(s.release());
    }
    ~DeviceBgzfSource() override {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
        if (reader_) kbbq_fastq_reader_destroy(reader_);
        for (int i = 0; i < 2; ++i) if (out_[i]) kbbq_host_free(out_[i]);
        if (in_) kbbq_host_free(in_);
        if (fd_ >= 0) ::close(fd_);
    }
    int read(void *dst, unsigned n) override {
        unsigned char *to = (unsigned char *)dst;
        unsigned done = 0;
        while (done < n) {
            if (have_ && pos_ < fill_[cur_]) {
                const size_t take = std::min<size_t>(n - done, fill_[cur_] - pos_);
                memcpy(to + done, out_[cur_] + pos_, take);
                pos_ += take;
                done += (unsigned)take;
                continue;
            }
            std::unique_lock<std::mutex> lk(mu_);
            if (have_) {          // the buffer is used up: give it back
                ready_[cur_] = false;
                have_ = false;
                cur_ ^= 1;
                cv_.notify_all();
            }
            cv_.wait(lk, [&] { return ready_[cur_] || eof_ || error_; });
            if (error_) return -1;
            if (!ready_[cur_]) break;      // end of file
            have_ = true;
            pos_ = 0;
        }
        return (int)done;
    }

private:
    bool init(const std::string &path) {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) return false;
        void *p = nullptr;
        if (kbbq_host_alloc(kIn + kCarry, &p) < 0) return false;
        in_ = (uint8_t *)p;
        for (int i = 0; i < 2; ++i) {
            if (kbbq_host_alloc(kOut, &p) < 0) return false;
            out_[i] = (uint8_t *)p;
        }
        if (kbbq_fastq_reader_create(0, &reader_) < 0) return false;
        th_ = std::thread([this] { run(); });
        return true;
    }
    void run() {
        uint64_t left = 0;      // unconsumed bytes at the front of in_ (the tail of the piece before)
        int k = 0;
        bool file_end = false;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return quit_ || !ready_[k]; });
                if (quit_) return;
            }
            // top the input buffer up
            while (!file_end && left < kIn) {
                const ssize_t r = ::read(fd_, in_ + left, (size_t)(kIn + kCarry - left));
                if (r < 0) { fail(); return; }
                if (r == 0) { file_end = true; break; }
                left += (uint64_t)r;
            }
            if (left == 0) break;
            uint64_t consumed = 0, produced = 0;
            if (kbbq_fastq_reader_inflate(reader_, in_, left, out_[k], kOut, &consumed, &produced) < 0) {
                std::cerr << "BGZF input: " << kbbq_last_error() << std::endl;
                fail();
                return;
            }
            if (consumed == 0) {      // no whole block in what is left: a truncated file
                if (file_end) { std::cerr << "BGZF input: the file ends inside a block." << std::endl; fail(); return; }
                fail();
                return;
            }
            memmove(in_, in_ + consumed, (size_t)(left - consumed));
            left -= consumed;
            if (produced) {
                std::lock_guard<std::mutex> lk(mu_);
                fill_[k] = produced;
                ready_[k] = true;
                k ^= 1;
            }
            cv_.notify_all();
        }
        std::lock_guard<std::mutex> lk(mu_);
        eof_ = true;
        cv_.notify_all();
    }
    void fail() {
        std::lock_guard<std::mutex> lk(mu_);
        error_ = true;
        cv_.notify_all();
    }
    static constexpr uint64_t kIn = 32ull << 20, kCarry = 1ull << 17, kOut = 256ull << 20;
    int fd_ = -1;
    kbbq_fastq_reader *reader_ = nullptr;
    uint8_t *in_ = nullptr, *out_[2] = {nullptr, nullptr};
    uint64_t fill_[2] = {0, 0};
    bool ready_[2] = {false, false}, have_ = false, eof_ = false, error_ = false, quit_ = false;
    int cur_ = 0;
    size_t pos_ = 0;
    std::mutex mu_;
    std::condition_variable cv_;
    std::thread th_;
};

// The input side on the GPU (include/kbbq_bgzf.h: kbbq_fastq_reader): a BGZF-compressed four-line FASTQ file goes to the
// device as it is, chunk by chunk -- inflate, line index, record rules and packing are kernels -- and every chunk becomes one
// resident batch.  Pass 4 feeds the same chunks again and the records' text is re-assembled there around the new qualities.
// What the reference does with kseq_read over bgzf_read once per pass (htsiter.cc:49-60).  Any shape this path does not
// take is reported by the reader and the caller starts over with the host parsers.
class DeviceFastqInput {
public:
    ~DeviceFastqInput() { close(); }
    bool active = false;
    bool text_kept = false;                   // every chunk's text and index stayed in HBM: pass 4 reads nothing
    uint64_t kept_bytes = 0;
    std::vector<uint64_t> chunk_records;      // records of every chunk of the first scan (pass 4 must meet the same)
    kbbq_fastq_reader *reader = nullptr;
    // BAM mode (open_bam): the same file feeding, the chunks go to a kbbq_bam_reader -- inflate, record chain, field decode and,
    // in pass 4, the records rewritten around the new qualities are kernels (sam_read1, the BAM
    // constructor of CReadData and BamFile::recalibrate / write, htsiter.cc:5-45, readutils.cc:13-61)
    kbbq_bam_reader *bam = nullptr;
    bool oq_unwritable = false;               // some record's OQ tag bam_aux_update_str could not update (--set-oq: host path)
    double wait_s = 0, device_s = 0, batch_s = 0;

    bool open_bam(const std::string &path, bool use_oq, int32_t n_ref, uint64_t header_bytes, const std::vector<std::string> &rg_ids) {
        if (!open_file(path)) return false;
        std::vector<const char *> ids;
        for (auto &id : rg_ids) ids.push_back(id.c_str());
        if (kbbq_bam_reader_create(0, use_oq ? 1 : 0, n_ref, header_bytes, ids.data(), (uint32_t)ids.size(), &bam) < 0) return false;
        start_pass();
        return true;
    }
    bool open(const std::string &path) {
        if (!open_file(path)) return false;
        if (kbbq_fastq_reader_create(0, &reader) < 0) return false;
        start_pass();
        return true;
    }
    bool open_file(const std::string &path) {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) return false;
        struct stat st;
        if (fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode)) return false;      // a pipe cannot be read twice
        size_ = (uint64_t)st.st_size;
        unsigned char magic[4] = {0, 0, 0, 0};
        if (pread(fd_, magic, 4, 0) != 4 || magic[0] != 0x1f || magic[1] != 0x8b || magic[2] != 8 || !(magic[3] & 4)) return false;      // not BGZF
        for (int i = 0; i < 2; ++i) {
            void *p = nullptr;
            if (kbbq_host_alloc(kFront + kPiece, &p) < 0) return false;
            buf_[i] = (uint8_t *)p;
        }
        return true;
    }
    void close() {
        stop_io();
        if (reader) kbbq_fastq_reader_destroy(reader);
        reader = nullptr;
        if (bam) kbbq_bam_reader_destroy(bam);
        bam = nullptr;
        for (int i = 0; i < 2; ++i) { if (buf_[i]) kbbq_host_free(buf_[i]); buf_[i] = nullptr; }
        if (fd_ >= 0) ::close(fd_);
        fd_ = -1;
    }
    // The file is read front to back in pieces of 256 MB by a thread of its own, into two page-locked buffers in turn; the
    // bytes the device did not take from one piece (the last, incomplete BGZF block: less than 64 KB) go in front of the
    // next one.  The same sequence of chunks comes out of every pass.
    void start_pass() {
        stop_io();
        next_piece_ = 0;
        taken_piece_ = 0;
        left_ = 0;
        io_error_ = false;
        quit_ = false;
        filled_[0] = filled_[1] = false;
        io_ = std::thread([this] {
            for (uint64_t k = 0;; ++k) {
                const uint64_t at = k * kPiece;
                if (at >= size_) break;
                const int b = (int)(k & 1);
                {
                    std::unique_lock<std::mutex> lk(mu_);
                    cv_.wait(lk, [&] { return quit_ || !filled_[b]; });
                    if (quit_) return;
                }
                const uint64_t n = std::min<uint64_t>(kPiece, size_ - at);
                uint64_t got = 0;
                bool ok = true;
                while (got < n) {
                    const ssize_t r = pread(fd_, buf_[b] + kFront + got, n - got, (off_t)(at + got));
                    if (r <= 0) { ok = false; break; }
                    got += (uint64_t)r;
                }
                // the piece starts for the device at once: its copy overlaps the kernels of the piece before it
                if (ok && !(getenv("KBBQ_PRELOAD") && atoi(getenv("KBBQ_PRELOAD")) == 0)) {
                    if (bam) (void)kbbq_bam_reader_preload(bam, buf_[b] + kFront, n, kFront);
                    else if (reader) (void)kbbq_fastq_reader_preload(reader, buf_[b] + kFront, n, kFront);
                }
                {
                    std::lock_guard<std::mutex> lk(mu_);
                    if (!ok) io_error_ = true;
                    filled_[b] = true;
                    piece_bytes_[b] = n;
                }
                cv_.notify_all();
                if (!ok) return;
            }
        });
    }
    // 1 = the next chunk is in `info` (its records, if any, are the reader's current chunk), 0 = end of file, -1 = I/O or device
    // error, -2 = a shape for the host parsers
    int next_chunk(kbbq_fastq_chunk &info) {
        const uint64_t at = taken_piece_ * kPiece;
        if (at >= size_) return 0;
        const int b = (int)(taken_piece_ & 1);
        const auto t0 = std::chrono::steady_clock::now();
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return filled_[b] || io_error_; });
            if (io_error_) return -1;
        }
        wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const uint64_t n = piece_bytes_[b];
        const int last = at + n >= size_;
        uint8_t *data = buf_[b] + kFront - left_;
        if (left_) memcpy(data, carry_, left_);
        const auto t1 = std::chrono::steady_clock::now();
        if ((bam ? kbbq_bam_reader_chunk(bam, data, left_ + n, last, &info) : kbbq_fastq_reader_chunk(reader, data, left_ + n, last, &info)) < 0) return -1;
        device_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        if (bam && (info.flags & 8)) oq_unwritable = true;
        if (info.flags & 7) return -2;      // (a read name that is too short included: the host path reports it)
        const uint64_t rest = left_ + n - info.consumed;
        if (rest > kFront || (rest && last)) return -2;      // a block that does not end: not a file this path reads
        if (rest) memcpy(carry_, data + info.consumed, rest);
        left_ = rest;
        {
            std::lock_guard<std::mutex> lk(mu_);
            filled_[b] = false;
        }
        cv_.notify_all();
        ++taken_piece_;
        return 1;
    }

private:
    void stop_io() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        if (io_.joinable()) io_.join();
    }
    static constexpr uint64_t kFront = 1ull << 16;
    // bytes of the file per chunk; KBBQ_READER_PIECE_KB shrinks it so that tests cross many chunk boundaries with small files
    const uint64_t kPiece = getenv("KBBQ_READER_PIECE_KB") ? std::max<uint64_t>(64, strtoull(getenv("KBBQ_READER_PIECE_KB"), nullptr, 10)) << 10 : 256ull << 20;
    int fd_ = -1;
    uint64_t size_ = 0, next_piece_ = 0, taken_piece_ = 0, left_ = 0, piece_bytes_[2] = {0, 0};
    uint8_t *buf_[2] = {nullptr, nullptr};
    uint8_t carry_[1 << 16];
    std::thread io_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool filled_[2] = {false, false}, io_error_ = false, quit_ = false;
};

// What the output pass needs of one batch besides the new qualities, kept from the first scan when it fits in
// host memory, so that the input is decoded once instead of twice: FASTQ name / comment / sequence text, or
// the BAM alignment blocks.
struct RecordStore {
    std::string blob;
    std::vector<uint32_t> lens;     // FASTQ: name, comment, sequence length per record; BAM: block length
    size_t bytes() const { return blob.capacity() + lens.capacity() * 4; }
    void add(const FastqRecord &r) {
        blob += r.name; blob += r.comment; blob += r.seq;
        lens.push_back((uint32_t)r.name.size()); lens.push_back((uint32_t)r.comment.size()); lens.push_back((uint32_t)r.seq.size());
    }
    void add(const BamRecord &r) {
        blob.append((const char *)r.data.data(), r.data.size());
        lens.push_back((uint32_t)r.data.size());
    }
};

// One batch of reads in the engine's layout, plus the records themselves for the output pass.
struct Batch {
    std::vector<FastqRecord> fq_recs;
    std::vector<BamRecord> bam_recs;
    std::vector<uint8_t> seq, qual, flags;
    std::vector<uint16_t> rg;
    std::vector<uint64_t> off, bases, nmask, offcase;
    kbbq_reads c;
    bool stop_at_empty = false;   // next_str() != "" loops end at the first empty read (kbbq.cc:234, htsiter.cc:95)
    bool fatal = false;
    bool saw_empty = false;       // an empty read went into this batch
    bool ended = false;           // stop_at_empty met its empty read: this pass over the source is over
    size_t longest = 0;           // longest read of this batch
    Item it;

    // returns false when no read was collected
    bool fill(Source &in, ReadGroups &groups, size_t max_reads, bool keep_records, bool is_bam = false) {
        fq_recs.clear(); bam_recs.clear(); seq.clear(); qual.clear(); flags.clear(); rg.clear();
        off.assign(1, 0);
        saw_empty = false;
        longest = 0;
        while (!ended && rg.size() < max_reads) {
            const int rc = in.next(it);
            if (rc == SRC_FATAL) { fatal = true; return false; }
            if (rc < 0) break;                       // -1 end of file; < -1 error: the reference's loops also just end
            if (stop_at_empty && it.seq.empty()) { ended = true; break; }
            if (it.seq.empty()) saw_empty = true;
            longest = std::max(longest, it.seq.size());
            seq.insert(seq.end(), it.seq.begin(), it.seq.end());
            qual.insert(qual.end(), it.qual.begin(), it.qual.end());
            qual.resize(seq.size(), 0);
            off.push_back(seq.size());
            flags.push_back(it.second ? 1 : 0);
            rg.push_back((uint16_t)groups.index_of(it.rg));
            if (keep_records) {
                if (is_bam) bam_recs.push_back(it.bam); else fq_recs.push_back(it.fq);
            }
        }
        return finish();
    }

    // The same batch from the block-parallel parsers (fastq_io.h: FastqChunkParser, strictly four-line FASTQ only; bam_io.h:
    // BamChunkParser).
    // Returns false when no read was collected; `complex` says the stream is not of that shape and the scan has to
    // start over with the serial reader.  The records go straight into `store` (when given).
    struct Fast {
        std::unique_ptr<ChunkPipeline> parser;      // FastqChunkParser or BamChunkParser
        std::shared_ptr<ReadPiece> cur;
        size_t at = 0;
        std::vector<int> rg_map;
        bool complex = false;
        int lens_per_record = 3;                    // RecordStore's layout: FASTQ 3 lengths per record, BAM 1
        double wait_s = 0;                          // time spent waiting for the parsers' next piece
    };
    // The pieces' arrays are copied into the batch's by a few threads at once (the segments and their places are known
    // first): one thread moved 580 bytes per BAM record -- sequence, qualities, the alignment block -- at 5 GB/s, which was
    // most of the first scan's wall time behind 16 parsers.
    struct Segment {
        std::shared_ptr<ReadPiece> piece;
        size_t a, b;                 // records [a, b) of the piece
        std::vector<int> rg_map;
        size_t read0, base0, blob0;  // where they go in the batch
    };
    bool fill_fast(Fast &f, ReadGroups &groups, size_t max_reads, RecordStore *store) {
        fq_recs.clear(); bam_recs.clear(); seq.clear(); qual.clear(); flags.clear(); rg.clear();
        off.assign(1, 0);
        saw_empty = false;
        longest = 0;
        std::vector<Segment> segs;
        size_t n_reads = 0, n_bases = 0, n_blob = 0;
        while (n_reads < max_reads) {
            if (f.cur && f.at == f.cur->n() && f.cur->fatal_at >= 0) {      // (the piece's parse stopped at that record)
                if (!f.cur->fatal_msg.empty()) std::cerr << f.cur->fatal_msg << std::flush;
                else std::cerr << put_now << " Error: read name '" << f.cur->fatal_name << "' is shorter than 2 characters before the first '_'." << std::endl;
                fatal = true;
                return false;
            }
            if (!f.cur || f.at == f.cur->n()) {
                const auto tw = std::chrono::steady_clock::now();
                f.cur = f.parser->next();
                f.wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
                f.at = 0;
                if (!f.cur) break;
                if (f.cur->complex) { f.complex = true; return false; }
                f.rg_map.clear();
                for (auto &name : f.cur->rg_names) f.rg_map.push_back(groups.index_of(name));
                continue;
            }
            const FastqPiece &P = *f.cur;
            const size_t take = std::min(P.n() - f.at, max_reads - n_reads);
            Segment g;
            g.piece = f.cur; g.a = f.at; g.b = f.at + take; g.rg_map = f.rg_map;
            g.read0 = n_reads; g.base0 = n_bases; g.blob0 = n_blob;
            n_reads += take;
            n_bases += (size_t)(P.off[g.b] - P.off[g.a]);
            if (store) n_blob += (size_t)(P.blob_off[g.b] - P.blob_off[g.a]);
            segs.push_back(std::move(g));
            f.at += take;
        }
        if (!n_reads) return finish();
        seq.resize(n_bases); qual.resize(n_bases); flags.resize(n_reads); rg.resize(n_reads); off.resize(n_reads + 1);
        const size_t blob_base = store ? store->blob.size() : 0, lens_base = store ? store->lens.size() : 0;
        if (store) { store->blob.resize(blob_base + n_blob); store->lens.resize(lens_base + (size_t)f.lens_per_record * n_reads); }
        std::vector<size_t> seg_longest(segs.size(), 0);
        std::vector<char> seg_empty(segs.size(), 0);
        auto copy_segment = [&](size_t i) {
            const Segment &g = segs[i];
            const ReadPiece &P = *g.piece;
            const uint64_t p0 = P.off[g.a];
            memcpy(seq.data() + g.base0, P.seq.data() + p0, (size_t)(P.off[g.b] - p0));
            memcpy(qual.data() + g.base0, P.qual.data() + p0, (size_t)(P.off[g.b] - p0));
            memcpy(flags.data() + g.read0, P.second.data() + g.a, g.b - g.a);
            size_t lg = 0;
            bool empty = false;
            for (size_t r = g.a; r < g.b; ++r) {
                const uint64_t l = P.off[r + 1] - P.off[r];
                off[g.read0 + (r - g.a) + 1] = g.base0 + (P.off[r + 1] - p0);
                lg = std::max<size_t>(lg, (size_t)l);
                empty = empty || l == 0;
                rg[g.read0 + (r - g.a)] = (uint16_t)g.rg_map[P.rg[r]];
            }
            seg_longest[i] = lg;
            seg_empty[i] = empty;
            if (store) {
                memcpy(&store->blob[blob_base + g.blob0], P.blob.data() + P.blob_off[g.a], (size_t)(P.blob_off[g.b] - P.blob_off[g.a]));
                memcpy(store->lens.data() + lens_base + (size_t)f.lens_per_record * g.read0, P.lens.data() + (size_t)f.lens_per_record * g.a,
                       (size_t)f.lens_per_record * (g.b - g.a) * sizeof(uint32_t));
            }
        };
        const size_t n_threads = std::min<size_t>(segs.size(), (size_t)std::max(1, std::min(g_io_threads, 8)));
        if (n_threads <= 1) {
            for (size_t i = 0; i < segs.size(); ++i) copy_segment(i);
        } else {
            std::atomic<size_t> next_seg{0};
            std::vector<std::thread> th;
            for (size_t t = 0; t < n_threads; ++t)
                th.emplace_back([&] { for (size_t i; (i = next_seg.fetch_add(1)) < segs.size();) copy_segment(i); });
            for (auto &x : th) x.join();
        }
        for (size_t i = 0; i < segs.size(); ++i) { longest = std::max(longest, seg_longest[i]); saw_empty = saw_empty || seg_empty[i]; }
        return finish();
    }

    // false: the batch is only going to be made resident with kbbq_reads_upload_text, which packs the bases on the device
    // (the first scan: packing 1.6e8 bases took this thread as long as everything else it does for a batch)
    bool pack_on_host = true;
    bool finish() {
        if (rg.empty()) return false;
        qual.resize(seq.size() + 16, 0);
        uint64_t n_offcase = 0;
        if (pack_on_host) {
            bases.assign(seq.size() / 32 + 2, 0);
            nmask.assign(seq.size() / 64 + 2, 0);
            // FASTQ text may be soft-masked: the raw case of a base matters to three comparisons of the reference
            // (include/kbbq_engine.h: kbbq_reads.offcase); the bit array only travels when some base is off-case.
            // BAM sequences come out of bam_seq_str upper-case (readutils.hh:30-42).
            offcase.assign(seq.size() / 64 + 2, 0);
            kbbq_pack_bases_case(seq.data(), seq.size(), bases.data(), nmask.data(), offcase.data(), &n_offcase);
        }
        memset(&c, 0, sizeof c);
        c.offcase = n_offcase ? offcase.data() : nullptr;
        c.n_reads = rg.size();
        c.n_bases = seq.size();
        c.bases = pack_on_host ? bases.data() : nullptr;
        c.nmask = pack_on_host ? nmask.data() : nullptr;
        c.qual = qual.data();
        c.offsets = off.data();
        c.flags = flags.data();
        c.rg = rg.data();
        // equally long reads (the usual Illumina file): the batch goes over as a uniform one -- no offsets array, the
        // engine's kernels for that shape (results do not depend on which form a batch takes)
        bool uniform = longest > 0;
        for (size_t r = 0; uniform && r + 1 < off.size(); ++r) uniform = off[r + 1] - off[r] == longest;
        if (uniform && longest <= 0xFFFFFFFFu) {
            c.offsets = nullptr;
            c.read_len = (uint32_t)longest;
        }
        return true;
    }
};

static uint64_t host_cache_budget() {
    if (const char *s = getenv("KBBQ_HOST_CACHE_MB")) return strtoull(s, nullptr, 10) << 20;
    const long pages = sysconf(_SC_PHYS_PAGES), psize = sysconf(_SC_PAGE_SIZE);
    const uint64_t ram = pages > 0 && psize > 0 ? (uint64_t)pages * (uint64_t)psize : 0;
    return std::min<uint64_t>(ram / 4, 64ULL << 30);
}

static int fail_engine(const char *what) {
    std::cerr << put_now << " Error: " << what << ": " << kbbq_last_error() << std::endl;
    return 1;
}

// hidden helpers for the CPU test-suite: exercise the reader, the name rules and the BGZF writer
// without touching the GPU
static int io_test(int argc, char *argv[]) {
    const std::string what = argc > 2 ? argv[2] : "";
    const int io_threads = getenv("KBBQ_IO_THREADS") ? atoi(getenv("KBBQ_IO_THREADS")) : 1;
    if (what == "parse" && argc > 3) {
        FastqReader in(argv[3], io_threads);
        if (!in.ok()) return 2;
        FastqRecord r;
        ReadGroups groups;
        int rc;
        while ((rc = in.next(r)) >= 0) {
            std::string rg, first;
            bool second = false;
            const bool ok = parse_read_name(r.name, rg, second, first);
            printf("%s\t%s\t%s\t%d\t%d\t%s\t%s\t%s\n", r.name.c_str(), r.comment.c_str(), ok ? rg.c_str() : "!", ok ? groups.index_of(rg) : -1,
                   (int)second, first.c_str(), r.seq.c_str(), r.qual.c_str());
        }
        printf("#end %d\n", rc);
        return 0;
    }
    if (what == "parse-fast" && argc > 3) {     // the block-parallel parser on the same file, same lines; "#complex" = not its shape
        FastqChunkParser in(argv[3], std::max(2, io_threads), argc > 4 ? atoi(argv[4]) : 4, true);
        if (!in.ok()) return 2;
        ReadGroups groups;
        while (auto piece = in.next()) {
            const FastqPiece &P = *piece;
            if (P.complex) { printf("#complex\n"); return 0; }
            std::vector<int> map;
            for (auto &n : P.rg_names) map.push_back(groups.index_of(n));
            for (size_t r = 0; r < P.n(); ++r) {
                const char *b = P.blob.data() + P.blob_off[r];
                const uint32_t nl = P.lens[3 * r], cl = P.lens[3 * r + 1], sl = P.lens[3 * r + 2];
                const std::string name(b, nl), comment(b + nl, cl), seq(b + nl + cl, sl);
                std::string rg, first, q(sl, ' ');
                bool second = false;
                parse_read_name(name, rg, second, first);
                for (uint32_t i = 0; i < sl; ++i) q[i] = (char)(P.qual[P.off[r] + i] + 33);
                if (seq != std::string((const char *)P.seq.data() + P.off[r], sl) || rg != P.rg_names[P.rg[r]] || second != (P.second[r] != 0)) return 3;
                printf("%s\t%s\t%s\t%d\t%d\t%s\t%s\t%s\n", name.c_str(), comment.c_str(), rg.c_str(), map[P.rg[r]], (int)second, first.c_str(),
                       seq.c_str(), q.c_str());
            }
            if (P.fatal_at >= 0) {
                printf("%s\t%s\t%s\t%d\n", P.fatal_name.c_str(), "?", "!", -1);
                break;
            }
        }
        printf("#end -1\n");
        return 0;
    }
    if (what == "bam" && argc > 3) {     // what the passes see of a BAM: --io-test bam FILE [use-oq]
        BamSource in(argv[3], argc > 4 && std::string(argv[4]) == "use-oq", io_threads);
        if (!in.ok()) return 2;
        printf("#text %zu genome %llu refs %zu\n", in.header().text.size(), (unsigned long long)in.header().genome_length(), in.header().refs.size());
        Item it;
        ReadGroups groups;
        int rc;
        while ((rc = in.next(it)) >= 0) {
            std::string q(it.qual.size(), ' ');
            for (size_t i = 0; i < q.size(); ++i) q[i] = (char)(it.qual[i] + 33);
            printf("%s\t%d\t%s\t%d\t%d\t%s\t%s\n", it.bam.name().c_str(), (int)it.bam.flag(), it.rg.c_str(), groups.index_of(it.rg), (int)it.second,
                   it.seq.c_str(), q.c_str());
        }
        printf("#end %d\n", rc);
        return 0;
    }
    if (what == "bam-scan" && argc > 3) {     // the first scan's host side alone, timed: --io-test bam-scan FILE [parse threads] [keep records 0/1]
        const int pt = argc > 4 ? atoi(argv[4]) : 4;
        const bool keep = argc > 5 ? atoi(argv[5]) != 0 : true;
        g_io_threads = io_threads;
        const auto t0 = std::chrono::steady_clock::now();
        Batch::Fast ff;
        auto *bp = new BamChunkParser(argv[3], false, std::max(2, io_threads), pt, keep);
        ff.parser.reset(bp);
        ff.lens_per_record = 1;
        if (!bp->ok()) return 2;
        Batch batch;
        batch.pack_on_host = false;
        ReadGroups groups;
        uint64_t reads = 0, bases = 0;
        double fill_s = 0;
        for (;;) {
            RecordStore st;
            const auto a = std::chrono::steady_clock::now();
            if (!batch.fill_fast(ff, groups, (size_t)1 << 20, keep ? &st : nullptr)) break;
            fill_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
            reads += batch.c.n_reads;
            bases += batch.c.n_bases;
        }
        const double all = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("reads %llu bases %llu wall %.3f s (in fill_fast %.3f s, of it waiting for pieces %.3f s) %.2f Gbases/s\n", (unsigned long long)reads,
               (unsigned long long)bases, all, fill_s, ff.wait_s, bases / all / 1e9);
        return 0;
    }
    if (what == "bam-fast" && argc > 3) {     // the same lines from the block-parallel BAM parser: --io-test bam-fast FILE [use-oq] [threads]
        BamChunkParser in(argv[3], argc > 4 && std::string(argv[4]) == "use-oq", std::max(2, io_threads), argc > 5 ? atoi(argv[5]) : 4, true);
        if (!in.ok()) return 2;
        printf("#text %zu genome %llu refs %zu\n", in.header().text.size(), (unsigned long long)in.header().genome_length(), in.header().refs.size());
        ReadGroups groups;
        int rc = -1;
        while (auto piece = in.next()) {
            const ReadPiece &P = *piece;
            std::vector<int> map;
            for (auto &n : P.rg_names) map.push_back(groups.index_of(n));
            BamRecord b;
            for (size_t r = 0; r < P.n(); ++r) {
                b.data.assign((const uint8_t *)P.blob.data() + P.blob_off[r], (const uint8_t *)P.blob.data() + P.blob_off[r + 1]);
                if (P.lens[r] != b.data.size()) return 3;
                const size_t len = P.off[r + 1] - P.off[r];
                std::string q(len, ' ');
                for (size_t i = 0; i < len; ++i) q[i] = (char)(P.qual[P.off[r] + i] + 33);
                printf("%s\t%d\t%s\t%d\t%d\t%s\t%s\n", b.name().c_str(), (int)b.flag(), P.rg_names[P.rg[r]].c_str(), map[P.rg[r]], (int)P.second[r],
                       std::string((const char *)P.seq.data() + P.off[r], len).c_str(), q.c_str());
            }
            if (P.fatal_at >= 0) { fflush(stdout); std::cerr << P.fatal_msg << std::flush; rc = SRC_FATAL; }
            else if (P.end_of_stream) rc = -2;
        }
        printf("#end %d\n", rc);
        return 0;
    }
    if (what == "bamcopy" && argc > 3) { // reader -> (OQ update) -> writer: --io-test bamcopy FILE [set-oq]
        BamReader in(argv[3], io_threads);
        if (!in.ok()) return 2;
        const bool set_oq = argc > 4 && std::string(argv[4]) == "set-oq";
        BgzfWriter out(stdout);
        BamWriter w(out);
        if (!w.write_header(in.header())) return 1;
        BamRecord b;
        std::string q;
        while (in.next(b) >= 0) {
            if (set_oq) {
                q.assign(b.l_seq(), ' ');
                for (size_t i = 0; i < q.size(); ++i) q[i] = (char)(b.qual()[i] + 33);
                int status = 0;
                if (!b.aux_update_string("OQ", q, status)) return 3;
            }
            if (!w.write(b)) return 1;
        }
        return out.close() ? 0 : 1;
    }
    if ((what == "synth-fastq" || what == "synth-bam") && argc > 4) {
        // The bench's own reads as a file on stdout: --io-test synth-fastq GENOME_LEN COVERAGE / synth-bam GENOME_LEN COVERAGE [oq]
        // (bench.py: seed 12345, 150-base reads, 100 N per million): k_synth -> record text / BAM records -> k_deflate, all on
        // the device.  The command line run on this file must log the bench's insert counts and write the bench's digest.
        const uint64_t G = strtoull(argv[3], nullptr, 10), cov = strtoull(argv[4], nullptr, 10);
        const bool bam = what == "synth-bam", oq = bam && argc > 5 && std::string(argv[5]) == "oq";
        kbbq_synth_params sp;
        memset(&sp, 0, sizeof sp);
        sp.seed = 12345; sp.genome_len = G; sp.read_len = 150; sp.n_reads = G * cov / 150; sp.n_rg = 1; sp.paired = 0; sp.n_per_million = 100;
        if (G < 150 || !sp.n_reads) return 2;
        kbbq_params prm;
        memset(&prm, 0, sizeof prm);
        prm.k = 32; prm.alpha = 0.1; prm.seed = 1; prm.n_rg = 1; prm.approx_kmers = 1000; prm.max_read_len = 150; prm.fpr_sampled = 0.01; prm.fpr_trusted = 0.0005;
        prm.bloom_seed = 0xA5A5A5A55A5A5A5AULL;
        kbbq_engine *e = nullptr;
        if (kbbq_engine_create(&prm, &e) < 0) { std::cerr << kbbq_last_error() << std::endl; return 1; }
        struct FreeEngine { kbbq_engine *e; ~FreeEngine() { kbbq_engine_destroy(e); } } free_engine{e};
        DeviceBgzfWriter out(stdout, 0);
        if (!out.ok()) return 1;
        if (bam) {
            BamHeader h;
            h.text = "@HD\tVN:1.6\tSO:unsorted\n@RG\tID:grp0\tSM:synth\n";
            h.refs.emplace_back("chr1", (uint32_t)std::min<uint64_t>(G, 0xFFFFFFFFull));
            BamWriter w(out);
            if (!w.write_header(h)) return 1;
        }
        const uint64_t step = (uint64_t)1 << 22;
        for (uint64_t first = 0; first < sp.n_reads; first += step)
            if (!out.synth_batch(e, &sp, first, std::min(step, sp.n_reads - first), bam ? (oq ? 2 : 1) : 0)) return 1;
        return out.close() ? 0 : 1;
    }
    if (what == "bgzf") {   // stdin -> BGZF on stdout: --io-test bgzf [threads]
        BgzfWriter out(stdout, argc > 3 ? atoi(argv[3]) : 1);
        std::vector<char> buf(1 << 16);
        size_t n;
        while ((n = fread(buf.data(), 1, buf.size(), stdin)) > 0)
            if (!out.write(buf.data(), n)) return 1;
        return out.close() ? 0 : 1;
    }
    return 2;
}

int main(int argc, char *argv[]) {
    // The runtime multiplexes a process's streams onto four hardware queues by default, and this process creates a dozen (reader,
    // engine, writer, copies): the engine's two streams then shared one, and what was queued "side by side" ran in order -- pass 3
    // of a 3e10-base file 0.55 s, with eight queues 0.47-0.50 (profiles/r04_e2e_3e10_hw_queues.txt).  Read once, at the first HIP call.
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    if (argc > 1 && std::string(argv[1]) == "--io-test") return io_test(argc, argv);
    int k = 32;
    long double alpha = 0;
    uint64_t genomelen = 0;
    unsigned coverage = 0;
    uint32_t seed = 0;
    bool set_oq = false, use_oq = false;
    int nthreads = 0;
    std::string fixedinput;
    int opt = 0, opt_idx = 0;
    while ((opt = getopt_long(argc, argv, "k:usg:c:f:a:t:", long_options, &opt_idx)) != -1) {
        switch (opt) {
            case 'k':
                k = std::stoi(std::string(optarg));
                if (k <= 0 || k > KBBQ_MAX_KMER) {
                    std::cerr << put_now << "  Error: k must be <= " << KBBQ_MAX_KMER << " and > 0." << std::endl;
                    return 1;   // the reference only prints this and goes on (kbbq.cc:102-104)
                }
                break;
            case 'u': use_oq = true; break;
            case 's': set_oq = true; break;
            case 'g': genomelen = std::stoull(std::string(optarg)); break;
            case 'c': coverage = (unsigned)std::stoul(std::string(optarg)); break;
            case 'f': fixedinput = std::string(optarg); break;
            case 'a': alpha = std::stold(std::string(optarg)); break;
            case 't':
                nthreads = std::stoi(std::string(optarg));
                if (nthreads < 0) std::cerr << put_now << " Error: threads must be >= 0." << std::endl;
                break;
            case '?':
            default:
                std::cerr << put_now << "  Unknown argument " << (char)opt << std::endl;
                return 1;
        }
    }
    // --threads sizes the BGZF compression pool like the reference's htslib pool (kbbq.cc:159-168); unlike the
    // reference, 0 does not mean "single-threaded" but "pick": the writer is the end-to-end bottleneck
    const int out_threads = nthreads > 0 ? nthreads : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    g_io_threads = out_threads;
    std::string filename("-");
    if (optind < argc) {
        filename = std::string(argv[optind]);
        while (++optind < argc) std::cerr << put_now << " Warning: Extra argument " << argv[optind] << " ignored." << std::endl;
    }
    const long double sampler_desiredfpr = 0.01, trusted_desiredfpr = 0.0005;   // kbbq.cc:155-156

    const Format fmt = sniff(filename);
    if (fmt == Format::unknown) {
        std::cerr << put_now << " Error opening file " << filename << std::endl;   // also: a pipe cannot be re-read by the passes
        return 1;
    }
    if (fmt == Format::cram) {
        std::cerr << put_now << " Error: CRAM input needs htslib, which this build does not have; use BAM or FASTQ." << std::endl;
        return 1;
    }
    const bool is_bam = fmt == Format::bam;
    // BGZF input that the host parsers read (BAM always) is inflated on the GPU as well
    if (!(getenv("KBBQ_HOST_DEFLATE") && atoi(getenv("KBBQ_HOST_DEFLATE")) != 0) && !(getenv("KBBQ_DEVICE_INFLATE") && atoi(getenv("KBBQ_DEVICE_INFLATE")) == 0))
        set_bgzf_source_factory(&DeviceBgzfSource::open);

    // One scan before the engine exists: total length (the reference's coverage pass, kbbq.cc:229-250),
    // read groups, longest read.  The packed batches of this scan are uploaded as they are made and stay
    // resident in HBM, so passes 1-4 run from device memory instead of decoding the file four more times
    // (288 GB of HBM hold a 30x human genome beside its filters); the records themselves are decoded once
    // more, for the output pass.  Falls back to re-reading per pass if the batches do not fit (or with
    // KBBQ_RESIDENT=0), and for the rare inputs where the passes would not see the same reads (an empty read
    // ends the reference's sampling and coverage loops but not the others).
    PhaseClock clock;
    ReadGroups groups;
    uint64_t seqlen = 0, n_reads = 0;
    size_t longest = 0;
    BamHeader bam_header;
    // reads per engine call; KBBQ_BATCH_READS shrinks it so that tests cross many batch boundaries with small files
    const size_t batch_reads = getenv("KBBQ_BATCH_READS") ? std::max<size_t>(1, strtoull(getenv("KBBQ_BATCH_READS"), nullptr, 10)) : (size_t)1 << 20;
    Batch batch;
    struct Resident {
        std::vector<kbbq_reads> dev;
        std::vector<RecordStore> recs;      // the records of the same batches, when they fit in host memory
        bool on = true, keep_recs = true;
        uint64_t bytes = 0, budget = 0, rec_bytes = 0, rec_budget = 0;
        void drop_recs() {
            std::vector<RecordStore>().swap(recs);
            keep_recs = false;
        }
        void drop() {
            for (auto &d : dev) { kbbq_reads_free_hints(&d); kbbq_reads_free(nullptr, &d); }
            dev.clear();
            drop_recs();
            on = false;
        }
    } resident;
    const bool fixed_mode = !fixedinput.empty();
    auto init_resident = [&]() {
        resident.on = true;
        resident.keep_recs = true;
        resident.bytes = resident.rec_bytes = 0;
        const char *env = getenv("KBBQ_RESIDENT");
        uint64_t free_b = 0, total_b = 0;
        if ((env && !strcmp(env, "0")) || fixed_mode || kbbq_device_memory(-1, &free_b, &total_b) < 0) resident.on = false;
        resident.budget = (uint64_t)(0.6 * (double)free_b);
        resident.rec_budget = host_cache_budget();
        if (!resident.on || !resident.rec_budget) resident.keep_recs = false;
    };
    init_resident();
    // The first scan parses its input with a pool: BAM always (bam_io.h: BamChunkParser), FASTQ when it is strictly
    // four-line (fastq_io.h: FastqChunkParser) -- anything else, found out while parsing, starts the scan over with the
    // serial reader.  KBBQ_SERIAL_PARSE=1: the serial readers at once.
    // A BGZF-compressed FASTQ file is read on the GPU (DeviceFastqInput): the compressed bytes go to the device, which
    // inflates, finds the records and packs them; every chunk of the file is one resident batch.  Anything that path does
    // not take -- another container, records that are not four lines, read groups in the names, reads that do not fit in HBM
    // -- starts over with the host parsers below.  KBBQ_DEVICE_READER=0: the host parsers at once.
    DeviceFastqInput dev_in;
    // (KBBQ_HOST_DEFLATE=1, the zlib writer of rounds 1-2, goes with the host readers: the A/B of the whole host I/O path)
    const bool host_io = getenv("KBBQ_HOST_DEFLATE") && atoi(getenv("KBBQ_HOST_DEFLATE")) != 0;
    if (!is_bam && !fixed_mode && !host_io && resident.on && filename != "-" && !(getenv("KBBQ_DEVICE_READER") && atoi(getenv("KBBQ_DEVICE_READER")) == 0) &&
        !(getenv("KBBQ_SERIAL_PARSE") && atoi(getenv("KBBQ_SERIAL_PARSE"))) && dev_in.open(filename)) {
        bool ok = true;
        // The inflated text stays in HBM beside the packed reads while both fit in three quarters of the free memory
        // (resident.budget is 60 %): pass 4 then takes the record text from there and the file is read and inflated once.
        // KBBQ_KEEP_TEXT=0: pass 4 reads the file again.
        bool keeping = !(getenv("KBBQ_KEEP_TEXT") && atoi(getenv("KBBQ_KEEP_TEXT")) == 0) && kbbq_fastq_reader_keep(dev_in.reader, 1) == 0;
        const uint64_t text_budget = resident.budget / 4 * 5;
        for (;;) {
            kbbq_fastq_chunk info;
            const int rc = dev_in.next_chunk(info);
            if (rc == 0) break;
            if (rc < 0) { ok = false; break; }
            dev_in.chunk_records.push_back(info.n_records);
            if (!info.n_records) continue;
            const uint64_t need = info.n_bases * 13 / 8 + info.n_records * 16 + (1 << 16);
            if (keeping) {
                uint64_t kept_chunks = 0, kept_bytes = 0;
                if (kbbq_fastq_reader_kept(dev_in.reader, &kept_chunks, &kept_bytes) < 0 || resident.bytes + need + kept_bytes > text_budget) {
                    kbbq_fastq_reader_keep(dev_in.reader, 0);
                    keeping = false;
                }
            }
            kbbq_reads d;
            const auto tb = std::chrono::steady_clock::now();
            if (info.longest > KBBQ_MAX_READ_LEN || resident.bytes + need > resident.budget || kbbq_fastq_reader_batch(dev_in.reader, &d) < 0) { ok = false; break; }
            if (kbbq_reads_alloc_hints(&d) < 0) { kbbq_reads_free(nullptr, &d); ok = false; break; }
            dev_in.batch_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tb).count();
            resident.dev.push_back(d);
            resident.bytes += need;
            seqlen += info.n_bases;
            n_reads += info.n_records;
            longest = std::max<size_t>(longest, info.longest);
        }
        if (ok && n_reads) {
            dev_in.active = true;
            uint64_t kept_chunks = 0;
            if (kbbq_fastq_reader_rewind(dev_in.reader) == 0 && kbbq_fastq_reader_kept(dev_in.reader, &kept_chunks, &dev_in.kept_bytes) == 0)
                dev_in.text_kept = kept_chunks == resident.dev.size();
            if (!dev_in.text_kept) { kbbq_fastq_reader_keep(dev_in.reader, 0); dev_in.kept_bytes = 0; }
            resident.keep_recs = false;      // the record text comes from the device's own copy of the input in pass 4
            groups.index_of(std::string());  // FASTQ without read-group fields: the one read group "" (readutils.cc:98-103)
        } else {
            seqlen = n_reads = 0;
            longest = 0;
            resident.drop();
            init_resident();
            dev_in.close();
        }
    }
    // A BAM file takes the same road (DeviceFastqInput::open_bam; include/kbbq_bgzf.h: kbbq_bam_reader): the header is
    // parsed here (its reference lengths are the genome length, kbbq.cc:196-216; its @RG ids are the table the record
    // kernel looks read groups up in), everything behind it on the device.  The compressed bytes of every chunk stay in HBM
    // for pass 4 while they fit; otherwise pass 4 reads the file again.  A shape that path does not take -- a read group
    // without an @RG line, a record the host codec would report or end the stream on -- starts over with BamChunkParser.
    if (is_bam && !fixed_mode && !host_io && resident.on && filename != "-" && !(getenv("KBBQ_DEVICE_READER") && atoi(getenv("KBBQ_DEVICE_READER")) == 0) &&
        !(getenv("KBBQ_SERIAL_PARSE") && atoi(getenv("KBBQ_SERIAL_PARSE")))) {
        BamReader head(filename, 1);
        bool ok = head.ok();
        if (ok) {
            bam_header = head.header();
            uint64_t header_bytes = 12 + bam_header.text.size();
            for (auto &r : bam_header.refs) header_bytes += 8 + r.first.size() + 1;
            // the ID fields of the @RG lines (SAMv1 1.3: tab-separated TAG:VALUE fields)
            std::vector<std::string> rg_ids;
            {
                const std::string &t = bam_header.text;
                for (size_t at = 0; at < t.size();) {
                    size_t eol = t.find('\n', at);
                    if (eol == std::string::npos) eol = t.size();
                    if (eol - at >= 3 && t.compare(at, 3, "@RG") == 0) {
                        for (size_t f = at; f < eol;) {
                            size_t tab = t.find('\t', f);
                            if (tab == std::string::npos || tab > eol) tab = eol;
                            if (tab - f >= 3 && t.compare(f, 3, "ID:") == 0) { rg_ids.push_back(t.substr(f + 3, tab - f - 3)); break; }
                            f = tab + 1;
                        }
                    }
                    at = eol + 1;
                }
            }
            ok = !rg_ids.empty() && rg_ids.size() < 65535 && dev_in.open_bam(filename, use_oq, (int32_t)bam_header.refs.size(), header_bytes, rg_ids);
            bool keeping = ok && !(getenv("KBBQ_KEEP_TEXT") && atoi(getenv("KBBQ_KEEP_TEXT")) == 0) && kbbq_bam_reader_keep(dev_in.bam, 1) == 0;
            const uint64_t text_budget = resident.budget / 4 * 5;
            while (ok) {
                kbbq_bam_chunk info;
                const int rc = dev_in.next_chunk(info);
                if (rc == 0) break;
                if (rc < 0) { ok = false; break; }
                dev_in.chunk_records.push_back(info.n_records);
                if (!info.n_records) continue;
                const uint64_t need = info.n_bases * 13 / 8 + info.n_records * 18 + (1 << 16);
                if (keeping) {
                    uint64_t kept_chunks = 0, kept_bytes = 0;
                    if (kbbq_bam_reader_kept(dev_in.bam, &kept_chunks, &kept_bytes) < 0 || resident.bytes + need + kept_bytes > text_budget) {
                        kbbq_bam_reader_keep(dev_in.bam, 0);
                        keeping = false;
                    }
                }
                kbbq_reads d;
                const auto tb = std::chrono::steady_clock::now();
                if (info.longest > KBBQ_MAX_READ_LEN || resident.bytes + need > resident.budget || kbbq_bam_reader_batch(dev_in.bam, &d) < 0) { ok = false; break; }
                if (kbbq_reads_alloc_hints(&d) < 0) { kbbq_reads_free(nullptr, &d); ok = false; break; }
                dev_in.batch_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tb).count();
                resident.dev.push_back(d);
                resident.bytes += need;
                seqlen += info.n_bases;
                n_reads += info.n_records;
                longest = std::max<size_t>(longest, info.longest);
                if (info.shortest == 0) ok = false;      // an empty read ends the reference's coverage and sampling loops: the host path's case
            }
            if (ok && set_oq && dev_in.oq_unwritable) ok = false;      // bam_aux_update_str would fail on some record: the host path reports it
            if (ok && n_reads) {
                // read groups in the order of their first records, as rg_to_int numbers them (readutils.cc:53-57)
                std::vector<uint32_t> order(rg_ids.size());
                uint32_t n_groups = 0;
                if (kbbq_bam_reader_read_groups(dev_in.bam, order.data(), (uint32_t)order.size(), &n_groups) < 0) ok = false;
                for (uint32_t g = 0; ok && g < n_groups; ++g) groups.index_of(rg_ids[order[g]]);
            }
        }
        if (ok && n_reads) {
            dev_in.active = true;
            uint64_t kept_chunks = 0;
            if (kbbq_bam_reader_rewind(dev_in.bam) == 0 && kbbq_bam_reader_kept(dev_in.bam, &kept_chunks, &dev_in.kept_bytes) == 0)
                dev_in.text_kept = kept_chunks == resident.dev.size();
            if (!dev_in.text_kept) { kbbq_bam_reader_keep(dev_in.bam, 0); dev_in.kept_bytes = 0; }
            resident.keep_recs = false;      // the records come from the device's own copy of the input in pass 4
        } else {
            groups = ReadGroups();
            seqlen = n_reads = 0;
            longest = 0;
            resident.drop();
            init_resident();
            dev_in.close();
        }
    }
    batch.pack_on_host = false;      // (the scan's batches only ever go to the device: packed there)
    bool scan_fast = false, any_empty = false;
    for (int attempt = 0; attempt < 2 && !dev_in.active; ++attempt) {
        const bool fast = attempt == 0 && g_io_threads > 1 && !(getenv("KBBQ_SERIAL_PARSE") && atoi(getenv("KBBQ_SERIAL_PARSE")));
        if (attempt == 1) {
            groups = ReadGroups();
            seqlen = n_reads = 0;
            longest = 0;
            resident.drop();
            init_resident();
        }
        std::unique_ptr<Source> in;
        Batch::Fast ff;
        if (fast && is_bam) {
            auto *bp = new BamChunkParser(filename, use_oq, g_io_threads, out_threads, resident.on && resident.keep_recs);
            ff.parser.reset(bp);
            ff.lens_per_record = 1;
            if (!bp->ok()) {
                std::cerr << put_now << " Error opening file " << filename << std::endl;
                return 1;
            }
            bam_header = bp->header();
        } else if (fast) {
            auto *fp = new FastqChunkParser(filename, g_io_threads, out_threads, resident.on && resident.keep_recs);
            ff.parser.reset(fp);
            if (!fp->ok()) {
                std::cerr << put_now << " Error opening file " << filename << std::endl;
                return 1;
            }
        } else {
            in = open_source(filename, is_bam, use_oq);
            if (!in->ok()) {
                std::cerr << put_now << " Error opening file " << filename << std::endl;
                return 1;
            }
            if (is_bam) bam_header = static_cast<BamSource *>(in.get())->header();
        }
        bool counting = true;    // the coverage pass stops at the first empty read; the other passes do not
        scan_fast = fast;
        for (;;) {
            const bool keep = resident.on && resident.keep_recs;
            RecordStore st;
            if (!(fast ? batch.fill_fast(ff, groups, batch_reads, keep ? &st : nullptr) : batch.fill(*in, groups, batch_reads, keep, is_bam))) break;
            for (size_t r = 0; r < batch.c.n_reads && counting; ++r) {
                const uint64_t l = batch.off[r + 1] - batch.off[r];
                if (l == 0) counting = false; else seqlen += l;
            }
            longest = std::max(longest, batch.longest);
            n_reads += batch.c.n_reads;
            if (batch.saw_empty) any_empty = true;
            if (batch.saw_empty && resident.on) resident.drop();
            if (resident.on && batch.longest <= KBBQ_MAX_READ_LEN) {
                // qualities 1 B + bases 1/4 + N mask 1/8 + two hint arrays 1/4 per base; offsets, flags, read groups per read
                const uint64_t need = batch.c.n_bases * 13 / 8 + batch.c.n_reads * 16 + (1 << 16);
                kbbq_reads d;
                if (resident.bytes + need > resident.budget ||
                    (batch.pack_on_host ? kbbq_reads_upload(nullptr, &batch.c, &d) : kbbq_reads_upload_text(nullptr, &batch.c, batch.seq.data(), &d)) < 0) {
                    resident.drop();
                } else if (kbbq_reads_alloc_hints(&d) < 0) {
                    kbbq_reads_free(nullptr, &d);
                    resident.drop();
                } else {
                    resident.dev.push_back(d);
                    resident.bytes += need;
                }
            }
            if (resident.on && resident.keep_recs) {
                if (!fast) {
                    st.lens.reserve(batch.c.n_reads * (is_bam ? 1 : 3));
                    if (is_bam) for (auto &b : batch.bam_recs) st.add(b);
                    else for (auto &f : batch.fq_recs) st.add(f);
                }
                resident.rec_bytes += st.bytes();
                if (resident.rec_bytes > resident.rec_budget) resident.drop_recs();
                else resident.recs.push_back(std::move(st));
            }
        }
        if (batch.fatal) return 1;
        if (fast && ff.complex) { scan_fast = false; continue; }
        break;
    }
    // Passes 1-3 of the streaming mode (the reads did not stay in HBM) read the file through the block-parallel parser as
    // well when the first scan found the stream of its shape and no empty read in it (an empty read ends the reference's
    // sampling loop: the serial reader's case); the serial reader otherwise.
    struct PassInput {
        std::unique_ptr<Source> serial;
        Batch::Fast fast;
        bool use_fast = false;
        bool next(Batch &b, ReadGroups &g, size_t max_reads) {
            return use_fast ? b.fill_fast(fast, g, max_reads, nullptr) : b.fill(*serial, g, max_reads, false);
        }
    };
    const bool passes_fast = scan_fast && !any_empty && !dev_in.active;
    auto open_pass = [&]() -> std::unique_ptr<PassInput> {
        std::unique_ptr<PassInput> in(new PassInput);
        if (passes_fast) {
            in->use_fast = true;
            if (is_bam) { in->fast.parser.reset(new BamChunkParser(filename, use_oq, g_io_threads, out_threads, false)); in->fast.lens_per_record = 1; }
            else in->fast.parser.reset(new FastqChunkParser(filename, g_io_threads, out_threads, false));
        } else {
            in->serial = open_source(filename, is_bam, use_oq);
        }
        return in;
    };
    batch.pack_on_host = true;       // (streaming passes hand host batches to the engine)
    if (longest > KBBQ_MAX_READ_LEN) {
        std::cerr << put_now << " Error: reads longer than " << KBBQ_MAX_READ_LEN << " bases are not supported by the GPU engine." << std::endl;
        return 1;
    }

    clock.mark("scan+pack+upload");
    kbbq_engine *e = nullptr;
    bool multi_device_done = false;      // KBBQ_DEVICES: passes 1-3 and the model ran sharded over several devices
    if (resident.on)
        std::cerr << put_now << " Reads are resident on the GPU: " << resident.dev.size() << " batches"
                  << (resident.keep_recs ? ", their records in host memory." : ".") << std::endl;

    if (!fixed_mode) {
        if (genomelen == 0) {
            if (is_bam) {   // kbbq.cc:196-216
                std::cerr << put_now << " Estimating genome length" << std::endl;
                genomelen = bam_header.genome_length();
                if (genomelen == 0) {
                    std::cerr << put_now << " Header does not contain genome information."
                              << " Unable to estimate genome length; please provide it on the command line"
                              << " using the --genomelen option." << std::endl;
                    return 1;
                }
                std::cerr << put_now << " Genome length is " << genomelen << " bp." << std::endl;
            } else {
                std::cerr << put_now << " Error: --genomelen must be specified if input is not a bam." << std::endl;
                return 1;
            }
        }
        if (alpha == 0) {   // kbbq.cc:227-252
            std::cerr << put_now << " Estimating alpha." << std::endl;
            if (coverage == 0) {
                std::cerr << put_now << " Estimating coverage." << std::endl;
                if (seqlen == 0) {
                    std::cerr << put_now << " Error: total sequence length in file " << filename << " is 0. Check that the file isn't empty." << std::endl;
                    return 1;
                }
                std::cerr << put_now << " Total Sequence length: " << seqlen << std::endl;
                std::cerr << put_now << " Genome length: " << genomelen << std::endl;
                coverage = (unsigned)(seqlen / genomelen);
                std::cerr << put_now << " Estimated coverage: " << coverage << std::endl;
                if (coverage == 0) {
                    std::cerr << put_now << " Error: estimated coverage is 0." << std::endl;
                    return 1;
                }
            }
            alpha = 7.0l / (long double)coverage;
        }
        if (coverage == 0) coverage = (unsigned)(7.0l / alpha);
        std::cerr << put_now << " Sampling kmers at rate " << alpha << std::endl;
        const unsigned long long approx_kmers = (unsigned long long)(genomelen * coverage * alpha);   // kbbq.cc:264
        if (const char *s = getenv("KBBQ_SEED")) seed = (uint32_t)strtoul(s, nullptr, 10);
        if (seed == 0) seed = time_pid_seed();
        std::cerr << put_now << " Seed: " << seed << std::endl;
        std::cerr << "p: " << (double)alpha << std::endl;   // KmerSubsampler ctor, htsiter.hh:143

        kbbq_params p;
        memset(&p, 0, sizeof p);
        p.k = k;
        p.device = 0;
        p.alpha = (double)alpha;
        p.seed = seed;
        p.n_rg = (int32_t)std::max<size_t>(1, groups.size());
        p.approx_kmers = approx_kmers;
        p.fpr_sampled = (double)sampler_desiredfpr;
        p.fpr_trusted = (double)trusted_desiredfpr;
        p.bloom_seed = KBBQ_DEFAULT_BLOOM_SEED;
        p.max_read_len = (int32_t)std::max<size_t>(1, longest);
        if (kbbq_engine_create(&p, &e) < 0) return fail_engine("cannot create the engine");

        // KBBQ_DEVICES=0,1,...: the hot path -- passes 1-3 and the model -- sharded over several GPUs of the node inside this
        // process (contiguous shards of the reads in file order, full filter replicas, three exchange steps
        // and a broadcast: include/kbbq_exchange.h).  The reads were made resident on the first device by the scan; every
        // other device gets a copy of its shard; one host thread per device runs the passes on its engine and meets the
        // others in the exchanges -- over RCCL when the devices are distinct, through device-to-device copies when one device
        // is listed several times (RCCL refuses that; KBBQ_EXCHANGE=local forces it).  The first device then holds the
        // global model and writes the output exactly as a one-device run does: same bytes, whatever the list.
        std::vector<int> devices;
        if (const char *dl = getenv("KBBQ_DEVICES")) {
            for (const char *q = dl; *q;) {
                char *end = nullptr;
                const long v = strtol(q, &end, 10);
                if (end == q) break;
                devices.push_back((int)v);
                q = *end == ',' ? end + 1 : end;
            }
        }
        if (devices.size() > 1 && (!resident.on || devices[0] != 0)) {
            std::cerr << put_now << " KBBQ_DEVICES needs the reads resident on device 0 (the first entry): running on one device." << std::endl;
            devices.clear();
        }
        if (devices.size() > 1) {
            const int N = (int)devices.size();
            const size_t nb = resident.dev.size();
            // global k-mer ordinals of the batches (the sampler's draw stream is one, in file order), shards balanced by bases
            std::vector<uint64_t> ordinal(nb + 1, 0), bases(nb + 1, 0);
            for (size_t b = 0; b < nb; ++b) {
                uint64_t nk = 0;
                if (kbbq_count_kmer_positions(e, &resident.dev[b], &nk) < 0) return fail_engine("sampling");
                ordinal[b + 1] = ordinal[b] + nk;
                bases[b + 1] = bases[b] + resident.dev[b].n_bases;
            }
            std::vector<size_t> first(N + 1, nb);
            first[0] = 0;
            for (int d = 1; d < N; ++d) {
                const uint64_t want = bases[nb] * (uint64_t)d / (uint64_t)N;
                size_t b = first[d - 1];
                while (b < nb && bases[b] < want) ++b;
                first[d] = b;
            }
            std::vector<kbbq_engine *> eng(N, nullptr);
            std::vector<std::vector<kbbq_reads>> shard(N);
            std::vector<kbbq_group *> grp(N, nullptr);
            eng[0] = e;
            bool distinct = true;
            for (int a = 0; a < N; ++a) for (int b = a + 1; b < N; ++b) if (devices[a] == devices[b]) distinct = false;
            const bool use_rccl = distinct && !(getenv("KBBQ_EXCHANGE") && !strcmp(getenv("KBBQ_EXCHANGE"), "local"));
            uint8_t uid[KBBQ_RCCL_ID_BYTES];
            if (use_rccl) { if (kbbq_group_rccl_unique_id(uid) < 0) return fail_engine("RCCL"); }
            else if (kbbq_group_local_create(N, grp.data()) < 0) return fail_engine("exchange group");
            for (int d = 1; d < N; ++d) {
                kbbq_params pd = p;
                pd.device = devices[d];
                if (kbbq_engine_create(&pd, &eng[d]) < 0) return fail_engine("cannot create an engine");
                for (size_t b = first[d]; b < first[d + 1]; ++b) {
                    kbbq_reads c;
                    if (kbbq_reads_clone(&resident.dev[b], devices[d], 1, &c) < 0) return fail_engine("copying a shard");
                    shard[d].push_back(c);
                }
            }
            for (size_t b = first[0]; b < first[1]; ++b) shard[0].push_back(resident.dev[b]);      // (views: owned by `resident`)
            std::cerr << put_now << " Passes 1-3 on " << N << " devices (" << (use_rccl ? "RCCL" : "in-process copies") << "): shards of";
            for (int d = 0; d < N; ++d) std::cerr << " " << first[d + 1] - first[d];
            std::cerr << " batches." << std::endl;
            char alpha_text[64];
            snprintf(alpha_text, sizeof alpha_text, "%.25Le", alpha);
            std::atomic<int> gate_seen(0);
            // one rank: returns 0, or exits the process on an error (a rank that fails must not leave the others in a collective)
            auto die = [&](const char *what) { std::cerr << put_now << " Error " << what << ": " << kbbq_last_error() << std::endl; _exit(1); };
            auto rank_body = [&](int d) {
                kbbq_engine *ed = eng[d];
                if (use_rccl && kbbq_group_rccl_create(uid, d, N, devices[d], &grp[d]) < 0) die("joining the RCCL group");
                for (size_t i = 0; i < shard[d].size(); ++i)
                    if (kbbq_sample_batch(ed, &shard[d][i], ordinal[first[d] + i]) < 0) die("sampling");
                uint64_t inserted = 0, total = 0;
                if (kbbq_sample_finish(ed, &inserted) < 0 || kbbq_exchange_filter(ed, 0, grp[d], 0, &total) < 0) die("sampling");
                if (d == 0) std::cerr << put_now << " Sampled " << total << " valid kmers." << std::endl;
                // every rank computes the same thresholds from the same (global) filter and count (kbbq.cc:304-331)
                char p_text[64];
                std::vector<int32_t> thresholds(k + 1);
                double fprd = 0;
                const int gate = kbbq_compute_thresholds(ed, alpha_text, thresholds.data(), &fprd, p_text, sizeof p_text);
                if (gate < 0) die("thresholds");
                if (d == 0) {
                    const long double fpr = fprd;
                    std::cerr << put_now << " Approximate false positive rate: " << fpr << std::endl;
                    if (gate != 1) {
                        const long double p_hit = strtold(p_text, nullptr);
                        std::cerr << put_now << " log CDF: [ ";
                        for (long double c : log_binom_cdf_values((unsigned long long)k, p_hit)) std::cerr << c << " ";
                        std::cerr << "]" << std::endl;
                    }
                }
                if (gate == 1) { gate_seen = 1; return; }      // (every rank sees the same gate)
                if (d == 0) { clock.mark("pass1"); std::cerr << put_now << " Finding trusted kmers" << std::endl; }
                for (auto &b : shard[d])
                    if (kbbq_trusted_batch(ed, &b, nullptr) < 0) die("finding trusted kmers");
                uint64_t trusted_inserted = 0;
                if (kbbq_trusted_finish(ed, nullptr) < 0 || kbbq_exchange_filter(ed, 1, grp[d], 0, &trusted_inserted) < 0) die("finding trusted kmers");
                if (d == 0 && getenv("KBBQ_QUAL_DIGEST") && atoi(getenv("KBBQ_QUAL_DIGEST"))) std::cerr << "[digest] trusted_inserted " << trusted_inserted << std::endl;
                if (d == 0) { clock.mark("pass2"); std::cerr << put_now << " Finding errors" << std::endl; }
                for (auto &b : shard[d])
                    if (kbbq_errors_batch(ed, &b, nullptr) < 0) die("finding errors");
                if (kbbq_exchange_histograms(ed, grp[d]) < 0) die("summing the histograms");
                if (d == 0) { clock.mark("pass3"); std::cerr << put_now << " Training model" << std::endl; }
                if (kbbq_exchange_dq(ed, grp[d]) < 0) die("training");      // rank 0 trains, the tables are broadcast
            };
            std::vector<std::thread> ranks;
            for (int d = 1; d < N; ++d) ranks.emplace_back(rank_body, d);
            rank_body(0);
            for (auto &t : ranks) t.join();
            for (int d = 0; d < N; ++d) if (grp[d]) kbbq_group_destroy(grp[d]);
            for (int d = 1; d < N; ++d) {
                for (auto &c : shard[d]) { kbbq_reads_free_hints(&c); kbbq_reads_free(eng[d], &c); }
                kbbq_engine_destroy(eng[d]);
            }
            if (gate_seen) {
                std::cerr << put_now << " Error: false positive rate is too high. "
                          << "Increase genomelen parameter and try again." << std::endl;
                return 1;
            }
            multi_device_done = true;
        }

        // pass 1, kbbq.cc:277-283
        if (!multi_device_done) {
            uint64_t ordinal = 0, nk = 0;
            if (resident.on) {
                for (auto &d : resident.dev) {
                    if (kbbq_sample_batch(e, &d, ordinal) < 0) return fail_engine("sampling");
                    if (kbbq_count_kmer_positions(e, &d, &nk) < 0) return fail_engine("sampling");
                    ordinal += nk;
                }
            } else {
                std::unique_ptr<PassInput> in = open_pass();
                batch.stop_at_empty = true;
                while (in->next(batch, groups, batch_reads)) {
                    if (kbbq_sample_batch(e, &batch.c, ordinal) < 0) return fail_engine("sampling");
                    if (kbbq_count_kmer_positions(e, &batch.c, &nk) < 0) return fail_engine("sampling");
                    ordinal += nk;
                }
                if (batch.fatal) return 1;
                batch.stop_at_empty = false;
                batch.ended = false;
            }
            uint64_t inserted = 0;
            if (kbbq_sample_finish(e, &inserted) < 0) return fail_engine("sampling");
            std::cerr << put_now << " Sampled " << inserted << " valid kmers." << std::endl;
        }
        if (!multi_device_done) {
        // kbbq.cc:304-331
        char alpha_text[64], p_text[64];
        snprintf(alpha_text, sizeof alpha_text, "%.25Le", alpha);
        std::vector<int32_t> thresholds(k + 1);
        double fprd = 0;
        const int gate = kbbq_compute_thresholds(e, alpha_text, thresholds.data(), &fprd, p_text, sizeof p_text);
        if (gate < 0) return fail_engine("thresholds");
        const long double fpr = fprd;
        std::cerr << put_now << " Approximate false positive rate: " << fpr << std::endl;
        if (gate == 1) {
            std::cerr << put_now << " Error: false positive rate is too high. "
                      << "Increase genomelen parameter and try again." << std::endl;
            return 1;
        }
        {
            const long double p_hit = strtold(p_text, nullptr);
            std::cerr << put_now << " log CDF: [ ";
            for (long double c : log_binom_cdf_values((unsigned long long)k, p_hit)) std::cerr << c << " ";
            std::cerr << "]" << std::endl;
        }
        // pass 2, kbbq.cc:333-337
        clock.mark("pass1");
        std::cerr << put_now << " Finding trusted kmers" << std::endl;
        {
            if (resident.on) {
                for (auto &d : resident.dev)
                    if (kbbq_trusted_batch(e, &d, nullptr) < 0) return fail_engine("finding trusted kmers");
            } else {
                std::unique_ptr<PassInput> in = open_pass();
                while (in->next(batch, groups, batch_reads))
                    if (kbbq_trusted_batch(e, &batch.c, nullptr) < 0) return fail_engine("finding trusted kmers");
                if (batch.fatal) return 1;
            }
            uint64_t trusted_inserted = 0;
            if (kbbq_trusted_finish(e, &trusted_inserted) < 0) return fail_engine("finding trusted kmers");
            if (getenv("KBBQ_QUAL_DIGEST") && atoi(getenv("KBBQ_QUAL_DIGEST")))      // (the reference prints no count here; bench.py's result.trusted_inserted)
                std::cerr << "[digest] trusted_inserted " << trusted_inserted << std::endl;
        }
        // pass 3, kbbq.cc:363-366
        clock.mark("pass2");
        std::cerr << put_now << " Finding errors" << std::endl;
        {
            if (resident.on) {
                for (auto &d : resident.dev)
                    if (kbbq_errors_batch(e, &d, nullptr) < 0) return fail_engine("finding errors");
            } else {
                std::unique_ptr<PassInput> in = open_pass();
                while (in->next(batch, groups, batch_reads))
                    if (kbbq_errors_batch(e, &batch.c, nullptr) < 0) return fail_engine("finding errors");
                if (batch.fatal) return 1;
            }
        }
        }      // (!multi_device_done)
    } else {
        // --fixed, kbbq.cc:367-378: errors = bases that differ from the corrected file
        std::cerr << put_now << " Using fixed file to find errors." << std::endl;
        kbbq_params p;
        memset(&p, 0, sizeof p);
        p.k = k; p.alpha = 0.5; p.seed = 1; p.approx_kmers = 1000;   // the filters are not used in this mode
        p.n_rg = (int32_t)std::max<size_t>(1, groups.size());
        p.fpr_sampled = 0.01; p.fpr_trusted = 0.0005; p.bloom_seed = KBBQ_DEFAULT_BLOOM_SEED;
        p.max_read_len = (int32_t)std::max<size_t>(1, longest);
        if (kbbq_engine_create(&p, &e) < 0) return fail_engine("cannot create the engine");
        // the second file is opened in the FIRST file's format (kbbq.cc:370 passes is_bam)
        std::unique_ptr<Source> in = open_source(filename, is_bam, use_oq), fixed = open_source(fixedinput, is_bam, use_oq);
        if (!fixed->ok()) {
            std::cerr << put_now << " Error opening file " << fixedinput << std::endl;
            return 1;
        }
        Batch fb;
        ReadGroups fixed_groups;
        while (batch.fill(*in, groups, batch_reads, false) && fb.fill(*fixed, fixed_groups, batch.c.n_reads, false)) {
            std::vector<uint64_t> err(batch.c.n_bases / 64 + 2, 0);
            const size_t nr = std::min<size_t>(batch.c.n_reads, fb.c.n_reads);
            for (size_t r = 0; r < nr; ++r) {
                const uint64_t a = batch.off[r], len = batch.off[r + 1] - a, b = fb.off[r], flen = fb.off[r + 1] - b;
                for (uint64_t i = 0; i < len && i < flen; ++i)
                    if (batch.seq[a + i] != fb.seq[b + i]) err[(a + i) >> 6] |= 1ULL << ((a + i) & 63);
            }
            if (fb.c.n_reads < batch.c.n_reads) {   // the fixed file ended first: the reference stops consuming there
                batch.c.n_reads = nr;
                batch.c.n_bases = batch.off[nr];
            }
            if (kbbq_tally_batch(e, &batch.c, err.data()) < 0) return fail_engine("tally");
        }
        if (batch.fatal || fb.fatal) return 1;
    }

    // kbbq.cc:405-407
    if (!multi_device_done) {
        clock.mark("pass3");
        std::cerr << put_now << " Training model" << std::endl;
        if (kbbq_train(e) < 0) return fail_engine("training");
    }

    // pass 4, kbbq.cc:455-457: recalibrate_and_write(file, dqs, "-")
    clock.mark("model");
    std::cerr << put_now << " Recalibrating file" << std::endl;
    uint64_t out_payload = 0, out_compressed = 0;
    double ms_format = 0, ms_deflate = 0, ms_gather = 0;
    {
        // The BGZF layer: the encoder on the GPU (DeviceBgzfWriter), or -- KBBQ_HOST_DEFLATE=1, the A/B switch -- zlib on
        // a pool of host threads as in rounds 1-2.  Same decompressed stream.
        const bool host_deflate = getenv("KBBQ_HOST_DEFLATE") && atoi(getenv("KBBQ_HOST_DEFLATE")) != 0;
        std::unique_ptr<DeviceBgzfWriter> dev_out;
        std::unique_ptr<BgzfWriter> host_out;
        if (host_deflate) {
            host_out.reset(new BgzfWriter(stdout, out_threads));
        } else {
            dev_out.reset(new DeviceBgzfWriter(stdout, 0));
            if (!dev_out->ok()) return fail_engine("cannot create the BGZF writer");
        }
        ByteSink &out = host_deflate ? static_cast<ByteSink &>(*host_out) : static_cast<ByteSink &>(*dev_out);
        BamWriter bam_out(out);
        if (is_bam && !bam_out.write_header(bam_header)) return 1;      // BamFile::open_out, htsiter.cc:35-42
        std::vector<uint8_t> newq;
        std::string qtext, line;
        // FastqFile::write, htsiter.cc:75-86 (the comment goes on the '+' line); qualities as text, htsiter.cc:61-65
        auto emit_fastq = [&](const char *name, size_t nl, const char *comment, size_t cl, const char *seq, size_t sl, const uint8_t *q) {
            line.clear();
            line += '@'; line.append(name, nl); line += '\n'; line.append(seq, sl); line += "\n+"; line.append(comment, cl); line += '\n';
            const size_t at = line.size();
            line.resize(at + sl);
            for (size_t i = 0; i < sl; ++i) line[at + i] = (char)(q[i] + 33);
            line += '\n';
            return out.write(line.data(), line.size());
        };
        // BamFile::recalibrate + write, htsiter.cc:11-45
        auto emit_bam = [&](BamRecord &b, const uint8_t *q) -> int {
            const size_t len = b.l_seq();
            if (set_oq) {
                qtext.resize(len);
                for (size_t i = 0; i < len; ++i) qtext[i] = (char)(b.qual()[i] + 33);
                int status = 0;
                if (!b.aux_update_string("OQ", qtext, status)) {
                    std::cerr << "Tag data is corrupt. Repair the tags and try again." << std::endl;
                    return -1;   // std::invalid_argument("Unable to update OQ tag.") in the reference
                }
            }
            if (b.reverse()) std::reverse_copy(q, q + len, b.qual());
            else std::copy(q, q + len, b.qual());
            return bam_out.write(b) ? 0 : -1;
        };
        if (dev_in.active && dev_out) {
            // The device path: the file's chunks again (inflate + record index, as in the first scan), pass 4 into a device
            // array, the text assembled from the device's own copy of the input, deflated, written.  Nothing but compressed
            // bytes crosses the host link in either direction.
            void *d_q[2] = {nullptr, nullptr};
            size_t d_q_bytes[2] = {0, 0};
            struct FreeQ { kbbq_engine *e; void **p; ~FreeQ() { for (int i = 0; i < 2; ++i) if (p[i]) kbbq_device_free(e, p[i]); } } free_q{e, d_q};
            size_t bi = 0;
            // KBBQ_QUAL_DIGEST=1: the sum of every recalibrated quality, taken on the device from the array the writer reads
            // (the number bench.py prints as recal_qual_sum for the same reads)
            const bool want_digest = getenv("KBBQ_QUAL_DIGEST") && atoi(getenv("KBBQ_QUAL_DIGEST"));
            if (!dev_in.text_kept) {
                dev_in.start_pass();
                if ((dev_in.bam ? kbbq_bam_reader_rewind(dev_in.bam) : kbbq_fastq_reader_rewind(dev_in.reader)) < 0) return fail_engine("recalibrating");
            }
            for (size_t ci = 0; ci < dev_in.chunk_records.size(); ++ci) {
                kbbq_fastq_chunk info;
                if (dev_in.text_kept && dev_in.bam) {
                    // the chunk's compressed bytes are still on the device: inflated and indexed again there
                    if (!dev_in.chunk_records[ci]) continue;
                    if (kbbq_bam_reader_select(dev_in.bam, bi, &info) < 0 || info.n_records != dev_in.chunk_records[ci]) return fail_engine("recalibrating");
                } else if (dev_in.text_kept) {
                    // the chunk's text and index are still on the device
                    if (!dev_in.chunk_records[ci]) continue;
                    if (kbbq_fastq_reader_select(dev_in.reader, bi, &info) < 0 || info.n_records != dev_in.chunk_records[ci] ||
                        kbbq_fastq_reader_attach(dev_in.reader, &resident.dev[bi]) < 0)
                        return fail_engine("recalibrating");
                } else if (dev_in.next_chunk(info) != 1 || info.n_records != dev_in.chunk_records[ci]) {
                    std::cerr << put_now << " Error: the input changed between the passes." << std::endl;
                    return 1;
                }
                if (!info.n_records) continue;
                const kbbq_reads &d = resident.dev[bi];
                const int t = (int)(bi & 1);
                ++bi;
                if (!dev_out->drain_to(1)) return 1;      // (the array this batch writes was read by the submission two back)
                if (d_q_bytes[t] < d.n_bases + 16) {
                    if (d_q[t] && kbbq_device_free(e, d_q[t]) < 0) return fail_engine("recalibrating");
                    d_q[t] = nullptr;
                    d_q_bytes[t] = d.n_bases + d.n_bases / 8 + 4096;
                    if (kbbq_device_alloc(e, d_q_bytes[t], &d_q[t]) < 0) return fail_engine("recalibrating");
                }
                if (kbbq_recalibrate_batch(e, &d, (uint8_t *)d_q[t]) < 0) return fail_engine("recalibrating");
                if (want_digest && kbbq_digest_add(e, (const uint8_t *)d_q[t], d.n_bases) < 0) return fail_engine("recalibrating");
                if (dev_in.bam ? !dev_out->bam_chunk(dev_in.bam, (const uint8_t *)d_q[t], set_oq, kbbq_engine_stream(e))
                               : !dev_out->reader_chunk(dev_in.reader, (const uint8_t *)d_q[t], kbbq_engine_stream(e)))
                    return 1;
            }
            if (!dev_out->drain()) return 1;
            if (want_digest) {
                uint64_t sum = 0;
                if (kbbq_digest_get(e, &sum, 1) < 0) return fail_engine("recalibrating");
                std::cerr << "[digest] recal_qual_sum " << sum << " reads " << n_reads << " bases " << seqlen << std::endl;
            }
        } else if (resident.on && resident.keep_recs && !is_bam && dev_out) {
            // FASTQ, every batch in HBM, its record text in host memory: the new qualities never leave the GPU.  Pass 4
            // writes them to a device array, the writer assembles "@name\nseq\n+comment\nqual\n" there (FastqFile::write,
            // htsiter.cc:75-86), deflates and hands back finished blocks; two batches are in flight, so the kernels of
            // one run while the blocks of the one before are written out.
            void *d_q[2] = {nullptr, nullptr};
            size_t d_q_bytes[2] = {0, 0};
            struct FreeQ { kbbq_engine *e; void **p; ~FreeQ() { for (int i = 0; i < 2; ++i) if (p[i]) kbbq_device_free(e, p[i]); } } free_q{e, d_q};
            for (size_t bi = 0; bi < resident.dev.size(); ++bi) {
                const kbbq_reads &d = resident.dev[bi];
                const RecordStore &st = resident.recs[bi];
                const int t = (int)(bi & 1);
                // the array this batch writes was read by the submission two batches ago: that one must be through
                if (!dev_out->drain_to(1)) return 1;      // (the array this batch writes was read by the submission two back)
                if (d_q_bytes[t] < d.n_bases + 16) {
                    if (d_q[t] && kbbq_device_free(e, d_q[t]) < 0) return fail_engine("recalibrating");
                    d_q[t] = nullptr;
                    d_q_bytes[t] = d.n_bases + d.n_bases / 8 + 4096;
                    if (kbbq_device_alloc(e, d_q_bytes[t], &d_q[t]) < 0) return fail_engine("recalibrating");
                }
                if (kbbq_recalibrate_batch(e, &d, (uint8_t *)d_q[t]) < 0) return fail_engine("recalibrating");
                if (!dev_out->fastq_batch(st.blob.data(), st.lens.data(), d.n_reads, (const uint8_t *)d_q[t], d.offsets, d.read_len,
                                          kbbq_engine_stream(e)))
                    return 1;
            }
            if (!dev_out->drain()) return 1;
        } else if (resident.on && resident.keep_recs && is_bam && dev_out) {
            // BAM, every batch in HBM, its alignment blocks in host memory: BamFile::recalibrate + write (htsiter.cc:11-45) for a
            // whole batch by a pool -- every thread rewrites a run of records (OQ tag, qualities, reversed for reverse-strand
            // reads) into a buffer of its own, the runs are copied side by side into one page-locked buffer, and the
            // encoder on the GPU takes it from there.
            const unsigned T = (unsigned)std::max(1, out_threads);
            std::vector<std::string> part(T);
            std::vector<int> part_rc(T, 0);
            char *pin = nullptr;
            size_t pin_bytes = 0;
            struct FreePin { char **p; ~FreePin() { if (*p) kbbq_host_free(*p); } } free_pin{&pin};
            for (size_t bi = 0; bi < resident.dev.size(); ++bi) {
                const kbbq_reads &d = resident.dev[bi];
                const RecordStore &st = resident.recs[bi];
                newq.assign(d.n_bases + 16, 0);
                if (kbbq_recalibrate_batch_host(e, &d, newq.data()) < 0) return fail_engine("recalibrating");
                // where every record's block and qualities start
                std::vector<uint64_t> rec_at(d.n_reads + 1), q_at(d.n_reads + 1);
                {
                    uint64_t at = 0, qa = 0;
                    for (size_t r = 0; r < d.n_reads; ++r) {
                        rec_at[r] = at; q_at[r] = qa;
                        at += st.lens[r];
                        const uint8_t *rec = (const uint8_t *)st.blob.data() + rec_at[r];
                        qa += (uint64_t)rec[16] | ((uint64_t)rec[17] << 8) | ((uint64_t)rec[18] << 16) | ((uint64_t)rec[19] << 24);      // l_seq
                    }
                    rec_at[d.n_reads] = at; q_at[d.n_reads] = qa;
                }
                std::vector<std::thread> pool;
                for (unsigned t = 0; t < T; ++t) {
                    pool.emplace_back([&, t] {
                        const size_t r0 = d.n_reads * t / T, r1 = d.n_reads * (t + 1) / T;
                        std::string &out_s = part[t];
                        out_s.clear();
                        out_s.reserve((size_t)(rec_at[r1] - rec_at[r0]) + (r1 - r0) * (set_oq ? 8 : 4) + (set_oq ? (size_t)(q_at[r1] - q_at[r0]) : 0));
                        BamRecord b;
                        std::string qtext_t;
                        for (size_t r = r0; r < r1; ++r) {
                            b.data.assign((const uint8_t *)st.blob.data() + rec_at[r], (const uint8_t *)st.blob.data() + rec_at[r + 1]);
                            const size_t len = b.l_seq();
                            const uint8_t *q = newq.data() + q_at[r];
                            if (set_oq) {
                                qtext_t.resize(len);
                                for (size_t i = 0; i < len; ++i) qtext_t[i] = (char)(b.qual()[i] + 33);
                                int status = 0;
                                if (!b.aux_update_string("OQ", qtext_t, status)) { part_rc[t] = -1; return; }
                            }
                            if (b.reverse()) std::reverse_copy(q, q + len, b.qual());
                            else std::copy(q, q + len, b.qual());
                            const uint32_t n = (uint32_t)b.data.size();
                            const char len4[4] = {(char)(n & 0xFF), (char)((n >> 8) & 0xFF), (char)((n >> 16) & 0xFF), (char)((n >> 24) & 0xFF)};
                            out_s.append(len4, 4);
                            out_s.append((const char *)b.data.data(), b.data.size());
                        }
                    });
                }
                for (auto &th : pool) th.join();
                size_t total = 0;
                for (unsigned t = 0; t < T; ++t) {
                    if (part_rc[t] < 0) {
                        std::cerr << "Tag data is corrupt. Repair the tags and try again." << std::endl;
                        return 1;      // std::invalid_argument("Unable to update OQ tag.") in the reference
                    }
                    total += part[t].size();
                }
                if (pin_bytes < total) {
                    if (pin) kbbq_host_free(pin);
                    pin = nullptr;
                    pin_bytes = total + total / 8 + 4096;
                    void *p = nullptr;
                    if (kbbq_host_alloc(pin_bytes, &p) < 0) return fail_engine("recalibrating");
                    pin = (char *)p;
                }
                {
                    std::vector<std::thread> copiers;
                    size_t at = 0;
                    for (unsigned t = 0; t < T; ++t) {
                        copiers.emplace_back([&, t, at] { memcpy(pin + at, part[t].data(), part[t].size()); });
                        at += part[t].size();
                    }
                    for (auto &th : copiers) th.join();
                }
                if (!dev_out->submit_buffer(pin, total)) return 1;
            }
            if (!dev_out->drain()) return 1;
        } else if (resident.on && resident.keep_recs) {
            // every batch is in HBM and its records are in host memory: nothing is decoded again
            BamRecord b;
            for (size_t bi = 0; bi < resident.dev.size(); ++bi) {
                const kbbq_reads &d = resident.dev[bi];
                const RecordStore &st = resident.recs[bi];
                newq.assign(d.n_bases + 16, 0);
                if (kbbq_recalibrate_batch_host(e, &d, newq.data()) < 0) return fail_engine("recalibrating");
                size_t at = 0, qa = 0;
                for (size_t r = 0; r < d.n_reads; ++r) {
                    if (is_bam) {
                        const uint32_t n = st.lens[r];
                        b.data.assign((const uint8_t *)st.blob.data() + at, (const uint8_t *)st.blob.data() + at + n);
                        at += n;
                        const size_t len = b.l_seq();
                        if (emit_bam(b, newq.data() + qa) < 0) return 1;
                        qa += len;
                    } else {
                        const uint32_t nl = st.lens[3 * r], cl = st.lens[3 * r + 1], sl = st.lens[3 * r + 2];
                        const char *p = st.blob.data() + at;
                        if (!emit_fastq(p, nl, p + nl, cl, p + nl + cl, sl, newq.data() + qa)) return 1;
                        at += (size_t)nl + cl + sl;
                        qa += sl;
                    }
                }
            }
        } else if (passes_fast && !is_bam && dev_out) {
            // FASTQ whose records were not kept (streaming mode, or the host cache was too small): the parsers decode the
            // file once more, and every batch goes the way of a resident one -- on the device (the resident copy, or
            // uploaded now with its bases packed there), pass 4 into a device array, the text assembled and deflated there.
            PassInput in;
            in.use_fast = true;
            in.fast.parser.reset(new FastqChunkParser(filename, g_io_threads, out_threads, true));
            void *d_q[2] = {nullptr, nullptr};
            size_t d_q_bytes[2] = {0, 0};
            kbbq_reads up[2];
            bool up_live[2] = {false, false};
            struct FreeAll {
                kbbq_engine *e; void **q; kbbq_reads *up; bool *live;
                ~FreeAll() { for (int i = 0; i < 2; ++i) { if (q[i]) kbbq_device_free(e, q[i]); if (live[i]) kbbq_reads_free(e, &up[i]); } }
            } free_all{e, d_q, up, up_live};
            batch.pack_on_host = false;
            size_t bi = 0;
            for (size_t n = 0;; ++n) {
                RecordStore st;
                if (!batch.fill_fast(in.fast, groups, batch_reads, &st)) break;
                const int t = (int)(n & 1);
                // the arrays of slot t were read by the submission two batches ago: that one must be through
                if (!dev_out->drain_to(1)) return 1;      // (the array this batch writes was read by the submission two back)
                if (up_live[t]) { kbbq_reads_free(e, &up[t]); up_live[t] = false; }
                const kbbq_reads *d = nullptr;
                if (resident.on) {
                    if (bi >= resident.dev.size() || resident.dev[bi].n_bases != batch.c.n_bases || resident.dev[bi].n_reads != batch.c.n_reads) {
                        std::cerr << put_now << " Error: the input changed between the passes." << std::endl;
                        return 1;
                    }
                    d = &resident.dev[bi++];
                } else {
                    if (kbbq_reads_upload_text(e, &batch.c, batch.seq.data(), &up[t]) < 0) return fail_engine("recalibrating");
                    up_live[t] = true;
                    d = &up[t];
                }
                if (d_q_bytes[t] < d->n_bases + 16) {
                    if (d_q[t] && kbbq_device_free(e, d_q[t]) < 0) return fail_engine("recalibrating");
                    d_q[t] = nullptr;
                    d_q_bytes[t] = d->n_bases + d->n_bases / 8 + 4096;
                    if (kbbq_device_alloc(e, d_q_bytes[t], &d_q[t]) < 0) return fail_engine("recalibrating");
                }
                if (kbbq_recalibrate_batch(e, d, (uint8_t *)d_q[t]) < 0) return fail_engine("recalibrating");
                if (!dev_out->fastq_batch(st.blob.data(), st.lens.data(), d->n_reads, (const uint8_t *)d_q[t], d->offsets, d->read_len, kbbq_engine_stream(e)))
                    return 1;
            }
            batch.pack_on_host = true;
            if (batch.fatal) return 1;
            if (!dev_out->drain()) return 1;
        } else {
            std::unique_ptr<Source> in = open_source(filename, is_bam, use_oq);
            size_t bi = 0;
            while (batch.fill(*in, groups, batch_reads, true, is_bam)) {
                newq.assign(batch.c.n_bases + 16, 0);
                if (resident.on) {
                    if (bi >= resident.dev.size() || resident.dev[bi].n_bases != batch.c.n_bases || resident.dev[bi].n_reads != batch.c.n_reads) {
                        std::cerr << put_now << " Error: the input changed between the passes." << std::endl;
                        return 1;
                    }
                    if (kbbq_recalibrate_batch_host(e, &resident.dev[bi], newq.data()) < 0) return fail_engine("recalibrating");
                    ++bi;
                } else if (kbbq_recalibrate_batch(e, &batch.c, newq.data()) < 0) {
                    return fail_engine("recalibrating");
                }
                for (size_t r = 0; r < batch.c.n_reads; ++r) {
                    const uint8_t *q = newq.data() + batch.off[r];
                    if (is_bam) {
                        if (emit_bam(batch.bam_recs[r], q) < 0) return 1;
                    } else {
                        const FastqRecord &f = batch.fq_recs[r];
                        if (!emit_fastq(f.name.data(), f.name.size(), f.comment.data(), f.comment.size(), f.seq.data(), f.seq.size(), q)) return 1;
                    }
                }
            }
            if (batch.fatal) return 1;
        }
        if (!out.close()) return 1;
        if (dev_out) {
            out_payload = dev_out->payload_bytes; out_compressed = dev_out->compressed_bytes;
            dev_out->kernel_ms(ms_format, ms_deflate, ms_gather);
        }
    }
    clock.mark("pass4+format+deflate+write");
    if (clock.on && dev_in.active) {
        double inf = 0, idx = 0;
        if (dev_in.bam) kbbq_bam_reader_kernel_ms(dev_in.bam, &inf, &idx); else kbbq_fastq_reader_kernel_ms(dev_in.reader, &inf, &idx);
        std::cerr << "[timing] " << (dev_in.bam ? "BAM" : "FASTQ") << " reader on the GPU (" << (dev_in.text_kept ? "one scan, the text kept in HBM: " : "both scans: ")
                  << (dev_in.text_kept ? std::to_string(dev_in.kept_bytes) + " bytes; " : std::string()) << "waiting for file reads " << dev_in.wait_s
                  << " s, device calls " << dev_in.device_s << " s, packing + batch arrays " << dev_in.batch_s << " s; kernels: inflate " << inf << " ms, index + pack " << idx << " ms" << std::endl;
    }
    if (clock.on && out_payload)
        std::cerr << "[timing] BGZF writer on the GPU: " << out_payload << " bytes -> " << out_compressed << " (ratio "
                  << (double)out_payload / (double)std::max<uint64_t>(1, out_compressed) << "); kernels: format " << ms_format
                  << " ms, deflate " << ms_deflate << " ms, gather " << ms_gather << " ms" << std::endl;
    // Everything is written and flushed.  Handing 200 GB of device memory back allocation by allocation takes 2.2 s at
    // BASELINE size (every hipFree waits for the device); the process ends here and the driver takes it all back at once.
    // KBBQ_RELEASE=1: the orderly way (leak checkers).
    if (!(getenv("KBBQ_RELEASE") && atoi(getenv("KBBQ_RELEASE")))) {
        clock.mark("end");
        clock.report();
        fflush(stdout);
        fflush(stderr);
        exit(0);      // (handlers registered with atexit still run: a profiler's, the runtime's)
    }
    resident.drop();
    kbbq_engine_destroy(e);
    clock.mark("release");
    return 0;
}
