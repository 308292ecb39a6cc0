// fastq_io.h -- host I/O of the FASTQ path, on plain zlib because
// htslib is not part of this image:
//   * FastqReader  : what kseq_read over bgzf_read delivers to FastqFile (htsiter.hh:101-126,
//                    htsiter.cc:49-59): name (up to the first blank), comment (rest of the header
//                    line), sequence, quality; plain, gzip and BGZF input alike (BGZF is multi-member gzip);
//   * parse_read_name : the read-group / second-in-pair rules of CReadData's FASTQ constructor
//                    (readutils.cc:64-104), quirks included;
//   * BgzfWriter / write_fastq_record : FastqFile::write (htsiter.cc:75-86): "@name\nseq\n+comment\nqual\n"
//                    through BGZF blocks (bgzf_open(.., "w"), htsiter.cc:67-72), ending with the BGZF EOF
//                    block.  Parity is promised on the DECOMPRESSED bytes, not on the compressed ones.
#pragma once
#include <zlib.h>

#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace kbbq {

// Decompressed bytes of a plain, gzip or BGZF file.  BGZF (every BAM; bgzip-ed FASTQ) is a series of
// independent <= 64 KiB gzip members whose size is in the header, so with threads > 1 its blocks are inflated
// by a pool ahead of the consumer (bgzf_mt of htslib, which the reference gets from hts_set_thread_pool,
// htsiter.hh:64-66,110-112); anything else goes through zlib's gzread on the calling thread.
class ByteSource {
public:
    virtual ~ByteSource() {}
    virtual int read(void *dst, unsigned n) = 0;   // bytes delivered (short only at the end), 0 at EOF, < 0 on error
};
std::unique_ptr<ByteSource> open_bytes(const std::string &path, int threads);
// Another inflater for BGZF files (the command line installs the one on the GPU, kbbq_cli.cc: DeviceBgzfSource): called by
// open_bytes for a regular BGZF file; a null result falls back to the thread pool.
typedef std::unique_ptr<ByteSource> (*BgzfSourceFactory)(const std::string &path);
void set_bgzf_source_factory(BgzfSourceFactory f);

struct FastqRecord {
    std::string name, comment, seq, qual;
};

class FastqReader {
public:
    explicit FastqReader(const std::string &path, int threads = 1);
    ~FastqReader();
    bool ok() const { return fh_ != nullptr; }
    // >= 0: sequence length; -1: end of file; -2: truncated quality (kseq_read's return values)
    int next(FastqRecord &rec);

private:
    int getc_();
    bool getline_(std::string &out, bool append);   // without the line terminator; false at EOF with nothing read
    std::unique_ptr<ByteSource> fh_;
    std::vector<unsigned char> buf_;
    size_t pos_ = 0, end_ = 0;
    int last_char_ = 0;
    bool eof_ = false;
};

// ---- block-parallel parse of the input ---------------------------------------------------------------------------
// A reader thread cuts the decompressed bytes of the input into pieces at record starts, a pool of workers parses
// the pieces, the consumer gets them back in file order (ChunkPipeline).  A piece carries what a batch of the engine
// and the output pass need of its records.
struct ReadPiece {
    std::vector<uint8_t> seq, qual;        // bases as text, qualities as numbers (readutils.cc:13-104), concatenated
    std::vector<uint64_t> off;             // n + 1 offsets into seq / qual
    std::vector<uint8_t> second;           // per read: second in pair
    std::vector<std::string> rg_names;     // read groups in order of first appearance in this piece
    std::vector<uint32_t> rg;              // per read: index into rg_names
    std::string blob;                      // FASTQ: name, comment, sequence text of every record; BAM: the alignment blocks
    std::vector<uint32_t> lens;            // FASTQ: three lengths per record; BAM: one
    std::vector<uint64_t> blob_off;        // n + 1 offsets into blob
    size_t longest = 0;
    bool complex = false;                  // FASTQ only: not strictly four-line records -- start over with FastqReader
    bool end_of_stream = false;            // BAM: a truncated or malformed record follows the last one of this piece (sam_read1 < -1)
    long fatal_at = -1;                    // the record after the last one of this piece is an error the command line reports
    std::string fatal_name, fatal_msg;     // FASTQ: the read name (parse_read_name failed); BAM: the text for stderr
    size_t n() const { return off.empty() ? 0 : off.size() - 1; }
};
using FastqPiece = ReadPiece;

class ChunkPipeline {
public:
    virtual ~ChunkPipeline();              // (derived destructors call stop() first: the threads call their virtuals)
    // the next piece in file order; null at the end of the stream and after a complex, fatal or end_of_stream piece
    std::shared_ptr<ReadPiece> next();

protected:
    struct Job {
        std::shared_ptr<std::vector<unsigned char>> chunk;
        size_t begin = 0, end = 0;
        bool last = false;                 // the stream ends with this piece
        bool force_complex = false;
        std::shared_ptr<ReadPiece> piece;
        bool done = false;
    };
    explicit ChunkPipeline(int parse_threads) : parse_threads_(parse_threads < 1 ? 1 : parse_threads) {}
    void start();                          // at the end of the derived constructor
    void stop();
    bool stopping();
    void submit(std::shared_ptr<Job> job); // blocks while too many pieces are in flight
    // A chunk buffer of at least `bytes` (its size is what it was left at: the caller tracks how much it filled).  Buffers
    // come back when their last job is done and are handed out again -- a fresh 32 MB vector per chunk is zero-filled and
    // faulted in page by page, which cost the reader thread more than reading the bytes.
    std::shared_ptr<std::vector<unsigned char>> chunk_buffer(size_t bytes);
    virtual void produce() = 0;            // reader thread: cut the stream into jobs
    virtual void parse(Job &job) = 0;      // worker thread

private:
    void run_reader();
    void worker();
    struct BufferPool {
        std::mutex mu;
        std::vector<std::vector<unsigned char>> free;
    };
    std::shared_ptr<BufferPool> buffers_ = std::make_shared<BufferPool>();
    int parse_threads_;
    std::vector<std::thread> pool_;
    std::deque<std::shared_ptr<Job>> order_, todo_;      // guarded by mu_
    std::mutex mu_;
    std::condition_variable cv_todo_, cv_done_, cv_room_;
    size_t in_flight_ = 0, max_in_flight_ = 64;
    bool stop_ = false, finished_ = false, ended_ = false, started_ = false;
    std::thread reader_;
};

// The usual FASTQ file is a run of strictly four-line records ("@header\nSEQ\n+...\nQUAL\n", |SEQ| = |QUAL| > 0,
// no carriage returns, nothing between records).  For such a stream the work of FastqReader + parse_read_name is done
// by the pool.  A cut is a GUESS ("\n@" whose line after next starts with '+'); it is proved by the piece before it,
// whose strict parse -- started at a proven record start -- must end exactly there.  Anything that is not strictly of
// that shape (multi-line records, FASTA records, empty reads, '\r', blank lines, a truncated tail) marks the piece
// `complex`: the caller then starts over with FastqReader, which is the definition.
class FastqChunkParser : public ChunkPipeline {
public:
    FastqChunkParser(const std::string &path, int io_threads, int parse_threads, bool keep_records);
    ~FastqChunkParser() override { stop(); }
    bool ok() const { return src_ != nullptr; }

private:
    void produce() override;
    void parse(Job &job) override;
    std::unique_ptr<ByteSource> src_;
    bool keep_records_;
};

// The dense read-group index in order of first appearance (CReadData::rg_to_int, readutils.cc:10-11,100-103).
class ReadGroups {
public:
    int index_of(const std::string &rg) {
        auto it = map_.find(rg);
        if (it != map_.end()) return it->second;
        const int id = (int)map_.size();   // g++ evaluates size() before the insertion (SURVEY hazard H4)
        map_.emplace(rg, id);
        names_.push_back(rg);
        return id;
    }
    size_t size() const { return names_.size(); }
    const std::vector<std::string> &names() const { return names_; }

private:
    std::unordered_map<std::string, int> map_;
    std::vector<std::string> names_;
};

// readutils.cc:74-97 with rg = "", second = 2 (infer), namedelimiter = "_".  Returns false where the
// reference would throw (name shorter than two characters, readutils.cc:90).
bool parse_read_name(const std::string &fullname, std::string &rg, bool &second, std::string &first_name);

// Where output bytes go: a BGZF writer -- host zlib (BgzfWriter below) or the device encoder (kbbq_cli.cc:
// DeviceBgzfWriter over include/kbbq_bgzf.h).  The decompressed stream is the same through either.
class ByteSink {
public:
    virtual ~ByteSink() {}
    virtual bool write(const char *data, size_t n) = 0;
    virtual bool close() = 0;   // flushes and appends the 28-byte EOF block; idempotent
};

// BGZF output.  Blocks are cut at fixed input boundaries (0xff00 bytes, htslib's BGZF_BLOCK_SIZE), so the
// compressed stream does not depend on the number of threads: with threads > 1 the blocks are deflated by a
// small pool (what hts_set_thread_pool buys the reference, htsiter.cc:69) and written in order.
class BgzfWriter : public ByteSink {
public:
    explicit BgzfWriter(FILE *out, int threads = 1);
    ~BgzfWriter() override;
    bool write(const char *data, size_t n) override;
    bool close() override;

private:
    static constexpr size_t kBlock = 0xff00;   // BGZF_BLOCK_SIZE of htslib
    struct Job {
        std::vector<unsigned char> in, out;
        bool done = false, ok = true;
    };
    bool submit();                 // pending_ -> a job (or compress inline when there is no pool)
    bool drain(size_t keep);       // write finished jobs in order until at most `keep` are outstanding
    void worker();
    FILE *out_;
    std::vector<unsigned char> pending_;
    bool closed_ = false, failed_ = false;
    // pool
    std::vector<std::thread> pool_;
    std::deque<std::shared_ptr<Job>> order_;     // submission order (writer thread only)
    std::deque<std::shared_ptr<Job>> todo_;      // guarded by mu_
    std::mutex mu_;
    std::condition_variable cv_todo_, cv_done_;
    bool stop_ = false;
};

inline bool write_fastq_record(ByteSink &w, const FastqRecord &r, const std::string &qual) {
    std::string s;
    s.reserve(r.name.size() + r.seq.size() + r.comment.size() + qual.size() + 8);
    s += '@'; s += r.name; s += '\n'; s += r.seq; s += "\n+"; s += r.comment; s += '\n'; s += qual; s += '\n';
    return w.write(s.data(), s.size());
}

}  // namespace kbbq
