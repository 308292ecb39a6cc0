// bam_io.h -- the BAM side of the host I/O, on plain zlib because htslib is
// not part of this image.  What the reference gets from htslib for this path, restated from the SAM/BAM
// specification (SAMv1 section 4) and from how the reference uses it:
//   * BamReader  : sam_open + sam_hdr_read + sam_read1 (htsiter.hh:60-68, htsiter.cc:5): header text,
//                  reference names/lengths (kbbq.cc:204-207 sums the lengths), one alignment block per call;
//   * BamRecord  : the accessors CReadData's BAM constructor uses (readutils.cc:13-61): query name, flag,
//                  4-bit bases, qualities, bam_aux_get / bam_aux2Z for RG:Z and OQ:Z, and bam_aux_update_str
//                  for --set-oq (htsiter.cc:13-26);
//   * BamWriter  : sam_hdr_write + sam_write1 (htsiter.cc:35-45) through BGZF.
// htslib's version is not pinned by the reference (only HTS_VERSION >= 101000, htsiter.hh:25), so parity for
// this file is promised on the DECOMPRESSED BAM stream and is "unpinned" (nothing in the reference tests it).
// CRAM is not supported.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <string>
#include <vector>

#include "fastq_io.h"

namespace kbbq {

struct BamHeader {
    std::string text;                                         // l_text bytes, verbatim
    std::vector<std::pair<std::string, uint32_t>> refs;       // name (without the NUL), l_ref
    uint64_t genome_length() const {                          // kbbq.cc:204-207
        uint64_t g = 0;
        for (auto &r : refs) g += r.second;
        return g;
    }
};

enum { BAM_AUX_OK = 0, BAM_AUX_MISSING = 1, BAM_AUX_CORRUPT = 2 };   // errno ENOENT / EINVAL of bam_aux_get

struct BamRecord {
    std::vector<uint8_t> data;   // one alignment block, without its block_size prefix

    uint32_t u32(size_t at) const { return (uint32_t)data[at] | (uint32_t)data[at + 1] << 8 | (uint32_t)data[at + 2] << 16 | (uint32_t)data[at + 3] << 24; }
    uint16_t u16(size_t at) const { return (uint16_t)(data[at] | data[at + 1] << 8); }
    uint32_t l_read_name() const { return data[8]; }
    uint32_t n_cigar() const { return u16(12); }
    uint16_t flag() const { return u16(14); }
    uint32_t l_seq() const { return u32(16); }
    bool reverse() const { return flag() & 16; }              // bam_is_rev
    bool second() const { return flag() & 128; }              // BAM_FREAD2, readutils.cc:59
    size_t name_at() const { return 32; }
    size_t seq_at() const { return 32 + l_read_name() + 4 * (size_t)n_cigar(); }
    size_t qual_at() const { return seq_at() + (l_seq() + 1) / 2; }
    size_t aux_at() const { return qual_at() + l_seq(); }
    bool well_formed() const { return data.size() >= 32 && aux_at() <= data.size() && l_read_name() >= 1; }
    std::string name() const { return std::string((const char *)data.data() + name_at()); }
    uint8_t *qual() { return data.data() + qual_at(); }
    const uint8_t *qual() const { return data.data() + qual_at(); }

    // bam_seq_str (readutils.hh:30-42): bases in sequencing orientation; on reverse-strand records every
    // code that is not A/C/G/T comes out as 'N', on forward ones as its "=ACMGRSVTWYHKDBN" letter
    void sequence(std::string &out) const;

    // bam_aux_get: offset of the tag's TYPE byte inside data, or 0; status says why not
    size_t aux_find(const char tag[2], int &status) const;
    // bam_aux2Z over bam_aux_get; false if the tag is missing, corrupt or not of type Z/H
    bool aux_string(const char tag[2], std::string &out, int &status) const;
    // bam_aux_update_str(r, tag, len + 1, text) of htslib >= 1.10: an existing Z tag is resized in place,
    // a missing one is appended; false (status = BAM_AUX_CORRUPT) if it exists with another type
    bool aux_update_string(const char tag[2], const std::string &text, int &status);
};

// What CReadData's BAM constructor takes from one record (readutils.cc:13-61): bases in sequencing orientation, the
// qualities (OQ:Z with use_oq) in the same orientation, the RG:Z tag, second-in-pair.  False with the text the command
// line prints (the reference's messages) when --use-oq finds no usable OQ tag or the record has no RG tag.
bool decode_bam_read(const BamRecord &b, bool use_oq, std::string &seq, std::vector<uint8_t> &qual, std::string &rg, bool &second,
                     std::string &err);

class BamReader {
public:
    explicit BamReader(const std::string &path, int threads = 1);
    ~BamReader();
    bool ok() const { return ok_; }
    const BamHeader &header() const { return header_; }
    int next(BamRecord &rec);   // >= 0 ok, -1 end of file, -2 truncated or malformed (sam_read1's convention)
    std::unique_ptr<ByteSource> release() { ok_ = false; return std::move(fh_); }   // the stream, positioned after the header

private:
    bool read_exact(void *dst, size_t n);
    std::unique_ptr<ByteSource> fh_;
    BamHeader header_;
    bool ok_ = false;
};

// The records of a BAM stream through the pool of fastq_io.h (ChunkPipeline): alignment blocks carry their length, so
// the reader thread cuts pieces exactly; the workers do decode_bam_read.  A truncated or malformed block ends the
// stream after the records before it, as sam_read1 < -1 ends the reference's loops.
class BamChunkParser : public ChunkPipeline {
public:
    BamChunkParser(const std::string &path, bool use_oq, int io_threads, int parse_threads, bool keep_records);
    ~BamChunkParser() override { stop(); }
    bool ok() const { return src_ != nullptr; }
    const BamHeader &header() const { return header_; }

private:
    void produce() override;
    void parse(Job &job) override;
    std::unique_ptr<ByteSource> src_;
    BamHeader header_;
    bool use_oq_, keep_records_;
};

class BamWriter {
public:
    explicit BamWriter(ByteSink &out) : out_(out) {}
    bool write_header(const BamHeader &h);
    bool write(const BamRecord &rec);

private:
    ByteSink &out_;
};

}  // namespace kbbq
