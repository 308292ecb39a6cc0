// ref_path_probe.cc -- TEST INFRASTRUCTURE.  Runs the REFERENCE's own read-level path on in-memory reads.
//
// Compiled only by `make -C oracle ref_path`, and only when the compiler finds a REAL htslib (<htslib/hts.h> and
// -lhts): the five path sources of the reference (bloom.cc, readutils.cc, covariateutils.cc, recalibrateutils.cc,
// htsiter.cc) and minion.cc are compiled IN PLACE from /root/reference together with this harness into
// oracle/_ref/libref_path.so.  No stand-in for any htslib header, table or function exists in this repository; on an
// image without htslib the target prints why it does nothing and tests/test_ref_parity_cpu.py skips.
//
// The harness feeds reads to the reference through its own record interface (htsiter::HTSFile, htsiter.hh:40-49) and
// follows main()'s sequence (kbbq.cc:258-457): Bloom ctors -> KmerSubsampler + subsample_kmers -> fprate,
// calculate_phit, calculate_thresholds -> find_trusted_kmers -> get_covariatedata -> get_dqs ->
// recalibrate_and_write.  Per-read observables (infer_read_errors flags, get_errors flags) are read off
// readutils::CReadData with the same calls the pass functions make (recalibrateutils.cc:22-24,50-52).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bloom.hh"
#include "covariateutils.hh"
#include "htsiter.hh"
#include "readutils.hh"
#include "recalibrateutils.hh"

namespace {

struct MemFile : public htsiter::HTSFile {
    uint64_t n_reads;
    const uint8_t *seq, *qual;
    const uint64_t *off;
    const int32_t *rg;
    const uint8_t *second;
    uint64_t cur = 0;          // index of the NEXT record
    bool have = false;
    std::string name_s, seq_s, qual_s;
    kseq::kseq_t rec;
    uint8_t *recal_out = nullptr;

    MemFile(uint64_t n, const uint8_t *s, const uint8_t *q, const uint64_t *o, const int32_t *g, const uint8_t *sec)
        : n_reads(n), seq(s), qual(q), off(o), rg(g), second(sec) {
        memset(&rec, 0, sizeof rec);
    }
    void rewind() { cur = 0; have = false; }
    uint64_t index() const { return cur - 1; }
    int next() override {
        if (cur >= n_reads) { have = false; return -1; }
        const uint64_t a = off[cur], b = off[cur + 1];
        name_s = "r" + std::to_string(cur);
        seq_s.assign(reinterpret_cast<const char *>(seq) + a, b - a);
        qual_s.resize(b - a);
        for (uint64_t i = a; i < b; ++i) qual_s[i - a] = (char)(qual[i] + 33);
        rec.name.s = &name_s[0]; rec.name.l = name_s.size(); rec.name.m = name_s.size() + 1;
        rec.seq.s = &seq_s[0]; rec.seq.l = seq_s.size(); rec.seq.m = seq_s.size() + 1;
        rec.qual.s = &qual_s[0]; rec.qual.l = qual_s.size(); rec.qual.m = qual_s.size() + 1;
        ++cur;
        have = true;
        return (int)(b - a);
    }
    std::string next_str() override { return next() >= 0 ? seq_s : std::string(""); }     // htsiter.cc: FastqFile::next_str
    readutils::CReadData get() override {
        const uint64_t i = index();
        return readutils::CReadData(&rec, "g" + std::to_string(rg ? rg[i] : 0), second ? (int)(second[i] & 1) : 0);
    }
    void recalibrate(const std::vector<uint8_t> &q) override {
        if (recal_out) memcpy(recal_out + off[index()], q.data(), q.size());
    }
    int open_out(std::string) override { return 0; }
    int write() override { return 0; }
};

}  // namespace

extern "C" {

// bits of the blocked filter the reference builds for (approx, fpr): sizes the table outputs of rp_run
uint64_t rp_filter_bits(uint64_t approx_kmers, double fpr) {
    bloom::Bloom b(approx_kmers, fpr);
    return b.bloom.size();
}

// 0 = ok, 1 = the reference would have stopped at its fpr gate (kbbq.cc:306-310; outputs up to there are valid)
int rp_run(int k, const char *alpha_text, uint64_t seed, uint64_t approx_kmers, double fpr_sampled, double fpr_trusted,
           uint64_t n_reads, const uint8_t *seq, const uint8_t *qual, const uint64_t *off, const int32_t *rg,
           const uint8_t *second, uint64_t *sampled_inserted, uint64_t *trusted_inserted, int32_t *thresholds,
           double *fpr_out, char *p_text, size_t p_text_len, uint8_t *infer_errors, uint8_t *errors, uint8_t *recal,
           uint64_t *sampled_table, uint64_t *trusted_table) {
    readutils::CReadData::rg_to_int.clear();
    readutils::CReadData::rg_to_pu.clear();
    const long double alpha = std::stold(alpha_text);      // kbbq.cc:122
    MemFile f(n_reads, seq, qual, off, rg, second);
    bloom::Bloom subsampled(approx_kmers, fpr_sampled);     // kbbq.cc:265-266
    bloom::Bloom trusted(approx_kmers, fpr_trusted);
    {
        htsiter::KmerSubsampler subsampler(&f, k, alpha, seed);   // kbbq.cc:277
        recalibrateutils::subsample_kmers(subsampler, subsampled);
    }
    *sampled_inserted = subsampled.inserted_elements();
    if (sampled_table) memcpy(sampled_table, subsampled.bloom.bit_table_.get(), subsampled.bloom.size() / 8);
    const long double fpr = subsampled.fprate();            // kbbq.cc:304
    *fpr_out = (double)fpr;
    const long double p = bloom::calculate_phit(subsampled, alpha);
    if (p_text && p_text_len) snprintf(p_text, p_text_len, "%.21Lg", p);
    const std::vector<int> thr = covariateutils::calculate_thresholds(k, p);
    for (size_t i = 0; i < thr.size(); ++i) thresholds[i] = thr[i];
    if (fpr > .15) return 1;
    // infer_read_errors flags, as find_trusted_kmers sees them (recalibrateutils.cc:22-24)
    if (infer_errors) {
        f.rewind();
        while (f.next() >= 0) {
            readutils::CReadData read = f.get();
            read.infer_read_errors(subsampled, thr, k);
            const uint64_t a = off[f.index()];
            for (size_t i = 0; i < read.errors.size(); ++i) infer_errors[a + i] = read.errors[i] ? 1 : 0;
        }
    }
    f.rewind();
    recalibrateutils::find_trusted_kmers(&f, trusted, subsampled, thr, k);     // kbbq.cc:337
    *trusted_inserted = trusted.inserted_elements();
    if (trusted_table) memcpy(trusted_table, trusted.bloom.bit_table_.get(), trusted.bloom.size() / 8);
    // get_errors flags, as get_covariatedata sees them (recalibrateutils.cc:50-52)
    if (errors) {
        f.rewind();
        while (f.next() >= 0) {
            readutils::CReadData read = f.get();
            read.get_errors(trusted, k, 6);
            const uint64_t a = off[f.index()];
            for (size_t i = 0; i < read.errors.size(); ++i) errors[a + i] = read.errors[i] ? 1 : 0;
        }
    }
    f.rewind();
    covariateutils::CCovariateData data = recalibrateutils::get_covariatedata(&f, trusted, k);   // kbbq.cc:407
    covariateutils::dq_t dqs = data.get_dqs();
    f.rewind();
    f.recal_out = recal;
    recalibrateutils::recalibrate_and_write(&f, dqs, "-");                      // kbbq.cc:457
    return 0;
}

}  // extern "C"
