// kbbq_cli.cc -- the `kbbq` command line over the MI355X engine.
//
// Mirrors main() of the reference (kbbq.cc:81-459): same flags, same defaults, same stderr lines with
// the same "[%F %T %Z]" stamps, recalibrated FASTQ through BGZF on stdout.  Every pass re-opens the
// input like the reference does (kbbq.cc:232,258,336,365,456) and hands batches of reads to the engine
// through the C ABI (include/kbbq_engine.h); nothing is computed on the host except what the reference
// also computes there (coverage, alpha, thresholds, the delta-Q model -- inside the library).
//
// Differences, all deliberate:
//   * BAM/CRAM input is refused: it needs htslib, which this image does not have (SURVEY risk R1);
//   * one extra read-only scan of the input sizes the histograms (read groups, longest read) before
//     the engine is created; the reference grows its tables on the fly;
//   * the sampler seed can be fixed with KBBQ_SEED=<u32> (the reference always draws it from time+pid,
//     kbbq.cc:268-270, so two of its own runs differ: SURVEY hazard H1);
//   * --threads is accepted and ignored (it only sizes htslib's BGZF pool, kbbq.cc:159-168);
//   * where the reference prints an error and then crashes (missing --genomelen on FASTQ, kbbq.cc:218)
//     this exits 1.
#include <getopt.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iomanip>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/kbbq_engine.h"
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

enum class Format { fastq, bam, unknown };
static Format sniff(const std::string &path) {   // hts_detect_format, as far as this tool needs it
    gzFile f = path == "-" ? nullptr : gzopen(path.c_str(), "rb");
    if (!f) return Format::unknown;
    unsigned char b[4] = {0, 0, 0, 0};
    const int n = gzread(f, b, 4);
    gzclose(f);
    if (n >= 4 && b[0] == 'B' && b[1] == 'A' && b[2] == 'M' && b[3] == 1) return Format::bam;
    if (n >= 4 && b[0] == 'C' && b[1] == 'R' && b[2] == 'A' && b[3] == 'M') return Format::bam;
    if (n >= 1 && b[0] == '@') return Format::fastq;
    return Format::unknown;
}

// One batch of reads in the engine's layout, plus the records themselves for the output pass.
struct Batch {
    std::vector<FastqRecord> recs;
    std::vector<uint8_t> seq, qual, flags;
    std::vector<uint16_t> rg;
    std::vector<uint64_t> off, bases, nmask;
    kbbq_reads c;
    bool stop_at_empty = false;   // next_str() != "" loops end at the first empty read (kbbq.cc:234, htsiter.cc:95)

    // returns false when no read was collected
    bool fill(FastqReader &in, ReadGroups &groups, size_t max_reads, bool keep_records, bool &bad_name) {
        recs.clear(); seq.clear(); qual.clear(); flags.clear(); rg.clear();
        off.assign(1, 0);
        FastqRecord r;
        while (rg.size() < max_reads) {
            const int rc = in.next(r);
            if (rc < 0) break;                       // -1 end of file; < -1 error: the reference's loops also just end
            if (stop_at_empty && r.seq.empty()) break;
            std::string group, first;
            bool second = false;
            if (!parse_read_name(r.name, group, second, first)) { bad_name = true; return false; }
            seq.insert(seq.end(), r.seq.begin(), r.seq.end());
            for (char ch : r.qual) qual.push_back((uint8_t)(ch - 33));       // readutils.cc:70-71
            off.push_back(seq.size());
            flags.push_back(second ? 1 : 0);
            rg.push_back((uint16_t)groups.index_of(group));
            if (keep_records) recs.push_back(r);
        }
        if (rg.empty()) return false;
        bases.assign(seq.size() / 32 + 2, 0);
        nmask.assign(seq.size() / 64 + 2, 0);
        qual.resize(seq.size() + 16, 0);
        kbbq_pack_bases(seq.data(), seq.size(), bases.data(), nmask.data());
        memset(&c, 0, sizeof c);
        c.n_reads = rg.size();
        c.n_bases = seq.size();
        c.bases = bases.data();
        c.nmask = nmask.data();
        c.qual = qual.data();
        c.offsets = off.data();
        c.flags = flags.data();
        c.rg = rg.data();
        return true;
    }
};

static int fail_engine(const char *what) {
    std::cerr << put_now << " Error: " << what << ": " << kbbq_last_error() << std::endl;
    return 1;
}

// hidden helpers for the CPU test-suite: exercise the reader, the name rules and the BGZF writer
// without touching the GPU
static int io_test(int argc, char *argv[]) {
    const std::string what = argc > 2 ? argv[2] : "";
    if (what == "parse" && argc > 3) {
        FastqReader in(argv[3]);
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
    if (what == "bgzf") {   // stdin -> BGZF on stdout
        BgzfWriter out(stdout);
        std::vector<char> buf(1 << 16);
        size_t n;
        while ((n = fread(buf.data(), 1, buf.size(), stdin)) > 0)
            if (!out.write(buf.data(), n)) return 1;
        return out.close() ? 0 : 1;
    }
    return 2;
}

int main(int argc, char *argv[]) {
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
    (void)set_oq; (void)use_oq; (void)nthreads;
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
    if (fmt == Format::bam) {
        std::cerr << put_now << " Error: BAM/CRAM input needs htslib, which this build does not have; only FASTQ is supported."
                  << std::endl;
        return 1;
    }

    // one read-only scan: total length (the reference's coverage pass, kbbq.cc:229-250), read groups, longest read
    ReadGroups groups;
    uint64_t seqlen = 0, n_reads = 0;
    size_t longest = 0;
    {
        FastqReader in(filename);
        FastqRecord r;
        std::string group, first;
        bool second;
        while (in.next(r) >= 0 && !r.seq.empty()) {
            seqlen += r.seq.length();
            longest = std::max(longest, r.seq.length());
            ++n_reads;
            if (!parse_read_name(r.name, group, second, first)) {
                std::cerr << put_now << " Error: read name '" << r.name << "' is shorter than 2 characters before the first '_'." << std::endl;
                return 1;   // std::out_of_range in the reference (readutils.cc:90)
            }
            groups.index_of(group);
        }
    }
    if (longest > KBBQ_MAX_READ_LEN) {
        std::cerr << put_now << " Error: reads longer than " << KBBQ_MAX_READ_LEN << " bases are not supported by the GPU engine." << std::endl;
        return 1;
    }

    const bool fixed_mode = !fixedinput.empty();
    kbbq_engine *e = nullptr;
    const size_t batch_reads = 1 << 20;
    Batch batch;
    bool bad_name = false;

    if (!fixed_mode) {
        if (genomelen == 0) {
            std::cerr << put_now << " Error: --genomelen must be specified if input is not a bam." << std::endl;
            return 1;
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

        // pass 1, kbbq.cc:277-283
        {
            FastqReader in(filename);
            uint64_t ordinal = 0, nk = 0;
            batch.stop_at_empty = true;
            while (batch.fill(in, groups, batch_reads, false, bad_name)) {
                if (kbbq_sample_batch(e, &batch.c, ordinal) < 0) return fail_engine("sampling");
                if (kbbq_count_kmer_positions(e, &batch.c, &nk) < 0) return fail_engine("sampling");
                ordinal += nk;
            }
            batch.stop_at_empty = false;
            uint64_t inserted = 0;
            if (kbbq_sample_finish(e, &inserted) < 0) return fail_engine("sampling");
            std::cerr << put_now << " Sampled " << inserted << " valid kmers." << std::endl;
        }
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
        std::cerr << put_now << " Finding trusted kmers" << std::endl;
        {
            FastqReader in(filename);
            while (batch.fill(in, groups, batch_reads, false, bad_name))
                if (kbbq_trusted_batch(e, &batch.c, nullptr) < 0) return fail_engine("finding trusted kmers");
            if (kbbq_trusted_finish(e, nullptr) < 0) return fail_engine("finding trusted kmers");
        }
        // pass 3, kbbq.cc:363-366
        std::cerr << put_now << " Finding errors" << std::endl;
        {
            FastqReader in(filename);
            while (batch.fill(in, groups, batch_reads, false, bad_name))
                if (kbbq_errors_batch(e, &batch.c, nullptr) < 0) return fail_engine("finding errors");
        }
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
        FastqReader in(filename), fixed(fixedinput);
        Batch fb;
        ReadGroups fixed_groups;
        bool bad2 = false;
        while (batch.fill(in, groups, batch_reads, false, bad_name) && fb.fill(fixed, fixed_groups, batch.c.n_reads, false, bad2)) {
            std::vector<uint64_t> err(batch.c.n_bases / 64 + 2, 0);
            const size_t nr = std::min<size_t>(batch.c.n_reads, fb.c.n_reads);
            for (size_t r = 0; r < nr; ++r) {
                const uint64_t a = batch.off[r], len = batch.off[r + 1] - a, b = fb.off[r], flen = fb.off[r + 1] - b;
                for (uint64_t i = 0; i < len && i < flen; ++i)
                    if (batch.seq[a + i] != fb.seq[b + i]) err[(a + i) >> 6] |= 1ULL << ((a + i) & 63);
            }
            if (kbbq_tally_batch(e, &batch.c, err.data()) < 0) return fail_engine("tally");
        }
    }
    if (bad_name) {
        std::cerr << put_now << " Error: a read name is shorter than 2 characters before the first '_'." << std::endl;
        return 1;
    }

    // kbbq.cc:405-407
    std::cerr << put_now << " Training model" << std::endl;
    if (kbbq_train(e) < 0) return fail_engine("training");

    // pass 4, kbbq.cc:455-457: recalibrate_and_write(file, dqs, "-")
    std::cerr << put_now << " Recalibrating file" << std::endl;
    {
        FastqReader in(filename);
        BgzfWriter out(stdout);
        std::vector<uint8_t> newq;
        std::string qtext;
        while (batch.fill(in, groups, batch_reads, true, bad_name)) {
            newq.assign(batch.c.n_bases + 16, 0);
            if (kbbq_recalibrate_batch(e, &batch.c, newq.data()) < 0) return fail_engine("recalibrating");
            for (size_t r = 0; r < batch.recs.size(); ++r) {
                const uint64_t a = batch.off[r], len = batch.off[r + 1] - a;
                qtext.resize(len);
                for (uint64_t i = 0; i < len; ++i) qtext[i] = (char)(newq[a + i] + 33);   // htsiter.cc:61-65
                if (!write_fastq_record(out, batch.recs[r], qtext)) return 1;
            }
        }
        if (!out.close()) return 1;
    }
    kbbq_engine_destroy(e);
    return 0;
}
