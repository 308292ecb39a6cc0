// kbbq_oracle.cc -- TEST INFRASTRUCTURE ONLY.
//
// A single-threaded CPU restatement of the k-mer BQSR hot path of adamjorr/kbbq
// (reference @ v0).  It exists to CHECK the MI355X engine (kbbq_amd/), never to
// be shipped or measured as the product: only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load it.
//
// PARITY STATUS: the reference's path sources #include <htslib/*.h>, which this
// image lacks, so the reference itself cannot be built here (no stand-in headers
// are written).  The restatement is pinned by
//   * the known-answer values recorded in SURVEY.md section 8c (outputs of the
//     reference captured by the survey), checked in tests/test_oracle_kat.py;
//   * oracle/_ref (the reference's own minionrng + bloom_filter.hpp, which DO
//     compile stand-alone) for the RNG, hash_ap, salts and filter sizing, and
//     real libstdc++-11 std::shuffle / uniform_int_distribution /
//     bernoulli_distribution for the restated Lemire/shuffle/draw rules.
// The read-level functions (infer_read_errors, get_errors, ...) have no
// reference fixture at all: for those the oracle is "parity unpinned" beyond a
// line-by-line reading of the cited sources.
//
// Every function cites the reference file:line it follows (paths relative to
// /root/reference).  Bases are handled as codes 0..3 = A,C,G,T and 4 = anything
// else (htslib seq_nt16_int[seq_nt16_table[ch]]), which is how every reference
// function on the path consumes them; the single char-level behaviour this
// loses (a lower-case base never compares equal to the upper-case candidates in
// find_longest_fix/adjust_right_anchor/correct_one) is documented in DESIGN.md.

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <utility>
#include <vector>

namespace {

// Branch counters: tests use them to prove that the seeded inputs of the parity suite reach the
// rare paths of get_errors (they do not influence any result).
enum {
    C_GET_ERRORS, C_NO_ANCHOR, C_CORRECT_ONE_FIXED, C_EARLY_PATCH_RETURN, C_ADJUST, C_ADJUST_FIRST_TRY,
    C_ADJUST_MOVED, C_ADJUST_STAYED, C_FIX_CALLS, C_EXTENSION_STEPS, C_TIE_STOP, C_TIE_CONTINUE, C_UNFIXABLE,
    C_PREFIX_RECURSION, C_SUFFIX_RECURSION, C_VETO, C_OVERCORRECTED, C_UNFLAG_BACKJUMP, C_FIX_ON_N, C_LEFT_FIX,
    C_RIGHT_FIX, C_OFFCASE_FIX_CAND, C_OFFCASE_FIX_WON, C_OFFCASE_ADJUST_CAND, C_OFFCASE_ADJUST_MULTIPLE, C_COUNT
};
uint64_t g_count[C_COUNT];
const char *const g_count_names[C_COUNT] = {
    "get_errors", "no_anchor", "correct_one_fixed", "early_patch_return", "adjust", "adjust_first_try",
    "adjust_moved", "adjust_stayed", "fix_calls", "extension_steps", "tie_stop", "tie_continue", "unfixable",
    "prefix_recursion", "suffix_recursion", "veto", "overcorrected", "unflag_backjump", "fix_on_n", "left_fix",
    "right_fix", "offcase_fix_candidate", "offcase_fix_won", "offcase_adjust_candidate", "offcase_adjust_multiple"};

constexpr size_t NPOS = static_cast<size_t>(-1);
constexpr int MAXQ = 93;            // covariateutils.hh:3  KBBQ_MAXQ: the largest quality the model proposes, and the output clamp
// Quality rows of the dense tables.  The reference's tables grow with the largest quality seen (`resize(q+1)`,
// covariateutils.cc:70,108,156) and a quality is a uint8_t (readutils.hh), so 256 rows hold whatever it can hold.
constexpr int NQ = 256;
constexpr int BAD_QUAL = 2;         // readutils.hh:16  INFER_ERROR_BAD_QUAL

// ---------------------------------------------------------------------------
// htslib lookup tables used by the path (hts.c: seq_nt16_table, seq_nt16_int).
// The reference reaches bases only through seq_nt16_int[seq_nt16_table[ch]].
// ---------------------------------------------------------------------------
struct Nt16 {
    uint8_t code[256];
    Nt16() {
        uint8_t t[256];
        memset(t, 15, sizeof t);
        t['='] = 0;
        t['0'] = 1; t['1'] = 2; t['2'] = 4; t['3'] = 8;
        const char *iupac = "=ACMGRSVTWYHKDBN";
        for (int i = 1; i < 16; ++i) {
            t[(unsigned char)iupac[i]] = (uint8_t)i;
            t[(unsigned char)(iupac[i] - 'A' + 'a')] = (uint8_t)i;
        }
        static const uint8_t to_int[16] = {4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4};
        for (int c = 0; c < 256; ++c) code[c] = to_int[t[c]];
    }
};
const Nt16 NT16;
inline uint8_t base_code(unsigned char ch) { return NT16.code[ch]; }

// ---------------------------------------------------------------------------
// xoshiro256** + minion seeding.  minion.hpp:72-143 (engine), :291-298
// (splitmix64), :320-335 (SeedSeq32::Generate), :355-375 (Random::Seed).
// ---------------------------------------------------------------------------
struct Xoshiro {
    uint64_t s[4];
    static uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
    uint64_t next() {
        const uint64_t out = rotl(s[1] * 5, 7) * 9;
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return out;
    }
    static uint64_t splitmix(uint64_t &st) {
        st += 0x9e3779b97f4a7c15ULL;
        uint64_t z = st;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    }
    // Random::Seed(uint32_t): SeedSeq32 holding the single value `seed`
    // hashes eight fixed 32-bit words; pairs of them are added to the fixed
    // initial state; 256 outputs are burnt.
    void seed32(uint32_t seed) {
        uint32_t v[8] = {0x9272B87Fu, 0xD9F64D09u, 0x6640D56Cu, 0x8CDA60ACu,
                         0xDEED25EDu, 0x8495FC63u, 0xAEA86A02u, 0x9F129AB9u};
        for (int i = 0; i < 8; ++i) {
            uint64_t st = v[i];
            uint64_t sum = splitmix(st);
            sum += splitmix(st) * seed;
            sum += splitmix(st) * 1u;
            v[i] = (uint32_t)(sum >> 32);
        }
        s[0] = 0x5FAF84EE2AA04CFFULL; s[1] = 0xB3A2EF3524D89987ULL;
        s[2] = 0x5A82B68EF098F79DULL; s[3] = 0x5D7AA03298486D6EULL;
        for (int j = 0; j < 4; ++j) s[j] += (uint64_t)v[2 * j] | ((uint64_t)v[2 * j + 1] << 32);
        if (!(s[0] | s[1] | s[2] | s[3])) s[1] = 0x1615CA18E55EE70CULL;
        for (int i = 0; i < 256; ++i) next();
    }
};

// libstdc++ 11 uniform_int_distribution on a 64-bit URBG: Lemire's method with
// a 128-bit product (bits/uniform_int_dist.h:246-268, 302-307).  Returns a
// value in [0, range).
uint64_t lemire_below(Xoshiro &g, uint64_t range) {
    unsigned __int128 prod = (unsigned __int128)g.next() * range;
    uint64_t low = (uint64_t)prod;
    if (low < range) {
        const uint64_t threshold = (0 - range) % range;
        while (low < threshold) {
            prod = (unsigned __int128)g.next() * range;
            low = (uint64_t)prod;
        }
    }
    return (uint64_t)(prod >> 64);
}

// std::bernoulli_distribution(p)(g) in libstdc++ 11 (bits/random.h:3635-3643
// over generate_canonical<double,53>, bits/random.tcc:3348-3380): one 64-bit
// output, converted to double (round to nearest), divided by 2^64, clamped
// below 1.  Used by KmerSubsampler::next, htsiter.cc:113-129.
inline bool bernoulli_draw(uint64_t u, double p) {
    double r = (double)u / 18446744073709551616.0;
    if (r >= 1.0) r = std::nextafter(1.0, 0.0);
    return r < p;
}

// ---------------------------------------------------------------------------
// bloom::Kmer, bloom.hh:339-379
// ---------------------------------------------------------------------------
struct Kmer {
    int k;
    uint64_t fwd, rev, mask;
    unsigned shift;
    size_t n;
    explicit Kmer(int k_)
        : k(k_), fwd(0), rev(0), mask(k_ < 32 ? ((1ULL << (2 * k_)) - 1) : ~0ULL),
          shift(2u * (unsigned)(k_ - 1)), n(0) {}
    size_t push(uint8_t c) {
        if (c < 4) {
            fwd = ((fwd << 2) | c) & mask;
            rev = (rev >> 2) | ((uint64_t)(3 - c) << shift);
            ++n;
        } else {
            clear();
        }
        return n;
    }
    void clear() { n = 0; fwd = rev = 0; }
    bool valid() const { return n >= (size_t)k; }
    uint64_t canon() const { return fwd < rev ? fwd : rev; }
};

// ---------------------------------------------------------------------------
// Bloom filter: bloom_filter.hpp:108-160 (sizing), :467-549 (salts), :551-608
// (hash_ap); bloom.hh:36-56 (blocked ctor), :99-105, :110-113, :189-231
// (pattern table), :250-292 (insert / contains), :318-330 (effective_fpp).
// ---------------------------------------------------------------------------
const uint32_t PREDEF_SALT[128] = {
    0xAAAAAAAA, 0x55555555, 0x33333333, 0xCCCCCCCC, 0x66666666, 0x99999999, 0xB5B5B5B5, 0x4B4B4B4B,
    0xAA55AA55, 0x55335533, 0x33CC33CC, 0xCC66CC66, 0x66996699, 0x99B599B5, 0xB54BB54B, 0x4BAA4BAA,
    0xAA33AA33, 0x55CC55CC, 0x33663366, 0xCC99CC99, 0x66B566B5, 0x994B994B, 0xB5AAB5AA, 0xAAAAAA33,
    0x555555CC, 0x33333366, 0xCCCCCC99, 0x666666B5, 0x9999994B, 0xB5B5B5AA, 0xFFFFFFFF, 0xFFFF0000,
    0xB823D5EB, 0xC1191CDF, 0xF623AEB3, 0xDB58499F, 0xC8D42E70, 0xB173F616, 0xA91A5967, 0xDA427D63,
    0xB1E8A2EA, 0xF6C0D155, 0x4909FEA3, 0xA68CC6A7, 0xC395E782, 0xA26057EB, 0x0CD5DA28, 0x467C5492,
    0xF15E6982, 0x61C6FAD3, 0x9615E352, 0x6E9E355A, 0x689B563E, 0x0C9831A8, 0x6753C18B, 0xA622689B,
    0x8CA63C47, 0x42CC2884, 0x8E89919B, 0x6EDBD7D3, 0x15B6796C, 0x1D6FDFE4, 0x63FF9092, 0xE7401432,
    0xEFFE9412, 0xAEAEDF79, 0x9F245A31, 0x83C136FC, 0xC3DA4A8C, 0xA5112C8C, 0x5271F491, 0x9A948DAB,
    0xCEE59A8D, 0xB5F525AB, 0x59D13217, 0x24E7C331, 0x697C2103, 0x84B0A460, 0x86156DA9, 0xAEF2AC68,
    0x23243DA5, 0x3F649643, 0x5FA495A8, 0x67710DF8, 0x9A6C499E, 0xDCFB0227, 0x46A43433, 0x1832B07A,
    0xC46AFF3C, 0xB9C8FFF0, 0xC9500467, 0x34431BDF, 0xB652432B, 0xE367F12B, 0x427F4C1B, 0x224C006E,
    0x2E7E5A89, 0x96F99AA5, 0x0BEB452A, 0x2FD87C39, 0x74B2E1FB, 0x222EFD24, 0xF357F60C, 0x440FCB1E,
    0x8BBE030F, 0x6704DC29, 0x1144D12F, 0x948B1355, 0x6D8FD7E9, 0x1C11A014, 0xADD1592F, 0xFB3C712E,
    0xFC77642F, 0xF9C4CE8C, 0x31312FB9, 0x08B0DD79, 0x318FA6E7, 0xC040D23D, 0xC0589AA7, 0x0CA5C075,
    0xF874B172, 0x0CF914D5, 0x784D3280, 0x4E8CFEBC, 0xC569F575, 0xCDB2A091, 0x2CC016B4, 0x5C5F4421};

// hash_ap on an 8-byte key: exactly one pass of the ">= 8 bytes" loop.
inline uint32_t hash_ap8(uint64_t key, uint32_t h) {
    const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
    h ^= (h << 7) ^ (lo * (h >> 3)) ^ (~((h << 11) + (hi ^ (h >> 5))));
    return h;
}

constexpr uint64_t BLOCK_BITS = 512;
constexpr uint64_t N_PATTERNS = 65536;

struct Filter {
    uint64_t projected = 0;
    double want_fpr = 0;
    unsigned nhash_opt = 0;       // compute_optimal_parameters: number_of_hashes
    uint64_t bits_opt = 0;        // compute_optimal_parameters: table_size
    unsigned nsalt = 0;
    uint64_t bits = 0;            // rounded up to whole 512-bit blocks
    uint64_t nblocks = 0;
    uint64_t random_seed = 0;
    std::vector<uint32_t> salt;
    std::vector<uint64_t> table;  // 8 words per block
    std::vector<uint64_t> pattern;  // 65536 x 8 words
    uint64_t inserted = 0;

    // bloom_parameters::compute_optimal_parameters, bloom_filter.hpp:108-160
    static void optimal(uint64_t n, double p, unsigned &nhash, uint64_t &tbits) {
        double min_m = std::numeric_limits<double>::infinity();
        double min_k = 0.0;
        double k = 1.0;
        while (k < 1000.0) {
            const double numerator = (-k * n);
            const double denominator = std::log(1.0 - std::pow(p, 1.0 / k));
            const double curr_m = numerator / denominator;
            if (curr_m < min_m) { min_m = curr_m; min_k = k; }
            k += 1.0;
        }
        nhash = (unsigned)min_k;
        tbits = (uint64_t)min_m;
        tbits += ((tbits % 8) != 0) ? (8 - (tbits % 8)) : 0;
        if (nhash < 1) nhash = 1;
        if (tbits < 1) tbits = 1;
    }

    // Which u64 word / bit of a 512-bit block a sampled bit number lands in.
    // bloom.hh:110-113 get_vector_unit: vec = (b/8)/32, unit = (b/8)%4, and the
    // bit inside the 64-bit unit is b%64 (bloom.hh:228) -- NOT a bijection;
    // restated as written.  A cell is four little-endian 64-bit units.
    static inline void bit_home(unsigned b, unsigned &word, unsigned &bit) {
        const unsigned vec = (b / 8) / 32;
        const unsigned unit = (b / 8) % 4;
        word = vec * 4 + unit;
        bit = b % 64;
    }

    void build(uint64_t n, double p, uint64_t seed, bool with_table) {
        projected = n;
        want_fpr = p;
        optimal(n, p, nhash_opt, bits_opt);
        random_seed = seed * 0xA5A5A5A5ULL + 1;                  // bloom.hh:39
        nsalt = std::max(nhash_opt, 2u);                         // bloom.hh:41
        bits = bits_opt;
        bits += (bits % BLOCK_BITS) ? BLOCK_BITS - (bits % BLOCK_BITS) : 0;  // bloom.hh:44
        nblocks = bits / BLOCK_BITS;
        // generate_unique_salt, bloom_filter.hpp:467-549 (salt_count_ <= 128 branch)
        salt.assign(PREDEF_SALT, PREDEF_SALT + nsalt);
        for (size_t i = 0; i < salt.size(); ++i)
            salt[i] = salt[i] * salt[(i + 3) % salt.size()] + (uint32_t)random_seed;
        if (with_table) table.assign(nblocks * 8, 0);
        inserted = 0;
        // pattern table, bloom.hh:189-231
        pattern.assign(N_PATTERNS * 8, 0);
        Xoshiro rng;
        rng.seed32((uint32_t)random_seed);                       // bloom.hh:203 (Seed(uint32_t))
        std::vector<size_t> slot(BLOCK_BITS);
        for (size_t i = 0; i < BLOCK_BITS; ++i) slot[i] = i;
        // std::shuffle, libstdc++ 11 bits/stl_algo.h:3731-3785: even length ->
        // one {0,1} draw for element 1, then two swap positions per draw.
        {
            size_t i = 1;
            std::swap(slot[i], slot[lemire_below(rng, 2)]);
            ++i;
            while (i != BLOCK_BITS) {
                const uint64_t r = i + 1;
                const uint64_t x = lemire_below(rng, r * (r + 1));
                std::swap(slot[i], slot[x / (r + 1)]);
                ++i;
                std::swap(slot[i], slot[x % (r + 1)]);
                ++i;
            }
        }
        for (uint64_t pnum = 0; pnum < N_PATTERNS; ++pnum) {
            for (unsigned j = 0; j < nsalt; ++j) {
                const size_t pick = (size_t)lemire_below(rng, BLOCK_BITS - j) + j;
                std::swap(slot[j], slot[pick]);
            }
            for (unsigned j = 0; j < nsalt; ++j) {
                unsigned w, b;
                bit_home((unsigned)slot[j], w, b);
                pattern[pnum * 8 + w] |= 1ULL << b;
            }
        }
    }

    inline uint64_t block_of(uint64_t key) const { return hash_ap8(key, salt[0]) % nblocks; }
    inline uint64_t pattern_of(uint64_t key) const { return hash_ap8(key, salt[1]) & (N_PATTERNS - 1); }

    void insert(uint64_t key) {                                  // bloom.hh:255-267
        uint64_t *b = &table[block_of(key) * 8];
        const uint64_t *p = &pattern[pattern_of(key) * 8];
        for (int i = 0; i < 8; ++i) b[i] |= p[i];
        ++inserted;
    }
    bool contains(uint64_t key) const {                          // bloom.hh:276-292
        const uint64_t *b = &table[block_of(key) * 8];
        const uint64_t *p = &pattern[pattern_of(key) * 8];
        for (int i = 0; i < 8; ++i)
            if ((b[i] & p[i]) != p[i]) return false;
        return true;
    }
    // Bloom::insert / Bloom::query, bloom.hh:395-398
    void insert(const Kmer &km) { if (km.valid()) insert(km.canon()); }
    bool query(const Kmer &km) const { return km.valid() && contains(km.canon()); }
};

// pattern_blocked_bf::effective_fpp, bloom.hh:318-330 (returns double).
double effective_fpp(uint64_t table_bits, uint64_t element_count, size_t nsalt) {
    // More elements than bits: the reference's integer division gives c = 0, lambda = inf and a loop that
    // never ends (bloom.hh:320-323).  Both sides of the parity pair report a saturated filter instead.
    if (element_count > table_bits) return 1.0;
    long double c = table_bits / element_count;                 // integer division
    long double lambda = BLOCK_BITS / c;
    long double fpp = 0;
    for (int i = 0; i < 3 * lambda; ++i) {
        long double p_block = std::pow(lambda, (long double)i) * std::exp(-lambda) / std::tgammal(i + 1);
        long double p_collision = 1.0l - std::pow(1.0l - 1.0l / (N_PATTERNS), (long double)i);
        long double fpr_inner = std::pow(1.0l - std::exp(-1.0l * nsalt * i / BLOCK_BITS), 1.0l * nsalt);
        fpr_inner = p_collision + (1.0l - p_collision) * fpr_inner;
        fpp += p_block * fpr_inner;
    }
    return (double)fpp;
}

// calculate_phit, bloom.cc:190-195.  `pow` there is the C double pow.
long double phit(double fprate, long double alpha) {
    long double fpr = fprate;
    double exponent = alpha < 0.1 ? 0.2 / alpha : 2;
    long double pa = 1 - ::pow((double)(1 - alpha), exponent);
    return pa + fpr - fpr * pa;
}

// covariateutils.hh:54-59
long double log_binom_pmf(unsigned long long k, unsigned long long n, long double p) {
    double coefficient = (::lgamma((double)(n + 1)) - (::lgamma((double)(k + 1)) + ::lgamma((double)(n - k + 1))));
    return (long double)coefficient + (long double)k * std::log(p) + (long double)(n - k) * std::log1p(-p);
}
// covariateutils.hh:61-68
std::vector<long double> log_binom_cdf(unsigned long long k, long double p) {
    std::vector<long double> ret(k + 1);
    ret[0] = log_binom_pmf(0, k, p);
    for (unsigned long long i = 1; i <= k; ++i)
        ret[i] = std::log(std::exp(ret[i - 1]) + std::exp(log_binom_pmf(i, k, p)));
    return ret;
}
// covariateutils.hh:71-85
std::vector<int> thresholds_for(unsigned long long k, long double p) {
    const long double quartile = .995l;
    std::vector<int> thr(k + 1, 0);
    for (unsigned long long i = 1; i <= k; ++i) {
        std::vector<long double> cdf = log_binom_cdf(i, p);
        for (size_t j = 0; j < cdf.size(); ++j) {
            if (cdf[j] >= std::log(quartile)) { thr[i] = (int)j; break; }
        }
    }
    return thr;
}

// recalibrateutils.hh:36-37
inline long double q_to_p(int q) { return std::pow(10.0l, -((long double)q / 10.0l)); }
inline int p_to_q(long double p, int maxscore = 42) { return p > 0 ? (int)(-10 * std::log10(p)) : maxscore; }

// NormalPrior::get_normal_prior, covariateutils.cc:7-19
long double normal_prior(size_t j) {
    static std::vector<long double> cache;
    if (j >= cache.size()) {
        for (size_t i = cache.size(); i < j + 1; ++i) {
            errno = 0;
            cache.push_back(std::log(.9l * std::exp(-(std::pow(((long double)i / .5l), 2.0l)) / 2.0l)));
            if (errno != 0) cache[i] = std::numeric_limits<long double>::lowest();
        }
    }
    return cache[j];
}

// The inner argmax shared by the four delta_q functions,
// covariateutils.cc:44-63, 78-100, 118-145, 166-191.
int map_q_minus_prior(unsigned long long err, unsigned long long tot, int prior) {
    int map_q = 0;
    long double best = std::numeric_limits<long double>::lowest();
    for (int possible = 0; possible < MAXQ + 1; ++possible) {
        int diff = std::abs(prior - possible);
        long double prior_prob = normal_prior((size_t)diff);
        long double p = q_to_p(possible);
        long double loglike = log_binom_pmf(err + 1, tot + 2, p);
        long double posterior = prior_prob + loglike;
        if (posterior > best) { map_q = possible; best = posterior; }
    }
    return map_q - prior;
}

// ---------------------------------------------------------------------------
// Read records and per-read kernels.
// ---------------------------------------------------------------------------
typedef std::vector<uint8_t> Codes;

struct Read {
    Codes seq;                  // base codes 0..4
    // 1 where the raw character is an ACGT base but not the upper-case letter itself ('a', 'c', 'g', 't', and the
    // digits '0'..'3' that seq_nt16_table also maps to bases).  K-mers, covariates and the apply step fold such
    // characters, but three loops of the reference compare the RAW character with the candidates 'A','C','G','T'
    // (bloom.cc:142 `c == unfixed_char`, bloom.cc:218,249 `seq[modified_idx] == c`, readutils.cc:202
    // `this->seq[i] == c`), so there the candidate that equals an off-case base is NOT skipped.  A fix writes the
    // upper-case letter (readutils.cc:327, :232), which clears the flag.  Only this->seq carries raw characters:
    // the left-hand walk runs on `revcomped`, which seq_nt16_str makes upper-case (readutils.cc:351-353).
    std::vector<uint8_t> offcase;
    std::vector<uint8_t> qual;
    std::vector<uint8_t> err;   // 0/1 per base (CReadData::errors)
    int rg = 0;
    bool second = false;
    Read sub(size_t pos, size_t count) const {      // CReadData::substr, readutils.cc:597-608
        Read r;
        r.rg = rg; r.second = second;
        size_t end = (count == NPOS || pos + count > seq.size()) ? seq.size() : pos + count;
        r.seq.assign(seq.begin() + pos, seq.begin() + end);
        r.offcase.assign(offcase.begin() + pos, offcase.begin() + end);
        r.qual.assign(qual.begin() + pos, qual.begin() + end);
        r.err.assign(err.begin() + pos, err.begin() + end);
        return r;
    }
};

// bloom.cc:28-67 + readutils.cc:173-193
void infer_read_errors(Read &rd, const Filter &sampled, const std::vector<int> &thr, int k) {
    const size_t len = rd.seq.size();
    rd.err.assign(len, 0);
    if (len < (size_t)k) return;        // reference: size_t underflow (UB); engine-defined: no k-mers, no flags
    Kmer km(k);
    std::vector<uint8_t> present(len - k + 1, 0);
    for (size_t i = 0; i < len; ++i) {
        km.push(rd.seq[i]);
        if (i + 1 >= (size_t)k) present[i + 1 - k] = km.valid() ? sampled.query(km) : 0;
    }
    size_t in = 0, out = 0;
    for (size_t i = 0; i < len; ++i) {
        if (i < len - k + 1) { if (present[i]) ++in; else ++out; }
        if (i >= (size_t)k) { if (present[i - k]) --in; else --out; }
        const size_t possible = in + out;
        rd.err[i] = ((long long)in <= (long long)thr[possible] || rd.qual[i] <= BAD_QUAL) ? 1 : 0;
    }
}

// recalibrateutils.cc:15-40 (body of the per-read loop)
void trusted_inserts(Read &rd, Filter &trusted, const Filter &sampled, const std::vector<int> &thr, int k) {
    infer_read_errors(rd, sampled, thr, k);
    int n_ok = 0;
    Kmer km(k);
    for (int i = 0; i < (int)rd.seq.size(); ++i) {
        km.push(rd.seq[i]);
        if (!rd.err[i]) ++n_ok;
        if (i >= k && !rd.err[i - k]) --n_ok;
        if (km.valid() && n_ok == k) trusted.insert(km);
    }
}

// bloom.cc:83-94
// returns the code of the first trusted extension (0..3) or -1
int next_trusted_code(const Kmer &km, const Filter &t, bool reverse_order) {
    for (int j = 0; j < 4; ++j) {
        const int c = reverse_order ? 3 - j : j;
        Kmer extra = km;
        extra.push((uint8_t)c);
        if (t.query(extra)) return c;
    }
    return -1;
}

// bloom.cc:96-128
void longest_trusted_seq(const Codes &seq, const Filter &t, int k, size_t &a_start, size_t &a_end) {
    Kmer km(k);
    size_t best = 0, cur = 0;
    a_start = a_end = NPOS;
    for (size_t i = 0; i < seq.size(); ++i) {
        if (km.push(seq[i]) >= (size_t)k) {
            if (t.query(km)) {
                ++cur;
            } else {
                if (cur > best) { best = cur; a_end = i - 1; a_start = i + 1 - k - cur; }
                cur = 0;
            }
        } else if (cur != 0) {
            if (cur > best) { best = cur; a_end = i - 1; a_start = i + 1 - k - cur; }
            cur = 0;
        }
    }
    if (cur > best) { best = cur; a_end = NPOS; a_start = seq.size() + 1 - k - cur; }
}

// bloom.cc:130-188.  `sub` is the read from (error position - k + 1) to its end.
struct Fix { std::vector<uint8_t> best; size_t stop; bool multiple; };
Fix longest_fix(Codes sub, const Filter &t, int k, bool reverse_order, bool unfixed_offcase = false) {
    Kmer km(k);
    ++g_count[C_FIX_CALLS];
    Fix out; out.stop = 0; out.multiple = false;
    bool single = false;
    const uint8_t unfixed = sub[k - 1];
    for (int jj = 0; jj < 4; ++jj) {
        const uint8_t c = (uint8_t)(reverse_order ? 3 - jj : jj);
        if (c == unfixed && !unfixed_offcase) continue;      // bloom.cc:142 compares raw characters
        if (c == unfixed) ++g_count[C_OFFCASE_FIX_CAND];
        sub[k - 1] = c;
        km.clear();
        size_t i;
        const size_t i_stop = std::max((size_t)(2 * k - 1), sub.size());
        for (i = 0; i < i_stop; ++i) {
            int n;
            if (i < sub.size()) n = sub[i];
            else {
                n = next_trusted_code(km, t, reverse_order);
                if (n < 0) break;
                ++g_count[C_EXTENSION_STEPS];
            }
            km.push((uint8_t)n);
            if (i + 1 >= (size_t)k) {
                if (km.valid()) {
                    if (!t.query(km)) break;
                    if (i == (size_t)(k - 1)) { if (single) out.multiple = true; single = true; }
                } else {
                    break;
                }
            }
        }
        if (i > out.stop) { out.best.clear(); out.best.push_back(c); out.stop = i; }
        else if (i == out.stop) out.best.push_back(c);
    }
    if (unfixed_offcase && std::find(out.best.begin(), out.best.end(), unfixed) != out.best.end()) ++g_count[C_OFFCASE_FIX_WON];
    return out;
}

// bloom.cc:208-277
std::pair<size_t, bool> adjust_right_anchor(size_t anchor, const Codes &seq, const Filter &t, int k,
                                            const std::vector<uint8_t> *offcase = nullptr) {
    Kmer km(k);
    ++g_count[C_ADJUST];
    bool multiple = false;
    size_t mod = anchor + 1;
    for (size_t i = mod - k + 1; i < mod; ++i) km.push(seq[i]);
    for (uint8_t c = 0; c < 4; ++c) {
        if (seq[mod] == c && !(offcase && (*offcase)[mod])) continue;      // bloom.cc:218, raw characters
        Kmer nk = km;
        nk.push(c);
        for (size_t i = 0; i <= (size_t)k; ++i) {
            if (!t.query(nk)) break;
            if (mod + i == seq.size() - 1 || i == (size_t)k) { ++g_count[C_ADJUST_FIRST_TRY]; return std::make_pair(anchor, multiple); }
            nk.push(seq[mod + i + 1]);
        }
    }
    for (int i = k / 2 - 1; i >= 0 && anchor > (size_t)(i + k - 1); --i) {
        km.clear();
        mod = anchor - i;
        for (size_t j = mod - k + 1; j < mod; ++j) km.push(seq[j]);
        for (uint8_t c = 0; c < 4; ++c) {
            if (seq[mod] == c && !(offcase && (*offcase)[mod])) continue;      // bloom.cc:249
            if (seq[mod] == c) ++g_count[C_OFFCASE_ADJUST_CAND];
            Kmer nk = km;
            nk.push(c);
            if (nk.valid() && t.query(nk)) {
                if (seq[mod] == c) ++g_count[C_OFFCASE_ADJUST_MULTIPLE];
                multiple = true;
                for (size_t j = 0; nk.valid() && t.query(nk) && mod + 1 + j < seq.size() && j <= (size_t)(k / 2); ++j) {
                    nk.push(seq[mod + 1 + j]);
                    if (j == (size_t)(k / 2) && nk.valid() && t.query(nk)) { ++g_count[C_ADJUST_MOVED]; return std::make_pair(mod - 1, multiple); }
                }
            }
        }
    }
    ++g_count[C_ADJUST_STAYED];
    return std::make_pair(anchor, multiple);
}

// bloom.cc:279-305
int biggest_trusted_block(const Codes &seq, const Filter &t, int k, int current_len) {
    Kmer km(k);
    int in = 0, out = 0, len = 0;
    for (size_t i = 0; i < seq.size(); ++i) {
        km.push(seq[i]);
        if (i + 1 >= (size_t)k) {
            if (t.query(km)) {
                ++in;
            } else {
                if (in > len) len = in;
                in = 0;
                ++out;
                if (k - out < current_len) break;
            }
        }
    }
    if (in > len) len = in;
    return len;
}

// readutils.cc:195-235
size_t correct_one(Read &rd, const Filter &t, int k) {
    int best_len = 0;
    uint8_t best_base = 0;
    size_t best_pos = NPOS;
    const size_t len = rd.seq.size();
    for (size_t i = 0; i < len; ++i) {
        Codes work(rd.seq);
        for (uint8_t c = 0; c < 4; ++c) {
            if (rd.seq[i] == c && !rd.offcase[i]) continue;      // readutils.cc:202, raw characters
            work[i] = c;
            const size_t start = i > (size_t)(k - 1) ? i - k + 1 : 0;
            const size_t magic_start = i > (size_t)(k / 2 - 1) ? std::min(i - k / 2 + 1, len - k) : 0;
            Kmer magic(k);
            for (size_t j = magic_start; j <= magic_start + k - 1; ++j) magic.push(work[j]);
            if (t.query(magic)) {
                const size_t wend = std::min(len, start + (size_t)(2 * k - 1));
                Codes window(work.begin() + start, work.begin() + wend);
                const int n_in = biggest_trusted_block(window, t, k, best_len);
                if (n_in > best_len) {
                    best_base = c; best_pos = i; best_len = n_in;
                } else if (n_in == best_len && rd.qual[i] < rd.qual[best_pos]) {
                    best_base = c; best_pos = i;
                }
            }
        }
    }
    if (best_len > 0) { rd.seq[best_pos] = best_base; rd.offcase[best_pos] = 0; }
    return best_pos;
}

// readutils.cc:238-570.  Leaves rd.err filled; rd.seq is restored to the
// original EXCEPT on the two early returns (:254, :264), as in the reference.
void get_errors(Read &rd, const Filter &t, int k, int minqual, bool first_call) {
    const Codes original(rd.seq);
    const std::vector<uint8_t> original_case(rd.offcase);
    const size_t len = rd.seq.size();
    if (len < (size_t)k) return;    // reference: UB (size_t underflow in correct_one); engine-defined: untouched
    size_t bad_prefix = 0;
    size_t bad_suffix = NPOS;
    bool multiple = false;
    size_t anchor[2];
    ++g_count[C_GET_ERRORS];
    longest_trusted_seq(rd.seq, t, k, anchor[0], anchor[1]);
    bool patched = false;
    if (anchor[0] == NPOS) {
        multiple = true;
        ++g_count[C_NO_ANCHOR];
        const size_t idx = correct_one(rd, t, k);
        if (idx == NPOS) return;
        ++g_count[C_CORRECT_ONE_FIXED];
        patched = true;
        longest_trusted_seq(rd.seq, t, k, anchor[0], anchor[1]);
        rd.err[idx] = 1;
    }
    if (anchor[0] == 0 && anchor[1] == NPOS) { if (patched) ++g_count[C_EARLY_PATCH_RETURN]; return; }
    const size_t anchor_len = std::min(anchor[1], len - 1) + 1 - anchor[0];
    bool corrected = false;
    // right side, :271-346
    if (anchor[1] != NPOS) {
        if (anchor_len - k + 1 >= (size_t)k) {
            std::pair<size_t, bool> adj = adjust_right_anchor(anchor[1], rd.seq, t, k, &rd.offcase);
            anchor[1] = adj.first;
            multiple = multiple || adj.second;
        }
        for (size_t i = anchor[1] + 1; i < len;) {
            const size_t start = i - k + 1;
            Fix fx = longest_fix(Codes(rd.seq.begin() + start, rd.seq.end()), t, k, false, rd.offcase[i] != 0);
            multiple = multiple || fx.multiple;
            const size_t next_untrusted = start + fx.stop;
            if (next_untrusted > i) {
                if (fx.best.size() > 1) {
                    multiple = true;
                    const size_t largest = std::min(i + k - 1, len - 1);
                    if (next_untrusted <= largest || largest - i + 1 < (size_t)k) {
                        bad_suffix = i;
                        ++g_count[C_TIE_STOP];
                        break;
                    }
                    ++g_count[C_TIE_CONTINUE];
                } else {
                    if (rd.seq[i] > 3) ++g_count[C_FIX_ON_N];
                    ++g_count[C_RIGHT_FIX];
                    rd.seq[i] = fx.best[0];
                    rd.offcase[i] = 0;
                    rd.err[i] = 1;
                }
                corrected = true;
                i += fx.stop - k + 1;
            } else {
                bad_suffix = i;
                ++g_count[C_UNFIXABLE];
                break;
            }
        }
    }
    // left side, :348-422
    if (anchor[0] != 0) {
        Codes rc(len);
        for (size_t i = 0; i < len; ++i) {
            const uint8_t c = rd.seq[len - 1 - i];
            rc[i] = c < 4 ? (uint8_t)(3 - c) : (uint8_t)4;
        }
        if (anchor_len - k + 1 >= (size_t)k) {
            std::pair<size_t, bool> adj = adjust_right_anchor(len - anchor[0] - 1, rc, t, k);
            anchor[0] = len - adj.first - 1;
            multiple = multiple || adj.second;
        }
        for (int i = (int)anchor[0] - 1; i >= 0;) {
            const int j = (int)len - i - 1;
            const size_t start = (size_t)(j - k + 1);
            Fix fx = longest_fix(Codes(rc.begin() + start, rc.end()), t, k, true);
            multiple = multiple || fx.multiple;
            const size_t next_untrusted = start + fx.stop;
            if (next_untrusted > (size_t)j) {
                if (fx.best.size() > 1) {
                    multiple = true;
                    const size_t largest = std::min((size_t)j + (size_t)k - 1, len - 1);
                    if (next_untrusted <= largest || largest - j + 1 < (size_t)k) {
                        bad_prefix = (size_t)i;
                        ++g_count[C_TIE_STOP];
                        break;
                    }
                    ++g_count[C_TIE_CONTINUE];
                } else {
                    ++g_count[C_LEFT_FIX];
                    rc[j] = fx.best[0];
                    rd.err[i] = 1;
                }
                corrected = true;
                i -= (int)(next_untrusted - j);
            } else {
                bad_prefix = (size_t)i;
                ++g_count[C_UNFIXABLE];
                break;
            }
        }
    }
    // over-correction check, :429-546
    if (corrected) {
        bool adjust = true;
        Kmer km(k);
        size_t t_start = NPOS, t_end = NPOS;
        for (size_t i = 0; i < len && adjust; ++i) {
            km.push(original[i]);
            if (km.valid() && t.query(km)) {
                t_start = std::min(t_start, i - k + 1);
                t_end = i;
            } else {
                if (i > t_end) {
                    for (size_t j = t_start; j <= t_end; ++j) {
                        if (rd.err[j]) { adjust = false; ++g_count[C_VETO]; break; }
                    }
                    t_start = NPOS;
                    t_end = NPOS;
                }
            }
        }
        adjust = adjust && !multiple;
        const int ocwindow = 20;
        const int base_threshold = 4;
        int threshold = base_threshold;
        double occount = 0;
        std::vector<int> over;
        for (int i = 0; i < (int)len; ++i) {
            if (rd.err[i] && original[i] < 4) {
                if (rd.qual[i] <= minqual) occount += 0.5; else ++occount;
            }
            if (i >= ocwindow && rd.err[i - ocwindow] && original[i - ocwindow] < 4) {
                if (rd.qual[i - ocwindow] <= minqual) occount -= 0.5; else --occount;
            }
            threshold = (adjust && i >= ocwindow && (size_t)(i + ocwindow - 1) < len) ? base_threshold + 1 : base_threshold;
            if (occount > threshold && rd.err[i]) { over.push_back(i); ++g_count[C_OVERCORRECTED]; }
        }
        for (size_t oi = 0; oi < over.size(); ++oi) {
            const int oc = over[oi];
            if (rd.err[oc]) {
                int start = oc - k + 1;
                start = start >= 0 ? start : 0;
                int end = oc + k;
                end = (size_t)end < len ? end : (int)len;
                for (int i = start; i < end; ++i) {
                    if (rd.err[i]) {
                        rd.err[i] = 0;
                        if (i + k > end) end = (size_t)(i + k) < len ? i + k : (int)len;
                        if (i - k < start) {
                            ++g_count[C_UNFLAG_BACKJUMP];
                            i = i - k + 1 >= 0 ? i - k : -1;
                            start = i;
                        }
                    }
                }
            }
        }
    }
    // one level of recursion on a long untouched prefix / suffix, :547-563
    if (first_call && bad_prefix > 0 && (bad_prefix >= len / 2 || bad_prefix >= (size_t)(2 * k))) {
        Read sub = rd.sub(0, bad_prefix + 1);
        ++g_count[C_PREFIX_RECURSION];
        get_errors(sub, t, k, minqual, false);
        std::copy(sub.err.begin(), sub.err.end(), rd.err.begin());
    }
    if (first_call && bad_suffix < NPOS && bad_suffix < len &&
        (len - bad_suffix > len / 2 || len - bad_suffix > (size_t)(2 * k))) {
        Read sub = rd.sub(bad_suffix, NPOS);
        ++g_count[C_SUFFIX_RECURSION];
        get_errors(sub, t, k, minqual, false);
        std::copy(sub.err.begin(), sub.err.end(), rd.err.begin() + bad_suffix);
    }
    rd.seq = original;
    rd.offcase = original_case;
}

// ---------------------------------------------------------------------------
// Covariate tallies (dense restatement of covariateutils.cc:30-42, 65-76,
// 102-116, 147-164, 193-202) and the delta-Q model (:204-230).
// Dense layout (shared with the engine's C ABI, see include/kbbq_engine.h):
//   rg   [R][2]             q     [R][256][2]
//   cyc  [R][256][2][C][2]  dinuc [R][256][16][2]      (last index: 0=errors 1=total)
// ---------------------------------------------------------------------------
struct Cov {
    size_t R = 0, C = 0;
    std::vector<uint64_t> rg, q, cyc, di;
    void ensure(size_t r, size_t c) {
        if (r <= R && c <= C) return;
        Cov n;
        n.R = std::max(R, r); n.C = std::max(C, c);
        n.rg.assign(n.R * 2, 0); n.q.assign(n.R * NQ * 2, 0);
        n.cyc.assign(n.R * NQ * 2 * n.C * 2, 0); n.di.assign(n.R * NQ * 16 * 2, 0);
        for (size_t a = 0; a < R; ++a) {
            n.rg[a * 2] = rg[a * 2]; n.rg[a * 2 + 1] = rg[a * 2 + 1];
            for (int b = 0; b < NQ; ++b) {
                for (int e = 0; e < 2; ++e) n.q[(a * NQ + b) * 2 + e] = q[(a * NQ + b) * 2 + e];
                for (int s = 0; s < 2; ++s)
                    for (size_t c2 = 0; c2 < C; ++c2)
                        for (int e = 0; e < 2; ++e)
                            n.cyc[(((a * NQ + b) * 2 + s) * n.C + c2) * 2 + e] = cyc[(((a * NQ + b) * 2 + s) * C + c2) * 2 + e];
                for (int d = 0; d < 16; ++d)
                    for (int e = 0; e < 2; ++e) n.di[((a * NQ + b) * 16 + d) * 2 + e] = di[((a * NQ + b) * 16 + d) * 2 + e];
            }
        }
        *this = n;
    }
    void consume(const Read &rd, int minscore) {
        ensure((size_t)rd.rg + 1, rd.seq.size());
        const size_t r = (size_t)rd.rg;
        // Any quality 0..255 is counted (a BAM can hold values above KBBQ_MAXQ = 93, 0xFF for "missing"): the
        // reference's tables grow to q + 1 rows (covariateutils.cc:70,108,156) and model such a base like any other;
        // only the OUTPUT is clamped to 93 (readutils.cc:592-594).
        for (size_t i = 0; i < rd.seq.size(); ++i) {
            const size_t qq = rd.qual[i];
            rg[r * 2] += rd.err[i];
            rg[r * 2 + 1] += 1;
            q[(r * NQ + qq) * 2] += rd.err[i];
            q[(r * NQ + qq) * 2 + 1] += 1;
            const size_t ci = (((r * NQ + qq) * 2 + (rd.second ? 1 : 0)) * C + i) * 2;
            cyc[ci] += rd.err[i];
            cyc[ci + 1] += 1;
            if (i >= 1 && rd.seq[i] < 4 && rd.seq[i - 1] < 4 && rd.qual[i] >= minscore) {
                const size_t d = 15 & ((rd.seq[i - 1] << 2) | rd.seq[i]);
                di[((r * NQ + qq) * 16 + d) * 2] += rd.err[i];
                di[((r * NQ + qq) * 16 + d) * 2 + 1] += 1;
            }
        }
    }
};

struct Dq {
    size_t R = 0, C = 0;
    std::vector<int32_t> meanq, rgdq, qdq, cydq, didq;
};

// CCovariateData::get_dqs, covariateutils.cc:204-230.  The reference's tables
// grow on demand; the extents it would have reached are recovered from the
// dense counts (every increment adds 1 to a total), and only cells inside those
// extents are evaluated; everything else stays 0.
Dq train(const Cov &cv) {
    Dq d;
    d.R = cv.R; d.C = cv.C;
    d.meanq.assign(cv.R, 0); d.rgdq.assign(cv.R, 0);
    d.qdq.assign(cv.R * NQ, 0); d.cydq.assign(cv.R * NQ * 2 * cv.C, 0); d.didq.assign(cv.R * NQ * 16, 0);
    for (size_t r = 0; r < cv.R; ++r) {
        // qcov[rg].size(): highest observed q + 1
        int qn = 0;
        for (int q = 0; q < NQ; ++q) if (cv.q[(r * NQ + q) * 2 + 1]) qn = q + 1;
        long double expected = 0;
        for (int q = 0; q < qn; ++q) expected += (q_to_p(q) * cv.q[(r * NQ + q) * 2 + 1]);
        d.meanq[r] = p_to_q(expected / cv.rg[r * 2 + 1]);
        d.rgdq[r] = map_q_minus_prior(cv.rg[r * 2], cv.rg[r * 2 + 1], d.meanq[r]);
        const int rgprior = d.meanq[r] + d.rgdq[r];
        std::vector<int> qprior(qn);
        for (int q = 0; q < qn; ++q) {
            d.qdq[r * NQ + q] = map_q_minus_prior(cv.q[(r * NQ + q) * 2], cv.q[(r * NQ + q) * 2 + 1], rgprior);
            qprior[q] = rgprior + d.qdq[r * NQ + q];
        }
        // cycov[rg].size() == qcov[rg].size(); per (q, strand) the vector reaches the last observed cycle
        for (int q = 0; q < qn; ++q)
            for (int s = 0; s < 2; ++s) {
                size_t cn = 0;
                for (size_t c = 0; c < cv.C; ++c)
                    if (cv.cyc[(((r * NQ + q) * 2 + s) * cv.C + c) * 2 + 1]) cn = c + 1;
                for (size_t c = 0; c < cn; ++c) {
                    const size_t ci = (((r * NQ + q) * 2 + s) * cv.C + c);
                    d.cydq[ci] = map_q_minus_prior(cv.cyc[ci * 2], cv.cyc[ci * 2 + 1], qprior[q]);
                }
            }
        // dicov[rg].size(): highest q with a tallied dinucleotide + 1; a q below
        // that with none keeps an empty vector
        int dn = 0;
        for (int q = 0; q < NQ; ++q)
            for (int x = 0; x < 16; ++x) if (cv.di[((r * NQ + q) * 16 + x) * 2 + 1]) dn = q + 1;
        for (int q = 0; q < dn && q < qn; ++q) {
            bool any = false;
            for (int x = 0; x < 16; ++x) if (cv.di[((r * NQ + q) * 16 + x) * 2 + 1]) any = true;
            if (!any) continue;
            for (int x = 0; x < 16; ++x) {
                const size_t di = (r * NQ + q) * 16 + x;
                d.didq[di] = map_q_minus_prior(cv.di[di * 2], cv.di[di * 2 + 1], qprior[q]);
            }
        }
    }
    return d;
}

// CReadData::recalibrate, readutils.cc:572-595
void recalibrate(const Read &rd, const Dq &d, int minqual, uint8_t *out) {
    const size_t r = (size_t)rd.rg;
    for (size_t i = 0; i < rd.seq.size(); ++i) {
        const int q = rd.qual[i];
        int v = q;
        if (q >= minqual && r < d.R && i < d.C) {
            v = d.meanq[r] + d.rgdq[r] + d.qdq[r * NQ + q] + d.cydq[((r * NQ + q) * 2 + (rd.second ? 1 : 0)) * d.C + i];
            if (i > 0) {
                const int a = rd.seq[i - 1], b = rd.seq[i];
                if (a < 4 && b < 4) v += d.didq[(r * NQ + q) * 16 + (15 & ((a << 2) | b))];
            }
        }
        out[i] = (uint8_t)(v < 0 ? 0 : (MAXQ < v ? MAXQ : v));
    }
}

// ---------------------------------------------------------------------------
// Context driving the four passes the way kbbq.cc:258-457 does.
// ---------------------------------------------------------------------------
struct Ctx {
    int k;
    long double alpha;
    uint32_t seed;
    Filter sampled, trusted;
    Xoshiro draw_rng;
    uint64_t draws = 0;
    std::vector<int> thr;
    Cov cov;
    Dq dq;
};

struct Batch {
    uint64_t n_reads;
    const uint8_t *seq;      // ASCII, concatenated
    const uint8_t *qual;     // phred values (no +33), concatenated
    const uint64_t *off;     // n_reads + 1
    const int32_t *rg;       // per read (may be null -> 0)
    const uint8_t *second;   // per read (may be null -> 0)
};

Read make_read(const Batch &b, uint64_t r) {
    Read rd;
    const uint64_t s = b.off[r], e = b.off[r + 1];
    rd.seq.resize(e - s);
    rd.offcase.resize(e - s);
    for (uint64_t i = s; i < e; ++i) {
        const uint8_t c = base_code(b.seq[i]);
        rd.seq[i - s] = c;
        rd.offcase[i - s] = c < 4 && b.seq[i] != (uint8_t)"ACGT"[c];
    }
    rd.qual.assign(b.qual + s, b.qual + e);
    rd.err.assign(e - s, 0);
    rd.rg = b.rg ? b.rg[r] : 0;
    rd.second = b.second ? b.second[r] != 0 : false;
    return rd;
}

}  // namespace

// ===========================================================================
// C interface for ctypes (tests, smoke, bench cpu_baseline only).
// ===========================================================================
extern "C" {

// alpha_text: the reference keeps alpha in long double (kbbq.cc:83,251); it is
// narrowed to double only for the sampler (htsiter.hh:143)
void *ko_new(int k, const char *alpha_text, uint32_t seed, uint64_t approx_kmers, double fpr_sampled,
             double fpr_trusted, uint64_t bloom_seed) {
    Ctx *c = new Ctx;
    c->k = k;
    c->alpha = strtold(alpha_text, nullptr);
    c->seed = seed;
    c->sampled.build(approx_kmers, fpr_sampled, bloom_seed, true);
    c->trusted.build(approx_kmers, fpr_trusted, bloom_seed, true);
    c->draw_rng.seed32(seed);      // KmerSubsampler ctor, htsiter.hh:143
    return c;
}
void ko_free(void *h) { delete (Ctx *)h; }

// which: 0 sampled, 1 trusted
static Filter &filt(void *h, int which) { return which ? ((Ctx *)h)->trusted : ((Ctx *)h)->sampled; }
uint64_t ko_filter_bits(void *h, int which) { return filt(h, which).bits; }
uint64_t ko_filter_bits_unblocked(void *h, int which) { return filt(h, which).bits_opt; }
uint32_t ko_filter_nhash(void *h, int which) { return filt(h, which).nhash_opt; }
uint32_t ko_filter_nsalt(void *h, int which) { return filt(h, which).nsalt; }
uint64_t ko_filter_random_seed(void *h, int which) { return filt(h, which).random_seed; }
uint64_t ko_filter_inserted(void *h, int which) { return filt(h, which).inserted; }
const uint32_t *ko_filter_salts(void *h, int which) { return filt(h, which).salt.data(); }
const uint64_t *ko_filter_table(void *h, int which) { return filt(h, which).table.data(); }
const uint64_t *ko_filter_patterns(void *h, int which) { return filt(h, which).pattern.data(); }
void ko_filter_insert_key(void *h, int which, uint64_t key) { filt(h, which).insert(key); }
int ko_filter_contains_key(void *h, int which, uint64_t key) { return filt(h, which).contains(key) ? 1 : 0; }
uint64_t ko_filter_block_of(void *h, int which, uint64_t key) { return filt(h, which).block_of(key); }
uint64_t ko_filter_pattern_of(void *h, int which, uint64_t key) { return filt(h, which).pattern_of(key); }

// pass 1: subsample_kmers over KmerSubsampler (recalibrateutils.cc:7-13,
// htsiter.cc:89-129): one draw per k-mer position of every read with len >= k,
// in file order; sampled and valid -> insert.
void ko_sample(void *h, uint64_t n_reads, const uint8_t *seq, const uint64_t *off) {
    Ctx *c = (Ctx *)h;
    const int k = c->k;
    const double p = (double)c->alpha;
    for (uint64_t r = 0; r < n_reads; ++r) {
        Kmer km(k);
        const uint64_t s = off[r], e = off[r + 1];
        for (uint64_t i = s; i < e; ++i) {
            km.push(base_code(seq[i]));
            if (i - s + 1 >= (uint64_t)k) {
                const uint64_t u = c->draw_rng.next();
                ++c->draws;
                if (bernoulli_draw(u, p) && km.valid()) c->sampled.insert(km);
            }
        }
    }
}
uint64_t ko_draws(void *h) { return ((Ctx *)h)->draws; }

double ko_sampled_fpr(void *h) {
    Ctx *c = (Ctx *)h;
    return effective_fpp(c->sampled.bits, c->sampled.inserted, c->sampled.salt.size());
}
// kbbq.cc:304-313: fpr -> p -> thresholds; p is returned as a decimal string too
int ko_compute_thresholds(void *h, int32_t *out, char *p_text, size_t p_text_len) {
    Ctx *c = (Ctx *)h;
    const double fpr = ko_sampled_fpr(h);
    const long double p = phit(fpr, c->alpha);
    c->thr = thresholds_for((unsigned long long)c->k, p);
    for (int i = 0; i <= c->k; ++i) out[i] = c->thr[i];
    if (p_text) snprintf(p_text, p_text_len, "%.21Lg", p);
    return fpr > .15 ? 1 : 0;      // kbbq.cc:306 gate
}
void ko_set_thresholds(void *h, const int32_t *thr) {
    Ctx *c = (Ctx *)h;
    c->thr.assign(thr, thr + c->k + 1);
}

// pass 2: find_trusted_kmers, recalibrateutils.cc:15-40.  err_out (optional):
// infer_read_errors' flag per base.
void ko_trusted(void *h, uint64_t n_reads, const uint8_t *seq, const uint8_t *qual, const uint64_t *off,
                uint8_t *err_out) {
    Ctx *c = (Ctx *)h;
    Batch b = {n_reads, seq, qual, off, nullptr, nullptr};
    for (uint64_t r = 0; r < n_reads; ++r) {
        Read rd = make_read(b, r);
        trusted_inserts(rd, c->trusted, c->sampled, c->thr, c->k);
        if (err_out) std::copy(rd.err.begin(), rd.err.end(), err_out + off[r]);
    }
}

// pass 3: get_covariatedata, recalibrateutils.cc:42-89: get_errors(trusted,k,6)
// then consume_read.  err_out (optional): CReadData::errors per base.
void ko_errors(void *h, uint64_t n_reads, const uint8_t *seq, const uint8_t *qual, const uint64_t *off,
               const int32_t *rg, const uint8_t *second, uint8_t *err_out, int tally) {
    Ctx *c = (Ctx *)h;
    Batch b = {n_reads, seq, qual, off, rg, second};
    for (uint64_t r = 0; r < n_reads; ++r) {
        Read rd = make_read(b, r);
        get_errors(rd, c->trusted, c->k, 6, true);
        if (err_out) std::copy(rd.err.begin(), rd.err.end(), err_out + off[r]);
        if (tally) c->cov.consume(rd, 6);
    }
}

// --fixed mode tally (kbbq.cc:367-378): errors supplied by the caller
void ko_tally(void *h, uint64_t n_reads, const uint8_t *seq, const uint8_t *qual, const uint64_t *off,
              const int32_t *rg, const uint8_t *second, const uint8_t *err) {
    Ctx *c = (Ctx *)h;
    Batch b = {n_reads, seq, qual, off, rg, second};
    for (uint64_t r = 0; r < n_reads; ++r) {
        Read rd = make_read(b, r);
        std::copy(err + off[r], err + off[r + 1], rd.err.begin());
        c->cov.consume(rd, 6);
    }
}

uint64_t ko_cov_nrg(void *h) { return ((Ctx *)h)->cov.R; }
uint64_t ko_cov_ncycle(void *h) { return ((Ctx *)h)->cov.C; }
const uint64_t *ko_cov_rg(void *h) { return ((Ctx *)h)->cov.rg.data(); }
const uint64_t *ko_cov_q(void *h) { return ((Ctx *)h)->cov.q.data(); }
const uint64_t *ko_cov_cycle(void *h) { return ((Ctx *)h)->cov.cyc.data(); }
const uint64_t *ko_cov_dinuc(void *h) { return ((Ctx *)h)->cov.di.data(); }
// load dense counts (lets tests train on counts produced elsewhere)
void ko_cov_set(void *h, uint64_t R, uint64_t C, const uint64_t *rg, const uint64_t *q, const uint64_t *cyc,
                const uint64_t *di) {
    Cov &cv = ((Ctx *)h)->cov;
    cv = Cov();
    cv.ensure(R, C);
    std::copy(rg, rg + R * 2, cv.rg.begin());
    std::copy(q, q + R * NQ * 2, cv.q.begin());
    std::copy(cyc, cyc + R * NQ * 2 * C * 2, cv.cyc.begin());
    std::copy(di, di + R * NQ * 16 * 2, cv.di.begin());
}

void ko_train(void *h) { Ctx *c = (Ctx *)h; c->dq = train(c->cov); }
const int32_t *ko_dq_meanq(void *h) { return ((Ctx *)h)->dq.meanq.data(); }
const int32_t *ko_dq_rg(void *h) { return ((Ctx *)h)->dq.rgdq.data(); }
const int32_t *ko_dq_q(void *h) { return ((Ctx *)h)->dq.qdq.data(); }
const int32_t *ko_dq_cycle(void *h) { return ((Ctx *)h)->dq.cydq.data(); }
const int32_t *ko_dq_dinuc(void *h) { return ((Ctx *)h)->dq.didq.data(); }

// pass 4: recalibrate_and_write's compute, recalibrateutils.cc:91-105
void ko_recalibrate(void *h, uint64_t n_reads, const uint8_t *seq, const uint8_t *qual, const uint64_t *off,
                    const int32_t *rg, const uint8_t *second, uint8_t *qual_out) {
    Ctx *c = (Ctx *)h;
    Batch b = {n_reads, seq, qual, off, rg, second};
    for (uint64_t r = 0; r < n_reads; ++r) {
        Read rd = make_read(b, r);
        recalibrate(rd, c->dq, 6, qual_out + off[r]);
    }
}

// branch counters
int ko_counter_count(void) { return C_COUNT; }
const char *ko_counter_name(int i) { return i >= 0 && i < C_COUNT ? g_count_names[i] : ""; }
uint64_t ko_counter_value(int i) { return i >= 0 && i < C_COUNT ? g_count[i] : 0; }
void ko_counters_reset(void) { memset(g_count, 0, sizeof g_count); }

// hooks for the sharded (multi-process) protocol test: install exchanged state
void ko_filter_set_inserted(void *h, int which, uint64_t n) { filt(h, which).inserted = n; }
uint64_t *ko_filter_table_mut(void *h, int which) { return filt(h, which).table.data(); }
void ko_skip_draws(void *h, uint64_t n) {
    Ctx *c = (Ctx *)h;
    for (uint64_t i = 0; i < n; ++i) c->draw_rng.next();
    c->draws += n;
}
void ko_dq_set(void *h, uint64_t R, uint64_t C, const int32_t *meanq, const int32_t *rgdq, const int32_t *qdq,
               const int32_t *cydq, const int32_t *didq) {
    Dq &d = ((Ctx *)h)->dq;
    d.R = R; d.C = C;
    d.meanq.assign(meanq, meanq + R); d.rgdq.assign(rgdq, rgdq + R); d.qdq.assign(qdq, qdq + R * NQ);
    d.cydq.assign(cydq, cydq + R * NQ * 2 * C); d.didq.assign(didq, didq + R * NQ * 16);
}

// ------------------------------------------------------------- unit probes
void ko_optimal_parameters(uint64_t n, double p, uint32_t *nhash, uint64_t *bits) {
    unsigned nh; uint64_t tb;
    Filter::optimal(n, p, nh, tb);
    *nhash = nh; *bits = tb;
}
uint32_t ko_hash_ap8(uint64_t key, uint32_t salt) { return hash_ap8(key, salt); }
void ko_rng_outputs(uint32_t seed, uint64_t n, uint64_t *out) {
    Xoshiro g; g.seed32(seed);
    for (uint64_t i = 0; i < n; ++i) out[i] = g.next();
}
uint64_t ko_bernoulli_count(uint32_t seed, double p, uint64_t n) {
    Xoshiro g; g.seed32(seed);
    uint64_t hits = 0;
    for (uint64_t i = 0; i < n; ++i) hits += bernoulli_draw(g.next(), p) ? 1 : 0;
    return hits;
}
int ko_bernoulli_one(uint64_t u, double p) { return bernoulli_draw(u, p) ? 1 : 0; }
// canonical k-mer after pushing an ASCII string; returns validity, size via *n
int ko_kmer(int k, const char *s, uint64_t *canon, uint64_t *n) {
    Kmer km(k);
    for (const char *p = s; *p; ++p) km.push(base_code((unsigned char)*p));
    *canon = km.canon(); *n = km.n;
    return km.valid() ? 1 : 0;
}
void ko_thresholds(int k, const char *p_text, int32_t *out) {
    long double p = strtold(p_text, nullptr);
    std::vector<int> t = thresholds_for((unsigned long long)k, p);
    for (int i = 0; i <= k; ++i) out[i] = t[i];
}
void ko_effective_fpp_text(uint64_t bits, uint64_t count, uint32_t nsalt, char *buf, size_t n) {
    long double v = effective_fpp(bits, count, nsalt);
    snprintf(buf, n, "%.21Lg", v);
}
void ko_phit_text(uint64_t bits, uint64_t count, uint32_t nsalt, const char *alpha_text, char *buf, size_t n) {
    long double a = strtold(alpha_text, nullptr);
    snprintf(buf, n, "%.21Lg", phit(effective_fpp(bits, count, nsalt), a));
}
void ko_normal_prior_text(uint64_t j, char *buf, size_t n) { snprintf(buf, n, "%.12Lg", normal_prior(j)); }
void ko_log_binom_pmf_text(uint64_t k, uint64_t nn, const char *p_text, char *buf, size_t n) {
    snprintf(buf, n, "%.15Lg", log_binom_pmf(k, nn, strtold(p_text, nullptr)));
}
int ko_p_to_q(const char *p_text) { return p_to_q(strtold(p_text, nullptr)); }
int ko_sizeof_long_double(void) { return (int)sizeof(long double); }
uint8_t ko_base_code(uint8_t ch) { return base_code(ch); }
int ko_map_q_minus_prior(uint64_t err, uint64_t tot, int prior) { return map_q_minus_prior(err, tot, prior); }

}  // extern "C"
