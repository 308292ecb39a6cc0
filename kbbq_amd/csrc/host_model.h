// host_model.h -- host-side scalar parts of the k-mer BQSR path.
//
// Everything here is small, serial and (for the statistics) x87 long double in
// the reference, so it stays on the host: filter sizing / salts / pattern table
// (bloom_filter.hpp:108-160,467-549; bloom.hh:36-56,189-231), the false-positive
// estimate and thresholds between passes (bloom.hh:318-330, bloom.cc:190-195,
// covariateutils.hh:54-85) and the delta-Q model (covariateutils.cc:7-19,
// 44-63,78-100,118-145,166-191,204-230).  It also owns the xoshiro256**
// jump-ahead tables the sampling kernel needs (the reference consumes its draw
// stream serially, htsiter.cc:113-129).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace kbbq {

constexpr int kNQ = 256;     // quality rows of every table: a quality is a uint8_t and the reference's tables grow with the largest seen (covariateutils.cc:65-76,102-116)
constexpr uint64_t kBlockBits = 512;
constexpr uint64_t kNumPatterns = 65536;

struct FilterSpec {
    uint64_t projected = 0;
    double fpr = 0;
    uint32_t n_hash = 0;
    uint64_t bits_unblocked = 0;
    uint64_t bits = 0;
    uint64_t n_blocks = 0;
    uint64_t random_seed = 0;
    uint32_t n_salt = 0;
    std::vector<uint32_t> salt;
    std::vector<uint64_t> patterns;  // 65536 x 8 words
};

// The engine's 128-bit form of a 512-bit block or pattern (device_common.h: FiltDev).  Word w of the
// reference only ever uses the 16 bits whose index has bits 3-4 equal to w&3 (get_vector_unit,
// bloom.hh:110-113,228): bytes w&3 and 4+(w&3) of the word.
constexpr uint64_t kEngineBlockBytes = 16;
inline uint64_t expand_field(uint16_t f, int unit) {
    return ((uint64_t)(f & 0xFF) << (8 * unit)) | ((uint64_t)(f >> 8) << (32 + 8 * unit));
}
inline void expand_block(const uint64_t in[2], uint64_t out[8]) {
    for (int w = 0; w < 8; ++w) out[w] = expand_field((uint16_t)(in[w >> 2] >> (16 * (w & 3))), w & 3);
}
// false if the block has a bit the engine's form cannot hold (never the case for a table built from patterns)
inline bool squeeze_block(const uint64_t in[8], uint64_t out[2]) {
    out[0] = out[1] = 0;
    bool exact = true;
    for (int w = 0; w < 8; ++w) {
        const int unit = w & 3;
        const uint16_t f = (uint16_t)(((in[w] >> (8 * unit)) & 0xFF) | (((in[w] >> (32 + 8 * unit)) & 0xFF) << 8));
        exact = exact && expand_field(f, unit) == in[w];
        out[w >> 2] |= (uint64_t)f << (16 * unit);
    }
    return exact;
}

// false when the parameters are ones the reference rejects (bloom.cc:18-21)
bool make_filter_spec(uint64_t projected, double fpr, uint64_t seed, FilterSpec &out);

// xoshiro256** as seeded by minion::Random::Seed(uint32_t)
struct Xoshiro256 {
    uint64_t s[4];
    void seed32(uint32_t seed);
    uint64_t next();
};

// x^(2^b) mod P for b = 0..63, P the characteristic polynomial of the
// xoshiro256 state transition; 4 words each, bit i of word w = coefficient of
// x^(64 w + i).  State after n steps = sum over set coefficients c_i of
// (x^n mod P) of M^i s.
const uint64_t *xoshiro_jump_table();  // [64][4]
void xoshiro_state_at(uint32_t seed, uint64_t ordinal, uint64_t out[4]);

// Largest T such that the draw rule of std::bernoulli_distribution(p) on one
// 64-bit output u (libstdc++ 11: (double)u / 2^64, clamped below 1, < p) holds
// exactly for u < T.  *always is set when every u is accepted.
uint64_t bernoulli_threshold(double p, bool *always);

// kbbq.cc:304-313.  Returns fpr; p as text with 21 significant digits.
double sampled_fpr(uint64_t table_bits, uint64_t inserted, uint32_t n_salt);
std::vector<int32_t> thresholds_from_counts(int k, uint64_t table_bits, uint64_t inserted, uint32_t n_salt,
                                            const char *alpha_text, double *fpr_out, std::string *p_text);

// covariateutils::log_binom_cdf(k, p) (covariateutils.hh:61-68): the values main() logs (kbbq.cc:327-331)
std::vector<long double> log_binom_cdf_values(unsigned long long k, long double p);

struct DqTables {
    uint64_t n_rg = 0, n_cycle = 0;
    std::vector<int32_t> meanq, rgdq, qdq, cycledq, dinucdq;
};
// cycle: [n_rg][256][2][n_cycle][2], dinuc: [n_rg][256][16][2]; q and rg totals are
// the sums the reference accumulates separately (covariateutils.cc:30-76).
void derive_q_rg(uint64_t n_rg, uint64_t n_cycle, const uint64_t *cycle, std::vector<uint64_t> &q,
                 std::vector<uint64_t> &rg);
DqTables train_model(uint64_t n_rg, uint64_t n_cycle, const uint64_t *rg, const uint64_t *q, const uint64_t *cycle,
                     const uint64_t *dinuc);

}  // namespace kbbq
