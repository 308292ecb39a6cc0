// ref_probe.cc -- TEST INFRASTRUCTURE ONLY (built into oracle/_ref/, git-ignored).
//
// A thin C interface over the parts of the reference that compile stand-alone
// in this image, taken in place from /root/reference (nothing is copied):
//   include/minionrng/minion.hpp + src/minionrng/minion.cc   (xoshiro256**, seeding)
//   include/kbbq/bloom_filter.hpp                            (sizing, salts, hash_ap)
// Everything else on the path #includes <htslib/*.h>, absent here, and is
// therefore unbuildable (no stand-in headers are written).
//
// ref_pattern_table() is NOT reference code: it follows bloom.hh:189-231 but
// drives the REAL minion::Random through the REAL libstdc++ std::shuffle and
// std::uniform_int_distribution<>, so it pins the oracle's explicit
// restatement of those library algorithms (SURVEY.md hazard H3).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

#include "bloom_filter.hpp"
#include "minion.hpp"

namespace {
struct Probe : public bloom_filter {
    explicit Probe(const bloom_parameters &p) : bloom_filter(p) {}
    using bloom_filter::hash_ap;
    const std::vector<bloom_type> &salts() const { return salt_; }
    unsigned long long seed() const { return random_seed_; }
};
}  // namespace

extern "C" {

void ref_rng_outputs(uint32_t seed, uint64_t n, uint64_t *out) {
    minion::Random rng;
    rng.Seed(seed);
    for (uint64_t i = 0; i < n; ++i) out[i] = rng();
}

uint64_t ref_bernoulli_count(uint32_t seed, double p, uint64_t n) {
    minion::Random rng;
    rng.Seed(seed);
    std::bernoulli_distribution d(p);
    uint64_t hits = 0;
    for (uint64_t i = 0; i < n; ++i) hits += d(rng) ? 1 : 0;
    return hits;
}

// per-draw outcomes, so the oracle's rule is compared draw by draw
void ref_bernoulli_bits(uint32_t seed, double p, uint64_t n, uint8_t *out) {
    minion::Random rng;
    rng.Seed(seed);
    std::bernoulli_distribution d(p);
    for (uint64_t i = 0; i < n; ++i) out[i] = d(rng) ? 1 : 0;
}

int ref_optimal_parameters(uint64_t n, double p, uint64_t seed, uint32_t *nhash, uint64_t *bits) {
    bloom_parameters bp;
    bp.projected_element_count = n;
    bp.false_positive_probability = p;
    bp.random_seed = seed;
    if (!bp) return -1;
    bp.compute_optimal_parameters();
    *nhash = bp.optimal_parameters.number_of_hashes;
    *bits = bp.optimal_parameters.table_size;
    return 0;
}

// salts of a plain bloom_filter with the given number of hashes (>= 2, where the
// blocked filter's max(nhash, 2) rule does not change the count); tiny table
int ref_salts(uint32_t nhash, uint64_t seed, uint32_t *out, uint64_t *random_seed) {
    bloom_parameters bp;
    bp.random_seed = seed;
    bp.optimal_parameters.number_of_hashes = nhash;
    bp.optimal_parameters.table_size = 64;
    Probe f(bp);
    for (size_t i = 0; i < f.salts().size(); ++i) out[i] = f.salts()[i];
    *random_seed = f.seed();
    return (int)f.salts().size();
}

uint32_t ref_hash_ap(const unsigned char *key, uint64_t len, uint32_t salt) {
    bloom_parameters bp;
    bp.optimal_parameters.number_of_hashes = 2;
    bp.optimal_parameters.table_size = 64;
    Probe f(bp);
    return f.hash_ap(key, len, salt);
}

// bloom.hh:189-231 procedure over the real RNG and real libstdc++ algorithms.
// out: 65536 * 8 u64 words, zeroed here.
void ref_pattern_table(uint32_t rng_seed, uint32_t nsalt, uint64_t *out) {
    const size_t block_size = 512, num_patterns = 65536;
    memset(out, 0, num_patterns * 64);
    minion::Random rng;
    rng.Seed(rng_seed);
    std::uniform_int_distribution<> d(0, block_size - 1);
    std::vector<size_t> possible_bits(block_size);
    std::iota(possible_bits.begin(), possible_bits.end(), 0);
    std::shuffle(possible_bits.begin(), possible_bits.end(), rng);
    for (size_t i = 0; i < num_patterns; ++i) {
        for (int j = 0; j < (int)nsalt; ++j) {
            size_t pick = d(rng, std::uniform_int_distribution<>::param_type{j, (int)block_size - 1});
            std::swap(possible_bits[j], possible_bits[pick]);
        }
        for (size_t j = 0; j < nsalt; ++j) {
            const size_t b = possible_bits[j];
            const size_t vec = (b / 8) / 32, unit = (b / 8) % 4;
            out[i * 8 + vec * 4 + unit] |= 1ULL << (b % 64);
        }
    }
}

}  // extern "C"
