// host_model.cc -- see host_model.h.  Built with g++ (x87 long double, glibc
// libm), the same arithmetic the reference's host code runs on.
#include "host_model.h"

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <thread>

namespace kbbq {

// ------------------------------------------------------------------ RNG ----
namespace {
inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t splitmix64(uint64_t &st) {   // minion.hpp:291-298
    st += 0x9e3779b97f4a7c15ULL;
    uint64_t z = st;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
}  // namespace

uint64_t Xoshiro256::next() {   // minion.hpp:99-117
    const uint64_t r = rotl64(s[1] * 5, 7) * 9;
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return r;
}

void Xoshiro256::seed32(uint32_t seed) {   // minion.hpp:355-375 over :320-335 and :120-137
    static const uint32_t start[8] = {0x9272B87Fu, 0xD9F64D09u, 0x6640D56Cu, 0x8CDA60ACu,
                                      0xDEED25EDu, 0x8495FC63u, 0xAEA86A02u, 0x9F129AB9u};
    uint32_t hashed[8];
    for (int i = 0; i < 8; ++i) {
        uint64_t st = start[i];
        uint64_t acc = splitmix64(st);
        acc += splitmix64(st) * (uint64_t)seed;
        acc += splitmix64(st);
        hashed[i] = (uint32_t)(acc >> 32);
    }
    static const uint64_t mixed[4] = {0x5FAF84EE2AA04CFFULL, 0xB3A2EF3524D89987ULL, 0x5A82B68EF098F79DULL,
                                      0x5D7AA03298486D6EULL};
    for (int j = 0; j < 4; ++j) s[j] = mixed[j] + ((uint64_t)hashed[2 * j] | ((uint64_t)hashed[2 * j + 1] << 32));
    if ((s[0] | s[1] | s[2] | s[3]) == 0) s[1] = 0x1615CA18E55EE70CULL;
    for (int i = 0; i < 256; ++i) next();
}

// ---- jump-ahead: polynomials x^(2^b) mod P over GF(2) -----------------------
namespace {

struct Poly512 { uint64_t w[8]; };
inline bool pbit(const uint64_t *w, int i) { return (w[i >> 6] >> (i & 63)) & 1; }
inline void pflip(uint64_t *w, int i) { w[i >> 6] ^= 1ULL << (i & 63); }

// Berlekamp-Massey over GF(2) on bit 0 of s[0]: connection polynomial of the
// output-bit sequence, then reversed into the characteristic polynomial P
// (degree 256, returned without its leading x^256 term).
void characteristic_poly(uint64_t p_low[4]) {
    const int N = 1024;
    std::vector<uint8_t> seq(N);
    Xoshiro256 g;
    g.s[0] = 0x0123456789abcdefULL; g.s[1] = 0xfedcba9876543210ULL;
    g.s[2] = 0x0f1e2d3c4b5a6978ULL; g.s[3] = 0x8796a5b4c3d2e1f0ULL;
    for (int t = 0; t < N; ++t) { seq[t] = (uint8_t)(g.s[0] & 1); g.next(); }
    std::vector<uint8_t> C(N + 1, 0), B(N + 1, 0), T;
    C[0] = B[0] = 1;
    int L = 0, m = 1;
    for (int n = 0; n < N; ++n) {
        int d = seq[n];
        for (int i = 1; i <= L; ++i) d ^= C[i] & seq[n - i];
        if (d == 0) {
            ++m;
        } else if (2 * L <= n) {
            T = C;
            for (int i = 0; i + m <= N; ++i) C[i + m] ^= B[i];
            L = n + 1 - L;
            B = T;
            m = 1;
        } else {
            for (int i = 0; i + m <= N; ++i) C[i + m] ^= B[i];
            ++m;
        }
    }
    if (L != 256) { fprintf(stderr, "kbbq: xoshiro256 linear complexity %d != 256\n", L); abort(); }
    // sum_{i=0..L} C[i] s[n-i] = 0  <=>  P(x) = sum C[i] x^(L-i)
    memset(p_low, 0, 32);
    for (int i = 1; i <= L; ++i)
        if (C[i]) pflip(p_low, L - i);
}

// r = a*b mod P, all polynomials of degree < 256
void mulmod(const uint64_t a[4], const uint64_t b[4], const uint64_t p_low[4], uint64_t r[4]) {
    uint64_t prod[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 256; ++i) {
        if (!pbit(a, i)) continue;
        const int ws = i >> 6, bs = i & 63;
        for (int j = 0; j < 4; ++j) {
            prod[ws + j] ^= b[j] << bs;
            if (bs) prod[ws + j + 1] ^= b[j] >> (64 - bs);
        }
    }
    for (int i = 511; i >= 256; --i) {
        if (!pbit(prod, i)) continue;
        pflip(prod, i);
        const int sh = i - 256, ws = sh >> 6, bs = sh & 63;
        for (int j = 0; j < 4; ++j) {
            prod[ws + j] ^= p_low[j] << bs;
            if (bs) prod[ws + j + 1] ^= p_low[j] >> (64 - bs);
        }
    }
    memcpy(r, prod, 32);
}

uint64_t g_jump[64][4];
std::once_flag g_jump_once;

void build_jump_table() {
    uint64_t p_low[4];
    characteristic_poly(p_low);
    uint64_t cur[4] = {2, 0, 0, 0};  // x
    for (int b = 0; b < 64; ++b) {
        memcpy(g_jump[b], cur, 32);
        uint64_t sq[4];
        mulmod(cur, cur, p_low, sq);
        memcpy(cur, sq, 32);
    }
}

}  // namespace

const uint64_t *xoshiro_jump_table() {
    std::call_once(g_jump_once, build_jump_table);
    return &g_jump[0][0];
}

void xoshiro_state_at(uint32_t seed, uint64_t ordinal, uint64_t out[4]) {
    const uint64_t *tab = xoshiro_jump_table();
    Xoshiro256 g;
    g.seed32(seed);
    for (int b = 0; b < 64; ++b) {
        if (!((ordinal >> b) & 1)) continue;
        const uint64_t *poly = tab + 4 * b;
        uint64_t acc[4] = {0, 0, 0, 0};
        for (int i = 0; i < 256; ++i) {
            if ((poly[i >> 6] >> (i & 63)) & 1)
                for (int j = 0; j < 4; ++j) acc[j] ^= g.s[j];
            g.next();
        }
        memcpy(g.s, acc, 32);
    }
    memcpy(out, g.s, 32);
}

uint64_t bernoulli_threshold(double p, bool *always) {
    // std::bernoulli_distribution::operator(), libstdc++ 11 bits/random.h:3635-3643
    // over generate_canonical<double,53> with a 64-bit engine (random.tcc:3348-3380)
    auto accept = [p](uint64_t u) {
        double r = (double)u / 18446744073709551616.0;
        if (r >= 1.0) r = std::nextafter(1.0, 0.0);
        return r < p;
    };
    *always = false;
    if (!accept(0)) return 0;
    if (accept(~0ULL)) { *always = true; return ~0ULL; }
    uint64_t lo = 0, hi = ~0ULL;  // accept(lo), !accept(hi)
    while (hi - lo > 1) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (accept(mid)) lo = mid; else hi = mid;
    }
    return hi;
}

// --------------------------------------------------------------- filters ----
namespace {
const uint32_t kPredefSalt[128] = {
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

// one value in [0, n) the way libstdc++ 11's uniform_int_distribution draws it
// from a full-range 64-bit engine (bits/uniform_int_dist.h:246-268)
uint64_t draw_below(Xoshiro256 &g, uint64_t n) {
    unsigned __int128 m = (unsigned __int128)g.next() * n;
    if ((uint64_t)m < n) {
        const uint64_t floor_ = (0 - n) % n;
        while ((uint64_t)m < floor_) m = (unsigned __int128)g.next() * n;
    }
    return (uint64_t)(m >> 64);
}
}  // namespace

bool make_filter_spec(uint64_t projected, double fpr, uint64_t seed, FilterSpec &f) {
    // bloom_parameters::operator!, bloom_filter.hpp:60-71 (with the defaults of :47-55)
    if (projected == 0 || fpr < 0.0 || std::isinf(fpr) || seed == 0 || seed == ~0ULL) return false;
    f.projected = projected;
    f.fpr = fpr;
    // compute_optimal_parameters, bloom_filter.hpp:108-160
    double best_m = std::numeric_limits<double>::infinity(), best_k = 0.0;
    for (double k = 1.0; k < 1000.0; k += 1.0) {
        const double num = (-k * projected);
        const double den = std::log(1.0 - std::pow(fpr, 1.0 / k));
        const double m = num / den;
        if (m < best_m) { best_m = m; best_k = k; }
    }
    f.n_hash = (uint32_t)best_k;
    uint64_t tbits = (uint64_t)best_m;
    if (tbits % 8) tbits += 8 - tbits % 8;
    if (f.n_hash < 1) f.n_hash = 1;
    if (tbits < 1) tbits = 1;
    f.bits_unblocked = tbits;
    // blocked_bloom_filter ctor, bloom.hh:36-45
    f.random_seed = seed * 0xA5A5A5A5ULL + 1;
    f.n_salt = std::max(f.n_hash, 2u);
    if (f.n_salt > 128) return false;   // the rand()-based branch (bloom_filter.hpp:530-548) is never reached by kbbq
    f.bits = tbits % kBlockBits ? tbits + (kBlockBits - tbits % kBlockBits) : tbits;
    f.n_blocks = f.bits / kBlockBits;
    // generate_unique_salt, bloom_filter.hpp:512-528
    f.salt.assign(kPredefSalt, kPredefSalt + f.n_salt);
    for (size_t i = 0; i < f.salt.size(); ++i)
        f.salt[i] = f.salt[i] * f.salt[(i + 3) % f.salt.size()] + (uint32_t)f.random_seed;
    // pattern table, bloom.hh:189-231 (libstdc++ 11 std::shuffle: bits/stl_algo.h:3731-3785)
    f.patterns.assign(kNumPatterns * 8, 0);
    Xoshiro256 g;
    g.seed32((uint32_t)f.random_seed);
    uint16_t order[kBlockBits];
    for (unsigned i = 0; i < kBlockBits; ++i) order[i] = (uint16_t)i;
    std::swap(order[1], order[draw_below(g, 2)]);
    for (unsigned i = 2; i < kBlockBits; i += 2) {
        const uint64_t span = i + 1;
        const uint64_t both = draw_below(g, span * (span + 1));
        std::swap(order[i], order[both / (span + 1)]);
        std::swap(order[i + 1], order[both % (span + 1)]);
    }
    for (uint64_t pn = 0; pn < kNumPatterns; ++pn) {
        uint64_t *pat = &f.patterns[pn * 8];
        for (unsigned j = 0; j < f.n_salt; ++j) std::swap(order[j], order[j + draw_below(g, kBlockBits - j)]);
        for (unsigned j = 0; j < f.n_salt; ++j) {
            const unsigned b = order[j];
            // get_vector_unit (bloom.hh:110-113): 32-byte cell (b/8)/32, 64-bit unit (b/8)%4, bit b%64
            pat[((b >> 3) >> 5) * 4 + ((b >> 3) & 3)] |= 1ULL << (b & 63);
        }
    }
    return true;
}

// ------------------------------------------------- between-pass statistics ----
double sampled_fpr(uint64_t table_bits, uint64_t inserted, uint32_t n_salt) {
    // pattern_blocked_bf::effective_fpp, bloom.hh:318-330
    if (inserted == 0) return 0.0;   // the reference divides by zero here
    // more elements than bits: c = 0 and lambda = inf below, and the reference's loop never ends; the
    // filter is saturated, which the caller's 0.15 gate (kbbq.cc:306-310) turns into its usual error
    if (inserted > table_bits) return 1.0;
    const size_t nsalt = n_salt;
    long double c = table_bits / inserted;
    long double lambda = kBlockBits / c;
    long double fpp = 0;
    for (int i = 0; i < 3 * lambda; ++i) {
        long double p_block = std::pow(lambda, (long double)i) * std::exp(-lambda) / std::tgammal(i + 1);
        long double p_collision = 1.0l - std::pow(1.0l - 1.0l / (kNumPatterns), (long double)i);
        long double fpr_inner = std::pow(1.0l - std::exp(-1.0l * nsalt * i / kBlockBits), 1.0l * nsalt);
        fpr_inner = p_collision + (1.0l - p_collision) * fpr_inner;
        fpp += p_block * fpr_inner;
    }
    return (double)fpp;
}

namespace {
long double binom_logpmf(unsigned long long k, unsigned long long n, long double p) {   // covariateutils.hh:54-59
    const double comb = ::lgamma((double)(n + 1)) - (::lgamma((double)(k + 1)) + ::lgamma((double)(n - k + 1)));
    return (long double)comb + (long double)k * std::log(p) + (long double)(n - k) * std::log1p(-p);
}
}  // namespace

std::vector<long double> log_binom_cdf_values(unsigned long long k, long double p) {
    std::vector<long double> ret(k + 1);
    ret[0] = binom_logpmf(0, k, p);
    for (unsigned long long i = 1; i <= k; ++i) ret[i] = std::log(std::exp(ret[i - 1]) + std::exp(binom_logpmf(i, k, p)));
    return ret;
}

std::vector<int32_t> thresholds_from_counts(int k, uint64_t table_bits, uint64_t inserted, uint32_t n_salt,
                                            const char *alpha_text, double *fpr_out, std::string *p_text) {
    const long double alpha = strtold(alpha_text, nullptr);
    const double fprate = sampled_fpr(table_bits, inserted, n_salt);
    if (fpr_out) *fpr_out = fprate;
    // calculate_phit, bloom.cc:190-195 (its unqualified pow is the C double pow)
    const long double fpr = fprate;
    const double exponent = alpha < 0.1 ? 0.2 / alpha : 2;
    const long double pa = 1 - ::pow((double)(1 - alpha), exponent);
    const long double p = pa + fpr - fpr * pa;
    if (p_text) {
        char buf[64];
        snprintf(buf, sizeof buf, "%.21Lg", p);
        *p_text = buf;
    }
    // calculate_thresholds over log_binom_cdf, covariateutils.hh:61-85
    const long double cut = std::log(.995l);
    std::vector<int32_t> thr((size_t)k + 1, 0);
    for (int n = 1; n <= k; ++n) {
        long double run = binom_logpmf(0, n, p);
        int j = 0;
        while (!(run >= cut) && j < n) {
            ++j;
            run = std::log(std::exp(run) + std::exp(binom_logpmf(j, n, p)));
        }
        thr[n] = run >= cut ? j : 0;
    }
    return thr;
}

// ------------------------------------------------------------ delta-Q model ----
namespace {
long double normal_log_prior(size_t j) {   // NormalPrior::get_normal_prior, covariateutils.cc:7-19
    static std::vector<long double> memo;
    while (memo.size() <= j) {
        const size_t i = memo.size();
        errno = 0;
        long double v = std::log(.9l * std::exp(-(std::pow(((long double)i / .5l), 2.0l)) / 2.0l));
        if (errno != 0) v = std::numeric_limits<long double>::lowest();
        memo.push_back(v);
    }
    return memo[j];
}
inline long double q_to_p(int q) { return std::pow(10.0l, -((long double)q / 10.0l)); }   // recalibrateutils.hh:36
inline int p_to_q(long double p) { return p > 0 ? (int)(-10 * std::log10(p)) : 42; }         // recalibrateutils.hh:37

// log(p) and log1p(-p) of the 94 candidate qualities: the same long double values binom_logpmf would compute for
// every cell, computed once
struct CandidateLogs {
    long double logp[94], log1mp[94];
    CandidateLogs() {
        for (int c = 0; c <= 93; ++c) { const long double p = q_to_p(c); logp[c] = std::log(p); log1mp[c] = std::log1p(-p); }
    }
};
const CandidateLogs &candidate_logs() {
    static const CandidateLogs t;
    return t;
}

// the argmax shared by the four delta_q members (covariateutils.cc:44-63 etc.): the binomial coefficient of
// log_binom_pmf (covariateutils.hh:54-59) does not depend on the candidate and is taken out of the loop; every
// term keeps the value and the order of additions it has in binom_logpmf
int delta_for(unsigned long long err, unsigned long long tot, int prior) {
    const CandidateLogs &L = candidate_logs();
    const unsigned long long k = err + 1, n = tot + 2;
    const double comb = ::lgamma((double)(n + 1)) - (::lgamma((double)(k + 1)) + ::lgamma((double)(n - k + 1)));
    int arg = 0;
    long double top = std::numeric_limits<long double>::lowest();
    for (int cand = 0; cand <= 93; ++cand) {
        const long double pmf = (long double)comb + (long double)k * L.logp[cand] + (long double)(n - k) * L.log1mp[cand];
        const long double score = normal_log_prior((size_t)std::abs(prior - cand)) + pmf;
        if (score > top) { top = score; arg = cand; }
    }
    return arg - prior;
}
}  // namespace

void derive_q_rg(uint64_t n_rg, uint64_t n_cycle, const uint64_t *cycle, std::vector<uint64_t> &q,
                 std::vector<uint64_t> &rg) {
    q.assign(n_rg * kNQ * 2, 0);
    rg.assign(n_rg * 2, 0);
    for (uint64_t r = 0; r < n_rg; ++r)
        for (int qq = 0; qq < kNQ; ++qq) {
            uint64_t e = 0, t = 0;
            const uint64_t *row = cycle + ((r * kNQ + qq) * 2) * n_cycle * 2;
            for (uint64_t i = 0; i < 2 * n_cycle; ++i) { e += row[2 * i]; t += row[2 * i + 1]; }
            q[(r * kNQ + qq) * 2] = e;
            q[(r * kNQ + qq) * 2 + 1] = t;
            rg[r * 2] += e;
            rg[r * 2 + 1] += t;
        }
}

DqTables train_model(uint64_t n_rg, uint64_t n_cycle, const uint64_t *rg, const uint64_t *q, const uint64_t *cycle,
                     const uint64_t *dinuc) {
    // CCovariateData::get_dqs, covariateutils.cc:204-230.  The reference's
    // tables grow as reads are consumed; a cell is evaluated here exactly when
    // it would exist there (extent = last observed index + 1), others stay 0.
    DqTables d;
    d.n_rg = n_rg;
    d.n_cycle = n_cycle;
    d.meanq.assign(n_rg, 0);
    d.rgdq.assign(n_rg, 0);
    d.qdq.assign(n_rg * kNQ, 0);
    d.cycledq.assign(n_rg * kNQ * 2 * n_cycle, 0);
    d.dinucdq.assign(n_rg * kNQ * 16, 0);
    normal_log_prior(1024);     // fill the memo before the threads read it
    candidate_logs();
    auto one_group = [&](uint64_t r) {
        const uint64_t *qr = q + r * kNQ * 2;
        int q_extent = 0;
        for (int i = 0; i < kNQ; ++i)
            if (qr[2 * i + 1]) q_extent = i + 1;
        long double expected = 0;
        for (int i = 0; i < q_extent; ++i) expected += (q_to_p(i) * qr[2 * i + 1]);
        d.meanq[r] = p_to_q(expected / rg[2 * r + 1]);
        d.rgdq[r] = delta_for(rg[2 * r], rg[2 * r + 1], d.meanq[r]);
        const int prior_rg = d.meanq[r] + d.rgdq[r];
        int prior_q[kNQ];
        for (int i = 0; i < q_extent; ++i) {
            d.qdq[r * kNQ + i] = delta_for(qr[2 * i], qr[2 * i + 1], prior_rg);
            prior_q[i] = prior_rg + d.qdq[r * kNQ + i];
        }
        for (int i = 0; i < q_extent; ++i)
            for (int s = 0; s < 2; ++s) {
                const uint64_t cell0 = ((r * kNQ + i) * 2 + s) * n_cycle;
                uint64_t extent = 0;
                for (uint64_t c = 0; c < n_cycle; ++c)
                    if (cycle[(cell0 + c) * 2 + 1]) extent = c + 1;
                for (uint64_t c = 0; c < extent; ++c)
                    d.cycledq[cell0 + c] = delta_for(cycle[(cell0 + c) * 2], cycle[(cell0 + c) * 2 + 1], prior_q[i]);
            }
        for (int i = 0; i < q_extent; ++i) {
            const uint64_t cell0 = (r * kNQ + i) * 16;
            bool seen = false;
            for (int x = 0; x < 16; ++x) seen = seen || dinuc[(cell0 + x) * 2 + 1];
            if (!seen) continue;   // CDinucCovariate keeps an empty vector for this q
            for (int x = 0; x < 16; ++x)
                d.dinucdq[cell0 + x] = delta_for(dinuc[(cell0 + x) * 2], dinuc[(cell0 + x) * 2 + 1], prior_q[i]);
        }
    };
    // read groups are independent (covariateutils.cc:204-230 loops over them): one thread each, up to 16 at a time
    const uint64_t n_threads = std::min<uint64_t>(n_rg, 16);
    if (n_threads <= 1) {
        for (uint64_t r = 0; r < n_rg; ++r) one_group(r);
    } else {
        std::vector<std::thread> pool;
        for (uint64_t t = 0; t < n_threads; ++t)
            pool.emplace_back([&, t] { for (uint64_t r = t; r < n_rg; r += n_threads) one_group(r); });
        for (auto &th : pool) th.join();
    }
    return d;
}

}  // namespace kbbq
