// correct_wave.h -- the trusted-k-mer error-finding scan (CReadData::get_errors and
// helpers), ONE READ PER WAVEFRONT.
//
// Reference: readutils.cc:195-235 (correct_one), :238-570 (get_errors);
// bloom.cc:83-94 (get_next_trusted_char), :96-128 (find_longest_trusted_seq),
// :130-188 (find_longest_fix), :208-277 (adjust_right_anchor), :279-305
// (biggest_consecutive_trusted_block).
//
// The reference walks a read base by base, re-querying the Bloom filter through
// std::string copies.  Here the 64 lanes of a wavefront own one read:
//   * the read (2-bit bases, non-ACGT mask, error flags, trusted-k-mer masks)
//     lives in wave-uniform 64-bit words; control flow is wave-uniform;
//   * "which alternative base at position m makes its k covering k-mers
//     trusted" -- the question behind adjust_right_anchor, find_longest_fix and
//     correct_one -- is answered by a PROBE: one lane per (candidate, k-mer), every
//     lane fetching its own 128-bit Bloom block, the answer returned as one bit mask
//     per candidate;
//   * k-mers that do not cover the modified base are never re-queried: their
//     status is a bit of the trusted mask the scan kernel already produced
//     (recomputed in parallel only after correct_one changed a base);
//   * the left-hand walk needs no reverse-complement copy: a k-mer's canonical
//     form is strand-symmetric, so the reference's walk over `revcomped` is the
//     same walk over mirrored indices with complemented candidate bases;
//   * the over-correction window sum and the trusted-region veto are range
//     pop-counts on ballot words.
// Results are bit-identical to the one-read-per-lane formulation in correct.h
// (kept as the k < 3 path) and to the oracle.
#pragma once
#include "device_common.h"

namespace kbbq {

template <int NB, int NN>   // NB: 64-bit words of 2-bit bases (32 each); NN: 64-bit words of flags
struct WaveCorrector {
    // The read's words live in a per-wavefront LDS slice (every lane reads the same words: LDS
    // broadcast; dynamic word indices cost nothing).  Each array carries one zero pad word.
    static constexpr int W = 0;                  // working sequence, NB words
    static constexpr int NM = W + NB + 1;        // its non-ACGT mask
    static constexpr int ON = NM + NN + 1;       // non-ACGT mask at entry of the current activation (original_seq)
    static constexpr int E = ON + NN + 1;        // CReadData::errors
    static constexpr int Te = E + NN + 1;        // trusted mask of the current activation's entry sequence
    static constexpr int Tc = Te + NN + 1;       // trusted mask of the working sequence (as of the last correct_one patch)
    static constexpr int H0 = Tc + NN + 1;       // scratch: 4 x (NN+1) candidate hit masks / weight masks
    static constexpr int WORDS = H0 + 4 * (NN + 1);
    uint64_t *S;         // this wavefront's slice
    FiltDev f;
    KParams K;
    const uint8_t *qual;
    int lane;
    unsigned queries;

    // Everything below that steers control flow is wave-uniform by construction (it derives from
    // ballots and from data every lane loaded identically).  The compiler cannot always prove that,
    // so such values go through readfirstlane: they then live in SGPRs, branches on them are scalar
    // branches, and the ballots that collect the answers of an `ask` see every lane.
    template <typename T>
    static __device__ __forceinline__ T uni(T v) {
        if constexpr (sizeof(T) == 8) {
            const uint64_t x = (uint64_t)v;
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32));
            return (T)(((uint64_t)hi << 32) | lo);
        } else {
            return (T)__builtin_amdgcn_readfirstlane((int)v);
        }
    }

    // ---- wave-uniform bit containers -------------------------------------------------
    __device__ __forceinline__ uint64_t ldw(int arr, int idx) const { return S[arr + idx]; }
    __device__ __forceinline__ void stw(int arr, int idx, uint64_t v) { S[arr + idx] = v; }
    __device__ __forceinline__ bool getbit(int arr, int i) const { return (S[arr + (i >> 6)] >> (i & 63)) & 1; }
    __device__ __forceinline__ void putbit(int arr, int i, bool v) {
        const uint64_t w = S[arr + (i >> 6)], b = 1ULL << (i & 63);
        S[arr + (i >> 6)] = v ? (w | b) : (w & ~b);
    }
    // first index in [from, limit) whose bit equals `one`; limit if none (wave-uniform arguments)
    __device__ __forceinline__ int find_next(int arr, int from, int limit, bool one) const {
        while (from < limit) {
            uint64_t x = uni(S[arr + (from >> 6)]);
            if (!one) x = ~x;
            x >>= (from & 63);
            if (x) {
                const int r = from + __ffsll((unsigned long long)x) - 1;
                return r < limit ? r : limit;
            }
            from = ((from >> 6) + 1) << 6;
        }
        return limit;
    }
    // last index in [floor, from] whose bit equals `one`; floor-1 if none
    __device__ __forceinline__ int find_prev(int arr, int from, int floor, bool one) const {
        while (from >= floor) {
            uint64_t x = uni(S[arr + (from >> 6)]);
            if (!one) x = ~x;
            x <<= 63 - (from & 63);
            if (x) {
                const int r = from - __clzll((long long)x);
                return r >= floor ? r : floor - 1;
            }
            from = ((from >> 6) << 6) - 1;
        }
        return floor - 1;
    }
    // popcount of bits lo..hi (inclusive, hi - lo < 64), lane-varying bounds
    __device__ __forceinline__ int range_popc(int arr, int lo, int hi) const {
        const int a = lo >> 6, b = hi >> 6;
        const uint64_t wa = S[arr + a];
        if (a == b) {
            const int n = hi - lo + 1;
            const uint64_t m = (n >= 64 ? ~0ULL : ((1ULL << n) - 1)) << (lo & 63);
            return __popcll(wa & m);
        }
        const uint64_t wb = S[arr + b];
        const int nb = (hi & 63) + 1;
        return __popcll(wa >> (lo & 63)) + __popcll(wb & (nb >= 64 ? ~0ULL : ((1ULL << nb) - 1)));
    }

    __device__ __forceinline__ int code(int i) const {
        if (getbit(NM, i)) return 4;
        return (int)((S[W + (i >> 5)] >> ((i & 31) * 2)) & 3);
    }
    __device__ __forceinline__ int code_u(int i) const { return uni(code(i)); }   // i wave-uniform
    __device__ __forceinline__ void setcode(int i, int c) {
        const int sh = (i & 31) * 2;
        S[W + (i >> 5)] = (S[W + (i >> 5)] & ~(3ULL << sh)) | ((uint64_t)(c > 3 ? 0 : c) << sh);
        putbit(NM, i, c > 3);
    }

    // ---- k-mers ------------------------------------------------------------------------
    struct Pair { uint64_t fw, rc; bool valid; };
    // forward / reverse-complement words of the k-mer starting at base st (lane-varying),
    // with base m replaced by c when it falls inside (m < 0: no replacement)
    __device__ __forceinline__ Pair kmer_pair(int st, int m, int c) const {
        const int wi = st >> 5, s2 = (st & 31) * 2;
        const uint64_t lo = S[W + wi], hi = S[W + wi + 1];
        uint64_t w = s2 ? ((lo >> s2) | (hi << (64 - s2))) : lo;
        const int ni = st >> 6, s1 = st & 63;
        const uint64_t nlo = S[NM + ni], nhi = S[NM + ni + 1];
        uint32_t nm = (uint32_t)(s1 ? ((nlo >> s1) | (nhi << (64 - s1))) : nlo) & K.nmask_bits;
        const int j = m - st;
        if (m >= 0 && j >= 0 && j < K.k) {
            w = (w & ~(3ULL << (2 * j))) | ((uint64_t)c << (2 * j));
            nm &= ~(1u << j);
        }
        Pair p;
        p.valid = nm == 0;
        p.rc = (~w) & K.mask;
        p.fw = rev2(w) >> (64 - 2 * K.k);
        return p;
    }
    // Bloom::query for one k-mer per lane
    __device__ __forceinline__ bool ask(bool active, const Pair &p) {
        const bool go = active && p.valid;
        const uint64_t key = p.fw < p.rc ? p.fw : p.rc;
        queries += __popcll(__ballot(go));
        return go && bloom_has(f, key);
    }

    // trusted mask of the working sequence for k-mer starts in [lo, lo+n-k]
    __device__ __forceinline__ void rescan(int lo, int n) {
        const int last = lo + n - K.k;
#pragma unroll 1
        for (int c = 0; c < NN; ++c) {
            if (c * 64 <= last && c * 64 + 63 >= lo) {
                const int s = c * 64 + lane;
                const bool in = s >= lo && s <= last;
                const Pair p = kmer_pair(in ? s : lo, -1, 0);
                const bool t = ask(in, p);
                const uint64_t got = __ballot(t), rng = __ballot(in);
                S[Tc + c] = (S[Tc + c] & ~rng) | got;
            }
        }
    }

    // find_longest_trusted_seq (bloom.cc:96-128) from a trusted mask, relative to [lo, lo+n).
    // a0 = -1: no trusted k-mer; a1 = -1: trusted to the end (npos).
    __device__ __forceinline__ void longest_run(int T, int lo, int n, int &a0, int &a1) const {
        const int k = K.k, end = lo + n - k + 1;   // one past the last k-mer start
        int best = 0;
        a0 = a1 = -1;
        int pos = find_next(T, lo, end, true);
        while (pos < end) {
            const int z = find_next(T, pos, end, false);
            const int len = z - pos;
            if (len > best) {
                best = len;
                a0 = pos - lo;
                a1 = z == end ? -1 : (z - lo) + k - 2;
            }
            pos = find_next(T, z, end, true);
        }
    }

    // ---- probe: candidates at one position ------------------------------------------------
    // M[y] bit jb = the k-mer that has position m at offset jb (start m - jb), with base m := y,
    // is trusted.  Two dependent rounds: the first k-mer of the walk for every candidate, then the
    // remaining covering k-mers of the candidates that survived.
    struct Probe { uint32_t M[4]; int first_ok; };   // first_ok: bit y = first-step k-mer trusted
    static __device__ __forceinline__ uint32_t mask_of(const Probe &pr, int y) {
        return y == 0 ? pr.M[0] : y == 1 ? pr.M[1] : y == 2 ? pr.M[2] : pr.M[3];
    }

    __device__ __forceinline__ Probe probe(int lo, int n, int m_abs, int dir) {
        const int k = K.k;
        Probe pr;
        pr.M[0] = pr.M[1] = pr.M[2] = pr.M[3] = 0;
        const int cur = code_u(m_abs);
        const int jb_first = dir > 0 ? k - 1 : 0;
        {
            const int y = lane & 3;
            const bool act = lane < 4 && y != cur;
            const Pair p = kmer_pair(m_abs - jb_first, m_abs, y);
            const bool t = ask(act, p);
            pr.first_ok = uni((int)(__ballot(t) & 0xF));
        }
        const int last_start = lo + n - k;
        int todo = pr.first_ok;
        while (todo) {
            const int ya = __ffs(todo) - 1;
            todo &= todo - 1;
            int yb = -1;
            if (todo) { yb = __ffs(todo) - 1; todo &= todo - 1; }
            const int half = lane >> 5, jb = lane & 31;
            const int y = half ? yb : ya;
            const int st = m_abs - jb;
            const bool act = y >= 0 && jb < k && st >= lo && st <= last_start;
            const Pair p = kmer_pair(act ? st : lo, m_abs, y < 0 ? 0 : y);
            const bool t = ask(act, p);
            const uint64_t bal = __ballot(t);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q == ya) pr.M[q] = uni((uint32_t)bal);
                if (q == yb) pr.M[q] = uni((uint32_t)(bal >> 32));
            }
        }
        return pr;
    }

    // number of consecutive trusted walk steps for candidate y at position m (relative rel_m),
    // find_longest_fix's `i - (k-1)` (bloom.cc:148-172): covering k-mers in walk order, then the
    // untouched k-mers beyond (from the trusted mask), then -- at the read end -- extension with
    // the first trusted next base.
    __device__ __forceinline__ int walk_steps(const Probe &pr, int y, int lo, int n, int rel_m, int dir, bool need_exact) {
        const int k = K.k;
        const uint32_t M = uni(mask_of(pr, y));
        int steps;
        if (dir > 0) {
            const int jmin = max(0, rel_m - (n - k));
            // consecutive ones from bit k-1 downwards to jmin
            const uint32_t top = k < 32 ? (M | ~((1u << k) - 1)) : M;       // pad above k-1 with ones
            const uint32_t inv = ~top;
            const int lead_all = inv ? __clz((int)inv) - (32 - k) : k;       // ones from bit k-1 down
            const int avail = k - jmin;
            const int lead = min(lead_all, avail);
            steps = lead;
            if (lead < avail) return steps;
            if (jmin == 0) {
                const int from = lo + rel_m + 1, end = lo + n - k + 1;
                const int z = find_next(Tc, from, end, false);
                return steps + (z - from);
            }
        } else {
            const int jmax = min(k - 1, rel_m);
            const uint32_t inv = ~M;
            const int lead_all = inv ? __ffs((int)inv) - 1 : 32;           // ones from bit 0 upwards
            const int avail = jmax + 1;
            const int lead = min(lead_all, avail);
            steps = lead;
            if (lead < avail) return steps;
            if (jmax == k - 1) {
                const int from = lo + rel_m - k;                            // next untouched k-mer start, going down
                if (from < lo) return steps;
                const int z = find_prev(Tc, from, lo, false);           // last zero at or below `from`
                return steps + (from - z);
            }
        }
        // read end reached with every k-mer trusted: extend (get_next_trusted_char, bloom.cc:83-94)
        const int iv = dir > 0 ? rel_m : n - 1 - rel_m;
        int ext_max = k - (n - iv);
        // The walk has reached the read end (steps >= n - iv), which already ends the caller's loop.
        // How far the extension goes only matters when another candidate could tie with this one.
        if (ext_max <= 0 || !need_exact) return steps;
        // k-mer at the read end, with the candidate in place
        Pair p = kmer_pair(dir > 0 ? lo + n - k : lo, lo + rel_m, y);
        uint64_t fw = uni(p.fw), rc = uni(p.rc);
        while (ext_max-- > 0) {
            const int x = lane & 3;
            Pair q;
            q.valid = true;
            if (dir > 0) {
                q.fw = ((fw << 2) | (uint64_t)x) & K.mask;
                q.rc = (rc >> 2) | ((uint64_t)(3 - x) << K.shift);
            } else {
                q.fw = (fw >> 2) | ((uint64_t)x << K.shift);
                q.rc = ((rc << 2) | (uint64_t)(3 - x)) & K.mask;
            }
            const bool t = ask(lane < 4, q);
            const int hit = uni((int)(__ballot(t) & 0xF));
            if (!hit) break;
            const int xb = __ffs(hit) - 1;   // A,C,G,T order when appending; T,G,C,A of the reverse strand = A,C,G,T prepended
            if (dir > 0) {
                fw = ((fw << 2) | (uint64_t)xb) & K.mask;
                rc = (rc >> 2) | ((uint64_t)(3 - xb) << K.shift);
            } else {
                fw = (fw >> 2) | ((uint64_t)xb << K.shift);
                rc = ((rc << 2) | (uint64_t)(3 - xb)) & K.mask;
            }
            ++steps;
        }
        return steps;
    }

    struct Fix { int nbest, best_y, steps; bool multiple; };
    // find_longest_fix's bookkeeping over the candidates (bloom.cc:142-186): the longest walk wins,
    // equal walks are collected (nbest), `multiple` = two candidates had a trusted first k-mer.
    __device__ __forceinline__ Fix longest_fix(const Probe &pr, int lo, int n, int rel_m, int dir) {
        Fix fx;
        fx.nbest = 0; fx.best_y = 0; fx.steps = 0;
        fx.multiple = __popc(pr.first_ok) >= 2;
        const int cur = code_u(lo + rel_m);
        bool first = true;
#pragma unroll 1
        for (int y = 0; y < 4; ++y) {
            if (y == cur) continue;
            const int s = ((pr.first_ok >> y) & 1) ? walk_steps(pr, y, lo, n, rel_m, dir, fx.multiple) : 0;
            if (first || s > fx.steps) { fx.nbest = 1; fx.best_y = y; fx.steps = s; first = false; }
            else if (s == fx.steps) ++fx.nbest;
        }
        return fx;
    }

    // adjust_right_anchor (bloom.cc:208-277) on the virtual string of direction dir; anchor and the
    // returned anchor are virtual indices relative to [lo, lo+n).  `pr_out`/`pr_m` return the probe
    // made at the first position so the walk can reuse it.
    __device__ __forceinline__ int adjust_anchor(int lo, int n, int dir, int anchor, bool &multiple, Probe &pr_out, int &pr_rel) {
        const int k = K.k;
        multiple = false;
        const int mod = anchor + 1;                                 // virtual
        const int rel = dir > 0 ? mod : n - 1 - mod;                // forward, relative
        Probe pr = probe(lo, n, lo + rel, dir);
        pr_out = pr;
        pr_rel = rel;
        const int cur = code_u(lo + rel);
        // first loop: some alternative keeps min(k+1, n - mod) consecutive k-mers trusted
        const int need = min(k + 1, n - mod);
#pragma unroll 1
        for (int y = 0; y < 4; ++y) {
            if (y == cur || !((pr.first_ok >> y) & 1)) continue;
            // covering k-mers + untouched ones, no extension: count without the extension part
            int got;
            {
                const uint32_t M = uni(mask_of(pr, y));
                if (dir > 0) {
                    const int jmin = max(0, rel - (n - k));
                    const uint32_t top = k < 32 ? (M | ~((1u << k) - 1)) : M;
                    const uint32_t inv = ~top;
                    const int lead_all = inv ? __clz((int)inv) - (32 - k) : k;
                    const int avail = k - jmin;
                    got = min(lead_all, avail);
                    if (got == avail && jmin == 0) {
                        const int from = lo + rel + 1, end = lo + n - k + 1;
                        got += find_next(Tc, from, end, false) - from;
                    }
                } else {
                    const int jmax = min(k - 1, rel);
                    const uint32_t inv = ~M;
                    const int lead_all = inv ? __ffs((int)inv) - 1 : 32;
                    const int avail = jmax + 1;
                    got = min(lead_all, avail);
                    if (got == avail && jmax == k - 1) {
                        const int from = lo + rel - k;
                        if (from >= lo) got += from - find_prev(Tc, from, lo, false);
                    }
                }
            }
            if (got >= need) return anchor;
        }
        // second loop: wind the anchor back by up to k/2 bases.  Its probes do not depend on one another (the
        // sequence does not change here) and inside an anchor an alternative base is almost never trusted, so the
        // first k-mer of every (position, candidate) is asked in ONE round -- lane = 4 * i + candidate, k/2 <= 16
        // positions -- and the loop below only probes the positions where something answered.
        uint64_t hits;
        {
            const int i_l = lane >> 2, y = lane & 3;
            const bool in = i_l <= k / 2 - 1 && anchor > i_l + k - 1;
            const int mod_l = in ? anchor - i_l : anchor;
            const int m_abs = lo + (dir > 0 ? mod_l : n - 1 - mod_l);
            const bool act = in && code(m_abs) != y;
            const Pair p = kmer_pair(m_abs - (dir > 0 ? k - 1 : 0), m_abs, y);
            hits = uni((uint64_t)__ballot(ask(act, p)));
        }
#pragma unroll 1
        for (int i = k / 2 - 1; i >= 0 && anchor > i + k - 1; --i) {
            if (!((hits >> (4 * i)) & 0xF)) continue;      // no candidate's first k-mer is trusted: nothing happens here
            const int mod2 = anchor - i;
            const int rel2 = dir > 0 ? mod2 : n - 1 - mod2;
            const Probe p2 = probe(lo, n, lo + rel2, dir);
            if (p2.first_ok) multiple = true;
            if (mod2 + 1 + k / 2 >= n) continue;      // the run of k/2+2 k-mers does not fit: never returns here
            const int cur2 = code_u(lo + rel2);
#pragma unroll 1
            for (int y = 0; y < 4; ++y) {
                if (y == cur2 || !((p2.first_ok >> y) & 1)) continue;
                const uint32_t M = uni(mask_of(p2, y));
                int lead;
                if (dir > 0) {
                    const uint32_t top = k < 32 ? (M | ~((1u << k) - 1)) : M;
                    const uint32_t inv = ~top;
                    lead = inv ? __clz((int)inv) - (32 - k) : k;
                } else {
                    const uint32_t inv = ~M;
                    lead = inv ? __ffs((int)inv) - 1 : 32;
                }
                if (lead >= k / 2 + 2) return mod2 - 1;
            }
        }
        return anchor;
    }

    // biggest_consecutive_trusted_block (bloom.cc:279-305) over the k-mers of the window
    // substr(max(0, i-k+1), 2k-1) with candidate y at rel_i, visited by increasing start: covering
    // k-mers come from the probe mask M, the others (only when rel_i < k-1) from the trusted mask.
    __device__ __forceinline__ int biggest_block(uint32_t M, int lo, int n, int rel_i, int current_len) const {
        const int k = K.k;
        const int s_lo = max(0, rel_i - k + 1), s_hi = min(s_lo + k - 1, n - k);
        int in = 0, out = 0, len = 0;
        for (int s = s_lo; s <= s_hi; ++s) {
            const bool t = s <= rel_i ? ((M >> (rel_i - s)) & 1) : getbit(Tc, lo + s);
            if (t) {
                ++in;
            } else {
                if (in > len) len = in;
                in = 0;
                ++out;
                if (k - out < current_len) break;
            }
        }
        if (in > len) len = in;
        return len;
    }

    // correct_one (readutils.cc:195-235) on [lo, lo+n); returns the relative index or -1
    __device__ __forceinline__ int correct_one(int lo, int n, int &fixed_base) {
        const int k = K.k;
        int best_len = 0, best_base = 0, best_pos = -1;
        // which (position, candidate) pairs have a trusted "magic" k-mer: one lane per position
#pragma unroll 1
        for (int y = 0; y < 4; ++y) {
#pragma unroll 1
            for (int c = 0; c < NN; ++c) {
                stw(H0 + y * (NN + 1), c, 0);
                if (c * 64 < n) {
                    const int i = c * 64 + lane;
                    const bool in = i < n;
                    const int ii = in ? i : 0;
                    const int magic = ii > k / 2 - 1 ? min(ii - k / 2 + 1, n - k) : 0;
                    const bool act = in && code(lo + ii) != y;
                    const Pair p = kmer_pair(lo + magic, lo + ii, y);
                    stw(H0 + y * (NN + 1), c, (uint64_t)__ballot(ask(act, p)));
                }
            }
        }
        // reference order: positions ascending, candidates A,C,G,T
        for (int i = 0; i < n; ++i) {
            const int hy = uni((getbit(H0, i) ? 1 : 0) | (getbit(H0 + (NN + 1), i) ? 2 : 0) |
                               (getbit(H0 + 2 * (NN + 1), i) ? 4 : 0) | (getbit(H0 + 3 * (NN + 1), i) ? 8 : 0));
            if (!hy) continue;
#pragma unroll 1
            for (int y = 0; y < 4; ++y) {
                if (!((hy >> y) & 1)) continue;
                // all covering k-mers of (i, y): lanes 0..k-1
                const int jb = lane & 31, st = lo + i - jb;
                const bool act = lane < 32 && jb < k && st >= lo && st <= lo + n - k;
                const Pair p = kmer_pair(act ? st : lo, lo + i, y);
                const uint32_t M = uni((uint32_t)__ballot(ask(act, p)));
                const int n_in = biggest_block(M, lo, n, i, best_len);
                if (n_in > best_len) {
                    best_base = y; best_pos = i; best_len = n_in;
                } else if (n_in == best_len && uni((int)qual[lo + i]) < uni((int)qual[lo + best_pos])) {
                    best_base = y; best_pos = i;
                }
            }
        }
        if (best_len > 0) setcode(lo + best_pos, best_base);
        fixed_base = best_base;
        return best_pos;
    }

    struct CallOut { int bad_prefix, bad_suffix, patch_pos, patch_base; };

    // One activation of get_errors (readutils.cc:238-546) on the sub-read [lo, lo+n).
    // On entry Tc describes the working sequence inside the range.
    __device__ __forceinline__ CallOut run_call(int lo, int n, int minqual) {
        const int k = K.k;
        CallOut res;
        res.bad_prefix = 0; res.bad_suffix = -1; res.patch_pos = -1; res.patch_base = 0;
        if (n < k) return res;   // engine-defined (undefined behaviour in the reference)
        for (int i = 0; i < NN; ++i) { stw(ON, i, ldw(NM, i)); stw(Te, i, ldw(Tc, i)); }
        bool multiple = false;
        int a0, a1;
        longest_run(Tc, lo, n, a0, a1);
        int patched_at = -1, patched_base = 0;
        if (a0 < 0) {
            multiple = true;
            patched_at = correct_one(lo, n, patched_base);
            if (patched_at < 0) return res;
            rescan(lo, n);
            longest_run(Tc, lo, n, a0, a1);
            putbit(E, lo + patched_at, true);
        }
        if (a0 == 0 && a1 < 0) {
            if (patched_at >= 0) { res.patch_pos = lo + patched_at; res.patch_base = patched_base; }
            return res;
        }
        const int anchor_len = (a1 < 0 ? n - 1 : min(a1, n - 1)) + 1 - a0;
        bool corrected = false;
        // the two walks, readutils.cc:271-346 (right of the anchor) and :348-422 (left of it, which the
        // reference runs on the reverse complement): one loop over virtual indices iv, mapped to
        // forward positions rel = iv (dir +1) or n-1-iv (dir -1)
#pragma unroll 1
        for (int side = 0; side < 2; ++side) {
            const int dir = side == 0 ? +1 : -1;
            if (dir > 0 ? a1 < 0 : a0 == 0) continue;
            int av = dir > 0 ? a1 : n - a0 - 1;
            Probe pr;
            int pr_rel = -1;
            if (anchor_len - k + 1 >= k) {
                bool m2;
                av = adjust_anchor(lo, n, dir, av, m2, pr, pr_rel);
                multiple = multiple || m2;
            }
            int bad = -1;
            for (int iv = av + 1; iv < n;) {
                const int rel = dir > 0 ? iv : n - 1 - iv;
                if (pr_rel != rel) { pr = probe(lo, n, lo + rel, dir); pr_rel = rel; }
                const Fix fx = longest_fix(pr, lo, n, rel, dir);
                multiple = multiple || fx.multiple;
                if (fx.steps > 0) {
                    if (fx.nbest > 1) {
                        multiple = true;
                        const int largest = min(iv + k - 1, n - 1);
                        if (iv + fx.steps <= largest || largest - iv + 1 < k) { bad = rel; break; }
                    } else {
                        setcode(lo + rel, fx.best_y);
                        putbit(E, lo + rel, true);
                    }
                    corrected = true;
                    iv += fx.steps;
                    pr_rel = -1;
                } else {
                    bad = rel;
                    break;
                }
            }
            if (bad >= 0) { if (dir > 0) res.bad_suffix = bad; else res.bad_prefix = bad; }
        }
        // over-correction control, readutils.cc:429-546.  A base is un-flagged only where the weighted
        // count of flags in a 20-base window exceeds the threshold (4 or 5, flags weigh 1/2 or 1): with
        // four flags or fewer in the whole range nothing can exceed it, and the trusted-region check
        // (which only selects between 4 and 5) need not run either.
        int n_flags = 0;
        if (corrected) {
            for (int c = 0; c < NN; ++c) n_flags += __popcll(ldw(E, c));
            n_flags = uni(n_flags);
        }
        if (corrected && n_flags > 4) {
            const int last = lo + n - k;   // last k-mer start
            // trusted regions of the entry sequence that end before the last k-mer must hold no flag
            const int tail_start = find_prev(Te, last, lo, false) + 1;   // first start of the final trusted run
            bool veto = false;
#pragma unroll 1
            for (int c = 0; c < NN; ++c) {
                if (c * 64 <= last && c * 64 + 63 >= lo) {
                    const int s = c * 64 + lane;
                    const bool in = s >= lo && s <= last && s < tail_start;
                    const bool hit = in && ((ldw(Te, c) >> lane) & 1) && range_popc(E, s, s + k - 1) != 0;
                    veto = veto || uni(__ballot(hit) != 0);
                }
            }
            const bool adjust = !veto && !multiple;
            const int ocwindow = 20, base_threshold = 4;
            // weights 1 (low quality) / 2 of flagged ACGT bases, as ballot words
            constexpr int w1 = H0, w2 = H0 + (NN + 1), OV = H0 + 2 * (NN + 1);
#pragma unroll 1
            for (int c = 0; c < NN; ++c) {
                uint64_t b1 = 0, b2 = 0;
                if (c * 64 < lo + n && c * 64 + 63 >= lo) {
                    const int a = c * 64 + lane;
                    const bool in = a >= lo && a < lo + n;
                    const bool flagged = in && ((ldw(E, c) >> lane) & 1) && !((ldw(ON, c) >> lane) & 1);
                    const bool lowq = flagged && qual[a] <= minqual;
                    b1 = __ballot(lowq);
                    b2 = __ballot(flagged && !lowq);
                }
                stw(w1, c, b1);
                stw(w2, c, b2);
                stw(OV, c, 0);
            }
#pragma unroll 1
            for (int c = 0; c < NN; ++c) {
                if (c * 64 < lo + n && c * 64 + 63 >= lo) {
                    const int a = c * 64 + lane;
                    const int rel = a - lo;
                    const bool in = rel >= 0 && rel < n;
                    bool over = false;
                    if (in) {
                        const int from = max(lo, a - ocwindow + 1);
                        const int occ2 = range_popc(w1, from, a) + 2 * range_popc(w2, from, a);
                        const int threshold = (adjust && rel >= ocwindow && rel + ocwindow - 1 < n) ? base_threshold + 1 : base_threshold;
                        over = occ2 > 2 * threshold && ((ldw(E, c) >> lane) & 1);
                    }
                    stw(OV, c, (uint64_t)__ballot(over));
                }
            }
            for (int oa = find_next(OV, lo, lo + n, true); oa < lo + n; oa = find_next(OV, oa + 1, lo + n, true)) {
                if (!getbit(E, oa)) continue;
                const int oc = oa - lo;
                int start = oc - k + 1;
                start = start >= 0 ? start : 0;
                int end = oc + k;
                end = end < n ? end : n;
                for (int i = start; i < end; ++i) {
                    if (getbit(E, lo + i)) {
                        putbit(E, lo + i, false);
                        if (i + k > end) end = i + k < n ? i + k : n;
                        if (i - k < start) {
                            i = i - k + 1 >= 0 ? i - k : -1;
                            start = i;
                        }
                    }
                }
            }
        }
        return res;
    }
};

}  // namespace kbbq
