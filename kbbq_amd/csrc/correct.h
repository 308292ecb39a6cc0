// correct.h -- the trusted-k-mer error-finding scan (CReadData::get_errors and
// its helpers) for ONE read per lane.
//
// Reference: readutils.cc:195-235 (correct_one), :238-570 (get_errors);
// bloom.cc:83-94 (get_next_trusted_char), :96-128 (find_longest_trusted_seq),
// :130-188 (find_longest_fix), :208-277 (adjust_right_anchor), :279-305
// (biggest_consecutive_trusted_block).
//
// Design (not a translation): the reference copies std::strings around and
// builds an explicit reverse complement for the left-hand walk.  Here a read
// lives once, 2-bit packed, in a per-lane LDS slice; both walks run over a
// "virtual string" V(lo, n, dir) that indexes that slice forwards (dir=+1) or
// reverse-complemented (dir=-1), and the one-level recursion on a bad
// prefix/suffix becomes up to two further calls over a sub-range of the same
// slice.  First scans and the trusted-region veto reuse the trusted-k-mer bit
// mask T that the wave-parallel scan kernel already computed, falling back to
// Bloom queries only once correct_one has changed a base.  Error flags leave
// the lane as a bit mask.
#pragma once
#include "device_common.h"

namespace kbbq {

struct Km {
    uint64_t fw, rc;
    int n;
};

struct FixResult {
    int nbest;      // size of the reference's best_c vector
    int best0;      // its first element
    int stop;       // best_i
    bool multiple;
};

struct CallResult {
    int bad_prefix;   // 0 = none
    int bad_suffix;   // -1 = none
    int patch_pos;    // >= 0: get_errors returned early with seq[patch_pos] replaced (readutils.cc:263-265)
    int patch_base;
};

template <int MAXL>
struct Corrector {
    // MAXL = 0: sized at run time (reads longer than 512 bases; the lane's words then live in global memory)
    int dyn_len;        // MAXL == 0 only: the capacity in bases, a multiple of 32
    __device__ __forceinline__ int NWB() const { return (MAXL ? MAXL : dyn_len) / 16; }   // 2-bit words
    __device__ __forceinline__ int NWN() const { return (MAXL ? MAXL : dyn_len) / 32; }   // 1-bit words
    __device__ __forceinline__ int OFF_W() const { return 0; }                            // working sequence
    __device__ __forceinline__ int OFF_NM() const { return NWB(); }                       // its non-ACGT mask
    __device__ __forceinline__ int OFF_O() const { return NWB() + NWN(); }                // sequence at entry of the current call
    __device__ __forceinline__ int OFF_ON() const { return 2 * NWB() + NWN(); }           // its non-ACGT mask
    __device__ __forceinline__ int OFF_E() const { return 2 * NWB() + 2 * NWN(); }        // error flags
    __device__ __forceinline__ int OFF_T() const { return 2 * NWB() + 3 * NWN(); }        // trusted k-mer mask of the original read
    // Off-case bases ('a','c','g','t' and the digits seq_nt16_table folds to bases): three loops of the reference
    // compare RAW characters with the candidates 'A','C','G','T' (bloom.cc:142,218,249; readutils.cc:202), so for
    // such a base the candidate equal to it is tried as well -- on this->seq only; the left-hand walk runs on the
    // upper-case `revcomped` string (readutils.cc:351-353).  A fix writes an upper-case letter: the bit is cleared.
    __device__ __forceinline__ int OFF_LC() const { return 2 * NWB() + 4 * NWN(); }
    static constexpr int words_for(int maxl) { return 2 * (maxl / 16) + 5 * (maxl / 32); }
    static constexpr int WORDS = words_for(MAXL);

    uint32_t *L;        // this lane's first LDS word
    int stride;         // distance between consecutive words of one lane
    FiltDev f;
    KParams K;
    const uint8_t *qual;
    bool t_ok;          // T still describes the working sequence (no correct_one patch yet)
    unsigned queries;

    __device__ __forceinline__ uint32_t &word(int off, int i) { return L[(off + i) * stride]; }
    __device__ __forceinline__ bool bit(int off, int i) { return (word(off, i >> 5) >> (i & 31)) & 1u; }
    __device__ __forceinline__ void setbit(int off, int i, bool v) {
        uint32_t &w = word(off, i >> 5);
        w = v ? (w | (1u << (i & 31))) : (w & ~(1u << (i & 31)));
    }
    __device__ __forceinline__ int code_in(int offb, int offn, int i) {
        if (bit(offn, i)) return 4;
        return (word(offb, i >> 4) >> ((i & 15) * 2)) & 3u;
    }
    __device__ __forceinline__ int code(int i) { return code_in(OFF_W(), OFF_NM(), i); }
    __device__ __forceinline__ void setcode(int i, int c) {   // c in 0..4
        uint32_t &w = word(OFF_W(), i >> 4);
        const int sh = (i & 15) * 2;
        w = (w & ~(3u << sh)) | ((uint32_t)(c & 3) << sh);
        setbit(OFF_NM(), i, c > 3);
        if (c > 3) w &= ~(3u << sh);
    }
    // virtual string: forward, or reverse-complemented (the `revcomped` string of readutils.cc:351-353)
    __device__ __forceinline__ int vcode(int lo, int n, int dir, int p) {
        const int c = code(dir > 0 ? lo + p : lo + n - 1 - p);
        return dir > 0 ? c : (c < 4 ? 3 - c : 4);
    }
    __device__ __forceinline__ void vset(int lo, int n, int dir, int p, int c) {
        if (dir > 0) setcode(lo + p, c); else setcode(lo + n - 1 - p, 3 - c);
    }

    __device__ __forceinline__ void kreset(Km &m) { m.fw = m.rc = 0; m.n = 0; }
    __device__ __forceinline__ void kpush(Km &m, int c) {   // Kmer::push_back, bloom.hh:350-358
        if (c < 4) {
            m.fw = ((m.fw << 2) | (uint64_t)c) & K.mask;
            m.rc = (m.rc >> 2) | ((uint64_t)(3 - c) << K.shift);
            ++m.n;
        } else {
            kreset(m);
        }
    }
    __device__ __forceinline__ bool kvalid(const Km &m) { return m.n >= K.k; }
    __device__ __forceinline__ bool query(const Km &m) {   // Bloom::query, bloom.hh:398
        if (!kvalid(m)) return false;
        ++queries;
        return bloom_has(f, m.fw < m.rc ? m.fw : m.rc);
    }

    // get_next_trusted_char, bloom.cc:83-94; -1 = none
    __device__ int next_trusted(const Km &m, bool reverse_order) {
        for (int j = 0; j < 4; ++j) {
            const int c = reverse_order ? 3 - j : j;
            Km e = m;
            kpush(e, c);
            if (query(e)) return c;
        }
        return -1;
    }

    // find_longest_trusted_seq, bloom.cc:96-128, on V(lo, n, +1).  a1 = -1 encodes npos
    // ("trusted to the end"), a0 = -1 encodes "no trusted k-mer".
    __device__ void longest_trusted(int lo, int n, bool from_mask, int &a0, int &a1) {
        int best = 0, cur = 0;
        a0 = a1 = -1;
        const int k = K.k;
        if (from_mask) {
            // T bit s <=> the k-mer starting at base s of the original read is trusted
            const int last = n - k;   // last k-mer start inside the range
            for (int s = 0; s <= last; ++s) {
                if (bit(OFF_T(), lo + s)) {
                    ++cur;
                } else {
                    if (cur > best) { best = cur; a1 = s + k - 2; a0 = s - cur; }
                    cur = 0;
                }
            }
            if (cur > best) { best = cur; a1 = -1; a0 = n + 1 - k - cur; }
            return;
        }
        Km m;
        kreset(m);
        for (int i = 0; i < n; ++i) {
            kpush(m, code(lo + i));
            if (m.n >= k) {
                if (query(m)) {
                    ++cur;
                } else {
                    if (cur > best) { best = cur; a1 = i - 1; a0 = i + 1 - k - cur; }
                    cur = 0;
                }
            } else if (cur != 0) {
                if (cur > best) { best = cur; a1 = i - 1; a0 = i + 1 - k - cur; }
                cur = 0;
            }
        }
        if (cur > best) { best = cur; a1 = -1; a0 = n + 1 - k - cur; }
    }

    // find_longest_fix, bloom.cc:130-188, on the suffix of V(lo, n, dir) that starts at `start`;
    // the base under test is V[start + k - 1].
    __device__ FixResult longest_fix(int lo, int n, int dir, int start) {
        const int k = K.k;
        const int sublen = n - start;
        const int i_stop = max(2 * k - 1, sublen);
        FixResult r;
        r.nbest = 0; r.best0 = 0; r.stop = 0; r.multiple = false;
        bool single = false;
        Km head;
        kreset(head);
        for (int i = 0; i < k - 1; ++i) kpush(head, vcode(lo, n, dir, start + i));
        const int unfixed = vcode(lo, n, dir, start + k - 1);
        const bool raw_differs = dir > 0 && bit(OFF_LC(), lo + start + k - 1);      // bloom.cc:142 on an off-case base
        for (int jj = 0; jj < 4; ++jj) {
            const int cand = dir > 0 ? jj : 3 - jj;
            if (cand == unfixed && !raw_differs) continue;
            Km m = head;
            int i = k - 1;
            kpush(m, cand);
            bool go = false;
            if (kvalid(m) && query(m)) {
                if (single) r.multiple = true;
                single = true;
                go = true;
                i = k;
            }
            while (go && i < i_stop) {
                if (i < sublen) {
                    kpush(m, vcode(lo, n, dir, start + i));
                    if (!(kvalid(m) && query(m))) break;
                } else {
                    const int c = next_trusted(m, dir < 0);
                    if (c < 0) break;
                    kpush(m, c);   // known trusted: next_trusted just queried this very k-mer
                }
                ++i;
            }
            if (i > r.stop) { r.nbest = 1; r.best0 = cand; r.stop = i; }
            else if (i == r.stop) { if (r.nbest == 0) r.best0 = cand; ++r.nbest; }
        }
        return r;
    }

    // adjust_right_anchor, bloom.cc:208-277, on V(lo, n, dir)
    __device__ int adjust_anchor(int lo, int n, int dir, int anchor, bool &multiple) {
        const int k = K.k;
        multiple = false;
        int mod = anchor + 1;
        Km m;
        kreset(m);
        for (int i = mod - k + 1; i < mod; ++i) kpush(m, vcode(lo, n, dir, i));
        const int at_mod = vcode(lo, n, dir, mod);
        for (int c = 0; c < 4; ++c) {
            if (at_mod == c && !(dir > 0 && bit(OFF_LC(), lo + mod))) continue;
            Km nk = m;
            kpush(nk, c);
            for (int i = 0; i <= k; ++i) {
                if (!query(nk)) break;
                if (mod + i == n - 1 || i == k) return anchor;
                kpush(nk, vcode(lo, n, dir, mod + i + 1));
            }
        }
        for (int i = k / 2 - 1; i >= 0 && anchor > i + k - 1; --i) {
            kreset(m);
            mod = anchor - i;
            for (int j = mod - k + 1; j < mod; ++j) kpush(m, vcode(lo, n, dir, j));
            const int here = vcode(lo, n, dir, mod);
            for (int c = 0; c < 4; ++c) {
                if (here == c && !(dir > 0 && bit(OFF_LC(), lo + mod))) continue;
                Km nk = m;
                kpush(nk, c);
                if (kvalid(nk) && query(nk)) {
                    multiple = true;
                    bool ok = true;
                    for (int j = 0; ok && mod + 1 + j < n && j <= k / 2; ++j) {
                        kpush(nk, vcode(lo, n, dir, mod + 1 + j));
                        ok = kvalid(nk) && query(nk);
                        if (j == k / 2 && ok) return mod - 1;
                    }
                }
            }
        }
        return anchor;
    }

    // biggest_consecutive_trusted_block, bloom.cc:279-305, on working bases [from, from+wlen)
    __device__ int biggest_block(int from, int wlen, int current_len) {
        const int k = K.k;
        Km m;
        kreset(m);
        int in = 0, out = 0, len = 0;
        for (int i = 0; i < wlen; ++i) {
            kpush(m, code(from + i));
            if (i >= k - 1) {
                if (query(m)) {
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

    // correct_one, readutils.cc:195-235, on working bases [lo, lo+n); -1 = npos
    __device__ int correct_one(int lo, int n, int &fixed_base) {
        const int k = K.k;
        int best_len = 0, best_base = 0, best_pos = -1;
        for (int i = 0; i < n; ++i) {
            const int orig = code(lo + i);
            for (int c = 0; c < 4; ++c) {
                if (orig == c && !bit(OFF_LC(), lo + i)) continue;
                setcode(lo + i, c);
                const int start = i > k - 1 ? i - k + 1 : 0;
                const int magic = i > k / 2 - 1 ? min(i - k / 2 + 1, n - k) : 0;
                Km m;
                kreset(m);
                for (int j = magic; j <= magic + k - 1; ++j) kpush(m, code(lo + j));
                if (query(m)) {
                    const int n_in = biggest_block(lo + start, min(2 * k - 1, n - start), best_len);
                    if (n_in > best_len) {
                        best_base = c; best_pos = i; best_len = n_in;
                    } else if (n_in == best_len && qual[lo + i] < qual[lo + best_pos]) {
                        best_base = c; best_pos = i;
                    }
                }
            }
            setcode(lo + i, orig);
        }
        if (best_len > 0) { setcode(lo + best_pos, best_base); setbit(OFF_LC(), lo + best_pos, false); }
        fixed_base = best_base;
        return best_pos;
    }

    // One activation of get_errors (readutils.cc:238-546) on the sub-read [lo, lo+n).
    // The recursion of :547-563 is driven by the caller from the returned bad_prefix / bad_suffix.
    __device__ CallResult run_call(int lo, int n, bool first_call, int minqual) {
        const int k = K.k;
        CallResult res;
        res.bad_prefix = 0; res.bad_suffix = -1; res.patch_pos = -1; res.patch_base = 0;
        if (n < k) return res;   // engine-defined (the reference has undefined behaviour here)
        const bool entry_mask_ok = first_call || t_ok;
        // snapshot = original_seq of this activation
        for (int i = 0; i < NWB(); ++i) word(OFF_O(), i) = word(OFF_W(), i);
        for (int i = 0; i < NWN(); ++i) word(OFF_ON(), i) = word(OFF_NM(), i);
        bool multiple = false;
        int a0, a1;
        longest_trusted(lo, n, entry_mask_ok, a0, a1);
        int patched_at = -1, patched_base = 0;
        if (a0 < 0) {
            multiple = true;
            patched_at = correct_one(lo, n, patched_base);
            if (patched_at < 0) return res;
            t_ok = false;
            longest_trusted(lo, n, false, a0, a1);
            setbit(OFF_E(), lo + patched_at, true);
        }
        if (a0 == 0 && a1 < 0) {
            if (patched_at >= 0) { res.patch_pos = lo + patched_at; res.patch_base = patched_base; }
            return res;
        }
        const int anchor_len = (a1 < 0 ? n - 1 : min(a1, n - 1)) + 1 - a0;
        bool corrected = false;
        // right-hand walk, readutils.cc:271-346
        if (a1 >= 0) {
            if (anchor_len - k + 1 >= k) {
                bool m2;
                a1 = adjust_anchor(lo, n, +1, a1, m2);
                multiple = multiple || m2;
            }
            for (int i = a1 + 1; i < n;) {
                const int start = i - k + 1;
                const FixResult fx = longest_fix(lo, n, +1, start);
                multiple = multiple || fx.multiple;
                const int next_untrusted = start + fx.stop;
                if (next_untrusted > i) {
                    if (fx.nbest > 1) {
                        multiple = true;
                        const int largest = min(i + k - 1, n - 1);
                        if (next_untrusted <= largest || largest - i + 1 < k) { res.bad_suffix = i; break; }
                    } else {
                        vset(lo, n, +1, i, fx.best0);
                        setbit(OFF_LC(), lo + i, false);
                        setbit(OFF_E(), lo + i, true);
                    }
                    corrected = true;
                    i += fx.stop - k + 1;
                } else {
                    res.bad_suffix = i;
                    break;
                }
            }
        }
        // left-hand walk on the reverse complement, readutils.cc:348-422
        if (a0 != 0) {
            if (anchor_len - k + 1 >= k) {
                bool m2;
                const int adj = adjust_anchor(lo, n, -1, n - a0 - 1, m2);
                a0 = n - adj - 1;
                multiple = multiple || m2;
            }
            for (int i = a0 - 1; i >= 0;) {
                const int j = n - i - 1;
                const int start = j - k + 1;
                const FixResult fx = longest_fix(lo, n, -1, start);
                multiple = multiple || fx.multiple;
                const int next_untrusted = start + fx.stop;
                if (next_untrusted > j) {
                    if (fx.nbest > 1) {
                        multiple = true;
                        const int largest = min(j + k - 1, n - 1);
                        if (next_untrusted <= largest || largest - j + 1 < k) { res.bad_prefix = i; break; }
                    } else {
                        vset(lo, n, -1, j, fx.best0);
                        setbit(OFF_E(), lo + i, true);
                    }
                    corrected = true;
                    i -= next_untrusted - j;
                } else {
                    res.bad_prefix = i;
                    break;
                }
            }
        }
        // over-correction control, readutils.cc:429-546
        if (corrected) {
            bool adjust = true;
            if (entry_mask_ok) {
                // maximal runs of trusted k-mers of the entry sequence that end before the last k-mer
                int run0 = -1;
                const int last = n - k;
                for (int s = 0; s <= last && adjust; ++s) {
                    if (bit(OFF_T(), lo + s)) {
                        if (run0 < 0) run0 = s;
                    } else if (run0 >= 0) {
                        for (int j = run0; j <= s - 1 + k - 1; ++j)
                            if (bit(OFF_E(), lo + j)) { adjust = false; break; }
                        run0 = -1;
                    }
                }
            } else {
                Km m;
                kreset(m);
                int ts = -1, te = -1;
                for (int i = 0; i < n && adjust; ++i) {
                    kpush(m, code_in(OFF_O(), OFF_ON(), lo + i));
                    if (kvalid(m) && query(m)) {
                        ts = ts < 0 ? i - k + 1 : min(ts, i - k + 1);
                        te = i;
                    } else if (te >= 0) {
                        for (int j = ts; j <= te; ++j)
                            if (bit(OFF_E(), lo + j)) { adjust = false; break; }
                        ts = te = -1;
                    }
                }
            }
            adjust = adjust && !multiple;
            const int ocwindow = 20, base_threshold = 4;
            int occ2 = 0;   // twice the reference's `occount` (it only ever moves by 0.5 or 1)
            // the snapshot's base words are free now: reuse them as the overcorrected-index set
            for (int i = 0; i < NWN(); ++i) word(OFF_O(), i) = 0;
            for (int i = 0; i < n; ++i) {
                const bool e = bit(OFF_E(), lo + i);
                if (e && !bit(OFF_ON(), lo + i)) occ2 += qual[lo + i] <= minqual ? 1 : 2;
                if (i >= ocwindow && bit(OFF_E(), lo + i - ocwindow) && !bit(OFF_ON(), lo + i - ocwindow))
                    occ2 -= qual[lo + i - ocwindow] <= minqual ? 1 : 2;
                const int threshold = (adjust && i >= ocwindow && i + ocwindow - 1 < n) ? base_threshold + 1 : base_threshold;
                if (occ2 > 2 * threshold && e) setbit(OFF_O(), i, true);
            }
            for (int oc = 0; oc < n; ++oc) {
                if (!bit(OFF_O(), oc)) continue;
                if (!bit(OFF_E(), lo + oc)) continue;
                int start = oc - k + 1;
                start = start >= 0 ? start : 0;
                int end = oc + k;
                end = end < n ? end : n;
                for (int i = start; i < end; ++i) {
                    if (bit(OFF_E(), lo + i)) {
                        setbit(OFF_E(), lo + i, false);
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
