// long_reads.h -- the streaming kernels for reads longer than 512 bases (gfx950).
//
// The reference has no read-length limit: get_errors works on any std::string (readutils.cc:238) and the cycle
// tables grow to the longest read (covariateutils.cc:102-116).  The short-read kernels (kernels.h) stage a whole
// read as one packed word per lane (Stage<NW>, at most 512 bases).  A longer read is walked here in WINDOWS of the
// same staging size, one after the other by the same wavefront:
//
//   k_insert_marked_long  k-mers are independent: windows of 512 k-mer starts, no overlap
//   k_scan_trusted_long   likewise; the read's trusted mask is written window by window
//   k_infer_long          infer_read_errors needs, for base i, the k-mers covering i, and the trusted-insert
//                         decision for a k-mer needs the flags of its k bases: a window of W = 512-(k-1) k-mer starts
//                         [a, b) yields exact flags for the bases [a+k-1, b-1] and exact decisions for the starts
//                         [a+k-1, b-k] (read ends extend both ranges), so consecutive windows advance by W-2(k-1)
//                         starts and every base and every start is decided by exactly one window that sees its
//                         whole neighbourhood
//
// Results are identical to the short-read kernels' where both apply (tests run reads <= 512 through both).
// These are the plain forms: no prefetch of the next window, direct inserts (the slice-bucketed emit of bucket.h
// is a short-read kernel); long reads are the rare case for this tool.
#pragma once
#include "kernels.h"

// ---- passes 1b / 2b ---------------------------------------------------------------------------------------
template <bool BY_BASE>
__global__ void __launch_bounds__(256) k_insert_marked_long(ReadsDev R, KParams K, FiltDev F, const uint64_t *mask,
                                                             uint64_t mask_words, const uint64_t *kofs,
                                                             unsigned long long *inserted) {
    constexpr int NW = 8;
    using S = Stage<NW>;
    __shared__ uint32_t lds[4][2 * S::WORDS];
    const int lane = threadIdx.x & 63;
    uint32_t *L32 = lds[threadIdx.x >> 6];
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const int k = K.k;
    unsigned long long mine = 0;
    for (uint64_t r = wave; r < R.n_reads; r += n_waves) {
        uint64_t off; uint32_t len;
        read_span(R, r, off, len);
        const int nk = (int)len - k + 1;
        if (nk <= 0) continue;
        const uint64_t kb0 = BY_BASE ? off : kmer_base(kofs, r, R.read_len, k);
        for (int w0 = 0; w0 < nk; w0 += NW * 64) {
            const uint64_t woff = off + (uint64_t)w0, wkb = kb0 + (uint64_t)w0;
            const int wnk = min(NW * 64, nk - w0);
            const uint64_t word = stage_fetch<NW>(R, nullptr, mask, wkb, mask_words - 1, woff, lane);
            __builtin_amdgcn_wave_barrier();
            if (lane < S::WORDS) stage_store(L32, lane, word);
            __builtin_amdgcn_wave_barrier();
            const int o31 = (int)(woff & 31), o63 = (int)(woff & 63), x63 = (int)(wkb & 63);
#pragma unroll 1
            for (int c = 0; c * 64 < wnk; ++c) {
                const int s = c * 64 + lane;
                bool take = false;
                if (s < wnk && lds_bit(L32 + 2 * S::X, x63 + s)) {
                    const bool valid = (lds_window32(L32 + 2 * S::M, o63 + s) & K.nmask_bits) == 0;
                    take = BY_BASE || valid;
                }
                if (take) {
                    const uint64_t key = canon_key(lds_window64(L32 + 2 * S::B, 2 * (o31 + s)), K);
                    bloom_put(F, block_of(F, key), pattern_of(F, key));
                }
                const unsigned long long bal = __ballot(take);
                if (!BY_BASE && R.hint_sampled) or_bits64(R.hint_sampled, woff + (uint64_t)c * 64, bal, lane);
                mine += __popcll(bal);
            }
        }
    }
    if (inserted && lane == 0 && mine) atomicAdd(inserted, mine);
}

// ---- pass 2a ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_infer_long(ReadsDev R, KParams K, FiltDev S, Thresholds thr, uint32_t *take_bits,
                                                     unsigned long long *inserted, uint32_t *err_out, uint32_t *qpresent,
                                                     unsigned long long *lookups) {
    constexpr int NW = 8;
    using St = Stage<NW>;
    __shared__ uint32_t lds[4][St::LDS_U32];
    __shared__ int thr_lds[KBBQ_MAX_KMER + 1];
    __shared__ uint32_t qseen[8];                  // quality values this block has met (k_infer)
    const int lane = threadIdx.x & 63;
    if (threadIdx.x <= KBBQ_MAX_KMER) thr_lds[threadIdx.x] = thr.v[threadIdx.x];
    if (threadIdx.x < 8) qseen[threadIdx.x] = 0;
    __syncthreads();
    uint32_t *L32 = lds[threadIdx.x >> 6];
    uint32_t *PW = L32 + 2 * St::WORDS;            // present bits: dword 0 = 0, dwords 1..2NW, then zeros
    uint32_t *EW = PW + St::RES;                   // error bits, same shape
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t *hint = reinterpret_cast<const uint64_t *>(R.hint_sampled);
    const int k = K.k;
    const int W = NW * 64 - (k - 1);               // k-mer starts per window: its bases fill the staged 512
    unsigned long long mine = 0, looked = 0;
    if (lane < St::RES) { PW[lane] = 0; EW[lane] = 0; }
    for (uint64_t r = wave; r < R.n_reads; r += n_waves) {
        uint64_t off; uint32_t len;
        read_span(R, r, off, len);
        const int Lr = (int)len, nk = Lr - k + 1;
        if (nk <= 0) continue;
        for (int a = 0;;) {
            const int b = min(nk, a + W);
            const int wnk = b - a, wlen = wnk + k - 1;            // this window's k-mer starts and bases (<= 512)
            // what this window decides: flags of the window's bases [e0, e1], insert decisions of its starts [e0, t1]
            const int e0 = a == 0 ? 0 : k - 1;
            const int e1 = b == nk ? wlen - 1 : wnk - 1;
            const int t1 = b == nk ? wnk - 1 : wnk - k;
            const uint64_t woff = off + (uint64_t)a;
            const uint64_t word = stage_fetch<NW>(R, hint, nullptr, 0, 0, woff, lane);
            __builtin_amdgcn_wave_barrier();
            if (lane < St::WORDS) stage_store(L32, lane, word);
            __builtin_amdgcn_wave_barrier();
            const int o31 = (int)(woff & 31), o63 = (int)(woff & 63);
            uint64_t V[NW];
            uint8_t q[NW];
#pragma unroll
            for (int c = 0; c < NW; ++c) {
                V[c] = 0;
                q[c] = 0;
                const int s = c * 64 + lane;
                if (c * 64 < wlen && s < wlen) {
                    q[c] = R.qual[woff + s];
                    qseen_note(qseen, q[c]);
                }
                bool valid = false, present = false;
                if (c * 64 < wnk && s < wnk) {
                    const uint64_t key = canon_key(lds_window64(L32 + 2 * St::B, 2 * (o31 + s)), K);
                    valid = (lds_window32(L32 + 2 * St::M, o63 + s) & K.nmask_bits) == 0;
                    const bool known = hint && lds_bit(L32 + 2 * St::H, o63 + s);
                    present = known;
                    if (valid && !known) {
                        present = bloom_has(S, key);
                        ++looked;
                    }
                }
                if (c * 64 < wnk) {
                    const uint64_t P = __ballot(present);
                    V[c] = __ballot(valid);
                    if (lane < 2) PW[1 + 2 * c + lane] = (uint32_t)(P >> (32 * lane));
                } else if (lane < 2) {
                    PW[1 + 2 * c + lane] = 0;
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < NW; ++c) {
                if (c * 64 < wlen) {
                    const int i = c * 64 + lane;          // base of the window; a + i in the read
                    bool err = false;
                    if (i < wlen) {
                        const int ia = a + i;
                        const int possible = min(ia, nk - 1) - max(0, ia - k + 1) + 1;
                        const int in = __popc(lds_window32(PW, i - k + 1 + 32) & K.nmask_bits);
                        err = in <= thr_lds[possible] || q[c] <= 2;
                    }
                    const uint64_t E = __ballot(err);
                    if (lane < 2) EW[1 + 2 * c + lane] = (uint32_t)(E >> (32 * lane));
                    if (err_out) {      // only the bases this window decides
                        const uint64_t own = __ballot(i >= e0 && i <= e1);
                        or_bits64(err_out, woff + (uint64_t)c * 64, E & own, lane);
                    }
                } else if (lane < 2) {
                    EW[1 + 2 * c + lane] = 0;
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < NW; ++c) {
                if (c * 64 < wnk) {
                    const int s = c * 64 + lane;
                    const bool take = s >= e0 && s <= t1 && ((V[c] >> lane) & 1) && (lds_window32(EW, s + 32) & K.nmask_bits) == 0;
                    const unsigned long long bal = __ballot(take);
                    or_bits64(take_bits, woff + (uint64_t)c * 64, bal, lane);
                    mine += __popcll(bal);
                }
            }
            if (b == nk) break;
            a = b - 2 * k + 2;
        }
    }
    // (per-lane counts of the lookups: summed over the wave here)
    for (int o = 32; o > 0; o >>= 1) looked += __shfl_down(looked, o);
    if (lane == 0 && mine) atomicAdd(inserted, mine);
    if (lane == 0 && looked) atomicAdd(lookups, looked);
    __syncthreads();
    if (threadIdx.x < 8 && qseen[threadIdx.x]) atomicOr(&qpresent[threadIdx.x], qseen[threadIdx.x]);
}

// ---- pass 3a ----------------------------------------------------------------------------------------------
// tw = words of trusted mask per read (a multiple of NW: every window writes NW words)
__global__ void __launch_bounds__(256) k_scan_trusted_long(ReadsDev R, KParams K, FiltDev T, uint64_t *tmask, int tw,
                                                            uint8_t *dirty) {
    constexpr int NW = 8;
    using S = Stage<NW>;
    __shared__ uint32_t lds[4][2 * S::WORDS];
    const int lane = threadIdx.x & 63;
    uint32_t *L32 = lds[threadIdx.x >> 6];
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t *hint = reinterpret_cast<const uint64_t *>(R.hint_trusted);
    const uint64_t oc_max = R.n_bases / 64 + 1;
    const int k = K.k;
    for (uint64_t r = wave; r < R.n_reads; r += n_waves) {
        uint64_t off; uint32_t len;
        read_span(R, r, off, len);
        const int nk = (int)len - k + 1;
        if (nk <= 0) { if (lane == 0) dirty[r] = 0; continue; }
        int trusted = 0;
        bool odd = false;
        for (int w0 = 0; w0 < nk; w0 += NW * 64) {
            const uint64_t woff = off + (uint64_t)w0;
            const int wnk = min(NW * 64, nk - w0);
            const uint64_t word = stage_fetch<NW>(R, hint, R.offcase, woff, oc_max, woff, lane);
            __builtin_amdgcn_wave_barrier();
            if (lane < S::WORDS) stage_store(L32, lane, word);
            __builtin_amdgcn_wave_barrier();
            const int o31 = (int)(woff & 31), o63 = (int)(woff & 63);
            uint64_t M[NW];
#pragma unroll
            for (int c = 0; c < NW; ++c) {
                M[c] = 0;
                if (c * 64 < wnk) {
                    const int s = c * 64 + lane;
                    bool ok = false;
                    if (s < wnk) {
                        const bool valid = (lds_window32(L32 + 2 * S::M, o63 + s) & K.nmask_bits) == 0;
                        const bool known = hint && lds_bit(L32 + 2 * S::H, o63 + s);
                        ok = known;
                        if (valid && !known) ok = bloom_has(T, canon_key(lds_window64(L32 + 2 * S::B, 2 * (o31 + s)), K));
                    }
                    M[c] = __ballot(ok);
                    trusted += __popcll(M[c]);
                }
            }
            if (R.offcase) {      // off-case bases among the (up to 512) bases staged for this window
                const int wb = min(NW * 64, (int)len - w0);
                for (int c = 0; c * 64 < wb; ++c) {
                    const int s = c * 64 + lane;
                    odd = odd || __ballot(s < wb && lds_bit(L32 + 2 * S::X, o63 + s)) != 0;
                }
            }
            if (lane < NW) tmask[r * (uint64_t)tw + (uint64_t)(w0 / 64) + lane] = sel_word<NW>(M, lane);
        }
        if (R.offcase) {          // the bases behind the last window's 512 (fewer than k)
            const int seen = ((nk + NW * 64 - 1) / (NW * 64)) * (NW * 64);
            bool tail = false;
            for (int s = seen + lane; s < (int)len; s += 64) {
                const uint64_t g = off + (uint64_t)s;
                tail = tail || ((R.offcase[g >> 6] >> (g & 63)) & 1);
            }
            odd = odd || __ballot(tail) != 0;
        }
        if (lane == 0) dirty[r] = (uint8_t)(trusted != nk ? (odd ? 3 : 1) : 0);
    }
}
