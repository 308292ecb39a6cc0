// bam_reader.h -- host side of kbbq_bam_reader (include/kbbq_bgzf.h): buffers, the chunk loop and the launches of
// bam_device.h's kernels.  Included once, at the end of bgzf_device.hip (it uses that file's Buf, Submission, device_scan,
// begin_submission and launch_deflate).
#pragma once

struct kbbq_bam_reader {
    Preload pre;                            // pieces of the file copied ahead of their chunk call (kbbq_bam_reader_preload)
    int device = 0;
    hipStream_t st = nullptr;
    hipEvent_t t0 = nullptr, t1 = nullptr, t2 = nullptr;
    int use_oq = 0;
    int32_t n_ref = 0;
    uint64_t header_left = 0;               // bytes of the BAM header still to skip at the front of the stream
    uint64_t header_bytes = 0;
    Buf comp, text, status, blk_meta, h_meta, h_small, carry, tile_sums;
    Buf seg_u32, seg_slots, seg_counts;     // BamSegs
    Buf idx_u32, idx_u16, idx_u64;          // BamIndex
    Buf d_out;                              // small device words: [0..1] chain flags, [4..6] record flags / longest / shortest
    Buf rg_ids, rg_off, rg_hash, first_seen, dense;  // the header's @RG table (+ hash slots), first appearance per chunk, table index -> dense index
    uint32_t rg_hash_mask = 0;
    Buf seq_text, counter;                  // scratch of kbbq_bam_reader_batch
    std::vector<uint8_t> h_ids;
    std::vector<uint32_t> h_id_off;
    std::vector<int32_t> dense_of;          // table index -> dense read-group index (first appearance, readutils.cc:53-57), -1: not met
    std::vector<uint32_t> order;            // dense index -> table index
    unsigned inflate_grid = 0;
    uint64_t carry_bytes = 0;
    // the current chunk
    uint64_t text_bytes = 0, n_records = 0, n_bases = 0, idx_cap = 0;
    uint32_t longest = 0, shortest = 0, chunk_flags = 0;
    bool have_chunk = false;
    double ms_inflate = 0, ms_index = 0, ms_rewrite = 0;
    // chunks of the first scan kept for pass 4: the COMPRESSED bytes (a third of the stream) with their block table and the
    // bytes the chunk before them left over; pass 4 inflates and indexes them again (kbbq_bam_reader_select)
    struct Kept {
        Buf comp, carry;
        std::vector<uint64_t> c_off, o_off;
        std::vector<uint32_t> c_len, o_len;
        uint64_t carry_bytes = 0, skip = 0, text = 0, n_records = 0;
    };
    std::vector<Kept> kept;
    bool keeping = false;
    uint64_t kept_bytes = 0;
};

namespace {

void bam_release_kept(kbbq_bam_reader *r) {
    for (auto &k : r->kept) { k.comp.release(); k.carry.release(); }
    r->kept.clear();
    r->kept_bytes = 0;
}

BamSegs bam_segs(kbbq_bam_reader *r, uint32_t n_segs) {
    BamSegs G;
    uint32_t *u = (uint32_t *)r->seg_u32.p;
    G.start = u; G.land = u + n_segs; G.count = u + 2 * (size_t)n_segs; G.bad = u + 3 * (size_t)n_segs;
    G.slots = (uint32_t *)r->seg_slots.p;
    G.n_segs = n_segs;
    return G;
}

BamIndex bam_index(kbbq_bam_reader *r) {
    BamIndex X;
    const size_t cap = r->idx_cap;
    uint32_t *u = (uint32_t *)r->idx_u32.p;
    X.rec_off = u; X.seq_off = u + cap; X.qual_off = u + 2 * cap; X.qsrc_off = u + 3 * cap; X.l_seq = u + 4 * cap;
    X.oq_at = u + 5 * cap; X.oq_vlen = u + 6 * cap;
    uint16_t *h = (uint16_t *)r->idx_u16.p;
    X.flag = h; X.rg = h + cap;
    uint64_t *q = (uint64_t *)r->idx_u64.p;
    X.base_sz = q; X.out_sz = q + (cap + 2);
    return X;
}

// the BGZF blocks at the front of file_bytes: where their DEFLATE streams lie and where their bytes go; false: not BGZF
bool bam_parse_blocks(const uint8_t *file_bytes, uint64_t n_bytes, uint64_t text0, std::vector<uint64_t> &c_off, std::vector<uint64_t> &o_off,
                      std::vector<uint32_t> &c_len, std::vector<uint32_t> &o_len, uint64_t *consumed, uint64_t *text_out) {
    uint64_t at = 0, text = text0;
    const uint64_t text_cap = 3500000000ull;      // record offsets travel in 32 bits
    while (at + 18 <= n_bytes) {
        const uint8_t *h = file_bytes + at;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return false;
        const uint32_t xlen = h[10] | (h[11] << 8);
        if (at + 12 + xlen > n_bytes) break;
        uint32_t bsize = 0;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            const uint8_t *sf = h + 12 + x;
            const uint32_t slen = sf[2] | (sf[3] << 8);
            if (sf[0] == 66 && sf[1] == 67 && slen == 2 && x + 6 <= xlen) bsize = (sf[4] | (sf[5] << 8)) + 1u;
            x += 4 + slen;
        }
        if (!bsize || bsize < 12 + xlen + 8) return false;
        if (at + bsize > n_bytes) break;      // the chunk ends inside this block
        const uint8_t *tail = h + bsize - 8;
        const uint32_t isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
        if (isize > 65536) return false;
        if (text + isize > text_cap) break;
        if (isize) {
            c_off.push_back(at + 12 + xlen);
            c_len.push_back(bsize - (12 + xlen) - 8);
            o_off.push_back(text);
            o_len.push_back(isize);
            text += isize;
        }
        at += bsize;
    }
    *consumed = at;
    *text_out = text;
    return true;
}

// inflate + checksum of the blocks described by the vectors, from the device copy `d_comp` of the compressed bytes into
// r->text (whose first carry bytes are in place); returns when every block's status has been read
int bam_inflate(kbbq_bam_reader *r, const void *d_comp, const std::vector<uint64_t> &c_off, const std::vector<uint64_t> &o_off,
                const std::vector<uint32_t> &c_len, const std::vector<uint32_t> &o_len, uint64_t text) {
    const uint32_t nb = (uint32_t)c_off.size();
    int rc;
    if ((rc = r->status.reserve((size_t)nb * 4 + 64))) return rc;
    const size_t meta_bytes = (size_t)nb * 24 + 64;
    if ((rc = r->blk_meta.reserve(meta_bytes))) return rc;
    if ((rc = r->h_meta.reserve(meta_bytes))) return rc;
    HIP_TRY(hipEventRecord(r->t0, r->st));
    if (nb) {
        uint64_t *hm = (uint64_t *)r->h_meta.p;
        memcpy(hm, c_off.data(), (size_t)nb * 8);
        memcpy(hm + nb, o_off.data(), (size_t)nb * 8);
        memcpy((uint32_t *)(hm + 2 * (size_t)nb), c_len.data(), (size_t)nb * 4);
        memcpy((uint32_t *)(hm + 2 * (size_t)nb) + nb, o_len.data(), (size_t)nb * 4);
        HIP_TRY(hipMemcpyAsync(r->blk_meta.p, hm, (size_t)nb * 24, hipMemcpyHostToDevice, r->st));
        InflateArgs A;
        A.comp = (const uint8_t *)d_comp;
        A.c_off = (const uint64_t *)r->blk_meta.p;
        A.o_off = A.c_off + nb;
        A.c_len = (const uint32_t *)(A.c_off + 2 * (size_t)nb);
        A.o_len = A.c_len + nb;
        A.out = (uint8_t *)r->text.p;
        A.n_blocks = nb;
        A.status = (uint32_t *)r->status.p;
        if (!r->inflate_grid && (rc = inflate_resident_waves(r->device, &r->inflate_grid))) return rc;
        hipLaunchKernelGGL(k_inflate, dim3(std::min<unsigned>(nb, r->inflate_grid)), dim3(64 * INF_WAVES), 0, r->st, A);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_block_crc, dim3(std::min<unsigned>((nb + 3) / 4, 256 * 16)), dim3(256), 0, r->st, A);      // as bgzf_read verifies them
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemsetAsync((char *)r->text.p + text, 0, 64, r->st));
    HIP_TRY(hipEventRecord(r->t1, r->st));
    if (nb) {
        std::vector<uint32_t> stt(nb);
        HIP_TRY(hipMemcpyAsync(stt.data(), r->status.p, (size_t)nb * 4, hipMemcpyDeviceToHost, r->st));
        HIP_TRY(hipStreamSynchronize(r->st));
        for (uint32_t b = 0; b < nb; ++b) {
            if (stt[b] == INF_BAD_CRC) return fail(KBBQ_EIO, "BGZF block %u of the chunk: CRC32 checksum mismatch", b);
            if (stt[b] != INF_OK) return fail(KBBQ_EIO, "BGZF block %u of the chunk does not inflate (code %u)", b, stt[b]);
        }
    }
    return KBBQ_OK;
}

// The records of the stream r->text[0, text): chain, index, fields.  skip: bytes in front of the first record (the BAM
// header in the first chunk).  Fills the current-chunk fields of r and info; leaves what the stream's end cut in r->carry.
int bam_index_stream(kbbq_bam_reader *r, uint64_t text, uint64_t skip, int32_t last, bool assign_groups, kbbq_bam_chunk *info) {
    int rc;
    r->n_records = 0; r->n_bases = 0; r->longest = r->shortest = 0;
    if ((rc = r->h_small.reserve(4096))) return rc;
    if ((rc = r->d_out.reserve(64))) return rc;
    uint64_t rec_end = skip;
    const uint8_t *t = (const uint8_t *)r->text.p;
    uint32_t *out = (uint32_t *)r->d_out.p;
    if (text > skip) {
        // The chain begins at `skip` (behind the BAM header in the first chunk): the kernels see the stream from the
        // aligned offset below it, so that segment 0 starts at a known place however long the header is.
        const uint64_t bias = skip & ~3ull;
        const uint8_t *tb = t + bias;
        const uint64_t nb = text - bias;
        const uint32_t n_segs = (uint32_t)((nb + BAM_SEG - 1) / BAM_SEG);
        if ((rc = r->seg_u32.reserve((size_t)n_segs * 16 + 64))) return rc;
        if ((rc = r->seg_slots.reserve((size_t)n_segs * BAM_SEG_SLOTS * 4 + 64))) return rc;
        if ((rc = r->seg_counts.reserve(((size_t)n_segs + 2) * 8))) return rc;
        const BamSegs G = bam_segs(r, n_segs);
        uint32_t *hs = (uint32_t *)r->h_small.p;
        const uint32_t init_out[8] = {0, 0, 0, 0, 0, 0, 0xFFFFFFFFu, 0};
        memcpy(hs, init_out, sizeof init_out);
        hs[8] = (uint32_t)(skip - bias);
        HIP_TRY(hipMemcpyAsync(out, hs, sizeof init_out, hipMemcpyHostToDevice, r->st));
        HIP_TRY(hipMemcpyAsync(G.start, hs + 8, 4, hipMemcpyHostToDevice, r->st));      // segment 0 starts where the caller says
        if (n_segs > 1) hipLaunchKernelGGL(k_bam_seg_guess, dim3((n_segs + 2) / 4), dim3(256), 0, r->st, tb, nb, r->n_ref, G);
        hipLaunchKernelGGL(k_bam_seg_walk, dim3((n_segs + 255) / 256), dim3(256), 0, r->st, tb, nb, G);
        HIP_TRY(hipGetLastError());
        if (n_segs > 1) {
            hipLaunchKernelGGL(k_bam_seg_check, dim3((n_segs + 254) / 256), dim3(256), 0, r->st, G, out);
            hipLaunchKernelGGL(k_bam_seg_repair, dim3(1), dim3(64), 0, r->st, tb, nb, G, out, 4096u);
            HIP_TRY(hipGetLastError());
        }
        uint64_t *counts = (uint64_t *)r->seg_counts.p;
        hipLaunchKernelGGL(k_bam_seg_counts, dim3((n_segs + 255) / 256), dim3(256), 0, r->st, G, counts, out);
        HIP_TRY(hipGetLastError());
        if ((rc = device_scan_on(r->tile_sums, r->st, counts, n_segs, counts + n_segs))) return rc;
        HIP_TRY(hipMemcpyAsync(hs + 16, counts + n_segs, 8, hipMemcpyDeviceToHost, r->st));
        HIP_TRY(hipMemcpyAsync(hs + 20, out, 8, hipMemcpyDeviceToHost, r->st));
        HIP_TRY(hipMemcpyAsync(hs + 24, G.land + (n_segs - 1), 4, hipMemcpyDeviceToHost, r->st));
        HIP_TRY(hipStreamSynchronize(r->st));
        const uint64_t n_rec = *(const uint64_t *)(hs + 16);
        const uint32_t chain_flags = hs[20];
        rec_end = hs[24] == BAM_NONE ? text + 1 : (uint64_t)hs[24] + bias;
        if (chain_flags & 6) info->flags |= BAMF_FALLBACK;      // too many repairs, or a malformed block: the host parsers' case
        if (rec_end > text) { info->flags |= BAMF_FALLBACK; rec_end = text; }
        if (n_rec && !(info->flags & BAMF_FALLBACK)) {
            if (r->idx_cap < n_rec) {
                const size_t cap = n_rec + n_rec / 8 + 1024;
                r->idx_cap = 0;
                if ((rc = r->idx_u32.reserve(cap * 7 * 4))) return rc;
                if ((rc = r->idx_u16.reserve(cap * 2 * 2))) return rc;
                if ((rc = r->idx_u64.reserve((cap + 2) * 2 * 8))) return rc;
                r->idx_cap = cap;
            }
            const BamIndex X = bam_index(r);
            hipLaunchKernelGGL(k_bam_rec_offsets, dim3(n_segs), dim3(256), 0, r->st, G, (const uint64_t *)counts, (uint32_t)bias, X.rec_off);
            BamRgTable T;
            T.ids = (const uint8_t *)r->rg_ids.p; T.id_off = (const uint32_t *)r->rg_off.p; T.n_ids = (uint32_t)r->dense_of.size();
            T.hash_slots = (const uint16_t *)r->rg_hash.p; T.hash_mask = r->rg_hash_mask;
            hipLaunchKernelGGL(k_bam_records, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, r->st, t, n_rec, r->use_oq, T, X, out + 4,
                               (unsigned long long *)r->first_seen.p);
            HIP_TRY(hipGetLastError());
            if ((rc = device_scan_on(r->tile_sums, r->st, X.base_sz, n_rec, X.base_sz + n_rec))) return rc;
            HIP_TRY(hipMemcpyAsync(hs + 32, X.base_sz + n_rec, 8, hipMemcpyDeviceToHost, r->st));
            HIP_TRY(hipMemcpyAsync(hs + 36, out + 4, 12, hipMemcpyDeviceToHost, r->st));
            const size_t n_ids = r->dense_of.size();
            std::vector<unsigned long long> seen(n_ids);
            if (assign_groups && n_ids) HIP_TRY(hipMemcpyAsync(seen.data(), r->first_seen.p, n_ids * 8, hipMemcpyDeviceToHost, r->st));
            HIP_TRY(hipStreamSynchronize(r->st));
            r->n_records = n_rec;
            r->n_bases = *(const uint64_t *)(hs + 32);
            info->flags |= hs[36];
            r->longest = hs[37];
            r->shortest = hs[38];
            if (assign_groups && n_ids) {
                // read groups in the order their first records appear (rg_to_int[rg] = rg_to_int.size(), readutils.cc:53-57)
                std::vector<std::pair<unsigned long long, uint32_t>> fresh;
                for (size_t i = 0; i < n_ids; ++i)
                    if (seen[i] != ~0ull && r->dense_of[i] < 0) fresh.emplace_back(seen[i], (uint32_t)i);
                std::sort(fresh.begin(), fresh.end());
                for (auto &f : fresh) { r->dense_of[f.second] = (int32_t)r->order.size(); r->order.push_back(f.second); }
                if (!fresh.empty()) {
                    std::vector<uint16_t> dn(n_ids);
                    for (size_t i = 0; i < n_ids; ++i) dn[i] = (uint16_t)(r->dense_of[i] < 0 ? 0 : r->dense_of[i]);
                    HIP_TRY(hipMemcpy(r->dense.p, dn.data(), n_ids * 2, hipMemcpyHostToDevice));
                }
                HIP_TRY(hipMemsetAsync(r->first_seen.p, 0xFF, n_ids * 8, r->st));
            }
        }
    }
    HIP_TRY(hipEventRecord(r->t2, r->st));
    // ---- what the chunk's end cut: kept for the next chunk
    const uint64_t left = text - rec_end;
    if (left) {
        if (last) info->flags |= BAMF_TRUNCATED;
        if ((rc = r->carry.reserve(left + 64))) return rc;
        HIP_TRY(hipMemcpyAsync(r->carry.p, (const char *)r->text.p + rec_end, left, hipMemcpyDeviceToDevice, r->st));
    }
    HIP_TRY(hipStreamSynchronize(r->st));
    r->carry_bytes = left;
    r->text_bytes = text;
    r->have_chunk = true;
    r->chunk_flags = info->flags;
    info->n_records = r->n_records;
    info->n_bases = r->n_bases;
    info->longest = r->n_records ? r->longest : 0;
    info->shortest = r->n_records ? r->shortest : 0;
    float a = 0, b = 0;
    if (hipEventElapsedTime(&a, r->t0, r->t1) == hipSuccess) r->ms_inflate += a;
    if (hipEventElapsedTime(&b, r->t1, r->t2) == hipSuccess) r->ms_index += b;
    return KBBQ_OK;
}

}  // namespace

extern "C" {

void kbbq_bam_reader_destroy(kbbq_bam_reader *r) {
    if (!r) return;
    KbbqDeviceGuard guard(r->device);
    if (r->st) (void)hipStreamSynchronize(r->st);
    Buf *all[] = {&r->comp, &r->text, &r->status, &r->blk_meta, &r->h_meta, &r->h_small, &r->carry, &r->tile_sums, &r->seg_u32, &r->seg_slots,
                  &r->seg_counts, &r->idx_u32, &r->idx_u16, &r->idx_u64, &r->d_out, &r->rg_ids, &r->rg_off, &r->rg_hash, &r->first_seen, &r->dense, &r->seq_text,
                  &r->counter};
    for (Buf *b : all) b->release();
    r->pre.release();
    bam_release_kept(r);
    hipEvent_t evs[] = {r->t0, r->t1, r->t2};
    for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    if (r->st) (void)hipStreamDestroy(r->st);
    delete r;
}

int kbbq_bam_reader_create(int32_t device, int32_t use_oq, int32_t n_ref, uint64_t header_bytes, const char *const *rg_ids, uint32_t n_rg_ids,
                           kbbq_bam_reader **out) {
    if (!out || (n_rg_ids && !rg_ids)) return fail(KBBQ_EINVAL, "null argument");
    if (n_rg_ids > 65535) return fail(KBBQ_ERANGE, "%u @RG lines: read-group indices travel in 16 bits", n_rg_ids);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(KBBQ_ENODEV, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(KBBQ_ENODEV, "device %d of %d", device, ndev);
    KbbqDeviceGuard guard(device);
    HIP_TRY(guard.err);
    kbbq_bam_reader *r = new kbbq_bam_reader;
    r->device = device;
    r->use_oq = use_oq ? 1 : 0;
    r->n_ref = n_ref;
    r->header_bytes = r->header_left = header_bytes;
    r->h_meta.host = r->h_small.host = true;
    r->h_id_off.push_back(0);
    for (uint32_t i = 0; i < n_rg_ids; ++i) {
        const char *s = rg_ids[i] ? rg_ids[i] : "";
        r->h_ids.insert(r->h_ids.end(), s, s + strlen(s));
        r->h_id_off.push_back((uint32_t)r->h_ids.size());
    }
    r->dense_of.assign(n_rg_ids, -1);
    hipError_t he = hipStreamCreateWithFlags(&r->st, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipEventCreate(&r->t0);
    if (he == hipSuccess) he = hipEventCreate(&r->t1);
    if (he == hipSuccess) he = hipEventCreate(&r->t2);
    int rc = KBBQ_OK;
    if (he == hipSuccess) {
        if (!(rc = r->rg_ids.reserve(r->h_ids.size() + 64)) && !(rc = r->rg_off.reserve(r->h_id_off.size() * 4 + 64)) &&
            !(rc = r->first_seen.reserve((size_t)n_rg_ids * 8 + 64)) && !(rc = r->dense.reserve((size_t)n_rg_ids * 2 + 64))) {
            if (!r->h_ids.empty()) he = hipMemcpy(r->rg_ids.p, r->h_ids.data(), r->h_ids.size(), hipMemcpyHostToDevice);
            if (he == hipSuccess) he = hipMemcpy(r->rg_off.p, r->h_id_off.data(), r->h_id_off.size() * 4, hipMemcpyHostToDevice);
            if (he == hipSuccess) he = hipMemset(r->first_seen.p, 0xFF, (size_t)n_rg_ids * 8 + 64);
            if (he == hipSuccess) he = hipMemset(r->dense.p, 0, (size_t)n_rg_ids * 2 + 64);
        }
        if (he == hipSuccess && !rc && n_rg_ids > 8) {
            // open addressing over the ids' hashes, at most half full (an id listed twice keeps its first index: the first wins
            // a linear comparison as well)
            uint32_t size = 16;
            while (size < 2 * n_rg_ids) size *= 2;
            std::vector<uint16_t> slots(size, 0xFFFF);
            for (uint32_t i = 0; i < n_rg_ids; ++i) {
                uint32_t h = 2166136261u;
                for (uint32_t j = r->h_id_off[i]; j < r->h_id_off[i + 1]; ++j) h = bam_fnv1a(h, r->h_ids[j]);
                uint32_t at = h & (size - 1);
                bool dup = false;
                while (slots[at] != 0xFFFF) {
                    const uint32_t o = slots[at];
                    const uint32_t la = r->h_id_off[o + 1] - r->h_id_off[o], lb = r->h_id_off[i + 1] - r->h_id_off[i];
                    if (la == lb && !memcmp(&r->h_ids[r->h_id_off[o]], &r->h_ids[r->h_id_off[i]], la)) { dup = true; break; }
                    at = (at + 1) & (size - 1);
                }
                if (!dup) slots[at] = (uint16_t)i;
            }
            if (!(rc = r->rg_hash.reserve((size_t)size * 2 + 64))) {
                he = hipMemcpy(r->rg_hash.p, slots.data(), (size_t)size * 2, hipMemcpyHostToDevice);
                r->rg_hash_mask = size - 1;
            }
        }
    }
    if (he != hipSuccess || rc) {
        kbbq_bam_reader_destroy(r);
        return rc ? rc : fail(KBBQ_EIO, "creating the BAM reader: %s", hipGetErrorString(he));
    }
    *out = r;
    return KBBQ_OK;
}

int kbbq_bam_reader_rewind(kbbq_bam_reader *r) {
    if (!r) return fail(KBBQ_EINVAL, "null argument");
    r->keeping = false;      // what was kept stays; a second scan keeps nothing more
    r->carry_bytes = 0;
    r->header_left = r->header_bytes;
    r->have_chunk = false;
    return KBBQ_OK;
}

int kbbq_bam_reader_keep(kbbq_bam_reader *r, int32_t on) {
    if (!r) return fail(KBBQ_EINVAL, "null argument");
    KbbqDeviceGuard guard(r->device);
    HIP_TRY(guard.err);
    if (on) {
        if (r->have_chunk || !r->kept.empty()) return fail(KBBQ_ESTATE, "keeping starts before the first chunk of a scan");
        r->keeping = true;
    } else {
        HIP_TRY(hipStreamSynchronize(r->st));
        bam_release_kept(r);
        r->keeping = false;
    }
    return KBBQ_OK;
}

int kbbq_bam_reader_kept(kbbq_bam_reader *r, uint64_t *n_chunks, uint64_t *n_bytes) {
    if (!r) return fail(KBBQ_EINVAL, "null argument");
    if (n_chunks) *n_chunks = r->kept.size();
    if (n_bytes) *n_bytes = r->kept_bytes;
    return KBBQ_OK;
}

int kbbq_bam_reader_read_groups(kbbq_bam_reader *r, uint32_t *table_index, uint32_t capacity, uint32_t *n) {
    if (!r || !n) return fail(KBBQ_EINVAL, "null argument");
    *n = (uint32_t)r->order.size();
    for (uint32_t i = 0; i < *n && i < capacity && table_index; ++i) table_index[i] = r->order[i];
    return KBBQ_OK;
}

int kbbq_bam_reader_chunk(kbbq_bam_reader *r, const uint8_t *file_bytes, uint64_t n_bytes, int32_t last, kbbq_bam_chunk *info) {
    if (!r || !info || (!file_bytes && n_bytes)) return fail(KBBQ_EINVAL, "bad argument");
    KbbqDeviceGuard guard(r->device);
    HIP_TRY(guard.err);
    memset(info, 0, sizeof *info);
    r->have_chunk = false;
    std::vector<uint64_t> c_off, o_off;
    std::vector<uint32_t> c_len, o_len;
    uint64_t at = 0, text = r->carry_bytes;
    if (!bam_parse_blocks(file_bytes, n_bytes, r->carry_bytes, c_off, o_off, c_len, o_len, &at, &text)) { info->flags |= BAMF_FALLBACK; return KBBQ_OK; }
    info->consumed = at;
    info->n_blocks = (uint32_t)c_off.size();
    if (at == 0 && n_bytes && !last && c_off.empty()) return fail(KBBQ_EINVAL, "the chunk holds no complete BGZF block");
    int rc;
    auto reserve_or_drop = [&](Buf &b, size_t need) -> int {
        int rc2 = b.reserve(need);
        if (rc2 == KBBQ_ENOMEM && (r->keeping || !r->kept.empty())) {
            (void)hipGetLastError();
            bam_release_kept(r);
            r->keeping = false;
            rc2 = b.reserve(need);
        }
        return rc2;
    };
    void *d_comp = r->pre.take(file_bytes, n_bytes, r->st);      // copied ahead by the caller's I/O thread?
    if (!d_comp && (rc = reserve_or_drop(r->comp, at + 4096))) return rc;
    if ((rc = reserve_or_drop(r->text, text + 4096))) return rc;
    const uint64_t carry_in = r->carry_bytes;
    if (carry_in) HIP_TRY(hipMemcpyAsync(r->text.p, r->carry.p, carry_in, hipMemcpyDeviceToDevice, r->st));
    if (!d_comp) {
        d_comp = r->comp.p;
        if (at) {
            HIP_TRY(hipMemcpyAsync(d_comp, file_bytes, at, hipMemcpyHostToDevice, r->st));
            HIP_TRY(hipMemsetAsync((char *)d_comp + at, 0, 4096, r->st));
        }
    }
    // the header's bytes come first in the stream; a header longer than this chunk's stream is not this path's case
    uint64_t skip = 0;
    if (r->header_left) {
        if (r->header_left > text) { info->flags |= BAMF_FALLBACK; return KBBQ_OK; }
        skip = r->header_left;
    }
    // kept for pass 4 (before the carry buffer is overwritten below): compressed bytes, block table, the bytes carried in
    kbbq_bam_reader::Kept k;
    bool keep_this = r->keeping && at;
    if (keep_this) {
        k.comp.exact = k.carry.exact = true;
        if (k.comp.reserve(at + 4096) || (carry_in && k.carry.reserve(carry_in + 64))) {
            (void)hipGetLastError();
            k.comp.release(); k.carry.release();
            bam_release_kept(r);
            r->keeping = false;
            keep_this = false;
        } else {
            HIP_TRY(hipMemcpyAsync(k.comp.p, d_comp, at, hipMemcpyDeviceToDevice, r->st));
            HIP_TRY(hipMemsetAsync((char *)k.comp.p + at, 0, 4096, r->st));
            if (carry_in) HIP_TRY(hipMemcpyAsync(k.carry.p, r->carry.p, carry_in, hipMemcpyDeviceToDevice, r->st));
        }
    }
    if ((rc = bam_inflate(r, d_comp, c_off, o_off, c_len, o_len, text))) { k.comp.release(); k.carry.release(); return rc; }
    info->text_bytes = text - carry_in;
    rc = bam_index_stream(r, text, skip, last, true, info);
    if (rc) { k.comp.release(); k.carry.release(); return rc; }
    r->header_left = 0;
    if (keep_this) {
        if (r->n_records && !(info->flags & (BAMF_FALLBACK | BAMF_TRUNCATED))) {
            k.c_off.swap(c_off); k.o_off.swap(o_off); k.c_len.swap(c_len); k.o_len.swap(o_len);
            k.carry_bytes = carry_in; k.skip = skip; k.text = text; k.n_records = r->n_records;
            r->kept_bytes += k.comp.bytes + k.carry.bytes;
            r->kept.push_back(std::move(k));
        } else {
            k.comp.release(); k.carry.release();
        }
    }
    return KBBQ_OK;
}

int kbbq_bam_reader_select(kbbq_bam_reader *r, uint64_t i, kbbq_bam_chunk *info) {
    if (!r) return fail(KBBQ_EINVAL, "null argument");
    KbbqDeviceGuard guard(r->device);
    HIP_TRY(guard.err);
    if (i >= r->kept.size()) return fail(KBBQ_EINVAL, "kept chunk %llu of %llu", (unsigned long long)i, (unsigned long long)r->kept.size());
    const kbbq_bam_reader::Kept &k = r->kept[(size_t)i];
    kbbq_bam_chunk local;
    if (!info) info = &local;
    memset(info, 0, sizeof *info);
    r->have_chunk = false;
    int rc;
    if ((rc = r->text.reserve(k.text + 4096))) return rc;
    if (k.carry_bytes) HIP_TRY(hipMemcpyAsync(r->text.p, k.carry.p, k.carry_bytes, hipMemcpyDeviceToDevice, r->st));
    if ((rc = bam_inflate(r, k.comp.p, k.c_off, k.o_off, k.c_len, k.o_len, k.text))) return rc;
    info->n_blocks = (uint32_t)k.c_off.size();
    info->text_bytes = k.text - k.carry_bytes;
    if ((rc = bam_index_stream(r, k.text, k.skip, 0, false, info))) return rc;
    if (r->n_records != k.n_records) return fail(KBBQ_EIO, "kept chunk %llu: %llu records where the scan found %llu", (unsigned long long)i,
                                                 (unsigned long long)r->n_records, (unsigned long long)k.n_records);
    return KBBQ_OK;
}

int kbbq_bam_reader_batch(kbbq_bam_reader *r, kbbq_reads *dev) {
    if (!r || !dev) return fail(KBBQ_EINVAL, "null argument");
    if (!r->have_chunk || !r->n_records) return fail(KBBQ_ESTATE, "no records in the current chunk");
    if (r->chunk_flags & BAMF_FALLBACK) return fail(KBBQ_ESTATE, "the chunk holds a shape this reader does not take (flags %u): the host parser's", r->chunk_flags);
    KbbqDeviceGuard guard(r->device);
    HIP_TRY(guard.err);
    const uint64_t n = r->n_records, nbases = r->n_bases;
    const BamIndex X = bam_index(r);
    memset(dev, 0, sizeof *dev);
    dev->n_reads = n;
    dev->n_bases = nbases;
    dev->on_device = 1;
    void *b = nullptr, *m = nullptr, *q = nullptr, *off = nullptr, *fl = nullptr, *rg = nullptr;
    auto release = [&]() { void *all[] = {b, m, q, off, fl, rg}; for (void *x : all) (void)hipFree(x); };
    int rc0;
    const uint64_t words = nbases / 64 + 1;
    if ((rc0 = r->seq_text.reserve(nbases + 64))) return rc0;
    if ((rc0 = r->counter.reserve((words + 2) * 8 + 64))) return rc0;      // [0..1] counts, then the (always empty) off-case words
    void *seq_text = r->seq_text.p;
    unsigned long long *cnt = (unsigned long long *)r->counter.p;
#define RB_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { release(); return fail(_e == hipErrorOutOfMemory ? KBBQ_ENOMEM : KBBQ_EIO, "%s: %s", #expr, hipGetErrorString(_e)); } } while (0)
    RB_TRY(hipMalloc(&b, (2 * words + 2) * 8));
    RB_TRY(hipMalloc(&m, (words + 2) * 8));
    RB_TRY(hipMalloc(&q, nbases + 16));
    RB_TRY(hipMalloc(&fl, n));
    RB_TRY(hipMalloc(&rg, n * 2 + 16));
    const bool uniform = r->longest == r->shortest;
    if (!uniform) RB_TRY(hipMalloc(&off, (n + 1) * 8));
    RB_TRY(hipMemsetAsync(cnt, 0, 16, r->st));
    RB_TRY(hipMemsetAsync((char *)b + 2 * words * 8, 0, 16, r->st));
    RB_TRY(hipMemsetAsync((char *)m + words * 8, 0, 16, r->st));
    RB_TRY(hipMemsetAsync((char *)q + nbases, 0, 16, r->st));
    hipLaunchKernelGGL(k_bam_gather, dim3((unsigned)std::min<uint64_t>((n + 3) / 4, 256 * 32)), dim3(256), 0, r->st, (const uint8_t *)r->text.p, X,
                       (const uint64_t *)X.base_sz, n, r->use_oq, (uint8_t *)seq_text, (uint8_t *)q);
    // (bam_seq_str gives upper-case letters only: no off-case bits; the words go to scratch)
    hipLaunchKernelGGL(k_pack_text, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, r->st, (const uint8_t *)seq_text, nbases, (uint64_t *)b,
                       (uint64_t *)m, (uint64_t *)(cnt + 2), cnt);
    hipLaunchKernelGGL(k_bam_read_meta, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, r->st, X, n, (const uint16_t *)r->dense.p, (uint8_t *)fl,
                       (uint16_t *)rg);
    RB_TRY(hipGetLastError());
    if (!uniform) RB_TRY(hipMemcpyAsync(off, X.base_sz, (n + 1) * 8, hipMemcpyDeviceToDevice, r->st));
    RB_TRY(hipStreamSynchronize(r->st));
#undef RB_TRY
    dev->bases = (const uint64_t *)b;
    dev->nmask = (const uint64_t *)m;
    dev->qual = (const uint8_t *)q;
    dev->offsets = (const uint64_t *)off;
    dev->flags = (const uint8_t *)fl;
    dev->rg = (const uint16_t *)rg;
    dev->read_len = uniform ? r->longest : 0;
    dev->offcase = nullptr;
    return KBBQ_OK;
}

int kbbq_bam_reader_write(kbbq_bam_reader *r, kbbq_bgzf *z, const uint8_t *d_qual, int32_t set_oq, void *after_stream) {
    if (!r || !z || !d_qual) return fail(KBBQ_EINVAL, "null argument");
    if (!r->have_chunk || !r->n_records) return fail(KBBQ_ESTATE, "no records in the current chunk");
    if (r->device != z->device) return fail(KBBQ_EINVAL, "reader and writer are on different devices");
    if (r->chunk_flags & BAMF_FALLBACK) return fail(KBBQ_ESTATE, "the chunk holds a shape this reader does not take (flags %u)", r->chunk_flags);
    if (set_oq && (r->chunk_flags & BAMF_OQ_UNWRITABLE)) return fail(KBBQ_EINVAL, "Tag data is corrupt: a record's OQ tag cannot be updated");
    KbbqDeviceGuard guard(z->device);
    HIP_TRY(guard.err);
    const uint64_t n = r->n_records;
    const BamIndex X = bam_index(r);
    int rc;
    // sizes of the rewritten records and where they go
    hipLaunchKernelGGL(k_bam_out_sizes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, r->st, (const uint8_t *)r->text.p, X, n, set_oq ? 1 : 0);
    HIP_TRY(hipGetLastError());
    if ((rc = device_scan_on(r->tile_sums, r->st, X.out_sz, n, X.out_sz + n))) return rc;
    uint64_t *hs = (uint64_t *)r->h_small.p;
    HIP_TRY(hipMemcpyAsync(hs, X.out_sz + n, 8, hipMemcpyDeviceToHost, r->st));
    HIP_TRY(hipStreamSynchronize(r->st));
    const uint64_t t = hs[0];
    Submission *sp;
    if ((rc = begin_submission(z, after_stream, &sp))) return rc;
    Submission &s = *sp;
    s.n = t;
    s.formatted = true;
    if ((rc = s.payload.reserve(t + 16))) return rc;
    HIP_TRY(hipMemsetAsync((char *)s.payload.p + t, 0, 16, z->st));
    HIP_TRY(hipEventRecord(s.t0, z->st));
    hipLaunchKernelGGL(k_bam_rewrite, dim3((unsigned)std::min<uint64_t>((n + 3) / 4, 256 * 32)), dim3(256), 0, z->st, (const uint8_t *)r->text.p, X,
                       (const uint64_t *)X.base_sz, (const uint64_t *)X.out_sz, n, set_oq ? 1 : 0, d_qual, (uint8_t *)s.payload.p);
    HIP_TRY(hipGetLastError());
    if ((rc = launch_deflate(z, s))) return rc;
    // the reader's stream and index are read by the kernel just queued: the next chunk must not overwrite them before it has run
    HIP_TRY(hipEventSynchronize(s.t1));
    return KBBQ_OK;
}

int kbbq_bgzf_submit_synth(kbbq_bgzf *z, kbbq_engine *e, const kbbq_synth_params *sp, uint64_t first_read, uint64_t n, int32_t format,
                           uint64_t *payload_bytes) {
    if (!z || !e || !sp || !n || format < 0 || format > 2) return fail(KBBQ_EINVAL, "bad argument");
    KbbqDeviceGuard guard(z->device);
    HIP_TRY(guard.err);
    kbbq_reads dev;
    int rc = kbbq_synth_reads(e, sp, first_read, n, &dev);
    if (rc) return rc;
    struct FreeBatch { kbbq_engine *e; kbbq_reads *d; ~FreeBatch() { kbbq_reads_free(e, d); } } free_batch{e, &dev};
    const uint32_t W = format == 0 ? synth_fastq_record(sp->read_len) : synth_bam_record(sp->read_len, format == 2);
    const uint64_t t = n * (uint64_t)W;
    Submission *sp2;
    if ((rc = begin_submission(z, kbbq_engine_stream(e), &sp2))) return rc;
    Submission &s = *sp2;
    s.n = t;
    s.formatted = true;
    if ((rc = s.payload.reserve(t + 16))) return rc;
    HIP_TRY(hipMemsetAsync((char *)s.payload.p + t, 0, 16, z->st));
    HIP_TRY(hipEventRecord(s.t0, z->st));
    SynthBatch B;
    B.bases = dev.bases; B.nmask = dev.nmask; B.qual = dev.qual; B.first = first_read; B.n = n; B.read_len = sp->read_len;
    const unsigned grid = (unsigned)std::min<uint64_t>((n + 3) / 4, 256 * 32);
    if (format == 0) hipLaunchKernelGGL(k_synth_fastq, dim3(grid), dim3(256), 0, z->st, B, (uint8_t *)s.payload.p);
    else hipLaunchKernelGGL(k_synth_bam, dim3(grid), dim3(256), 0, z->st, B, format == 2 ? 1 : 0, (uint8_t *)s.payload.p);
    HIP_TRY(hipGetLastError());
    if ((rc = launch_deflate(z, s))) return rc;
    HIP_TRY(hipEventSynchronize(s.t1));      // the batch is freed on return: the formatting kernel must be through with it
    if (payload_bytes) *payload_bytes = t;
    return KBBQ_OK;
}

int kbbq_bam_reader_preload(kbbq_bam_reader *r, const uint8_t *file_bytes, uint64_t n_bytes, uint64_t front_room) {
    if (!r || !file_bytes || !n_bytes) return fail(KBBQ_EINVAL, "bad argument");
    KbbqDeviceGuard guard(r->device);
    HIP_TRY(guard.err);
    return r->pre.start(r->device, file_bytes, n_bytes, front_room);
}

int kbbq_bam_reader_kernel_ms(kbbq_bam_reader *r, double *inflate_ms, double *index_ms) {
    if (!r) return fail(KBBQ_EINVAL, "null argument");
    if (inflate_ms) *inflate_ms = r->ms_inflate;
    if (index_ms) *index_ms = r->ms_index;
    return KBBQ_OK;
}

}  // extern "C"
