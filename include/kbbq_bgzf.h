/* kbbq_bgzf.h -- C ABI of the MI355X BGZF writer (libkbbq_engine.so): record assembly, DEFLATE and CRC-32 as HIP
 * kernels, compressed blocks back to the host in file order.
 *
 * What it replaces: the reference hands every output record to htslib's BGZF layer -- FastqFile::write builds
 * "@name\nseq\n+comment\nqual\n" and calls bgzf_write (htsiter.cc:75-86; the comment goes on the '+' line),
 * BamFile::write calls sam_write1 on a BGZF handle (htsiter.cc:45) -- which deflates the stream in blocks of 0xff00
 * payload bytes on the host.  At the engine's rate the host deflate is 85 % of the command line's wall time, so the
 * writer moves to the GPU: the caller submits a payload (bytes it has formatted itself, or the pieces of FASTQ
 * records whose text the device assembles around the recalibrated qualities it already holds) and gets back finished
 * BGZF blocks to write out in order.  The DECOMPRESSED stream is byte-identical to what the host writer produces; the
 * compressed bytes are this encoder's own (a valid RFC 1951 / RFC 1952 / SAM-spec BGZF stream, any inflater reads it).
 *
 * Plain pointers and sizes; 0 or a negative errno-style code; kbbq_last_error() (kbbq_engine.h) has the text.
 * One host thread per kbbq_bgzf.  Up to two submissions may be in flight: submit(i + 1) is queued while the caller
 * writes the blocks of collect(i).
 */
#ifndef KBBQ_BGZF_H
#define KBBQ_BGZF_H

#include <stddef.h>
#include <stdint.h>

#include "kbbq_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

#define KBBQ_BGZF_PAYLOAD 0xff00   /* payload bytes per block: htslib's BGZF_BLOCK_SIZE */

typedef struct kbbq_bgzf kbbq_bgzf;

int kbbq_bgzf_create(int32_t device, kbbq_bgzf **out);
void kbbq_bgzf_destroy(kbbq_bgzf *z);

/* Queue the compression of n payload bytes (n > 0) as consecutive BGZF blocks of KBBQ_BGZF_PAYLOAD bytes, the last one
 * shorter (no EOF block: kbbq_bgzf_eof_block).  payload is device memory if payload_on_device, host memory otherwise
 * (page-locked memory, kbbq_host_alloc, is copied by DMA); a host payload is the caller's again when the call returns.
 * after_stream: a hipStream_t (or NULL) whose work queued so far must finish first -- e.g. kbbq_engine_stream(e)
 * when the payload is produced by the engine.  KBBQ_ESTATE when two submissions are already in flight. */
int kbbq_bgzf_submit(kbbq_bgzf *z, const void *payload, uint64_t n, int32_t payload_on_device, void *after_stream);

/* One batch of FASTQ records whose text the device assembles (FastqFile::write, htsiter.cc:75-86):
 *   blob   host memory: for every record its name, comment and sequence text back to back (no separators)
 *   lens   host memory: 3 x n_records lengths (name, comment, sequence)
 *   d_qual DEVICE memory: the batch's new qualities as phred values, read r at d_qual[qual_offset(r)]; the text gets q + 33
 *   d_qual_offsets DEVICE memory: n_records + 1 base offsets, or NULL for equally long reads of uniform_len bases
 * The record text is "@" name "\n" seq "\n+" comment "\n" qual "\n".  Then as kbbq_bgzf_submit. */
int kbbq_bgzf_submit_fastq(kbbq_bgzf *z, const char *blob, const uint32_t *lens, uint64_t n_records, const uint8_t *d_qual,
                           const uint64_t *d_qual_offsets, uint32_t uniform_len, void *after_stream);

/* Wait for the oldest submission: *blocks points at its BGZF blocks, back to back in file order, *n_bytes long, in
 * page-locked memory of the writer, valid until the third kbbq_bgzf_collect from now (three buffers in turn: a caller
 * may hand the blocks to a writing thread and go on submitting and collecting).  *payload_bytes (optional) = the
 * uncompressed size. */
int kbbq_bgzf_collect(kbbq_bgzf *z, const uint8_t **blocks, uint64_t *n_bytes, uint64_t *payload_bytes);

/* The 28-byte empty block that ends a BGZF file. */
const uint8_t *kbbq_bgzf_eof_block(void);
/* Upper bound of the compressed size of n payload bytes (every block stored). */
uint64_t kbbq_bgzf_bound(uint64_t n);

/* Milliseconds the device spent in the writer's kernels since creation (format, deflate, gather), for reports. */
int kbbq_bgzf_kernel_ms(kbbq_bgzf *z, double *format_ms, double *deflate_ms, double *gather_ms);

/* ---- the input side: a BGZF-compressed four-line FASTQ file read on the device ---------------------------------------
 * What it replaces: kseq_read over bgzf_read (htsiter.hh:101-126, htsiter.cc:49-60) and the FASTQ constructor of
 * CReadData (readutils.cc:64-104), once per chunk of the file instead of once per record and pass.  The caller feeds the
 * file's bytes in chunks; the device inflates every BGZF block (one wavefront each), finds the lines, checks that the
 * records have kseq's four-line shape, and packs them into the engine's read layout.  The same chunks fed again in pass 4
 * give the output: the records' text re-assembled on the device around the new qualities and deflated by a kbbq_bgzf.
 * Shapes this path does not take -- multi-line records, FASTA, empty reads, carriage returns, "RG:" fields in read names,
 * a file that is not BGZF -- are reported (flags bit 0) and left to the caller's serial reader, which stays the definition. */
typedef struct kbbq_fastq_reader kbbq_fastq_reader;
typedef struct kbbq_fastq_chunk {
    uint64_t consumed;      /* bytes of the input that were taken: whole BGZF blocks (feed the rest again with the next chunk) */
    uint64_t n_records;     /* complete records of this chunk (a record cut by the chunk's end is carried into the next one) */
    uint64_t n_bases;
    uint32_t longest, shortest;
    uint32_t flags;         /* bit 0: a shape the device path does not take; bit 1: a read name shorter than 2 characters
                             * (readutils.cc:90 throws); bit 2: the input ended inside a record */
    uint32_t n_blocks;      /* BGZF blocks inflated */
    uint64_t text_bytes;    /* inflated bytes of this chunk */
} kbbq_fastq_chunk;
int kbbq_fastq_reader_create(int32_t device, kbbq_fastq_reader **out);
void kbbq_fastq_reader_destroy(kbbq_fastq_reader *r);
/* Restart at the beginning of a file (pass 4 feeds the same chunks again). */
int kbbq_fastq_reader_rewind(kbbq_fastq_reader *r);
/* Keep what the first scan inflates: with on != 0 (before the scan's first chunk) the text and record index of every chunk
 * with records stay in device memory, so that pass 4 selects them (kbbq_fastq_reader_select) instead of reading and
 * inflating the file a second time -- the reference reads its input once per pass (htsiter.cc:49-60); here the second
 * reading costs nothing while the text fits in HBM (about 2.3 bytes per base).  A chunk whose buffers can no longer be
 * allocated releases everything kept (kbbq_fastq_reader_kept then reports 0 chunks) and the scan goes on; on == 0 does
 * the same on the caller's word (its own budget is used up).  kept: chunks and bytes held now. */
int kbbq_fastq_reader_keep(kbbq_fastq_reader *r, int32_t on);
int kbbq_fastq_reader_kept(kbbq_fastq_reader *r, uint64_t *n_chunks, uint64_t *n_bytes);
/* Kept chunk i (in the order of the first scan, chunks without records not counted) becomes the current chunk for
 * kbbq_fastq_reader_write; info (may be NULL) gets its counts.
 * A chunk whose batch was built (kbbq_fastq_reader_batch) and whose sequence lines hold nothing but ACGTN and acgt is kept
 * in a short form -- names and comments only, a seventh of the memory -- because the packed batch gives those lines back
 * exactly; kbbq_fastq_reader_attach hands the selected chunk its batch (the one kbbq_fastq_reader_batch returned for it)
 * before kbbq_fastq_reader_write.  Attaching is harmless for a chunk kept whole. */
int kbbq_fastq_reader_select(kbbq_fastq_reader *r, uint64_t i, kbbq_fastq_chunk *info);
int kbbq_fastq_reader_attach(kbbq_fastq_reader *r, const kbbq_reads *batch);
/* The next bytes of the file (host memory; page-locked memory is copied by DMA).  last != 0: nothing follows.
 * Every block's CRC-32 and ISIZE are checked as bgzf_read checks them; a block that does not inflate to them is
 * KBBQ_EIO ("CRC32 checksum mismatch" / "does not inflate"). */
int kbbq_fastq_reader_chunk(kbbq_fastq_reader *r, const uint8_t *file_bytes, uint64_t n_bytes, int32_t last, kbbq_fastq_chunk *info);
/* Optional: start copying a piece of the file to the device AHEAD of the kbbq_fastq_reader_chunk call that will take it --
 * from the caller's I/O thread, the moment the piece has been read (page-locked memory) -- so that the host link moves piece
 * i + 1 while the kernels of piece i run.  The chunk call recognises the piece by its address: file_bytes of that call may
 * start up to front_room bytes BEFORE these bytes (what the call before left unconsumed, put in front by the caller) and must
 * end where they end.  The memory must stay as it is until that chunk call has returned.  Two pieces may be ahead. */
int kbbq_fastq_reader_preload(kbbq_fastq_reader *r, const uint8_t *file_bytes, uint64_t n_bytes, uint64_t front_room);
/* kbbq_reads_upload (kbbq_engine.h) for a batch whose bases are still text: host's bases / nmask / offcase are ignored,
 * seq_text holds its n_bases sequence characters and is packed on the device (kbbq_pack_bases_case's table).  e may be
 * NULL like there. */
int kbbq_reads_upload_text(kbbq_engine *e, const kbbq_reads *host, const uint8_t *seq_text, kbbq_reads *dev);
/* The inflater alone, for callers that parse on the host (BAM; FASTQ shapes the record kernels do not take): the whole
 * BGZF blocks at the front of file_bytes[0, n_bytes) whose inflated bytes fit in `capacity` are inflated on the device,
 * checked (CRC-32, ISIZE) and copied to host_out (page-locked memory makes the copy DMA).  *consumed: bytes of the input
 * taken (feed the rest again in front of the next piece), *produced: bytes written.  What bgzf_read does under
 * sam_read1 / kseq_read (htsiter.hh:64-66,101-126), at the device's rate instead of a thread pool's. */
int kbbq_fastq_reader_inflate(kbbq_fastq_reader *r, const uint8_t *file_bytes, uint64_t n_bytes, uint8_t *host_out, uint64_t capacity,
                              uint64_t *consumed, uint64_t *produced);
/* The current chunk's records as a device batch (arrays owned by the library: kbbq_reads_free): bases, N mask, qualities,
 * offsets (NULL and read_len for equally long reads), second-in-pair flags, off-case bits when a base is not upper-case. */
int kbbq_fastq_reader_batch(kbbq_fastq_reader *r, kbbq_reads *dev);
/* Pass 4: the current chunk's records as "@name\nseq\n+comment\nqual\n" (FastqFile::write, htsiter.cc:75-86) with
 * d_qual (device: the batch's new qualities, in the batch's base order) on the quality lines, submitted to writer z
 * (kbbq_bgzf_collect returns the blocks).  after_stream as in kbbq_bgzf_submit. */
int kbbq_fastq_reader_write(kbbq_fastq_reader *r, kbbq_bgzf *z, const uint8_t *d_qual, void *after_stream);
/* Milliseconds the device spent inflating / indexing + packing since creation. */
int kbbq_fastq_reader_kernel_ms(kbbq_fastq_reader *r, double *inflate_ms, double *index_ms);

/* ---- the input side: a BAM file read on the device (round 4) ----------------------------------------------------------
 * What it replaces: sam_read1 (BamFile::next, htsiter.cc:5), the BAM constructor of CReadData (readutils.cc:13-61, with
 * bam_seq_str, readutils.hh:30-42) and -- pass 4 -- BamFile::recalibrate + sam_write1 (htsiter.cc:11-45), once per chunk of
 * the file instead of once per record and pass.  The caller parses the BAM header itself (it needs the reference lengths,
 * kbbq.cc:196-216) and feeds the file's bytes in chunks, as for kbbq_fastq_reader; the device inflates every BGZF block,
 * finds the alignment records (a length-prefixed chain: segments of the stream are walked in parallel from guessed starts
 * and accepted only where each segment starts exactly where the one before it lands, csrc/bam_device.h), decodes them --
 * 4-bit bases to the engine's 2-bit layout + N mask, reverse-strand records reverse-complemented with their qualities
 * reversed, qualities from OQ:Z with use_oq, second-in-pair from FLAG 0x80, RG:Z looked up in the header's @RG ids -- and
 * pass 4 rewrites them in place around the new qualities (with set_oq the old ones become OQ:Z, replaced or appended as
 * bam_aux_update_str does) for a kbbq_bgzf to deflate.  Shapes this path does not take are reported in flags and left to
 * the caller's host parser (bam_io.cc), which stays the definition:
 *   bit 0  not BGZF; a malformed alignment block; a record without a usable RG:Z tag, or one the header has no @RG line
 *          for; with use_oq a missing / corrupt OQ tag or one whose length is not l_seq; more than 4096 misjudged segments
 *   bit 2  the stream ended inside a record
 *   bit 3  some record's OQ tag could not be updated by bam_aux_update_str (other type, or malformed tags in front of the
 *          place it would be appended): the caller must not use kbbq_bam_reader_write with set_oq on this file
 * Read groups get dense indices in the order of their first record, as rg_to_int does (readutils.cc:53-57). */
typedef struct kbbq_bam_reader kbbq_bam_reader;
typedef struct kbbq_fastq_chunk kbbq_bam_chunk;      /* same fields; text_bytes = inflated bytes, flags as above */
/* header_bytes: size of the BAM header in the inflated stream (magic, l_text, text, n_ref, references);
 * n_ref: number of references (refID must be in [-1, n_ref)); rg_ids: the ID fields of the header's @RG lines. */
int kbbq_bam_reader_create(int32_t device, int32_t use_oq, int32_t n_ref, uint64_t header_bytes, const char *const *rg_ids, uint32_t n_rg_ids,
                           kbbq_bam_reader **out);
void kbbq_bam_reader_destroy(kbbq_bam_reader *r);
int kbbq_bam_reader_rewind(kbbq_bam_reader *r);
/* Keep the COMPRESSED bytes of every chunk with records in device memory (about a third of the stream's size), so that
 * pass 4 inflates and indexes them again there (kbbq_bam_reader_select) instead of reading the file a second time. */
int kbbq_bam_reader_keep(kbbq_bam_reader *r, int32_t on);
int kbbq_bam_reader_kept(kbbq_bam_reader *r, uint64_t *n_chunks, uint64_t *n_bytes);
int kbbq_bam_reader_select(kbbq_bam_reader *r, uint64_t i, kbbq_bam_chunk *info);
int kbbq_bam_reader_chunk(kbbq_bam_reader *r, const uint8_t *file_bytes, uint64_t n_bytes, int32_t last, kbbq_bam_chunk *info);
int kbbq_bam_reader_preload(kbbq_bam_reader *r, const uint8_t *file_bytes, uint64_t n_bytes, uint64_t front_room);   /* as kbbq_fastq_reader_preload */
/* The read groups met so far in dense-index order: table_index[d] = index into rg_ids of the group with dense index d. */
int kbbq_bam_reader_read_groups(kbbq_bam_reader *r, uint32_t *table_index, uint32_t capacity, uint32_t *n);
/* The current chunk's records as a device batch (arrays owned by the library: kbbq_reads_free), rg = dense indices. */
int kbbq_bam_reader_batch(kbbq_bam_reader *r, kbbq_reads *dev);
/* Pass 4: the current chunk's records with d_qual (device: the batch's new qualities in the batch's base order) in their
 * quality fields, submitted to writer z (kbbq_bgzf_collect returns the blocks).  after_stream as in kbbq_bgzf_submit. */
int kbbq_bam_reader_write(kbbq_bam_reader *r, kbbq_bgzf *z, const uint8_t *d_qual, int32_t set_oq, void *after_stream);
int kbbq_bam_reader_kernel_ms(kbbq_bam_reader *r, double *inflate_ms, double *index_ms);

/* ---- test and bench data as files: reads [first_read, first_read + n) of the synthetic data set (kbbq_synth_params: the
 * generator bench.py measures on) formatted on the device and submitted to z like any payload.  format 0: four-line FASTQ,
 * names "r%010llu"; 1: unaligned BAM records with RG:Z:grp0, about half of them reverse-flagged (stored reverse-complemented,
 * qualities reversed); 2: the same with the true qualities in OQ:Z and 11s in the quality field (BASELINE configs[3]:
 * --use-oq).  Every record has the same size.  The caller writes the BAM header itself.  `kbbq --io-test synth-fastq|synth-bam`. */
int kbbq_bgzf_submit_synth(kbbq_bgzf *z, kbbq_engine *e, const kbbq_synth_params *sp, uint64_t first_read, uint64_t n, int32_t format,
                           uint64_t *payload_bytes);

/* ---- host-only twin (no GPU touched): the same scalar pieces (Huffman lengths, header, token bits, CRC chaining,
 * framing) around a serial match finder; lets the CPU test-suite inflate what those pieces produce. */
int kbbq_host_bgzf_compress(const uint8_t *payload, uint64_t n, uint8_t *out, uint64_t out_capacity, uint64_t *out_bytes);

#ifdef __cplusplus
}
#endif
#endif /* KBBQ_BGZF_H */
