/* kbbq_engine.h -- C ABI of the MI355X k-mer BQSR engine (libkbbq_engine.so).
 *
 * This is the drop-in boundary for kbbq's hot path.  The reference has no FFI;
 * its seam is the five pass functions main() calls (recalibrateutils.hh:29-44,
 * covariateutils.hh:171, called at kbbq.cc:280,337,366,407,457).  Each entry
 * point below is the batch-level equivalent of one of them and cites what it
 * replaces.  Plain pointers and sizes only; no exceptions cross the boundary;
 * every function returns 0 (KBBQ_OK) or a negative errno-style code, and
 * kbbq_last_error() gives the text.
 *
 * Threading: one host thread per engine (the reference's passes are
 * single-threaded and non-reentrant).  All device work of
 * an engine runs on its own HIP streams (kbbq_engine_stream()).  Every entry point
 * works on the engine's device and leaves the calling thread's current HIP device
 * as it found it.
 *
 * Results are independent of batch size, batch order inside a pass (given each
 * batch's first_kmer_ordinal) and GPU count: Bloom inserts are bitwise ORs and
 * histograms are integer sums.
 */
#ifndef KBBQ_ENGINE_H
#define KBBQ_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KBBQ_OK 0
#define KBBQ_EINVAL (-22)   /* bad argument */
#define KBBQ_ENOMEM (-12)   /* host or device allocation failed (reference: std::bad_alloc, bloom.hh:49-51) */
#define KBBQ_EIO (-5)       /* HIP runtime error */
#define KBBQ_ERANGE (-34)   /* read longer than KBBQ_MAX_READ_LEN, k out of 1..32, ... */
#define KBBQ_ENODEV (-19)   /* no usable GPU */
#define KBBQ_ESTATE (-1)    /* call out of order (e.g. trusted pass before thresholds are set) */

#define KBBQ_MAX_KMER 32        /* bloom.hh:15 */
#define KBBQ_MAXQ 93            /* covariateutils.hh:3: the largest quality the MODEL proposes and the output clamp (readutils.cc:592-594) */
#define KBBQ_NQ 256             /* quality rows of every table: an input quality is any uint8_t (a BAM can hold up to 255) and the
                                 * reference's tables grow with the largest one seen (covariateutils.cc:65-76,102-116,147-164) */
#define KBBQ_MAX_READ_LEN 8388607 /* 2^23 - 1: a read's positions travel in 23 bits of a word (the reference has no limit:
                                   * covariateutils.cc:102-116; the longest nanopore reads published are about half of this);
                                   * reads of up to 512 bases take the staged fast kernels, longer ones the windowed forms.
                                   * The tables are [n_rg][256][2][params.max_read_len]: 8 KiB of histogram per cycle and read group */
#define KBBQ_DEFAULT_BLOOM_SEED 0xA5A5A5A55A5A5A5AULL   /* bloom.hh:389 */

#define KBBQ_SAMPLED 0
#define KBBQ_TRUSTED 1

typedef struct kbbq_engine kbbq_engine;

/* Replaces the scalar set-up in main(), kbbq.cc:251-271. */
typedef struct kbbq_params {
    int32_t k;              /* --ksize, 1..32 (kbbq.cc:82,101) */
    int32_t device;         /* HIP device ordinal */
    double alpha;           /* sampling rate as the double KmerSubsampler receives (htsiter.hh:143) */
    uint32_t seed;          /* sampler seed (kbbq.cc:86,268-271); must be given: the engine never draws one */
    int32_t n_rg;           /* number of read groups (>= 1): first dimension of the histograms */
    uint64_t approx_kmers;  /* genomelen*coverage*alpha (kbbq.cc:264) */
    double fpr_sampled;     /* (double)0.01L  (kbbq.cc:155,265) */
    double fpr_trusted;     /* (double)0.0005L (kbbq.cc:156,266) */
    uint64_t bloom_seed;    /* KBBQ_DEFAULT_BLOOM_SEED unless testing */
    int32_t max_read_len;   /* longest read that will be submitted: cycle dimension of the histograms */
    int32_t flags;          /* KBBQ_F_* */
} kbbq_params;

#define KBBQ_F_PROFILE 1    /* time every kernel with HIP events (kbbq_profile_get) */
/* Behavioural switches (none changes a result), fixed when the engine is created.  Each also has an environment
 * variable, read at kbbq_engine_create when the flag is not given (README.md). */
#define KBBQ_F_NO_OVERLAP 2          /* every kernel in order on one stream: exclusive kernel durations for profiles (KBBQ_NO_OVERLAP=1) */
#define KBBQ_F_BUCKET_OFF 4          /* direct Bloom inserts whatever the filter size (KBBQ_BUCKET=0) */
#define KBBQ_F_BUCKET_ON 8           /* slice-bucketed inserts whatever the filter size (KBBQ_BUCKET=1; default: filters of 256 MB and more) */
#define KBBQ_F_LANE_WALK 16          /* the one-read-per-lane form of the correction walk for every read (KBBQ_CORRECT=lane) */
#define KBBQ_F_NO_FASTPATH 32        /* no in-scan fast path: every read with untrusted k-mers takes the walk (KBBQ_NO_FASTPATH=1) */
#define KBBQ_F_PASS2_INORDER 64      /* the insert side of pass 2 behind k_infer instead of beside it (KBBQ_PASS2_SIDE=0) */
#define KBBQ_F_NO_PASS4_PIPELINE 128 /* pass 4 of a host batch in one piece (KBBQ_NO_PASS4_PIPELINE=1) */

/* One batch of reads, structure of arrays.  All pointers are device pointers if
 * on_device != 0, host pointers otherwise.  A host batch is copied into one of the
 * engine's device staging slots (a ring of three, no allocation per call) with
 * hipMemcpyAsync on the engine's copy stream -- DMA straight out of the caller's memory
 * when that is page-locked (kbbq_host_alloc), staged by the runtime otherwise -- and the
 * pass's kernels wait for the copy by event.  The call returns when the COPY has landed:
 * the caller may reuse the batch's memory at once, while the kernels are still queued or
 * running, so the copy of batch i+1 overlaps the kernels of batch i.  Results a call hands
 * back in host memory (error masks, recalibrated qualities) are complete on return; results
 * that stay in the engine (filters, histograms) are complete after the pass's *_finish or
 * kbbq_engine_sync.  Calls with device batches only queue work (see the individual functions).
 *   bases   2 bits per base, base i of the batch in bits [2*(i%32), +2) of word
 *           i/32; A=0 C=1 G=2 T=3 (seq_nt16_int[seq_nt16_table[ch]], bloom.hh:351);
 *           anything else is stored as 0 with its nmask bit set.
 *   nmask   1 bit per base, bit i%64 of word i/64.
 *   qual    phred value per base (FASTQ char - 33, readutils.cc:70-71).
 *   offsets n_reads+1 base offsets, or NULL for uniform reads of read_len bases.
 *   flags   per read: bit 0 = second-in-pair (readutils.cc:59,89-97); NULL = 0.
 *   rg      per read dense read-group index (order of first appearance,
 *           readutils.cc:54-58,100-103); NULL = 0.
 *   hint_*  see below (optional).
 * bases and nmask must be allocated with at least one extra zero u64 word past
 * the last used one (the kernels read unaligned 64-bit windows). */
typedef struct kbbq_reads {
    uint64_t n_reads;
    uint64_t n_bases;
    const uint64_t *bases;
    const uint64_t *nmask;
    const uint8_t *qual;
    const uint64_t *offsets;
    const uint8_t *flags;
    const uint16_t *rg;
    uint32_t read_len;
    int32_t on_device;
    /* Optional, device batches only (NULL = off): two caller-owned, caller-zeroed bit arrays, 1 bit per
     * base of the batch (layout of nmask, n_bases/64+2 words), that live as long as the batch is
     * re-submitted across passes.  Pass 1 marks the k-mer starts this read inserted into the sampled
     * filter, pass 2 skips those lookups (an inserted k-mer is contained) and marks the starts it
     * inserted into the trusted filter, pass 3 skips those.  Results are unchanged. */
    uint64_t *hint_sampled;
    uint64_t *hint_trusted;
    /* Optional (NULL = every base is upper-case): 1 bit per base, layout of nmask, set where the raw character is an
     * ACGT base but not the upper-case letter -- 'a','c','g','t' of a soft-masked FASTQ, and the digits '0'..'3'
     * that seq_nt16_table also folds to bases.  K-mers, covariates and the apply step fold case (bloom.hh:351), but
     * three loops of the reference compare RAW characters with 'A','C','G','T' (bloom.cc:142,218,249;
     * readutils.cc:202), so for such a base they also try the candidate equal to it; the engine reproduces that.
     * kbbq_pack_bases_case fills the array.  Device or host like the batch. */
    const uint64_t *offcase;
} kbbq_reads;

typedef struct kbbq_filter_info {
    uint64_t bits;            /* blocked size: multiple of 512 (bloom.hh:44) */
    uint64_t bits_unblocked;  /* bloom_parameters::optimal_parameters.table_size */
    uint64_t n_blocks;
    uint64_t random_seed;     /* bloom.hh:39 */
    uint64_t inserted;        /* inserted_element_count_ (counts duplicates) */
    uint32_t n_hash;          /* optimal_parameters.number_of_hashes */
    uint32_t n_salt;          /* max(n_hash, 2) */
    uint32_t salt[128];
    uint64_t table_bytes;     /* size of the DEVICE bit array: n_blocks * 16 (the engine keeps the 128 bits of a
                               * block the reference's patterns can reach, see kbbq_filter_device_table) */
} kbbq_filter_info;

/* ---- engine life cycle ------------------------------------------------ */

/* Replaces `bloom::Bloom subsampled(...), trusted(...)` (kbbq.cc:265-266) and
 * `KmerSubsampler subsampler(file, k, alpha, seed)` (kbbq.cc:277). */
int kbbq_engine_create(const kbbq_params *params, kbbq_engine **out);
void kbbq_engine_destroy(kbbq_engine *e);
/* Zero both filters, counters, histograms, delta-Q tables (a new run). */
int kbbq_engine_reset(kbbq_engine *e);
/* Block until everything submitted so far has finished. */
int kbbq_engine_sync(kbbq_engine *e);
/* Numeric knobs of an engine (tests and measurements; none changes a result): "bucket_records" = records gathered per
 * flush of the slice-bucketed inserts (before the first batch; KBBQ_BUCKET_RECORDS), "pass4_piece" = bases per piece of
 * the pass-4 pipeline of a host batch (KBBQ_PASS4_PIECE), "no_overlap" = 1 / 0: KBBQ_F_NO_OVERLAP switched on or off between
 * two runs (the call waits for everything queued; bench.py takes its exclusive kernel durations this way), "infer_subset"
 * = 1 / 0 (KBBQ_INFER_SUBSET), "pass2_side" = 0 / 1 / 2: where the insert side of pass 2 runs (KBBQ_PASS2_SIDE; waits likewise),
 * "scan_blocks" / "walk_blocks" / "infer_blocks" = workgroups per CU of the kernels that stay resident for a whole batch while
 * the other stream has work (0 = as many as fit ... 16, 17 = the engine's own choice; KBBQ_SCAN_BLOCKS, KBBQ_WALK_BLOCKS,
 * KBBQ_INFER_BLOCKS).  KBBQ_EINVAL for an unknown name or a value out of range.
 * The engine runs kernels of neighbouring batches on two streams of its own: a host process with many streams should set
 * GPU_MAX_HW_QUEUES >= 8 in its environment before its first HIP call, or the two may share a hardware queue. */
int kbbq_engine_tune(kbbq_engine *e, const char *name, uint64_t value);
/* hipStream_t of the engine, as void*. */
void *kbbq_engine_stream(kbbq_engine *e);
/* First and last dimension of the histograms and delta-Q tables: read groups, cycles (kbbq_params.n_rg / max_read_len). */
int kbbq_engine_dims(kbbq_engine *e, uint64_t *n_rg, uint64_t *n_cycle);
const char *kbbq_last_error(void);

int kbbq_filter_info_get(kbbq_engine *e, int which, kbbq_filter_info *out);
/* Device address of a filter's bit array, for collectives: kbbq_filter_info.table_bytes bytes in the ENGINE's
 * layout.  The reference sets bit b (0..511) of a pattern in 64-bit word (b>>8)*4 + ((b>>3)&3) at bit b&63
 * (get_vector_unit, bloom.hh:110-113,228), so only 16 bits of every word -- bytes w&3 and 4+(w&3) of word w --
 * can ever be set; the engine stores those: block = 2 x u64, word w of the reference = 16-bit field w&3 of
 * u64 w>>2 (low byte = byte w&3 of the word, high byte = byte 4+(w&3)).  Bitwise OR commutes with that
 * mapping, so the multi-GPU exchange works on this array directly. */
void *kbbq_filter_device_table(kbbq_engine *e, int which);
/* Device address of the 8-byte insert counter of a filter. */
void *kbbq_filter_device_counter(kbbq_engine *e, int which);
/* The bit array / the pattern table in the REFERENCE's layout (n_blocks*8 resp. 65536*8 words of 64 bits,
 * pattern_blocked_bf's table and patterns, bloom.hh:175-231), expanded from the device copies. */
int kbbq_filter_download(kbbq_engine *e, int which, uint64_t *host_words, uint64_t n_words);
int kbbq_filter_patterns_download(kbbq_engine *e, int which, uint64_t *host_words /* 65536*8 */);
/* dst |= src over a range of a filter's device bit array; src is a device buffer in the same (engine)
 * layout, offsets and counts in u64 words of it (the OR step of the multi-GPU all-reduce; RCCL has no
 * bitwise OR). */
int kbbq_filter_or_from(kbbq_engine *e, int which, const void *src_device, uint64_t word_offset, uint64_t n_words);
/* dst |= src for two arbitrary 16-byte-aligned device buffers of n_words u64
 * (reducing the pieces received in the OR all-reduce). */
int kbbq_device_or(kbbq_engine *e, void *dst_device, const void *src_device, uint64_t n_words);
/* dst |= OR over p != skip of src[p*piece_words .. (p+1)*piece_words): all received pieces in one launch. */
int kbbq_device_or_pieces(kbbq_engine *e, void *dst_device, const void *src_device, uint64_t piece_words,
                          int32_t n_pieces, int32_t skip);
int kbbq_filter_set_inserted(kbbq_engine *e, int which, uint64_t inserted);
/* A running byte sum on the engine's stream: the digest of recalibrated qualities that never leave the device uncompressed
 * (`kbbq` with KBBQ_QUAL_DIGEST=1 prints it; bench.py's recal_qual_sum is the same number).  add: queued behind the
 * kernels that produce device_bytes; get: waits, returns the sum since the last reset. */
int kbbq_digest_add(kbbq_engine *e, const uint8_t *device_bytes, uint64_t n);
int kbbq_digest_get(kbbq_engine *e, uint64_t *sum, int32_t reset);

/* ---- read staging ------------------------------------------------------ */

/* Host helper: ASCII FASTQ-style arrays -> the packed layout above.
 * seq/qual_ascii hold n_bases characters; bases_out/nmask_out must have
 * n_bases/32+2 and n_bases/64+2 words; qual_out n_bases bytes (qual_ascii-33). */
int kbbq_pack_bases(const uint8_t *seq, uint64_t n_bases, uint64_t *bases_out, uint64_t *nmask_out);
/* Same, and the off-case bit array of kbbq_reads.offcase (n_bases/64+2 words); *n_offcase (optional) = bits set, so
 * that a driver can leave kbbq_reads.offcase NULL for the usual all-upper-case batch. */
int kbbq_pack_bases_case(const uint8_t *seq, uint64_t n_bases, uint64_t *bases_out, uint64_t *nmask_out, uint64_t *offcase_out,
                         uint64_t *n_offcase);
/* Copy a host batch to device memory owned by the engine library; *dev gets
 * device pointers (on_device=1).  Free with kbbq_reads_free. */
int kbbq_reads_upload(kbbq_engine *e, const kbbq_reads *host, kbbq_reads *dev);
int kbbq_reads_free(kbbq_engine *e, kbbq_reads *dev);
/* A device batch copied onto `device` (arrays owned by the library; free with that device current, or through an engine
 * of that device): a single-process multi-device caller gives every engine its shard this way.  with_hints != 0: with zeroed
 * hint arrays (kbbq_reads_alloc_hints). */
int kbbq_reads_clone(const kbbq_reads *src, int32_t device, int32_t with_hints, kbbq_reads *out);
/* Page-locked host memory for batch arrays and outputs (hipHostMalloc): what makes the copies of host batches
 * true DMA at the link's rate.  kbbq_measure_host_link times one `bytes`-sized copy each way from such memory. */
int kbbq_host_alloc(size_t bytes, void **out);
int kbbq_host_free(void *p);
/* Plain device memory on the engine's device (e.g. the output array of kbbq_recalibrate_batch for a resident batch
 * whose new qualities stay on the GPU for the BGZF writer, kbbq_bgzf.h). */
int kbbq_device_alloc(kbbq_engine *e, size_t bytes, void **out);
int kbbq_device_free(kbbq_engine *e, void *p);
int kbbq_measure_host_link(int32_t device, uint64_t bytes, double *h2d_gbps, double *d2h_gbps);
/* The same with both directions running at once (three copies each way on two streams): the rates pass 4 of the
 * host-batch mode, which copies in and out at the same time, can hope for. */
int kbbq_measure_host_link_duplex(int32_t device, uint64_t bytes, double *h2d_gbps, double *d2h_gbps);
/* For both, e may be NULL (current device): a driver can make its batches resident in HBM while it is
 * still counting the bases that size the engine (kbbq.cc:229-264), then run every pass from HBM.
 * kbbq_reads_alloc_hints gives a device batch zeroed hint arrays (see kbbq_reads.hint_*), owned by the
 * library until kbbq_reads_free_hints.  kbbq_device_memory: hipMemGetInfo of a device (-1 = current). */
int kbbq_reads_alloc_hints(kbbq_reads *dev);
int kbbq_reads_free_hints(kbbq_reads *dev);
int kbbq_device_memory(int32_t device, uint64_t *free_bytes, uint64_t *total_bytes);

/* ---- pass 1: recalibrateutils::subsample_kmers (recalibrateutils.cc:7-13) -- */

/* first_kmer_ordinal = number of k-mer positions (sum of max(0,len-k+1)) in all
 * reads of the file that precede this batch: the sampler consumes exactly one
 * draw per k-mer position in file order (htsiter.cc:89-129). */
int kbbq_sample_batch(kbbq_engine *e, const kbbq_reads *reads, uint64_t first_kmer_ordinal);
/* Number of k-mer positions in a batch (to advance first_kmer_ordinal). */
int kbbq_count_kmer_positions(kbbq_engine *e, const kbbq_reads *reads, uint64_t *out);
/* Syncs; *inserted = Bloom::inserted_elements() of the sampled filter (kbbq.cc:283). */
int kbbq_sample_finish(kbbq_engine *e, uint64_t *inserted);

/* ---- between passes: kbbq.cc:304-313 ------------------------------------ */

/* fpr = subsampled.fprate(); p = calculate_phit(subsampled, alpha);
 * thresholds = calculate_thresholds(k, p).  alpha_text is the long double alpha
 * as decimal text (the reference keeps alpha in long double, kbbq.cc:83,251).
 * thresholds_out: k+1 ints.  Returns 1 when fpr > .15 (the reference exits 1,
 * kbbq.cc:306-310), 0 otherwise; thresholds are installed either way. */
int kbbq_compute_thresholds(kbbq_engine *e, const char *alpha_text, int32_t *thresholds_out, double *fpr_out,
                            char *p_text_out, size_t p_text_len);
int kbbq_set_thresholds(kbbq_engine *e, const int32_t *thresholds, int32_t n);

/* ---- pass 2: recalibrateutils::find_trusted_kmers (recalibrateutils.cc:15-40) */

/* infer_errors_out (optional, device or host like the batch): 1 bit per base,
 * CReadData::infer_read_errors' flags, same bit layout as nmask. */
int kbbq_trusted_batch(kbbq_engine *e, const kbbq_reads *reads, uint64_t *infer_errors_out);
int kbbq_trusted_finish(kbbq_engine *e, uint64_t *inserted);

/* ---- pass 3: recalibrateutils::get_covariatedata (recalibrateutils.cc:42-89) */

/* get_errors(trusted, k, 6) then CCovariateData::consume_read for every read.
 * errors_out (optional): 1 bit per base, CReadData::errors.
 * Asynchronous for device-resident batches without errors_out: the call returns once the batch is queued (its
 * correction walk and tally run on a side stream beside the next batch's Bloom scan) and the batch's arrays
 * must stay valid until the next synchronising call -- kbbq_engine_sync, kbbq_train, kbbq_covariates_get,
 * kbbq_stats_get or kbbq_engine_reset.  Host batches and calls with errors_out complete before returning. */
int kbbq_errors_batch(kbbq_engine *e, const kbbq_reads *reads, uint64_t *errors_out);
/* --fixed mode (kbbq.cc:367-378): tally with caller-supplied error bits. */
int kbbq_tally_batch(kbbq_engine *e, const kbbq_reads *reads, const uint64_t *errors);

/* Dense covariate histograms, {errors,total} pairs of u64 (KBBQ_NQ = 256 quality rows):
 *   rg    [n_rg][2]                 q     [n_rg][256][2]
 *   cycle [n_rg][256][2][max_read_len][2]   (third index: 0 first-in-pair, 1 second)
 *   dinuc [n_rg][256][16][2]         (dinuc = 4*prev + cur, covariateutils.hh:92-96) */
typedef struct kbbq_covariates {
    uint64_t n_rg, n_cycle;
    uint64_t *rg, *q, *cycle, *dinuc;   /* caller-allocated host arrays */
} kbbq_covariates;
int kbbq_covariates_get(kbbq_engine *e, kbbq_covariates *out);
/* Device addresses + word counts of the two tallied histograms (cycle, dinuc),
 * contiguous, for the multi-GPU sum all-reduce.  Call kbbq_engine_sync first: tallies of the last batches may
 * still be running (see kbbq_errors_batch). */
void *kbbq_covariates_device(kbbq_engine *e, uint64_t *n_words);

/* ---- CCovariateData::get_dqs (covariateutils.cc:204-230) ------------------ */

typedef struct kbbq_dq {
    uint64_t n_rg, n_cycle;
    int32_t *meanq;   /* [n_rg] */
    int32_t *rgdq;    /* [n_rg] */
    int32_t *qdq;     /* [n_rg][256] */
    int32_t *cycledq; /* [n_rg][256][2][n_cycle] */
    int32_t *dinucdq; /* [n_rg][256][16] */
} kbbq_dq;
/* Host-side model on the engine's histograms; installs the tables on the device. */
int kbbq_train(kbbq_engine *e);
int kbbq_dq_get(kbbq_engine *e, kbbq_dq *out);      /* caller-allocated arrays */
int kbbq_set_dq(kbbq_engine *e, const kbbq_dq *dq); /* e.g. tables computed on another rank */

/* ---- pass 4: CReadData::recalibrate (readutils.cc:572-595) ---------------- */

/* qual_out: n_bases bytes, device or host like the batch.  A device output is queued on the engine's stream
 * (complete after kbbq_engine_sync, or order your own work behind kbbq_engine_stream); a host output is complete
 * when the call returns. */
int kbbq_recalibrate_batch(kbbq_engine *e, const kbbq_reads *reads, uint8_t *qual_out);
/* Same, but qual_out is always HOST memory (a device-resident batch whose new qualities go to a writer). */
int kbbq_recalibrate_batch_host(kbbq_engine *e, const kbbq_reads *reads, uint8_t *host_qual_out);

/* ---- asynchronous submission of host batches --------------------------------
 * The calls above return when their host batch has left the caller's memory, so batch i+1 is copied only after batch i
 * has landed: a bubble per batch on the link.  The *_submit forms only QUEUE the batch -- copy into a staging slot,
 * kernels, and for pass 4 the copy of the new qualities back to qual_out -- and return a ticket; kbbq_batch_wait(ticket)
 * returns when the batch's arrays are the caller's again and (pass 4) qual_out is complete.  A caller that cycles through
 * two or three page-locked batches keeps the host link busy back to back in both directions (bench.py: pcie_inclusive).
 * Up to three host batches are in flight (the staging ring); a fourth submit waits inside the call for the oldest batch's
 * kernels.  Device batches are queued exactly as by the plain calls and get ticket 0.  The optional outputs of passes 2
 * and 3 (flag arrays) are not available in this form. */
typedef uint64_t kbbq_ticket;
int kbbq_sample_batch_submit(kbbq_engine *e, const kbbq_reads *reads, uint64_t first_kmer_ordinal, kbbq_ticket *ticket);
int kbbq_trusted_batch_submit(kbbq_engine *e, const kbbq_reads *reads, kbbq_ticket *ticket);
int kbbq_errors_batch_submit(kbbq_engine *e, const kbbq_reads *reads, kbbq_ticket *ticket);
int kbbq_recalibrate_batch_submit(kbbq_engine *e, const kbbq_reads *reads, uint8_t *qual_out, kbbq_ticket *ticket);
int kbbq_batch_wait(kbbq_engine *e, kbbq_ticket ticket);

/* ---- synthetic input and measurement ------------------------------------- */

typedef struct kbbq_synth_params {
    uint64_t seed;
    uint64_t genome_len;
    uint64_t n_reads;
    uint32_t read_len;
    uint32_t n_rg;
    uint32_t paired;      /* second-in-pair flag = read index & 1 */
    uint32_t n_per_million; /* N bases per million */
} kbbq_synth_params;
/* Threshold tables of the synthetic generator (quality profile per cycle, error
 * probability per quality), so that the host twin uses the very same numbers. */
int kbbq_synth_tables(const kbbq_synth_params *sp, uint32_t *qcum /* [read_len][4] */, uint32_t *errthr /* [94] */);
/* Generate reads [first_read, first_read+n) of the synthetic data set straight
 * into device memory (uniform-length layout); identical to kbbq_amd.synth on
 * the host.  Free with kbbq_reads_free. */
int kbbq_synth_reads(kbbq_engine *e, const kbbq_synth_params *sp, uint64_t first_read, uint64_t n, kbbq_reads *dev);

typedef struct kbbq_profile_entry {
    char name[48];
    uint64_t launches;
    double total_ms;
} kbbq_profile_entry;
int kbbq_profile_get(kbbq_engine *e, kbbq_profile_entry *out, int32_t max_entries, int32_t *n_out);
int kbbq_profile_reset(kbbq_engine *e);
/* Counters: [0] reads sent to the correction kernel and [1] Bloom queries issued there (pass 3), [2] reads of pass 3,
 * [3] Bloom blocks fetched by pass 2, [4],[5] flushes of the slice-bucketed inserts per filter, [6] records inserted
 * directly because a region was full, [7] records gathered per flush, [8] always 0 (rounds 1-2: a quality above 93 was
 * met and left out of the model; every uint8_t quality is modelled now, as the reference's growing tables do).
 * Writes min(n, 9) values. */
int kbbq_stats_get(kbbq_engine *e, uint64_t *out, int32_t n);

/* ---- host-only entry points (no GPU touched) --------------------------------
 * The scalar parts of the path, for integrators that keep their own device code
 * and for the CPU test-suite.  Same arithmetic as the engine uses internally. */
/* Filter sizing, salts and pattern table (bloom_filter.hpp:108-160,467-549; bloom.hh:36-56,189-231). */
int kbbq_host_filter_spec(uint64_t approx_kmers, double fpr, uint64_t bloom_seed, kbbq_filter_info *info,
                          uint64_t *patterns_out /* 65536*8 words or NULL */);
/* hash % n_blocks exactly as the kernels compute it (get_block, bloom.hh:99-105; kbbq_amd/csrc/modmath.h). */
uint32_t kbbq_host_block_index(uint32_t hash, uint64_t n_blocks);
/* Between the reference's 512-bit blocks (8 words) and the engine's 128-bit ones (2 words), see
 * kbbq_filter_device_table.  Squeezing fails with KBBQ_ERANGE on a bit no pattern can set. */
int kbbq_host_blocks_squeeze(const uint64_t *reference_words, uint64_t n_blocks, uint64_t *engine_words);
int kbbq_host_blocks_expand(const uint64_t *engine_words, uint64_t n_blocks, uint64_t *reference_words);
/* kbbq.cc:304-313 from the sampled filter's size and insert count; returns 1 when fpr > .15. */
int kbbq_host_thresholds(int32_t k, uint64_t filter_bits, uint64_t inserted, uint32_t n_salt, const char *alpha_text,
                         int32_t *thresholds_out, double *fpr_out, char *p_text_out, size_t p_text_len);
/* CCovariateData::get_dqs (covariateutils.cc:204-230) on dense histograms; cov->rg and cov->q are
 * ignored (they are sums of cov->cycle). */
int kbbq_host_train(const kbbq_covariates *cov, kbbq_dq *out);
/* Largest T with: std::bernoulli_distribution(p) accepts a 64-bit draw u  <=>  u < T. */
uint64_t kbbq_host_bernoulli_threshold(double p, int32_t *always);

/* Test hook: xoshiro256** state after `ordinal` draws of a sampler seeded with
 * `seed` (jump-ahead used by pass 1). */
int kbbq_rng_state_at(uint32_t seed, uint64_t ordinal, uint64_t state_out[4]);

#ifdef __cplusplus
}
#endif
#endif /* KBBQ_ENGINE_H */
