/* kbbq_exchange.h -- C ABI of the multi-GPU exchange steps of the k-mer BQSR path (libkbbq_engine.so, round 4).
 *
 * What it belongs to: the reference's passes loop over one file in one thread; its pass functions only note that the loops
 * "can be parallelized" (recalibrateutils.hh:32,40).  Here reads shard over the GPUs of a node as contiguous ranges in file
 * order, every GPU holds full replicas of both Bloom filters, and between the pass functions of recalibrateutils.hh:29-44
 * the path has exactly three exchange steps and one broadcast:
 *
 *   after subsample_kmers     OR  all-reduce of the sampled bit array + SUM of its insert counter   kbbq_exchange_filter(e, 0, g)
 *   after find_trusted_kmers  OR  all-reduce of the trusted bit array + SUM of its insert counter   kbbq_exchange_filter(e, 1, g)
 *   after get_covariatedata   SUM all-reduce of the u64 covariate histograms                        kbbq_exchange_histograms(e, g)
 *   after get_dqs             rank 0 trains (host, x87 long double), the int32 delta-Q tables are
 *                             broadcast and installed on every rank                                 kbbq_exchange_dq(e, g)
 *
 * RCCL has no bitwise-OR reduction, so the OR all-reduce is built from what it has, in the direct one-hop form that suits a
 * fully connected xGMI node: the bit array goes in slabs (512 MB by default); a slab is cut into n pieces; one grouped
 * ncclSend/ncclRecv round hands piece j of every rank to rank j (all n-1 links of a GPU carry 1/n of the slab at once),
 * rank j ORs the n pieces with the engine's kernel (kbbq_device_or_pieces), one ncclAllGather returns the reduced pieces.
 * Per rank that moves 2 (n-1)/n of the array.  Everything is queued on the engine's own HIP stream (kbbq_engine_stream): RCCL
 * calls and OR kernels are ordered by the stream, the host does not wait inside the loop.
 *
 * A kbbq_group is one rank's handle of a group of n ranks:
 *   rccl   one process (or thread) per GPU over RCCL -- created from a unique id that the caller distributes itself
 *          (torch.distributed, MPI, a file), or wrapped around an ncclComm_t the caller already has.  RCCL is loaded
 *          with dlopen when the first such group is made: the library has no link-time dependency on it.
 *   local  n ranks = n host threads of ONE process, each with its own engine (on its own device, or several on one): the
 *          same steps through device-to-device copies (peer access over xGMI) and barriers.  What a single-process
 *          multi-device caller uses, and what lets the slab / piece / padding logic be tested with n > 1 on one GPU, where
 *          RCCL refuses two ranks.
 * Results are rank-count invariant (OR and integer sums commute): tests/test_exchange_gpu.py.
 *
 * Plain pointers and sizes; 0 or a negative errno-style code (kbbq_last_error()).  Every rank calls the same functions in the
 * same order; a call returns when the step is complete on this rank (the engine is synchronised).
 */
#ifndef KBBQ_EXCHANGE_H
#define KBBQ_EXCHANGE_H

#include <stddef.h>
#include <stdint.h>

#include "kbbq_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kbbq_group kbbq_group;

#define KBBQ_RCCL_ID_BYTES 128      /* sizeof(ncclUniqueId) */

/* ncclGetUniqueId: call on one rank, hand the bytes to the others. */
int kbbq_group_rccl_unique_id(uint8_t id[KBBQ_RCCL_ID_BYTES]);
/* ncclCommInitRank on `device`; collective: every rank of the group calls it. */
int kbbq_group_rccl_create(const uint8_t id[KBBQ_RCCL_ID_BYTES], int32_t rank, int32_t n_ranks, int32_t device, kbbq_group **out);
/* A communicator the caller made itself (ncclComm_t as void*); not destroyed with the group. */
int kbbq_group_from_nccl_comm(void *nccl_comm, int32_t rank, int32_t n_ranks, kbbq_group **out);
/* n_ranks handles of a group inside this process, one per host thread; out[r] is rank r's.  Destroy every handle. */
int kbbq_group_local_create(int32_t n_ranks, kbbq_group **out);
void kbbq_group_destroy(kbbq_group *g);
int kbbq_group_rank(const kbbq_group *g, int32_t *rank, int32_t *n_ranks);

/* slab_words: 64-bit words of the bit array per round (0: 2^26 = 512 MB).  *inserted_total (optional): the group's count,
 * which is also installed in the engine (kbbq_filter_set_inserted). */
int kbbq_exchange_filter(kbbq_engine *e, int which, kbbq_group *g, uint64_t slab_words, uint64_t *inserted_total);
int kbbq_exchange_histograms(kbbq_engine *e, kbbq_group *g);
int kbbq_exchange_dq(kbbq_engine *e, kbbq_group *g);
/* Milliseconds of the last call of each step on this rank: [0] filter 0, [1] filter 1, [2] histograms, [3] delta-Q. */
int kbbq_exchange_ms(const kbbq_group *g, double out[4]);

#ifdef __cplusplus
}
#endif
#endif /* KBBQ_EXCHANGE_H */
