/*
 * msd_sharded_hip.h -- the multi-GPU entry points of the MI355X in-place MSD radix sort behind a plain C ABI
 * (libinpmsdradix_hip_rccl.so: links libinpmsdradix_hip.so and RCCL).
 *
 * The reference sorts ONE (keys, rids) pair of arrays per memory node in one call -- sort(keys, rids, size, threads,
 * numa, ...), src/msb_64.c:2261-2263: contiguous key ranges per node (numa_dest, :1596-1607), blocks balanced and
 * swapped between the nodes (:1952-2153), then purely local sorting (:2200-2255).  Here a node is a GPU, the block
 * exchange is ONE all-to-all over xGMI (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd), and the entry points
 * come in two shapes:
 *
 *   msd_shard_*            one rank = one GPU = one process (or thread) of the caller, who owns the RCCL communicator
 *                          (ncclComm_t, passed as void *): what inplacemsdradixsort_amd/dist.py does through
 *                          torch.distributed, without Python.
 *   msd_sort_u32_multi     the reference's own calling shape: ONE call, one array per device, all devices of this
 *                          process (ncclCommInitAll, one host thread per device inside the call).
 *
 * Uniform keys only (the top bits of a key name its rank, like the reference's radix bounds (p << 58) - 1, :1555-1557);
 * skewed keys take the sampled splitters of dist.py.  The number of ranks must be a power of two <= 256.
 * All d_* pointers are device pointers on the rank's device, 16-byte aligned.
 * Give the context a stream of its own with HIGH priority (hipStreamCreateWithPriority), or raise GPU_MAX_HW_QUEUES to 8:
 * HIP multiplexes streams onto 4 hardware queues by default, and a compute stream that shares its queue with the stream
 * RCCL launches on serialises the exchange with the local work (measured: DESIGN.md section 6).  No single ncclSend /
 * ncclRecv of these entry points is larger than 512 MiB (RCCL 2.26 moves only half of a 2 GiB message, silently).
 * Return value: 0, a negative MSD_E* code (msd_radix_hip.h), or MSD_EOVERFLOW: some rank's receive buffer is too small
 * for its key range -- decided from the all-gathered send matrix BEFORE the exchange, so every rank returns it and no
 * rank is left alone in a collective; nothing has been exchanged then (the rank's own keys are partly ordered).
 */
#ifndef MSD_SHARDED_HIP_H_
#define MSD_SHARDED_HIP_H_

#include <stddef.h>
#include <stdint.h>

#include "msd_radix_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { MSD_EOVERFLOW = -5, MSD_ERCCL = -6 };

typedef struct msd_shard msd_shard;

/* One rank of a sharded sort: `ctx` (its device, its stream) + the caller's communicator `nccl_comm` (an ncclComm_t
 * whose rank on this process is bound to ctx's device).  Rank and number of ranks are taken from the communicator.
 * nccl_comm == NULL: a single rank (no communicator needed; the calls below then sort locally). */
int msd_shard_create(msd_shard **out, msd_ctx *ctx, void *nccl_comm);
int msd_shard_destroy(msd_shard *sh);
/* "force_exchange" (tests): 1 = a single rank WITH a communicator runs the whole exchange -- count all-gather, grouped
 * send / receive to itself, the leaf over the "arrived" extents -- instead of sorting locally: all of the N > 1 code that
 * one GPU can execute through RCCL. */
/* "low16" (default 1): the fine scheme exchanges only the low halves of the keys -- the upper half of a key is its
 * bucket's number once the shard is ordered by it (msd_pack_low16_u32 / msd_merge_buckets_u32_low16): half the bytes over
 * xGMI for one more pass over the shard.  0 = whole keys travel.  All ranks must use the same value.
 * "hist" (default 1): at up to "hist_max_world" ranks (default 4: at 8 ranks a pair's low halves take less time over their link
 * than the local work they hide behind), when every rank holds at least "hist_min_keys" keys (default 3 * 2^28) and d_recv / d_work hold 2^16
 * records, the buckets travel as HISTOGRAMS of their low halves (msd_hist2_pack_u32 / msd_merge_buckets_u32_hist2 of
 * msd_radix_hip.h: 17408 bytes per source and bucket, a quarter of the whole keys' bytes at 2^30 keys per rank); a rank
 * whose packing overflows (a bucket of more than 65535 keys, more than 255 values with three copies in one bucket) says
 * so with its counts and EVERY rank sends low halves instead.  0 = never.  All ranks must use the same values. */
int msd_shard_set_option(msd_shard *sh, const char *name, int64_t value);
int msd_shard_rank(const msd_shard *sh);
int msd_shard_world(const msd_shard *sh);
const char *msd_shard_last_error(const msd_shard *sh);

/* Sorts the union of all ranks' d_keys (n u32 keys on this rank; ranks may hold different numbers).  Afterwards
 * *d_out points at this rank's sorted key range (*n_out keys; rank r's keys precede rank r + 1's) -- inside d_work
 * when the fine scheme ran, inside d_recv otherwise, d_keys itself for a single rank.
 *   fine scheme   (d_work != NULL, 2..8 ranks, n >= 2^27 on every rank, or forced): the shard is ordered by its top 16
 *                 bits, the 2^16 bucket counts are all-gathered, one all-to-all (of the keys' low halves, packed into
 *                 d_work: option "low16"), one counting pass over the arrived extents writes the sorted buckets into
 *                 d_work (msd_sort_u32_top / msd_bucket_bounds_u32 / msd_pack_low16_u32 / msd_merge_buckets_u32_low16 of
 *                 msd_radix_hip.h).
 *   coarse scheme one in-place top-digit pass, the 256 bucket counts are all-gathered, one all-to-all, the arrived keys
 *                 are sorted in d_recv on their low 32 - log2(ranks) bits (msd_partition_u32 / msd_sort_u32_bits).
 * recv_cap / work_cap: elements; the call returns MSD_EOVERFLOW on every rank if some rank's range does not fit.
 * scheme: 0 = choose, 1 = fine, 2 = coarse (all ranks must pass the same value).  Blocking. */
int msd_sort_u32_sharded(msd_shard *sh, uint32_t *d_keys, uint64_t n, uint32_t *d_recv, uint64_t recv_cap,
			 uint32_t *d_work, uint64_t work_cap, int scheme, uint32_t **d_out, uint64_t *n_out);

/* The reference's own element type: (u64 key, u64 rid) tuples, one (keys, rids) pair per rank.  One in-place pass on
 * the top 8 key bits moves keys and rids together, one count exchange, keys and rids travel in the same all-to-all
 * group, the tuples that arrived are sorted on their low 64 - log2(ranks) bits (the reference's `bits`, :2242).  The
 * result is the first *n_out tuples of d_recv_keys / d_recv_rids (d_keys / d_rids for a single rank: *d_out_* say which). */
int msd_sort_pairs_u64_sharded(msd_shard *sh, uint64_t *d_keys, uint64_t *d_rids, uint64_t n, uint64_t *d_recv_keys,
			       uint64_t *d_recv_rids, uint64_t recv_cap, uint64_t **d_out_keys, uint64_t **d_out_rids,
			       uint64_t *n_out);

/* ONE call for all devices of this process, the reference's calling shape (one array per "node"): device devices[i]
 * holds d_keys[i] (n[i] keys) and owns d_recv[i] / d_work[i] (recv_cap / work_cap elements each; d_work may be NULL:
 * coarse scheme).  Creates the contexts and the communicator (ncclCommInitAll), runs msd_sort_u32_sharded on one host
 * thread per device and tears everything down again; d_out[i] / n_out[i] as above.  ndev: a power of two. */
int msd_sort_u32_multi(int ndev, const int *devices, uint32_t **d_keys, const uint64_t *n, uint32_t **d_recv,
		       uint64_t recv_cap, uint32_t **d_work, uint64_t work_cap, int scheme, uint32_t **d_out, uint64_t *n_out);

#ifdef __cplusplus
}
#endif

#endif
