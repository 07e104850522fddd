/*
 * msd_radix_hip.h -- typed, device-resident entry points of the MI355X in-place
 * MSD radix sort (libinpmsdradix_hip.so).  Plain C ABI: pointers and sizes only.
 *
 * The reference exposes one host-pointer call (include/msb_64.h:37-39) and leaks
 * its building blocks as accidental global symbols.  The
 * entry points below are the device-side counterparts of those building blocks;
 * each one names the reference function it stands in for (paths relative to
 * /root/reference).
 *
 * All d_* pointers are DEVICE pointers on the context's device.  All kernels run on the
 * context's stream.  msd_sort_* and msd_partition_* plan every round on the host from the
 * previous round's child list: they block the calling thread once per round (one small
 * device-to-host copy; two rounds for 2^30 u32 keys) and once behind the counting leaf, and
 * return when the last kernels have been launched -- the data are final once the stream has
 * drained.  The building blocks (histogram, scan, generators, sample, splitters) are fully
 * asynchronous; msd_check_* are synchronous (they return values).
 * Return value: 0 on success, a negative MSD_E* code otherwise; msd_last_error() gives the
 * message.  After MSD_EINTERNAL (an invariant check failed between two rounds: a bug) the array
 * holds a permutation of its input that is partitioned by the digits processed so far but not
 * sorted; after MSD_EINVAL / MSD_ENOMEM it is untouched.
 */
#ifndef MSD_RADIX_HIP_H_
#define MSD_RADIX_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct msd_ctx msd_ctx;

enum {
	MSD_OK = 0,
	MSD_EINVAL = -1,   /* bad argument (null pointer, misaligned buffer, bad bit range) */
	MSD_ENOMEM = -2,   /* workspace allocation failed */
	MSD_EHIP = -3,     /* a HIP runtime call failed */
	MSD_EINTERNAL = -4 /* an internal invariant check failed (bug) */
};

/* ---- context ------------------------------------------------------------ */

/* Create a sorting context on `device`; `stream` is a hipStream_t (NULL = the
 * default stream).  The context owns the auxiliary workspace (block map, block
 * lists, stripe leftovers; about 12 % of the data size at 2^30 u32 keys, grown on
 * demand and reused between calls). */
int msd_create(msd_ctx **ctx, int device, void *stream);
int msd_destroy(msd_ctx *ctx);
int msd_set_stream(msd_ctx *ctx, void *stream);
void *msd_get_stream(const msd_ctx *ctx); /* the hipStream_t the context launches on */
int msd_get_device(const msd_ctx *ctx);
/* Pre-allocate the workspace for sorting n elements of key_bytes (+val_bytes)
 * so that the first timed call does not allocate. */
int msd_reserve(msd_ctx *ctx, uint64_t n, int key_bytes, int val_bytes);
uint64_t msd_workspace_bytes(const msd_ctx *ctx);
const char *msd_last_error(const msd_ctx *ctx);
const char *msd_version(void);

/* ---- the sort (reference: sort() src/msb_64.c:2261, core :2232-2244) ----- */

/* In-place MSD radix sort of n keys in device memory.  Keys must be aligned to
 * 16 bytes (the reference asserts the same, src/msb_64.c:2273-2276). */
int msd_sort_u32(msd_ctx *ctx, uint32_t *d_keys, uint64_t n);
int msd_sort_u64(msd_ctx *ctx, uint64_t *d_keys, uint64_t n);
/* (key,rid) tuples in two arrays, the reference's layout (SoA, 64-bit each).
 * Unstable like the reference: equal keys may appear in any order. */
int msd_sort_pairs_u64(msd_ctx *ctx, uint64_t *d_keys, uint64_t *d_rids, uint64_t n);
/* Same, restricted to key bits [0, end_bit): bits at and above end_bit must be
 * equal in all keys (what the reference's `bits` argument of schedule_passes
 * means, src/msb_64.c:1334, :2242 passes 58). */
int msd_sort_u32_bits(msd_ctx *ctx, uint32_t *d_keys, uint64_t n, int end_bit);
int msd_sort_u64_bits(msd_ctx *ctx, uint64_t *d_keys, uint64_t n, int end_bit);
int msd_sort_pairs_u64_bits(msd_ctx *ctx, uint64_t *d_keys, uint64_t *d_rids, uint64_t n, int end_bit);

/* ---- building blocks ----------------------------------------------------- */

/* count[(key >> shift) & (2^radix_bits - 1)]++ over n keys; d_count has
 * 2^radix_bits uint64 entries and is zeroed first; radix_bits <= 12.
 * Reference: histogram() src/msb_64.c:701-738. */
int msd_histogram_u32(msd_ctx *ctx, const uint32_t *d_keys, uint64_t n,
		      unsigned shift, unsigned radix_bits, uint64_t *d_count);
int msd_histogram_u64(msd_ctx *ctx, const uint64_t *d_keys, uint64_t n,
		      unsigned shift, unsigned radix_bits, uint64_t *d_count);

/* Device-wide exclusive prefix sum (single pass, decoupled look-back).
 * Reference: the bucket-offset prefix sums src/msb_64.c:747-750, 799-823 and
 * the cross-thread offset computation 1076-1082.  d_out may equal d_in.  Synchronous: the call
 * waits for the scan and returns MSD_EINTERNAL if a tile's look-back gave up (it polls a bounded
 * number of times so that a lost predecessor cannot hang the device). */
int msd_exclusive_scan_u64(msd_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint64_t n);

/* One in-place digit pass: permute keys so that they are grouped by
 * digit = (key >> shift) & (2^radix_bits - 1), buckets in ascending digit order
 * (unstable).  radix_bits <= 8.  If d_count != NULL it receives the 2^radix_bits
 * bucket sizes (uint64).  Reference: histogram + partition_ip / partition_ip_buf,
 * src/msb_64.c:1023-1027 (740-770, 785-978). */
int msd_partition_u32(msd_ctx *ctx, uint32_t *d_keys, uint64_t n,
		      unsigned shift, unsigned radix_bits, uint64_t *d_count);
int msd_partition_u64(msd_ctx *ctx, uint64_t *d_keys, uint64_t n,
		      unsigned shift, unsigned radix_bits, uint64_t *d_count);
int msd_partition_pairs_u64(msd_ctx *ctx, uint64_t *d_keys, uint64_t *d_rids, uint64_t n,
			    unsigned shift, unsigned radix_bits, uint64_t *d_count);

/* ---- segmented sort and run gather: a rank of the multi-GPU sort after its exchange ----
 * The reference's nodes sort their key ranges locally after the blocks have been balanced and swapped
 * (src/msb_64.c:2200-2255: every node's buckets are whole, local_radixsort takes the remaining `bits`, :2242).
 * Here a rank receives, from every source rank, that source's buckets of the rank's key range in order (one
 * all-to-all); msd_gather_runs_* copies all those runs in ONE launch to a second buffer where every bucket is
 * contiguous (run i: `len[i]` elements from d_src + src_off[i] to d_dst + dst_off[i]; the three arrays are HOST
 * arrays; runs must not overlap in d_dst; d_dst and d_src must not overlap), and msd_sort_*_segments sorts
 * nseg independent segments [seg_off[i], seg_off[i+1]) (HOST array of nseg + 1 ascending element offsets,
 * seg_off[nseg] <= n) on their low `end_bit` bits -- all keys of a segment must agree above them -- in one
 * call: the segments are the parents of the first round, so the local sort starts where the top-digit pass
 * before the exchange left off instead of repeating it. */
int msd_sort_u32_segments(msd_ctx *ctx, uint32_t *d_keys, uint64_t n, const uint64_t *seg_off, uint32_t nseg, int end_bit);
int msd_sort_u64_segments(msd_ctx *ctx, uint64_t *d_keys, uint64_t n, const uint64_t *seg_off, uint32_t nseg, int end_bit);
int msd_sort_pairs_u64_segments(msd_ctx *ctx, uint64_t *d_keys, uint64_t *d_rids, uint64_t n,
				const uint64_t *seg_off, uint32_t nseg, int end_bit);
int msd_gather_runs_u32(msd_ctx *ctx, uint32_t *d_dst, const uint32_t *d_src, const uint64_t *src_off,
			const uint64_t *dst_off, const uint64_t *len, uint32_t nruns);
int msd_gather_runs_u64(msd_ctx *ctx, uint64_t *d_dst, const uint64_t *d_src, const uint64_t *src_off,
			const uint64_t *dst_off, const uint64_t *len, uint32_t nruns);

/* ---- fine-grained sharding: the top digits are sorted BEFORE the exchange, the open bits are counted after it ----
 * A rank of the multi-GPU sort whose keys are evenly spread orders its shard by the top 16 key bits first (two
 * direct-placement rounds -- they run at full speed there, whereas what ARRIVES after an exchange is run-structured),
 * learns the boundaries of its 2^16 buckets, sends every destination rank its range of buckets in ONE all-to-all, and
 * finishes every bucket it received -- nsrc extents, one per source rank -- with one counting pass that reads the
 * extents where they arrived and writes the sorted bucket to its place in a second buffer.  The reference's nodes do
 * the same after their block exchange: whole buckets are sorted locally on the bits that are left
 * (src/msb_64.c:2200-2255, `bits` :2242; range boundaries from the histograms :1546-1564).
 *   msd_sort_*_top: like msd_sort_*_bits, but stops once the keys are ordered by key >> begin_bit (keys that agree
 *     above begin_bit end up adjacent, in any order).  begin_bit = 0 is the full sort.
 *   msd_bucket_bounds_*: d_bounds[b] = first index i with (d_keys[i] >> shift) >= first + b, b = 0 .. nbuckets
 *     (nbuckets + 1 uint64 on the device); the keys must be ordered by key >> shift.  Asynchronous.
 *   msd_merge_buckets_u32: d_counts = nsrc x nbuckets uint64 ON THE DEVICE (row x: the lengths of source x's
 *     extents, bucket by bucket); source x's extents lie back to back in d_src from element src_base[x] on (HOST
 *     array of nsrc offsets: where the all-to-all put source x's keys).  Bucket j holds the keys whose bits above
 *     open_bits equal first_prefix + j; all its keys differ only in their low open_bits <= 16 bits.  The sorted
 *     buckets are written back to back to d_dst (n_expected = the sum of all counts <= dst_cap elements; the call
 *     fails with MSD_EINVAL and writes nothing if the counts do not add up to it).  d_src (src_cap elements) and
 *     d_dst must not overlap.  nsrc <= 8.  Blocks the calling thread until the leaf has run (one small readback:
 *     buckets it did not take -- longer than 17408 keys, more than 255 copies of one key -- are finished by the
 *     general leaves, msd_stat "merge_rejected").
 *   msd_pack_low16_u32 / msd_merge_buckets_u32_low16: once a shard is ordered by its keys' UPPER halves, the upper half of
 *     every key is the number of its bucket, which sender and receiver know from the counts: only the LOW halves need to
 *     cross the links.  msd_pack_low16_u32 writes the low 16 bits of the n keys, in order, to d_out (n uint16; no overlap
 *     with d_keys; asynchronous); msd_merge_buckets_u32_low16 is msd_merge_buckets_u32 with open_bits = 16 for extents of
 *     such low halves (d_src, src_cap and src_base in uint16 elements) and writes whole keys.  Half the exchange volume --
 *     the exchange is what bounds a step at 2 and 4 GPUs (one xGMI link per pair) -- for one more pass over the shard
 *     (6 bytes per key) before it.
 *   msd_hist2_pack_u32 / msd_merge_buckets_u32_hist2: dense buckets (2^14 keys and more per source and bucket: 2^30 keys per
 *     rank) travel as HISTOGRAMS of their low halves instead: per bucket one record of msd_hist2_record_bytes() = 17408
 *     bytes -- 2^16 2-bit counters (0, 1, 2, "3 or more") + up to 255 (value, copies) entries for the values with three or
 *     more copies -- whatever the bucket holds; a quarter of the whole keys' bytes at 2^14 keys per bucket.
 *     msd_hist2_pack_u32 writes the records of buckets 0 .. nbuckets - 1 (d_bounds: nbuckets + 1 boundaries, as from
 *     msd_bucket_bounds_u32 with shift 16) back to back to d_rec and sets *d_overflow (device; cleared first) if a bucket
 *     has more than 65535 keys or more than 255 such values -- the caller must then send the low halves themselves.
 *     Asynchronous.  msd_merge_buckets_u32_hist2: source x's records of this rank's nbuckets buckets lie at
 *     d_rec + x * nbuckets * 17408; d_counts as above (it gives the buckets' places in d_dst); the sum of the
 *     histograms of a bucket IS the sorted bucket. */
int msd_sort_u32_top(msd_ctx *ctx, uint32_t *d_keys, uint64_t n, int end_bit, int begin_bit);
int msd_sort_u64_top(msd_ctx *ctx, uint64_t *d_keys, uint64_t n, int end_bit, int begin_bit);
int msd_sort_pairs_u64_top(msd_ctx *ctx, uint64_t *d_keys, uint64_t *d_rids, uint64_t n, int end_bit, int begin_bit);
int msd_bucket_bounds_u32(msd_ctx *ctx, const uint32_t *d_keys, uint64_t n, unsigned shift, uint64_t first, uint32_t nbuckets, uint64_t *d_bounds);
int msd_bucket_bounds_u64(msd_ctx *ctx, const uint64_t *d_keys, uint64_t n, unsigned shift, uint64_t first, uint32_t nbuckets, uint64_t *d_bounds);
int msd_merge_buckets_u32(msd_ctx *ctx, const uint32_t *d_src, uint64_t src_cap, const uint64_t *d_counts, const uint64_t *src_base,
			  uint32_t nsrc, uint32_t nbuckets, int open_bits, uint32_t first_prefix, uint32_t *d_dst, uint64_t dst_cap,
			  uint64_t n_expected);
int msd_pack_low16_u32(msd_ctx *ctx, const uint32_t *d_keys, uint64_t n, uint16_t *d_out);
/* msd_sort_u32_top(.., 32, 16) + msd_bucket_bounds_u32 + msd_pack_low16_u32 in one go and two passes less: ONE in-place round
 * on the top 8 bits, exact counts of all 2^16 upper halves (d_counts: 65536 uint64 on the device -- the bucket sizes the
 * exchange needs), and the low halves scattered OUT OF PLACE, bucket after bucket (bucket b from the sum of d_counts[0 .. b)
 * on, in any order inside the bucket), into d_out (n uint16, no overlap with d_keys).  d_keys is left ordered by its top 8
 * bits.  Blocks the calling thread for the in-place round; the rest is asynchronous. */
int msd_order_low16_u32(msd_ctx *ctx, uint32_t *d_keys, uint64_t n, uint16_t *d_out, uint64_t *d_counts);
/* ... in two halves: d_counts is complete (in stream order) after the first, so that the caller can start exchanging the
 * counts with the other ranks while the second -- the scatter, 2 ms per 2^30 keys -- runs.  The scatter must be the
 * context's next call after the counts, on the same keys (MSD_EINVAL otherwise). */
int msd_order_low16_counts_u32(msd_ctx *ctx, uint32_t *d_keys, uint64_t n, uint64_t *d_counts);
int msd_order_low16_scatter_u32(msd_ctx *ctx, const uint32_t *d_keys, uint64_t n, uint16_t *d_out);
uint64_t msd_hist2_record_bytes(void);
int msd_hist2_pack_u32(msd_ctx *ctx, const uint32_t *d_keys, uint64_t n, const uint64_t *d_bounds, uint32_t nbuckets, void *d_rec,
		       uint64_t rec_bytes, uint32_t *d_overflow);
/* ... from the LOW HALVES msd_order_low16_u32 has written (d_low: n uint16, bucket after bucket; d_bounds from its counts:
 * msd_bounds_from_counts16 writes the 65537 prefix sums of 65536 counts, asynchronously) -- half the bytes to read. */
int msd_hist2_pack_u32_low16(msd_ctx *ctx, const uint16_t *d_low, uint64_t n, const uint64_t *d_bounds, uint32_t nbuckets, void *d_rec,
			     uint64_t rec_bytes, uint32_t *d_overflow);
int msd_bounds_from_counts16(msd_ctx *ctx, const uint64_t *d_counts, uint64_t *d_bounds);
int msd_merge_buckets_u32_hist2(msd_ctx *ctx, const void *d_rec, uint64_t rec_bytes, const uint64_t *d_counts, uint32_t nsrc,
				uint32_t nbuckets, uint32_t first_prefix, uint32_t *d_dst, uint64_t dst_cap, uint64_t n_expected);
int msd_merge_buckets_u32_low16(msd_ctx *ctx, const uint16_t *d_src, uint64_t src_cap, const uint64_t *d_counts, const uint64_t *src_base,
				uint32_t nsrc, uint32_t nbuckets, uint32_t first_prefix, uint32_t *d_dst, uint64_t dst_cap, uint64_t n_expected);

/* ---- splitter service: sample -> sort (msd_sort_u32) -> delimiters -> range partition ----
 * The reference's front end for skewed keys (src/msb_64.c:1511-1564): a random sample of the
 * UNSORTED data (:1511-1521, index = mulhi(rand64, n); here a counter-based generator, seed + i),
 * the sample is sorted, equi-depth delimiters are picked with the duplicate rule of
 * extract_delimiters (:1304-1322), and the data is partitioned by the lower-bound range
 * function: range p = keys in (delim[p-1], delim[p]] (binary_search_64 :188-204, SIMD :239-351).
 * Used by the multi-GPU path on skewed keys (inplacemsdradixsort_amd/dist.py): one range per GPU.
 *   msd_sample_u32: d_sample[i] = d_keys[mulhi(splitmix64(seed + i), n)], i < m.
 *   msd_splitters_u32: d_delims[0 .. parts-2] from a SORTED sample of m keys; parts <= 256.
 *   msd_partition_by_splitters_u32: one in-place pass; afterwards range 0's keys come first, then
 *     range 1's, ... (unsorted inside a range); d_count (may be NULL) receives `parts` range sizes. */
int msd_sample_u32(msd_ctx *ctx, const uint32_t *d_keys, uint64_t n, uint64_t m, uint64_t seed, uint32_t *d_sample);
int msd_splitters_u32(msd_ctx *ctx, const uint32_t *d_sorted_sample, uint64_t m, unsigned parts, uint32_t *d_delims);
int msd_partition_by_splitters_u32(msd_ctx *ctx, uint32_t *d_keys, uint64_t n, const uint32_t *d_delims,
				   unsigned parts, uint64_t *d_count);
/* the same for what the reference itself sorts: 64-bit keys, alone or with their rids (its sample, its delimiters and
 * its range function all work on 64-bit keys, src/msb_64.c:1511-1564, :1304-1322, :188-204) */
int msd_sample_u64(msd_ctx *ctx, const uint64_t *d_keys, uint64_t n, uint64_t m, uint64_t seed, uint64_t *d_sample);
int msd_splitters_u64(msd_ctx *ctx, const uint64_t *d_sorted_sample, uint64_t m, unsigned parts, uint64_t *d_delims);
int msd_partition_by_splitters_u64(msd_ctx *ctx, uint64_t *d_keys, uint64_t n, const uint64_t *d_delims,
				   unsigned parts, uint64_t *d_count);
int msd_partition_by_splitters_pairs_u64(msd_ctx *ctx, uint64_t *d_keys, uint64_t *d_rids, uint64_t n,
					 const uint64_t *d_delims, unsigned parts, uint64_t *d_count);

/* Verifier, the device form of check() (src/msb_64.c:2432-2505): counts order
 * violations (key[i] < key[i-1]) and, when d_rids != NULL, key != rid
 * mismatches; returns wrap-around sum and xor of the keys.  Synchronous (the
 * three results are written to host memory). */
int msd_check_u32(msd_ctx *ctx, const uint32_t *d_keys, uint64_t n,
		  uint64_t *violations, uint64_t *sum, uint64_t *xr);
int msd_check_u64(msd_ctx *ctx, const uint64_t *d_keys, const uint64_t *d_rids, uint64_t n,
		  uint64_t *violations, uint64_t *sum, uint64_t *xr);

/* Synthetic inputs of SURVEY.md section 8d, generated on the device:
 * key[i] = splitmix64(seed + first + i) >> 32 (u32) or the full word (u64);
 * Zipf(theta=1): key = floor((2^32+1)^u) - 1. */
int msd_gen_uniform_u32(msd_ctx *ctx, uint32_t *d_keys, uint64_t n, uint64_t seed, uint64_t first);
int msd_gen_uniform_u64(msd_ctx *ctx, uint64_t *d_keys, uint64_t n, uint64_t seed, uint64_t first, int shift_right);
int msd_gen_zipf_u32(msd_ctx *ctx, uint32_t *d_keys, uint64_t n, uint64_t seed, uint64_t first);
int msd_gen_iota_u64(msd_ctx *ctx, uint64_t *d_vals, uint64_t n, uint64_t first);
/* duplicates: `distinct` different values with evenly spread digits, each about n / distinct times:
 * key[i] = splitmix64((splitmix64(seed + first + i) mod distinct) ^ 0xD0B1E5) >> 32 */
int msd_gen_dup_u32(msd_ctx *ctx, uint32_t *d_keys, uint64_t n, uint64_t seed, uint64_t first, uint64_t distinct);
/* the reference's own generator, MT19937-64 (src/rand.c:47-86): d_keys[i] = the i-th rand64_next() after
 * rand64_init(seed), shifted right by shift_right -- for cross-checks against a caller that fills its arrays
 * with the reference's RNG.  One workgroup walks the stream (a few GB/s); not a bulk generator. */
int msd_gen_mt19937_64(msd_ctx *ctx, uint64_t *d_keys, uint64_t n, uint64_t seed, int shift_right);

/* ---- pass planner (reference: schedule_passes(), src/msb_64.c:1334-1400) -------------
 * Host-only (no device needed).  The reference plans 1-3 leading passes of <= 9 bits
 * to reach <= 6500-tuple pieces; this planner cuts segments by <= 8-bit digits until they
 * fit the LDS leaf sorters.  Describes the FIRST round for n elements with `end_bit`
 * open key bits; later rounds depend on the data and are planned from the child counts. */
typedef struct msd_plan {
	uint32_t digit_width;      /* bits of the first digit (0: no partition round, LDS leaf only) */
	uint32_t digit_shift;      /* digit = (key >> digit_shift) & (2^digit_width - 1) */
	uint32_t block_elems;      /* elements per 256-byte block */
	uint32_t tile_elems;       /* elements per classify tile */
	uint64_t stripe_elems;     /* elements per stripe (one classify workgroup) */
	uint64_t stripes;          /* stripes of the first round */
	uint64_t leaf_capacity;    /* largest segment the LDS leaf sorter takes */
	uint32_t leaf_count_bits;  /* open bits the one-pass counting leaf can finish (0: not used) */
	uint32_t expected_rounds;  /* rounds for uniformly distributed keys */
	uint64_t workspace_bytes;  /* per-round + per-call auxiliary memory for this shape */
} msd_plan;
int msd_plan_first_round(uint64_t n, int key_bytes, int val_bytes, int end_bit, int compute_units, msd_plan *out);

/* ---- tuning knobs (no counterpart in the reference) --------------------------------
 * "direct_mode": 0 = every round classifies into the workgroup's own stripe and permutes
 *   all blocks afterwards; 1 (default) = the first round of a large input whose sampled
 *   top-digit buckets are about equally big writes its blocks straight into the bucket's
 *   estimated region and permutes only the misplaced ones; 2 = the same without the sample test.
 *   Rounds after the first follow (from exact per-parent digit counts) if the first round did.
 * "direct_min": smallest round (elements) direct placement is tried on (default 2^22).
 * "direct_min_parent": rounds after the first: smallest parent segment (default 2^17).
 * "direct_kernel": accepted and ignored (round 1's first version of the direct kernel is gone;
 *   profiles/r02_sq_counters.json and r02_stamps_classify_direct_before.json keep its measurements).
 * "count16": u32 keys with 16 open bits: 1 (default) = count_place16_kernel for segments of about 2^14
 *   keys, 2 = always, 0 = never (count_place_kernel).
 * "leaf17": u64 keys and tuples: 1 (default) = segments of <= 17408 elements are finished in one pass by leaf17_kernel (read
 *   once, sorted in registers and LDS, written once), 0 = tuples: register partition + the small leaves, u64 keys:
 *   leaf_count_sort_kernel (round 2).
 * "stream_kernel": the streaming classify of rounds that do not place directly: 2 (default) = classify_stream2_kernel
 *   (one fetch-add per key, no per-key second pass), 1 = round 2's classify_kernel.
 * "mid_leaf": u32 keys: 1 (default) = counting-leaf segments the register-resident kernels do not take (17 Ki .. 128 Ki
 *   keys, crowded ones) are finished by the 16-bit-counter leaf (merge_count_kernel) instead of count_walk_kernel; 0 = never.
 * "merge_leaf": msd_merge_buckets_u32: 0 (default) = by bucket size, 1 = the register-resident leaf, 2 = the 16-bit-counter leaf.
 * "regpart": u64 keys / tuples: 1 (default) = segments of <= 17408 elements take the register-resident
 *   partition pass (csrc/msd_regpart.hpp) instead of a general round, 0 = never. */
int msd_set_option(msd_ctx *ctx, const char *name, int64_t value);

/* ---- phase report (reference: description[]/times[], src/msb_64.c:2402-2412) */

/* Enable per-phase hipEvent timing for subsequent sorts on this context
 * (adds synchronisation; off by default). */
int msd_set_profiling(msd_ctx *ctx, int enabled);
/* Number of phases recorded by the last sort; names/us arrays of that length. */
int msd_phase_count(const msd_ctx *ctx);
const char *msd_phase_name(const msd_ctx *ctx, int i);
double msd_phase_us(const msd_ctx *ctx, int i);
/* Counters of the last sort: rounds of in-place partitioning, blocks moved,
 * small segments, ... (for tests and DESIGN.md tables). */
int msd_stat(const msd_ctx *ctx, const char *name, uint64_t *value);

#ifdef __cplusplus
}
#endif

#endif
