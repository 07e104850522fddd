/*
 * msb_64.h -- drop-in declaration of the reference library's public surface,
 * served by the MI355X implementation (libinpmsdradix_hip.so).
 *
 * Replaces /root/reference/include/msb_64.h:37-41 (sort, mamalloc) and adds the
 * reference's de-facto test hook check() (defined at src/msb_64.c:2470 but
 * missing from its header).  Unlike the reference header this one includes
 * <stdint.h>/<stddef.h> and is extern "C" safe.
 *
 * Semantics kept from the reference (src/msb_64.c:2261-2430):
 *   - keys[0..numa), rids[0..numa): caller-owned host arrays, array a holds
 *     size[a] valid (key,rid) tuples; after the call the concatenation
 *     keys[0] || keys[1] || ... is non-decreasing and rids carry the same
 *     permutation; the sort is unstable; sum of size[] is preserved.
 *   - description[0..9] receive pointers to static phase labels,
 *     description[10] = NULL; times[0..9] receive microseconds per phase.
 *   - void return; a violated precondition aborts with a message (the
 *     reference asserts, src/msb_64.c:2266, 2273-2276).
 * Re-interpreted (documented in INTEGRATION.md):
 *   - threads is accepted and ignored (the reference demands 64); numa is the
 *     number of caller arrays (all of them are sorted on device 0: one GPU
 *     holds 2^33 tuples); fudge only has to be >= 1.0.
 *   - size[] IS rewritten, as the reference does (src/msb_64.c:2180, sum preserved :2379-2383): the
 *     reference gives every node whole key ranges; here array a keeps the cut at the end of its
 *     input share unless a run of equal keys straddles it -- then the cut moves to the nearer end
 *     of that run, provided the growing array stays within its capacity size[a] * fudge (what the
 *     reference requires the caller to allocate, :1574-1578).  So with fudge > 1 no key value is
 *     split between two arrays; with fudge = 1.0 size[] comes back unchanged.
 *   - the phases reported are the GPU pipeline's: times[0] host-to-device staging (pinned,
 *     chunked, several copy streams), [1..7] the device phases (every device phase has a slot:
 *     they add up to the device time), [8] device-to-host staging, [9] the whole call.
 *   - out of device memory (for the two arrays, or for the sort's workspace): nothing is sorted, the
 *     caller's arrays are untouched, a message goes to stderr, times[] come back zero and
 *     msb_64_last_error() says why.  sort() is void, as in the reference: a caller MUST look at
 *     msb_64_last_error() after sort() (empty string = sorted) -- or set MSB_64_ABORT_ON_ERROR=1 in the
 *     environment to get the reference's behaviour (it would have died in an assert) instead.
 */
#ifndef MSB_64_H_HIP_
#define MSB_64_H_HIP_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference: include/msb_64.h:37-39, src/msb_64.c:2261-2263 */
void sort(uint64_t **keys, uint64_t **rids, uint64_t *size,
	  int threads, int numa, double fudge,
	  char **description, uint64_t *times);

/* empty string after a successful sort(), else why the last one did nothing (no counterpart in the reference).
 * Check it after every sort(): the call is void and returns normally when the device memory does not suffice. */
const char *msb_64_last_error(void);

/* reference: include/msb_64.h:41, src/msb_64.c:111-115 (64-byte aligned, free()) */
void *mamalloc(size_t size);

/* reference: src/msb_64.c:2470-2505.  Returns the wrap-around sum of all keys;
 * aborts with a message if keys are not non-decreasing over the concatenation
 * of the arrays or (same != 0) some key != rid.  The order check covers every
 * adjacent pair, including the slice boundaries the reference skips (:2458). */
uint64_t check(uint64_t **keys, uint64_t **rids, uint64_t *size, int numa, int same);

#ifdef __cplusplus
}
#endif

#endif
