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
 *     number of caller arrays; fudge only has to be >= 1.0 because the GPU
 *     path needs no slack inside the caller's arrays; size[] is left as it was
 *     (the reference redistributes it by sampled splitters, :2180, which is
 *     implementation-defined).
 *   - the phases reported are the GPU pipeline's (H2D, digit passes, local
 *     sort, D2H), not the CPU block machinery's.
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
