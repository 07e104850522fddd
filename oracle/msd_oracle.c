/*
 * msd_oracle.c -- CPU restatement of the reference's single-thread in-place MSD
 * radix core.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP path.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it; the product library
 * (inplacemsdradixsort_amd/csrc) never links, loads or calls anything in here.
 *
 * Every function restates (in plain scalar C99, no SSE, no inline asm) what a
 * function of the reference does and cites it as src/msb_64.c:<lines> relative
 * to /root/reference.  Parity status: PINNED -- tests/test_oracle_vs_reference.py
 * checks this restatement against the reference itself compiled from its own
 * sources into oracle/_ref/ (recipe: oracle/Makefile), and tests/golden/ holds
 * vectors produced by that reference build (script: tests/golden/make_golden.py).
 * The reference ships no golden vectors of its own.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define ORC_CACHE_LIMIT 6500u /* tuples that "fit cache": src/msb_64.c:1337 */
#define ORC_SMALL_CUTOFF 20u  /* insertion-sort cutoff:   src/msb_64.c:1011 */
#define ORC_HIST_CAP 4096u    /* hist/offset scratch:     src/msb_64.c:2234-2238 */
#define ORC_LEVELS 5

/* ------------------------------------------------------------------ helpers */

/* smallest p with 2^p >= x  (src/msb_64.c:1324-1330) */
static int orc_ceil_log2(uint64_t x)
{
	int p = 0;
	while (((uint64_t)1 << p) < x) ++p;
	return p;
}

static uint64_t orc_div_up(uint64_t a, uint64_t b) /* src/msb_64.c:1332 */
{
	return (a + b - 1) / b;
}

static inline uint64_t orc_digit(uint64_t key, unsigned shift, uint64_t mask)
{
	return (key >> shift) & mask;
}

/* ------------------------------------------------------------- pass planner */

/*
 * Pass plan for one call of the single-thread core (src/msb_64.c:1334-1400).
 * Emits p radix passes (width radix_bits[i], buffered[i] in {0,1}), then the
 * sentinel buffered[p] = -1 with radix_bits[p] = number of low "tail" bits that
 * are left to comb/insertion sort.  Returns p.  The leading passes cut the input
 * into pieces of <= 6500 tuples; the last pass is an in-cache pass of
 * ceil_log2(piece) - 2 bits.
 */
int orc_schedule_passes(uint64_t size, int bits, int8_t *radix_bits, int8_t *buffered)
{
	int np = 0;
	int lg = orc_ceil_log2(orc_div_up(size, ORC_CACHE_LIMIT));
#define ORC_PASS(width, is_buf)                      \
	do {                                         \
		radix_bits[np] = (int8_t)(width);    \
		buffered[np] = (int8_t)(is_buf);     \
		++np;                                \
	} while (0)
	if (size <= ORC_CACHE_LIMIT) {
		/* no leading pass */
	} else if (lg <= 5) { /* one in-cache split, 3..5 bits */
		int w = lg < 3 ? 3 : lg;
		if (w > bits) w = bits;
		ORC_PASS(w, 0);
	} else if (lg <= 9) { /* one buffered split */
		ORC_PASS(lg, 1);
	} else if (lg <= 12) { /* 3 in-cache bits, then buffered */
		ORC_PASS(3, 0);
		ORC_PASS(lg - 3, 1);
	} else if (lg <= 14) { /* buffered, then 5 in-cache bits */
		ORC_PASS(lg - 5, 1);
		ORC_PASS(5, 0);
	} else if (lg <= 18) { /* two buffered halves */
		ORC_PASS(lg >> 1, 1);
		ORC_PASS((lg + 1) >> 1, 1);
	} else { /* lg <= 27: three buffered passes */
		int first = lg / 3;
		int rest = lg - first;
		ORC_PASS(first, 1);
		ORC_PASS(rest >> 1, 1);
		ORC_PASS((rest + 1) >> 1, 1);
	}
	for (int i = 0; i < np; ++i) {
		size >>= radix_bits[i];
		bits -= radix_bits[i];
	}
	int last = orc_ceil_log2(size) - 2;
	if (last > bits) last = bits;
	bits -= last;
	ORC_PASS(last, 0);
#undef ORC_PASS
	buffered[np] = -1;
	radix_bits[np] = (int8_t)bits;
	return np;
}

/* ---------------------------------------------------------------- histogram */

/* count[(key >> shift) & (2^radix_bits - 1)]++ ; count zeroed first
 * (src/msb_64.c:701-738; the SSE body is an unrolled form of this loop) */
void orc_histogram(const uint64_t *keys, uint64_t n, uint64_t *count,
		   unsigned shift, unsigned radix_bits)
{
	uint64_t parts = (uint64_t)1 << radix_bits, mask = parts - 1;
	memset(count, 0, parts * sizeof(uint64_t));
	for (uint64_t i = 0; i < n; ++i)
		count[orc_digit(keys[i], shift, mask)]++;
}

/* u32-key form used for the GPU histogram parity tests (same arithmetic on a
 * 32-bit key; the reference has no 32-bit entry point) */
void orc_histogram_u32(const uint32_t *keys, uint64_t n, uint64_t *count,
		       unsigned shift, unsigned radix_bits)
{
	uint64_t parts = (uint64_t)1 << radix_bits, mask = parts - 1;
	memset(count, 0, parts * sizeof(uint64_t));
	for (uint64_t i = 0; i < n; ++i)
		count[((uint64_t)keys[i] >> shift) & mask]++;
}

/* exclusive prefix sum (bucket *start* offsets); the reference keeps inclusive
 * sums = bucket ends (src/msb_64.c:747-750, 799-823); start[p] = end[p]-count[p] */
void orc_exclusive_scan(const uint64_t *count, uint64_t *start, uint64_t parts)
{
	uint64_t run = 0;
	for (uint64_t p = 0; p < parts; ++p) {
		start[p] = run;
		run += count[p];
	}
}

/* ------------------------------------------------ in-cache cycle-leader pass */

/*
 * In-place partition by one digit, American-flag style, buckets filled from
 * their ends (src/msb_64.c:740-770).  cursor[p] starts at the END of bucket p
 * and moves down.  A cycle starts at the first slot `head` of the first bucket
 * that is not finished; the carried tuple is dropped at --cursor[digit] and the
 * tuple found there is carried on, until the drop lands on `head` itself.
 */
void orc_partition_ip(uint64_t *keys, uint64_t *rids, uint64_t n,
		      const uint64_t *sizes, uint64_t *cursor,
		      unsigned shift, unsigned radix_bits)
{
	uint64_t parts = (uint64_t)1 << radix_bits, mask = parts - 1;
	uint64_t run = 0, p;
	for (p = 0; p < parts; ++p) {
		run += sizes[p];
		cursor[p] = run;
	}
	if (n == 0) return;
	uint64_t head = 0;
	p = 0;
	while (sizes[p] == 0) ++p;
	for (;;) {
		uint64_t ck = keys[head], cr = rids[head], at;
		do {
			at = --cursor[orc_digit(ck, shift, mask)];
			uint64_t tk = keys[at], tr = rids[at];
			keys[at] = ck;
			rids[at] = cr;
			ck = tk;
			cr = tr;
		} while (at != head);
		/* skip buckets whose cursor has come all the way down */
		do {
			head += sizes[p++];
		} while (p != parts && head == cursor[p]);
		if (p == parts) break;
	}
}

/* --------------------------------------------- out-of-cache (buffered) pass */

/*
 * The reference's out-of-cache variant (src/msb_64.c:785-978) runs the same
 * cycle-leader permutation but serves every bucket's *current 8-tuple line*
 * from a 128-byte software write-combining buffer, streams the line out when
 * it is full and pre-loads the 8 tuples below it.  Functionally that is a
 * one-line-per-bucket write-back cache in front of the arrays.  The restatement
 * below models exactly that: a line cache keyed by bucket, lines aligned to
 * 8 tuples of the (virtually 64-byte aligned) array, write-back on line change
 * and at the end.  Result: the same partition; the tie order inside a bucket is
 * whatever the cycle-leader order gives, which the sort's output does not pin
 * (the sort is unstable).
 */
typedef struct {
	uint64_t k[8], r[8];
	uint64_t line; /* index>>3 of the cached line, or ~0 */
	uint8_t dirty;
} orc_line_t;

typedef struct {
	uint64_t *keys, *rids;
	const uint64_t *beg, *end; /* bucket extents */
	orc_line_t *ln;
} orc_cache_t;

static void orc_line_flush(orc_cache_t *c, uint64_t b)
{
	orc_line_t *l = &c->ln[b];
	if (l->line == ~(uint64_t)0 || !l->dirty) return;
	uint64_t lo = l->line << 3;
	for (unsigned j = 0; j < 8; ++j) {
		uint64_t pos = lo + j;
		if (pos >= c->beg[b] && pos < c->end[b]) {
			c->keys[pos] = l->k[j];
			c->rids[pos] = l->r[j];
		}
	}
	l->dirty = 0;
}

static orc_line_t *orc_line_get(orc_cache_t *c, uint64_t b, uint64_t pos)
{
	orc_line_t *l = &c->ln[b];
	if (l->line != (pos >> 3)) {
		orc_line_flush(c, b);
		l->line = pos >> 3;
		uint64_t lo = l->line << 3;
		for (unsigned j = 0; j < 8; ++j) {
			uint64_t q = lo + j;
			if (q >= c->beg[b] && q < c->end[b]) {
				l->k[j] = c->keys[q];
				l->r[j] = c->rids[q];
			}
		}
	}
	return l;
}

void orc_partition_ip_buf(uint64_t *keys, uint64_t *rids, uint64_t n,
			  const uint64_t *sizes, unsigned shift, unsigned radix_bits)
{
	uint64_t parts = (uint64_t)1 << radix_bits, mask = parts - 1;
	if (n == 0) return;
	uint64_t *beg = malloc(parts * sizeof(uint64_t));
	uint64_t *end = malloc(parts * sizeof(uint64_t));
	uint64_t *cursor = malloc(parts * sizeof(uint64_t));
	orc_line_t *ln = malloc(parts * sizeof(orc_line_t));
	uint64_t run = 0, p;
	for (p = 0; p < parts; ++p) {
		beg[p] = run;
		run += sizes[p];
		end[p] = cursor[p] = run;
		ln[p].line = ~(uint64_t)0;
		ln[p].dirty = 0;
	}
	orc_cache_t c = { keys, rids, beg, end, ln };
	uint64_t head = 0;
	p = 0;
	while (sizes[p] == 0) ++p;
	for (;;) {
		/* the cycle head lies in bucket p: read it through p's line */
		orc_line_t *hl = orc_line_get(&c, p, head);
		uint64_t ck = hl->k[head & 7], cr = hl->r[head & 7], at;
		do {
			uint64_t b = orc_digit(ck, shift, mask);
			at = --cursor[b];
			orc_line_t *l = orc_line_get(&c, b, at);
			uint64_t tk = l->k[at & 7], tr = l->r[at & 7];
			l->k[at & 7] = ck;
			l->r[at & 7] = cr;
			l->dirty = 1;
			ck = tk;
			cr = tr;
		} while (at != head);
		do {
			head += sizes[p++];
		} while (p != parts && head == cursor[p]);
		if (p == parts) break;
	}
	for (p = 0; p < parts; ++p) orc_line_flush(&c, p);
	free(ln);
	free(cursor);
	free(end);
	free(beg);
}

/* -------------------------------------------------------- tiny-bucket sorts */

/* insertion sort on (key,rid), used for <= 20 tuples (src/msb_64.c:126-149) */
void orc_insertsort(uint64_t *keys, uint64_t *rids, uint64_t n)
{
	for (uint64_t i = 1; i < n; ++i) {
		uint64_t k = keys[i], r = rids[i], j = i;
		while (j > 0 && keys[j - 1] > k) {
			keys[j] = keys[j - 1];
			rids[j] = rids[j - 1];
			--j;
		}
		keys[j] = k;
		rids[j] = r;
	}
}

/* comb sort, shrink 0.77 in single precision (src/msb_64.c:980-1005) */
void orc_combsort(uint64_t *keys, uint64_t *rids, uint64_t n)
{
	const float shrink = 0.77f;
	uint64_t gap = (uint64_t)(n * shrink);
	if (n < 2) return;
	for (;;) {
		int swapped = 0;
		for (uint64_t i = 0, j = gap; j < n; ++i, ++j) {
			if (keys[i] > keys[j]) {
				uint64_t t = keys[i];
				keys[i] = keys[j];
				keys[j] = t;
				t = rids[i];
				rids[i] = rids[j];
				rids[j] = t;
				swapped = 1;
			}
		}
		if (gap > 1)
			gap = (uint64_t)(gap * shrink);
		else if (!swapped)
			break;
		if (gap == 0) gap = 1;
	}
}

/* --------------------------------------------------------- recursion driver */

typedef struct {
	const int8_t *bits;     /* cumulative: bits[d] = key bits still unsorted at depth d */
	const int8_t *buffered; /* 1 buffered, 0 in-cache, -1 tail */
	uint64_t *hist[ORC_LEVELS];
	uint64_t *offs[ORC_LEVELS];
} orc_plan_t;

/* DFS over buckets (src/msb_64.c:1007-1035) */
static void orc_recurse(uint64_t *keys, uint64_t *rids, uint64_t n,
			const orc_plan_t *pl, int depth)
{
	if (n <= ORC_SMALL_CUTOFF) {
		orc_insertsort(keys, rids, n);
		return;
	}
	if (pl->buffered[depth] < 0) {
		orc_combsort(keys, rids, n);
		return;
	}
	unsigned shift = (unsigned)pl->bits[depth + 1];
	unsigned width = (unsigned)(pl->bits[depth] - pl->bits[depth + 1]);
	uint64_t parts = (uint64_t)1 << width;
	uint64_t *h = pl->hist[depth];
	orc_histogram(keys, n, h, shift, width);
	if (pl->buffered[depth])
		orc_partition_ip_buf(keys, rids, n, h, shift, width);
	else
		orc_partition_ip(keys, rids, n, h, pl->offs[depth], shift, width);
	if (shift == 0) return;
	uint64_t at = 0;
	for (uint64_t b = 0; b < parts; ++b) {
		orc_recurse(keys + at, rids + at, h[b], pl, depth + 1);
		at += h[b];
	}
}

/*
 * One call of the reference's single-thread core on (keys,rids)[0..n) whose
 * keys differ only in their low `bits` bits: plan, make the plan cumulative,
 * recurse -- the driver lines src/msb_64.c:2232-2244.  Writes the plan it used
 * to plan_bits/plan_buf (8 entries each) when they are non-NULL.  Returns the
 * number of radix passes.
 */
int orc_sort_pairs_u64(uint64_t *keys, uint64_t *rids, uint64_t n, int bits,
		       int8_t *plan_bits, int8_t *plan_buf)
{
	int8_t rb[8] = { 0 }, bf[8] = { 0 };
	if (n == 0) return 0;
	int np = orc_schedule_passes(n, bits, rb, bf);
	if (plan_bits) memcpy(plan_bits, rb, 8);
	if (plan_buf) memcpy(plan_buf, bf, 8);
	for (int i = np; i-- > 0;) rb[i] = (int8_t)(rb[i] + rb[i + 1]);
	orc_plan_t pl;
	pl.bits = rb;
	pl.buffered = bf;
	for (int i = 0; i < ORC_LEVELS; ++i) {
		pl.hist[i] = malloc(ORC_HIST_CAP * sizeof(uint64_t));
		pl.offs[i] = malloc(ORC_HIST_CAP * sizeof(uint64_t));
	}
	orc_recurse(keys, rids, n, &pl, 0);
	for (int i = 0; i < ORC_LEVELS; ++i) {
		free(pl.hist[i]);
		free(pl.offs[i]);
	}
	return np;
}

/* u32 keys the way SURVEY.md section 8c prescribes for config C1: zero-extend
 * to u64, rid = key, bits = 32, sort, narrow back */
int orc_sort_u32(uint32_t *keys, uint64_t n)
{
	if (n == 0) return 0;
	uint64_t *k = malloc(n * sizeof(uint64_t));
	uint64_t *r = malloc(n * sizeof(uint64_t));
	if (!k || !r) {
		free(k);
		free(r);
		return -1;
	}
	for (uint64_t i = 0; i < n; ++i) k[i] = r[i] = keys[i];
	int np = orc_sort_pairs_u64(k, r, n, 32, NULL, NULL);
	for (uint64_t i = 0; i < n; ++i) keys[i] = (uint32_t)k[i];
	free(k);
	free(r);
	return np;
}

/* u64 keys only (rid = key), all 64 bits significant */
int orc_sort_u64(uint64_t *keys, uint64_t n)
{
	if (n == 0) return 0;
	uint64_t *r = malloc(n * sizeof(uint64_t));
	if (!r) return -1;
	memcpy(r, keys, n * sizeof(uint64_t));
	int np = orc_sort_pairs_u64(keys, r, n, 64, NULL, NULL);
	free(r);
	return np;
}

/* ------------------------------------------------------------- verification */

/*
 * The reference's acceptance check (src/msb_64.c:2432-2505) restated for `numa`
 * caller arrays: keys non-decreasing inside every array and across array
 * boundaries, key == rid when `same`, returns the wrap-around sum of keys.
 * Unlike the reference (whose asserts vanish under NDEBUG and which restarts
 * the order check at every thread slice, :2458) this reports violations:
 * *violations = number of order breaks + key!=rid mismatches.
 */
uint64_t orc_check(uint64_t **keys, uint64_t **rids, const uint64_t *size,
		   int numa, int same, uint64_t *violations, uint64_t *xor_out)
{
	uint64_t sum = 0, x = 0, bad = 0, prev = 0;
	for (int a = 0; a < numa; ++a) {
		for (uint64_t i = 0; i < size[a]; ++i) {
			uint64_t k = keys[a][i];
			if (k < prev) ++bad;
			if (same && rids && rids[a][i] != k) ++bad;
			sum += k;
			x ^= k;
			prev = k;
		}
	}
	if (violations) *violations = bad;
	if (xor_out) *xor_out = x;
	return sum;
}

/* ----------------------------------------------------- synthetic generators */

/* counter-based generator of SURVEY.md section 8d */
static inline uint64_t orc_splitmix64(uint64_t x)
{
	x += 0x9E3779B97F4A7C15ull;
	uint64_t z = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

void orc_gen_uniform_u32(uint32_t *out, uint64_t n, uint64_t seed, uint64_t first)
{
	for (uint64_t i = 0; i < n; ++i)
		out[i] = (uint32_t)(orc_splitmix64(seed + first + i) >> 32);
}

void orc_gen_uniform_u64(uint64_t *out, uint64_t n, uint64_t seed, uint64_t first)
{
	for (uint64_t i = 0; i < n; ++i) out[i] = orc_splitmix64(seed + first + i);
}

/* Zipf(theta = 1) over ranks 1..2^32 by the inverse CDF of the continuous 1/x
 * law: rank = floor((U+1)^u), key = rank-1 */
void orc_gen_zipf_u32(uint32_t *out, uint64_t n, uint64_t seed, uint64_t first)
{
	const double U = 4294967296.0;
	const double lnU1 = log(U + 1.0);
	for (uint64_t i = 0; i < n; ++i) {
		double u = (double)(orc_splitmix64(seed + first + i) >> 11) * 0x1.0p-53;
		double r = floor(exp(u * lnU1));
		if (r < 1.0) r = 1.0;
		if (r > U) r = U;
		out[i] = (uint32_t)((uint64_t)r - 1);
	}
}

/* ------------------------------------------------ splitter front end (skew) */

/* high 64 bits of a 64 x 64 bit product (src/msb_64.c:178-186: mulq) */
static inline uint64_t orc_mulhi(uint64_t a, uint64_t b)
{
	return (uint64_t)(((unsigned __int128)a * b) >> 64);
}

/* random sample of an unsorted array: sample[p] = keys[mulhi(rand64, size)] (src/msb_64.c:1511-1521).
 * The reference draws rand64 from its MT19937-64 with an uninitialised seed;
 * the build's counter-based generator stands in: rand64 = splitmix64(seed + p). */
void orc_sample_u32(const uint32_t *keys, uint64_t n, uint64_t m, uint64_t seed, uint32_t *out)
{
	for (uint64_t p = 0; p < m; ++p) out[p] = keys[orc_mulhi(orc_splitmix64(seed + p), n)];
}

/* the same for the 64-bit keys the reference itself sorts (its sample is 64-bit, src/msb_64.c:1511-1521) */
void orc_sample_u64(const uint64_t *keys, uint64_t n, uint64_t m, uint64_t seed, uint64_t *out)
{
	for (uint64_t p = 0; p < m; ++p) out[p] = keys[orc_mulhi(orc_splitmix64(seed + p), n)];
}

/* extract_delimiters (src/msb_64.c:1304-1322) for `parts` ranges (parts - 1 delimiters):
 * delimiter i = sample[(uint64)(percentile * (i+1) - 0.001)], percentile = sample_size / parts; if the run
 * of equal values around the pick extends further behind it than in front of it (and the value is
 * not 0) the delimiter is value - 1. */
void orc_extract_delimiters(const uint64_t *sample, uint64_t sample_size, uint64_t parts, uint64_t *delimiter)
{
	double percentile = sample_size * 1.0 / parts;
	for (uint64_t i = 0; i + 1 < parts; ++i) {
		uint64_t index = (uint64_t)(percentile * (i + 1) - 0.001);
		uint64_t start, end;
		delimiter[i] = sample[index];
		for (start = index; start; --start)
			if (sample[start] != delimiter[i]) break;
		for (end = index; end != sample_size; ++end)
			if (sample[end] != delimiter[i]) break;
		if (index - start < end - index && delimiter[i]) delimiter[i]--;
	}
}

/* lower-bound range function (binary_search_64, src/msb_64.c:188-204): number of delimiters < key,
 * i.e. range p holds the keys in (delimiter[p-1], delimiter[p]] */
uint64_t orc_range_of(const uint64_t *delimiter, uint64_t ndelim, uint64_t key)
{
	uint64_t low = 0, high = ndelim;
	while (low < high) {
		uint64_t mid = (low + high) >> 1;
		if (key > delimiter[mid]) low = mid + 1;
		else high = mid;
	}
	return low;
}

/* range sizes of a u32 array under `ndelim` delimiters (counts[0 .. ndelim]) */
void orc_range_histogram_u32(const uint32_t *keys, uint64_t n, const uint64_t *delimiter, uint64_t ndelim, uint64_t *counts)
{
	for (uint64_t p = 0; p <= ndelim; ++p) counts[p] = 0;
	for (uint64_t i = 0; i < n; ++i) counts[orc_range_of(delimiter, ndelim, keys[i])]++;
}

/* range sizes of a u64 array (the reference's own key type) under `ndelim` delimiters */
void orc_range_histogram_u64(const uint64_t *keys, uint64_t n, const uint64_t *delimiter, uint64_t ndelim, uint64_t *counts)
{
	for (uint64_t p = 0; p <= ndelim; ++p) counts[p] = 0;
	for (uint64_t i = 0; i < n; ++i) counts[orc_range_of(delimiter, ndelim, keys[i])]++;
}

/* duplicates generator of include/msd_radix_hip.h (msd_gen_dup_u32) */
void orc_gen_dup_u32(uint32_t *out, uint64_t n, uint64_t seed, uint64_t first, uint64_t distinct)
{
	for (uint64_t i = 0; i < n; ++i)
		out[i] = (uint32_t)(orc_splitmix64((orc_splitmix64(seed + first + i) % distinct) ^ 0xD0B1E5ull) >> 32);
}

/* MT19937-64 as the reference implements it (src/rand.c:47-86): rand64_init(seed), then n x rand64_next() */
void orc_mt19937_64(uint64_t *out, uint64_t n, uint64_t seed)
{
	uint64_t num[313], x;
	size_t index = 312, i;
	num[0] = seed;
	for (i = 0; i != 311; ++i) num[i + 1] = 6364136223846793005ull * (num[i] ^ (num[i] >> 62)) + i + 1;
	for (uint64_t k = 0; k < n; ++k) {
		if (index == 312) {
			i = 0;
			do {
				x = (num[i] & 0xffffffff80000000ull) | (num[i + 1] & 0x7fffffffull);
				num[i] = num[i + 156] ^ (x >> 1) ^ (0xb5026f5aa96619e9ull & (0ull - (x & 1)));
			} while (++i != 156);
			num[312] = num[0];
			do {
				x = (num[i] & 0xffffffff80000000ull) | (num[i + 1] & 0x7fffffffull);
				num[i] = num[i - 156] ^ (x >> 1) ^ (0xb5026f5aa96619e9ull & (0ull - (x & 1)));
			} while (++i != 312);
			index = 0;
		}
		x = num[index++];
		x ^= (x >> 29) & 0x5555555555555555ull;
		x ^= (x << 17) & 0x71d67fffeda60000ull;
		x ^= (x << 37) & 0xfff7eee000000000ull;
		x ^= (x >> 43);
		out[k] = x;
	}
}
