/*
 * benchmark_msb_64.c -- stand-in for the reference's benchmark driver, which its Makefile names
 * (/root/reference/Makefile:20-21) but which is absent from the repository.  Plain C against
 * include/msb_64.h: allocate with mamalloc, fill (key, rid = key) tuples, sort(), check(), print the
 * phase report the way a caller of the reference would.  Links against libinpmsdradix_hip.so:
 *
 *   gcc -O2 -Iinclude examples/benchmark_msb_64.c -Linplacemsdradixsort_amd -linpmsdradix_hip \
 *       -Wl,-rpath,$PWD/inplacemsdradixsort_amd -o benchmark_msb_64
 *   ./benchmark_msb_64 [log2_tuples=24] [arrays=2]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/time.h>

#include "msb_64.h"

static uint64_t splitmix64(uint64_t x) /* generator of SURVEY.md section 8d */
{
	x += 0x9E3779B97F4A7C15ull;
	uint64_t z = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

static double now(void)
{
	struct timeval t;
	gettimeofday(&t, NULL);
	return t.tv_sec + t.tv_usec * 1e-6;
}

int main(int argc, char **argv)
{
	int logn = argc > 1 ? atoi(argv[1]) : 24;
	int numa = argc > 2 ? atoi(argv[2]) : 2;
	double fudge = 2.0; /* what the reference needs; this library touches only size[a] tuples */
	uint64_t n = (uint64_t)1 << logn, per = n / numa, sum = 0;
	uint64_t **keys = malloc(numa * sizeof *keys), **rids = malloc(numa * sizeof *rids);
	uint64_t *size = malloc(numa * sizeof *size);
	for (int a = 0; a < numa; ++a) {
		size[a] = a + 1 == numa ? n - per * a : per;
		keys[a] = mamalloc((size_t)(size[a] * fudge) * sizeof(uint64_t));
		rids[a] = mamalloc((size_t)(size[a] * fudge) * sizeof(uint64_t));
		if (!keys[a] || !rids[a]) return 2;
		for (uint64_t i = 0; i < size[a]; ++i) {
			uint64_t k = splitmix64(0x5EED0005ull + per * a + i);
			keys[a][i] = rids[a][i] = k;
			sum += k;
		}
	}
	char *description[11];
	uint64_t times[10];
	{ /* warm-up on a throw-away array: device context, workspace and pinned staging buffers are created once per process */
		uint64_t wn = 1 << 16, *wk = mamalloc(wn * 2 * sizeof(uint64_t)), *wr = mamalloc(wn * 2 * sizeof(uint64_t));
		uint64_t *wks[1] = { wk }, *wrs[1] = { wr }, ws[1] = { wn };
		if (!wk || !wr) return 2;
		for (uint64_t i = 0; i < wn; ++i) wk[i] = wr[i] = splitmix64(i);
		sort(wks, wrs, ws, 64, 1, fudge, description, times);
		free(wk);
		free(wr);
	}
	double t0 = now();
	sort(keys, rids, size, 64, numa, fudge, description, times);
	double dt = now() - t0;
	uint64_t checksum = check(keys, rids, size, numa, 1); /* aborts on an order or key != rid violation */
	for (int i = 0; description[i]; ++i) printf("%s%10lu us\n", description[i], (unsigned long)times[i]);
	printf("tuples %lu  wall %.3f s  %.1f Mtuples/s (host arrays, PCIe copies included)\n",
	       (unsigned long)n, dt, n / dt / 1e6);
	printf("checksum %s\n", checksum == sum ? "ok" : "MISMATCH");
	for (int a = 0; a < numa; ++a) {
		free(keys[a]);
		free(rids[a]);
	}
	return checksum == sum ? 0 : 1;
}
