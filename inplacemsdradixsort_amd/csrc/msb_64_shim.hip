// msb_64_shim.hip -- the reference library's public symbols (include/msb_64.h)
// served by the GPU sort.  Replaces /root/reference/src/msb_64.c:2261-2430 sort(),
// :111-115 mamalloc(), :2470-2505 check().
//
// Host arrays in, host arrays out: stage to the device, run the device-resident
// pair sort, stage back.  The PCIe copies are reported as their own phases and
// are never part of the roofline figure (bench.py times the device-resident
// entry points).  There is no CPU sorting path: without a GPU sort() aborts.
#include "../../include/msb_64.h"
#include "../../include/msd_radix_hip.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace {

[[noreturn]] void die(const char *what, const char *detail)
{
	// the reference asserts (Debug) on contract violations, src/msb_64.c:2266, 2273-2276
	fprintf(stderr, "inpmsdradix_hip: %s%s%s\n", what, detail ? ": " : "", detail ? detail : "");
	abort();
}

#define HIP_OR_DIE(call)                                          \
	do {                                                      \
		hipError_t e_ = (call);                           \
		if (e_ != hipSuccess) die(#call, hipGetErrorString(e_)); \
	} while (0)

std::mutex g_mu;
msd_ctx *g_ctx = nullptr;

msd_ctx *shared_ctx()
{
	if (!g_ctx) {
		int rc = msd_create(&g_ctx, 0, nullptr);
		if (rc != MSD_OK) die("no usable MI355X device for sort()", "msd_create failed (there is no CPU fallback)");
	}
	return g_ctx;
}

uint64_t now_us()
{
	using namespace std::chrono;
	return (uint64_t)duration_cast<microseconds>(steady_clock::now().time_since_epoch()).count();
}

// labels in the reference's style (src/msb_64.c:2402-2411): text padded, ends in ": "
const char *const kLabels[10] = {
	"Host to device copy time: ",
	"Plan and upload time:     ",
	"Classify to blocks time:  ",
	"Block metadata time:      ",
	"Block permutation time:   ",
	"Cleanup heads/tails time: ",
	"Round readback time:      ",
	"LDS local sort time:      ",
	"Device to host copy time: ",
	"Total sort() time:        ",
};
const char *const kPhaseOf[10] = { nullptr, "plan+upload", "A classify", "B metadata", "B block permute",
				   "C cleanup", "readback", "LDS sort", nullptr, nullptr };

} // namespace

extern "C" {

void *mamalloc(size_t size)
{
	void *p = nullptr;
	return posix_memalign(&p, 64, size) ? nullptr : p;
}

void sort(uint64_t **keys, uint64_t **rids, uint64_t *size, int threads, int numa, double fudge,
	  char **description, uint64_t *times)
{
	(void)threads; // the reference demands 64 CPU threads (src/msb_64.c:2266); meaningless here
	if (!keys || !rids || !size) die("sort(): null argument", nullptr);
	if (numa < 1) die("sort(): numa (number of caller arrays) must be >= 1", nullptr);
	if (!(fudge >= 1.0)) die("sort(): fudge must be >= 1.0", nullptr);
	uint64_t total = 0;
	for (int a = 0; a < numa; ++a) {
		if (size[a] && (!keys[a] || !rids[a])) die("sort(): null array", nullptr);
		if (((uintptr_t)keys[a] & 15) || ((uintptr_t)rids[a] & 15))
			die("sort(): arrays must be 16-byte aligned (use mamalloc)", nullptr);
		total += size[a];
	}
	std::lock_guard<std::mutex> lock(g_mu);
	const uint64_t t_begin = now_us();
	msd_ctx *ctx = shared_ctx();
	uint64_t tm[10] = { 0 };
	if (total) {
		uint64_t *dk = nullptr, *dr = nullptr;
		HIP_OR_DIE(hipSetDevice(0));
		HIP_OR_DIE(hipMalloc((void **)&dk, total * sizeof(uint64_t)));
		HIP_OR_DIE(hipMalloc((void **)&dr, total * sizeof(uint64_t)));
		uint64_t t0 = now_us(), off = 0;
		for (int a = 0; a < numa; ++a) {
			if (!size[a]) continue;
			HIP_OR_DIE(hipMemcpy(dk + off, keys[a], size[a] * sizeof(uint64_t), hipMemcpyHostToDevice));
			HIP_OR_DIE(hipMemcpy(dr + off, rids[a], size[a] * sizeof(uint64_t), hipMemcpyHostToDevice));
			off += size[a];
		}
		tm[0] = now_us() - t0;
		msd_set_profiling(ctx, 1);
		int rc = msd_sort_pairs_u64(ctx, dk, dr, total);
		if (rc != MSD_OK) die("sort(): device sort failed", msd_last_error(ctx));
		for (int i = 0; i < msd_phase_count(ctx); ++i)
			for (int j = 1; j <= 7; ++j)
				if (!strcmp(msd_phase_name(ctx, i), kPhaseOf[j]) ||
				    (j == 2 && !strncmp(msd_phase_name(ctx, i), "A ", 2))) // sampling / direct placement count as classify
					tm[j] += (uint64_t)msd_phase_us(ctx, i);
		msd_set_profiling(ctx, 0);
		t0 = now_us();
		off = 0;
		// size[] is left as the caller set it; the reference rewrites it according to its
		// sampled splitters (src/msb_64.c:2180), which callers cannot rely on
		for (int a = 0; a < numa; ++a) {
			if (!size[a]) continue;
			HIP_OR_DIE(hipMemcpy(keys[a], dk + off, size[a] * sizeof(uint64_t), hipMemcpyDeviceToHost));
			HIP_OR_DIE(hipMemcpy(rids[a], dr + off, size[a] * sizeof(uint64_t), hipMemcpyDeviceToHost));
			off += size[a];
		}
		tm[8] = now_us() - t0;
		HIP_OR_DIE(hipFree(dk));
		HIP_OR_DIE(hipFree(dr));
	}
	tm[9] = now_us() - t_begin;
	if (times)
		for (int i = 0; i < 10; ++i) times[i] = tm[i];
	if (description) {
		for (int i = 0; i < 10; ++i) description[i] = const_cast<char *>(kLabels[i]);
		description[10] = nullptr;
	}
}

uint64_t check(uint64_t **keys, uint64_t **rids, uint64_t *size, int numa, int same)
{
	if (!keys || !size || numa < 1) die("check(): bad argument", nullptr);
	uint64_t total = 0;
	for (int a = 0; a < numa; ++a) total += size[a];
	if (!total) return 0;
	std::lock_guard<std::mutex> lock(g_mu);
	msd_ctx *ctx = shared_ctx();
	uint64_t *dk = nullptr, *dr = nullptr;
	HIP_OR_DIE(hipSetDevice(0));
	HIP_OR_DIE(hipMalloc((void **)&dk, total * sizeof(uint64_t)));
	if (same && rids) HIP_OR_DIE(hipMalloc((void **)&dr, total * sizeof(uint64_t)));
	uint64_t off = 0;
	for (int a = 0; a < numa; ++a) {
		if (!size[a]) continue;
		HIP_OR_DIE(hipMemcpy(dk + off, keys[a], size[a] * sizeof(uint64_t), hipMemcpyHostToDevice));
		if (dr) HIP_OR_DIE(hipMemcpy(dr + off, rids[a], size[a] * sizeof(uint64_t), hipMemcpyHostToDevice));
		off += size[a];
	}
	uint64_t bad = 0, sum = 0, xr = 0;
	if (msd_check_u64(ctx, dk, dr, total, &bad, &sum, &xr) != MSD_OK) die("check(): device check failed", msd_last_error(ctx));
	HIP_OR_DIE(hipFree(dk));
	if (dr) HIP_OR_DIE(hipFree(dr));
	if (bad) { // the reference asserts key >= previous key and key == rid (src/msb_64.c:2461-2462)
		char msg[96];
		snprintf(msg, sizeof msg, "%llu order/rid violations", (unsigned long long)bad);
		die("check() failed", msg);
	}
	return sum;
}

} // extern "C"
