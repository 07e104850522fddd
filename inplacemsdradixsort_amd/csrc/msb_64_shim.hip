// msb_64_shim.hip -- the reference library's public symbols (include/msb_64.h)
// served by the GPU sort.  Replaces /root/reference/src/msb_64.c:2261-2430 sort(),
// :111-115 mamalloc(), :2470-2505 check().
//
// Host arrays in, host arrays out: stage to the device (pinned bounce buffers, several
// copy streams, the host-side copies on worker threads), run the device-resident pair
// sort, stage back.  The PCIe copies are reported as their own phases and are never part
// of the roofline figure (bench.py times the device-resident entry points).  There is no
// CPU sorting path: without a GPU sort() aborts.
#include "../../include/msb_64.h"
#include "../../include/msd_radix_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

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
std::string g_last_error;

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
	"Plan, scan and upload time:",
	"Classify to blocks time:  ",
	"Block metadata time:      ",
	"Block permutation time:   ",
	"Cleanup heads/tails time: ",
	"Round readback time:      ",
	"Leaf sorts time:          ",
	"Device to host copy time: ",
	"Total sort() time:        ",
};
// device phase (msd_phase_name) -> slot of times[]; every phase has a slot, so that times[1..7] add up to the
// device time ("A ..." = sampling / histogram / classify; the three leaf sorters share one slot)
int slot_of(const char *phase)
{
	if (!strcmp(phase, "plan+upload") || !strcmp(phase, "bit skip")) return 1;
	if (!strncmp(phase, "A ", 2)) return 2;
	if (!strcmp(phase, "B metadata")) return 3;
	if (!strcmp(phase, "B block permute")) return 4;
	if (!strcmp(phase, "C cleanup")) return 5;
	if (!strcmp(phase, "readback")) return 6;
	return 7; // "LDS sort", "count sort", "big count sort", anything new
}

// ---- staging: host array <-> device, through pinned bounce buffers.  kWorkers threads, each with its own
// stream and two pinned chunks: while chunk i travels by DMA the thread copies chunk i+1 between the caller's
// (pageable) array and its other buffer; several streams keep the PCIe link busy in both phases.
constexpr size_t kChunk = (size_t)16 << 20;
constexpr int kWorkers = 8;

struct Piece { // one contiguous copy: host <-> device
	char *host;
	char *dev;
	size_t bytes;
};

struct Stager {
	char *pinned[kWorkers][2] = {};
	hipStream_t stream[kWorkers] = {};
	hipEvent_t done[kWorkers][2] = {};
	bool ready = false;
	void init()
	{
		if (ready) return;
		for (int w = 0; w < kWorkers; ++w) {
			HIP_OR_DIE(hipStreamCreateWithFlags(&stream[w], hipStreamNonBlocking));
			for (int b = 0; b < 2; ++b) {
				HIP_OR_DIE(hipHostMalloc((void **)&pinned[w][b], kChunk, hipHostMallocDefault));
				HIP_OR_DIE(hipEventCreateWithFlags(&done[w][b], hipEventDisableTiming));
			}
		}
		ready = true;
	}
	// to_device: host -> device, else device -> host
	void run(const std::vector<Piece> &pieces, bool to_device)
	{
		init();
		struct Job { char *host, *dev; size_t bytes; };
		std::vector<Job> jobs;
		for (auto &p : pieces)
			for (size_t o = 0; o < p.bytes; o += kChunk) jobs.push_back({ p.host + o, p.dev + o, std::min(kChunk, p.bytes - o) });
		std::atomic<size_t> next{ 0 };
		std::atomic<int> failed{ 0 };
		auto worker = [&](int w) {
			if (hipSetDevice(0) != hipSuccess) { failed = 1; return; }
			Job prev[2] = {};
			bool pend[2] = { false, false };
			// a buffer is free again when its transfer is over and (towards the host) its content has been copied out
			auto finish = [&](int q) -> bool {
				if (!pend[q]) return true;
				if (hipEventSynchronize(done[w][q]) != hipSuccess) return false;
				if (!to_device) memcpy(prev[q].host, pinned[w][q], prev[q].bytes);
				pend[q] = false;
				return true;
			};
			int b = 0;
			for (;;) {
				const size_t j = next.fetch_add(1);
				if (j >= jobs.size()) break;
				const Job &job = jobs[j];
				if (!finish(b)) { failed = 1; return; }
				hipError_t e;
				if (to_device) {
					memcpy(pinned[w][b], job.host, job.bytes); // (the other buffer's DMA runs meanwhile)
					e = hipMemcpyAsync(job.dev, pinned[w][b], job.bytes, hipMemcpyHostToDevice, stream[w]);
				} else
					e = hipMemcpyAsync(pinned[w][b], job.dev, job.bytes, hipMemcpyDeviceToHost, stream[w]);
				if (e != hipSuccess || hipEventRecord(done[w][b], stream[w]) != hipSuccess) { failed = 1; return; }
				prev[b] = job;
				pend[b] = true;
				b ^= 1;
			}
			if (!finish(0) || !finish(1)) failed = 1;
		};
		const int nw = (int)std::min<size_t>(kWorkers, std::max<size_t>(1, jobs.size()));
		std::vector<std::thread> th;
		for (int w = 1; w < nw; ++w) th.emplace_back(worker, w);
		worker(0);
		for (auto &t : th) t.join();
		if (failed) die("sort()/check(): a staging copy failed", nullptr);
	}
} g_stager;

// first index in [lo, hi) of the sorted device array whose key is >= v (upper = false) or > v (upper = true)
__global__ void bound_kernel(const uint64_t *__restrict__ k, uint64_t lo, uint64_t hi, uint64_t v, int upper, uint64_t *__restrict__ out)
{
	while (lo < hi) {
		const uint64_t mid = (lo + hi) >> 1;
		const bool right = upper ? k[mid] <= v : k[mid] < v;
		if (right) lo = mid + 1; else hi = mid;
	}
	*out = lo;
}

} // namespace

extern "C" {

void *mamalloc(size_t size)
{
	void *p = nullptr;
	return posix_memalign(&p, 64, size) ? nullptr : p;
}

const char *msb_64_last_error(void) { return g_last_error.c_str(); }

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
	g_last_error.clear();
	const uint64_t t_begin = now_us();
	msd_ctx *ctx = shared_ctx();
	uint64_t tm[10] = { 0 };
	if (total) {
		uint64_t *dk = nullptr, *dr = nullptr;
		HIP_OR_DIE(hipSetDevice(0));
		uint64_t t0 = now_us();
		// Out of device memory -- for the two arrays or, further down, for the sort's workspace -- is ONE case with ONE
		// policy: not a contract violation, so nothing is sorted, the caller's arrays stay as they are, times[] come back
		// zero and msb_64_last_error() says why (the API is void, like the reference's).  A caller that would rather stop
		// than go on with unsorted data -- the reference would have died in an assert -- sets MSB_64_ABORT_ON_ERROR=1.
		auto out_of_memory = [&](const char *what) {
			if (dk) (void)hipFree(dk);
			if (dr) (void)hipFree(dr);
			(void)hipGetLastError();
			g_last_error = std::string("sort(): ") + what + "; nothing was sorted";
			const char *ab = getenv("MSB_64_ABORT_ON_ERROR");
			if (ab && atoi(ab) != 0) die(g_last_error.c_str(), nullptr);
			fprintf(stderr, "inpmsdradix_hip: %s\n", g_last_error.c_str());
			if (times) memset(times, 0, 10 * sizeof(uint64_t));
			if (description) {
				for (int i = 0; i < 10; ++i) description[i] = const_cast<char *>(kLabels[i]);
				description[10] = nullptr;
			}
		};
		if (hipMalloc((void **)&dk, total * sizeof(uint64_t)) != hipSuccess ||
		    hipMalloc((void **)&dr, total * sizeof(uint64_t)) != hipSuccess) {
			out_of_memory("the arrays do not fit the device memory");
			return;
		}
		std::vector<Piece> pieces;
		uint64_t off = 0;
		for (int a = 0; a < numa; ++a) {
			if (!size[a]) continue;
			pieces.push_back({ (char *)keys[a], (char *)(dk + off), size[a] * sizeof(uint64_t) });
			pieces.push_back({ (char *)rids[a], (char *)(dr + off), size[a] * sizeof(uint64_t) });
			off += size[a];
		}
		g_stager.run(pieces, true);
		tm[0] = now_us() - t0;
		msd_set_profiling(ctx, 1);
		int rc = msd_sort_pairs_u64(ctx, dk, dr, total);
		if (rc == MSD_ENOMEM) { // (the caller's arrays have not been touched yet: only copies went to the device)
			msd_set_profiling(ctx, 0);
			out_of_memory("no device memory for the sort's workspace");
			return;
		}
		if (rc != MSD_OK) die("sort(): device sort failed", msd_last_error(ctx));
		for (int i = 0; i < msd_phase_count(ctx); ++i) tm[slot_of(msd_phase_name(ctx, i))] += (uint64_t)msd_phase_us(ctx, i);
		msd_set_profiling(ctx, 0);
		t0 = now_us();
		// ---- size[] is rewritten like the reference's (src/msb_64.c:2180: every node ends up with whole key ranges,
		// the sum is preserved, :2379-2383).  Rule here: array a keeps the cut at the end of its input share unless a run of
		// equal keys straddles it; then the cut moves to the nearer end of that run if the growing array's capacity
		// (its input size x fudge, what the reference requires of the caller, :1574-1578) allows, so that -- like the
		// reference's ranges -- no key value is split between two arrays.  fudge = 1.0 leaves size[] unchanged.
		std::vector<uint64_t> cut(numa + 1, 0), cap(numa);
		for (int a = 0; a < numa; ++a) {
			cut[a + 1] = cut[a] + size[a];
			cap[a] = (uint64_t)((double)size[a] * fudge);
		}
		if (numa > 1 && fudge > 1.0) {
			uint64_t *dres = nullptr, h[2];
			HIP_OR_DIE(hipMalloc((void **)&dres, 2 * sizeof(uint64_t)));
			for (int a = 0; a + 1 < numa; ++a) {
				const uint64_t p = cut[a + 1];
				if (p == 0 || p >= total || p <= cut[a]) continue;
				uint64_t edge[2];
				HIP_OR_DIE(hipMemcpy(edge, dk + p - 1, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost));
				if (edge[0] != edge[1]) continue; // the cut already lies between two key values
				hipLaunchKernelGGL(bound_kernel, dim3(1), dim3(1), 0, 0, dk, cut[a], p, edge[0], 0, dres);
				hipLaunchKernelGGL(bound_kernel, dim3(1), dim3(1), 0, 0, dk, p, total, edge[0], 1, dres + 1);
				HIP_OR_DIE(hipMemcpy(h, dres, sizeof h, hipMemcpyDeviceToHost));
				const uint64_t lo = h[0], hi = h[1]; // the run of equal keys is [lo, hi)
				const bool can_hi = hi - cut[a] <= cap[a] && hi <= cut[a + 2 <= numa ? a + 2 : numa];
				const bool can_lo = cut[a + 2 <= numa ? a + 2 : numa] - lo <= cap[a + 1] && lo >= cut[a];
				if (can_hi && (!can_lo || hi - p <= p - lo)) cut[a + 1] = hi;
				else if (can_lo) cut[a + 1] = lo;
			}
			HIP_OR_DIE(hipFree(dres));
		}
		pieces.clear();
		for (int a = 0; a < numa; ++a) {
			size[a] = cut[a + 1] - cut[a];
			if (!size[a]) continue;
			pieces.push_back({ (char *)keys[a], (char *)(dk + cut[a]), size[a] * sizeof(uint64_t) });
			pieces.push_back({ (char *)rids[a], (char *)(dr + cut[a]), size[a] * sizeof(uint64_t) });
		}
		g_stager.run(pieces, false);
		HIP_OR_DIE(hipFree(dk));
		HIP_OR_DIE(hipFree(dr));
		tm[8] = now_us() - t0;
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
	std::vector<Piece> pieces;
	uint64_t off = 0;
	for (int a = 0; a < numa; ++a) {
		if (!size[a]) continue;
		pieces.push_back({ (char *)keys[a], (char *)(dk + off), size[a] * sizeof(uint64_t) });
		if (dr) pieces.push_back({ (char *)rids[a], (char *)(dr + off), size[a] * sizeof(uint64_t) });
		off += size[a];
	}
	g_stager.run(pieces, true);
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
