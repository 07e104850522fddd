// msd_sharded.hip -- multi-GPU entry points behind the C ABI (include/msd_sharded_hip.h): what
// inplacemsdradixsort_amd/dist.py does through torch.distributed, with RCCL called directly.
// Built into libinpmsdradix_hip_rccl.so (links libinpmsdradix_hip.so and librccl); the single-GPU library does not
// depend on RCCL.
//
// Reference: sort() takes one (keys, rids) pair per memory node (src/msb_64.c:2261-2263), gives every node a contiguous
// group of key ranges (numa_dest, :1596-1607), exchanges blocks between the nodes (:1952-2153) and sorts locally
// (:2200-2255).  Here: node = GPU, the exchange = one ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd over xGMI.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/msd_sharded_hip.h"

namespace {

constexpr int kFineBits = 16;                  // the fine scheme orders a shard by its top 16 bits before the exchange
constexpr uint32_t kFineBuckets = 1u << kFineBits;
constexpr uint64_t kFineMinKeys = 1ull << 27;  // ... when every rank holds at least this many keys
constexpr uint64_t kMaxPieceBytes = 1ull << 29; // largest single ncclSend / ncclRecv (see all_to_all)
constexpr uint64_t kHistMinKeys = 3ull << 28;   // the buckets travel as histogram records when every rank holds at least this many keys
constexpr uint64_t kHistReady = 1ull << 63;     // ... and says so in the top bit of the "keys held" word of its row

} // namespace

struct msd_shard {
	msd_ctx *ctx = nullptr;
	ncclComm_t comm = nullptr;
	int rank = 0, world = 1, device = 0;
	hipStream_t stream = nullptr;
	// device scratch: [own counts | capacity | n][all ranks' rows][send matrix + capacities + sizes][mine: world x buckets / world]
	uint64_t *d_row = nullptr, *d_all = nullptr, *d_small = nullptr, *d_mine = nullptr, *d_bounds = nullptr;
	uint64_t *h_small = nullptr; // pinned
	bool force_exchange = false; // (tests) a single rank with a communicator goes through the whole exchange instead of sorting locally
	bool low16 = true;           // fine scheme: only the low halves of the keys are exchanged ("low16" = 0: whole keys; all ranks alike)
	bool hist = true;            // fine scheme: dense buckets are exchanged as histogram records ("hist" = 0: never; all ranks alike)
	uint64_t hist_min = 3ull << 28; // ("hist_min_keys": tests)
	int hist_max_world = 4;         // ("hist_max_world")
	uint32_t *d_flag = nullptr;
	std::string err;
};

namespace {

int fail(msd_shard *sh, int code, const char *fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	if (sh) sh->err = buf;
	return code;
}

#define SH_HIP(sh, call)                                                                                       \
	do {                                                                                                   \
		hipError_t e_ = (call);                                                                        \
		if (e_ != hipSuccess) return fail(sh, MSD_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
	} while (0)
#define SH_NCCL(sh, call)                                                                                        \
	do {                                                                                                     \
		ncclResult_t r_ = (call);                                                                        \
		if (r_ != ncclSuccess) return fail(sh, MSD_ERCCL, "%s failed: %s", #call, ncclGetErrorString(r_)); \
	} while (0)
#define SH_MSD(sh, call)                                                                        \
	do {                                                                                    \
		int rc_ = (call);                                                               \
		if (rc_ != MSD_OK) return fail(sh, rc_, "%s: %s", #call, msd_last_error(sh->ctx)); \
	} while (0)

constexpr uint32_t kRowLen = kFineBuckets + 2; // bucket counts, receive capacity, keys held

// row[b] = bounds[b + 1] - bounds[b]; row[nb] = capacity; row[nb + 1] = n
__global__ void counts_row_kernel(const uint64_t *__restrict__ bounds, uint32_t nb, uint64_t cap, uint64_t n, uint64_t *__restrict__ row)
{
	const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b < nb) row[b] = bounds[b + 1] - bounds[b];
	if (b == nb) row[nb] = cap;
	if (b == nb + 1) row[nb + 1] = n;
}

// the row's "keys held" word gets the mark "my buckets are ready as histogram records" unless the packing overflowed
__global__ void hist_ready_kernel(const uint32_t *__restrict__ overflow, uint64_t *__restrict__ held)
{
	if (*overflow == 0) *held |= kHistReady;
}

__global__ void row_tail_kernel(uint32_t nb, uint64_t cap, uint64_t n, uint64_t *__restrict__ row)
{
	row[nb] = cap;
	row[nb + 1] = n;
}

// small[src * world + dst] = keys source src holds for destination dst; small[world^2 + r] = capacity of rank r;
// small[world^2 + world + r] = keys rank r holds.  One workgroup per (src, dst).
__global__ __launch_bounds__(256) void send_matrix_kernel(const uint64_t *__restrict__ all, uint32_t row_len, uint32_t nb, uint32_t world,
	uint64_t *__restrict__ small)
{
	__shared__ uint64_t part[256];
	const uint32_t src = blockIdx.x / world, dst = blockIdx.x % world, per = nb / world;
	uint64_t s = 0;
	for (uint32_t j = threadIdx.x; j < per; j += 256) s += all[(size_t)src * row_len + dst * per + j];
	part[threadIdx.x] = s;
	__syncthreads();
	for (uint32_t o = 128; o; o >>= 1) {
		if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		small[blockIdx.x] = part[0];
		if (dst == 0) {
			small[(size_t)world * world + src] = all[(size_t)src * row_len + nb];
			small[(size_t)world * world + world + src] = all[(size_t)src * row_len + nb + 1];
		}
	}
}

// mine[src][j] = all[src][me * per + j]: the extents this rank receives, source-major
__global__ void mine_kernel(const uint64_t *__restrict__ all, uint32_t row_len, uint32_t per, uint32_t me, uint32_t world, uint64_t *__restrict__ mine)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= per * world) return;
	const uint32_t src = i / per, j = i % per;
	mine[i] = all[(size_t)src * row_len + (size_t)me * per + j];
}

int log2_exact(int g)
{
	int b = 0;
	while ((1 << b) < g) ++b;
	return (1 << b) == g ? b : -1;
}

// All ranks learn the send matrix (rows of `row_len` uint64: nb bucket counts, the receive capacity, the keys held); this
// rank's row and column come back as element counts.  Returns MSD_EOVERFLOW -- on every rank -- if some rank's total
// exceeds its capacity.
int exchange_counts(msd_shard *sh, uint32_t nb, uint32_t row_len, const char *what, std::vector<uint64_t> &send_cnt, std::vector<uint64_t> &recv_cnt)
{
	const int W = sh->world;
	SH_NCCL(sh, ncclAllGather(sh->d_row, sh->d_all, row_len, ncclUint64, sh->comm, sh->stream));
	hipLaunchKernelGGL(send_matrix_kernel, dim3(W * W), dim3(256), 0, sh->stream, (const uint64_t *)sh->d_all, row_len, nb, (uint32_t)W, sh->d_small);
	SH_HIP(sh, hipGetLastError());
	SH_HIP(sh, hipMemcpyAsync(sh->h_small, sh->d_small, ((size_t)W * W + 2 * W) * sizeof(uint64_t), hipMemcpyDeviceToHost, sh->stream));
	SH_HIP(sh, hipStreamSynchronize(sh->stream));
	const uint64_t *m = sh->h_small, *cap = m + (size_t)W * W;
	send_cnt.assign(W, 0);
	recv_cnt.assign(W, 0);
	std::string over;
	for (int r = 0; r < W; ++r) {
		uint64_t tot = 0;
		for (int s = 0; s < W; ++s) tot += m[(size_t)s * W + r];
		if (tot > cap[r]) {
			char b[96];
			snprintf(b, sizeof b, "%s%d: %llu %s for capacity %llu", over.empty() ? "" : ", ", r, (unsigned long long)tot, what, (unsigned long long)cap[r]);
			over += b;
		}
		send_cnt[r] = m[(size_t)sh->rank * W + r];
		recv_cnt[r] = m[(size_t)r * W + sh->rank];
	}
	if (!over.empty()) return fail(sh, MSD_EOVERFLOW, "receive buffer too small on rank(s) %s", over.c_str());
	return MSD_OK;
}

// the coarse schemes' row: 256 top-digit counts (written by msd_partition_*), then capacity and size
int finish_coarse_row(msd_shard *sh, uint64_t cap, uint64_t n)
{
	sh->h_small[0] = cap;
	sh->h_small[1] = n;
	SH_HIP(sh, hipMemcpyAsync(sh->d_row + 256, sh->h_small, 2 * sizeof(uint64_t), hipMemcpyHostToDevice, sh->stream));
	SH_HIP(sh, hipStreamSynchronize(sh->stream)); // (the pinned words are reused by the count exchange)
	return MSD_OK;
}

template <typename T>
int all_to_all(msd_shard *sh, const T *src, T *dst, const std::vector<uint64_t> &send_cnt, const std::vector<uint64_t> &recv_cnt, bool group_open)
{
	// (16-bit elements travel as twice as many bytes: RCCL has no 16-bit integer type)
	const ncclDataType_t ty = sizeof(T) == 4 ? ncclUint32 : sizeof(T) == 8 ? ncclUint64 : ncclUint8;
	const uint64_t per = sizeof(T) == 2 ? 2 : 1; // RCCL elements per element of T (uint8_t: bytes as they are)
	if (!group_open) SH_NCCL(sh, ncclGroupStart());
	// RCCL 2.26 (ROCm 7) moves only half of a single send / receive of >= 2 GiB, silently (tools/debug/a2a_big.py: measured on
	// the GPU box, through torch.distributed and through ncclSend / ncclRecv alike): every pair's block goes in pieces of
	// at most 512 MiB -- several sends to one peer inside a group are matched with its receives in order.
	const uint64_t lim = kMaxPieceBytes / sizeof(T);
	uint64_t so = 0, ro = 0;
	for (int p = 0; p < sh->world; ++p) {
		for (uint64_t a = 0; a < send_cnt[p]; a += lim)
			SH_NCCL(sh, ncclSend(src + so + a, std::min(lim, send_cnt[p] - a) * per, ty, p, sh->comm, sh->stream));
		for (uint64_t a = 0; a < recv_cnt[p]; a += lim)
			SH_NCCL(sh, ncclRecv(dst + ro + a, std::min(lim, recv_cnt[p] - a) * per, ty, p, sh->comm, sh->stream));
		so += send_cnt[p];
		ro += recv_cnt[p];
	}
	if (!group_open) SH_NCCL(sh, ncclGroupEnd());
	return MSD_OK;
}

} // namespace

extern "C" {

int msd_shard_create(msd_shard **out, msd_ctx *ctx, void *nccl_comm)
{
	if (!out || !ctx) return MSD_EINVAL;
	*out = nullptr;
	msd_shard *sh = new msd_shard();
	sh->ctx = ctx;
	sh->comm = (ncclComm_t)nccl_comm;
	sh->device = msd_get_device(ctx);
	sh->stream = (hipStream_t)msd_get_stream(ctx);
	if (sh->comm) {
		if (ncclCommUserRank(sh->comm, &sh->rank) != ncclSuccess || ncclCommCount(sh->comm, &sh->world) != ncclSuccess) {
			delete sh;
			return MSD_ERCCL;
		}
	}
	if (log2_exact(sh->world) < 0 || sh->world > 256) {
		delete sh;
		return MSD_EINVAL; // the radix split needs a power of two
	}
	if (hipSetDevice(sh->device) != hipSuccess) {
		delete sh;
		return MSD_EHIP;
	}
	const size_t W = (size_t)sh->world;
	bool ok = hipMalloc((void **)&sh->d_row, kRowLen * sizeof(uint64_t)) == hipSuccess &&
		  hipMalloc((void **)&sh->d_all, W * kRowLen * sizeof(uint64_t)) == hipSuccess &&
		  hipMalloc((void **)&sh->d_small, (W * W + 2 * W) * sizeof(uint64_t)) == hipSuccess &&
		  hipMalloc((void **)&sh->d_mine, (size_t)kFineBuckets * sizeof(uint64_t)) == hipSuccess &&
		  hipMalloc((void **)&sh->d_bounds, ((size_t)kFineBuckets + 3) * sizeof(uint64_t)) == hipSuccess && // (+ the packing's overflow flag)
		  hipHostMalloc((void **)&sh->h_small, (W * W + 2 * W) * sizeof(uint64_t), hipHostMallocDefault) == hipSuccess;
	if (!ok) {
		msd_shard_destroy(sh);
		return MSD_ENOMEM;
	}
	sh->d_flag = reinterpret_cast<uint32_t *>(sh->d_bounds + kFineBuckets + 1);
	*out = sh;
	return MSD_OK;
}

int msd_shard_destroy(msd_shard *sh)
{
	if (!sh) return MSD_EINVAL;
	(void)hipSetDevice(sh->device);
	(void)hipStreamSynchronize(sh->stream);
	for (uint64_t *p : { sh->d_row, sh->d_all, sh->d_small, sh->d_mine, sh->d_bounds })
		if (p) (void)hipFree(p);
	if (sh->h_small) (void)hipHostFree(sh->h_small);
	delete sh;
	return MSD_OK;
}

int msd_shard_set_option(msd_shard *sh, const char *name, int64_t value)
{
	if (!sh || !name) return MSD_EINVAL;
	if (!strcmp(name, "force_exchange")) {
		sh->force_exchange = value != 0;
		return MSD_OK;
	}
	if (!strcmp(name, "low16")) {
		sh->low16 = value != 0;
		return MSD_OK;
	}
	if (!strcmp(name, "hist")) {
		sh->hist = value != 0;
		return MSD_OK;
	}
	if (!strcmp(name, "hist_min_keys")) {
		sh->hist_min = (uint64_t)value;
		return MSD_OK;
	}
	if (!strcmp(name, "hist_max_world")) {
		sh->hist_max_world = (int)value;
		return MSD_OK;
	}
	return fail(sh, MSD_EINVAL, "unknown option %s", name);
}

int msd_shard_rank(const msd_shard *sh) { return sh ? sh->rank : -1; }
int msd_shard_world(const msd_shard *sh) { return sh ? sh->world : 0; }
const char *msd_shard_last_error(const msd_shard *sh) { return sh ? sh->err.c_str() : "null shard"; }

int msd_sort_u32_sharded(msd_shard *sh, uint32_t *d_keys, uint64_t n, uint32_t *d_recv, uint64_t recv_cap, uint32_t *d_work,
			 uint64_t work_cap, int scheme, uint32_t **d_out, uint64_t *n_out)
{
	if (!sh || !d_out || !n_out) return MSD_EINVAL;
	if (n && !d_keys) return fail(sh, MSD_EINVAL, "sort_u32_sharded: null keys");
	if (scheme < 0 || scheme > 2) return fail(sh, MSD_EINVAL, "sort_u32_sharded: scheme must be 0, 1 or 2");
	SH_HIP(sh, hipSetDevice(sh->device));
	sh->stream = (hipStream_t)msd_get_stream(sh->ctx);
	const int W = sh->world, lg = log2_exact(W);
	if (W == 1 && !(sh->force_exchange && sh->comm)) {
		SH_MSD(sh, msd_sort_u32(sh->ctx, d_keys, n));
		SH_HIP(sh, hipStreamSynchronize(sh->stream));
		*d_out = d_keys;
		*n_out = n;
		return MSD_OK;
	}
	if (!d_recv) return fail(sh, MSD_EINVAL, "sort_u32_sharded: null receive buffer");
	std::vector<uint64_t> send_cnt, recv_cnt;
	// The scheme must be the same on every rank: it is decided from what every rank can see before any exchange -- the
	// arguments all ranks pass alike (scheme, work buffer or not) -- and, for scheme 0, from the smallest shard, which
	// every rank learns from the all-gathered rows; the fine scheme's rows are therefore produced only after a first
	// tiny all-gather of the shard sizes when scheme == 0.
	bool fine = scheme == 1 || (scheme == 0 && d_work != nullptr && W <= 8);
	if (scheme == 0 && fine) {
		sh->h_small[0] = n;
		SH_HIP(sh, hipMemcpyAsync(sh->d_row, sh->h_small, sizeof(uint64_t), hipMemcpyHostToDevice, sh->stream));
		SH_HIP(sh, hipStreamSynchronize(sh->stream));
		SH_NCCL(sh, ncclAllGather(sh->d_row, sh->d_all, 1, ncclUint64, sh->comm, sh->stream));
		SH_HIP(sh, hipMemcpyAsync(sh->h_small, sh->d_all, (size_t)W * sizeof(uint64_t), hipMemcpyDeviceToHost, sh->stream));
		SH_HIP(sh, hipStreamSynchronize(sh->stream));
		for (int r = 0; r < W; ++r) fine = fine && sh->h_small[r] >= kFineMinKeys;
	}
	if (fine && (!d_work || W > 8)) return fail(sh, MSD_EINVAL, "sort_u32_sharded: the fine scheme needs a work buffer and at most 8 ranks");
	if (fine) {
		const uint32_t per = kFineBuckets / (uint32_t)W;
		const uint64_t rec_total = (uint64_t)kFineBuckets * msd_hist2_record_bytes();
		// Dense buckets travel as HISTOGRAMS of their low halves (msd_hist2_pack_u32: one record of 17408 bytes per bucket,
		// whatever it holds -- a quarter of the whole keys' bytes at 2^14 keys per bucket), packed into the work buffer; a rank
		// whose packing did not overflow says so in its row, and the records travel only if every rank's did not.
		// (up to 4 ranks: there a pair's low halves take longer over their link than the local work they hide behind; at 8 they
		// do not, and records cost 1.3 ms more to prepare -- the figures are in inplacemsdradixsort_amd/dist.py at FINE_HIST_MAX_WORLD)
		const bool want_hist = sh->hist && W <= sh->hist_max_world && n >= sh->hist_min && work_cap * 4 >= rec_total && recv_cap * 4 >= rec_total;
		// Otherwise only the keys' LOW halves travel: once the shard is ordered by the upper halves, the upper half of a key is
		// its bucket's number, which the receiver knows from the counts -- half the bytes over the links (one xGMI link per
		// pair of GPUs: at 2^30 keys per rank the exchange of whole keys outlasts a rank's local work at 2 and 4 GPUs) and for
		// the leaf to read.  msd_order_low16_u32 orders and packs in one go (one in-place round, exact counts, the low halves
		// scattered into the work buffer -- dead until the leaf writes it).
		const bool low16 = sh->low16 && work_cap * 2 >= n;
		bool packed = false;
		// (records are packed from the low halves, behind them in the work buffer, when both fit)
		const uint64_t rec_at = (2 * n + 255) / 256 * 256;
		unsigned char *d_rec = (unsigned char *)d_work;
		if (low16 && (!want_hist || work_cap * 4 >= rec_at + rec_total)) {
			SH_MSD(sh, msd_order_low16_u32(sh->ctx, d_keys, n, (uint16_t *)d_work, sh->d_row)); // (the row's first 2^16 words: the bucket sizes)
			hipLaunchKernelGGL(row_tail_kernel, dim3(1), dim3(1), 0, sh->stream, kFineBuckets, recv_cap < work_cap ? recv_cap : work_cap, n, sh->d_row);
			SH_HIP(sh, hipGetLastError());
			packed = true;
			if (want_hist) {
				d_rec += rec_at;
				SH_MSD(sh, msd_bounds_from_counts16(sh->ctx, sh->d_row, sh->d_bounds));
				SH_MSD(sh, msd_hist2_pack_u32_low16(sh->ctx, (const uint16_t *)d_work, n, sh->d_bounds, kFineBuckets, d_rec, work_cap * 4 - rec_at, sh->d_flag));
			}
		} else {
			SH_MSD(sh, msd_sort_u32_top(sh->ctx, d_keys, n, 32, 32 - kFineBits));
			SH_MSD(sh, msd_bucket_bounds_u32(sh->ctx, d_keys, n, 32 - kFineBits, 0, kFineBuckets, sh->d_bounds));
			hipLaunchKernelGGL(counts_row_kernel, dim3((kRowLen + 255) / 256), dim3(256), 0, sh->stream, (const uint64_t *)sh->d_bounds, kFineBuckets,
					   recv_cap < work_cap ? recv_cap : work_cap, n, sh->d_row);
			SH_HIP(sh, hipGetLastError());
		}
		if (want_hist) {
			if (!packed) SH_MSD(sh, msd_hist2_pack_u32(sh->ctx, d_keys, n, sh->d_bounds, kFineBuckets, d_work, work_cap * 4, sh->d_flag));
			hipLaunchKernelGGL(hist_ready_kernel, dim3(1), dim3(1), 0, sh->stream, (const uint32_t *)sh->d_flag, sh->d_row + kFineBuckets + 1);
			SH_HIP(sh, hipGetLastError());
		}
		int rc = exchange_counts(sh, kFineBuckets, kRowLen, "keys", send_cnt, recv_cnt);
		if (rc) return rc;
		bool use_hist = true;
		for (int r = 0; r < W; ++r) use_hist = use_hist && (sh->h_small[(size_t)W * W + W + r] & kHistReady) != 0;
		if (use_hist) {
			std::vector<uint64_t> blocks(W, rec_total / (uint64_t)W);
			rc = all_to_all<uint8_t>(sh, (const uint8_t *)d_rec, (uint8_t *)d_recv, blocks, blocks, false);
			if (rc) return rc;
			hipLaunchKernelGGL(mine_kernel, dim3((kFineBuckets + 255) / 256), dim3(256), 0, sh->stream, (const uint64_t *)sh->d_all, kRowLen, per,
					   (uint32_t)sh->rank, (uint32_t)W, sh->d_mine);
			SH_HIP(sh, hipGetLastError());
			uint64_t m = 0;
			for (int s2 = 0; s2 < W; ++s2) m += recv_cnt[s2];
			SH_MSD(sh, msd_merge_buckets_u32_hist2(sh->ctx, d_recv, recv_cap * 4, sh->d_mine, (uint32_t)W, per, (uint32_t)sh->rank * per, d_work, work_cap, m));
			SH_HIP(sh, hipStreamSynchronize(sh->stream));
			*d_out = d_work;
			*n_out = m;
			return MSD_OK;
		}
		if (low16) {
			if (!packed) SH_MSD(sh, msd_pack_low16_u32(sh->ctx, d_keys, n, (uint16_t *)d_work)); // (the records did not travel after all)
			rc = all_to_all<uint16_t>(sh, (const uint16_t *)d_work, (uint16_t *)d_recv, send_cnt, recv_cnt, false);
		} else
			rc = all_to_all<uint32_t>(sh, d_keys, d_recv, send_cnt, recv_cnt, false);
		if (rc) return rc;
		hipLaunchKernelGGL(mine_kernel, dim3((kFineBuckets + 255) / 256), dim3(256), 0, sh->stream, (const uint64_t *)sh->d_all, kRowLen, per,
				   (uint32_t)sh->rank, (uint32_t)W, sh->d_mine);
		SH_HIP(sh, hipGetLastError());
		std::vector<uint64_t> base(W);
		uint64_t m = 0;
		for (int s = 0; s < W; ++s) {
			base[s] = m;
			m += recv_cnt[s];
		}
		if (low16)
			SH_MSD(sh, msd_merge_buckets_u32_low16(sh->ctx, (const uint16_t *)d_recv, recv_cap * 2, sh->d_mine, base.data(), (uint32_t)W, per,
							       (uint32_t)sh->rank * per, d_work, work_cap, m));
		else
			SH_MSD(sh, msd_merge_buckets_u32(sh->ctx, d_recv, recv_cap, sh->d_mine, base.data(), (uint32_t)W, per, 32 - kFineBits,
							 (uint32_t)sh->rank * per, d_work, work_cap, m));
		SH_HIP(sh, hipStreamSynchronize(sh->stream));
		*d_out = d_work;
		*n_out = m;
		return MSD_OK;
	}
	// coarse: one top-digit pass; the 256 bucket counts go into the row in front of capacity and size
	SH_MSD(sh, msd_partition_u32(sh->ctx, d_keys, n, 24, 8, sh->d_row));
	int rc = finish_coarse_row(sh, recv_cap, n);
	if (!rc) rc = exchange_counts(sh, 256, 258, "keys", send_cnt, recv_cnt);
	if (rc) return rc;
	rc = all_to_all<uint32_t>(sh, d_keys, d_recv, send_cnt, recv_cnt, false);
	if (rc) return rc;
	uint64_t m = 0;
	for (int s = 0; s < W; ++s) m += recv_cnt[s];
	SH_MSD(sh, msd_sort_u32_bits(sh->ctx, d_recv, m, 32 - lg));
	SH_HIP(sh, hipStreamSynchronize(sh->stream));
	*d_out = d_recv;
	*n_out = m;
	return MSD_OK;
}

int msd_sort_pairs_u64_sharded(msd_shard *sh, uint64_t *d_keys, uint64_t *d_rids, uint64_t n, uint64_t *d_recv_keys,
			       uint64_t *d_recv_rids, uint64_t recv_cap, uint64_t **d_out_keys, uint64_t **d_out_rids, uint64_t *n_out)
{
	if (!sh || !d_out_keys || !d_out_rids || !n_out) return MSD_EINVAL;
	if (n && (!d_keys || !d_rids)) return fail(sh, MSD_EINVAL, "sort_pairs_u64_sharded: null arrays");
	SH_HIP(sh, hipSetDevice(sh->device));
	sh->stream = (hipStream_t)msd_get_stream(sh->ctx);
	const int W = sh->world, lg = log2_exact(W);
	if (W == 1 && !(sh->force_exchange && sh->comm)) {
		SH_MSD(sh, msd_sort_pairs_u64(sh->ctx, d_keys, d_rids, n));
		SH_HIP(sh, hipStreamSynchronize(sh->stream));
		*d_out_keys = d_keys;
		*d_out_rids = d_rids;
		*n_out = n;
		return MSD_OK;
	}
	if (!d_recv_keys || !d_recv_rids) return fail(sh, MSD_EINVAL, "sort_pairs_u64_sharded: null receive buffers");
	SH_MSD(sh, msd_partition_pairs_u64(sh->ctx, d_keys, d_rids, n, 56, 8, sh->d_row));
	std::vector<uint64_t> send_cnt, recv_cnt;
	int rc = finish_coarse_row(sh, recv_cap, n);
	if (!rc) rc = exchange_counts(sh, 256, 258, "tuples", send_cnt, recv_cnt);
	if (rc) return rc;
	uint64_t m = 0;
	for (int s = 0; s < W; ++s) m += recv_cnt[s];
	// keys and rids in ONE group: 2 x (world - 1) sends and receives in flight together
	SH_NCCL(sh, ncclGroupStart());
	rc = all_to_all<uint64_t>(sh, d_keys, d_recv_keys, send_cnt, recv_cnt, true);
	if (!rc) rc = all_to_all<uint64_t>(sh, d_rids, d_recv_rids, send_cnt, recv_cnt, true);
	SH_NCCL(sh, ncclGroupEnd());
	if (rc) return rc;
	SH_MSD(sh, msd_sort_pairs_u64_bits(sh->ctx, d_recv_keys, d_recv_rids, m, 64 - lg));
	SH_HIP(sh, hipStreamSynchronize(sh->stream));
	*d_out_keys = d_recv_keys;
	*d_out_rids = d_recv_rids;
	*n_out = m;
	return MSD_OK;
}

int msd_sort_u32_multi(int ndev, const int *devices, uint32_t **d_keys, const uint64_t *n, uint32_t **d_recv, uint64_t recv_cap,
		       uint32_t **d_work, uint64_t work_cap, int scheme, uint32_t **d_out, uint64_t *n_out)
{
	if (ndev < 1 || !devices || !d_keys || !n || !d_out || !n_out || log2_exact(ndev) < 0) return MSD_EINVAL;
	if (ndev > 1 && !d_recv) return MSD_EINVAL;
	std::vector<ncclComm_t> comms(ndev, nullptr);
	if (ndev > 1 && ncclCommInitAll(comms.data(), ndev, devices) != ncclSuccess) return MSD_ERCCL;
	std::vector<int> rcs(ndev, MSD_OK);
	std::vector<std::thread> th;
	for (int i = 0; i < ndev; ++i) {
		th.emplace_back([&, i]() {
			msd_ctx *ctx = nullptr;
			msd_shard *sh = nullptr;
			hipStream_t st = nullptr;
			// (a high-priority stream: it gets a hardware queue of its own -- on the default stream the compute kernels and
			// RCCL's kernels shared one queue on the GPU box and ran strictly one after the other)
			int lo_p = 0, hi_p = 0;
			int rc = hipSetDevice(devices[i]) == hipSuccess && hipDeviceGetStreamPriorityRange(&lo_p, &hi_p) == hipSuccess &&
					 hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi_p) == hipSuccess
				 ? MSD_OK
				 : MSD_EHIP;
			if (!rc) rc = msd_create(&ctx, devices[i], st);
			if (!rc) rc = msd_shard_create(&sh, ctx, comms[i]);
			if (!rc)
				rc = msd_sort_u32_sharded(sh, d_keys[i], n[i], d_recv ? d_recv[i] : nullptr, recv_cap, d_work ? d_work[i] : nullptr, work_cap,
							  scheme, &d_out[i], &n_out[i]);
			if (rc && sh) fprintf(stderr, "msd_sort_u32_multi: device %d: %s\n", devices[i], msd_shard_last_error(sh));
			if (sh) msd_shard_destroy(sh);
			if (ctx) msd_destroy(ctx);
			if (st) (void)hipStreamDestroy(st);
			rcs[i] = rc;
		});
	}
	for (auto &t : th) t.join();
	for (auto c : comms)
		if (c) (void)ncclCommDestroy(c);
	for (int rc : rcs)
		if (rc) return rc;
	return MSD_OK;
}

} // extern "C"
