// Streaming ceiling of one MI355X: what a kernel that does nothing but move bytes reaches.
// The sort's big kernels are judged against these numbers, not against
// torch.Tensor.copy_ or the 8 TB/s spec figure alone.
//
//   hipcc --offload-arch=gfx950 -O3 stream_copy.hip -o stream_copy && ./stream_copy [GiB]
//
// Variants: copy / read-only / write-only; 256- and 1024-thread workgroups; U 16-byte loads in
// flight per lane (all U issued before the first store); plain and nontemporal accesses; one
// workgroup per chunk ("chunk": as many workgroups as chunks, the dispatcher balances) or a
// persistent grid-stride grid ("stride": CUs x k workgroups).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4 *p)
{
	if constexpr (NT) return __builtin_nontemporal_load(p);
	else return *p;
}
template <bool NT> __device__ __forceinline__ void st(u32x4 *p, u32x4 v)
{
	if constexpr (NT) __builtin_nontemporal_store(v, p);
	else *p = v;
}

// mode 0 copy, 1 read-only, 2 write-only.  nvec = number of 16-byte vectors; every workgroup moves
// tiles of TH*U vectors: tile t covers vectors [t*TH*U, (t+1)*TH*U), lane-contiguous per load.
template <int TH, int U, bool NTL, bool NTS, int MODE>
__global__ __launch_bounds__(TH) void stream_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t nvec,
	unsigned *__restrict__ sink)
{
	const size_t tiles = nvec / ((size_t)TH * U);
	u32x4 acc = { 0, 0, 0, 0 };
	for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
		const size_t base = t * (size_t)TH * U + threadIdx.x;
		u32x4 v[U];
		if constexpr (MODE != 2) {
#pragma unroll
			for (int u = 0; u < U; ++u) v[u] = ld<NTL>(src + base + (size_t)u * TH);
		} else {
#pragma unroll
			for (int u = 0; u < U; ++u) v[u] = u32x4{ (unsigned)t, (unsigned)u, threadIdx.x, 7u };
		}
		if constexpr (MODE != 1) {
#pragma unroll
			for (int u = 0; u < U; ++u) st<NTS>(dst + base + (size_t)u * TH, v[u]);
		} else {
#pragma unroll
			for (int u = 0; u < U; ++u) acc ^= v[u];
		}
	}
	if constexpr (MODE == 1)
		if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = 1; // keeps the loads alive
}

static int g_cus = 256;

template <int TH, int U, bool NTL, bool NTS, int MODE>
static void run(const char *label, const u32x4 *src, u32x4 *dst, size_t bytes, unsigned *sink, int per_cu /* 0 = one workgroup per tile */)
{
	const size_t nvec = bytes / 16;
	const size_t tiles = nvec / ((size_t)TH * U);
	const unsigned grid = per_cu ? (unsigned)std::min<size_t>(tiles, (size_t)g_cus * per_cu) : (unsigned)tiles;
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	float best = 1e9f, sum = 0;
	const int reps = 6;
	for (int rep = 0; rep < reps + 1; ++rep) {
		CK(hipEventRecord(e0));
		hipLaunchKernelGGL((stream_kernel<TH, U, NTL, NTS, MODE>), dim3(grid), dim3(TH), 0, 0, src, dst, nvec, sink);
		CK(hipEventRecord(e1));
		CK(hipEventSynchronize(e1));
		float ms;
		CK(hipEventElapsedTime(&ms, e0, e1));
		if (rep) { // first launch warms up
			best = std::min(best, ms);
			sum += ms;
		}
	}
	const double moved = (MODE == 0 ? 2.0 : 1.0) * (double)bytes;
	printf("{\"kernel\": \"%s\", \"threads\": %d, \"loads_in_flight\": %d, \"nt_load\": %d, \"nt_store\": %d, \"grid\": \"%s\", "
	       "\"workgroups\": %u, \"GiB\": %.1f, \"best_ms\": %.3f, \"avg_ms\": %.3f, \"best_TBps\": %.3f, \"avg_TBps\": %.3f}\n",
	       label, TH, U, (int)NTL, (int)NTS, per_cu ? "stride" : "chunk", grid, bytes / 1073741824.0, best, sum / reps,
	       moved / best / 1e9, moved / (sum / reps) / 1e9);
	fflush(stdout);
	CK(hipEventDestroy(e0));
	CK(hipEventDestroy(e1));
}


// Block permutation ceiling: block i of BLK bytes goes from slot (i * MS) mod nblocks to slot (i * MD) mod nblocks
// (nblocks a power of two, MS / MD odd: two bijections, no index arrays); BLK / 16 lanes move one block,
// U blocks in flight per lane.  MS = MD = 1 is the sequential copy in the same geometry.
template <int TH, int U, int BLK>
__global__ __launch_bounds__(TH) void permute_kernel(const char *__restrict__ src, char *__restrict__ dst, unsigned nblocks,
	unsigned ms, unsigned md)
{
	constexpr unsigned LPB = BLK / 16, GPW = TH / LPB; // lane groups per workgroup
	const unsigned grp = threadIdx.x / LPB, sub = threadIdx.x % LPB;
	const unsigned mask = nblocks - 1;
	for (unsigned b0 = blockIdx.x * GPW * U + grp; b0 < nblocks; b0 += gridDim.x * GPW * U) {
		u32x4 v[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const unsigned i = b0 + u * GPW;
			v[u] = *reinterpret_cast<const u32x4 *>(src + (size_t)((i * ms) & mask) * BLK + sub * 16);
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const unsigned i = b0 + u * GPW;
			*reinterpret_cast<u32x4 *>(dst + (size_t)((i * md) & mask) * BLK + sub * 16) = v[u];
		}
	}
}

template <int TH, int U, int BLK>
static void run_permute(const char *src, char *dst, size_t bytes, unsigned ms, unsigned md, int per_cu)
{
	const unsigned nblocks = (unsigned)(bytes / BLK);
	constexpr unsigned GPW = TH / (BLK / 16);
	const unsigned tiles = nblocks / (GPW * U);
	const unsigned grid = per_cu ? std::min<unsigned>(tiles, (unsigned)g_cus * per_cu) : tiles;
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	float best = 1e9f, sum = 0;
	const int reps = 5;
	for (int rep = 0; rep < reps + 1; ++rep) {
		CK(hipEventRecord(e0));
		hipLaunchKernelGGL((permute_kernel<TH, U, BLK>), dim3(grid), dim3(TH), 0, 0, src, dst, nblocks, ms, md);
		CK(hipEventRecord(e1));
		CK(hipEventSynchronize(e1));
		float t;
		CK(hipEventElapsedTime(&t, e0, e1));
		if (rep) {
			best = std::min(best, t);
			sum += t;
		}
	}
	printf("{\"kernel\": \"permute\", \"block_bytes\": %d, \"threads\": %d, \"blocks_in_flight_per_lane\": %d, \"src\": \"%s\", \"dst\": \"%s\", "
	       "\"grid\": \"%s\", \"workgroups\": %u, \"best_ms\": %.3f, \"best_TBps\": %.3f, \"avg_TBps\": %.3f}\n",
	       BLK, TH, U, ms == 1 ? "sequential" : "random", md == 1 ? "sequential" : "random", per_cu ? "stride" : "chunk", grid, best,
	       2.0 * bytes / best / 1e9, 2.0 * bytes / (sum / reps) / 1e9);
	fflush(stdout);
}
template <int BLK> static void permute_suite(const char *src, char *dst, size_t bytes)
{
	const unsigned R1 = 2654435761u, R2 = 0x9E3779B1u;
	run_permute<1024, 1, BLK>(src, dst, bytes, 1, 1, 0);
	run_permute<1024, 1, BLK>(src, dst, bytes, R1, R2, 0);
	run_permute<1024, 2, BLK>(src, dst, bytes, R1, R2, 0);
	run_permute<1024, 4, BLK>(src, dst, bytes, R1, R2, 2);
	run_permute<256, 4, BLK>(src, dst, bytes, R1, R2, 8);
	run_permute<256, 8, BLK>(src, dst, bytes, R1, R2, 8);
	run_permute<1024, 2, BLK>(src, dst, bytes, R1, 1, 0);
	run_permute<1024, 2, BLK>(src, dst, bytes, 1, R2, 0);
}

// The direct classify kernel's access pattern without any of its work: 1024 workgroups (two per CU, LDS-limited like the
// kernel), each owns a 64-slot piece of each of 256 regions; per tile it reads 128 slots (pieces walked from a per-piece
// starting offset, SPV consecutive slots per visit of a piece) and writes 128 blocks into slots of the same pieces it has
// read earlier (in place: write behind read; OOP: into a second buffer at the same positions).  What this reaches is the
// memory-side ceiling of that kernel.
template <int TH, int NV, int SPV, bool OOP>
__global__ __launch_bounds__(TH) void pieces_kernel(char *__restrict__ data, char *__restrict__ other, unsigned nslots, int lag)
{
	extern __shared__ unsigned pad_lds[]; // occupancy control only
	constexpr unsigned LPB = 16, GPW = TH / LPB, GPT = GPW * NV;
	const unsigned W = gridDim.x, me = blockIdx.x;
	const unsigned region = nslots / 256, piece = region / W; // slots
	const unsigned grp = threadIdx.x / LPB, sub = threadIdx.x % LPB;
	const unsigned tiles = 256 * piece / GPT;
	char *wbase = OOP ? other : data;
	if (threadIdx.x == 0) pad_lds[0] = 0;
	// running slot number j of this workgroup -> slot: visits of SPV consecutive slots, pieces round-robin
	auto slot_of = [&](unsigned j) -> unsigned {
		const unsigned visit = j / SPV, k = j % SPV;
		const unsigned p = visit & 255u, idx = (visit >> 8) * SPV + k;
		const unsigned rot = (((p * 2654435761u) ^ (me * 40503u + (me >> 3))) % piece) / SPV * SPV;
		return p * region + me * piece + (rot + idx) % piece;
	};
	for (unsigned t = 0; t < tiles + lag; ++t) {
		u32x4 v[NV];
		unsigned long long wa[NV];
#pragma unroll
		for (int k = 0; k < NV; ++k) {
			const unsigned j = t * GPT + k * GPW + grp;
			if (t < tiles) v[k] = *reinterpret_cast<const u32x4 *>(data + (size_t)slot_of(j) * 256 + sub * 16);
			wa[k] = (size_t)slot_of(j - lag * GPT) * 256 + sub * 16; // where this lane group read `lag` tiles ago
		}
		if (t >= (unsigned)lag) {
#pragma unroll
			for (int k = 0; k < NV; ++k) *reinterpret_cast<u32x4 *>(wbase + wa[k]) = (t < tiles) ? v[k] : u32x4{ 1, 2, 3, 4 };
		}
	}
}

template <int TH, int NV, int SPV, bool OOP> static void run_pieces(char *data, char *other, size_t bytes, int lag, int lds_kb)
{
	const unsigned nslots = (unsigned)(bytes / 256);
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&pieces_kernel<TH, NV, SPV, OOP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
	float best = 1e9f, sum = 0;
	const int reps = 5;
	for (int rep = 0; rep < reps + 1; ++rep) {
		CK(hipEventRecord(e0));
		hipLaunchKernelGGL((pieces_kernel<TH, NV, SPV, OOP>), dim3(1024), dim3(TH), lds_kb * 1024, 0, data, other, nslots, lag);
		CK(hipEventRecord(e1));
		CK(hipEventSynchronize(e1));
		float t;
		CK(hipEventElapsedTime(&t, e0, e1));
		if (rep) {
			best = std::min(best, t);
			sum += t;
		}
	}
	printf("{\"kernel\": \"pieces_%s\", \"threads\": %d, \"slots_per_thread_per_tile\": %d, \"consecutive_slots_per_piece_visit\": %d, "
	       "\"write_lag_tiles\": %d, \"lds_KiB\": %d, \"best_ms\": %.3f, \"best_TBps\": %.3f, \"avg_TBps\": %.3f}\n",
	       OOP ? "out_of_place" : "in_place", TH, NV, SPV, lag, lds_kb, best, 2.0 * bytes / best / 1e9, 2.0 * bytes / (sum / reps) / 1e9);
	fflush(stdout);
}

int main(int argc, char **argv)
{
	const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 4;
	const size_t bytes = gib << 30;
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	g_cus = prop.multiProcessorCount;
	printf("{\"device\": \"%s\", \"cus\": %d, \"buffer_GiB\": %zu}\n", prop.name, g_cus, gib);
	u32x4 *src, *dst;
	unsigned *sink;
	CK(hipMalloc(&src, bytes));
	CK(hipMalloc(&dst, bytes));
	CK(hipMalloc(&sink, 256));
	CK(hipMemset(src, 0x5a, bytes));
	CK(hipMemset(dst, 0, bytes));
	CK(hipDeviceSynchronize());

	// ---- copy
	run<256, 4, false, false, 0>("copy", src, dst, bytes, sink, 0);
	run<256, 8, false, false, 0>("copy", src, dst, bytes, sink, 0);
	run<1024, 4, false, false, 0>("copy", src, dst, bytes, sink, 0);
	run<1024, 1, false, false, 0>("copy", src, dst, bytes, sink, 0);
	run<1024, 2, false, false, 0>("copy", src, dst, bytes, sink, 0);
	run<256, 4, false, false, 0>("copy", src, dst, bytes, sink, 8);
	run<256, 8, false, false, 0>("copy", src, dst, bytes, sink, 8);
	run<256, 8, false, false, 0>("copy", src, dst, bytes, sink, 4);
	run<1024, 4, false, false, 0>("copy", src, dst, bytes, sink, 2);
	run<1024, 8, false, false, 0>("copy", src, dst, bytes, sink, 2);
	run<512, 8, false, false, 0>("copy", src, dst, bytes, sink, 4);
	run<256, 4, true, true, 0>("copy", src, dst, bytes, sink, 0);
	run<256, 8, true, true, 0>("copy", src, dst, bytes, sink, 8);
	run<256, 8, false, true, 0>("copy", src, dst, bytes, sink, 8);
	run<256, 8, true, false, 0>("copy", src, dst, bytes, sink, 8);
	run<1024, 4, true, true, 0>("copy", src, dst, bytes, sink, 2);
	// ---- read only
	run<256, 8, false, false, 1>("read", src, dst, bytes, sink, 8);
	run<256, 8, true, false, 1>("read", src, dst, bytes, sink, 8);
	run<1024, 4, false, false, 1>("read", src, dst, bytes, sink, 2);
	run<256, 4, false, false, 1>("read", src, dst, bytes, sink, 0);
	// ---- write only
	run<256, 8, false, false, 2>("write", src, dst, bytes, sink, 8);
	run<256, 8, false, true, 2>("write", src, dst, bytes, sink, 8);
	run<1024, 4, false, false, 2>("write", src, dst, bytes, sink, 2);
	run<256, 4, false, false, 2>("write", src, dst, bytes, sink, 0);
	// ---- the direct classify kernel's in-place pattern
	run_pieces<512, 4, 1, false>((char *)src, (char *)dst, bytes, 2, 78);
	run_pieces<1024, 2, 1, false>((char *)src, (char *)dst, bytes, 2, 78);
	run_pieces<512, 4, 1, true>((char *)src, (char *)dst, bytes, 2, 78);
	run_pieces<512, 4, 2, false>((char *)src, (char *)dst, bytes, 2, 78);
	run_pieces<512, 4, 4, false>((char *)src, (char *)dst, bytes, 2, 78);
	run_pieces<512, 4, 8, false>((char *)src, (char *)dst, bytes, 2, 78);
	run_pieces<512, 4, 4, true>((char *)src, (char *)dst, bytes, 2, 78);
	run_pieces<512, 4, 1, false>((char *)src, (char *)dst, bytes, 8, 78);
	run_pieces<256, 4, 1, false>((char *)src, (char *)dst, bytes, 2, 38);
	// ---- block permutation (the sort's 256-byte blocks and what larger ones would buy)
	permute_suite<128>((const char *)src, (char *)dst, bytes);
	permute_suite<256>((const char *)src, (char *)dst, bytes);
	permute_suite<512>((const char *)src, (char *)dst, bytes);
	permute_suite<1024>((const char *)src, (char *)dst, bytes);
	permute_suite<4096>((const char *)src, (char *)dst, bytes);
	CK(hipFree(src));
	CK(hipFree(dst));
	CK(hipFree(sink));
	return 0;
}
