// Ceiling for the block-permutation access pattern: copy N blocks of BLK bytes from a random
// permutation of source slots to a random permutation of destination slots (no atomics, no chains).
//   hipcc --offload-arch=gfx950 -O3 block_copy.hip -o block_copy && ./block_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int BLK>  // bytes per block; each lane moves 16 B, LPB = BLK/16 lanes per block
__global__ __launch_bounds__(256) void copy_blocks(const char *__restrict__ src, char *__restrict__ dst,
	const unsigned *__restrict__ sidx, const unsigned *__restrict__ didx, unsigned nblocks, int per_lane_blocks)
{
	constexpr int LPB = BLK / 16;
	const unsigned tid = blockIdx.x * 256 + threadIdx.x;
	const unsigned group = tid / LPB, sub = tid % LPB;
	const unsigned ngroups = gridDim.x * 256 / LPB;
	for (unsigned b0 = group; b0 < nblocks; b0 += ngroups * per_lane_blocks) {
		u32x4 v[16];
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			const unsigned b = b0 + k * ngroups;
			if (k < per_lane_blocks && b < nblocks) v[k] = *reinterpret_cast<const u32x4 *>(src + (size_t)sidx[b] * BLK + sub * 16);
		}
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			const unsigned b = b0 + k * ngroups;
			if (k < per_lane_blocks && b < nblocks) *reinterpret_cast<u32x4 *>(dst + (size_t)didx[b] * BLK + sub * 16) = v[k];
		}
	}
}

template <int BLK> void run(size_t bytes, bool random_order)
{
	const unsigned nb = (unsigned)(bytes / BLK);
	std::vector<unsigned> s(nb), d(nb);
	std::iota(s.begin(), s.end(), 0u);
	std::iota(d.begin(), d.end(), 0u);
	if (random_order) {
		std::mt19937 g(1);
		std::shuffle(s.begin(), s.end(), g);
		std::shuffle(d.begin(), d.end(), g);
	}
	char *src, *dst;
	unsigned *ds, *dd;
	CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes));
	CK(hipMalloc(&ds, nb * 4)); CK(hipMalloc(&dd, nb * 4));
	CK(hipMemset(src, 1, bytes));
	CK(hipMemcpy(ds, s.data(), nb * 4, hipMemcpyHostToDevice));
	CK(hipMemcpy(dd, d.data(), nb * 4, hipMemcpyHostToDevice));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int per : { 4, 16 }) {
		float best = 1e9f;
		for (int rep = 0; rep < 4; ++rep) {
			CK(hipEventRecord(e0));
			hipLaunchKernelGGL((copy_blocks<BLK>), dim3(256 * 16), dim3(256), 0, 0, src, dst, ds, dd, nb, per);
			CK(hipEventRecord(e1));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			best = std::min(best, ms);
		}
		printf("block %4d B  %s  %2d blocks in flight per lane group: %7.3f ms  %6.2f TB/s (read+write)\n", BLK,
		       random_order ? "random    " : "sequential", per, best, 2.0 * bytes / best / 1e9);
	}
	CK(hipFree(src)); CK(hipFree(dst)); CK(hipFree(ds)); CK(hipFree(dd));
}

int main()
{
	const size_t bytes = (size_t)4 << 30;
	run<256>(bytes, false);
	run<256>(bytes, true);
	run<512>(bytes, true);
	run<1024>(bytes, true);
	run<128>(bytes, true);
	return 0;
}
