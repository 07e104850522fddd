// What the classify kernels' LDS traffic costs: cycles per wave-instruction of the LDS pipe of one CU for
//   * ds_add_rtn_u32 on 256 counters with random digits (the per-key rank),
//   * ds_write_b32 to kbuf[d * 64 + r] (the per-key buffer write; its bank is r mod 32),
//   * ds_read_b128 of whole blocks (the flush),
// with every CU running `waves` waves (one or two workgroups), all of them issuing back to back.
//   hipcc --offload-arch=gfx950 -O3 lds_ops.hip -o lds_ops && ./lds_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned mix(unsigned x)
{
	x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
	return x;
}

// MODE 0: atomic add rtn, random counter of 256; 1: same, 4 counter copies by lane quarter (bank-disjoint);
// 2: write b32 random (d*64 + r); 3: read b128 sequential blocks; 4: atomic add without return;
// 5: atomic rtn on counters spread to stride 33 words (bank = f(d) unchanged, control);
// 6: write b32 with bank = lane (conflict-free, control); 7: atomic rtn conflict-free (counter = lane)
template <int MODE, int TH>
__global__ __launch_bounds__(TH) void lds_kernel(unsigned *out, int iters, unsigned long long *cycles)
{
	extern __shared__ unsigned lds[];
	const unsigned tid = threadIdx.x;
	for (unsigned i = tid; i < 16384 + 2048; i += TH) lds[i] = 0;
	__syncthreads();
	unsigned acc = 0, x = mix(tid * 2654435761u + blockIdx.x);
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int u = 0; u < 16; ++u) {
			x = x * 1664525u + 1013904223u;
			const unsigned d = (x >> 24) & 255u, r = (x >> 8) & 63u;
			if constexpr (MODE == 0) acc += atomicAdd(&lds[d], 1u);
			else if constexpr (MODE == 1) acc += atomicAdd(&lds[((tid >> 4) & 3u) * 264u + d], 1u);
			else if constexpr (MODE == 2) lds[2048 + d * 64 + r] = x;
			else if constexpr (MODE == 3) {
				const uint4 q = *reinterpret_cast<const uint4 *>(&lds[2048 + (((tid >> 4) * 17u + u * 29u + it) & 255u) * 64 + (tid & 15u) * 4]);
				acc += q.x ^ q.y ^ q.z ^ q.w;
			} else if constexpr (MODE == 4) __hip_atomic_fetch_add(&lds[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			else if constexpr (MODE == 5) acc += atomicAdd(&lds[d * 33u % 8192u], 1u);
			else if constexpr (MODE == 6) lds[2048 + ((x >> 13) & 255u) * 64 + (tid & 31u) + (x & 32u)] = x;
			else if constexpr (MODE == 7) acc += atomicAdd(&lds[(tid & 63u) + ((x >> 20) & 3u) * 64u], 1u);
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	if (acc == 0x12345678u) out[0] = acc;
	if (MODE == 2 || MODE == 4 || MODE == 6) { __syncthreads(); if (lds[2048 + tid] == 0x1234567u && lds[tid] == 77u) out[1] = 1; }
	if (tid == 0) atomicAdd(cycles, t1 - t0);
}

template <int MODE, int TH> static void run(const char *label, int wg_per_cu)
{
	unsigned *out;
	unsigned long long *cyc, h = 0;
	CK(hipMalloc(&out, 64));
	CK(hipMalloc(&cyc, 8));
	CK(hipMemset(cyc, 0, 8));
	const int iters = 2000, grid = 256 * wg_per_cu;
	CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&lds_kernel<MODE, TH>), hipFuncAttributeMaxDynamicSharedMemorySize, 73728));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	hipLaunchKernelGGL((lds_kernel<MODE, TH>), dim3(grid), dim3(TH), 73728, 0, out, 10, cyc);
	CK(hipMemset(cyc, 0, 8));
	CK(hipEventRecord(e0));
	hipLaunchKernelGGL((lds_kernel<MODE, TH>), dim3(grid), dim3(TH), 73728, 0, out, iters, cyc);
	CK(hipEventRecord(e1));
	CK(hipEventSynchronize(e1));
	float ms;
	CK(hipEventElapsedTime(&ms, e0, e1));
	CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
	const double per_wg_cycles = (double)h / grid;
	const double wave_ops_per_cu = (double)iters * 16 * (TH / 64) * wg_per_cu;
	printf("{\"op\": \"%s\", \"threads\": %d, \"wg_per_cu\": %d, \"waves_per_cu\": %d, \"ms\": %.3f, \"cycles_per_wave_op_per_cu\": %.2f, "
	       "\"lane_ops_per_cycle_per_cu\": %.2f}\n",
	       label, TH, wg_per_cu, TH / 64 * wg_per_cu, ms, per_wg_cycles / wave_ops_per_cu, wave_ops_per_cu * 64 / per_wg_cycles);
	fflush(stdout);
	CK(hipFree(out));
	CK(hipFree(cyc));
}

int main()
{
	run<0, 512>("atomic_add_rtn random of 256 counters", 2);
	run<0, 1024>("atomic_add_rtn random of 256 counters", 2);
	run<0, 256>("atomic_add_rtn random of 256 counters", 2);
	run<1, 512>("atomic_add_rtn, 4 bank-disjoint counter copies by lane quarter", 2);
	run<4, 512>("atomic_add (no return) random of 256 counters", 2);
	run<5, 512>("atomic_add_rtn random, counters at stride 33", 2);
	run<7, 512>("atomic_add_rtn conflict-free (counter = lane)", 2);
	run<2, 512>("write_b32 to kbuf[d*64 + r] random", 2);
	run<2, 1024>("write_b32 to kbuf[d*64 + r] random", 2);
	run<6, 512>("write_b32 conflict-free", 2);
	run<3, 512>("read_b128 whole blocks", 2);
	return 0;
}
