// msd_scatter16.hpp -- the second half of "order a shard by its upper halves and keep only the low halves"
// (msd_order_low16_u32, included by msd_device.hpp): what a rank of the multi-GPU sort sends when the keys' low halves
// travel.
//
// Round 3's first version ordered the shard IN PLACE by its top 16 bits (two direct-placement rounds, 5.45 ms per 2^30
// keys) and then copied the low halves out (1.27 ms).  But the second round's output is read exactly once more -- by that
// copy -- and the copy's target is a second buffer anyway: so the second round can be an OUT-OF-PLACE scatter of the low
// halves straight into that buffer.  No slot grid, no misplaced blocks, no metadata, no block permutation, and 2 instead
// of 4 bytes written per key:
//   1. one in-place round on the top 8 bits (msd_sort_u32_top with begin_bit 24: the direct-placement round 0);
//   2. the 257 boundaries of the top-byte parents (bucket_bounds_kernel: binary searches);
//   3. hist16_kernel: a workgroup counts the second byte over its sixteenth of a parent (a read-only pass);
//   4. scan16_kernel (one workgroup per parent): the sizes of all 2^16 buckets -- what the exchange needs -- and where
//      every workgroup's share of every bucket starts in the output;
//   5. scatter_low16_kernel: the same workgroup streams the same chunk, collects low halves per bucket in LDS (256 buffers
//      of 128 values, one fetch-add per key gives the place) and writes whole 128-byte blocks to ITS part of the bucket:
//      no device-scope atomics, a deterministic result.  The order of the keys inside a bucket is that of the chunks, and
//      arbitrary inside a chunk (the receiver's counting leaf does not care).
// The reference does the same thing when it partitions tuples to the nodes' buffers: counted, then scattered through
// per-partition software write-combining buffers (src/msb_64.c:611-667, `range_partition_to_blocks`).
#pragma once

namespace msd {

#ifndef MSD_S16_VEC // (overridable for experiments, tools/variant_run.py)
#define MSD_S16_VEC 2
#define MSD_S16_CHUNKS 16
#endif
constexpr int kS16Th = 512;         // scatter: threads per workgroup (two workgroups per CU)
constexpr uint32_t kS16Cap = 128;   // values a bucket's LDS buffer holds: a block of 64 + what one tile can add
constexpr uint32_t kS16Chunks = MSD_S16_CHUNKS; // workgroups per top-byte parent
constexpr int kS16Vec = MSD_S16_VEC;            // 16-byte vectors per thread and tile
constexpr int kS16Kpt = 4 * kS16Vec;            // keys per thread and tile
constexpr uint32_t kS16Tile = kS16Th * kS16Kpt;
constexpr size_t kS16Lds = (size_t)256 * kS16Cap * 2 + 256 * 8 + 256 * 8 + 256 * 4 + 256 * 4 + (kS16Th / 64) * 64; // rings | places | ends | fill | skip | a wave's job lanes
static_assert(2 * kS16Lds <= 160 * 1024, "two workgroups per CU");

// chunk j of parent p (the keys whose top byte is p lie in [pb[p], pb[p + 1])): whole 16-byte vectors of the parent's range
__device__ __forceinline__ void s16_chunk(const uint64_t *__restrict__ pb, uint32_t p, uint32_t j, uint64_t &a, uint64_t &b)
{
	auto rfl64 = [](uint64_t x) -> uint64_t {
		return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)x) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(x >> 32)) << 32);
	};
	const uint64_t ps = rfl64(pb[p]), pe = rfl64(pb[p + 1]);
	const uint64_t per = ((pe - ps + kS16Chunks - 1) / kS16Chunks + 3) & ~3ull;
	a = ps + (uint64_t)j * per < pe ? ps + (uint64_t)j * per : pe;
	b = a + per < pe ? a + per : pe;
}

// a thread's keys of the tile that starts at grid element t0: kS16Vec 16-byte vectors (the array's last vector may be
// short: element by element); ok = the key belongs to the chunk [a, b)
__device__ __forceinline__ void s16_load(const uint32_t *__restrict__ keys, uint64_t n, uint64_t t0, uint64_t a, uint64_t b, uint32_t tid,
	uint32_t (&k)[kS16Kpt], uint32_t &okmask)
{
	okmask = 0;
#pragma unroll
	for (int u = 0; u < kS16Vec; ++u) {
		const uint64_t e = t0 + ((uint64_t)u * kS16Th + tid) * 4;
		if (e + 4 <= n && e < b) {
			const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(keys + e));
			k[4 * u + 0] = q.x; k[4 * u + 1] = q.y; k[4 * u + 2] = q.z; k[4 * u + 3] = q.w;
		} else {
#pragma unroll
			for (int i = 0; i < 4; ++i) k[4 * u + i] = e + i < n && e + i < b ? keys[e + i] : 0u;
		}
#pragma unroll
		for (int i = 0; i < 4; ++i) okmask |= (e + i >= a && e + i < b) ? 1u << (4 * u + i) : 0u;
	}
}

// wg[(p * 256 + c) * 16 + j] = keys of chunk j of parent p whose second byte is c
__global__ __launch_bounds__(kS16Th) void hist16_kernel(const uint32_t *__restrict__ keys, uint64_t n, const uint64_t *__restrict__ pb,
	uint32_t *__restrict__ wg)
{
	__shared__ uint32_t h[256];
	const uint32_t tid = threadIdx.x, p = blockIdx.x / kS16Chunks, j = blockIdx.x % kS16Chunks;
	uint64_t a, b;
	s16_chunk(pb, p, j, a, b);
	if (tid < 256) h[tid] = 0;
	__syncthreads();
	const uint64_t va = a & ~3ull;
	for (uint64_t t0 = va; t0 < b; t0 += 2 * kS16Tile) { // (sixteen keys in flight per thread)
		uint32_t k0[kS16Kpt], k1[kS16Kpt], m0, m1;
		s16_load(keys, n, t0, a, b, tid, k0, m0);
		s16_load(keys, n, t0 + kS16Tile, a, b, tid, k1, m1);
#pragma unroll
		for (int u = 0; u < kS16Kpt; ++u)
			if ((m0 >> u) & 1u) atomicAdd(&h[(k0[u] >> 16) & 255u], 1u);
#pragma unroll
		for (int u = 0; u < kS16Kpt; ++u)
			if ((m1 >> u) & 1u) atomicAdd(&h[(k1[u] >> 16) & 255u], 1u);
	}
	__syncthreads();
	if (tid < 256) wg[((size_t)p * 256u + tid) * kS16Chunks + j] = h[tid];
}

// per parent (one workgroup, thread c = second byte): counts[p * 256 + c] = the bucket's keys; base[(p * 256 + c) * 16 + j] =
// where chunk j's share of the bucket starts in the output
__global__ __launch_bounds__(256) void scan16_kernel(const uint32_t *__restrict__ wg, const uint64_t *__restrict__ pb,
	unsigned long long *__restrict__ counts, unsigned long long *__restrict__ base)
{
	__shared__ unsigned long long wsum[4];
	const uint32_t p = blockIdx.x, c = threadIdx.x, lane = c & 63u, w = c >> 6;
	uint32_t part[kS16Chunks];
	unsigned long long mine = 0;
#pragma unroll
	for (uint32_t j = 0; j < kS16Chunks; ++j) {
		part[j] = wg[((size_t)p * 256u + c) * kS16Chunks + j];
		mine += part[j];
	}
	counts[p * 256u + c] = mine;
	const unsigned long long inc = wave_incl_scan64(mine);
	if (lane == 63) wsum[w] = inc;
	__syncthreads();
	unsigned long long at = pb[p] + inc - mine;
	for (uint32_t ww = 0; ww < w; ++ww) at += wsum[ww];
#pragma unroll
	for (uint32_t j = 0; j < kS16Chunks; ++j) {
		base[((size_t)p * 256u + c) * kS16Chunks + j] = at;
		at += part[j];
	}
	if (p == 255u && c == 255u) base[(size_t)65536 * kS16Chunks] = at; // (one behind the last: a share ends where the next begins)
}

// Write alignment.  A workgroup's share of a bucket starts wherever the shares before it end -- at any 2-byte address --, and
// 128-byte blocks written from there straddle three 64-byte granules instead of two (PMC: 2.99 GB written per 2^30 keys
// where 2.15 are stored; 2.18 GB since -- profiles/r03_pmc_order_low16.txt).  So a ring starts its life as if it already held k values, k = the share's distance from the
// 128-byte boundary below it: every half-ring then leaves for a 128-byte-aligned place, as whole dwords; only the first one
// of a share skips its k phantom values (2-byte stores, once per share), and the last, partial one is written in pieces.
__global__ __launch_bounds__(kS16Th, 2) void scatter_low16_kernel(const uint32_t *__restrict__ keys, uint64_t n, const uint64_t *__restrict__ pb,
	const unsigned long long *__restrict__ base, uint16_t *__restrict__ out)
{
	constexpr int TH = kS16Th;
	constexpr uint32_t CAP = kS16Cap;
	static_assert(CAP == 128 && kS16Th == 512, "a bucket's buffer is two halves of 64 values; a wave looks after 32 buckets");
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint16_t *buf = reinterpret_cast<uint16_t *>(smem);                                        // 256 rings of CAP values
	unsigned long long *dstp = reinterpret_cast<unsigned long long *>(smem + (size_t)256 * CAP * 2); // the place of the ring's first half (aligned; before the share's start at first)
	unsigned long long *tail = dstp + 256;                                                     // the share's end: keys that find the ring full go there, backwards
	uint32_t *cnt = reinterpret_cast<uint32_t *>(tail + 256);                                  // per bucket: values in the ring (phantoms included) | first half-ring << 16
	uint32_t *skip = cnt + 256;                                                                // phantom values at the start of the ring's first half
	uint8_t *joblane = reinterpret_cast<uint8_t *>(skip + 256) + 64 * (threadIdx.x >> 6);     // this wave's compacted job lanes
	const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
	const uint32_t p = blockIdx.x / kS16Chunks, j = blockIdx.x % kS16Chunks;
	uint64_t a, b;
	s16_chunk(pb, p, j, a, b);
	if (a >= b) return;
	if (tid < 256) {
		const size_t at = ((size_t)p * 256u + tid) * kS16Chunks + j;
		const unsigned long long b0 = base[at];
		const uint32_t ph = (uint32_t)((reinterpret_cast<uintptr_t>(out + b0) >> 1) & 63u);
		cnt[tid] = ph;
		skip[tid] = ph;
		dstp[tid] = b0 - ph; // (may lie before the array: nothing is written below b0)
		tail[tid] = base[at + 1];
	}
	__syncthreads();
	const uint64_t va = a & ~3ull; // the chunk on the array's 16-byte grid
	uint32_t k[kS16Kpt], okm;
	s16_load(keys, n, va, a, b, tid, k, okm);
	// a wave looks after buckets 32 w .. 32 w + 31: lane l after the first (l < 32) or second (l >= 32) unwritten half-ring
	// of bucket 32 w + (l & 31)
	const uint32_t myc = 32u * w + (lane & 31u), myhalf = lane >> 5;
	MSD_STAMP_DECL(10);
	MSD_STAMP_START();
	auto lane64 = [&](unsigned long long x, int l) -> unsigned long long {
		return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, l) |
		       ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), l) << 32);
	};
	for (uint64_t t0 = va; t0 < b; t0 += kS16Tile) {
		MSD_STAMP(9);
		MSD_STAMP_TICK(11);
#ifdef MSD_STAMPS
		if constexpr (kStampThis) { // (the wait for this tile's keys, apart from the work on them)
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			MSD_STAMP(0);
		}
#endif
		uint32_t got[kS16Kpt];
#pragma unroll
		for (int u = 0; u < kS16Kpt; ++u) got[u] = atomicAdd(&cnt[(k[u] >> 16) & 255u], (okm >> u) & 1u); // (a key of another chunk adds nothing)
#pragma unroll
		for (int u = 0; u < kS16Kpt; ++u) {
			if ((okm >> u) & 1u) {
				const uint32_t c = (k[u] >> 16) & 255u, in_ring = got[u] & 0xFFFFu, first = got[u] >> 16;
				if (in_ring < CAP)
					buf[c * CAP + ((64u * first + in_ring) & (CAP - 1u))] = (uint16_t)k[u];
				else { // (its ring is full -- a bucket that takes more than 64 of a tile's keys, tile after tile: one by one, from
					// the end of the share backwards)
					const unsigned long long g = atomicAdd(&tail[c], ~0ull) - 1ull;
					out[g] = (uint16_t)k[u];
				}
			}
		}
		// the next tile's keys are on their way while this one's blocks are written
		MSD_STAMP(1); // fetch-adds + ring stores
		uint32_t kn[kS16Kpt], okn;
		s16_load(keys, n, t0 + kS16Tile, a, b, tid, kn, okn);
		MSD_STAMP(2); // next tile's loads issued
		__syncthreads();
		MSD_STAMP(3); // barrier
		{
			// (keys that found the ring full took numbers beyond it: given back)
			const uint32_t cc = cnt[myc], first = cc >> 16, have = min(cc & 0xFFFFu, CAP), full = have >> 6; // whole half-rings: 0, 1 or 2
			const unsigned long long g0 = dstp[myc];
			const uint32_t sk0 = skip[myc];
			const bool job = myhalf < full;
			// a share's first block has phantoms in front (once per share): one job per step, 2-byte stores
			unsigned long long slow = __ballot(job && myhalf == 0 && sk0 != 0);
			const unsigned long long fast = __ballot(job) & ~slow;
			while (slow) {
				const int l = __builtin_ctzll(slow);
				slow &= slow - 1;
				const uint32_t c = 32u * w + ((uint32_t)l & 31u);
				const uint32_t f0 = (uint32_t)__builtin_amdgcn_readlane((int)first, l), sk = (uint32_t)__builtin_amdgcn_readlane((int)sk0, l);
				const unsigned long long g = lane64(g0, l);
				const uint16_t *src = buf + c * CAP + 64u * (f0 & 1u);
				if (lane >= sk) out[g + lane] = src[lane];
			}
			// every other half-ring = 64 values = 128 bytes to a 128-byte-aligned place: eight jobs per step, eight lanes of 16
			// bytes per job (one job per step, a dependent LDS round trip each, took 3.6 of a tile's 9.4 thousand cycles)
			const uint32_t nfast = (uint32_t)__builtin_popcountll(fast);
			if (nfast) { // (uniform)
				if ((fast >> lane) & 1ull) joblane[__builtin_popcountll(fast & ((1ull << lane) - 1ull))] = (uint8_t)lane;
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				for (uint32_t j0 = 0; j0 < nfast; j0 += 8) {
					const uint32_t jq = j0 + (lane >> 3);
					const uint32_t sl = joblane[min(jq, nfast - 1u)];
					const uint32_t c = 32u * w + (sl & 31u), hf = sl >> 5;
					const uint32_t f0 = (uint32_t)__shfl((int)first, (int)sl);
					const unsigned long long g = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)g0, (int)sl) |
								      ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(g0 >> 32), (int)sl) << 32)) + 64u * hf;
					if (jq < nfast) {
						const u32x4 v = *reinterpret_cast<const u32x4 *>(buf + c * CAP + 64u * ((f0 + hf) & 1u) + 8u * (lane & 7u));
						*reinterpret_cast<u32x4 *>(out + g + 8u * (lane & 7u)) = v;
					}
				}
			}
			if (myhalf == 0) {
				cnt[myc] = (have - 64u * full) | (((first + full) & 1u) << 16);
				if (full) {
					dstp[myc] = g0 + 64u * full;
					skip[myc] = 0;
				}
			}
		}
		MSD_STAMP(4); // half-rings out
		__syncthreads();
		MSD_STAMP(5); // barrier
#pragma unroll
		for (int u = 0; u < kS16Kpt; ++u) k[u] = kn[u];
		okm = okn;
	}
	MSD_STAMP_FLUSH(TH / 64);
	// ---- what is left: less than a half-ring per bucket
	for (uint32_t c = 32u * w; c < 32u * w + 32u; ++c) {
		const uint32_t cc = cnt[c], rem = cc & 0xFFFFu, first = cc >> 16, sk = skip[c]; // (rem < 64)
		const unsigned long long g = dstp[c];
		if (lane >= sk && lane < rem) out[g + lane] = buf[c * CAP + 64u * first + lane];
	}
}

} // namespace msd
