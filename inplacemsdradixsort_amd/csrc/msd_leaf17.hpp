// msd_leaf17.hpp -- the leaf for (u64 key, u64 rid) segments of up to 17408 tuples: the whole segment is sorted in the
// registers and the LDS of ONE 1024-thread workgroup, read once and written once (included by msd_device.hpp).
//
// Round 2 finished such a segment (2^30 tuples: 65536 of them after two 8-bit rounds, 48 open bits) in two passes:
// regpart_kernel split it by 2-3 bits (read + write of every tuple), leaf_count_sort_kernel sorted the <= 3072-tuple
// children (read + write again): 139 B per tuple for the whole sort, four full passes.  This kernel
// is regpart_kernel's data movement -- 16-byte loads on the arrays' grid into registers, keys and then payloads through
// one LDS staging buffer to their places and out as whole vectors -- around leaf_count_sort_kernel's ranking: ONE
// unstable counting pass over the top bits that vary in the segment (16-bit LDS counters, the fetch-add's return value is
// the rank among equal values, an in-place scan gives positions), the keys go to those positions in LDS, and every element
// then finds its own place inside its group of equal counted bits from its two neighbours on either side (one LDS round
// trip; longer groups continue element by element; a group longer than 48 rejects the segment,
// untouched, to regpart + the small leaves).  Payloads never pass through LDS before their final permutation: they wait
// in registers.  This is the reference's in-cache finish of a bucket (local_radixsort down to insertion sort,
// src/msb_64.c:1007-1035, :126-149) with the register file and LDS as the cache.
//
// How many bits are counted decides the fix-up.  With 13 bits (the first version: the counters beside the staging buffer)
// a 2^14-element segment has two elements per counter value: an element in three sits in a group of four or more, some lane
// of nearly every wave walked its group element by element, and the fix-up took 48 of the segment's 95 thousand cycles
// (profiles/r03_stamps_leaf17.json).  Now the counters use the staging buffer itself -- the keys are in registers until the
// positions are known -- so there is room for 2^16 of them: ceil(log2 n) + 2 bits are counted, nine elements in ten are
// alone with their counted bits and a window of two neighbours covers all but one group in a thousand.
#pragma once

namespace msd {

constexpr int kL17Th = 1024;
constexpr int kL17Vec = 8;                                           // 16-byte vectors (2 elements) per thread and array
constexpr uint32_t kL17Cap = kL17Th * (kL17Vec * 2 + 1);              // 17408 elements on the 16-byte grid
#ifndef MSD_L17_BITS // (overridable for experiments, tools/variant_run.py)
#define MSD_L17_BITS 16
#define MSD_L17_WIN 2
#endif
#ifndef MSD_L17_PREFETCH // (0: every segment's keys are requested when its turn comes; A/B comparisons)
#define MSD_L17_PREFETCH 1
#endif
constexpr int kL17BitsMin = 13, kL17BitsMax = MSD_L17_BITS;          // counted bits: ceil(log2 n) + 2 within these bounds
constexpr size_t kL17Side = kL17Cap;                                 // a byte per position (the distances of the fix-up)
// The 16-bit counters (two per word) lie where the staging buffer and the distances will be -- the keys wait in registers
// while they are counted.  A thread scans a contiguous run of 4..32 words: 4 padding words behind every 32 keep the
// 16-byte accesses of neighbouring lanes on different banks.
__device__ __forceinline__ uint32_t l17_at(uint32_t w) { return w + ((w >> 5) << 2); }
constexpr size_t kL17CwBytes = ((((size_t)1 << kL17BitsMax) / 2) / 32 * 36) * 4;
constexpr uint32_t kL17ListCap = 3072;                               // groups of two or more the list holds (2^14 uniform keys: about 1750)
constexpr size_t kL17Lds = (size_t)kL17Cap * 8 + kL17Side + 64 * 8 + 256 + kL17ListCap * 2; // staging | distances | junk | misc | group list
static_assert(kL17CwBytes <= (size_t)kL17Cap * 8 + kL17Side, "the counters fit the staging buffer + the distances");
static_assert(kL17Lds <= 160 * 1024, "one workgroup per CU");
static_assert(kL17BitsMax <= 16 && kL17BitsMin >= 13 && (((size_t)1 << kL17BitsMin) / 2) / kL17Th >= 4, "a thread scans whole 16-byte vectors");
constexpr uint32_t kL17MaxGroup = 48;

template <typename V>
__global__ __launch_bounds__(kL17Th, 4) void leaf17_kernel(uint64_t *__restrict__ keys, uint64_t *__restrict__ vals,
	const Segment *__restrict__ segs, uint32_t nsegs, Segment *__restrict__ rejected, uint32_t *__restrict__ nrejected,
	Counters *__restrict__ ctr, uint32_t min_count)
{
	constexpr bool HV = has_val<V>::value;
	constexpr int TH = kL17Th, NV = kL17Vec, NK = NV * 2 + 1;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint64_t *stage = reinterpret_cast<uint64_t *>(smem);                        // kL17Cap elements
	uint32_t *cw = reinterpret_cast<uint32_t *>(smem);                           // before that: 2 x 16-bit counters per word (l17_at)
	int8_t *dl = reinterpret_cast<int8_t *>(smem + (size_t)kL17Cap * 8);         // distance to the final place, per position
	uint64_t *junk = reinterpret_cast<uint64_t *>(smem + (size_t)kL17Cap * 8 + kL17Side); // per-lane junk word
	uint32_t *wtot = reinterpret_cast<uint32_t *>(junk + 64);                    // 16 wave totals, [16] [18] flags, [17] listed groups
	uint64_t *s_or = reinterpret_cast<uint64_t *>(wtot + 32);                    // OR / AND of the keys
	uint16_t *list = reinterpret_cast<uint16_t *>(smem + (size_t)kL17Cap * 8 + kL17Side + 64 * 8 + 256); // first positions of the groups
	const uint32_t tid0 = threadIdx.x;
	auto rfl = [](uint32_t x) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane(x); };

	// The keys of a workgroup's NEXT segment are requested as soon as the current ones have left their registers for the
	// staging buffer.  Measured (2^30 u64 keys / tuples, -DMSD_L17_PREFETCH=0 against 1 on one box): no difference -- the
	// 7 thousand cycles the keys' arrival loses reappear where the loads are issued.  With every CU at it a workgroup's share
	// of the memory bandwidth is about 9 bytes per cycle: 139 KiB take 15 thousand cycles to arrive and as long to leave
	// whenever they are requested, which is all of the tuple leaf's 60 thousand cycles per segment (5.3 TB/s) and 31 of the
	// key leaf's 45.  Hiding the rest needs the loads in flight during the counting phases, i.e. a second set of key
	// registers: tried (= 2), the compiler parks exactly those 34 registers in scratch memory.
	// (segment sj's keys: vector v of thread t = grid elements (v * TH + t) * 2, + 1; tail element NV * TH * 2 + t; lanes beyond
	// the segment read its last vector again; behind the list's end the last segment is read once more, unused)
	uint64_t k[kL17Vec * 2 + 1];
	auto load_keys = [&](uint64_t(&k)[kL17Vec * 2 + 1], uint32_t sj) {
		uint32_t tl = tid0;
		asm volatile("" : "+v"(tl));
		const Segment g2 = segs[min(sj, nsegs - 1u)];
		const uint64_t start2 = (uint64_t)rfl((uint32_t)g2.start) | ((uint64_t)rfl((uint32_t)(g2.start >> 32)) << 32);
		const uint32_t off2 = (uint32_t)(start2 & 1u);
		const uint32_t tot2 = max(rfl((uint32_t)g2.count) + off2, 1u);
		const uint64_t *kb2 = keys + (start2 - off2);
		const uint32_t lastv2 = (tot2 - 1u) >> 1;
#pragma unroll
		for (int v = 0; v < kL17Vec; ++v) {
			const uint32_t q = min((uint32_t)(v * kL17Th) + tl, lastv2) * 2u;
			const u32x4 a = *reinterpret_cast<const u32x4 *>(kb2 + q);
			k[2 * v] = (uint64_t)a.x | ((uint64_t)a.y << 32);
			k[2 * v + 1] = (uint64_t)a.z | ((uint64_t)a.w << 32);
		}
		k[kL17Vec * 2] = kb2[min((uint32_t)(kL17Vec * kL17Th * 2) + tl, tot2 - 1u)];
	};
	constexpr bool kAhead = MSD_L17_PREFETCH != 0;
	if (kAhead && blockIdx.x < nsegs) load_keys(k, blockIdx.x);
	MSD_STAMP_DECL(9);
	MSD_STAMP_START();
	for (uint32_t si = blockIdx.x; si < nsegs; si += gridDim.x) {
		MSD_STAMP(9);
		MSD_STAMP_TICK(11);
		// (the thread index is made opaque per segment: addresses and predicates derived from it would otherwise be hoisted
		// out of this loop, kept in registers for its whole life and spilled)
		uint32_t tid = tid0;
		asm volatile("" : "+v"(tid));
		const uint32_t lane = tid & 63u, w = tid >> 6;
		const Segment g = segs[si];
		const uint64_t start = (uint64_t)rfl((uint32_t)g.start) | ((uint64_t)rfl((uint32_t)(g.start >> 32)) << 32);
		const uint64_t cnt64 = (uint64_t)rfl((uint32_t)g.count) | ((uint64_t)rfl((uint32_t)(g.count >> 32)) << 32);
		const uint32_t bits = rfl(g.bits);
		const uint32_t off = (uint32_t)(start & 1u);
		if (cnt64 + off > kL17Cap || cnt64 < min_count || cnt64 < 2 || bits == 0 || bits > 64) {
			// too long for the staging buffer (with its first element on an odd index), or shorter than this kernel is worth:
			// left, untouched, to whoever takes the rejected segments (fewer than two elements or no open bit: nothing to do)
			if (tid == 0 && cnt64 >= 2 && bits != 0 && bits <= 64) rejected[atomicAdd(nrejected, 1u)] = g;
			if constexpr (MSD_L17_PREFETCH == 1) load_keys(k, si + gridDim.x);
			continue;
		}
		const uint32_t n = (uint32_t)cnt64, tot = n + off; // the segment on the 16-byte grid: elements [off, tot)
		uint64_t *kb = keys + (start - off), *vb = HV ? vals + (start - off) : nullptr;
		const uint32_t lastv = (tot - 1u) >> 1;
		if constexpr (!kAhead) load_keys(k, si);
		auto elem = [&](int u) -> uint32_t { return u < NV * 2 ? (uint32_t)((u / 2) * TH * 2) + tid * 2 + (u % 2) : (uint32_t)(NV * TH * 2) + tid; };
		(void)elem;
		// ---- which bits vary (OR / AND over the segment), counters cleared
		// (counted bits: about four counters per element -- nine elements in ten are then alone with their counted bits;
		// from n alone, so that the counters can be cleared while the keys are still on their way)
		const uint32_t lb = min((uint32_t)kL17BitsMax, max((uint32_t)kL17BitsMin, 34u - (uint32_t)__builtin_clz(n - 1u)));
		{
			const uint32_t pw = l17_at(((uint32_t)1 << lb) / 2); // padded words, a multiple of 4
			const u32x4 zero = { 0u, 0u, 0u, 0u };
			for (uint32_t j = tid * 4u; j < pw; j += TH * 4u) *reinterpret_cast<u32x4 *>(cw + j) = zero;
		}
		if (tid == 0) {
			s_or[0] = 0;
			s_or[1] = ~0ull;
			wtot[16] = wtot[17] = wtot[18] = 0;
		}
		uint64_t k_or = 0, k_and = ~0ull;
#pragma unroll
		for (int u = 0; u < NK; ++u) {
			const uint32_t el = elem(u);
			if (el >= off && el < tot) {
				k_or |= k[u];
				k_and &= k[u];
			}
		}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			k_or |= __shfl_xor(k_or, o);
			k_and &= __shfl_xor(k_and, o);
		}
		__syncthreads();
		if (lane == 0) {
			atomicOr(reinterpret_cast<unsigned long long *>(&s_or[0]), (unsigned long long)k_or);
			atomicAnd(reinterpret_cast<unsigned long long *>(&s_or[1]), (unsigned long long)k_and);
		}
		__syncthreads();
		MSD_STAMP(0); // keys arrive, OR / AND, clear, two barriers
		const uint64_t openmask = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
		const uint64_t vopen = (s_or[0] ^ s_or[1]) & openmask;
		if (vopen == 0) { // (uniform) constant on the open bits: already sorted
			__syncthreads();
			if constexpr (MSD_L17_PREFETCH == 1) load_keys(k, si + gridDim.x);
			continue;
		}
		const uint32_t nbits = (uint32_t)(64 - __builtin_clzll((unsigned long long)vopen));
		const uint32_t shift = nbits > lb ? nbits - lb : 0;
		const uint32_t mask = (1u << (nbits - shift)) - 1u;
		// ---- rank among the elements with equal counted bits: one LDS fetch-add per element
		uint32_t pr[NK]; // rank, then position; bit 31: not an element of the segment
#pragma unroll
		for (int u = 0; u < NK; ++u) {
			const uint32_t el = elem(u);
			const bool in = el >= off && el < tot;
			const uint32_t v = (uint32_t)(k[u] >> shift) & mask, sh = 16u * (v & 1u);
			uint32_t *a = in ? cw + l17_at(v >> 1) : reinterpret_cast<uint32_t *>(junk + lane);
			pr[u] = ((atomicAdd(a, 1u << sh) >> sh) & 0xFFFFu) | (in ? 0u : 0x80000000u);
			if (u % 6 == 5) __builtin_amdgcn_sched_barrier(0); // (six fetch-adds in flight: all seventeen with their addresses spill)
		}
		MSD_STAMP(1); // fetch-adds
		__syncthreads();
		// ---- counts -> exclusive positions, in place; thread t owns wpt = 4..32 consecutive words (inside one padding chunk)
		{
			const uint32_t wpt = max((((uint32_t)1 << (nbits - shift)) / 2) / (uint32_t)TH, 4u);
			uint32_t *mine = cw + l17_at(tid * wpt);
			uint32_t acc = 0; // (both halves at once: no half ever exceeds the segment's 17408 elements)
#pragma unroll 1
			for (uint32_t j = 0; j < wpt; j += 4) {
				const u32x4 q = *reinterpret_cast<const u32x4 *>(mine + j);
				acc += q.x + q.y + q.z + q.w;
			}
			const uint32_t tsum = (acc & 0xFFFFu) + (acc >> 16);
			const uint32_t inc = wave_incl_scan(tsum);
			if (lane == 63) wtot[w] = inc;
			__syncthreads();
			uint32_t run = inc - tsum;
#pragma clang loop vectorize(disable) unroll(disable) // (vectorised, this 16-step loop costs eight registers -- spilled)
			for (uint32_t ww = 0; ww < w; ++ww) run += wtot[ww];
#pragma unroll 1
			for (uint32_t j = 0; j < wpt; j += 4) {
				u32x4 q = *reinterpret_cast<const u32x4 *>(mine + j);
				uint32_t t;
				t = q.x; q.x = run | ((run + (t & 0xFFFFu)) << 16); run += (t & 0xFFFFu) + (t >> 16);
				t = q.y; q.y = run | ((run + (t & 0xFFFFu)) << 16); run += (t & 0xFFFFu) + (t >> 16);
				t = q.z; q.z = run | ((run + (t & 0xFFFFu)) << 16); run += (t & 0xFFFFu) + (t >> 16);
				t = q.w; q.w = run | ((run + (t & 0xFFFFu)) << 16); run += (t & 0xFFFFu) + (t >> 16);
				*reinterpret_cast<u32x4 *>(mine + j) = q;
			}
		}
		__syncthreads();
		MSD_STAMP(2); // scan (three barriers)
		// ---- every element's position (groups of equal counted bits become contiguous); the element of rank 1 of every group
		// of two or more enters the group in a list
		// (the counter addresses are those of the fetch-adds: computed from an opaque copy of the shift, or the compiler keeps
		// all seventeen in registers across the scan -- and spills)
		const uint64_t lowmask = shift ? (1ull << shift) - 1ull : 0ull;
		const bool groups = shift && (vopen & lowmask) != 0; // more bits vary than were counted
		uint32_t shift2 = rfl(shift);
		asm volatile("" : "+s"(shift2));
		uint32_t m1 = 0;
#pragma unroll
		for (int u = 0; u < NK; ++u) {
			const uint32_t v = (uint32_t)(k[u] >> shift2) & mask;
			const uint32_t p = ((cw[l17_at(v >> 1)] >> (16u * (v & 1u))) & 0xFFFFu) + (pr[u] & 0xFFFFu);
			m1 |= pr[u] == 1u ? 1u << u : 0u; // (rank 1, an element of the segment)
			pr[u] = (pr[u] & 0x80000000u) | p;
			if (u % 6 == 5) __builtin_amdgcn_sched_barrier(0);
		}
		__builtin_amdgcn_sched_barrier(0);
		if (groups) { // (uniform)
			const uint32_t c1 = (uint32_t)__builtin_popcount(m1), inc = wave_incl_scan(c1);
			uint32_t wb = 0;
			if (lane == 63) wb = atomicAdd(&wtot[17], inc);
			uint32_t o = (uint32_t)__shfl((int)wb, 63) + inc - c1;
#pragma unroll
			for (int u = 0; u < NK; ++u) {
				if ((m1 >> u) & 1u) {
					if (o < kL17ListCap) list[o] = (uint16_t)((pr[u] & 0xFFFFu) - 1u);
					++o;
				}
			}
		}
		__syncthreads(); // ... and only then the keys go there, on the array's 16-byte grid: the staging buffer is where the counters were
#pragma unroll
		for (int u = 0; u < NK; ++u) {
			uint64_t *o = (pr[u] >> 31) ? junk + lane : stage + (pr[u] & 0x7FFFFFFFu) + off;
			*o = k[u];
			if (u % 6 == 5) __builtin_amdgcn_sched_barrier(0);
		}
		if constexpr (MSD_L17_PREFETCH == 1) load_keys(k, si + gridDim.x); // (the key registers are free from here on)
		if (HV && groups) { // the distances start at zero
			const u32x4 zero = { 0u, 0u, 0u, 0u };
			for (uint32_t j = tid * 16u; j < (uint32_t)kL17Side; j += TH * 16u) *reinterpret_cast<u32x4 *>(dl + j) = zero;
		}
		__syncthreads();
		MSD_STAMP(3); // keys into LDS
		if (groups) { // (uniform)
			// (same group <=> the keys agree above `shift` <=> their XOR is below 2^shift: one 64-bit compare against a uniform
			// bound instead of two 64-bit shifts, which run at a quarter of the rate)
			const uint64_t glim = 1ull << shift;
			const uint64_t *sg = stage + off; // the segment's elements by position
			// two elements 6 apart share a group only if the group has 7 members or more
			bool long7 = false;
			for (uint32_t i = tid; i + 6u < n; i += TH)
				if ((sg[i] ^ sg[i + 6u]) < glim) long7 = true;
			if (long7) wtot[16] = 1;
			__syncthreads();
			const uint32_t ngr = wtot[17];
			const bool slow = wtot[16] != 0 || ngr > kL17ListCap;
			MSD_STAMP(4); // long-group check
			if (!slow) {
				// ---- one lane per listed group (<= 6 members, of which the first two are known): all of it into registers, every
				// member's rank from the 15 comparisons, back in order; the distance each member moved is left for its owner
				const uint32_t lastp = n - 1u;
#pragma unroll 1
				for (uint32_t i = tid; i < ngr; i += TH) {
					const uint32_t gq = list[i];
					uint64_t x[6];
					bool m[6];
					uint32_t bf[6];
#pragma unroll
					for (uint32_t j = 0; j < 6; ++j) {
						x[j] = sg[min(gq + j, lastp)];
						bf[j] = 0;
					}
					m[0] = m[1] = true;
#pragma unroll
					for (uint32_t j = 2; j < 6; ++j) m[j] = m[j - 1] && gq + j <= lastp && (x[j] ^ x[0]) < glim;
#pragma unroll
					for (uint32_t a = 0; a < 6; ++a) {
#pragma unroll
						for (uint32_t b2 = a + 1; b2 < 6; ++b2) {
							const bool gt = x[a] > x[b2]; // (equal keys keep their order)
							bf[a] += m[b2] && gt ? 1u : 0u;
							bf[b2] += m[b2] && !gt ? 1u : 0u;
						}
					}
#pragma unroll
					for (uint32_t j = 0; j < 6; ++j) {
						if (m[j]) {
							stage[off + gq + bf[j]] = x[j];
							if constexpr (HV) dl[gq + j] = (int8_t)((int)bf[j] - (int)j);
						}
					}
				}
				__syncthreads();
				if constexpr (HV) {
#pragma unroll
					for (int u = 0; u < NK; ++u) {
						const uint32_t idx = (pr[u] >> 31) ? 0u : (pr[u] & 0x7FFFFFFFu);
						pr[u] = (pr[u] & 0x80000000u) | (uint32_t)((int)idx + (int)dl[idx]);
					}
				}
			} else {
				// ---- a group of 7 or more, or more groups than the list holds (keys with many copies): every position looks
				// for its element's place itself.
				if (tid == 0) atomicAdd(&ctr->l17_slow, 1u);
				// two elements kL17MaxGroup apart share a group only if the group is longer than that
				bool too_long = false;
				for (uint32_t i = tid; i + kL17MaxGroup < n; i += TH)
					if ((sg[i] ^ sg[i + kL17MaxGroup]) < glim) too_long = true;
				if (too_long) wtot[18] = 1;
				__syncthreads();
				if (wtot[18] != 0) { // (nothing has been written back: the segment goes to the register partition + the small leaves)
					if (tid == 0) rejected[atomicAdd(nrejected, 1u)] = g;
					__syncthreads();
					continue;
				}
				// every position finds its element's place inside its group: first slot of the group + the members with a smaller
				// key (or an equal key further left); LDS is only read.  The loop runs over POSITIONS (not unrolled: done for a
				// thread's own 17 elements in registers, the look-ups of all of them are in flight at once and a hundred registers
				// spill) and leaves the distance to the final place -- within +-48 -- as a byte; the elements' owners pick it up
				// behind the barrier.
				constexpr uint32_t WIN = MSD_L17_WIN;
#pragma unroll 1
				for (uint32_t idx = tid; idx < n; idx += TH) {
					const uint64_t me = sg[idx];
					uint64_t lk[WIN], rk[WIN];
#pragma unroll
					for (uint32_t d = 0; d < WIN; ++d) {
						lk[d] = sg[idx > d ? idx - d - 1 : 0u];
						rk[d] = sg[min(idx + d + 1, n - 1u)];
					}
					uint32_t left = 0, before = 0, right = 0;
					bool ml = true, mr = true;
#pragma unroll
					for (uint32_t d = 0; d < WIN; ++d) {
						ml = ml && idx > d && (lk[d] ^ me) < glim;             // members to the left: those <= me come first
						before += ml && lk[d] <= me ? 1u : 0u;
						left += ml ? 1u : 0u;
						mr = mr && idx + d + 1 < n && (rk[d] ^ me) < glim;     // members to the right: those < me come first
						before += mr && rk[d] < me ? 1u : 0u;
						right += mr ? 1u : 0u;
					}
					if (ml) { // the group goes on beyond the window
						while (idx > left) {
							const uint64_t o = sg[idx - left - 1];
							if ((o ^ me) >= glim) break;
							before += o <= me ? 1u : 0u;
							++left;
						}
					}
					if (mr) {
						for (uint32_t e = idx + right + 1; e < n; ++e) {
							const uint64_t o = sg[e];
							if ((o ^ me) >= glim) break;
							before += o < me ? 1u : 0u;
						}
					}
					dl[idx] = (int8_t)((int)before - (int)left);
				}
				__syncthreads();
				// (the elements' owners take their keys back from the staging buffer: the key registers hold the next segment's)
				uint64_t kk[NK];
#pragma unroll
				for (int u = 0; u < NK; ++u) {
					const uint32_t idx = (pr[u] >> 31) ? 0u : (pr[u] & 0x7FFFFFFFu);
					kk[u] = sg[idx];
					pr[u] = (pr[u] & 0x80000000u) | (uint32_t)((int)idx + (int)dl[idx]);
				}
				__syncthreads(); // every look-up is done: the staging buffer is free
#pragma unroll
				for (int u = 0; u < NK; ++u) {
					uint64_t *o = (pr[u] >> 31) ? junk + lane : stage + (pr[u] & 0x7FFFFFFFu) + off;
					*o = kk[u];
				}
			}
		}
		// ---- keys, then payloads: to their final places in the staging buffer (on the array's 16-byte grid), out as whole vectors
		auto place = [&](const uint64_t(&x)[NK]) {
#pragma unroll
			for (int u = 0; u < NK; ++u) {
				uint64_t *o = (pr[u] >> 31) ? junk + lane : stage + (pr[u] & 0x7FFFFFFFu) + off;
				*o = x[u];
			}
		};
		auto store_out = [&](uint64_t *gb) {
			// (an opaque copy of the thread index per use: the nine vector addresses, common to the key loads, the payload
			// loads and both stores, are otherwise kept in registers across the whole segment -- and spilled)
			uint32_t tq = tid;
			asm volatile("" : "+v"(tq));
			const uint32_t v_first = off, v_end = tot >> 1; // whole vectors: [v_first, v_end)
#pragma unroll
			for (int v = 0; v < NV; ++v) {
				const uint32_t q = (uint32_t)(v * TH) + tq;
				if (q >= v_first && q < v_end) reinterpret_cast<u32x4 *>(gb)[q] = reinterpret_cast<const u32x4 *>(stage)[q];
			}
			{
				const uint32_t q = (uint32_t)(NV * TH) + tq; // (at most kL17Cap / 2 - NV * TH = 512 more vectors)
				if (q < v_end) reinterpret_cast<u32x4 *>(gb)[q] = reinterpret_cast<const u32x4 *>(stage)[q];
			}
			if (tq == 0) { // the single elements at both ends
				if (off && tot > 1) gb[1] = stage[1];
				if ((tot & 1u) && tot - 1u >= off && (tot - 1u != 1u || !off)) gb[tot - 1u] = stage[tot - 1u];
			}
		};
		MSD_STAMP(5); // fix-up
		// (the keys lie at their final places in the staging buffer)
		// The payloads start travelling once the keys have left their registers, i.e. while the keys are stored.  (Loaded
		// earlier -- before the fix-up, or before the keys' last LDS write -- they are live together with the 34 key
		// registers: the compiler parks registers in scratch memory, which profiles/pmc_traffic_c5a.json showed as 13 GB of
		// HBM traffic per launch on top of the 34 GB of tuples, and the kernel runs slower, not faster.)
		uint64_t r[HV ? NK : 1];
		if constexpr (HV) {
			uint32_t tq = tid;
			asm volatile("" : "+v"(tq));
#pragma unroll
			for (int v = 0; v < NV; ++v) {
				const uint32_t q = min((uint32_t)(v * TH) + tq, lastv) * 2u;
				const u32x4 b = *reinterpret_cast<const u32x4 *>(vb + q);
				r[2 * v] = (uint64_t)b.x | ((uint64_t)b.y << 32);
				r[2 * v + 1] = (uint64_t)b.z | ((uint64_t)b.w << 32);
			}
			r[NK - 1] = vb[min((uint32_t)(NV * TH * 2) + tq, tot - 1u)];
		}
		__syncthreads();
		store_out(kb);
		__syncthreads();
		MSD_STAMP(6); // keys out
		if constexpr (HV) {
			place(r);
			__syncthreads();
			store_out(vb);
			__syncthreads();
		}
		MSD_STAMP(7); // payloads out
	}
	MSD_STAMP_FLUSH(TH / 64);
}

} // namespace msd
