// msd_regpart.hpp -- one digit pass over segments that fit the registers of one workgroup (included by msd_device.hpp).
//
// The general round (classify -> block metadata -> block permutation -> cleanup) is built for parents far larger
// than a workgroup's stripe.  On the last partition round of the tuple sort (2^30 tuples: 65536 parents of about
// 2^14 tuples, 2-3 bit digits) it moves every tuple three times and spends a third of the whole sort
// (profiles/r02_kernel_stats_c5a.csv: classify 8 ms + block permutation and metadata about 10 ms).  Such a parent --
// up to 17408 elements -- fits the registers of one 1024-thread workgroup: it is read once (16 bytes per lane, on the
// array's 16-byte grid like count_place16_kernel), every element gets digit and rank from one LDS fetch-add,
// bucket starts come from a scan of the (at most 256) counters, and keys, then payloads, go through an LDS staging
// buffer to their places and back out as whole vectors.  One read and one write per element, no block map, no lists.
// This is the reference's in-cache partition_ip (src/msb_64.c:740-770: histogram, prefix sum, permute inside the cache)
// with the register file and LDS as the cache; like it, the pass is unstable.
#pragma once

namespace msd {

constexpr int kRpTh = 1024;
constexpr int kRpVec = 8;                                        // 16-byte vectors (2 elements) per thread and array
constexpr uint32_t kRpCap = kRpTh * (kRpVec * 2 + 1);             // 17408 elements on the 16-byte grid (+ 1 scalar per thread)
constexpr size_t kRpLds = (size_t)kRpCap * 8 + kP * 4 * 2 + 64 * 8 + 64; // staging | counters, starts | junk | misc

// The narrowest digit whose children fit the leaf sorter with a quarter to spare (the leaf's per-segment costs favour
// few large segments; a child that turns out too big simply takes another pass).  Host and device use the same rule.
__host__ __device__ inline uint32_t regpart_width(uint64_t count, uint32_t bits, uint64_t small_max)
{
	uint32_t w = 1;
	while (w < 8 && w < bits && (count >> w) > small_max - small_max / 4) ++w;
	return w < bits ? w : bits;
}

// Plans a register-resident round on the device: the previous round's next-parent list -> Parent records (digit width
// by the rule above, child ranges handed out by one fetch-add each; their order does not matter).  Saves the host the
// round trip of the list (65536 segments at 2^30 tuples: 1.5 ms).
__global__ __launch_bounds__(256) void regpart_plan_kernel(const Segment *__restrict__ segs, uint32_t n, uint64_t small_max,
	Parent *__restrict__ parents, Counters *__restrict__ ctr)
{
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const Segment s = segs[i];
	Parent p;
	p.start = s.start;
	p.count = s.count;
	p.width = regpart_width(s.count, s.bits, small_max);
	p.shift = s.bits - p.width;
	p.child_base = atomicAdd(&ctr->rp_children, 1u << p.width);
	p.stripe_lo = p.stripe_hi = 0;
	p.pad = 0;
	parents[i] = p;
}

template <typename V>
__global__ __launch_bounds__(kRpTh, 4) void regpart_kernel(uint64_t *__restrict__ keys, uint64_t *__restrict__ vals,
	const Parent *__restrict__ parents, uint32_t nparents, ChildArrays ca, Counters *__restrict__ ctr)
{
	constexpr bool HV = has_val<V>::value;
	constexpr int TH = kRpTh, NV = kRpVec, NK = NV * 2 + 1;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint64_t *stage = reinterpret_cast<uint64_t *>(smem);            // kRpCap elements, on the array's 16-byte grid
	uint32_t *cnt = reinterpret_cast<uint32_t *>(smem + (size_t)kRpCap * 8); // per-bucket counters
	uint32_t *bstart = cnt + kP;                                      // per-bucket first place
	uint64_t *junk = reinterpret_cast<uint64_t *>(bstart + kP);      // per-lane junk word
	uint32_t *misc = reinterpret_cast<uint32_t *>(junk + 64);
	const uint32_t tid = threadIdx.x, lane = tid & 63;
	auto rfl = [](uint32_t x) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane(x); };

	for (uint32_t pi = blockIdx.x; pi < nparents; pi += gridDim.x) {
		const Parent pa = parents[pi];
		// (uniform values in scalar registers: addresses below are base + 32-bit lane offset)
		const uint64_t start = (uint64_t)rfl((uint32_t)pa.start) | ((uint64_t)rfl((uint32_t)(pa.start >> 32)) << 32);
		const uint32_t n = rfl((uint32_t)pa.count), shift = rfl(pa.shift), width = rfl(pa.width), cbase = rfl(pa.child_base);
		const uint32_t mask = (1u << width) - 1u, nb = 1u << width;
		const uint32_t off = (uint32_t)(start & 1u), tot = n + off; // the parent on the 16-byte grid: elements [off, tot)
		uint64_t *kb = keys + (start - off), *vb = HV ? vals + (start - off) : nullptr;
		if (tot > kRpCap || pa.count > kRpCap) { // (the host only sends parents that fit)
			if (tid == 0) msd_note_error(ctr, 11u);
			continue;
		}
		if (tid < nb) cnt[tid] = 0;
		// ---- read: vector v of thread t = grid elements (v * TH + t) * 2, + 1; tail element NV * TH * 2 + t
		uint64_t k[NK], r[HV ? NK : 1];
		const uint32_t lastv = (tot - 1u) >> 1;
#pragma unroll
		for (int v = 0; v < NV; ++v) {
			const uint32_t q = min((uint32_t)(v * TH) + tid, lastv) * 2u; // (beyond the parent: its last vector again, ignored)
			const u32x4 a = *reinterpret_cast<const u32x4 *>(kb + q);
			k[2 * v] = (uint64_t)a.x | ((uint64_t)a.y << 32);
			k[2 * v + 1] = (uint64_t)a.z | ((uint64_t)a.w << 32);
			if constexpr (HV) {
				const u32x4 b = *reinterpret_cast<const u32x4 *>(vb + q);
				r[2 * v] = (uint64_t)b.x | ((uint64_t)b.y << 32);
				r[2 * v + 1] = (uint64_t)b.z | ((uint64_t)b.w << 32);
			}
		}
		{
			const uint32_t q = min((uint32_t)(NV * TH * 2) + tid, tot - 1u);
			k[NK - 1] = kb[q];
			if constexpr (HV) r[NK - 1] = vb[q];
		}
		__syncthreads();
		// ---- digit and rank inside the bucket: one LDS fetch-add per element (elements outside the parent: junk)
		uint32_t dr[NK]; // digit | rank << 8, later the place; bit 31: not an element of the parent
#pragma unroll
		for (int u = 0; u < NK; ++u) {
			const uint32_t el = u < NV * 2 ? (uint32_t)((u / 2) * TH * 2) + tid * 2 + (u % 2) : (uint32_t)(NV * TH * 2) + tid;
			const bool in = el >= off && el < tot;
			const uint32_t d = (uint32_t)(k[u] >> shift) & mask;
			uint32_t *a = in ? cnt + d : reinterpret_cast<uint32_t *>(junk + lane);
			dr[u] = d | (atomicAdd(a, 1u) << 8) | (in ? 0u : 0x80000000u);
		}
		__syncthreads();
		// ---- bucket starts (one wave scans the counters) and the children's geometry
		if (tid < 64) {
			uint32_t run = 0;
			for (uint32_t b0 = 0; b0 < nb; b0 += 64) {
				const uint32_t c = b0 + tid < nb ? cnt[b0 + tid] : 0u;
				const uint32_t inc = wave_incl_scan(c);
				if (b0 + tid < nb) {
					bstart[b0 + tid] = run + inc - c;
					ca.start[cbase + b0 + tid] = start + run + inc - c;
					ca.count[cbase + b0 + tid] = c;
				}
				run += (uint32_t)__shfl((int)inc, 63);
			}
			if (tid == 0) misc[0] = run;
		}
		__syncthreads();
		if (misc[0] != n && tid == 0) msd_note_error(ctr, 12u);
#pragma unroll
		for (int u = 0; u < NK; ++u) dr[u] = (dr[u] & 0x80000000u) | (bstart[dr[u] & 0xFFu] + ((dr[u] >> 8) & 0x7FFFFFu) + off);
		// ---- keys, then payloads: to their places in the staging buffer, out as whole vectors
		auto permute = [&](const uint64_t(&x)[NK], uint64_t *gb) {
			// (opaque copy of the thread id: the vector offsets below, hoisted out of the parent loop as invariants, are
			// spilled and their reloads wait for the stores in flight)
			uint32_t tq = tid;
			asm volatile("" : "+v"(tq));
#pragma unroll
			for (int u = 0; u < NK; ++u) {
				uint64_t *o = (dr[u] >> 31) ? junk + lane : stage + (dr[u] & 0x7FFFFFFFu);
				*o = x[u];
			}
			__syncthreads();
			const uint32_t v_first = off, v_end = tot >> 1; // whole vectors: [v_first, v_end)
#pragma unroll
			for (int v = 0; v < NV; ++v) {
				const uint32_t q = (uint32_t)(v * TH) + tq;
				if (q >= v_first && q < v_end) reinterpret_cast<u32x4 *>(gb)[q] = reinterpret_cast<const u32x4 *>(stage)[q];
			}
			{
				const uint32_t q = (uint32_t)(NV * TH) + tq; // (at most kRpCap / 2 - NV * TH = 512 more vectors)
				if (q < v_end) reinterpret_cast<u32x4 *>(gb)[q] = reinterpret_cast<const u32x4 *>(stage)[q];
			}
			if (tq == 0) { // the single elements at both ends
				if (off && tot > 1) gb[1] = stage[1];
				if ((tot & 1u) && tot - 1u >= off && (tot - 1u != 1u || !off)) gb[tot - 1u] = stage[tot - 1u];
			}
			__syncthreads();
		};
		permute(k, kb);
		if constexpr (HV) permute(r, vb);
	}
}

} // namespace msd
