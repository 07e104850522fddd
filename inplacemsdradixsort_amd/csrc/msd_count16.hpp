// msd_count16.hpp -- the one-pass counting leaf for the case the planner aims at: u32 keys, 16 open bits (or a few
// less: the values are then spread over the 2^16 counters by a shift), a segment that fits the registers of one
// 512-thread workgroup (included by msd_device.hpp).
//
// Same algorithm as count_place_kernel (2^16 byte counters in LDS, the fetch-add's return value is the key's
// rank among equal keys, place = base[owner of the value] + prefix[value] + rank, keys are re-generated as
// prefix | value), rebuilt around what profiles/r02_stamps_count_place_before.json showed: 17 four-byte
// loads and 17 four-byte stores per thread cost 12 of a segment's 37 thousand cycles (vector memory
// instructions issue at a fixed rate, whatever their width), the counters were read three times, and
// every conditional fetch-add waited for the one before it.  Here:
//   * global loads and stores are 16 bytes per lane: the segment is handled on the 16-byte grid of the array
//     (a segment starts at any element; the up to three elements in front of it are masked), the re-generated
//     keys are staged in LDS relative to that grid and leave as whole vectors;
//   * a thread's 32 counter words (128 values) are read ONCE (ds_read_b128), summed, scanned across the
//     workgroup and turned into byte prefixes in registers, then written back (layout: four words of padding
//     per 64 words, which keeps 16-byte alignment and makes the b128 accesses bank-conflict free);
//   * fetch-adds, position look-ups and output writes are branch-free and issued back to back; an element
//     outside the segment uses the lane's junk words;
//   * 512-thread workgroups, two per CU, 128 VGPRs.
// What this kernel does not take -- other bit counts, longer segments, a thread with more than 255 keys, an
// overflowing byte -- is queued untouched for count_place_kernel / count_walk_kernel.
#pragma once

namespace msd {

#ifndef MSD_C16_TH // (overridable for experiments)
#define MSD_C16_TH 512
#endif
constexpr int kC16Th = MSD_C16_TH;
constexpr int kC16Vec = 4096 / kC16Th;                       // 16-byte vectors per thread: TH * NV * 4 = 16384 elements
constexpr int kC16Tail = 1024 / kC16Th;                      // + scalar elements per thread behind them
constexpr uint32_t kC16Cap = kC16Th * (kC16Vec * 4 + kC16Tail); // 17408 elements on the 16-byte grid
constexpr uint32_t kC16MinBits = 9;                          // fewer open bits: more than 255 copies per value are the rule (byte counters)
constexpr uint32_t kC16Words = 16384;                        // counter words (4 byte counters each)
constexpr uint32_t kC16CwWords = kC16Words + (kC16Words >> 6) * 4; // with 4 words of padding per 64
static_assert(kC16CwWords == kC16Cap, "counters and output buffer share one LDS area");
// [counters | output buffer][per-thread bases][junk words: 64 counters + 64 keys][wave totals, flags]
constexpr size_t kC16Lds = (size_t)kC16Cap * 4 + kC16Th * 4 + 128 * 4 + 128;
__device__ __forceinline__ uint32_t c16_at(uint32_t w) { return w + ((w >> 6) << 2); }

__global__ __launch_bounds__(kC16Th, (kC16Th >= 1024 ? 8 : 4)) void count_place16_kernel(uint32_t *__restrict__ keys,
	const Segment *__restrict__ segs, uint32_t nsegs, Segment *__restrict__ rejected, Counters *__restrict__ ctr,
	uint64_t n_total)
{
	constexpr int TH = kC16Th, NV = kC16Vec, NT = kC16Tail, NK = NV * 4 + NT;
	constexpr int CH = TH >= 1024 ? 4 : 8; // fetch-adds / look-ups in flight per thread (register budget: 64 or 128 VGPRs)
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint32_t *cw = reinterpret_cast<uint32_t *>(smem);  // packed byte counters (padded layout) ...
	uint32_t *out = reinterpret_cast<uint32_t *>(smem); // ... later the output buffer, on the array's 16-byte grid
	uint32_t *tbase = cw + kC16Cap;                     // per-thread output base
	uint32_t *junkc = tbase + TH;                       // per-lane junk counter / junk output word
	uint32_t *junko = junkc + 64;
	uint32_t *wtot = junko + 64;                        // 8 wave totals
	uint32_t *nexti = wtot + 9, *hi_l = wtot + 10, *crowded = wtot + 11;
	const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	if (blockIdx.x >= nsegs) return;
	if (tid == 0) wtot[13] = 0; // tickets in hand (only thread 0 uses them)
	// segment descriptors are the same in every lane: keep them in scalar registers (loaded through vector
	// memory or LDS they would make every address a per-lane 64-bit computation)
	auto uniform = [](Segment g) -> Segment {
		Segment u;
		// (the builtin returns a signed int: without the cast to uint32_t a low word >= 2^31 sign-extends into the high
		// word -- element offsets that large exist only in arrays of more than 2^31 keys.  THIS was round 2's GPU fault
		// at 2^32 keys: `keys + start` pointed 16 GiB below the array and the segment's first 16-byte load faulted.)
		auto rfl = [](uint32_t x) -> uint64_t { return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(x); };
		u.start = rfl((uint32_t)g.start) | (rfl((uint32_t)(g.start >> 32)) << 32);
		u.count = rfl((uint32_t)g.count) | (rfl((uint32_t)(g.count >> 32)) << 32);
		u.bits = (uint32_t)rfl(g.bits);
		u.pad = 0;
		return u;
	};
	Segment sg = uniform(segs[blockIdx.x]);

	// ONE register per key for its whole life: the key (prefetched) -> value | rank << 16 -> value | place << 16 -> dead,
	// by which time the next segment's keys are loaded into the same registers
	uint32_t rk[NK];
	// the segment's elements on the 16-byte grid: vector v of thread t = grid elements (v * TH + t) * 4 .. + 3,
	// tail element s of thread t = grid element NV * TH * 4 + s * TH + t.  Branch-free: lanes beyond the
	// segment re-read its last vector / element; a segment this kernel will not take is not read.
	auto prefetch = [&](const Segment &g) {
		const uint32_t off = (uint32_t)(g.start & 3u);
		const uint32_t *base = keys + (g.start - off);
		const uint64_t tot = g.count + off;
		// (the only other access of this kernel that can leave the array: the 16-byte vector holding the LAST elements of
		// the array's last segment may extend up to 12 bytes behind the allocation -- such a segment is not taken, n_total
		// is the array's length)
		const bool take = g.bits >= kC16MinBits && g.bits <= 16 && tot <= (uint64_t)kC16Cap && ((g.start - off + tot + 3) & ~3ull) <= n_total;
		const uint32_t totc = take ? (uint32_t)tot : 1u;
		const uint32_t lastv = (totc - 1u) >> 2;
#pragma unroll
		for (int v = 0; v < NV; ++v) {
			const u32x4 q = *reinterpret_cast<const u32x4 *>(base + min((uint32_t)(v * TH) + tid, lastv) * 4u);
			rk[v * 4 + 0] = q.x; rk[v * 4 + 1] = q.y; rk[v * 4 + 2] = q.z; rk[v * 4 + 3] = q.w;
		}
#pragma unroll
		for (int s = 0; s < NT; ++s) rk[NV * 4 + s] = base[min((uint32_t)(NV * TH * 4 + s * TH) + tid, totc - 1u)];
	};
	prefetch(sg);
	MSD_STAMP_DECL(2);
	MSD_STAMP_START();
	for (;;) {
		const uint32_t off = (uint32_t)(sg.start & 3u);
		const uint32_t tot = (uint32_t)min(sg.count + off, (uint64_t)0xFFFFFFFFu), n = tot - off;
		const bool fits = sg.bits >= kC16MinBits && sg.bits <= 16 && sg.count + off <= (uint64_t)kC16Cap && ((sg.start + sg.count + 3) & ~3ull) <= n_total;
		const uint32_t vsh = 16u - (fits ? sg.bits : 16u); // a value of b < 16 bits is counted as value << (16 - b): every thread's counters get their share
		uint32_t *segb = keys + (sg.start - off); // 16-byte aligned
		MSD_STAMP(9);
		MSD_STAMP_TICK(11);
		// ---- clear the counters (16-byte stores)
		// (loop invariants of one use per segment -- a vector of zeros, a lane's slot in the wave totals -- are made
		// opaque: hoisted out of the loop they are spilled, and their reload waits for the stores in flight)
		uint32_t zero = 0, tq = tid;
		asm volatile("" : "+v"(zero), "+v"(tq));
#pragma unroll
		for (uint32_t j = 0; j < (kC16Cap / 4 + TH - 1) / TH; ++j) {
			const uint32_t q = j * TH + tq;
			if (q < kC16Cap / 4) reinterpret_cast<u32x4 *>(cw)[q] = u32x4{ zero, zero, zero, zero };
		}
		if (tid == 0) {
			// (tickets two at a time when there are many segments: fewer fetch-adds on the one ticket word; the pair
			// in hand lives in LDS, wtot[12] next / wtot[13] how many)
			if (wtot[13] == 0) {
				const uint32_t take = nsegs > 64u * gridDim.x ? 2u : 1u;
				wtot[12] = atomicAdd(&ctr->count_ticket3, take) + gridDim.x;
				wtot[13] = take;
			}
			*nexti = wtot[12];
			wtot[12] += 1;
			wtot[13] -= 1;
			*crowded = 0;
			const uint32_t k0 = off == 0 ? rk[0] : off == 1 ? rk[1] : off == 2 ? rk[2] : rk[3]; // first key
			*hi_l = k0 & ~((1u << sg.bits) - 1u); // common prefix of the whole segment
		}
		MSD_STAMP(0); // clear
		__syncthreads();
		MSD_STAMP(1);
		// ---- one fetch-add per key: value (low 16 bits) | rank among equal keys << 16; elements outside the segment
		// bump the lane's junk counter
		if (fits) {
			// (eight fetch-adds in flight at a time)
#pragma unroll
			for (int u0 = 0; u0 < NK; u0 += CH) {
				uint32_t old[CH];
#pragma unroll
				for (int i = 0; i < CH; ++i) {
					const int u = u0 + i;
					if (u < NK) {
						const uint32_t el = u < NV * 4 ? (uint32_t)((u / 4) * TH * 4) + tid * 4 + (u % 4) : (uint32_t)(NV * TH * 4 + (u - NV * 4) * TH) + tid;
						const uint32_t val = (rk[u] << vsh) & 0xFFFFu; // (fewer than 16 open bits: spread over the counters, order kept)
						const bool in = el >= off && el < tot;
						const uint32_t a = in ? c16_at(val >> 2) : (uint32_t)(junkc - cw) + lane; // word index from cw
						old[i] = atomicAdd(cw + a, 1u << ((val & 3u) * 8u));
						rk[u] = val | (in ? 0u : 0x80000000u);
					}
				}
#pragma unroll
				for (int i = 0; i < CH; ++i) {
					const int u = u0 + i;
					if (u < NK) rk[u] |= ((old[i] >> ((rk[u] & 3u) * 8u)) & 0xFFu) << 16;
				}
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		MSD_STAMP(2); // fetch-adds (incl. the wait for the keys)
		__syncthreads();
		MSD_STAMP(3);
		const uint32_t nxt = (uint32_t)__builtin_amdgcn_readfirstlane(*nexti), hi = (uint32_t)__builtin_amdgcn_readfirstlane(*hi_l);
		// the next segment's descriptor travels during the counter phases (loaded where its keys are prefetched,
		// its whole memory latency would sit in front of that prefetch)
		const Segment nraw = segs[nxt < nsegs ? nxt : blockIdx.x];
		// ---- the thread's 32 counter words, once: byte sums -> scan over the workgroup -> byte prefixes
		constexpr int WPT = (int)(kC16Words / TH); // counter words per thread (32 or 16: whole 16-byte vectors, inside one 64-word group)
		u32x4 *cq = reinterpret_cast<u32x4 *>(cw + c16_at(tid * (uint32_t)WPT)); // (16-byte aligned)
		uint32_t cr[WPT];
		uint32_t totk = 0;
		if (fits) {
#pragma unroll
			for (int j = 0; j < WPT / 4; ++j) {
				const u32x4 q = cq[j];
				cr[4 * j + 0] = q.x; cr[4 * j + 1] = q.y; cr[4 * j + 2] = q.z; cr[4 * j + 3] = q.w;
			}
#pragma unroll
			for (int j = 0; j < WPT; ++j) totk = __builtin_amdgcn_sad_u8(cr[j], 0u, totk);
		}
		if (totk > 255u) *crowded = 1; // (a byte prefix would not fit)
		const uint32_t inc = wave_incl_scan(totk);
		if ((tq & 63u) == 63u) wtot[tq >> 6] = inc;
		MSD_STAMP(4); // counters read + byte sums + wave scan
		__syncthreads();
		uint32_t pos = inc - totk, all = 0;
#pragma unroll
		for (uint32_t ww = 0; ww < TH / 64; ++ww) {
			const uint32_t t = wtot[ww];
			if (ww < w) pos += t;
			all += t;
		}
		// a byte that overflowed carried into its neighbour: the sum of all bytes then falls short of n
		const bool ok = fits && all == n && *crowded == 0;
		MSD_STAMP(5);
		if (ok) {
			uint32_t run = 0;
#pragma unroll
			for (int j = 0; j < WPT / 4; ++j) {
#pragma unroll
				for (int e = 0; e < 4; ++e) {
					const uint32_t x = cr[4 * j + e], y = x * 0x01010101u; // bytes of y: inclusive sums inside the word
					cr[4 * j + e] = (y - x) + run * 0x01010101u;
					run += y >> 24;
				}
				cq[j] = u32x4{ cr[4 * j + 0], cr[4 * j + 1], cr[4 * j + 2], cr[4 * j + 3] };
			}
			tbase[tid] = pos + off; // (output positions are on the array's 16-byte grid)
			MSD_STAMP(6); // byte prefixes
			__syncthreads();
			// ---- place of every key: base[owner of its value] + prefix[value] + rank
			// (the place, < 2^15, replaces the rank in bits 16..30)
#pragma unroll
			for (int u0 = 0; u0 < NK; u0 += CH) {
				uint32_t tb[CH], cv[CH];
#pragma unroll
				for (int i = 0; i < CH; ++i) {
					if (u0 + i < NK) {
						const uint32_t wi = (rk[u0 + i] & 0xFFFFu) >> 2;
						tb[i] = tbase[wi / (uint32_t)WPT];
						cv[i] = cw[c16_at(wi)];
					}
				}
#pragma unroll
				for (int i = 0; i < CH; ++i) {
					if (u0 + i < NK) {
						const uint32_t r = rk[u0 + i];
						const uint32_t pl = tb[i] + ((cv[i] >> ((r & 3u) * 8u)) & 0xFFu) + ((r >> 16) & 0xFFu);
						rk[u0 + i] = (r & 0x8000FFFFu) | (pl << 16);
					}
				}
				__builtin_amdgcn_sched_barrier(0);
			}
			MSD_STAMP(7); // positions
			__syncthreads(); // counters are dead: the area is the output buffer now
#pragma unroll
			for (int u = 0; u < NK; ++u) {
				uint32_t *o = (rk[u] >> 31) ? junko + lane : out + ((rk[u] >> 16) & 0x7FFFu);
				*o = hi | ((rk[u] & 0xFFFFu) >> vsh);
			}
			__syncthreads();
			MSD_STAMP(8); // output into LDS
		} else if (tid == 0)
			rejected[atomicAdd(&ctr->nslow16, 1u)] = sg; // untouched
		Segment nsg = sg;
		if (nxt < nsegs) { // the next segment's keys travel while this one is stored
			nsg = uniform(nraw);
			prefetch(nsg);
		}
		if (ok) {
			// whole vectors inside [off, tot) leave as 16-byte stores, the first and the last vector element-wise
			const uint32_t v_first = off ? 1u : 0u, v_end = tot >> 2; // full vectors: [v_first, v_end)
#pragma unroll
			for (int v0 = 0; v0 < NV; v0 += 4) { // (four vectors in flight: the next segment's keys occupy 34 registers by now)
				u32x4 t4[4];
#pragma unroll
				for (int i = 0; i < 4; ++i) t4[i] = reinterpret_cast<const u32x4 *>(out)[(uint32_t)((v0 + i) * TH) + tid];
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const uint32_t q = (uint32_t)((v0 + i) * TH) + tid;
					if (q >= v_first && q < v_end) reinterpret_cast<u32x4 *>(segb)[q] = t4[i];
				}
				__builtin_amdgcn_sched_barrier(0);
			}
			// tail elements of full-size segments are a run of whole vectors too
			{
				const uint32_t q = (uint32_t)(NV * TH) + tq; // (at most kC16Cap / 4 - NV * TH = 256 of them)
				if (q < v_end) reinterpret_cast<u32x4 *>(segb)[q] = reinterpret_cast<const u32x4 *>(out)[q];
			}
			if (tq < 4) { // the partial vectors at both ends
				if (off && tq >= off && tq < tot) segb[tq] = out[tq];
				const uint32_t el = (v_end << 2) + tq;
				if (el < tot && el >= off && (el >= 4u || !off)) segb[el] = out[el];
			}
		}
		MSD_STAMP(10); // prefetch issue + store
		if (nxt >= nsegs) break;
		sg = nsg;
		__syncthreads(); // the output buffer is cleared next
	}
	MSD_STAMP_FLUSH(TH / 64);
}

} // namespace msd
