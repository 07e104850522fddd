// msd_bigcount.hpp -- counting sort for keys-only segments of any size with <= 16 open bits (included by msd_device.hpp).
//
// The heavy buckets of skewed inputs (Zipf keys: half of 2^30 keys share the top 16 bits), inputs of a small key range
// and low-cardinality inputs end here: ONE read of the segment counts its values, a scan turns counts into positions,
// and the segment is then re-generated from the prefix array and written once -- all remaining partition rounds are
// replaced, whatever the distribution.  (The reference has no such leaf: skew goes through its range splitters,
// src/msb_64.c:1304-1322, and duplicates degrade its comb/insertion sorts)
//
//   bigcount_hist_kernel   persistent workgroups (one per CU), each a contiguous share of the 2^18-key chunks: all 2^16
//                          values are counted in LDS in ONE pass with 16-bit counters packed in pairs; a counter that
//                          reaches 2^14 hands that much to the segment's histogram in HBM and continues (the lane whose
//                          fetch-add crossed the mark does it), so a share may hold any number of equal keys.  16-byte
//                          loads on the array's 16-byte grid, two batches of four in flight; about 11 instructions per
//                          key (the kernel is bound by instruction issue and by the CU's load queue, not by LDS).
//   bigcount_scan_kernel   one workgroup per segment: exclusive prefix over the 2^16 counts, in place; the common key
//                          prefix; and, per output tile, the value whose run covers the tile's first element -- the
//                          writers then start from one table look-up instead of a search over the prefix array.
//   bigcount_write_kernel  one 256-thread workgroup per group of consecutive 30 KiB tiles, four per CU; every look-up
//                          of the group (tile table, prefix entries -- into LDS) happens before its first store.  A
//                          tile inside one value's run is stored straight from registers; otherwise runs are laid out
//                          in LDS -- short ones by one lane each, long ones by whole waves or the whole workgroup -- and
//                          leave as 16-byte vectors.
//
// Round 1 read every chunk twice (32-bit counters for half of the value range per pass) with one workgroup per 2^22
// keys -- half of the CUs idle on a 2^29-key segment --, took ~60 instructions per key on skewed values (a ballot loop
// that grouped equal values, always on for Zipf keys) and filled a tile covered by one long run with a single wave.
// Zipf 2^30 (866 M keys in 369 such segments, 3.5 GB): count 2.04 -> 0.80 ms (4.3 TB/s), write 1.33 -> 0.71 ms
// (4.7 TB/s; write-only streaming measures 5.7-6.0), profiles/r02_kernel_stats_c3.csv.
#pragma once

namespace msd {

constexpr uint32_t kBigChunk = 1u << 18;      // keys per unit of histogram work
constexpr int kBigHistTh = 1024;
constexpr size_t kBigHistLds = 32768 * 4 + 256; // 2^16 16-bit counters, a spare word per lane
constexpr uint32_t kBigFlush = 0x4000;        // a counter hands this much over when it gets there
constexpr uint32_t kBigTile = 7680;           // 4-byte words staged per output tile
constexpr int kBigWriteTh = 256;
#ifndef MSD_BIG_GROUP
#define MSD_BIG_GROUP 4
#endif
constexpr uint32_t kBigGroup = MSD_BIG_GROUP; // consecutive tiles per workgroup
constexpr uint32_t kBigPrefixCap = 1024;      // prefix entries of a group's values kept in LDS
#ifndef MSD_BIG_HOT
#define MSD_BIG_HOT 4
#endif
constexpr int kBigHotSet = MSD_BIG_HOT;       // values a wave counts in scalar registers instead of LDS
constexpr uint32_t kBigHotLanes = 16;         // ... once so many of its lanes hold the value at once
constexpr uint32_t kBigRun = 32;              // longer runs are filled cooperatively
constexpr uint32_t kBigHeavyCap = kBigTile / kBigRun;
constexpr size_t kBigWriteLds = (size_t)kBigTile * 4 + 16 + kBigHeavyCap * 12 + 64 + kBigPrefixCap * 4; // four workgroups per CU

template <typename K> constexpr uint32_t big_tile_elems() { return kBigTile * 4 / (uint32_t)sizeof(K); }

// work item -> segment: first[i] = index of segment i's first work item (first[nb] = their number); uniform
__device__ __forceinline__ uint32_t big_find(const uint32_t *__restrict__ first, uint32_t nb, uint32_t item)
{
	uint32_t lo = 0, hi = nb; // first[lo] <= item < first[hi]
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (first[mid] <= item) lo = mid; else hi = mid;
	}
	return lo;
}

template <typename K>
__global__ __launch_bounds__(kBigHistTh) void bigcount_hist_kernel(const K *__restrict__ keys,
	const Segment *__restrict__ segs, const uint32_t *__restrict__ first, uint32_t nb, uint32_t *__restrict__ ghist)
{
	constexpr int TH = kBigHistTh, VEC = 16 / (int)sizeof(K), NV = 4, NH = kBigHotSet;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint32_t *cw = reinterpret_cast<uint32_t *>(smem);
	const uint32_t tid = threadIdx.x;
	uint32_t *spare = cw + 32768 + (tid & 63u); // a word per lane that only ever receives zeros
	if (tid < 64) *spare = 0;
	for (uint32_t j = tid; j < 32768 / 4; j += TH) reinterpret_cast<uint4 *>(cw)[j] = make_uint4(0u, 0u, 0u, 0u);
	// persistent workgroups, each a contiguous share of the chunks: consecutive chunks of one segment keep counting in
	// the same LDS table, which is added to the segment's histogram once (one atomic per non-zero counter) -- a table
	// per chunk would cost 2^16 atomics per 2^18 keys
	MSD_STAMP_DECL(5);
	MSD_STAMP_START();
	const uint32_t nchunks = first[nb];
	const uint32_t i0 = (uint32_t)((uint64_t)nchunks * blockIdx.x / gridDim.x), i1 = (uint32_t)((uint64_t)nchunks * (blockIdx.x + 1) / gridDim.x);
	uint32_t si = i0 < i1 ? big_find(first, nb, i0) : 0u;
	for (uint32_t item = i0; item < i1;) {
		const Segment sg = segs[si];
		const uint32_t item_end = first[si + 1] < i1 ? first[si + 1] : i1; // this workgroup's chunks of segment si
		const uint32_t nv = 1u << sg.bits, mask = nv - 1u;
		uint32_t *gh = ghist + (size_t)si * 65536;
		__syncthreads(); // the table is clear
		// a key of value v: the fetch-add, then the hand-over if it took the counter across the mark
		auto add_rtn = [&](uint32_t v) -> uint32_t {
			const uint32_t sh = (v & 1u) << 4;
			return (atomicAdd(&cw[v >> 1], 1u << sh) >> sh) & 0xFFFFu;
		};
		auto hand_over = [&](uint32_t v, uint32_t old) {
			if (old == kBigFlush - 1u) {
				atomicSub(&cw[v >> 1], kBigFlush << ((v & 1u) << 4));
				atomicAdd(&gh[v], kBigFlush);
			}
		};
		// A value that a quarter of a wave's lanes share (low-cardinality keys) would serialise in the LDS atomic unit,
		// one lane after the other.  Every wave keeps up to NH such values in scalar registers with their counts: a key
		// slot costs one compare + ballot + popcount per entry, and lanes that matched skip the fetch-add.  A value
		// enters the set when the wave's first lane holds it together with 15 more lanes.  (The kernel is bound by
		// instruction issue, about 25 per key slot at the HBM rate: the set is not worth its instructions for milder
		// skew -- Zipf keys, 6 lanes on the hottest value, run faster without.)
		uint32_t hv[NH], hc[NH], nh = 0; // (wave-uniform)
#pragma unroll
		for (int j = 0; j < NH; ++j) {
			hv[j] = 0xFFFFFFFFu;
			hc[j] = 0;
		}
		auto values = [&](const u32x4 &q, uint32_t (&v)[VEC]) {
			if constexpr (sizeof(K) == 4) {
				v[0] = q.x & mask; v[1] = q.y & mask; v[2] = q.z & mask; v[3] = q.w & mask;
			} else {
				v[0] = q.x & mask; v[1] = q.z & mask; // (<= 16 open bits: they are in the low word)
			}
		};
		auto hand_over_any = [&](const uint32_t (&v)[VEC], const uint32_t (&old)[VEC]) {
			bool any = false;
#pragma unroll
			for (int k = 0; k < VEC; ++k) any |= old[k] == kBigFlush - 1u;
			if (any) {
#pragma unroll
				for (int k = 0; k < VEC; ++k) hand_over(v[k], old[k]);
			}
		};
		// the usual case -- every lane has a key, no value is counted in registers: all fetch-adds of the vector before
		// the first result is looked at
		auto count_vec_plain = [&](const u32x4 &q) {
			uint32_t v[VEC], old[VEC];
			values(q, v);
#pragma unroll
			for (int k = 0; k < VEC; ++k) old[k] = atomicAdd(&cw[v[k] >> 1], 1u << ((v[k] & 1u) << 4));
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int k = 0; k < VEC; ++k) old[k] = (old[k] >> ((v[k] & 1u) << 4)) & 0xFFFFu;
			hand_over_any(v, old);
		};
		auto probe_vec = [&](const u32x4 &q, bool in) {
			if (nh >= (uint32_t)NH) return;
			const uint32_t v0 = q.x & mask, vl = (uint32_t)__builtin_amdgcn_readfirstlane((int)v0);
			bool known = false;
#pragma unroll
			for (int j = 0; j < NH; ++j) known |= hv[j] == vl;
			if (!known && __popcll(__ballot(in && v0 == vl)) >= (int)kBigHotLanes) {
#pragma unroll
				for (int j = 0; j < NH; ++j)
					if ((uint32_t)j == nh) hv[j] = vl;
				++nh;
			}
		};
		// lanes without a key (the chunk's end), values counted in registers: branch-free all the same (a lane that has
		// nothing to count adds zero to its own spare word; behind a branch the compiler waits for every single result)
		auto count_vec = [&](const u32x4 &q, bool in) {
			uint32_t v[VEC], old[VEC];
			values(q, v);
			bool mine[VEC];
#pragma unroll
			for (int k = 0; k < VEC; ++k) {
				mine[k] = in;
#pragma unroll
				for (int j = 0; j < NH; ++j) {
					if (nh <= (uint32_t)j) break; // (uniform)
					const bool eq = v[k] == hv[j];
					hc[j] += (uint32_t)__popcll(__ballot(mine[k] && eq));
					mine[k] = mine[k] && !eq;
				}
			}
#pragma unroll
			for (int k = 0; k < VEC; ++k) {
				const uint32_t sh = (v[k] & 1u) << 4;
				uint32_t *at = mine[k] ? cw + (v[k] >> 1) : spare;
				old[k] = (atomicAdd(at, mine[k] ? 1u << sh : 0u) >> sh) & 0xFFFFu;
			}
			hand_over_any(v, old);
		};
		for (; item < item_end; ++item) {
			const uint64_t off = (uint64_t)(item - first[si]) * kBigChunk;
			const uint32_t len = (uint32_t)(sg.count - off < (uint64_t)kBigChunk ? sg.count - off : (uint64_t)kBigChunk);
			const K *src = keys + sg.start + off;
			// the chunk on the array's 16-byte grid: `lead` single keys, whole vectors, single keys again
			const uint32_t mis = (uint32_t)((sg.start + off) & (uint64_t)(VEC - 1));
			const uint32_t lead = min(len, (uint32_t)(VEC - mis) & (uint32_t)(VEC - 1));
			const uint32_t nvec = (len - lead) / VEC, tail0 = lead + nvec * VEC;
			if (tid < lead) {
				const uint32_t v = (uint32_t)src[tid] & mask;
				hand_over(v, add_rtn(v));
			}
			if (tail0 + tid < len) {
				const uint32_t v = (uint32_t)src[tail0 + tid] & mask;
				hand_over(v, add_rtn(v));
			}
			const K *vsrc = src + lead;
			auto load = [&](u32x4 (&a)[NV], uint32_t q0) {
#pragma unroll
				for (int u = 0; u < NV; ++u) // (beyond the chunk: its last vector, ignored)
					a[u] = *reinterpret_cast<const u32x4 *>(vsrc + (size_t)min(q0 + u * TH + tid, nvec - 1u) * VEC);
			};
			auto process = [&](const u32x4 (&a)[NV], uint32_t q0) {
				probe_vec(a[0], q0 + tid < nvec);
				if (nh == 0 && q0 + NV * TH <= nvec) { // (uniform)
#pragma unroll
					for (int u = 0; u < NV; ++u) count_vec_plain(a[u]);
				} else {
#pragma unroll
					for (int u = 0; u < NV; ++u) count_vec(a[u], q0 + u * TH + tid < nvec);
				}
			};
			// two register sets: the next batch of loads is in flight while this one is counted
			u32x4 a[NV], b[NV];
			if (nvec) load(a, 0);
			MSD_STAMP(0); // chunk set-up, single keys
			MSD_STAMP_TICK(11);
			for (uint32_t q0 = 0; q0 < nvec; q0 += 2 * NV * TH) {
				const bool more = q0 + NV * TH < nvec;
				if (more) load(b, q0 + NV * TH);
				MSD_STAMP(1); // load issue
				process(a, q0);
				MSD_STAMP(2); // counting (incl. the wait for the keys)
				if (!more) break;
				if (q0 + 2 * NV * TH < nvec) load(a, q0 + 2 * NV * TH);
				MSD_STAMP(1);
				process(b, q0 + NV * TH);
				MSD_STAMP(2);
			}
		}
		if ((tid & 63u) == 0) {
#pragma unroll
			for (int j = 0; j < NH; ++j)
				if (hc[j]) atomicAdd(&gh[hv[j]], hc[j]);
		}
		__syncthreads();
		for (uint32_t j = tid; j < (nv + 1u) / 2u; j += TH) { // (leaves the table clear)
			const uint32_t w = cw[j];
			if (w) cw[j] = 0;
			if (w & 0xFFFFu) atomicAdd(&gh[2 * j], w & 0xFFFFu);
			if (w >> 16) atomicAdd(&gh[2 * j + 1], w >> 16);
		}
		++si; // (item == first[si + 1], or the share is done)
		MSD_STAMP(3); // merge
	}
	MSD_STAMP_FLUSH(TH / 64);
}

// counts -> exclusive prefix (in place); the segment's common key prefix; the first value of every output tile.
// Wave w owns the contiguous values [w*nv/16, (w+1)*nv/16) and walks them 64 at a time (coalesced).
template <typename K>
__global__ __launch_bounds__(1024) void bigcount_scan_kernel(const K *__restrict__ keys,
	const Segment *__restrict__ segs, const uint32_t *__restrict__ first_group, uint32_t *__restrict__ ghist,
	K *__restrict__ seg_hi, uint16_t *__restrict__ tile_v, Counters *__restrict__ ctr)
{
	constexpr uint32_t TILE = big_tile_elems<K>();
	__shared__ uint32_t wtot[16];
	const uint32_t si = blockIdx.x;
	const Segment sg = segs[si];
	const uint32_t nv = 1u << sg.bits, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	uint32_t *gh = ghist + (size_t)si * 65536;
	const uint32_t per = (nv + 15) / 16;                  // values per wave
	const uint32_t v0 = w * per < nv ? w * per : nv, v1 = v0 + per < nv ? v0 + per : nv;
	uint32_t tot = 0;
	for (uint32_t v = v0 + lane; v < v1; v += 64) tot += gh[v];
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
	if (lane == 0) wtot[w] = tot;
	__syncthreads();
	uint32_t run = 0;
	for (uint32_t ww = 0; ww < w; ++ww) run += wtot[ww];
	const uint32_t total = (uint32_t)sg.count, ntiles = (total + TILE - 1) / TILE;
	for (uint32_t vb = v0; vb < v1; vb += 64) {
		const uint32_t v = vb + lane;
		const uint32_t c = v < v1 ? gh[v] : 0u;
		const uint32_t inc = wave_incl_scan(c);
		if (v < v1) gh[v] = run + inc - c;
		run += __shfl(inc, 63);
	}
	__syncthreads(); // the prefix array is complete
	// tile t starts inside the run of the value v with P[v] <= t * TILE < P[v + 1].  Values go to the waves in turn
	// (the long runs of a skewed segment are neighbours: one wave would fill most of the table)
	uint16_t *tv = tile_v + (size_t)first_group[si] * kBigGroup + si; // (one entry more than tiles per segment)
	for (uint32_t vb = w * 64; vb < nv; vb += 1024) {
		const uint32_t v = vb + lane;
		const uint32_t pb = v < nv ? gh[v] : total, pe = v + 1 < nv ? gh[v + 1] : total;
		const uint32_t t0 = (pb + TILE - 1) / TILE, t1 = pe > pb ? (pe + TILE - 1) / TILE : t0; // tiles starting in [pb, pe)
		const bool wide = t1 - t0 > 4;
		if (!wide)
			for (uint32_t t = t0; t < t1; ++t) tv[t] = (uint16_t)v;
		for (uint64_t m = __ballot(wide); m; m &= m - 1) { // long runs: the whole wave
			const int l = __ffsll((long long)m) - 1;
			const uint32_t a = (uint32_t)__shfl((int)t0, l), b = (uint32_t)__shfl((int)t1, l), vv = (uint32_t)__shfl((int)v, l);
			for (uint32_t t = a + lane; t < b; t += 64) tv[t] = (uint16_t)vv;
		}
	}
	if (w == 15 && lane == 0) {
		if (run != total) msd_note_error(ctr, 13u); // every key was counted exactly once
		const K mask = (K)nv - 1;
		seg_hi[si] = keys[sg.start] & ~mask;
		tv[ntiles] = (uint16_t)(nv - 1u);
	}
}

template <typename K>
__global__ __launch_bounds__(kBigWriteTh) void bigcount_write_kernel(K *__restrict__ keys,
	const Segment *__restrict__ segs, const uint32_t *__restrict__ first_group, uint32_t nb,
	const uint32_t *__restrict__ ghist, const K *__restrict__ seg_hi, const uint16_t *__restrict__ tile_v)
{
	constexpr int TH = kBigWriteTh, NW = TH / 64, VEC = 16 / (int)sizeof(K);
	constexpr uint32_t TILE = big_tile_elems<K>(), G = kBigGroup;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	K *stage = reinterpret_cast<K *>(smem);                                       // TILE elements + alignment offset
	uint32_t *heavy = reinterpret_cast<uint32_t *>(smem + (size_t)kBigTile * 4 + 16); // (value, begin, count) triples
	uint32_t *misc = heavy + kBigHeavyCap * 3;                                    // [0] number of triples
	uint32_t *pl = misc + 16;                                                      // the group's part of the prefix array
	const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	MSD_STAMP_DECL(4);
	MSD_STAMP_START();
	const uint32_t si = big_find(first_group, nb, blockIdx.x);
	const Segment sg = segs[si];
	const uint32_t nv = 1u << sg.bits, total = (uint32_t)sg.count, ntiles = (total + TILE - 1) / TILE;
	const uint32_t *P = ghist + (size_t)si * 65536; // exclusive prefix per value
	const uint16_t *tv = tile_v + (size_t)first_group[si] * G + si;
	const K hi = seg_hi[si];
	const uint32_t tg = (blockIdx.x - first_group[si]) * G;
	// Every look-up happens before the first store: a load issued behind stores waits for them (one counter for
	// both), and under write load that is the latency of a full write queue -- per tile, 19 us measured.
	uint32_t tvr[G + 1]; // first value of this workgroup's tiles (and of the one behind them)
#pragma unroll
	for (uint32_t g = 0; g <= G; ++g) tvr[g] = tv[min(tg + g, ntiles)];
	uint32_t pe[G];      // end of that value's run
#pragma unroll
	for (uint32_t g = 0; g < G; ++g) pe[g] = tvr[g] + 1 < nv ? P[tvr[g] + 1] : total;
	// prefix entries of the values tvr[0] .. tvr[G] + 1 (P[nv] = total) into LDS, if they fit
	const uint32_t pl0 = tvr[0], npl = tvr[G] + 2 - pl0;
	const bool in_lds = npl <= kBigPrefixCap;
	if (in_lds)
		for (uint32_t j = tid; j < npl; j += TH) pl[j] = pl0 + j < nv ? P[pl0 + j] : total;
	MSD_STAMP(0); // look-ups
#pragma unroll
	for (uint32_t g = 0; g < G; ++g) {
		const uint32_t t = tg + g;
		if (t >= ntiles) break;
		MSD_STAMP_TICK(11);
		const uint32_t off = t * TILE, len = total - off < TILE ? total - off : TILE;
		const uint32_t v_lo = tvr[g], v_hi = tvr[g + 1]; // runs of v_lo .. v_hi overlap the tile (that of v_hi perhaps not)
		K *dst = keys + sg.start + off;
		// the tile on the array's 16-byte grid
		const uint32_t mis = (uint32_t)((sg.start + off) & (uint64_t)(VEC - 1));
		const uint32_t lead = min(len, (uint32_t)(VEC - mis) & (uint32_t)(VEC - 1));
		const uint32_t nvec = (len - lead) / VEC, tail0 = lead + nvec * VEC;
		if (pe[g] >= off + len) { // inside one value's run: straight from registers
			const K kv = hi | (K)v_lo;
			if (tid < lead) dst[tid] = kv;
			if (tail0 + tid < len) dst[tail0 + tid] = kv;
			u32x4 q;
			if constexpr (sizeof(K) == 4) {
				q.x = q.y = q.z = q.w = (uint32_t)kv;
			} else {
				q.x = q.z = (uint32_t)kv;
				q.y = q.w = (uint32_t)((uint64_t)kv >> 32);
			}
			u32x4 *dv = reinterpret_cast<u32x4 *>(dst + lead);
			for (uint32_t i = tid; i < nvec; i += TH) dv[i] = q;
			MSD_STAMP(1); // tile inside one run
			MSD_STAMP_TICK(10);
			continue;
		}
		K *sb = stage + mis; // element i of the tile: same offset inside a 16-byte vector in LDS as in the array
		__syncthreads(); // (the previous tile has left LDS; pl is written)
		if (tid == 0) misc[0] = 0;
		__syncthreads();
		MSD_STAMP(2); // two barriers
		for (uint32_t v = v_lo + tid; v <= v_hi; v += TH) {
			uint32_t pb, pn;
			if (in_lds) {
				pb = pl[v - pl0];
				pn = pl[v - pl0 + 1];
			} else {
				pb = P[v];
				pn = v + 1 < nv ? P[v + 1] : total;
			}
			const uint32_t b = pb > off ? pb : off, e = pn < off + len ? pn : off + len;
			if (e > b) {
				const K kv = hi | (K)v;
				if (e - b <= kBigRun) {
					for (uint32_t i = b; i < e; ++i) sb[i - off] = kv;
				} else {
					const uint32_t at = atomicAdd(&misc[0], 1u);
					heavy[3 * at] = v;
					heavy[3 * at + 1] = b - off;
					heavy[3 * at + 2] = e - b;
				}
			}
		}
		MSD_STAMP(3); // runs of the tile
		__syncthreads();
		const uint32_t nheavy = misc[0];
		if (nheavy >= (uint32_t)NW) {
			for (uint32_t hidx = w; hidx < nheavy; hidx += NW) {
				const K kv = hi | (K)heavy[3 * hidx];
				const uint32_t s0 = heavy[3 * hidx + 1], c = heavy[3 * hidx + 2];
				for (uint32_t i = lane; i < c; i += 64) sb[s0 + i] = kv;
			}
		} else {
			for (uint32_t hidx = 0; hidx < nheavy; ++hidx) {
				const K kv = hi | (K)heavy[3 * hidx];
				const uint32_t s0 = heavy[3 * hidx + 1], c = heavy[3 * hidx + 2];
				for (uint32_t i = tid; i < c; i += TH) sb[s0 + i] = kv;
			}
		}
		MSD_STAMP(4); // barrier + long runs
		__syncthreads();
		MSD_STAMP(5); // barrier
		if (tid < lead) dst[tid] = sb[tid];
		if (tail0 + tid < len) dst[tail0 + tid] = sb[tail0 + tid];
		u32x4 *dv = reinterpret_cast<u32x4 *>(dst + lead);
		const u32x4 *sv = reinterpret_cast<const u32x4 *>(sb + lead);
		for (uint32_t i = tid; i < nvec; i += TH) dv[i] = sv[i];
		MSD_STAMP(6); // LDS -> array
	}
	MSD_STAMP_FLUSH(TH / 64);
}

} // namespace msd
