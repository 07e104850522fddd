// msd_merge16.hpp -- the counting leaf of a rank of the multi-GPU sort (included by msd_device.hpp).
//
// Fine-grained sharding (inplacemsdradixsort_amd/dist.py, csrc/msd_sharded.hip): every rank orders its shard by the
// top 16 key bits BEFORE the exchange (two direct-placement rounds on evenly spread keys -- where they run at full
// speed), so that the one all-to-all delivers, per source rank, that source's 16-bit-prefix buckets of the receiver's
// key range in ascending order.  A bucket of the receiver is then G extents (one per source) of about 2^14 / G keys
// with 16 open bits: this kernel reads the G extents where they arrived, counts all open bits at once and writes the
// sorted bucket to its place in a second buffer -- no gather pass, no run-structured in-place round.  The reference's
// nodes do the same job after their block exchange: every node sorts its whole buckets locally on the remaining bits
// (local_radixsort on the ranges of a node, src/msb_64.c:2200-2255).
//
// The algorithm is count_place16_kernel's (msd_count16.hpp): 2^16 byte counters in LDS, the fetch-add's return value is
// the key's rank among equal keys, place = base[owner of the value] + prefix[value] + rank, keys are re-generated as
// prefix | value.  What differs: the keys of a bucket come from G extents (16-byte loads on each extent's own grid in
// the receive buffer: per extent VPE vectors and SPE scalar tail elements per thread), the output goes to another array,
// and the common prefix is the bucket's number.  A bucket this kernel does not take (an extent or the whole bucket too
// long, a crowded thread, an overflowing byte counter) is copied to its place unsorted and queued for the general leaves.
#pragma once

namespace msd {

template <int G> struct Merge16Cfg {
	static_assert(G == 2 || G == 4 || G == 8, "2, 4 or 8 extents per bucket");
	static constexpr int TH = kC16Th;
	static constexpr int VPE = 4096 / (TH * G) > 0 ? 4096 / (TH * G) : 1; // 16-byte vectors per thread and extent
	static constexpr int SPE = G == 2 ? 2 : 1;                            // scalar tail elements per thread and extent
	static constexpr int RPE = VPE * 4 + SPE;                             // registers per extent
	static constexpr int NK = G * RPE;
	static constexpr uint32_t ECAP = (uint32_t)TH * RPE; // elements of one extent on its 16-byte grid
};
static_assert(kC16Th == 512, "the extent geometry assumes 512-thread workgroups");

// Per source row x and bucket j: where the extent starts in the receive buffer (src_off), how long it is (cnt32), and
// where the bucket starts in the output (dst_off) -- exclusive prefix sums along the buckets, one workgroup per row
// (row nsrc: the totals).  status[0] = 1 if the grand total is not what the host expects or a count does not fit 32 bits
// (the leaf kernel then does nothing).
struct MergeBase {
	uint64_t b[8];
};
__global__ __launch_bounds__(1024) void merge_plan_kernel(const uint64_t *__restrict__ counts, MergeBase base, uint32_t nsrc, uint32_t nb,
	uint64_t n_expected, uint32_t *__restrict__ cnt32, uint64_t *__restrict__ src_off, uint64_t *__restrict__ dst_off,
	uint32_t *__restrict__ status)
{
	__shared__ uint64_t tmp[8];
	__shared__ uint64_t wsum[16];
	const uint32_t row = blockIdx.x, tid = threadIdx.x;
	const uint32_t per = (nb + 1023u) / 1024u, j0 = tid * per, j1 = min(j0 + per, nb);
	auto val = [&](uint32_t j) -> uint64_t {
		if (row < nsrc) return counts[(size_t)row * nb + j];
		uint64_t s = 0;
		for (uint32_t x = 0; x < nsrc; ++x) s += counts[(size_t)x * nb + j];
		return s;
	};
	uint64_t mine = 0;
	bool bad = false;
	for (uint32_t j = j0; j < j1; ++j) {
		const uint64_t v = val(j);
		bad |= v >= 0xFFFFFFFFull;
		mine += v;
	}
	// exclusive scan of `mine` over the 1024 threads
	const uint64_t inc = wave_incl_scan64(mine);
	if ((tid & 63u) == 63u) wsum[tid >> 6] = inc;
	__syncthreads();
	uint64_t pre = inc - mine, total = 0;
	for (uint32_t w = 0; w < 16; ++w) {
		if (w < (tid >> 6)) pre += wsum[w];
		total += wsum[w];
	}
	(void)tmp;
	uint64_t at = pre + (row < nsrc ? base.b[row] : 0ull);
	for (uint32_t j = j0; j < j1; ++j) {
		const uint64_t v = val(j);
		if (row < nsrc) {
			src_off[(size_t)row * nb + j] = at;
			cnt32[(size_t)row * nb + j] = (uint32_t)v;
		} else
			dst_off[j] = at;
		at += v;
	}
	if (bad) atomicOr(status, 1u);
	if (row == nsrc && tid == 0) {
		dst_off[nb] = total; // (nb + 1 entries: a bucket's size is the difference of two neighbours)
		if (total != n_expected) atomicOr(status, 1u);
	}
}

template <int G>
__global__ __launch_bounds__(kC16Th, 4) void merge_place16_kernel(const uint32_t *__restrict__ src, uint64_t src_cap,
	uint32_t *__restrict__ dst, const uint32_t *__restrict__ cnt32, const uint64_t *__restrict__ src_off,
	const uint64_t *__restrict__ dst_off, uint32_t nsrc, uint32_t nb, uint32_t bits, uint32_t prefix0,
	Segment *__restrict__ rejected, Counters *__restrict__ ctr, const uint32_t *__restrict__ status)
{
	using M = Merge16Cfg<G>;
	constexpr int TH = M::TH, VPE = M::VPE, SPE = M::SPE, RPE = M::RPE, NK = M::NK;
	constexpr int CH = 8; // fetch-adds / look-ups in flight per thread
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint32_t *cw = reinterpret_cast<uint32_t *>(smem);  // packed byte counters (padded layout) ...
	uint32_t *out = reinterpret_cast<uint32_t *>(smem); // ... later the output buffer, on the OUTPUT's 16-byte grid
	uint32_t *tbase = cw + kC16Cap;                     // per-thread output base
	uint32_t *junkc = tbase + TH;                       // per-lane junk counter / junk output word
	uint32_t *junko = junkc + 64;
	uint32_t *wtot = junko + 64;                        // 8 wave totals
	uint32_t *nexti = wtot + 9, *crowded = wtot + 11;
	const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	if (*status != 0 || blockIdx.x >= nb) return;
	auto rfl = [](uint32_t x) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane(x); };
	auto rfl64 = [&](uint64_t x) -> uint64_t { return (uint64_t)rfl((uint32_t)x) | ((uint64_t)rfl((uint32_t)(x >> 32)) << 32); };

	// a bucket's extents, uniform: packed o | (o + len) << 8 per extent (o: misalignment of the extent's first
	// element on the 16-byte grid of `src`, o + len <= ECAP < 2^24 when the bucket is taken)
	struct Desc {
		uint32_t ot[G];  // o | tot << 8  (tot = o + len; an extent that is too long: o | 0xFFFFFF00)
		uint64_t d;      // first output element
		uint32_t n;      // keys in the bucket (saturated)
		bool take;
	};
	uint32_t rk[NK];
	// loads of bucket j's keys into rk (branch-free: lanes beyond an extent re-read its last vector / element; an
	// extent that is empty or too long reads the first words of `src`)
	auto load_bucket = [&](uint32_t j) -> Desc {
		Desc ds;
		uint64_t n64 = 0;
		bool take = bits >= kC16MinBits && bits <= 16;
		uint64_t sx[G];
		uint32_t lx[G];
#pragma unroll
		for (int x = 0; x < G; ++x) {
			const bool have = (uint32_t)x < nsrc;
			sx[x] = have ? rfl64(src_off[(size_t)x * nb + j]) : 0ull;
			lx[x] = have ? rfl(cnt32[(size_t)x * nb + j]) : 0u;
			n64 += lx[x];
		}
		ds.d = rfl64(dst_off[j]);
#pragma unroll
		for (int x = 0; x < G; ++x) {
			const uint32_t o = (uint32_t)(sx[x] & 3u);
			const uint64_t tot = (uint64_t)lx[x] + o;
			// (the last vector of an extent may reach beyond the receive buffer's end: such a bucket is not taken)
			take = take && tot <= (uint64_t)M::ECAP && ((sx[x] - o + tot + 3) & ~3ull) <= src_cap;
		}
		take = take && n64 + (ds.d & 3u) <= (uint64_t)kC16Cap;
		ds.take = take;
		ds.n = (uint32_t)(n64 < 0xFFFFFFFFull ? n64 : 0xFFFFFFFFull);
#pragma unroll
		for (int x = 0; x < G; ++x) {
			const uint32_t o = (uint32_t)(sx[x] & 3u);
			const bool rd = take && lx[x] != 0;
			const uint32_t tot = rd ? lx[x] + o : 1u;
			ds.ot[x] = rd ? (o | (tot << 8)) : 0u; // (not read: no element is "in")
			const uint32_t *base = rd ? src + (sx[x] - o) : src;
			const uint32_t lastv = (tot - 1u) >> 2;
#pragma unroll
			for (int v = 0; v < VPE; ++v) {
				const u32x4 q = *reinterpret_cast<const u32x4 *>(base + min((uint32_t)(v * TH) + tid, lastv) * 4u);
				rk[x * RPE + v * 4 + 0] = q.x; rk[x * RPE + v * 4 + 1] = q.y; rk[x * RPE + v * 4 + 2] = q.z; rk[x * RPE + v * 4 + 3] = q.w;
			}
#pragma unroll
			for (int s = 0; s < SPE; ++s) rk[x * RPE + VPE * 4 + s] = base[min((uint32_t)(VPE * TH * 4 + s * TH) + tid, tot - 1u)];
		}
		return ds;
	};
	if (tid == 0) wtot[13] = 0;
	uint32_t cur = blockIdx.x;
	Desc sg = load_bucket(cur);
	for (;;) {
		const uint32_t off = (uint32_t)(sg.d & 3u);
		const uint32_t n = sg.n, tot = sg.take ? n + off : 0u;
		const bool fits = sg.take && n != 0;
		const uint32_t vsh = 16u - (fits ? bits : 16u);
		uint32_t *segb = dst + (sg.d - off); // 16-byte aligned when dst is
		uint32_t zero = 0, tq = tid;
		asm volatile("" : "+v"(zero), "+v"(tq));
#pragma unroll
		for (uint32_t jj = 0; jj < (kC16Cap / 4 + TH - 1) / TH; ++jj) {
			const uint32_t q = jj * TH + tq;
			if (q < kC16Cap / 4) reinterpret_cast<u32x4 *>(cw)[q] = u32x4{ zero, zero, zero, zero };
		}
		if (tid == 0) {
			if (wtot[13] == 0) {
				const uint32_t take = nb > 64u * gridDim.x ? 2u : 1u;
				wtot[12] = atomicAdd(&ctr->count_ticket3, take) + gridDim.x;
				wtot[13] = take;
			}
			*nexti = wtot[12];
			wtot[12] += 1;
			wtot[13] -= 1;
			*crowded = 0;
		}
		__syncthreads();
		if (fits) {
#pragma unroll
			for (int u0 = 0; u0 < NK; u0 += CH) {
				uint32_t old[CH];
#pragma unroll
				for (int i = 0; i < CH; ++i) {
					const int u = u0 + i;
					if (u < NK) {
						const int x = u / RPE, r = u % RPE;
						const uint32_t el = r < VPE * 4 ? (uint32_t)((r / 4) * TH * 4) + tid * 4 + (r % 4) : (uint32_t)(VPE * TH * 4 + (r - VPE * 4) * TH) + tid;
						const uint32_t val = (rk[u] << vsh) & 0xFFFFu;
						const bool in = el >= (sg.ot[x] & 0xFFu) && el < (sg.ot[x] >> 8);
						const uint32_t a = in ? c16_at(val >> 2) : (uint32_t)(junkc - cw) + lane;
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
		__syncthreads();
		const uint32_t nxt = rfl(*nexti);
		const uint32_t hi = (prefix0 + cur) << bits;
		// the thread's 32 counter words are read in two halves, twice (byte sums, then byte prefixes): all 32 in registers
		// beside this kernel's 36-40 key registers spill
		constexpr int WPT = (int)(kC16Words / TH), HW = WPT / 2;
		u32x4 *cq = reinterpret_cast<u32x4 *>(cw + c16_at(tid * (uint32_t)WPT));
		uint32_t totk = 0;
		if (fits) {
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				uint32_t cr[HW];
#pragma unroll
				for (int jj = 0; jj < HW / 4; ++jj) {
					const u32x4 q = cq[h * (HW / 4) + jj];
					cr[4 * jj + 0] = q.x; cr[4 * jj + 1] = q.y; cr[4 * jj + 2] = q.z; cr[4 * jj + 3] = q.w;
				}
#pragma unroll
				for (int jj = 0; jj < HW; ++jj) totk = __builtin_amdgcn_sad_u8(cr[jj], 0u, totk);
			}
		}
		if (totk > 255u) *crowded = 1;
		const uint32_t inc = wave_incl_scan(totk);
		if ((tq & 63u) == 63u) wtot[tq >> 6] = inc;
		__syncthreads();
		uint32_t pos = inc - totk, all = 0;
#pragma unroll
		for (uint32_t ww = 0; ww < TH / 64; ++ww) {
			const uint32_t t = wtot[ww];
			if (ww < w) pos += t;
			all += t;
		}
		const bool ok = fits && all == n && *crowded == 0;
		if (ok) {
			uint32_t run = 0;
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				uint32_t cr[HW];
#pragma unroll
				for (int jj = 0; jj < HW / 4; ++jj) {
					const u32x4 q = cq[h * (HW / 4) + jj];
					cr[4 * jj + 0] = q.x; cr[4 * jj + 1] = q.y; cr[4 * jj + 2] = q.z; cr[4 * jj + 3] = q.w;
				}
#pragma unroll
				for (int jj = 0; jj < HW / 4; ++jj) {
#pragma unroll
					for (int e = 0; e < 4; ++e) {
						const uint32_t x = cr[4 * jj + e], y = x * 0x01010101u;
						cr[4 * jj + e] = (y - x) + run * 0x01010101u;
						run += y >> 24;
					}
					cq[h * (HW / 4) + jj] = u32x4{ cr[4 * jj + 0], cr[4 * jj + 1], cr[4 * jj + 2], cr[4 * jj + 3] };
				}
				__builtin_amdgcn_sched_barrier(0);
			}
			tbase[tid] = pos + off;
			__syncthreads();
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
			__syncthreads(); // counters are dead: the area is the output buffer now
#pragma unroll
			for (int u = 0; u < NK; ++u) {
				uint32_t *o = (rk[u] >> 31) ? junko + lane : out + ((rk[u] >> 16) & 0x7FFFu);
				*o = hi | ((rk[u] & 0xFFFFu) >> vsh);
			}
			__syncthreads();
		} else if (n != 0) {
			// not taken: the bucket goes to its place as it is (extent after extent) and is queued for the general leaves
			uint64_t at = sg.d;
#pragma unroll 1
			for (uint32_t x = 0; x < nsrc; ++x) {
				const uint64_t s0 = rfl64(src_off[(size_t)x * nb + cur]);
				const uint32_t len = rfl(cnt32[(size_t)x * nb + cur]);
				for (uint32_t i = tid; i < len; i += TH) dst[at + i] = src[s0 + i];
				at += len;
			}
			if (tid == 0) {
				Segment r;
				r.start = sg.d;
				r.count = at - sg.d;
				r.bits = bits;
				r.pad = 0;
				rejected[atomicAdd(&ctr->nslow16, 1u)] = r;
			}
		}
		Desc nsg = sg;
		if (nxt < nb) nsg = load_bucket(nxt); // the next bucket's keys travel while this one is stored
		if (ok) {
			const uint32_t v_first = off ? 1u : 0u, v_end = tot >> 2; // full vectors: [v_first, v_end)
			constexpr int NVO = 4096 / TH;
#pragma unroll
			for (int v0 = 0; v0 < NVO; v0 += 4) {
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
			{
				const uint32_t q = (uint32_t)(NVO * TH) + tq;
				if (q < v_end) reinterpret_cast<u32x4 *>(segb)[q] = reinterpret_cast<const u32x4 *>(out)[q];
			}
			if (tq < 4) {
				if (off && tq >= off && tq < tot) segb[tq] = out[tq];
				const uint32_t el = (v_end << 2) + tq;
				if (el < tot && el >= off && (el >= 4u || !off)) segb[el] = out[el];
			}
		}
		if (nxt >= nb) break;
		sg = nsg;
		cur = nxt;
		__syncthreads();
	}
}

// ------------------------------------------------------------------------------------------------------------------
// merge_count_kernel -- the counting leaf for buckets of 2^15 .. 2^19 keys with <= 16 open bits: one 1024-thread
// workgroup per bucket (one per CU), persistent, buckets by ticket.
//
// At 2^30 keys per rank a bucket of the fine-grained exchange holds G x 2^14 keys -- too many for the registers of a
// workgroup (merge_place16_kernel takes the buckets of smaller shards), too few to pay for a 256 KiB histogram in HBM
// (bigcount_*).  Here: 2^16 16-bit counters in LDS (packed in pairs, padded layout) count the bucket's keys as they
// stream in from its extents (16-byte loads, no register copy of the keys, fetch-adds without return value); a scan
// turns the counters in place into 16-bit offsets inside groups of 256 values plus a table of 257 group bases
// (position of value v = base[v >> 8] + offset[v]); the sorted bucket is then re-generated tile by tile, VALUE-parallel:
// for every output tile of 5120 keys the values whose runs touch it -- found by one two-level search per tile, all
// tiles of a bucket at once -- are dealt out to all 1024 threads, every thread writes its values' (short) runs into the
// LDS stage, long runs are filled by the whole wave, and the tile leaves as 16-byte vectors on the output's grid.
// (Round 2's negative result was the value-OWNER walk -- a thread emitting its own 64 values tile after tile, one or two
// waves busy per tile)  A counter that overflows (>= 65536 copies of a key) or a group that holds
// >= 65536 keys makes the sum of the counters or a group offset go wrong, which the scan notices: such a bucket is not
// taken (copied to its place / left where it is, and queued for the general leaves).
// LIST = false: buckets of the multi-GPU exchange (extents from merge_plan_kernel's tables, output to another buffer).
// LIST = true: segments of a list, sorted in place (one extent; all of it is read before the first tile is written) --
// the 17 Ki .. 512 Ki-key segments of skewed inputs that count_walk_kernel used to take.
// inclusive max-scan inside a wave (DPP row shifts + row broadcasts, as wave_incl_scan)
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v)
{
	v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false)); // row_shr:1
	v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false)); // row_shr:2
	v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false)); // row_shr:4
	v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false)); // row_shr:8
	v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false)); // row_bcast:15 into rows 1, 3
	v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false)); // row_bcast:31 into rows 2, 3
	return v;
}

#ifndef MSD_MC_NB // (overridable for experiments)
#define MSD_MC_NB 4
#endif
constexpr int kMcTh = 1024;
constexpr uint32_t kMcWords = 32768;                                 // counter words, two 16-bit counters each
constexpr uint32_t kMcCwWords = kMcWords + (kMcWords >> 6) * 4;      // with 4 words of padding per 64 (c16_at)
constexpr uint32_t kMcSeg = 256;                                     // output positions a wave re-generates at a time
constexpr uint32_t kMcMaxSegs = 2048;                                // buckets of more segments are not taken
constexpr uint64_t kMcMaxKeys = (uint64_t)kMcSeg * (kMcMaxSegs - 1);
constexpr size_t kMcLds = ((size_t)kMcCwWords + 272 + (kMcTh / 64) * kMcSeg + 16 + 4 + 48 + (kMcMaxSegs + 4) / 2) * 4;
static_assert(kMcLds <= 160 * 1024, "one workgroup per CU");

// IN = Hist2 (extent mode, 16 open bits): what arrived is not keys at all but, per source and bucket, the HISTOGRAM of the
// bucket's low halves (hist2_pack_kernel below): 2^16 2-bit counters + the values with three or more copies.  A bucket's
// keys are the sum of its sources' histograms: every thread adds up its 64 values' fields in registers -- no fetch-adds --
// and the scan and the output run as for keys.  17 KiB per source and bucket whatever the bucket holds: at 2^14 keys
// per source and bucket (2^30 keys per rank) a quarter of the whole keys' bytes.
struct Hist2 { unsigned char b; };
constexpr uint32_t kH2Fields = 16384;                    // bytes of 2-bit fields: value v = bits 2 (v & 15) of word v >> 4
constexpr uint32_t kH2MaxExc = 255;                      // values with >= 3 copies a record can name
constexpr uint32_t kH2Rec = kH2Fields + (kH2MaxExc + 1) * 4; // + [number of entries][value << 16 | copies] ...
static_assert(kH2Rec % 16 == 0, "records are read and written as 16-byte vectors");

// IN = uint16_t (extent mode, 16 open bits): the extents hold only the keys' LOW halves -- the upper half of a key is its
// bucket's number, which the receiver of a multi-GPU exchange knows; half the bytes cross the links and half are read here.
template <bool LIST, typename IN = uint32_t>
__global__ __launch_bounds__(kMcTh) void merge_count_kernel(const IN *src, uint32_t *dst,
	const uint32_t *__restrict__ cnt32, const uint64_t *__restrict__ src_off, const uint64_t *__restrict__ dst_off,
	uint32_t nsrc, uint32_t nb, uint32_t bits_arg, uint32_t prefix0,
	const Segment *__restrict__ segs, const uint32_t *__restrict__ nsegs_dev,
	Segment *__restrict__ rejected, uint32_t *__restrict__ nrejected, uint32_t *__restrict__ ticket,
	const uint32_t *__restrict__ status)
{
	constexpr int TH = kMcTh;
	constexpr bool HIST = sizeof(IN) == 1;
	constexpr bool IN16 = sizeof(IN) == 2;
	constexpr uint32_t VE = 16 / sizeof(IN); // elements per 16-byte vector
	static_assert(!LIST || (!IN16 && !HIST), "a list's segments are sorted where they are: whole keys");
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint32_t *cw = reinterpret_cast<uint32_t *>(smem); // 2 x 16-bit counters per word, later offsets inside the value's group
	uint32_t *gbase = cw + kMcCwWords;                 // 257 group bases
	uint32_t *stage = gbase + 272;                     // 256 output positions per wave
	uint32_t *wtot = stage + (kMcTh / 64) * kMcSeg;    // 16 wave totals
	uint32_t *flags = wtot + 16;                       // [0] next ticket, [1] crowded, [2] next output segment
	uint32_t *ext = flags + 4;                         // the bucket's extents: start (2 words), length; up to 16
	uint16_t *seg_va = reinterpret_cast<uint16_t *>(ext + 48); // the value whose run holds every segment's first key (+ one behind the last)
	const uint32_t tid0 = threadIdx.x;
	if (status && *status != 0) return;
	auto rfl = [](uint32_t x) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane(x); };
	auto rfl64 = [&](uint64_t x) -> uint64_t { return (uint64_t)rfl((uint32_t)x) | ((uint64_t)rfl((uint32_t)(x >> 32)) << 32); };
	const uint32_t nbk = LIST ? rfl(*nsegs_dev) : nb;
	const uint32_t nx = LIST ? 1u : nsrc;
	uint32_t cur = blockIdx.x;
	MSD_STAMP_DECL(8);
	MSD_STAMP_START();
	while (cur < nbk) {
		MSD_STAMP(9);
		MSD_STAMP_TICK(11);
		// (the thread index is made opaque per bucket: everything derived from it -- dozens of LDS addresses, masks and
		// predicates -- would otherwise be hoisted out of this loop, kept in registers for its whole life and spilled)
		uint32_t tid = tid0;
		asm volatile("" : "+v"(tid));
		const uint32_t lane = tid & 63u, w = tid >> 6;
		// ---- the bucket (uniform)
		uint32_t bits = bits_arg, hi;
		uint64_t d, n64;
		Segment lsg = {};
		if constexpr (LIST) {
			const Segment g = segs[cur];
			lsg.start = rfl64(g.start);
			lsg.count = rfl64(g.count);
			lsg.bits = rfl(g.bits);
			bits = lsg.bits;
			d = lsg.start;
			n64 = lsg.count;
		} else {
			d = rfl64(dst_off[cur]);
			n64 = rfl64(dst_off[cur + 1]) - d;
		}
		const bool bits_ok = bits >= 1 && bits <= 16 && ((!IN16 && !HIST) || bits == 16);
		const uint32_t mask = bits_ok ? (1u << bits) - 1u : 0u;
		if constexpr (LIST)
			hi = n64 ? rfl(src[d]) & ~mask : 0u; // (read before anything is written: the sort is in place)
		else
			hi = (prefix0 + cur) << (bits_ok ? bits : 0u);
		const uint32_t off = (uint32_t)(d & 3u);
		const bool take = bits_ok && n64 >= 1 && n64 <= kMcMaxKeys;
		const uint32_t n = take ? (uint32_t)n64 : 0u;
		const uint32_t nseg = (n + off + kMcSeg - 1) / kMcSeg;
		// (the extents' descriptors go through LDS: fetched from global memory where they are needed they would make the
		// count loop wait for ALL its key loads in flight -- one counter for every vector-memory operation)
		auto extent = [&](uint32_t x, uint64_t &s, uint32_t &len) {
			if constexpr (LIST) {
				s = d;
				len = n;
			} else {
				s = (uint64_t)rfl(ext[3 * x]) | ((uint64_t)rfl(ext[3 * x + 1]) << 32);
				len = rfl(ext[3 * x + 2]);
			}
		};
		if constexpr (!LIST) {
			if (tid < nx) {
				const uint64_t s = src_off[(size_t)tid * nb + cur];
				ext[3 * tid] = (uint32_t)s;
				ext[3 * tid + 1] = (uint32_t)(s >> 32);
				ext[3 * tid + 2] = cnt32[(size_t)tid * nb + cur];
			}
		}
		// ---- clear the counters (histograms: every counter is written below), take the next ticket
		if constexpr (!HIST)
			for (uint32_t j = tid; j < kMcCwWords / 4; j += TH) reinterpret_cast<u32x4 *>(cw)[j] = u32x4{ 0u, 0u, 0u, 0u };
		if (tid == 0) {
			flags[0] = atomicAdd(ticket, 1u) + gridDim.x;
			flags[1] = 0;
			flags[2] = TH / 64; // the next output segment a wave may take (the first TH / 64 are the waves' own)
		}
		MSD_STAMP(0); // clear + ticket
		__syncthreads();
		MSD_STAMP(1);
		const uint32_t nxt = rfl(flags[0]);
		if (n64 == 0) { // (empty bucket)
			cur = nxt;
			__syncthreads();
			continue;
		}
		bool ok = take;
		// (source x's record of this bucket)
		auto record = [&](uint32_t x) -> const unsigned char * {
			return reinterpret_cast<const unsigned char *>(src) + ((size_t)x * nb + cur) * kH2Rec;
		};
		if constexpr (HIST) {
			// ---- count: thread t owns values [64 t, 64 t + 64) = 16 bytes of every source's fields = 32 counter words
			uint32_t acc[32];
#pragma unroll
			for (int j = 0; j < 32; ++j) acc[j] = 0;
			for (uint32_t x = 0; x < nx; ++x) {
				const u32x4 q = *reinterpret_cast<const u32x4 *>(record(x) + 16u * tid);
				const uint32_t wq[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
				for (int i = 0; i < 4; ++i)
#pragma unroll
					for (int pp = 0; pp < 8; ++pp)
						acc[8 * i + pp] += ((wq[i] >> (4 * pp)) & 3u) | (((wq[i] >> (4 * pp + 2)) & 3u) << 16);
			}
			u32x4 *cq0 = reinterpret_cast<u32x4 *>(cw + c16_at(tid * 32u));
#pragma unroll
			for (int j = 0; j < 8; ++j) cq0[j] = u32x4{ acc[4 * j + 0], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3] };
			__syncthreads();
			// ... and the values with three or more copies: the field says 3, the entry the rest
			for (uint32_t x = 0; x < nx; ++x) {
				const uint32_t *ex = reinterpret_cast<const uint32_t *>(record(x) + kH2Fields);
				const uint32_t ne = min(rfl(ex[0]), kH2MaxExc);
				if (tid < ne) {
					const uint32_t e = ex[1u + tid], v = e >> 16, c = e & 0xFFFFu;
					if (c > 3u) (void)__hip_atomic_fetch_add(&cw[c16_at(v >> 1)], (c - 3u) << ((v & 1u) << 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
			}
		} else if (take) {
			auto count = [&](uint32_t key) {
				const uint32_t v = key & mask;
				(void)__hip_atomic_fetch_add(&cw[c16_at(v >> 1)], 1u << ((v & 1u) << 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			};
			// ---- count: the partial vectors at both ends of every extent element by element ...
			if (tid < VE) {
				for (uint32_t x = 0; x < nx; ++x) {
					uint64_t s;
					uint32_t len;
					extent(x, s, len);
					if (len == 0) continue;
					const uint32_t o = (uint32_t)(s & (VE - 1u)), tot = len + o, vend = tot / VE;
					const IN *base = src + (s - o);
					if (o && tid >= o && tid < tot) count(base[tid]);
					const uint32_t el = vend * VE + tid;
					if (el < tot && (vend > 0 || o == 0)) count(base[el]);
				}
			}
			// ... and the whole vectors in batches of four per thread, the next batch in flight while this one is counted
			constexpr int NB = MSD_MC_NB;
			uint32_t bx = 0, bi = 0, bend = 0; // the batch being loaded: extent bx, vectors bi .. of [.., bend)
			const u32x4 *bbase = nullptr;
			auto open_extent = [&]() { // advance bx to the next extent that has whole vectors
				for (; bx < nx; ++bx) {
					uint64_t s;
					uint32_t len;
					extent(bx, s, len);
					const uint32_t o = (uint32_t)(s & (VE - 1u)), tot = len + o;
					const uint32_t vfirst = o ? 1u : 0u, vend = tot / VE;
					if (len && vend > vfirst) {
						bbase = reinterpret_cast<const u32x4 *>(src + (s - o));
						bi = vfirst;
						bend = vend;
						return;
					}
				}
				bend = 0; // none left
			};
			u32x4 qa[NB], qb[NB];
			uint32_t lim_a = 0, lim_b = 0, i_a = 0, i_b = 0; // (batch: vectors i + u * TH + tid < lim)
			auto load = [&](u32x4 (&q)[NB], uint32_t &i0, uint32_t &lim) {
				i0 = bi;
				lim = bend;
				// (branch-free: registers that are only conditionally written become loop-carried values and are spilled;
				// with no batch left the first vector of `src` is read and ignored)
				const u32x4 *bb = bend ? bbase : reinterpret_cast<const u32x4 *>(src);
#pragma unroll
				for (int u = 0; u < NB; ++u) q[u] = bb[bend ? min(bi + (uint32_t)u * TH + tid, bend - 1u) : 0u];
				if (bend) {
					bi += NB * TH;
					if (bi >= bend) {
						++bx;
						open_extent();
					}
				}
			};
			auto eat = [&](const u32x4 (&q)[NB], uint32_t i0, uint32_t lim) {
#pragma unroll
				for (int u = 0; u < NB; ++u) {
					if (i0 + (uint32_t)u * TH + tid < lim) {
						count(q[u].x);
						count(q[u].y);
						count(q[u].z);
						count(q[u].w);
						if constexpr (IN16) { // (eight low halves per vector; count() masks the value)
							count(q[u].x >> 16);
							count(q[u].y >> 16);
							count(q[u].z >> 16);
							count(q[u].w >> 16);
						}
					}
				}
			};
			open_extent();
			load(qa, i_a, lim_a);
			while (lim_a) {
				load(qb, i_b, lim_b);
				eat(qa, i_a, lim_a);
				if (!lim_b) break;
				load(qa, i_a, lim_a);
				eat(qb, i_b, lim_b);
			}
		}
		MSD_STAMP(2); // count
		__syncthreads();
		MSD_STAMP(3);
		// ---- scan: thread t owns values [64 t, 64 t + 64) = 32 consecutive words
		uint32_t cr[32];
		uint32_t tot = 0;
		u32x4 *cq = reinterpret_cast<u32x4 *>(cw + c16_at(tid * 32u));
		{ // (unconditional -- a bucket that is not taken has clear counters --: conditionally written registers are spilled)
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				const u32x4 q = cq[j];
				cr[4 * j + 0] = q.x; cr[4 * j + 1] = q.y; cr[4 * j + 2] = q.z; cr[4 * j + 3] = q.w;
			}
#pragma unroll
			for (int j = 0; j < 32; ++j) tot += (cr[j] & 0xFFFFu) + (cr[j] >> 16);
		}
		const uint32_t inc = wave_incl_scan(tot);
		if (lane == 63) wtot[w] = inc;
		__syncthreads();
		uint32_t pos = inc - tot, all = 0;
#pragma unroll
		for (uint32_t ww = 0; ww < TH / 64; ++ww) {
			const uint32_t t = wtot[ww];
			if (ww < w) pos += t;
			all += t;
		}
		ok = ok && all == n; // (an overflowing counter loses or misplaces 2^16)
		if (ok) {
			const uint32_t lead = (uint32_t)__shfl((int)pos, (int)(lane & ~3u)); // the group's first thread
			if ((tid & 3u) == 0) gbase[tid >> 2] = pos;
			if (tid == TH - 1) gbase[256] = pos + tot;
			uint32_t r = pos - lead;
#pragma unroll
			for (int j = 0; j < 32; ++j) {
				const uint32_t lo = cr[j] & 0xFFFFu, hh = cr[j] >> 16;
				const uint32_t a = r;
				r += lo;
				cr[j] = a | (r << 16);
				r += hh;
			}
			if (r > 0xFFFFu) flags[1] = 1; // (an offset inside the group would not fit 16 bits)
#pragma unroll
			for (int j = 0; j < 8; ++j) cq[j] = u32x4{ cr[4 * j + 0], cr[4 * j + 1], cr[4 * j + 2], cr[4 * j + 3] };
		}
		__syncthreads();
		ok = ok && flags[1] == 0;
		MSD_STAMP(4); // scan (two barriers inside)
		if (ok) {
			auto half = [&](uint32_t v) -> uint32_t { return (cw[c16_at(v >> 1)] >> ((v & 1u) << 4)) & 0xFFFFu; };
			// ---- the value whose run holds every output segment's first key: largest v with position(v) <= the segment's start
			for (uint32_t i = tid; i <= nseg; i += TH) {
				const uint32_t q0 = i * kMcSeg;
				const uint32_t P = q0 > off ? q0 - off : 0u;
				uint32_t g = 0;
#pragma unroll
				for (uint32_t step = 128; step; step >>= 1)
					if (gbase[g + step] <= P) g += step;
				const uint32_t rel = P - gbase[g];
				uint32_t ii = 0;
#pragma unroll
				for (uint32_t step = 128; step; step >>= 1)
					if (half(g * 256u + ii + step) <= rel) ii += step;
				seg_va[i] = (uint16_t)(g * 256u + ii);
			}
			__syncthreads();
			MSD_STAMP(5); // segment search + barrier
			// ---- every wave re-generates segments of 256 output positions on its own (no workgroup barrier from here on):
			// the values whose runs touch the segment -- two per lane and step -- put (value + 1) at their run's first
			// position in the wave's 256 LDS words (cleared before), an inclusive max-scan over the positions (values
			// ascend with the position: four positions per lane, then across the wave) fills the runs, and every lane
			// stores its four keys as one 16-byte vector on the output's grid
			uint32_t *segb = dst + (d - off);
			uint32_t *wst = stage + (w << 8);
			// (segments by ticket, not round robin: the waves of a workgroup finished 20 thousand cycles apart)
			for (uint32_t sgi = w; sgi < nseg; sgi = rfl(lane == 0 ? atomicAdd(&flags[2], 1u) : 0u)) {
				const uint32_t va = rfl(seg_va[sgi]), vb = min(rfl(seg_va[sgi + 1]), mask); // (no key has a value above the mask)
				const uint32_t q0 = sgi * kMcSeg;
				const uint32_t plo = (q0 > off ? q0 : off) - off, phi = min(q0 + kMcSeg, n + off) - off; // the segment in bucket positions
				reinterpret_cast<u32x4 *>(wst)[lane] = u32x4{ 0u, 0u, 0u, 0u };
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				const uint32_t pbase = off - q0;
				// value v owns [start(v), start(v + 1)): its head goes to the first of its positions inside the segment
				auto head = [&](uint32_t s, uint32_t e, uint32_t v) {
					const uint32_t pos = max(s, plo);
					if (e > pos && s < phi) wst[pos + pbase] = v + 1u;
				};
				if (vb - va < 4u * 64u) {
					// dense buckets (about two keys per value: 128 values per segment): a lane takes four neighbouring values = two
					// counter words (one 8-byte look-up) + the word behind them; one step covers the segment
					const uint32_t ga = va >> 2, gz = vb >> 2;
					for (uint32_t g0 = ga; g0 <= gz; g0 += 64) {
						// (lanes behind the last group repeat it: the same heads to the same places)
						const uint32_t w2 = min(g0 + lane, gz) * 2u, wn = min(w2 + 2u, kMcWords - 1u);
						const uint2 c = *reinterpret_cast<const uint2 *>(cw + c16_at(w2));
						const uint32_t cn = cw[c16_at(wn)], gb = gbase[w2 >> 7], gn = gbase[(w2 + 2u) >> 7];
						const uint32_t s0 = gb + (c.x & 0xFFFFu), s1 = gb + (c.x >> 16), s2 = gb + (c.y & 0xFFFFu), s3 = gb + (c.y >> 16);
						const uint32_t s4 = w2 + 2u < kMcWords ? gn + (cn & 0xFFFFu) : n; // (behind the last value: the bucket's end)
						head(s0, s1, 2u * w2);
						head(s1, s2, 2u * w2 + 1u);
						head(s2, s3, 2u * w2 + 2u);
						head(s3, s4, 2u * w2 + 3u);
					}
				} else {
					// sparse buckets: eight neighbouring values = four counter words per lane (one 16-byte look-up)
					const uint32_t ga = va >> 3, gz = vb >> 3;
					for (uint32_t g0 = ga; g0 <= gz; g0 += 64) {
						const uint32_t w4 = min(g0 + lane, gz) * 4u, wn = min(w4 + 4u, kMcWords - 1u);
						const u32x4 c = *reinterpret_cast<const u32x4 *>(cw + c16_at(w4)); // (4 | 64: the four words are neighbours in the padded layout too)
						const uint32_t cn = cw[c16_at(wn)], gb = gbase[w4 >> 7], gn = gbase[(w4 + 4u) >> 7];
						uint32_t st[9];
						st[0] = gb + (c.x & 0xFFFFu); st[1] = gb + (c.x >> 16);
						st[2] = gb + (c.y & 0xFFFFu); st[3] = gb + (c.y >> 16);
						st[4] = gb + (c.z & 0xFFFFu); st[5] = gb + (c.z >> 16);
						st[6] = gb + (c.w & 0xFFFFu); st[7] = gb + (c.w >> 16);
						st[8] = w4 + 4u < kMcWords ? gn + (cn & 0xFFFFu) : n;
#pragma unroll
						for (int j = 0; j < 8; ++j) head(st[j], st[j + 1], 2u * w4 + (uint32_t)j);
					}
				}
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				const u32x4 hv = reinterpret_cast<const u32x4 *>(wst)[lane];
				const uint32_t m1 = max(hv.x, hv.y), m2 = max(m1, hv.z), m3 = max(m2, hv.w);
				const uint32_t incl = wave_incl_max(m3);
				uint32_t carry = (uint32_t)__shfl_up((int)incl, 1);
				if (lane == 0) carry = 0;
				// (positions outside the bucket -- before `off` in its first vector, behind its end in the last -- hold no run: masked below)
				const u32x4 kv = { hi | (max(carry, hv.x) - 1u), hi | (max(carry, m1) - 1u), hi | (max(carry, m2) - 1u), hi | (max(carry, m3) - 1u) };
				const uint32_t e0 = q0 + 4u * lane;
				if (e0 >= off && e0 + 4u <= n + off)
					*reinterpret_cast<u32x4 *>(segb + e0) = kv;
				else {
					if (e0 + 0u >= off && e0 + 0u < n + off) segb[e0 + 0u] = kv.x;
					if (e0 + 1u >= off && e0 + 1u < n + off) segb[e0 + 1u] = kv.y;
					if (e0 + 2u >= off && e0 + 2u < n + off) segb[e0 + 2u] = kv.z;
					if (e0 + 3u >= off && e0 + 3u < n + off) segb[e0 + 3u] = kv.w;
				}
			}
			MSD_STAMP(6); // output
		} else {
			// not taken: the bucket goes to its place as it is (extent after extent; a list's segment stays where it is)
			// and is queued for the general leaves
			uint64_t at = d;
			if constexpr (HIST) {
				// every source's histogram is written out as that source's keys, in order (the general leaves merge them)
				for (uint32_t x = 0; x < nx; ++x) {
					const u32x4 q = *reinterpret_cast<const u32x4 *>(record(x) + 16u * tid);
					const uint32_t wq[4] = { q.x, q.y, q.z, q.w };
					const uint32_t *ex = reinterpret_cast<const uint32_t *>(record(x) + kH2Fields);
					const uint32_t ne = min(rfl(ex[0]), kH2MaxExc);
					auto copies = [&](uint32_t f) -> uint32_t { // of value 64 tid + f
						uint32_t c = (wq[f >> 4] >> (2u * (f & 15u))) & 3u;
						if (c == 3u)
							for (uint32_t k = 0; k < ne; ++k)
								if ((ex[1u + k] >> 16) == tid * 64u + f) c = ex[1u + k] & 0xFFFFu;
						return c;
					};
					uint32_t mine = 0;
					for (uint32_t f = 0; f < 64; ++f) mine += copies(f);
					const uint32_t inc2 = wave_incl_scan(mine);
					__syncthreads();
					if (lane == 63) wtot[w] = inc2;
					__syncthreads();
					uint32_t o = inc2 - mine, all2 = 0;
					for (uint32_t ww = 0; ww < TH / 64; ++ww) {
						const uint32_t t = wtot[ww];
						if (ww < w) o += t;
						all2 += t;
					}
					for (uint32_t f = 0; f < 64; ++f)
						for (uint32_t c = copies(f); c; --c) dst[at + o++] = hi | (tid * 64u + f);
					at += all2;
				}
			} else if constexpr (!LIST) {
				for (uint32_t x = 0; x < nx; ++x) {
					uint64_t s;
					uint32_t len;
					extent(x, s, len);
					for (uint32_t i = tid; i < len; i += TH) dst[at + i] = IN16 ? (hi | (uint32_t)src[s + i]) : (uint32_t)src[s + i];
					at += len;
				}
			}
			if (tid == 0) {
				Segment r;
				r.start = d;
				r.count = LIST ? n64 : at - d;
				r.bits = bits;
				r.pad = 0;
				rejected[atomicAdd(nrejected, 1u)] = r;
			}
		}
		cur = nxt;
		__syncthreads();
	}
	MSD_STAMP_FLUSH(TH / 64);
}

// the low halves of n u32 keys, in order (what a rank sends once its shard is ordered by the keys' upper halves)
__global__ __launch_bounds__(256) void pack_low16_kernel(const uint32_t *__restrict__ keys, uint64_t n, uint16_t *__restrict__ out)
{
	const uint64_t nv = n >> 3, step = (uint64_t)gridDim.x * 256;
	for (uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x; v < nv; v += step) {
		const u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(keys) + 2 * v);
		const u32x4 b = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(keys) + 2 * v + 1);
		const u32x4 o = { (a.x & 0xFFFFu) | (a.y << 16), (a.z & 0xFFFFu) | (a.w << 16), (b.x & 0xFFFFu) | (b.y << 16), (b.z & 0xFFFFu) | (b.w << 16) };
		__builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(out) + v);
	}
	if (blockIdx.x == 0 && threadIdx.x < (n & 7u)) out[(nv << 3) + threadIdx.x] = (uint16_t)keys[(nv << 3) + threadIdx.x];
}

// hist2_pack_kernel -- what a rank sends of a bucket in the "histogram" form of the fine exchange: the bucket's keys
// (contiguous: the shard is ordered by its upper halves; bounds[b] .. bounds[b + 1]) are counted on their low halves in
// 2^16 byte counters in LDS -- 72 KiB with their padding, so that two workgroups share a CU and one's key loads run under
// the other's LDS phases --, every thread packs its 128 counters into 2-bit fields (0, 1, 2, "3 or more") and the values
// with three or more copies are listed behind them.  A bucket of more than 65535 keys, a value with more than 255 copies
// (its byte counter has spilled into the neighbour: the bytes no longer add up to the bucket's keys) or more than 255 listed values
// set *overflow: the caller then sends the low halves themselves (every rank learns the flag with the counts).
// (first version: 2^16 16-bit counters, one 1024-thread workgroup per CU: 2.2 ms per 2^30 keys; this one: see DESIGN.md)
constexpr int kH2Th = 512;
constexpr uint32_t kH2Words = 16384 + (16384 >> 5) * 4;   // byte counters, four per word, 4 words of padding per 32
__device__ __forceinline__ uint32_t h2_at(uint32_t w) { return w + ((w >> 5) << 2); }
constexpr size_t kH2Lds = ((size_t)kH2Words + (kH2MaxExc + 1) + 8) * 4;
static_assert(2 * kH2Lds <= 160 * 1024, "two workgroups per CU");
// IN = uint16_t: the buckets' LOW HALVES, as msd_order_low16_u32 leaves them (half the bytes to read).
template <typename IN>
__global__ __launch_bounds__(kH2Th, 2) void hist2_pack_kernel(const IN *__restrict__ keys, const uint64_t *__restrict__ bounds, uint32_t nb,
	unsigned char *__restrict__ rec, uint32_t *__restrict__ overflow)
{
	constexpr int TH = kH2Th;
	constexpr uint32_t VE = 16 / sizeof(IN); // values per 16-byte vector
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint32_t *cw = reinterpret_cast<uint32_t *>(smem); // byte counters (h2_at)
	uint32_t *exc = cw + kH2Words;                     // [0] entries, then the entries
	uint32_t *bad = exc + (kH2MaxExc + 1);             // a byte counter has overflowed
	const uint32_t tid0 = threadIdx.x;
	auto rfl = [](uint32_t x) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane(x); };
	constexpr int U = sizeof(IN) == 2 ? 4 : 8; // 16-byte vectors in flight per thread: all of a 2^14-key bucket at once
	// a bucket on the array's 16-byte grid: single elements at both ends, whole vectors [vfirst, vend) between them
	struct Where {
		const IN *base;
		uint32_t n, o, tot, vfirst, vend;
		bool fits;
	};
	auto where = [&](uint32_t bb) -> Where {
		const uint64_t s0 = bounds[bb], s1 = bounds[bb + 1];
		const uint64_t s = (uint64_t)rfl((uint32_t)s0) | ((uint64_t)rfl((uint32_t)(s0 >> 32)) << 32);
		const uint64_t e = (uint64_t)rfl((uint32_t)s1) | ((uint64_t)rfl((uint32_t)(s1 >> 32)) << 32);
		Where wq;
		wq.fits = e - s <= 65535u;
		wq.n = wq.fits ? (uint32_t)(e - s) : 0u;
		wq.o = (uint32_t)(s & (VE - 1u));
		wq.tot = wq.n + wq.o;
		wq.vend = wq.tot / VE;
		wq.vfirst = wq.o ? 1u : 0u;
		wq.base = keys + (s - wq.o);
		return wq;
	};
	// (branch-free loads: lanes behind the end read the bucket's last vector -- or, for an empty bucket, the array's first -- again)
	auto load_batch = [&](const Where &wq, uint32_t v0, uint32_t tq, u32x4 (&q)[U]) {
		const u32x4 *bp = wq.vend > 0 ? reinterpret_cast<const u32x4 *>(wq.base) : reinterpret_cast<const u32x4 *>(keys);
		const uint32_t last = wq.vend > 0 ? wq.vend - 1u : 0u;
#pragma unroll
		for (int u = 0; u < U; ++u) q[u] = __builtin_nontemporal_load(bp + min(v0 + (uint32_t)u * TH + tq, last));
	};
	u32x4 q[U]; // the first batch of the bucket at hand: requested while the bucket before it is packed
	if (blockIdx.x < nb) load_batch(where(blockIdx.x), where(blockIdx.x).vfirst, tid0, q);
	MSD_STAMP_DECL(11);
	MSD_STAMP_START();
	for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
		MSD_STAMP(9);
		MSD_STAMP_TICK(11);
		uint32_t tid = tid0;
		asm volatile("" : "+v"(tid));
		const Where wb = where(b);
		const bool fits = wb.fits;
		const uint32_t n = wb.n;
		for (uint32_t j = tid; j < kH2Words / 4; j += TH) reinterpret_cast<u32x4 *>(cw)[j] = u32x4{ 0u, 0u, 0u, 0u };
		if (tid < (kH2MaxExc + 1) / 4) reinterpret_cast<u32x4 *>(exc)[tid] = u32x4{ 0u, 0u, 0u, 0u };
		if (tid == 0) {
			bad[0] = fits ? 0u : 1u;
			bad[1] = 0;
		}
		MSD_STAMP(0); // clear
		__syncthreads();
		MSD_STAMP(1); // barrier
#ifdef MSD_STAMPS
		if constexpr (kStampThis) {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			MSD_STAMP(2); // wait for the bucket's keys
		}
#endif
		// (fetch-adds without return value; a counter that has spilled into its neighbour shows in the sum of all bytes below)
		auto count = [&](uint32_t key) {
			const uint32_t v = key & 0xFFFFu;
			(void)__hip_atomic_fetch_add(&cw[h2_at(v >> 2)], 1u << ((v & 3u) << 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		};
		auto eat = [&](const u32x4 (&qq)[U], uint32_t v0) {
#pragma unroll
			for (int u = 0; u < U; ++u) {
				if (v0 + (uint32_t)u * TH + tid < wb.vend) {
					count(qq[u].x);
					count(qq[u].y);
					count(qq[u].z);
					count(qq[u].w);
					if constexpr (sizeof(IN) == 2) { // (eight low halves per vector; count() masks the value)
						count(qq[u].x >> 16);
						count(qq[u].y >> 16);
						count(qq[u].z >> 16);
						count(qq[u].w >> 16);
					}
				}
			}
		};
		if (n && tid < VE) {
			if (wb.o && tid >= wb.o && tid < wb.tot) count(wb.base[tid]);
			const uint32_t el = wb.vend * VE + tid;
			if (el < wb.tot && (wb.vend > 0 || wb.o == 0)) count(wb.base[el]);
		}
		eat(q, wb.vfirst);
		for (uint32_t v0 = wb.vfirst + U * TH; v0 < wb.vend; v0 += U * TH) { // (buckets of more than 2^14 keys)
			u32x4 q2[U];
			load_batch(wb, v0, tid, q2);
			eat(q2, v0);
		}
		MSD_STAMP(3); // fetch-adds
		{ // the next bucket's first batch is on its way while this one is packed (past the last bucket: the last one again, unused)
			const Where wn = where(min(b + gridDim.x, nb - 1u));
			load_batch(wn, wn.vfirst, tid, q);
		}
		MSD_STAMP(4); // next loads issued
		__syncthreads();
		MSD_STAMP(5); // barrier
		// ---- thread t packs values [128 t, 128 t + 128) = 32 counter words into 32 bytes of fields
		const u32x4 *cq = reinterpret_cast<const u32x4 *>(cw + h2_at(tid * 32u));
		uint32_t out[8], bsum = 0, many = 0; // many: bit = a word of mine that holds a value with three or more copies
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const u32x4 qc = cq[j];
			const uint32_t wq[4] = { qc.x, qc.y, qc.z, qc.w };
			uint32_t o4 = 0;
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				// per byte min(count, 3): the four bytes as two pairs of 16-bit lanes, one packed minimum each (the first
				// version clamped bytes with seven SWAR operations: the packing loop was VALU-bound, 7-11 of a bucket's 20
				// thousand cycles)
				typedef unsigned short us2 __attribute__((ext_vector_type(2)));
				const uint32_t w = wq[i];
				bsum = __builtin_amdgcn_sad_u8(w, 0u, bsum); // + the four bytes
				const uint32_t e02 = w & 0x00FF00FFu, e13 = (w >> 8) & 0x00FF00FFu; // counts of values 0, 2 | 1, 3 of the word
				const us2 three = { 3, 3 };
				const us2 m02 = __builtin_elementwise_min(__builtin_bit_cast(us2, e02), three), m13 = __builtin_elementwise_min(__builtin_bit_cast(us2, e13), three);
				const uint32_t t = __builtin_bit_cast(uint32_t, m02) | (__builtin_bit_cast(uint32_t, m13) << 2); // bits 0..3: values 0, 1; bits 16..19: values 2, 3
				o4 |= ((t & 0xFu) | ((t >> 12) & 0xF0u)) << (8 * i);
				many |= (t & (t >> 1) & 0x00050005u) != 0u ? 1u << (4 * j + i) : 0u;
			}
			out[j] = o4;
		}
		// (one value in five hundred has three or more copies: listed behind the loop -- inside it, some lane of nearly every
		// wave took the branch at nearly every word)
		while (many) {
			const uint32_t wi = (uint32_t)__builtin_ctz(many);
			many &= many - 1u;
			const uint32_t w = cw[h2_at(tid * 32u + wi)];
			for (uint32_t k2 = 0; k2 < 4; ++k2) {
				const uint32_t c = (w >> (8 * k2)) & 0xFFu;
				if (c >= 3u) {
					const uint32_t k = atomicAdd(&exc[0], 1u);
					if (k < kH2MaxExc) exc[1u + k] = ((tid * 128u + 4u * wi + k2) << 16) | c;
				}
			}
		}
		{ // the bytes must add up to the bucket's keys: a counter that passed 255 has carried into its neighbour (or out of its word)
			uint32_t t = bsum;
#pragma unroll
			for (int o2 = 32; o2 > 0; o2 >>= 1) t += (uint32_t)__shfl_xor((int)t, o2);
			if ((tid & 63u) == 0) atomicAdd(&bad[1], t);
		}
		unsigned char *r = rec + (size_t)b * kH2Rec;
		__builtin_nontemporal_store(u32x4{ out[0], out[1], out[2], out[3] }, reinterpret_cast<u32x4 *>(r) + 2 * tid);
		__builtin_nontemporal_store(u32x4{ out[4], out[5], out[6], out[7] }, reinterpret_cast<u32x4 *>(r) + 2 * tid + 1);
		MSD_STAMP(6); // pack + entries + fields out
		__syncthreads();
		MSD_STAMP(7); // barrier
		if (tid < (kH2MaxExc + 1) / 4) {
			const u32x4 q = reinterpret_cast<const u32x4 *>(exc)[tid];
			if (tid == 0 && (q.x > kH2MaxExc || bad[0] != 0 || bad[1] != n)) atomicOr(overflow, 1u);
			reinterpret_cast<u32x4 *>(r + kH2Fields)[tid] = q;
		}
		__syncthreads();
		MSD_STAMP(8); // entries out + barrier
	}
	MSD_STAMP_FLUSH(TH / 64);
}

// first index i with (keys[i] >> shift) >= first + b, b = 0 .. nbuckets: the boundaries of the buckets of an array that
// is ordered by key >> shift (the reference knows its ranges' boundaries from its histograms, src/msb_64.c:1546-1564;
// here a binary search per boundary: 2^16 searches of about 30 dependent look-ups)
template <typename K>
__global__ __launch_bounds__(256) void bucket_bounds_kernel(const K *__restrict__ keys, uint64_t n, uint32_t shift, uint64_t first,
	uint32_t nbuckets, uint64_t *__restrict__ bounds)
{
	const uint32_t b = blockIdx.x * 256 + threadIdx.x;
	if (b > nbuckets) return;
	const uint64_t want = first + b; // (may be 2^(bits - shift): nothing reaches it)
	uint64_t lo = 0, hi = n;
	while (lo < hi) {
		const uint64_t mid = lo + ((hi - lo) >> 1);
		if ((uint64_t)(keys[mid] >> shift) < want) lo = mid + 1; else hi = mid;
	}
	bounds[b] = lo;
}

} // namespace msd
