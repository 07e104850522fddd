// msd_device.hpp -- gfx950 kernels of the in-place MSD radix sort.
//
// One "round" partitions every big segment ("parent") in place by one digit of
// up to 8 bits, in three phases that mirror the reference's parallel block
// machinery (src/msb_64.c:497-699 range_partition_to_blocks, :2022-2093 block
// cycle-leader swap with fetch-add claims, :1220-1302 combine/inject) rather
// than its single sequential cycle (partition_ip_buf, :785-978):
//
//   A  classify_kernel   each workgroup streams its stripe through LDS, appends
//                        keys to 256 per-bucket LDS buffers (rank = LDS fetch-add)
//                        and flushes every full buffer as one aligned 256-byte
//                        block BEHIND its own read cursor; the digit histogram
//                        falls out of this pass (no separate counting read).
//   B  chains_kernel     block-granular permutation: a lane owns a hole, claims
//                        the next misplaced block of the hole's bucket with a
//                        fetch-add on that bucket's list cursor, moves it into
//                        the hole and inherits the vacated slot.
//   C  cleanup_kernel    bucket heads/tails are filled from the stripes'
//                        partial buffers.
//   A' classify_direct2_kernel, msd_direct.hpp (rounds whose buckets are about equally big and free of
//                        runs): bucket boundaries are known before the pass (sampled /
//                        exactly counted), every workgroup reads its own share of each
//                        bucket's region and writes completed blocks straight into it,
//                        so that B only moves the few misplaced blocks.
// Leaves: count_place_kernel / count_walk_kernel (keys only, <= 16 open bits: one counting
// pass over all of them), bigcount_* (the same for segments of any size, msd_bigcount.hpp),
// leaf_count_sort_kernel (everything else that fits LDS: one counting pass over the top
// varying bits + group fix-up), lds_sort_kernel (general fallback: stable LSD passes
// inside LDS; ranks from wavefront ballot/popcount match-any).
//
// Everything here is integer/byte work bound by HBM; there is no MFMA use.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace msd {

constexpr int kP = 256;              // buckets per digit pass (8-bit digits)
constexpr uint32_t kXBase = 0x80000000u; // slot ids >= kXBase live in the side block store
constexpr uint32_t kNoOwner = 0xFFFFFFFFu;
constexpr uint32_t kRposStride = 32;    // claim cursors sit on separate 128-byte lines

struct NoVal {};
typedef uint32_t u32x4 __attribute__((ext_vector_type(4))); // one 16-byte register quad

template <typename V> struct has_val { static constexpr bool value = true; };
template <> struct has_val<NoVal> { static constexpr bool value = false; };

// Geometry per (key, payload) type.  B = elements per block (256-byte key
// blocks), T = elements per classify tile, TH = classify threads.
template <typename K, typename V> struct Cfg;
template <> struct Cfg<uint32_t, NoVal> {
	static constexpr int B = 64, T = 4096, TH = 1024;
	static constexpr int SORT_TH = 1024, SORT_KPT = 24; // LDS sort capacity 24576
};
template <> struct Cfg<uint64_t, NoVal> {
	static constexpr int B = 32, T = 4096, TH = 1024;
	// 17408 u64 keys (139 KiB of the 160 KiB LDS): the 2^14-key children of two 8-bit rounds over 2^30 keys
	// are leaves, no third round
	static constexpr int SORT_TH = 1024, SORT_KPT = 17;
};
template <> struct Cfg<uint64_t, uint64_t> {
	static constexpr int B = 32, T = 2048, TH = 1024;
#ifndef MSD_PAIR_LEAF_TH // (overridable for experiments)
#define MSD_PAIR_LEAF_TH 512
#endif
	static constexpr int SORT_TH = MSD_PAIR_LEAF_TH, SORT_KPT = 6; // 3072 tuples: two leaf workgroups per CU (1024 threads / 6144 tuples: one, 9.2 vs 8.0 ms at 2^30)
};

struct Parent {
	uint64_t start, count;
	uint32_t shift;      // digit = (key >> shift) & ((1 << width) - 1)
	uint32_t width;      // 1..8
	uint32_t child_base; // global child index of digit 0
	uint32_t stripe_lo, stripe_hi;
	uint32_t pad;        // range partitioning: number of delimiters, else 0
};

struct Stripe {
	uint64_t begin, end; // element range; begin is B-aligned except for a parent's first stripe
	uint64_t lo_base;    // element offset of this stripe's leftover area
	uint32_t parent;
	uint32_t slot_lo, slot_hi; // aligned block slots fully inside [begin, end)
	uint32_t pad;
};

struct Segment { // a finished-partition child that still needs sorting
	uint64_t start, count;
	uint32_t bits; // low bits still unsorted
	uint32_t pad;
};

struct ListEntry {
	uint32_t slot;  // where the misplaced block sits (>= kXBase: side store)
	uint32_t owner; // global child whose interior contains `slot`, kNoOwner for fringe
};

struct Counters { // one per sort call, zeroed per round where noted
	uint32_t nholes;       // per round
	uint32_t hole_cursor;  // per round
	uint32_t next_parents; // per round
	uint32_t nsmall;       // cumulative
	uint32_t errors;       // cumulative: internal invariant violations
	uint32_t chain_steps;  // cumulative (stat)
	uint32_t ncount;       // cumulative: segments for the one-pass counting sort (<= 16 bits left)
	uint32_t nfallback;    // counting-sort segments handed to the general LDS sort (byte counter overflow)
	uint32_t nbig;         // cumulative: segments of any size with <= 16 bits left (multi-workgroup counting sort)
	uint32_t count_ticket; // work ticket of the persistent counting-sort workgroups
	uint32_t direct_uneven; // per round: parents whose children are too unequal for direct placement
	uint32_t nslow;         // counting-sort segments the fast kernel left to count_walk_kernel
	uint32_t count_ticket2; // work ticket of count_walk_kernel
	uint32_t nevict;        // per round: side-store blocks handed out for evictions
	uint32_t leaf_ticket[2]; // work tickets of the two leaf_count_sort launches
	uint32_t count_ticket3;  // work ticket of count_place16_kernel
	uint32_t nslow16;        // segments count_place16_kernel left to count_place_kernel
	uint32_t next_max;       // per round: largest next parent (saturated to 32 bits)
	uint32_t rp_children;    // per round: children handed out by regpart_plan_kernel
	uint32_t nhot;           // per round: lists with sharded claim cursors (chains_kernel)
	uint32_t nslow2;         // counting-leaf segments merge_count_kernel (list mode) left to count_walk_kernel
	uint32_t count_ticket4;  // work ticket of merge_count_kernel (list mode)
	uint32_t l17_slow;       // leaf17_kernel: segments whose groups took the position-by-position fix-up
	uint32_t err_sites;      // cumulative: which checks raised `errors` (bit = site, msd_note_error)
	uint32_t pad_;
};
// an internal invariant does not hold: counted, and the place remembered for the error message
__device__ __forceinline__ void msd_note_error(Counters *ctr, uint32_t site)
{
	atomicAdd(&ctr->errors, 1u);
	atomicOr(&ctr->err_sites, 1u << site);
}
static_assert(sizeof(Counters) % 8 == 0, "the words behind the counters are used for 64-bit atomics");

// ---------------------------------------------------------------- diagnostics
// -DMSD_STAMPS (tools/stamps_build.sh, never the shipped library): wave 0 (a "bucket wave") and the last wave of
// every classify_direct workgroup add the shader cycles they spend in each section of the tile loop to
// g_stamps[which wave][section]; read back with msd_debug_stamps().
#ifdef MSD_STAMPS // = 1: classify_direct kernels, 2: count_place kernels, 3: leaf_count_sort_kernel, 4/5: bigcount write/hist, 6: classify_kernel
__device__ unsigned long long g_stamps[2][16];
#define MSD_STAMP_DECL(id)                          \
	constexpr bool kStampThis = MSD_STAMPS == (id); \
	unsigned long long st_acc[12] = {}, st_last = 0
#define MSD_STAMP_START()                                                                              \
	do {                                                                                            \
		if constexpr (kStampThis) {                                                             \
			__builtin_amdgcn_sched_barrier(0);                                              \
			asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory"); \
			__builtin_amdgcn_sched_barrier(0);                                              \
		}                                                                                       \
	} while (0)
#define MSD_STAMP(i)                                                                                    \
	do {                                                                                            \
		if constexpr (kStampThis) {                                                             \
			unsigned long long now_;                                                        \
			__builtin_amdgcn_sched_barrier(0);                                              \
			asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");    \
			__builtin_amdgcn_sched_barrier(0);                                              \
			st_acc[i] += now_ - st_last;                                                    \
			st_last = now_;                                                                 \
		}                                                                                       \
	} while (0)
#define MSD_STAMP_TICK(i)                            \
	do {                                         \
		if constexpr (kStampThis) st_acc[i] += 1; \
	} while (0)
#define MSD_STAMP_FLUSH(nwaves)                                                                         \
	do {                                                                                            \
		if constexpr (kStampThis) {                                                             \
			const unsigned wv_ = threadIdx.x >> 6;                                          \
			if ((threadIdx.x & 63) == 0 && (wv_ == 0 || wv_ == (nwaves) - 1)) {            \
				for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&g_stamps[wv_ ? 1 : 0][i_], st_acc[i_]); \
			}                                                                               \
		}                                                                                       \
	} while (0)
#else
#define MSD_STAMP_DECL(id)
#define MSD_STAMP_START() do {} while (0)
#define MSD_STAMP(i) do {} while (0)
#define MSD_STAMP_TICK(i) do {} while (0)
#define MSD_STAMP_FLUSH(nwaves) do {} while (0)
#endif

// ---------------------------------------------------------------- utilities

template <typename K> __device__ __forceinline__ uint32_t digit_of(K key, uint32_t shift, uint32_t mask)
{
	return (uint32_t)(key >> shift) & mask;
}

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63; }

// inclusive scan inside a wave
// (DPP row shifts + row broadcasts: no lane-index registers, unlike a bpermute-based shuffle whose
// six source-lane addresses the compiler keeps -- and spills -- as loop invariants)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); // row_shr:4
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); // row_shr:8
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1, 3
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2, 3
	return v;
}
__device__ __forceinline__ uint64_t wave_incl_scan64(uint64_t v)
{
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		uint64_t t = __shfl_up(v, o);
		if ((int)lane_id() >= o) v += t;
	}
	return v;
}

// Exclusive scan over the first 256 threads' values; every thread of the block
// must call it (it contains barriers).  tmp: >= 5 uint32 of LDS.  Returns the
// exclusive prefix for threads < 256 and the grand total in `total`.
// popcount of the bits of `m` below this lane / this lane's bit of `m`, without a lane mask
// (a hoisted 64-bit (1 << lane) - 1 costs two registers for the whole kernel, and a spill that is
// reloaded behind an outstanding prefetch stalls on vmcnt)
__device__ __forceinline__ uint32_t popc_below_lane(uint64_t m)
{
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ bool lane_bit(uint64_t m)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t half = lane < 32u ? (uint32_t)m : (uint32_t)(m >> 32);
	return (half >> (lane & 31u)) & 1u;
}

__device__ __forceinline__ uint32_t block_excl_scan256(uint32_t v, uint32_t *tmp, uint32_t &total)
{
	const uint32_t tid = threadIdx.x;
	if (tid >= 256) v = 0;
	uint32_t inc = wave_incl_scan(v);
	if (tid < 256 && lane_id() == 63) tmp[tid >> 6] = inc;
	__syncthreads();
	uint32_t base = 0;
	if (tid < 256) {
		const uint32_t w = tid >> 6;
		if (w > 0) base += tmp[0];
		if (w > 1) base += tmp[1];
		if (w > 2) base += tmp[2];
	}
	total = tmp[0] + tmp[1] + tmp[2] + tmp[3];
	__syncthreads();
	return base + inc - v;
}
__device__ __forceinline__ uint64_t block_excl_scan256_64(uint64_t v, uint64_t *tmp, uint64_t &total)
{
	const uint32_t tid = threadIdx.x;
	if (tid >= 256) v = 0;
	uint64_t inc = wave_incl_scan64(v);
	if (tid < 256 && lane_id() == 63) tmp[tid >> 6] = inc;
	__syncthreads();
	uint64_t base = 0;
	if (tid < 256) {
		const uint32_t w = tid >> 6;
		if (w > 0) base += tmp[0];
		if (w > 1) base += tmp[1];
		if (w > 2) base += tmp[2];
	}
	total = tmp[0] + tmp[1] + tmp[2] + tmp[3];
	__syncthreads();
	return base + inc - v;
}

// 16-byte vector of K
template <typename K> struct Vec16;
template <> struct Vec16<uint32_t> { static constexpr int N = 4; };
template <> struct Vec16<uint64_t> { static constexpr int N = 2; };

// ------------------------------------------------------------ A: classify

template <typename K, typename V> struct ClassifyLds {
	using C = Cfg<K, V>;
	static constexpr bool HV = has_val<V>::value;
	static constexpr size_t kbuf = (size_t)(kP * C::B) * sizeof(K); // per-bucket partial buffers
	static constexpr size_t vbuf = HV ? (size_t)(kP * C::B) * sizeof(uint64_t) : 0;
	static constexpr size_t head = (size_t)C::B * sizeof(K) + (HV ? (size_t)C::B * sizeof(uint64_t) : 0);
	static constexpr int JOBS = kP + 8;
	// meta, cnt, hc, loff : 4*kP u32 ; jobs ; tmp 16
	static constexpr size_t small = (size_t)(4 * kP + JOBS + 16 + 64 + 64) * sizeof(uint32_t);
	static constexpr size_t bytes = kbuf + vbuf + head + small;
};

// RANGE: the bucket of a key is not a digit but its range among `splitters` (2^width - 1 ascending delimiters,
// padded with the largest key): bucket p = number of delimiters < key, i.e. range p holds the keys in
// (delim[p-1], delim[p]] -- the reference's lower-bound range function (binary_search_64, src/msb_64.c:188-204;
// SIMD form :239-351).  Everything behind the classification (blocks, maps, permutation) is the same.
template <typename K, typename V, bool RANGE = false>
__global__ __launch_bounds__((Cfg<K, V>::TH), (Cfg<K, V>::TH >= 1024 ? (has_val<V>::value ? 4 : 8) : 1)) void classify_kernel(
	K *__restrict__ keys, uint64_t *__restrict__ vals, const Stripe *__restrict__ stripes,
	const Parent *__restrict__ parents, uint8_t *__restrict__ block_map,
	uint32_t *__restrict__ fb, uint32_t *__restrict__ lo_cnt, uint32_t *__restrict__ lo_off,
	K *__restrict__ lo_keys, uint64_t *__restrict__ lo_vals, uint32_t *__restrict__ nfull,
	const K *__restrict__ splitters = nullptr,
	// launched behind a direct-placement attempt: runs only if that declined (Counters::direct_uneven != 0) -- the
	// decision stays on the device, the host does not wait for it
	const uint32_t *__restrict__ run_if_nonzero = nullptr)
{
	if (run_if_nonzero && *run_if_nonzero == 0) return;
	using C = Cfg<K, V>;
	constexpr bool HV = has_val<V>::value;
	constexpr int B = C::B, T = C::T, TH = C::TH;
	constexpr int VEC = Vec16<K>::N;
	constexpr int KPT = T / TH;    // keys per thread per tile
	constexpr int NV = KPT / VEC;  // 16-byte vectors per thread per tile
	constexpr int LPB = B / VEC;   // lanes that move one block
	constexpr int PB = kP * B;
	static_assert(KPT % VEC == 0 && NV >= 1, "tile geometry");
	using L = ClassifyLds<K, V>;

	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	K *kbuf = reinterpret_cast<K *>(smem); // [0,PB) partial buffers
	uint64_t *vbuf = reinterpret_cast<uint64_t *>(smem + L::kbuf);
	K *headk = reinterpret_cast<K *>(smem + L::kbuf + L::vbuf);
	uint64_t *headv = reinterpret_cast<uint64_t *>(smem + L::kbuf + L::vbuf + (size_t)B * sizeof(K));
	uint32_t *meta = reinterpret_cast<uint32_t *>(smem + L::kbuf + L::vbuf + L::head);
	uint32_t *cnt = meta + kP;   // tile count per bucket
	uint32_t *hc = cnt + kP;     // head keys per bucket
	uint32_t *loff = hc + kP;    // leftover offsets
	uint32_t *jobs = loff + kP;  // flush job table: bucket whose buffer goes to slot wslot+g
	uint32_t *tmp = jobs + L::JOBS; // [0..1] slots | jobs << 16 claimed per tile (ping-pong), [4..7] scan scratch, [8..9] skew
	uint32_t *multi = tmp + 16;     // buckets that completed more than one block in this tile: bucket | blocks << 8 | first slot << 16
	uint32_t *spare = multi + 64;   // a word per lane that only ever receives zeros
	K *spl = reinterpret_cast<K *>(smem + L::bytes); // RANGE only: the delimiters (the launch adds kP keys of LDS)

	const uint32_t tid = threadIdx.x;
	const Stripe st = stripes[blockIdx.x];
	const Parent pa = parents[st.parent];
	const uint32_t shift = pa.shift, mask = (1u << pa.width) - 1u;
	if constexpr (RANGE) {
		if (tid < kP) spl[tid] = tid < pa.pad ? splitters[tid] : ~(K)0; // pa.pad: number of delimiters
	}
	// bucket of a key: digit, or number of delimiters below it (branch-free binary search in LDS)
	auto digit_of = [&](K key, uint32_t sh, uint32_t mk) -> uint32_t {
		if constexpr (RANGE) {
			uint32_t p = 0;
			for (uint32_t step = (mk + 1u) >> 1; step; step >>= 1)
				if (spl[p + step - 1u] < key) p += step;
			return p;
		} else
			return msd::digit_of(key, sh, mk);
	};

	if (tid < kP) {
		meta[tid] = 0;
		cnt[tid] = 0;
		hc[tid] = 0;
	}
	if (tid < 16) tmp[tid] = 0;
	if (tid < 64) spare[tid] = 0;
	const uint64_t a0 = (uint64_t)st.slot_lo * B; // first aligned position >= begin
	// ---- head keys (only a parent's first stripe has them): parked in LDS until the end
	const uint32_t h = (uint32_t)((a0 < st.end ? a0 : st.end) - st.begin);
	if (tid < h) {
		headk[tid] = keys[st.begin + tid];
		if (HV) headv[tid] = vals[st.begin + tid];
	}
	__syncthreads();
	if (tid < h) atomicAdd(&hc[digit_of(headk[tid], shift, mask)], 1u);

	uint32_t wslot = st.slot_lo; // next output slot (uniform)
	uint32_t fill_r = 0;         // thread d<kP: fill of bucket d (register copy)
	uint32_t fb_r = 0;

	K kreg[KPT];
	uint64_t vreg[HV ? KPT : 1];
	// remainders of buckets flushed in the previous tile: written during this tile's scatter
	K dkey[KPT];
	uint64_t dval[HV ? KPT : 1];
	uint32_t dat[KPT];
#pragma unroll
	for (int i = 0; i < KPT; ++i) dat[i] = 0xFFFFFFFFu;

	// (uniform 64-bit base + 32-bit per-thread offset: a per-thread 64-bit index would be kept in two
	// registers across the tile loop, spilled, and its reload would wait for the prefetch in flight)
	auto load_tile = [&](uint64_t pos, K *kr, uint64_t *vr) {
		const K *kp = keys + pos;
		const uint64_t *vp = vals + pos;
		const uint32_t rem = pos < st.end ? (uint32_t)(st.end - pos < (uint64_t)T ? st.end - pos : (uint64_t)T) : 0u;
#pragma unroll
		for (int v = 0; v < NV; ++v) {
			const uint32_t off = (uint32_t)(v * TH + tid) * VEC;
			if (off + VEC <= rem) {
				if constexpr (sizeof(K) == 4) {
					const uint4 q = *reinterpret_cast<const uint4 *>(kp + off);
					kr[v * VEC + 0] = q.x; kr[v * VEC + 1] = q.y; kr[v * VEC + 2] = q.z; kr[v * VEC + 3] = q.w;
				} else {
					const ulonglong2 q = *reinterpret_cast<const ulonglong2 *>(kp + off);
					kr[v * VEC + 0] = q.x; kr[v * VEC + 1] = q.y;
				}
				if constexpr (HV) {
					const ulonglong2 q = *reinterpret_cast<const ulonglong2 *>(vp + off);
					vr[v * VEC + 0] = q.x; vr[v * VEC + 1] = q.y;
				}
			} else {
#pragma unroll
				for (int e = 0; e < VEC; ++e) {
					if (off + e < rem) {
						kr[v * VEC + e] = kp[off + e];
						if constexpr (HV) vr[v * VEC + e] = vp[off + e];
					}
				}
			}
		}
	};

	uint64_t pos = a0;
	K kregB[KPT];
	uint64_t vregB[HV ? KPT : 1];
	if (pos < st.end) load_tile(pos, kreg, vreg);
	if (pos + T < st.end) load_tile(pos + T, kregB, vregB);
	uint32_t par = 0; // tile parity: which pair of claim counters is live
	MSD_STAMP_DECL(6);
	MSD_STAMP_START();

	// One tile: [ranks] B1 [per-bucket bookkeeping] B2 [scatter] B3 [prefetch tile t+2 into the
	// registers this tile just vacated] [flush].  Two register sets alternate (no copies), so a load
	// is issued two tiles before its first use and, like the flush stores, a whole tile before the
	// explicit vmcnt(0) (vmcnt counts loads and stores together).  The flush of tile t overlaps the
	// rank phase of tile t+1; nothing it reads is written before B2 of tile t+1.
	auto tile = [&](K (&kc)[KPT], uint64_t (&vc)[HV ? KPT : 1]) {
		const uint64_t npos = pos + T;
		const bool full = npos <= st.end;                  // uniform: every key of the tile exists
		MSD_STAMP(9);
		MSD_STAMP_TICK(11);

		// ---- rank every key inside its bucket for this tile (LDS fetch-add)
		uint32_t dr[KPT]; // digit | rank<<8
		const uint32_t hflag = tmp[8 + (par ^ 1)];
		if (full && hflag && sizeof(K) == 4) {
			// the previous tile was skewed -- bucket hflag - 1 took more than a sixteenth of it: the lanes of a wave that
			// hold a key of that bucket take ONE fetch-add together (a same-address LDS atomic serialises per lane), their
			// first lane for all of them.  Branch-free, all fetch-adds of the tile before the first result is looked at:
			// a lane whose key is counted by another one adds zero to its own spare word.
			const uint32_t h = hflag - 1u;
			uint32_t old[KPT];
			uint64_t hm[KPT];
#pragma unroll
			for (int i = 0; i < KPT; ++i) {
				const uint32_t d = digit_of(kc[i], shift, mask);
				hm[i] = __ballot(d == h);
				const bool rides = d == h && popc_below_lane(hm[i]) != 0;
				uint32_t *at = rides ? spare + lane_id() : cnt + d;
				old[i] = atomicAdd(at, rides ? 0u : (d == h ? (uint32_t)__popcll(hm[i]) : 1u));
				dr[i] = d;
			}
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int i = 0; i < KPT; ++i) {
				const int lead = hm[i] ? __ffsll((long long)hm[i]) - 1 : 0;
				const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)old[i], lead);
				dr[i] |= (dr[i] == h ? base + popc_below_lane(hm[i]) : old[i]) << 8;
			}
		} else if (full && hflag) {
			// (8-byte keys: the batched form above spills at this kernel's register budget) lanes that share lane 0's
			// digit take one fetch-add together
#pragma unroll
			for (int i = 0; i < KPT; ++i) {
				const uint32_t d = digit_of(kc[i], shift, mask);
				const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
				const uint64_t same = __ballot(d == d0);
				uint32_t r;
				if (__popcll(same) >= 8) {
					uint32_t base = 0;
					if (lane_id() == 0) base = atomicAdd(&cnt[d0], (uint32_t)__popcll(same));
					base = __builtin_amdgcn_readfirstlane(base);
					r = d == d0 ? base + popc_below_lane(same) : atomicAdd(&cnt[d], 1u);
				} else
					r = atomicAdd(&cnt[d], 1u);
				dr[i] = d | (r << 8);
			}
		} else if (full) {
#pragma unroll
			for (int i = 0; i < KPT; ++i) {
				const uint32_t d = digit_of(kc[i], shift, mask);
				dr[i] = d | (atomicAdd(&cnt[d], 1u) << 8);
			}
		} else {
			const uint32_t rem = (uint32_t)(st.end - pos); // < T here (partial last tile)
#pragma unroll
			for (int i = 0; i < KPT; ++i) {
				const uint32_t off = (uint32_t)((i / VEC) * TH + tid) * VEC + (i % VEC);
				dr[i] = 0xFFFFFFFFu;
				if (off < rem) {
					const uint32_t d = digit_of(kc[i], shift, mask);
					dr[i] = d | (atomicAdd(&cnt[d], 1u) << 8);
				}
			}
		}
		MSD_STAMP(0); // ranks (incl. the wait for the keys)
		__syncthreads(); // B1
		MSD_STAMP(1);

		// ---- per bucket: blocks completed by this tile claim consecutive output slots
		if (tid < kP) {
			const uint32_t ct = cnt[tid];
			if (ct > (uint32_t)T / 16) tmp[8 + par] = 1u + tid; // skewed tile: the next one counts this bucket's keys per wave
			const uint32_t L_r = fill_r + ct;
			cnt[tid] = 0;
			const uint32_t nb_r = L_r / B;
			uint32_t bbase = 0;
			if (nb_r) {
				// ONE fetch-add claims the slots wslot+bbase .. +nb_r-1 (low half) and a place in the job table (high half);
				// the bucket's LDS buffer becomes the first block, further ones (skewed tiles only) are written straight
				// from registers; the flush writes the map entries of all of them
				const uint32_t claim = atomicAdd(&tmp[par], nb_r | 0x10000u);
				bbase = claim & 0xFFFFu;
				jobs[claim >> 16] = tid | (bbase << 8);
				if (nb_r > 1) multi[atomicAdd(&tmp[10 + par], 1u)] = tid | (nb_r << 8) | (bbase << 16);
			}
			meta[tid] = fill_r | (nb_r << 8) | (bbase << 20);
			fill_r = L_r - nb_r * B;
			fb_r += nb_r;
		}
		if (tid == 0) { // the other parity's counters were last read before B1
			tmp[par ^ 1] = 0;
			tmp[8 + (par ^ 1)] = 0;
			tmp[10 + (par ^ 1)] = 0;
		}
		MSD_STAMP(2); // bookkeeping
		__syncthreads(); // B2
		MSD_STAMP(3);
		const uint32_t nbtot = tmp[par] & 0xFFFFu, njobs = tmp[par] >> 16, nmulti = tmp[10 + par];

		// ---- scatter: first the remainders deferred from the previous tile, then this tile's keys
#pragma unroll
		for (int i = 0; i < KPT; ++i) {
			if (dat[i] != 0xFFFFFFFFu) {
				kbuf[dat[i]] = dkey[i];
				if constexpr (HV) vbuf[dat[i]] = dval[i];
				dat[i] = 0xFFFFFFFFu;
			}
		}
		uint32_t mt[KPT]; // (all look-ups before the first use: one LDS round trip, not one per key)
		if constexpr (sizeof(K) == 4) { // (8-byte keys: spills at this kernel's register budget)
#pragma unroll
			for (int i = 0; i < KPT; ++i) mt[i] = meta[dr[i] & 0xFFu];
		}
#pragma unroll
		for (int i = 0; i < KPT; ++i) {
			if (dr[i] != 0xFFFFFFFFu) {
				const uint32_t d = dr[i] & 0xFFu, r = dr[i] >> 8;
				const uint32_t m = sizeof(K) == 4 ? mt[i] : meta[d];
				const uint32_t vp = (m & 0xFFu) + r, nb = (m >> 8) & 0xFFFu;
				if (nb == 0 || vp < (uint32_t)B) { // tops up the bucket's buffer
					kbuf[d * B + vp] = kc[i];
					if constexpr (HV) vbuf[d * B + vp] = vc[i];
				} else if (vp < nb * B) { // skewed tile: a further whole block of this bucket, straight to its slot
					const uint64_t at = (uint64_t)(wslot + (m >> 20)) * B + vp;
					keys[at] = kc[i];
					if constexpr (HV) vals[at] = vc[i];
				} else { // remainder of a flushed bucket: its buffer is still being flushed, park it
					dat[i] = d * B + vp - nb * B;
					dkey[i] = kc[i];
					if constexpr (HV) dval[i] = vc[i];
				}
			}
		}
		MSD_STAMP(4); // scatter
		__syncthreads(); // B3
		MSD_STAMP(5);

		// ---- everything outstanding is a tile old: drain it, then refill the vacated registers
		__builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0) only
		MSD_STAMP(6); // drain
		if (npos + T < st.end) load_tile(npos + T, kc, vc);
		MSD_STAMP(7); // refill issue

		// ---- flush the completed buffers to their slots behind the read cursor
		for (uint32_t g = tid / LPB; g < njobs; g += TH / LPB) {
			const uint32_t j = jobs[g], slot = j >> 8;
			const uint32_t src = (j & 0xFFu) * B + (tid % LPB) * VEC;
			const uint64_t dst = (uint64_t)(wslot + slot) * B + (tid % LPB) * VEC;
			*reinterpret_cast<uint4 *>(keys + dst) = *reinterpret_cast<const uint4 *>(kbuf + src);
			if constexpr (HV)
				*reinterpret_cast<uint4 *>(vals + dst) = *reinterpret_cast<const uint4 *>(vbuf + src);
			if ((tid % LPB) == 0) block_map[wslot + slot] = (uint8_t)(j & 0xFFu);
		}
		for (uint32_t e = 0; e < nmulti; ++e) { // (skewed tiles only) the map entries of a bucket's further blocks
			const uint32_t m = multi[e], nb = (m >> 8) & 0xFFu;
			for (uint32_t q = tid; q + 1u < nb; q += TH) block_map[wslot + (m >> 16) + 1u + q] = (uint8_t)(m & 0xFFu);
		}
		wslot += nbtot;
		pos = npos;
		par ^= 1;
		MSD_STAMP(8); // flush
	};
	while (pos < st.end) {
		tile(kreg, vreg);
		if (pos >= st.end) break;
		tile(kregB, vregB);
	}
	MSD_STAMP_FLUSH(TH / 64);
	__syncthreads();
	// remainders deferred by the last tile
#pragma unroll
	for (int i = 0; i < KPT; ++i) {
		if (dat[i] != 0xFFFFFFFFu) {
			kbuf[dat[i]] = dkey[i];
			if constexpr (HV) vbuf[dat[i]] = dval[i];
		}
	}
	if (tid < kP) meta[tid] = fill_r;
	__syncthreads();

	// ---- stripe epilogue: leftovers (partial buffers + head keys) to the side area
	uint32_t lc = 0;
	if (tid < kP) lc = fill_r + hc[tid];
	uint32_t ltot;
	const uint32_t lex = block_excl_scan256(lc, tmp + 4, ltot);
	const size_t so = (size_t)blockIdx.x * kP + tid;
	if (tid < kP) {
		loff[tid] = lex;
		lo_cnt[so] = lc;
		lo_off[so] = lex;
		fb[so] = fb_r;
		hc[tid] = 0; // reused as head cursor
	}
	if (tid == 0) nfull[blockIdx.x] = wslot - st.slot_lo;
	__syncthreads();
	for (uint32_t idx = tid; idx < (uint32_t)PB; idx += TH) {
		const uint32_t d = idx / B, j = idx % B;
		if (j < (meta[d] & 0xFFu)) {
			lo_keys[st.lo_base + loff[d] + j] = kbuf[idx];
			if constexpr (HV) lo_vals[st.lo_base + loff[d] + j] = vbuf[idx];
		}
	}
	if (tid < h) {
		const uint32_t d = digit_of(headk[tid], shift, mask);
		const uint32_t r = atomicAdd(&hc[d], 1u);
		const uint64_t at = st.lo_base + loff[d] + (meta[d] & 0xFFu) + r;
		lo_keys[at] = headk[tid];
		if constexpr (HV) lo_vals[at] = headv[tid];
	}
}

// ------------------------------------ A': classify with direct block placement

// Variant of phase A for parents whose children are about equally big (the kernel itself: msd_direct.hpp).
// A strided sample (first round) or an exact count estimates the child boundaries; every workgroup owns, for each child, a *piece*
// (run of slots) of that child's estimated region and reads its pieces round-robin, one 256-byte
// slot per piece per row.  A completed block goes straight into the workgroup's own piece of the
// block's child whenever that piece has a slot that was already read ("write behind read" holds
// per piece); otherwise it takes any read slot of the workgroup (misplaced).  Slots that stay
// unwritten are empty.  The block permutation (B) then only has to fix misplaced / empty slots.
struct DirectPlan {
	uint32_t est_cnt[kP];   // sampled digit counts
	uint32_t bound[kP + 1]; // estimated child boundaries in slots (bound[0] = first slot of the parent)
	// neighbouring keys compared / found with equal digits: sorted, reversed or run-structured input (a
	// workgroup then reads one bucket at a time and nothing finds a slot in its own piece)
	uint32_t adj_seen, adj_same;
};

template <typename K>
__global__ __launch_bounds__(256) void direct_sample_kernel(const K *__restrict__ keys, const Parent *__restrict__ parents,
	DirectPlan *__restrict__ plan, uint32_t every)
{
	__shared__ uint32_t h[kP];
	const Parent pa = parents[0];
	const uint32_t mask = (1u << pa.width) - 1u;
	h[threadIdx.x] = 0;
	uint32_t seen = 0, same = 0; // per wave (identical in all its lanes)
	__syncthreads();
	// every `every`-th run of 256 consecutive keys (coalesced), four runs in flight per thread
	const uint64_t nruns = pa.count / 256;
	const uint64_t step = (uint64_t)gridDim.x * every;
	for (uint64_t r = (uint64_t)blockIdx.x * every; r < nruns; r += 4 * step) {
		K k4[4];
#pragma unroll
		for (int u = 0; u < 4; ++u)
			if (r + u * step < nruns) k4[u] = keys[pa.start + (r + u * step) * 256 + threadIdx.x];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			if (r + u * step < nruns) { // (uniform)
				const uint32_t d = digit_of(k4[u], pa.shift, mask);
				atomicAdd(&h[d], 1u);
				if (u == 0 && (blockIdx.x & 7u) == 0) { // (a small part of the sample is plenty)
					same += __popcll(__ballot(d == (uint32_t)__shfl_down((int)d, 1) && (threadIdx.x & 63) != 63));
					seen += 63;
				}
			}
		}
	}
	__syncthreads();
	if (h[threadIdx.x]) atomicAdd(&plan->est_cnt[threadIdx.x], h[threadIdx.x]);
	if ((threadIdx.x & 63) == 0 && seen) {
		atomicAdd(&plan->adj_seen, seen);
		atomicAdd(&plan->adj_same, same);
	}
}

// Exact digit counts per parent for the rounds after the first (a sample will not do there: the
// error of a sampled boundary is a random walk over the children before it and soon reaches the
// size of a whole child).  One workgroup per stripe, 16-byte loads, counts in LDS, then one
// device-scope add per non-empty digit.  A read-only pass: 4 B per u32 key.
template <typename K>
__global__ __launch_bounds__(1024) void direct_hist_kernel(const K *__restrict__ keys, const Stripe *__restrict__ stripes,
	const Parent *__restrict__ parents, DirectPlan *__restrict__ plans,
	// != nullptr: also OR and AND of all keys (vres[0] |= OR, vres[1] &= AND, as vary_kernel does) -- the exact check behind
	// a leading-bit skip that was decided from a sample rides on this read
	unsigned long long *__restrict__ vres)
{
	constexpr int VEC = Vec16<K>::N;
	K v_or = 0, v_and = ~(K)0;
	auto seen_key = [&](K k) {
		v_or |= k;
		v_and &= k;
		return k;
	};
	__shared__ uint32_t h[kP];
	const Stripe st = stripes[blockIdx.x];
	const Parent pa = parents[st.parent];
	const uint32_t shift = pa.shift, mask = (1u << pa.width) - 1u, tid = threadIdx.x;
	if (tid < kP) h[tid] = 0;
	uint32_t seen = 0, same = 0;
	__syncthreads();
	const uint64_t a0 = (st.begin + VEC - 1) / VEC * VEC, a1 = st.end / VEC * VEC; // 16-byte aligned part
	if (a0 < a1) {
		for (uint64_t i = st.begin + tid; i < a0; i += 1024) atomicAdd(&h[digit_of(seen_key(keys[i]), shift, mask)], 1u);
		const uint32_t nvec = (uint32_t)((a1 - a0) / VEC); // (a stripe has at most 2^20 elements)
		const K *kp = keys + a0;                           // uniform base, 32-bit offsets, loads branch-free
		constexpr int U = 8;
		for (uint32_t v = tid; v < nvec; v += U * 1024) {
			K kk[U][VEC];
#pragma unroll
			for (int u = 0; u < U; ++u) {
				const uint32_t vv = min(v + (uint32_t)u * 1024u, nvec - 1u);
				if constexpr (sizeof(K) == 4) {
					// (nontemporal: a read-once stream; tools/microbench/stream_copy.hip reads 7.0 TB/s this way, 6.3 plain)
					const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(kp + (size_t)vv * VEC));
					kk[u][0] = q.x; kk[u][1] = q.y; kk[u][2] = q.z; kk[u][3] = q.w;
				} else {
					const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(kp + (size_t)vv * VEC));
					kk[u][0] = (K)q.x | ((K)q.y << 32); kk[u][1] = (K)q.z | ((K)q.w << 32);
				}
			}
#pragma unroll
			for (int u = 0; u < U; ++u) {
				if (v + (uint32_t)u * 1024u < nvec) {
#pragma unroll
					for (int e = 0; e < VEC; ++e) atomicAdd(&h[digit_of(seen_key(kk[u][e]), shift, mask)], 1u);
					seen += 1;
					same += digit_of(kk[u][0], shift, mask) == digit_of(kk[u][VEC - 1], shift, mask) ? 1u : 0u;
				}
			}
		}
		for (uint64_t i = a1 + tid; i < st.end; i += 1024) atomicAdd(&h[digit_of(seen_key(keys[i]), shift, mask)], 1u);
	} else
		for (uint64_t i = st.begin + tid; i < st.end; i += 1024) atomicAdd(&h[digit_of(seen_key(keys[i]), shift, mask)], 1u);
	__syncthreads();
	if (tid < kP && h[tid]) atomicAdd(&plans[st.parent].est_cnt[tid], h[tid]);
	if (vres) { // (uniform)
		unsigned long long o = (unsigned long long)v_or, a = (unsigned long long)v_and | (sizeof(K) == 4 ? 0xFFFFFFFF00000000ull : 0ull);
#pragma unroll
		for (int s = 32; s > 0; s >>= 1) {
			o |= __shfl_xor(o, s);
			a &= __shfl_xor(a, s);
		}
		if ((tid & 63) == 0) {
			atomicOr(&vres[0], o);
			atomicAnd(&vres[1], a);
		}
	}
	// first and last key of every 16-byte vector: equal digits nearly always <=> locally sorted / runs
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		seen += (uint32_t)__shfl_xor((int)seen, o);
		same += (uint32_t)__shfl_xor((int)same, o);
	}
	if ((tid & 63) == 0 && seen) {
		atomicAdd(&plans[st.parent].adj_seen, seen);
		atomicAdd(&plans[st.parent].adj_same, same);
	}
}

// One workgroup per parent: counts (sampled or exact) -> child boundaries on the slot grid; a parent
// whose biggest child has more than 1.25 x the keys of its smallest one is reported as uneven.
template <int B>
__global__ __launch_bounds__(256) void direct_plan_kernel(const Parent *__restrict__ parents, DirectPlan *__restrict__ plans,
	Counters *__restrict__ ctr)
{
	__shared__ uint64_t tmp[8];
	__shared__ uint32_t s_mn, s_mx;
	const Parent pa = parents[blockIdx.x];
	DirectPlan *plan = plans + blockIdx.x;
	const uint32_t d = threadIdx.x;
	if (d == 0) {
		s_mn = 0xFFFFFFFFu;
		s_mx = 0;
	}
	const uint64_t s0 = (pa.start + B - 1) / B, s1 = (pa.start + pa.count) / B; // aligned slots of the parent
	uint64_t total;
	const uint64_t c = plan->est_cnt[d];
	const uint64_t ex = block_excl_scan256_64(c, tmp, total);
	const uint64_t ns = s1 > s0 ? s1 - s0 : 0;
	plan->bound[d] = (uint32_t)(s0 + (total ? ex * ns / total : (uint64_t)d * ns / kP));
	if (d == 0) plan->bound[kP] = (uint32_t)s1;
	if (d < (1u << pa.width)) {
		atomicMin(&s_mn, (uint32_t)c);
		atomicMax(&s_mx, (uint32_t)c);
	}
	__syncthreads();
	// (equal neighbours: 1/256 of the pairs on random keys; an eighth of them means runs)
	const bool runs = (uint64_t)plan->adj_same * 8u > plan->adj_seen;
#ifndef MSD_UNEVEN_PCT // (overridable for experiments)
#define MSD_UNEVEN_PCT 125
#endif
	if (d == 0 && (runs || !(s_mn > 0 && (double)s_mx * 100.0 <= (double)MSD_UNEVEN_PCT * (double)s_mn))) atomicAdd(&ctr->direct_uneven, 1u);
}

// The `need` lanes with the smallest 10-bit value among the eligible ones (ties: lower lane first).
// Wave-uniform radix select on ballots; returns this lane's verdict, `mask_out` = all selected lanes.
__device__ __forceinline__ bool wave_select_smallest(uint32_t v, bool elig, int need, uint64_t &mask_out)
{
	uint64_t cand = __ballot(elig), selm = 0;
	if ((int)__popcll(cand) > need) {
#pragma unroll
		for (int bit = 9; bit >= 0; --bit) {
			const uint64_t zeros = cand & ~__ballot((v >> bit) & 1u);
			const int cz = __popcll(zeros);
			if (cz >= need)
				cand = zeros;
			else {
				selm |= zeros;
				need -= cz;
				cand &= ~zeros;
			}
		}
		// the candidates left are equal: the lowest `need` lanes of them
		const bool mine = lane_bit(cand) && (int)popc_below_lane(cand) < need;
		selm |= __ballot(mine);
	} else
		selm = cand;
	mask_out = selm;
	return lane_bit(selm);
}

} // namespace msd
#include "msd_direct.hpp"
#include "msd_stream2.hpp"
namespace msd {

// ------------------------------------------------- child geometry per parent

struct ChildArrays {
	uint64_t *start;  // absolute first element
	uint64_t *count;
	uint32_t *F;      // full blocks produced for the child
	uint32_t *is;     // first interior slot
	uint32_t *I;      // interior slots (<= F)
	uint32_t *lsum;   // leftover elements from all stripes
	uint32_t *n_int;  // misplaced blocks sitting in some interior
	uint32_t *n_fr;   // misplaced blocks sitting in fringe slots
	uint32_t *n_int0; // n_int before the eviction adjustment
	uint32_t *cur_int, *cur_fr; // scatter cursors
	uint64_t *list_len, *list_base;
	uint32_t *rpos;   // claim cursor of the child's list
	uint32_t *flags;  // bit0 evict, bit1 excess
	uint32_t *rot;    // the chain-continuing entries [0, n_int) of the list are claimed from entry rot on, wrapping around
	uint32_t *nev;    // blocks of this child's list parked in the side store (each opens a chain start)
	uint32_t *xfirst; // first side-store block of those
	uint32_t *hot_cur; // kHotMax x kHotShards claim cursors (one 128-byte line each) of the longest lists
	u32x4 *lmeta;     // per list, for the block permutation: first entry, length, chain-continuing entries, rotation -- one load
};
// A list of at least kHotLen entries is claimed from kHotShards cursors instead of one: on skewed inputs half of all
// claims go to ONE list (Zipf keys: the bucket of the small keys), and a single device-scope fetch-add word saturates
// at about 88 claims per microsecond (MI355X guide) -- 1 ms of a 2.1 ms block permutation.  flags bits 8..: hot id + 1.
constexpr uint32_t kHotLen = 1u << 15, kHotMax = 256, kHotShards = 8;

template <int B>
__global__ __launch_bounds__(1024) void child_scan_kernel(const Parent *__restrict__ parents,
	const uint32_t *__restrict__ fb, const uint32_t *__restrict__ lo_cnt,
	uint32_t *__restrict__ lo_dst, ChildArrays ca)
{
	// thread (g, d): stripe group g of 4 walks a contiguous quarter of the parent's stripes for bucket d
	__shared__ uint64_t tmp[8];
	__shared__ uint32_t gF[4][kP], gL[4][kP];
	const Parent pa = parents[blockIdx.x];
	const uint32_t d = threadIdx.x & 255, g = threadIdx.x >> 8;
	const uint32_t ns = pa.stripe_hi - pa.stripe_lo, per = (ns + 3) / 4;
	const uint32_t s0 = pa.stripe_lo + g * per, s1 = s0 + per < pa.stripe_hi ? s0 + per : pa.stripe_hi;
	uint32_t F = 0, ls = 0;
#pragma unroll 8
	for (uint32_t s = s0; s < s1; ++s) {
		const size_t o = (size_t)s * kP + d;
		F += fb[o];
		ls += lo_cnt[o];
	}
	gF[g][d] = F;
	gL[g][d] = ls;
	__syncthreads();
	uint32_t pre = 0;
	for (uint32_t gg = 0; gg < g; ++gg) pre += gL[gg][d];
#pragma unroll 8
	for (uint32_t s = s0; s < s1; ++s) { // second walk: where each stripe's leftovers go inside the child
		const size_t o = (size_t)s * kP + d;
		lo_dst[o] = pre;
		pre += lo_cnt[o];
	}
	uint64_t Ft = 0, lt = 0, c = 0;
	if (g == 0) {
		Ft = (uint64_t)gF[0][d] + gF[1][d] + gF[2][d] + gF[3][d];
		lt = (uint64_t)gL[0][d] + gL[1][d] + gL[2][d] + gL[3][d];
		c = Ft * B + lt;
	}
	uint64_t total;
	const uint64_t ex = block_excl_scan256_64(c, tmp, total);
	if (g == 0 && d < (1u << pa.width)) {
		const uint32_t ci = pa.child_base + d;
		const uint64_t st = pa.start + ex, en = st + c;
		const uint64_t is = (st + B - 1) / B, ie = en / B;
		const uint64_t room = ie > is ? ie - is : 0;
		const uint64_t I = Ft < room ? Ft : room;
		ca.start[ci] = st;
		ca.count[ci] = c;
		ca.F[ci] = (uint32_t)Ft;
		ca.is[ci] = (uint32_t)is;
		ca.I[ci] = (uint32_t)I;
		ca.lsum[ci] = (uint32_t)lt;
		ca.n_int[ci] = 0;
		ca.n_fr[ci] = 0;
		ca.cur_int[ci] = 0;
		ca.cur_fr[ci] = 0;
		ca.rpos[(size_t)ci * kRposStride] = 0;
	}
}

// owner of slot i among the W children of a parent (LDS copies of is/ie), -1 if fringe
__device__ __forceinline__ int slot_owner(const uint32_t *is, const uint32_t *ie, uint32_t W, uint32_t i)
{
	// smallest c with ie[c] > i   (ie is non-decreasing)
	uint32_t lo = 0, hi = W;
	while (lo < hi) {
		const uint32_t mid = (lo + hi) >> 1;
		if (ie[mid] > i) hi = mid; else lo = mid + 1;
	}
	if (lo < W && is[lo] <= i) return (int)lo;
	return -1;
}

// In a direct-placement round nearly every slot lies in the interior of the bucket its piece belongs
// to (block_map holds that bucket for every slot of the piece, written or not): try it before searching.
__device__ __forceinline__ int slot_owner_guess(const uint32_t *is, const uint32_t *ie, uint32_t W, uint32_t i, uint32_t g)
{
	if (g < W && is[g] <= i && i < ie[g]) return (int)g;
	return slot_owner(is, ie, W, i);
}

// Pass 1 (SCATTER = false): count misplaced blocks per (child, class) and record holes.
// Pass 2 (SCATTER = true): write the per-child lists, interior-class entries first.
// Global atomics are issued once per (workgroup, child, class) / once per wave (holes);
// positions inside a reservation come from LDS fetch-adds.
#ifndef MSD_SLOT_PARTS // (overridable for experiments)
#define MSD_SLOT_PARTS 2
#endif
constexpr uint32_t kSlotParts = MSD_SLOT_PARTS;
template <bool SCATTER>
__global__ __launch_bounds__(256) void slot_classify_kernel(const Stripe *__restrict__ stripes,
	const Parent *__restrict__ parents, const uint8_t *__restrict__ block_map,
	const uint32_t *__restrict__ nfull, ChildArrays ca, ListEntry *__restrict__ list,
	ListEntry *__restrict__ holes, Counters *__restrict__ ctr, const uint8_t *__restrict__ slot_full_arg, uint32_t force)
{
	// slot_full == nullptr: a stripe's first nfull slots hold blocks (streaming classify);
	// otherwise a byte per slot says whether it holds a block (direct placement; the map counts only if the
	// direct kernel ran: the plan did not decline, or the mode forces it)
	const uint8_t *slot_full = (slot_full_arg && (force || ctr->direct_uneven == 0)) ? slot_full_arg : nullptr;
	__shared__ uint32_t s_is[kP], s_ie[kP], s_cls[2][kP], s_base[2][kP];
	// kSlotParts workgroups share a stripe (each a contiguous part of its slots); counts and list ranges are combined
	// across workgroups by atomics anyway.  Measured per launch pair at 2^30 u32 keys: 1 part 0.43 ms, 2 parts 0.37,
	// 4 parts 0.44, 8 parts 0.53 -- more workgroups shorten each sweep but every one pays the same chain of dependent
	// look-ups (stripe, parent, child geometry) and its fetch-adds on the shared counters
	Stripe st = stripes[blockIdx.x / kSlotParts];
	const Parent pa = parents[st.parent];
	const uint32_t W = 1u << pa.width, tid = threadIdx.x;
	if (tid < W) {
		s_is[tid] = ca.is[pa.child_base + tid];
		s_ie[tid] = s_is[tid] + ca.I[pa.child_base + tid];
	}
	s_cls[0][tid] = 0;
	s_cls[1][tid] = 0;
	__syncthreads();
	uint32_t nf = nfull[blockIdx.x / kSlotParts]; // streaming classify: the stripe's first nf slots hold blocks
	{
		const uint32_t all = st.slot_hi - st.slot_lo, part = blockIdx.x % kSlotParts;
		const uint32_t a = (uint32_t)((uint64_t)all * part / kSlotParts), b = (uint32_t)((uint64_t)all * (part + 1) / kSlotParts);
		st.slot_hi = st.slot_lo + b;
		st.slot_lo += a;
		nf = nf > a ? nf - a : 0u;
	}
	const uint32_t nsl = st.slot_hi - st.slot_lo;
	// Every sweep over the stripe's slots takes groups of 8 consecutive slots per thread, read with one
	// aligned 8-byte load per map (byte loads are processed lane by lane: sixteen of them per 8 slots
	// made the kernel's time), two groups in flight; the few slots before the first 8-aligned one and
	// after the last whole group go one per thread.
	auto sweep = [&](auto body) {
		const uint32_t lead = min(nsl, (8u - (st.slot_lo & 7u)) & 7u);
		const uint32_t ngr = (nsl - lead) / 8u, rest = lead + 8u * ngr;
		auto one = [&](uint32_t o) {
			const uint32_t i = st.slot_lo + o;
			body(i, (uint32_t)block_map[i], slot_full ? slot_full[i] != 0 : o < nf);
		};
		if (tid < lead) one(tid);
		if (tid < nsl - rest) one(rest + tid);
		for (uint32_t g0 = tid; g0 < ngr; g0 += 2 * 256) {
			uint2 bm[2], sf[2];
#pragma unroll
			for (int k = 0; k < 2; ++k) { // branch-free: a thread without a second group re-reads the last one
				const uint32_t g = min(g0 + k * 256u, ngr - 1u);
				const size_t at = (size_t)st.slot_lo + lead + 8u * g;
				bm[k] = *reinterpret_cast<const uint2 *>(block_map + at);
				sf[k] = slot_full ? *reinterpret_cast<const uint2 *>(slot_full + at) : make_uint2(0u, 0u);
			}
#pragma unroll
			for (int k = 0; k < 2; ++k) {
				const uint32_t g = g0 + k * 256u;
				if (g < ngr) {
#pragma unroll
					for (int u = 0; u < 8; ++u) {
						const uint32_t o = lead + 8u * g + u;
						const uint32_t b = ((u < 4 ? bm[k].x : bm[k].y) >> (8 * (u & 3))) & 0xFFu;
						const uint32_t f = ((u < 4 ? sf[k].x : sf[k].y) >> (8 * (u & 3))) & 0xFFu;
						body(st.slot_lo + o, b, slot_full ? f != 0 : o < nf);
					}
				}
			}
		}
	};
	auto owner_of = [&](uint32_t i, uint32_t d) -> int {
		return slot_full ? slot_owner_guess(s_is, s_ie, W, i, d) : slot_owner(s_is, s_ie, W, i);
	};
	// ---- count (both passes need the per-workgroup counts)
	uint32_t my_holes = 0;
	sweep([&](uint32_t i, uint32_t d, bool full) {
		const int own = owner_of(i, d);
		if (full) {
			if ((int)d != own) atomicAdd(&s_cls[own >= 0 ? 0 : 1][d], 1u);
		} else if (own >= 0)
			++my_holes;
	});
	if (!SCATTER) { // holes: one global fetch-add per workgroup, positions from an LDS cursor
		__shared__ uint32_t s_hcount, s_hbase;
		if (tid == 0) s_hcount = 0;
		__syncthreads();
		if (my_holes) atomicAdd(&s_hcount, my_holes);
		__syncthreads();
		const uint32_t nh = s_hcount;
		if (nh) { // (uniform)
			if (tid == 0) {
				s_hbase = atomicAdd(&ctr->nholes, nh);
				s_hcount = 0;
			}
			__syncthreads();
			sweep([&](uint32_t i, uint32_t d, bool full) {
				if (full) return;
				const int own = owner_of(i, d);
				if (own < 0) return;
				ListEntry e;
				e.slot = i;
				e.owner = pa.child_base + (uint32_t)own;
				holes[s_hbase + atomicAdd(&s_hcount, 1u)] = e;
			});
		}
	}
	__syncthreads();
	if (!SCATTER) {
		if (tid < W) {
			if (s_cls[0][tid]) atomicAdd(&ca.n_int[pa.child_base + tid], s_cls[0][tid]);
			if (s_cls[1][tid]) atomicAdd(&ca.n_fr[pa.child_base + tid], s_cls[1][tid]);
		}
		return;
	}
	// ---- reserve list ranges for this workgroup, then place entries
	if (tid < W) {
		const uint32_t ci = pa.child_base + tid;
		const uint32_t lb = (uint32_t)ca.list_base[ci];
		s_base[0][tid] = lb + (s_cls[0][tid] ? atomicAdd(&ca.cur_int[ci], s_cls[0][tid]) : 0);
		s_base[1][tid] = lb + ca.n_int0[ci] + (s_cls[1][tid] ? atomicAdd(&ca.cur_fr[ci], s_cls[1][tid]) : 0);
		s_cls[0][tid] = 0;
		s_cls[1][tid] = 0;
	}
	__syncthreads();
	sweep([&](uint32_t i, uint32_t d, bool full) {
		if (!full) return;
		const int own = owner_of(i, d);
		if ((int)d != own) {
			const uint32_t cls = own >= 0 ? 0 : 1;
			const uint32_t p = s_base[cls][d] + atomicAdd(&s_cls[cls][d], 1u);
			ListEntry e;
			e.slot = i;
			e.owner = own >= 0 ? pa.child_base + (uint32_t)own : kNoOwner;
			list[p] = e;
		}
	});
}

// A chain starts at a hole and ends at a chain-ending list entry; there are as many of one as of the
// other, and that number is the parallelism of the block permutation.  Inputs whose stripes hold few
// buckets (sorted, reversed, long runs) leave about one empty slot per stripe: a thousand chains of
// thousands of steps each, one wave instruction at a time (measured: 100 ms instead of 1).  Below
// kMinChains the lists therefore park more of their blocks in the side store -- every parked block ends
// a chain and the slot it left starts one -- in proportion to their length.
constexpr uint32_t kMinChains = 1u << 16;

// Per child: list length, eviction / excess decisions.
__global__ __launch_bounds__(256) void list_prepare_kernel(uint32_t nchildren, ChildArrays ca, Counters *__restrict__ ctr,
	uint32_t round_slots, uint32_t pool_cap)
{
	const uint32_t ci = blockIdx.x * 256 + threadIdx.x;
	if (ci >= nchildren) return;
	const uint32_t ni = ca.n_int[ci], nf = ca.n_fr[ci];
	const uint32_t nholes = ctr->nholes; // complete: the counting pass is over
	uint32_t ev = (ni > 0 && nf == 0) ? 1u : 0u; // a list without a chain-ending entry gets one (hole-free cycles)
	if (nholes < kMinChains && ni > ev) {
		const uint64_t want = ((uint64_t)(kMinChains - nholes) * (ni + nf) + round_slots - 1) / (round_slots ? round_slots : 1u);
		ev = (uint32_t)(want + ev < ni ? want + ev : ni);
	}
	uint32_t first = 0;
	if (ev) {
		first = atomicAdd(&ctr->nevict, ev);
		if (first + ev > pool_cap) { // (cannot happen: the pool holds kMinChains + two blocks per child)
			msd_note_error(ctr, 0u);
			ev = 0;
		}
	}
	const uint32_t ex = ca.F[ci] > ca.I[ci] ? 2u : 0u;
	ca.n_int0[ci] = ni;
	ca.n_int[ci] = ni - ev; // entries [0, n_int) keep a chain going, the rest end it
	ca.nev[ci] = ev;
	ca.xfirst[ci] = first;
	// Lists are ordered by slot and all lists are consumed at about the same pace: on locally sorted or
	// run-structured inputs entry k of every list lies at the same offset inside its child's region, a
	// whole number of child regions apart -- every access of the chip then agrees in the address bits that
	// select the memory channel (measured: 12 x slower).  A per-child starting entry breaks the lockstep.
	ca.rot[ci] = ni - ev ? (uint32_t)((ci * 2654435761u) ^ (ci >> 7) * 40503u) % (ni - ev) : 0u;
	ca.list_len[ci] = ni + nf;
	uint32_t hot = 0;
	if (ni + nf >= kHotLen) {
		const uint32_t h = atomicAdd(&ctr->nhot, 1u);
		if (h < kHotMax) {
			hot = h + 1;
			for (uint32_t sh = 0; sh < kHotShards; ++sh) ca.hot_cur[(size_t)(h * kHotShards + sh) * kRposStride] = 0;
		}
	}
	ca.flags[ci] = (ev ? 1u : 0u) | ex | (hot << 8);
}

// address of a block slot: real slots inside the key array, virtual ones in the side store
template <typename T, int B>
__device__ __forceinline__ T *slot_ptr(T *base, T *xbase, uint32_t slot)
{
	return slot < kXBase ? base + (uint64_t)slot * B : xbase + (uint64_t)(slot - kXBase) * B;
}

// One wave per child.  (a) The last nev chain-continuing entries of the child's list are parked in the
// side store: the block is copied there, the entry now points at the copy and ends a chain, the slot it
// left becomes a hole (owned by the interior it lies in).  (b) A child with more full blocks than
// interior slots opens the virtual slot that takes its excess block.
// Side store layout: block 2*ci+1 = child ci's excess block; blocks from 2*nchildren on = the eviction pool.
template <typename K, typename V>
__global__ __launch_bounds__(256) void evict_kernel(uint32_t nchildren, ChildArrays ca,
	ListEntry *__restrict__ list, ListEntry *__restrict__ holes, Counters *__restrict__ ctr,
	K *__restrict__ keys, uint64_t *__restrict__ vals, K *__restrict__ xkeys, uint64_t *__restrict__ xvals,
	uint32_t waves_per_child)
{
	constexpr int B = Cfg<K, V>::B;
	constexpr bool HV = has_val<V>::value;
	constexpr int VEC = Vec16<K>::N;
	constexpr int LPB = B / VEC; // lanes per block (16)
	if (waves_per_child == 0) { // many children, next to nothing to do for any of them: one thread per child
		const uint32_t ci = blockIdx.x * 256 + threadIdx.x;
		if (ci >= nchildren) return;
		const uint32_t fl = ca.flags[ci];
		if (fl == 0) return;
		if (fl & 1u) {
			const uint32_t nev = ca.nev[ci], xf = ca.xfirst[ci];
			const uint64_t e0 = ca.list_base[ci] + ca.n_int[ci];
			const uint32_t hbase = atomicAdd(&ctr->nholes, nev);
			for (uint32_t j = 0; j < nev; ++j) {
				const ListEntry ent = list[e0 + j];
				const uint32_t xs = kXBase + 2 * nchildren + xf + j;
				K *dk = slot_ptr<K, B>(keys, xkeys, xs);
				const K *sk = slot_ptr<K, B>(keys, xkeys, ent.slot);
				for (int e = 0; e < B; ++e) dk[e] = sk[e];
				if constexpr (HV) {
					uint64_t *dv = slot_ptr<uint64_t, B>(vals, xvals, xs);
					const uint64_t *sv = slot_ptr<uint64_t, B>(vals, xvals, ent.slot);
					for (int e = 0; e < B; ++e) dv[e] = sv[e];
				}
				ListEntry ne;
				ne.slot = xs;
				ne.owner = kNoOwner;
				list[e0 + j] = ne;
				holes[hbase + j] = ent;
			}
		}
		if (fl & 2u) {
			ListEntry hsl;
			hsl.slot = kXBase + 2 * ci + 1;
			hsl.owner = ci;
			holes[atomicAdd(&ctr->nholes, 1u)] = hsl;
		}
		return;
	}
	// waves_per_child waves share a child (few children with many blocks to park: low-cardinality keys)
	const uint32_t wid = blockIdx.x * 4 + threadIdx.x / 64, lane = threadIdx.x & 63;
	const uint32_t ci = wid / waves_per_child, part = wid % waves_per_child;
	if (ci >= nchildren) return;
	const uint32_t fl = ca.flags[ci];
	if (fl == 0) return;
	if (fl & 1u) {
		const uint32_t nev = ca.nev[ci], xf = ca.xfirst[ci];
		const uint64_t e0 = ca.list_base[ci] + ca.n_int[ci]; // entries [n_int, n_int0) are parked
		// this wave's share [j0, j1) of the parked entries; its holes go to a range of their own
		const uint32_t j0 = (uint32_t)((uint64_t)nev * part / waves_per_child), j1 = (uint32_t)((uint64_t)nev * (part + 1) / waves_per_child);
		uint32_t hbase = 0;
		if (lane == 0 && j1 > j0) hbase = atomicAdd(&ctr->nholes, j1 - j0);
		hbase = __shfl(hbase, 0) - j0;
		for (uint32_t j = j0 + lane / LPB; j < j1; j += 64 / LPB) {
			const ListEntry ent = list[e0 + j];
			const uint32_t xs = kXBase + 2 * nchildren + xf + j, sub = lane % LPB;
			*reinterpret_cast<uint4 *>(slot_ptr<K, B>(keys, xkeys, xs) + sub * VEC) =
				*reinterpret_cast<const uint4 *>(slot_ptr<K, B>(keys, xkeys, ent.slot) + sub * VEC);
			if constexpr (HV)
				*reinterpret_cast<uint4 *>(slot_ptr<uint64_t, B>(vals, xvals, xs) + sub * VEC) =
					*reinterpret_cast<const uint4 *>(slot_ptr<uint64_t, B>(vals, xvals, ent.slot) + sub * VEC);
			if (sub == 0) {
				ListEntry ne;
				ne.slot = xs;
				ne.owner = kNoOwner;
				list[e0 + j] = ne;
				holes[hbase + j] = ent; // the vacated slot, owned by the interior it lies in
			}
		}
	}
	if ((fl & 2u) && lane == 0 && part == 0) {
		ListEntry hsl;
		hsl.slot = kXBase + 2 * ci + 1;
		hsl.owner = ci;
		holes[atomicAdd(&ctr->nholes, 1u)] = hsl;
	}
}

// What a chain step needs to know about a list, in one 16-byte load (five separate arrays cost the step two more
// dependent round trips).
__global__ __launch_bounds__(256) void list_pack_kernel(uint32_t nchildren, ChildArrays ca)
{
	const uint32_t ci = blockIdx.x * 256 + threadIdx.x;
	if (ci >= nchildren) return;
	ca.lmeta[ci] = u32x4{ (uint32_t)ca.list_base[ci], (uint32_t)ca.list_len[ci], ca.n_int[ci], ca.rot[ci] };
}

// ------------------------------------------------------ B: block permutation

// Every lane runs one chain: own a hole -> claim the next misplaced block of the
// hole's bucket (fetch-add on the bucket's list cursor) -> the wave moves the 64
// claimed blocks -> the vacated slot is the lane's next hole, unless the claimed
// entry was fringe-class, which ends the chain.  No lane ever waits for another.
// Lanes of a wave that claim from the same bucket share one fetch-add (skewed inputs
// put most chains into one bucket).
template <typename K, typename V>
__global__ __launch_bounds__(256) void chains_kernel(ChildArrays ca, const ListEntry *__restrict__ list,
	const ListEntry *__restrict__ holes, Counters *__restrict__ ctr,
	K *__restrict__ keys, uint64_t *__restrict__ vals, K *__restrict__ xkeys, uint64_t *__restrict__ xvals,
	uint32_t real_slots, uint32_t side_slots)
{
	constexpr int B = Cfg<K, V>::B;
	constexpr bool HV = has_val<V>::value;
	constexpr int VEC = Vec16<K>::N;
	constexpr int LPB = B / VEC;      // lanes per block (16)
	constexpr int BPI = 64 / LPB;     // blocks per wave instruction (4)
	// a slot id from the lists is an address: refuse what lies outside the array / the side store
	// (corrupted metadata must end in an error code, never in a stray 256-byte write)
	auto slot_ok = [&](uint32_t s) { return s < kXBase ? s < real_slots : s - kXBase < side_slots; };
	constexpr int NI = 64 / BPI;      // instructions to move 64 blocks (16)
	constexpr int GROUPS = 4;         // shared claims are formed for the 4 most common buckets
	const uint32_t lane = lane_id();
	const uint32_t nholes = ctr->nholes;
	uint32_t hole = 0, owner = 0;
	// the list of `owner`: {first entry, length, chain-continuing entries, rotation} and its flags -- loaded when the
	// lane learns its next owner, i.e. while the blocks of the current step move
	u32x4 om = { 0u, 0u, 0u, 0u };
	uint32_t ofl = 0;
	bool active = false, exhausted = false;
	uint32_t steps = 0;
	// claims from a hot list: which of its shards this lane tries next (first choice by workgroup and wave, so that the
	// waves of the chip spread over the shards), and how many it has found exhausted
	const uint32_t shard0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) % kHotShards;
	uint32_t tries = 0;

	// The first 64 holes of every wave are its own, by position: one fetch-add per wave on the one cursor word would be
	// most of this kernel's time when there are few holes (direct placement: 2.6e5 holes for 8192 waves, 0.15 ms of
	// claims at the rate a single word takes them).  The cursor hands out what lies behind those.
	const uint32_t nwaves = gridDim.x * (blockDim.x >> 6), wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	bool first = true;
	MSD_STAMP_DECL(7);
	MSD_STAMP_START();
	for (;;) {
		MSD_STAMP_TICK(11);
		// ---- lanes without a hole fetch the next chain start
		if (!exhausted) {
			const uint64_t need = __ballot(!active);
			if (need) {
				const uint32_t n = __popcll(need);
				const int ldr = __ffsll((long long)need) - 1;
				uint32_t base = wave * 64u;
				if (!first) {
					if ((int)lane == ldr) base = atomicAdd(&ctr->hole_cursor, n);
					base = __shfl(base, ldr) + nwaves * 64u;
				}
				first = false;
				if (nwaves * 64u >= nholes) exhausted = true; // (every hole is some wave's own)
				if (!active) {
					const uint32_t my = base + popc_below_lane(need);
					if (my < nholes) {
						const ListEntry e = holes[my];
						hole = e.slot;
						owner = e.owner;
						om = ca.lmeta[owner];
						ofl = ca.flags[owner];
						active = true;
					}
				}
				if (base + n >= nholes) exhausted = true;
			}
		}
		const uint64_t act = __ballot(active);
		if (!act) break;
		MSD_STAMP(0); // chain starts

		// ---- claim one source block per active lane; lanes with equal owner (and shard) share a fetch-add.
		// A hot list's chain-continuing entries [0, n_int) are split over kHotShards cursors; its chain-ending entries
		// keep the list's own cursor and are handed out only to a lane that has found every shard used up -- a chain
		// may only end where no chain-continuing entry of the list is left (otherwise the
		// remaining entries can form cycles no chain enters).
		const uint32_t fl = active ? ofl : 0u;
		const uint32_t hot = fl >> 8;                                  // hot id + 1, or 0
		const uint32_t shard = !hot ? 0u : tries < kHotShards ? (shard0 + tries) % kHotShards : kHotShards; // kHotShards: the ending entries
		const uint32_t ckey = owner * 16u + shard;                     // (owner < 2^28)
		uint64_t rem = act;
		int my_ldr = (int)lane;
		uint32_t my_rank = 0, my_cnt = 1;
#pragma unroll
		for (int g = 0; g < GROUPS; ++g) {
			if (rem) {
				const int l = __ffsll((long long)rem) - 1;
				const uint32_t o = __shfl(ckey, l);
				const uint64_t same = __ballot(active && ckey == o) & rem;
				if (lane_bit(same)) {
					my_ldr = l;
					my_rank = popc_below_lane(same);
					my_cnt = __popcll(same);
				}
				rem &= ~same;
			}
		}
		MSD_STAMP(1); // owner flags + grouping
		uint32_t idx = 0;
		if (active && my_ldr == (int)lane)
			idx = atomicAdd(hot && shard < kHotShards ? &ca.hot_cur[(size_t)((hot - 1) * kHotShards + shard) * kRposStride]
								  : &ca.rpos[(size_t)owner * kRposStride], my_cnt);
		idx = __shfl(idx, my_ldr) + my_rank;

		MSD_STAMP(2); // claim
		uint32_t src = 0, src_owner = kNoOwner;
		bool last = false, retry = false;
		u32x4 nom = om;
		uint32_t nfl = ofl;
		if (active) {
			const uint32_t len = om.y, nint = om.z;
			if (hot) {
				if (shard < kHotShards) { // shard `shard` holds the chain-continuing entries [n_int * shard / S, n_int * (shard + 1) / S)
					const uint32_t sb = (uint32_t)((uint64_t)nint * shard / kHotShards), se = (uint32_t)((uint64_t)nint * (shard + 1) / kHotShards);
					if (idx < se - sb)
						idx += sb;
					else { // used up: the lane keeps its hole and asks the next shard (then the ending entries) in the next step
						retry = true;
						++tries;
					}
				} else
					idx = idx < len - nint ? idx + nint : len; // the list's own cursor counts its chain-ending entries (len: error below)
			}
			if (retry) {
			} else if (idx >= len) { // cannot happen when the bookkeeping is right
				msd_note_error(ctr, 1u);
				active = false;
			} else {
				tries = 0;
				last = idx >= nint;
				uint32_t at = idx;
				if (!last) { // chain-continuing entries: rotated order
					at += om.w;
					if (at >= nint) at -= nint;
				}
				const ListEntry e = list[om.x + at];
				src = e.slot;
				src_owner = e.owner;
				if (!slot_ok(src) || !slot_ok(hole)) {
					msd_note_error(ctr, 2u);
					active = false;
				} else if (!last) { // the next step's list: on its way while this step's blocks move
					nom = ca.lmeta[src_owner];
					nfl = ca.flags[src_owner];
				}
			}
		}
		const uint64_t mv = __ballot(active && !retry);
		++steps;
		MSD_STAMP(3); // list geometry + entry
#ifdef MSD_STAMPS
		if constexpr (kStampThis) st_acc[10] += __popcll(mv);
#endif

		// ---- move the claimed blocks: lane group g of instruction i serves chain i*BPI+g.
		// Loads are unconditional (idle chains read slot 0) so that the 16 vectors stay in registers.
		const uint32_t sub = lane % LPB;
		const uint32_t src_safe = active && !retry ? src : 0u;
		u32x4 kd[NI];
		u32x4 vd[HV ? NI : 1];
#pragma unroll
		for (int i = 0; i < NI; ++i) {
			const int chain = i * BPI + (int)(lane / LPB);
			const uint32_t s = __shfl(src_safe, chain);
			kd[i] = *reinterpret_cast<const u32x4 *>(slot_ptr<K, B>(keys, xkeys, s) + sub * VEC);
			if constexpr (HV)
				vd[i] = *reinterpret_cast<const u32x4 *>(slot_ptr<uint64_t, B>(vals, xvals, s) + sub * VEC);
		}
#pragma unroll
		for (int i = 0; i < NI; ++i) {
			const int chain = i * BPI + (int)(lane / LPB);
			const uint32_t hdst = __shfl(hole, chain);
			if ((mv >> chain) & 1ull) {
				*reinterpret_cast<u32x4 *>(slot_ptr<K, B>(keys, xkeys, hdst) + sub * VEC) = kd[i];
				if constexpr (HV)
					*reinterpret_cast<u32x4 *>(slot_ptr<uint64_t, B>(vals, xvals, hdst) + sub * VEC) = vd[i];
			}
		}
		if (active && !retry) {
			if (last)
				active = false;
			else {
				hole = src;
				owner = src_owner;
				om = nom;
				ofl = nfl;
			}
		}
		MSD_STAMP(4); // block loads + stores
	}
	MSD_STAMP_FLUSH(4);
	if (lane == 0 && steps) atomicAdd(&ctr->chain_steps, steps);
}

// every list must have been consumed exactly
__global__ __launch_bounds__(256) void chains_verify_kernel(uint32_t nchildren, ChildArrays ca, Counters *ctr)
{
	const uint32_t ci = blockIdx.x * 256 + threadIdx.x;
	if (ci >= nchildren) return;
	const uint32_t len = (uint32_t)ca.list_len[ci], hot = ca.flags[ci] >> 8;
	if (!hot) {
		if (ca.rpos[(size_t)ci * kRposStride] != len) msd_note_error(ctr, 3u);
		return;
	}
	// sharded cursors: every shard of the chain-continuing entries used up (lanes that found a shard empty have bumped
	// its cursor beyond its length), the list's own cursor = its chain-ending entries
	const uint32_t nint = ca.n_int[ci];
	for (uint32_t sh = 0; sh < kHotShards; ++sh) {
		const uint32_t sb = (uint32_t)((uint64_t)nint * sh / kHotShards), se = (uint32_t)((uint64_t)nint * (sh + 1) / kHotShards);
		if (ca.hot_cur[(size_t)((hot - 1) * kHotShards + sh) * kRposStride] < se - sb) msd_note_error(ctr, 4u);
	}
	if (ca.rpos[(size_t)ci * kRposStride] != len - nint) msd_note_error(ctr, 5u);
}

// ------------------------------------------------------------- C: cleanup

// position of logical fringe offset o of a child
template <int B>
__device__ __forceinline__ uint64_t fringe_pos(uint64_t start, uint32_t is, uint32_t I, uint64_t o)
{
	if (I == 0) return start + o;
	const uint64_t head = (uint64_t)is * B - start;
	return o < head ? start + o : (uint64_t)(is + I) * B + (o - head);
}

template <typename K, typename V>
__global__ __launch_bounds__(256) void cleanup_kernel(const Stripe *__restrict__ stripes,
	const Parent *__restrict__ parents, const uint32_t *__restrict__ lo_cnt,
	const uint32_t *__restrict__ lo_off, const uint32_t *__restrict__ lo_dst, ChildArrays ca,
	const K *__restrict__ lo_keys, const uint64_t *__restrict__ lo_vals,
	K *__restrict__ keys, uint64_t *__restrict__ vals)
{
	constexpr int B = Cfg<K, V>::B;
	constexpr bool HV = has_val<V>::value;
	__shared__ uint32_t s_off[kP + 1], s_dst[kP], s_is[kP], s_I[kP];
	__shared__ uint64_t s_start[kP];
	const Stripe st = stripes[blockIdx.x];
	const Parent pa = parents[st.parent];
	const uint32_t tid = threadIdx.x, W = 1u << pa.width;
	const size_t so = (size_t)blockIdx.x * kP + tid;
	s_off[tid] = lo_off[so];
	s_dst[tid] = lo_dst[so];
	if (tid == kP - 1) s_off[kP] = lo_off[so] + lo_cnt[so];
	if (tid < W) {
		s_is[tid] = ca.is[pa.child_base + tid];
		s_I[tid] = ca.I[pa.child_base + tid];
		s_start[tid] = ca.start[pa.child_base + tid];
	}
	__syncthreads();
	const uint32_t total = s_off[kP];
	for (uint32_t idx = tid; idx < total; idx += 256) {
		// last d with s_off[d] <= idx
		uint32_t lo = 0, hi = kP;
		while (hi - lo > 1) {
			const uint32_t mid = (lo + hi) >> 1;
			if (s_off[mid] <= idx) lo = mid; else hi = mid;
		}
		const uint32_t d = lo;
		const uint64_t o = (uint64_t)s_dst[d] + (idx - s_off[d]);
		const uint64_t p = fringe_pos<B>(s_start[d], s_is[d], s_I[d], o);
		keys[p] = lo_keys[st.lo_base + idx];
		if constexpr (HV) vals[p] = lo_vals[st.lo_base + idx];
	}
}

// excess blocks (one per child at most) go behind the stripes' leftovers
template <typename K, typename V>
__global__ __launch_bounds__(64) void excess_kernel(uint32_t nchildren, ChildArrays ca,
	const K *__restrict__ xkeys, const uint64_t *__restrict__ xvals,
	K *__restrict__ keys, uint64_t *__restrict__ vals)
{
	constexpr int B = Cfg<K, V>::B;
	constexpr bool HV = has_val<V>::value;
	const uint32_t ci = blockIdx.x;
	if (ci >= nchildren || !(ca.flags[ci] & 2u)) return;
	const uint64_t st = ca.start[ci];
	const uint32_t is = ca.is[ci], I = ca.I[ci], ls = ca.lsum[ci];
	for (uint32_t j = threadIdx.x; j < (uint32_t)B; j += 64) {
		const uint64_t p = fringe_pos<B>(st, is, I, (uint64_t)ls + j);
		keys[p] = xkeys[(uint64_t)(2 * ci + 1) * B + j];
		if constexpr (HV) vals[p] = xvals[(uint64_t)(2 * ci + 1) * B + j];
	}
}

// children -> next round's parents / the small-segment lists / done.  A workgroup serves 256 >> wmax parents (wmax:
// the round's widest digit) and takes its places in each list with ONE fetch-add per list: the list counters are single
// words, and a fetch-add per wave on one word is what the device serialises (65536 narrow parents of a register-resident
// round, one wave each: 0.7 ms).
__global__ __launch_bounds__(256) void collect_kernel(const Parent *__restrict__ parents, uint32_t nparents, uint32_t wmax, ChildArrays ca,
	uint64_t small_max, uint64_t med_max, uint32_t small_cap, uint32_t count_bits, Segment *__restrict__ next_parents,
	Segment *__restrict__ small, Segment *__restrict__ small_count, Segment *__restrict__ big, uint32_t big_cap,
	Counters *__restrict__ ctr, uint64_t *__restrict__ count_out, uint32_t count_n,
	// a sort that stops early (msd_sort_*_top: the keys are to be ordered by key >> stop_bits only): children whose open
	// bits all lie below stop_bits are done
	uint32_t stop_bits = 0)
{
	__shared__ uint32_t s_n[4], s_base[4], s_max; // lists: 0 small, 1 counting leaf, 2 big counting sort, 3 next parents
	const uint32_t tid = threadIdx.x;
	if (tid < 4) s_n[tid] = 0;
	if (tid == 0) s_max = 0;
	__syncthreads();
	const uint32_t pi = blockIdx.x * (256u >> wmax) + (tid >> wmax), d = tid & ((1u << wmax) - 1u);
	int which = -1;
	uint32_t rank = 0;
	Segment s = {};
	if (pi < nparents) {
		const Parent pa = parents[pi];
		if (d < (1u << pa.width)) {
			const uint32_t ci = pa.child_base + d;
			const uint64_t c = ca.count[ci];
			if (count_out && ci < count_n) count_out[ci] = c;
			if (c > 1 && pa.shift > stop_bits) {
				s.start = ca.start[ci];
				s.count = c;
				s.bits = pa.shift;
				s.pad = 0;
				const bool countable = pa.shift <= count_bits; // all open bits fit one counting pass
				if (countable && c >= 64 && c <= med_max)
					which = 1; // one workgroup: LDS byte counters
				else if (c > small_max)
					which = (big && countable && c < 0xFFFF0000ull) ? 2 : 3; // no further round: the multi-workgroup counting sort
				else
					which = 0;
				rank = atomicAdd(&s_n[which], 1u);
				if (which == 3) atomicMax(&s_max, (uint32_t)(c < 0xFFFFFFFFull ? c : 0xFFFFFFFFull));
			}
		}
	}
	__syncthreads();
	if (tid < 4 && s_n[tid]) {
		uint32_t *cnt = tid == 0 ? &ctr->nsmall : tid == 1 ? &ctr->ncount : tid == 2 ? &ctr->nbig : &ctr->next_parents;
		s_base[tid] = atomicAdd(cnt, s_n[tid]);
		if (tid == 3) atomicMax(&ctr->next_max, s_max);
	}
	__syncthreads();
	if (which < 0) return;
	uint32_t at = s_base[which] + rank;
	if (which == 2 && at >= big_cap) { // the big list is full: another round instead
		which = 3;
		at = atomicAdd(&ctr->next_parents, 1u);
		atomicMax(&ctr->next_max, (uint32_t)(s.count < 0xFFFFFFFFull ? s.count : 0xFFFFFFFFull));
	}
	if (which == 3)
		next_parents[at] = s;
	else if (which == 2)
		big[at] = s;
	else if (at < small_cap)
		(which == 0 ? small : small_count)[at] = s;
	else
		msd_note_error(ctr, 6u);
}

} // namespace msd
#include "msd_regpart.hpp"
namespace msd {

// ------------------------------------------------- one-pass counting sort

// A segment whose keys differ only in their low `bits` <= 16 bits is finished in ONE
// pass over all of those bits, so the pass needs no stability: 2^bits byte counters in
// LDS are bumped with LDS fetch-adds and, the keys being nothing but (common prefix |
// counted value), the sorted segment is re-generated from the counters -- no key is
// moved at all.  A value that occurs more than 255 times overflows its byte: the
// workgroup then leaves the segment untouched and queues it for the general LDS sort.
constexpr int kCountTh = 1024;
constexpr int kCountMaxBits = 16;
#ifndef MSD_COUNT_MED_LOG // (overridable for experiments)
#define MSD_COUNT_MED_LOG 17
#endif
constexpr uint64_t kCountMedMax = 1ull << MSD_COUNT_MED_LOG; // largest segment one workgroup counts by itself
// counter word i lives at i + i/32: a thread's 16 consecutive words and its neighbours' then
// fall on different LDS banks (unpadded, the stride-16 walk is a 32-way bank conflict)
constexpr size_t kCountCwBytes = (((size_t)1 << kCountMaxBits) / 4 + ((size_t)1 << kCountMaxBits) / 128 + 4) * 4;
constexpr int kCountStageBytes = 14080; // output window; with the counters: two workgroups per CU
constexpr size_t kCountLds = kCountCwBytes + kCountStageBytes + 128; // 128: wave totals, flag, ticket, prefix
__device__ __forceinline__ uint32_t cw_at(uint32_t i) { return i + (i >> 5); }

// Persistent workgroups: each takes segments by ticket and loads the first keys of its NEXT segment
// while it stores the current one.  Two kernels share the counting phase:
//  * count_place_kernel (the fast path): a segment whose keys all sit in registers (<= 17 Ki u32) keeps
//    each fetch-add's return value, the key's rank among equal keys.  Every thread then turns its counter
//    bytes into exclusive prefix sums relative to its own first output position (they fit a byte as long
//    as a thread owns <= 255 keys) and publishes that position; a key's place is
//    base[owner of its value] + prefix[value] + rank, and all keys are written at once into an output
//    buffer that takes over the counters' LDS.  Segments it cannot take (too long, a thread with > 255
//    keys, a byte overflow) are queued untouched for
//  * count_walk_kernel: no ranks; every thread walks its own counters and re-generates its run of the
//    output window by window (the windows grow into the counter words already consumed).
struct CountLds {
	uint32_t *cw;     // packed byte counters (padded layout)
	uint32_t *wtot;   // 16 wave totals
	uint32_t *nexti;  // next ticket
	uint32_t *tfree;  // walk: the thread cut by the window's end
	uint32_t *crowded; // place: some thread owns more than 255 keys
};

template <typename K>
__global__ __launch_bounds__(kCountTh, 8) void count_place_kernel(K *__restrict__ keys,
	const Segment *__restrict__ segs, uint32_t nsegs_host, const uint32_t *__restrict__ nsegs_dev,
	Segment *__restrict__ slow, Counters *__restrict__ ctr)
{
	const uint32_t nsegs = nsegs_dev ? *nsegs_dev : nsegs_host; // (the list may come from count_place16_kernel)
	// keys per thread held in registers: a little more than 2^14 / 1024, the typical segment
	constexpr int PF = sizeof(K) == 4 ? 17 : 9;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint32_t *cw = reinterpret_cast<uint32_t *>(smem);
	uint32_t *tbase = reinterpret_cast<uint32_t *>(smem + kCountCwBytes); // per-thread output base (before the buffer is used)
	K *out = reinterpret_cast<K *>(smem);                                  // counters + stage as one output buffer
	uint32_t *wtot = reinterpret_cast<uint32_t *>(smem + kCountCwBytes + kCountStageBytes);
	uint32_t *nexti = wtot + 17;
	K *hi_l = reinterpret_cast<K *>(wtot + 18); // 8 bytes
	uint32_t *crowded = wtot + 21;
	static_assert((size_t)PF * kCountTh * sizeof(K) <= kCountCwBytes + kCountStageBytes, "output buffer");
	const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	if (blockIdx.x >= nsegs) return;
	Segment sg = segs[blockIdx.x];
	K pk[PF];
	auto prefetch = [&](const Segment &g) {
		const K *src = keys + g.start;
		// branch-free (a conditional prefetch makes the compiler wait for the loads where they are issued):
		// out-of-range lanes re-read the last key, a segment too long for the fast path only its first key
		const uint32_t cnt = g.count <= (uint64_t)(PF * kCountTh) ? (uint32_t)g.count : 1u;
#pragma unroll
		for (int u = 0; u < PF; ++u) {
			const uint32_t idx = min((uint32_t)(u * kCountTh) + tid, cnt - 1u);
			pk[u] = src[idx];
		}
	};
	prefetch(sg);
	MSD_STAMP_DECL(2);
	MSD_STAMP_START();
	for (;;) {
		const uint32_t n = (uint32_t)sg.count;
		const uint32_t nv = 1u << sg.bits, mask = nv - 1u;
		const uint32_t nwords = nv >= 4 ? nv / 4 : 1;
		K *seg = keys + sg.start;
		const bool in_regs = n <= (uint32_t)(PF * kCountTh);
		MSD_STAMP(9);
		MSD_STAMP_TICK(11);

		if (in_regs)
			for (uint32_t j = tid; j < cw_at(nwords) + 1; j += kCountTh) cw[j] = 0;
		if (tid == 0) {
			*nexti = atomicAdd(&ctr->count_ticket, 1u) + gridDim.x;
			*crowded = 0;
			*hi_l = pk[0] & ~(K)mask; // common prefix of the whole segment
		}
		MSD_STAMP(0); // clear
		__syncthreads();
		MSD_STAMP(1);
		// value (low 16 bits) | rank among equal keys, later | output position (high 16 bits)
		uint32_t rk[PF];
		if (in_regs) {
#pragma unroll
			for (int u = 0; u < PF; ++u) {
				const uint32_t left = n > (uint32_t)(u * kCountTh) ? n - u * kCountTh : 0u;
				rk[u] = 0;
				if (tid < left) {
					const uint32_t v = (uint32_t)pk[u] & mask, sh = (v & 3u) * 8u;
					rk[u] = v | (((atomicAdd(&cw[cw_at(v >> 2)], 1u << sh) >> sh) & 0xFFu) << 16);
				}
			}
		}
		MSD_STAMP(2); // fetch-adds (incl. the wait for the keys)
		__syncthreads();
		MSD_STAMP(3);
		const uint32_t nxt = *nexti;
		const K hi = *hi_l;
		// ---- exclusive prefix of the counters; thread t owns words [t*wpt, (t+1)*wpt), wpt <= 16
		// (wpt divides 32, so the owned words are consecutive in the padded layout too)
		const uint32_t wpt = nwords >= kCountTh ? nwords / kCountTh : 1;
		const uint32_t w0 = tid * wpt;
		uint32_t *cww = cw + cw_at(w0);
		uint32_t tot = 0;
		if (in_regs && w0 < nwords) {
#pragma unroll 4
			for (uint32_t j = 0; j < wpt; ++j) tot = __builtin_amdgcn_sad_u8(cww[j], 0u, tot); // byte sum
		}
		if (tot > 255u) *crowded = 1;
		const uint32_t inc = wave_incl_scan(tot);
		if (lane == 63) wtot[w] = inc;
		MSD_STAMP(4); // byte sums + wave scan
		__syncthreads();
		uint32_t pos = inc - tot, all = 0;
#pragma unroll 2
		for (uint32_t ww = 0; ww < kCountTh / 64; ++ww) {
			const uint32_t t = wtot[ww];
			if (ww < w) pos += t;
			all += t;
		}
		// a byte that overflowed carried into its neighbour: the sum of all bytes then falls short of n
		const bool ok = in_regs && all == n && *crowded == 0;
		MSD_STAMP(5); // barrier + block scan
		if (ok) {
			uint32_t run = 0;
			if (w0 < nwords) {
#pragma unroll 4
				for (uint32_t j = 0; j < wpt; ++j) {
					const uint32_t x = cww[j], y = x * 0x01010101u; // bytes of y: inclusive sums inside the word
					cww[j] = (y - x) + run * 0x01010101u;
					run += y >> 24;
				}
			}
			tbase[tid] = pos;
			MSD_STAMP(6); // prefix inside the thread's words
			__syncthreads();
			const uint32_t lgw = (uint32_t)__builtin_ctz(wpt);
#pragma unroll
			for (int u = 0; u < PF; ++u) {
				const uint32_t left = n > (uint32_t)(u * kCountTh) ? n - u * kCountTh : 0u;
				if (tid < left) { // (positions are below 2^15: the segment has at most 17 Ki keys)
					const uint32_t v = rk[u] & 0xFFFFu, wi = v >> 2;
					rk[u] += (tbase[wi >> lgw] + ((cw[cw_at(wi)] >> ((v & 3u) * 8u)) & 0xFFu)) << 16;
				}
			}
			MSD_STAMP(7); // positions
			__syncthreads(); // counters and bases are dead: everything up to the stage's end is the output buffer
#pragma unroll
			for (int u = 0; u < PF; ++u) {
				const uint32_t left = n > (uint32_t)(u * kCountTh) ? n - u * kCountTh : 0u;
				if (tid < left) out[rk[u] >> 16] = hi | (K)(rk[u] & 0xFFFFu);
			}
			__syncthreads();
			MSD_STAMP(8); // barrier + output into LDS + barrier
		} else if (tid == 0)
			slow[atomicAdd(&ctr->nslow, 1u)] = sg; // untouched
		Segment nsg = sg;
		if (nxt < nsegs) { // the next segment's keys travel while this one is stored
			nsg = segs[nxt];
			prefetch(nsg);
		}
		if (ok) {
			for (uint32_t i = tid; i < n; i += kCountTh) seg[i] = out[i];
		}
		MSD_STAMP(10); // prefetch issue + store
		if (nxt >= nsegs) break;
		sg = nsg;
		__syncthreads(); // the output buffer is cleared next
	}
	MSD_STAMP_FLUSH(kCountTh / 64);
}

template <typename K>
__global__ __launch_bounds__(kCountTh, 8) void count_walk_kernel(K *__restrict__ keys,
	const Segment *__restrict__ segs, const uint32_t *__restrict__ nsegs_dev, Segment *__restrict__ fallback,
	uint32_t fallback_base, uint32_t lds_cap, Segment *__restrict__ big, uint32_t big_cap, Counters *__restrict__ ctr)
{
	constexpr uint32_t WS = kCountStageBytes / sizeof(K); // keys per output window
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint32_t *cw = reinterpret_cast<uint32_t *>(smem);    // packed byte counters (padded layout)
	K *stage = reinterpret_cast<K *>(smem + kCountCwBytes);
	uint32_t *wtot = reinterpret_cast<uint32_t *>(smem + kCountCwBytes + kCountStageBytes); // 16 wave totals
	uint32_t *nexti = wtot + 17;
	K *hi_l = reinterpret_cast<K *>(wtot + 18); // 8 bytes
	uint32_t *tfree = wtot + 20;
	const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const uint32_t nsegs = *nsegs_dev;
	if (blockIdx.x >= nsegs) return;
	Segment sg = segs[blockIdx.x];
	for (;;) {
		const uint32_t n = (uint32_t)sg.count;
		const uint32_t nv = 1u << sg.bits, mask = nv - 1u;
		const uint32_t nwords = nv >= 4 ? nv / 4 : 1;
		K *seg = keys + sg.start;

		for (uint32_t j = tid; j < cw_at(nwords) + 1; j += kCountTh) cw[j] = 0;
		if (tid == 0) {
			*nexti = atomicAdd(&ctr->count_ticket2, 1u) + gridDim.x;
			*hi_l = seg[0] & ~(K)mask; // common prefix of the whole segment
		}
		__syncthreads();
		// A byte that overflows carries into its neighbour: the sum of all bytes then falls short of
		// n (every carry loses 255 or 256), which the prefix sums below notice -- no per-key check.
		constexpr int LB = 16 / (int)sizeof(K) * 4; // loads in flight per thread (branch-free, clamped)
		for (uint32_t i0 = 0; i0 < n; i0 += LB * kCountTh) {
			K kb[LB];
#pragma unroll
			for (int u = 0; u < LB; ++u) kb[u] = seg[min(i0 + u * kCountTh + tid, n - 1u)];
#pragma unroll
			for (int u = 0; u < LB; ++u) {
				if (i0 + u * kCountTh + tid < n) {
					const uint32_t v = (uint32_t)kb[u] & mask;
					atomicAdd(&cw[cw_at(v >> 2)], 1u << ((v & 3u) * 8u));
				}
			}
		}
		__syncthreads();
		const uint32_t nxt = *nexti;
		const K hi = *hi_l;
		// ---- exclusive prefix of the counters; thread t owns words [t*wpt, (t+1)*wpt), wpt <= 16
		// (wpt divides 32, so the owned words are consecutive in the padded layout too)
		const uint32_t wpt = nwords >= kCountTh ? nwords / kCountTh : 1;
		const uint32_t w0 = tid * wpt;
		const uint32_t *cwp = cw + cw_at(w0);
		uint32_t tot = 0;
		uint64_t nz = 0; // one bit per non-empty byte counter of this thread
		if (w0 < nwords) {
#pragma unroll 4
			for (uint32_t j = 0; j < wpt; ++j) {
				const uint32_t x = cwp[j];
				tot = __builtin_amdgcn_sad_u8(x, 0u, tot); // byte sum
				const uint32_t hb = (x | ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu)) & 0x80808080u; // bit 7 of each non-zero byte
				nz |= (uint64_t)((((hb >> 7) * 0x01020408u) >> 24) & 0xFu) << (4u * j);
			}
		}
		const uint32_t inc = wave_incl_scan(tot);
		if (lane == 63) wtot[w] = inc;
		__syncthreads();
		uint32_t pos = inc - tot, all = 0;
#pragma unroll 2
		for (uint32_t ww = 0; ww < kCountTh / 64; ++ww) {
			const uint32_t t = wtot[ww];
			if (ww < w) pos += t;
			all += t;
		}
		if (all != n) { // some value occurs > 255 times: hand the untouched segment on
			if (tid == 0) {
				if (n <= lds_cap) { // ... to the general LDS sort
					const uint32_t at = atomicAdd(&ctr->nfallback, 1u);
					fallback[fallback_base + at] = sg;
				} else {            // ... to the multi-workgroup counting sort (32-bit counters)
					const uint32_t at = big ? atomicAdd(&ctr->nbig, 1u) : 0xFFFFFFFFu;
					if (at < big_cap) big[at] = sg; else msd_note_error(ctr, 7u);
				}
			}
		} else {
			const uint32_t end = pos + tot;
			const K hi4 = hi | (K)(w0 * 4u);
			// ---- re-generate the sorted keys window by window through LDS (coalesced stores).
			// A thread's keys form one ascending run [pos, end); it resumes where the last window cut it.
			// Only the few waves whose runs fall into a window work on it, so the number of windows is
			// what this phase costs: besides the stage a window also uses the counter words of the
			// threads that are already done (everything below the first unfinished thread's words), so
			// the windows grow -- 3 instead of 5 for a 16 Ki-key segment.
			uint32_t c = 0;
			K cur = 0;
			K *cwk = reinterpret_cast<K *>(cw);
			auto window = [&](uint32_t wbeg, uint32_t wend) -> uint32_t {
				if (pos <= wend && wend < end) *tfree = tid; // the thread cut by this window's end
				while (pos < end && pos < wend) {
					if (c == 0) { // next non-empty counter (one exists because pos < end)
						const uint32_t i = (uint32_t)__ffsll((long long)nz) - 1u;
						nz &= nz - 1;
						c = (cwp[i >> 2] >> (8u * (i & 3u))) & 0xFFu;
						cur = hi4 + (K)i;
					}
					const uint32_t r = pos - wbeg;
					if (r < WS) stage[r]= cur; else cwk[r - WS] = cur;
					++pos;
					--c;
				}
				__syncthreads();
				const uint32_t tf = *tfree;
				for (uint32_t i = tid; i < wend - wbeg; i += kCountTh) seg[wbeg + i] = i < WS ? stage[i] : cwk[i - WS];
				__syncthreads();
				return (uint32_t)(cw_at(tf * wpt) * 4u / sizeof(K)); // keys that fit below thread tf's counters
			};
			uint32_t wbeg = 0, wend = WS < n ? WS : n;
			for (;;) {
				const uint32_t room = window(wbeg, wend);
				if (wend >= n) break;
				wbeg = wend;
				wend = wbeg + WS + room < n ? wbeg + WS + room : n;
			}
		}
		if (nxt >= nsegs) break;
		sg = segs[nxt];
		__syncthreads(); // nexti / hi_l / wtot are rewritten next
	}
}

} // namespace msd
#include "msd_count16.hpp"
#include "msd_merge16.hpp"
#include "msd_scatter16.hpp"
namespace msd {

// ------------------------------------------- counting leaf (keys or tuples)

// Leaf for segments that fit the LDS exchange buffers: ONE unstable counting pass over the top 14
// of the bits that actually vary in the segment.  16-bit counters (a segment has < 65536 elements,
// so they cannot overflow) are bumped by LDS fetch-adds whose return value is the element's rank
// among equal counted values; an in-place scan turns counts into positions; elements are scattered
// into LDS.  If more than 14 bits vary, the few groups of equal counted bits are put in order by
// whole-key insertion (their first element does it); a group longer than 48 sends the untouched
// segment to the general LDS sort.  Output is stored coalesced.
constexpr int kLeafCountBits = 14;
// counted bits of the leaf per type: u64 keys trade one bit for a larger exchange buffer (see Cfg)
template <typename K, typename V> struct LeafBits {
	static constexpr int value = (sizeof(K) == 8 && (!has_val<V>::value || Cfg<K, V>::SORT_TH < 1024)) ? 13 : kLeafCountBits;
};
template <typename K, typename V> struct LeafCountLds {
	static constexpr int CAP = Cfg<K, V>::SORT_TH * Cfg<K, V>::SORT_KPT;
	static constexpr size_t bytes = (size_t)CAP * (sizeof(K) + (has_val<V>::value ? 8 : 0)) +
					((size_t)1 << LeafBits<K, V>::value) * 2 + 192;
};

template <typename K, typename V>
__global__ __launch_bounds__((Cfg<K, V>::SORT_TH)) void leaf_count_sort_kernel(K *__restrict__ keys,
	uint64_t *__restrict__ vals, const Segment *__restrict__ segs, uint32_t nsegs,
	Segment *__restrict__ fallback, Counters *__restrict__ ctr, uint32_t *__restrict__ ticket,
	const uint32_t *__restrict__ nsegs_dev = nullptr)
{
	if (nsegs_dev) nsegs = *nsegs_dev; // (the list was written by the launch before this one)
	// Persistent workgroups (one per CU fits the LDS): segments by ticket; the next segment's elements are
	// loaded (branch-free) as soon as the current ones have left the registers for LDS, so its load
	// latency is hidden behind the fix-up and the write-back of the current one.
	using C = Cfg<K, V>;
	constexpr bool HV = has_val<V>::value;
	constexpr int TH = C::SORT_TH, KPT = C::SORT_KPT, CAP = TH * KPT, LB = LeafBits<K, V>::value;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	K *xk = reinterpret_cast<K *>(smem);
	uint64_t *xv = reinterpret_cast<uint64_t *>(smem + (size_t)CAP * sizeof(K));
	uint32_t *cw = reinterpret_cast<uint32_t *>(smem + (size_t)CAP * (sizeof(K) + (HV ? 8 : 0))); // 2 x 16-bit counters per word
	uint32_t *wtot = cw + ((size_t)1 << LB) / 2; // 16 wave totals, [16] flag, [17] next ticket, [18..19] tickets in hand
	K *s_or = reinterpret_cast<K *>(wtot + 32);              // [2] OR / AND of the keys
	if (blockIdx.x >= nsegs) return;
	const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	Segment sg = segs[blockIdx.x];
	K kr[KPT];
	uint64_t vr[HV ? KPT : 1];
	auto prefetch = [&](const Segment &g) {
		const uint32_t cnt = (uint32_t)g.count;
#pragma unroll
		for (int i = 0; i < KPT; ++i) { // past-the-end lanes re-read the last element (ignored later)
			if ((uint32_t)(i * TH) < cnt) { // (uniform: rows beyond the segment are not loaded at all)
				const uint32_t idx = min((uint32_t)(i * TH) + tid, cnt - 1u);
				kr[i] = keys[g.start + idx];
				if constexpr (HV) vr[i] = vals[g.start + idx];
			}
		}
	};
	prefetch(sg);
	if (tid == 0) wtot[19] = 0; // tickets in hand (only thread 0 uses them)
	MSD_STAMP_DECL(3);
	MSD_STAMP_START();
	for (;;) {
		MSD_STAMP(9); // loop
		MSD_STAMP_TICK(11);
		const uint32_t n = (uint32_t)sg.count;
		// (uniform) all open bits fit the counters: which of them vary does not matter, the OR/AND reduction and its
		// barrier are skipped (tuples whose rounds left 13 bits: a fifth of the segment's time)
#ifndef MSD_LEAF_NARROW // (0: experiments)
#define MSD_LEAF_NARROW 1
#endif
		const bool narrow = MSD_LEAF_NARROW && sg.bits <= (uint32_t)LB;
		K k_or = 0, k_and = ~(K)0;
		if (!narrow) {
#pragma unroll
			for (int i = 0; i < KPT; ++i) {
				if ((uint32_t)(i * TH) + tid < n) {
					k_or |= kr[i];
					k_and &= kr[i];
				}
			}
		}
		if (tid == 0) {
			s_or[0] = 0;
			s_or[1] = ~(K)0;
			wtot[16] = 0;
			// (tickets four at a time when there are many segments: 2^19 leaves of 2^11 tuples finish at 75 per
			// microsecond, about what a single word takes in fetch-adds)
			// (kept in LDS, [18] next / [19] how many: two more live registers would spill)
			if (wtot[19] == 0) {
				const uint32_t take = nsegs > 64u * gridDim.x ? 4u : 1u;
				wtot[18] = atomicAdd(ticket, take) + gridDim.x;
				wtot[19] = take;
			}
			wtot[17] = wtot[18];
			wtot[18] += 1;
			wtot[19] -= 1;
		}
		for (uint32_t j = tid; j < ((uint32_t)1 << LB) / 2; j += TH) cw[j] = 0;
		MSD_STAMP(0); // wait for the keys + OR/AND + clear
		__syncthreads();
		if (!narrow) {
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) {
				k_or |= __shfl_xor(k_or, o);
				k_and &= __shfl_xor(k_and, o);
			}
			if (lane == 0) {
				if constexpr (sizeof(K) == 4) {
					atomicOr(reinterpret_cast<unsigned int *>(&s_or[0]), (unsigned int)k_or);
					atomicAnd(reinterpret_cast<unsigned int *>(&s_or[1]), (unsigned int)k_and);
				} else {
					atomicOr(reinterpret_cast<unsigned long long *>(&s_or[0]), (unsigned long long)k_or);
					atomicAnd(reinterpret_cast<unsigned long long *>(&s_or[1]), (unsigned long long)k_and);
				}
			}
			__syncthreads();
		}
		MSD_STAMP(1); // B + merge + B
		const uint32_t nxt = wtot[17];
		// the next segment's descriptor travels during the counting phases (loaded where its elements are prefetched,
		// its whole memory latency would sit in front of that prefetch)
		const Segment nraw = segs[nxt < nsegs ? nxt : blockIdx.x];
		const K openmask = sg.bits >= sizeof(K) * 8 ? ~(K)0 : (((K)1 << sg.bits) - 1);
		const K vopen = narrow ? openmask : (s_or[0] ^ s_or[1]) & openmask;
		Segment nsg = sg;
		bool fetched = false;
		if (vopen != 0) { // (uniform) otherwise constant on the open bits: already sorted
			const uint32_t nbits = (uint32_t)(64 - __builtin_clzll((unsigned long long)vopen));
			const uint32_t shift = nbits > (uint32_t)LB ? nbits - LB : 0;
			const uint32_t mask = (1u << (nbits - shift)) - 1u;
			uint32_t rk[KPT];
#pragma unroll
			for (int i = 0; i < KPT; ++i) {
				rk[i] = 0;
				if ((uint32_t)(i * TH) + tid < n) {
					const uint32_t v = (uint32_t)(kr[i] >> shift) & mask, sh = 16u * (v & 1u);
					rk[i] = (atomicAdd(&cw[v >> 1], 1u << sh) >> sh) & 0xFFFFu;
				}
			}
			MSD_STAMP(2); // fetch-adds
			__syncthreads();
			// counts -> exclusive positions, in place; thread t owns 8 words
			constexpr uint32_t WPT = (((uint32_t)1 << LB) / 2) / TH;
			static_assert(WPT >= 1, "counter words per thread");
			const uint32_t w0 = tid * WPT;
			uint32_t tot = 0;
#pragma unroll
			for (uint32_t j = 0; j < WPT; ++j) {
				const uint32_t x = cw[w0 + j];
				tot += (x & 0xFFFFu) + (x >> 16);
			}
			const uint32_t inc = wave_incl_scan(tot);
			if (lane == 63) wtot[w] = inc;
			__syncthreads();
			uint32_t run = inc - tot;
			for (uint32_t ww = 0; ww < w; ++ww) run += wtot[ww];
#pragma unroll
			for (uint32_t j = 0; j < WPT; ++j) {
				const uint32_t x = cw[w0 + j];
				const uint32_t lo = x & 0xFFFFu, hi = x >> 16;
				cw[w0 + j] = run | ((run + lo) << 16);
				run += lo + hi;
			}
			MSD_STAMP(3); // B + sums + scan + B + prefix
			__syncthreads();
#pragma unroll
			for (int i = 0; i < KPT; ++i) {
				if ((uint32_t)(i * TH) + tid < n) {
					const uint32_t v = (uint32_t)(kr[i] >> shift) & mask;
					const uint32_t p = ((cw[v >> 1] >> (16u * (v & 1u))) & 0xFFFFu) + rk[i];
					xk[p] = kr[i];
					if constexpr (HV) xv[p] = vr[i];
				}
			}
			MSD_STAMP(4); // B + scatter into LDS
			// the registers are free: the next segment's elements start travelling
			if (nxt < nsegs) {
				nsg = nraw;
				prefetch(nsg);
				fetched = true;
			}
			MSD_STAMP(5); // descriptor + prefetch issue
			__syncthreads();
			bool bad = false;
			const K lowmask = shift ? ((K)1 << shift) - 1 : (K)0;
			const bool groups = shift && (vopen & lowmask) != 0; // more bits vary than were counted
			constexpr uint32_t kMaxGroup = 48;
			// (same group <=> the keys agree above `shift` <=> their XOR is below 2^shift: one compare against a uniform bound
			// instead of two shifts -- 64-bit shifts run at a quarter of the rate)
			const K glim = (K)1 << shift;
			if (groups) {
				// a group (elements equal on the counted bits; contiguous by now) longer than kMaxGroup sends the untouched
				// segment to the general LDS sort: with the groups in order, two elements kMaxGroup apart share a group
				// only if the group is longer than that
				bool too_long = false;
				for (uint32_t i = tid; i + kMaxGroup < n; i += TH)
					if ((K)(xk[i] ^ xk[i + kMaxGroup]) < glim) too_long = true;
				if (too_long) wtot[16] = 1;
				__syncthreads();
				bad = wtot[16] != 0;
				if (bad && tid == 0) fallback[atomicAdd(&ctr->nfallback, 1u)] = sg; // (nothing has been written back)
			}
			MSD_STAMP(6); // B + long-group check
			if (!bad && groups) {
				// Write-back with the groups put in order on the way: every element finds its own place inside its group --
				// first slot of the group + the members with a smaller key (or an equal key further left) -- by scanning
				// its few neighbours in LDS, and goes straight to that place in the array.  LDS is only read, so no element
				// waits for another (one thread per group running an insertion sort through dependent LDS round trips was
				// half of this kernel's time for tuples and four fifths for u64 keys, profiles/r02_stamps_leaf_before.json).
				// (the three neighbours on either side are read at once, clamped at the segment's ends -- one LDS round trip
				// covers nearly every group; only longer groups continue element by element)
				constexpr uint32_t WIN = 3;
#pragma unroll 1
				for (uint32_t idx = tid; idx < n; idx += TH) {
					const K me = xk[idx];
					K lk[WIN], rkk[WIN];
#pragma unroll
					for (uint32_t d = 0; d < WIN; ++d) {
						lk[d] = xk[idx > d ? idx - d - 1 : 0u];
						rkk[d] = xk[min(idx + d + 1, n - 1u)];
					}
					uint32_t left = 0, before = 0, right = 0;
					bool ml = true, mr = true;
#pragma unroll
					for (uint32_t d = 0; d < WIN; ++d) {
						ml = ml && idx > d && (K)(lk[d] ^ me) < glim;      // members to the left: those <= me come first
						before += ml && lk[d] <= me ? 1u : 0u;
						left += ml ? 1u : 0u;
						mr = mr && idx + d + 1 < n && (K)(rkk[d] ^ me) < glim; // members to the right: those < me come first
						before += mr && rkk[d] < me ? 1u : 0u;
						right += mr ? 1u : 0u;
					}
					if (ml) { // the group goes on beyond the window
						while (idx > left) {
							const K o = xk[idx - left - 1];
							if ((K)(o ^ me) >= glim) break;
							before += o <= me ? 1u : 0u;
							++left;
						}
					}
					if (mr) {
						for (uint32_t e = idx + right + 1; e < n; ++e) {
							const K o = xk[e];
							if ((K)(o ^ me) >= glim) break;
							before += o < me ? 1u : 0u;
						}
					}
					const uint64_t at = sg.start + idx - left + before;
					keys[at] = me;
					if constexpr (HV) vals[at] = xv[idx];
				}
			} else if (!bad) {
				for (uint32_t idx = tid; idx < n; idx += TH) {
					keys[sg.start + idx] = xk[idx];
					if constexpr (HV) vals[sg.start + idx] = xv[idx];
				}
			}
		}
		MSD_STAMP(7); // write-back
		if (nxt >= nsegs) break;
		if (!fetched) {
			nsg = nraw;
			prefetch(nsg);
		}
		sg = nsg;
		__syncthreads(); // LDS (exchange buffers, counters, flags) is reused
	}
	MSD_STAMP_FLUSH(TH / 64);
}

} // namespace msd
#include "msd_leaf17.hpp"
#include "msd_bigcount.hpp"
namespace msd {

// One launch clears what a round starts from (a handful of separate memsets cost a few microseconds of
// idle GPU each): the per-round counters, the per-parent plans of a direct round, the scan's tile state.
__global__ __launch_bounds__(256) void round_init_kernel(Counters *__restrict__ ctr, uint32_t *__restrict__ plan_words,
	uint64_t nplan_words, unsigned long long *__restrict__ scan_state, uint64_t ntiles, uint32_t *__restrict__ scan_ctr)
{
	const uint64_t i0 = (uint64_t)blockIdx.x * 256 + threadIdx.x, step = (uint64_t)gridDim.x * 256;
	if (i0 == 0) {
		ctr->nholes = 0;
		ctr->hole_cursor = 0;
		ctr->next_parents = 0;
		ctr->nevict = 0;
		ctr->direct_uneven = 0;
		ctr->next_max = 0;
		ctr->rp_children = 0;
		ctr->nhot = 0;
		scan_ctr[0] = scan_ctr[1] = scan_ctr[2] = scan_ctr[3] = 0;
	}
	for (uint64_t i = i0; i < nplan_words; i += step) plan_words[i] = 0;
	for (uint64_t i = i0; i < ntiles; i += step) scan_state[i] = 0;
}

// ------------------------------------------------------- LDS segment sort

// One workgroup sorts one segment of <= SORT_TH*SORT_KPT elements on its low
// `bits` bits: stable LSD passes of 8 bits inside LDS.  Keys live in registers in
// wave-striped order; the per-wave rank of a key among equal digits comes from a
// match-any built with 8 wavefront ballots and a popcount of the lower lanes, plus
// a per-wave running digit counter in LDS (read by all peers, bumped by the lowest).
template <typename K, typename V>
__global__ __launch_bounds__((Cfg<K, V>::SORT_TH)) void lds_sort_kernel(K *__restrict__ keys,
	uint64_t *__restrict__ vals, const Segment *__restrict__ segs, uint32_t nsegs,
	const uint32_t *__restrict__ nsegs_dev)
{
	using C = Cfg<K, V>;
	constexpr bool HV = has_val<V>::value;
	constexpr int TH = C::SORT_TH, KPT = C::SORT_KPT, NW = TH / 64, CAP = TH * KPT;
	if (nsegs_dev) nsegs = *nsegs_dev; // list filled by an earlier kernel of this stream
	for (uint32_t sgi = blockIdx.x; sgi < nsegs; sgi += gridDim.x) { // (grid may be smaller than the list)
	__syncthreads();
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	K *xk = reinterpret_cast<K *>(smem);
	uint64_t *xv = reinterpret_cast<uint64_t *>(smem + (size_t)CAP * sizeof(K));
	uint32_t *wcnt = reinterpret_cast<uint32_t *>(smem + (size_t)CAP * sizeof(K) + (HV ? (size_t)CAP * 8 : 0));
	uint32_t *dbase = wcnt + NW * kP; // 256
	uint32_t *tmp = dbase + kP;       // 8
	K *s_or = reinterpret_cast<K *>(tmp + 8); // [2]: OR and AND of the keys (varying-bit detection)

	const Segment sg = segs[sgi];
	const uint32_t n = (uint32_t)sg.count;
	const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const int nitems = (int)((n + TH - 1) / TH);         // items per thread actually in use
	const uint32_t wbase = w * (uint32_t)nitems * 64;    // wave-striped layout over [0, nitems*TH)
	const uint64_t lt_mask = (1ull << lane) - 1ull;

	K kr[KPT];
	uint64_t vr[HV ? KPT : 1];
	K k_or = 0, k_and = ~(K)0;
#pragma unroll
	for (int i = 0; i < KPT; ++i) {
		kr[i] = ~(K)0;
		if constexpr (HV) vr[i] = 0;
		if (i < nitems) {
			const uint32_t idx = wbase + i * 64 + lane;
			if (idx < n) {
				kr[i] = keys[sg.start + idx];
				if constexpr (HV) vr[i] = vals[sg.start + idx];
				k_or |= kr[i];
				k_and &= kr[i];
			}
		}
	}
	// which bits vary inside the segment (passes over constant digits are skipped)
	if (tid == 0) { s_or[0] = 0; s_or[1] = ~(K)0; }
	__syncthreads();
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		k_or |= __shfl_xor(k_or, o);
		k_and &= __shfl_xor(k_and, o);
	}
	if (lane == 0) {
		if constexpr (sizeof(K) == 4) {
			atomicOr(reinterpret_cast<unsigned int *>(&s_or[0]), (unsigned int)k_or);
			atomicAnd(reinterpret_cast<unsigned int *>(&s_or[1]), (unsigned int)k_and);
		} else {
			atomicOr(reinterpret_cast<unsigned long long *>(&s_or[0]), (unsigned long long)k_or);
			atomicAnd(reinterpret_cast<unsigned long long *>(&s_or[1]), (unsigned long long)k_and);
		}
	}
	__syncthreads();
	const K vary = s_or[0] ^ s_or[1];
	__syncthreads();

	uint32_t *mycnt = wcnt + w * kP;
	bool in_lds = false;
	// With many open bits (u64 keys) few of them are needed to tell <= CAP keys apart: sort on the
	// TOP 16 open bits only (2 stable passes), then put the few groups of equal top bits in order
	// by comparing whole keys.  Falls through to the remaining low passes if a group is too long.
	const K openmask = sg.bits >= sizeof(K) * 8 ? ~(K)0 : (((K)1 << sg.bits) - 1);
	const K vopen = vary & openmask;
	// bits above the highest varying one need no pass at all
	const uint32_t nbits = vopen ? (uint32_t)(64 - __builtin_clzll((unsigned long long)vopen)) : 0u;
	const uint32_t first_shift = nbits > 24 ? nbits - 16 : 0;
	bool msd_mode = first_shift != 0;
	for (uint32_t shift = first_shift; shift < nbits; shift += 8) {
		const uint32_t width = nbits - shift < 8 ? nbits - shift : 8;
		const uint32_t mask = (1u << width) - 1u;
		if (((uint32_t)(vary >> shift) & mask) == 0) continue; // digit constant over the segment
		for (uint32_t j = tid; j < NW * kP; j += TH) wcnt[j] = 0;
		__syncthreads();
		uint32_t rk[(KPT + 1) / 2]; // two 16-bit ranks per register
#pragma unroll
		for (int i = 0; i < KPT; ++i) {
			if (i < nitems) {
				const uint32_t d = digit_of(kr[i], shift, mask);
				uint32_t plo = ~0u, phi = ~0u; // match-any: lanes of this wave with the same digit
#pragma unroll
				for (int b = 0; b < 8; ++b) {
					const uint32_t bit = (d >> b) & 1u;
					const uint64_t m = __ballot(bit != 0);
					const uint32_t ext = bit - 1u; // 0 when the bit is set, ~0 otherwise
					plo &= (uint32_t)m ^ ext;
					phi &= (uint32_t)(m >> 32) ^ ext;
				}
				const uint32_t below = __popc(plo & (uint32_t)lt_mask) + __popc(phi & (uint32_t)(lt_mask >> 32));
				const uint32_t old = mycnt[d];                          // same value for all peers
				if (below == 0) mycnt[d] = old + __popc(plo) + __popc(phi); // lowest peer bumps the counter
				const uint32_t r = old + below;                         // < nitems*64 <= 1536
				if (i & 1) rk[i / 2] |= r << 16; else rk[i / 2] = r;
			}
			__builtin_amdgcn_sched_barrier(0); // keep one item's ballots live at a time
		}
		__syncthreads();
		// exclusive offsets: digit-major, wave-minor
		uint32_t tot_d = 0;
		if (tid < kP) {
#pragma unroll
			for (int ww = 0; ww < NW; ++ww) {
				const uint32_t t = wcnt[ww * kP + tid];
				wcnt[ww * kP + tid] = tot_d;
				tot_d += t;
			}
		}
		uint32_t gt;
		const uint32_t ex = block_excl_scan256(tot_d, tmp, gt);
		if (tid < kP) dbase[tid] = ex;
		__syncthreads();
#pragma unroll
		for (int i = 0; i < KPT; ++i) {
			if (i < nitems) {
				const uint32_t d = digit_of(kr[i], shift, mask);
				const uint32_t r = (i & 1) ? rk[i / 2] >> 16 : rk[i / 2] & 0xFFFFu;
				const uint32_t p = dbase[d] + mycnt[d] + r;
				xk[p] = kr[i];
				if constexpr (HV) xv[p] = vr[i];
			}
		}
		__syncthreads();
		in_lds = true;
		// is there another pass with a varying digit?
		bool more = false;
		for (uint32_t s2 = shift + 8; s2 < nbits; s2 += 8) {
			const uint32_t w2 = nbits - s2 < 8 ? nbits - s2 : 8;
			if (((uint32_t)(vary >> s2) & ((1u << w2) - 1u)) != 0) more = true;
		}
		if (!more && msd_mode) {
			// ---- groups of equal top bits: the first element of each group orders it by whole keys
			const K lowmask = ((K)1 << first_shift) - 1;
			if (tid == 0) tmp[0] = 0;
			__syncthreads();
			bool too_long = false;
			if ((vary & lowmask) != 0) {
				for (uint32_t i = tid; i < n; i += TH) {
					const K hi = xk[i] >> first_shift;
					if ((i == 0 || (xk[i - 1] >> first_shift) != hi) && i + 1 < n && (xk[i + 1] >> first_shift) == hi) {
						uint32_t e = i + 2;
						while (e < n && (xk[e] >> first_shift) == hi) ++e;
						if (e - i > 48) {
							too_long = true;
						} else {
							for (uint32_t a = i + 1; a < e; ++a) { // insertion sort of [i, e)
								const K ka = xk[a];
								uint64_t va = 0;
								if constexpr (HV) va = xv[a];
								uint32_t b = a;
								while (b > i && xk[b - 1] > ka) {
									xk[b] = xk[b - 1];
									if constexpr (HV) xv[b] = xv[b - 1];
									--b;
								}
								xk[b] = ka;
								if constexpr (HV) xv[b] = va;
							}
						}
					}
				}
			}
			if (too_long) tmp[0] = 1;
			__syncthreads();
			if (tmp[0]) { // rare: redo as a plain LSD sort over all open bits, starting from the current order
				msd_mode = false;
				more = true;
				shift = (uint32_t)0 - 8u; // the loop increment brings it to 0
			}
			__syncthreads();
		}
		if (!more) break;
#pragma unroll
		for (int i = 0; i < KPT; ++i) {
			if (i < nitems) {
				kr[i] = xk[wbase + i * 64 + lane];
				if constexpr (HV) vr[i] = xv[wbase + i * 64 + lane];
			}
		}
		__syncthreads();
		in_lds = false;
	}
	if (in_lds) {
		for (uint32_t idx = tid; idx < n; idx += TH) {
			keys[sg.start + idx] = xk[idx];
			if constexpr (HV) vals[sg.start + idx] = xv[idx];
		}
	}
	// (no varying digit at all: the segment is constant on the bits in question, nothing to do)
	} // segment loop
}

template <typename K, typename V> struct SortLds {
	using C = Cfg<K, V>;
	static constexpr int CAP = C::SORT_TH * C::SORT_KPT;
	static constexpr size_t bytes = (size_t)CAP * sizeof(K) + (has_val<V>::value ? (size_t)CAP * 8 : 0) +
					(size_t)((C::SORT_TH / 64) * kP + kP + 8) * 4 + 16;
};

// ------------------------------------------------------------- histogram

template <typename K>
__global__ __launch_bounds__(256) void histogram_kernel(const K *__restrict__ keys, uint64_t n,
	uint32_t shift, uint32_t radix_bits, unsigned long long *__restrict__ count)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint32_t *h = reinterpret_cast<uint32_t *>(smem);
	const uint32_t parts = 1u << radix_bits, mask = parts - 1u;
	for (uint32_t j = threadIdx.x; j < parts; j += 256) h[j] = 0;
	__syncthreads();
	constexpr int VEC = Vec16<K>::N;
	const uint64_t nvec = n / VEC;
	const uint64_t stride = (uint64_t)gridDim.x * 256;
	for (uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += stride) {
		if constexpr (sizeof(K) == 4) {
			const uint4 q = reinterpret_cast<const uint4 *>(keys)[v];
			atomicAdd(&h[(q.x >> shift) & mask], 1u);
			atomicAdd(&h[(q.y >> shift) & mask], 1u);
			atomicAdd(&h[(q.z >> shift) & mask], 1u);
			atomicAdd(&h[(q.w >> shift) & mask], 1u);
		} else {
			const ulonglong2 q = reinterpret_cast<const ulonglong2 *>(keys)[v];
			atomicAdd(&h[(uint32_t)(q.x >> shift) & mask], 1u);
			atomicAdd(&h[(uint32_t)(q.y >> shift) & mask], 1u);
		}
	}
	if (blockIdx.x == 0)
		for (uint64_t i = nvec * VEC + threadIdx.x; i < n; i += 256)
			atomicAdd(&h[(uint32_t)(keys[i] >> shift) & mask], 1u);
	__syncthreads();
	for (uint32_t j = threadIdx.x; j < parts; j += 256)
		if (h[j]) atomicAdd(&count[j], (unsigned long long)h[j]);
}

// ------------------------------------ device-wide scan (decoupled look-back)

constexpr int kScanTh = 256, kScanIpt = 8, kScanTile = kScanTh * kScanIpt;
constexpr uint64_t kFlagAgg = 1ull << 62, kFlagPre = 2ull << 62, kFlagMask = 3ull << 62;

// tile_state[ntiles] and tile_counter[1] must be zero on entry.
__global__ __launch_bounds__(kScanTh) void scan_lookback_kernel(const uint64_t *__restrict__ in,
	uint64_t *__restrict__ out, uint64_t n, unsigned long long *__restrict__ tile_state,
	uint32_t *__restrict__ tile_counter, uint32_t *__restrict__ err)
{
	__shared__ uint64_t tmp[8];
	__shared__ uint32_t s_tile;
	__shared__ uint64_t s_prefix;
	const uint32_t tid = threadIdx.x;
	if (tid == 0) s_tile = atomicAdd(tile_counter, 1u); // dynamic id: predecessors have started
	__syncthreads();
	const uint32_t tile = s_tile;
	const uint64_t base = (uint64_t)tile * kScanTile + (uint64_t)tid * kScanIpt;
	uint64_t v[kScanIpt], sum = 0;
#pragma unroll
	for (int i = 0; i < kScanIpt; ++i) {
		v[i] = base + i < n ? in[base + i] : 0;
		sum += v[i];
	}
	// block scan of the per-thread sums (256 threads)
	uint64_t agg;
	const uint64_t tex = block_excl_scan256_64(sum, tmp, agg);
	if (tid == 0) {
		uint64_t prefix = 0;
		if (tile == 0) {
			__hip_atomic_store(&tile_state[0], kFlagPre | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		} else {
			__hip_atomic_store(&tile_state[tile], kFlagAgg | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			uint32_t t = tile - 1;
			uint32_t spins = 0;
			for (;;) {
				const uint64_t s = __hip_atomic_load(&tile_state[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const uint64_t f = s & kFlagMask;
				if (f == 0) {
					if (++spins > (1u << 26)) { // bounded: report instead of hanging
						atomicAdd(err, 1u);
						break;
					}
					__builtin_amdgcn_s_sleep(1);
					continue;
				}
				prefix += s & ~kFlagMask;
				if (f == kFlagPre) break;
				--t;
			}
			__hip_atomic_store(&tile_state[tile], kFlagPre | (prefix + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		s_prefix = prefix;
	}
	__syncthreads();
	uint64_t run = s_prefix + tex;
#pragma unroll
	for (int i = 0; i < kScanIpt; ++i) {
		if (base + i < n) out[base + i] = run;
		run += v[i];
	}
}

// ------------------------------------------------- run gather (multi-GPU: source-major -> bucket-major)

// After the all-to-all a rank holds, per source rank, that source's buckets in order; the local sort wants every bucket
// contiguous.  One launch copies all runs (nruns of them, any lengths) to their places in a second buffer.
struct GatherRun {
	uint64_t src, dst, len;  // element offsets and length
	uint32_t first_chunk;    // index of the run's first chunk of kGatherChunk elements
	uint32_t pad;
};
#ifndef MSD_GATHER_CHUNK
#define MSD_GATHER_CHUNK 2048
#endif
constexpr uint32_t kGatherChunk = MSD_GATHER_CHUNK;

template <typename K>
__global__ __launch_bounds__(256) void gather_runs_kernel(K *__restrict__ dst, const K *__restrict__ src,
	const GatherRun *__restrict__ runs, uint32_t nruns, uint32_t nchunks, const uint32_t *__restrict__ coarse)
{
	constexpr int VEC = 16 / (int)sizeof(K);
	for (uint32_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
		// coarse[i] = the run of chunk 64 i: a workgroup lives for a few microseconds, a search over all runs from
		// scratch (eleven dependent look-ups for 2048 runs) would be a good part of them
		uint32_t lo = coarse[c >> 6], hi = min(coarse[(c >> 6) + 1] + 1u, nruns); // runs[lo].first_chunk <= c < runs[hi].first_chunk
		while (hi - lo > 1) {
			const uint32_t mid = (lo + hi) >> 1;
			if (runs[mid].first_chunk <= c) lo = mid; else hi = mid;
		}
		const GatherRun r = runs[lo];
		const uint64_t off = (uint64_t)(c - r.first_chunk) * kGatherChunk;
		const uint32_t len = (uint32_t)(r.len - off < (uint64_t)kGatherChunk ? r.len - off : (uint64_t)kGatherChunk);
		K *d = dst + r.dst + off;
		const K *s = src + r.src + off;
		// 16-byte stores on the destination's grid
		const uint32_t lead = min(len, (uint32_t)((VEC - (int)(((uintptr_t)d / sizeof(K)) % VEC)) % VEC));
		const uint32_t nvec = (len - lead) / VEC, tail0 = lead + nvec * VEC;
		if (threadIdx.x < lead) d[threadIdx.x] = s[threadIdx.x];
		if (tail0 + threadIdx.x < len) d[tail0 + threadIdx.x] = s[tail0 + threadIdx.x];
		// 16-byte loads on the SOURCE's grid too (an unaligned 16-byte load is split by the hardware: 3.6 TB/s): vector q
		// of the output is words k .. k+3 of the aligned source vectors q and q + 1, k = the run's misalignment in
		// 4-byte words (uniform); the next vector's words come from the next lane (the wave's last lane loads them)
		const uint32_t kw = (uint32_t)(((uintptr_t)(s + lead) & 15u) >> 2);
		const u32x4 *sa = reinterpret_cast<const u32x4 *>(reinterpret_cast<const unsigned char *>(s + lead) - 4u * kw);
		u32x4 *dv = reinterpret_cast<u32x4 *>(d + lead);
		const bool last_lane = (threadIdx.x & 63u) == 63u;
#ifndef MSD_GATHER_NF
#define MSD_GATHER_NF 2
#endif
		constexpr int NF = MSD_GATHER_NF; // loads in flight per thread
		for (uint32_t q0 = 0; q0 < nvec; q0 += NF * 256) { // (every lane stays in the loop: its neighbour needs its words)
			const uint32_t q = q0 + threadIdx.x;
			u32x4 a[NF], nx[NF];
#pragma unroll
			for (int u = 0; u < NF; ++u) { // (beyond the chunk: its last vector again, ignored; with kw > 0 vector nvec holds the run's last words)
				const uint32_t qq = min(q + u * 256u, kw ? nvec : nvec - 1u);
				a[u] = sa[qq];
				if (kw && last_lane) nx[u] = sa[min(qq + 1u, nvec)];
			}
#pragma unroll
			for (int u = 0; u < NF; ++u) {
				u32x4 o = a[u];
				if (kw) { // (uniform)
					// (the shuffles run with every lane active -- a lane that sat out would hand its neighbour zero)
					const uint32_t sx = (uint32_t)__shfl_down((int)a[u].x, 1), sy = (uint32_t)__shfl_down((int)a[u].y, 1),
						       sz = (uint32_t)__shfl_down((int)a[u].z, 1);
					const uint32_t bx = last_lane ? nx[u].x : sx, by = last_lane ? nx[u].y : sy, bz = last_lane ? nx[u].z : sz;
					o = kw == 1 ? u32x4{ a[u].y, a[u].z, a[u].w, bx } : kw == 2 ? u32x4{ a[u].z, a[u].w, bx, by } : u32x4{ a[u].w, bx, by, bz };
				}
				if (q + u * 256u < nvec) dv[q + u * 256u] = o;
			}
		}
	}
}

// ------------------------------------------------- varying-bit reduction

// OR and AND over keys[0], keys[stride], keys[2*stride], ...: bits where OR and AND differ
// vary somewhere in the (sampled) input.  res[0] |= OR, res[1] &= AND (res preset to 0 / ~0).
template <typename K>
__global__ __launch_bounds__(256) void vary_kernel(const K *__restrict__ keys, uint64_t n, uint64_t stride,
	unsigned long long *__restrict__ res)
{
	K o = 0, a = ~(K)0;
	const uint64_t cnt = (n + stride - 1) / stride;
	const uint64_t step = (uint64_t)gridDim.x * 256, t0 = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	if (stride == 1) { // every key: 16-byte nontemporal loads on the array's 16-byte grid, four in flight (read-only stream)
		constexpr int VEC = 16 / (int)sizeof(K);
		const uint64_t lead = min(n, (uint64_t)((VEC - (int)(((uintptr_t)keys / sizeof(K)) % VEC)) % VEC));
		const uint64_t nvec = (n - lead) / VEC;
		const u32x4 *v = reinterpret_cast<const u32x4 *>(keys + lead);
		auto take = [&](const u32x4 &q) {
			if constexpr (sizeof(K) == 4) {
				o |= (K)(q.x | q.y | q.z | q.w);
				a &= (K)(q.x & q.y & q.z & q.w);
			} else {
				const K k0 = (K)q.x | ((K)q.y << 32), k1 = (K)q.z | ((K)q.w << 32);
				o |= k0 | k1;
				a &= k0 & k1;
			}
		};
		uint64_t i = t0;
		for (; i + 3 * step < nvec; i += 4 * step) {
			const u32x4 q0 = __builtin_nontemporal_load(v + i), q1 = __builtin_nontemporal_load(v + i + step);
			const u32x4 q2 = __builtin_nontemporal_load(v + i + 2 * step), q3 = __builtin_nontemporal_load(v + i + 3 * step);
			take(q0); take(q1); take(q2); take(q3);
		}
		for (; i < nvec; i += step) take(__builtin_nontemporal_load(v + i));
		const uint64_t tail0 = lead + nvec * VEC; // the single keys at both ends
		if (t0 < lead) {
			o |= keys[t0];
			a &= keys[t0];
		}
		if (tail0 + t0 < n) {
			o |= keys[tail0 + t0];
			a &= keys[tail0 + t0];
		}
	} else
		for (uint64_t i = t0; i < cnt; i += step) {
			const K k = keys[i * stride];
			o |= k;
			a &= k;
		}
#pragma unroll
	for (int s = 32; s > 0; s >>= 1) {
		o |= __shfl_xor(o, s);
		a &= __shfl_xor(a, s);
	}
	if (lane_id() == 0) {
		atomicOr(&res[0], (unsigned long long)o);
		atomicAnd(&res[1], (unsigned long long)a | (sizeof(K) == 4 ? 0xFFFFFFFF00000000ull : 0ull));
	}
}

// ------------------------------------------------------------- verifier

struct CheckResult {
	unsigned long long violations, sum, xr;
};

template <typename K>
__global__ __launch_bounds__(256) void check_kernel(const K *__restrict__ keys, const uint64_t *__restrict__ rids,
	uint64_t n, CheckResult *__restrict__ res)
{
	unsigned long long bad = 0, sum = 0, xr = 0;
	const uint64_t stride = (uint64_t)gridDim.x * 256;
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
		const K k = keys[i];
		if (i > 0 && k < keys[i - 1]) ++bad;
		if (rids && rids[i] != (uint64_t)k) ++bad;
		sum += (unsigned long long)k;
		xr ^= (unsigned long long)k;
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		bad += __shfl_xor(bad, o);
		sum += __shfl_xor(sum, o);
		xr ^= __shfl_xor(xr, o);
	}
	if (lane_id() == 0) {
		if (bad) atomicAdd(&res->violations, bad);
		atomicAdd(&res->sum, sum);
		atomicXor(&res->xr, xr);
	}
}

// ------------------------------------------------------------- generators

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
	x += 0x9E3779B97F4A7C15ull;
	uint64_t z = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

__global__ void gen_uniform_u32_kernel(uint32_t *out, uint64_t n, uint64_t seed)
{
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
		out[i] = (uint32_t)(splitmix64(seed + i) >> 32);
}
__global__ void gen_uniform_u64_kernel(uint64_t *out, uint64_t n, uint64_t seed, int shr)
{
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
		out[i] = splitmix64(seed + i) >> shr;
}
__global__ void gen_zipf_u32_kernel(uint32_t *out, uint64_t n, uint64_t seed)
{
	const double U = 4294967296.0;
	const double lnU1 = log(U + 1.0);
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const double u = (double)(splitmix64(seed + i) >> 11) * 0x1.0p-53;
		double r = floor(exp(u * lnU1));
		r = r < 1.0 ? 1.0 : (r > U ? U : r);
		out[i] = (uint32_t)((uint64_t)r - 1);
	}
}
__global__ void gen_iota_u64_kernel(uint64_t *out, uint64_t n, uint64_t first)
{
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
		out[i] = first + i;
}

// `distinct` different key values, each about n / distinct times: value j = the top half of splitmix64(j ^ salt),
// j = splitmix64(seed + i) mod distinct -- duplicates whose digits are still evenly spread (the reference's
// comb/insertion sorts degrade on duplicates; the counting leaves here must not)
__global__ void gen_dup_u32_kernel(uint32_t *out, uint64_t n, uint64_t seed, uint64_t distinct)
{
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
		out[i] = (uint32_t)(splitmix64((splitmix64(seed + i) % distinct) ^ 0xD0B1E5ull) >> 32);
}

// The reference's own generator, MT19937-64 (src/rand.c:47-86: rand64_init / rand64_next), as a stream on the
// device: out[i] = the i-th rand64_next() after rand64_init(seed), shifted right by shr.  The twist of a block of
// 312 words has two parallel halves (:65-78): words [0, 156) need only old words, words [156, 312) the new
// words [0, 156) and (word 311) the new word 0.  One workgroup walks the stream (the recurrence is sequential
// from block to block); it is a cross-checking aid, not a bulk generator.
__global__ __launch_bounds__(320) void gen_mt19937_64_kernel(uint64_t *out, uint64_t n, uint64_t seed, int shr)
{
	__shared__ uint64_t st[313];
	const uint32_t t = threadIdx.x;
	if (t == 0) { // rand64_init, src/rand.c:47-58
		st[0] = seed;
		for (uint32_t i = 0; i != 311; ++i) st[i + 1] = 6364136223846793005ull * (st[i] ^ (st[i] >> 62)) + i + 1;
	}
	__syncthreads();
	auto twist = [](uint64_t a, uint64_t b, uint64_t far) -> uint64_t {
		const uint64_t x = (a & 0xffffffff80000000ull) | (b & 0x7fffffffull);
		return far ^ (x >> 1) ^ (0xb5026f5aa96619e9ull & (0ull - (x & 1ull)));
	};
	for (uint64_t base = 0; base < n; base += 312) {
		uint64_t v = 0;
		if (t < 156) v = twist(st[t], st[t + 1], st[t + 156]);
		__syncthreads();
		if (t < 156) st[t] = v;
		__syncthreads();
		if (t >= 156 && t < 312) v = twist(st[t], t == 311 ? st[0] : st[t + 1], st[t - 156]);
		__syncthreads();
		if (t >= 156 && t < 312) st[t] = v;
		__syncthreads();
		if (t < 312 && base + t < n) { // tempering, src/rand.c:80-85
			uint64_t x = st[t];
			x ^= (x >> 29) & 0x5555555555555555ull;
			x ^= (x << 17) & 0x71d67fffeda60000ull;
			x ^= (x << 37) & 0xfff7eee000000000ull;
			x ^= (x >> 43);
			out[base + t] = x >> shr;
		}
	}
}

// ------------------------------------------------------- splitter service

// Random sample of an (unsorted) array: out[i] = keys[mulhi(rand64, n)], as the reference draws its sample
// (src/msb_64.c:1511-1521, mulhi :178-186) -- with a counter-based generator in place of its MT19937 stream
// (which it seeds from an uninitialised field).
template <typename K>
__global__ __launch_bounds__(256) void sample_kernel(const K *__restrict__ keys, uint64_t n, uint64_t m, uint64_t seed,
	K *__restrict__ out)
{
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride)
		out[i] = keys[__umul64hi(splitmix64(seed + i), n)];
}

// parts-1 equi-depth delimiters of a sorted sample with the reference's duplicate rule (extract_delimiters,
// src/msb_64.c:1304-1322): pick sample[(uint64)(m / parts * (i+1) - 0.001)]; if more repetitions of the picked
// value lie at and after the pick than before it (and the value is not 0), use value-1, so that a heavy value
// does not straddle two ranges.  The reference walks to both ends of the run of equal values; here they are found
// by binary search (same result: `start` = last index before the run, or 0; `end` = first index behind it).
template <typename K>
__global__ __launch_bounds__(256) void splitters_kernel(const K *__restrict__ sample, uint64_t m, uint32_t parts, K *__restrict__ delim)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i + 1 >= parts) return;
	const double pct = __ddiv_rn((double)m, (double)parts);
	const uint64_t idx = (uint64_t)__dsub_rn(__dmul_rn(pct, (double)(i + 1)), 0.001); // (no fused multiply-add: the reference rounds twice)
	K v = sample[idx];
	uint64_t lo = 0, hi = idx; // first index holding v
	while (lo < hi) {
		const uint64_t mid = (lo + hi) >> 1;
		if (sample[mid] < v) lo = mid + 1; else hi = mid;
	}
	const uint64_t start = lo ? lo - 1 : 0;
	uint64_t lo2 = idx, hi2 = m; // first index behind the run
	while (lo2 < hi2) {
		const uint64_t mid = (lo2 + hi2) >> 1;
		if (sample[mid] <= v) lo2 = mid + 1; else hi2 = mid;
	}
	if (idx - start < lo2 - idx && v) --v;
	delim[i] = v;
}

} // namespace msd
