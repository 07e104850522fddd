// msd_direct.hpp -- classify with direct block placement, lean tile loop (included by msd_device.hpp).
//
// Same contract as the streaming classify (block_map, slot_full, fb, leftovers), same idea as DESIGN.md
// section 2 A': bucket boundaries on the slot grid are known before the pass, every workgroup owns a
// *piece* of every bucket's region, reads only its own pieces and writes a completed block straight into
// a slot its piece of the block's bucket has already given up.  What differs from the first version of this
// kernel is the tile loop, which was bound by instruction issue (profiles/r02_stamps_classify_direct_before.json:
// about 520 instructions per wave and 4096-key tile at 8 waves per SIMD), not by memory:
//
//   * one LDS fetch-add per key does everything: the per-bucket counters are never reset, they hold the
//     bucket's fill, so the value returned IS the key's place in the bucket's buffer -- no second phase
//     behind a barrier, no per-bucket metadata read per key.  A key whose place is beyond the buffer
//     (its bucket completes a block in this tile) stays in its register and is written behind the flush.
//   * a bucket flushes at most one block per round of a tile; a tile that needs more (a bucket received
//     more than a block's worth of keys: skew) repeats the round -- rare by construction, direct placement
//     is only chosen for evenly spread digits.
//   * blocks without a consumed slot in their own piece ("steals") are handled by the bucket waves BEHIND
//     the job barrier, together with the choice of the next reads, while the other waves flush; the
//     tile loop has three workgroup barriers and no conditional fourth.
//   * 512-thread workgroups, two per CU (128 VGPRs per lane instead of 64: nothing spills), tiles of 4096
//     u32 / 2048 u64 keys (two 16-byte vectors per thread; the sweep is recorded at Direct2Cfg below).
#pragma once

namespace msd {

template <typename K, typename V> struct Direct2Cfg;
// (overridable for experiments, tools/variant_run.py.  2^30 u32 keys, per launch: 512 threads x 2 vectors 1.88 ms,
// x 4 vectors 1.97, x 1 vector 2.12; 384 x 4: 2.26; 640 x 2: 1.98; 1024 x 1: 2.33)
#ifndef MSD_D2_TH
#define MSD_D2_TH 512
#define MSD_D2_NV 2
#define MSD_D2_WPE 4
#endif
template <> struct Direct2Cfg<uint32_t, NoVal> { static constexpr int TH = MSD_D2_TH, NV = MSD_D2_NV, WPE = MSD_D2_WPE; };
#ifndef MSD_D2K_TH // (u64 keys, overridable for experiments; 2^30 keys: two vectors per thread 3.97 ms per round, four 4.15, one 4.68)
#define MSD_D2K_TH 512
#define MSD_D2K_NV 2
#define MSD_D2K_WPE 4
#endif
template <> struct Direct2Cfg<uint64_t, NoVal> { static constexpr int TH = MSD_D2K_TH, NV = MSD_D2K_NV, WPE = MSD_D2K_WPE; };
// tuples: key and payload buffers fill the LDS, one workgroup per CU
#ifndef MSD_D2P_NV // (tuples, overridable for experiments; 2^30 tuples: two vectors per thread 6.15 ms per round, one 6.31, four 9.9)
#define MSD_D2P_NV 2
#endif
template <> struct Direct2Cfg<uint64_t, uint64_t> { static constexpr int TH = 1024, NV = MSD_D2P_NV, WPE = 4; };

template <typename K, typename V> struct Direct2Lds {
	using C = Cfg<K, V>;
	using D = Direct2Cfg<K, V>;
	static constexpr bool HV = has_val<V>::value;
	static constexpr int LPB = C::B / Vec16<K>::N;      // lanes that move one block
	static constexpr int GPW = D::TH / LPB;             // lane groups per workgroup
	static constexpr int GPT = GPW * D::NV;             // slots read per tile
	// (+ 64 elements: the junk words, one per lane, that take the keys without a place in their buffer)
	static constexpr size_t kbuf = (size_t)(kP * C::B + 64) * sizeof(K);
	static constexpr size_t vbuf = HV ? (size_t)(kP * C::B + 64) * sizeof(uint64_t) : 0;
	static constexpr size_t head = (size_t)2 * C::B * sizeof(K) + (HV ? (size_t)2 * C::B * sizeof(uint64_t) : 0);
	// cnt, hc, loff, plo, pnl, rot, cw, bst, fbc : 9*kP ; jobs 2*kP ; steal list kP ; sel GPT ; tmp 32
	static constexpr size_t small = (size_t)(9 * kP + 2 * kP + kP + GPT + 32) * sizeof(uint32_t);
	static constexpr size_t bytes = kbuf + vbuf + head + small;
};

template <typename K, typename V>
__global__ __launch_bounds__((Direct2Cfg<K, V>::TH), (Direct2Cfg<K, V>::WPE)) void classify_direct2_kernel(
	K *__restrict__ keys, uint64_t *__restrict__ vals, const Stripe *__restrict__ stripes,
	const Parent *__restrict__ parents, const DirectPlan *__restrict__ plans, uint8_t *__restrict__ block_map,
	uint8_t *__restrict__ slot_full, uint32_t *__restrict__ fb, uint32_t *__restrict__ lo_cnt,
	uint32_t *__restrict__ lo_off, K *__restrict__ lo_keys, uint64_t *__restrict__ lo_vals,
	uint32_t *__restrict__ nfull, Counters *__restrict__ ctr, uint32_t force)
{
	if (!force && ctr->direct_uneven) return; // the plan declined: the streaming kernel behind this launch runs instead
	using C = Cfg<K, V>;
	using D = Direct2Cfg<K, V>;
	using L = Direct2Lds<K, V>;
	constexpr bool HV = has_val<V>::value;
	constexpr int B = C::B, TH = D::TH, NV = D::NV;
	constexpr int VEC = Vec16<K>::N;
	constexpr int LPB = L::LPB, GPW = L::GPW, GPT = L::GPT;
	constexpr int SPW = GPT / 4;   // slots each of the 4 bucket waves may request per tile
	constexpr int KPT = NV * VEC;  // keys per thread per tile
	constexpr int PB = kP * B;
	constexpr uint32_t NONE = 0xFFFFFFFFu;
	constexpr int NJ = 0, NS = 2, NSEL = 4, SKEW = 6, MORE = 8, SCAN = 16; // tmp[] words (the first four pairs: by tile parity)
	static_assert(TH > kP && TH % 64 == 0 && GPT % 4 == 0 && SPW <= 64 && GPT <= kP, "tile geometry");
	static_assert((TH - kP) % LPB == 0, "the waves behind the bucket waves flush whole blocks");

	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	// the small arrays come first: their LDS addresses then fit the 16-bit offset field of the DS instructions
	uint32_t *cnt = reinterpret_cast<uint32_t *>(smem); // fill of bucket d's buffer (+ the ranks handed out in this tile)
	uint64_t *headv = reinterpret_cast<uint64_t *>(smem + L::small);
	K *headk = reinterpret_cast<K *>(smem + L::small + (HV ? (size_t)2 * B * sizeof(uint64_t) : 0));
	K *kbuf = reinterpret_cast<K *>(smem + L::small + L::head);
	uint64_t *vbuf = reinterpret_cast<uint64_t *>(smem + L::small + L::head + L::kbuf);
	uint32_t *hc = cnt + kP;
	uint32_t *loff = hc + kP;
	uint32_t *plo = loff + kP;     // first slot of my piece of child d
	uint32_t *pnl = plo + kP;      // slots in the piece
	uint32_t *rot = pnl + kP;      // the piece is used from slot rot[d] on, wrapping around (see phys)
	uint32_t *cw = rot + kP;       // slots of the piece consumed (low 16) | blocks written into it (high 16)
	uint32_t *bst = cw + kP;       // reads issued (low 16) | has a slot in the next tile (bit 16) / the one after (bit 17)
	uint32_t *fbc = bst + kP;      // full blocks produced
	uint32_t *jobs = fbc + kP;     // [2*g] bucket, [2*g+1] destination slot (blocks with a slot in their own piece)
	uint32_t *stl = jobs + 2 * kP; // buckets whose block has to take a slot of another piece
	uint32_t *sel = stl + kP;      // slots to read for the tile after next
	uint32_t *tmp = sel + GPT;

	const uint32_t tid = threadIdx.x, grp = tid / LPB, sub = tid % LPB, lane = tid & 63;
	// junk word of this lane, behind the buffers (one word for all lanes would make every branch-free "no place" write
	// a 64-way same-address conflict: 1500 cycles per tile)
	const uint32_t junk = (uint32_t)(kP * C::B) + lane;
	const Stripe st = stripes[blockIdx.x];
	const Parent pa = parents[st.parent];
	const uint32_t shift = pa.shift, mask = (1u << pa.width) - 1u;
	const uint32_t W = pa.stripe_hi - pa.stripe_lo, me = blockIdx.x - pa.stripe_lo;
	const DirectPlan *plan = plans + st.parent;
	const uint32_t slot0 = plan->bound[0], slotN = plan->bound[kP];

	if (tid < kP) {
		const uint32_t b0 = plan->bound[tid], b1 = plan->bound[tid + 1], len = b1 - b0;
		const uint32_t p0 = b0 + (uint32_t)((uint64_t)me * len / W);
		const uint32_t pn0 = b0 + (uint32_t)((uint64_t)(me + 1) * len / W) - p0;
		plo[tid] = p0;
		pnl[tid] = pn0;
		// All workgroups advance through their pieces at about the same pace.  Were every piece used
		// from its first slot on, the addresses in flight at any moment would agree in the bits that
		// select the memory channel; a per-piece starting offset spreads them over all channels.
		rot[tid] = pn0 ? ((tid * 2654435761u) ^ (me * 40503u + (me >> 3))) % pn0 : 0u;
		bst[tid] = 0;
		fbc[tid] = 0;
		cw[tid] = 0;
		cnt[tid] = 0;
		hc[tid] = 0;
	}
	if (tid < 32) tmp[tid] = 0;
	// head / tail keys of the parent (outside every aligned slot): first / last workgroup parks them
	const uint64_t pend = pa.start + pa.count;
	uint32_t h = 0;
	if (me == 0) {
		h = (uint32_t)((uint64_t)slot0 * B - pa.start);
		if (tid < h) {
			headk[tid] = keys[pa.start + tid];
			if (HV) headv[tid] = vals[pa.start + tid];
		}
	}
	uint32_t h2 = 0;
	if (me == W - 1) {
		h2 = (uint32_t)(pend - (uint64_t)slotN * B);
		if (tid < h2) {
			headk[h + tid] = keys[(uint64_t)slotN * B + tid];
			if (HV) headv[h + tid] = vals[(uint64_t)slotN * B + tid];
		}
	}
	const uint32_t hh = h + h2;
	__syncthreads();
	if (tid < hh) atomicAdd(&hc[digit_of(headk[tid], shift, mask)], 1u);
	// A block written into piece d holds bucket d unless it was stolen by another bucket: mark the
	// whole piece up front (64-byte runs), the rare stolen slot is overwritten when it is handed out;
	// which slots hold a block at all (the first `written` of each piece) is recorded at the end.
	for (uint32_t d = tid >> 6; d < (uint32_t)kP; d += TH / 64) {
		const uint32_t p0 = plo[d], pn = pnl[d];
		for (uint32_t j = lane; j < pn; j += 64) block_map[p0 + j] = (uint8_t)d;
	}

	// idx-th slot of piece d in the order the piece is used
	auto phys = [&](uint32_t d, uint32_t idx) -> uint32_t {
		const uint32_t pn = pnl[d];
		uint32_t j = rot[d] + idx;
		if (j >= pn) j -= pn;
		return plo[d] + j;
	};
	// Which slots to read next (bucket waves, thread d = bucket d): the pieces whose buckets have the least
	// room (read-ahead slots plus free buffer space) go first, so that a bucket's next slot has been read
	// before its buffer fills.
	auto select_reads = [&](uint32_t *count_out) {
		const uint32_t wr = cw[tid] >> 16, bs = bst[tid];
		uint32_t rd = bs & 0xFFFFu, q = (bs >> 16) & 3u;
		const uint32_t fill = min(cnt[tid], (uint32_t)B);
		const bool elig = rd < pnl[tid];
		const uint32_t v = min(1023u, (rd - wr) * B + B - fill);
		uint64_t sm;
		const bool mine = wave_select_smallest(v, elig, SPW, sm);
		const uint32_t n = (uint32_t)__popcll(sm);
		uint32_t *selw = sel + __builtin_amdgcn_readfirstlane(tid >> 6) * SPW;
		if (mine) {
			selw[popc_below_lane(sm)] = phys(tid, rd);
			++rd;
			q |= 2; // bit0: has a slot in the next tile, bit1: in the tile after it
		}
		// (lane id recomputed: kept in a register across the tile loop it gets spilled, and a reload behind the
		// stores in flight waits for them)
		uint32_t l; // (volatile: a plain mbcnt is hoisted out of the tile loop as an invariant -- and spilled all the same)
		asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
		if (l < (uint32_t)SPW && l >= n) selw[l] = NONE;
		if (l == 0 && n) atomicAdd(count_out, n);
		bst[tid] = rd | (q << 16);
	};
	// NV slots' worth of keys into one register set; bit v of the result: vector v holds keys
	auto load_tile = [&](K *kr, uint64_t *vr) -> uint32_t {
		uint32_t okm = 0, sl[NV];
#pragma unroll
		for (int v = 0; v < NV; ++v) sl[v] = sel[v * GPW + grp];
#pragma unroll
		for (int v = 0; v < NV; ++v) {
			// Unconditional loads (a vector without a slot re-reads the array's first block and is ignored):
			// with a fixed number of loads per refill the compiler can count the vector memory operations
			// issued behind a register set's loads and wait for exactly those loads (vmcnt(n)) where the set
			// is used a tile later; a conditional load makes it wait for everything (vmcnt(0)), including the
			// refill of the other set issued just before -- the whole memory latency, once per tile.
			const uint32_t slot = sl[v];
			const bool ok = slot >= slot0 && slot < slotN; // NONE fails the test
			okm |= ok ? 1u << v : 0u;
			const uint64_t at = (ok ? (uint64_t)slot * B : 0ull) + sub * VEC;
			if constexpr (sizeof(K) == 4) {
				const uint4 q = *reinterpret_cast<const uint4 *>(keys + at);
				kr[v * VEC + 0] = q.x; kr[v * VEC + 1] = q.y; kr[v * VEC + 2] = q.z; kr[v * VEC + 3] = q.w;
			} else {
				const ulonglong2 q = *reinterpret_cast<const ulonglong2 *>(keys + at);
				kr[v * VEC + 0] = q.x; kr[v * VEC + 1] = q.y;
			}
			if constexpr (HV) {
				const ulonglong2 q = *reinterpret_cast<const ulonglong2 *>(vals + at);
				vr[v * VEC + 0] = q.x; vr[v * VEC + 1] = q.y;
			}
		}
		return okm;
	};
	// one block: the bucket's LDS buffer -> its slot (a lane group moves it, 16 bytes per lane)
	auto flush_block = [&](uint32_t d, uint32_t slot) {
		if (slot >= slot0 && slot < slotN) {
			const uint64_t dst = (uint64_t)slot * B + sub * VEC;
			*reinterpret_cast<uint4 *>(keys + dst) = *reinterpret_cast<const uint4 *>(kbuf + d * B + sub * VEC);
			if constexpr (HV)
				*reinterpret_cast<uint4 *>(vals + dst) = *reinterpret_cast<const uint4 *>(vbuf + d * B + sub * VEC);
		} else if (sub == 0)
			msd_note_error(ctr, 8u);
	};
	// per bucket: a full buffer becomes a block; it takes the next consumed slot of the bucket's own piece
	// or joins the steal list.  `consumed_now`: the bucket's slot of this tile is in registers by now.
	auto bucket_round = [&](uint32_t par, uint32_t consumed_now) {
		const uint32_t c = cnt[tid], cwv = cw[tid];
		const uint32_t cons = (cwv & 0xFFFFu) + consumed_now;
		uint32_t wr = cwv >> 16;
		if (c >= (uint32_t)B) {
			cnt[tid] = c - B;
			fbc[tid] += 1;
			if (wr < cons) {
				const uint32_t j = atomicAdd(&tmp[NJ + par], 1u);
				jobs[2 * j] = tid;
				jobs[2 * j + 1] = phys(tid, wr);
				++wr;
			} else
				stl[atomicAdd(&tmp[NS + par], 1u)] = tid;
		}
		cw[tid] = cons | (wr << 16);
	};
	// behind the job barrier.  Bucket waves: blocks without a slot of their own take a consumed slot of another
	// piece (one lane group per block: its first lane claims, the group moves the block).  Other waves: the jobs.
	auto flush_round = [&](uint32_t par) {
		if (tid < kP) {
			const uint32_t ns = tmp[NS + par];
			for (uint32_t g = grp; g < ns; g += kP / LPB) {
				const uint32_t d = stl[g];
				uint32_t slot = NONE;
				if (sub == 0) {
					uint32_t e = (d * 37u + 1u) & (kP - 1);
					for (uint32_t tries = 0; tries < 3 * kP; ++tries, e = (e + 1) & (kP - 1)) {
						const uint32_t c0 = cw[e];
						if ((c0 >> 16) >= (c0 & 0xFFFFu)) continue;
						const uint32_t old = atomicAdd(&cw[e], 1u << 16);
						if ((old >> 16) < (old & 0xFFFFu)) {
							slot = phys(e, old >> 16);
							break;
						}
						atomicSub(&cw[e], 1u << 16);
					}
					if (slot >= slot0 && slot < slotN) block_map[slot] = (uint8_t)d;
				}
				slot = (uint32_t)__shfl((int)slot, (int)(lane & ~(uint32_t)(LPB - 1)));
				flush_block(d, slot); // (no slot cannot happen: consumed slots >= blocks produced; counted as an error)
			}
		} else {
			// four blocks in flight per lane group: descriptors, then the LDS reads, then the stores
			const uint32_t nj = tmp[NJ + par];
			constexpr uint32_t GS = (TH - kP) / LPB, U = 4;
			for (uint32_t g0 = grp - kP / LPB; g0 < nj; g0 += GS * U) {
				uint32_t dd[U], ss[U];
				u32x4 q[U], qv[U];
#pragma unroll
				for (uint32_t u = 0; u < U; ++u) {
					const uint32_t g = min(g0 + u * GS, nj - 1u);
					dd[u] = jobs[2 * g];
					ss[u] = g0 + u * GS < nj ? jobs[2 * g + 1] : NONE;
				}
#pragma unroll
				for (uint32_t u = 0; u < U; ++u) {
					q[u] = *reinterpret_cast<const u32x4 *>(kbuf + dd[u] * B + sub * VEC);
					if constexpr (HV) qv[u] = *reinterpret_cast<const u32x4 *>(vbuf + dd[u] * B + sub * VEC);
				}
#pragma unroll
				for (uint32_t u = 0; u < U; ++u) {
					if (ss[u] >= slot0 && ss[u] < slotN) {
						const uint64_t dst = (uint64_t)ss[u] * B + sub * VEC;
						*reinterpret_cast<u32x4 *>(keys + dst) = q[u];
						if constexpr (HV) *reinterpret_cast<u32x4 *>(vals + dst) = qv[u];
					} else if (ss[u] != NONE && sub == 0)
						msd_note_error(ctr, 9u);
				}
			}
		}
	};

	K kreg[KPT], kregB[KPT];
	uint64_t vreg[HV ? KPT : 1], vregB[HV ? KPT : 1];
	uint32_t dat[KPT]; // where a key that waits for its bucket's flush goes (LDS index), NONE otherwise
#pragma unroll
	for (int i = 0; i < KPT; ++i) dat[i] = NONE;

	// prologue: the first two tiles
	if (tid < kP) {
		select_reads(&tmp[NSEL]);
		bst[tid] = (bst[tid] & 0xFFFFu) | (((bst[tid] >> 17) & 1u) << 16); // -> bit0: in tile 0
	}
	__syncthreads();
	uint32_t nA = tmp[NSEL];
	uint32_t okA = load_tile(kreg, vreg);
	__syncthreads();
	if (tid < kP) select_reads(&tmp[NSEL + 1]); // bit1: in tile 1
	__syncthreads();
	uint32_t nB = tmp[NSEL + 1];
	uint32_t okB = load_tile(kregB, vregB);
	__syncthreads();
	if (tid == 0) tmp[NSEL] = tmp[NSEL + 1] = 0;
	uint32_t par = 0;
	__syncthreads();

	MSD_STAMP_DECL(1);
	MSD_STAMP_START();
	auto tile = [&](K (&kc)[KPT], uint64_t (&vc)[HV ? KPT : 1], uint32_t &okc, uint32_t &nc) {
		MSD_STAMP(9); // refill of the previous tile + loop overhead
		MSD_STAMP_TICK(11);
		// ---- every key takes its place in its bucket's buffer
		// (all fetch-adds are issued before the first result is used, and nothing below branches per key:
		// written key by key the compiler waits for every fetch-add in turn -- 16 LDS round trips per thread,
		// 5900 of the tile's 15900 cycles.  A key without a place in the buffer is written to a junk word.)
		uint32_t rmax = 0;
		if (__all(okc == (1u << NV) - 1u)) {
			uint32_t at[KPT];
#pragma unroll
			for (int i = 0; i < KPT; ++i) {
				const uint32_t d = digit_of(kc[i], shift, mask);
				at[i] = d * B;
				at[i] += atomicAdd(&cnt[d], 1u);
			}
#pragma unroll
			for (int i = 0; i < KPT; ++i) {
				const uint32_t r = at[i] & (uint32_t)(B - 1); // (only meaningful below B; rmax needs the real rank)
				(void)r;
				const uint32_t d = digit_of(kc[i], shift, mask);
				const uint32_t rank = at[i] - d * B;
				rmax = max(rmax, rank);
				const bool in = rank < (uint32_t)B;
				const uint32_t w = in ? at[i] : junk;
				kbuf[w] = kc[i];
				if constexpr (HV) vbuf[w] = vc[i];
				dat[i] = in ? NONE : at[i] - B; // behind this tile's flush (valid if rank < 2B, else the tile repeats its round)
			}
		} else { // a tile at the end of the pieces: some vectors hold no keys
#pragma unroll
			for (int i = 0; i < KPT; ++i) {
				if ((okc >> (i / VEC)) & 1u) {
					const uint32_t d = digit_of(kc[i], shift, mask);
					const uint32_t r = atomicAdd(&cnt[d], 1u);
					const uint32_t at = d * B + r;
					rmax = max(rmax, r);
					if (r < (uint32_t)B) {
						kbuf[at] = kc[i];
						if constexpr (HV) vbuf[at] = vc[i];
					} else
						dat[i] = at - B;
				}
			}
		}
		if (rmax >= 2u * B - 1u) tmp[SKEW + par] = 1; // some bucket completes two blocks or more
		MSD_STAMP(0); // scatter (incl. the wait for this tile's keys)
		__syncthreads(); // B1: the counters are final
		MSD_STAMP(1);
		if (tid < kP) {
			const uint32_t bs = bst[tid], q = (bs >> 16) & 3u;
			bst[tid] = (bs & 0xFFFFu) | ((q >> 1) << 16);
			bucket_round(par, q & 1u);
		}
		if (tid == kP) { // (a thread outside the bucket waves) the other parity's words were last read before B1
			tmp[NJ + (par ^ 1)] = 0;
			tmp[NS + (par ^ 1)] = 0;
			tmp[NSEL + (par ^ 1)] = 0;
			tmp[SKEW + (par ^ 1)] = 0;
		}
		MSD_STAMP(2); // per-bucket bookkeeping
		__syncthreads(); // B2: the jobs are posted
		MSD_STAMP(3);
		flush_round(par);
		MSD_STAMP(4); // flush
		if (tid < kP) select_reads(&tmp[NSEL + par]);
		MSD_STAMP(5); // read selection
		__syncthreads(); // B3: the buffers of the flushed buckets are free again
		MSD_STAMP(7);
		if (tmp[SKEW + par]) { // (uniform, rare) repeat the round until no bucket holds a full buffer
			for (;;) {
				// waiting keys whose place is inside the buffer now go there; the others move up one block
				// (their bucket has a full buffer again and flushes it in the next round)
#pragma unroll
				for (int i = 0; i < KPT; ++i) {
					if (dat[i] != NONE) {
						const uint32_t d = digit_of(kc[i], shift, mask);
						if (dat[i] - d * B < (uint32_t)B) {
							kbuf[dat[i]] = kc[i];
							if constexpr (HV) vbuf[dat[i]] = vc[i];
							dat[i] = NONE;
						} else
							dat[i] -= B;
					}
				}
				if (tid == 0) {
					tmp[NJ + par] = 0;
					tmp[NS + par] = 0;
				}
				__syncthreads();
				if (tid < kP) bucket_round(par, 0u);
				__syncthreads();
				if (tmp[NJ + par] + tmp[NS + par] == 0) break; // (uniform) every buffer is below a block again
				flush_round(par);
				__syncthreads();
			}
		} else {
#pragma unroll
			for (int i = 0; i < KPT; ++i) { // (branch-free: the others write their junk word)
				const uint32_t w = dat[i] != NONE ? dat[i] : junk;
				kbuf[w] = kc[i];
				if constexpr (HV) vbuf[w] = vc[i];
				dat[i] = NONE;
			}
		}
		MSD_STAMP(6); // waiting keys
		nc = tmp[NSEL + par];
		okc = load_tile(kc, vc);
		par ^= 1;
	};
	for (;;) {
		if (!nA) break;
		tile(kreg, vreg, okA, nA);
		if (!nB) break;
		tile(kregB, vregB, okB, nB);
	}
	__syncthreads();
	MSD_STAMP(9);
	// ---- epilogue: leftovers (partial buffers + head/tail keys) to the side area
	uint32_t fill_r = 0, lc = 0;
	if (tid < kP) {
		fill_r = cnt[tid]; // < B: the last round left no full buffer
		if (fill_r >= (uint32_t)B) msd_note_error(ctr, 10u);
		lc = fill_r + hc[tid];
	}
	uint32_t ltot;
	const uint32_t lex = block_excl_scan256(lc, tmp + SCAN, ltot);
	const size_t so = (size_t)blockIdx.x * kP + tid;
	if (tid < kP) {
		loff[tid] = lex;
		lo_cnt[so] = lc;
		lo_off[so] = lex;
		fb[so] = fbc[tid];
		hc[tid] = 0;
	}
	if (tid == 0) nfull[blockIdx.x] = 0;
	__syncthreads();
	for (uint32_t d = tid >> 6; d < (uint32_t)kP; d += TH / 64) {
		const uint32_t p0 = plo[d], pn = pnl[d], wr = cw[d] >> 16, r0 = rot[d];
		for (uint32_t j = lane; j < pn; j += 64) slot_full[p0 + j] = (j >= r0 ? j - r0 : j + pn - r0) < wr ? 1 : 0;
	}
	for (uint32_t idx = tid; idx < (uint32_t)PB; idx += TH) {
		const uint32_t d = idx / B, j = idx % B;
		if (j < min(cnt[d], (uint32_t)B)) {
			lo_keys[st.lo_base + loff[d] + j] = kbuf[idx];
			if constexpr (HV) lo_vals[st.lo_base + loff[d] + j] = vbuf[idx];
		}
	}
	if (tid < hh) {
		const uint32_t d = digit_of(headk[tid], shift, mask);
		const uint32_t r = atomicAdd(&hc[d], 1u);
		const uint64_t at = st.lo_base + loff[d] + min(cnt[d], (uint32_t)B) + r;
		lo_keys[at] = headk[tid];
		if constexpr (HV) lo_vals[at] = headv[tid];
	}
	MSD_STAMP(10); // epilogue
	MSD_STAMP_FLUSH(TH / 64);
}

} // namespace msd
