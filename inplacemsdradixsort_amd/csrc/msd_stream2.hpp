// msd_stream2.hpp -- the streaming classify (phase A of a round whose buckets are NOT evenly spread: skewed keys, sorted
// or run-structured input, range partitioning by splitters) with the lean tile loop of classify_direct2_kernel
// (included by msd_device.hpp).
//
// Same contract as classify_kernel (msd_device.hpp): one workgroup per stripe, keys stream through registers into 256
// per-bucket LDS buffers of one block each, every completed block is flushed to the next slot BEHIND the workgroup's own
// read cursor (cf. range_partition_to_blocks, src/msb_64.c:611-667), block_map records the bucket of every slot, partial
// buffers and head keys go to the stripe's leftover area, the digit histogram falls out of the pass.  What differs is
// the tile loop, rebuilt like the direct kernel's (profiles/r02_stamps_classify_stream.json showed
// 11.1 thousand cycles per 4096-key tile, a third of them in a per-bucket bookkeeping section between two barriers
// and a second per-key pass that looked up that section's results):
//   * the per-bucket counters are never reset: they hold the bucket's fill, so ONE LDS fetch-add per key returns the
//     key's place in the bucket's buffer, and a key whose place is inside the buffer is written there at once -- no
//     second pass over the keys, no per-bucket look-up per key;
//   * a key whose place lies beyond the buffer (its bucket completes a block in this tile) stays in its register and is
//     written behind the flush; only in a SKEWED tile -- some bucket completes two blocks or more -- do those keys look
//     their bucket's claim up: keys of a further whole block go straight from registers to that block's slot, the
//     remainder waits for the flush like everywhere else;
//   * the bucket round between the barriers is a dozen instructions for the four bucket waves: buckets with a full
//     buffer claim their slots and a place in the job table with one packed fetch-add;
//   * 512-thread workgroups, two per CU, 128 VGPRs (tuples: 1024 threads, one per CU); all fetch-adds of a tile are
//     issued before the first result is used; the refill loads are unconditional for whole tiles.
// A bucket that took more than a sixteenth of the previous tile is counted per wave in this one (the lanes of a wave
// that hold one of its keys take ONE fetch-add together), as in classify_kernel.
#pragma once

namespace msd {

template <typename K, typename V> struct Stream2Cfg;
#ifndef MSD_S2_TH // (overridable for experiments)
#define MSD_S2_TH 512
#define MSD_S2_NV 2
#endif
template <> struct Stream2Cfg<uint32_t, NoVal> { static constexpr int TH = MSD_S2_TH, NV = MSD_S2_NV, WPE = 2048 / MSD_S2_TH; };
template <> struct Stream2Cfg<uint64_t, NoVal> { static constexpr int TH = 512, NV = 2, WPE = 4; };
template <> struct Stream2Cfg<uint64_t, uint64_t> { static constexpr int TH = 1024, NV = 2, WPE = 4; };

template <typename K, typename V> struct Stream2Lds {
	using C = Cfg<K, V>;
	static constexpr bool HV = has_val<V>::value;
	static constexpr size_t kbuf = (size_t)(kP * C::B + 64) * sizeof(K); // + a junk word per lane
	static constexpr size_t vbuf = HV ? (size_t)(kP * C::B + 64) * sizeof(uint64_t) : 0;
	static constexpr size_t head = (size_t)C::B * sizeof(K) + (HV ? (size_t)C::B * sizeof(uint64_t) : 0);
	static constexpr int JOBS = kP + 8;
	// cnt, hc, loff, meta : 4 kP ; jobs ; multi 64 ; spare 64 ; tmp 32
	static constexpr size_t small = (size_t)(4 * kP + JOBS + 64 + 64 + 32) * sizeof(uint32_t);
	static constexpr size_t bytes = kbuf + vbuf + head + small;
};

template <typename K, typename V, bool RANGE = false>
__global__ __launch_bounds__((Stream2Cfg<K, V>::TH), (Stream2Cfg<K, V>::WPE)) void classify_stream2_kernel(
	K *__restrict__ keys, uint64_t *__restrict__ vals, const Stripe *__restrict__ stripes,
	const Parent *__restrict__ parents, uint8_t *__restrict__ block_map,
	uint32_t *__restrict__ fb, uint32_t *__restrict__ lo_cnt, uint32_t *__restrict__ lo_off,
	K *__restrict__ lo_keys, uint64_t *__restrict__ lo_vals, uint32_t *__restrict__ nfull,
	const K *__restrict__ splitters = nullptr,
	// launched behind a direct-placement attempt: runs only if that declined (Counters::direct_uneven != 0)
	const uint32_t *__restrict__ run_if_nonzero = nullptr)
{
	if (run_if_nonzero && *run_if_nonzero == 0) return;
	using C = Cfg<K, V>;
	using S = Stream2Cfg<K, V>;
	using L = Stream2Lds<K, V>;
	constexpr bool HV = has_val<V>::value;
	constexpr int B = C::B, TH = S::TH, NV = S::NV;
	constexpr int VEC = Vec16<K>::N;
	constexpr int KPT = NV * VEC;     // keys per thread per tile
	constexpr int T = TH * KPT;       // keys per tile
	constexpr int LPB = B / VEC;      // lanes that move one block
	constexpr int PB = kP * B;
	constexpr uint32_t NONE = 0xFFFFFFFFu;
	constexpr int CL = 0, HOT = 2, NM = 4, SKEW = 6, SCAN = 16; // tmp[] words (pairs: by tile parity)
	static_assert(TH > kP && TH % 64 == 0 && T / B <= 254, "tile geometry (a tile completes at most 255 blocks: 8-bit fields)");

	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	// the small arrays first: their LDS addresses fit the 16-bit offset field of the DS instructions
	uint32_t *cnt = reinterpret_cast<uint32_t *>(smem); // fill of bucket d's buffer (+ the places handed out in this tile)
	uint32_t *hc = cnt + kP;      // head keys per bucket
	uint32_t *loff = hc + kP;     // leftover offsets (epilogue)
	uint32_t *meta = loff + kP;   // per bucket and tile: blocks completed | first claimed slot << 8
	uint32_t *jobs = meta + kP;   // flush jobs: bucket | slot << 8
	uint32_t *multi = jobs + L::JOBS; // buckets that completed more than one block: bucket | blocks << 8 | first slot << 16
	uint32_t *spare = multi + 64; // a word per lane that only ever receives zeros
	uint32_t *tmp = spare + 64;
	uint64_t *headv = reinterpret_cast<uint64_t *>(smem + L::small);
	K *headk = reinterpret_cast<K *>(smem + L::small + (HV ? (size_t)B * sizeof(uint64_t) : 0));
	K *kbuf = reinterpret_cast<K *>(smem + L::small + L::head);
	uint64_t *vbuf = reinterpret_cast<uint64_t *>(smem + L::small + L::head + L::kbuf);
	K *spl = reinterpret_cast<K *>(smem + L::bytes); // RANGE only: the delimiters (the launch adds kP keys of LDS)

	const uint32_t tid = threadIdx.x, lane = tid & 63;
	const uint32_t junk = (uint32_t)PB + lane; // this lane's junk word behind the buffers
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
		cnt[tid] = 0;
		hc[tid] = 0;
		meta[tid] = 0;
	}
	if (tid < 32) tmp[tid] = 0;
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
	uint32_t fill_r = 0;         // thread d < kP: fill of bucket d's buffer before the tile
	uint32_t fb_r = 0;           // ... full blocks produced

	K kreg[KPT], kregB[KPT];
	uint64_t vreg[HV ? KPT : 1], vregB[HV ? KPT : 1];
	uint32_t dat[KPT]; // where a key that waits for its bucket's flush goes (LDS index), NONE otherwise
#pragma unroll
	for (int i = 0; i < KPT; ++i) dat[i] = NONE;

	// (uniform 64-bit base + 32-bit per-thread offset; whole tiles load unconditionally so that the compiler can count
	// the loads behind a register set's and wait for exactly those)
	auto load_tile = [&](uint64_t pos, K *kr, uint64_t *vr) {
		const K *kp = keys + pos;
		const uint64_t *vp = vals + pos;
		const uint32_t rem = pos < st.end ? (uint32_t)(st.end - pos < (uint64_t)T ? st.end - pos : (uint64_t)T) : 0u;
		if (rem == (uint32_t)T) { // (uniform)
#pragma unroll
			for (int v = 0; v < NV; ++v) {
				const uint32_t off = (uint32_t)(v * TH + tid) * VEC;
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
			}
		} else {
#pragma unroll
			for (int v = 0; v < NV; ++v) {
				const uint32_t off = (uint32_t)(v * TH + tid) * VEC;
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
	if (pos < st.end) load_tile(pos, kreg, vreg);
	if (pos + T < st.end) load_tile(pos + T, kregB, vregB);
	uint32_t par = 0;
	MSD_STAMP_DECL(6);
	MSD_STAMP_START();

	auto tile = [&](K (&kc)[KPT], uint64_t (&vc)[HV ? KPT : 1]) {
		const uint64_t npos = pos + T;
		const bool full = npos <= st.end; // uniform: every key of the tile exists
		MSD_STAMP(9);
		MSD_STAMP_TICK(11);
		// ---- every key takes its place in its bucket's buffer: dr = bucket | place << 8 for a key that has to wait
		uint32_t dr[KPT];
		const uint32_t hflag = tmp[HOT + (par ^ 1)];
		if (full && hflag && sizeof(K) == 4 && !RANGE) {
			// the previous tile was skewed -- bucket hflag - 1 took more than a sixteenth of it: the lanes of a wave that hold
			// a key of that bucket take ONE fetch-add together, their first lane for all of them (a same-address LDS atomic
			// serialises per lane).  Branch-free; a lane whose key another one counts adds zero to its spare word.
			const uint32_t hb = hflag - 1u;
			uint32_t old[KPT], dd[KPT];
			uint64_t hm[KPT];
			uint32_t before[KPT], hot_total = 0; // (wave-uniform: keys of the hot bucket in the slots before i / in the whole tile)
#pragma unroll
			for (int i = 0; i < KPT; ++i) {
				dd[i] = digit_of(kc[i], shift, mask);
				hm[i] = __ballot(dd[i] == hb);
				before[i] = hot_total;
				hot_total += (uint32_t)__popcll(hm[i]);
			}
			// ONE fetch-add per wave and tile for all its keys of the hot bucket (lane 0, with the first key's slot); every
			// other key takes its own; a key of the hot bucket adds zero to its lane's spare word
#pragma unroll
			for (int i = 0; i < KPT; ++i) {
				const bool hot = dd[i] == hb;
				uint32_t *at = hot ? spare + lane : cnt + dd[i];
				uint32_t add = hot ? 0u : 1u;
				if (i == 0 && lane == 0) { // (lane 0's own first key is counted here as well if it is not hot)
					if (hot) {
						at = cnt + hb;
						add = hot_total;
					}
				}
				old[i] = atomicAdd(at, add);
			}
			uint32_t hbase;
			{
				// lane 0 could not take both fetch-adds in slot 0 when its own first key is cold: it takes the hot one behind
				const bool lane0_cold = __builtin_amdgcn_readfirstlane((int)(dd[0] != hb));
				uint32_t hv = old[0];
				if (lane0_cold) { // (uniform)
					hv = 0;
					if (lane == 0 && hot_total) hv = atomicAdd(cnt + hb, hot_total);
				}
				hbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)hv);
			}
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int i = 0; i < KPT; ++i) {
				const uint32_t r = dd[i] == hb ? hbase + before[i] + popc_below_lane(hm[i]) : old[i];
				const bool in = r < (uint32_t)B;
				const uint32_t w = in ? dd[i] * B + r : junk;
				kbuf[w] = kc[i];
				if constexpr (HV) vbuf[w] = vc[i];
				dr[i] = in ? NONE : (dd[i] | (r << 8));
			}
		} else if (full) {
			uint32_t at[KPT];
#pragma unroll
			for (int i = 0; i < KPT; ++i) {
				const uint32_t d = digit_of(kc[i], shift, mask);
				at[i] = d | (atomicAdd(&cnt[d], 1u) << 8);
			}
#pragma unroll
			for (int i = 0; i < KPT; ++i) {
				const uint32_t d = at[i] & 0xFFu, r = at[i] >> 8;
				const bool in = r < (uint32_t)B;
				const uint32_t w = in ? d * B + r : junk;
				kbuf[w] = kc[i];
				if constexpr (HV) vbuf[w] = vc[i];
				dr[i] = in ? NONE : at[i];
			}
		} else {
			const uint32_t rem = (uint32_t)(st.end - pos); // < T here (partial last tile)
#pragma unroll
			for (int i = 0; i < KPT; ++i) {
				const uint32_t off = (uint32_t)((i / VEC) * TH + tid) * VEC + (i % VEC);
				dr[i] = NONE;
				if (off < rem) {
					const uint32_t d = digit_of(kc[i], shift, mask);
					const uint32_t r = atomicAdd(&cnt[d], 1u);
					if (r < (uint32_t)B) {
						kbuf[d * B + r] = kc[i];
						if constexpr (HV) vbuf[d * B + r] = vc[i];
					} else
						dr[i] = d | (r << 8);
				}
			}
		}
		MSD_STAMP(0); // places (incl. the wait for the keys)
		__syncthreads(); // B1: the counters are final
		MSD_STAMP(1);
		// ---- bucket round: buckets whose buffer is full claim consecutive output slots and a place in the job table
		if (tid < kP) {
			const uint32_t c = cnt[tid];
			if (c - fill_r > (uint32_t)T / 16) tmp[HOT + par] = 1u + tid; // skewed tile: the next one counts this bucket's keys per wave
			const uint32_t nb = c / B;
			// the wave claims for all its buckets with ONE packed fetch-add -- slots wslot + base .. (low half) and places
			// in the job table (high half); sixty buckets claiming on their own serialise on that word (2000 cycles of the
			// tile in profiles/r03_stamps_stream2.json).  The bucket's LDS buffer becomes its first block, further ones
			// (skewed tiles only) are written straight from registers; the flush writes the map entries of all of them.
			const uint64_t claimers = __ballot(nb != 0);
			uint32_t bbase = 0;
			if (claimers) { // (uniform per wave)
				const uint32_t incl = wave_incl_scan(nb);
				const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
				uint32_t claim = 0;
				if (lane == 0) claim = atomicAdd(&tmp[CL + par], total | ((uint32_t)__popcll(claimers) << 16));
				claim = (uint32_t)__builtin_amdgcn_readfirstlane((int)claim);
				bbase = (claim & 0xFFFFu) + incl - nb;
				if (nb) {
					jobs[(claim >> 16) + popc_below_lane(claimers)] = tid | (bbase << 8);
					if (nb > 1) {
						multi[atomicAdd(&tmp[NM + par], 1u)] = tid | (nb << 8) | (bbase << 16);
						tmp[SKEW + par] = 1;
					}
					cnt[tid] = c - nb * B;
				}
			}
			meta[tid] = nb | (bbase << 8);
			fill_r = c - nb * B;
			fb_r += nb;
		}
		if (tid == kP) { // (a thread outside the bucket waves) the other parity's words were last read before B1
			tmp[CL + (par ^ 1)] = 0;
			tmp[HOT + (par ^ 1)] = 0;
			tmp[NM + (par ^ 1)] = 0;
			tmp[SKEW + (par ^ 1)] = 0;
		}
		MSD_STAMP(2); // bucket round
		__syncthreads(); // B2: the jobs are posted
		MSD_STAMP(3);
		const uint32_t nbtot = tmp[CL + par] & 0xFFFFu, njobs = tmp[CL + par] >> 16, nmulti = tmp[NM + par];
		// ---- waiting keys: where they go
		if (tmp[SKEW + par]) { // (uniform) some bucket completed two blocks or more: look the claims up
			// (uniform 64-bit base + 32-bit per-lane offset: the tile's slots lie within 2^16 blocks of wslot)
			K *wk = keys + (uint64_t)wslot * B;
			uint64_t *wv = vals + (uint64_t)wslot * B;
			uint32_t mm[KPT];
#pragma unroll
			for (int i = 0; i < KPT; ++i) mm[i] = meta[dr[i] & 0xFFu]; // (all look-ups before the first use; NONE reads bucket 255's word)
#pragma unroll
			for (int i = 0; i < KPT; ++i) {
				const uint32_t d = dr[i] & 0xFFu, r = dr[i] >> 8, nb = mm[i] & 0xFFu;
				const bool waits = dr[i] != NONE;
				if (waits && r < nb * B) { // a further whole block of this bucket: straight to its slot
					const uint32_t off = (mm[i] >> 8) * B + r;
					wk[off] = kc[i];
					if constexpr (HV) wv[off] = vc[i];
				}
				dat[i] = waits && r >= nb * B ? d * B + r - nb * B : NONE; // remainder: behind the flush
			}
		} else {
#pragma unroll
			for (int i = 0; i < KPT; ++i) dat[i] = dr[i] != NONE ? (dr[i] & 0xFFu) * B + (dr[i] >> 8) - B : NONE;
		}
		// ---- flush the completed buffers to their slots behind the read cursor (four blocks in flight per lane group)
		{
			const uint32_t grp = tid / LPB, sub = tid % LPB;
			constexpr uint32_t GS = TH / LPB, U = 4;
			for (uint32_t g0 = grp; g0 < njobs; g0 += GS * U) {
				uint32_t jj[U];
				u32x4 q[U], qv[U];
#pragma unroll
				for (uint32_t u = 0; u < U; ++u) jj[u] = jobs[min(g0 + u * GS, njobs - 1u)];
#pragma unroll
				for (uint32_t u = 0; u < U; ++u) {
					q[u] = *reinterpret_cast<const u32x4 *>(kbuf + (jj[u] & 0xFFu) * B + sub * VEC);
					if constexpr (HV) qv[u] = *reinterpret_cast<const u32x4 *>(vbuf + (jj[u] & 0xFFu) * B + sub * VEC);
				}
#pragma unroll
				for (uint32_t u = 0; u < U; ++u) {
					if (g0 + u * GS < njobs) {
						const uint32_t slot = wslot + (jj[u] >> 8);
						const uint64_t dst = (uint64_t)slot * B + sub * VEC;
						*reinterpret_cast<u32x4 *>(keys + dst) = q[u];
						if constexpr (HV) *reinterpret_cast<u32x4 *>(vals + dst) = qv[u];
						if (sub == 0) block_map[slot] = (uint8_t)(jj[u] & 0xFFu);
					}
				}
			}
			for (uint32_t e = 0; e < nmulti; ++e) { // (skewed tiles only) the map entries of a bucket's further blocks
				const uint32_t m = multi[e], nb = (m >> 8) & 0xFFu;
				for (uint32_t q2 = tid; q2 + 1u < nb; q2 += TH) block_map[wslot + (m >> 16) + 1u + q2] = (uint8_t)(m & 0xFFu);
			}
		}
		MSD_STAMP(4); // waiting keys' places + flush
		__syncthreads(); // B3: the buffers of the flushed buckets are free again
		MSD_STAMP(5);
#pragma unroll
		for (int i = 0; i < KPT; ++i) { // (branch-free: the others write their junk word)
			const uint32_t w = dat[i] != NONE ? dat[i] : junk;
			kbuf[w] = kc[i];
			if constexpr (HV) vbuf[w] = vc[i];
		}
		MSD_STAMP(6); // waiting keys
		if (npos + T < st.end) load_tile(npos + T, kc, vc);
		MSD_STAMP(7); // refill issue
		wslot += nbtot;
		pos = npos;
		par ^= 1;
	};
	while (pos < st.end) {
		tile(kreg, vreg);
		if (pos >= st.end) break;
		tile(kregB, vregB);
	}
	MSD_STAMP_FLUSH(TH / 64);
	__syncthreads();

	// ---- stripe epilogue: leftovers (partial buffers + head keys) to the side area
	uint32_t lc = 0;
	if (tid < kP) lc = fill_r + hc[tid];
	uint32_t ltot;
	const uint32_t lex = block_excl_scan256(lc, tmp + SCAN, ltot);
	const size_t so = (size_t)blockIdx.x * kP + tid;
	if (tid < kP) {
		loff[tid] = lex;
		lo_cnt[so] = lc;
		lo_off[so] = lex;
		fb[so] = fb_r;
		hc[tid] = 0; // reused as head cursor
		meta[tid] = fill_r;
	}
	if (tid == 0) nfull[blockIdx.x] = wslot - st.slot_lo;
	__syncthreads();
	for (uint32_t idx = tid; idx < (uint32_t)PB; idx += TH) {
		const uint32_t d = idx / B, j = idx % B;
		if (j < meta[d]) {
			lo_keys[st.lo_base + loff[d] + j] = kbuf[idx];
			if constexpr (HV) lo_vals[st.lo_base + loff[d] + j] = vbuf[idx];
		}
	}
	if (tid < h) {
		const uint32_t d = digit_of(headk[tid], shift, mask);
		const uint32_t r = atomicAdd(&hc[d], 1u);
		const uint64_t at = st.lo_base + loff[d] + meta[d] + r;
		lo_keys[at] = headk[tid];
		if constexpr (HV) lo_vals[at] = headv[tid];
	}
}

} // namespace msd
