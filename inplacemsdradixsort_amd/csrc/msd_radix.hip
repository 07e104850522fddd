// msd_radix.hip -- host side of the MI355X in-place MSD radix sort and its C ABI
// (include/msd_radix_hip.h).  The host plans rounds (the role of the reference's
// schedule_passes, src/msb_64.c:1334-1400, re-parameterised for LDS capacity and
// <= 8-bit digits) and launches the kernels of msd_device.hpp; it never touches
// key data itself and has no CPU fallback.
#include "msd_device.hpp"
#include "../../include/msd_radix_hip.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

using namespace msd;

#define MSD_VERSION "inplacemsdradixsort_amd 0.2 (gfx950)"

// range partitioning (classify_kernel<.., true>) is built for every element type: the multi-GPU path shards u32 keys,
// u64 keys and the reference's own (u64 key, u64 rid) tuples by sampled splitters when the keys are skewed
template <typename K, typename V> constexpr bool kHasRange = true;

struct PhaseRec {
	const char *name;
	hipEvent_t ev; // recorded at the END of the phase
};

struct msd_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	char *slab = nullptr; // device workspace of the current round (dead between rounds)
	size_t slab_bytes = 0;
	char *keep = nullptr; // device workspace that lives for the whole call
	size_t keep_bytes = 0;
	Segment *lists = nullptr; // leaf-segment lists: [general + fallbacks behind it: 3*cap][counting sort: cap]
	size_t lists_cap = 0;
	void *pinned = nullptr; // small host staging (pinned)
	size_t pinned_bytes = 0;
	std::string err;
	bool profiling = false;
	hipEvent_t ev_start = nullptr;
	std::vector<PhaseRec> phases;
	std::vector<hipEvent_t> ev_pool;
	size_t ev_used = 0;
	std::vector<std::pair<std::string, double>> phase_us;
	std::vector<std::pair<std::string, uint64_t>> stats;
	int sm_count = 256;
	int chains_per_cu[3] = { 4, 4, 2 }; // resident chains_kernel workgroups per CU: u32 keys, u64 keys, tuples (measured at msd_create)
	// direct block placement in the first round: 0 off, 1 when the sampled
	// children are about equally big, 2 whenever the geometry allows (tests)
	int direct_mode = 1;
	uint64_t direct_min = 1ull << 22; // smallest round (elements) it is tried on (tools/size_sweep.py: from 2^22 on it is never slower by more than 7 %, and up to 47 % faster)
	uint64_t direct_min_parent = 1ull << 17; // rounds after the first: smallest parent
	int regpart = 1;       // u64 keys / tuples: rounds of small parents as one register-resident pass (0: A/B comparisons)
	int count16 = 1;       // u32 keys: count_place16_kernel in front of count_place_kernel (0: A/B comparisons)
	int leaf17 = 1;        // u64 keys and tuples: segments of <= 17408 elements are finished by leaf17_kernel (0: tuples: register partition + small leaves, keys: leaf_count_sort_kernel; A/B comparisons)
	int stream_kernel = 2; // streaming classify: 2 = classify_stream2_kernel (lean tile loop), 1 = classify_kernel (round 2; A/B comparisons)
	int mid_leaf = 1;      // u32 keys: merge_count_kernel (list mode) in front of count_walk_kernel (0: A/B comparisons)
	const uint32_t *order_keys = nullptr; // msd_order_low16_counts_u32 has run on these keys and its tables are still in the slab
	uint64_t order_n = 0;
	int merge_leaf = 0;    // msd_merge_buckets_u32: 0 = by bucket size, 1 = merge_place16_kernel, 2 = merge_count_kernel (tests)
};

static int fail(msd_ctx *c, int code, const char *fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	if (c) c->err = buf;
	return code;
}

#define HIPCHK(c, call)                                                                   \
	do {                                                                              \
		hipError_t e_ = (call);                                                   \
		if (e_ != hipSuccess)                                                     \
			return fail(c, MSD_EHIP, "%s failed: %s (%s:%d)", #call,          \
				    hipGetErrorString(e_), __FILE__, __LINE__);           \
	} while (0)

static void set_stat(msd_ctx *c, const char *name, uint64_t v)
{
	for (auto &s : c->stats)
		if (s.first == name) {
			s.second = v;
			return;
		}
	c->stats.emplace_back(name, v);
}
static void add_stat(msd_ctx *c, const char *name, uint64_t v)
{
	for (auto &s : c->stats)
		if (s.first == name) {
			s.second += v;
			return;
		}
	c->stats.emplace_back(name, v);
}

// ------------------------------------------------------------ workspace slab

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// (exact: msd_reserve's sizes are the planner's own worst case for the shape; a buffer that has to grow in the middle of a
// sort gets an eighth on top, so that the next slightly bigger round does not allocate again)
static int dev_reserve(msd_ctx *c, char *&p, size_t &have, size_t bytes, bool exact = false)
{
	if (bytes <= have) return MSD_OK;
	if (p) {
		HIPCHK(c, hipStreamSynchronize(c->stream));
		HIPCHK(c, hipFree(p));
		p = nullptr;
		have = 0;
	}
	bytes = align_up(exact ? bytes : bytes + bytes / 8, 1 << 20);
	hipError_t e = hipMalloc((void **)&p, bytes);
	if (e != hipSuccess) return fail(c, MSD_ENOMEM, "workspace hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
	have = bytes;
	return MSD_OK;
}
static int slab_reserve(msd_ctx *c, size_t bytes, bool exact = false)
{
	c->order_keys = nullptr; // (whoever reserves the slab overwrites the tables a pending msd_order_low16_scatter_u32 would read)
	return dev_reserve(c, c->slab, c->slab_bytes, bytes, exact);
}
static int keep_reserve(msd_ctx *c, size_t bytes, bool exact = false) { return dev_reserve(c, c->keep, c->keep_bytes, bytes, exact); }

// Leaf lists grow between rounds (the host knows how many children a round can add);
// live entries are carried over.
static int lists_reserve(msd_ctx *c, size_t need, size_t live_general, size_t live_count, bool exact = false)
{
	if (need <= c->lists_cap) return MSD_OK;
	const size_t cap = align_up(exact ? need : need + need / 2, 4096);
	Segment *nb = nullptr;
	hipError_t e = hipMalloc((void **)&nb, 4 * cap * sizeof(Segment));
	if (e != hipSuccess) return fail(c, MSD_ENOMEM, "leaf list hipMalloc failed: %s", hipGetErrorString(e));
	if (c->lists) {
		HIPCHK(c, hipStreamSynchronize(c->stream));
		if (live_general) HIPCHK(c, hipMemcpy(nb, c->lists, live_general * sizeof(Segment), hipMemcpyDeviceToDevice));
		if (live_count) HIPCHK(c, hipMemcpy(nb + 3 * cap, c->lists + 3 * c->lists_cap, live_count * sizeof(Segment), hipMemcpyDeviceToDevice));
		HIPCHK(c, hipFree(c->lists));
	}
	c->lists = nb;
	c->lists_cap = cap;
	return MSD_OK;
}

struct Bump { // sizing pass (base == nullptr) or carving pass
	char *base;
	size_t off = 0;
	explicit Bump(char *b) : base(b) {}
	template <typename T> T *take(size_t n)
	{
		off = align_up(off, 256);
		T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
		off += n * sizeof(T);
		return p;
	}
};

static int pinned_reserve(msd_ctx *c, size_t bytes)
{
	if (bytes <= c->pinned_bytes) return MSD_OK;
	if (c->pinned) {
		HIPCHK(c, hipStreamSynchronize(c->stream));
		HIPCHK(c, hipHostFree(c->pinned));
		c->pinned = nullptr;
	}
	bytes = align_up(bytes * 2, 4096);
	HIPCHK(c, hipHostMalloc(&c->pinned, bytes, hipHostMallocDefault));
	c->pinned_bytes = bytes;
	return MSD_OK;
}

// ------------------------------------------------------------ phase timing

static void phase_begin(msd_ctx *c)
{
	c->phases.clear();
	c->ev_used = 0;
	c->phase_us.clear();
	if (!c->profiling) return;
	if (!c->ev_start) (void)hipEventCreate(&c->ev_start);
	(void)hipEventRecord(c->ev_start, c->stream);
}
static void phase_mark(msd_ctx *c, const char *name)
{
	if (!c->profiling) return;
	if (c->ev_used == c->ev_pool.size()) {
		hipEvent_t e;
		(void)hipEventCreate(&e);
		c->ev_pool.push_back(e);
	}
	hipEvent_t e = c->ev_pool[c->ev_used++];
	(void)hipEventRecord(e, c->stream);
	c->phases.push_back({ name, e });
}
static void phase_end(msd_ctx *c)
{
	if (!c->profiling) return;
	(void)hipStreamSynchronize(c->stream);
	hipEvent_t prev = c->ev_start;
	for (auto &p : c->phases) {
		float ms = 0;
		(void)hipEventElapsedTime(&ms, prev, p.ev);
		bool found = false;
		for (auto &q : c->phase_us)
			if (q.first == p.name) {
				q.second += ms * 1000.0;
				found = true;
			}
		if (!found) c->phase_us.emplace_back(p.name, ms * 1000.0);
		prev = p.ev;
	}
}

// ------------------------------------------------------------ round planning

struct RoundPlan {
	std::vector<Parent> parents;
	std::vector<Stripe> stripes;
	uint32_t nchildren = 0;
	uint64_t lo_elems = 0;   // leftover area, elements
	uint64_t nslots = 0;     // slots covered by the stripes
	uint64_t round_keys = 0;
};

static uint32_t ceil_log2_u64(uint64_t x)
{
	uint32_t p = 0;
	while (((uint64_t)1 << p) < x) ++p;
	return p;
}

// Digit width for a parent: 8 bits for big parents, fewer when that already brings
// the children down to about half the LDS-sort capacity.  `leaf_bits` > 0 (keys without
// payload): children that are leaves should keep <= leaf_bits open bits so that the
// one-pass counting sort can finish them.
static uint32_t pick_width(uint64_t count, uint32_t bits, uint64_t small_max, uint32_t leaf_bits)
{
	uint64_t target = small_max / 2;
	uint32_t w = ceil_log2_u64((count + target - 1) / target);
	w = std::max(1u, std::min(8u, w));
	w = std::min(w, bits);
	// widen only while the leaves stay big enough to amortise the counting sort's 64 KiB of counters
	if (leaf_bits && bits - w > leaf_bits && bits <= leaf_bits + 8 && (count >> (bits - leaf_bits)) <= small_max &&
	    (count >> (bits - leaf_bits)) >= 2048)
		w = bits - leaf_bits;
	return w;
}

template <typename K, typename V>
static void plan_round(const std::vector<Segment> &segs, uint64_t small_max, int sm_count, RoundPlan &rp, uint32_t leaf_bits = 0,
		       uint32_t forced_width = 0, uint32_t nsplit = 0)
{
	using C = Cfg<K, V>;
	constexpr uint64_t B = C::B, T = C::T;
	rp = RoundPlan();
	uint64_t total = 0;
	for (auto &s : segs) total += s.count;
	rp.round_keys = total;
	// stripe length: enough stripes to fill the chip a few times over, whole tiles
	// (experiment knobs, compiled in with -DMSD_STRIPE_WANT= / -DMSD_STRIPE_CAP_LOG= for variant builds only:
	// tools/variant_run.py.  They used to be read from the environment, unchecked, by every context of the process.)
#ifndef MSD_STRIPE_WANT
#define MSD_STRIPE_WANT 4
#endif
#ifndef MSD_STRIPE_CAP_LOG
#define MSD_STRIPE_CAP_LOG 20
#endif
	static_assert(MSD_STRIPE_WANT >= 1 && MSD_STRIPE_WANT <= 64 && MSD_STRIPE_CAP_LOG >= 12 && MSD_STRIPE_CAP_LOG <= 20, "stripe geometry knobs out of range (a stripe has at most 2^20 elements)");
	constexpr int want_mul = MSD_STRIPE_WANT, cap_log = MSD_STRIPE_CAP_LOG;
	uint64_t want = std::max<uint64_t>(1, (uint64_t)sm_count * want_mul);
	uint64_t slen = (total + want - 1) / want;
	slen = std::max<uint64_t>(slen, 4 * T);
	slen = std::min<uint64_t>(slen, (uint64_t)1 << cap_log);
	slen = (slen + T - 1) / T * T;
	for (auto &s : segs) {
		Parent p;
		p.start = s.start;
		p.count = s.count;
		p.width = forced_width ? forced_width : pick_width(s.count, s.bits, small_max, leaf_bits);
		p.shift = s.bits - p.width;
		p.child_base = rp.nchildren;
		p.stripe_lo = (uint32_t)rp.stripes.size();
		p.pad = nsplit; // range partitioning: number of delimiters
		rp.nchildren += 1u << p.width;
		const uint64_t end = s.start + s.count;
		const uint64_t a0 = (s.start + B - 1) / B * B; // first aligned position
		uint64_t b = s.start;
		while (b < end) {
			uint64_t e = (b == s.start ? a0 : b) + slen;
			if (e + slen / 2 > end) e = end; // do not leave a short last stripe
			Stripe st;
			st.begin = b;
			st.end = e;
			st.parent = (uint32_t)rp.parents.size();
			st.slot_lo = (uint32_t)((b + B - 1) / B);
			st.slot_hi = (uint32_t)(e / B);
			if (st.slot_hi < st.slot_lo) st.slot_hi = st.slot_lo;
			st.lo_base = rp.lo_elems;
			st.pad = 0;
			// leftovers: < B per bucket (2^width buckets) from the stream, plus < B head keys
			rp.lo_elems += std::min<uint64_t>(e - b, ((uint64_t)1 << p.width) * (B - 1) + 2 * B);
			rp.nslots += st.slot_hi - st.slot_lo;
			rp.stripes.push_back(st);
			b = e;
		}
		p.stripe_hi = (uint32_t)rp.stripes.size();
		rp.parents.push_back(p);
	}
}

struct RoundBufs {
	Parent *parents;
	Stripe *stripes;
	uint32_t *fb, *lo_cnt, *lo_off, *lo_dst, *nfull;
	void *lo_keys;
	uint64_t *lo_vals;
	ChildArrays ca;
	ListEntry *list, *holes;
	void *xkeys;
	uint64_t *xvals;
	unsigned long long *scan_state;
	uint32_t *scan_ctr;
	Segment *next_parents;
	DirectPlan *plans; // per parent (direct placement)
};

// direct placement is tried on rounds of at most this many parents
constexpr size_t kDirectMaxParents = 4096;

template <typename K, typename V>
static void carve_round(Bump &b, const RoundPlan &rp, uint64_t small_max, RoundBufs &rb)
{
	using C = Cfg<K, V>;
	constexpr bool HV = has_val<V>::value;
	const size_t np = rp.parents.size(), ns = rp.stripes.size(), nc = rp.nchildren;
	rb.parents = b.take<Parent>(np);
	rb.stripes = b.take<Stripe>(ns);
	rb.fb = b.take<uint32_t>(ns * kP);
	rb.lo_cnt = b.take<uint32_t>(ns * kP);
	rb.lo_off = b.take<uint32_t>(ns * kP);
	rb.lo_dst = b.take<uint32_t>(ns * kP);
	rb.nfull = b.take<uint32_t>(ns);
	rb.lo_keys = b.take<K>(rp.lo_elems);
	rb.lo_vals = HV ? b.take<uint64_t>(rp.lo_elems) : nullptr;
	rb.ca.start = b.take<uint64_t>(nc);
	rb.ca.count = b.take<uint64_t>(nc);
	rb.ca.F = b.take<uint32_t>(nc);
	rb.ca.is = b.take<uint32_t>(nc);
	rb.ca.I = b.take<uint32_t>(nc);
	rb.ca.lsum = b.take<uint32_t>(nc);
	rb.ca.n_int = b.take<uint32_t>(nc);
	rb.ca.n_fr = b.take<uint32_t>(nc);
	rb.ca.n_int0 = b.take<uint32_t>(nc);
	rb.ca.cur_int = b.take<uint32_t>(nc);
	rb.ca.cur_fr = b.take<uint32_t>(nc);
	rb.ca.list_len = b.take<uint64_t>(nc);
	rb.ca.list_base = b.take<uint64_t>(nc);
	rb.ca.rpos = b.take<uint32_t>((size_t)nc * kRposStride);
	rb.ca.flags = b.take<uint32_t>(nc);
	rb.ca.rot = b.take<uint32_t>(nc);
	rb.ca.nev = b.take<uint32_t>(nc);
	rb.ca.xfirst = b.take<uint32_t>(nc);
	rb.ca.hot_cur = b.take<uint32_t>((size_t)kHotMax * kHotShards * kRposStride);
	rb.ca.lmeta = b.take<u32x4>(nc);
	rb.list = b.take<ListEntry>(rp.nslots + 1);
	// holes: tail slots (< kP + 2 per stripe) + one eviction + one excess per child
	// + the eviction pool: what it takes to reach kMinChains chains (shared out in proportion to the list
	// lengths, rounded up per child) plus the one parked block a list without a chain-ending entry needs
	const uint64_t pool = 2ull * nc + kMinChains;
	const uint64_t hmax = std::min<uint64_t>(rp.nslots, (uint64_t)ns * (kP + 2)) + 2ull * nc + pool + 1;
	rb.holes = b.take<ListEntry>(hmax);
	rb.xkeys = b.take<K>((size_t)(2 * nc + pool) * C::B);
	rb.xvals = HV ? b.take<uint64_t>((size_t)(2 * nc + pool) * C::B) : nullptr;
	const size_t ntiles = (nc + kScanTile - 1) / kScanTile + 1;
	rb.scan_state = b.take<unsigned long long>(ntiles);
	rb.scan_ctr = b.take<uint32_t>(4);
	// children that stay big: each has > small_max elements
	rb.next_parents = b.take<Segment>(rp.round_keys / (small_max + 1) + 2);
	rb.plans = b.take<DirectPlan>(np <= kDirectMaxParents ? np : 1);
}

// scan helper on the context stream
static int run_scan(msd_ctx *c, const uint64_t *in, uint64_t *out, uint64_t n,
		    unsigned long long *state, uint32_t *ctr, uint32_t *err, bool cleared = false)
{
	if (n == 0) return MSD_OK;
	const size_t ntiles = (n + kScanTile - 1) / kScanTile;
	if (!cleared) { // (the sort's rounds clear the state in round_init_kernel)
		HIPCHK(c, hipMemsetAsync(state, 0, ntiles * sizeof(unsigned long long), c->stream));
		HIPCHK(c, hipMemsetAsync(ctr, 0, 16, c->stream));
	}
	hipLaunchKernelGGL(scan_lookback_kernel, dim3((unsigned)ntiles), dim3(kScanTh), 0, c->stream, in, out, n, state, ctr, err);
	HIPCHK(c, hipGetLastError());
	return MSD_OK;
}

// ------------------------------------------------------------------ the sort

// Leaf-list entries msd_reserve() provides up front: what evenly spread keys need (two 8-bit rounds over 2^30 u32 keys
// leave 2^16 counting segments; a round reserves room for one entry per child on top of the live ones), not the most
// any input could need -- the lists grow between rounds when an input leaves more, smaller segments (lists_reserve).
// (Round 2 reserved n / 1024 + 6 n / leaf capacity entries, 1.5 times over: 189 MB of the 646 MB at 2^30 u32 keys.)
template <typename K, typename V> static uint64_t leaf_list_guess(uint64_t n)
{
	return std::min<uint64_t>(n / 8192 + 4096, n / 2 + 16);
}

template <typename K, typename V> static size_t keep_bytes_for(uint64_t n)
{
	Bump b(nullptr);
	b.take<uint8_t>(n / Cfg<K, V>::B + 2);
	b.take<uint8_t>(n / Cfg<K, V>::B + 2); // slot_full (direct placement)
	b.take<Counters>(2); // counters + scratch for the varying-bit reduction
	b.take<Segment>(n / ((uint64_t)Cfg<K, V>::SORT_TH * Cfg<K, V>::SORT_KPT) + 16); // big counting-sort segments
	if (sizeof(K) == 8) b.take<Segment>(n / ((uint64_t)Cfg<K, V>::SORT_TH * Cfg<K, V>::SORT_KPT + 1) + 2); // parents of a device-planned round
	return b.off + 4096;
}

// Per-round workspace for the two shapes the headline sizes produce: one parent
// covering everything, and 256 equal parents.  Other shapes grow the slab between
// rounds (nothing in it is live there).
template <typename K, typename V>
static size_t round_bytes_estimate(uint64_t n, int sm_count)
{
	using C = Cfg<K, V>;
	const uint64_t small_max = (uint64_t)C::SORT_TH * C::SORT_KPT;
	size_t worst = 0;
	for (int shape = 0; shape < 2; ++shape) {
		std::vector<Segment> segs;
		if (shape == 0)
			segs.push_back({ 0, n, (uint32_t)(sizeof(K) * 8), 0 });
		else if (n > 512 * small_max)
			for (int i = 0; i < 256; ++i) segs.push_back({ n / 256 * i, n / 256, (uint32_t)(sizeof(K) * 8 - 8), 0 });
		if (segs.empty()) continue;
		RoundPlan rp;
		plan_round<K, V>(segs, small_max, sm_count, rp);
		Bump b(nullptr);
		RoundBufs rb;
		carve_round<K, V>(b, rp, small_max, rb);
		worst = std::max(worst, b.off);
	}
	return worst + 4096;
}

template <typename K, typename V>
static int sort_impl(msd_ctx *c, K *keys, uint64_t *vals, uint64_t n, int end_bit,
		     // single-pass mode (msd_partition_*): one round, caller-chosen digit
		     bool single_pass, unsigned sp_shift, unsigned sp_width, uint64_t *sp_count,
		     // ... or, instead of a digit, the key's range among nsplit ascending delimiters (msd_partition_by_splitters_*)
		     const K *splitters = nullptr, uint32_t nsplit = 0,
		     // ... or a sort of nseg independent segments [seg_off[i], seg_off[i + 1]) on their low end_bit bits (msd_sort_*_segments)
		     const uint64_t *seg_off = nullptr, uint32_t nseg = 0,
		     // ... or of an explicit list of disjoint segments, each with its own number of open bits (internal: what the merge leaf rejected)
		     const std::vector<Segment> *seg_list = nullptr,
		     // stop early: the keys are only to be ordered by key >> stop_bits (msd_sort_*_top)
		     uint32_t stop_bits = 0)
{
	using C = Cfg<K, V>;
	constexpr bool HV = has_val<V>::value;
	constexpr int B = C::B;
	const uint64_t small_max = (uint64_t)C::SORT_TH * C::SORT_KPT;
	if (n == 0) return MSD_OK;
	if (!keys || (HV && !vals)) return fail(c, MSD_EINVAL, "null data pointer");
	if (((uintptr_t)keys & 15) || (HV && ((uintptr_t)vals & 15)))
		return fail(c, MSD_EINVAL, "keys/rids must be 16-byte aligned (the reference asserts the same, src/msb_64.c:2273)");
	if (end_bit < 0 || end_bit > (int)sizeof(K) * 8) return fail(c, MSD_EINVAL, "end_bit out of range");
	if (n >= ((uint64_t)1 << 36)) return fail(c, MSD_EINVAL, "n too large for 32-bit block slots");
	HIPCHK(c, hipSetDevice(c->device));
	c->stats.clear();
	phase_begin(c);

	std::vector<Segment> cur;
	if (single_pass) {
		if (sp_width < 1 || sp_width > 8 || sp_shift + sp_width > sizeof(K) * 8)
			return fail(c, MSD_EINVAL, "partition: radix_bits must be 1..8 and shift+radix_bits within the key");
		cur.push_back({ 0, n, sp_shift + sp_width, 0 });
	} else if (nseg) {
		for (uint32_t i = 0; i < nseg; ++i) {
			if (seg_off[i] > seg_off[i + 1] || seg_off[i + 1] > n) return fail(c, MSD_EINVAL, "segments: offsets must ascend and stay within n");
			if (end_bit > 0 && seg_off[i + 1] - seg_off[i] > 1) cur.push_back({ seg_off[i], seg_off[i + 1] - seg_off[i], (uint32_t)end_bit, 0 });
		}
	} else if (seg_list) {
		for (auto &sg : *seg_list) {
			if (sg.start + sg.count > n || sg.bits > sizeof(K) * 8) return fail(c, MSD_EINVAL, "segments: a segment lies outside the array");
			if (sg.bits > 0 && sg.count > 1) cur.push_back(sg);
		}
		std::sort(cur.begin(), cur.end(), [](const Segment &a, const Segment &b) { return a.start < b.start; });
	} else if (end_bit > 0 && n > 1 && (uint32_t)end_bit > stop_bits)
		cur.push_back({ 0, n, (uint32_t)end_bit, 0 });
	const bool segmented = nseg != 0 || seg_list != nullptr;

	// ---- buffers that live for the whole call
	{
		int rc = keep_reserve(c, keep_bytes_for<K, V>(n), true);
		if (!rc) rc = pinned_reserve(c, 1 << 16);
		if (!rc) rc = lists_reserve(c, 4096, 0, 0);
		if (rc) return rc;
	}
	Bump kb(c->keep);
	uint8_t *block_map = kb.take<uint8_t>(n / B + 2);
	uint8_t *slot_full = kb.take<uint8_t>(n / B + 2);
	Counters *ctr = kb.take<Counters>(2);
	const uint32_t big_cap = (uint32_t)std::min<uint64_t>(n / small_max + 16, 0x7FFFFFFFu);
	Segment *big = kb.take<Segment>(big_cap);
	Segment *dev_list = sizeof(K) == 8 ? kb.take<Segment>(n / (small_max + 1) + 2) : nullptr; // parents of a device-planned round
	uint32_t dev_np = 0;
	Segment *small = c->lists, *small_count = c->lists + 3 * c->lists_cap;
	HIPCHK(c, hipMemsetAsync(ctr, 0, sizeof(Counters), c->stream));
	// keys without payload whose last <= 16 bits are open are finished by the counting sort
	const uint32_t count_bits = HV ? (uint32_t)kLeafCountBits : (uint32_t)kCountMaxBits;

	// ---- leading-bit skipping: a cheap strided sample decides whether an exact OR/AND pass over
	// all keys can pay off (it does when whole leading digits are constant, e.g. keys whose upper
	// half is zero); all-equal inputs are finished here.
	unsigned long long *vres = reinterpret_cast<unsigned long long *>(ctr + 1);
	auto vres_init = [&]() -> int { // OR accumulator 0, AND accumulator all ones (no host round trip)
		HIPCHK(c, hipMemsetAsync(vres, 0x00, sizeof(unsigned long long), c->stream));
		HIPCHK(c, hipMemsetAsync(vres + 1, 0xFF, sizeof(unsigned long long), c->stream));
		return MSD_OK;
	};
	auto run_vary = [&](uint64_t stride, uint64_t *vary_out) -> int {
		int rc0 = vres_init();
		if (rc0) return rc0;
		const uint64_t cnt = (n + stride - 1) / stride;
		const unsigned grid = (unsigned)std::min<uint64_t>((uint64_t)c->sm_count * 8, (cnt + 255) / 256);
		hipLaunchKernelGGL((vary_kernel<K>), dim3(grid), dim3(256), 0, c->stream, keys, n, stride, vres);
		HIPCHK(c, hipGetLastError());
		HIPCHK(c, hipMemcpyAsync(c->pinned, vres, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipStreamSynchronize(c->stream));
		const unsigned long long *h = (const unsigned long long *)c->pinned;
		*vary_out = h[0] ^ h[1];
		return MSD_OK;
	};
	const uint64_t low_mask = end_bit >= 64 ? ~0ull : ((1ull << end_bit) - 1ull);
	// A skip decided from the sample alone is `unverified`: the rounds only permute keys, so the exact check may come
	// later -- on the exact histogram pass of the second round if that reads every key (it costs that pass nothing:
	// 1.4 ms less for 2^30 tuples with 32 constant key bits), else in a pass of its own before the leaves, which
	// re-generate keys from a common prefix and must not run on a wrong one.  If the check fails (some key differs
	// in a bit the sample found constant) the sort starts over on all bits the exact pass found varying: the data is
	// still the same multiset.
	bool unverified = false;
	uint64_t claimed_const = 0; // bits below end_bit the sample found constant
	if (!single_pass && !segmented && !cur.empty() && n >= 4096) {
		uint64_t vary = 0;
		int rc = run_vary(std::max<uint64_t>(1, n / 8192), &vary);
		if (rc) return rc;
		vary &= low_mask;
		const int top_sample = vary ? 64 - __builtin_clzll(vary) : 0;
		if (top_sample + 8 <= end_bit) { // at least one whole leading digit looks constant
			if (top_sample > 0 && c->direct_mode != 0 && n >= c->direct_min && n >= ((uint64_t)1 << 24)) {
				unverified = true;
				claimed_const = low_mask & ~(((uint64_t)1 << top_sample) - 1ull);
				cur[0].bits = (uint32_t)top_sample;
				set_stat(c, "skipped_bits", (uint64_t)(end_bit - top_sample));
			} else { // make sure at once
				rc = run_vary(1, &vary);
				if (rc) return rc;
				vary &= low_mask;
				const int top = vary ? 64 - __builtin_clzll(vary) : 0;
				set_stat(c, "skipped_bits", (uint64_t)(end_bit - top));
				if (top == 0)
					cur.clear(); // every key is the same on the bits in question: already sorted
				else
					cur[0].bits = (uint32_t)top;
			}
		}
		phase_mark(c, "bit skip");
		if (!cur.empty() && cur[0].bits <= stop_bits) cur.clear(); // (a sort that stops above every varying bit)
	}

	uint32_t nsmall_host = 0, ncount_host = 0, nbig_host = 0;
	if (segmented && !cur.empty()) {
		// segments that need no partition round go to the leaf lists at once, by collect_kernel's rules
		std::vector<Segment> l_small, l_count, l_big, parents;
		for (auto &sg : cur) {
			const bool countable = !HV && sg.bits <= count_bits;
			if (countable && sg.count >= 64 && sg.count <= std::max<uint64_t>(small_max, kCountMedMax))
				l_count.push_back(sg);
			else if (sg.count > small_max) {
				if (countable && sg.count < 0xFFFF0000ull && l_big.size() < big_cap) l_big.push_back(sg); else parents.push_back(sg);
			} else
				l_small.push_back(sg);
		}
		int rc = lists_reserve(c, std::max(l_small.size(), l_count.size()) + 16, 0, 0);
		if (!rc) rc = pinned_reserve(c, (l_small.size() + l_count.size() + l_big.size()) * sizeof(Segment) + sizeof(Counters));
		if (rc) return rc;
		small = c->lists;
		small_count = c->lists + 3 * c->lists_cap;
		HIPCHK(c, hipStreamSynchronize(c->stream));
		Segment *ps = (Segment *)c->pinned;
		auto up = [&](const std::vector<Segment> &v, Segment *dst) -> int {
			if (v.empty()) return MSD_OK;
			memcpy(ps, v.data(), v.size() * sizeof(Segment));
			HIPCHK(c, hipMemcpyAsync(dst, ps, v.size() * sizeof(Segment), hipMemcpyHostToDevice, c->stream));
			ps += v.size();
			return MSD_OK;
		};
		if ((rc = up(l_small, small)) || (rc = up(l_count, small_count)) || (rc = up(l_big, big))) return rc;
		nsmall_host = (uint32_t)l_small.size();
		ncount_host = (uint32_t)l_count.size();
		nbig_host = (uint32_t)l_big.size();
		Counters *hc0 = (Counters *)ps; // the lists' counters continue from here
		memset(hc0, 0, sizeof(Counters));
		hc0->nsmall = nsmall_host;
		hc0->ncount = ncount_host;
		hc0->nbig = nbig_host;
		HIPCHK(c, hipMemcpyAsync(ctr, hc0, sizeof(Counters), hipMemcpyHostToDevice, c->stream));
		cur = parents;
	}
	if constexpr (!HV) { // <= 16 open bits from the start (small key range): no partition round at all
		if (!single_pass && !segmented && !cur.empty() && n > small_max && cur[0].bits <= count_bits && n < 0xFFFF0000ull) {
			HIPCHK(c, hipStreamSynchronize(c->stream));
			memcpy(c->pinned, &cur[0], sizeof(Segment));
			HIPCHK(c, hipMemcpyAsync(big, c->pinned, sizeof(Segment), hipMemcpyHostToDevice, c->stream));
			nbig_host = 1;
			cur.clear();
		}
	}
	if (!single_pass && !segmented && !cur.empty() && n <= small_max) { // fits LDS: no partition round at all
		HIPCHK(c, hipStreamSynchronize(c->stream));
		memcpy(c->pinned, &cur[0], sizeof(Segment));
		HIPCHK(c, hipMemcpyAsync(small, c->pinned, sizeof(Segment), hipMemcpyHostToDevice, c->stream));
		nsmall_host = 1;
		cur.clear();
	}

	int round = 0;
	// the previous round placed its blocks directly (its digit was evenly spread); segments handed in by the caller are
	// taken to be such a round's children (the shards of a multi-GPU sort after their top-digit pass and exchange)
	bool prev_direct = nseg != 0;
	// the sort starts over from here if the exact check behind a sampled leading-bit skip fails: on `exact_vary`, the
	// bits that really vary
	auto start_over = [&](uint64_t exact_vary) -> int {
		exact_vary &= low_mask;
		const int top = exact_vary ? 64 - __builtin_clzll(exact_vary) : 0;
		set_stat(c, "skipped_bits", (uint64_t)(end_bit - top));
		add_stat(c, "bit_skip_restarts", 1);
		cur.clear();
		if (top > 0 && (uint32_t)top > stop_bits) cur.push_back({ 0, n, (uint32_t)top, 0 });
		nsmall_host = ncount_host = nbig_host = 0;
		dev_np = 0;
		prev_direct = false;
		unverified = false;
		HIPCHK(c, hipMemsetAsync(ctr, 0, sizeof(Counters), c->stream));
		return MSD_OK;
	};
	bool leaf17_ok = true; // (tuples) leaf17_kernel has rejected nothing yet in this call
	bool again = true;
	while (again) {
	again = false;
	while (!cur.empty() || dev_np) {
		// ---- segments that fit the registers of one workgroup take ONE register-resident pass (msd_regpart.hpp) instead
		// of a general round: the last partition round of the tuple sort (65536 parents of about 2^14 tuples at 2^30)
		if constexpr (sizeof(K) == 8) {
			if (!single_pass && c->regpart) {
				// (dev_np: the previous round left only parents that fit, and their list stayed on the device -- it is
				// planned there too, regpart_plan_kernel; otherwise the host sorts the fitting parents out of its list)
				std::vector<Segment> fit, rest;
				if (!dev_np)
					for (auto &sg : cur) (sg.count + 1 <= kRpCap && sg.count > small_max ? fit : rest).push_back(sg);
				if constexpr (HV) {
					// ---- tuples: such a segment is FINISHED in one pass by leaf17_kernel (msd_leaf17.hpp: read once, sorted in
					// registers and LDS, written once) instead of a register partition + the small leaves; what it rejects
					// (a group of > 48 tuples equal on the counted bits) takes that way.  A leaf must not run behind an
					// unconfirmed leading-bit skip.
					// (segments with <= 16 open bits -- tuples whose upper key half is constant, config 5b -- too: the leaf counts up to
					// 16 bits, such a segment has no groups to put in order at all)
					if (c->leaf17 && leaf17_ok && !unverified && (dev_np || fit.size() >= 64 || (!fit.empty() && rest.empty()))) {
						const bool on_device = dev_np != 0;
						const uint32_t np = on_device ? dev_np : (uint32_t)fit.size();
						dev_np = 0;
						int rc = slab_reserve(c, 2 * (size_t)np * sizeof(Segment) + 4096);
						if (!rc) rc = pinned_reserve(c, (size_t)np * sizeof(Segment) + 4096);
						if (rc) return rc;
						Segment *d_segs = reinterpret_cast<Segment *>(c->slab), *d_rej = d_segs + np;
						if (on_device)
							HIPCHK(c, hipMemcpyAsync(d_segs, dev_list, (size_t)np * sizeof(Segment), hipMemcpyDeviceToDevice, c->stream));
						else {
							HIPCHK(c, hipStreamSynchronize(c->stream)); // the staging buffer may still be in flight
							memcpy(c->pinned, fit.data(), (size_t)np * sizeof(Segment));
							HIPCHK(c, hipMemcpyAsync(d_segs, c->pinned, (size_t)np * sizeof(Segment), hipMemcpyHostToDevice, c->stream));
						}
						HIPCHK(c, hipMemsetAsync(&ctr->nslow2, 0, sizeof(uint32_t), c->stream));
						HIPCHK(c, hipMemsetAsync(&ctr->l17_slow, 0, sizeof(uint32_t), c->stream));
						phase_mark(c, "plan+upload");
						hipLaunchKernelGGL((leaf17_kernel<V>), dim3(std::min<uint32_t>(np, (uint32_t)c->sm_count)), dim3(kL17Th), kL17Lds, c->stream,
								   (uint64_t *)keys, vals, (const Segment *)d_segs, np, d_rej, &ctr->nslow2, ctr, 0u);
						HIPCHK(c, hipGetLastError());
						phase_mark(c, "leaf17");
						Counters hc;
						HIPCHK(c, hipMemcpyAsync(c->pinned, ctr, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
						HIPCHK(c, hipStreamSynchronize(c->stream));
						memcpy(&hc, c->pinned, sizeof hc);
						if (hc.errors) return fail(c, MSD_EINTERNAL, "leaf17: %u internal invariant violations (checks 0x%x)", hc.errors, hc.err_sites);
						add_stat(c, "leaf17_segments", np - hc.nslow2);
						add_stat(c, "leaf17_slow_segments", hc.l17_slow);
						cur = rest;
						if (hc.nslow2) { // rejected segments: the register partition + the small leaves finish them
							rc = pinned_reserve(c, (size_t)hc.nslow2 * sizeof(Segment));
							if (rc) return rc;
							HIPCHK(c, hipMemcpyAsync(c->pinned, d_rej, (size_t)hc.nslow2 * sizeof(Segment), hipMemcpyDeviceToHost, c->stream));
							HIPCHK(c, hipStreamSynchronize(c->stream));
							cur.insert(cur.end(), (Segment *)c->pinned, (Segment *)c->pinned + hc.nslow2);
							std::sort(cur.begin(), cur.end(), [](const Segment &a, const Segment &b) { return a.start < b.start; });
							leaf17_ok = false; // (for the rest of this call)
							add_stat(c, "leaf17_rejected", hc.nslow2);
						}
						phase_mark(c, "readback");
						continue;
					}
				}
				if (dev_np || fit.size() >= 64 || (!fit.empty() && rest.empty())) {
					const bool on_device = dev_np != 0;
					std::vector<Parent> ps(fit.size());
					uint32_t nc = 0;
					for (size_t i = 0; i < fit.size(); ++i) {
						Parent &p = ps[i];
						p.start = fit[i].start;
						p.count = fit[i].count;
						p.width = regpart_width(fit[i].count, fit[i].bits, small_max);
						p.shift = fit[i].bits - p.width;
						p.child_base = nc;
						p.stripe_lo = p.stripe_hi = 0;
						p.pad = 0;
						nc += 1u << p.width;
					}
					if (on_device) nc = dev_np << regpart_width(kRpCap, 64, small_max); // (an upper bound: the widest digit of the rule)
					const uint32_t np = on_device ? dev_np : (uint32_t)ps.size();
					dev_np = 0;
					const size_t next_cap = (size_t)nc + 2; // (children above the leaf capacity: none on sane input, all at worst)
					Bump sz(nullptr), *bp = &sz;
					Parent *d_parents = nullptr;
					ChildArrays ca = {};
					Segment *d_next = nullptr;
					uint32_t *d_scr = nullptr;
					auto carve = [&]() {
						d_parents = bp->take<Parent>(np);
						ca.start = bp->take<uint64_t>(nc);
						ca.count = bp->take<uint64_t>(nc);
						d_next = bp->take<Segment>(next_cap);
						d_scr = bp->take<uint32_t>(64);
					};
					carve();
					int rc = slab_reserve(c, sz.off + 4096);
					if (!rc) rc = lists_reserve(c, (size_t)std::max(nsmall_host, ncount_host) + nc + 16, nsmall_host, ncount_host);
					if (!rc) rc = pinned_reserve(c, std::max<size_t>(on_device ? 0 : np * sizeof(Parent), 256 + 2048 * sizeof(Segment)));
					if (rc) return rc;
					small = c->lists;
					small_count = c->lists + 3 * c->lists_cap;
					Bump real(c->slab);
					bp = &real;
					carve();
					hipLaunchKernelGGL(round_init_kernel, dim3(1), dim3(256), 0, c->stream, ctr, d_scr + 16, (uint64_t)0,
							   reinterpret_cast<unsigned long long *>(d_scr + 32), (uint64_t)0, d_scr);
					if (on_device)
						hipLaunchKernelGGL(regpart_plan_kernel, dim3((np + 255) / 256), dim3(256), 0, c->stream, (const Segment *)dev_list, np,
								   small_max, d_parents, ctr);
					else {
						HIPCHK(c, hipStreamSynchronize(c->stream)); // the staging buffer may still be in flight
						memcpy(c->pinned, ps.data(), np * sizeof(Parent));
						HIPCHK(c, hipMemcpyAsync(d_parents, c->pinned, np * sizeof(Parent), hipMemcpyHostToDevice, c->stream));
					}
					phase_mark(c, "plan+upload");
					hipLaunchKernelGGL((regpart_kernel<V>), dim3(std::min<uint32_t>(np, (uint32_t)c->sm_count)), dim3(kRpTh), kRpLds, c->stream,
							   (uint64_t *)keys, vals, (const Parent *)d_parents, np, ca, ctr);
					phase_mark(c, "A register partition");
					const uint32_t wmax_rp = regpart_width(kRpCap, 64, small_max); // (the widest digit of the rule)
					hipLaunchKernelGGL(collect_kernel, dim3((np + (256u >> wmax_rp) - 1) / (256u >> wmax_rp)), dim3(256), 0, c->stream, (const Parent *)d_parents, np, wmax_rp, ca, small_max, small_max,
							   (uint32_t)std::min<size_t>(c->lists_cap, 0xFFFFFFFFu), count_bits, d_next, small, small_count,
							   HV ? (Segment *)nullptr : big, big_cap, ctr, (uint64_t *)nullptr, nc, stop_bits);
					HIPCHK(c, hipGetLastError());
					phase_mark(c, "C cleanup");
					Counters hc;
					const size_t ahead = std::min<size_t>(next_cap, 2048);
					HIPCHK(c, hipMemcpyAsync(c->pinned, ctr, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
					HIPCHK(c, hipMemcpyAsync((char *)c->pinned + 256, d_next, ahead * sizeof(Segment), hipMemcpyDeviceToHost, c->stream));
					HIPCHK(c, hipStreamSynchronize(c->stream));
					memcpy(&hc, c->pinned, sizeof hc);
					if (hc.errors) return fail(c, MSD_EINTERNAL, "register partition round: %u internal invariant violations (checks 0x%x)", hc.errors, hc.err_sites);
					nsmall_host = hc.nsmall;
					ncount_host = hc.ncount;
					nbig_host = hc.nbig;
					add_stat(c, "rounds", 1);
					add_stat(c, "regpart_rounds", 1);
					add_stat(c, "parents", np);
					add_stat(c, "children", nc);
					cur = rest;
					if (hc.next_parents) {
						if (hc.next_parents > ahead) {
							rc = pinned_reserve(c, (size_t)hc.next_parents * sizeof(Segment));
							if (rc) return rc;
							HIPCHK(c, hipMemcpyAsync(c->pinned, d_next, (size_t)hc.next_parents * sizeof(Segment), hipMemcpyDeviceToHost, c->stream));
							HIPCHK(c, hipStreamSynchronize(c->stream));
							cur.insert(cur.end(), (Segment *)c->pinned, (Segment *)c->pinned + hc.next_parents);
						} else
							cur.insert(cur.end(), (Segment *)((char *)c->pinned + 256), (Segment *)((char *)c->pinned + 256) + hc.next_parents);
						std::sort(cur.begin(), cur.end(), [](const Segment &a, const Segment &b) { return a.start < b.start; });
					}
					phase_mark(c, "readback");
					prev_direct = false;
					++round;
					continue;
				}
			}
		}
		RoundPlan rp;
		plan_round<K, V>(cur, small_max, c->sm_count, rp, count_bits, single_pass ? sp_width : 0u, splitters ? nsplit : 0u);
		RoundBufs rb;
		{
			Bump sz(nullptr);
			carve_round<K, V>(sz, rp, small_max, rb);
			size_t need = sz.off + 4096;
			if (round == 0 && !single_pass) need = std::max(need, round_bytes_estimate<K, V>(n, c->sm_count));
			int rc = slab_reserve(c, need, round == 0); // between rounds nothing in the slab is live
			if (rc) return rc;
			Bump b(c->slab);
			carve_round<K, V>(b, rp, small_max, rb);
		}
		const uint32_t np = (uint32_t)rp.parents.size(), ns = (uint32_t)rp.stripes.size(), nc = rp.nchildren;
		{ // every child of this round may become a leaf
			int rc = lists_reserve(c, (size_t)std::max(nsmall_host, ncount_host) + nc + 16, nsmall_host, ncount_host);
			if (rc) return rc;
			small = c->lists;
			small_count = c->lists + 3 * c->lists_cap;
		}
		const uint32_t small_cap = (uint32_t)std::min<size_t>(c->lists_cap, 0xFFFFFFFFu);
		{ // per-round counters, per-parent plans, scan state: one launch
			const uint64_t plan_words = (np <= kDirectMaxParents ? np : 1) * sizeof(DirectPlan) / sizeof(uint32_t);
			const uint64_t ntiles = (nc + kScanTile - 1) / kScanTile + 1;
			const unsigned grid = (unsigned)std::min<uint64_t>(1024, (std::max(plan_words, ntiles) + 255) / 256 + 1);
			hipLaunchKernelGGL(round_init_kernel, dim3(grid), dim3(256), 0, c->stream, ctr, reinterpret_cast<uint32_t *>(rb.plans),
					   plan_words, rb.scan_state, ntiles, rb.scan_ctr);
		}
		// ---- upload tables
		{
			const size_t bytes = np * sizeof(Parent) + ns * sizeof(Stripe);
			int rc = pinned_reserve(c, bytes);
			if (rc) return rc;
			HIPCHK(c, hipStreamSynchronize(c->stream)); // the staging buffer may still be in flight
			memcpy(c->pinned, rp.parents.data(), np * sizeof(Parent));
			memcpy((char *)c->pinned + np * sizeof(Parent), rp.stripes.data(), ns * sizeof(Stripe));
			HIPCHK(c, hipMemcpyAsync(rb.parents, c->pinned, np * sizeof(Parent), hipMemcpyHostToDevice, c->stream));
			HIPCHK(c, hipMemcpyAsync(rb.stripes, (char *)c->pinned + np * sizeof(Parent), ns * sizeof(Stripe), hipMemcpyHostToDevice, c->stream));
		}
		phase_mark(c, "plan+upload");

		// ---- A: classify (histogram falls out of it)
		bool tried_direct = false, hist_checks = false;
		// Direct placement (DESIGN.md section 2, A'): the first round from a sample, later rounds -- only
		// after a direct first round -- from exact counts (a read-only pass).
		// (the read schedule hands a bucket one slot per tile: with fewer than 256 buckets the tiles
		// are not filled, so narrower digits keep the streaming kernel unless forced)
		bool try_direct = c->direct_mode != 0 && rp.round_keys >= c->direct_min && !splitters;
		for (size_t i = 0; i < np && try_direct; ++i) try_direct = rp.parents[i].width == 8 || c->direct_mode == 2;
		if (try_direct && np > 1) {
			try_direct = prev_direct && np <= kDirectMaxParents;
			for (size_t i = 0; i < np && try_direct; ++i) try_direct = rp.parents[i].count >= c->direct_min_parent;
		}
		// A workgroup of a direct round reads its pieces, not its stripe: up to one slot per bucket more than
		// the stripe holds.  Its leftovers (< B per bucket + head/tail) must fit the stripe's leftover area,
		// which is capped by the stripe's own size: no direct placement for stripes smaller than that bound.
		for (size_t i = 0; i < ns && try_direct; ++i)
			try_direct = rp.stripes[i].end - rp.stripes[i].begin >= (((uint64_t)1 << rp.parents[rp.stripes[i].parent].width) * (B - 1) + 2 * B);
		uint64_t max_piece = 0; // a piece is at most one stripe's share of its parent's slots; the kernel counts it in 16 bits
		for (size_t i = 0; i < np && try_direct; ++i)
			max_piece = std::max<uint64_t>(max_piece, rp.parents[i].count / B / (rp.parents[i].stripe_hi - rp.parents[i].stripe_lo) + 2);
		if (try_direct && max_piece < 65535) {
			if (np == 1) {
				// sample about 2^22 keys or more, as runs of 256 spread evenly over the parent
				const uint64_t nruns = rp.parents[0].count / 256;
				const uint32_t every = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(1, nruns / 16384));
				const uint32_t sgrid = (uint32_t)std::min<uint64_t>(1024, std::max<uint64_t>(1, nruns / every / 4));
				hipLaunchKernelGGL((direct_sample_kernel<K>), dim3(sgrid), dim3(256), 0, c->stream, (const K *)keys, rb.parents, rb.plans, every);
			} else {
				// (the exact check behind a sampled leading-bit skip rides on this pass if it reads every key)
				hist_checks = unverified && rp.round_keys == n;
				if (hist_checks) {
					int rcv = vres_init();
					if (rcv) return rcv;
					add_stat(c, "bit_skip_checked_by_histogram", 1);
				}
				hipLaunchKernelGGL((direct_hist_kernel<K>), dim3(ns), dim3(1024), 0, c->stream, (const K *)keys, rb.stripes, rb.parents, rb.plans,
						   hist_checks ? vres : (unsigned long long *)nullptr);
			}
			hipLaunchKernelGGL((direct_plan_kernel<B>), dim3(np), dim3(256), 0, c->stream, rb.parents, rb.plans, ctr);
			phase_mark(c, np == 1 ? "A sample" : "A histogram");
			// The plan's verdict (Counters::direct_uneven: some parent's children are too unequal, or its keys come in
			// runs) stays on the device: the direct kernel returns at once if it is non-zero, the streaming kernel
			// launched behind it if it is zero.  The host learns it with the round's summary.
			tried_direct = true;
			const uint32_t force = c->direct_mode == 2 ? 1u : 0u;
			{
				constexpr size_t direct_lds = Direct2Lds<K, V>::bytes;
				hipLaunchKernelGGL((classify_direct2_kernel<K, V>), dim3(ns), dim3(Direct2Cfg<K, V>::TH), direct_lds, c->stream,
						   keys, vals, rb.stripes, rb.parents, (const DirectPlan *)rb.plans, block_map, slot_full,
						   rb.fb, rb.lo_cnt, rb.lo_off, (K *)rb.lo_keys, rb.lo_vals, rb.nfull, ctr, force);
			}
			HIPCHK(c, hipGetLastError());
			phase_mark(c, "A classify direct");
		}
		if (!tried_direct || c->direct_mode != 2) {
			const uint32_t *run_if = tried_direct ? (const uint32_t *)&ctr->direct_uneven : (const uint32_t *)nullptr;
			if (c->stream_kernel == 2) { // the lean tile loop (msd_stream2.hpp)
				constexpr size_t s2_lds = Stream2Lds<K, V>::bytes;
				if (splitters)
					hipLaunchKernelGGL((classify_stream2_kernel<K, V, true>), dim3(ns), dim3(Stream2Cfg<K, V>::TH), s2_lds + kP * sizeof(K), c->stream,
							   keys, vals, rb.stripes, rb.parents, block_map, rb.fb, rb.lo_cnt, rb.lo_off,
							   (K *)rb.lo_keys, rb.lo_vals, rb.nfull, splitters, run_if);
				else
					hipLaunchKernelGGL((classify_stream2_kernel<K, V, false>), dim3(ns), dim3(Stream2Cfg<K, V>::TH), s2_lds, c->stream,
							   keys, vals, rb.stripes, rb.parents, block_map, rb.fb, rb.lo_cnt, rb.lo_off,
							   (K *)rb.lo_keys, rb.lo_vals, rb.nfull, (const K *)nullptr, run_if);
			} else { // round 2's kernel (A/B comparisons)
				constexpr size_t classify_lds = ClassifyLds<K, V>::bytes;
				if (splitters)
					hipLaunchKernelGGL((classify_kernel<K, V, true>), dim3(ns), dim3(C::TH), classify_lds + kP * sizeof(K), c->stream,
							   keys, vals, rb.stripes, rb.parents, block_map, rb.fb, rb.lo_cnt, rb.lo_off,
							   (K *)rb.lo_keys, rb.lo_vals, rb.nfull, splitters, run_if);
				else
					hipLaunchKernelGGL((classify_kernel<K, V, false>), dim3(ns), dim3(C::TH), classify_lds, c->stream,
							   keys, vals, rb.stripes, rb.parents, block_map, rb.fb, rb.lo_cnt, rb.lo_off,
							   (K *)rb.lo_keys, rb.lo_vals, rb.nfull, (const K *)nullptr, run_if);
			}
			HIPCHK(c, hipGetLastError());
		}
		const uint8_t *full_map = tried_direct ? (const uint8_t *)slot_full : (const uint8_t *)nullptr;
		const uint32_t force_map = c->direct_mode == 2 ? 1u : 0u;
		phase_mark(c, "A classify");

		// ---- block metadata: child geometry, misplaced-block lists, holes
		hipLaunchKernelGGL((child_scan_kernel<B>), dim3(np), dim3(1024), 0, c->stream, rb.parents, rb.fb, rb.lo_cnt, rb.lo_dst, rb.ca);
		hipLaunchKernelGGL((slot_classify_kernel<false>), dim3(ns * kSlotParts), dim3(256), 0, c->stream, rb.stripes, rb.parents,
				   block_map, rb.nfull, rb.ca, rb.list, rb.holes, ctr, full_map, force_map);
		hipLaunchKernelGGL(list_prepare_kernel, dim3((nc + 255) / 256), dim3(256), 0, c->stream, nc, rb.ca, ctr,
				   (uint32_t)std::min<uint64_t>(rp.nslots, 0xFFFFFFFFu), (uint32_t)(2 * nc + kMinChains));
		{
			int rc = run_scan(c, rb.ca.list_len, rb.ca.list_base, nc, rb.scan_state, rb.scan_ctr, &ctr->errors, true);
			if (rc) return rc;
		}
		hipLaunchKernelGGL((slot_classify_kernel<true>), dim3(ns * kSlotParts), dim3(256), 0, c->stream, rb.stripes, rb.parents,
				   block_map, rb.nfull, rb.ca, rb.list, rb.holes, ctr, full_map, force_map);
		// per child: up to 64 waves when there are few children, one thread when there are very many
		const uint32_t evict_waves = nc > 16384 ? 0u : (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(1, 16384 / nc));
		const unsigned evict_grid = evict_waves ? (unsigned)(((uint64_t)nc * evict_waves + 3) / 4) : (unsigned)((nc + 255) / 256);
		hipLaunchKernelGGL((evict_kernel<K, V>), dim3(evict_grid), dim3(256), 0, c->stream, nc, rb.ca,
				   rb.list, rb.holes, ctr, keys, vals, (K *)rb.xkeys, rb.xvals, evict_waves);
		HIPCHK(c, hipGetLastError());
		phase_mark(c, "B metadata");

		// ---- B: block permutation
		{
			// Exactly the workgroups the chip holds at once: a wave's first 64 chain starts are its own by position and the
			// rest come from the cursor as its chains end, so every hole is in the hands of a running wave from the start.
			// (Twice as many workgroups: those of the second half whose share held holes started when the first finished
			// -- 2^30 Zipf keys: 1.3 ms where a wave's own work takes 0.7.)
			const unsigned grid = (unsigned)std::min<uint64_t>((uint64_t)c->sm_count * c->chains_per_cu[HV ? 2 : sizeof(K) == 8 ? 1 : 0],
									 std::max<uint64_t>(1, (rp.nslots + 255) / 256));
			hipLaunchKernelGGL(list_pack_kernel, dim3((nc + 255) / 256), dim3(256), 0, c->stream, nc, rb.ca);
			hipLaunchKernelGGL((chains_kernel<K, V>), dim3(grid), dim3(256), 0, c->stream, rb.ca, rb.list, rb.holes, ctr,
					   keys, vals, (K *)rb.xkeys, rb.xvals, (uint32_t)(n / B), (uint32_t)(4 * nc + kMinChains));
			hipLaunchKernelGGL(chains_verify_kernel, dim3((nc + 255) / 256), dim3(256), 0, c->stream, nc, rb.ca, ctr);
			HIPCHK(c, hipGetLastError());
		}
		phase_mark(c, "B block permute");

		// ---- C: cleanup
		hipLaunchKernelGGL((cleanup_kernel<K, V>), dim3(ns), dim3(256), 0, c->stream, rb.stripes, rb.parents, rb.lo_cnt,
				   rb.lo_off, rb.lo_dst, rb.ca, (const K *)rb.lo_keys, rb.lo_vals, keys, vals);
		hipLaunchKernelGGL((excess_kernel<K, V>), dim3(nc), dim3(64), 0, c->stream, nc, rb.ca, (const K *)rb.xkeys, rb.xvals, keys, vals);
		uint32_t wmax = 1;
		for (size_t i = 0; i < np; ++i) wmax = std::max(wmax, rp.parents[i].width);
		hipLaunchKernelGGL(collect_kernel, dim3((np + (256u >> wmax) - 1) / (256u >> wmax)), dim3(256), 0, c->stream, (const Parent *)rb.parents, np, wmax, rb.ca,
				   single_pass ? ~0ull : small_max, (HV || single_pass) ? small_max : std::max<uint64_t>(small_max, kCountMedMax),
				   small_cap, single_pass ? 0u : count_bits,
				   rb.next_parents, small, small_count, (HV || single_pass) ? (Segment *)nullptr : big, big_cap, ctr,
				   (single_pass && sp_count) ? sp_count : (uint64_t *)nullptr, splitters ? nsplit + 1u : nc, stop_bits);
		HIPCHK(c, hipGetLastError());
		phase_mark(c, "C cleanup");

		// ---- round summary + next parents back to the host (which plans the next round): ONE synchronisation; the
		// first kReadAhead next parents travel with the counters (more than that only on the odd input: fetched then)
		constexpr size_t kReadAhead = 2048, kSegOff = 256;
		const size_t np_cap = rp.round_keys / (small_max + 1) + 2, ahead = single_pass ? 0 : std::min(np_cap, kReadAhead);
		{
			int rc = pinned_reserve(c, kSegOff + kReadAhead * sizeof(Segment));
			if (rc) return rc;
		}
		Counters hc;
		HIPCHK(c, hipMemcpyAsync(c->pinned, ctr, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
		if (ahead)
			HIPCHK(c, hipMemcpyAsync((char *)c->pinned + kSegOff, rb.next_parents, ahead * sizeof(Segment), hipMemcpyDeviceToHost, c->stream));
		static_assert(sizeof(Counters) + 2 * sizeof(unsigned long long) <= kSegOff, "counters and the OR/AND words share the head of the staging buffer");
		if (hist_checks)
			HIPCHK(c, hipMemcpyAsync((char *)c->pinned + kSegOff - 2 * sizeof(unsigned long long), vres, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipStreamSynchronize(c->stream));
		memcpy(&hc, c->pinned, sizeof hc);
		if (hc.errors) return fail(c, MSD_EINTERNAL, "round %d: %u internal invariant violations (checks 0x%x: bit = site of msd_note_error in csrc/; 0 = a scan tile's look-back timed out; "
									"%u parents, %u stripes, direct placement %s)",
							     round, hc.errors, hc.err_sites, np, ns, tried_direct ? (hc.direct_uneven ? "declined" : "used") : "not tried");
		if (hist_checks) {
			const unsigned long long *h = (const unsigned long long *)((char *)c->pinned + kSegOff - 2 * sizeof(unsigned long long));
			const uint64_t exact_vary = h[0] ^ h[1];
			if (exact_vary & claimed_const) { // some key differs in a bit the sample found constant: all over again, on every varying bit
				int rcs = start_over(exact_vary);
				if (rcs) return rcs;
				round = 0;
				phase_mark(c, "readback");
				continue;
			}
			unverified = false;
		}
		const bool direct = tried_direct && (hc.direct_uneven == 0 || c->direct_mode == 2);
		if (direct) add_stat(c, "direct_rounds", 1);
		prev_direct = direct;
		nsmall_host = hc.nsmall;
		ncount_host = hc.ncount;
		nbig_host = hc.nbig;
		add_stat(c, "rounds", 1);
		add_stat(c, "parents", np);
		add_stat(c, "stripes", ns);
		add_stat(c, "children", nc);
		add_stat(c, "slots", rp.nslots);
		add_stat(c, "holes", hc.nholes);
		set_stat(c, "chain_steps", hc.chain_steps);
		cur.clear();
		if (single_pass) break;
		bool stays_on_device = false;
		if constexpr (sizeof(K) == 8) {
			// every next parent fits a workgroup's registers: the list stays on the device and is planned there
			if (c->regpart && hc.next_parents >= 64 && (uint64_t)hc.next_max + 1 <= kRpCap) {
				HIPCHK(c, hipMemcpyAsync(dev_list, rb.next_parents, (size_t)hc.next_parents * sizeof(Segment), hipMemcpyDeviceToDevice, c->stream));
				dev_np = hc.next_parents;
				stays_on_device = true;
			}
		}
		if (hc.next_parents && !stays_on_device) {
			if (hc.next_parents <= ahead)
				cur.assign((Segment *)((char *)c->pinned + kSegOff), (Segment *)((char *)c->pinned + kSegOff) + hc.next_parents);
			else {
				int rc = pinned_reserve(c, (size_t)hc.next_parents * sizeof(Segment));
				if (rc) return rc;
				HIPCHK(c, hipMemcpyAsync(c->pinned, rb.next_parents, (size_t)hc.next_parents * sizeof(Segment), hipMemcpyDeviceToHost, c->stream));
				HIPCHK(c, hipStreamSynchronize(c->stream));
				cur.assign((Segment *)c->pinned, (Segment *)c->pinned + hc.next_parents);
			}
			// atomic appends arrive in any order; make the plan deterministic
			std::sort(cur.begin(), cur.end(), [](const Segment &a, const Segment &b) { return a.start < b.start; });
		}
		phase_mark(c, "readback");
		++round;
	}
	if (unverified) { // no round's histogram pass carried the check: a pass of its own, before any leaf runs
		uint64_t vary = 0;
		int rc = run_vary(1, &vary);
		if (rc) return rc;
		phase_mark(c, "bit skip");
		if (vary & claimed_const) {
			if ((rc = start_over(vary))) return rc;
			again = true;
		} else
			unverified = false;
	}
	}

	// ---- leaves, stage 1: one unstable counting pass over all remaining bits (one workgroup per segment)
	constexpr size_t sort_lds = SortLds<K, V>::bytes;
	uint32_t nfallback_known = 0xFFFFFFFFu; // segments the counting leaves have handed to the general LDS sort, once the host has seen it
	if constexpr (!HV) {
		if (ncount_host && !single_pass) {
			// persistent workgroups (two per CU fit the LDS), segments handed out by ticket; what the fast
			// kernel cannot place directly is queued (in the round slab, dead by now) for the walking kernel
			int rcs = slab_reserve(c, 3 * (size_t)ncount_host * sizeof(Segment) + 4096);
			if (rcs) return rcs;
			Segment *slow = reinterpret_cast<Segment *>(c->slab), *rej16 = slow + ncount_host, *slow2 = rej16 + ncount_host;
			const uint32_t count_grid = std::min<uint32_t>(ncount_host, (uint32_t)c->sm_count * 2);
			if constexpr (sizeof(K) == 4) {
				// u32 keys with 16 open bits (what the planner aims for): the specialised kernel first, the general one
				// takes what that leaves (other bit counts, long or crowded segments)
				// (it pays for segments of about 2^14 keys -- 2^30-key inputs; on shorter ones the per-segment work on the
				// 2^16 counters dominates and 1024-thread workgroups hide its latency better: 1.15 vs 1.44 ms at 2^28)
				if (c->count16 == 2 || (c->count16 == 1 && n / ncount_host >= 12000)) {
					hipLaunchKernelGGL(count_place16_kernel, dim3(count_grid), dim3(kC16Th), kC16Lds, c->stream,
							   keys, small_count, ncount_host, rej16, ctr, n);
					hipLaunchKernelGGL((count_place_kernel<K>), dim3(count_grid), dim3(kCountTh), kCountLds, c->stream,
							   keys, rej16, 0u, (const uint32_t *)&ctr->nslow16, slow, ctr);
				} else
					hipLaunchKernelGGL((count_place_kernel<K>), dim3(count_grid), dim3(kCountTh), kCountLds, c->stream,
							   keys, small_count, ncount_host, (const uint32_t *)nullptr, slow, ctr);
			} else
				hipLaunchKernelGGL((count_place_kernel<K>), dim3(count_grid), dim3(kCountTh), kCountLds, c->stream,
						   keys, small_count, ncount_host, (const uint32_t *)nullptr, slow, ctr);
			const uint32_t *walk_n = &ctr->nslow;
			const Segment *walk_list = slow;
			if constexpr (sizeof(K) == 4) {
				// What the register-resident kernels left -- segments of 17 Ki .. 128 Ki keys (the mid-size buckets of skewed
				// inputs), crowded ones -- takes the 16-bit-counter leaf (msd_merge16.hpp, list mode: one workgroup per
				// segment, all of it counted before the first key is written back, so the sort is in place); only what that
				// does not take either (a key with >= 2^16 copies, a 256-value group with >= 2^16 keys) walks its counters.
				if (c->mid_leaf) {
					hipLaunchKernelGGL((merge_count_kernel<true>), dim3(std::min<uint32_t>(ncount_host, (uint32_t)c->sm_count)), dim3(kMcTh), kMcLds, c->stream,
							   (const uint32_t *)keys, (uint32_t *)keys, (const uint32_t *)nullptr, (const uint64_t *)nullptr, (const uint64_t *)nullptr,
							   0u, 0u, 0u, 0u, (const Segment *)slow, (const uint32_t *)&ctr->nslow, slow2, &ctr->nslow2, &ctr->count_ticket4,
							   (const uint32_t *)nullptr);
					walk_n = &ctr->nslow2;
					walk_list = slow2;
				}
			}
			hipLaunchKernelGGL((count_walk_kernel<K>), dim3(count_grid), dim3(kCountTh), kCountLds, c->stream,
					   keys, walk_list, walk_n, small, nsmall_host, (uint32_t)small_max, big, big_cap, ctr);
			HIPCHK(c, hipGetLastError());
			phase_mark(c, "count sort");
			// byte-counter overflows of segments above the LDS-sort capacity joined the big list
			Counters hc3;
			HIPCHK(c, hipMemcpyAsync(c->pinned, ctr, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
			HIPCHK(c, hipStreamSynchronize(c->stream));
			memcpy(&hc3, c->pinned, sizeof hc3);
			if (hc3.errors) return fail(c, MSD_EINTERNAL, "counting leaf: %u segments could not be queued", hc3.errors);
			nbig_host = hc3.nbig;
			nfallback_known = hc3.nfallback;
		}
	}
	// ---- keys-only segments of any size with <= 16 open bits: multi-workgroup counting sort
	if constexpr (!HV) {
		if (nbig_host && !single_pass) {
			int rc = pinned_reserve(c, (size_t)nbig_host * sizeof(Segment));
			if (rc) return rc;
			HIPCHK(c, hipMemcpyAsync(c->pinned, big, (size_t)nbig_host * sizeof(Segment), hipMemcpyDeviceToHost, c->stream));
			HIPCHK(c, hipStreamSynchronize(c->stream));
			std::vector<Segment> bs((Segment *)c->pinned, (Segment *)c->pinned + nbig_host);
			const uint32_t batch_max = 4096; // 256 KiB of histogram per segment: 1 GiB per batch
			for (uint32_t b0 = 0; b0 < nbig_host; b0 += batch_max) {
				const uint32_t nb = std::min(batch_max, nbig_host - b0);
				// work items -> segments: first[i] = index of segment i's first histogram chunk / group of output tiles
				constexpr uint32_t tile_elems = big_tile_elems<K>() * kBigGroup;
				std::vector<uint32_t> first(2 * ((size_t)nb + 1));
				uint32_t *first_chunk = first.data(), *first_tile = first.data() + nb + 1;
				uint64_t nchunks = 0, ntiles = 0;
				for (uint32_t i = 0; i < nb; ++i) {
					first_chunk[i] = (uint32_t)nchunks;
					first_tile[i] = (uint32_t)ntiles;
					nchunks += (bs[b0 + i].count + kBigChunk - 1) / kBigChunk;
					ntiles += (bs[b0 + i].count + tile_elems - 1) / tile_elems;
				}
				first_chunk[nb] = (uint32_t)nchunks;
				first_tile[nb] = (uint32_t)ntiles;
				uint32_t *ghist = nullptr, *d_first = nullptr;
				K *seg_hi = nullptr;
				uint16_t *tile_v = nullptr;
				Bump sz(nullptr), *bp = &sz;
				auto carve = [&]() {
					ghist = bp->take<uint32_t>((size_t)nb * 65536);
					seg_hi = bp->take<K>(nb);
					d_first = bp->take<uint32_t>(first.size());
					tile_v = bp->take<uint16_t>(ntiles * kBigGroup + nb + 8);
				};
				carve();
				rc = slab_reserve(c, sz.off + 4096); // the round slab is dead by now
				if (!rc) rc = pinned_reserve(c, first.size() * sizeof(uint32_t));
				if (rc) return rc;
				Bump bb(c->slab);
				bp = &bb;
				carve();
				HIPCHK(c, hipStreamSynchronize(c->stream));
				memcpy(c->pinned, first.data(), first.size() * sizeof(uint32_t));
				HIPCHK(c, hipMemcpyAsync(d_first, c->pinned, first.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
				HIPCHK(c, hipMemsetAsync(ghist, 0, (size_t)nb * 65536 * sizeof(uint32_t), c->stream));
				hipLaunchKernelGGL((bigcount_hist_kernel<K>), dim3((unsigned)std::min<uint64_t>(nchunks, (uint64_t)c->sm_count)), dim3(kBigHistTh), kBigHistLds, c->stream,
						   (const K *)keys, big + b0, (const uint32_t *)d_first, nb, ghist);
				hipLaunchKernelGGL((bigcount_scan_kernel<K>), dim3(nb), dim3(1024), 0, c->stream,
						   (const K *)keys, big + b0, (const uint32_t *)(d_first + nb + 1), ghist, seg_hi, tile_v, ctr);
				hipLaunchKernelGGL((bigcount_write_kernel<K>), dim3((unsigned)ntiles), dim3(kBigWriteTh), kBigWriteLds, c->stream,
						   keys, big + b0, (const uint32_t *)(d_first + nb + 1), nb, (const uint32_t *)ghist, (const K *)seg_hi,
						   (const uint16_t *)tile_v);
				HIPCHK(c, hipGetLastError());
			}
			phase_mark(c, "big count sort");
			Counters hc2;
			HIPCHK(c, hipMemcpyAsync(c->pinned, ctr, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
			HIPCHK(c, hipStreamSynchronize(c->stream));
			memcpy(&hc2, c->pinned, sizeof hc2);
			if (hc2.errors) return fail(c, MSD_EINTERNAL, "counting sort: %u histogram totals disagree with segment sizes", hc2.errors);
		}
	}
	set_stat(c, "big_count_segments", nbig_host);

	// ---- leaves, stage 2: everything else that fits LDS (payloads, > 16 open bits): counting leaf on the
	// top varying bits; its rare failures (long groups of equal top bits) and the byte-counter overflows
	// of stage 1 go to the general LDS sort (their number is only known on the device)
	if (!single_pass) {
		const uint32_t nsm = nsmall_host + (HV ? ncount_host : 0u); // tuples: both lists hold leaf_count work
		// persistent workgroups with prefetch of the next segment: as many per CU as the LDS holds
		constexpr size_t leaf_lds = LeafCountLds<K, V>::bytes;
		const uint32_t leaf_per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(2048 / C::SORT_TH, (160 * 1024) / (leaf_lds + 512)));
		bool small_done = false;
		if constexpr (!HV && sizeof(K) == 8) {
			// u64 keys, segments of about 2^14 (what two 8-bit rounds leave of 2^30 keys): leaf17_kernel -- counters for up to
			// 16 bits where the staging buffer will be, one lane per group of equal counted bits -- finishes a segment in 0.6
			// of this leaf's time; what it leaves (segments shorter than 4096 keys, a group of more than 48) goes on to it.
			if (c->leaf17 && nsmall_host && n / nsmall_host >= 8192) {
				int rc = slab_reserve(c, (size_t)nsmall_host * sizeof(Segment) + 4096);
				if (rc) return rc;
				Segment *d_rej = reinterpret_cast<Segment *>(c->slab);
				HIPCHK(c, hipMemsetAsync(&ctr->nslow2, 0, sizeof(uint32_t), c->stream));
				HIPCHK(c, hipMemsetAsync(&ctr->l17_slow, 0, sizeof(uint32_t), c->stream));
				hipLaunchKernelGGL((leaf17_kernel<NoVal>), dim3(std::min<uint32_t>(nsmall_host, (uint32_t)c->sm_count)), dim3(kL17Th), kL17Lds, c->stream,
						   (uint64_t *)keys, (uint64_t *)nullptr, (const Segment *)small, nsmall_host, d_rej, &ctr->nslow2, ctr, 4096u);
				hipLaunchKernelGGL((leaf_count_sort_kernel<K, V>), dim3(std::min<uint32_t>(nsmall_host, (uint32_t)c->sm_count * leaf_per_cu)), dim3(C::SORT_TH), leaf_lds, c->stream,
						   keys, vals, (const Segment *)d_rej, nsmall_host, small + nsmall_host, ctr, &ctr->leaf_ticket[0], (const uint32_t *)&ctr->nslow2);
				HIPCHK(c, hipGetLastError());
				add_stat(c, "leaf17_launches", 1);
				small_done = true;
			}
		}
		if (nsmall_host && !small_done) {
			hipLaunchKernelGGL((leaf_count_sort_kernel<K, V>), dim3(std::min<uint32_t>(nsmall_host, (uint32_t)c->sm_count * leaf_per_cu)), dim3(C::SORT_TH), leaf_lds, c->stream,
					   keys, vals, small, nsmall_host, small + nsmall_host, ctr, &ctr->leaf_ticket[0]);
			HIPCHK(c, hipGetLastError());
		}
		if (HV && ncount_host) { // (tuples whose last <= 16 bits are open -- 5b after its rounds: same kernel, its own ticket)
			hipLaunchKernelGGL((leaf_count_sort_kernel<K, V>), dim3(std::min<uint32_t>(ncount_host, (uint32_t)c->sm_count * leaf_per_cu)), dim3(C::SORT_TH), leaf_lds, c->stream,
					   keys, vals, small_count, ncount_host, small + nsmall_host, ctr, &ctr->leaf_ticket[1]);
			HIPCHK(c, hipGetLastError());
		}
		(void)nsm;
		// (keys only, no small segments, and the counting leaves are known to have handed nothing on: no launch -- 2^30 uniform
		// u32 keys: 0.03 ms for workgroups that look at an empty list)
		const bool may_fall_back = HV || nsmall_host != 0 || nfallback_known != 0;
		if (may_fall_back && nsmall_host + ncount_host) {
			hipLaunchKernelGGL((lds_sort_kernel<K, V>), dim3(std::min<uint32_t>(nsmall_host + ncount_host, 2 * c->sm_count)), dim3(C::SORT_TH), sort_lds, c->stream,
					   keys, vals, small + nsmall_host, 0u, (const uint32_t *)&ctr->nfallback);
			HIPCHK(c, hipGetLastError());
		}
		phase_mark(c, "LDS sort");
	}
	set_stat(c, "count_segments", ncount_host);
	set_stat(c, "small_segments", nsmall_host);
	set_stat(c, "workspace_bytes", c->slab_bytes + c->keep_bytes + 4 * c->lists_cap * sizeof(Segment));
	phase_end(c);
	return MSD_OK;
}

template <typename K, typename V> static int set_lds_attrs(msd_ctx *c)
{
	{ // the block permutation is launched with exactly the workgroups the chip holds at once (see chains_grid)
		int per_cu = 0;
		HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chains_kernel<K, V>, 256, 0));
		c->chains_per_cu[has_val<V>::value ? 2 : sizeof(K) == 8 ? 1 : 0] = std::max(1, per_cu);
	}
	HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&classify_kernel<K, V, false>),
				      hipFuncAttributeMaxDynamicSharedMemorySize, (int)ClassifyLds<K, V>::bytes));
	if constexpr (kHasRange<K, V>)
		HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&classify_kernel<K, V, true>),
					      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ClassifyLds<K, V>::bytes + kP * sizeof(K))));
	HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&classify_stream2_kernel<K, V, false>),
				      hipFuncAttributeMaxDynamicSharedMemorySize, (int)Stream2Lds<K, V>::bytes));
	HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&classify_stream2_kernel<K, V, true>),
				      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(Stream2Lds<K, V>::bytes + kP * sizeof(K))));
	HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&lds_sort_kernel<K, V>),
				      hipFuncAttributeMaxDynamicSharedMemorySize, (int)SortLds<K, V>::bytes));
	HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&classify_direct2_kernel<K, V>),
				      hipFuncAttributeMaxDynamicSharedMemorySize, (int)Direct2Lds<K, V>::bytes));
	if constexpr (sizeof(K) == 8) {
		HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&regpart_kernel<V>),
					      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRpLds));
		HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&leaf17_kernel<V>),
					      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kL17Lds));
	}
	HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&leaf_count_sort_kernel<K, V>),
				      hipFuncAttributeMaxDynamicSharedMemorySize, (int)LeafCountLds<K, V>::bytes));
	if constexpr (!has_val<V>::value) {
		if constexpr (sizeof(K) == 4) {
			HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&count_place16_kernel),
						      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kC16Lds));
			HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&merge_place16_kernel<2>),
						      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kC16Lds));
			HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&merge_place16_kernel<4>),
						      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kC16Lds));
			HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&merge_place16_kernel<8>),
						      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kC16Lds));
			HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&merge_count_kernel<false, uint32_t>),
						      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMcLds));
			HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&merge_count_kernel<false, uint16_t>),
						      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMcLds));
			HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&merge_count_kernel<false, Hist2>),
						      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMcLds));
			HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&hist2_pack_kernel<uint32_t>),
						      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kH2Lds));
			HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&hist2_pack_kernel<uint16_t>),
						      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kH2Lds));
			HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&scatter_low16_kernel),
						      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kS16Lds));
			HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&merge_count_kernel<true, uint32_t>),
						      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMcLds));
		}
		HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&count_place_kernel<K>),
					      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCountLds));
		HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&count_walk_kernel<K>),
					      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCountLds));
		HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&bigcount_hist_kernel<K>),
					      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBigHistLds));
		HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&bigcount_write_kernel<K>),
					      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBigWriteLds));
	}
	return MSD_OK;
}

template <typename K>
static int gather_impl(msd_ctx *c, K *dst, const K *src, const uint64_t *src_off, const uint64_t *dst_off, const uint64_t *len, uint32_t nruns)
{
	if (!c) return MSD_EINVAL;
	if (nruns == 0) return MSD_OK;
	if (!dst || !src || !src_off || !dst_off || !len) return fail(c, MSD_EINVAL, "gather: null pointer");
	HIPCHK(c, hipSetDevice(c->device));
	std::vector<GatherRun> runs;
	uint64_t nchunks = 0;
	for (uint32_t i = 0; i < nruns; ++i) {
		if (!len[i]) continue;
		runs.push_back({ src_off[i], dst_off[i], len[i], (uint32_t)nchunks, 0 });
		nchunks += (len[i] + kGatherChunk - 1) / kGatherChunk;
	}
	if (runs.empty()) return MSD_OK;
	if (nchunks >= 0xFFFFFFFFull) return fail(c, MSD_EINVAL, "gather: too many elements");
	runs.push_back({ 0, 0, 0, (uint32_t)nchunks, 0 }); // sentinel
	const size_t nreal = runs.size() - 1, ncoarse = (size_t)(nchunks >> 6) + 2;
	std::vector<uint32_t> coarse(ncoarse); // the run of every 64th chunk
	for (size_t i = 0, r = 0; i < ncoarse; ++i) {
		const uint64_t ch = std::min<uint64_t>((uint64_t)i << 6, nchunks - 1);
		while (r + 1 < nreal && runs[r + 1].first_chunk <= ch) ++r;
		coarse[i] = (uint32_t)r;
	}
	const size_t runs_bytes = align_up(runs.size() * sizeof(GatherRun), 256), bytes = runs_bytes + ncoarse * sizeof(uint32_t);
	int rc = pinned_reserve(c, bytes);
	if (!rc) rc = slab_reserve(c, bytes + 4096); // (between sorts nothing in the slab is live)
	if (rc) return rc;
	HIPCHK(c, hipStreamSynchronize(c->stream)); // the staging buffer may still be in flight
	memcpy(c->pinned, runs.data(), runs.size() * sizeof(GatherRun));
	memcpy((char *)c->pinned + runs_bytes, coarse.data(), ncoarse * sizeof(uint32_t));
	GatherRun *d_runs = reinterpret_cast<GatherRun *>(c->slab);
	const uint32_t *d_coarse = reinterpret_cast<const uint32_t *>(c->slab + runs_bytes);
	HIPCHK(c, hipMemcpyAsync(d_runs, c->pinned, bytes, hipMemcpyHostToDevice, c->stream));
	// (one workgroup per 8 KiB chunk, no loop: 4.3-4.5 TB/s; persistent workgroups with larger chunks: 3.6)
	const unsigned grid = (unsigned)nchunks;
	hipLaunchKernelGGL((gather_runs_kernel<K>), dim3(grid), dim3(256), 0, c->stream, dst, src, (const GatherRun *)d_runs,
			   (uint32_t)runs.size() - 1, (uint32_t)nchunks, d_coarse);
	HIPCHK(c, hipGetLastError());
	return MSD_OK;
}

// ------------------------------------------------------------------ C ABI

extern "C" {

const char *msd_version(void) { return MSD_VERSION; }

int msd_create(msd_ctx **out, int device, void *stream)
{
	if (!out) return MSD_EINVAL;
	*out = nullptr;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return MSD_EHIP; // no CPU fallback exists
	if (device < 0 || device >= ndev) return MSD_EINVAL;
	msd_ctx *c = new msd_ctx();
	c->device = device;
	c->stream = (hipStream_t)stream;
	if (hipSetDevice(device) != hipSuccess) {
		delete c;
		return MSD_EHIP;
	}
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->sm_count = prop.multiProcessorCount;
	// A/B switches for benchmarks: the same knobs as msd_set_option, through the same range checks (an out-of-range
	// value is reported and ignored)
	for (const char *name : { "direct_mode", "regpart", "count16" }) {
		const char *env = !strcmp(name, "direct_mode") ? "MSD_DIRECT" : !strcmp(name, "regpart") ? "MSD_REGPART" : "MSD_COUNT16";
		const char *v = getenv(env);
		if (!v) continue;
		char *end = nullptr;
		const long x = strtol(v, &end, 10);
		if (end == v || *end || msd_set_option(c, name, x) != MSD_OK) fprintf(stderr, "msd_create: ignoring %s=%s\n", env, v);
	}
	int rc = set_lds_attrs<uint32_t, NoVal>(c);
	if (!rc) rc = set_lds_attrs<uint64_t, NoVal>(c);
	if (!rc) rc = set_lds_attrs<uint64_t, uint64_t>(c);
	if (rc) {
		fprintf(stderr, "msd_create: %s\n", c->err.c_str());
		delete c;
		return rc;
	}
	*out = c;
	return MSD_OK;
}

int msd_destroy(msd_ctx *c)
{
	if (!c) return MSD_EINVAL;
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize(c->stream);
	if (c->slab) (void)hipFree(c->slab);
	if (c->keep) (void)hipFree(c->keep);
	if (c->lists) (void)hipFree(c->lists);
	if (c->pinned) (void)hipHostFree(c->pinned);
	if (c->ev_start) (void)hipEventDestroy(c->ev_start);
	for (auto e : c->ev_pool) (void)hipEventDestroy(e);
	delete c;
	return MSD_OK;
}

int msd_set_stream(msd_ctx *c, void *stream)
{
	if (!c) return MSD_EINVAL;
	c->stream = (hipStream_t)stream;
	return MSD_OK;
}

void *msd_get_stream(const msd_ctx *c) { return c ? (void *)c->stream : nullptr; }
int msd_get_device(const msd_ctx *c) { return c ? c->device : -1; }

int msd_reserve(msd_ctx *c, uint64_t n, int key_bytes, int val_bytes)
{
	if (!c) return MSD_EINVAL;
	HIPCHK(c, hipSetDevice(c->device));
	size_t round_b, keep_b, list_n;
	if (key_bytes == 4 && val_bytes == 0) {
		round_b = round_bytes_estimate<uint32_t, NoVal>(n, c->sm_count);
		keep_b = keep_bytes_for<uint32_t, NoVal>(n);
		list_n = leaf_list_guess<uint32_t, NoVal>(n);
	} else if (key_bytes == 8 && val_bytes == 0) {
		round_b = round_bytes_estimate<uint64_t, NoVal>(n, c->sm_count);
		keep_b = keep_bytes_for<uint64_t, NoVal>(n);
		list_n = leaf_list_guess<uint64_t, NoVal>(n);
	} else if (key_bytes == 8 && val_bytes == 8) {
		round_b = round_bytes_estimate<uint64_t, uint64_t>(n, c->sm_count);
		keep_b = keep_bytes_for<uint64_t, uint64_t>(n);
		list_n = leaf_list_guess<uint64_t, uint64_t>(n);
	} else
		return fail(c, MSD_EINVAL, "unsupported element layout %d+%d bytes", key_bytes, val_bytes);
	int rc = slab_reserve(c, round_b, true);
	if (!rc) rc = keep_reserve(c, keep_b, true);
	if (!rc) rc = pinned_reserve(c, 1 << 20);
	if (!rc) rc = lists_reserve(c, list_n, 0, 0, true);
	return rc;
}

uint64_t msd_workspace_bytes(const msd_ctx *c) { return c ? c->slab_bytes + c->keep_bytes + 4 * c->lists_cap * sizeof(Segment) : 0; }
const char *msd_last_error(const msd_ctx *c) { return c ? c->err.c_str() : "null context"; }

int msd_sort_u32_bits(msd_ctx *c, uint32_t *k, uint64_t n, int end_bit)
{
	if (!c) return MSD_EINVAL;
	return sort_impl<uint32_t, NoVal>(c, k, nullptr, n, end_bit, false, 0, 0, nullptr);
}
int msd_sort_u64_bits(msd_ctx *c, uint64_t *k, uint64_t n, int end_bit)
{
	if (!c) return MSD_EINVAL;
	return sort_impl<uint64_t, NoVal>(c, k, nullptr, n, end_bit, false, 0, 0, nullptr);
}
int msd_sort_pairs_u64_bits(msd_ctx *c, uint64_t *k, uint64_t *r, uint64_t n, int end_bit)
{
	if (!c) return MSD_EINVAL;
	return sort_impl<uint64_t, uint64_t>(c, k, r, n, end_bit, false, 0, 0, nullptr);
}
int msd_sort_u32(msd_ctx *c, uint32_t *k, uint64_t n) { return msd_sort_u32_bits(c, k, n, 32); }
int msd_sort_u64(msd_ctx *c, uint64_t *k, uint64_t n) { return msd_sort_u64_bits(c, k, n, 64); }
int msd_sort_pairs_u64(msd_ctx *c, uint64_t *k, uint64_t *r, uint64_t n) { return msd_sort_pairs_u64_bits(c, k, r, n, 64); }

int msd_partition_u32(msd_ctx *c, uint32_t *k, uint64_t n, unsigned shift, unsigned rb, uint64_t *cnt)
{
	if (!c) return MSD_EINVAL;
	HIPCHK(c, hipSetDevice(c->device));
	if (cnt && rb <= 8) HIPCHK(c, hipMemsetAsync(cnt, 0, sizeof(uint64_t) << rb, c->stream));
	return sort_impl<uint32_t, NoVal>(c, k, nullptr, n, 32, true, shift, rb, cnt);
}
int msd_partition_u64(msd_ctx *c, uint64_t *k, uint64_t n, unsigned shift, unsigned rb, uint64_t *cnt)
{
	if (!c) return MSD_EINVAL;
	HIPCHK(c, hipSetDevice(c->device));
	if (cnt && rb <= 8) HIPCHK(c, hipMemsetAsync(cnt, 0, sizeof(uint64_t) << rb, c->stream));
	return sort_impl<uint64_t, NoVal>(c, k, nullptr, n, 64, true, shift, rb, cnt);
}
int msd_partition_pairs_u64(msd_ctx *c, uint64_t *k, uint64_t *r, uint64_t n, unsigned shift, unsigned rb, uint64_t *cnt)
{
	if (!c) return MSD_EINVAL;
	HIPCHK(c, hipSetDevice(c->device));
	if (cnt && rb <= 8) HIPCHK(c, hipMemsetAsync(cnt, 0, sizeof(uint64_t) << rb, c->stream));
	return sort_impl<uint64_t, uint64_t>(c, k, r, n, 64, true, shift, rb, cnt);
}

// ---- a sort that stops early: afterwards the keys are ordered by key >> begin_bit (the top-digit passes of a rank of the
// multi-GPU sort before its exchange; keys that agree above begin_bit may be in any order)
int msd_sort_u32_top(msd_ctx *c, uint32_t *k, uint64_t n, int end_bit, int begin_bit)
{
	if (!c) return MSD_EINVAL;
	if (begin_bit < 0 || begin_bit > end_bit) return fail(c, MSD_EINVAL, "begin_bit must lie in [0, end_bit]");
	return sort_impl<uint32_t, NoVal>(c, k, nullptr, n, end_bit, false, 0, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, (uint32_t)begin_bit);
}
int msd_sort_u64_top(msd_ctx *c, uint64_t *k, uint64_t n, int end_bit, int begin_bit)
{
	if (!c) return MSD_EINVAL;
	if (begin_bit < 0 || begin_bit > end_bit) return fail(c, MSD_EINVAL, "begin_bit must lie in [0, end_bit]");
	return sort_impl<uint64_t, NoVal>(c, k, nullptr, n, end_bit, false, 0, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, (uint32_t)begin_bit);
}
int msd_sort_pairs_u64_top(msd_ctx *c, uint64_t *k, uint64_t *r, uint64_t n, int end_bit, int begin_bit)
{
	if (!c) return MSD_EINVAL;
	if (begin_bit < 0 || begin_bit > end_bit) return fail(c, MSD_EINVAL, "begin_bit must lie in [0, end_bit]");
	return sort_impl<uint64_t, uint64_t>(c, k, r, n, end_bit, false, 0, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, (uint32_t)begin_bit);
}

} // extern "C"

template <typename K>
static int bounds_impl(msd_ctx *c, const K *k, uint64_t n, unsigned shift, uint64_t first, uint32_t nbuckets, uint64_t *bounds)
{
	if (!c) return MSD_EINVAL;
	if (!bounds || (n && !k)) return fail(c, MSD_EINVAL, "bucket_bounds: null pointer");
	if (shift >= sizeof(K) * 8 || nbuckets == 0 || nbuckets > (1u << 24)) return fail(c, MSD_EINVAL, "bucket_bounds: shift or bucket count out of range");
	HIPCHK(c, hipSetDevice(c->device));
	hipLaunchKernelGGL((bucket_bounds_kernel<K>), dim3((nbuckets + 1 + 255) / 256), dim3(256), 0, c->stream, k, n, (uint32_t)shift, first, nbuckets, bounds);
	HIPCHK(c, hipGetLastError());
	return MSD_OK;
}

// The counting leaf of a rank after a fine-grained exchange (msd_merge16.hpp): every bucket = nsrc extents in d_src.
template <typename IN>
static int merge_impl(msd_ctx *c, const IN *src, uint64_t src_cap, const uint64_t *d_counts, const uint64_t *src_base, uint32_t nsrc,
		      uint32_t nb, int open_bits, uint32_t first_prefix, uint32_t *dst, uint64_t dst_cap, uint64_t n_expected)
{
	constexpr bool IN16 = sizeof(IN) == 2, HIST = sizeof(IN) == 1; // (HIST: src = records of hist2_pack_kernel, src_cap in bytes)
	if (!c) return MSD_EINVAL;
	if (!src || !dst || !d_counts || !src_base) return fail(c, MSD_EINVAL, "merge_buckets: null pointer");
	if (nsrc < 1 || nsrc > 8) return fail(c, MSD_EINVAL, "merge_buckets: 1..8 source runs per bucket");
	if (nb == 0 || nb > (1u << 24)) return fail(c, MSD_EINVAL, "merge_buckets: bucket count out of range");
	if (open_bits < 1 || open_bits > 16) return fail(c, MSD_EINVAL, "merge_buckets: 1..16 open bits");
	if ((IN16 || HIST) && open_bits != 16) return fail(c, MSD_EINVAL, "merge_buckets: extents of low halves need 16 open bits");
	if (HIST && src_cap < (uint64_t)nsrc * nb * kH2Rec) return fail(c, MSD_EINVAL, "merge_buckets: %u x %u records of %u bytes do not fit the source buffer", nsrc, nb, kH2Rec);
	if (((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return fail(c, MSD_EINVAL, "merge_buckets: buffers must be 16-byte aligned");
	if (n_expected > dst_cap) return fail(c, MSD_EINVAL, "merge_buckets: the output buffer is too small");
	if ((uint64_t)first_prefix + nb > (1ull << (32 - open_bits))) return fail(c, MSD_EINVAL, "merge_buckets: bucket numbers exceed the key's prefix");
	{ // the buffers must not overlap (the leaf reads extents while other workgroups write finished buckets)
		const uintptr_t s0 = (uintptr_t)src, s1 = s0 + src_cap * sizeof(IN), d0 = (uintptr_t)dst, d1 = d0 + dst_cap * 4;
		if (s0 < d1 && d0 < s1) return fail(c, MSD_EINVAL, "merge_buckets: source and destination overlap");
	}
	HIPCHK(c, hipSetDevice(c->device));
	c->stats.clear();
	phase_begin(c);
	if (n_expected == 0) return MSD_OK;
	Bump sz(nullptr), *bp = &sz;
	Counters *ctr = nullptr;
	uint32_t *status = nullptr, *cnt32 = nullptr;
	uint64_t *soff = nullptr, *doff = nullptr;
	Segment *rej = nullptr;
	auto carve = [&]() {
		ctr = bp->take<Counters>(1);
		status = bp->take<uint32_t>(64);
		cnt32 = bp->take<uint32_t>((size_t)nsrc * nb);
		soff = bp->take<uint64_t>((size_t)nsrc * nb);
		doff = bp->take<uint64_t>((size_t)nb + 1);
		rej = bp->take<Segment>(nb);
	};
	carve();
	int rc = slab_reserve(c, sz.off + 4096);
	if (!rc) rc = pinned_reserve(c, 4096);
	if (rc) return rc;
	Bump real(c->slab);
	bp = &real;
	carve();
	HIPCHK(c, hipMemsetAsync(ctr, 0, (char *)(status + 64) - (char *)ctr, c->stream));
	MergeBase mb = {};
	for (uint32_t x = 0; x < nsrc; ++x) mb.b[x] = src_base[x];
	hipLaunchKernelGGL(merge_plan_kernel, dim3(nsrc + 1), dim3(1024), 0, c->stream, d_counts, mb, nsrc, nb, n_expected, cnt32, soff, doff, status);
	// buckets that fit the registers of a workgroup (shards of <= 2^27 keys at 8 ranks) take merge_place16_kernel, larger
	// ones (2^30 keys per rank: nsrc x 2^14 keys per bucket) merge_count_kernel
	// (low halves: merge_count_kernel at every bucket size -- shorter buckets cost it more per key, the exchange it follows
	// cost half)
	const bool in_regs = !IN16 && !HIST && (c->merge_leaf == 1 || (c->merge_leaf == 0 && n_expected / nb <= 12000 && open_bits >= (int)kC16MinBits));
	if constexpr (IN16 || HIST) {
		const unsigned grid = (unsigned)std::min<uint64_t>(nb, (uint64_t)c->sm_count);
		hipLaunchKernelGGL((merge_count_kernel<false, IN>), dim3(grid), dim3(kMcTh), kMcLds, c->stream, src, dst, (const uint32_t *)cnt32,
				   (const uint64_t *)soff, (const uint64_t *)doff, nsrc, nb, (uint32_t)open_bits, first_prefix, (const Segment *)nullptr,
				   (const uint32_t *)nullptr, rej, &ctr->nslow16, &ctr->count_ticket3, (const uint32_t *)status);
	} else if (in_regs) {
		const unsigned grid = (unsigned)std::min<uint64_t>(nb, (uint64_t)c->sm_count * 2);
		const uint32_t G = nsrc <= 2 ? 2 : nsrc <= 4 ? 4 : 8;
#define MSD_MERGE_LAUNCH(GG)                                                                                                              \
	hipLaunchKernelGGL((merge_place16_kernel<GG>), dim3(grid), dim3(kC16Th), kC16Lds, c->stream, (const uint32_t *)src, src_cap, dst, (const uint32_t *)cnt32, \
			   (const uint64_t *)soff, (const uint64_t *)doff, nsrc, nb, (uint32_t)open_bits, first_prefix, rej, ctr, (const uint32_t *)status)
		if (G == 2) MSD_MERGE_LAUNCH(2); else if (G == 4) MSD_MERGE_LAUNCH(4); else MSD_MERGE_LAUNCH(8);
#undef MSD_MERGE_LAUNCH
	} else {
		const unsigned grid = (unsigned)std::min<uint64_t>(nb, (uint64_t)c->sm_count);
		hipLaunchKernelGGL((merge_count_kernel<false, uint32_t>), dim3(grid), dim3(kMcTh), kMcLds, c->stream, (const uint32_t *)src, dst, (const uint32_t *)cnt32,
				   (const uint64_t *)soff, (const uint64_t *)doff, nsrc, nb, (uint32_t)open_bits, first_prefix, (const Segment *)nullptr,
				   (const uint32_t *)nullptr, rej, &ctr->nslow16, &ctr->count_ticket3, (const uint32_t *)status);
	}
	HIPCHK(c, hipGetLastError());
	phase_mark(c, "merge leaf");
	// what the leaf did not take (rare: long or crowded buckets) lies unsorted at its place in dst: the general leaves finish it
	HIPCHK(c, hipMemcpyAsync(c->pinned, ctr, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipMemcpyAsync((char *)c->pinned + 1024, status, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	Counters hc;
	memcpy(&hc, c->pinned, sizeof hc);
	if (*(const uint32_t *)((char *)c->pinned + 1024)) return fail(c, MSD_EINVAL, "merge_buckets: the counts do not add up to the expected %llu keys (or a count exceeds 32 bits)", (unsigned long long)n_expected);
	const uint32_t nrej = hc.nslow16;
	phase_end(c);
	if (nrej) {
		rc = pinned_reserve(c, (size_t)nrej * sizeof(Segment));
		if (rc) return rc;
		HIPCHK(c, hipMemcpyAsync(c->pinned, rej, (size_t)nrej * sizeof(Segment), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipStreamSynchronize(c->stream));
		std::vector<Segment> segs((Segment *)c->pinned, (Segment *)c->pinned + nrej);
		rc = sort_impl<uint32_t, NoVal>(c, dst, nullptr, n_expected, 32, false, 0, 0, nullptr, nullptr, 0, nullptr, 0, &segs);
		if (rc) return rc;
	}
	set_stat(c, "merge_rejected", nrej);
	return MSD_OK;
}

extern "C" {

int msd_bucket_bounds_u32(msd_ctx *c, const uint32_t *k, uint64_t n, unsigned shift, uint64_t first, uint32_t nbuckets, uint64_t *bounds)
{
	return bounds_impl<uint32_t>(c, k, n, shift, first, nbuckets, bounds);
}
int msd_bucket_bounds_u64(msd_ctx *c, const uint64_t *k, uint64_t n, unsigned shift, uint64_t first, uint32_t nbuckets, uint64_t *bounds)
{
	return bounds_impl<uint64_t>(c, k, n, shift, first, nbuckets, bounds);
}
int msd_merge_buckets_u32(msd_ctx *c, const uint32_t *d_src, uint64_t src_cap, const uint64_t *d_counts, const uint64_t *src_base, uint32_t nsrc,
			  uint32_t nbuckets, int open_bits, uint32_t first_prefix, uint32_t *d_dst, uint64_t dst_cap, uint64_t n_expected)
{
	return merge_impl<uint32_t>(c, d_src, src_cap, d_counts, src_base, nsrc, nbuckets, open_bits, first_prefix, d_dst, dst_cap, n_expected);
}
int msd_merge_buckets_u32_low16(msd_ctx *c, const uint16_t *d_src, uint64_t src_cap, const uint64_t *d_counts, const uint64_t *src_base, uint32_t nsrc,
				uint32_t nbuckets, uint32_t first_prefix, uint32_t *d_dst, uint64_t dst_cap, uint64_t n_expected)
{
	return merge_impl<uint16_t>(c, d_src, src_cap, d_counts, src_base, nsrc, nbuckets, 16, first_prefix, d_dst, dst_cap, n_expected);
}
int msd_merge_buckets_u32_hist2(msd_ctx *c, const void *d_rec, uint64_t rec_bytes, const uint64_t *d_counts, uint32_t nsrc, uint32_t nbuckets,
				uint32_t first_prefix, uint32_t *d_dst, uint64_t dst_cap, uint64_t n_expected)
{
	const uint64_t zero[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; // (records lie at fixed places: source x's at x * nbuckets * record size)
	return merge_impl<Hist2>(c, (const Hist2 *)d_rec, rec_bytes, d_counts, zero, nsrc, nbuckets, 16, first_prefix, d_dst, dst_cap, n_expected);
}
uint64_t msd_hist2_record_bytes(void) { return kH2Rec; }
} // extern "C"
template <typename IN>
static int hist2_pack_impl(msd_ctx *c, const IN *d_keys, uint64_t n, const uint64_t *d_bounds, uint32_t nbuckets, void *d_rec, uint64_t rec_bytes,
			   uint32_t *d_overflow)
{
	if (!c) return MSD_EINVAL;
	if (!d_keys || !d_bounds || !d_rec || !d_overflow) return fail(c, MSD_EINVAL, "hist2_pack: null pointer");
	if (nbuckets == 0 || nbuckets > 65536) return fail(c, MSD_EINVAL, "hist2_pack: 1..65536 buckets");
	if (((uintptr_t)d_keys & 15) || ((uintptr_t)d_rec & 15)) return fail(c, MSD_EINVAL, "hist2_pack: buffers must be 16-byte aligned");
	if (rec_bytes < (uint64_t)nbuckets * kH2Rec) return fail(c, MSD_EINVAL, "hist2_pack: %u records of %u bytes do not fit the output buffer", nbuckets, kH2Rec);
	{
		const uintptr_t s0 = (uintptr_t)d_keys, s1 = s0 + n * sizeof(IN), d0 = (uintptr_t)d_rec, d1 = d0 + (uint64_t)nbuckets * kH2Rec;
		if (s0 < d1 && d0 < s1) return fail(c, MSD_EINVAL, "hist2_pack: source and destination overlap");
	}
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipMemsetAsync(d_overflow, 0, sizeof(uint32_t), c->stream));
	const unsigned grid = (unsigned)std::min<uint64_t>(nbuckets, (uint64_t)c->sm_count * 2);
	hipLaunchKernelGGL((hist2_pack_kernel<IN>), dim3(grid), dim3(kH2Th), kH2Lds, c->stream, d_keys, d_bounds, nbuckets, (unsigned char *)d_rec, d_overflow);
	HIPCHK(c, hipGetLastError());
	return MSD_OK;
}
// starts[b] = sum of counts[0 .. b), b = 0 .. 65536 (one workgroup)
__global__ __launch_bounds__(1024) void bounds16_kernel(const uint64_t *__restrict__ counts, uint64_t *__restrict__ bounds)
{
	__shared__ uint64_t wsum[16];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
	uint64_t mine = 0;
	for (uint32_t j = 0; j < 64; ++j) mine += counts[tid * 64u + j];
	const uint64_t inc = wave_incl_scan64(mine);
	if (lane == 63) wsum[w] = inc;
	__syncthreads();
	uint64_t at = inc - mine;
	for (uint32_t ww = 0; ww < w; ++ww) at += wsum[ww];
	for (uint32_t j = 0; j < 64; ++j) {
		bounds[tid * 64u + j] = at;
		at += counts[tid * 64u + j];
	}
	if (tid == 1023) bounds[65536] = at;
}
extern "C" {
int msd_hist2_pack_u32(msd_ctx *c, const uint32_t *d_keys, uint64_t n, const uint64_t *d_bounds, uint32_t nbuckets, void *d_rec, uint64_t rec_bytes,
		       uint32_t *d_overflow)
{
	return hist2_pack_impl<uint32_t>(c, d_keys, n, d_bounds, nbuckets, d_rec, rec_bytes, d_overflow);
}
int msd_hist2_pack_u32_low16(msd_ctx *c, const uint16_t *d_low, uint64_t n, const uint64_t *d_bounds, uint32_t nbuckets, void *d_rec, uint64_t rec_bytes,
			     uint32_t *d_overflow)
{
	return hist2_pack_impl<uint16_t>(c, d_low, n, d_bounds, nbuckets, d_rec, rec_bytes, d_overflow);
}
int msd_bounds_from_counts16(msd_ctx *c, const uint64_t *d_counts, uint64_t *d_bounds)
{
	if (!c) return MSD_EINVAL;
	if (!d_counts || !d_bounds) return fail(c, MSD_EINVAL, "bounds_from_counts16: null pointer");
	HIPCHK(c, hipSetDevice(c->device));
	hipLaunchKernelGGL(bounds16_kernel, dim3(1), dim3(1024), 0, c->stream, d_counts, d_bounds);
	HIPCHK(c, hipGetLastError());
	return MSD_OK;
}
// msd_order_low16_u32 in two halves: the counts are ready (asynchronously) after the first, so that the caller can start its
// count exchange with the other ranks while the second -- the scatter, 2 ms per 2^30 keys -- runs
int msd_order_low16_counts_u32(msd_ctx *c, uint32_t *d_keys, uint64_t n, uint64_t *d_counts)
{
	if (!c) return MSD_EINVAL;
	c->order_keys = nullptr;
	if (!d_counts || (n && !d_keys)) return fail(c, MSD_EINVAL, "order_low16: null pointer");
	if ((uintptr_t)d_keys & 15) return fail(c, MSD_EINVAL, "order_low16: buffers must be 16-byte aligned");
	if (n >= (1ull << 40)) return fail(c, MSD_EINVAL, "order_low16: too many keys");
	// one in-place round on the top 8 bits (the direct-placement round 0) ...
	int rc = sort_impl<uint32_t, NoVal>(c, d_keys, nullptr, n, 32, false, 0, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, 24u);
	if (rc) return rc;
	// ... exact counts of all upper halves and the place of every workgroup's share of every bucket
	HIPCHK(c, hipSetDevice(c->device));
	Bump sz(nullptr), *bp = &sz;
	uint64_t *pb = nullptr;
	uint32_t *wg = nullptr;
	unsigned long long *base = nullptr;
	auto carve = [&]() {
		pb = bp->take<uint64_t>(257);
		wg = bp->take<uint32_t>((size_t)65536 * kS16Chunks);
		base = bp->take<unsigned long long>((size_t)65536 * kS16Chunks + 1);
	};
	carve();
	rc = slab_reserve(c, sz.off + 4096);
	if (rc) return rc;
	Bump real(c->slab);
	bp = &real;
	carve();
	hipLaunchKernelGGL((bucket_bounds_kernel<uint32_t>), dim3(2), dim3(256), 0, c->stream, (const uint32_t *)d_keys, n, 24u, (uint64_t)0, 256u, pb);
	hipLaunchKernelGGL(hist16_kernel, dim3(256 * kS16Chunks), dim3(kS16Th), 0, c->stream, (const uint32_t *)d_keys, n, (const uint64_t *)pb, wg);
	hipLaunchKernelGGL(scan16_kernel, dim3(256), dim3(256), 0, c->stream, (const uint32_t *)wg, (const uint64_t *)pb, (unsigned long long *)d_counts, base);
	HIPCHK(c, hipGetLastError());
	c->order_keys = d_keys; // (the tables of the scatter lie in the slab: the scatter must be this context's next call)
	c->order_n = n;
	return MSD_OK;
}
int msd_order_low16_scatter_u32(msd_ctx *c, const uint32_t *d_keys, uint64_t n, uint16_t *d_out)
{
	if (!c) return MSD_EINVAL;
	if (!d_out || (n && !d_keys)) return fail(c, MSD_EINVAL, "order_low16: null pointer");
	if (c->order_keys != d_keys || c->order_n != n) return fail(c, MSD_EINVAL, "order_low16_scatter: not preceded by msd_order_low16_counts_u32 on the same keys");
	c->order_keys = nullptr;
	if ((uintptr_t)d_out & 15) return fail(c, MSD_EINVAL, "order_low16: buffers must be 16-byte aligned");
	{
		const uintptr_t s0 = (uintptr_t)d_keys, s1 = s0 + n * 4, d0 = (uintptr_t)d_out, d1 = d0 + n * 2;
		if (s0 < d1 && d0 < s1) return fail(c, MSD_EINVAL, "order_low16: source and destination overlap");
	}
	HIPCHK(c, hipSetDevice(c->device));
	Bump real(c->slab); // (as carved by msd_order_low16_counts_u32)
	const uint64_t *pb = real.take<uint64_t>(257);
	(void)real.take<uint32_t>((size_t)65536 * kS16Chunks);
	const unsigned long long *base = real.take<unsigned long long>((size_t)65536 * kS16Chunks + 1);
	if (n)
		hipLaunchKernelGGL(scatter_low16_kernel, dim3(256 * kS16Chunks), dim3(kS16Th), kS16Lds, c->stream, d_keys, n, pb, base, d_out);
	HIPCHK(c, hipGetLastError());
	return MSD_OK;
}
int msd_order_low16_u32(msd_ctx *c, uint32_t *d_keys, uint64_t n, uint16_t *d_out, uint64_t *d_counts)
{
	if (!c) return MSD_EINVAL;
	if (!d_out) return fail(c, MSD_EINVAL, "order_low16: null pointer");
	int rc = msd_order_low16_counts_u32(c, d_keys, n, d_counts);
	if (!rc) rc = msd_order_low16_scatter_u32(c, d_keys, n, d_out);
	return rc;
}
int msd_pack_low16_u32(msd_ctx *c, const uint32_t *d_keys, uint64_t n, uint16_t *d_out)
{
	if (!c) return MSD_EINVAL;
	if (n == 0) return MSD_OK;
	if (!d_keys || !d_out) return fail(c, MSD_EINVAL, "pack_low16: null pointer");
	if (((uintptr_t)d_keys & 15) || ((uintptr_t)d_out & 15)) return fail(c, MSD_EINVAL, "pack_low16: buffers must be 16-byte aligned");
	{
		const uintptr_t s0 = (uintptr_t)d_keys, s1 = s0 + n * 4, d0 = (uintptr_t)d_out, d1 = d0 + n * 2;
		if (s0 < d1 && d0 < s1) return fail(c, MSD_EINVAL, "pack_low16: source and destination overlap");
	}
	HIPCHK(c, hipSetDevice(c->device));
	const unsigned grid = (unsigned)std::min<uint64_t>((uint64_t)c->sm_count * 16, (n / 8 + 255) / 256 + 1);
	hipLaunchKernelGGL(pack_low16_kernel, dim3(grid), dim3(256), 0, c->stream, d_keys, n, d_out);
	HIPCHK(c, hipGetLastError());
	return MSD_OK;
}

// ---- segmented sort and run gather: what a rank of the multi-GPU sort does with the keys it received

int msd_sort_u32_segments(msd_ctx *c, uint32_t *k, uint64_t n, const uint64_t *seg_off, uint32_t nseg, int end_bit)
{
	if (!c) return MSD_EINVAL;
	if (nseg == 0) return MSD_OK;
	if (!seg_off) return fail(c, MSD_EINVAL, "segments: null offsets");
	return sort_impl<uint32_t, NoVal>(c, k, nullptr, n, end_bit, false, 0, 0, nullptr, nullptr, 0, seg_off, nseg);
}
int msd_sort_u64_segments(msd_ctx *c, uint64_t *k, uint64_t n, const uint64_t *seg_off, uint32_t nseg, int end_bit)
{
	if (!c) return MSD_EINVAL;
	if (nseg == 0) return MSD_OK;
	if (!seg_off) return fail(c, MSD_EINVAL, "segments: null offsets");
	return sort_impl<uint64_t, NoVal>(c, k, nullptr, n, end_bit, false, 0, 0, nullptr, nullptr, 0, seg_off, nseg);
}
int msd_sort_pairs_u64_segments(msd_ctx *c, uint64_t *k, uint64_t *r, uint64_t n, const uint64_t *seg_off, uint32_t nseg, int end_bit)
{
	if (!c) return MSD_EINVAL;
	if (nseg == 0) return MSD_OK;
	if (!seg_off) return fail(c, MSD_EINVAL, "segments: null offsets");
	return sort_impl<uint64_t, uint64_t>(c, k, r, n, end_bit, false, 0, 0, nullptr, nullptr, 0, seg_off, nseg);
}

int msd_gather_runs_u32(msd_ctx *c, uint32_t *dst, const uint32_t *src, const uint64_t *src_off, const uint64_t *dst_off, const uint64_t *len, uint32_t nruns)
{
	return gather_impl<uint32_t>(c, dst, src, src_off, dst_off, len, nruns);
}
int msd_gather_runs_u64(msd_ctx *c, uint64_t *dst, const uint64_t *src, const uint64_t *src_off, const uint64_t *dst_off, const uint64_t *len, uint32_t nruns)
{
	return gather_impl<uint64_t>(c, dst, src, src_off, dst_off, len, nruns);
}

// ---- splitter service (reference: sampling src/msb_64.c:1511-1521, extract_delimiters :1304-1322, range function :188-204)

} // extern "C"

template <typename K> static int sample_impl(msd_ctx *c, const K *k, uint64_t n, uint64_t m, uint64_t seed, K *out)
{
	if (!c) return MSD_EINVAL;
	if (m && (!k || !out || n == 0)) return fail(c, MSD_EINVAL, "sample: null pointer or empty input");
	HIPCHK(c, hipSetDevice(c->device));
	if (m == 0) return MSD_OK;
	const unsigned grid = (unsigned)std::min<uint64_t>((uint64_t)c->sm_count * 8, (m + 255) / 256);
	hipLaunchKernelGGL((sample_kernel<K>), dim3(grid), dim3(256), 0, c->stream, k, n, m, seed, out);
	HIPCHK(c, hipGetLastError());
	return MSD_OK;
}

template <typename K> static int splitters_impl(msd_ctx *c, const K *sorted_sample, uint64_t m, unsigned parts, K *delims)
{
	if (!c) return MSD_EINVAL;
	if (parts < 1 || parts > 256) return fail(c, MSD_EINVAL, "splitters: parts must be 1..256");
	if (parts == 1) return MSD_OK;
	if (!sorted_sample || !delims || m == 0) return fail(c, MSD_EINVAL, "splitters: null pointer or empty sample");
	HIPCHK(c, hipSetDevice(c->device));
	hipLaunchKernelGGL((splitters_kernel<K>), dim3(1), dim3(256), 0, c->stream, sorted_sample, m, parts, delims);
	HIPCHK(c, hipGetLastError());
	return MSD_OK;
}

template <typename K, typename V>
static int range_partition_impl(msd_ctx *c, K *k, uint64_t *r, uint64_t n, const K *delims, unsigned parts, uint64_t *cnt)
{
	if (!c) return MSD_EINVAL;
	if (parts < 1 || parts > 256) return fail(c, MSD_EINVAL, "partition_by_splitters: parts must be 1..256");
	if (parts > 1 && !delims) return fail(c, MSD_EINVAL, "partition_by_splitters: null delimiters");
	HIPCHK(c, hipSetDevice(c->device));
	if (parts == 1) { // one range: nothing moves; its size goes to the device on the context's stream (ordered with the caller's work there)
		if (cnt) {
			int rcp = pinned_reserve(c, 64);
			if (rcp) return rcp;
			HIPCHK(c, hipStreamSynchronize(c->stream)); // the staging buffer may still be in flight
			memcpy(c->pinned, &n, sizeof n);
			HIPCHK(c, hipMemcpyAsync(cnt, c->pinned, sizeof n, hipMemcpyHostToDevice, c->stream));
			HIPCHK(c, hipStreamSynchronize(c->stream));
		}
		return MSD_OK;
	}
	if (cnt) HIPCHK(c, hipMemsetAsync(cnt, 0, sizeof(uint64_t) * parts, c->stream));
	unsigned width = 1;
	while ((1u << width) < parts) ++width;
	// one in-place round whose buckets are the ranges (parts - 1 delimiters; the buckets beyond `parts` stay empty)
	return sort_impl<K, V>(c, k, r, n, (int)sizeof(K) * 8, true, 0, width, cnt, delims, parts - 1);
}

extern "C" {

int msd_sample_u32(msd_ctx *c, const uint32_t *k, uint64_t n, uint64_t m, uint64_t seed, uint32_t *out) { return sample_impl<uint32_t>(c, k, n, m, seed, out); }
int msd_sample_u64(msd_ctx *c, const uint64_t *k, uint64_t n, uint64_t m, uint64_t seed, uint64_t *out) { return sample_impl<uint64_t>(c, k, n, m, seed, out); }
int msd_splitters_u32(msd_ctx *c, const uint32_t *s, uint64_t m, unsigned parts, uint32_t *d) { return splitters_impl<uint32_t>(c, s, m, parts, d); }
int msd_splitters_u64(msd_ctx *c, const uint64_t *s, uint64_t m, unsigned parts, uint64_t *d) { return splitters_impl<uint64_t>(c, s, m, parts, d); }
int msd_partition_by_splitters_u32(msd_ctx *c, uint32_t *k, uint64_t n, const uint32_t *delims, unsigned parts, uint64_t *cnt)
{
	return range_partition_impl<uint32_t, NoVal>(c, k, nullptr, n, delims, parts, cnt);
}
int msd_partition_by_splitters_u64(msd_ctx *c, uint64_t *k, uint64_t n, const uint64_t *delims, unsigned parts, uint64_t *cnt)
{
	return range_partition_impl<uint64_t, NoVal>(c, k, nullptr, n, delims, parts, cnt);
}
int msd_partition_by_splitters_pairs_u64(msd_ctx *c, uint64_t *k, uint64_t *r, uint64_t n, const uint64_t *delims, unsigned parts, uint64_t *cnt)
{
	if (c && !r) return fail(c, MSD_EINVAL, "partition_by_splitters: null rids");
	return range_partition_impl<uint64_t, uint64_t>(c, k, r, n, delims, parts, cnt);
}

} // extern "C"

template <typename K>
static int histogram_impl(msd_ctx *c, const K *k, uint64_t n, unsigned shift, unsigned rb, uint64_t *cnt)
{
	if (!c) return MSD_EINVAL;
	if (!cnt || (n && !k)) return fail(c, MSD_EINVAL, "null pointer");
	if (rb < 1 || rb > 12 || shift + rb > sizeof(K) * 8) return fail(c, MSD_EINVAL, "radix_bits must be 1..12 and shift+radix_bits within the key");
	if ((uintptr_t)k & 15) return fail(c, MSD_EINVAL, "keys must be 16-byte aligned");
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipMemsetAsync(cnt, 0, sizeof(uint64_t) << rb, c->stream));
	if (n == 0) return MSD_OK;
	const unsigned grid = (unsigned)std::min<uint64_t>((uint64_t)c->sm_count * 8, (n + 4095) / 4096);
	hipLaunchKernelGGL((histogram_kernel<K>), dim3(grid), dim3(256), sizeof(uint32_t) << rb, c->stream, k, n, shift, rb,
			   (unsigned long long *)cnt);
	HIPCHK(c, hipGetLastError());
	return MSD_OK;
}
extern "C" {

int msd_histogram_u32(msd_ctx *c, const uint32_t *k, uint64_t n, unsigned s, unsigned rb, uint64_t *cnt) { return histogram_impl(c, k, n, s, rb, cnt); }
int msd_histogram_u64(msd_ctx *c, const uint64_t *k, uint64_t n, unsigned s, unsigned rb, uint64_t *cnt) { return histogram_impl(c, k, n, s, rb, cnt); }

int msd_exclusive_scan_u64(msd_ctx *c, const uint64_t *in, uint64_t *out, uint64_t n)
{
	if (!c) return MSD_EINVAL;
	if (n && (!in || !out)) return fail(c, MSD_EINVAL, "null pointer");
	HIPCHK(c, hipSetDevice(c->device));
	if (n == 0) return MSD_OK;
	const size_t ntiles = (n + kScanTile - 1) / kScanTile;
	int rc = slab_reserve(c, ntiles * 8 + 4096);
	if (rc) return rc;
	unsigned long long *state = (unsigned long long *)(c->slab + 256);
	uint32_t *ctr = (uint32_t *)c->slab; // [0] tile counter, [2] error flag
	HIPCHK(c, hipMemsetAsync(c->slab, 0, 256, c->stream));
	HIPCHK(c, hipMemsetAsync(state, 0, ntiles * 8, c->stream));
	hipLaunchKernelGGL(scan_lookback_kernel, dim3((unsigned)ntiles), dim3(kScanTh), 0, c->stream, in, out, n, state, ctr, ctr + 2);
	HIPCHK(c, hipGetLastError());
	// the look-back gives up after a bounded number of polls and sets a flag: report it (one small readback)
	rc = pinned_reserve(c, 64);
	if (rc) return rc;
	HIPCHK(c, hipMemcpyAsync(c->pinned, ctr + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	if (*(const uint32_t *)c->pinned) return fail(c, MSD_EINTERNAL, "exclusive scan: a tile's look-back timed out");
	return MSD_OK;
}

} // extern "C"

template <typename K>
static int check_impl(msd_ctx *c, const K *k, const uint64_t *r, uint64_t n, uint64_t *viol, uint64_t *sum, uint64_t *xr)
{
	if (!c) return MSD_EINVAL;
	HIPCHK(c, hipSetDevice(c->device));
	int rc = slab_reserve(c, 4096);
	if (!rc) rc = pinned_reserve(c, 4096);
	if (rc) return rc;
	CheckResult *res = (CheckResult *)c->slab;
	HIPCHK(c, hipMemsetAsync(res, 0, sizeof *res, c->stream));
	if (n) {
		const unsigned grid = (unsigned)std::min<uint64_t>((uint64_t)c->sm_count * 8, (n + 255) / 256);
		hipLaunchKernelGGL((check_kernel<K>), dim3(grid), dim3(256), 0, c->stream, k, r, n, res);
		HIPCHK(c, hipGetLastError());
	}
	HIPCHK(c, hipMemcpyAsync(c->pinned, res, sizeof *res, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	const CheckResult *h = (const CheckResult *)c->pinned;
	if (viol) *viol = h->violations;
	if (sum) *sum = h->sum;
	if (xr) *xr = h->xr;
	return MSD_OK;
}
extern "C" {

int msd_check_u32(msd_ctx *c, const uint32_t *k, uint64_t n, uint64_t *v, uint64_t *s, uint64_t *x) { return check_impl<uint32_t>(c, k, nullptr, n, v, s, x); }
int msd_check_u64(msd_ctx *c, const uint64_t *k, const uint64_t *r, uint64_t n, uint64_t *v, uint64_t *s, uint64_t *x) { return check_impl<uint64_t>(c, k, r, n, v, s, x); }

static unsigned gen_grid(const msd_ctx *c, uint64_t n) { return (unsigned)std::min<uint64_t>((uint64_t)c->sm_count * 16, (n + 255) / 256 + 1); }

int msd_gen_uniform_u32(msd_ctx *c, uint32_t *k, uint64_t n, uint64_t seed, uint64_t first)
{
	if (!c) return MSD_EINVAL;
	HIPCHK(c, hipSetDevice(c->device));
	hipLaunchKernelGGL(gen_uniform_u32_kernel, dim3(gen_grid(c, n)), dim3(256), 0, c->stream, k, n, seed + first);
	HIPCHK(c, hipGetLastError());
	return MSD_OK;
}
int msd_gen_uniform_u64(msd_ctx *c, uint64_t *k, uint64_t n, uint64_t seed, uint64_t first, int shr)
{
	if (!c) return MSD_EINVAL;
	HIPCHK(c, hipSetDevice(c->device));
	hipLaunchKernelGGL(gen_uniform_u64_kernel, dim3(gen_grid(c, n)), dim3(256), 0, c->stream, k, n, seed + first, shr);
	HIPCHK(c, hipGetLastError());
	return MSD_OK;
}
int msd_gen_zipf_u32(msd_ctx *c, uint32_t *k, uint64_t n, uint64_t seed, uint64_t first)
{
	if (!c) return MSD_EINVAL;
	HIPCHK(c, hipSetDevice(c->device));
	hipLaunchKernelGGL(gen_zipf_u32_kernel, dim3(gen_grid(c, n)), dim3(256), 0, c->stream, k, n, seed + first);
	HIPCHK(c, hipGetLastError());
	return MSD_OK;
}
int msd_gen_dup_u32(msd_ctx *c, uint32_t *k, uint64_t n, uint64_t seed, uint64_t first, uint64_t distinct)
{
	if (!c) return MSD_EINVAL;
	if (distinct == 0) return fail(c, MSD_EINVAL, "gen_dup: distinct must be positive");
	HIPCHK(c, hipSetDevice(c->device));
	hipLaunchKernelGGL(gen_dup_u32_kernel, dim3(gen_grid(c, n)), dim3(256), 0, c->stream, k, n, seed + first, distinct);
	HIPCHK(c, hipGetLastError());
	return MSD_OK;
}
int msd_gen_mt19937_64(msd_ctx *c, uint64_t *k, uint64_t n, uint64_t seed, int shr)
{
	if (!c) return MSD_EINVAL;
	if (shr < 0 || shr > 63) return fail(c, MSD_EINVAL, "gen_mt19937_64: shift_right must be 0..63");
	HIPCHK(c, hipSetDevice(c->device));
	if (n == 0) return MSD_OK;
	hipLaunchKernelGGL(gen_mt19937_64_kernel, dim3(1), dim3(320), 0, c->stream, k, n, seed, shr);
	HIPCHK(c, hipGetLastError());
	return MSD_OK;
}
int msd_gen_iota_u64(msd_ctx *c, uint64_t *v, uint64_t n, uint64_t first)
{
	if (!c) return MSD_EINVAL;
	HIPCHK(c, hipSetDevice(c->device));
	hipLaunchKernelGGL(gen_iota_u64_kernel, dim3(gen_grid(c, n)), dim3(256), 0, c->stream, v, n, first);
	HIPCHK(c, hipGetLastError());
	return MSD_OK;
}

} // extern "C"

template <typename K, typename V>
static int plan_describe(uint64_t n, int end_bit, int cus, msd_plan *out)
{
	using C = Cfg<K, V>;
	constexpr bool HV = has_val<V>::value;
	const uint64_t small_max = (uint64_t)C::SORT_TH * C::SORT_KPT;
	const uint32_t count_bits = HV ? (uint32_t)kLeafCountBits : (uint32_t)kCountMaxBits;
	memset(out, 0, sizeof *out);
	out->block_elems = C::B;
	out->tile_elems = C::T;
	out->leaf_capacity = small_max;
	out->leaf_count_bits = count_bits;
	if (n <= small_max || end_bit <= 0) return MSD_OK; // a single LDS leaf
	std::vector<Segment> segs(1);
	segs[0] = { 0, n, (uint32_t)end_bit, 0 };
	RoundPlan rp;
	plan_round<K, V>(segs, small_max, cus, rp, count_bits);
	out->digit_width = rp.parents[0].width;
	out->digit_shift = rp.parents[0].shift;
	out->stripes = rp.stripes.size();
	out->stripe_elems = rp.stripes[0].end - rp.stripes[0].begin;
	// uniform keys: every round divides the segment size by 2^width
	uint64_t sz = n;
	uint32_t bits = (uint32_t)end_bit, rounds = 0;
	while (sz > small_max && bits > 0) {
		const uint32_t w = pick_width(sz, bits, small_max, count_bits);
		sz >>= w;
		bits -= w;
		++rounds;
	}
	out->expected_rounds = rounds;
	out->workspace_bytes = round_bytes_estimate<K, V>(n, cus) + keep_bytes_for<K, V>(n) + 4 * leaf_list_guess<K, V>(n) * sizeof(Segment);
	return MSD_OK;
}

extern "C" {

int msd_plan_first_round(uint64_t n, int key_bytes, int val_bytes, int end_bit, int cus, msd_plan *out)
{
	if (!out || cus <= 0 || end_bit < 0 || end_bit > key_bytes * 8) return MSD_EINVAL;
	if (key_bytes == 4 && val_bytes == 0) return plan_describe<uint32_t, NoVal>(n, end_bit, cus, out);
	if (key_bytes == 8 && val_bytes == 0) return plan_describe<uint64_t, NoVal>(n, end_bit, cus, out);
	if (key_bytes == 8 && val_bytes == 8) return plan_describe<uint64_t, uint64_t>(n, end_bit, cus, out);
	return MSD_EINVAL;
}

int msd_set_option(msd_ctx *c, const char *name, int64_t value)
{
	if (!c || !name) return MSD_EINVAL;
	if (!strcmp(name, "direct_mode")) {
		if (value < 0 || value > 2) return fail(c, MSD_EINVAL, "direct_mode must be 0, 1 or 2");
		c->direct_mode = (int)value;
	} else if (!strcmp(name, "direct_min")) {
		if (value < 1) return fail(c, MSD_EINVAL, "direct_min must be positive");
		c->direct_min = (uint64_t)value;
	} else if (!strcmp(name, "regpart")) {
		c->regpart = value != 0;
	} else if (!strcmp(name, "count16")) {
		if (value < 0 || value > 2) return fail(c, MSD_EINVAL, "count16 must be 0, 1 or 2");
		c->count16 = (int)value;
	} else if (!strcmp(name, "leaf17")) {
		c->leaf17 = value != 0;
	} else if (!strcmp(name, "stream_kernel")) {
		if (value < 1 || value > 2) return fail(c, MSD_EINVAL, "stream_kernel must be 1 or 2");
		c->stream_kernel = (int)value;
	} else if (!strcmp(name, "mid_leaf")) {
		c->mid_leaf = value != 0;
	} else if (!strcmp(name, "merge_leaf")) {
		if (value < 0 || value > 2) return fail(c, MSD_EINVAL, "merge_leaf must be 0, 1 or 2");
		c->merge_leaf = (int)value;
	} else if (!strcmp(name, "direct_kernel")) {
		(void)value; // (round 1's kernel is gone; the option is accepted for old callers)
	} else if (!strcmp(name, "direct_min_parent")) {
		if (value < 1) return fail(c, MSD_EINVAL, "direct_min_parent must be positive");
		c->direct_min_parent = (uint64_t)value;
	} else
		return fail(c, MSD_EINVAL, "unknown option %s", name);
	return MSD_OK;
}

#ifdef MSD_STAMPS
// diagnostic build only: per-section shader cycles of classify_direct_kernel, [wave 0 | last wave][section]; resets them
int msd_debug_stamps(uint64_t *out)
{
	unsigned long long h[2][16];
	if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof h) != hipSuccess) return MSD_EHIP;
	memcpy(out, h, sizeof h);
	memset(h, 0, sizeof h);
	if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), h, sizeof h) != hipSuccess) return MSD_EHIP;
	return MSD_OK;
}
#endif

int msd_set_profiling(msd_ctx *c, int on)
{
	if (!c) return MSD_EINVAL;
	c->profiling = on != 0;
	return MSD_OK;
}
int msd_phase_count(const msd_ctx *c) { return c ? (int)c->phase_us.size() : 0; }
const char *msd_phase_name(const msd_ctx *c, int i) { return (c && i >= 0 && i < (int)c->phase_us.size()) ? c->phase_us[i].first.c_str() : ""; }
double msd_phase_us(const msd_ctx *c, int i) { return (c && i >= 0 && i < (int)c->phase_us.size()) ? c->phase_us[i].second : 0.0; }
int msd_stat(const msd_ctx *c, const char *name, uint64_t *v)
{
	if (!c || !name || !v) return MSD_EINVAL;
	for (auto &s : c->stats)
		if (s.first == name) {
			*v = s.second;
			return MSD_OK;
		}
	return MSD_EINVAL;
}

} // extern "C"
