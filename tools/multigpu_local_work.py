#!/usr/bin/env python3
"""What one rank of the N-GPU sort does per step, timed on ONE GPU (RCCL cannot run here).

Two schemes (inplacemsdradixsort_amd/dist.py):
  fine   -- before the exchange the rank orders its (uniform, unmasked) shard by the top 16 bits and finds the bucket
            boundaries; after it, the counting leaf reads every bucket's G extents where they arrived (msd_merge_buckets_u32).
  coarse -- one top-digit pass before the exchange; after it the arrived runs are gathered bucket-major and sorted as
            segments on 24 bits (round 2's scheme), or sorted where they are on 32 - log2(N) bits.
"before the exchange" is timed on the rank's own shard (uniform over all 32 bits -- round 2's version of this tool
masked the keys first and so timed a partition whose keys sat in 256 / N buckets); "after the exchange" on keys laid out
as the all-to-all delivers them: N sources, each with its buckets of this rank's range in order.

    python tools/multigpu_local_work.py [ranks=8] [logn=30]     # one JSON line
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
if os.environ.get("MSD_VARIANT"):  # an experimental build (inplacemsdradixsort_amd._build.build_variant)
    from inplacemsdradixsort_amd import _build
    _build.LIB = os.path.join(_build.HERE, f"libinpmsdradix_hip_{os.environ['MSD_VARIANT']}.so")
    _build.stale = lambda: False
from inplacemsdradixsort_amd import MsdContext  # noqa: E402
from inplacemsdradixsort_amd.dist import bucket_major  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 30
lg = G.bit_length() - 1
n, per = 1 << logn, 256 // G
nbl = 65536 // G
ctx = MsdContext(0)
ctx.use_torch_stream()
keys = torch.empty(n, dtype=torch.int32, device="cuda")
arrived = torch.empty(n + 64, dtype=torch.int32, device="cuda")
work = torch.empty(n, dtype=torch.int32, device="cuda")
ctx.reserve(n + n // 8, 4, 0)


def timed(f):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


best = {}


def note(name, ms):
    best[name] = min(best.get(name, 1e9), ms)


stamps = None
from inplacemsdradixsort_amd import _lib  # noqa: E402
L = _lib.load(build_if_missing=False)


def read_stamps():
    """a -DMSD_STAMPS=8 build: cycles per section of merge_count_kernel for one more run of the leaf"""
    import ctypes as C
    NAMES = ["clear+ticket", "B", "count", "B", "scan", "segment search", "output (per-wave segments)", "-", "-", "loop", "-", "buckets"]
    L.msd_debug_stamps.argtypes = [C.POINTER(C.c_uint64)]
    buf = (C.c_uint64 * 32)()
    L.msd_debug_stamps(buf)
    ctx.merge_buckets(arrived, counts, base, 16, 0, work, n)
    torch.cuda.synchronize()
    L.msd_debug_stamps(buf)
    out = {}
    for wv, label in ((0, "wave0"), (1, "last_wave")):
        v = [int(buf[wv * 16 + i]) for i in range(12)]
        bk = max(1, v[11])
        out[label] = {f"{i}:{NAMES[i]}": round(v[i] / bk) for i in range(11)}
        out[label]["cycles_per_bucket"] = round(sum(v[:11]) / bk)
        out[label]["buckets"] = v[11]
    return out


for it in range(3):
    # ---- before the exchange: the rank's own shard, uniform over all 32 bits
    ctx.gen_uniform_u32(keys, seed=it)
    note("fine: order the shard by its top 16 bits", timed(lambda: ctx.sort_top(keys, 16)))
    note("fine: 65536 bucket boundaries", timed(lambda: ctx.bucket_bounds(keys, 16, 65536)))
    ctx.gen_uniform_u32(keys, seed=it)
    note("coarse: top-digit pass", timed(lambda: ctx.partition(keys, 24, 8)))

    # ---- after the exchange, fine: N sources, each n / N keys of this rank's range ordered by their top 16 bits
    ctx.gen_uniform_u32(keys, seed=100 + it)
    if G > 1:
        keys &= (1 << (32 - lg)) - 1                              # rank 0's range: 16-bit prefixes [0, 65536 / N)
    c0 = ctx.check(keys)
    a = arrived[:n]
    a.copy_(keys)
    chunk = n // G
    rows, base = [], []
    for s in range(G):
        part = a[s * chunk:(s + 1) * chunk]
        ctx.sort_top(part, 16)
        b = ctx.bucket_bounds(part, 16, nbl)
        rows.append(b[1:] - b[:-1])
        base.append(s * chunk)
    counts = torch.stack(rows).contiguous()
    work.fill_(-1)
    note("fine: counting leaf over the arrived extents", timed(lambda: ctx.merge_buckets(arrived, counts, base, 16, 0, work, n)))
    c1 = ctx.check(work)
    assert c1[0] == 0 and c1[1:] == c0[1:], (c0, c1)
    rejected = ctx.stats().get("merge_rejected", 0)
    # ---- the same with only the keys' low halves travelling: pack before the exchange, the leaf reads uint16 extents
    low = torch.empty(n + 64, dtype=torch.int16, device="cuda")
    note("fine, low halves: pack", timed(lambda: ctx.pack_low16(a, low)))
    work.fill_(-1)
    note("fine, low halves: counting leaf over the arrived extents", timed(lambda: ctx.merge_buckets(low, counts, base, 16, 0, work, n)))
    c1 = ctx.check(work)
    assert c1[0] == 0 and c1[1:] == c0[1:], (c0, c1)
    del low
    # ---- ... ordered and packed in one go (msd_order_low16_u32: one in-place round, exact counts, out-of-place scatter)
    if it == 0:
        low2 = torch.empty(n + 64, dtype=torch.int16, device="cuda")
        ok2 = torch.empty(n, dtype=torch.int32, device="cuda")
    ctx.gen_uniform_u32(ok2, seed=it)
    note("fine, low halves: order + pack in one go", timed(lambda: ctx.order_low16(ok2, low2)))
    # ---- ... and with the buckets travelling as histogram records: packed from the ordered shard before the exchange, the
    # leaf sums the G records of a bucket
    if it == 0:
        RB = ctx.HIST2_RECORD_BYTES
        rec_send = torch.empty(65536 * RB, dtype=torch.uint8, device="cuda")
        rec_recv = torch.empty(G * nbl * RB, dtype=torch.uint8, device="cuda")
        hk = torch.empty(n, dtype=torch.int32, device="cuda")
    ctx.gen_uniform_u32(hk, seed=it)
    ctx.sort_top(hk, 16)
    bnd = ctx.bucket_bounds(hk, 16, 65536)
    flag = [None]
    note("fine, histograms: pack", timed(lambda: flag.__setitem__(0, ctx.hist2_pack(hk, bnd, rec_send))))
    assert int(flag[0].item()) == 0
    cnt2 = ctx.order_low16(ok2, low2)      # (ok2 holds this iteration's shard ordered by its top 8 bits by now: any order does)
    bnd2 = ctx.bounds_from_counts16(cnt2)
    note("fine, histograms: pack from the low halves", timed(lambda: flag.__setitem__(0, ctx.hist2_pack(low2[:n], bnd2, rec_send))))
    assert int(flag[0].item()) == 0
    for s_ in range(G):   # what rank 0 receives: per source the records of ITS buckets (here: of the arrived extents)
        part = a[s_ * chunk:(s_ + 1) * chunk]
        bp = ctx.bucket_bounds(part, 16, nbl)
        f2 = ctx.hist2_pack(part, bp, rec_recv[s_ * nbl * RB:(s_ + 1) * nbl * RB])
        assert int(f2.item()) == 0
    work.fill_(-1)
    note("fine, histograms: leaf summing the arrived records", timed(lambda: ctx.merge_buckets(rec_recv, counts, [0] * G, 16, 0, work, n)))
    c1 = ctx.check(work)
    assert c1[0] == 0 and c1[1:] == c0[1:], (c0, c1)
    if hasattr(L, "msd_debug_stamps") and it == 2:
        stamps = read_stamps()

    # ---- after the exchange, coarse (round 2): per source its 256 / N top-digit buckets in order
    ctx.partition(keys, 24, 8)
    cnt = ctx.partition(keys, 24, 8).cpu().numpy()[:per]           # (idempotent: the keys are partitioned already)
    mine = [[int(c) // G + (int(c) % G if s == G - 1 else 0) for c in cnt] for s in range(G)]
    src_off, dst_off, lens, seg_off = bucket_major(mine)
    ctx.gather_runs(a, keys, dst_off, src_off, lens)                # bucket-major -> source-major: "as it arrived"
    note("coarse: gather runs bucket-major", timed(lambda: ctx.gather_runs(work, a, src_off, dst_off, lens)))
    note("coarse: segmented sort on 24 bits", timed(lambda: ctx.sort_segments(work, seg_off, 24)))
    note("coarse, in place: sort on %d bits" % (32 - lg), timed(lambda: ctx.sort_u32(a, end_bit=32 - lg)))
    c2, c3 = ctx.check(work), ctx.check(a)
    assert c2[0] == 0 and c2[1:] == c0[1:] and c3[0] == 0 and c3[1:] == c0[1:], (c0, c2, c3)

ms = {k: round(v, 3) for k, v in best.items()}
fine_pre = ms["fine: order the shard by its top 16 bits"] + ms["fine: 65536 bucket boundaries"]
fine_post = ms["fine: counting leaf over the arrived extents"]
coarse_post = min(ms["coarse: gather runs bucket-major"] + ms["coarse: segmented sort on 24 bits"], ms["coarse, in place: sort on %d bits" % (32 - lg)])
print(json.dumps({"ranks": G, "keys_per_rank": n, "ms_best_of_3": ms,
                  "fine": {"pre_ms": round(fine_pre, 3), "post_ms": round(fine_post, 3), "local_ms_per_step": round(fine_pre + fine_post, 3),
                           "merge_rejected_buckets": rejected},
                  "fine_low16": {"pre_ms": round(fine_pre + ms["fine, low halves: pack"], 3),
                                 "post_ms": ms["fine, low halves: counting leaf over the arrived extents"],
                                 "local_ms_per_step": round(fine_pre + ms["fine, low halves: pack"] + ms["fine, low halves: counting leaf over the arrived extents"], 3),
                                 "exchange_bytes_per_key": 2},
                  "fine_low16_fused": {"pre_ms": ms["fine, low halves: order + pack in one go"],
                                       "post_ms": ms["fine, low halves: counting leaf over the arrived extents"],
                                       "local_ms_per_step": round(ms["fine, low halves: order + pack in one go"] + ms["fine, low halves: counting leaf over the arrived extents"], 3),
                                       "exchange_bytes_per_key": 2},
                  "fine_hist_fused": {"pre_ms": round(ms["fine, low halves: order + pack in one go"] + ms["fine, histograms: pack from the low halves"], 3),
                                      "post_ms": ms["fine, histograms: leaf summing the arrived records"],
                                      "local_ms_per_step": round(ms["fine, low halves: order + pack in one go"] + ms["fine, histograms: pack from the low halves"]
                                                                 + ms["fine, histograms: leaf summing the arrived records"], 3),
                                      "exchange_bytes_per_key": round(65536 * 17408 / n, 3)},
                  "fine_hist": {"pre_ms": round(fine_pre + ms["fine, histograms: pack"], 3),
                                "post_ms": ms["fine, histograms: leaf summing the arrived records"],
                                "local_ms_per_step": round(fine_pre + ms["fine, histograms: pack"] + ms["fine, histograms: leaf summing the arrived records"], 3),
                                "exchange_bytes_per_key": round(65536 * 17408 / n, 3)},
                  "coarse": {"pre_ms": ms["coarse: top-digit pass"], "post_ms": round(coarse_post, 3),
                             "local_ms_per_step": round(ms["coarse: top-digit pass"] + coarse_post, 3)},
                  **({"stamps_merge_count_kernel": stamps} if stamps else {})}))
