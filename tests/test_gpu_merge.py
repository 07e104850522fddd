"""Fine-grained sharding building blocks (include/msd_radix_hip.h): msd_sort_*_top (a sort that stops at a bit),
msd_bucket_bounds_* and msd_merge_buckets_u32 (the counting leaf that reads a bucket's extents where the all-to-all put
them) -- against the oracle / numpy, through the C ABI."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def dev(a):
    import torch
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.int32 if a.dtype == np.uint32 else np.int64)).cuda()


def host(t, dt):
    return t.cpu().numpy().view(dt)


@pytest.mark.parametrize("begin_bit", [0, 8, 16, 20, 32])
@pytest.mark.parametrize("n", [1, 1000, 30_000, (1 << 20) + 13, 1 << 23])
def test_sort_u32_top(ctx, n, begin_bit):
    from oracle import oracle as O
    k = O.gen_uniform_u32(n, seed=n + begin_bit)
    t = dev(k)
    ctx.sort_top(t, begin_bit)
    out = host(t, np.uint32)
    top = out.astype(np.uint64) >> np.uint64(begin_bit)
    assert (np.diff(top.astype(np.int64)) >= 0).all()              # ordered by key >> begin_bit
    assert (np.sort(out) == O.sort_u32(k)).all()                   # the same keys


def test_sort_u32_top_skewed_and_constant(ctx):
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    for k in (O.gen_zipf_u32(1 << 21, seed=3), np.full(1 << 20, 0xDEADBEEF, dtype=np.uint32),
              (rng.integers(0, 1 << 12, 1 << 21, dtype=np.uint32)),   # every varying bit below begin_bit: nothing to do
              (rng.integers(0, 1 << 12, 1 << 21, dtype=np.uint32) << np.uint32(18))):
        t = dev(k)
        ctx.sort_top(t, 16)
        out = host(t, np.uint32)
        assert (np.diff((out >> np.uint32(16)).astype(np.int64)) >= 0).all()
        assert (np.sort(out) == np.sort(k)).all()


def test_sort_top_u64_and_pairs(ctx):
    from oracle import oracle as O
    n = (1 << 21) + 5
    k = O.gen_uniform_u64(n, seed=77)
    t = dev(k)
    ctx.sort_top(t, 40)
    out = host(t, np.uint64)
    assert (np.diff((out >> np.uint64(40)).astype(np.int64)) >= 0).all()
    assert (np.sort(out) == O.sort_u64(k)).all()
    tk, tr = dev(k), dev(k)
    ctx.sort_top(tk, 48, rids=tr)
    ok, orr = host(tk, np.uint64), host(tr, np.uint64)
    assert (ok == orr).all() and (np.diff((ok >> np.uint64(48)).astype(np.int64)) >= 0).all()
    assert (np.sort(ok) == O.sort_u64(k)).all()


@pytest.mark.parametrize("shift,nb,first", [(16, 1 << 16, 0), (24, 256, 0), (16, 8192, 8192 * 3), (20, 100, 4000)])
def test_bucket_bounds(ctx, shift, nb, first):
    rng = np.random.default_rng(shift + nb)
    k = np.sort(rng.integers(0, 1 << 32, 500_000, dtype=np.uint64).astype(np.uint32))
    b = ctx.bucket_bounds(dev(k), shift, nb, first).cpu().numpy()
    want = np.searchsorted(k.astype(np.uint64) >> np.uint64(shift), np.arange(first, first + nb + 1, dtype=np.uint64), side="left")
    assert (b == want).all()
    e = ctx.bucket_bounds(dev(np.zeros(0, dtype=np.uint32)), shift, nb, first).cpu().numpy()   # empty input
    assert (e == 0).all()


def _arrivals(rng, nsrc, nb, first, open_bits, per_bucket, gaps, heavy=None):
    """What a rank holds after a fine-grained exchange: per source, that source's buckets first .. first + nb - 1 in
    order (unsorted inside a bucket), sources back to back with ``gaps[x]`` elements of junk in front of source x."""
    rows, counts, base, at = [], np.zeros((nsrc, nb), dtype=np.int64), [], 0
    parts = []
    for x in range(nsrc):
        c = rng.poisson(per_bucket / nsrc, nb).astype(np.int64)
        if heavy is not None:
            c[heavy[0]] = heavy[1] // nsrc
        counts[x] = c
        pre = np.repeat(np.arange(first, first + nb, dtype=np.uint64), c)
        low = rng.integers(0, 1 << open_bits, int(c.sum()), dtype=np.uint64)
        keys = ((pre << np.uint64(open_bits)) | low).astype(np.uint32)
        parts.append(np.full(gaps[x], 0xFFFFFFFF, dtype=np.uint32))
        at += gaps[x]
        base.append(at)
        parts.append(keys)
        rows.append(keys)
        at += keys.size
    parts.append(np.full(8, 0xFFFFFFFF, dtype=np.uint32))
    return np.concatenate(parts), counts, base, np.concatenate(rows)


@pytest.mark.parametrize("nsrc,nb,per_bucket,open_bits", [
    (2, 512, 16384, 16), (4, 512, 16384, 16), (8, 512, 16384, 16),     # the multi-GPU shape: 2^14-key buckets
    (3, 300, 9000, 16), (8, 2000, 700, 12), (5, 64, 17000, 16), (8, 128, 17300, 16),   # odd source counts, narrow values, nearly full buckets
    (1, 100, 5000, 16), (8, 4096, 3, 9), (2, 100, 500, 6),
])
def test_merge_buckets_equals_sort(ctx, nsrc, nb, per_bucket, open_bits):
    import torch
    rng = np.random.default_rng(nsrc * 1000 + nb)
    first = 5 * nb if (5 * nb + nb) <= (1 << (32 - open_bits)) else 0
    gaps = [int(g) for g in rng.integers(0, 7, nsrc)]
    src, counts, base, allk = _arrivals(rng, nsrc, nb, first, open_bits, per_bucket, gaps)
    n = int(counts.sum())
    dst = torch.full((n + 5,), -1, dtype=torch.int32, device="cuda")
    ctx.merge_buckets(dev(src), torch.from_numpy(counts).cuda(), base, open_bits, first, dst, n)
    out = host(dst, np.uint32)
    assert (out[:n] == np.sort(allk)).all()
    assert (out[n:] == 0xFFFFFFFF).all()                              # nothing written behind the last bucket


@pytest.mark.parametrize("nsrc,nb,per_bucket,open_bits", [
    (8, 24, 131072, 16), (4, 40, 65536, 16), (2, 64, 32768, 16),       # 2^30 keys per rank: nsrc x 2^14 keys per bucket
    (3, 50, 40000, 16), (8, 300, 3000, 12), (1, 10, 200000, 16), (8, 2000, 5, 7), (5, 6, 700000, 16), (2, 3, 900000, 16),
])
def test_merge_buckets_counting_kernel(ctx, nsrc, nb, per_bucket, open_bits):
    """merge_count_kernel forced (it is chosen by bucket size otherwise): 16-bit LDS counters, value-parallel tiles."""
    import torch
    rng = np.random.default_rng(nsrc * 77 + nb)
    first = 3 * nb
    gaps = [int(g) for g in rng.integers(0, 7, nsrc)]
    src, counts, base, allk = _arrivals(rng, nsrc, nb, first, open_bits, per_bucket, gaps)
    counts[:, nb // 2] = 0                                            # an empty bucket in the middle
    src, allk = _rebuild(src, counts, base, gaps, first, open_bits, rng)
    n = int(counts.sum())
    dst = torch.full((n + 5,), -1, dtype=torch.int32, device="cuda")
    ctx.set_option("merge_leaf", 2)
    try:
        ctx.merge_buckets(dev(src), torch.from_numpy(counts).cuda(), base, open_bits, first, dst, n)
    finally:
        ctx.set_option("merge_leaf", 0)
    out = host(dst, np.uint32)
    assert (out[:n] == np.sort(allk)).all()
    assert (out[n:] == 0xFFFFFFFF).all()


def _rebuild(src, counts, base, gaps, first, open_bits, rng):
    """Arrival buffer for edited ``counts`` (same layout rules as _arrivals)."""
    nsrc, nb = counts.shape
    parts, rows, at = [], [], 0
    for x in range(nsrc):
        c = counts[x]
        pre = np.repeat(np.arange(first, first + nb, dtype=np.uint64), c)
        low = rng.integers(0, 1 << open_bits, int(c.sum()), dtype=np.uint64)
        keys = ((pre << np.uint64(open_bits)) | low).astype(np.uint32)
        parts.append(np.full(gaps[x], 0xFFFFFFFF, dtype=np.uint32))
        at += gaps[x]
        base[x] = at
        parts.append(keys)
        rows.append(keys)
        at += keys.size
    parts.append(np.full(8, 0xFFFFFFFF, dtype=np.uint32))
    return np.concatenate(parts), np.concatenate(rows)


def test_merge_counting_kernel_duplicates(ctx):
    """Long runs of equal keys (filled by whole waves), a key with more copies than a 16-bit counter holds and a
    256-value group with more keys than a 16-bit offset addresses (both: not taken, finished by the general leaves)."""
    import torch
    rng = np.random.default_rng(4)
    nsrc, nb = 4, 6
    counts = np.full((nsrc, nb), 30000, dtype=np.int64)
    base, gaps = [0] * nsrc, [1, 0, 3, 2]
    src, allk = _rebuild(None, counts, base, gaps, 0, 16, rng)
    def bucket_slice(x, j):
        a = base[x] + int(counts[x, :j].sum())
        return slice(a, a + int(counts[x, j]))
    for x in range(nsrc):
        src[bucket_slice(x, 1)] &= np.uint32(0xFFFF3F00)              # bucket 1: 64 distinct keys in 64 groups, ~1900 copies each
        src[bucket_slice(x, 2)] = np.uint32((2 << 16) | 77)           # bucket 2: one key, 120000 copies
        src[bucket_slice(x, 4)] &= np.uint32(0xFFFF00FF)              # bucket 4: all keys in one 256-value group
    allk = np.concatenate([src[base[x]:base[x] + int(counts[x].sum())] for x in range(nsrc)])
    n = int(counts.sum())
    dst = torch.empty(n, dtype=torch.int32, device="cuda")
    ctx.set_option("merge_leaf", 2)
    try:
        ctx.merge_buckets(dev(src), torch.from_numpy(counts).cuda(), base, 16, 0, dst, n)
    finally:
        ctx.set_option("merge_leaf", 0)
    assert (host(dst, np.uint32) == np.sort(allk)).all()
    assert ctx.stats().get("merge_rejected", 0) == 2


def _low16_arrivals(src, counts, base):
    """The same arrival buffer holding only the keys' low halves: every offset counts uint16 elements."""
    return (src & np.uint32(0xFFFF)).astype(np.uint16), base


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 1000, (1 << 20) + 5, 1 << 24])
def test_pack_low16(ctx, n):
    import torch
    from oracle import oracle as O
    k = O.gen_uniform_u32(n, seed=n + 1) if n else np.zeros(0, dtype=np.uint32)
    out = torch.full((n + 8,), -1, dtype=torch.int16, device="cuda")
    ctx.pack_low16(dev(k), out)
    got = out.cpu().numpy().view(np.uint16)
    assert (got[:n] == (k & np.uint32(0xFFFF)).astype(np.uint16)).all() and (got[n:] == 0xFFFF).all()


def _check_order_low16(ctx, k):
    """out holds, bucket after bucket (bucket = upper half), exactly the low halves of that bucket's keys"""
    import torch
    n = k.size
    t = dev(k)
    out = torch.full((n + 8,), -1, dtype=torch.int16, device="cuda")
    counts = ctx.order_low16(t, out).cpu().numpy()
    want = np.bincount(k >> np.uint32(16), minlength=65536)
    assert (counts == want).all()
    got = out.cpu().numpy().view(np.uint16)
    assert (got[n:] == 0xFFFF).all()                                  # nothing written behind the last bucket
    # rebuild the keys from (bucket, low half) and compare the sorted multisets
    rebuilt = (np.repeat(np.arange(65536, dtype=np.uint32), want) << np.uint32(16)) | got[:n].astype(np.uint32)
    assert (np.sort(rebuilt) == np.sort(k)).all()
    assert (np.sort(host(t, np.uint32)) == np.sort(k)).all()          # the keys themselves are only reordered ...
    top = host(t, np.uint32) >> np.uint32(24)
    assert (np.diff(top.astype(np.int64)) >= 0).all()                 # ... by their top 8 bits


@pytest.mark.parametrize("n", [1, 5, 1000, 70_001, (1 << 20) + 3, 1 << 23, (1 << 25) + 77])
def test_order_low16_uniform(ctx, n):
    from oracle import oracle as O
    _check_order_low16(ctx, O.gen_uniform_u32(n, seed=n))


def test_order_low16_skewed_and_narrow(ctx):
    """Zipf keys (one bucket holds a quarter of the keys: its LDS buffer overflows tile after tile), keys with a constant
    upper half, keys in a handful of top bytes, sorted and reversed keys."""
    from oracle import oracle as O
    rng = np.random.default_rng(61)
    n = (1 << 21) + 11
    for k in (O.gen_zipf_u32(n, seed=5), (rng.integers(0, 1 << 16, n, dtype=np.uint32) | np.uint32(0xABCD0000)),
              rng.integers(0, 1 << 26, n, dtype=np.uint32), np.sort(rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)),
              np.sort(rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32))[::-1].copy(), np.full(n, 0x12345678, dtype=np.uint32)):
        _check_order_low16(ctx, k)


@pytest.mark.parametrize("nsrc,nb,per_bucket", [
    (8, 24, 131072), (4, 40, 65536), (2, 64, 32768),                   # 2^30 keys per rank: nsrc x 2^14 keys per bucket
    (3, 50, 40000), (8, 300, 3000), (1, 10, 200000), (8, 2000, 5), (5, 6, 700000), (2, 512, 4096), (8, 512, 16384),
])
def test_merge_buckets_from_low_halves(ctx, nsrc, nb, per_bucket):
    """msd_merge_buckets_u32_low16: the extents hold uint16 low halves (what the fine exchange moves), the leaf puts the
    bucket numbers back: the output equals the sort of the whole keys."""
    import torch
    rng = np.random.default_rng(nsrc * 91 + nb)
    first = 3 * nb
    gaps = [int(g) for g in rng.integers(0, 13, nsrc)]
    src, counts, base, allk = _arrivals(rng, nsrc, nb, first, 16, per_bucket, gaps)
    counts[:, nb // 2] = 0                                            # an empty bucket in the middle
    src, allk = _rebuild(src, counts, base, gaps, first, 16, rng)
    s16, base16 = _low16_arrivals(src, counts, base)
    n = int(counts.sum())
    dst = torch.full((n + 5,), -1, dtype=torch.int32, device="cuda")
    t16 = torch.from_numpy(s16.view(np.int16)).cuda()
    ctx.merge_buckets(t16, torch.from_numpy(counts).cuda(), base16, 16, first, dst, n)
    out = host(dst, np.uint32)
    assert (out[:n] == np.sort(allk)).all()
    assert (out[n:] == 0xFFFFFFFF).all()


def test_merge_low_halves_duplicates_are_finished(ctx):
    """Buckets the counting leaf does not take (a key with more copies than a 16-bit counter holds, a crowded 256-value
    group) are written out as whole keys and finished by the general leaves."""
    import torch
    rng = np.random.default_rng(41)
    nsrc, nb = 4, 6
    counts = np.full((nsrc, nb), 30000, dtype=np.int64)
    base, gaps = [0] * nsrc, [1, 0, 3, 2]
    src, allk = _rebuild(None, counts, base, gaps, 0, 16, rng)
    def bucket_slice(x, j):
        a = base[x] + int(counts[x, :j].sum())
        return slice(a, a + int(counts[x, j]))
    for x in range(nsrc):
        src[bucket_slice(x, 2)] = np.uint32((2 << 16) | 77)           # bucket 2: one key, 120000 copies
        src[bucket_slice(x, 4)] &= np.uint32(0xFFFF00FF)              # bucket 4: all keys in one 256-value group
    allk = np.concatenate([src[base[x]:base[x] + int(counts[x].sum())] for x in range(nsrc)])
    n = int(counts.sum())
    dst = torch.empty(n, dtype=torch.int32, device="cuda")
    s16, base16 = _low16_arrivals(src, counts, base)
    ctx.merge_buckets(torch.from_numpy(s16.view(np.int16)).cuda(), torch.from_numpy(counts).cuda(), base16, 16, 0, dst, n)
    assert (host(dst, np.uint32) == np.sort(allk)).all()
    assert ctx.stats().get("merge_rejected", 0) == 2


def _hist2_sources(ctx, rng, nsrc, nb, first, per_bucket, edit=None):
    """Per source a shard ordered by its upper halves over buckets first .. first + nb - 1, its histogram records and bucket
    sizes: (records back to back source-major, counts[nsrc, nb], all keys, overflow flags)."""
    import torch
    recs, counts, allk, flags = [], np.zeros((nsrc, nb), dtype=np.int64), [], []
    for x in range(nsrc):
        c = rng.poisson(per_bucket, nb).astype(np.int64)
        pre = np.repeat(np.arange(first, first + nb, dtype=np.uint64), c)
        keys = ((pre << np.uint64(16)) | rng.integers(0, 1 << 16, int(c.sum()), dtype=np.uint64)).astype(np.uint32)
        if edit is not None:
            keys = edit(x, keys)
            keys = keys[np.argsort(keys >> np.uint32(16), kind="stable")]
        t = dev(keys)
        b = ctx.bucket_bounds(t, 16, nb, first)
        rec = torch.empty(nb * ctx.HIST2_RECORD_BYTES, dtype=torch.uint8, device="cuda")
        flags.append(int(ctx.hist2_pack(t, b, rec).item()))
        recs.append(rec)
        counts[x] = (b[1:] - b[:-1]).cpu().numpy()
        allk.append(keys)
    return torch.cat(recs), counts, np.concatenate(allk), flags


@pytest.mark.parametrize("nsrc,nb,per_bucket", [(8, 24, 16384), (4, 40, 16384), (2, 64, 16384), (1, 30, 17000), (3, 50, 3), (8, 16, 12000), (5, 33, 1000)])
def test_hist2_records_sum_to_the_sorted_bucket(ctx, nsrc, nb, per_bucket):
    """msd_hist2_pack_u32 + msd_merge_buckets_u32_hist2: the histogram form of the fine exchange.  Every source packs its
    buckets into records; the sum of a bucket's records, written out, is the sort of all its keys."""
    import torch
    rng = np.random.default_rng(nsrc * 131 + nb)
    first = 3 * nb
    rec, counts, allk, flags = _hist2_sources(ctx, rng, nsrc, nb, first, per_bucket)
    assert flags == [0] * nsrc
    n = int(counts.sum())
    dst = torch.full((n + 5,), -1, dtype=torch.int32, device="cuda")
    ctx.merge_buckets(rec, torch.from_numpy(counts).cuda(), [0] * nsrc, 16, first, dst, n)
    out = host(dst, np.uint32)
    assert (out[:n] == np.sort(allk)).all() and (out[n:] == 0xFFFFFFFF).all()


def test_hist2_records_from_low_halves(ctx):
    """msd_hist2_pack_u32_low16: the records packed from what msd_order_low16_u32 wrote equal those packed from the keys
    ordered in place (the fields are a function of the bucket's multiset; the entries may come in another order)."""
    import torch
    from oracle import oracle as O
    n = (1 << 22) + 5
    k = O.gen_uniform_u32(n, seed=123) & np.uint32(0x00FFFFFF)        # 256 buckets of about 2^14 keys
    k[:900] = np.uint32(0x00340000) | (np.arange(900, dtype=np.uint32) % np.uint32(50))   # 50 values with 18 copies each
    RB = ctx.HIST2_RECORD_BYTES
    t1 = dev(k)
    ctx.sort_top(t1, 16)
    b1 = ctx.bucket_bounds(t1, 16, 256)
    r1 = torch.zeros(256 * RB, dtype=torch.uint8, device="cuda")
    f1 = ctx.hist2_pack(t1, b1, r1)
    t2 = dev(k)
    low = torch.empty(n, dtype=torch.int16, device="cuda")
    counts = ctx.order_low16(t2, low)
    b2 = ctx.bounds_from_counts16(counts)
    assert (b2[:257].cpu().numpy() == b1.cpu().numpy()).all() and int(b2[65536]) == n
    r2 = torch.zeros(256 * RB, dtype=torch.uint8, device="cuda")
    f2 = ctx.hist2_pack(low, b2[:257].contiguous(), r2)
    assert int(f1.item()) == int(f2.item()) == 0
    a1, a2 = r1.cpu().numpy().reshape(256, RB), r2.cpu().numpy().reshape(256, RB)
    assert (a1[:, :16384] == a2[:, :16384]).all()
    e1, e2 = a1[:, 16384:].copy().view(np.uint32), a2[:, 16384:].copy().view(np.uint32)
    assert (e1[:, 0] == e2[:, 0]).all() and int(e1[:, 0].max()) > 100
    for j in range(256):
        c = int(e1[j, 0])
        assert (np.sort(e1[j, 1:1 + c]) == np.sort(e2[j, 1:1 + c])).all()


def test_hist2_many_copies_and_overflow(ctx):
    """Values with three and more copies travel as (value, copies) entries; a 256-value group with more keys in all than a
    16-bit offset addresses makes the receiver write the sources' histograms out one after the other (finished by the general
    leaves); more than 255 such values in one bucket, a value with more than 255 copies, or a bucket of more than 65535
    keys set the sender's overflow flag."""
    import torch
    rng = np.random.default_rng(17)
    nsrc, nb, first = 4, 8, 100

    def edit(x, keys):
        b = keys >> np.uint32(16)
        k = keys.copy()
        sel = np.flatnonzero(b == first + 1)
        k[sel[:200]] = np.uint32(((first + 1) << 16) | 4242)                 # 200 copies per source: 800 in all
        sel = np.flatnonzero(b == first + 2)
        k[sel[:250]] = np.uint32(((first + 2) << 16) | 7)                    # ... and 250 of another (a byte counter holds 255)
        sel = np.flatnonzero(b == first + 3)
        k[sel] = (k[sel] & np.uint32(0xFFFF0000)) | ((k[sel] & np.uint32(0xFFFF)) % np.uint32(200))   # bucket 3: 200 values only, 80 copies each
        keep = k[b != first + 5]                                             # bucket 5: 250 values of ONE 256-value group, 240 copies each
        crowd = np.uint32((first + 5) << 16) | np.repeat(np.arange(250, dtype=np.uint32), 240)      # per source: 240000 keys in the group in all
        return np.concatenate([keep, crowd])

    rec, counts, allk, flags = _hist2_sources(ctx, rng, nsrc, nb, first, 16384, edit)
    assert flags == [0] * nsrc
    n = int(counts.sum())
    dst = torch.empty(n, dtype=torch.int32, device="cuda")
    ctx.merge_buckets(rec, torch.from_numpy(counts).cuda(), [0] * nsrc, 16, first, dst, n)
    assert (host(dst, np.uint32) == np.sort(allk)).all()
    assert ctx.stats().get("merge_rejected", 0) == 1                         # bucket 5: the group's 240000 keys do not fit a 16-bit offset
    # the sender's limits
    k = np.sort(rng.integers(0, 1 << 16, 3000, dtype=np.uint32).repeat(3))   # 3000 values with three copies each
    t = dev(k)
    b = ctx.bucket_bounds(t, 16, 1, 0)
    r1 = torch.empty(ctx.HIST2_RECORD_BYTES, dtype=torch.uint8, device="cuda")
    assert int(ctx.hist2_pack(t, b, r1).item()) != 0
    k = rng.integers(0, 1 << 16, 70000, dtype=np.uint32)                     # a bucket of 70000 keys
    t = dev(k)
    assert int(ctx.hist2_pack(t, ctx.bucket_bounds(t, 16, 1, 0), r1).item()) != 0
    k = np.concatenate([rng.permutation(30000).astype(np.uint32), np.full(256, 31000, dtype=np.uint32)])   # a value with 256 copies
    t = dev(k)
    assert int(ctx.hist2_pack(t, ctx.bucket_bounds(t, 16, 1, 0), r1).item()) != 0
    k = rng.permutation(60000).astype(np.uint32)                             # 60000 different keys: fits
    t = dev(k)
    assert int(ctx.hist2_pack(t, ctx.bucket_bounds(t, 16, 1, 0), r1).item()) == 0


def test_merge_buckets_rejections_are_finished(ctx):
    """Buckets the leaf does not take -- longer than it holds, more than 255 copies of one key -- go through the
    general leaves and still come out sorted."""
    import torch
    rng = np.random.default_rng(9)
    src, counts, base, allk = _arrivals(rng, 4, 256, 0, 16, 8000, [0, 1, 2, 3], heavy=(17, 60000))
    # ... and a bucket that is all one key (byte counters overflow)
    x0, j = 0, 40
    a = base[x0] + int(counts[x0, :j].sum())
    src[a:a + counts[x0, j]] = (np.uint32(j) << np.uint32(16)) | np.uint32(0x1234)
    allk = np.concatenate([src[base[x]:base[x] + int(counts[x].sum())] for x in range(4)])
    n = int(counts.sum())
    dst = torch.empty(n, dtype=torch.int32, device="cuda")
    ctx.merge_buckets(dev(src), torch.from_numpy(counts).cuda(), base, 16, 0, dst, n)
    assert (host(dst, np.uint32) == np.sort(allk)).all()
    assert ctx.stats().get("merge_rejected", 0) >= 1


def test_merge_buckets_argument_errors(ctx):
    import torch
    from inplacemsdradixsort_amd.api import MsdError
    src = torch.zeros(1024, dtype=torch.int32, device="cuda")
    dst = torch.zeros(1024, dtype=torch.int32, device="cuda")
    counts = torch.full((2, 4), 100, dtype=torch.int64, device="cuda")
    with pytest.raises(MsdError):
        ctx.merge_buckets(src, counts, [0, 400], 16, 0, dst, 801)     # the counts add up to 800
    with pytest.raises(MsdError):
        ctx.merge_buckets(src, counts, [0, 400], 16, 0, src, 800)     # overlap
    with pytest.raises(MsdError):
        ctx.merge_buckets(src, counts, [0, 400], 17, 0, dst, 800)     # open bits
    with pytest.raises(MsdError):
        ctx.merge_buckets(src, counts, [0, 400], 16, 0, dst[:100], 800)   # output too small
