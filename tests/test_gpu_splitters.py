"""Device splitter service (msd_sample_u32 / msd_splitters_u32 / msd_partition_by_splitters_u32) against the oracle's
restatement of the reference's front end -- sampling src/msb_64.c:1511-1521, extract_delimiters :1304-1322, the
lower-bound range function :188-204 -- and against the reference's own extract_delimiters (oracle/_ref) on the
same sample."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from inplacemsdradixsort_amd import MsdContext
    c = MsdContext(0)
    yield c
    c.close()


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()


def host(t):
    return t.cpu().numpy().view(np.uint32)


def make(kind, n, seed):
    if kind == "zipf":
        return O.gen_zipf_u32(n, seed=seed)
    if kind == "uniform":
        return O.gen_uniform_u32(n, seed=seed)
    if kind == "dup5":
        return (O.gen_uniform_u32(n, seed=seed) % 5).astype(np.uint32) * np.uint32(100000)
    if kind == "const":
        return np.full(n, 12345, np.uint32)
    raise ValueError(kind)


@pytest.mark.parametrize("n,m", [(1000, 64), (1 << 20, 65536), ((1 << 22) + 3, 500000)])
def test_sample_equals_oracle(ctx, n, m):
    k = O.gen_uniform_u32(n, seed=5)
    got = host(ctx.sample_u32(dev(k), m, seed=0xABCDEF))
    assert (got == O.sample_u32(k, m, seed=0xABCDEF)).all()


@pytest.mark.parametrize("kind", ["zipf", "uniform", "dup5", "const"])
@pytest.mark.parametrize("parts", [2, 4, 8, 64, 256])
def test_splitters_equal_oracle_and_reference(ctx, kind, parts):
    k = make(kind, 300000, seed=21)
    s = np.sort(O.sample_u32(k, 20000, seed=3))
    got = host(ctx.splitters_u32(dev(s), parts)).astype(np.uint64)
    want = O.extract_delimiters(s.astype(np.uint64), parts)
    assert (got == want).all()
    if O.have_ref():
        assert (got == O.ref_extract_delimiters(s.astype(np.uint64), parts)).all()


@pytest.mark.parametrize("kind", ["zipf", "uniform", "dup5", "const"])
@pytest.mark.parametrize("n,parts", [(5000, 2), (70001, 8), ((1 << 21) + 77, 8), (1 << 22, 64), (1 << 20, 256), (300000, 3)])
def test_partition_by_splitters_equals_oracle(ctx, kind, n, parts):
    """One in-place pass: range sizes == the oracle's range histogram, range p's keys are exactly the oracle's range-p
    multiset, ranges are contiguous and ascending."""
    k = make(kind, n, seed=parts + 1)
    s = np.sort(O.sample_u32(k, min(n, 20000), seed=9))
    d = O.extract_delimiters(s.astype(np.uint64), parts).astype(np.uint32)
    t = dev(k)
    cnt = ctx.partition_by_splitters(t, dev(d) if parts > 1 else None, parts).cpu().numpy().astype(np.uint64)
    want = O.range_histogram_u32(k, d.astype(np.uint64))
    assert (cnt == want).all()
    out = host(t)
    rid_in = O.range_of_u32(k, d)
    order = np.argsort(rid_in, kind="stable")
    exp_sorted_per_range = k[order]
    pos = 0
    for p in range(parts):
        c = int(cnt[p])
        assert (np.sort(out[pos:pos + c]) == np.sort(exp_sorted_per_range[pos:pos + c])).all(), p
        pos += c
    assert pos == n


def test_partition_by_splitters_full_size_zipf(ctx):
    """2^30 Zipf keys (BASELINE config C3's input) cut into 8 ranges by sampled splitters: sizes sum to n, every range
    within its delimiters, checksums preserved, balance far better than the top-3-bit radix split (75 % in range 0)."""
    import torch
    n = 1 << 30
    t = torch.empty(n, dtype=torch.int32, device="cuda")
    ctx.gen_zipf_u32(t)
    v0, s0, x0 = ctx.check(t)
    sample = ctx.sample_u32(t, 500000)          # the reference's sample size cap (src/msb_64.c:2320-2322)
    ctx.sort_u32(sample)
    d = ctx.splitters_u32(sample, 8)
    cnt = ctx.partition_by_splitters(t, d, 8).cpu().numpy()
    assert int(cnt.sum()) == n
    _, s1, x1 = ctx.check(t)
    assert (s1, x1) == (s0, x0)
    dh = host(d).astype(np.int64)
    pos = 0
    for p in range(8):
        c = int(cnt[p])
        if c:
            seg = t[pos:pos + c]
            u = (seg.to(torch.int64) & 0xFFFFFFFF)
            assert p == 0 or int(u.min().item()) > dh[p - 1]
            assert p == 7 or int(u.max().item()) <= dh[p]
            del u
        pos += c
    assert cnt.max() < 0.3 * n, cnt
    del t
    torch.cuda.empty_cache()


# ---- the same service on the reference's own key type: 64-bit keys, alone or with their rids

def dev64(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def host64(t):
    return t.cpu().numpy().view(np.uint64)


def make64(kind, n, seed):
    if kind == "zipf":        # Zipf ranks spread over 64 bits (small keys stay heavy)
        z = O.gen_zipf_u32(n, seed=seed).astype(np.uint64)
        return z * z + (z >> np.uint64(3))
    if kind == "uniform":
        return O.gen_uniform_u64(n, seed=seed)
    if kind == "dup5":
        return (O.gen_uniform_u64(n, seed=seed) % np.uint64(5)) * np.uint64(0x0123456789AB)
    if kind == "const":
        return np.full(n, 0xDEADBEEF12345678, np.uint64)
    raise ValueError(kind)


@pytest.mark.parametrize("n,m", [(1000, 64), ((1 << 21) + 3, 100000)])
def test_sample_u64_equals_oracle(ctx, n, m):
    k = O.gen_uniform_u64(n, seed=5)
    assert (host64(ctx.sample(dev64(k), m, seed=0xABCDEF)) == O.sample_u64(k, m, seed=0xABCDEF)).all()


@pytest.mark.parametrize("kind", ["zipf", "uniform", "dup5", "const"])
@pytest.mark.parametrize("parts", [2, 8, 64, 256])
def test_splitters_u64_equal_oracle_and_reference(ctx, kind, parts):
    k = make64(kind, 300000, seed=21)
    s = np.sort(O.sample_u64(k, 20000, seed=3))
    got = host64(ctx.splitters(dev64(s), parts))
    assert (got == O.extract_delimiters(s, parts)).all()
    if O.have_ref():
        assert (got == O.ref_extract_delimiters(s, parts)).all()   # the reference's own function, on its own key type


@pytest.mark.parametrize("kind", ["zipf", "uniform", "dup5", "const"])
@pytest.mark.parametrize("n,parts,pairs", [(5000, 2, False), (70001, 8, True), ((1 << 21) + 77, 8, False), (1 << 21, 64, True), (300000, 3, True)])
def test_partition_by_splitters_u64_equals_oracle(ctx, kind, n, parts, pairs):
    k = make64(kind, n, seed=parts + 1)
    s = np.sort(O.sample_u64(k, min(n, 20000), seed=9))
    d = O.extract_delimiters(s, parts)
    t = dev64(k)
    r = dev64(k ^ np.uint64(0x5A5A5A5A5A5A5A5A)) if pairs else None
    cnt = ctx.partition_by_splitters(t, dev64(d) if parts > 1 else None, parts, rids=r).cpu().numpy().astype(np.uint64)
    assert (cnt == O.range_histogram_u64(k, d)).all()
    out = host64(t)
    if pairs:
        assert (host64(r) == (out ^ np.uint64(0x5A5A5A5A5A5A5A5A))).all()   # every rid still with its key
    rid_in = O.range_of_u32(k, d)
    exp = k[np.argsort(rid_in, kind="stable")]
    pos = 0
    for p in range(parts):
        c = int(cnt[p])
        assert (np.sort(out[pos:pos + c]) == np.sort(exp[pos:pos + c])).all(), p
        pos += c
    assert pos == n
