"""Every kernel variant a tuning knob selects gives the oracle's result: the specialised 16-bit counting leaf
(count16 = 0 / 1 / 2), the register-resident partition round (regpart = 0 / 1), direct placement on and off --
so that A/B switches used for measurements cannot hide a wrong path."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    from inplacemsdradixsort_amd import MsdContext
    c = MsdContext(0)
    yield c
    c.close()


def dev(a):
    import torch
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.int32 if a.dtype == np.uint32 else np.int64)).cuda()


def host(t):
    a = t.cpu().numpy()
    return a.view(np.uint32) if a.dtype == np.int32 else a.view(np.uint64)


@pytest.mark.parametrize("count16", [0, 1, 2])
@pytest.mark.parametrize("n,kind", [((1 << 22) + 5, "uniform"), (1 << 24, "uniform"), ((1 << 23) + 3, "dup"), (3_000_001, "zipf")])
def test_count16_modes(ctx, count16, n, kind):
    """u32 keys whose leaves have 16 open bits: the same result whichever counting leaf takes them (segments of 2^6 ..
    2^14 keys, duplicates that overflow byte counters, segments starting at any element)."""
    k = {"uniform": lambda: O.gen_uniform_u32(n, seed=41), "zipf": lambda: O.gen_zipf_u32(n, seed=42),
         "dup": lambda: O.gen_dup_u32(n, 5000, seed=43)}[kind]()
    ctx.set_option("count16", count16)
    ctx.set_option("direct_min", 1 << 20)
    t = dev(k)
    ctx.sort_u32(t)
    assert (host(t) == np.sort(k)).all()
    # a sub-array that starts 16 bytes into the allocation and ends on an odd element
    t2 = dev(k)
    ctx.sort_u32(t2[4:n - 3])
    assert (host(t2[4:n - 3]) == np.sort(k[4:n - 3])).all() and (host(t2[:4]) == k[:4]).all() and (host(t2[n - 3:]) == k[n - 3:]).all()


@pytest.mark.parametrize("regpart,leaf17", [(0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("n,shr", [(300_001, 0), ((1 << 21) + 7, 0), (1 << 22, 40), (4_000_000, 0), (4_000_000, 30)])
def test_regpart_modes_pairs(ctx, regpart, leaf17, n, shr):
    """(key, rid) tuples whose last partition round has parents of a few thousand tuples: with the general round, with
    the register-resident pass + small leaves, and with the one-pass leaf for segments of <= 17408 tuples (leaf17_kernel)
    the key sequence is the oracle's and the rids follow their keys."""
    import torch
    k = O.gen_uniform_u64(n, seed=51) >> np.uint64(shr)
    ctx.set_option("regpart", regpart)
    ctx.set_option("leaf17", leaf17)
    try:
        tk = dev(k)
        tr = torch.arange(n, dtype=torch.int64, device="cuda")
        ctx.sort_pairs_u64(tk, tr)
    finally:
        ctx.set_option("regpart", 1)
        ctx.set_option("leaf17", 1)
    out, rid = host(tk), tr.cpu().numpy()
    assert (out == np.sort(k)).all()
    assert (k[rid] == out).all() and (np.sort(rid) == np.arange(n)).all()
    st = ctx.stats()
    if regpart == 0:
        assert st.get("regpart_rounds", 0) == 0 and st.get("leaf17_segments", 0) == 0
    elif n == 4_000_000:               # the first round leaves 256 parents of about 15.6 Ki tuples: they fit
        if leaf17:
            assert st.get("leaf17_segments", 0) >= 1 and st.get("regpart_rounds", 0) == 0, st
        else:
            assert st.get("regpart_rounds", 0) >= 1, st


def test_leaf17_duplicates_and_rejected_segments(ctx):
    """leaf17_kernel's corners: duplicates inside a segment (ties are put in order by the neighbour fix-up) and segments
    with runs of equal counted bits longer than the fix-up follows (rejected: the register partition + the small leaves
    finish them)."""
    import torch
    rng = np.random.default_rng(77)
    n = 4_000_001
    k = rng.integers(0, 1 << 64, n, dtype=np.uint64)                # (256 first-round buckets of about 15.6 Ki tuples: they fit the leaf)
    k[5::5] = k[4::5][: k[5::5].size]                                # every fifth key repeats its neighbour
    # one top-byte bucket whose 20 varying bits take only 8 values on their top 13 (the counted ones): thousands of tuples
    # per counted value, seven more bits to put in order
    sel = (k >> np.uint64(56)) == np.uint64(9)
    c = (k[sel] >> np.uint64(20)) & np.uint64(7)
    k[sel] = (np.uint64(9) << np.uint64(56)) | ((c * np.uint64(0x1249)) & np.uint64(0x1FFF)) << np.uint64(7) | (k[sel] & np.uint64(127))
    tk, tr = dev(k), torch.arange(n, dtype=torch.int64, device="cuda")
    ctx.sort_pairs_u64(tk, tr)
    out, rid = host(tk), tr.cpu().numpy()
    assert (out == np.sort(k)).all() and (k[rid] == out).all() and (np.sort(rid) == np.arange(n)).all()
    st = ctx.stats()
    assert st.get("leaf17_segments", 0) >= 1 and st.get("leaf17_rejected", 0) >= 1 and st.get("leaf17_slow_segments", 0) >= 1, st


def test_leaf17_long_groups_and_many_groups(ctx):
    """leaf17_kernel's position-by-position fix-up (the fallback of the one-lane-per-group fix-up): a segment whose groups
    of equal counted bits have 7..48 members, and a segment with more groups of two or more than the group list holds."""
    import torch
    rng = np.random.default_rng(78)
    n = 4_000_001
    k = rng.integers(0, 1 << 64, n, dtype=np.uint64)                # (256 first-round buckets of about 15.6 Ki tuples)
    low = np.uint64((1 << 20) - 1)
    for bucket, distinct in ((10, 1024), (11, 8192)):               # about 15 tuples per counted value; about 4600 groups >= 2
        sel = (k >> np.uint64(56)) == np.uint64(bucket)
        c = (k[sel] >> np.uint64(20)) % np.uint64(distinct)
        k[sel] = (np.uint64(bucket) << np.uint64(56)) | (c << np.uint64(30)) | (k[sel] & low)
    tk, tr = dev(k), torch.arange(n, dtype=torch.int64, device="cuda")
    ctx.sort_pairs_u64(tk, tr)
    out, rid = host(tk), tr.cpu().numpy()
    assert (out == np.sort(k)).all() and (k[rid] == out).all() and (np.sort(rid) == np.arange(n)).all()
    st = ctx.stats()
    assert st.get("leaf17_segments", 0) >= 256 and st.get("leaf17_slow_segments", 0) >= 2 and st.get("leaf17_rejected", 0) == 0, st


@pytest.mark.parametrize("regpart", [0, 1])
def test_regpart_modes_u64_keys(ctx, regpart):
    k = O.gen_uniform_u64(3_000_003, seed=52)
    ctx.set_option("regpart", regpart)
    t = dev(k)
    ctx.sort_u64(t)
    assert (host(t) == np.sort(k)).all()


@pytest.mark.parametrize("leaf17", [0, 1])
@pytest.mark.parametrize("kind", ["uniform", "dups", "groups", "short"])
def test_leaf17_u64_keys(ctx, leaf17, kind):
    """u64 keys: the segments two rounds leave (about 2^14 keys) are finished by leaf17_kernel<NoVal>; what it does not take
    (short segments, groups of more than 48 keys equal on the counted bits) goes on to leaf_count_sort_kernel."""
    rng = np.random.default_rng(79)
    n = 4_000_001
    k = rng.integers(0, 1 << 64, n, dtype=np.uint64)
    if kind == "dups":
        k[5::5] = k[4::5][: k[5::5].size]
    elif kind == "groups":   # bucket 10: about 15 keys per counted value; bucket 11: more groups than the list holds; bucket 12: rejected
        low = np.uint64((1 << 20) - 1)
        for bucket, distinct in ((10, 1024), (11, 8192), (12, 8)):
            sel = (k >> np.uint64(56)) == np.uint64(bucket)
            c = (k[sel] >> np.uint64(20)) % np.uint64(distinct)
            k[sel] = (np.uint64(bucket) << np.uint64(56)) | (c << np.uint64(30)) | (k[sel] & low)
    elif kind == "short":    # half of the first-round buckets are split once more into segments of a few hundred keys
        sel = ((k >> np.uint64(56)) & np.uint64(1)) == np.uint64(1)
        k[sel] &= ~(np.uint64(0xFFFF) << np.uint64(32))
    ctx.set_option("leaf17", leaf17)
    try:
        t = dev(k)
        ctx.sort_u64(t)
    finally:
        ctx.set_option("leaf17", 1)
    assert (host(t) == np.sort(k)).all()
    st = ctx.stats()
    assert st.get("leaf17_launches", 0) == (1 if leaf17 else 0), st


def test_last_error_of_the_reference_api_is_empty_after_a_good_sort():
    import ctypes as C
    import inplacemsdradixsort_amd as M
    from inplacemsdradixsort_amd import _lib
    n = 50_000
    k = O.gen_uniform_u64(n, seed=53)
    keys = [M.mamalloc(n * 8).view(np.uint64)]
    rids = [M.mamalloc(n * 8).view(np.uint64)]
    keys[0][:] = k
    rids[0][:] = k
    size = [n]
    M.sort(keys, rids, size, threads=64, numa=1, fudge=1.0)
    L = _lib.load()
    L.msb_64_last_error.restype = C.c_char_p
    assert L.msb_64_last_error() == b"" and (keys[0] == np.sort(k)).all() and size == [n]
