"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on
the same seeded inputs -- bit-exact for keys, same key sequence + same multiset
of (key, rid) tuples for pairs (the reference is unstable).
Full BASELINE.json sizes are covered by size-independent properties: sortedness,
sum and xor checksums, idempotence."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def dev(a):
    import torch
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        return torch.from_numpy(a.view(np.int32)).cuda()
    assert a.dtype == np.uint64
    return torch.from_numpy(a.view(np.int64)).cuda()


def host(t):
    a = t.cpu().numpy()
    return a.view(np.uint32) if a.dtype == np.int32 else a.view(np.uint64)


def make_u32(n, kind, seed=1):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return O.gen_uniform_u32(n, seed=0x5EED0001 + seed)
    if kind == "zipf":
        return O.gen_zipf_u32(n, seed=0x5EED0003 + seed)
    if kind == "dup256":
        return rng.integers(0, 256, n, dtype=np.uint32) * np.uint32(0x01010101)
    if kind == "const":
        return np.full(n, 0xDEADBEEF, np.uint32)
    if kind == "sorted":
        return np.sort(O.gen_uniform_u32(n, seed=seed))
    if kind == "reverse":
        return np.sort(O.gen_uniform_u32(n, seed=seed))[::-1].copy()
    if kind == "skew8":
        return (rng.random(n) ** 8 * 2**32).astype(np.uint32)
    if kind == "lowbits":
        return rng.integers(0, 1 << 12, n, dtype=np.uint32)
    raise ValueError(kind)


SIZES = [0, 1, 2, 20, 21, 63, 64, 65, 4097, 6500, 6501, 24576, 24577, 70001, 1 << 20, (1 << 21) + 77]
KINDS = ["uniform", "zipf", "dup256", "const", "sorted", "reverse", "skew8", "lowbits"]


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("kind", KINDS)
def test_sort_u32_equals_oracle(ctx, n, kind):
    k = make_u32(n, kind, seed=n % 97 + 1)
    t = dev(k)
    ctx.sort_u32(t)
    assert (host(t) == O.sort_u32(k)).all()


@pytest.mark.parametrize("n", [1 << 22, (1 << 23) + 12345, 1 << 24, 3 << 23])
@pytest.mark.parametrize("kind", ["uniform", "zipf", "dup256"])
def test_sort_u32_mid_sizes_equal_oracle(ctx, n, kind):
    """Sizes where the round planner mixes digit widths (children of very different sizes)."""
    k = make_u32(n, kind, seed=5)
    t = dev(k)
    ctx.sort_u32(t)
    assert (host(t) == O.sort_u32(k)).all()


@pytest.mark.parametrize("kind", ["lowbits12", "range16", "heavy", "zipf", "twovalues", "tenpercent", "fourvalues"])
@pytest.mark.parametrize("n", [30000, (1 << 20) + 7, 1 << 23])
def test_big_counting_sort_paths(ctx, n, kind):
    """Segments with <= 16 open bits that exceed the LDS leaf: multi-workgroup counting sort
    (small key ranges from the start, and the heavy buckets of skewed inputs after two rounds)."""
    rng = np.random.default_rng(n)
    if kind == "lowbits12":
        k = rng.integers(0, 1 << 12, n, dtype=np.uint32)
    elif kind == "range16":
        k = rng.integers(0, 1 << 16, n, dtype=np.uint32) | np.uint32(0x12340000)
    elif kind == "heavy":      # one value holds half of the keys, the rest is spread over 16 bits
        k = rng.integers(0, 1 << 16, n, dtype=np.uint32)
        k[rng.random(n) < 0.5] = 777
    elif kind == "zipf":
        k = O.gen_zipf_u32(n, seed=n)
    elif kind == "tenpercent":  # too few lanes of a wave share the value for the per-wave register count, enough keys for
        k = rng.integers(0, 1 << 16, n, dtype=np.uint32)  # its packed 16-bit LDS counter to hand over to the histogram in HBM
        k[rng.random(n) < 0.1] = 4242
    elif kind == "fourvalues":  # 40 / 30 / 20 / 10 %: values counted in registers next to values counted in LDS
        k = rng.choice(np.array([3, 70000 & 0xFFFF, 65535, 12], dtype=np.uint32), n, p=[0.4, 0.3, 0.2, 0.1]).astype(np.uint32)
    else:
        k = np.where(rng.random(n) < 0.3, np.uint32(5), np.uint32(0xFFFF0005)).astype(np.uint32)
    t = dev(k)
    ctx.sort_u32(t)
    assert (host(t) == np.sort(k)).all()
    k64 = k.astype(np.uint64) | (np.uint64(0xABCD) << np.uint64(48))
    t = dev(k64)
    ctx.sort_u64(t)
    assert (host(t) == np.sort(k64)).all()


@pytest.mark.parametrize("mid_leaf,hot_share", [(1, 0.3), (0, 0.3), (1, 0.9)])
def test_counting_leaf_overflow_escalates(ctx, mid_leaf, hot_share):
    """A medium segment (above the LDS-sort capacity, ~100 K keys) whose hot value overflows the 8-bit LDS counters of
    the register-resident leaves: the 16-bit-counter leaf (merge_count_kernel, list mode) finishes it -- unless it is
    switched off or the hot value has more copies than a 16-bit counter holds (hot_share 0.9: ~90 K copies); then the
    segment escalates to the multi-workgroup counting sort."""
    rng = np.random.default_rng(42)
    n = 1 << 22
    k = rng.integers(0, 1 << 21, n, dtype=np.uint32)
    hot = rng.random(n) < 0.025                      # one child of the first round gets ~100 K keys ...
    k[hot] = (np.uint32(0x55) << np.uint32(13)) | rng.integers(0, 1 << 13, int(hot.sum()), dtype=np.uint32)
    hotter = hot & (rng.random(n) < hot_share)       # ... a good part of them one single value
    k[hotter] = (np.uint32(0x55) << np.uint32(13)) | np.uint32(77)
    t = dev(k)
    ctx.set_option("mid_leaf", mid_leaf)
    try:
        ctx.sort_u32(t)
    finally:
        ctx.set_option("mid_leaf", 1)
    st = ctx.stats()
    assert (host(t) == np.sort(k)).all()
    assert st.get("count_segments", 0) >= 1, st
    if mid_leaf == 0 or hot_share > 0.8:
        assert st.get("big_count_segments", 0) >= 1, st
    else:
        assert st.get("big_count_segments", 0) == 0, st


def test_sort_u32_config_c1(ctx):
    """BASELINE.json configs[0]: 2^20 uniform u32; digest produced by the reference."""
    import hashlib
    d = json.load(open(os.path.join(G, "golden_c1_digest.json")))
    t = dev(O.gen_uniform_u32(d["n"], seed=d["seed"]))
    ctx.sort_u32(t)
    assert hashlib.sha256(host(t).tobytes()).hexdigest() == d["sha256_sorted"]


@pytest.mark.parametrize("name", ["golden_u32_4096.npz", "golden_zipf_4096.npz"])
def test_golden_u32(ctx, name):
    g = np.load(os.path.join(G, name))
    t = dev(g["keys_in"])
    ctx.sort_u32(t)
    assert (host(t) == g["keys_out"]).all()


def test_golden_pairs(ctx):
    g = np.load(os.path.join(G, "golden_pairs_8192.npz"))
    k, r = dev(g["keys_in"]), dev(g["rids_in"])
    ctx.sort_pairs_u64(k, r)
    ko, ro = host(k), host(r)
    assert (ko == g["keys_out"]).all()
    assert (g["keys_in"][ro] == ko).all() and (np.sort(ro) == g["rids_in"]).all()


@pytest.mark.parametrize("n", [0, 1, 2, 33, 1000, 12288, 12289, 17408, 17409, 100003, 1 << 20, (1 << 22) + 5])
@pytest.mark.parametrize("kind", ["full", "hi32zero", "dup", "const"])
def test_sort_u64_equals_oracle(ctx, n, kind):
    if kind == "full":
        k = O.gen_uniform_u64(n, seed=n)
    elif kind == "hi32zero":
        k = O.gen_uniform_u64(n, seed=n) >> np.uint64(32)
    elif kind == "dup":
        k = (O.gen_uniform_u64(n, seed=n) & np.uint64(0xFF)) * np.uint64(0x0101010101010101)
    else:
        k = np.full(n, 0x0123456789ABCDEF, np.uint64)
    t = dev(k)
    ctx.sort_u64(t)
    assert (host(t) == O.sort_u64(k)).all()


@pytest.mark.parametrize("kind", ["lowvary", "groups", "twolevel"])
@pytest.mark.parametrize("n", [3000, 12288, 200000])
def test_sort_u64_msd_leaf_paths(ctx, n, kind):
    """u64 leaves with many open bits: top-16-bit LDS passes + group fix-up, its long-group
    fallback, and constant leading digits."""
    rng = np.random.default_rng(n)
    if kind == "lowvary":      # only the low 20 bits vary: leading digits of the open range are constant
        k = rng.integers(0, 1 << 20, n, dtype=np.uint64) | np.uint64(0x00AB000000000000)
    elif kind == "groups":     # few distinct top parts, long runs that differ only in low bits (fallback)
        k = (rng.integers(0, 40, n, dtype=np.uint64) << np.uint64(44)) | rng.integers(0, 1 << 30, n, dtype=np.uint64)
    else:                      # random top, tiny random bottom: many short groups
        k = (rng.integers(0, 1 << 12, n, dtype=np.uint64) << np.uint64(50)) | rng.integers(0, 4, n, dtype=np.uint64)
    t = dev(k)
    ctx.sort_u64(t)
    assert (host(t) == np.sort(k)).all()
    r = np.arange(n, dtype=np.uint64)
    tk, tr = dev(k), dev(r)
    ctx.sort_pairs_u64(tk, tr)
    ko, ro = host(tk), host(tr)
    assert (ko == np.sort(k)).all() and (k[ro] == ko).all() and (np.sort(ro) == r).all()


@pytest.mark.parametrize("n", [0, 1, 2, 21, 6144, 6145, 50001, 1 << 19, (1 << 21) + 3])
@pytest.mark.parametrize("kind", ["full", "hi32zero", "dup"])
def test_sort_pairs_parity(ctx, n, kind):
    if kind == "full":
        k = O.gen_uniform_u64(n, seed=n + 5)
    elif kind == "hi32zero":
        k = O.gen_uniform_u64(n, seed=n + 5) >> np.uint64(32)
    else:
        k = O.gen_uniform_u64(n, seed=n + 5) & np.uint64(0x3FF)
    r = np.arange(n, dtype=np.uint64)
    tk, tr = dev(k), dev(r)
    ctx.sort_pairs_u64(tk, tr)
    ko, ro = host(tk), host(tr)
    ek, er = O.sort_pairs_u64(k, r, 64)
    assert (ko == ek).all()                               # same key sequence as the reference path
    assert (k[ro] == ko).all()                            # every rid still travels with its key
    assert (np.sort(ro) == r).all()                       # rids are a permutation: same multiset of tuples
    # rid == key convention of the reference's own check(..., same=1), src/msb_64.c:2461
    tk, tr = dev(k), dev(k)
    ctx.sort_pairs_u64(tk, tr)
    assert (host(tk) == host(tr)).all()
    v, s, x = ctx.check(tk, tr)
    assert v == 0


def test_sort_bits_argument(ctx):
    # keys that differ only in their low 20 bits: the reference's `bits` (src/msb_64.c:1334, 2242)
    k = (O.gen_uniform_u32(300000, seed=9) & np.uint32(0xFFFFF)) | np.uint32(0xABC00000)
    t = dev(k)
    ctx.sort_u32(t, end_bit=20)
    assert (host(t) == np.sort(k)).all()
    k64 = O.gen_uniform_u64(200000, seed=3) >> np.uint64(6)
    t = dev(k64)
    ctx.sort_u64(t, end_bit=58)
    assert (host(t) == O.sort_pairs_u64(k64, k64, 58)[0]).all()


@pytest.mark.parametrize("off", [4, 8, 12, 20, 64])
def test_unaligned_subarray(ctx, off):
    # recursion hands the reference arbitrary bucket starts (virtual_add, src/msb_64.c:795-797);
    # here: any 16-byte aligned sub-array
    k = O.gen_uniform_u32(200000 + off, seed=off)
    t = dev(k)
    ctx.sort_u32(t[off:])
    out = host(t)
    assert (out[:off] == k[:off]).all()
    assert (out[off:] == np.sort(k[off:])).all()


@pytest.mark.parametrize("where", ["first", "last", "vector_tail", "middle"])
@pytest.mark.parametrize("n", [4099, 100003, (1 << 20) + 2])
def test_bit_skip_sees_every_key(ctx, n, where):
    """The exact OR/AND pass behind leading-bit skipping reads 16-byte vectors plus single keys at both ends: one key
    that differs from all others in a leading bit, at the positions that are easy to miss, must still be sorted right."""
    k = O.gen_uniform_u32(n, seed=n) & np.uint32(0xFFFF)       # 16 constant leading bits ...
    at = {"first": 0, "last": n - 1, "vector_tail": (n // 4) * 4 - 1, "middle": n // 2 + 1}[where]
    k[at] |= np.uint32(1 << 30)                                 # ... except in one key
    t = dev(k)
    ctx.sort_u32(t)
    assert (host(t) == np.sort(k)).all()
    k64 = (O.gen_uniform_u32(n, seed=n + 1).astype(np.uint64) & np.uint64(0xFFFFF))
    k64[at] |= np.uint64(1 << 61)
    t = dev(k64)
    ctx.sort_u64(t)
    assert (host(t) == np.sort(k64)).all()


@pytest.mark.parametrize("outlier", [False, True])
@pytest.mark.parametrize("n,opts,path", [((1 << 26) + 3, {}, "histogram"), ((1 << 24) + 7, {"direct_min": 1 << 20}, "own pass")])
def test_sampled_bit_skip_is_verified(n, opts, path, outlier):
    """Big inputs skip leading bits on the word of a sample and check it exactly later -- on the second round's
    histogram pass if that reads every key, else in a pass of its own before the leaves.  One key that differs in a
    skipped bit, at a position the sample does not visit, must make the sort start over on all varying bits."""
    from inplacemsdradixsort_amd import MsdContext
    c = MsdContext(0)
    try:
        for k_, v_ in opts.items():
            c.set_option(k_, v_)
        k = O.gen_uniform_u32(n, seed=n) & np.uint32(0x00FFFFFF)       # 8 constant leading bits
        if outlier:
            k[12345] |= np.uint32(1 << 29)
        t = dev(k)
        c.sort_u32(t)
        st = c.stats()
        assert (host(t) == np.sort(k)).all()
        assert st.get("direct_rounds", 0) >= 1, st
        if outlier:
            assert st.get("bit_skip_restarts", 0) == 1 and st.get("skipped_bits", 0) == 2, (path, st)
        else:
            assert st.get("bit_skip_restarts", 0) == 0 and st.get("skipped_bits", 0) == 8, (path, st)
        k64 = k.astype(np.uint64) | (np.uint64(0x7B) << np.uint64(40))   # u64 keys and tuples: 23 constant leading bits
        if outlier:
            k64[777] ^= np.uint64(1 << 60)
        t = dev(k64)
        r = dev(k64 ^ np.uint64(0xFFFF))
        c.sort_pairs_u64(t, r)
        st = c.stats()
        assert (host(t) == np.sort(k64)).all() and (host(r) == (host(t) ^ np.uint64(0xFFFF))).all()
        assert st.get("bit_skip_restarts", 0) == (1 if outlier else 0), (path, st)
        assert st.get("bit_skip_checked_by_histogram", 0) == (1 if path == "histogram" else 0), (path, st)
    finally:
        c.close()


def test_misaligned_pointer_is_rejected(ctx):
    from inplacemsdradixsort_amd import MsdError
    t = dev(O.gen_uniform_u32(1000))
    with pytest.raises(MsdError):
        ctx.sort_u32(t[1:])


@pytest.mark.parametrize("shift,rb", [(24, 8), (0, 8), (13, 11), (20, 5), (31, 1), (10, 12)])
@pytest.mark.parametrize("n", [0, 1, 1000, (1 << 20) + 3])
def test_histogram_equals_oracle(ctx, n, shift, rb):
    k = O.gen_zipf_u32(n, seed=n + shift)
    h = ctx.histogram(dev(k), shift, rb).cpu().numpy().view(np.uint64)
    assert (h == O.histogram(k, shift, rb)).all()
    k64 = O.gen_uniform_u64(n, seed=n + 1)
    h = ctx.histogram(dev(k64), shift + 32, rb).cpu().numpy().view(np.uint64)
    assert (h == O.histogram(k64, shift + 32, rb)).all()


def test_histogram_golden(ctx):
    g = np.load(os.path.join(G, "golden_hist.npz"))
    t = dev(g["keys"])
    for shift, rb in ((24, 8), (0, 8), (13, 11), (20, 5)):
        assert (ctx.histogram(t, shift, rb).cpu().numpy().view(np.uint64) == g[f"h_s{shift}_r{rb}"]).all()


@pytest.mark.parametrize("n", [1, 255, 2048, 2049, 100000, (1 << 22) + 11])
def test_exclusive_scan_equals_oracle(ctx, n):
    x = (O.gen_uniform_u64(n, seed=n) >> np.uint64(40)).astype(np.uint64)
    out = host(ctx.exclusive_scan(dev(x)))
    assert (out == O.exclusive_scan(x)).all()


@pytest.mark.parametrize("n", [10, 1000, 70000, (1 << 21) + 9])
@pytest.mark.parametrize("shift,rb", [(24, 8), (0, 8), (27, 5), (16, 3)])
def test_partition_pass_equals_oracle(ctx, n, shift, rb):
    """One in-place digit pass: same bucket sizes and same bucket contents
    (as multisets; the pass is unstable) as the reference's histogram+partition."""
    k = O.gen_zipf_u32(n, seed=n) if shift else O.gen_uniform_u32(n, seed=n)
    t = dev(k)
    cnt = ctx.partition(t, shift, rb).cpu().numpy().view(np.uint64)
    ek, _, eh = O.partition(k.astype(np.uint64), k.astype(np.uint64), shift, rb, buffered=True)
    assert (cnt == eh).all()
    out = host(t)
    digits = (out >> np.uint32(shift)) & np.uint32((1 << rb) - 1)
    assert (np.diff(digits.astype(np.int64)) >= 0).all()
    start = 0
    for b, c in enumerate(eh.tolist()):
        assert (np.sort(out[start:start + c]) == np.sort(ek[start:start + c]).astype(np.uint32)).all()
        start += c


def test_partition_pairs_pass(ctx):
    g = np.load(os.path.join(G, "golden_partition.npz"))
    tk, tr = dev(g["keys"]), dev(g["rids"])
    cnt = ctx.partition(tk, 24, 8, rids=tr).cpu().numpy().view(np.uint64)
    assert (cnt == g["buf_hist"]).all()
    ko, ro = host(tk), host(tr)
    assert (g["keys"][ro] == ko).all() and (np.sort(ro) == g["rids"]).all()
    start = 0
    for c in cnt.tolist():
        assert (np.sort(ko[start:start + c]) == np.sort(g["buf_keys"][start:start + c])).all()
        start += c


def test_check_counts_violations(ctx):
    k = np.sort(O.gen_uniform_u32(100000, seed=2))
    v, s, x = ctx.check(dev(k))
    es, ex, eb = O.check([k.astype(np.uint64)], None, False)
    assert (v, s, x) == (0, es, ex)
    k2 = k.copy()
    k2[500], k2[70000] = k2[70000], k2[500]
    v, s, x = ctx.check(dev(k2))
    assert v == O.check([k2.astype(np.uint64)], None, False)[2] and s == es and x == ex


def test_device_generators_match_oracle(ctx):
    import torch
    n = 100000
    t = torch.empty(n, dtype=torch.int32, device="cuda")
    ctx.gen_uniform_u32(t, seed=0x5EED0001, first=12345)
    assert (host(t) == O.gen_uniform_u32(n, seed=0x5EED0001, first=12345)).all()
    t64 = torch.empty(n, dtype=torch.int64, device="cuda")
    ctx.gen_uniform_u64(t64, seed=0x5EED0005, first=7, shift_right=32)
    assert (host(t64) == O.gen_uniform_u64(n, seed=0x5EED0005, first=7) >> np.uint64(32)).all()
    ctx.gen_zipf_u32(t, seed=0x5EED0003)
    z, ez = host(t).astype(np.int64), O.gen_zipf_u32(n, seed=0x5EED0003).astype(np.int64)
    # floating-point pow may differ in the last ulp between host libm and the device: allow rank +-1 on a few keys
    assert (np.abs(z - ez) <= np.maximum(1, ez >> 40)).all() and (z != ez).mean() < 1e-3


def test_duplicates_and_mt19937_generators_match_oracle(ctx):
    """msd_gen_dup_u32 and the MT19937-64 stream (the reference's own RNG, src/rand.c:47-86) are bit-identical to the
    oracle's restatements (the MT one is pinned to the reference's rand.c in tests/test_oracle.py); duplicate-heavy keys sort."""
    import torch
    n = 200001
    t = torch.empty(n, dtype=torch.int32, device="cuda")
    for distinct in (1, 7, 1000, 1 << 20):
        ctx.gen_dup_u32(t, distinct, seed=77, first=5)
        want = O.gen_dup_u32(n, distinct, seed=77, first=5)
        assert (host(t) == want).all()
        ctx.sort_u32(t)
        assert (host(t) == np.sort(want)).all()
    m = torch.empty(5000, dtype=torch.int64, device="cuda")
    for seed, shr in ((5489, 0), (0x5EED0001, 32)):
        ctx.gen_mt19937_64(m, seed, shift_right=shr)
        assert (host(m) == (O.mt19937_64(5000, seed) >> np.uint64(shr))).all()


def test_reference_api_sort_and_check(ctx):
    """sort()/check()/mamalloc of include/msb_64.h on host arrays, two caller arrays ('numa' = 2)."""
    import inplacemsdradixsort_amd as M
    n0, n1 = 150000, 90001
    k = O.gen_uniform_u64(n0 + n1, seed=21)
    # (capacity size * fudge per array, what the reference requires of its caller, src/msb_64.c:1574-1578)
    keys = [M.mamalloc(2 * n0 * 8).view(np.uint64), M.mamalloc(2 * n1 * 8).view(np.uint64)]
    rids = [M.mamalloc(2 * n0 * 8).view(np.uint64), M.mamalloc(2 * n1 * 8).view(np.uint64)]
    keys[0][:n0], keys[1][:n1] = k[:n0], k[n0:]
    rids[0][:n0], rids[1][:n1] = k[:n0], k[n0:]
    size = np.array([n0, n1], dtype=np.uint64)        # a numpy array is written back like a list
    desc, times = M.sort(keys, rids, size, threads=64, numa=2, fudge=2.0)
    assert len(desc) == 11 and desc[10] is None and all(d.endswith(": ") or d.rstrip().endswith(":") for d in desc[:10])
    assert int(size.sum()) == n0 + n1
    cat = np.concatenate([keys[a][:int(size[a])] for a in range(2)])
    assert (cat == np.sort(k)).all() and (np.concatenate([rids[a][:int(size[a])] for a in range(2)]) == cat).all()
    assert M.check(keys, rids, size, numa=2, same=True) == int(k.sum(dtype=np.uint64))
    assert int(times[9]) >= int(times[0])
    with pytest.raises(M.MsdError):                   # size[] is rewritten: a tuple cannot take the result
        M.sort(keys, rids, (n0, n1), threads=64, numa=2, fudge=1.0)
    with pytest.raises(M.MsdError):                   # fudge > 1 needs the room it promises
        M.sort([keys[0][:n0], keys[1][:n1]], [rids[0][:n0], rids[1][:n1]], [n0, n1], threads=64, numa=2, fudge=2.0)


def test_reference_api_rewrites_size_at_key_boundaries():
    """sort() rewrites size[] like the reference (src/msb_64.c:2180; the sum is preserved, :2379-2383): with fudge > 1 a run
    of equal keys that straddles two arrays moves whole into one of them (the reference hands every node whole key
    ranges); times[0..8] add up to times[9]; fudge = 1.0 leaves size[] alone."""
    import inplacemsdradixsort_amd as M
    n0, n1, n2 = 100000, 60000, 30001
    k = (O.gen_uniform_u64(n0 + n1 + n2, seed=33) % np.uint64(50)) * np.uint64(0x0101010101)   # 50 values, runs of ~3800
    for fudge in (2.0, 1.0):
        caps = [int(x * fudge) for x in (n0, n1, n2)]
        keys = [M.mamalloc(c * 8).view(np.uint64) for c in caps]
        rids = [M.mamalloc(c * 8).view(np.uint64) for c in caps]
        parts = [k[:n0], k[n0:n0 + n1], k[n0 + n1:]]
        for a in range(3):
            keys[a][:parts[a].size] = parts[a]
            rids[a][:parts[a].size] = parts[a]
        size = [n0, n1, n2]
        desc, times = M.sort(keys, rids, size, threads=64, numa=3, fudge=fudge)
        assert sum(size) == n0 + n1 + n2 and all(size[a] <= caps[a] for a in range(3))
        cat = np.concatenate([keys[a][:size[a]] for a in range(3)])
        assert (cat == np.sort(k)).all() and (np.concatenate([rids[a][:size[a]] for a in range(3)]) == cat).all()
        if fudge > 1.0:
            assert size != [n0, n1, n2]
            for a in range(2):   # no key value is split between two arrays
                assert keys[a][size[a] - 1] < keys[a + 1][0]
        else:
            assert size == [n0, n1, n2]
        assert M.check(keys, rids, size, numa=3, same=True) == int(k.sum(dtype=np.uint64))
        tm = [int(x) for x in times]
        assert 0.6 * tm[9] <= sum(tm[:9]) <= 1.05 * tm[9] + 2000, tm


def _independent_check(t, rids=None, chunk=1 << 26):
    """(order violations, key sum mod 2^64, key != rid) by chunked torch reductions -- independent of the library's own
    check kernel (the full-size tests relied on msd_check_* alone).  u32 / u64 bit patterns in int32 / int64."""
    import torch
    n, viol, total, mism, prev = t.numel(), 0, 0, 0, None
    for a in range(0, n, chunk):
        c = t[a:a + chunk]
        if t.element_size() == 4:
            u = c.to(torch.int64) & 0xFFFFFFFF
            total += int(u.sum().item())
        else:
            u = c ^ (-(1 << 63))                     # unsigned order as signed order
            total += int(c.sum().item())             # (wraps like the device's sum)
        viol += int((u[1:] < u[:-1]).sum().item())
        if prev is not None and int(u[0].item()) < prev:
            viol += 1
        prev = int(u[-1].item())
        if rids is not None:
            mism += int((c != rids[a:a + chunk]).sum().item())
    return viol, total & ((1 << 64) - 1), mism


@pytest.mark.parametrize("logn,kind", [(26, "uniform"), (26, "zipf"), (30, "uniform"), (30, "zipf"), (32, "uniform")])
def test_full_size_properties(ctx, logn, kind):
    """BASELINE.json configs[1], [2] at full size: sorted, checksums preserved, idempotent."""
    import torch
    n = 1 << logn
    t = torch.empty(n, dtype=torch.int32, device="cuda")
    (ctx.gen_uniform_u32 if kind == "uniform" else ctx.gen_zipf_u32)(t)
    v0, s0, x0 = ctx.check(t)
    assert v0 > 0
    ctx.sort_u32(t)
    v, s, x = ctx.check(t)
    assert (v, s, x) == (0, s0, x0)
    assert _independent_check(t) == (0, s0, 0)          # the same verdict without the library's check kernel
    if logn <= 26:
        first = t.clone()
        ctx.sort_u32(t)  # idempotence
        assert torch.equal(first, t)
        assert (host(t) == np.sort(host(first))).all()
    # spot-check against the oracle on a prefix of the sorted output: the smallest 2^20 keys
    m = 1 << 20
    h = host(t[:m])
    assert (np.diff(h.astype(np.int64)) >= 0).all()
    del t
    torch.cuda.empty_cache()


@pytest.mark.parametrize("shr,name", [(0, "5a: full 64-bit keys"), (32, "5b: upper 32 key bits zero")])
def test_pairs_at_baseline_size(ctx, shr, name):
    """BASELINE.json configs[4] at its stated size, 2^30 (u64 key, u64 rid) tuples = 32 GiB: sorted, key == rid (the
    reference's own check(..., same=1) convention, src/msb_64.c:2461), key sum and xor preserved.  At this size the
    second round of 5a places its blocks directly at default thresholds (parents of 2^22 tuples)."""
    import torch
    n = 1 << 30
    k = torch.empty(n, dtype=torch.int64, device="cuda")
    ctx.gen_uniform_u64(k, shift_right=shr)
    r = k.clone()
    v0, s0, x0 = ctx.check(k)
    assert v0 > 0
    ctx.sort_pairs_u64(k, r)
    st = ctx.stats()
    v, s, x = ctx.check(k, r)  # order + key == rid
    assert (v, s, x) == (0, s0, x0)
    assert _independent_check(k, r) == (0, s0, 0)       # order, sum and key == rid without the library's check kernel
    assert st.get("direct_rounds", 0) >= 2, st
    if shr:
        assert st.get("skipped_bits", 0) == 32, st
    m = 1 << 20   # spot-check against the oracle's order on the smallest 2^20 tuples
    h = host(k[:m])
    assert (np.diff(h.astype(np.float64)) >= 0).all() and (h == host(r[:m])).all()
    del k, r
    torch.cuda.empty_cache()


def test_pairs_full_size_properties(ctx):
    """BASELINE.json configs[4] shape at 2^26 tuples (5a full 64-bit keys, 5b upper half zero)."""
    import torch
    n = 1 << 26
    for shr in (0, 32):
        k = torch.empty(n, dtype=torch.int64, device="cuda")
        ctx.gen_uniform_u64(k, shift_right=shr)
        r = k.clone()
        v0, s0, x0 = ctx.check(k)
        ctx.sort_pairs_u64(k, r)
        v, s, x = ctx.check(k, r)  # order + key == rid
        assert (v, s, x) == (0, s0, x0)
        del k, r
    torch.cuda.empty_cache()
