"""GPU parity of the first round's direct block placement.

By default the path is only tried on inputs of 2^22 elements and more whose sampled top-digit
buckets are about equally big; here `direct_min` is lowered so that it runs at sizes numpy
sorts in a moment, and `direct_mode` 2 drops the sample test so that it also meets inputs it
is not meant for (skew, few distinct values, sorted runs) -- slow there, but still exact.
The reference has no counterpart; the oracle is numpy's sort of the same keys (bit-exact).
"""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def dev(a):
    import torch
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.int32 if a.dtype == np.uint32 else np.int64)).cuda()


def host(t):
    a = t.cpu().numpy()
    return a.view(np.uint32) if a.dtype == np.int32 else a.view(np.uint64)


@pytest.fixture()
def dctx(ctx):
    ctx.set_option("direct_min", 1 << 16)
    ctx.set_option("direct_min_parent", 1 << 12)
    ctx.set_option("direct_mode", 1)
    yield ctx
    ctx.set_option("direct_min", 1 << 26)
    ctx.set_option("direct_min_parent", 1 << 17)
    ctx.set_option("direct_mode", 1)


def shapes(rng, n, kind, bits):
    full = (1 << bits) - 1
    dt = np.uint32 if bits == 32 else np.uint64
    if kind == "uniform":
        k = rng.integers(0, full, n, dtype=np.uint64, endpoint=True)
    elif kind == "sorted":
        k = np.sort(rng.integers(0, full, n, dtype=np.uint64, endpoint=True))
    elif kind == "reversed":
        k = np.sort(rng.integers(0, full, n, dtype=np.uint64, endpoint=True))[::-1].copy()
    elif kind == "stride":      # top digit cycles with the position: every stripe sees one bucket at a time
        k = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15 & full)) & np.uint64(full)
    elif kind == "runs":        # long runs of one top digit each (a piece is read long before it can be written)
        top = (np.arange(n, dtype=np.uint64) // np.uint64(4096)) % np.uint64(256)
        k = (top << np.uint64(bits - 8)) | rng.integers(0, (1 << (bits - 8)) - 1, n, dtype=np.uint64)
    elif kind == "lowbits":     # uniform top digit, everything below it constant
        k = rng.integers(0, 255, n, dtype=np.uint64, endpoint=True) << np.uint64(bits - 8)
    elif kind == "zipf":
        k = O.gen_zipf_u32(n, seed=int(rng.integers(1, 1 << 30))).astype(np.uint64)
        if bits == 64:
            k = k << np.uint64(32) | k
    elif kind == "heavy":       # one value takes a third of the input
        k = rng.integers(0, full, n, dtype=np.uint64, endpoint=True)
        k[rng.random(n) < 0.33] = np.uint64(0x5A5A5A5A5A5A5A5A & full)
    else:
        raise ValueError(kind)
    return (k & np.uint64(full)).astype(dt)


KINDS_EVEN = ["uniform", "sorted", "reversed", "stride", "runs", "lowbits"]


@pytest.mark.parametrize("kind", KINDS_EVEN)
@pytest.mark.parametrize("logn", [22, 23, 24])
def test_direct_u32_even_buckets(dctx, kind, logn):
    rng = np.random.default_rng(logn * 100 + len(kind))
    n = (1 << logn) + int(rng.integers(0, 200))
    k = shapes(rng, n, kind, 32)
    t = dev(k)
    dctx.sort_u32(t)
    assert (host(t) == np.sort(k)).all()
    if kind in ("uniform", "stride"):  # the others may need no 8-bit round at all (few varying bits)
        assert dctx.stats().get("direct_rounds", 0) >= 1, "the direct path did not run"


@pytest.mark.parametrize("kind", ["uniform", "zipf", "heavy", "runs", "sorted"])
@pytest.mark.parametrize("typ", ["u32", "u64", "pairs"])
def test_direct_forced_any_distribution(dctx, kind, typ):
    dctx.set_option("direct_mode", 2)
    rng = np.random.default_rng(7 + len(kind) * 13 + len(typ))
    n = (1 << 21) + int(rng.integers(1, 77))
    k = shapes(rng, n, kind, 32 if typ == "u32" else 64)
    t = dev(k)
    if typ == "u32":
        dctx.sort_u32(t)
        assert (host(t) == np.sort(k)).all()
    elif typ == "u64":
        dctx.sort_u64(t)
        assert (host(t) == np.sort(k)).all()
    else:
        r = np.arange(n, dtype=np.uint64)
        tr = dev(r)
        dctx.sort_pairs_u64(t, tr)
        ko, ro = host(t), host(tr)
        assert (ko == np.sort(k)).all()
        assert (k[ro] == ko).all() and (np.sort(ro) == r).all()


def test_direct_second_round_u32(dctx):
    """2^26 keys: the 256 children of the first round (2^18 keys each) are partitioned again, from exact counts
    (their digit is 5 bits wide at this size, which only the forced mode places directly)."""
    dctx.set_option("direct_mode", 2)
    rng = np.random.default_rng(31)
    n = (1 << 26) + 12345
    k = shapes(rng, n, "uniform", 32)
    t = dev(k)
    dctx.sort_u32(t)
    assert dctx.stats().get("direct_rounds", 0) == 2, dctx.stats()
    out = host(t)
    k.sort()
    assert (out == k).all()


@pytest.mark.parametrize("kind", ["uniform", "stride", "runs", "heavy"])
def test_direct_second_round_u64(dctx, kind):
    """u64 keys need more rounds: at 2^24 the second round has 256 parents of 2^16 keys."""
    dctx.set_option("direct_mode", 2)  # the later rounds' digits are narrower than 8 bits at this size
    rng = np.random.default_rng(41 + len(kind))
    n = (1 << 24) + 777
    k = shapes(rng, n, kind, 64)
    t = dev(k)
    dctx.sort_u64(t)
    if kind == "uniform":
        assert dctx.stats().get("direct_rounds", 0) >= 2, dctx.stats()
    out = host(t)
    k.sort()
    assert (out == k).all()


def test_direct_second_round_uneven_children_fall_back(dctx):
    """Top digit uniform, second digit heavily skewed: the exact counts say no, the round streams."""
    rng = np.random.default_rng(51)
    n = 1 << 26
    top = rng.integers(0, 255, n, dtype=np.uint64, endpoint=True) << np.uint64(24)
    low = (rng.random(n) ** 6 * float(1 << 24)).astype(np.uint64)
    k = (top | low).astype(np.uint32)
    t = dev(k)
    dctx.sort_u32(t)
    assert dctx.stats().get("direct_rounds", 0) == 1, dctx.stats()
    out = host(t)
    k.sort()
    assert (out == k).all()


def test_direct_one_digit_partition(dctx):
    """msd_partition_u32 with an 8-bit digit (the pass before the multi-GPU exchange) places directly too."""
    rng = np.random.default_rng(61)
    n = (1 << 22) + 99
    k = shapes(rng, n, "uniform", 32)
    t = dev(k)
    counts = dctx.partition(t, 24, 8).cpu().numpy()
    assert dctx.stats().get("direct_rounds", 0) == 1, dctx.stats()
    out = host(t)
    assert (np.bincount(k >> 24, minlength=256) == counts).all()
    assert ((out >> 24)[1:] >= (out >> 24)[:-1]).all()
    assert (np.sort(out) == np.sort(k)).all()


def test_direct_forced_on_heavy_duplicates_deep_recursion(dctx):
    """Tuples with one value on 40 % of the keys, direct placement forced down to tiny parents: the heavy value's
    segment goes through all eight digits.  (A round whose stripes are smaller than the leftover bound must
    stream: a direct workgroup reads up to a slot per bucket more than its stripe holds -- found by tools/soak.py.)"""
    dctx.set_option("direct_mode", 2)
    dctx.set_option("direct_min", 1 << 14)
    dctx.set_option("direct_min_parent", 1 << 10)
    rng = np.random.default_rng(314)
    n = 5592250
    k = rng.integers(0, (1 << 64) - 1, n, dtype=np.uint64, endpoint=True)
    k[rng.random(n) < 0.4] = k[0]
    r = np.arange(n, dtype=np.uint64)
    t, tr = dev(k), dev(r)
    dctx.sort_pairs_u64(t, tr)
    ko, ro = host(t), host(tr)
    assert (ko == np.sort(k)).all()
    assert (k[ro] == ko).all() and (np.sort(ro) == r).all()


def test_direct_unaligned_start_and_odd_length(dctx):
    rng = np.random.default_rng(99)
    for off, n in [(4, (1 << 22) + 1), (8, (1 << 22) - 63), (12, (1 << 20) + 64)]:
        k = shapes(rng, n + off, "uniform", 32)
        t = dev(k)
        dctx.sort_u32(t[off:])
        out = host(t)
        assert (out[:off] == k[:off]).all() and (out[off:] == np.sort(k[off:])).all()


def test_direct_repeated_runs_identical(dctx):
    rng = np.random.default_rng(5)
    k = shapes(rng, (1 << 22) + 3, "uniform", 32)
    exp = np.sort(k)
    for _ in range(4):
        t = dev(k)
        dctx.sort_u32(t)
        assert (host(t) == exp).all()


def test_direct_off_matches(dctx):
    rng = np.random.default_rng(6)
    k = shapes(rng, 1 << 21, "uniform", 32)
    dctx.set_option("direct_mode", 0)
    t = dev(k)
    dctx.sort_u32(t)
    assert dctx.stats().get("direct_rounds", 0) == 0
    assert (host(t) == np.sort(k)).all()


def test_set_option_rejects_unknown(ctx):
    with pytest.raises(Exception):
        ctx.set_option("no_such_option", 1)
    with pytest.raises(Exception):
        ctx.set_option("direct_mode", 7)
