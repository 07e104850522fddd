"""What a rank of the multi-GPU sort does with the keys it received: msd_gather_runs_* (source-major -> bucket-major,
one launch) and msd_sort_*_segments (the buckets are the parents of the first local round), against numpy."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from inplacemsdradixsort_amd import MsdContext
    c = MsdContext(0)
    yield c
    c.close()


def dev(a):
    import torch
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.int32 if a.dtype == np.uint32 else np.int64)).cuda()


def host(t, dt):
    return t.cpu().numpy().view(dt)


def _segments(rng, n, sizes):
    """Offsets of len(sizes) segments scaled to tile [0, n) (some empty), ascending."""
    w = np.array(sizes, dtype=np.float64)
    cuts = np.concatenate([[0], np.floor(np.cumsum(w) / w.sum() * n).astype(np.int64)])
    cuts[-1] = n
    return [int(x) for x in cuts]


@pytest.mark.parametrize("end_bit", [24, 16, 11])
@pytest.mark.parametrize("n,sizes", [
    (1 << 22, [1] * 32),                                   # the multi-GPU shape: equal buckets, parents of a direct round
    (3_000_000, [5, 0, 1, 40, 0.001, 0.0001, 8, 0, 3]),    # every list at once: parents, leaves of all kinds, tiny and empty segments
    (70_000, [1, 2, 3]),
    (100, [1, 1]),
])
def test_sort_u32_segments(ctx, n, sizes, end_bit):
    rng = np.random.default_rng(n + end_bit)
    off = _segments(rng, n, sizes)
    k = rng.integers(0, 1 << end_bit, n, dtype=np.uint32)
    for i in range(len(off) - 1):                          # all keys of a segment agree above end_bit; later segments need not be larger
        k[off[i]:off[i + 1]] |= np.uint32(((i * 37) % 200) << end_bit) if end_bit <= 24 else np.uint32(0)
    t = dev(k)
    ctx.sort_segments(t, off, end_bit)
    out = host(t, np.uint32)
    for i in range(len(off) - 1):
        assert (out[off[i]:off[i + 1]] == np.sort(k[off[i]:off[i + 1]])).all(), (i, off[i], off[i + 1])


def test_sort_segments_leaves_gaps_alone(ctx):
    rng = np.random.default_rng(5)
    n = 500_000
    k = rng.integers(0, 1 << 20, n, dtype=np.uint32)
    off = [1000, 200_000, 200_000, 450_000]                # [0, 1000) and [450000, n) belong to no segment
    t = dev(k)
    ctx.sort_segments(t, off, 20)
    out = host(t, np.uint32)
    assert (out[:1000] == k[:1000]).all() and (out[450_000:] == k[450_000:]).all()
    assert (out[1000:200_000] == np.sort(k[1000:200_000])).all() and (out[200_000:450_000] == np.sort(k[200_000:450_000])).all()


@pytest.mark.parametrize("pairs", [False, True])
def test_sort_u64_and_tuple_segments(ctx, pairs):
    rng = np.random.default_rng(9 + pairs)
    n, end_bit = 1 << 21, 56
    off = _segments(rng, n, [3, 1, 0, 0.01, 6, 2])
    k = rng.integers(0, 1 << 56, n, dtype=np.uint64)
    for i in range(len(off) - 1):
        k[off[i]:off[i + 1]] |= np.uint64((7 * i + 1) % 256) << np.uint64(56)
    t = dev(k)
    if pairs:
        r = dev(k ^ np.uint64(0x1234))
        ctx.sort_segments(t, off, end_bit, rids=r)
        assert (host(r, np.uint64) == (host(t, np.uint64) ^ np.uint64(0x1234))).all()
    else:
        ctx.sort_segments(t, off, end_bit)
    out = host(t, np.uint64)
    for i in range(len(off) - 1):
        assert (out[off[i]:off[i + 1]] == np.sort(k[off[i]:off[i + 1]])).all(), i


def test_segment_offsets_are_checked(ctx):
    from inplacemsdradixsort_amd import MsdError
    import torch
    t = torch.zeros(1000, dtype=torch.int32, device="cuda")
    with pytest.raises(MsdError):
        ctx.sort_segments(t, [0, 600, 500], 32)
    with pytest.raises(MsdError):
        ctx.sort_segments(t, [0, 500, 2000], 32)


@pytest.mark.parametrize("dt", [np.uint32, np.uint64])
def test_gather_runs(ctx, dt):
    """Runs of any length and alignment, source-major -> bucket-major as after the all-to-all (8 sources x 32 buckets)."""
    import torch
    rng = np.random.default_rng(3)
    sources, buckets = 8, 32
    lens = rng.integers(0, 40_000, (sources, buckets))
    lens[2, 5] = 0
    lens[0, 0] = 1
    lens[7, 31] = 300_001
    n = int(lens.sum())
    src = rng.integers(0, 1 << 31, n).astype(dt)
    src_off = np.concatenate([[0], np.cumsum(lens.reshape(-1))[:-1]]).reshape(sources, buckets)      # source-major
    dst_off = np.concatenate([[0], np.cumsum(lens.T.reshape(-1))[:-1]]).reshape(buckets, sources).T   # bucket-major
    d = torch.full((n + 3,), -1, dtype=torch.int32 if dt == np.uint32 else torch.int64, device="cuda")
    ctx.gather_runs(d[3:] if dt == np.uint32 else d[1:-2], dev(src), src_off.reshape(-1).tolist(), dst_off.reshape(-1).tolist(),
                    lens.reshape(-1).tolist())
    got = host(d[3:] if dt == np.uint32 else d[1:-2], dt)[:n]
    want = np.empty(n, dtype=dt)
    for s in range(sources):
        for b in range(buckets):
            want[dst_off[s, b]:dst_off[s, b] + lens[s, b]] = src[src_off[s, b]:src_off[s, b] + lens[s, b]]
    assert (got == want).all()
