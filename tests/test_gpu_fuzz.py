"""Randomised GPU parity: many small-to-medium random cases (size, key type, bit pattern, skew,
alignment offset) against numpy's sort, plus repeated runs of one input (the block permutation
claims slots with atomics, so its internal order differs from run to run -- the output must not)."""
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


def random_keys(rng, n, bits):
    dt = np.uint32 if bits == 32 else np.uint64
    kind = rng.integers(0, 7)
    full = (1 << bits) - 1
    if kind == 0:      # uniform over all bits
        k = rng.integers(0, full, n, dtype=np.uint64, endpoint=True)
    elif kind == 1:    # random AND-mask: arbitrary constant / varying bit positions
        m = int(rng.integers(0, full, dtype=np.uint64, endpoint=True))
        k = rng.integers(0, full, n, dtype=np.uint64, endpoint=True) & np.uint64(m)
    elif kind == 2:    # few distinct values
        vals = rng.integers(0, full, int(rng.integers(1, 300)), dtype=np.uint64, endpoint=True)
        k = vals[rng.integers(0, len(vals), n)]
    elif kind == 3:    # power-law skew
        k = (rng.random(n) ** float(rng.integers(2, 12)) * full).astype(np.uint64)
    elif kind == 4:    # narrow window high up
        base = int(rng.integers(0, full >> 1, dtype=np.uint64))
        k = np.uint64(base) + rng.integers(0, 1 << int(rng.integers(1, 20)), n, dtype=np.uint64)
    elif kind == 5:    # already sorted / reversed
        k = np.sort(rng.integers(0, full, n, dtype=np.uint64, endpoint=True))
        if rng.integers(0, 2):
            k = k[::-1].copy()
    else:              # one heavy value plus noise
        k = rng.integers(0, full, n, dtype=np.uint64, endpoint=True)
        k[rng.random(n) < rng.random()] = np.uint64(int(rng.integers(0, full, dtype=np.uint64, endpoint=True)))
    return (k & np.uint64(full)).astype(dt)


@pytest.mark.parametrize("seed", range(40))
def test_random_cases(ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(2 ** rng.uniform(0, 22.5)) + int(rng.integers(0, 3))
    typ = ["u32", "u64", "pairs"][seed % 3]
    off = int(rng.integers(0, 4)) * (4 if typ == "u32" else 2)          # 16-byte aligned sub-array start
    k = random_keys(rng, n + off, 32 if typ == "u32" else 64)
    t = dev(k)
    if typ == "u32":
        ctx.sort_u32(t[off:])
        out = host(t)
        assert (out[:off] == k[:off]).all() and (out[off:] == np.sort(k[off:])).all()
    elif typ == "u64":
        ctx.sort_u64(t[off:])
        out = host(t)
        assert (out[:off] == k[:off]).all() and (out[off:] == np.sort(k[off:])).all()
    else:
        r = np.arange(n + off, dtype=np.uint64)
        tr = dev(r)
        ctx.sort_pairs_u64(t[off:], tr[off:])
        ko, ro = host(t), host(tr)
        assert (ko[:off] == k[:off]).all() and (ko[off:] == np.sort(k[off:])).all()
        assert (k[ro[off:]] == ko[off:]).all() and (np.sort(ro[off:]) == r[off:]).all()


def test_repeated_runs_give_identical_keys(ctx):
    k = O.gen_zipf_u32((1 << 22) + 5, seed=77)
    exp = np.sort(k)
    for _ in range(6):
        t = dev(k)
        ctx.sort_u32(t)
        assert (host(t) == exp).all()


def test_non_default_stream(ctx):
    import torch
    k = O.gen_uniform_u32(3_000_000, seed=5)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ctx.use_torch_stream()
        t = dev(k)
        ctx.sort_u32(t)
        out = t.clone()
    s.synchronize()
    torch.cuda.synchronize()
    ctx.use_torch_stream()  # back to the default stream for the other tests
    assert (host(out) == np.sort(k)).all()
