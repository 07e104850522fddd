"""Multi-GPU range partition + one all-to-all.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Rank g
holds a shard of keys in HBM.  Like the reference's NUMA phase -- contiguous key
ranges per node (numa_dest, src/msb_64.c:1596-1607), blocks balanced across nodes
(:1952-1997), then purely local sorting (:2200-2255) -- the distributed sort is:

  1. one in-place top-digit pass on every rank (``engine.partition``): buckets end up
     contiguous in ascending digit order, so the slice for destination rank d
     (digits [d*256/G, (d+1)*256/G)) is already packed for sending;
  2. a G x G count exchange and ONE all-to-all(v) of the keys;
  3. each rank sorts what it received; all its keys share the top log2(G) bits, which
     are passed on as ``end_bit`` (the reference passes bits=58 after its 6-bit split,
     src/msb_64.c:2242).

``engine`` is an :class:`inplacemsdradixsort_amd.MsdContext`; the CPU gloo tests pass a
stand-in with the same three methods to exercise the exchange logic without a GPU.
The receive buffer needs slack over n/G under skew (the reference's ``fudge``).
"""
from __future__ import annotations


def _log2(g: int) -> int:
    b = g.bit_length() - 1
    if g < 1 or (1 << b) != g or g > 256:
        raise ValueError("number of ranks must be a power of two <= 256")
    return b


def sort_sharded_u32(engine, keys, recv, dist, world: int, group=None):
    """Sorts the union of all ranks' ``keys`` (int32 tensors holding u32 bit patterns).
    Returns this rank's sorted range as a view of ``recv``; rank r's range precedes rank r+1's."""
    import torch
    lg = _log2(world)
    if world == 1:
        engine.sort_u32(keys)
        return keys
    counts = engine.partition(keys, 24, 8)                       # int64[256], device of `keys`
    send = counts.view(world, 256 // world).sum(dim=1)           # keys per destination rank
    got = torch.empty_like(send)
    dist.all_to_all_single(got, send, group=group)               # count exchange
    send_l, got_l = send.tolist(), got.tolist()
    total = int(sum(got_l))
    if total > recv.numel():
        raise RuntimeError(f"receive buffer too small: {total} keys for capacity {recv.numel()} "
                           "(raise the slack, the reference's fudge)")
    out = recv[:total]
    dist.all_to_all_single(out, keys, output_split_sizes=got_l, input_split_sizes=send_l, group=group)
    engine.sort_u32(out, end_bit=32 - lg)
    return out


class ShardedSorter:
    """Pipelined form of :func:`sort_sharded_u32` for a stream of shards (``bench.py --gpus N``).

    ``submit(keys)`` runs the top-digit pass on a shard, exchanges the counts and STARTS the
    all-to-all into the next receive buffer; ``collect()`` waits for the oldest exchange and sorts
    what arrived.  Calling ``submit(s)`` before ``collect(s-1)`` lets the exchange of shard s run on
    RCCL's stream (xGMI) while the compute stream sorts shard s-1 -- the exchange is as long as a
    local sort, so hiding it is what weak scaling needs.  ``recv_bufs``: at least two buffers (one
    per exchange in flight plus the one being sorted); the tensor returned by ``collect`` is a view
    of one of them and is overwritten two submissions later.  ``keys`` must stay untouched until the
    matching ``collect`` returns.
    """

    def __init__(self, engine, dist, world: int, recv_bufs, group=None):
        self.engine, self.dist, self.world, self.group = engine, dist, world, group
        self.lg = _log2(world)
        self.recv = list(recv_bufs)
        if world > 1 and len(self.recv) < 2:
            raise ValueError("ShardedSorter needs two receive buffers")
        self._slot = 0
        self._pending = []

    def _all_to_all(self, out, keys, got_l, send_l):
        try:      # torch.distributed: returns a Work whose wait() orders the current stream after it
            return self.dist.all_to_all_single(out, keys, output_split_sizes=got_l, input_split_sizes=send_l,
                                               group=self.group, async_op=True)
        except TypeError:  # stand-ins without async_op (tests)
            self.dist.all_to_all_single(out, keys, output_split_sizes=got_l, input_split_sizes=send_l, group=self.group)
            return None

    def submit(self, keys) -> None:
        import torch
        if self.world == 1:
            self._pending.append((keys, None, keys))
            return
        if len(self._pending) >= len(self.recv):
            raise RuntimeError("collect() before submitting more shards than there are receive buffers")
        counts = self.engine.partition(keys, 24, 8)
        send = counts.view(self.world, 256 // self.world).sum(dim=1)
        got = torch.empty_like(send)
        self.dist.all_to_all_single(got, send, group=self.group)
        send_l, got_l = send.tolist(), got.tolist()
        total = int(sum(got_l))
        recv = self.recv[self._slot]
        self._slot = (self._slot + 1) % len(self.recv)
        if total > recv.numel():
            raise RuntimeError(f"receive buffer too small: {total} keys for capacity {recv.numel()} "
                               "(raise the slack, the reference's fudge)")
        out = recv[:total]
        self._pending.append((out, self._all_to_all(out, keys, got_l, send_l), keys))

    def collect(self):
        out, work, _keys = self._pending.pop(0)
        if work is not None:
            work.wait()
        self.engine.sort_u32(out, end_bit=32 - self.lg)
        return out

    def pending(self) -> int:
        return len(self._pending)


def sort_sharded_u32_sampled(engine, keys, recv, dist, world: int, group=None, sample_per_rank: int = 65536):
    """Skew-robust variant (the reference's own scheme, src/msb_64.c:1511-1564): every rank sorts
    its shard, contributes an equidistant sample of it, all ranks derive the same world-1 equi-depth
    splitters with the reference's duplicate rule (:func:`splitters_equi_depth`), the sorted shard
    is cut at the splitters (range p = keys in (delim[p-1], delim[p]]), ONE all-to-all(v) moves the
    ranges, and each rank sorts what it received (world sorted runs).  A single key value heavier
    than 1/world of the data cannot be split (the reference's limitation too): balance degrades,
    the result stays correct as long as ``recv`` is large enough."""
    import torch
    if world == 1:
        engine.sort_u32(keys)
        return keys
    n = keys.numel()
    engine.sort_u32(keys)
    # ---- sample: equidistant picks of the sorted shard (as unsigned values in int64)
    m = min(sample_per_rank, n)
    idx = (torch.arange(m, device=keys.device, dtype=torch.int64) * n) // max(m, 1)
    mine = keys[idx].to(torch.int64) & 0xFFFFFFFF
    cnt = torch.tensor([m], dtype=torch.int64, device=keys.device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    mmax = int(max(int(c.item()) for c in cnts))
    pad = torch.full((mmax,), -1, dtype=torch.int64, device=keys.device)
    pad[:m] = mine
    allp = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(allp, pad, group=group)
    sample = torch.cat([p[: int(c.item())] for p, c in zip(allp, cnts)]).sort().values.cpu().numpy()
    delim = splitters_equi_depth(sample, world) if len(sample) else [0] * (world - 1)
    # ---- cut the sorted shard: keys <= delim[p] belong to ranges <= p
    ukeys = keys.to(torch.int64) & 0xFFFFFFFF
    d = torch.tensor(delim, dtype=torch.int64, device=keys.device)
    cuts = torch.searchsorted(ukeys, d, right=True)
    bounds = torch.cat([torch.zeros(1, dtype=torch.int64, device=keys.device), cuts,
                        torch.tensor([n], dtype=torch.int64, device=keys.device)])
    send = bounds[1:] - bounds[:-1]
    got = torch.empty_like(send)
    dist.all_to_all_single(got, send, group=group)
    send_l, got_l = send.tolist(), got.tolist()
    total = int(sum(got_l))
    if total > recv.numel():
        raise RuntimeError(f"receive buffer too small: {total} keys for capacity {recv.numel()}")
    out = recv[:total]
    dist.all_to_all_single(out, keys, output_split_sizes=got_l, input_split_sizes=send_l, group=group)
    engine.sort_u32(out)
    return out


def splitters_equi_depth(sorted_sample, parts: int):
    """parts-1 equi-depth delimiters from a sorted sample with the reference's duplicate
    rule (extract_delimiters, src/msb_64.c:1304-1322): if more repetitions of the picked
    value lie after the pick than before it, use value-1 so a heavy value does not
    straddle two ranges.  Range p = keys in (delim[p-1], delim[p]]."""
    n = len(sorted_sample)
    out = []
    pct = n * 1.0 / parts
    for i in range(parts - 1):
        idx = int(pct * (i + 1) - 0.001)
        v = int(sorted_sample[idx])
        start = idx
        while start > 0 and int(sorted_sample[start]) == v:
            start -= 1
        end = idx
        while end < n and int(sorted_sample[end]) == v:
            end += 1
        if idx - start < end - idx and v:
            v -= 1
        out.append(v)
    return out
