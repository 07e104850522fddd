"""Multi-GPU range partition + one all-to-all.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Rank g
holds a shard of keys in HBM.  Like the reference's NUMA phase -- contiguous key
ranges per node (numa_dest, src/msb_64.c:1596-1607), blocks balanced across nodes
(:1952-1997), then purely local sorting (:2200-2255) -- the distributed sort is:

  1. one in-place partition pass on every rank: buckets end up contiguous in
     ascending order, so the slice for destination rank d is already packed for sending.
     Uniform keys: the top digit (``engine.partition``), destination = top log2(G) bits.
     Skewed keys: sampled splitters (``sort_sharded_u32_sampled``), destination = range.
  2. the G x G send matrix is all-gathered (every rank learns every rank's total, so a
     receive buffer that is too small fails on ALL ranks, not on one) and ONE
     all-to-all(v) moves the keys;
  3. each rank sorts what it received; with the radix split all its keys share the top
     log2(G) bits, which are passed on as ``end_bit`` (the reference passes bits=58 after
     its 6-bit split, src/msb_64.c:2242).

Large evenly spread shards (>= 2^27 keys per rank, with a second buffer) take the FINE scheme instead: the shard is
ordered by its top 16 bits BEFORE the exchange (``engine.sort_top`` -- two direct-placement rounds, which run at full
speed on a rank's own uniform keys, whereas what ARRIVES after an exchange is run-structured), the 65536 bucket counts
are all-gathered, the one all-to-all moves every rank's range of buckets, and ``engine.merge_buckets`` finishes every
bucket -- ``world`` extents, one per source -- with one counting pass that reads the extents where they arrived and
writes the sorted bucket to its place in the second buffer: no gather pass, no run-structured round, 7.5-8.5 ms of
local work per 2^30-key step at 8, 4 and 2 ranks instead of 10.9-11.5 (tools/multigpu_local_work.py).

u64 keys and (u64 key, u64 rid) tuples -- what the reference's ``sort()`` takes, one pair of arrays per memory node --
shard the same way (``sort_sharded_u64``, ``sort_sharded_pairs_u64``).
``engine`` is an :class:`inplacemsdradixsort_amd.MsdContext`; the CPU gloo tests pass a
stand-in with the same methods to exercise the exchange logic without a GPU.
The receive buffer needs slack over n/G under skew (the reference's ``fudge``).
"""
from __future__ import annotations


class ReceiveOverflow(RuntimeError):
    """Some rank's receive buffer cannot take its range; raised on EVERY rank (the decision is made
    from the all-gathered send matrix, before the data exchange, so no rank is left in a collective)."""


class HostStagedDist:
    """torch.distributed's collectives for CUDA tensors over a CPU-only backend (gloo): every tensor is staged
    through host memory.  For rehearsing the multi-rank code paths where RCCL cannot run -- several ranks sharing
    one GPU (tests/test_gpu_dist.py, ``bench.py --backend gloo``); the real runs use backend "nccl" directly."""

    def __init__(self, dist):
        self._d = dist

    def get_rank(self, group=None):
        return self._d.get_rank(group) if group is not None else self._d.get_rank()

    def barrier(self, group=None):
        self._d.barrier(group=group)

    def all_gather(self, outs, t, group=None):
        import torch
        tmp = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
        self._d.all_gather(tmp, t.cpu(), group=group)
        for o, x in zip(outs, tmp):
            o.copy_(x)

    def all_reduce(self, t, op=None, group=None):
        h = t.cpu()
        self._d.all_reduce(h, op=op if op is not None else self._d.ReduceOp.SUM, group=group)
        t.copy_(h)

    def all_to_all_single(self, output, input, output_split_sizes=None, input_split_sizes=None, group=None):
        import torch
        o = torch.empty(output.shape, dtype=output.dtype)
        self._d.all_to_all_single(o, input.cpu(), output_split_sizes=output_split_sizes,
                                  input_split_sizes=input_split_sizes, group=group)
        output.copy_(o)

    def all_to_all(self, outs, ins, group=None):
        import torch
        o = torch.empty(sum(x.numel() for x in outs), dtype=outs[0].dtype)
        self._d.all_to_all_single(o, torch.cat([x.reshape(-1) for x in ins]).cpu(), output_split_sizes=[x.numel() for x in outs],
                                  input_split_sizes=[x.numel() for x in ins], group=group)
        at = 0
        for x in outs:
            x.copy_(o[at:at + x.numel()])
            at += x.numel()

    def __getattr__(self, name):   # ReduceOp, get_world_size, ...
        return getattr(self._d, name)


# RCCL 2.26 (the build ROCm 7 / torch 2.10 ship) moves only HALF of a single send / receive of >= 2 GiB -- silently: a
# one-rank all_to_all_single of 2^29 int32 comes back with its second GiB untouched (tools/debug/a2a_big.py, measured on
# the GPU box in round 3; the same through ncclSend / ncclRecv directly).  At 2^30 keys per rank a pair's block is 2 GiB at
# 2 ranks, 1 GiB at 4, 512 MiB at 8.  Every exchange therefore goes in pieces of at most this many bytes per pair.
A2A_MAX_BYTES = 1 << 29


class _Splits(list):
    """Split sizes of one rank + the largest block any PAIR of ranks exchanges (every rank must run the same number of
    exchange rounds)."""
    max_block = 0


def all_to_all_v(dist, out, inp, got_l, send_l, group=None, async_op: bool = False):
    """``out`` <- all-to-all(v) of ``inp`` with the given split sizes, in rounds of at most A2A_MAX_BYTES per pair (one
    ``all_to_all_single`` when every block is small).  Returns the list of Work handles (``async_op``) or []."""
    if inp.element_size() == 2:   # (neither RCCL nor gloo moves 16-bit integers: the same bytes as uint8, twice the counts)
        import torch
        send2 = _Splits([2 * int(x) for x in send_l])
        send2.max_block = 2 * getattr(send_l, "max_block", 0)
        return all_to_all_v(dist, out.view(torch.uint8), inp.view(torch.uint8), [2 * int(x) for x in got_l], send2, group, async_op)
    lim = A2A_MAX_BYTES // inp.element_size()
    world = len(send_l)
    big = max(getattr(send_l, "max_block", 0), max(list(send_l) + list(got_l) + [0]))
    if big <= lim:
        if async_op:
            return [dist.all_to_all_single(out, inp, output_split_sizes=list(got_l), input_split_sizes=list(send_l), group=group, async_op=True)]
        dist.all_to_all_single(out, inp, output_split_sizes=list(got_l), input_split_sizes=list(send_l), group=group)
        return []
    soff, goff, a, b = [], [], 0, 0
    for p in range(world):
        soff.append(a)
        goff.append(b)
        a += int(send_l[p])
        b += int(got_l[p])
    handles = []
    for j in range((big + lim - 1) // lim):
        ins = [inp[soff[p] + min(j * lim, int(send_l[p])):soff[p] + min((j + 1) * lim, int(send_l[p]))] for p in range(world)]
        outs = [out[goff[p] + min(j * lim, int(got_l[p])):goff[p] + min((j + 1) * lim, int(got_l[p]))] for p in range(world)]
        if _no_list_all_to_all(dist, group):
            _packed_all_to_all(dist, outs, ins, group)
        elif async_op:
            handles.append(dist.all_to_all(outs, ins, group=group, async_op=True))
        else:
            dist.all_to_all(outs, ins, group=group)
    return handles


def _no_list_all_to_all(dist, group) -> bool:
    """gloo (the CPU tests' backend) has all_to_all_single but not the list form RCCL has."""
    try:
        return str(dist.get_backend(group)) == "gloo" and not isinstance(dist, HostStagedDist)
    except Exception:
        return False


def _packed_all_to_all(dist, outs, ins, group):
    import torch
    o = torch.empty(sum(x.numel() for x in outs), dtype=outs[0].dtype, device=outs[0].device)
    dist.all_to_all_single(o, torch.cat([x.reshape(-1) for x in ins]), output_split_sizes=[x.numel() for x in outs],
                           input_split_sizes=[x.numel() for x in ins], group=group)
    at = 0
    for x in outs:
        x.copy_(o[at:at + x.numel()])
        at += x.numel()


def _log2(g: int) -> int:
    b = g.bit_length() - 1
    if g < 1 or (1 << b) != g or g > 256:
        raise ValueError("number of ranks must be a power of two <= 256")
    return b


def _rank(dist, group=None) -> int:
    return int(dist.get_rank(group)) if group is not None else int(dist.get_rank())


def exchange_counts(dist, send, capacity: int, world: int, group=None):
    """All ranks learn the whole G x G send matrix and every rank's receive capacity
    (one all-gather of G+1 int64 per rank).  Returns (send_list, recv_list) of this rank;
    raises :class:`ReceiveOverflow` on every rank if any rank's total exceeds its capacity."""
    import torch
    row = torch.cat([send.to(torch.int64), torch.tensor([capacity], dtype=torch.int64, device=send.device)])
    rows = [torch.empty_like(row) for _ in range(world)]
    dist.all_gather(rows, row, group=group)
    mat = torch.stack(rows).cpu()                                # [sender, destination | capacity]: one D2H
    totals, caps = mat[:, :world].sum(dim=0), mat[:, world]
    over = [(r, int(totals[r]), int(caps[r])) for r in range(world) if int(totals[r]) > int(caps[r])]
    if over:
        raise ReceiveOverflow("receive buffer too small on rank(s) " +
                              ", ".join(f"{r}: {t} keys for capacity {c}" for r, t, c in over) +
                              " (raise the slack -- the reference's fudge -- or use sort_sharded_u32_sampled)")
    me = _rank(dist, group)
    send_l = _Splits(mat[me, :world].tolist())
    send_l.max_block = int(mat[:, :world].max())
    return send_l, mat[:, me].tolist()


def exchange_bucket_counts(dist, counts, capacity: int, world: int, group=None):
    """Like :func:`exchange_counts`, at the granularity of the 256 top-digit buckets: every rank learns every rank's
    bucket sizes (one all-gather of 257 int64 per rank).  Returns (send_list, recv_list, mine): ``mine[s][j]`` = keys of
    this rank's j-th bucket that source rank s holds -- the lengths of the runs this rank receives, in arrival order."""
    import torch
    row = torch.cat([counts.to(torch.int64), torch.tensor([capacity], dtype=torch.int64, device=counts.device)])
    rows = [torch.empty_like(row) for _ in range(world)]
    dist.all_gather(rows, row, group=group)
    mat = torch.stack(rows).cpu()                                # [sender, bucket | capacity]: one D2H
    per = 256 // world
    to_rank = mat[:, :256].view(world, world, per).sum(dim=2)    # [sender, destination]
    totals, caps = to_rank.sum(dim=0), mat[:, 256]
    over = [(r, int(totals[r]), int(caps[r])) for r in range(world) if int(totals[r]) > int(caps[r])]
    if over:
        raise ReceiveOverflow("receive buffer too small on rank(s) " +
                              ", ".join(f"{r}: {t} keys for capacity {c}" for r, t, c in over) +
                              " (raise the slack -- the reference's fudge -- or use sort_sharded_u32_sampled)")
    me = _rank(dist, group)
    send_l = _Splits(to_rank[me].tolist())
    send_l.max_block = int(to_rank.max())
    return send_l, to_rank[:, me].tolist(), mat[:, me * per:(me + 1) * per].tolist()


FINE_BITS = 16          # the fine scheme orders a shard by its top 16 bits before the exchange ...
FINE_MIN_KEYS = 1 << 27  # ... when a rank holds at least this many keys (buckets of >= 2^11 keys per source)


def use_fine(n_keys: int, world: int, have_work: bool, scheme=None, _force_exchange: bool = False) -> bool:
    """Which scheme ``sort_sharded_u32`` / ``ShardedSorter`` take: ``scheme`` "fine" / "coarse" forces one (tests,
    experiments); by default the fine scheme runs when there is a second buffer, 2..8 ranks and a large shard.  Every
    rank must come to the same answer: shards of one call should be equally long (they are in bench.py)."""
    if scheme == "coarse" or (world < 2 and not _force_exchange) or not have_work:
        return False
    if scheme == "fine":
        return True
    return world <= 8 and n_keys >= FINE_MIN_KEYS


def _takes_async(fn) -> bool:
    import inspect
    try:
        return "async_op" in inspect.signature(fn).parameters
    except (TypeError, ValueError):
        return False


def exchange_fine_counts(dist, counts, capacity: int, world: int, group=None, hist_ok=None, between=None):
    """The fine scheme's count exchange: ``counts`` = this rank's 2^16 bucket sizes (int64, on the device).  One
    all-gather of 2^16 + 1 int64 per rank; the world x world send matrix and the capacities come to the host in ONE
    small copy, the per-bucket counts stay on the device.  Returns (send_list, recv_list, mine): ``mine`` = int64
    [world, 2^16 / world] on the device -- row s: the lengths of source s's extents of this rank's buckets, in arrival
    order.  Raises :class:`ReceiveOverflow` on every rank if any rank's total exceeds its capacity."""
    import torch
    nb = 1 << FINE_BITS
    # (hist_ok: a one-element tensor on the device, non-zero = this rank has its buckets ready as histogram records; the
    # records travel only if EVERY rank has -- the verdict is returned as send_l.use_hist)
    flag = (hist_ok.to(torch.int64).reshape(1) != 0).to(torch.int64) if hist_ok is not None else torch.zeros(1, dtype=torch.int64, device=counts.device)
    row = torch.cat([counts.to(torch.int64), torch.tensor([capacity], dtype=torch.int64, device=counts.device), flag])
    rows = [torch.empty_like(row) for _ in range(world)]
    # (between: device work that does not depend on the other ranks' counts -- the scatter of the low halves -- is queued while
    # the counts travel and the host waits for them; with a backend whose collectives block the host it is queued first)
    if between is not None and _takes_async(dist.all_gather):
        h = dist.all_gather(rows, row, group=group, async_op=True)
        between()
        h.wait()
    else:
        if between is not None:
            between()
        dist.all_gather(rows, row, group=group)
    allc = torch.stack(rows)                                      # [sender, bucket | capacity | records ready], on the device
    nbl = nb // world
    to_rank = allc[:, :nb].view(world, world, nbl).sum(dim=2)     # [sender, destination]
    small = torch.cat([to_rank.reshape(-1), allc[:, nb], allc[:, nb + 1]]).cpu()   # one D2H: world^2 + 2 world numbers
    to_rank_h, caps = small[:world * world].view(world, world), small[world * world:world * world + world]
    all_hist = bool((small[world * world + world:] != 0).all())
    totals = to_rank_h.sum(dim=0)
    over = [(r, int(totals[r]), int(caps[r])) for r in range(world) if int(totals[r]) > int(caps[r])]
    if over:
        raise ReceiveOverflow("receive buffer too small on rank(s) " +
                              ", ".join(f"{r}: {t} keys for capacity {c}" for r, t, c in over) +
                              " (raise the slack -- the reference's fudge -- or use sort_sharded_u32_sampled)")
    me = _rank(dist, group)
    mine = allc[:, me * nbl:(me + 1) * nbl].contiguous()
    send_l = _Splits(to_rank_h[me].tolist())
    send_l.max_block = int(to_rank_h.max())
    send_l.use_hist = all_hist
    return send_l, to_rank_h[:, me].tolist(), mine


# The fine scheme exchanges only the keys' LOW halves: once the shard is ordered by its top 16 bits the upper half of a
# key is its bucket's number, which every receiver knows from the all-gathered counts.  Half the bytes cross xGMI -- one
# link per pair of GPUs: at 2^30 keys per rank a pair exchanges 2 GiB at 2 ranks, 1 GiB at 4, which takes longer than
# all of a rank's local work -- and the counting leaf reads half as much; the price is one more pass over the shard
# (read 4, write 2 bytes per key: engine.pack_low16).  (False: whole keys travel; tests and A/B comparisons)
FINE_LOW16 = True


# Dense buckets travel as HISTOGRAMS of their low halves (engine.hist2_pack): one record of 17408 bytes per source and
# bucket -- 2^16 2-bit counters + the values with three or more copies -- whatever the bucket holds, a quarter of the
# whole keys' bytes at 2^14 keys per source and bucket (2^30 keys per rank), and the receiver adds histograms instead of
# counting keys.  Used when every rank holds at least FINE_HIST_MIN_KEYS keys and no bucket overflows its record (every
# rank reports that with its counts; otherwise the low halves travel).  (False: never; tests and A/B comparisons)
FINE_HIST = True
FINE_HIST_MIN_KEYS = 3 << 28     # 12288 keys per bucket: a record is then 0.7 of the bucket's low halves
# ... and only up to this many ranks.  One GPU's measurements at 2^30 keys per rank (tools/multigpu_local_work.py,
# profiles/r03_multigpu_local_work.jsonl): a rank's local work per step is 8.8 / 8.2 / 7.9 ms at 2 / 4 / 8 ranks with
# records (order_low16 4.85 + pack 1.3 + leaf), 7.6 / 6.9 / 6.7 with low halves (order_low16 4.85 + leaf), while a pair of
# GPUs -- ONE xGMI link, 76.8 GB/s each way at best, 55-60 expected of an all-to-all -- exchanges 4 n / G bytes as whole
# keys, half of that as low halves, 1.06 n / G as records: at 2^30 keys and 58 GB/s 37 / 18.5 / 9.8 ms at 2 ranks, 18.5 / 9.3
# / 4.9 at 4, 9.3 / 4.6 / 2.5 at 8.  Records where the low halves would take longer over the link than the local work they
# hide behind (2 ranks: 9.8 against 18.5 ms per step; 4 ranks: 8.5 against 9.3 -- a tie if the links reach 66 GB/s);
# low halves where the exchange hides either way (8 ranks).
FINE_HIST_MAX_WORLD = 4
HIST2_RECORD_BYTES = 17408


def _bytes_view(buf):
    import torch
    return buf.view(torch.uint8)


def _hist_splits(world: int):
    nbytes = ((1 << FINE_BITS) // world) * HIST2_RECORD_BYTES
    sp = _Splits([nbytes] * world)
    sp.max_block = nbytes
    return sp


def _as_low16(buf):
    """The int16 view of an int32 buffer (twice the elements)."""
    import torch
    return buf.view(torch.int16)


def _fine_counts(engine, keys, rec=None, world: int = 1, send16=None):
    """What a rank does before the fine exchange.  Returns (its 2^16 bucket sizes -- int64, on the device --, the flag "my
    buckets are ready in ``rec`` as histogram records" or None, "the low halves are ready in ``send16``": True, False, or a
    callable that makes them ready and is to be called once -- while the counts are exchanged).
    * histogram records wanted (``rec`` holds 2^16 of them, few ranks, a big shard): ``engine.order_low16`` as below, then
      the records are packed from the low halves (``engine.hist2_pack``); without ``send16`` the shard is ordered in place by
      its top 16 bits (``engine.sort_top``) and the records are packed from the keys;
    * else, with an int16 buffer ``send16``: ``engine.order_low16`` -- one in-place round on the top 8 bits, exact counts,
      the low halves scattered out of place into ``send16``: two passes over the shard less than ordering in place and
      packing afterwards (5.5 against 6.75 ms per 2^30 keys);
    * else the shard is ordered in place and whole keys travel."""
    want_hist = (rec is not None and FINE_HIST and world <= FINE_HIST_MAX_WORLD and keys.numel() >= FINE_HIST_MIN_KEYS
                 and rec.numel() >= (1 << FINE_BITS) * HIST2_RECORD_BYTES)
    if not want_hist and FINE_LOW16 and send16 is not None and send16.numel() >= keys.numel() and hasattr(engine, "order_low16"):
        if hasattr(engine, "order_low16_counts"):   # the scatter runs while the counts are exchanged (exchange_fine_counts' `between`)
            return engine.order_low16_counts(keys), None, (lambda: engine.order_low16_scatter(keys, send16))
        return engine.order_low16(keys, send16), None, True
    if want_hist and FINE_LOW16 and send16 is not None and send16.numel() >= keys.numel() and hasattr(engine, "order_low16"):
        # (the records are packed from the low halves order_low16 has written -- which are then also ready for the case that
        # some rank's records overflow and every rank sends low halves after all)
        counts = engine.order_low16(keys, send16)
        ok = engine.hist2_pack(send16[:keys.numel()], engine.bounds_from_counts16(counts), rec) == 0
        return counts, ok, True
    engine.sort_top(keys, 32 - FINE_BITS)
    b = engine.bucket_bounds(keys, 32 - FINE_BITS, 1 << FINE_BITS)
    ok = (engine.hist2_pack(keys, b, rec) == 0) if want_hist else None
    return b[1:] - b[:-1], ok, False


def _fine_finish(engine, arrived, out, mine, got_l, rank: int, world: int):
    """The counting leaf over what arrived: source s's extents lie back to back from sum(got_l[:s]) on."""
    base, at = [], 0
    for g in got_l:
        base.append(at)
        at += int(g)
    engine.merge_buckets(arrived, mine, base, 32 - FINE_BITS, rank * ((1 << FINE_BITS) // world), out, at)
    return out[:at]


def bucket_major(mine):
    """The runs a rank received (``mine[s][j]``: source-major, as the all-to-all delivers them) and where each belongs
    when every bucket is to be contiguous: (src_off, dst_off, lens) for ``engine.gather_runs`` and the bucket
    boundaries ``seg_off`` for ``engine.sort_segments``."""
    world, per = len(mine), len(mine[0])
    src_off, dst_off, lens = [], [], []
    bucket_start, at = [], 0
    for j in range(per):
        bucket_start.append(at)
        at += sum(int(mine[s][j]) for s in range(world))
    seg_off = bucket_start + [at]
    pos = 0
    fill = list(bucket_start)
    for s in range(world):
        for j in range(per):
            n = int(mine[s][j])
            src_off.append(pos)
            dst_off.append(fill[j])
            lens.append(n)
            pos += n
            fill[j] += n
    return src_off, dst_off, lens, seg_off


def sort_sharded_u32(engine, keys, recv, dist, world: int, group=None, work=None, scheme=None, _force_exchange: bool = False):
    """Sorts the union of all ranks' ``keys`` (int32 tensors holding u32 bit patterns).
    Returns this rank's sorted range; rank r's range precedes rank r+1's.

    ``scheme``: None = choose (:func:`use_fine`), "fine" / "coarse" = force.  The fine scheme (module docstring)
    needs ``work``; its result is a view of ``work``, and ``recv`` must hold what arrives plus up to 3 elements
    (the leaf reads whole 16-byte vectors).

    With a second buffer ``work`` (as large as ``recv``) the local sort does not repeat the top-digit pass: the runs
    that arrived (per source, that source's buckets of this rank's range) are gathered bucket-major into ``work`` in
    one launch and sorted there as segments on the remaining 24 bits -- the result is a view of ``work``.  Without it
    the received keys are sorted in ``recv`` on all ``32 - log2(world)`` bits."""
    lg = _log2(world)
    if world == 1 and not _force_exchange:   # (_force_exchange: tests run the whole exchange over a one-rank process group)
        engine.sort_u32(keys)
        return keys
    if use_fine(keys.numel(), world, work is not None, scheme, _force_exchange):
        # (low halves and records wait in the work buffer -- dead until the leaf writes it --, the records behind the low halves)
        w16 = _as_low16(work)[:keys.numel()] if FINE_LOW16 and 2 * work.numel() >= keys.numel() else None
        rec_at = (2 * keys.numel() + 255) // 256 * 256 if w16 is not None else 0
        rec, s16 = _bytes_view(work)[rec_at:], w16
        if rec.numel() < (1 << FINE_BITS) * HIST2_RECORD_BYTES:   # no room for both: records from the keys, low halves only if needed
            rec = _bytes_view(work)
            if FINE_HIST and world <= FINE_HIST_MAX_WORLD and keys.numel() >= FINE_HIST_MIN_KEYS and rec.numel() >= (1 << FINE_BITS) * HIST2_RECORD_BYTES:
                s16 = None
        counts, hist_ok, packed = _fine_counts(engine, keys, rec, world, s16)
        finish = packed if callable(packed) else None
        send_l, got_l, mine = exchange_fine_counts(dist, counts, min(recv.numel(), work.numel()), world, group, hist_ok, finish)
        packed = True if finish is not None else packed
        m = int(sum(got_l))
        if send_l.use_hist and 4 * recv.numel() >= (1 << FINE_BITS) * HIST2_RECORD_BYTES:
            sp = _hist_splits(world)
            r8 = _bytes_view(recv)
            all_to_all_v(dist, r8[:sum(sp)], rec[:sum(sp)], list(sp), sp, group)
            return _fine_finish(engine, r8, work, mine, got_l, _rank(dist, group), world)
        if w16 is not None:
            # the low halves (packed into the work buffer by now, or packed here from the shard ordered in place) arrive in
            # the receive buffer as int16; the leaf puts the upper halves back
            r16 = _as_low16(recv)
            if not packed:
                engine.pack_low16(keys, w16)
            all_to_all_v(dist, r16[:m], w16, got_l, send_l, group)
            return _fine_finish(engine, r16, work, mine, got_l, _rank(dist, group), world)
        all_to_all_v(dist, recv[:m], keys, got_l, send_l, group)
        return _fine_finish(engine, recv, work, mine, got_l, _rank(dist, group), world)
    counts = engine.partition(keys, 24, 8)                       # int64[256], device of `keys`
    if work is None or world > 4:                                # (8 ranks and more: see ShardedSorter)
        send = counts.view(world, 256 // world).sum(dim=1)       # keys per destination rank
        send_l, got_l = exchange_counts(dist, send, recv.numel(), world, group)
        out = recv[:int(sum(got_l))]
        all_to_all_v(dist, out, keys, got_l, send_l, group)
        engine.sort_u32(out, end_bit=32 - lg)
        return out
    send_l, got_l, mine = exchange_bucket_counts(dist, counts, min(recv.numel(), work.numel()), world, group)
    m = int(sum(got_l))
    all_to_all_v(dist, recv[:m], keys, got_l, send_l, group)
    return _gather_and_sort(engine, recv[:m], work[:m], mine)


def _gather_and_sort(engine, arrived, out, mine):
    src_off, dst_off, lens, seg_off = bucket_major(mine)
    engine.gather_runs(out, arrived, src_off, dst_off, lens)
    engine.sort_segments(out, seg_off, 24)
    return out


def sort_sharded_u64(engine, keys, recv, dist, world: int, group=None):
    """:func:`sort_sharded_u32` for u64 keys (int64 tensors holding the bit patterns): top 8 bits, one all-to-all,
    local sort with ``end_bit = 64 - log2(world)``."""
    lg = _log2(world)
    if world == 1:
        engine.sort_u64(keys)
        return keys
    counts = engine.partition(keys, 56, 8)
    send = counts.view(world, 256 // world).sum(dim=1)
    send_l, got_l = exchange_counts(dist, send, recv.numel(), world, group)
    out = recv[:int(sum(got_l))]
    all_to_all_v(dist, out, keys, got_l, send_l, group)
    engine.sort_u64(out, end_bit=64 - lg)
    return out


def sort_sharded_pairs_u64(engine, keys, rids, recv_keys, recv_rids, dist, world: int, group=None):
    """The reference's own job across devices: (u64 key, u64 rid) tuples held in one (keys, rids) pair of arrays per
    memory node (``sort(keys, rids, size, threads, numa, ...)``, src/msb_64.c:2261; contiguous key ranges per node
    :1596-1607, local sorting :2200-2255) -- here one pair per GPU.  One in-place pass on the top 8 key bits moves keys
    and rids together, the counts are exchanged once, keys and rids travel in two all-to-alls with the same splits,
    and every rank sorts the tuples it received on the remaining bits (``end_bit = 64 - log2(world)``, the
    reference's ``bits`` after its split, :2242).  Returns (keys, rids) views of the receive buffers; rank r's keys
    precede rank r+1's; like the reference's, the sort is unstable."""
    lg = _log2(world)
    if world == 1:
        engine.sort_pairs_u64(keys, rids)
        return keys, rids
    if recv_keys.numel() != recv_rids.numel():
        raise ValueError("receive buffers for keys and rids differ in length")
    counts = engine.partition(keys, 56, 8, rids=rids)
    send = counts.view(world, 256 // world).sum(dim=1)
    send_l, got_l = exchange_counts(dist, send, recv_keys.numel(), world, group)
    m = int(sum(got_l))
    out_k, out_r = recv_keys[:m], recv_rids[:m]
    all_to_all_v(dist, out_k, keys, got_l, send_l, group)
    all_to_all_v(dist, out_r, rids, got_l, send_l, group)
    engine.sort_pairs_u64(out_k, out_r, end_bit=64 - lg)
    return out_k, out_r


class ShardedSorter:
    """Pipelined form of :func:`sort_sharded_u32` for a stream of shards (``bench.py --gpus N``).

    ``submit(keys)`` runs the top-digit pass on a shard, exchanges the counts and STARTS the
    all-to-all into the next receive buffer; ``collect()`` waits for the oldest exchange and sorts
    what arrived.  Calling ``submit(s)`` before ``collect(s-1)`` lets the exchange of shard s run on
    RCCL's stream (xGMI) while the compute stream sorts shard s-1 -- the exchange is as long as a
    local sort, so hiding it is what weak scaling needs.  ``recv_bufs``: at least two buffers (one
    per exchange in flight plus the one being sorted); the tensor returned by ``collect`` is a view
    of one of them and is overwritten ``len(recv_bufs)`` submissions later.  ``keys`` must stay
    untouched until the matching ``collect`` returns.
    """

    def __init__(self, engine, dist, world: int, recv_bufs, group=None, work_bufs=None, scheme=None, _force_exchange: bool = False):
        self.engine, self.dist, self.world, self.group = engine, dist, world, group
        self.scheme = scheme              # None: by shard size (use_fine); "fine" / "coarse": forced
        self._force = _force_exchange     # (tests) one rank goes through the whole exchange
        self.fine_work = list(work_bufs) if work_bufs else []   # the fine scheme uses the work buffers at any rank count
        self.lg = _log2(world)
        self.recv = list(recv_bufs)
        if (world > 1 or _force_exchange) and len(self.recv) < 2:
            raise ValueError("ShardedSorter needs two receive buffers")
        # work buffers: the arrived runs are gathered bucket-major into one of them and sorted there as segments on
        # 24 bits (the local sort does not repeat the top-digit pass); collect() then returns a view of a WORK buffer,
        # overwritten len(work_bufs) collects later.  Without them the keys are sorted where they arrived.
        # (8 ranks and more: the 32 buckets of a rank hold 2^25 keys each at 2^30 keys per rank, one more 8-bit round leaves
        # 2^17-key segments, beyond the fast counting leaf -- sorting the arrived keys on 29 bits measures faster: 8.5 against
        # 1.6 + 9.7 ms, tools/multigpu_local_work.py; 2 and 4 ranks: 9.2 and 8.9 against 13.0 and 12.5 ms)
        self.work = list(work_bufs) if work_bufs and world <= 4 else []
        self._wslot = 0
        self._slot = 0
        self._pending = []
        self._async = None
        self._send16 = []                 # fine scheme: the packed low halves of the shards whose exchange is in flight
        self._send8 = []                  # ... or their buckets' histogram records
        self.last_format = None           # what the last fine exchange moved (reporting)

    def _all_to_all(self, out, keys, got_l, send_l):
        # torch.distributed returns Work handles whose wait() orders the current stream after the exchange; stand-ins without
        # ``async_op`` (HostStagedDist, tests) exchange synchronously.  Decided once from the signature -- a TypeError
        # raised by the real backend (bad split list, dtype) is an error, not a reason to run the collective again.
        if self._async is None:
            import inspect
            try:
                self._async = "async_op" in inspect.signature(self.dist.all_to_all_single).parameters
            except (TypeError, ValueError):
                self._async = False
        return all_to_all_v(self.dist, out, keys, got_l, send_l, self.group, async_op=self._async)

    def submit(self, keys) -> None:
        if self.world == 1 and not self._force:
            self._pending.append((keys, None, None))
            return
        if len(self._pending) >= len(self.recv):
            raise RuntimeError("collect() before submitting more shards than there are receive buffers")
        recv = self.recv[self._slot]
        if use_fine(keys.numel(), self.world, bool(self.fine_work), self.scheme, self._force):
            cap = min(recv.numel(), min(w.numel() for w in self.fine_work))
            slot = self._slot
            rec = None
            if (FINE_HIST and self.world <= FINE_HIST_MAX_WORLD and keys.numel() >= FINE_HIST_MIN_KEYS
                    and 4 * recv.numel() >= (1 << FINE_BITS) * HIST2_RECORD_BYTES):
                while len(self._send8) < len(self.recv):
                    self._send8.append(None)
                if self._send8[slot] is None:
                    import torch
                    self._send8[slot] = torch.empty((1 << FINE_BITS) * HIST2_RECORD_BYTES, dtype=torch.uint8, device=keys.device)
                rec = self._send8[slot]
            send16 = None
            if FINE_LOW16:   # the low halves of the shards whose exchange is in flight: one send buffer per receive buffer
                while len(self._send16) < len(self.recv):
                    self._send16.append(None)
                if self._send16[slot] is None or self._send16[slot].numel() < keys.numel():
                    import torch
                    self._send16[slot] = torch.empty(keys.numel(), dtype=torch.int16, device=keys.device)
                send16 = self._send16[slot][:keys.numel()]
            counts, hist_ok, packed = _fine_counts(self.engine, keys, rec, self.world, send16)
            finish = packed if callable(packed) else None
            send_l, got_l, mine = exchange_fine_counts(self.dist, counts, cap, self.world, self.group, hist_ok, finish)  # raises on all ranks
            packed = True if finish is not None else packed
            self._slot = (self._slot + 1) % len(self.recv)
            self.last_format = "histogram records" if send_l.use_hist else "low halves" if FINE_LOW16 else "whole keys"
            if send_l.use_hist:
                # every rank's buckets are ready as histogram records (module comment at FINE_HIST): equal blocks travel
                sp = _hist_splits(self.world)
                r8 = _bytes_view(recv)
                out = r8[:sum(sp)]
                self._pending.append((out, self._all_to_all(out, rec[:sum(sp)], list(sp), sp), ("fine", r8, mine, got_l)))
                return
            if send16 is not None:
                # only the low halves travel (module comment at FINE_LOW16), received as int16
                if not packed:
                    self.engine.pack_low16(keys, send16)
                r16 = _as_low16(recv)
                out = r16[:int(sum(got_l))]
                self._pending.append((out, self._all_to_all(out, send16, got_l, send_l), ("fine", r16, mine, got_l)))
                return
            out = recv[:int(sum(got_l))]
            self._pending.append((out, self._all_to_all(out, keys, got_l, send_l), ("fine", recv, mine, got_l)))
            return
        counts = self.engine.partition(keys, 24, 8)
        mine = None
        if self.work:
            cap = min(recv.numel(), min(w.numel() for w in self.work))
            send_l, got_l, mine = exchange_bucket_counts(self.dist, counts, cap, self.world, self.group)  # raises on all ranks
        else:
            send = counts.view(self.world, 256 // self.world).sum(dim=1)
            send_l, got_l = exchange_counts(self.dist, send, recv.numel(), self.world, self.group)        # raises on all ranks
        self._slot = (self._slot + 1) % len(self.recv)
        out = recv[:int(sum(got_l))]
        self._pending.append((out, self._all_to_all(out, keys, got_l, send_l), mine))

    def collect(self):
        out, handles, mine = self._pending.pop(0)
        for h in handles or []:
            h.wait()
        if isinstance(mine, tuple):   # fine scheme: the counting leaf reads the extents in the receive buffer, writes a work buffer
            _, recv, counts, got_l = mine
            final = self.fine_work[self._wslot]
            self._wslot = (self._wslot + 1) % len(self.fine_work)
            return _fine_finish(self.engine, recv, final, counts, got_l, _rank(self.dist, self.group), self.world)
        if mine is None:
            if self.world > 1 or self._force:
                self.engine.sort_u32(out, end_bit=32 - self.lg)
            else:
                self.engine.sort_u32(out)
            return out
        final = self.work[self._wslot][:out.numel()]
        self._wslot = (self._wslot + 1) % len(self.work)
        return _gather_and_sort(self.engine, out, final, mine)

    def pending(self) -> int:
        return len(self._pending)


def _sort_sharded_sampled(engine, keys, rids, recv, recv_rids, dist, world: int, group, sample_per_rank: int, seed: int):
    import torch
    pairs = rids is not None
    es = keys.element_size()

    def local_sort(k, r):
        if pairs:
            engine.sort_pairs_u64(k, r)
        elif es == 4:
            engine.sort_u32(k)
        else:
            engine.sort_u64(k)

    if world == 1:
        local_sort(keys, rids)
        return (keys, rids) if pairs else keys
    if pairs and recv.numel() != recv_rids.numel():
        raise ValueError("receive buffers for keys and rids differ in length")
    n = keys.numel()
    m = min(sample_per_rank, n)
    mine = engine.sample(keys, m, seed + 7919 * _rank(dist, group))
    # ranks may hold different numbers of keys: gather the sample sizes, then the (padded) samples
    cnt = torch.tensor([m], dtype=torch.int64, device=keys.device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    sizes = [int(c.item()) for c in cnts]
    mmax = max(sizes)
    pad = torch.zeros(max(mmax, 1), dtype=keys.dtype, device=keys.device)
    pad[:m] = mine
    allp = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(allp, pad, group=group)
    sample = torch.cat([p[:s] for p, s in zip(allp, sizes)]).contiguous()
    if sample.numel():
        (engine.sort_u32 if es == 4 else engine.sort_u64)(sample)
        delim = engine.splitters(sample, world)                         # [world-1] key bit patterns, on the device
    else:
        delim = torch.zeros(world - 1, dtype=keys.dtype, device=keys.device)
    send = engine.partition_by_splitters(keys, delim, world, rids=rids) if pairs else engine.partition_by_splitters(keys, delim, world)
    send_l, got_l = exchange_counts(dist, send, recv.numel(), world, group)   # int64[world]: range sizes, ranges contiguous
    m_out = int(sum(got_l))
    out = recv[:m_out]
    all_to_all_v(dist, out, keys, got_l, send_l, group)
    out_r = None
    if pairs:
        out_r = recv_rids[:m_out]
        all_to_all_v(dist, out_r, rids, got_l, send_l, group)
    local_sort(out, out_r)
    return (out, out_r) if pairs else out


def sort_sharded_u32_sampled(engine, keys, recv, dist, world: int, group=None, sample_per_rank: int = 65536,
                             seed: int = 0x5EED0007):
    """Skew-robust variant, the reference's own scheme (src/msb_64.c:1511-1564) with one range per rank:

      * every rank draws ``sample_per_rank`` keys at random from its UNSORTED shard
        (``engine.sample``: index = mulhi(rand64, n), :1511-1521) -- nothing is sorted before the exchange;
      * the samples are all-gathered and sorted (``engine.sort_u32``, the reference sorts its sample with
        eight passes of partition_keys, :1526-1541);
      * world-1 equi-depth delimiters with the reference's duplicate rule (``engine.splitters``,
        extract_delimiters :1304-1322) -- identical on every rank, the sample being the same;
      * ONE in-place pass cuts the shard into the ranges (delim[p-1], delim[p]]
        (``engine.partition_by_splitters``, the lower-bound range function :188-204);
      * count exchange, ONE all-to-all(v), ONE local sort of what arrived.

    A single key value heavier than 1/world of the data cannot be split (the reference's limitation too):
    balance degrades, the result stays correct as long as ``recv`` is large enough (else
    :class:`ReceiveOverflow` on every rank)."""
    return _sort_sharded_sampled(engine, keys, None, recv, None, dist, world, group, sample_per_rank, seed)


def sort_sharded_u64_sampled(engine, keys, recv, dist, world: int, group=None, sample_per_rank: int = 65536,
                             seed: int = 0x5EED0007):
    """:func:`sort_sharded_u32_sampled` for u64 keys (int64 tensors holding the bit patterns)."""
    return _sort_sharded_sampled(engine, keys, None, recv, None, dist, world, group, sample_per_rank, seed)


def sort_sharded_pairs_u64_sampled(engine, keys, rids, recv_keys, recv_rids, dist, world: int, group=None,
                                   sample_per_rank: int = 65536, seed: int = 0x5EED0007):
    """What the reference actually sorts -- (u64 key, u64 rid) tuples, one (keys, rids) pair of arrays per memory node
    (src/msb_64.c:2261) -- sharded by the reference's own skew front end (sample :1511-1521, extract_delimiters
    :1304-1322, range function :188-204) with one range per GPU: the radix split of :func:`sort_sharded_pairs_u64`
    puts ~75 % of Zipf-distributed keys on one rank, this keeps every rank near n / world.  Keys and rids move
    together in the one in-place range pass and travel in two all-to-alls with the same splits.  Returns (keys, rids)
    views of the receive buffers; rank r's keys precede rank r+1's; unstable, like the reference."""
    return _sort_sharded_sampled(engine, keys, rids, recv_keys, recv_rids, dist, world, group, sample_per_rank, seed)


def splitters_equi_depth(sorted_sample, parts: int):
    """parts-1 equi-depth delimiters from a sorted sample with the reference's duplicate
    rule (extract_delimiters, src/msb_64.c:1304-1322): if more repetitions of the picked
    value lie after the pick than before it, use value-1 so a heavy value does not
    straddle two ranges.  Range p = keys in (delim[p-1], delim[p]].  Host restatement (numpy / lists) of
    what ``msd_splitters_u32`` computes on the device; the tests check both against the compiled reference."""
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
