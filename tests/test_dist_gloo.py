"""world_size-2 and -4 gloo tests (CPU) of the multi-GPU exchange logic: the same
``sort_sharded_u32`` the GPU bench calls, with a numpy stand-in engine (test
infrastructure) in place of the HIP context."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class NumpyEngine:
    """Stand-in with MsdContext's partition / sort_u32 semantics (tests only)."""

    def partition(self, keys, shift, radix_bits, rids=None):
        a = keys.numpy().view(np.uint32 if keys.element_size() == 4 else np.uint64)
        d = ((a >> a.dtype.type(shift)) & a.dtype.type((1 << radix_bits) - 1)).astype(np.int64)
        order = np.argsort(d, kind="stable")
        a[:] = a[order]
        if rids is not None:
            r = rids.numpy()
            r[:] = r[order]
        return torch.from_numpy(np.bincount(d, minlength=1 << radix_bits).astype(np.int64))

    def gather_runs(self, dst, src, src_off, dst_off, lens):
        d, s_ = dst.numpy(), src.numpy()
        for so, do, n in zip(src_off, dst_off, lens):
            d[do:do + n] = s_[so:so + n]

    def sort_segments(self, keys, seg_off, end_bit, rids=None):
        a = keys.numpy().view(np.uint32 if keys.element_size() == 4 else np.uint64)
        for lo, hi in zip(seg_off[:-1], seg_off[1:]):
            if hi > lo:
                hb = a[lo:hi] >> a.dtype.type(end_bit)
                assert int(hb.min()) == int(hb.max())           # a segment's keys agree above end_bit
                a[lo:hi].sort()

    # fine scheme: order by the top bits, bucket boundaries, the counting leaf over the arrived extents
    def sort_top(self, keys, begin_bit, end_bit=None, rids=None):
        a = keys.numpy().view(np.uint32 if keys.element_size() == 4 else np.uint64)
        order = np.argsort(a >> a.dtype.type(begin_bit), kind="stable")   # (keys that agree above begin_bit stay in input order)
        a[:] = a[order]
        if rids is not None:
            r = rids.numpy()
            r[:] = r[order]

    def bucket_bounds(self, keys, shift, nbuckets, first=0):
        a = keys.numpy().view(np.uint32 if keys.element_size() == 4 else np.uint64).astype(np.uint64) >> np.uint64(shift)
        assert (np.diff(a.astype(np.int64)) >= 0).all()
        return torch.from_numpy(np.searchsorted(a, np.arange(first, first + nbuckets + 1, dtype=np.uint64), side="left").astype(np.int64))

    # histogram records (include/msd_radix_hip.h, msd_hist2_pack_u32): 2^16 2-bit fields + [count][value << 16 | copies] x <= 255
    REC = 17408

    def bounds_from_counts16(self, counts):
        return torch.from_numpy(np.concatenate([[0], np.cumsum(counts.numpy())]).astype(np.int64))

    def hist2_pack(self, keys, bounds, rec):
        a, b, r = keys.numpy().view(np.uint32 if keys.element_size() == 4 else np.uint16), bounds.numpy(), rec.numpy()
        nb, over = b.size - 1, 0
        shifts = (2 * np.arange(16, dtype=np.uint32))
        for j in range(nb):
            v = a[b[j]:b[j + 1]].astype(np.uint32) & np.uint32(0xFFFF)
            out = r[j * self.REC:(j + 1) * self.REC]
            out[:] = 0
            if v.size > 65535:
                over = 1
                continue
            cnt = np.bincount(v, minlength=65536).astype(np.uint32)
            words = (np.minimum(cnt, 3).reshape(4096, 16) << shifts).sum(axis=1, dtype=np.uint32)
            out[:16384] = words.view(np.uint8)
            big = np.flatnonzero(cnt >= 3)
            ent = out[16384:].view(np.uint32)
            ent[0] = big.size
            if big.size > 255:
                over = 1
                big = big[:255]
            ent[1:1 + big.size] = (big.astype(np.uint32) << np.uint32(16)) | cnt[big]
        return torch.tensor([over], dtype=torch.int32)

    def _hist2_counts(self, r):
        """the copies of every value a record stands for"""
        shifts = (2 * np.arange(16, dtype=np.uint32))
        cnt = ((r[:16384].view(np.uint32)[:, None] >> shifts) & np.uint32(3)).reshape(65536).astype(np.int64)
        ent = r[16384:].view(np.uint32)
        for e in ent[1:1 + min(int(ent[0]), 255)]:
            cnt[int(e) >> 16] = int(e) & 0xFFFF
        return cnt

    def order_low16(self, keys, out):
        a = keys.numpy().view(np.uint32)
        order = np.argsort(a >> np.uint32(16), kind="stable")
        out.numpy().view(np.uint16)[:a.size] = a[order].astype(np.uint16)
        counts = np.bincount(a >> np.uint32(16), minlength=65536).astype(np.int64)
        a[:] = a[np.argsort(a >> np.uint32(24), kind="stable")]       # (the shard is left ordered by its top 8 bits)
        return torch.from_numpy(counts)

    def order_low16_counts(self, keys):
        a = keys.numpy().view(np.uint32)
        return torch.from_numpy(np.bincount(a >> np.uint32(16), minlength=65536).astype(np.int64))

    def order_low16_scatter(self, keys, out):
        self.order_low16(keys, out)

    def pack_low16(self, keys, out):
        out.numpy().view(np.uint16)[:keys.numel()] = keys.numpy().view(np.uint32).astype(np.uint16)

    def merge_buckets(self, src, counts, src_base, open_bits, first_prefix, dst, n_expected):
        if src.element_size() == 1:              # histogram records: source x's at x * nb * REC; their sum is the sorted bucket
            r, d, c = src.numpy(), dst.numpy().view(np.uint32), counts.numpy()
            nsrc, nb = c.shape
            assert open_bits == 16 and int(c.sum()) == n_expected <= d.size
            at = 0
            for j in range(nb):
                cnt = sum(self._hist2_counts(r[(x * nb + j) * self.REC:(x * nb + j + 1) * self.REC]) for x in range(nsrc))
                assert int(cnt.sum()) == int(c[:, j].sum())
                k = np.repeat(np.arange(65536, dtype=np.uint32), cnt) | np.uint32((first_prefix + j) << 16)
                d[at:at + k.size] = k
                at += k.size
            return
        low16 = src.element_size() == 2          # extents of low halves: the upper half of a key is its bucket's number
        s_, d, c = src.numpy().view(np.uint16 if low16 else np.uint32), dst.numpy().view(np.uint32), counts.numpy()
        nsrc, nb = c.shape
        assert int(c.sum()) == n_expected <= d.size
        pos = [int(b) for b in src_base]
        at = 0
        for j in range(nb):
            parts = []
            for x in range(nsrc):
                parts.append(s_[pos[x]:pos[x] + int(c[x, j])])
                pos[x] += int(c[x, j])
            b = np.concatenate(parts) if parts else np.zeros(0, np.uint32)
            if low16:
                assert open_bits == 16
                b = b.astype(np.uint32) | np.uint32((first_prefix + j) << 16)
            assert ((b >> np.uint32(open_bits)) == first_prefix + j).all()   # every extent holds keys of its bucket only
            d[at:at + b.size] = np.sort(b)
            at += b.size

    def sort_u64(self, keys, end_bit=64):
        a = keys.numpy().view(np.uint64)
        if a.size and end_bit < 64:
            assert int((a >> np.uint64(end_bit)).min()) == int((a >> np.uint64(end_bit)).max())
        a.sort()

    def sort_pairs_u64(self, keys, rids, end_bit=64):
        a, r = keys.numpy().view(np.uint64), rids.numpy()
        if a.size and end_bit < 64:
            assert int((a >> np.uint64(end_bit)).min()) == int((a >> np.uint64(end_bit)).max())
        order = np.argsort(a, kind="stable")
        a[:] = a[order]
        r[:] = r[order]

    def sort_u32(self, keys, end_bit=32):
        a = keys.numpy().view(np.uint32)
        if a.size:
            assert int((a >> np.uint32(end_bit)).min()) == int((a >> np.uint32(end_bit)).max()) if end_bit < 32 else True
        a.sort()

    # splitter service: the oracle's restatement of the reference's front end (src/msb_64.c:1511-1521, :1304-1322, :188-204)
    def sample(self, keys, m, seed=0x5EED0007):
        from oracle import oracle as O
        if keys.element_size() == 4:
            return torch.from_numpy(O.sample_u32(keys.numpy().view(np.uint32), m, seed).view(np.int32).copy())
        return torch.from_numpy(O.sample_u64(keys.numpy().view(np.uint64), m, seed).view(np.int64).copy())

    def splitters(self, sorted_sample, parts):
        from oracle import oracle as O
        if sorted_sample.element_size() == 4:
            d = O.extract_delimiters(sorted_sample.numpy().view(np.uint32).astype(np.uint64), parts)
            return torch.from_numpy(d.astype(np.uint32).view(np.int32).copy())
        return torch.from_numpy(O.extract_delimiters(sorted_sample.numpy().view(np.uint64), parts).view(np.int64).copy())

    def partition_by_splitters(self, keys, delims, parts, rids=None):
        from oracle import oracle as O
        dt = np.uint32 if keys.element_size() == 4 else np.uint64
        a = keys.numpy().view(dt)
        r = O.range_of_u32(a, delims.numpy().view(dt))
        order = np.argsort(r, kind="stable")
        a[:] = a[order]
        if rids is not None:
            v = rids.numpy()
            v[:] = v[order]
        return torch.from_numpy(np.bincount(r, minlength=parts).astype(np.int64))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, kind, q, sampled=False, piece_bytes=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if piece_bytes:   # exchanges in several rounds of small pieces (the product's limit is 512 MiB per pair and round)
        import inplacemsdradixsort_amd.dist as D
        D.A2A_MAX_BYTES = piece_bytes
    from inplacemsdradixsort_amd.dist import sort_sharded_u32, sort_sharded_u32_sampled
    from oracle import oracle as O
    if kind == "uniform":
        k = O.gen_uniform_u32(n, first=rank * n)
    else:
        k = O.gen_zipf_u32(n, first=rank * n)
    keys = torch.from_numpy(k.view(np.int32).copy())
    recv = torch.empty(n * world, dtype=torch.int32)
    if sampled == "work":   # gather bucket-major into a second buffer, segmented local sort
        out = sort_sharded_u32(NumpyEngine(), keys, recv, dist, world, work=torch.empty(n * world, dtype=torch.int32))
    elif sampled == "fine":  # top 16 bits before the exchange, counting leaf over the arrived extents after it
        out = sort_sharded_u32(NumpyEngine(), keys, recv, dist, world, work=torch.empty(n * world, dtype=torch.int32), scheme="fine")
    else:
        fn = sort_sharded_u32_sampled if sampled else sort_sharded_u32
        out = fn(NumpyEngine(), keys, recv, dist, world)
    q.put((rank, out.numpy().view(np.uint32).copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,kind,work", [(2, "uniform", False), (4, "uniform", False), (2, "zipf", False), (8, "uniform", False),
                                             (2, "zipf", True), (4, "uniform", True), (8, "uniform", True),
                                             (2, "uniform", "fine"), (4, "zipf", "fine"), (8, "uniform", "fine")])
def test_sharded_sort_over_gloo(world, kind, work):
    n = 20000 if world < 8 else 6000   # (8 ranks: each owns 32 top-digit buckets, BASELINE config C4's geometry)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, kind, q, work if isinstance(work, str) else ("work" if work else False))) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle import oracle as O
    gen = O.gen_uniform_u32 if kind == "uniform" else O.gen_zipf_u32
    allk = np.concatenate([gen(n, first=r * n) for r in range(world)])
    got = np.concatenate([res[r] for r in range(world)])
    assert (got == O.sort_u32(allk)).all()
    for r in range(world):  # rank r owns top bits == r
        if res[r].size:
            lg = world.bit_length() - 1
            assert ((res[r] >> np.uint32(32 - lg)) == r).all()


@pytest.mark.parametrize("world,kind,how,piece_bytes", [(2, "uniform", False, 4096), (4, "zipf", "fine", 1000), (2, "zipf", "work", 20000),
                                                        (4, "zipf", True, 512), (8, "uniform", "fine", 400)])
def test_exchange_in_pieces_over_gloo(world, kind, how, piece_bytes):
    """The all-to-all in rounds of bounded pieces (RCCL 2.26 moves only half of a >= 2 GiB message): ragged blocks, blocks
    smaller than a piece, piece sizes that do not divide the blocks, every scheme."""
    n = 20000 if world < 8 else 6000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, kind, q, how, piece_bytes)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle import oracle as O
    gen = O.gen_uniform_u32 if kind == "uniform" else O.gen_zipf_u32
    allk = np.concatenate([gen(n, first=r * n) for r in range(world)])
    assert (np.concatenate([res[r] for r in range(world)]) == O.sort_u32(allk)).all()


def _hist_worker(rank, world, port, n, q, pipelined, dups):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import inplacemsdradixsort_amd.dist as D
    from oracle import oracle as O
    D.FINE_HIST_MIN_KEYS = 0                     # (the product asks for 3 * 2^28 keys per rank)
    D.A2A_MAX_BYTES = 1 << 28                    # the records of a pair go in several pieces
    k = O.gen_uniform_u32(n, seed=9, first=rank * n)
    if dups:                                     # rank 1 holds a bucket with 300 values of three copies each: its record overflows,
        if rank == 1:                            # EVERY rank must fall back to the low halves
            k[:900] = (np.uint32(0x12340000) | np.repeat(np.arange(300, dtype=np.uint32), 3))
    keys = torch.from_numpy(k.view(np.int32).copy())
    nrec = (65536 * 17408 + 3) // 4
    recv = [torch.empty(max(nrec, n * world), dtype=torch.int32) for _ in range(2 if pipelined else 1)]
    work = [torch.empty(max(nrec, n * world), dtype=torch.int32) for _ in range(2 if pipelined else 1)]
    eng = NumpyEngine()
    calls = {"hist": 0, "low16": 0}
    orig = eng.merge_buckets

    def spy(src, *a, **kw):
        calls["hist" if src.element_size() == 1 else "low16" if src.element_size() == 2 else "keys"] = calls.get("hist" if src.element_size() == 1 else "low16", 0) + 1
        return orig(src, *a, **kw)

    eng.merge_buckets = spy
    if pipelined:
        sorter = D.ShardedSorter(eng, dist, world, recv, work_bufs=work, scheme="fine")
        sorter.submit(keys)
        out = sorter.collect()
    else:
        out = D.sort_sharded_u32(eng, keys, recv[0], dist, world, work=work[0], scheme="fine")
    q.put((rank, out.numpy().view(np.uint32).copy(), dict(calls), k))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("pipelined,dups", [(False, False), (True, False), (False, True)])
def test_histogram_exchange_over_gloo(pipelined, dups):
    """The fine scheme's histogram form (dist.FINE_HIST): every rank packs its 2^16 buckets into records, equal blocks
    travel, the receiver sums histograms; a record that overflows on ONE rank sends every rank back to the low halves."""
    world, n = 2, 150_000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hist_worker, args=(r, world, port, n, q, pipelined, dups)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, out, calls, k = q.get(timeout=600)
        res[r] = (out, calls, k)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    allk = np.concatenate([res[r][2] for r in range(world)])
    got = np.concatenate([res[r][0] for r in range(world)])
    assert (got == np.sort(allk)).all()
    for r in range(world):
        assert res[r][1].get("hist", 0) == (0 if dups else 1) and res[r][1].get("low16", 0) == (1 if dups else 0), res[r][1]


def test_all_to_all_v_rounds_match_one_call():
    """all_to_all_v's slicing, alone: one rank, a recording stand-in for torch.distributed."""
    import inplacemsdradixsort_amd.dist as D

    class Rec:
        calls = 0

        def all_to_all(self, outs, ins, group=None, async_op=False):
            Rec.calls += 1
            assert all(o.numel() == i.numel() and i.numel() * i.element_size() <= D.A2A_MAX_BYTES for o, i in zip(outs, ins))
            for o, i in zip(outs, ins):
                o.copy_(i)
            return "h"

        def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
            Rec.calls += 1
            out[:sum(output_split_sizes)].copy_(inp[:sum(input_split_sizes)])
            return "h"

    old = D.A2A_MAX_BYTES
    try:
        inp = torch.arange(1000, dtype=torch.int32)
        for lim_b, want_calls in ((1 << 29, 1), (400, 10), (4000, 1), (396, 11)):
            D.A2A_MAX_BYTES, Rec.calls = lim_b, 0
            out = torch.full((1200,), -1, dtype=torch.int32)
            h = D.all_to_all_v(Rec(), out, inp, [1000], D._Splits([1000]), async_op=True)
            assert Rec.calls == want_calls and len(h) == want_calls
            assert (out[:1000] == inp).all() and (out[1000:] == -1).all()
    finally:
        D.A2A_MAX_BYTES = old


def _pipeline_worker(rank, world, port, n, shards, q, scheme=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from inplacemsdradixsort_amd.dist import ShardedSorter
    from oracle import oracle as O
    bufs = [torch.from_numpy(O.gen_uniform_u32(n, seed=100 + s, first=rank * n).view(np.int32).copy()) for s in range(shards)]
    recv = [torch.empty(n * world, dtype=torch.int32) for _ in range(2)]
    work = [torch.empty(n * world, dtype=torch.int32) for _ in range(2)] if (world != 2 or scheme) else None   # (2 ranks: sorted where they arrive)
    sorter = ShardedSorter(NumpyEngine(), dist, world, recv, work_bufs=work, scheme=scheme)
    outs = []
    for s in range(shards):        # the order bench.py uses: submit shard s, then finish shard s-1
        sorter.submit(bufs[s])
        if s:
            outs.append(sorter.collect().numpy().view(np.uint32).copy())
    outs.append(sorter.collect().numpy().view(np.uint32).copy())
    assert sorter.pending() == 0
    q.put((rank, outs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,scheme", [(2, None), (4, None), (8, None), (2, "fine"), (8, "fine")])
def test_pipelined_sharded_sorter_over_gloo(world, scheme):
    """ShardedSorter (exchange of shard s in flight while shard s-1 is sorted) gives every shard's sorted ranges."""
    n, shards = (12000, 4) if world < 8 else (4000, 3)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, n, shards, q, scheme)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle import oracle as O
    for s in range(shards):
        allk = np.concatenate([O.gen_uniform_u32(n, seed=100 + s, first=r * n) for r in range(world)])
        got = np.concatenate([res[r][s] for r in range(world)])
        assert (got == O.sort_u32(allk)).all(), s


@pytest.mark.parametrize("world,kind", [(2, "zipf"), (4, "zipf"), (4, "uniform"), (8, "zipf")])
def test_sampled_splitter_sort_over_gloo(world, kind):
    """Skew path: random sample of the unsorted shards, equi-depth splitters with the reference's duplicate rule, one
    range partition, one exchange, one local sort; ranks stay balanced on Zipf keys."""
    n = 30000 if world < 8 else 8000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, kind, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle import oracle as O
    gen = O.gen_uniform_u32 if kind == "uniform" else O.gen_zipf_u32
    allk = np.concatenate([gen(n, first=r * n) for r in range(world)])
    got = np.concatenate([res[r] for r in range(world)])
    assert (got == O.sort_u32(allk)).all()
    sizes = [res[r].size for r in range(world)]
    assert max(sizes) < (1.35 if world < 8 else 1.6) * n, sizes   # the radix split would put ~75 % of Zipf keys on rank 0


def _overflow_worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from inplacemsdradixsort_amd.dist import ReceiveOverflow, ShardedSorter, sort_sharded_u32
    from oracle import oracle as O
    keys = torch.from_numpy(O.gen_zipf_u32(n, first=rank * n).view(np.int32).copy())   # ~75 % of the keys go to rank 0
    recv = torch.empty(n + n // 8, dtype=torch.int32)                                  # bench.py's 12.5 % slack
    got = []
    for fn in (lambda: sort_sharded_u32(NumpyEngine(), keys.clone(), recv, dist, world),
               lambda: ShardedSorter(NumpyEngine(), dist, world, [recv, recv.clone()]).submit(keys.clone()),
               lambda: sort_sharded_u32(NumpyEngine(), keys.clone(), recv, dist, world, work=recv.clone(), scheme="fine")):
        try:
            fn()
            got.append("no error")
        except ReceiveOverflow as e:
            got.append(str(e))
    dist.barrier()        # every rank is still in step: nobody entered the data exchange alone
    q.put((rank, got))
    dist.destroy_process_group()


def test_receive_overflow_is_raised_on_every_rank():
    """A receive buffer that is too small for ONE rank's range fails the call on ALL ranks, before the data
    exchange (the decision comes from the all-gathered send matrix): no rank is left waiting in the collective."""
    world, n = 4, 20000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overflow_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert len(res[r]) == 3 and all("receive buffer too small on rank(s) 0:" in m for m in res[r]), res[r]


def test_splitters_follow_the_reference_duplicate_rule():
    from inplacemsdradixsort_amd.dist import splitters_equi_depth
    from oracle import oracle as O
    if not O.have_ref():
        pytest.skip("needs oracle/_ref")
    import ctypes as C
    s = np.sort(O.gen_zipf_u32(4000, seed=3).astype(np.uint64))
    parts = 8
    delim = np.zeros(parts, np.uint64)
    delim[parts - 1] = np.uint64(2**64 - 1)  # terminator the reference scans for (src/msb_64.c:1307)
    L = O.ref().lib
    L.extract_delimiters.argtypes = [C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint64)]
    L.extract_delimiters(s.ctypes.data_as(C.POINTER(C.c_uint64)), s.size, delim.ctypes.data_as(C.POINTER(C.c_uint64)))
    assert splitters_equi_depth(s, parts) == delim[:parts - 1].tolist()


def _wide_worker(rank, world, port, n, pairs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from inplacemsdradixsort_amd.dist import sort_sharded_pairs_u64, sort_sharded_u64
    from oracle import oracle as O
    k = O.gen_uniform_u64(n, first=rank * n)
    keys = torch.from_numpy(k.view(np.int64).copy())
    recv_k = torch.empty(2 * n, dtype=torch.int64)
    if pairs:
        rids = torch.from_numpy((k ^ np.uint64(0x5A5A5A5A5A5A5A5A)).view(np.int64).copy())   # rid = f(key): the pairing can be checked
        out_k, out_r = sort_sharded_pairs_u64(NumpyEngine(), keys, rids, recv_k, torch.empty(2 * n, dtype=torch.int64), dist, world)
        q.put((rank, out_k.numpy().view(np.uint64).copy(), out_r.numpy().view(np.uint64).copy()))
    else:
        out = sort_sharded_u64(NumpyEngine(), keys, recv_k, dist, world)
        q.put((rank, out.numpy().view(np.uint64).copy(), None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,pairs", [(2, False), (4, True), (8, True)])
def test_sharded_u64_and_tuple_sort_over_gloo(world, pairs):
    """u64 keys and (u64 key, u64 rid) tuples -- the reference's sort() with one pair of arrays per memory node,
    src/msb_64.c:2261 -- across ranks: global order, every rid still with its key, rank r owns top bits == r."""
    n = 15000 if world < 8 else 5000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wide_worker, args=(r, world, port, n, pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, k, v = q.get(timeout=120)
        res[r] = (k, v)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle import oracle as O
    allk = np.concatenate([O.gen_uniform_u64(n, first=r * n) for r in range(world)])
    got = np.concatenate([res[r][0] for r in range(world)])
    assert (got == np.sort(allk)).all()
    lg = world.bit_length() - 1
    for r in range(world):
        assert ((res[r][0] >> np.uint64(64 - lg)) == r).all()
        if pairs:
            assert (res[r][1] == (res[r][0] ^ np.uint64(0x5A5A5A5A5A5A5A5A))).all()


def _zipf64(n, rank):
    """Zipf-distributed u64 keys: the u32 Zipf ranks spread over 64 bits by squaring (small keys stay heavy)."""
    from oracle import oracle as O
    z = O.gen_zipf_u32(n, first=rank * n).astype(np.uint64)
    return z * z + (z >> np.uint64(3))


def _sampled64_worker(rank, world, port, n, pairs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from inplacemsdradixsort_amd.dist import sort_sharded_pairs_u64_sampled, sort_sharded_u64_sampled
    k = _zipf64(n, rank)
    keys = torch.from_numpy(k.view(np.int64).copy())
    recv_k = torch.empty(2 * n, dtype=torch.int64)
    if pairs:
        rids = torch.from_numpy((k ^ np.uint64(0x5A5A5A5A5A5A5A5A)).view(np.int64).copy())
        out_k, out_r = sort_sharded_pairs_u64_sampled(NumpyEngine(), keys, rids, recv_k, torch.empty(2 * n, dtype=torch.int64), dist, world)
        q.put((rank, out_k.numpy().view(np.uint64).copy(), out_r.numpy().view(np.uint64).copy()))
    else:
        out = sort_sharded_u64_sampled(NumpyEngine(), keys, recv_k, dist, world)
        q.put((rank, out.numpy().view(np.uint64).copy(), None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,pairs", [(2, True), (4, False), (4, True), (8, True)])
def test_sampled_sharded_u64_and_tuple_sort_over_gloo(world, pairs):
    """The reference's skew front end on its own key type across ranks: Zipf-distributed u64 keys (and tuples), sampled
    splitters, one range pass, one exchange; global order, every rid with its key, and no rank above 1.6 n where the radix
    split would put most keys on rank 0."""
    n = 20000 if world < 8 else 8000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sampled64_worker, args=(r, world, port, n, pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, k, v = q.get(timeout=120)
        res[r] = (k, v)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    allk = np.concatenate([_zipf64(n, r) for r in range(world)])
    got = np.concatenate([res[r][0] for r in range(world)])
    assert (got == np.sort(allk)).all()
    assert max(res[r][0].size for r in range(world)) < 1.6 * n, [res[r][0].size for r in range(world)]
    top = world.bit_length() - 1
    assert (allk >> np.uint64(64 - top) == 0).mean() > 0.7            # (the radix split would send > 70 % to rank 0)
    if pairs:
        for r in range(world):
            assert (res[r][1] == (res[r][0] ^ np.uint64(0x5A5A5A5A5A5A5A5A))).all()
