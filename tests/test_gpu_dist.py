"""Multi-rank sort with the REAL HIP engine: two (and four) processes share cuda:0 and run
inplacemsdradixsort_amd.dist.sort_sharded_u32; the exchange goes through a gloo-via-CPU stand-in
for torch.distributed (RCCL cannot run several ranks on one device; the driver's N-GPU bench uses
the same function with backend "nccl")."""
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
pytestmark = pytest.mark.gpu


class GlooViaCpu:
    """all_to_all_single / all_gather for CUDA tensors over the gloo backend (copies through host memory)."""

    get_rank = staticmethod(dist.get_rank)

    @staticmethod
    def all_gather(outs, t, group=None):
        tmp = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
        dist.all_gather(tmp, t.cpu(), group=group)
        for o, x in zip(outs, tmp):
            o.copy_(x)

    @staticmethod
    def all_to_all_single(output, input, output_split_sizes=None, input_split_sizes=None, group=None):
        o = torch.empty(output.shape, dtype=output.dtype)
        dist.all_to_all_single(o, input.cpu(), output_split_sizes=output_split_sizes,
                               input_split_sizes=input_split_sizes, group=group)
        output.copy_(o)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, kind, q, sampled=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from inplacemsdradixsort_amd import MsdContext
    from inplacemsdradixsort_amd.dist import sort_sharded_u32, sort_sharded_u32_sampled
    ctx = MsdContext(0)
    keys = torch.empty(n, dtype=torch.int32, device="cuda:0")
    (ctx.gen_uniform_u32 if kind == "uniform" else ctx.gen_zipf_u32)(keys, first=rank * n)
    v0, s0, x0 = ctx.check(keys)
    recv = torch.empty(n * world, dtype=torch.int32, device="cuda:0")
    if sampled == "work":   # runs gathered bucket-major into a second buffer, segmented local sort (round 2's N > 1 path)
        out = sort_sharded_u32(ctx, keys, recv, GlooViaCpu, world, work=torch.empty(n * world, dtype=torch.int32, device="cuda:0"), scheme="coarse")
    elif sampled == "fine":  # top 16 bits before the exchange, counting leaf over the arrived extents (bench.py's N > 1 path)
        out = sort_sharded_u32(ctx, keys, recv, GlooViaCpu, world, work=torch.empty(n * world, dtype=torch.int32, device="cuda:0"), scheme="fine")
    else:
        out = (sort_sharded_u32_sampled if sampled else sort_sharded_u32)(ctx, keys, recv, GlooViaCpu, world)
    v, s, x = ctx.check(out)
    lo = int(out[0].item()) & 0xFFFFFFFF if out.numel() else -1
    hi = int(out[-1].item()) & 0xFFFFFFFF if out.numel() else -1
    q.put((rank, out.numel(), v, s0, x0, s, x, lo, hi, out.cpu().numpy().view(np.uint32).copy()))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


def _pipeline_worker(rank, world, port, n, shards, q, scheme=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from inplacemsdradixsort_amd import MsdContext
    from inplacemsdradixsort_amd.dist import ShardedSorter
    ctx = MsdContext(0)
    ctx.set_option("direct_min", 1 << 16)   # the pre-exchange pass and the local sorts place directly at this size
    bufs, sums = [], []
    for s in range(shards):
        t = torch.empty(n, dtype=torch.int32, device="cuda:0")
        ctx.gen_uniform_u32(t, seed=500 + s, first=rank * n)
        bufs.append(t)
        sums.append(ctx.check(t)[1:])
    recv = [torch.empty(n * world, dtype=torch.int32, device="cuda:0") for _ in range(2)]
    work = [torch.empty(n * world, dtype=torch.int32, device="cuda:0") for _ in range(2)]
    sorter = ShardedSorter(ctx, GlooViaCpu, world, recv, work_bufs=work, scheme=scheme)
    res = []

    def take(out):
        v, s_, x_ = ctx.check(out)
        lo = int(out[0].item()) & 0xFFFFFFFF if out.numel() else -1
        hi = int(out[-1].item()) & 0xFFFFFFFF if out.numel() else -1
        res.append((out.numel(), v, s_, x_, lo, hi, out.cpu().numpy().view(np.uint32).copy()))

    for s in range(shards):
        sorter.submit(bufs[s])
        if s:
            take(sorter.collect())
    take(sorter.collect())
    q.put((rank, res, sums))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


@pytest.mark.parametrize("scheme", ["coarse", "fine"])
def test_pipelined_sharded_sorter_real_engine(scheme):
    """bench.py's N > 1 loop (ShardedSorter) with the HIP engine: 2 ranks on one device, 3 shards in a row; the
    concatenated outputs equal the sorted union of the regenerated inputs."""
    world, n, shards = 2, 1 << 22, 3
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_pipeline_worker, args=(r, world, port, n, shards, q, scheme)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for s in range(shards):
        per_rank = [got[r][1][s] for r in range(world)]
        assert sum(x[0] for x in per_rank) == n * world
        assert all(x[1] == 0 for x in per_rank)                       # every rank's range is sorted
        in_sum = sum(got[r][2][s][0] for r in range(world)) & (2 ** 64 - 1)
        in_xor = 0
        for r in range(world):
            in_xor ^= got[r][2][s][1]
        out_xor = 0
        for x in per_rank:
            out_xor ^= x[3]
        assert sum(x[2] for x in per_rank) & (2 ** 64 - 1) == in_sum and out_xor == in_xor
        assert per_rank[0][5] < per_rank[1][4]                        # rank 0's range precedes rank 1's
        from oracle import oracle as O
        allk = np.concatenate([O.gen_uniform_u32(n, seed=500 + s, first=r * n) for r in range(world)])
        assert (np.concatenate([x[6] for x in per_rank]) == np.sort(allk)).all()


@pytest.mark.parametrize("world,kind,n,work", [(2, "uniform", 1 << 22, False), (4, "uniform", 1 << 20, False), (2, "zipf", 1 << 21, False),
                                               (2, "uniform", 1 << 22, True), (4, "zipf", 1 << 20, True),
                                               (2, "uniform", 1 << 22, "fine"), (4, "uniform", 1 << 21, "fine"), (4, "zipf", 1 << 20, "fine")])
def test_sharded_sort_real_engine(world, kind, n, work):
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, world, port, n, kind, q, work if isinstance(work, str) else ("work" if work else False))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    total = sum(r[1] for r in res)
    assert total == n * world
    assert all(r[2] == 0 for r in res)                                 # every rank's range is sorted
    M = 1 << 64
    assert sum(r[3] for r in res) % M == sum(r[5] for r in res) % M   # key sum preserved across the exchange
    x_in = x_out = 0
    for r in res:
        x_in ^= r[4]
        x_out ^= r[6]
    assert x_in == x_out
    lg = world.bit_length() - 1
    prev_hi = -1
    for r in res:                                                      # rank r owns top bits == r, ranges ascend
        if r[1]:
            assert (r[7] >> (32 - lg)) == r[0] and (r[8] >> (32 - lg)) == r[0] and r[7] > prev_hi
            prev_hi = r[8]
    # the exact result: the ranks' outputs, concatenated, are the sorted union of the (regenerated) inputs
    from oracle import oracle as O
    gen = O.gen_uniform_u32 if kind == "uniform" else O.gen_zipf_u32
    allk = np.concatenate([gen(n, first=r * n) for r in range(world)])
    assert (np.concatenate([r[9] for r in res]) == np.sort(allk)).all()


@pytest.mark.parametrize("world,kind,n", [(2, "zipf", 1 << 21), (4, "zipf", 1 << 20)])
def test_sampled_splitter_sort_real_engine(world, kind, n):
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, world, port, n, kind, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sum(r[1] for r in res) == n * world and all(r[2] == 0 for r in res)
    M = 1 << 64
    assert sum(r[3] for r in res) % M == sum(r[5] for r in res) % M
    prev_hi = -1
    for r in res:
        if r[1]:
            assert r[7] > prev_hi                                      # ranges (delim[p-1], delim[p]] do not share a value
            prev_hi = r[8]
    assert max(r[1] for r in res) < 1.35 * n      # balanced although 75 % of the keys share the top byte
    from oracle import oracle as O
    allk = np.concatenate([O.gen_zipf_u32(n, first=r * n) for r in range(world)])
    assert (np.concatenate([r[9] for r in res]) == np.sort(allk)).all()


def _pairs_worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from inplacemsdradixsort_amd import MsdContext
    from inplacemsdradixsort_amd.dist import sort_sharded_pairs_u64
    ctx = MsdContext(0)
    keys = torch.empty(n, dtype=torch.int64, device="cuda:0")
    ctx.gen_uniform_u64(keys, first=rank * n)
    rids = keys.clone()                     # the reference's own check convention: rid == key (src/msb_64.c:2461)
    v0, s0, x0 = ctx.check(keys)
    rk = torch.empty(2 * n, dtype=torch.int64, device="cuda:0")
    rr = torch.empty(2 * n, dtype=torch.int64, device="cuda:0")
    out_k, out_r = sort_sharded_pairs_u64(ctx, keys, rids, rk, rr, GlooViaCpu, world)
    v, s, x = ctx.check(out_k, out_r)       # order + key == rid
    M = (1 << 64) - 1
    lo = int(out_k[0].item()) & M if out_k.numel() else -1
    hi = int(out_k[-1].item()) & M if out_k.numel() else -1
    q.put((rank, out_k.numel(), v, s0, x0, s, x, lo, hi, out_k.cpu().numpy().view(np.uint64).copy()))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


@pytest.mark.parametrize("world,n", [(2, 1 << 21), (4, 1 << 19)])
def test_sharded_tuple_sort_real_engine(world, n):
    """The reference's sort() across devices: (u64 key, u64 rid) tuples, one (keys, rids) pair per rank; sorted per
    rank, key == rid everywhere, key sum and xor preserved, rank r owns the keys whose top bits are r."""
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_pairs_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sum(r[1] for r in res) == n * world and all(r[2] == 0 for r in res)
    M = 1 << 64
    assert sum(r[3] for r in res) % M == sum(r[5] for r in res) % M
    x_in = x_out = 0
    for r in res:
        x_in ^= r[4]
        x_out ^= r[6]
    assert x_in == x_out
    lg = world.bit_length() - 1
    for r in res:
        if r[1]:
            assert (r[7] >> (64 - lg)) == r[0] and (r[8] >> (64 - lg)) == r[0]
    from oracle import oracle as O
    allk = np.concatenate([O.gen_uniform_u64(n, first=r * n) for r in range(world)])
    assert (np.concatenate([r[9] for r in res]) == np.sort(allk)).all()   # the exact key sequence (rid == key was checked on the device)
