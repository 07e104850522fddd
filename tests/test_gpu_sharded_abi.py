"""The multi-GPU entry points behind the C ABI (include/msd_sharded_hip.h) as far as ONE GPU allows: a single rank
without a communicator, a single rank WITH a real RCCL communicator of one rank (ncclCommInitRank: the all-gather and
the grouped send/receive to itself run through RCCL), and the one-call form msd_sort_u32_multi on one device.  More
than one rank per device is not possible with RCCL; the N > 1 exchange logic is covered by the gloo tests of dist.py."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def dev(a):
    import torch
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.int32 if a.dtype == np.uint32 else np.int64)).cuda()


def host(t, dt):
    return t.cpu().numpy().view(dt)


class NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


@pytest.fixture(scope="module")
def comm1():
    """A real RCCL communicator with one rank on device 0."""
    import torch  # noqa: F401  (loads the process's RCCL)
    from inplacemsdradixsort_amd import _lib
    _lib.load_rccl()
    R = C.CDLL("librccl.so.1")
    R.ncclGetUniqueId.argtypes = [C.POINTER(NcclUniqueId)]
    R.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, NcclUniqueId, C.c_int]
    R.ncclCommDestroy.argtypes = [C.c_void_p]
    uid = NcclUniqueId()
    assert R.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    torch.cuda.set_device(0)
    assert R.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    yield comm.value
    R.ncclCommDestroy(comm)


@pytest.mark.parametrize("with_comm", [False, True])
@pytest.mark.parametrize("n", [1000, (1 << 22) + 3])
def test_single_rank_u32_and_pairs(ctx, comm1, with_comm, n):
    import torch
    from inplacemsdradixsort_amd import MsdShard
    from oracle import oracle as O
    sh = MsdShard(ctx, comm1 if with_comm else None)
    assert (sh.rank, sh.world) == (0, 1)
    k = O.gen_uniform_u32(n, seed=n)
    t = dev(k)
    out = sh.sort_u32(t, None)
    assert out.data_ptr() == t.data_ptr() and (host(out, np.uint32) == O.sort_u32(k)).all()
    k64 = O.gen_uniform_u64(n, seed=n + 1)
    tk, tr = dev(k64), dev(k64)
    ok, orr = sh.sort_pairs_u64(tk, tr, None, None)
    assert (host(ok, np.uint64) == O.sort_u64(k64)).all() and torch.equal(ok, orr)
    sh.close()


def test_multi_call_on_one_device(ctx):
    """msd_sort_u32_multi with ndev = 1: contexts, threads and teardown of the one-call form (no communicator for one device)."""
    import torch
    from inplacemsdradixsort_amd import _lib
    from oracle import oracle as O
    R = _lib.load_rccl()
    n = (1 << 20) + 77
    k = O.gen_uniform_u32(n, seed=5)
    t = dev(k)
    devs = (C.c_int * 1)(0)
    keys = (C.c_void_p * 1)(t.data_ptr())
    ns = (C.c_uint64 * 1)(n)
    outp = (C.c_void_p * 1)()
    nout = (C.c_uint64 * 1)()
    torch.cuda.synchronize()
    rc = R.msd_sort_u32_multi(1, devs, keys, ns, None, 0, None, 0, 0, outp, nout)
    assert rc == 0 and outp[0] == t.data_ptr() and nout[0] == n
    assert (host(t, np.uint32) == O.sort_u32(k)).all()


def test_bad_arguments(ctx, comm1):
    from inplacemsdradixsort_amd import MsdError, MsdShard, _lib
    R = _lib.load_rccl()
    h = C.c_void_p()
    assert R.msd_shard_create(C.byref(h), None, None) != 0           # no context
    sh = MsdShard(ctx, comm1)
    import torch
    t = torch.zeros(16, dtype=torch.int32, device="cuda")
    out, n_out = C.c_void_p(), C.c_uint64()
    assert R.msd_sort_u32_sharded(sh._h, C.c_void_p(t.data_ptr()), 16, None, 0, None, 0, 7, C.byref(out), C.byref(n_out)) != 0   # scheme
    assert b"scheme" in R.msd_shard_last_error(sh._h)
    devs = (C.c_int * 3)(0, 0, 0)
    assert R.msd_sort_u32_multi(3, devs, None, None, None, 0, None, 0, 0, None, None) != 0   # not a power of two / null arrays
    sh.close()


@pytest.mark.parametrize("scheme", ["fine", "fine-whole-keys", "fine-histograms", "coarse"])
@pytest.mark.parametrize("n", [(1 << 20) + 3, 1 << 24])
def test_whole_exchange_through_rccl_on_one_rank(ctx, comm1, scheme, n):
    """msd_shard_set_option("force_exchange"): the single rank does NOT take the local shortcut -- top-digit passes,
    bucket boundaries, ncclAllGather of the counts, the send-matrix kernels, ncclGroupStart / ncclSend / ncclRecv to
    itself / ncclGroupEnd, and the leaf over the arrived extents (fine) or the sort of what arrived (coarse) all run,
    through a real RCCL communicator: everything of the N > 1 path that one GPU can execute."""
    import torch
    from inplacemsdradixsort_amd import MsdShard
    from oracle import oracle as O
    sh = MsdShard(ctx, comm1)
    sh.set_option("force_exchange", 1)
    if scheme == "fine-whole-keys":    # (the default fine exchange moves only the keys' low halves: option "low16")
        sh.set_option("low16", 0)
        scheme = "fine"
    cap = n + 64
    if scheme == "fine-histograms":    # (2^30 keys per rank: the buckets travel as histogram records; forced here at any size)
        sh.set_option("hist_min_keys", 0)
        cap = max(cap, 65536 * 17408 // 4)
        scheme = "fine"
    k = O.gen_uniform_u32(n, seed=n + len(scheme))
    t = dev(k)
    recv = torch.full((cap,), -1, dtype=torch.int32, device="cuda")
    work = torch.full((cap,), -1, dtype=torch.int32, device="cuda")
    out = sh.sort_u32(t, recv, work, scheme=scheme)
    assert out.data_ptr() == (work if scheme == "fine" else recv).data_ptr() and out.numel() == n
    assert (host(out, np.uint32) == O.sort_u32(k)).all()
    # tuples (coarse scheme: keys and rids in one group of sends and receives)
    m = n // 4
    k64 = O.gen_uniform_u64(m, seed=m)
    tk, tr = dev(k64), dev(k64 ^ np.uint64(0x5A5A5A5A5A5A5A5A))
    rk = torch.empty(m + 16, dtype=torch.int64, device="cuda")
    rr = torch.empty(m + 16, dtype=torch.int64, device="cuda")
    ok, orr = sh.sort_pairs_u64(tk, tr, rk, rr)
    assert ok.data_ptr() == rk.data_ptr() and ok.numel() == m
    assert (host(ok, np.uint64) == O.sort_u64(k64)).all()
    assert (host(orr, np.uint64) == (host(ok, np.uint64) ^ np.uint64(0x5A5A5A5A5A5A5A5A))).all()
    # a receive buffer that is too small: MSD_EOVERFLOW before anything is exchanged
    from inplacemsdradixsort_amd.dist import ReceiveOverflow
    with pytest.raises(ReceiveOverflow):
        sh.sort_u32(dev(k), recv[: n // 2], work, scheme=scheme)
    if cap > n + 64:
        # a bucket that does not fit a record (300 values with three copies each): the low halves travel instead
        k[:900] = np.uint32(0x12340000) | np.repeat(np.arange(300, dtype=np.uint32), 3)
        out = sh.sort_u32(dev(k), recv, work, scheme=scheme)
        assert (host(out, np.uint32) == np.sort(k)).all()
    sh.close()


def _nccl_worker(port, q):
    """torch.distributed with backend "nccl" (= RCCL) and ONE rank: dist.py's exchange code -- all_gather, the asynchronous
    all_to_all_single and its Work handle, the pipelined ShardedSorter -- on the real backend, and the C entry point on the
    communicator torch has set up."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from inplacemsdradixsort_amd import MsdContext, MsdShard, torch_nccl_comm
    from inplacemsdradixsort_amd.dist import ShardedSorter, sort_sharded_u32
    from oracle import oracle as O
    res = {}
    try:
        ctx = MsdContext(0)
        ctx.use_torch_stream()
        n = (1 << 22) + 5
        for scheme in ("fine", "coarse"):
            k = O.gen_uniform_u32(n, seed=7)
            t = dev(k)
            recv = torch.empty(n + 64, dtype=torch.int32, device="cuda")
            work = torch.empty(n + 64, dtype=torch.int32, device="cuda")
            out = sort_sharded_u32(ctx, t, recv, dist, 1, work=work, scheme=scheme, _force_exchange=True)
            res["oneshot " + scheme] = bool((host(out, np.uint32) == O.sort_u32(k)).all())
            # the pipelined form bench.py --gpus N times: exchange of shard s in flight while shard s - 1 is finished
            shards = [dev(O.gen_uniform_u32(n, seed=20 + s)) for s in range(3)]
            sorter = ShardedSorter(ctx, dist, 1, [torch.empty(n + 64, dtype=torch.int32, device="cuda") for _ in range(2)],
                                   work_bufs=[torch.empty(n + 64, dtype=torch.int32, device="cuda") for _ in range(3)], scheme=scheme,
                                   _force_exchange=True)
            outs = []
            for s in range(3):
                sorter.submit(shards[s])
                if s:
                    outs.append(host(sorter.collect(), np.uint32).copy())
            outs.append(host(sorter.collect(), np.uint32).copy())
            res["pipelined " + scheme] = all((outs[s] == O.sort_u32(O.gen_uniform_u32(n, seed=20 + s))).all() for s in range(3))
        # the histogram form of the fine exchange (2^30 keys per rank in production; forced here), one-shot and pipelined
        import inplacemsdradixsort_amd.dist as D
        D.FINE_HIST_MIN_KEYS = 0
        cap = 65536 * 17408 // 4
        k = O.gen_uniform_u32(n, seed=11)
        out = sort_sharded_u32(ctx, dev(k), torch.empty(cap, dtype=torch.int32, device="cuda"), dist, 1,
                               work=torch.empty(cap, dtype=torch.int32, device="cuda"), scheme="fine", _force_exchange=True)
        res["oneshot histograms"] = bool((host(out, np.uint32) == O.sort_u32(k)).all()) and ctx.stats().get("merge_rejected", 1) == 0
        sorter = ShardedSorter(ctx, dist, 1, [torch.empty(cap, dtype=torch.int32, device="cuda") for _ in range(2)],
                               work_bufs=[torch.empty(cap, dtype=torch.int32, device="cuda") for _ in range(2)], scheme="fine", _force_exchange=True)
        outs = []
        for s in range(3):
            sorter.submit(dev(O.gen_uniform_u32(n, seed=40 + s)))
            if s:
                outs.append(host(sorter.collect(), np.uint32).copy())
        outs.append(host(sorter.collect(), np.uint32).copy())
        res["pipelined histograms"] = all((outs[s] == O.sort_u32(O.gen_uniform_u32(n, seed=40 + s))).all() for s in range(3)) and len(sorter._send8) == 2
        D.FINE_HIST_MIN_KEYS = 3 << 28
        # the C entry point on torch's own communicator
        dist.barrier()
        sh = MsdShard(ctx, torch_nccl_comm(0))
        sh.set_option("force_exchange", 1)
        k = O.gen_uniform_u32(n, seed=9)
        out = sh.sort_u32(dev(k), torch.empty(n + 64, dtype=torch.int32, device="cuda"), torch.empty(n + 64, dtype=torch.int32, device="cuda"), scheme="fine")
        res["native on torch's communicator"] = bool((host(out, np.uint32) == O.sort_u32(k)).all()) and (sh.rank, sh.world) == (0, 1)
        sh.close()
        ctx.close()
    except Exception as e:  # report instead of hanging the parent
        res["exception"] = f"{type(e).__name__}: {e}"
    q.put(res)
    dist.destroy_process_group()


def test_dist_py_over_the_nccl_backend_on_one_rank():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    p = mpc.Process(target=_nccl_worker, args=(port, q))
    p.start()
    res = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert "exception" not in res, res
    assert len(res) == 7 and all(res.values()), res
