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
