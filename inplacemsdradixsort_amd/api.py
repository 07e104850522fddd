"""Host-side mirror of the reference's interface for the hot path.

The reference is a C library with one public call, ``sort(keys, rids, size,
threads, numa, fudge, description, times)`` (include/msb_64.h:37-39) plus
``mamalloc`` and the test hook ``check`` (src/msb_64.c:2470).  :func:`sort`,
:func:`mamalloc` and :func:`check` below keep those names, argument meaning and
error behaviour on numpy host arrays; :class:`MsdContext` wraps the typed
device-resident entry points of include/msd_radix_hip.h on torch tensors (torch
only supplies device memory and the stream).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib


class MsdError(RuntimeError):
    pass


def _torch():
    import torch
    return torch


class MsdContext:
    """One sorting context (device + stream + auxiliary workspace)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self._L = _lib.load()
        h = C.c_void_p()
        rc = self._L.msd_create(C.byref(h), device, C.c_void_p(stream or 0))
        if rc != 0 or not h:
            raise MsdError(f"msd_create(device={device}) failed with {rc}: no usable HIP device "
                           "(this library has no CPU fallback)")
        self._h = h
        self.device = device

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.msd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers
    def _ok(self, rc: int) -> None:
        if rc != 0:
            raise MsdError(f"error {rc}: {self._L.msd_last_error(self._h).decode()}")

    def _ptr(self, t, dtype_size: int) -> C.c_void_p:
        torch = _torch()
        if not t.is_cuda or t.device.index != self.device:
            raise MsdError("tensor must live on the context's GPU")
        if not t.is_contiguous() or t.element_size() != dtype_size:
            raise MsdError("tensor must be contiguous with the expected element size")
        return C.c_void_p(t.data_ptr())

    def use_torch_stream(self) -> None:
        torch = _torch()
        self._ok(self._L.msd_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def reserve(self, n: int, key_bytes: int, val_bytes: int = 0) -> None:
        self._ok(self._L.msd_reserve(self._h, n, key_bytes, val_bytes))

    @property
    def workspace_bytes(self) -> int:
        return int(self._L.msd_workspace_bytes(self._h))

    # ---- the sort
    def sort_u32(self, keys, end_bit: int = 32) -> None:
        self._ok(self._L.msd_sort_u32_bits(self._h, self._ptr(keys, 4), keys.numel(), end_bit))

    def sort_u64(self, keys, end_bit: int = 64) -> None:
        self._ok(self._L.msd_sort_u64_bits(self._h, self._ptr(keys, 8), keys.numel(), end_bit))

    def sort_pairs_u64(self, keys, rids, end_bit: int = 64) -> None:
        if keys.numel() != rids.numel():
            raise MsdError("keys and rids differ in length")
        self._ok(self._L.msd_sort_pairs_u64_bits(self._h, self._ptr(keys, 8), self._ptr(rids, 8), keys.numel(), end_bit))

    # ---- fine-grained sharding (include/msd_radix_hip.h): top digits before the exchange, open bits after it
    def sort_top(self, keys, begin_bit: int, end_bit: Optional[int] = None, rids=None) -> None:
        """Orders ``keys`` by ``key >> begin_bit`` only (keys that agree above ``begin_bit`` end up adjacent, in any order)."""
        eb = keys.element_size() * 8 if end_bit is None else end_bit
        if rids is not None:
            self._ok(self._L.msd_sort_pairs_u64_top(self._h, self._ptr(keys, 8), self._ptr(rids, 8), keys.numel(), eb, begin_bit))
        elif keys.element_size() == 4:
            self._ok(self._L.msd_sort_u32_top(self._h, self._ptr(keys, 4), keys.numel(), eb, begin_bit))
        else:
            self._ok(self._L.msd_sort_u64_top(self._h, self._ptr(keys, 8), keys.numel(), eb, begin_bit))

    def bucket_bounds(self, keys, shift: int, nbuckets: int, first: int = 0):
        """int64[nbuckets + 1] on the device: bounds[b] = first index whose ``key >> shift`` is >= first + b
        (``keys`` ordered by ``key >> shift``)."""
        torch = _torch()
        out = torch.empty(nbuckets + 1, dtype=torch.int64, device=keys.device)
        f = self._L.msd_bucket_bounds_u32 if keys.element_size() == 4 else self._L.msd_bucket_bounds_u64
        self._ok(f(self._h, self._ptr(keys, keys.element_size()), keys.numel(), shift, first, nbuckets, C.c_void_p(out.data_ptr())))
        return out

    def merge_buckets(self, src, counts, src_base, open_bits: int, first_prefix: int, dst, n_expected: int) -> None:
        """Finishes the buckets a rank received: ``counts`` = int64[nsrc, nbuckets] on the device (extent lengths per
        source row and bucket), source x's extents lie back to back in ``src`` from ``src_base[x]`` on; the sorted
        buckets are written back to back into ``dst``."""
        nsrc, nb = int(counts.shape[0]), int(counts.shape[1])
        if len(src_base) != nsrc:
            raise MsdError("merge_buckets: one base offset per source row")
        if src.element_size() == 1:   # histogram records (hist2_pack): source x's at x * nb * HIST2_RECORD_BYTES
            if open_bits != 16:
                raise MsdError("merge_buckets: histogram records need 16 open bits")
            self._ok(self._L.msd_merge_buckets_u32_hist2(self._h, self._ptr(src, 1), src.numel(), self._ptr(counts, 8), nsrc, nb,
                                                         first_prefix, self._ptr(dst, 4), dst.numel(), n_expected))
            return
        if src.element_size() == 2:   # extents of low halves (pack_low16): the upper half of a key is its bucket's number
            if open_bits != 16:
                raise MsdError("merge_buckets: extents of low halves need 16 open bits")
            self._ok(self._L.msd_merge_buckets_u32_low16(self._h, self._ptr(src, 2), src.numel(), self._ptr(counts, 8), self._u64arr(src_base),
                                                         nsrc, nb, first_prefix, self._ptr(dst, 4), dst.numel(), n_expected))
            return
        self._ok(self._L.msd_merge_buckets_u32(self._h, self._ptr(src, 4), src.numel(), self._ptr(counts, 8), self._u64arr(src_base),
                                               nsrc, nb, open_bits, first_prefix, self._ptr(dst, 4), dst.numel(), n_expected))

    HIST2_RECORD_BYTES = 17408

    def hist2_pack(self, keys, bounds, rec):
        """``rec`` (uint8, >= (bounds.numel() - 1) * HIST2_RECORD_BYTES) <- one histogram record per bucket of the u32 ``keys``
        (ordered by their upper halves; ``bounds`` from :meth:`bucket_bounds` with shift 16).  Returns a one-element int32
        tensor on the device: non-zero = some bucket does not fit a record (send the low halves instead)."""
        torch = _torch()
        nb = bounds.numel() - 1
        es = keys.element_size()        # 4: whole keys; 2: the low halves order_low16 has written (half the bytes to read)
        if es not in (2, 4) or rec.element_size() != 1 or rec.numel() < nb * self.HIST2_RECORD_BYTES:
            raise MsdError("hist2_pack: u32 keys or their low halves, a uint8 buffer of one record per bucket")
        flag = torch.zeros(1, dtype=torch.int32, device=keys.device)
        f = self._L.msd_hist2_pack_u32 if es == 4 else self._L.msd_hist2_pack_u32_low16
        self._ok(f(self._h, self._ptr(keys, es), keys.numel(), self._ptr(bounds, 8), nb, self._ptr(rec, 1), rec.numel(), C.c_void_p(flag.data_ptr())))
        return flag

    def bounds_from_counts16(self, counts):
        """int64[65537] on the device: the prefix sums of 65536 bucket sizes."""
        torch = _torch()
        out = torch.empty(65537, dtype=torch.int64, device=counts.device)
        self._ok(self._L.msd_bounds_from_counts16(self._h, self._ptr(counts, 8), C.c_void_p(out.data_ptr())))
        return out

    def order_low16(self, keys, out):
        """Orders the u32 ``keys`` by their upper halves and writes only their low halves: ``out`` (int16, >= keys.numel())
        holds bucket after bucket (bucket = upper half), in any order inside a bucket; returns the 2^16 bucket sizes (int64,
        on the device).  ``keys`` is left ordered by its top 8 bits."""
        torch = _torch()
        if keys.element_size() != 4 or out.element_size() != 2 or out.numel() < keys.numel():
            raise MsdError("order_low16: u32 keys, an int16 buffer at least as long")
        counts = torch.empty(65536, dtype=torch.int64, device=keys.device)
        self._ok(self._L.msd_order_low16_u32(self._h, self._ptr(keys, 4), keys.numel(), self._ptr(out, 2), C.c_void_p(counts.data_ptr())))
        return counts

    def order_low16_counts(self, keys):
        """First half of :meth:`order_low16`: the 2^16 bucket sizes (ready in stream order); :meth:`order_low16_scatter` must
        be this context's next call."""
        torch = _torch()
        counts = torch.empty(65536, dtype=torch.int64, device=keys.device)
        self._ok(self._L.msd_order_low16_counts_u32(self._h, self._ptr(keys, 4), keys.numel(), C.c_void_p(counts.data_ptr())))
        return counts

    def order_low16_scatter(self, keys, out) -> None:
        if out.element_size() != 2 or out.numel() < keys.numel():
            raise MsdError("order_low16: an int16 buffer at least as long as the keys")
        self._ok(self._L.msd_order_low16_scatter_u32(self._h, self._ptr(keys, 4), keys.numel(), self._ptr(out, 2)))

    def pack_low16(self, keys, out) -> None:
        """``out`` (int16, >= keys.numel() elements) <- the low 16 bits of the u32 ``keys``, in order."""
        if keys.element_size() != 4 or out.element_size() != 2 or out.numel() < keys.numel():
            raise MsdError("pack_low16: u32 keys, an int16 buffer at least as long")
        self._ok(self._L.msd_pack_low16_u32(self._h, self._ptr(keys, 4), keys.numel(), self._ptr(out, 2)))

    # ---- building blocks
    def histogram(self, keys, shift: int, radix_bits: int):
        torch = _torch()
        out = torch.empty(1 << radix_bits, dtype=torch.int64, device=keys.device)
        f = self._L.msd_histogram_u32 if keys.element_size() == 4 else self._L.msd_histogram_u64
        self._ok(f(self._h, self._ptr(keys, keys.element_size()), keys.numel(), shift, radix_bits, C.c_void_p(out.data_ptr())))
        return out

    def exclusive_scan(self, x):
        torch = _torch()
        out = torch.empty_like(x)
        self._ok(self._L.msd_exclusive_scan_u64(self._h, self._ptr(x, 8), C.c_void_p(out.data_ptr()), x.numel()))
        return out

    def partition(self, keys, shift: int, radix_bits: int, rids=None):
        """One in-place digit pass; returns the bucket sizes (int64 tensor)."""
        torch = _torch()
        cnt = torch.zeros(1 << radix_bits, dtype=torch.int64, device=keys.device)
        if rids is not None:
            self._ok(self._L.msd_partition_pairs_u64(self._h, self._ptr(keys, 8), self._ptr(rids, 8), keys.numel(),
                                                      shift, radix_bits, C.c_void_p(cnt.data_ptr())))
        elif keys.element_size() == 4:
            self._ok(self._L.msd_partition_u32(self._h, self._ptr(keys, 4), keys.numel(), shift, radix_bits, C.c_void_p(cnt.data_ptr())))
        else:
            self._ok(self._L.msd_partition_u64(self._h, self._ptr(keys, 8), keys.numel(), shift, radix_bits, C.c_void_p(cnt.data_ptr())))
        return cnt

    # ---- a rank of the multi-GPU sort after its exchange (reference: local sorting of whole key ranges, src/msb_64.c:2200-2255)
    @staticmethod
    def _u64arr(xs):
        """Host array of uint64 for the C ABI (a numpy array goes through without a per-element conversion)."""
        a = np.ascontiguousarray(xs, dtype=np.uint64)
        p = a.ctypes.data_as(C.POINTER(C.c_uint64))
        p._keep = a   # the array must outlive the call
        return p

    def sort_segments(self, keys, seg_off, end_bit: int, rids=None) -> None:
        """Sorts the independent segments [seg_off[i], seg_off[i+1]) on their low ``end_bit`` bits in one call
        (``seg_off``: nseg + 1 ascending element offsets on the host)."""
        nseg = len(seg_off) - 1
        if nseg <= 0:
            return
        off = self._u64arr(seg_off)
        if rids is not None:
            self._ok(self._L.msd_sort_pairs_u64_segments(self._h, self._ptr(keys, 8), self._ptr(rids, 8), keys.numel(), off, nseg, end_bit))
        elif keys.element_size() == 4:
            self._ok(self._L.msd_sort_u32_segments(self._h, self._ptr(keys, 4), keys.numel(), off, nseg, end_bit))
        else:
            self._ok(self._L.msd_sort_u64_segments(self._h, self._ptr(keys, 8), keys.numel(), off, nseg, end_bit))

    def gather_runs(self, dst, src, src_off, dst_off, lens) -> None:
        """Copies run i (``lens[i]`` elements) from ``src[src_off[i]:]`` to ``dst[dst_off[i]:]``, all runs in one launch."""
        if not (len(src_off) == len(dst_off) == len(lens)):
            raise MsdError("gather_runs: the three lists differ in length")
        if dst.element_size() != src.element_size():
            raise MsdError("gather_runs: element sizes differ")
        f = self._L.msd_gather_runs_u32 if src.element_size() == 4 else self._L.msd_gather_runs_u64
        self._ok(f(self._h, self._ptr(dst, dst.element_size()), self._ptr(src, src.element_size()),
                   self._u64arr(src_off), self._u64arr(dst_off), self._u64arr(lens), len(lens)))

    # ---- splitter service (reference: src/msb_64.c:1511-1521, :1304-1322, :188-204)
    def sample(self, keys, m: int, seed: int = 0x5EED0007):
        """m keys drawn from the (unsorted) tensor at pseudo-random positions mulhi(splitmix64(seed + i), n);
        u32 (int32 tensor) or u64 (int64 tensor) keys."""
        torch = _torch()
        es = keys.element_size()
        out = torch.empty(m, dtype=keys.dtype, device=keys.device)
        f = self._L.msd_sample_u32 if es == 4 else self._L.msd_sample_u64
        self._ok(f(self._h, self._ptr(keys, es), keys.numel(), m, seed, C.c_void_p(out.data_ptr())))
        return out

    def splitters(self, sorted_sample, parts: int):
        """parts-1 equi-depth delimiters (bit patterns in a tensor of the sample's dtype) with the reference's duplicate rule."""
        torch = _torch()
        es = sorted_sample.element_size()
        out = torch.empty(max(parts - 1, 0), dtype=sorted_sample.dtype, device=sorted_sample.device)
        f = self._L.msd_splitters_u32 if es == 4 else self._L.msd_splitters_u64
        self._ok(f(self._h, self._ptr(sorted_sample, es), sorted_sample.numel(), parts, C.c_void_p(out.data_ptr())))
        return out

    sample_u32 = sample          # (round 2's names)
    splitters_u32 = splitters

    def partition_by_splitters(self, keys, delims, parts: int, rids=None):
        """One in-place pass: range p = keys in (delims[p-1], delims[p]] (rids move with their keys); returns the range
        sizes (int64 tensor)."""
        torch = _torch()
        es = keys.element_size()
        cnt = torch.zeros(parts, dtype=torch.int64, device=keys.device)
        dp = self._ptr(delims, es) if parts > 1 else C.c_void_p(0)
        if rids is not None:
            self._ok(self._L.msd_partition_by_splitters_pairs_u64(self._h, self._ptr(keys, 8), self._ptr(rids, 8), keys.numel(), dp, parts,
                                                                  C.c_void_p(cnt.data_ptr())))
        else:
            f = self._L.msd_partition_by_splitters_u32 if es == 4 else self._L.msd_partition_by_splitters_u64
            self._ok(f(self._h, self._ptr(keys, es), keys.numel(), dp, parts, C.c_void_p(cnt.data_ptr())))
        return cnt

    def check(self, keys, rids=None) -> Tuple[int, int, int]:
        """(violations, sum, xor): device form of the reference's check()."""
        v, s, x = C.c_uint64(), C.c_uint64(), C.c_uint64()
        if keys.element_size() == 4:
            self._ok(self._L.msd_check_u32(self._h, self._ptr(keys, 4), keys.numel(), C.byref(v), C.byref(s), C.byref(x)))
        else:
            rp = self._ptr(rids, 8) if rids is not None else C.c_void_p(0)
            self._ok(self._L.msd_check_u64(self._h, self._ptr(keys, 8), rp, keys.numel(), C.byref(v), C.byref(s), C.byref(x)))
        return int(v.value), int(s.value), int(x.value)

    # ---- synthetic inputs
    def gen_uniform_u32(self, keys, seed: int = 0x5EED0001, first: int = 0) -> None:
        self._ok(self._L.msd_gen_uniform_u32(self._h, self._ptr(keys, 4), keys.numel(), seed, first))

    def gen_uniform_u64(self, keys, seed: int = 0x5EED0005, first: int = 0, shift_right: int = 0) -> None:
        self._ok(self._L.msd_gen_uniform_u64(self._h, self._ptr(keys, 8), keys.numel(), seed, first, shift_right))

    def gen_zipf_u32(self, keys, seed: int = 0x5EED0003, first: int = 0) -> None:
        self._ok(self._L.msd_gen_zipf_u32(self._h, self._ptr(keys, 4), keys.numel(), seed, first))

    def gen_dup_u32(self, keys, distinct: int, seed: int = 0x5EED0009, first: int = 0) -> None:
        self._ok(self._L.msd_gen_dup_u32(self._h, self._ptr(keys, 4), keys.numel(), seed, first, distinct))

    def gen_mt19937_64(self, keys, seed: int, shift_right: int = 0) -> None:
        """The reference's own RNG stream (src/rand.c:47-86) into an int64 tensor."""
        self._ok(self._L.msd_gen_mt19937_64(self._h, self._ptr(keys, 8), keys.numel(), seed, shift_right))

    def gen_iota_u64(self, vals, first: int = 0) -> None:
        self._ok(self._L.msd_gen_iota_u64(self._h, self._ptr(vals, 8), vals.numel(), first))

    def set_option(self, name: str, value: int) -> None:
        """Tuning knob of include/msd_radix_hip.h (``direct_mode``, ``direct_min``)."""
        self._ok(self._L.msd_set_option(self._h, name.encode(), int(value)))

    # ---- phase report
    def set_profiling(self, on: bool) -> None:
        self._ok(self._L.msd_set_profiling(self._h, int(on)))

    def phases(self) -> List[Tuple[str, float]]:
        n = self._L.msd_phase_count(self._h)
        return [(self._L.msd_phase_name(self._h, i).decode(), float(self._L.msd_phase_us(self._h, i))) for i in range(n)]

    def stats(self) -> Dict[str, int]:
        out = {}
        for name in ("rounds", "parents", "stripes", "children", "slots", "holes", "chain_steps",
                     "small_segments", "count_segments", "big_count_segments", "direct_rounds", "regpart_rounds", "skipped_bits", "bit_skip_restarts", "bit_skip_checked_by_histogram",
                     "merge_rejected", "leaf17_segments", "leaf17_rejected", "leaf17_slow_segments", "leaf17_launches", "workspace_bytes"):
            v = C.c_uint64()
            if self._L.msd_stat(self._h, name.encode(), C.byref(v)) == 0:
                out[name] = int(v.value)
        return out


class MsdShard:
    """One rank of a sharded sort behind the C ABI (include/msd_sharded_hip.h): RCCL is called from C, no
    torch.distributed on the data path.  ``nccl_comm``: the address of the caller's ncclComm_t -- e.g.
    :func:`torch_nccl_comm` for the process group torch.distributed has set up -- or None for a single rank."""

    def __init__(self, ctx: MsdContext, nccl_comm: Optional[int] = None):
        self._L = _lib.load_rccl()
        self.ctx = ctx
        h = C.c_void_p()
        rc = self._L.msd_shard_create(C.byref(h), ctx._h, C.c_void_p(nccl_comm or 0))
        if rc != 0 or not h:
            raise MsdError(f"msd_shard_create failed with {rc}")
        self._h = h
        self.rank = int(self._L.msd_shard_rank(h))
        self.world = int(self._L.msd_shard_world(h))

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.msd_shard_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ok(self, rc: int) -> None:
        if rc == -5:
            from .dist import ReceiveOverflow
            raise ReceiveOverflow(self._L.msd_shard_last_error(self._h).decode())
        if rc != 0:
            raise MsdError(f"error {rc}: {self._L.msd_shard_last_error(self._h).decode()}")

    def set_option(self, name: str, value: int) -> None:
        self._ok(self._L.msd_shard_set_option(self._h, name.encode(), int(value)))

    def sort_u32(self, keys, recv, work=None, scheme: Optional[str] = None):
        """``msd_sort_u32_sharded``: returns this rank's sorted key range -- a view of ``work`` (fine scheme), ``recv``
        (coarse) or ``keys`` (single rank)."""
        p = self.ctx._ptr
        out, n_out = C.c_void_p(), C.c_uint64()
        sc = {None: 0, "fine": 1, "coarse": 2}[scheme]
        self._ok(self._L.msd_sort_u32_sharded(self._h, p(keys, 4), keys.numel(), p(recv, 4) if recv is not None else None,
                                              recv.numel() if recv is not None else 0, p(work, 4) if work is not None else None,
                                              work.numel() if work is not None else 0, sc, C.byref(out), C.byref(n_out)))
        for t in (work, recv, keys):
            if t is not None and t.data_ptr() == out.value:
                return t[:n_out.value]
        raise MsdError("msd_sort_u32_sharded returned an unknown buffer")

    def sort_pairs_u64(self, keys, rids, recv_keys, recv_rids):
        p = self.ctx._ptr
        ok_, or_, n_out = C.c_void_p(), C.c_void_p(), C.c_uint64()
        cap = min(recv_keys.numel(), recv_rids.numel()) if recv_keys is not None else 0
        self._ok(self._L.msd_sort_pairs_u64_sharded(self._h, p(keys, 8), p(rids, 8), keys.numel(),
                                                    p(recv_keys, 8) if recv_keys is not None else None,
                                                    p(recv_rids, 8) if recv_rids is not None else None, cap,
                                                    C.byref(ok_), C.byref(or_), C.byref(n_out)))
        if ok_.value == keys.data_ptr():
            return keys[:n_out.value], rids[:n_out.value]
        return recv_keys[:n_out.value], recv_rids[:n_out.value]


def torch_nccl_comm(device: int) -> int:
    """Address of the ncclComm_t (RCCL) behind torch.distributed's default process group on ``device`` (backend "nccl";
    the communicator exists once a collective has run on it)."""
    import torch
    import torch.distributed as dist
    pg = dist.distributed_c10d._get_default_group()
    be = pg._get_backend(torch.device("cuda", device))
    return int(be._comm_ptr())


def plan_first_round(n: int, key_bytes: int = 4, val_bytes: int = 0, end_bit: Optional[int] = None,
                     compute_units: int = 256) -> Dict[str, int]:
    """Host-only pass planner (the counterpart of the reference's schedule_passes,
    src/msb_64.c:1334-1400); needs no GPU."""
    L = _lib.load()
    p = _lib.MsdPlan()
    rc = L.msd_plan_first_round(n, key_bytes, val_bytes, key_bytes * 8 if end_bit is None else end_bit,
                                compute_units, C.byref(p))
    if rc != 0:
        raise MsdError(f"msd_plan_first_round failed with {rc}")
    return {f: int(getattr(p, f)) for f, _ in p._fields_}


# ---------------------------------------------------------------------------
# the reference's own surface, on host numpy arrays
# ---------------------------------------------------------------------------

_u64p = C.POINTER(C.c_uint64)


def mamalloc(n_bytes: int) -> np.ndarray:
    """64-byte aligned host buffer, as the reference's mamalloc (src/msb_64.c:111-115)."""
    L = _lib.load()
    p = L.mamalloc(n_bytes)
    if not p:
        raise MemoryError(n_bytes)
    buf = (C.c_uint8 * n_bytes).from_address(p)
    arr = np.frombuffer(buf, dtype=np.uint8)
    return arr  # freed by the C library's allocator only at process exit (tests use small sizes)


def sort(keys: Sequence[np.ndarray], rids: Sequence[np.ndarray], size: Sequence[int], threads: int = 64,
         numa: Optional[int] = None, fudge: float = 1.0):
    """``sort(keys, rids, size, threads, numa, fudge, description, times)`` of
    include/msb_64.h:37-39 on lists of uint64 numpy arrays (sorted in place).
    Returns (description, times) as the reference fills them."""
    L = _lib.load()
    numa = len(keys) if numa is None else numa
    for a in list(keys) + list(rids):
        if a.dtype != np.uint64 or not a.flags.c_contiguous:
            raise MsdError("arrays must be contiguous uint64")
    # size[] is REWRITTEN by the call (the reference does the same, src/msb_64.c:2180): it must be writable ...
    if isinstance(size, tuple) or (isinstance(size, np.ndarray) and not size.flags.writeable):
        raise MsdError("size must be a mutable sequence: sort() rewrites it")
    # ... and with fudge > 1 an array may come back with up to size[a] * fudge tuples (the capacity the reference
    # requires of the caller, :1574-1578): the arrays must have that room
    for a in range(numa):
        cap = int(float(size[a]) * fudge) if fudge > 1.0 else int(size[a])
        if keys[a].size < cap or rids[a].size < cap:
            raise MsdError(f"array {a} holds {min(keys[a].size, rids[a].size)} elements but needs room for int(size * fudge) = {cap}")
    KA = (_u64p * numa)(*[a.ctypes.data_as(_u64p) for a in keys[:numa]])
    RA = (_u64p * numa)(*[a.ctypes.data_as(_u64p) for a in rids[:numa]])
    sz = np.array(list(size)[:numa], dtype=np.uint64)
    desc = (C.c_char_p * 11)()
    times = np.zeros(10, dtype=np.uint64)
    L.sort(KA, RA, sz.ctypes.data_as(_u64p), threads, numa, fudge, desc, times.ctypes.data_as(_u64p))
    for i, s in enumerate(sz):
        size[i] = int(s)
    err = L.msb_64_last_error()
    if err:
        raise MsdError(err.decode())
    return [d.decode() if d is not None else None for d in desc], times


def check(keys: Sequence[np.ndarray], rids: Optional[Sequence[np.ndarray]], size: Sequence[int], numa: Optional[int] = None,
          same: bool = True) -> int:
    """``check(keys, rids, size, numa, same)`` of src/msb_64.c:2470: returns the key
    checksum; aborts the process on an order or key!=rid violation, as the
    reference's asserts do."""
    L = _lib.load()
    numa = len(keys) if numa is None else numa
    KA = (_u64p * numa)(*[a.ctypes.data_as(_u64p) for a in keys[:numa]])
    RA = (_u64p * numa)(*[a.ctypes.data_as(_u64p) for a in rids[:numa]]) if rids is not None else None
    sz = np.array(list(size)[:numa], dtype=np.uint64)
    return int(L.check(KA, RA, sz.ctypes.data_as(_u64p), numa, int(same)))
