"""ctypes binding of libinpmsdradix_hip.so (the C ABI of include/*.h).

Fails loudly: a missing library raises, a missing GPU makes ``msd_create`` fail
and :class:`~inplacemsdradixsort_amd.api.MsdContext` raises.  Nothing here falls
back to a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os

from . import _build

# every symbol include/msd_radix_hip.h and include/msb_64.h declare
EXPORTS = [
    "msd_create", "msd_destroy", "msd_set_stream", "msd_get_stream", "msd_get_device", "msd_reserve", "msd_workspace_bytes",
    "msd_last_error", "msd_version",
    "msd_sort_u32", "msd_sort_u64", "msd_sort_pairs_u64",
    "msd_sort_u32_bits", "msd_sort_u64_bits", "msd_sort_pairs_u64_bits",
    "msd_histogram_u32", "msd_histogram_u64", "msd_exclusive_scan_u64",
    "msd_partition_u32", "msd_partition_u64", "msd_partition_pairs_u64",
    "msd_sample_u32", "msd_splitters_u32", "msd_partition_by_splitters_u32",
    "msd_sample_u64", "msd_splitters_u64", "msd_partition_by_splitters_u64", "msd_partition_by_splitters_pairs_u64",
    "msd_sort_u32_top", "msd_sort_u64_top", "msd_sort_pairs_u64_top", "msd_bucket_bounds_u32", "msd_bucket_bounds_u64", "msd_merge_buckets_u32", "msd_pack_low16_u32", "msd_order_low16_u32", "msd_order_low16_counts_u32", "msd_order_low16_scatter_u32", "msd_merge_buckets_u32_low16", "msd_hist2_record_bytes", "msd_hist2_pack_u32", "msd_hist2_pack_u32_low16", "msd_bounds_from_counts16", "msd_merge_buckets_u32_hist2",
    "msd_sort_u32_segments", "msd_sort_u64_segments", "msd_sort_pairs_u64_segments", "msd_gather_runs_u32", "msd_gather_runs_u64",
    "msd_check_u32", "msd_check_u64",
    "msd_gen_uniform_u32", "msd_gen_uniform_u64", "msd_gen_zipf_u32", "msd_gen_iota_u64",
    "msd_gen_dup_u32", "msd_gen_mt19937_64",
    "msd_plan_first_round",
    "msd_set_option", "msd_set_profiling", "msd_phase_count", "msd_phase_name", "msd_phase_us", "msd_stat",
    "sort", "mamalloc", "check", "msb_64_last_error",
]

# every symbol include/msd_sharded_hip.h declares (libinpmsdradix_hip_rccl.so)
RCCL_EXPORTS = ["msd_shard_create", "msd_shard_destroy", "msd_shard_set_option", "msd_shard_rank", "msd_shard_world", "msd_shard_last_error",
                "msd_sort_u32_sharded", "msd_sort_pairs_u64_sharded", "msd_sort_u32_multi"]

_lib = None
_rccl = None


class MsdPlan(C.Structure):
    _fields_ = [("digit_width", C.c_uint32), ("digit_shift", C.c_uint32), ("block_elems", C.c_uint32),
                ("tile_elems", C.c_uint32), ("stripe_elems", C.c_uint64), ("stripes", C.c_uint64),
                ("leaf_capacity", C.c_uint64), ("leaf_count_bits", C.c_uint32), ("expected_rounds", C.c_uint32),
                ("workspace_bytes", C.c_uint64)]

_vp = C.c_void_p
_u64 = C.c_uint64
_u64p = C.POINTER(C.c_uint64)


def load(build_if_missing: bool = True) -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if build_if_missing and _build.stale():
        _build.build()
    if not os.path.exists(_build.LIB):
        raise RuntimeError(f"{_build.LIB} is missing: run inplacemsdradixsort_amd._build.build()")
    try:  # share the process's HIP runtime with torch when torch is in use
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(_build.LIB, mode=C.RTLD_GLOBAL if False else C.DEFAULT_MODE)
    L.msd_create.argtypes = [C.POINTER(_vp), C.c_int, _vp]
    L.msd_destroy.argtypes = [_vp]
    L.msd_set_stream.argtypes = [_vp, _vp]
    L.msd_get_stream.argtypes = [_vp]
    L.msd_get_stream.restype = _vp
    L.msd_get_device.argtypes = [_vp]
    L.msd_reserve.argtypes = [_vp, _u64, C.c_int, C.c_int]
    L.msd_workspace_bytes.argtypes = [_vp]
    L.msd_workspace_bytes.restype = _u64
    L.msd_last_error.argtypes = [_vp]
    L.msd_last_error.restype = C.c_char_p
    L.msd_version.restype = C.c_char_p
    for f in ("msd_sort_u32", "msd_sort_u64"):
        getattr(L, f).argtypes = [_vp, _vp, _u64]
    L.msd_sort_pairs_u64.argtypes = [_vp, _vp, _vp, _u64]
    for f in ("msd_sort_u32_bits", "msd_sort_u64_bits"):
        getattr(L, f).argtypes = [_vp, _vp, _u64, C.c_int]
    L.msd_sort_pairs_u64_bits.argtypes = [_vp, _vp, _vp, _u64, C.c_int]
    for f in ("msd_histogram_u32", "msd_histogram_u64"):
        getattr(L, f).argtypes = [_vp, _vp, _u64, C.c_uint, C.c_uint, _vp]
    L.msd_exclusive_scan_u64.argtypes = [_vp, _vp, _vp, _u64]
    for f in ("msd_partition_u32", "msd_partition_u64"):
        getattr(L, f).argtypes = [_vp, _vp, _u64, C.c_uint, C.c_uint, _vp]
    L.msd_partition_pairs_u64.argtypes = [_vp, _vp, _vp, _u64, C.c_uint, C.c_uint, _vp]
    L.msd_sample_u32.argtypes = [_vp, _vp, _u64, _u64, _u64, _vp]
    L.msd_splitters_u32.argtypes = [_vp, _vp, _u64, C.c_uint, _vp]
    L.msd_partition_by_splitters_u32.argtypes = [_vp, _vp, _u64, _vp, C.c_uint, _vp]
    L.msd_sample_u64.argtypes = [_vp, _vp, _u64, _u64, _u64, _vp]
    L.msd_splitters_u64.argtypes = [_vp, _vp, _u64, C.c_uint, _vp]
    L.msd_partition_by_splitters_u64.argtypes = [_vp, _vp, _u64, _vp, C.c_uint, _vp]
    L.msd_partition_by_splitters_pairs_u64.argtypes = [_vp, _vp, _vp, _u64, _vp, C.c_uint, _vp]
    for f in ("msd_sort_u32_segments", "msd_sort_u64_segments"):
        getattr(L, f).argtypes = [_vp, _vp, _u64, _u64p, C.c_uint32, C.c_int]
    L.msd_sort_pairs_u64_segments.argtypes = [_vp, _vp, _vp, _u64, _u64p, C.c_uint32, C.c_int]
    for f in ("msd_gather_runs_u32", "msd_gather_runs_u64"):
        getattr(L, f).argtypes = [_vp, _vp, _vp, _u64p, _u64p, _u64p, C.c_uint32]
    for f in ("msd_sort_u32_top", "msd_sort_u64_top"):
        getattr(L, f).argtypes = [_vp, _vp, _u64, C.c_int, C.c_int]
    L.msd_sort_pairs_u64_top.argtypes = [_vp, _vp, _vp, _u64, C.c_int, C.c_int]
    for f in ("msd_bucket_bounds_u32", "msd_bucket_bounds_u64"):
        getattr(L, f).argtypes = [_vp, _vp, _u64, C.c_uint, _u64, C.c_uint32, _vp]
    L.msd_merge_buckets_u32.argtypes = [_vp, _vp, _u64, _vp, _u64p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, _vp, _u64, _u64]
    L.msd_merge_buckets_u32_low16.argtypes = [_vp, _vp, _u64, _vp, _u64p, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _u64, _u64]
    L.msd_pack_low16_u32.argtypes = [_vp, _vp, _u64, _vp]
    L.msd_order_low16_u32.argtypes = [_vp, _vp, _u64, _vp, _vp]
    L.msd_order_low16_counts_u32.argtypes = [_vp, _vp, _u64, _vp]
    L.msd_order_low16_scatter_u32.argtypes = [_vp, _vp, _u64, _vp]
    L.msd_hist2_record_bytes.restype = C.c_uint64
    L.msd_hist2_record_bytes.argtypes = []
    L.msd_hist2_pack_u32.argtypes = [_vp, _vp, _u64, _vp, C.c_uint32, _vp, _u64, _vp]
    L.msd_hist2_pack_u32_low16.argtypes = [_vp, _vp, _u64, _vp, C.c_uint32, _vp, _u64, _vp]
    L.msd_bounds_from_counts16.argtypes = [_vp, _vp, _vp]
    L.msd_merge_buckets_u32_hist2.argtypes = [_vp, _vp, _u64, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _u64, _u64]
    L.msd_check_u32.argtypes = [_vp, _vp, _u64, _u64p, _u64p, _u64p]
    L.msd_check_u64.argtypes = [_vp, _vp, _vp, _u64, _u64p, _u64p, _u64p]
    L.msd_gen_uniform_u32.argtypes = [_vp, _vp, _u64, _u64, _u64]
    L.msd_gen_uniform_u64.argtypes = [_vp, _vp, _u64, _u64, _u64, C.c_int]
    L.msd_gen_zipf_u32.argtypes = [_vp, _vp, _u64, _u64, _u64]
    L.msd_gen_iota_u64.argtypes = [_vp, _vp, _u64, _u64]
    L.msd_gen_dup_u32.argtypes = [_vp, _vp, _u64, _u64, _u64, _u64]
    L.msd_gen_mt19937_64.argtypes = [_vp, _vp, _u64, _u64, C.c_int]
    L.msd_plan_first_round.argtypes = [_u64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(MsdPlan)]
    L.msd_set_option.argtypes = [_vp, C.c_char_p, C.c_int64]
    L.msd_set_profiling.argtypes = [_vp, C.c_int]
    L.msd_phase_count.argtypes = [_vp]
    L.msd_phase_name.argtypes = [_vp, C.c_int]
    L.msd_phase_name.restype = C.c_char_p
    L.msd_phase_us.argtypes = [_vp, C.c_int]
    L.msd_phase_us.restype = C.c_double
    L.msd_stat.argtypes = [_vp, C.c_char_p, _u64p]
    # reference surface (include/msb_64.h)
    L.sort.argtypes = [C.POINTER(_u64p), C.POINTER(_u64p), _u64p, C.c_int, C.c_int, C.c_double,
                       C.POINTER(C.c_char_p), _u64p]
    L.sort.restype = None
    L.msb_64_last_error.restype = C.c_char_p
    L.mamalloc.argtypes = [C.c_size_t]
    L.mamalloc.restype = _vp
    L.check.argtypes = [C.POINTER(_u64p), C.POINTER(_u64p), _u64p, C.c_int, C.c_int]
    L.check.restype = _u64
    _lib = L
    return L


def load_rccl(build_if_missing: bool = True) -> C.CDLL:
    """The multi-GPU entry points (include/msd_sharded_hip.h).  Loading it loads RCCL (the process's own if torch has
    loaded one: same soname)."""
    global _rccl
    if _rccl is not None:
        return _rccl
    load(build_if_missing)
    if build_if_missing and _build.rccl_stale():
        _build.build_rccl()
    if not os.path.exists(_build.RCCL_LIB):
        raise RuntimeError(f"{_build.RCCL_LIB} is missing: run inplacemsdradixsort_amd._build.build_rccl()")
    L = C.CDLL(_build.RCCL_LIB)
    _vpp = C.POINTER(_vp)
    L.msd_shard_create.argtypes = [_vpp, _vp, _vp]
    L.msd_shard_destroy.argtypes = [_vp]
    L.msd_shard_set_option.argtypes = [_vp, C.c_char_p, C.c_int64]
    L.msd_shard_rank.argtypes = [_vp]
    L.msd_shard_world.argtypes = [_vp]
    L.msd_shard_last_error.argtypes = [_vp]
    L.msd_shard_last_error.restype = C.c_char_p
    L.msd_sort_u32_sharded.argtypes = [_vp, _vp, _u64, _vp, _u64, _vp, _u64, C.c_int, _vpp, _u64p]
    L.msd_sort_pairs_u64_sharded.argtypes = [_vp, _vp, _vp, _u64, _vp, _vp, _u64, _vpp, _vpp, _u64p]
    L.msd_sort_u32_multi.argtypes = [C.c_int, C.POINTER(C.c_int), _vpp, _u64p, _vpp, _u64, _vpp, _u64, C.c_int, _vpp, _u64p]
    _rccl = L
    return L
