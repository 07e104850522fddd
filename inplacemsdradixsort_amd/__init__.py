"""MI355X-native in-place MSD radix sort behind the C ABI of
MichaelAxtmann/InPlaceMSDradixsort (see include/msb_64.h, include/msd_radix_hip.h)."""
from ._build import build, LIB  # noqa: F401
from .api import MsdContext, MsdError, MsdShard, torch_nccl_comm, sort, mamalloc, check, plan_first_round  # noqa: F401

__all__ = ["build", "LIB", "MsdContext", "MsdError", "MsdShard", "torch_nccl_comm", "sort", "mamalloc", "check", "plan_first_round"]
