"""ctypes front end of the parity oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() import this
module; nothing under inplacemsdradixsort_amd/ does (tests/test_no_oracle_in_product.py
enforces it).

Two libraries live behind it:

* ``liborc.so``              -- oracle/msd_oracle.c, the CPU restatement.
* ``_ref/libref_msb64.so``   -- the reference compiled from its own sources by
  oracle/Makefile (only present if it was built in the build container; it
  travels to the GPU box as a prebuilt file).  Symbols bound here are the ones
  the reference exports: src/msb_64.c:1334 schedule_passes, :1007 local_radixsort,
  :701 histogram, :740 partition_ip, :785 partition_ip_buf, :2261 sort, :111 mamalloc.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORC_PATH = os.path.join(_HERE, "liborc.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libref_msb64.so")

_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_i8p = C.POINTER(C.c_int8)


def build(force: bool = False) -> None:
    """Compile liborc.so (and _ref when /root/reference is present)."""
    if force or not os.path.exists(_ORC_PATH) or (
        os.path.getmtime(_ORC_PATH) < os.path.getmtime(os.path.join(_HERE, "msd_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "liborc.so"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/src/msb_64.c") and (force or not os.path.exists(_REF_PATH)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def _ptr(a: np.ndarray, ty):
    return a.ctypes.data_as(ty)


def aligned(n: int, dtype=np.uint64, align: int = 64) -> np.ndarray:
    """numpy array whose data pointer is `align`-byte aligned (the reference
    asserts 16 B and effectively needs 64 B, src/msb_64.c:505-506, 789)."""
    item = np.dtype(dtype).itemsize
    raw = np.empty(n * item + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n * item].view(dtype)


class _Orc:
    def __init__(self):
        build()
        self.lib = C.CDLL(_ORC_PATH)
        L = self.lib
        L.orc_schedule_passes.restype = C.c_int
        L.orc_schedule_passes.argtypes = [C.c_uint64, C.c_int, _i8p, _i8p]
        L.orc_histogram.argtypes = [_u64p, C.c_uint64, _u64p, C.c_uint, C.c_uint]
        L.orc_histogram_u32.argtypes = [_u32p, C.c_uint64, _u64p, C.c_uint, C.c_uint]
        L.orc_exclusive_scan.argtypes = [_u64p, _u64p, C.c_uint64]
        L.orc_partition_ip.argtypes = [_u64p, _u64p, C.c_uint64, _u64p, _u64p, C.c_uint, C.c_uint]
        L.orc_partition_ip_buf.argtypes = [_u64p, _u64p, C.c_uint64, _u64p, C.c_uint, C.c_uint]
        L.orc_insertsort.argtypes = [_u64p, _u64p, C.c_uint64]
        L.orc_combsort.argtypes = [_u64p, _u64p, C.c_uint64]
        L.orc_sort_pairs_u64.restype = C.c_int
        L.orc_sort_pairs_u64.argtypes = [_u64p, _u64p, C.c_uint64, C.c_int, _i8p, _i8p]
        L.orc_sort_u32.restype = C.c_int
        L.orc_sort_u32.argtypes = [_u32p, C.c_uint64]
        L.orc_sort_u64.restype = C.c_int
        L.orc_sort_u64.argtypes = [_u64p, C.c_uint64]
        L.orc_check.restype = C.c_uint64
        L.orc_check.argtypes = [C.POINTER(_u64p), C.POINTER(_u64p), _u64p, C.c_int, C.c_int, _u64p, _u64p]
        for g, ty in (("orc_gen_uniform_u32", _u32p), ("orc_gen_uniform_u64", _u64p), ("orc_gen_zipf_u32", _u32p)):
            getattr(L, g).argtypes = [ty, C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_gen_dup_u32.argtypes = [_u32p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_mt19937_64.argtypes = [_u64p, C.c_uint64, C.c_uint64]
        L.orc_sample_u32.argtypes = [_u32p, C.c_uint64, C.c_uint64, C.c_uint64, _u32p]
        L.orc_extract_delimiters.argtypes = [_u64p, C.c_uint64, C.c_uint64, _u64p]
        L.orc_range_of.restype = C.c_uint64
        L.orc_range_of.argtypes = [_u64p, C.c_uint64, C.c_uint64]
        L.orc_range_histogram_u32.argtypes = [_u32p, C.c_uint64, _u64p, C.c_uint64, _u64p]
        L.orc_sample_u64.argtypes = [_u64p, C.c_uint64, C.c_uint64, C.c_uint64, _u64p]
        L.orc_range_histogram_u64.argtypes = [_u64p, C.c_uint64, _u64p, C.c_uint64, _u64p]


_orc = None


def orc() -> _Orc:
    global _orc
    if _orc is None:
        _orc = _Orc()
    return _orc


# ---------------------------------------------------------------- restatement

def schedule_passes(size: int, bits: int):
    rb = np.zeros(8, np.int8)
    bf = np.zeros(8, np.int8)
    p = orc().lib.orc_schedule_passes(size, bits, _ptr(rb, _i8p), _ptr(bf, _i8p))
    return p, rb[: p + 1].tolist(), bf[: p + 1].tolist()


def histogram(keys: np.ndarray, shift: int, radix_bits: int) -> np.ndarray:
    out = np.zeros(1 << radix_bits, np.uint64)
    keys = np.ascontiguousarray(keys)
    if keys.dtype == np.uint32:
        orc().lib.orc_histogram_u32(_ptr(keys, _u32p), keys.size, _ptr(out, _u64p), shift, radix_bits)
    else:
        assert keys.dtype == np.uint64
        orc().lib.orc_histogram(_ptr(keys, _u64p), keys.size, _ptr(out, _u64p), shift, radix_bits)
    return out


def exclusive_scan(count: np.ndarray) -> np.ndarray:
    count = np.ascontiguousarray(count, np.uint64)
    out = np.zeros_like(count)
    orc().lib.orc_exclusive_scan(_ptr(count, _u64p), _ptr(out, _u64p), count.size)
    return out


def partition(keys: np.ndarray, rids: np.ndarray, shift: int, radix_bits: int, buffered: bool):
    """One in-place digit pass (histogram + cycle-leader permute) on copies."""
    k = np.array(keys, np.uint64)
    r = np.array(rids, np.uint64)
    h = histogram(k, shift, radix_bits)
    if buffered:
        orc().lib.orc_partition_ip_buf(_ptr(k, _u64p), _ptr(r, _u64p), k.size, _ptr(h, _u64p), shift, radix_bits)
    else:
        offs = np.zeros_like(h)
        orc().lib.orc_partition_ip(_ptr(k, _u64p), _ptr(r, _u64p), k.size, _ptr(h, _u64p), _ptr(offs, _u64p), shift, radix_bits)
    return k, r, h


def sort_pairs_u64(keys: np.ndarray, rids: np.ndarray, bits: int = 64):
    k = np.array(keys, np.uint64)
    r = np.array(rids, np.uint64)
    orc().lib.orc_sort_pairs_u64(_ptr(k, _u64p), _ptr(r, _u64p), k.size, bits, None, None)
    return k, r


def sort_u32(keys: np.ndarray) -> np.ndarray:
    k = np.array(keys, np.uint32)
    rc = orc().lib.orc_sort_u32(_ptr(k, _u32p), k.size)
    assert rc >= 0
    return k


def sort_u64(keys: np.ndarray) -> np.ndarray:
    k = np.array(keys, np.uint64)
    rc = orc().lib.orc_sort_u64(_ptr(k, _u64p), k.size)
    assert rc >= 0
    return k


def sort_u32_inplace(keys: np.ndarray) -> None:
    """Timed by bench.py's cpu_baseline leg ("port")."""
    assert keys.dtype == np.uint32 and keys.flags.c_contiguous
    orc().lib.orc_sort_u32(_ptr(keys, _u32p), keys.size)


def check(key_arrays, rid_arrays, same: bool):
    """(sum, xor, violations) over the concatenation of the caller arrays."""
    n = len(key_arrays)
    KA = (_u64p * n)(*[_ptr(a, _u64p) for a in key_arrays])
    RA = (_u64p * n)(*[_ptr(a, _u64p) for a in rid_arrays]) if rid_arrays is not None else None
    sizes = np.array([a.size for a in key_arrays], np.uint64)
    bad = C.c_uint64(0)
    x = C.c_uint64(0)
    s = orc().lib.orc_check(KA, RA, _ptr(sizes, _u64p), n, int(same), C.byref(bad), C.byref(x))
    return int(s), int(x.value), int(bad.value)


def gen_uniform_u32(n: int, seed: int = 0x5EED0001, first: int = 0) -> np.ndarray:
    out = np.empty(n, np.uint32)
    orc().lib.orc_gen_uniform_u32(_ptr(out, _u32p), n, seed, first)
    return out


def gen_uniform_u64(n: int, seed: int = 0x5EED0005, first: int = 0) -> np.ndarray:
    out = np.empty(n, np.uint64)
    orc().lib.orc_gen_uniform_u64(_ptr(out, _u64p), n, seed, first)
    return out


def gen_zipf_u32(n: int, seed: int = 0x5EED0003, first: int = 0) -> np.ndarray:
    out = np.empty(n, np.uint32)
    orc().lib.orc_gen_zipf_u32(_ptr(out, _u32p), n, seed, first)
    return out


def gen_dup_u32(n: int, distinct: int, seed: int = 0x5EED0009, first: int = 0) -> np.ndarray:
    out = np.empty(n, np.uint32)
    orc().lib.orc_gen_dup_u32(_ptr(out, _u32p), n, seed, first, distinct)
    return out


def mt19937_64(n: int, seed: int) -> np.ndarray:
    """rand64_init(seed) + n x rand64_next() as the reference implements them (src/rand.c:47-86)."""
    out = np.empty(n, np.uint64)
    orc().lib.orc_mt19937_64(_ptr(out, _u64p), n, seed)
    return out


def ref_mt19937_64(n: int, seed: int) -> np.ndarray:
    """The same stream from the reference's own rand.c (oracle/_ref)."""
    L = ref().lib
    L.rand64_init.restype = C.c_void_p
    L.rand64_init.argtypes = [C.c_uint64]
    L.rand64_next.restype = C.c_uint64
    L.rand64_next.argtypes = [C.c_void_p]
    st = L.rand64_init(seed)
    return np.array([L.rand64_next(st) for _ in range(n)], dtype=np.uint64)


# ------------------------------------------------ splitter front end (skew)

def sample_u32(keys: np.ndarray, m: int, seed: int = 0x5EED0007) -> np.ndarray:
    """sample[p] = keys[mulhi(splitmix64(seed + p), n)] (sampling of src/msb_64.c:1511-1521)."""
    keys = np.ascontiguousarray(keys, np.uint32)
    out = np.empty(m, np.uint32)
    orc().lib.orc_sample_u32(_ptr(keys, _u32p), keys.size, m, seed, _ptr(out, _u32p))
    return out


def sample_u64(keys: np.ndarray, m: int, seed: int = 0x5EED0007) -> np.ndarray:
    """The same for 64-bit keys, the reference's own key type."""
    keys = np.ascontiguousarray(keys, np.uint64)
    out = np.empty(m, np.uint64)
    orc().lib.orc_sample_u64(_ptr(keys, _u64p), keys.size, m, seed, _ptr(out, _u64p))
    return out


def range_histogram_u64(keys: np.ndarray, delimiters: np.ndarray) -> np.ndarray:
    keys = np.ascontiguousarray(keys, np.uint64)
    d = np.ascontiguousarray(delimiters, np.uint64)
    out = np.zeros(d.size + 1, np.uint64)
    orc().lib.orc_range_histogram_u64(_ptr(keys, _u64p), keys.size, _ptr(d, _u64p), d.size, _ptr(out, _u64p))
    return out


def extract_delimiters(sorted_sample: np.ndarray, parts: int) -> np.ndarray:
    """parts-1 delimiters of a sorted sample (extract_delimiters, src/msb_64.c:1304-1322)."""
    s = np.ascontiguousarray(sorted_sample, np.uint64)
    out = np.zeros(max(parts - 1, 0), np.uint64)
    if parts > 1:
        orc().lib.orc_extract_delimiters(_ptr(s, _u64p), s.size, parts, _ptr(out, _u64p))
    return out


def range_histogram_u32(keys: np.ndarray, delimiters: np.ndarray) -> np.ndarray:
    """Range sizes under the lower-bound range function (binary_search_64, src/msb_64.c:188-204)."""
    keys = np.ascontiguousarray(keys, np.uint32)
    d = np.ascontiguousarray(delimiters, np.uint64)
    out = np.zeros(d.size + 1, np.uint64)
    orc().lib.orc_range_histogram_u32(_ptr(keys, _u32p), keys.size, _ptr(d, _u64p), d.size, _ptr(out, _u64p))
    return out


def range_of_u32(keys: np.ndarray, delimiters: np.ndarray) -> np.ndarray:
    """Range id per key (numpy form of the same lower bound: number of delimiters < key)."""
    return np.searchsorted(np.asarray(delimiters, np.uint64), np.asarray(keys, np.uint64), side="left")


def ref_extract_delimiters(sorted_sample: np.ndarray, parts: int) -> np.ndarray:
    """The reference's own extract_delimiters (oracle/_ref), which finds `parts` from a ~0 terminator (:1307)."""
    s = np.ascontiguousarray(sorted_sample, np.uint64)
    delim = np.zeros(parts, np.uint64)
    delim[parts - 1] = np.uint64(2**64 - 1)
    L = ref().lib
    L.extract_delimiters.argtypes = [_u64p, C.c_uint64, _u64p]
    L.extract_delimiters(_ptr(s, _u64p), s.size, _ptr(delim, _u64p))
    return delim[:parts - 1].copy()


# ------------------------------------------------------------ real reference

class _Ref:
    """The reference itself (oracle/_ref/libref_msb64.so)."""

    def __init__(self):
        build()
        if not os.path.exists(_REF_PATH):
            raise FileNotFoundError(_REF_PATH)
        self.lib = C.CDLL(_REF_PATH, mode=os.RTLD_LOCAL if hasattr(os, "RTLD_LOCAL") else 0)
        L = self.lib
        L.schedule_passes.restype = C.c_int
        L.schedule_passes.argtypes = [C.c_uint64, C.c_int8, _i8p, _i8p]
        L.histogram.argtypes = [_u64p, C.c_uint64, _u64p, C.c_uint8, C.c_uint8]
        L.partition_ip.argtypes = [_u64p, _u64p, C.c_uint64, _u64p, _u64p, C.c_uint8, C.c_uint8]
        L.partition_ip_buf.argtypes = [_u64p, _u64p, C.c_uint64, _u64p, C.c_uint8, C.c_uint8]
        L.local_radixsort.argtypes = [_u64p, _u64p, C.c_uint64, _i8p, _i8p, C.c_int,
                                      C.POINTER(_u64p), C.POINTER(_u64p)]
        L.sort.argtypes = [C.POINTER(_u64p), C.POINTER(_u64p), _u64p, C.c_int, C.c_int, C.c_double,
                           C.POINTER(C.c_char_p), _u64p]
        L.mamalloc.restype = C.c_void_p
        L.mamalloc.argtypes = [C.c_size_t]


_ref = None


def have_ref() -> bool:
    try:
        build()
    except Exception:
        pass
    return os.path.exists(_REF_PATH)


def ref() -> _Ref:
    global _ref
    if _ref is None:
        _ref = _Ref()
    return _ref


def ref_schedule_passes(size: int, bits: int):
    rb = np.zeros(8, np.int8)
    bf = np.zeros(8, np.int8)
    p = ref().lib.schedule_passes(size, bits, _ptr(rb, _i8p), _ptr(bf, _i8p))
    return p, rb[: p + 1].tolist(), bf[: p + 1].tolist()


def ref_histogram(keys: np.ndarray, shift: int, radix_bits: int) -> np.ndarray:
    k = aligned(keys.size)
    k[:] = keys
    out = np.zeros(1 << radix_bits, np.uint64)
    ref().lib.histogram(_ptr(k, _u64p), k.size, _ptr(out, _u64p), shift, radix_bits)
    return out


def ref_partition(keys, rids, shift: int, radix_bits: int, buffered: bool):
    k = aligned(len(keys))
    r = aligned(len(keys))
    k[:] = keys
    r[:] = rids
    h = np.zeros(1 << radix_bits, np.uint64)
    ref().lib.histogram(_ptr(k, _u64p), k.size, _ptr(h, _u64p), shift, radix_bits)
    if buffered:
        ref().lib.partition_ip_buf(_ptr(k, _u64p), _ptr(r, _u64p), k.size, _ptr(h, _u64p), shift, radix_bits)
    else:
        offs = np.zeros_like(h)
        ref().lib.partition_ip(_ptr(k, _u64p), _ptr(r, _u64p), k.size, _ptr(h, _u64p), _ptr(offs, _u64p), shift, radix_bits)
    return np.array(k), np.array(r), h


def ref_sort_pairs_u64_inplace(k: np.ndarray, r: np.ndarray, bits: int) -> None:
    """The reference's single-thread core exactly as its driver calls it
    (src/msb_64.c:2232-2244) on 64-byte-aligned arrays, in place."""
    n = k.size
    if n == 0:
        return
    assert k.ctypes.data % 64 == 0 and r.ctypes.data % 64 == 0
    rb = np.zeros(8, np.int8)
    bf = np.zeros(8, np.int8)
    L = ref().lib
    p = L.schedule_passes(n, bits, _ptr(rb, _i8p), _ptr(bf, _i8p))
    while p > 0:
        p -= 1
        rb[p] += rb[p + 1]
    bufs = [aligned(4096) for _ in range(10)]
    H = (_u64p * 5)(*[_ptr(b, _u64p) for b in bufs[:5]])
    O = (_u64p * 5)(*[_ptr(b, _u64p) for b in bufs[5:]])
    L.local_radixsort(_ptr(k, _u64p), _ptr(r, _u64p), n, _ptr(rb, _i8p), _ptr(bf, _i8p), 0, H, O)


def ref_sort_pairs_u64(keys, rids, bits: int = 64):
    k = aligned(len(keys))
    r = aligned(len(keys))
    k[:] = keys
    r[:] = rids
    ref_sort_pairs_u64_inplace(k, r, bits)
    return np.array(k), np.array(r)


def ref_sort_u32(keys: np.ndarray) -> np.ndarray:
    k, _ = ref_sort_pairs_u64(keys.astype(np.uint64), keys.astype(np.uint64), 32)
    return k.astype(np.uint32)
