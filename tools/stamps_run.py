#!/usr/bin/env python3
"""Per-section shader cycles of classify_direct_kernel from the -DMSD_STAMPS diagnostic build
(inplacemsdradixsort_amd._build.build_stamps(); the stamps cost about 10 % themselves).

    python tools/stamps_run.py [logn]
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from inplacemsdradixsort_amd import _build, _lib  # noqa: E402

_build.LIB = _build.STAMPS_LIB  # load the diagnostic library instead of the product
assert os.path.exists(_build.LIB), "build it first: python -c 'from inplacemsdradixsort_amd import _build; _build.build_stamps()'"
from inplacemsdradixsort_amd import MsdContext  # noqa: E402

NAMES = ["ranks", "B1", "bookkeeping", "B2a", "steal+B2b", "select", "scatter", "B3", "vmcnt0", "load+flush", "epilogue", "tiles"]
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << logn
ctx = MsdContext(0)
L = _lib.load(build_if_missing=False)
L.msd_debug_stamps.argtypes = [C.POINTER(C.c_uint64)]
t = torch.empty(n, dtype=torch.int32, device="cuda")
buf = (C.c_uint64 * 32)()
for it in range(2):
    ctx.gen_uniform_u32(t, seed=0x5EED0001 + it)
    torch.cuda.synchronize()
    L.msd_debug_stamps(buf)  # reset
    ctx.sort_u32(t)
    torch.cuda.synchronize()
    L.msd_debug_stamps(buf)
out = {}
for w, label in ((0, "wave0(bucket wave)"), (1, "last wave")):
    v = [int(buf[w * 16 + i]) for i in range(12)]
    tiles = max(1, v[11])
    out[label] = {NAMES[i]: round(v[i] / tiles, 1) for i in range(11)}
    out[label]["tiles_per_wave_total"] = v[11]
    out[label]["cycles_per_tile"] = round(sum(v[:10]) / tiles, 1)
print(json.dumps(out, indent=1))
