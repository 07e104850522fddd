#!/usr/bin/env python3
"""Time one sort of 2^logn uniform u32 keys with an experimental build of the library
(inplacemsdradixsort_amd._build.build_variant / build_stamps) and print its phase table.

    python tools/variant_run.py <library suffix, e.g. "stamps" or "v1"> [logn] [u32|zipf|dup<k>|u64|pairs|pairs5b]
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from inplacemsdradixsort_amd import _build, _lib  # noqa: E402

name = sys.argv[1]
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 30
kind = sys.argv[3] if len(sys.argv) > 3 else "u32"
_build.LIB = os.path.join(_build.HERE, f"libinpmsdradix_hip_{name}.so") if name != "product" else _build.LIB
assert os.path.exists(_build.LIB), _build.LIB
_build.stale = lambda: False
from inplacemsdradixsort_amd import MsdContext  # noqa: E402

n = 1 << logn
ctx = MsdContext(0)
L = _lib.load(build_if_missing=False)
if kind == "u32":
    t = torch.empty(n, dtype=torch.int32, device="cuda")
    gen = lambda s: ctx.gen_uniform_u32(t, seed=0x5EED0001 + s)
    run = lambda: ctx.sort_u32(t)
elif kind == "zipf":
    t = torch.empty(n, dtype=torch.int32, device="cuda")
    gen = lambda s: ctx.gen_zipf_u32(t, seed=0x5EED0003 + s)
    run = lambda: ctx.sort_u32(t)
elif kind.startswith("dup"):  # dup<distinct>: that many distinct values
    t = torch.empty(n, dtype=torch.int32, device="cuda")
    gen = lambda s: ctx.gen_dup_u32(t, int(kind[3:]), seed=0x5EED0004 + s)
    run = lambda: ctx.sort_u32(t)
elif kind == "u64":
    t = torch.empty(n, dtype=torch.int64, device="cuda")
    gen = lambda s: ctx.gen_uniform_u64(t, seed=0x5EED0005 + s)
    run = lambda: ctx.sort_u64(t)
else:
    t = torch.empty(n, dtype=torch.int64, device="cuda")
    r = torch.empty(n, dtype=torch.int64, device="cuda")
    def gen(s):
        ctx.gen_uniform_u64(t, seed=0x5EED0005 + s, shift_right=32 if kind == "pairs5b" else 0)
        ctx.gen_iota_u64(r)
    run = lambda: ctx.sort_pairs_u64(t, r)
times = []
for it in range(4):
    gen(it)
    c0 = ctx.check(t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx.use_torch_stream()
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1))
    c1 = ctx.check(t)
    assert os.environ.get("MSD_VARIANT_NOCHECK") or (c1[0] == 0 and c1[1:] == c0[1:]), (c0, c1)  # (timing-only experiments)
gen(9)
ctx.set_profiling(True)
run()
torch.cuda.synchronize()
out = {"lib": name, "kind": kind, "logn": logn, "ms": [round(x, 3) for x in times], "phases_us": {k: round(v, 1) for k, v in ctx.phases()},
       "stats": ctx.stats()}
if hasattr(L, "msd_debug_stamps"):
    NAMES = ["scatter", "B1", "bookkeeping", "B2", "flush", "select", "waiting-keys", "B3", "-", "refill+loop", "epilogue", "tiles"]
    if name.startswith("lstamps"):  # leaf_count_sort_kernel sections
        NAMES = ["keys wait+OR/AND+clear", "B+merge+B", "fetch-adds", "B+sums+scan+B+prefix", "B+scatter", "descriptor+prefetch", "B+fix-up",
                 "write-back", "-", "loop", "-", "segments"]
    if name.startswith("l17stamps"):  # leaf17_kernel sections
        NAMES = ["keys arrive+OR/AND+clear+2B", "fetch-adds", "scan (3B)", "keys into LDS", "long-group check", "fix-up", "payload loads + keys out", "payloads out", "-", "loop", "-", "segments"]
    if name.startswith("cstamps"):  # count_place_kernel sections
        NAMES = ["clear", "B", "fetch-adds", "B", "byte sums", "B+block scan", "prefix", "positions", "B+LDS out+B", "loop", "prefetch+store", "segments"]
    if name.startswith("sstamps"):  # classify_kernel (streaming) sections
        NAMES = ["places", "B1", "bucket round", "B2", "waiting keys' places + flush", "B3", "waiting keys", "refill issue", "-", "loop", "-", "tiles"]  # classify_stream2_kernel
    if name.startswith("chstamps"):  # chains_kernel sections (per wave step); 10: blocks moved per step
        NAMES = ["chain starts", "owner flags+grouping", "claim", "list geometry+entry", "block loads+stores", "-", "-", "-", "-", "-", "blocks moved", "steps"]
    if name.startswith("wstamps"):  # bigcount_write_kernel sections
        NAMES = ["look-ups", "tile inside one run", "2 barriers", "runs of the tile", "B+long runs", "B", "LDS->array", "-", "-", "-", "fast tiles", "tiles"]
    if name.startswith("hstamps"):  # bigcount_hist_kernel sections
        NAMES = ["chunk set-up", "load issue", "counting", "merge", "-", "-", "-", "-", "-", "-", "-", "chunks"]
    L.msd_debug_stamps.argtypes = [C.POINTER(C.c_uint64)]
    buf = (C.c_uint64 * 32)()
    ctx.set_profiling(False)
    gen(10)
    torch.cuda.synchronize()
    L.msd_debug_stamps(buf)
    run()
    torch.cuda.synchronize()
    L.msd_debug_stamps(buf)
    for w, label in ((0, "wave0"), (1, "last_wave")):
        v = [int(buf[w * 16 + i]) for i in range(12)]
        tiles = max(1, v[11])
        out[label] = {f"{i}:{NAMES[i]}": round(v[i] / tiles, 1 if v[i] / tiles < 1e4 else 0) for i in range(11) if NAMES[i] != "-"}
        out[label]["cycles_per_tile"] = round(sum(v[:11]) / tiles, 1)
        out[label]["tiles"] = v[11]
print(json.dumps(out))
