#!/usr/bin/env python3
"""A/B of one msd_set_option knob on one workload, inside one process (same box, same buffers).

    python tools/ab_option.py <u32|zipf|dup<k>|u64|pairs|pairs5b> <logn> <option> <value> [<value> ...]

Prints one JSON line per value: best-of-5 ms, the phase table and the stats of a profiled run; every result is checked
(order, sum, xor) against the input's checksums.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from inplacemsdradixsort_amd import MsdContext  # noqa: E402

kind, logn, opt = sys.argv[1], int(sys.argv[2]), sys.argv[3]
values = [int(v) for v in sys.argv[4:]]
n = 1 << logn
ctx = MsdContext(0)
ctx.use_torch_stream()
wide = kind in ("u64", "pairs", "pairs5b")
t = torch.empty(n, dtype=torch.int64 if wide else torch.int32, device="cuda")
r = torch.empty(n, dtype=torch.int64, device="cuda") if kind.startswith("pairs") else None


def gen(s):
    if kind == "u32":
        ctx.gen_uniform_u32(t, seed=0x5EED0001 + s)
    elif kind == "zipf":
        ctx.gen_zipf_u32(t, seed=0x5EED0003 + s)
    elif kind.startswith("dup"):
        ctx.gen_dup_u32(t, int(kind[3:]), seed=0x5EED0004 + s)
    else:
        ctx.gen_uniform_u64(t, seed=0x5EED0005 + s, shift_right=32 if kind == "pairs5b" else 0)
        if r is not None:
            r.copy_(t)


def run():
    if r is not None:
        ctx.sort_pairs_u64(t, r)
    elif wide:
        ctx.sort_u64(t)
    else:
        ctx.sort_u32(t)


ctx.reserve(n + n // 8, 8 if wide else 4, 8 if r is not None else 0)
for v in values:
    ctx.set_option(opt, v)
    ms = []
    for it in range(6):
        gen(it)
        c0 = ctx.check(t)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run()
        e1.record()
        torch.cuda.synchronize()
        if it:
            ms.append(e0.elapsed_time(e1))
        c1 = ctx.check(t, r) if r is not None else ctx.check(t)
        assert c1[0] == 0 and c1[1:] == c0[1:], (c0, c1)
    gen(9)
    ctx.set_profiling(True)
    run()
    torch.cuda.synchronize()
    ctx.set_profiling(False)
    print(json.dumps({"kind": kind, "logn": logn, opt: v, "ms_best": round(min(ms), 3), "ms": [round(x, 3) for x in ms],
                      "phases_us": {k: round(x, 1) for k, x in ctx.phases()}, "stats": ctx.stats()}), flush=True)
