#!/usr/bin/env python3
"""Sort time over input sizes (uniform keys; KIND=u32|u64|pairs, LOGN_LO / LOGN_HI), default options and with a lowered direct_min: where does direct
placement start to pay?    python tools/size_sweep.py [direct_min log2 ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
if os.environ.get("MSD_VARIANT"):  # an experimental build (inplacemsdradixsort_amd._build.build_variant)
    from inplacemsdradixsort_amd import _build
    _build.LIB = os.path.join(_build.HERE, f"libinpmsdradix_hip_{os.environ['MSD_VARIANT']}.so")
    _build.stale = lambda: False
from inplacemsdradixsort_amd import MsdContext  # noqa: E402

ctx = MsdContext(0)
ctx.use_torch_stream()
mins = [int(a) for a in sys.argv[1:]] or [26]
for logn in range(int(os.environ.get('LOGN_LO', 18)), int(os.environ.get('LOGN_HI', 28)) + 1):
    n = 1 << logn
    kind = os.environ.get("KIND", "u32")
    t = torch.empty(n, dtype=torch.int32 if kind == "u32" else torch.int64, device="cuda")
    r = torch.empty(n, dtype=torch.int64, device="cuda") if kind == "pairs" else None
    row = {"logn": logn, "kind": kind}
    for m in mins:
        ctx.set_option("direct_min", 1 << m)
        best = 1e9
        for it in range(5):
            if kind == "u32":
                ctx.gen_uniform_u32(t, seed=it)
            else:
                ctx.gen_uniform_u64(t, seed=it)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if kind == "u32":
                ctx.sort_u32(t)
            elif kind == "u64":
                ctx.sort_u64(t)
            else:
                ctx.sort_pairs_u64(t, r)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        assert ctx.check(t)[0] == 0
        row[f"direct_min=2^{m}"] = {"ms": round(best, 4), "Gkeys/s": round(n / best / 1e6, 2), "direct_rounds": ctx.stats().get("direct_rounds", 0)}
    print(json.dumps(row))
