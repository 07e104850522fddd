#!/usr/bin/env python3
"""What one rank of the N-GPU sort does per step, timed on ONE GPU (RCCL cannot run here): the top-digit pass before
the exchange, then -- on keys laid out as they arrive, per source rank that source's buckets of the rank's range --
the run gather and the segmented sort, next to the sort on 32 - log2(N) bits they replace.

    python tools/multigpu_local_work.py [ranks=8] [logn=30]     # one JSON line
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
if os.environ.get("MSD_VARIANT"):  # an experimental build (inplacemsdradixsort_amd._build.build_variant)
    from inplacemsdradixsort_amd import _build
    _build.LIB = os.path.join(_build.HERE, f"libinpmsdradix_hip_{os.environ['MSD_VARIANT']}.so")
    _build.stale = lambda: False
from inplacemsdradixsort_amd import MsdContext  # noqa: E402
from inplacemsdradixsort_amd.dist import bucket_major  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n, per = 1 << logn, 256 // G
ctx = MsdContext(0)
ctx.use_torch_stream()
keys = torch.empty(n, dtype=torch.int32, device="cuda")
arrived = torch.empty(n, dtype=torch.int32, device="cuda")
work = torch.empty(n, dtype=torch.int32, device="cuda")


def timed(f):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


res = {"ranks": G, "keys_per_rank": n, "ms": {}}
for it in range(3):
    # keys of ONE rank's range (top log2(G) bits fixed) in the order the all-to-all delivers them: G sources, each with
    # its `per` buckets in order.  Made from a top-digit pass over the rank's own uniform keys, cut into G parts per bucket.
    ctx.gen_uniform_u32(keys, seed=it)
    keys &= (1 << (32 - (G.bit_length() - 1))) - 1 if G > 1 else -1
    t_part = timed(lambda: ctx.partition(keys, 24, 8))
    cnt = ctx.partition(keys, 24, 8).cpu().numpy()[:per]           # (idempotent: the keys are partitioned already)
    mine = [[int(c) // G + (int(c) % G if s == G - 1 else 0) for c in cnt] for s in range(G)]
    src_off, dst_off, lens, seg_off = bucket_major(mine)
    ctx.gather_runs(arrived, keys, dst_off, src_off, lens)          # bucket-major -> source-major: "as it arrived"
    c0 = ctx.check(keys)
    t_gather = timed(lambda: ctx.gather_runs(work, arrived, src_off, dst_off, lens))
    t_seg = timed(lambda: ctx.sort_segments(work, seg_off, 24))
    t_legacy = timed(lambda: ctx.sort_u32(arrived, end_bit=32 - (G.bit_length() - 1)))
    c1, c2 = ctx.check(work), ctx.check(arrived)
    assert c1[0] == 0 and c1[1:] == c0[1:] and c2[0] == 0 and c2[1:] == c0[1:], (c0, c1, c2)
    res["ms"] = {"top-digit pass before the exchange": round(t_part, 3), "gather runs bucket-major": round(t_gather, 3),
                 "segmented sort on 24 bits": round(t_seg, 3), "replaced: sort on %d bits" % (32 - (G.bit_length() - 1)): round(t_legacy, 3)}
res["local_ms_per_step"] = round(sum(v for k, v in res["ms"].items() if not k.startswith("replaced")), 3)
res["local_ms_per_step_before"] = round(res["ms"]["top-digit pass before the exchange"] + [v for k, v in res["ms"].items() if k.startswith("replaced")][0], 3)
print(json.dumps(res))
