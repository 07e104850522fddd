"""Per-kernel cost per key against the size of the array: do the second round and the leaf run faster when their data fits
the 256 MiB Infinity Cache?  Emulates "what is left after round 0 of a 2^30-key sort" for 2^logn keys: segments of 2^22 keys
sorted on their low 24 bits (one direct round on bits 16..23 with exact counts, then the 16-bit counting leaf).  Run under
rocprofv3, one process per size:
    rocprofv3 --kernel-trace --stats -d gpurun_out/cp25 -- python3 tools/cache_probe.py --logn 25
(tools/cache_probe_table.py reads the stats files)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import inplacemsdradixsort_amd as M  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--logn", type=int, default=25)
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--logseg", type=int, default=22)
ap.add_argument("--whole", action="store_true", help="a whole sort of 2^logn keys instead")
a = ap.parse_args()
n = 1 << a.logn
ctx = M.MsdContext()
buf = torch.empty(n, dtype=torch.int32, device="cuda")
seg = [i << a.logseg for i in range((n >> a.logseg) + 1)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
best = 1e9
for r in range(a.reps):
    ctx.gen_uniform_u32(buf, seed=r + 1)
    ev[0].record()
    if a.whole:
        ctx.sort_u32(buf)
    else:
        ctx.sort_segments(buf, seg, 24)
    ev[1].record()
    torch.cuda.synchronize()
    best = min(best, ev[0].elapsed_time(ev[1]))
if a.whole:
    assert ctx.check(buf)[0] == 0
else:
    low = (buf[: 1 << a.logseg] & 0xFFFFFF)
    assert bool((low[1:] >= low[:-1]).all())
print(f"logn {a.logn}: {best:.3f} ms = {best * 1e6 / n:.4f} ns/key (host-inclusive)", {k: v for k, v in ctx.stats().items() if "round" in k or "leaf" in k or "count" in k})
