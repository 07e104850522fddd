"""Looks for data-dependent failures of msd_order_low16_u32 / msd_sort_u32_top(begin_bit = 24): many seeds at one size, one process."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from inplacemsdradixsort_amd import MsdContext  # noqa: E402
from inplacemsdradixsort_amd.api import MsdError  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 28
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 500
n = 1 << logn
ctx = MsdContext(0)
ctx.reserve(n + n // 8, 4, 0)
t = torch.empty(n, dtype=torch.int32, device="cuda")
out = torch.empty(n, dtype=torch.int16, device="cuda")
bad = 0
for i in range(iters):
    rank = i % 4
    ctx.gen_uniform_u32(t, seed=0x5EED0001 + 300 + i // 4, first=rank * n)
    try:
        if i % 2:
            ctx.order_low16(t, out)
        else:
            ctx.sort_top(t, 24)
    except MsdError as e:
        bad += 1
        print("iteration", i, "rank", rank, e, flush=True)
torch.cuda.synchronize()
print("done", iters, "failures", bad)
