"""Profiling driver (run under rocprofv3): a few device-resident sorts at 2^LOGN."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from inplacemsdradixsort_amd import MsdContext
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 28
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kind = sys.argv[3] if len(sys.argv) > 3 else "uniform"
ctx = MsdContext(0)
n = 1 << logn
t = torch.empty(n, dtype=torch.int32, device="cuda")
ctx.reserve(n, 4, 0)
for r in range(reps):
    (ctx.gen_uniform_u32 if kind == "uniform" else ctx.gen_zipf_u32)(t, seed=0x5EED0001 + r)
    ctx.sort_u32(t)
torch.cuda.synchronize()
print(ctx.check(t)[0], ctx.stats())
