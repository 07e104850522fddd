"""Profiling driver (run under rocprofv3): sorts of one structured input at 2^LOGN."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from inplacemsdradixsort_amd import MsdContext
logn = int(sys.argv[1]); kind = sys.argv[2]
ctx = MsdContext(0)
n = 1 << logn
base = torch.empty(n, dtype=torch.int32, device="cuda")
ctx.reserve(n, 4, 0)
ctx.gen_uniform_u32(base)
if kind in ("sorted", "reversed"):
    ctx.sort_u32(base)
    if kind == "reversed":
        base = torch.flip(base, dims=[0]).contiguous()
elif kind == "runs64k":
    base = (base & 0x00FFFFFF) | (((torch.arange(n, device="cuda", dtype=torch.int64) >> 16) & 0xFF) << 24).to(torch.int32)
for r in range(2):
    t = base.clone()
    ctx.sort_u32(t)
torch.cuda.synchronize()
print(ctx.check(t)[0], ctx.stats())
