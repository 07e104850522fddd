import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inplacemsdradixsort_amd import MsdContext
n = 1 << 30
ctx = MsdContext(0)
ctx.use_torch_stream()
ctx.reserve(n, 4, 0)
t = torch.empty(n, dtype=torch.int32, device="cuda")
out = torch.empty(n, dtype=torch.int16, device="cuda")
for it in range(3):
    ctx.gen_uniform_u32(t, seed=it)
    ctx.set_profiling(it == 2)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    c = ctx.order_low16_counts(t)
    e[1].record()
    ctx.order_low16_scatter(t, out)
    e[2].record()
    torch.cuda.synchronize()
    print("counts half %.3f ms, scatter %.3f ms" % (e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
print({k: round(v) for k, v in ctx.phases()})
