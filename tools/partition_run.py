"""Development driver: time the one-digit partition (the pass before the multi-GPU exchange) at 2^LOGN."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from inplacemsdradixsort_amd import MsdContext
ctx = MsdContext(0)
n = 1 << int(sys.argv[1])
t = torch.empty(n, dtype=torch.int32, device="cuda")
ctx.reserve(n, 4, 0)
for rep in range(3):
    ctx.gen_uniform_u32(t, seed=9 + rep)
    torch.cuda.synchronize()
    ctx.set_profiling(rep == 2)
    t0 = time.time()
    cnt = ctx.partition(t, 24, 8)
    torch.cuda.synchronize()
    dt = time.time() - t0
d = (t.to(torch.int64) & 0xFFFFFFFF) >> 24
print(f"partition 2^{sys.argv[1]}: {dt*1e3:.2f} ms, grouped={bool((d[1:] >= d[:-1]).all())}, counts_ok={int(cnt.sum().item()) == n}", ctx.stats().get("direct_rounds"))
print({a: round(b) for a, b in ctx.phases()})
