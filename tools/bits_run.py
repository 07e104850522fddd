"""Development driver: local sort of keys with `end_bit` open bits (what a rank sorts after the multi-GPU exchange)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from inplacemsdradixsort_amd import MsdContext

ctx = MsdContext(0)
logn = int(sys.argv[1]); bits = [int(b) for b in sys.argv[2].split(",")]
n = 1 << logn
t = torch.empty(n, dtype=torch.int32, device="cuda")
ctx.reserve(n, 4, 0)
for eb in bits:
    for rep in range(3):
        ctx.gen_uniform_u32(t, seed=5 + rep)
        if eb < 32:
            t.bitwise_and_((1 << eb) - 1)
        torch.cuda.synchronize()
        v0, s0, x0 = ctx.check(t)
        ctx.set_profiling(rep == 2)
        t0 = time.time()
        ctx.sort_u32(t, end_bit=eb)
        torch.cuda.synchronize()
        dt = time.time() - t0
        v, s, x = ctx.check(t)
    print(f"2^{logn} end_bit={eb}: {dt*1e3:.2f} ms {n/dt/1e9:.1f} Gkeys/s viol={v} ok={s==s0 and x==x0}")
    print("  stats", {k: ctx.stats()[k] for k in ("rounds", "parents", "children", "direct_rounds") if k in ctx.stats()})
    print("  phases(us):", {a: round(b) for a, b in ctx.phases()})
