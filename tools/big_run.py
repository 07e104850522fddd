"""Development driver: time the device-resident sort at large sizes with the per-phase report."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from inplacemsdradixsort_amd import MsdContext

ctx = MsdContext(0)
dev = torch.device("cuda:0")
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["uniform"]
for logn in [int(x) for x in sys.argv[1].split(",")]:
    n = 1 << logn
    for kind in kinds:
        t = torch.empty(n, dtype=torch.int32, device=dev)
        ctx.reserve(n, 4, 0)
        def gen():
            if kind == "uniform": ctx.gen_uniform_u32(t)
            elif kind == "zipf": ctx.gen_zipf_u32(t)
        gen(); torch.cuda.synchronize()
        v0, s0, x0 = ctx.check(t)
        for rep in range(3):
            gen(); torch.cuda.synchronize()
            ctx.set_profiling(rep == 2)
            t0 = time.time()
            ctx.sort_u32(t)
            torch.cuda.synchronize()
            dt = time.time() - t0
            v, s, x = ctx.check(t)
            print(f"2^{logn} {kind} rep{rep}: {dt*1e3:.2f} ms  {n/dt/1e9:.2f} Gkeys/s  viol={v} sum_ok={s==s0} xor_ok={x==x0}", flush=True)
        print("  stats", ctx.stats())
        ph = ctx.phases()
        print("  phases(us):", {k: round(v) for k, v in ph}, "total", round(sum(v for _, v in ph)))
        del t
