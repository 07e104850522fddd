"""Development driver: time u64-key and (u64 key, u64 rid) pair sorts at large sizes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from inplacemsdradixsort_amd import MsdContext

ctx = MsdContext(0)
logns = [int(x) for x in sys.argv[1].split(",")]
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["u64", "pairs", "pairs32"]
for logn in logns:
    n = 1 << logn
    for mode in modes:
        shr = 32 if mode.endswith("32") else 0
        k = torch.empty(n, dtype=torch.int64, device="cuda")
        r = torch.empty(n, dtype=torch.int64, device="cuda") if mode.startswith("pairs") else None
        ctx.reserve(n, 8, 8 if r is not None else 0)
        for rep in range(3):
            ctx.gen_uniform_u64(k, shift_right=shr)
            if r is not None:
                r.copy_(k)
            torch.cuda.synchronize()
            v0, s0, x0 = ctx.check(k)
            ctx.set_profiling(rep == 2)
            t0 = time.time()
            if r is None:
                ctx.sort_u64(k)
            else:
                ctx.sort_pairs_u64(k, r)
            torch.cuda.synchronize()
            dt = time.time() - t0
            v, s, x = ctx.check(k, r)
            print(f"2^{logn} {mode} rep{rep}: {dt*1e3:.2f} ms  {n/dt/1e9:.2f} Gelem/s  viol={v} sum_ok={s==s0} xor_ok={x==x0}", flush=True)
        print("  stats", ctx.stats())
        ph = ctx.phases()
        print("  phases(us):", {a: round(b) for a, b in ph}, "total", round(sum(b for _, b in ph)))
        del k, r
        torch.cuda.empty_cache()
