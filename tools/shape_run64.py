"""Development driver: u64 keys and (u64,u64) tuples on structured inputs."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from inplacemsdradixsort_amd import MsdContext
ctx = MsdContext(0)
logn = int(sys.argv[1]); n = 1 << logn
base = torch.empty(n, dtype=torch.int64, device="cuda")
ctx.reserve(n, 8, 8)
ctx.gen_uniform_u64(base)
srt = base.clone(); ctx.sort_u64(srt)
idx = torch.arange(n, device="cuda", dtype=torch.int64)
shapes = {
    "uniform": base, "sorted": srt, "reversed": torch.flip(srt, dims=[0]).contiguous(),
    "runs64k": (base & 0x00FFFFFFFFFFFFFF) | (((idx >> 16) & 0x7F) << 56),
    "few16": (base & 0xF) * 0x0101010101010101,
    "low16": base & 0xFFFF,
}
for mode in ("u64", "pairs"):
    for name, src in shapes.items():
        for rep in range(2):
            k = src.clone(); r = src.clone() if mode == "pairs" else None
            torch.cuda.synchronize()
            v0, s0, x0 = ctx.check(k)
            ctx.set_profiling(rep == 1)
            t0 = time.time()
            ctx.sort_u64(k) if r is None else ctx.sort_pairs_u64(k, r)
            torch.cuda.synchronize(); dt = time.time() - t0
            v, s, x = ctx.check(k, r)
        print(f"2^{logn} {mode:5s} {name:9s}: {dt*1e3:8.2f} ms {n/dt/1e9:6.2f} G/s viol={v} ok={s==s0 and x==x0}", flush=True)
        print("    ", {a: round(b) for a, b in ctx.phases()})
