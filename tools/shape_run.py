"""Development driver: time the sort on structured inputs (sorted, reversed, per-stripe runs) at 2^LOGN."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from inplacemsdradixsort_amd import MsdContext

ctx = MsdContext(0)
logn = int(sys.argv[1])
n = 1 << logn
base = torch.empty(n, dtype=torch.int32, device="cuda")
ctx.reserve(n, 4, 0)
ctx.gen_uniform_u32(base)
srt = base.clone(); ctx.sort_u32(srt)
shapes = {
    "uniform": lambda: base.clone(),
    "sorted": lambda: srt.clone(),
    "reversed": lambda: torch.flip(srt, dims=[0]).contiguous(),
    # runs of 2^16 keys with the same top byte, top byte cycling: every stripe holds few buckets
    "runs64k": lambda: (base & 0x00FFFFFF) | (((torch.arange(n, device="cuda", dtype=torch.int64) >> 16) & 0xFF) << 24).to(torch.int32),
    # sorted within blocks of 2^20, blocks shuffled (locally sorted data)
    "blocksorted": lambda: base.view(-1, 1 << 20).sort(dim=1).values.contiguous().view(-1),
    "const": lambda: torch.full_like(base, 0x12345678),
    "two": lambda: torch.where((base & 1) == 0, torch.full_like(base, 7), torch.full_like(base, -9)),
    "few16": lambda: (base & 0xF) * 0x01010101,
    "low8": lambda: base & 0xFF,
    "high8": lambda: base & (-0x1000000),
    "heavy50": lambda: torch.where((base & 1) == 0, torch.full_like(base, 0x5A5A5A5A), base),
    "mid16": lambda: base & 0x00FFFF00,
}
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
if only:
    shapes = {k: v for k, v in shapes.items() if k in only}
for name, mk in shapes.items():
    for rep in range(2):
        t = mk()
        torch.cuda.synchronize()
        v0, s0, x0 = ctx.check(t)
        ctx.set_profiling(rep == 1)
        t0 = time.time(); ctx.sort_u32(t); torch.cuda.synchronize(); dt = time.time() - t0
        v, s, x = ctx.check(t)
    st = ctx.stats()
    print(f"2^{logn} {name:12s}: {dt*1e3:7.2f} ms {n/dt/1e9:6.1f} Gkeys/s viol={v} ok={s==s0 and x==x0} direct_rounds={st.get('direct_rounds',0)} chain_steps={st.get('chain_steps')}", flush=True)
    print("   ", {a: round(b) for a, b in ctx.phases()})
    del t
