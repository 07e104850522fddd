import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inplacemsdradixsort_amd import MsdContext
from inplacemsdradixsort_amd.api import MsdError
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << logn
ctx = MsdContext(0)
ctx.reserve(n + n // 8, 4, 0)
t = torch.empty(n, dtype=torch.int32, device="cuda")
out = torch.empty(n, dtype=torch.int16, device="cuda")
for rank in range(4):
    for s in list(range(0, 4)) + list(range(300, 304)):
        for what in ("top16", "top24", "order", "sort"):
            ctx.gen_uniform_u32(t, seed=0x5EED0001 + 1000003 * s, first=rank * n)
            try:
                if what == "top16":
                    ctx.sort_top(t, 16)
                elif what == "top24":
                    ctx.sort_top(t, 24)
                elif what == "order":
                    ctx.order_low16(t, out)
                else:
                    ctx.sort_u32(t)
            except MsdError as e:
                print("FAIL rank", rank, "s", s, what, e, flush=True)
torch.cuda.synchronize()
print("done")
