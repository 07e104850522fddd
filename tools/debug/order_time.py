"""Times msd_order_low16_u32 with an experimental build: python tools/debug/order_time.py <library suffix | product> [logn]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inplacemsdradixsort_amd import _build
name = sys.argv[1]
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 30
if name != "product":
    _build.LIB = os.path.join(_build.HERE, f"libinpmsdradix_hip_{name}.so")
    _build.stale = lambda: False
from inplacemsdradixsort_amd import MsdContext
n = 1 << logn
ctx = MsdContext(0)
ctx.use_torch_stream()
ctx.reserve(n, 4, 0)
t = torch.empty(n, dtype=torch.int32, device="cuda")
out = torch.empty(n, dtype=torch.int16, device="cuda")
best = 1e9
for it in range(5):
    ctx.gen_uniform_u32(t, seed=it)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ctx.order_low16(t, out)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print(name, f"{best:.3f} ms")
