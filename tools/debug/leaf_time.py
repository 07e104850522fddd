"""Times the counting leaf over low halves laid out as they arrive at G ranks with an experimental build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inplacemsdradixsort_amd import _build
name, G = sys.argv[1], int(sys.argv[2])
if name != "product":
    _build.LIB = os.path.join(_build.HERE, f"libinpmsdradix_hip_{name}.so")
    _build.stale = lambda: False
from inplacemsdradixsort_amd import MsdContext
n = 1 << 30
lg = G.bit_length() - 1
ctx = MsdContext(0)
ctx.use_torch_stream()
keys = torch.empty(n, dtype=torch.int32, device="cuda")
low = torch.empty(n + 64, dtype=torch.int16, device="cuda")
work = torch.empty(n, dtype=torch.int32, device="cuda")
ctx.gen_uniform_u32(keys, seed=1)
keys &= (1 << (32 - lg)) - 1
chunk, nbl = n // G, 65536 // G
rows, base = [], []
for s in range(G):
    part = keys[s * chunk:(s + 1) * chunk]
    c = ctx.order_low16(part, low[s * chunk:(s + 1) * chunk])
    rows.append(c[:nbl].clone())
    base.append(s * chunk)
counts = torch.stack(rows).contiguous()
best = 1e9
for it in range(4):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ctx.merge_buckets(low, counts, base, 16, 0, work, n)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
assert ctx.check(work)[0] == 0
print(name, G, f"{best:.3f} ms")
import ctypes as C
from inplacemsdradixsort_amd import _lib
L = _lib.load(build_if_missing=False)
if hasattr(L, "msd_debug_stamps"):
    NAMES = ["clear+ticket", "B", "count", "B", "scan", "segment search", "output (per-wave segments)", "-", "-", "loop", "-", "buckets"]
    L.msd_debug_stamps.argtypes = [C.POINTER(C.c_uint64)]
    buf = (C.c_uint64 * 32)()
    L.msd_debug_stamps(buf)
    ctx.merge_buckets(low, counts, base, 16, 0, work, n)
    torch.cuda.synchronize()
    L.msd_debug_stamps(buf)
    for wv, label in ((0, "wave0"), (1, "last_wave")):
        v = [int(buf[wv * 16 + i]) for i in range(12)]
        bk = max(1, v[11])
        print(label, {f"{i}:{NAMES[i]}": round(v[i] / bk) for i in range(11) if v[i]}, "cycles per bucket", round(sum(v[:11]) / bk), "buckets", v[11])
