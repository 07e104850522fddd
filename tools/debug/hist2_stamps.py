import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inplacemsdradixsort_amd import _build
_build.LIB = os.path.join(_build.HERE, "libinpmsdradix_hip_h2stamps.so")
_build.stale = lambda: False
from inplacemsdradixsort_amd import MsdContext, _lib
n = 1 << 30
ctx = MsdContext(0)
ctx.use_torch_stream()
L = _lib.load(build_if_missing=False)
t = torch.empty(n, dtype=torch.int32, device="cuda")
low = torch.empty(n, dtype=torch.int16, device="cuda")
rec = torch.empty(65536 * ctx.HIST2_RECORD_BYTES, dtype=torch.uint8, device="cuda")
ctx.gen_uniform_u32(t, seed=1)
c = ctx.order_low16(t, low)
bnd = ctx.bounds_from_counts16(c)
ctx.hist2_pack(low, bnd, rec)
torch.cuda.synchronize()
L.msd_debug_stamps.argtypes = [C.POINTER(C.c_uint64)]
buf = (C.c_uint64 * 32)()
L.msd_debug_stamps(buf)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ctx.hist2_pack(low, bnd, rec)
e1.record()
torch.cuda.synchronize()
L.msd_debug_stamps(buf)
print("%.3f ms" % e0.elapsed_time(e1))
NAMES = ["clear", "barrier", "wait for the keys", "fetch-adds", "next loads issued", "barrier", "pack + fields out", "barrier", "entries out + barrier", "loop", "-", "buckets"]
for wv, label in ((0, "wave0"), (1, "last_wave")):
    v = [int(buf[wv * 16 + i]) for i in range(12)]
    tl = max(1, v[11])
    print(label, {f"{i}:{NAMES[i]}": round(v[i] / tl) for i in range(11) if v[i]}, "cycles per bucket", round(sum(v[:11]) / tl), "buckets", v[11])
