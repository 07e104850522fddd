import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inplacemsdradixsort_amd import _build
_build.LIB = os.path.join(_build.HERE, "libinpmsdradix_hip_s16stamps.so")
_build.stale = lambda: False
from inplacemsdradixsort_amd import MsdContext, _lib
n = 1 << 30
ctx = MsdContext(0)
L = _lib.load(build_if_missing=False)
t = torch.empty(n, dtype=torch.int32, device="cuda")
out = torch.empty(n, dtype=torch.int16, device="cuda")
ctx.gen_uniform_u32(t, seed=1)
ctx.order_low16(t, out)
torch.cuda.synchronize()
L.msd_debug_stamps.argtypes = [C.POINTER(C.c_uint64)]
buf = (C.c_uint64 * 32)()
L.msd_debug_stamps(buf)
ctx.gen_uniform_u32(t, seed=2)
ctx.order_low16(t, out)
torch.cuda.synchronize()
L.msd_debug_stamps(buf)
NAMES = ["wait for the tile's keys", "fetch-adds + ring stores", "next loads issued", "barrier", "half-rings out", "barrier", "-", "-", "-", "loop", "-", "tiles"]
for wv, label in ((0, "wave0"), (1, "last_wave")):
    v = [int(buf[wv * 16 + i]) for i in range(12)]
    tl = max(1, v[11])
    print(label, {f"{i}:{NAMES[i]}": round(v[i] / tl) for i in range(11) if v[i]}, "cycles per tile", round(sum(v[:11]) / tl), "tiles", v[11])
