"""msd_order_low16_u32 on large inputs of several shapes, checked on the device with torch (multiset of (bucket, low half))."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inplacemsdradixsort_amd import MsdContext
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << logn
ctx = MsdContext(0)
ctx.use_torch_stream()
t = torch.empty(n, dtype=torch.int32, device="cuda")
out = torch.empty(n, dtype=torch.int16, device="cuda")
for kind in ("uniform", "zipf", "dup1000", "narrow", "sorted"):
    if kind == "uniform":
        ctx.gen_uniform_u32(t, seed=3)
    elif kind == "zipf":
        ctx.gen_zipf_u32(t, seed=4)
    elif kind == "dup1000":
        ctx.gen_dup_u32(t, 1000, seed=5)
    elif kind == "narrow":
        ctx.gen_uniform_u32(t, seed=6)
        t &= 0x000FFFFF
    else:
        ctx.gen_uniform_u32(t, seed=7)
        ctx.sort_u32(t)
    ref = torch.sort(t.to(torch.int64) & 0xFFFFFFFF).values
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    counts = ctx.order_low16(t, out)
    e1.record()
    torch.cuda.synchronize()
    assert int(counts.sum()) == n
    bucket = torch.repeat_interleave(torch.arange(65536, device="cuda", dtype=torch.int64), counts)
    got = torch.sort((bucket << 16) | (out.to(torch.int64) & 0xFFFF)).values
    ok = bool((got == ref).all())
    print(kind, "ok" if ok else "MISMATCH", f"{e0.elapsed_time(e1):.2f} ms", flush=True)
    assert ok
    del ref, bucket, got
