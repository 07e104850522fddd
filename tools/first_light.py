"""Development driver: sort a few inputs on the GPU, compare with the oracle, print phases."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from inplacemsdradixsort_amd import MsdContext
from oracle import oracle as O

ctx = MsdContext(0)
dev = torch.device("cuda:0")

def as_u32(t): return t.cpu().numpy().view(np.uint32)

def run(n, kind):
    if kind == "uniform": k = O.gen_uniform_u32(n)
    elif kind == "zipf": k = O.gen_zipf_u32(n)
    elif kind == "dup": k = (O.gen_uniform_u32(n) & 0xFF) * 0x01010101
    elif kind == "const": k = np.full(n, 0xDEADBEEF, np.uint32)
    elif kind == "sorted": k = np.sort(O.gen_uniform_u32(n))
    else: raise ValueError(kind)
    t = torch.from_numpy(k.view(np.int32)).to(dev)
    ctx.set_profiling(True)
    t0 = time.time()
    ctx.sort_u32(t)
    torch.cuda.synchronize()
    dt = time.time() - t0
    out = as_u32(t)
    exp = np.sort(k)
    ok = bool((out == exp).all())
    print(f"n={n:>10} {kind:8s} ok={ok} {dt*1e3:8.2f} ms stats={ctx.stats()}")
    if not ok:
        bad = np.nonzero(out != exp)[0]
        print("   first mismatches at", bad[:10], "count", bad.size, "multiset ok:", bool((np.sort(out) == exp).all()))
    return ok

if __name__ == "__main__":
    allok = True
    sizes = [1, 2, 63, 64, 65, 1000, 24576, 24577, 30000, 100000, 1 << 20, (1 << 20) + 13, 1 << 22, 3 * (1 << 22) + 5, 1 << 24]
    for n in sizes:
        for kind in ("uniform", "zipf", "dup", "const", "sorted"):
            try:
                allok &= run(n, kind)
            except Exception as e:
                print(f"n={n} {kind}: EXC {e}")
                allok = False
    print("phases(last):", ctx.phases())
    print("ALL OK" if allok else "FAILURES")
    sys.exit(0 if allok else 1)
