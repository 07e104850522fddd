"""Does an asynchronous (chunked) all-to-all on the nccl backend run concurrently with this library's kernels on the compute
stream?  One rank exchanging 4 GiB with itself + one 2^30-key sort, apart and together."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from inplacemsdradixsort_amd import MsdContext
from inplacemsdradixsort_amd.dist import all_to_all_v
ctx = MsdContext(0); ctx.use_torch_stream()
n = 1 << 30
a = torch.empty(n, dtype=torch.int32, device="cuda"); ctx.gen_uniform_u32(a, seed=1)
b = torch.empty(n, dtype=torch.int32, device="cuda")
k = torch.empty(n, dtype=torch.int32, device="cuda")
ctx.reserve(n + n // 8, 4, 0)
def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        ctx.gen_uniform_u32(k, seed=5)
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return round(best * 1e3, 3)
def xch():
    for h in all_to_all_v(dist, b, a, [n], [n], async_op=True): h.wait()
def both():
    hs = all_to_all_v(dist, b, a, [n], [n], async_op=True)
    ctx.sort_u32(k)
    for h in hs: h.wait()
print({"exchange alone ms": t(xch), "sort alone ms": t(lambda: ctx.sort_u32(k)), "exchange started, then sort, then wait ms": t(both)}, flush=True)
dist.destroy_process_group()
