"""Which pairs of (side-stream work, compute-stream work) overlap on this device?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29579")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from inplacemsdradixsort_amd import MsdContext
from inplacemsdradixsort_amd.dist import all_to_all_v
ctx = MsdContext(0); ctx.use_torch_stream()
n = 1 << 30
a = torch.empty(n, dtype=torch.int32, device="cuda"); ctx.gen_uniform_u32(a, seed=1)
b = torch.empty(n, dtype=torch.int32, device="cuda")
k = torch.empty(n, dtype=torch.int32, device="cuda")
k2 = torch.empty(n, dtype=torch.int32, device="cuda")
ctx.reserve(n + n // 8, 4, 0)
side = torch.cuda.Stream()
def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        ctx.gen_uniform_u32(k, seed=5)
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return round(best * 1e3, 3)
def side_copy():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): b.copy_(a)
def elementwise():
    for _ in range(8): k2.add_(1)
def xch_async():
    return all_to_all_v(dist, b, a, [n], [n], async_op=True)
res = {}
res["side copy alone"] = t(side_copy)
res["elementwise alone"] = t(elementwise)
res["sort alone"] = t(lambda: ctx.sort_u32(k))
res["exchange alone"] = t(lambda: [h.wait() for h in xch_async()])
res["side copy || elementwise"] = t(lambda: (side_copy(), elementwise()))
res["side copy || sort"] = t(lambda: (side_copy(), ctx.sort_u32(k)))
def f1():
    hs = xch_async(); elementwise(); [h.wait() for h in hs]
res["exchange || elementwise"] = t(f1)
def f2():
    hs = xch_async(); ctx.sort_u32(k); [h.wait() for h in hs]
res["exchange || sort"] = t(f2)
# the sort on a NON-default stream
s3 = torch.cuda.Stream()
def f3():
    hs = xch_async()
    with torch.cuda.stream(s3):
        ctx.use_torch_stream(); ctx.sort_u32(k)
    ctx.use_torch_stream()
    [h.wait() for h in hs]
res["exchange || sort on a pool stream"] = t(f3)
print(res, flush=True)
print("env", {k_: os.environ.get(k_) for k_ in ("GPU_MAX_HW_QUEUES", "HIP_FORCE_DEV_KERNARG", "NCCL_MIN_NCHANNELS", "HSA_ENABLE_SDMA")}, flush=True)
dist.destroy_process_group()
