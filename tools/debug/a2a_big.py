"""Does a single all-to-all block of >= 2^31 bytes survive the nccl backend?  One rank exchanging with itself."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for logn in (28, 29, 30):
    n = 1 << logn
    a = torch.arange(n, dtype=torch.int32, device="cuda")
    b = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    w = dist.all_to_all_single(b, a, output_split_sizes=[n], input_split_sizes=[n], async_op=True)
    w.wait(); torch.cuda.synchronize()
    bad = int((a != b).sum().item())
    first_bad = int((a != b).nonzero()[0].item()) if bad else -1
    print(f"all_to_all_single 2^{logn} int32 ({n*4/2**30:.0f} GiB): mismatches={bad} first={first_bad}", flush=True)
    outs, ins = [torch.full((n,), -1, dtype=torch.int32, device="cuda")], [a]
    dist.all_to_all(outs, ins)
    torch.cuda.synchronize()
    print(f"all_to_all (lists) 2^{logn}: mismatches={int((outs[0] != a).sum().item())}", flush=True)
    del a, b, outs, ins
from inplacemsdradixsort_amd import MsdContext, MsdShard, torch_nccl_comm
ctx = MsdContext(0); ctx.use_torch_stream()
sh = MsdShard(ctx, torch_nccl_comm(0)); sh.set_option("force_exchange", 1)
n = 1 << 30
k = torch.empty(n, dtype=torch.int32, device="cuda"); ctx.gen_uniform_u32(k, seed=3)
c0 = ctx.check(k)
recv = torch.empty(n + 64, dtype=torch.int32, device="cuda")
out = sh.sort_u32(k, recv, None, scheme="coarse")
c1 = ctx.check(out)
print("native coarse 2^30 through ncclSend/ncclRecv:", c1[0] == 0 and c1[1:] == c0[1:], c1[0], flush=True)
dist.destroy_process_group()
