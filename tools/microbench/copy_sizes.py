"""Device copy rate vs buffer size (does the 256 MB memory-side cache help a read+write stream?)."""
import torch, time
for mb in (4, 16, 64, 128, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.int32, device="cuda").random_()
    b = torch.empty_like(a)
    reps = max(5, 20000 // mb)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{mb:5d} MB copy: {ms*1e3:9.1f} us  {2*mb/1024/(ms/1e3):8.1f} GB/s (read+write)")
    # read-only: sum
    for _ in range(2): a.sum()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): a.sum()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{mb:5d} MB sum : {ms*1e3:9.1f} us  {mb/1024/(ms/1e3):8.1f} GB/s (read)")
