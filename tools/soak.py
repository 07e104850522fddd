"""One-off soak run: many random cases (sizes, types, distributions, option settings) against torch.sort.
Not part of the test-suite (takes minutes); prints the first failing case and exits non-zero."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from inplacemsdradixsort_amd import MsdContext

ctx = MsdContext(0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 12345
g = torch.Generator(device="cuda"); g.manual_seed(seed)
torch.manual_seed(seed)  # the CPU-side choices (sizes, kinds, options) too: a run can be repeated exactly
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
types = sys.argv[3].split(",") if len(sys.argv) > 3 else ["u32", "u64", "pairs"]

def rnd(n, bits):
    hi = torch.randint(0, 1 << 31, (n,), device="cuda", generator=g, dtype=torch.int64)
    lo = torch.randint(0, 1 << 31, (n,), device="cuda", generator=g, dtype=torch.int64)
    x = (hi << 33) ^ (lo << 2) ^ (hi >> 29)
    return x if bits == 64 else (x & 0xFFFFFFFF)

def make(kind, n, bits):
    x = rnd(n, bits)
    full = (1 << bits) - 1
    if kind == 0: return x
    if kind == 1: return torch.sort(x.view(torch.int64)).values
    if kind == 2: return torch.flip(torch.sort(x).values, dims=[0]).contiguous()
    if kind == 3:
        run = 1 << int(torch.randint(8, 18, (1,)).item())
        top = (torch.arange(n, device="cuda", dtype=torch.int64) // run) & 0xFF
        return (x & (full >> 8)) | (top << (bits - 8))
    if kind == 4: return x & int(torch.randint(0, 1 << 31, (1,)).item() * (1 << (bits - 31)) | 0xFF)
    if kind == 5:
        vals = rnd(int(torch.randint(1, 40, (1,)).item()), bits)
        return vals[torch.randint(0, vals.numel(), (n,), device="cuda", generator=g)]
    if kind == 6:
        y = x.clone(); y[torch.rand(n, device="cuda", generator=g) < 0.4] = int(x[0].item()); return y
    if kind == 7: return (x.double() / float(full)).pow(8).mul(float(full)).long() & full
    blk = 1 << int(torch.randint(10, 16, (1,)).item())
    m = n // blk * blk
    y = x.clone()
    if m: y[:m] = x[:m].view(-1, blk).sort(dim=1).values.view(-1)
    return y

t_start = time.time()
for c in range(cases):
    typ = types[c % len(types)]
    bits = 32 if typ == "u32" else 64
    lo_, hi_ = float(os.environ.get("SOAK_LOGN_MIN", 8)), float(os.environ.get("SOAK_LOGN_MAX", 27.3))
    logn = float(torch.empty(1).uniform_(lo_, hi_ if typ != "pairs" else hi_ - 1).item())
    n = int(2 ** logn) + int(torch.randint(0, 5, (1,)).item())
    kind = int(torch.randint(0, 9, (1,)).item())
    mode = int(torch.randint(0, 3, (1,)).item())
    ctx.set_option("direct_mode", mode)
    if not os.environ.get("SOAK_DEFAULT_OPTS"):
        ctx.set_option("count16", int(torch.randint(0, 3, (1,)).item()))
        ctx.set_option("regpart", int(torch.randint(0, 4, (1,)).item() != 0))
        ctx.set_option("leaf17", int(torch.randint(0, 4, (1,)).item() != 0))          # round 3's kernels: mostly on
        ctx.set_option("mid_leaf", int(torch.randint(0, 4, (1,)).item() != 0))
        ctx.set_option("stream_kernel", 2 if int(torch.randint(0, 4, (1,)).item()) != 0 else 1)
        ctx.set_option("direct_min", 1 << int(torch.randint(14, 27, (1,)).item()))
        ctx.set_option("direct_min_parent", 1 << int(torch.randint(10, 18, (1,)).item()))
    x = make(kind, n, bits)
    # sometimes: fewer open bits (end_bit) and a 16-byte aligned sub-array start
    eb = bits
    if int(torch.randint(0, 4, (1,)).item()) == 0:
        eb = int(torch.randint(1, bits + 1, (1,)).item())
        if eb < bits:
            x = (x & ((1 << eb) - 1)) | (int(rnd(1, bits)[0].item()) & ~((1 << eb) - 1) & ((1 << bits) - 1) if bits == 32 else (x[0] & ~((1 << eb) - 1)))
    off = int(torch.randint(0, 4, (1,)).item()) * (4 if bits == 32 else 2) if int(torch.randint(0, 3, (1,)).item()) == 0 else 0
    if off:
        x = torch.cat([rnd(off, bits), x])
    print(f"case {c}: {typ} n={n} kind={kind} mode={mode}", file=sys.stderr, flush=True) if os.environ.get("SOAK_VERBOSE") else None
    if bits == 64:
        # torch has no uint64 sort: order by (high, low) halves as unsigned via bias
        ref = torch.sort(x ^ (-(1 << 63))).values ^ (-(1 << 63))
        k = x.clone()
    else:
        ref = torch.sort(x).values
        k = x.to(torch.int32) if False else (x - ((x >> 31) << 32)).to(torch.int32)
    if c % 7 == 3 and not off and eb == bits:  # the one-digit partition (the pass before the multi-GPU exchange)
        rb = int(torch.randint(1, 9, (1,)).item())
        sh = int(torch.randint(0, bits - rb + 1, (1,)).item())
        r = torch.arange(n, device="cuda", dtype=torch.int64) if typ == "pairs" else None
        cnt = ctx.partition(k, sh, rb, r)
        ku = (k.to(torch.int64) & 0xFFFFFFFF) if bits == 32 else k
        xu = x
        d = (ku >> sh) & ((1 << rb) - 1)
        ok = bool((d[1:] >= d[:-1]).all()) if n > 1 else True
        d0 = (xu >> sh) & ((1 << rb) - 1)
        ok = ok and bool((torch.bincount(d0, minlength=1 << rb) == cnt).all())
        srt = (lambda t: torch.sort(t ^ (-(1 << 63))).values) if bits == 64 else (lambda t: torch.sort(t).values)
        ok = ok and bool((srt(ku) == srt(xu)).all())
        if r is not None:
            ok = ok and bool(((r >= 0) & (r < n)).all()) and bool((x[r] == k).all())
        if not ok:
            print(f"FAIL partition case {c}: typ={typ} n={n} kind={kind} mode={mode} shift={sh} radix_bits={rb}", ctx.stats(), flush=True)
            sys.exit(1)
        del x, ref, k
        continue
    if off:  # the head must stay untouched, the tail sorted
        if bits == 64:
            ref = torch.cat([x[:off], torch.sort(x[off:] ^ (-(1 << 63))).values ^ (-(1 << 63))])
        else:
            ref = torch.cat([x[:off], torch.sort(x[off:]).values])
    if typ == "u32":
        ctx.sort_u32(k[off:], end_bit=eb)
        out = k.to(torch.int64) & 0xFFFFFFFF
        ok = bool((out == ref).all())
    elif typ == "u64":
        ctx.sort_u64(k[off:], end_bit=eb)
        ok = bool((k == ref).all())
    else:
        r = torch.arange(n + off, device="cuda", dtype=torch.int64)
        ctx.sort_pairs_u64(k[off:], r[off:], end_bit=eb)
        n = n + off
        # (check the rids' range first: indexing with a corrupted rid would fault inside torch)
        ok = bool(((r >= 0) & (r < n)).all()) and bool((torch.sort(r).values == torch.arange(n, device="cuda")).all())
        ok = ok and bool((k == ref).all()) and bool((x[r] == k).all())
    if not ok:
        print(f"FAIL case {c}: typ={typ} n={n} kind={kind} mode={mode} end_bit={eb} off={off}", ctx.stats(), flush=True)
        sys.exit(1)
    del x, ref, k
    if c % int(os.environ.get("SOAK_EVERY", 20)) == 0:
        print(f"case {c} ok ({typ} n={n} kind={kind} mode={mode}) t={time.time()-t_start:.0f}s", flush=True)
print(f"all {cases} cases ok in {time.time()-t_start:.0f}s")
