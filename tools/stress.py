"""Repeats full-size sorts of skewed keys (2^30 and 2^28 Zipf, 2^29 keys of a handful of distinct values) and checks
every result: order, key sum and xor.  These inputs drive the block permutation's hot lists (sharded claim cursors);
a timing-dependent bookkeeping error there shows only under load.  python tools/stress.py <iterations>"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from inplacemsdradixsort_amd import MsdContext
ctx = MsdContext(0)
bad = 0
for it in range(int(sys.argv[1])):
    for logn, gen in ((30, ctx.gen_zipf_u32), (28, ctx.gen_zipf_u32), (29, ctx.gen_dup_u32)):
        t = torch.empty(1 << logn, dtype=torch.int32, device="cuda")
        if gen == ctx.gen_dup_u32:
            gen(t, 3 + it, seed=it)
        else:
            gen(t, seed=1000 + it)
        c0 = ctx.check(t)
        try:
            ctx.sort_u32(t)
        except Exception as e:
            print("EXC", it, logn, e, flush=True); bad += 1; continue
        c1 = ctx.check(t)
        if c1[0] != 0 or c1[1:] != c0[1:]:
            print("BAD", it, logn, c0, c1, ctx.stats(), flush=True); bad += 1
        del t
print("stress done, bad =", bad)
sys.exit(1 if bad else 0)
