#!/usr/bin/env python3
"""Random cases of the fine exchange's building blocks, all "ranks" in one process: every rank's shard goes through
msd_order_low16_u32 (and msd_hist2_pack_u32_low16), what each destination would receive is laid out as the all-to-all
delivers it, and msd_merge_buckets_u32_low16 / _hist2 must give the sort of all keys of the destination's range.

    python tools/soak_exchange.py <seed> <cases>
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from inplacemsdradixsort_amd import MsdContext  # noqa: E402

seed, cases = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
ctx = MsdContext(0)
RB = ctx.HIST2_RECORD_BYTES
done = {"low16": 0, "hist": 0, "hist_overflow": 0, "rejected": 0}
for case in range(cases):
    G = int(rng.choice([1, 2, 4, 8]))
    nbl = 65536 // G
    kind = rng.choice(["uniform", "dups", "narrow", "fewbuckets", "sorted"])
    shards, lows, counts, recs, flags = [], [], [], [], []
    for r in range(G):
        n = int(rng.integers(1, 1 << int(rng.integers(10, 22))))
        if kind == "uniform":
            k = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        elif kind == "dups":
            k = rng.choice(rng.integers(0, 1 << 32, int(rng.integers(1, 2000)), dtype=np.uint64), n).astype(np.uint32)
        elif kind == "narrow":
            k = (rng.integers(0, 1 << int(rng.integers(1, 32)), n, dtype=np.uint64)).astype(np.uint32)
        elif kind == "fewbuckets":   # dense buckets: a handful of upper halves
            k = ((rng.integers(0, int(rng.integers(1, 40)), n, dtype=np.uint64) * 1657 % 65536) << 16 | rng.integers(0, 1 << 16, n, dtype=np.uint64)).astype(np.uint32)
        else:
            k = np.sort(rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32))
        shards.append(k)
        t = torch.from_numpy(k.view(np.int32).copy()).cuda()
        low = torch.empty(n + 8, dtype=torch.int16, device="cuda")
        c = ctx.order_low16(t, low)
        assert (c.cpu().numpy() == np.bincount(k >> np.uint32(16), minlength=65536)).all(), (case, kind, "counts")
        rec = torch.empty(65536 * RB, dtype=torch.uint8, device="cuda") if G <= 2 and n >= 4096 else None   # (memory: 1.1 GB per rank)
        if rec is not None:
            flags.append(int(ctx.hist2_pack(low[:n], ctx.bounds_from_counts16(c), rec).item()))
        lows.append(low)
        counts.append(c)
        recs.append(rec)
    allk = np.concatenate(shards)
    use_hist = all(r is not None for r in recs) and not any(flags)
    if any(flags):
        done["hist_overflow"] += 1
    for d in range(G):
        cm = torch.stack([c[d * nbl:(d + 1) * nbl] for c in counts]).contiguous()
        want = np.sort(allk[(allk >> np.uint32(16)) // nbl == d])
        m = want.size
        # low halves: source-major, every source's range of buckets back to back
        parts, base, at = [], [], 0
        for r in range(G):
            s0 = int(counts[r][:d * nbl].sum())
            ln = int(cm[r].sum())
            parts.append(lows[r][s0:s0 + ln])
            base.append(at)
            at += ln
        arrived = torch.cat(parts + [torch.zeros(16, dtype=torch.int16, device="cuda")])
        out = torch.full((m + 4,), -1, dtype=torch.int32, device="cuda")
        ctx.merge_buckets(arrived, cm, base, 16, d * nbl, out, m)
        got = out.cpu().numpy().view(np.uint32)
        assert (got[:m] == want).all() and (got[m:] == 0xFFFFFFFF).all(), (case, kind, G, d, "low16")
        done["low16"] += 1
        done["rejected"] += ctx.stats().get("merge_rejected", 0)
        if use_hist:
            rr = torch.cat([recs[r][d * nbl * RB:(d + 1) * nbl * RB] for r in range(G)])
            out.fill_(-1)
            ctx.merge_buckets(rr, cm, [0] * G, 16, d * nbl, out, m)
            got = out.cpu().numpy().view(np.uint32)
            assert (got[:m] == want).all() and (got[m:] == 0xFFFFFFFF).all(), (case, kind, G, d, "hist")
            done["hist"] += 1
            done["rejected"] += ctx.stats().get("merge_rejected", 0)
    if case % 20 == 19:
        print(case + 1, done, flush=True)
print("passed", cases, done)
