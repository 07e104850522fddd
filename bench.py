#!/usr/bin/env python3
"""bench.py -- headline benchmark: Gkeys/s of the device-resident in-place MSD radix
sort on 2^30 uniform u32 keys per GPU (BASELINE.json configs[1]; configs[3] for N>1).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" sorts one fresh array of 2^LOGN uniform keys that is already resident in
HBM (arrays for all W+K steps are generated before the timed region).  For N>1
every rank holds its own 2^LOGN shard (weak scaling): one in-place top-digit pass
packs the send side, one RCCL all-to-all exchanges key ranges, each rank sorts
what it received; value = all ranks' keys / max-over-ranks time.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     -- dominant kernel: algorithmic bytes / HIP-event time vs 8 TB/s
  cpu_baseline -- the reference's single-thread core (oracle/_ref) on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# ALGORITHMIC bytes, SURVEY.md section 8d: 3 touches x 4 B per key per 8-bit digit pass
BYTES_PER_KEY_PASS = 12
# share of one partition round's 12 B/key credited to each of its two streaming kernels
# (each really moves 4 B in + 4 B out per key), and digit passes done per launch
KERNEL_ALGO = {
    "A classify": ("classify_kernel<u32>", 6.0),       # per key per round
    # direct rounds: the permute read + write of the round's 12 B (the histogram read is the sample / the
    # exact counting pass, "A histogram"); the few misplaced blocks chains still moves are not credited
    "A classify direct": ("classify_direct_kernel<u32>", 8.0),
    "A histogram": ("direct_hist_kernel<u32>", 4.0),
    "B block permute": ("chains_kernel<u32>", 6.0),    # per key per round
    "LDS sort": ("lds_sort_kernel<u32>", None),        # remaining passes x 12 B per key
    "count sort": ("count_place_kernel<u32>", None),   # remaining passes x 12 B per key
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--logn", type=int, default=30, help="log2 keys per GPU (default 30 = BASELINE config)")
    ap.add_argument("--dist", choices=["uniform", "zipf"], default="uniform")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-logn", type=int, default=28, help="log2 tuples for the reference's 64-thread sort()")
    ap.add_argument("--cpu-logn-1t", type=int, default=27, help="log2 keys for the single-thread core")
    return ap.parse_args()


def cpu_baseline_single(logn: int):
    """The reference's single-thread core (schedule_passes + local_radixsort,
    src/msb_64.c:2232-2244) from oracle/_ref, timed on this host; falls back to the
    C restatement ("port") if the prebuilt reference is absent."""
    import numpy as np
    from oracle import oracle as O
    n = 1 << logn
    k32 = O.gen_uniform_u32(n, seed=0x5EED0001)
    if O.have_ref():
        k = O.aligned(n)
        r = O.aligned(n)
        k[:] = k32
        r[:] = k32
        t0 = time.perf_counter()
        O.ref_sort_pairs_u64_inplace(k, r, 32)
        dt = time.perf_counter() - t0
        ok = bool((np.diff(k.view(np.int64)) >= 0).all()) and bool((k == r).all()) and int(k.sum(dtype=np.uint64)) == int(k32.sum(dtype=np.uint64))
        kind = "reference"
    else:
        t0 = time.perf_counter()
        O.sort_u32_inplace(k32)
        dt = time.perf_counter() - t0
        ok = bool((np.diff(k32.astype(np.int64)) >= 0).all())
        kind = "port"
    return {
        "value": round(n / dt / 1e9, 5), "unit": "Gkeys/s", "cores": 1, "kind": kind,
        "sample": f"2^{logn} uniform u32 keys (zero-extended to u64, rid=key, bits=32), "
                  f"single-thread schedule_passes+local_radixsort, {dt:.2f} s, output verified={ok}, "
                  f"host has {os.cpu_count()} logical cpus",
    }


def cpu_baseline(logn_mt: int, logn_1t: int):
    """CPU baseline beside the GPU number: the reference's own pthreads sort() (src/msb_64.c:2261,
    64 threads as it demands) from oracle/_ref, run in a subprocess (the reference is fragile,
    SURVEY.md section 0.9) and verified; if it is absent, crashes or fails verification, the
    reference's single-thread core is reported instead."""
    import subprocess
    single = cpu_baseline_single(logn_1t)
    script = os.path.join(ROOT, "oracle", "ref_sort_mt.py")
    ref_lib = os.path.join(ROOT, "oracle", "_ref", "libref_msb64.so")
    note = "oracle/_ref absent"
    if os.path.exists(ref_lib):
        try:
            p = subprocess.run([sys.executable, script, str(logn_mt), "2", "3"], capture_output=True, text=True, timeout=420)
            if p.returncode == 0 and p.stdout.strip():
                r = json.loads(p.stdout.strip().splitlines()[-1])
                if r.get("verified"):
                    return {
                        "value": round(r["gkeys_per_s"], 5), "unit": "Gkeys/s", "cores": 64, "kind": "reference",
                        "sample": f"reference sort() with its mandatory 64 pthreads on 2^{logn_mt} (u32<<32 key, rid=key) tuples, "
                                  f"numa=2 arrays, fudge=2.0, median of 3 runs {r['median_s']:.2f} s, every run verified "
                                  f"(order, key==rid, sum, xor); host has {r['logical_cpus']} logical cpus",
                        "single_thread_core": single,
                    }
                note = "reference sort() produced a wrong result (it is nondeterministic, SURVEY.md section 0.9)"
            else:
                note = f"reference sort() exited with {p.returncode}"
        except Exception as e:  # timeout, crash
            note = f"reference sort() did not finish: {type(e).__name__}"
    single["sample"] += f"; multi-thread sort() not reported: {note}"
    return single


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from inplacemsdradixsort_amd import MsdContext

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    N = world
    torch.cuda.set_device(local_rank)
    if N > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    ctx = MsdContext(local_rank)
    ctx.use_torch_stream()
    n = 1 << args.logn
    W, K = args.warmup, args.steps
    gen = ctx.gen_uniform_u32 if args.dist == "uniform" else ctx.gen_zipf_u32
    seed = 0x5EED0001 if args.dist == "uniform" else 0x5EED0003

    # ---- inputs for every step, resident before the clock starts
    bufs = []
    for s in range(W + K):
        t = torch.empty(n, dtype=torch.int32, device="cuda")
        gen(t, seed=seed + 1000003 * s, first=rank * n)  # C4: global index over all shards
        bufs.append(t)
    ctx.reserve(n + n // 8, 4, 0)
    # N > 1: two receive buffers with 12.5 % slack (the reference's fudge); the exchange of step s runs
    # (RCCL stream, xGMI) while step s-1 is sorted locally -- inplacemsdradixsort_amd.dist.ShardedSorter
    recv = [torch.empty(n + n // 8, dtype=torch.int32, device="cuda") for _ in range(2)] if N > 1 else None
    checks0 = [ctx.check(t) for t in bufs[W:]] if N == 1 else None

    from inplacemsdradixsort_amd.dist import ShardedSorter
    sorter = ShardedSorter(ctx, dist, N, recv) if N > 1 else None

    def run_steps(shards):
        if N == 1:
            for t in shards:
                ctx.sort_u32(t)
            return list(shards)
        outs = []
        for i, t in enumerate(shards):
            sorter.submit(t)
            if i:
                outs.append(sorter.collect())
        if shards:
            outs.append(sorter.collect())
        return outs

    run_steps(bufs[:W])
    torch.cuda.synchronize()
    if N > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = run_steps(bufs[W:W + K])
    torch.cuda.synchronize()
    if N > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if N > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- verify every timed step's output (outside the clock)
    verified = True
    if N == 1:
        for o, c0 in zip(outs, checks0):
            v, s_, x_ = ctx.check(o)
            verified &= (v == 0 and s_ == c0[1] and x_ == c0[2])
    else:
        v, _, _ = ctx.check(outs[-1])
        verified &= v == 0

    # ---- per-kernel HIP-event timing of one more step (profiling adds event records, so it is separate)
    roofline = None
    if N == 1:
        t = bufs[0]
        gen(t, seed=seed + 77, first=0)
        ctx.set_profiling(True)
        ctx.sort_u32(t)
        torch.cuda.synchronize()
        ctx.set_profiling(False)
        ph = dict(ctx.phases())
        st = ctx.stats()
        rounds = st.get("rounds", 0)
        dom = max((p for p in ph if p in KERNEL_ALGO), key=lambda p: ph[p], default=None)
        if dom:
            name, per_key = KERNEL_ALGO[dom]
            if per_key is None:  # LDS sort: the digit passes the partition rounds left over
                passes_left = max(0, 4 - rounds) if args.dist == "uniform" else 2
                launches, algo = 1, n * BYTES_PER_KEY_PASS * passes_left
            else:
                direct = st.get("direct_rounds", 0)
                if dom == "A classify direct":
                    launches = max(1, direct)
                elif dom == "A histogram":
                    launches = max(1, direct - 1)
                elif dom == "A classify":
                    launches = max(1, rounds - direct)
                else:
                    launches = max(1, rounds)
                algo = n * per_key
            avg_us = ph[dom] / launches
            ach = algo / (avg_us * 1e-6) / 1e9
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get(name, {}).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            roofline = {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                        "moved_GBps": None if not traffic else round(traffic / (avg_us * 1e-6) / 1e9, 1),
                        "avg_launch_us": round(avg_us, 1), "launches_per_sort": launches,
                        "algorithmic_bytes_per_launch": int(algo),
                        "phases_us": {k: round(v, 1) for k, v in ph.items()}}

    # ---- achievable copy rate on this device, same run (second denominator)
    copy_gbps = None
    if N == 1:
        a, b = bufs[0], bufs[1] if len(bufs) > 1 else torch.empty_like(bufs[0])
        b.copy_(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        copy_gbps = 5 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9

    total_keys = N * n * K
    value = total_keys / dt / 1e9
    whole = N * n * 4 * BYTES_PER_KEY_PASS / (dt / K) / 1e9 / N  # per-GPU algorithmic GB/s (48 B/key)
    out = {
        "metric": "Gkeys/s + achieved HBM GB/s, 2^30 uniform u32 keys, 1/2/4/8 MI355X",
        "value": round(value, 3), "unit": "Gkeys/s", "n_gpus": N, "steps": K, "warmup": W,
        "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"2^{args.logn} {args.dist} u32 keys per GPU, in-place MSD radix sort, 8-bit digits"
                               + (f", range-partitioned over {N} GPUs by one RCCL all-to-all per step (overlapped with the previous step's local sort)" if N > 1 else ""),
                   "keys_per_gpu": n, "distribution": args.dist, "verified": bool(verified),
                   "workspace_bytes": ctx.workspace_bytes},
        "whole_sort": {"algorithmic_GBps_per_gpu": round(whole, 1), "frac_of_peak": round(whole / HBM_PEAK_GBS, 4),
                       "bytes_per_key": 4 * BYTES_PER_KEY_PASS,
                       "device_copy_GBps_same_run": None if copy_gbps is None else round(copy_gbps, 1),
                       "frac_of_device_copy": None if copy_gbps is None else round(whole / copy_gbps, 4)},
        "roofline": roofline,
    }
    if rank == 0:
        out["cpu_baseline"] = None if (args.no_cpu_baseline or N > 1) else cpu_baseline(args.cpu_logn, args.cpu_logn_1t)
        print(json.dumps(out), flush=True)
    if N > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
