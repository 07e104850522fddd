#!/usr/bin/env python3
"""bench.py -- headline benchmark: Gkeys/s of the device-resident in-place MSD radix
sort on 2^30 uniform u32 keys per GPU (BASELINE.json configs[1]; configs[3] for N>1).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" sorts one fresh array of 2^LOGN uniform keys that is already resident in
HBM (arrays for all W+K steps are generated before the timed region).  For N>1
every rank holds its own 2^LOGN shard (weak scaling): the shard is ordered by its
upper halves, one RCCL all-to-all exchanges key ranges -- as the keys' low halves,
at 2 ranks as histogram records of the buckets --, one counting
pass finishes what arrived; exchange s runs under the local work of steps s - 1 and
s + 1; value = all ranks' keys / max-over-ranks time.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      -- dominant kernel: algorithmic bytes / HIP-event time vs 8 TB/s (N > 1: the whole local sort per GPU,
                   the exchange's bytes and time beside it)
  cpu_baseline  -- the reference's own 64-thread sort() (oracle/_ref) on 2^30 (or 2^28) tuples on this host, verified,
                   with its single-thread core on 2^27 keys beside it (rank 0; a reported baseline, not the target)
  other_configs -- (default single-GPU run) configs c3, c5a, c5b: three verified steps each, after the headline loop
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
# Measured ceilings of the same device kind (tools/microbench/stream_copy.hip, profiles/r02_stream_ceiling.jsonl):
# sequential copy 6.1-6.2 TB/s, random 256-byte block permutation 5.3 TB/s, the direct classify kernel's in-place
# pattern with no work at all 4.9 TB/s.  Reported beside the spec peak, never instead of it.
CEILINGS_GBS = {"sequential_copy": 6150.0, "block_permutation_256B": 5300.0, "in_place_pieces_pattern": 4900.0}

# ---- workloads (BASELINE.json configs; SURVEY.md section 8d gives the ALGORITHMIC bytes: per 8-bit digit pass one
# histogram read of the key + one permute read + one permute write of key and payload, P = significant key bits / 8)
CONFIGS = {
    "c2": dict(title="uniform u32 keys", key=4, val=0, passes=4, dist="uniform", dtype="u32",
               passes_note="8+8 partition rounds (direct placement), then one 16-bit counting leaf that re-generates the keys"),
    "c3": dict(title="Zipf(theta=1) u32 keys (skewed-bucket path)", key=4, val=0, passes=4, dist="zipf", dtype="u32",
               passes_note="8+8 streaming partition rounds; heavy buckets finish in the multi-workgroup counting sort"),
    "c5a": dict(title="(u64 key, u64 rid) tuples, full 64-bit keys", key=8, val=8, passes=8, dist="uniform", dtype="u64+u64",
                passes_note="8+8 direct rounds, one narrow round, LDS counting leaf on the varying bits"),
    "c5b": dict(title="(u64 key, u64 rid) tuples, upper 32 key bits zero (bit skipping)", key=8, val=8, passes=4, dist="uniform",
                dtype="u64+u64", passes_note="one OR/AND scan finds 32 constant bits (4 digit passes skipped), then as c5a"),
    "u64": dict(title="uniform u64 keys", key=8, val=0, passes=8, dist="uniform", dtype="u64",
                passes_note="8+8 direct rounds, LDS counting leaf"),
}


def algo_bytes_per_elem(cfg):
    return cfg["passes"] * (cfg["key"] + 2 * (cfg["key"] + cfg["val"]))


def tname(cfg):
    return {"u32": "u32", "u64": "u64", "u64+u64": "u64,u64"}[cfg["dtype"]]


def kernel_algo(cfg):
    """phase of msd_phase_* -> (kernel, algorithmic bytes per element and launch).  A round's
    K + 2 (K+V) bytes are shared between its kernels: direct rounds = histogram read (sample / exact count) +
    permute read and write (classify_direct); streaming rounds = half each for classify and the block permutation.
    Leaves are credited the digit passes the rounds left over (None: computed from the round count)."""
    K, V, t = cfg["key"], cfg["val"], tname(cfg)
    rnd = K + 2 * (K + V)
    return {
        "A classify": (f"classify_stream2_kernel<{t}>", rnd / 2),
        "A classify direct": (f"classify_direct2_kernel<{t}>", 2 * (K + V)),
        "A histogram": (f"direct_hist_kernel<{t}>", K),
        "B block permute": (f"chains_kernel<{t}>", rnd / 2),
        # (u64 keys: the segments two rounds leave are finished by leaf17_kernel<NoVal>, launched in this phase)
        "LDS sort": ("leaf17_kernel<NoVal>" if t == "u64" else f"leaf_count_sort_kernel<{t}>", None),
        "leaf17": ("leaf17_kernel<u64>", None),
        "count sort": ("count_place16_kernel" if t == "u32" else f"count_place_kernel<{t}>", None),
        "big count sort": (f"bigcount_write_kernel<{t}>", None),
    }


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--logn", type=int, default=30, help="log2 elements per GPU (default 30 = BASELINE config)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2",
                    help="c2 (default) = the headline: 2^30 uniform u32; c3 Zipf; c5a/c5b pairs; u64 keys")
    ap.add_argument("--dist", choices=["uniform", "zipf"], default=None, help="(older spelling) --dist zipf = --config c3")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: REHEARSAL of the N > 1 path with all ranks on cuda:0 and the collectives staged through "
                         "host memory (RCCL cannot run several ranks on one device); its numbers mean nothing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="headline only (the default c2 run also times c3, c5a, c5b)")
    ap.add_argument("--scheme", choices=["fine", "coarse"], default=None, help="N > 1: force the sharding scheme (dist.use_fine)")
    ap.add_argument("--whole-keys", action="store_true", help="N > 1, fine scheme: whole keys travel (A/B; dist.FINE_LOW16 = dist.FINE_HIST = False)")
    ap.add_argument("--low-halves", action="store_true", help="N > 1, fine scheme: the keys' low halves travel, never histogram records (A/B; dist.FINE_HIST = False)")
    ap.add_argument("--compute-stream", choices=["pool", "high", "default"], default="high",
                    help="N > 1: the stream the local work runs on: a stream of its own (pool; high = with high priority) or torch's default stream")
    ap.add_argument("--one-rank-exchange", action="store_true",
                    help="REHEARSAL on one GPU with the real RCCL backend: a single rank runs the N > 1 loop -- pre-exchange pass, count "
                         "exchange, asynchronous all-to-all to itself, counting leaf, pipelined -- instead of the plain sort "
                         "(start it like an N > 1 run: torch.distributed.run --nproc-per-node 1); the exchange is a device-local "
                         "copy, so the line says nothing about xGMI")
    ap.add_argument("--cpu-logn", type=int, default=None, help="log2 tuples for the reference's 64-thread sort() "
                    "(default: 30 when the host has the memory for it, else 28)")
    ap.add_argument("--cpu-logn-1t", type=int, default=27, help="log2 keys for the single-thread core")
    args = ap.parse_args()
    if args.dist == "zipf" and args.config == "c2":
        args.config = "c3"
    return args


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


def host_mem_available_gib() -> float:
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / (1 << 20)
    except OSError:
        pass
    return 0.0


def cpu_baseline(logn_mt, logn_1t: int):
    """CPU baseline beside the GPU number: the reference's own pthreads sort() (src/msb_64.c:2261,
    64 threads as it demands) from oracle/_ref, run in a subprocess (the reference is fragile,
    SURVEY.md section 0.9) and verified; if it is absent, crashes or fails verification, the
    reference's single-thread core is reported instead.  BASELINE.md section 3 asks for n = 2^30 tuples
    (16 B x fudge 2.0 = 32 GiB + the verification's copies): used when the host has the memory, else 2^28."""
    import subprocess
    single = cpu_baseline_single(logn_1t)
    script = os.path.join(ROOT, "oracle", "ref_sort_mt.py")
    ref_lib = os.path.join(ROOT, "oracle", "_ref", "libref_msb64.so")
    note = "oracle/_ref absent"
    mem = host_mem_available_gib()
    why_n = ""
    if logn_mt is None:
        logn_mt = 30 if mem >= 96 else 28
        if logn_mt != 30:
            why_n = f" (2^30 tuples need about 80 GiB of host memory with verification, {mem:.0f} GiB available)"
    tries = [logn_mt] + ([28] if logn_mt > 28 else [])
    if os.path.exists(ref_lib):
        for ln in tries:
            try:
                reps = 2 if ln >= 30 else 3
                p = subprocess.run([sys.executable, script, str(ln), "2", str(reps)], capture_output=True, text=True,
                                   timeout=600 if ln >= 30 else 420)
                if p.returncode == 0 and p.stdout.strip():
                    r = json.loads(p.stdout.strip().splitlines()[-1])
                    if r.get("verified"):
                        best = min(r["seconds"])
                        return {
                            "value": round(r["n"] / best / 1e9, 5), "unit": "Gkeys/s", "cores": 64, "kind": "reference",
                            "sample": f"reference sort() with its mandatory 64 pthreads on 2^{ln} (u32<<32 key, rid=key) tuples{why_n}, "
                                      f"numa=2 arrays, fudge=2.0, best of {reps} runs {best:.2f} s (all: "
                                      f"{', '.join(f'{x:.2f}' for x in r['seconds'])}; the first touches its memory), every run verified "
                                      f"(order, key==rid, sum, xor); host has {r['logical_cpus']} logical cpus, {mem:.0f} GiB free",
                            "phases_us_last_run": r.get("phases_us_last"),
                            "single_thread_core": single,
                        }
                    note = "reference sort() produced a wrong result (it is nondeterministic, SURVEY.md section 0.9)"
                else:
                    note = f"reference sort() exited with {p.returncode} at 2^{ln}"
            except Exception as e:  # timeout, crash
                note = f"reference sort() did not finish at 2^{ln}: {type(e).__name__}"
    single["sample"] += f"; multi-thread sort() not reported: {note}"
    return single


def git_sha() -> str:
    try:
        import subprocess
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or "?"
    except Exception:
        return "?"


def torch_check(torch, t, rids=None, chunk: int = 1 << 26):
    """Independent of the library's own check kernel: (order violations, key sum mod 2^64[, key != rid]) of a tensor of
    u32 / u64 bit patterns, by chunked torch reductions (neighbours compared as unsigned, chunk boundaries included)."""
    n = t.numel()
    viol, total, mism = 0, 0, 0
    prev_last = None
    for a in range(0, n, chunk):
        c = t[a:a + chunk]
        if t.element_size() == 4:
            u = c.to(torch.int64) & 0xFFFFFFFF                 # unsigned value
            total += int(u.sum().item())
        else:
            u = c ^ (-(1 << 63))                                # unsigned order as signed order
            total += int(c.sum().item())                        # (wraps like the device sum)
        viol += int((u[1:] < u[:-1]).sum().item())
        if prev_last is not None and int(u[0].item()) < prev_last:
            viol += 1
        prev_last = int(u[-1].item())
        if rids is not None:
            mism += int((c != rids[a:a + chunk]).sum().item())
    return viol, total & ((1 << 64) - 1), mism


def load_pmc(config_id: str):
    pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json" if config_id == "c2" else f"pmc_traffic_{config_id}.json")
    if os.path.exists(pmc_file):
        try:
            return json.load(open(pmc_file)), pmc_file
        except Exception:
            pass
    return None, pmc_file


def profile_one_sort(ctx, torch, cfg, config_id, n, gen, sort, t, r):
    """Per-kernel HIP-event timing of one more sort (profiling adds event records, so it is separate from the timed
    loop) -> the `roofline` object for the dominant kernel, and the bytes one whole sort really moves (committed PMC)."""
    gen(t, 77)
    if r is not None:
        r.copy_(t)
    ctx.set_profiling(True)
    sort(t, r)
    torch.cuda.synchronize()
    ctx.set_profiling(False)
    ph = dict(ctx.phases())
    st = ctx.stats()
    rounds = st.get("rounds", 0)
    direct = st.get("direct_rounds", 0)
    KA = kernel_algo(cfg)
    rnd_bytes = cfg["key"] + 2 * (cfg["key"] + cfg["val"])
    dom = max((p for p in ph if p in KA), key=lambda p: ph[p], default=None)
    pmc, pmc_file = load_pmc(config_id)
    roofline, real = None, None
    if dom:
        name, per_elem = KA[dom]
        if per_elem is None:  # leaves: the digit passes the partition rounds left over
            launches, algo = 1, n * rnd_bytes * max(0, cfg["passes"] - rounds)
        else:
            launches = {"A classify direct": max(1, direct), "A histogram": max(1, direct - 1),
                        "A classify": max(1, rounds - direct)}.get(dom, max(1, rounds))
            algo = n * per_elem
        avg_us = ph[dom] / launches
        ach = algo / (avg_us * 1e-6) / 1e9
        rec = (pmc or {}).get(name)
        if rec is None and pmc:  # extra template arguments in the profiler's name: classify_kernel<u32, false>
            rec = next((v for k, v in pmc.items() if k.startswith(name[:-1] + ",")), None)
        traffic = (rec or {}).get("hbm_bytes_per_launch")
        sha = (pmc or {}).get("__meta__", {}).get("git_sha") or git_sha()
        roofline = {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    # the PMC bytes are a committed measurement of the same kernel and workload (separate rocprofv3
                    # --pmc passes, FETCH_SIZE x2 as the guide prescribes), not collected inside this run
                    "traffic_source": None if traffic is None else f"{os.path.relpath(pmc_file, ROOT)} @ {sha} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this command, committed)",
                    "moved_GBps": None if not traffic else round(traffic / (avg_us * 1e-6) / 1e9, 1),
                    "avg_launch_us": round(avg_us, 1), "launches_per_sort": launches,
                    "algorithmic_bytes_per_launch": int(algo),
                    "measured_ceilings_GBps": CEILINGS_GBS,
                    "phases_us": {k: round(v, 1) for k, v in ph.items()}}
    if pmc:  # bytes one whole sort really moves, from the same committed PMC profile
        real = pmc.get("__per_sort__", {}).get("hbm_bytes")
    return roofline, real


def whole_sort_block(cfg, n, sec_per_step, real, copy_gbps=None):
    bpe = algo_bytes_per_elem(cfg)
    whole = n * bpe / sec_per_step / 1e9
    return {"algorithmic_GBps_per_gpu": round(whole, 1), "algorithmic_frac_of_peak": round(whole / HBM_PEAK_GBS, 4),
            "algorithmic_bytes_per_element": bpe,
            "note": "the algorithmic figure is SURVEY.md section 8d's fixed 3-touches-per-8-bit-pass model; the sort moves fewer "
                    "bytes (real_*), so the algorithmic rate is a speed-up over that model, not achieved bandwidth",
            "real_bytes_per_element": None if not real else round(real / n, 2),
            "real_GBps": None if not real else round(real / sec_per_step / 1e9, 1),
            "real_frac_of_peak": None if not real else round(real / sec_per_step / 1e9 / HBM_PEAK_GBS, 4),
            "device_copy_GBps_same_run": None if copy_gbps is None else round(copy_gbps, 1)}


def make_workload(ctx, torch, config_id, rank, n):
    cfg = CONFIGS[config_id]
    pairs = cfg["val"] > 0

    def gen(t, s):
        if config_id == "c2":
            ctx.gen_uniform_u32(t, seed=0x5EED0001 + 1000003 * s, first=rank * n)  # C4: global index over all shards
        elif config_id == "c3":
            ctx.gen_zipf_u32(t, seed=0x5EED0003 + 1000003 * s, first=rank * n)
        else:
            ctx.gen_uniform_u64(t, seed=0x5EED0005 + 1000003 * s, first=rank * n, shift_right=32 if config_id == "c5b" else 0)

    def sort(t, r=None):
        if cfg["dtype"] == "u32":
            ctx.sort_u32(t)
        elif pairs:
            ctx.sort_pairs_u64(t, r)
        else:
            ctx.sort_u64(t)

    return cfg, pairs, (torch.int32 if cfg["key"] == 4 else torch.int64), gen, sort


def run_other_config(config_id, logn, steps=3, warmup=1):
    """One more single-GPU config inside the default run: `bench.py --config <id>` itself, as a child
    process (its own HIP context: a failure there cannot take the headline line with it), with `steps` timed sorts -- every
    one verified by the library's check AND the independent torch reduction -- and one profiled sort for the dominant
    kernel.  Returns the fields of its line that matter here."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--config", config_id, "--logn", str(logn), "--steps", str(steps), "--warmup", str(warmup),
           "--no-cpu-baseline", "--no-other-configs"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": f"exit code {p.returncode}: {p.stderr.strip()[-300:]}"}
    d = json.loads(lines[-1])
    ws = d["whole_sort"]
    ws.pop("note", None)
    return {"workload": d["config"]["workload"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "value": d["value"], "unit": d["unit"],
            "dtype": d["dtype"], "verified": d["config"]["verified"], "steps_verified": d["config"]["steps_verified"], "whole_sort": ws,
            "roofline": d["roofline"]}


def main():
    args = parse()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.one_rank_exchange:
        # HIP multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; with the default the compute stream and
        # RCCL's stream shared one queue on the GPU box and the exchange ran strictly BETWEEN the compute kernels (one-rank
        # rehearsal: 14.1 ms per step = the sum of its parts; with 8 queues, or a high-priority compute stream, 12.2).
        # Must be set before the HIP runtime starts.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    from inplacemsdradixsort_amd import MsdContext
    from inplacemsdradixsort_amd.dist import ShardedSorter, use_fine
    if args.whole_keys or args.low_halves:
        import inplacemsdradixsort_amd.dist as _d
        _d.FINE_HIST = False
        _d.FINE_LOW16 = not args.whole_keys

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    N = world
    multi = N > 1 or args.one_rank_exchange      # the sharded loop runs (one rank: rehearsal over the real backend)
    if multi and CONFIGS[args.config]["dtype"] != "u32":
        raise SystemExit("the multi-GPU path shards u32 keys (configs c2 / c3)")
    rehearsal = (N > 1 and args.backend == "gloo") or args.one_rank_exchange
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal and args.backend == "gloo":
            import torch.distributed as tdist
            from inplacemsdradixsort_amd.dist import HostStagedDist
            tdist.init_process_group("gloo")
            dist = HostStagedDist(tdist)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    # N > 1: the local work runs on a stream of its own, not on the default stream.  HIP multiplexes streams onto a few
    # hardware queues; on the GPU box the default stream and the stream RCCL launches its kernels on shared ONE queue, and
    # the exchange ran strictly between the compute kernels instead of under them (rocprofv3 kernel trace of the one-rank
    # rehearsal, profiles/r03_onerank_trace_*.txt).  --compute-stream default restores the old behaviour for comparison.
    side = None
    if multi and args.compute_stream != "default":
        side = torch.cuda.Stream(device=local_rank, priority=-1 if args.compute_stream == "high" else 0)
        torch.cuda.set_stream(side)
    ctx = MsdContext(local_rank)
    ctx.use_torch_stream()
    n = 1 << args.logn
    W, K = args.warmup, args.steps
    cfg, pairs, tdt, gen, sort = make_workload(ctx, torch, args.config, rank, n)

    # ---- inputs for every step, resident before the clock starts
    bufs, rids = [], []
    for s in range(W + K):
        t = torch.empty(n, dtype=tdt, device="cuda")
        gen(t, s)
        bufs.append(t)
        rids.append(t.clone() if pairs else None)   # rid = key, the reference's check(..., same=1) convention (src/msb_64.c:2461)
    ctx.reserve(n + n // 8 if multi else n, cfg["key"], cfg["val"])   # (N > 1: a rank may receive up to n / 8 more than it sent)
    checks0 = [ctx.check(t) for t in bufs]          # (violations, sum, xor) of every input
    # N > 1: receive buffers with 12.5 % slack (the reference's fudge); the exchange of step s runs (RCCL stream, xGMI)
    # while step s-1 is finished locally -- inplacemsdradixsort_amd.dist.ShardedSorter.
    # Fine scheme (shards of >= 2^27 keys): the shard is ordered by its top 16 bits before the exchange, the counting leaf
    # reads what arrived in one of TWO receive buffers (one being filled, one being read) and writes a work buffer -- one
    # per timed step when the memory allows, so that EVERY timed step's output is still there when the clock has stopped
    # and is verified.  Coarse scheme (small shards): round 2's path (gather + segmented sort at <= 4 ranks, sort on
    # 32 - log2 N bits where the keys arrived at 8).
    fine = multi and use_fine(n, N, True, args.scheme, args.one_rank_exchange)
    cap = n + n // 8
    nkeep = 0
    recv = work = None
    if multi:
        free_b, _ = torch.cuda.mem_get_info()
        nkeep = int(max(2, min(K, 24, (free_b - (6 << 30)) // (cap * 4) - 2)))
        gathered = fine or N <= 4
        recv = [torch.empty(cap, dtype=torch.int32, device="cuda") for _ in range(2 if gathered else nkeep)]
        work = [torch.empty(cap, dtype=torch.int32, device="cuda") for _ in range(nkeep)] if gathered else None
    sorter = ShardedSorter(ctx, dist, N, recv, work_bufs=work, scheme=args.scheme, _force_exchange=args.one_rank_exchange) if multi else None

    def run_steps(lo, hi):
        if not multi:
            for i in range(lo, hi):
                sort(bufs[i], rids[i])
            return [(i, bufs[i]) for i in range(lo, hi)]
        outs = []
        for i in range(lo, hi):
            sorter.submit(bufs[i])
            if i > lo:
                outs.append((i - 1, sorter.collect()))
        if hi > lo:
            outs.append((hi - 1, sorter.collect()))
        return outs

    run_steps(0, W)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = run_steps(W, W + K)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if multi:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- verify the timed steps' outputs (outside the clock): the library's check kernel AND an independent torch reduction
    verified, steps_verified, fail_detail = True, 0, []
    if not multi:
        for i, o in outs:
            v, s_, x_ = ctx.check(o, rids[i]) if pairs else ctx.check(o)   # pairs: order and key == rid
            tv, ts, tm = torch_check(torch, o, rids[i] if pairs else None)
            verified &= (v == 0 and s_ == checks0[i][1] and x_ == checks0[i][2] and tv == 0 and ts == checks0[i][1] and tm == 0)
            steps_verified += 1
    else:
        # Every output still in a buffer (all timed steps when the memory allowed one work buffer each, else the last
        # nkeep): sorted on its rank (two independent checks), key sum and xor of all ranks' outputs == those of all
        # ranks' inputs, rank r's keys carry top bits r and follow rank r-1's (what the reference's check() verifies
        # across its numa arrays, src/msb_64.c:2432-2505).
        lg = N.bit_length() - 1
        M64 = (1 << 64) - 1
        for i, o in outs[-nkeep:]:
            v, s_, x_ = ctx.check(o)
            tv, ts, _ = torch_check(torch, o)
            v += tv + (0 if ts == s_ else 1)
            cnt = o.numel()
            lo_k = (int(o[0].item()) & 0xFFFFFFFF) if cnt else -1
            hi_k = (int(o[-1].item()) & 0xFFFFFFFF) if cnt else -1
            # sums travel as two 32-bit halves in int64 (exact), xors and boundaries as they are
            row = torch.tensor([v, cnt, s_ & 0xFFFFFFFF, s_ >> 32, x_ & 0xFFFFFFFF, x_ >> 32,
                                checks0[i][1] & 0xFFFFFFFF, checks0[i][1] >> 32, checks0[i][2] & 0xFFFFFFFF, checks0[i][2] >> 32,
                                lo_k, hi_k], dtype=torch.int64, device="cuda")
            rows = [torch.empty_like(row) for _ in range(N)]
            dist.all_gather(rows, row)
            R = [r_.tolist() for r_ in rows]
            ok = all(r_[0] == 0 for r_ in R) and sum(r_[1] for r_ in R) == N * n
            s_out = sum(r_[2] + (r_[3] << 32) for r_ in R) & M64
            s_in = sum(r_[6] + (r_[7] << 32) for r_ in R) & M64
            x_out = x_in = 0
            for r_ in R:
                x_out ^= r_[4] | (r_[5] << 32)
                x_in ^= r_[8] | (r_[9] << 32)
            ok &= s_out == s_in and x_out == x_in
            prev = -1
            for g, r_ in enumerate(R):
                if r_[1]:
                    ok &= (r_[10] >> (32 - lg)) == g and (r_[11] >> (32 - lg)) == g and r_[10] > prev
                    prev = r_[11]
            verified &= bool(ok)
            if not ok:   # say what failed (rank 0's view), for the record
                fail_detail.append({"step": i, "violations_per_rank": [r_[0] for r_ in R], "keys_out": sum(r_[1] for r_ in R), "keys_in": N * n,
                                    "sum_ok": s_out == s_in, "xor_ok": x_out == x_in, "first_last_per_rank": [(r_[10], r_[11]) for r_ in R]})
            steps_verified += 1
        flag = torch.tensor([1 if verified else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        verified = bool(flag.item())

    roofline, real, copy_gbps, mg = None, None, None, None
    if not multi:
        r = bufs[0].clone() if pairs else None
        roofline, real = profile_one_sort(ctx, torch, cfg, args.config, n, gen, sort, bufs[0], r)
        del r
        # ---- achievable copy rate on this device, same run (second denominator)
        a, b = bufs[0], bufs[1] if len(bufs) > 1 else torch.empty_like(bufs[0])
        b.copy_(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        copy_gbps = 5 * 2 * a.numel() * a.element_size() / (e0.elapsed_time(e1) * 1e-3) / 1e9
    else:
        # ---- N > 1: what a step consists of, measured apart after the clock has stopped (3 repetitions, max over ranks):
        # the local work alone (before + after the exchange, nothing in flight) and the exchange alone.  The timed loop
        # overlaps them (ms_per_step); their sum is what it would cost without the overlap.
        def max_over_ranks(ms):
            x = torch.tensor([ms], dtype=torch.float64, device="cuda")
            dist.all_reduce(x, op=dist.ReduceOp.MAX)
            return float(x.item())

        pre_ms, post_ms, xch_ms, xch_bytes = [], [], [], 0
        for rep in range(3):
            gen(bufs[0], 300 + rep)
            torch.cuda.synchronize()
            dist.barrier()
            t1 = time.perf_counter()
            sorter.submit(bufs[0])                      # pre-exchange pass + count exchange (blocks: the counts come to the host) + the all-to-all is STARTED
            t2 = time.perf_counter()                    # (no device synchronisation here: it would wait for the exchange as well)
            out_view, handles, _ = sorter._pending[0]
            for h in handles or []:
                h.wait()
            torch.cuda.synchronize()
            dist.barrier()
            t3 = time.perf_counter()
            sorter.collect()
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            pre_ms.append((t2 - t1) * 1e3)
            xch_ms.append((t3 - t2) * 1e3)
            post_ms.append((t4 - t3) * 1e3)
            xch_bytes = out_view.numel() * out_view.element_size()   # (fine scheme: the keys' low halves, 2 bytes each)
        k_algo = algo_bytes_per_elem(cfg)
        sec = dt / K
        mg = {"scheme": (f"fine (top 16 bits before the exchange, {sorter.last_format} travel, counting leaf over what arrived)"
                         if fine else "coarse (top digit before the exchange)"),
                 "local_before_exchange_ms": round(max_over_ranks(min(pre_ms)), 3),
                 "local_after_exchange_ms": round(max_over_ranks(min(post_ms)), 3),
                 "exchange_alone_ms": round(max_over_ranks(min(xch_ms)), 3),
                 "note_before": "includes the count exchange (one all-gather + one small device-to-host copy) and the launch of the all-to-all; "
                                "the all-to-all then runs alone (not overlapped) and is what exchange_alone_ms times",
                 "exchange_bytes_received_per_gpu": int(xch_bytes),
                 "exchange_GBps_per_gpu_alone": round(xch_bytes / (max(min(xch_ms), 1e-6) * 1e-3) / 1e9, 1)}
        # the `roofline` object of an N > 1 line: the whole per-GPU step against the HBM peak (algorithmic bytes as for one
        # GPU: the rank sorts 2^logn keys per step whatever their origin); the exchange is reported beside it
        roofline = {"bound": "hbm", "kernel": "whole per-GPU step (pre-exchange pass + counting leaf; exchange overlapped)",
                    "achieved": round(n * k_algo / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(n * k_algo / sec / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                    "algorithmic_bytes_per_launch": int(n * k_algo), "avg_launch_us": round(sec * 1e6, 1), "multi_gpu": mg}

    total = N * n * K
    value = total / dt / 1e9
    unit = "Gtuples/s" if pairs else "Gkeys/s"
    headline = "Gkeys/s + achieved HBM GB/s, 2^30 uniform u32 keys, 1/2/4/8 MI355X"
    if multi:
        how = (f", range-partitioned over {N} GPUs by one RCCL all-to-all per step (overlapped with the previous step's local work; "
               + (f"fine scheme: shard ordered by its top 16 bits before the exchange, {sorter.last_format} travel, one counting pass over what arrived after it)" if fine
                  else (f"the arrived runs are gathered bucket-major and sorted as {256 // N} segments on 24 bits)" if N <= 4
                        else f"the arrived keys are sorted on {32 - N.bit_length() + 1} bits)")))
    else:
        how = ""
    out = {
        "metric": headline if args.config == "c2" else f"{unit} + achieved HBM GB/s, 2^{args.logn} {cfg['title']}, 1 MI355X",
        "value": round(value, 3), "unit": unit if args.config != "c2" else "Gkeys/s", "n_gpus": N, "steps": K, "warmup": W,
        "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": cfg["dtype"], "data": "synthetic" + (" (REHEARSAL: one rank exchanging with itself over RCCL)" if args.one_rank_exchange else
                                                       " (REHEARSAL: gloo via host memory, all ranks on one GPU)" if rehearsal else ""),
        "config": {"workload": f"2^{args.logn} {cfg['title']} per GPU, in-place MSD radix sort, 8-bit digits" + how,
                   "config_id": args.config, "elements_per_gpu": n, "passes": cfg["passes_note"],
                   "verified": bool(verified), "steps_verified": steps_verified, **({"verify_failures": fail_detail[:4]} if fail_detail else {}),
                   "verified_by": "msd_check_* (device) and an independent chunked torch reduction (order, sum" + (", key == rid)" if pairs else ")"),
                   "workspace_bytes": ctx.workspace_bytes},
        "whole_sort": whole_sort_block(cfg, n, dt / K, real, copy_gbps),
        "roofline": roofline,
    }
    # ---- the other single-GPU configs in the same record (default run only): headline buffers are freed first
    if not multi and args.config == "c2" and not args.no_other_configs and args.logn == 30:
        a = b = None
        del bufs, rids, outs, a, b
        torch.cuda.empty_cache()
        others = {}
        for cid in ("c3", "c5a", "c5b"):
            try:
                others[cid] = run_other_config(cid, args.logn)
            except Exception as e:  # a failing side config must not take the headline line with it
                others[cid] = {"error": f"{type(e).__name__}: {e}"}
        out["other_configs"] = others
    if rank == 0:
        out["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(args.cpu_logn, args.cpu_logn_1t)
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()            # the other ranks wait for rank 0's CPU baseline
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
