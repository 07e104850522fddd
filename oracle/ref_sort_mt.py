"""Times the reference's own multi-threaded sort() (src/msb_64.c:2261) from oracle/_ref.
TEST / BENCH INFRASTRUCTURE ONLY -- run as a subprocess because the reference is fragile
(out-of-bounds read on every run, crashes outside a narrow envelope):

    python oracle/ref_sort_mt.py LOGN [NUMA] [REPS]   ->  one JSON line on stdout

Envelope used (the one SURVEY.md section 8c found reliable): threads = 64, numa = 2 "virtual"
nodes, fudge = 2.0, full-width keys (u32 << 32), rid = key, buffers from the reference's mamalloc.
Every run's output is verified (global order across the arrays, key == rid, sum and xor of keys).
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != HERE]  # `oracle` must resolve to the package dir
sys.path.insert(0, os.path.dirname(HERE))


def main():
    logn = int(sys.argv[1])
    numa = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    fudge = 2.0
    from oracle import oracle as O
    L = O.ref().lib
    n = 1 << logn
    per = n // numa
    cap = int(per * fudge)
    u64p = C.POINTER(C.c_uint64)

    def arr(nbytes):
        p = L.mamalloc(nbytes)
        assert p, "mamalloc failed"
        return np.frombuffer((C.c_uint8 * nbytes).from_address(p), dtype=np.uint64)

    keys = [arr(cap * 8) for _ in range(numa)]
    rids = [arr(cap * 8) for _ in range(numa)]
    times, ok_all, phases = [], True, None
    for rep in range(reps):
        s0 = x0 = 0
        for a in range(numa):
            k32 = O.gen_uniform_u32(per, seed=0x5EED0001 + rep, first=a * per).astype(np.uint64)
            k = k32 << np.uint64(32)
            keys[a][:per] = k
            rids[a][:per] = k
            s0 = (s0 + int(k.sum(dtype=np.uint64))) & (2**64 - 1)
            x0 ^= int(np.bitwise_xor.reduce(k))
        size = np.array([per] * numa, dtype=np.uint64)
        KA = (u64p * numa)(*[a.ctypes.data_as(u64p) for a in keys])
        RA = (u64p * numa)(*[a.ctypes.data_as(u64p) for a in rids])
        desc = (C.c_char_p * 11)()
        tm = np.zeros(10, dtype=np.uint64)
        t0 = time.perf_counter()
        L.sort(KA, RA, size.ctypes.data_as(u64p), 64, numa, C.c_double(fudge), desc, tm.ctypes.data_as(u64p))
        dt = time.perf_counter() - t0
        # verify: sizes, global order, key == rid, checksums
        ok = int(size.sum()) == n
        s1 = x1 = 0
        prev = 0
        for a in range(numa):
            m = int(size[a])
            k = keys[a][:m]
            if m:
                ok &= bool((np.diff(k.view(np.int64)) >= 0).all()) if int(k.max()) < 2**63 else bool((k[1:] >= k[:-1]).all())
                ok &= int(k[0]) >= prev
                prev = int(k[-1])
                ok &= bool((k == rids[a][:m]).all())
                s1 = (s1 + int(k.sum(dtype=np.uint64))) & (2**64 - 1)
                x1 ^= int(np.bitwise_xor.reduce(k))
        ok &= (s0 == s1) and (x0 == x1)
        if os.environ.get("REF_MT_DEBUG"):
            print("rep", rep, "sizes", size.tolist(), "sum", s0 == s1, "xor", x0 == x1, "ok", ok, file=sys.stderr)
        ok_all &= bool(ok)
        times.append(dt)
        phases = {desc[i].decode().strip(): int(tm[i]) for i in range(10) if desc[i]}
    med = sorted(times)[len(times) // 2]
    print(json.dumps({"n": n, "numa": numa, "threads": 64, "fudge": fudge, "reps": reps, "seconds": times, "median_s": med,
                      "gkeys_per_s": n / med / 1e9, "verified": ok_all, "logical_cpus": os.cpu_count(),
                      "phases_us_last": phases}))


if __name__ == "__main__":
    main()
