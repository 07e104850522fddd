"""Generates tests/golden/*.npz and golden_plans.json from the REFERENCE ITSELF.

Run in the build container (needs /root/reference, compiled by oracle/Makefile into
oracle/_ref/libref_msb64.so).  The reference ships no golden vectors (SURVEY.md
section 8c), so these are outputs of the reference's own functions on seeded inputs:

* golden_plans.json      schedule_passes(size, bits)            src/msb_64.c:1334
* golden_u32_4096.npz    single-thread core on 4096 u32 keys    src/msb_64.c:2232-2244
* golden_pairs_8192.npz  same on 8192 (u64 key, rid=index) tuples (pair order as the
                         reference's deterministic single-thread core leaves it)
* golden_hist.npz        histogram(keys, shift, radix_bits)     src/msb_64.c:701
* golden_partition.npz   histogram + partition_ip / partition_ip_buf  :740 / :785
* golden_c1_digest.json  sha256 of the sorted 2^20 uniform u32 keys of config C1
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def main():
    assert O.have_ref(), "reference library not built"
    plans = []
    for bits in (32, 58, 64):
        for n in (1, 4, 20, 21, 100, 6500, 6501, 13000, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 23, 1 << 24,
                  1 << 26, 1 << 30, 1 << 33):
            p, rb, bf = O.ref_schedule_passes(n, bits)
            plans.append({"size": n, "bits": bits, "passes": p, "radix_bits": rb, "buffered": bf})
    json.dump(plans, open(os.path.join(HERE, "golden_plans.json"), "w"), indent=0)

    k = O.gen_uniform_u32(4096, seed=0x5EED0001)
    np.savez_compressed(os.path.join(HERE, "golden_u32_4096.npz"), keys_in=k, keys_out=O.ref_sort_u32(k))

    kz = O.gen_zipf_u32(4096, seed=0x5EED0003)
    np.savez_compressed(os.path.join(HERE, "golden_zipf_4096.npz"), keys_in=kz, keys_out=O.ref_sort_u32(kz))

    k64 = O.gen_uniform_u64(8192, seed=0x5EED0005)
    r64 = np.arange(8192, dtype=np.uint64)
    ko, ro = O.ref_sort_pairs_u64(k64, r64, 64)
    np.savez_compressed(os.path.join(HERE, "golden_pairs_8192.npz"), keys_in=k64, rids_in=r64, keys_out=ko, rids_out=ro)

    kk = O.gen_uniform_u32(1 << 14, seed=7).astype(np.uint64)
    np.savez_compressed(os.path.join(HERE, "golden_hist.npz"), keys=kk.astype(np.uint32),
                        h_s24_r8=O.ref_histogram(kk, 24, 8), h_s0_r8=O.ref_histogram(kk, 0, 8),
                        h_s13_r11=O.ref_histogram(kk, 13, 11), h_s20_r5=O.ref_histogram(kk, 20, 5))

    kp = O.gen_uniform_u32(1 << 13, seed=11).astype(np.uint64)
    rp = np.arange(kp.size, dtype=np.uint64)
    a_k, a_r, a_h = O.ref_partition(kp, rp, 24, 8, buffered=True)
    b_k, b_r, b_h = O.ref_partition(kp, rp, 27, 5, buffered=False)
    np.savez_compressed(os.path.join(HERE, "golden_partition.npz"), keys=kp, rids=rp,
                        buf_keys=a_k, buf_rids=a_r, buf_hist=a_h, ip_keys=b_k, ip_rids=b_r, ip_hist=b_h)

    c1 = O.gen_uniform_u32(1 << 20, seed=0x5EED0001)
    out = O.ref_sort_u32(c1)
    json.dump({"n": 1 << 20, "seed": 0x5EED0001, "sha256_sorted": hashlib.sha256(out.tobytes()).hexdigest(),
               "sum": int(out.astype(np.uint64).sum()), "first": out[:4].tolist(), "last": out[-4:].tolist()},
              open(os.path.join(HERE, "golden_c1_digest.json"), "w"))
    print("golden vectors written")


if __name__ == "__main__":
    main()
