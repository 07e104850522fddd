"""CPU tests of the parity oracle: against the golden vectors the reference
produced (tests/golden, made by make_golden.py) and, when the reference build is
present (oracle/_ref), against the reference itself on fresh inputs."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def test_plans_match_reference_golden():
    for e in json.load(open(os.path.join(G, "golden_plans.json"))):
        p, rb, bf = O.schedule_passes(e["size"], e["bits"])
        assert (p, rb, bf) == (e["passes"], e["radix_bits"], e["buffered"]), e


def test_plans_quoted_in_survey():
    # SURVEY.md section 3.2 (probe of the reference): bits=32
    assert O.schedule_passes(1 << 20, 32)[1:] == ([8, 10, 14], [1, 0, -1])
    assert O.schedule_passes(1 << 24, 32)[1:] == ([3, 9, 10, 10], [0, 1, 0, -1])
    assert O.schedule_passes(1 << 26, 32)[1:] == ([9, 5, 10, 8], [1, 0, 0, -1])
    assert O.schedule_passes(1 << 30, 32)[1:] == ([9, 9, 10, 4], [1, 1, 0, -1])
    assert O.schedule_passes(1 << 33, 58)[1:] == ([7, 7, 7, 10, 27], [1, 1, 1, 0, -1])


@pytest.mark.parametrize("name", ["golden_u32_4096.npz", "golden_zipf_4096.npz"])
def test_u32_golden(name):
    g = np.load(os.path.join(G, name))
    assert (O.sort_u32(g["keys_in"]) == g["keys_out"]).all()
    assert (np.sort(g["keys_in"]) == g["keys_out"]).all()


def test_pairs_golden_bit_exact():
    g = np.load(os.path.join(G, "golden_pairs_8192.npz"))
    k, r = O.sort_pairs_u64(g["keys_in"], g["rids_in"], 64)
    assert (k == g["keys_out"]).all()
    # the single-thread core is deterministic: the restatement reproduces its tie order too
    assert (r == g["rids_out"]).all()


def test_histogram_golden():
    g = np.load(os.path.join(G, "golden_hist.npz"))
    keys = g["keys"]
    for shift, rb in ((24, 8), (0, 8), (13, 11), (20, 5)):
        exp = g[f"h_s{shift}_r{rb}"]
        assert (O.histogram(keys, shift, rb) == exp).all()
        assert (O.histogram(keys.astype(np.uint64), shift, rb) == exp).all()
        assert int(exp.sum()) == keys.size


def test_partition_golden():
    g = np.load(os.path.join(G, "golden_partition.npz"))
    k, r, h = O.partition(g["keys"], g["rids"], 24, 8, buffered=True)
    assert (h == g["buf_hist"]).all() and (k == g["buf_keys"]).all() and (r == g["buf_rids"]).all()
    k, r, h = O.partition(g["keys"], g["rids"], 27, 5, buffered=False)
    assert (h == g["ip_hist"]).all() and (k == g["ip_keys"]).all() and (r == g["ip_rids"]).all()


def test_c1_digest():
    d = json.load(open(os.path.join(G, "golden_c1_digest.json")))
    k = O.gen_uniform_u32(d["n"], seed=d["seed"])
    out = O.sort_u32(k)
    assert hashlib.sha256(out.tobytes()).hexdigest() == d["sha256_sorted"]
    assert int(out.astype(np.uint64).sum()) == d["sum"]
    assert out[:4].tolist() == d["first"] and out[-4:].tolist() == d["last"]


@pytest.mark.parametrize("n", [0, 1, 2, 20, 21, 6500, 6501, 70000])
@pytest.mark.parametrize("kind", ["uniform", "zipf", "dup", "const", "reverse"])
def test_oracle_sorts(n, kind):
    rng = np.random.default_rng(n + 1)
    if kind == "uniform":
        k = O.gen_uniform_u32(n, seed=n)
    elif kind == "zipf":
        k = O.gen_zipf_u32(n, seed=n)
    elif kind == "dup":
        k = rng.integers(0, 256, n, dtype=np.uint32) * np.uint32(0x01010101)
    elif kind == "const":
        k = np.full(n, 77, np.uint32)
    else:
        k = np.sort(O.gen_uniform_u32(n, seed=3))[::-1].copy()
    assert (O.sort_u32(k) == np.sort(k)).all()
    k64 = k.astype(np.uint64) << np.uint64(32) | k.astype(np.uint64)
    assert (O.sort_u64(k64) == np.sort(k64)).all()


def test_check_reports_violations():
    a = np.array([1, 2, 3], np.uint64)
    b = np.array([3, 5, 4], np.uint64)
    s, x, bad = O.check([a, b], [a, b], True)
    assert s == 18 and x == (1 ^ 2 ^ 3 ^ 3 ^ 5 ^ 4) and bad == 1
    s, x, bad = O.check([b, a], None, False)
    assert bad == 2  # 5>4 inside b, 4>1 across the array boundary (the reference skips slice seams, src/msb_64.c:2458)


def test_generators_are_the_survey_ones():
    # key[i] = splitmix64(seed + i) >> 32
    def sm(x):
        x = (x + 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        return z ^ (z >> 31)
    k = O.gen_uniform_u32(5, seed=0x5EED0001, first=10)
    assert k.tolist() == [sm(0x5EED0001 + 10 + i) >> 32 for i in range(5)]
    z = O.gen_zipf_u32(1 << 16)
    assert 0.70 < (z < (1 << 24)).mean() < 0.80  # ~75 % of keys have a zero top byte


needs_ref = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (no /root/reference here)")


@needs_ref
@pytest.mark.parametrize("n", [1, 21, 4096, 6501, 1 << 16, (1 << 18) + 7])
@pytest.mark.parametrize("bits,kind", [(32, "u32"), (64, "u64"), (58, "u58"), (32, "dup"), (32, "skew")])
def test_restatement_equals_reference(n, bits, kind):
    rng = np.random.default_rng(n * 7 + bits)
    if kind == "u32":
        k = rng.integers(0, 2**32, n, dtype=np.uint64)
    elif kind == "u64":
        k = rng.integers(0, 2**64, n, dtype=np.uint64)
    elif kind == "u58":
        k = rng.integers(0, 2**58, n, dtype=np.uint64) | np.uint64(0x2A << 58)
    elif kind == "dup":
        k = rng.integers(0, 256, n, dtype=np.uint64) * np.uint64(0x01010101)
    else:
        k = (rng.random(n) ** 8 * 2**32).astype(np.uint64)
    r = np.arange(n, dtype=np.uint64)
    ok, orr = O.sort_pairs_u64(k, r, bits)
    rk, rr = O.ref_sort_pairs_u64(k, r, bits)
    assert (ok == rk).all() and (ok == np.sort(k)).all()
    assert (orr == rr).all()


@needs_ref
def test_histogram_and_partition_equal_reference():
    k = O.gen_uniform_u64(50000, seed=5)
    r = np.arange(k.size, dtype=np.uint64)
    for shift, rb in ((56, 8), (0, 8), (30, 9), (61, 3)):
        assert (O.histogram(k, shift, rb) == O.ref_histogram(k, shift, rb)).all()
    for shift, rb, buf in ((56, 8, True), (55, 9, True), (59, 5, False), (61, 3, False)):
        a = O.partition(k, r, shift, rb, buf)
        b = O.ref_partition(k, r, shift, rb, buf)
        assert all((x == y).all() for x, y in zip(a, b))


@pytest.mark.parametrize("kind,parts", [("zipf", 2), ("zipf", 8), ("zipf", 64), ("uniform", 8), ("dup", 8), ("const", 4)])
def test_splitter_front_end_equals_reference(kind, parts):
    """extract_delimiters (src/msb_64.c:1304-1322) restated in the oracle == the reference's own, on sorted samples
    drawn the way the build draws them (mulhi(rand64, n) indices, :1511-1521); the range function (:188-204) agrees
    with numpy's lower bound."""
    if not O.have_ref():
        pytest.skip("needs oracle/_ref")
    n, m = 200000, 5000
    if kind == "zipf":
        k = O.gen_zipf_u32(n, seed=11)
    elif kind == "uniform":
        k = O.gen_uniform_u32(n, seed=12)
    elif kind == "dup":
        k = (O.gen_uniform_u32(n, seed=13) % 5).astype(np.uint32) * 1000
    else:
        k = np.full(n, 77, np.uint32)
    s = np.sort(O.sample_u32(k, m, seed=99)).astype(np.uint64)
    d = O.extract_delimiters(s, parts)
    assert (d == O.ref_extract_delimiters(s, parts)).all()
    cnt = O.range_histogram_u32(k, d)
    assert int(cnt.sum()) == n and (cnt == np.bincount(O.range_of_u32(k, d), minlength=parts)).all()


def test_mt19937_64_restatement_equals_reference_rand_c():
    """src/rand.c:47-86 restated in the oracle == the reference's rand64_init / rand64_next (across a block boundary)."""
    if not O.have_ref():
        pytest.skip("needs oracle/_ref")
    for seed in (0, 1, 0x5EED0001, 2**64 - 1):
        assert (O.mt19937_64(1000, seed) == O.ref_mt19937_64(1000, seed)).all()
    # known-answer: std::mt19937_64's 10000th output for the default seed 5489 (the C++ standard pins it)
    assert int(O.mt19937_64(10000, 5489)[-1]) == 9981545732273789042
