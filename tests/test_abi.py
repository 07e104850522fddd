"""The C-ABI library loads here (no GPU) and exports every symbol include/*.h
declares; creating a context without a GPU fails loudly (no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b([a-z_0-9]+)\s*\(", text)) - {"defined"}


def test_every_declared_symbol_is_exported():
    from inplacemsdradixsort_amd import _lib
    L = _lib.load()
    declared = _declared("msd_radix_hip.h") | _declared("msb_64.h")
    assert {"sort", "mamalloc", "check", "msd_sort_u32", "msd_histogram_u32", "msd_exclusive_scan_u64"} <= declared
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert b"gfx950" in L.msd_version()


def test_every_declared_sharded_symbol_is_exported():
    """include/msd_sharded_hip.h: the multi-GPU entry points live in a library of their own that links the single-GPU
    library and RCCL; it loads here (no GPU) and exports what the header declares."""
    from inplacemsdradixsort_amd import _build, _lib
    R = _lib.load_rccl()
    declared = _declared("msd_sharded_hip.h")
    assert {"msd_shard_create", "msd_sort_u32_sharded", "msd_sort_pairs_u64_sharded", "msd_sort_u32_multi"} <= declared
    missing = [s for s in sorted(declared) if not hasattr(R, s)]
    assert not missing, missing
    assert declared == set(_lib.RCCL_EXPORTS), declared ^ set(_lib.RCCL_EXPORTS)
    import subprocess
    needed = subprocess.run(["readelf", "-d", _build.RCCL_LIB], capture_output=True, text=True).stdout
    assert "librccl.so" in needed and "libinpmsdradix_hip.so" in needed
    # ... and the single-GPU library does not depend on RCCL
    assert "rccl" not in subprocess.run(["readelf", "-d", _build.LIB], capture_output=True, text=True).stdout


def test_mamalloc_is_64_byte_aligned():
    from inplacemsdradixsort_amd import _lib
    L = _lib.load()
    for sz in (1, 100, 4096, 1 << 20):
        p = L.mamalloc(sz)
        assert p and p % 64 == 0  # src/msb_64.c:111-115


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from inplacemsdradixsort_amd import MsdContext, MsdError
    with pytest.raises(MsdError):
        MsdContext(0)


def test_product_never_touches_the_oracle():
    bad = []
    pkg = os.path.join(ROOT, "inplacemsdradixsort_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                t = open(os.path.join(dp, f)).read()
                if re.search(r"\boracle\b|liborc|_ref/|msd_oracle", t):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_code_object_is_gfx950_only():
    from inplacemsdradixsort_amd import LIB
    data = open(LIB, "rb").read()
    assert b"gfx950" in data
    for other in (b"gfx90a", b"gfx942", b"sm_80"):
        assert other not in data


def _build_c_driver(tmp_path):
    import subprocess
    exe = str(tmp_path / "benchmark_msb_64")
    lib_dir = os.path.join(ROOT, "inplacemsdradixsort_amd")
    from inplacemsdradixsort_amd import build
    build()
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "benchmark_msb_64.c"), "-L" + lib_dir, "-linpmsdradix_hip",
                           "-Wl,-rpath," + lib_dir, "-o", exe])
    return exe


def test_plain_c_caller_compiles_and_links_against_the_abi(tmp_path):
    """A C caller of the reference library (include/msb_64.h: sort, mamalloc, check) builds unchanged."""
    assert os.path.exists(_build_c_driver(tmp_path))


@pytest.mark.gpu
def test_plain_c_caller_runs(tmp_path):
    import subprocess
    out = subprocess.run([_build_c_driver(tmp_path), "22", "2"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "checksum ok" in out.stdout and "Total sort() time" in out.stdout
