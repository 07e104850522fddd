"""Build recipe of the HIP library (gfx950 only, in tree).

``build()`` compiles ``csrc/*.hip`` into ``libinpmsdradix_hip.so`` next to this
file with ``hipcc --offload-arch=gfx950``.  hipcc cross-compiles without a GPU,
so this also runs in the CPU-only build container.
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libinpmsdradix_hip.so")
# diagnostic build with in-kernel cycle stamps (tools/variant_run.py); never what tests or bench.py load
STAMPS_LIB = os.path.join(HERE, "libinpmsdradix_hip_stamps.so")
SOURCES = ["msd_radix.hip", "msb_64_shim.hip"]
DEPS = SOURCES + ["msd_device.hpp", "msd_direct.hpp", "msd_stream2.hpp", "msd_count16.hpp", "msd_merge16.hpp", "msd_leaf17.hpp", "msd_regpart.hpp", "msd_bigcount.hpp",
                  os.path.join("..", "..", "include", "msd_radix_hip.h"), os.path.join("..", "..", "include", "msb_64.h")]
# the multi-GPU entry points (include/msd_sharded_hip.h): a library of its own, linked against the one above and RCCL
RCCL_LIB = os.path.join(HERE, "libinpmsdradix_hip_rccl.so")
RCCL_SOURCES = ["msd_sharded.hip"]
RCCL_DEPS = RCCL_SOURCES + [os.path.join("..", "..", "include", "msd_sharded_hip.h"), os.path.join("..", "..", "include", "msd_radix_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
         "-Wno-unused-result", "-pthread"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


def _rocm_lib() -> str:
    return os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def rccl_stale() -> bool:
    if not os.path.exists(RCCL_LIB):
        return True
    t = os.path.getmtime(RCCL_LIB)
    return os.path.getmtime(LIB) > t or any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in RCCL_DEPS)


def build_rccl(force: bool = False, verbose: bool = False) -> str:
    """libinpmsdradix_hip_rccl.so: csrc/msd_sharded.hip against libinpmsdradix_hip.so (found next to it at run time,
    $ORIGIN) and librccl.  hipcc cross-compiles and links it without a GPU; RCCL is only needed when it is loaded."""
    build(force=False)
    if not force and not rccl_stale():
        return RCCL_LIB
    cmd = [_hipcc(), *FLAGS, *[os.path.join(CSRC, s) for s in RCCL_SOURCES], "-o", RCCL_LIB + ".tmp",
           "-L" + HERE, "-l:libinpmsdradix_hip.so", "-L" + _rocm_lib(), "-lrccl", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + _rocm_lib()]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(RCCL_LIB + ".tmp", RCCL_LIB)
    return RCCL_LIB


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the library if it is missing or older than its sources."""
    if not force and not stale():
        return LIB
    cmd = [_hipcc(), *FLAGS, *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


def build_stamps(extra=(), out: str = STAMPS_LIB, which: int = 1) -> str:
    """which: 1 = the direct classify kernels, 2 = count_place_kernel carry the stamps."""
    cmd = [_hipcc(), *FLAGS, f"-DMSD_STAMPS={which}", *extra, *[os.path.join(CSRC, s) for s in SOURCES], "-o", out]
    subprocess.check_call(cmd)
    return out


def build_variant(name: str, extra=()) -> str:
    """An experimental build with extra -D flags next to the product library (tools/variant_run.py)."""
    out = os.path.join(HERE, f"libinpmsdradix_hip_{name}.so")
    subprocess.check_call([_hipcc(), *FLAGS, *extra, *[os.path.join(CSRC, s) for s in SOURCES], "-o", out])
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
