"""Build libadsb_hip.so (HIP kernels + C ABI) for gfx950, in-tree.

    python -m dump1090_rs_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU
box with the tree.  -ffp-contract=off is load-bearing: the magnitude pipeline
(reference src/utils.rs:53-55) has one separately rounded multiply that must not
be fused.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libadsb_hip.so"
SOURCES = [CSRC / "adsb_scan_fast.hip", CSRC / "adsb_scan_simple.hip", CSRC / "adsb_aux.hip",
           *(CSRC / f for f in ("adsb_context.cpp", "adsb_pass.cpp", "adsb_collect.cpp", "adsb_ring.cpp",
                                "adsb_shard.cpp", "adsb_multi.cpp", "adsb_selftest.cpp", "adsb_replay_host.cpp"))]
HEADERS = [CSRC / "adsb_ctx.h", CSRC / "adsb_device.h", CSRC / "adsb_dev_common.h", CSRC / "adsb_scan_geometry.h",
           CSRC / "adsb_tables.h", CSRC / "adsb_tail_dev.h", CSRC / "adsb_record.h", CSRC / "adsb_replay_host.h", CSRC / "mode_s_host.hpp", PKG.parent / "include" / "adsb_hip.h"]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
    "-Wall", "-Wextra", "-x", "hip",
]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libadsb_hip.so cannot be built")
    return exe


def stale() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + HEADERS + [Path(__file__)])


FEED = PKG / "adsb_feed"
FEED_SRC = CSRC / "adsb_feed.cpp"


def build_library(force: bool = False, verbose: bool = False) -> Path:
    if not force and not stale() and FEED.exists() and FEED.stat().st_mtime >= FEED_SRC.stat().st_mtime:
        return LIB
    # ADSB_HIPCC_FLAGS: extra flags, e.g. -DADSB_KERNEL_ACCT for the in-kernel phase accounting
    extra = os.environ.get("ADSB_HIPCC_FLAGS", "").split()
    cmd = [hipcc(), *FLAGS, *extra, *map(str, SOURCES), "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    # the file/pipe -> "*hex;" feeder (host only, plain C++ over the C ABI); finds the library
    # next to itself
    cmd = ["g++", "-O3", "-std=c++17", "-Wall", "-Wextra", str(FEED_SRC), "-o", str(FEED),
           f"-L{PKG}", "-ladsb_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


ABI_HOST_SRC = PKG.parent / "tests" / "abi_host.c"
ABI_HOST = PKG.parent / "tests" / "abi_host"


def build_abi_host(verbose: bool = False, force: bool = False) -> Path:
    """tests/abi_host.c: the reference's test routine from a plain-C host over include/adsb_hip.h
    (gcc, no Python in between); the GPU tests run it against the golden frames.  Test
    infrastructure: built by __graft_entry__.build() and the test fixtures, never by
    build_library() -- the product does not depend on the test tree or on gcc."""
    if not force and ABI_HOST.exists() and ABI_HOST.stat().st_mtime >= max(
            ABI_HOST_SRC.stat().st_mtime, LIB.stat().st_mtime if LIB.exists() else 0,
            (PKG.parent / "include" / "adsb_hip.h").stat().st_mtime):
        return ABI_HOST
    cmd = ["gcc", "-O2", "-std=c11", "-Wall", "-Wextra", f"-I{PKG.parent / 'include'}", str(ABI_HOST_SRC),
           "-o", str(ABI_HOST), f"-L{PKG}", "-ladsb_hip", f"-Wl,-rpath,{PKG}", "-Wl,-rpath,$ORIGIN/../dump1090_rs_amd"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return ABI_HOST


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, verbose=True)
    if ABI_HOST_SRC.exists():
        build_abi_host(verbose=True)
    print(LIB)
