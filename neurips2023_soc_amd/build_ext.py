"""Compile libsoc_hip.so (the four gfx950 kernels behind include/soc_hip.h) in-tree with hipcc.

    python -m neurips2023_soc_amd.build_ext [--force]

hipcc cross-compiles for gfx950 without a GPU; the built .so is git-ignored but travels to the
GPU box with the repo snapshot.  No CUDA shims, no hipify, gfx950 only.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libsoc_hip.so")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-Wall",
         "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libsoc_hip.so (no CPU fallback exists)")
    # one builder at a time (the ranks of a multi-GPU job all load the library): the others wait on the lock and
    # then find the library fresh
    import fcntl
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not stale():
                return LIB
            tmp = f"{LIB}.{os.getpid()}.tmp"
            cmd = [hipcc, *FLAGS, "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", tmp, *sources()]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            os.replace(tmp, LIB)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
