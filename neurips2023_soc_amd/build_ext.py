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
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
OBJ = os.path.join(CSRC, "_obj")           # per-source objects (git-ignored): a changed kernel file rebuilds alone


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))


def obj_of(src: str) -> str:
    return os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")


def stale_objects():
    """Sources whose object is missing or older than the source or any header."""
    newest_header = max((os.path.getmtime(h) for h in headers()), default=0.0)
    out = []
    for src in sources():
        o = obj_of(src)
        if not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(src), newest_header):
            out.append(src)
    return out


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in sources() + headers())


def jobs() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("SOC_BUILD_JOBS", "8"))))


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libsoc_hip.so (no CPU fallback exists)")
    # one builder at a time (the ranks of a multi-GPU job all load the library): the others wait on the lock and
    # then find the library fresh
    import fcntl
    from concurrent.futures import ThreadPoolExecutor
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not stale():
                return LIB
            os.makedirs(OBJ, exist_ok=True)
            inc = ["-I", os.path.join(ROOT, "include"), "-I", CSRC]
            todo = sources() if force else stale_objects()

            def compile_one(src):
                tmp = f"{obj_of(src)}.{os.getpid()}.tmp"
                cmd = [hipcc, *FLAGS, *inc, "-c", "-o", tmp, src]
                if verbose:
                    print(" ".join(cmd), flush=True)
                subprocess.run(cmd, check=True)
                os.replace(tmp, obj_of(src))

            # the kernel files are independent translation units (no device code is shared across them): compile them side by
            # side -- the longest one (~90 s: the template forms of K23 / K24) sets the wall time, not their sum
            with ThreadPoolExecutor(max_workers=jobs()) as pool:
                list(pool.map(compile_one, todo))
            for o in glob.glob(os.path.join(OBJ, "*.o")):               # objects of deleted sources
                if o not in {obj_of(s_) for s_ in sources()}:
                    os.remove(o)
            tmp = f"{LIB}.{os.getpid()}.tmp"
            cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", tmp, *[obj_of(s_) for s_ in sources()]]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            os.replace(tmp, LIB)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
