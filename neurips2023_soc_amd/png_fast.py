"""ctypes binding of libsoc_host.so (include/soc_host.h): the run-length PNG encoder of the drivers' output side.

The reference saves every mask through Pillow (infer_refytb.py:269-277, infer_davis.py:285-291).  Pixels, mode ('L' / 'P'),
size and palette of the files written here are the same; the bytes of the compressed stream are not (one fixed-Huffman
deflate block with byte-run matches only): 0.1-0.3 ms of CPU per 720p mask instead of zlib's 1.9 (level 1) / 3.0 (Pillow's
default), files of 6-10 KB instead of 6 / 3 KB.  `SOC_PNG=pillow` (or `use_pillow=True`) goes back to Pillow with
`compress_level` = SOC_PNG_LEVEL (default: Pillow's own default, i.e. what the reference writes)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading
from typing import Optional, Sequence

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)
LIB_PATH = os.path.join(_PKG, "libsoc_host.so")
SOURCES = [os.path.join(_PKG, "csrc_host", "png_runs.c")]
HEADER = os.path.join(ROOT, "include", "soc_host.h")
EXPORTS = ("soc_host_abi_version", "soc_png_bound", "soc_png_encode_u8")
ABI_VERSION = 1

_lib = None
_lock = threading.Lock()


def stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(f) > t for f in SOURCES + [HEADER])


def build(force: bool = False) -> str:
    """gcc -O3 -shared -> neurips2023_soc_amd/libsoc_host.so (in-tree, git-ignored; travels to the GPU box)."""
    if not force and not stale():
        return LIB_PATH
    import fcntl
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or stale():
                tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
                subprocess.run(["gcc", "-O3", "-std=c99", "-Wall", "-Wextra", "-fPIC", "-shared", "-I",
                                os.path.join(ROOT, "include"), "-o", tmp, *SOURCES], check=True)
                os.replace(tmp, LIB_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            build()
            lib = C.CDLL(LIB_PATH)
            for name in EXPORTS:
                if not hasattr(lib, name):
                    raise RuntimeError(f"libsoc_host.so does not export {name}")
            lib.soc_host_abi_version.restype = C.c_int
            lib.soc_png_bound.restype = C.c_size_t
            lib.soc_png_bound.argtypes = [C.c_int, C.c_int]
            lib.soc_png_encode_u8.restype = C.c_long
            lib.soc_png_encode_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_int,
                                              C.c_void_p, C.c_size_t]
            if lib.soc_host_abi_version() != ABI_VERSION:
                raise RuntimeError("libsoc_host.so ABI version mismatch; rebuild it")
            _lib = lib
    return _lib


_scratch = threading.local()           # one output buffer per writer thread, grown on demand


def encode(img: np.ndarray, binarize: bool = False, palette: Optional[Sequence[int]] = None) -> bytes:
    """[H, W] uint8 / bool array (any row stride, contiguous along W) -> the bytes of a PNG file: mode 'L', or 'P' when
    `palette` (flat RGB list, as PIL's putpalette takes it) is given.  binarize: non-zero -> 255 (a bool mask -> 0 / 255)."""
    lib = load()
    a = np.asarray(img)
    if a.dtype == np.bool_:
        a = a.view(np.uint8)
    if a.dtype != np.uint8 or a.ndim != 2 or a.shape[0] < 1 or a.shape[1] < 1:
        raise ValueError(f"encode: expected a non-empty [H, W] uint8 / bool array, got {a.dtype} {a.shape}")
    if a.strides[1] != 1 or a.strides[0] < a.shape[1]:
        a = np.ascontiguousarray(a)
    h, w = a.shape
    cap = lib.soc_png_bound(h, w)
    buf = getattr(_scratch, "buf", None)
    if buf is None or buf.size < cap:
        buf = _scratch.buf = np.empty(cap, np.uint8)
    pal_ptr, n_colors = None, 0
    if palette is not None:
        pal = np.asarray(list(palette), dtype=np.uint8)
        n_colors = min(len(pal) // 3, 256)
        if n_colors < 1:
            raise ValueError("encode: empty palette")
        pal = np.ascontiguousarray(pal[:3 * n_colors])
        pal_ptr = pal.ctypes.data
    n = lib.soc_png_encode_u8(a.ctypes.data, h, w, a.strides[0], int(bool(binarize)), pal_ptr, n_colors, buf.ctypes.data, buf.size)
    if n <= 0:
        raise RuntimeError(f"soc_png_encode_u8 failed ({n})")
    return buf[:n].tobytes()


_fallback = None                       # None: not tried yet; True: libsoc_host.so could not be built / loaded -> Pillow


def use_pillow() -> bool:
    """True when the PNG files are to be written by Pillow: SOC_PNG=pillow, or libsoc_host.so cannot be built or loaded on this
    box (no gcc, a read-only package directory).  The drivers ask BEFORE the forwards run, so a missing library costs a warning
    and the slower writer, not the whole inference run at the first f.result() of the writer pool."""
    global _fallback
    if os.environ.get("SOC_PNG", "runs") == "pillow":
        return True
    if _fallback is None:
        try:
            load()
            _fallback = False
        except (OSError, subprocess.CalledProcessError, RuntimeError) as exc:      # FileNotFoundError (no gcc) is an OSError
            import warnings
            warnings.warn(f"libsoc_host.so unavailable ({exc}); PNG files are written by Pillow (slower, same pixels)")
            _fallback = True
    return _fallback


def pillow_save_kwargs() -> dict:
    """compress_level for the Pillow path: SOC_PNG_LEVEL, or nothing (= Pillow's default, the reference's files)."""
    level = os.environ.get("SOC_PNG_LEVEL")
    return {"compress_level": int(level)} if level not in (None, "") else {}


def save(path: str, img: np.ndarray, binarize: bool = False, palette: Optional[Sequence[int]] = None) -> None:
    data = encode(img, binarize, palette)
    with open(path, "wb") as f:
        f.write(data)
