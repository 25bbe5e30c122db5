"""ctypes access to oracle/_build/libsoc_oracle.so (plain-C MSDA restatement) -- test infrastructure."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libsoc_oracle.so")


def build() -> str:
    subprocess.run(["make", "-s", "-C", HERE], check=True)
    return LIB


def msda(value: np.ndarray, shapes: np.ndarray, lsi: np.ndarray, loc: np.ndarray, w: np.ndarray) -> np.ndarray:
    if not os.path.exists(LIB):
        build()
    lib = C.CDLL(LIB)
    dt = value.dtype
    fn = lib.soc_oracle_msda_f64 if dt == np.float64 else lib.soc_oracle_msda_f32
    value, loc, w = (np.ascontiguousarray(a, dtype=dt) for a in (value, loc, w))
    shapes, lsi = (np.ascontiguousarray(a, dtype=np.int64) for a in (shapes, lsi))
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = np.empty((N, Lq, M * D), dtype=dt)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    fn.restype = None
    fn(ptr(value), ptr(shapes), ptr(lsi), ptr(loc), ptr(w), ptr(out), *(C.c_int(v) for v in (N, S, M, D, L, Lq, P)))
    return out
