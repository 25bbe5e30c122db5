"""The arithmetic mode of a forward ("split": f32 products as six bf16 MFMA products of an exact three-way operand split;
"f32": the f32-input MFMA path) and the routing rules that depend on it.  Split out of hot_ops.py, which
re-exports every name here."""
from __future__ import annotations

import os as _os

# "split" (default): the pixel-sized linear layers listed in split_wins() run on K20; "f32": every GEMM stays on the f32
# MFMA path (K13 / K12 / library), i.e. the round-2 arithmetic.  bench.py reports both.
#
# The mode is NOT process state: it belongs to a model (SOC.matmul_mode) and reaches the ops through a thread-local that the
# model's forward sets for its own duration (use_matmul_mode), and it reaches the C ABI as an argument of each launch.  Two
# models with different modes can therefore run from two threads of one process (tests/test_gpu_forward.py).  The environment
# variable only supplies the default a model is built with.
import contextlib as _contextlib
import threading as _threading

DEFAULT_MATMUL_MODE = _os.environ.get("SOC_MATMUL", "split")
_mode_tls = _threading.local()


def matmul_mode() -> str:
    """The arithmetic of the calling thread's current forward: "split" or "f32"."""
    return getattr(_mode_tls, "mode", None) or DEFAULT_MATMUL_MODE


@_contextlib.contextmanager
def use_matmul_mode(mode):
    """`with use_matmul_mode("f32"):` -- the ops called from THIS thread inside the block run in that arithmetic (None =
    leave it as it is).  Nests; other threads are unaffected."""
    if mode is None:
        yield
        return
    if mode not in ("split", "f32"):
        raise ValueError(f"matmul mode {mode!r}: expected 'split' or 'f32'")
    prev = getattr(_mode_tls, "mode", None)
    _mode_tls.mode = mode
    try:
        yield
    finally:
        _mode_tls.mode = prev


def split_enabled() -> bool:
    return matmul_mode() == "split"


def k1_split_enabled() -> bool:
    """K1 (full 8x7x7 windows) on the bf16 matrix cores with the exact three-way split; SOC_SPLIT_OFF=k1 or
    SOC_MATMUL=f32 keep the f32-input MFMA form."""
    return split_enabled() and "k1" not in _os.environ.get("SOC_SPLIT_OFF", "").split(",")


_SPLIT_OFF = set(filter(None, _os.environ.get("SOC_SPLIT_OFF", "").split(",")))   # debugging: sites forced back to f32


def split_wins(rows: int, N: int, K: int, fused_passes: int = 0, site: str = "plain") -> bool:
    """Does K20 beat the f32 path for a [rows, K] x [N, K]^T layer?  From tools/split_probe.py on MI355X (round 3): K20
    runs at 100-125 TFLOP/s f32-equivalent once the grid fills the chip and K is short, about what the tuned f32 library
    GEMM reaches, so it wins where it also removes separate passes (`fused_passes`: LayerNorm, GELU, residual / mul /
    positional adds), and loses on long-K layers with few row tiles (K >= 768 with < 30 000 rows)."""
    if not split_enabled() or K % 8 or N % 4 or rows < 1024 or site in _SPLIT_OFF:
        return False
    if K > 768:
        # long reductions (Video-Swin stage-3 fc2 + shortcut: K = 3072; the last merging reduction: K = 1536), round 6,
        # tools/experiments/longk_probe.py: at the 19 200 rows of a ten-clip launch group K20 runs them at 180 TFLOP/s where the
        # f32 library GEMM (+ its add pass) reaches 110; a tie at 7 680 rows (four clips), a loss at one clip's 1 920 (35 against 95)
        return rows >= 12288 and N >= 512
    if rows * N < 5_500_000:                    # too few tiles to fill 256 CUs (stage-2/3 proj, stage-3 qkv, coarse levels)
        return False
    if K > 512:                                 # K = 768: only the wide, GELU-fused fc1 of stage 3 (1920 x 768 -> 3072)
        return fused_passes >= 1 and N >= 1024
    if fused_passes == 0 and rows < 16384:      # bare GEMM on a short token map: a tie at best
        return False
    return True


def k13_split_enabled() -> bool:
    return split_enabled() and "k13" not in _SPLIT_OFF
