"""Library-GEMM algorithm selection for the hot path's shapes.

~0.65 of the ~0.77 TFLOP per clip are plain fp32 GEMMs (Swin qkv/proj/MLP, deformable-encoder
linears/FFN) that run in hipBLASLt/rocBLAS through torch.  Their default heuristics are up to
35 % off the best available kernel on the tall-skinny shapes of this model (M = 115 200 tokens,
K = 96), so the package ships the winners measured on an MI355X (`tunableop_gfx950.csv`,
produced by tools/tune_gemms.py with PyTorch TunableOp) and loads them with tuning DISABLED:
no start-up cost, no run-to-run variation.  Shapes not in the table use the library default.
The file is validated by TunableOp against the torch / hipBLASLt / rocBLAS versions and the GPU
architecture it was recorded with; on a mismatch it is ignored.
Set SOC_DISABLE_TUNED_GEMMS=1 to run with library defaults.
"""
from __future__ import annotations

import os

import torch

TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")
_done = False


def enable_tuned_gemms() -> bool:
    """Idempotent.  Returns True when the shipped table was loaded."""
    global _done
    if _done:
        return True
    if os.environ.get("SOC_DISABLE_TUNED_GEMMS") == "1" or not torch.cuda.is_available():
        return False
    if not os.path.exists(TABLE):
        return False
    import torch.cuda.tunable as tunable
    if os.environ.get("PYTORCH_TUNABLEOP_TUNING") == "1":
        return False  # a tuning session (tools/tune_gemms.py) is in charge
    tunable.enable(True)
    tunable.tuning_enable(False)
    try:
        ok = bool(tunable.read_file(TABLE))
    except Exception:
        ok = False
    # whatever TunableOp may ever want to write goes to a per-process scratch file, never to the
    # shipped table (N ranks share the package directory)
    import tempfile
    tunable.set_filename(os.path.join(tempfile.gettempdir(), f"soc_tunableop_{os.getpid()}.csv"))
    _done = ok
    return ok
