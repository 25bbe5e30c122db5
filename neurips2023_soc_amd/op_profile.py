"""Measurement hooks of the op wrappers (bench.py's rooflines, the route tests): per-kernel HIP-event timing and recording of a
forward's launches per kernel family.  Split out of hot_ops.py, which re-exports the public names.

Both are process-wide switches that a measuring caller turns on around ONE eager forward; neither may be on while a graph is
captured (graph_runner asserts `not op_profile.active()`)."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

# ---- per-kernel timing: HIP events on the stream the kernels run on
_prof: Optional[Dict[str, list]] = None


def active() -> bool:
    return _prof is not None


def profile_begin() -> None:
    global _prof
    _prof = {}


def profile_end() -> dict:
    """-> {kernel: {"launches", "ms", "work", "unit"}} summed over everything since profile_begin()."""
    global _prof
    rec, _prof = _prof or {}, None
    torch.cuda.synchronize()
    out = {}
    for name, items in rec.items():
        ms = sum(s.elapsed_time(e) for s, e, _ in items)
        out[name] = {"launches": len(items), "ms": ms, "work": float(sum(w for _, _, w in items)),
                     "unit": "flop" if name == "win_attn3d" else "byte"}
    return out


class timed:
    """`with timed("xs_linear", flops_or_bytes):` around a launch -- a pair of events while profiling is on, nothing otherwise."""

    def __init__(self, name: str, work: float):
        self.name, self.work = name, work

    def __enter__(self):
        if _prof is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *exc):
        if _prof is not None:
            self.e.record()
            _prof.setdefault(self.name, []).append((self.s, self.e, self.work))
        return False


# ---- recording of the launches of a forward, per kernel family ("k1", "k13", "k20", "k23", "k24"): bench.py replays exactly
# those launches -- same tensors, same geometry -- back to back between ONE pair of HIP events (per-launch event pairs add
# host / queue latency to ~50 us kernels); tests/test_routes.py checks which kernel every layer took
_calls: Dict[str, List] = {}


def record_calls(family: str, on: bool):
    """Start (True) / stop (False: returns the list) recording the arguments of every call of a kernel family; the recorded
    tensors are kept alive by the list."""
    if on:
        _calls[family] = []
        return None
    return _calls.pop(family, None)


def recording(family: str) -> Optional[List]:
    """The list the wrappers of `family` append their arguments to, or None when that family is not being recorded."""
    return _calls.get(family)
