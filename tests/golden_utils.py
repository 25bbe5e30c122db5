"""Helpers shared by the golden generator and the tests (sub-sampling of large tensors)."""
import numpy as np
import torch


def sub(t: torch.Tensor, n: int = 4096) -> np.ndarray:
    f = t.detach().float().flatten()
    step = max(1, f.numel() // n)
    return f[::step][:n].cpu().numpy().copy()


def t(x) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(x))
