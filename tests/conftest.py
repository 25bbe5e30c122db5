"""Shared fixtures.  `-m "not gpu"` runs in the build container (no GPU); `-m gpu` on an MI355X."""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_npz(name)
        return cache[name]
    return get


@pytest.fixture(scope="session")
def ref_shapes():
    def get(tag="t"):
        with open(os.path.join(GOLDEN, f"state_shapes_{tag}.json")) as f:
            return json.load(f)
    return get


@pytest.fixture(scope="session")
def synthetic_sd(ref_shapes):
    """Synthetic Swin-T SOC checkpoint (same generator/seed the goldens were made with)."""
    from neurips2023_soc_amd import weights as W
    shapes = {k: v[0] for k, v in ref_shapes("t").items() if v[1].startswith("float")}
    return W.synthetic_state_dict(shapes, seed=2023)


