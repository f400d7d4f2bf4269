import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    return torch.cuda.is_available()


@pytest.fixture(scope="session")
def golden():
    """Captures of the reference itself (tests/golden/make_golden.py)."""
    with np.load(os.path.join(GOLDEN_DIR, "hotpath_b4.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def ref_state_names():
    with open(os.path.join(GOLDEN_DIR, "state_dict_names.json")) as f:
        return [(k, tuple(s)) for k, s in json.load(f)]


def to_torch_scene(scene, device=None):
    """numpy scene tree -> torch tensors (like data.from_numpy), optionally on a device."""
    def conv(x):
        if isinstance(x, dict):
            return {k: conv(v) for k, v in x.items()}
        if isinstance(x, list):
            return [conv(v) for v in x]
        if isinstance(x, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(x))
            return t.to(device) if device is not None else t
        return x
    return conv(scene)
