import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_cases():
    """objective fixtures (make_golden.py)"""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and not f.startswith("fwd_"))


def forward_cases():
    """inference forward() with missing modalities (make_golden_forward.py)"""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("fwd_"))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    d = {k: z[k] for k in z.files}
    meta = json.loads(bytes(d.pop("meta")).decode())
    return meta, d


def load_full_grads(name):
    """every element of every parameter gradient of the reference for `name` (tests/golden/full_grads, one small case
    per mixer), or None"""
    path = os.path.join(GOLDEN_DIR, "full_grads", name + ".npz")
    if not os.path.exists(path):
        return None
    z = np.load(path)
    return {k: z[k] for k in z.files}


def full_grad_cases():
    d = os.path.join(GOLDEN_DIR, "full_grads")
    return sorted(f[:-4] for f in os.listdir(d) if f.endswith(".npz")) if os.path.isdir(d) else []


@pytest.fixture(scope="session")
def hip_lib():
    """The C-ABI library; skips (CPU box) are not allowed to hide a missing build on the GPU box."""
    import torch
    from multimodal_vae_comparison_amd import hipops
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return hipops.lib()
