"""Deterministic, name-keyed weights for golden fixtures (test infrastructure).

The fixtures under tests/golden/ do not store the ~1 M-parameter state_dict of every case (4 MB each);
both the generating script (which loads them into the *reference* model) and the tests (which load them
into the oracle / the HIP path) rebuild the identical tensors from (key, shape, seed) with this function.
numpy's PCG64 bit stream + Generator.random() are stable across numpy versions.
"""
import zlib

import numpy as np
import torch


def make_tensor(key, shape, seed=0):
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode())]))
    n = int(np.prod(shape))
    u = rng.random(n, dtype=np.float64) * 2.0 - 1.0
    if key.endswith("_pz_params.1"):
        v = 0.5 * u                                     # non-trivial trainable prior
    elif ("norm" in key or ".bn" in key or "downsample.1" in key) and key.endswith("weight"):
        v = 1.0 + 0.1 * u                               # LayerNorm / BatchNorm scales
    elif "embedding" in key:
        v = u
    elif key.endswith("bias"):
        v = 0.1 * u
    else:
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        v = u * (1.5 / np.sqrt(fan_in))
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def make_params(shapes, seed=0, requires_grad=False):
    out = {}
    for k, s in shapes.items():
        t = make_tensor(k, tuple(s), seed)
        if requires_grad:
            t.requires_grad_(True)
        out[k] = t
    return out


def summary_indices(n, k=96):
    return np.unique(np.linspace(0, n - 1, num=min(n, k)).astype(np.int64))


def summarize(t, k=96):
    """Compact pin of a tensor: [l2 norm, sum, abs-sum] + values at evenly spaced flat indices."""
    a = t.detach().double().reshape(-1).cpu().numpy()
    idx = summary_indices(a.size, k)
    return np.concatenate([[np.sqrt((a * a).sum()), a.sum(), np.abs(a).sum()], a[idx]])
