"""Import harness for the *reference* (gabinsane/multimodal-vae-comparison) on a CPU-only box.

ONLY used by tests/golden/make_golden.py, in the build container where /root/reference exists.
It never travels to the GPU box and nothing in the product, the `-m gpu` tests, smoke() or bench.py
imports it.  Recipe = SURVEY.md Appendix C:

  * stub modules for the reference's unavailable third-party imports (cv2, h5py, torchvision,
    pytorch_lightning, ...): none of them does arithmetic on the hot path;
  * numpy-2 removals used by the reference (np.product, np.float);
  * `.cuda()` / `.to("cuda")` -> identity so the reference's hard-coded device moves
    (models/mmvae_models.py:45,51,173,...; models/objectives.py:165,...,406) run on the CPU;
  * recording / replaying of the standard-normal draws behind `Normal.rsample`
    (torch/distributions/normal.py -> torch.distributions.utils._standard_normal) and of the draws behind
    `Laplace.rsample` (recorded as the standard-Laplace variate e = -sign(u) log1p(-|u|) of torch's uniform u, so that
    z = loc + scale * e exactly as for the Normal case);
  * `install_tuple_cuda_shim()` (iwae fixtures only, recorded in their `meta["shims"]`): MultimodalObjective.iwae calls
    `data["pz_params"].cuda()` on the TUPLE that `MOE.pz_params` returns (models/objectives.py:353) -- the same class of
    hard-coded device move as the tensor `.cuda()` calls above, only spelled on a tuple; the property hands back a tuple
    subclass whose `.cuda()` is the identity, nothing else changes.
"""
import importlib.machinery
import os
import sys
import types

REF_ROOT = os.environ.get("MMVAE_REFERENCE", "/root/reference/multimodal_compare")

_STUBS = [
    "cv2", "h5py", "imageio", "seaborn", "wget", "umap", "statsmodels", "statsmodels.api",
    "torchnet", "torchnet.dataset", "pytorch_fid", "pytorch_fid.inception",
    "torchvision", "torchvision.models", "torchvision.utils", "torchvision.transforms",
    "pytorch_lightning", "pytorch_lightning.loggers", "pytorch_lightning.callbacks",
    "pytorch_lightning.profiler", "adabelief_pytorch", "tensorboard",
]


class _Permissive(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        obj = type(name, (), {})
        setattr(self, name, obj)
        return obj


def install():
    """Make `import models` resolve to the reference package. Idempotent."""
    import numpy as np
    import torch
    import torch.nn as nn

    if getattr(install, "_done", False):
        return
    sys.dont_write_bytecode = True
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    for name in _STUBS:
        if name in sys.modules:
            continue
        try:
            __import__(name)
            continue
        except Exception:
            pass
        m = _Permissive(name)
        m.__path__ = []
        m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
        sys.modules[name] = m
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(sys.modules[parent], child, m)
    pl = sys.modules["pytorch_lightning"]
    pl.LightningModule = nn.Module
    pl.LightningDataModule = object
    sys.modules["pytorch_fid.inception"].InceptionV3 = type(
        "InceptionV3", (), {"BLOCK_INDEX_BY_DIM": {2048: 3}})
    if not hasattr(np, "product"):
        np.product = np.prod
    if not hasattr(np, "float"):
        np.float = float

    ident = lambda self, *a, **k: self
    torch.Tensor.cuda = ident
    nn.Module.cuda = ident
    _orig_to = torch.Tensor.to

    def _to(self, *args, **kwargs):
        args = tuple("cpu" if (isinstance(a, str) and a.startswith("cuda")) else a for a in args)
        if isinstance(kwargs.get("device"), str) and kwargs["device"].startswith("cuda"):
            kwargs["device"] = "cpu"
        return _orig_to(self, *args, **kwargs)

    torch.Tensor.to = _to
    _orig_mto = nn.Module.to

    def _mto(self, *args, **kwargs):
        args = tuple("cpu" if (isinstance(a, str) and a.startswith("cuda")) else a for a in args)
        return _orig_mto(self, *args, **kwargs)

    nn.Module.to = _mto
    install._done = True


class CudaTuple(tuple):
    """a tuple whose `.cuda()` is the identity (objectives.py:353 calls it on MOE.pz_params)"""

    def cuda(self, *a, **k):
        return self


def install_tuple_cuda_shim():
    """MOE.pz_params -> CudaTuple(same two tensors).  Idempotent; returns the shim's name for fixture metadata."""
    import models.mmvae_models as mm
    if not getattr(install_tuple_cuda_shim, "_done", False):
        orig = mm.MOE.pz_params
        mm.MOE.pz_params = property(lambda self: CudaTuple(orig.fget(self)))
        install_tuple_cuda_shim._done = True
    return "MOE.pz_params returns a tuple subclass with an identity .cuda() (objectives.py:353)"


class EpsTape:
    """Record (or replay) every standard-normal draw made through torch.distributions."""

    def __init__(self, replay=None):
        self.draws = []
        self.replay = list(replay) if replay is not None else None

    def __enter__(self):
        import torch
        import torch.distributions.utils as du
        import torch.distributions.normal as dn
        self._du, self._dn = du, dn
        self._orig = du._standard_normal

        def rec(shape, dtype, device):
            if self.replay is not None:
                e = self.replay.pop(0)
                assert tuple(e.shape) == tuple(shape), (e.shape, shape)
                e = e.to(dtype)
            else:
                e = self._orig(shape, dtype, device)
            self.draws.append(e.detach().clone())
            return e

        du._standard_normal = rec
        dn._standard_normal = rec

        # Laplace.rsample draws u ~ U(eps - 1, 1) with Tensor.uniform_ and returns loc - scale * sign(u) * log1p(-|u|)
        # (torch/distributions/laplace.py).  Record: run the original, then redraw the same u from the saved
        # generator state; replay: loc + scale * e (bit-identical, sign(u) is +-1).
        import torch.distributions.laplace as dl
        self._dl = dl
        self._orig_lap = dl.Laplace.rsample
        tape = self

        def lap_rsample(d, sample_shape=torch.Size()):
            shape = d._extended_shape(sample_shape)
            if tape.replay is not None:
                e = tape.replay.pop(0)
                assert tuple(e.shape) == tuple(shape), (e.shape, shape)
                tape.draws.append(e.detach().clone())
                return d.loc + d.scale * e.to(d.loc.dtype)
            st = torch.get_rng_state()
            out = tape._orig_lap(d, sample_shape)
            end = torch.get_rng_state()
            torch.set_rng_state(st)
            u = d.loc.new(shape).uniform_(torch.finfo(d.loc.dtype).eps - 1, 1)
            torch.set_rng_state(end)
            e = -(u.sign() * torch.log1p(-u.abs()))
            assert torch.equal(d.loc + d.scale * e, out)
            tape.draws.append(e.detach().clone())
            return out

        dl.Laplace.rsample = lap_rsample
        return self

    def __exit__(self, *exc):
        self._du._standard_normal = self._orig
        self._dn._standard_normal = self._orig
        self._dl.Laplace.rsample = self._orig_lap
        return False
