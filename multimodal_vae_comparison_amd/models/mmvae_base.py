"""TorchMMVAE: base class of the multimodal mixers (reference: models/mmvae_base.py)."""
import abc

import numpy as np
import torch
import torch.distributions as dist
import torch.nn as nn

from .. import ops
from .nn_modules import DropoutState
from .objectives import MultimodalObjective
from .output_storage import VAEOutput

# the decoders' first dropout-counter advance rides on the first encoder's launch (DropoutState.link;
# MMVAE_DROPOUT_LINK=0: one one-thread launch at the head of every tower's chain)
LINK_DROPOUT_ADVANCE = True


def normal(loc, scale):
    """torch.distributions.Normal without the argument validation pass (it forces a device->host sync)."""
    return dist.Normal(loc, scale, validate_args=False)


def packed_head(mu, lv):
    """(B, 2D) = [mu | lv].  The towers of this package return mu / lv as the two halves of one tensor
    (VaeComponent.process_output); recover it without a copy, else concatenate."""
    base = mu._base
    if base is not None and base is lv._base and base.dim() == 2 and base.shape[1] == 2 * mu.shape[1] \
            and mu.data_ptr() == base.data_ptr() and lv.data_ptr() == base.data_ptr() + 4 * mu.shape[1] \
            and base.is_contiguous():
        return base
    return torch.cat([mu, lv], dim=-1).contiguous()


class TorchMMVAE(nn.Module):
    """mmvae_base.py:12-240.  Plugin contract: subclass, implement modality_mixing(mods) and
    objective(mods) -> {"loss": scalar, ["reconstruction_loss": ..., "kld": ...]}."""

    def __init__(self, vaes, n_latents: int, obj: str, beta=1, K=1):
        super().__init__()
        self.vaes = nn.ModuleDict(vaes)
        self.modelName = "TorchMMVAE"
        self.qz_x = self.px_z = self.pz = dist.Normal
        self.n_latents = n_latents
        self.K = K
        self.obj_fn = MultimodalObjective(obj, beta)
        self._pz_params = nn.ParameterList([
            nn.Parameter(torch.zeros(1, self.n_latents), requires_grad=False),  # mu
            nn.Parameter(torch.zeros(1, self.n_latents), requires_grad=True)])  # logvar (trainable, mmvae_base.py:35-38)
        self.set_likelihood_scales()
        for name, vae in self.vaes.items():      # names of the dropout sites (tests map them onto the oracle's)
            for part in ("enc", "dec"):
                st = getattr(getattr(vae, part), "drop_state", None)
                if st is not None:
                    st.prefix = f"vaes.{name}.{part}"
        self.eps_override = None    # list of (K,B,D) noise tensors consumed in draw order (parity tests)
        # counter-based device noise generator {seed, call counter, ticket} (ops.randn)
        self.register_buffer("_rng_state", torch.tensor([torch.initial_seed() & 0x7FFFFFFF, 0, 0], dtype=torch.int32),
                             persistent=False)

    def set_likelihood_scales(self):
        """mmvae_base.py:41-47"""
        min_dim = min([np.prod(vae.enc.data_dim) for vae in self.vaes.values()])
        for vae in self.vaes.values():
            if vae.llik_scaling == "auto":
                vae.llik_scaling = min_dim / np.prod(vae.enc.data_dim)
            else:
                vae.llik_scaling = float(vae.llik_scaling)

    def _require_normal_priors(self):
        """`prior:` names the per-VAE prior / posterior / likelihood classes (models/trainer.py:104).  MoPoE and DMVAE
        build their posteriors with a hard-coded dist.Normal (mmvae_models.py:363-365,480-485), so there the config's
        family only reaches the LIKELIHOOD `vae.px_z` (:369,495-501): `normal` and `laplace` are on this path (laplace =
        a Laplace log-prob under recon_loss lprob; every other loss only reads the likelihood's loc), pinned by the
        reference fixtures *_lprob_laplace.  Anything else must not silently compute Normal terms."""
        for name, vae in self.vaes.items():
            if vae.prior_str not in ("normal", "gaussian", "laplace"):
                raise NotImplementedError(f"{self.modelName}: prior '{vae.prior_str}' ({name}) is not on the MI355X path "
                                          f"(normal, laplace are)")

    def _lap(self, vae):
        """is this VAE's likelihood `px_z` a Laplace (`prior: laplace`)?"""
        return vae.prior_str == "laplace"

    def _px(self, vae, loc, scale):
        """`vae.px_z(loc, scale)` (mmvae_models.py:369,495): the likelihood object of the inference forward()"""
        return dist.Laplace(loc, scale, validate_args=False) if self._lap(vae) else normal(loc, scale)

    @property
    def latent_factorization(self):
        return any(v.private_latents is not None for v in self.vaes.values())

    def _begin_step(self):
        """start of an objective() call: per-step dropout call counters back to 0"""
        ops.GradReducer.begin_step(self._rng_state.device)
        groups = ([], [])
        for vae in self.vaes.values():
            for k, part in enumerate((vae.enc, vae.dec)):
                st = getattr(part, "drop_state", None)
                if st is not None:
                    st.reset_calls()
                    if part.training:
                        groups[k].append(st)
        if LINK_DROPOUT_ADVANCE:
            DropoutState.link(*groups)

    # ---- noise ----------------------------------------------------------------------------------
    def _draw(self, B, D, device):
        """one standard-normal draw of shape (B, D): replayed from `eps_override` or from the device generator"""
        if self.eps_override is not None:
            e = self.eps_override.pop(0)
            return e.reshape(B, D).to(device=device, dtype=torch.float32).contiguous()
        return ops.randn((B, D), self._rng_state)

    def _draw_many(self, n, B, D, device):
        """n draws of shape (B, D) in one launch (`eps_override`: n consecutive recorded draws)"""
        if self.eps_override is not None:
            return [self._draw(B, D, device) for _ in range(n)]
        return list(ops.randn((n, B, D), self._rng_state).unbind(0))

    # ---- tower-level concurrency -------------------------------------------------------------------
    def _tower_streams(self, device, main=0):
        """stream per modality: modality `main` stays on the current stream (None), the others get side streams
        (ops.StreamPlan) so that independent towers overlap -- each of them alone cannot fill the chip at batch 128"""
        names = list(self.vaes.keys())
        key = device.index if device.index is not None else (torch.cuda.current_device() if device.type == "cuda" else 0)
        ops.StreamPlan.pair.pop(key, None)
        if not (ops.StreamPlan.enabled and device.type == "cuda") or len(names) < 2:
            return [None] * len(names)
        # (the two streams of this step, for ops that park weight-gradient launches on the stream they are NOT on)
        ops.StreamPlan.pair[key] = (torch.cuda.current_stream(device), ops.StreamPlan.get("tower1", device))
        # ONE side stream however many towers there are: a captured step with three parallel branches (image, text,
        # actions on their own streams) crashes ROCm 7.2's hipGraphInstantiate (segmentation fault inside capture_end;
        # two branches and the eager three-stream path are fine) -- the towers beyond the first two share a stream
        max_side = 1
        if len(names) >= 3 and max_side == 1:
            # two streams for three or more towers: the tower with the longest launch chain (number of sub-modules as the
            # proxy: the 8 + 4-layer action Transformers, a ResNet-50) gets the side stream to itself, the others share
            # the capture stream (BASELINE configs[4]: 4.48 vs 4.68 ms/step with text + actions sharing the side stream)
            heavy = max(range(len(names)), key=lambda i: sum(1 for _ in self.vaes[names[i]].modules()))
            s = ops.StreamPlan.get("tower1", device)
            ops.GradReducer.note_stream(device, s)
            return [s if i == heavy else None for i in range(len(names))]
        out, side = [], 0
        for i in range(len(names)):
            if i == main % len(names):
                out.append(None)
                continue
            side = min(side + 1, max_side)
            s = ops.StreamPlan.get(f"tower{side}", device)
            ops.GradReducer.note_stream(device, s)
            out.append(s)
        return out

    def _fork(self, streams, device, batch=None):
        """the side streams wait for what is queued on the current one.  `batch`: the step's input dict -- a tower on a
        side stream reads its modality's data / masks (encoder input, loss target) from tensors that were allocated on
        the caller's stream: they are registered with the side stream, or the caching allocator hands their memory out
        again as soon as the caller drops the batch -- while the side stream's last kernels (the text encoder's embedding
        backward) have yet to read it.  (Found in round 4: `objective(to_device(batch))` with a temporary dict, two
        streams, a step following smaller ones in the same process: wrong embedding gradient.)"""
        cur = torch.cuda.current_stream(device)
        for s in streams:
            if s is not None:
                s.wait_stream(cur)
        if batch is not None:      # (every side stream: MoPoE's decoders rotate streams after the fusion)
            side = [s for s in streams if s is not None]
            for entry in batch.values():
                for t in entry.values():
                    if torch.is_tensor(t) and t.is_cuda:
                        for s in side:
                            t.record_stream(s)

    @staticmethod
    def _join(streams, device):
        cur = torch.cuda.current_stream(device)
        for s in streams:
            if s is not None:
                cur.wait_stream(s)

    # ---- plumbing shared by the mixers ----------------------------------------------------------
    def make_output_dict(self, encoder_dist=None, decoder_dist=None, latent_samples=None, joint_dist=None,
                         enc_dist_private=None, dec_dist_private=None, joint_decoder_dist=None,
                         cross_decoder_dist=None):
        out = VAEOutput()
        for v in ["encoder_dist", "decoder_dist", "latent_samples", "joint_dist", "enc_dist_private",
                  "dec_dist_private", "joint_decoder_dist", "cross_decoder_dist"]:
            out.set_with_dict(locals()[v], v)
        return out

    def encode(self, inputs):
        """mmvae_base.py:139-159"""
        qz_xs = {}
        for modality, vae in self.vaes.items():
            if modality in inputs and inputs[modality]["data"] is not None:
                qz_x = vae.enc(inputs[modality])
                if not self.latent_factorization:
                    qz_xs[modality] = {"shared": qz_x, "private": None}
                else:
                    n = vae.n_latents
                    qz_xs[modality] = {"shared": [qz_x[0][:, :n], qz_x[1][:, :n]],
                                       "private": [qz_x[0][:, n:], qz_x[1][:, n:]]}
            elif modality in inputs and inputs[modality]["data"] is None:
                qz_xs[modality] = {"shared": None, "private": None}
        return qz_xs

    def decode(self, samples):
        """mmvae_base.py:185-201"""
        pz_xs = {}
        for modality, vae in self.vaes.items():
            if modality in samples and samples[modality]["latents"] is not None:
                pz_xs[modality] = vae.dec(samples[modality])
            elif modality in samples and samples[modality]["latents"] is None:
                pz_xs[modality] = None
        return pz_xs

    @abc.abstractmethod
    def modality_mixing(self, mods):
        pass

    @abc.abstractmethod
    def objective(self, mods):
        pass

    def product_of_experts(self, mu, logvar, with_prior=False):
        """mmvae_base.py:203-222 on the fused kernel: mu/logvar are lists of (B,D) tensors; returns
        (mu, VARIANCE) of the product (the reference returns the variance as `pd_logvar`)."""
        packed = [packed_head(m, l) for m, l in zip(mu, logvar)]
        joint, _, _ = ops.poe_reparam_kl(self._pz_params[1], packed, [], with_prior, 0, self._pz_params[1].grad)
        return joint[0], joint[1]

    def get_missing_modalities(self, mods):
        keys, keys_with_val = [], []
        for modality, val in mods.items():
            (keys if val["data"] is None else keys_with_val).append(modality)
        return keys, keys_with_val
