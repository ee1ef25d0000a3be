"""Per-modality VAE wrapper + tower factory by name (reference: models/vae.py)."""
import torch
import torch.distributions as dist
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from . import decoders, encoders
from .decoders import VaeDecoder
from .encoders import VaeEncoder


# towers of the reference's registry that are outside SURVEY section 8's hot-path scope: selecting one fails loudly,
# by name, instead of with the reference's generic "Did not find encoder" assert
OUT_OF_SCOPE_TOWERS = ("FNN", "PolyMNIST", "VideoGPT", "VIT", "TransformerIMG", "Audio", "AudioConv",
                       "CNN_CUB", "Fashion_CNN", "Sprites")


class DencoderFactory(object):
    @classmethod
    def get_nework_classes(cls, enc_name, dec_name, n_latents, private_latents, data_dim: tuple, enc_mu_logvar: bool):
        """vae.py:15-30: `Enc_<name>` / `Dec_<name>` looked up by string"""
        for name in (enc_name, dec_name):
            if name in OUT_OF_SCOPE_TOWERS:
                raise NotImplementedError(f"tower '{name}' is outside the MI355X hot-path scope (SURVEY.md section 8: "
                                          f"CNN2/CNN, TxtTransformer, TxtRNN, Transformer, MNIST, SVHN are built)")
        assert hasattr(encoders, "Enc_{}".format(enc_name)), "Did not find encoder {}".format(enc_name)
        enc_obj = getattr(encoders, "Enc_{}".format(enc_name))(n_latents, data_dim, private_latents, enc_mu_logvar)
        assert hasattr(decoders, "Dec_{}".format(dec_name)), "Did not find decoder {}".format(dec_name)
        dec_obj = getattr(decoders, "Dec_{}".format(dec_name))(n_latents, data_dim, private_latents)
        return enc_obj, dec_obj


class BaseVae(nn.Module):
    """vae.py:33-59"""

    def __init__(self, enc, dec, prior_dist=dist.Normal, likelihood_dist=dist.Normal, post_dist=dist.Normal):
        super().__init__()
        assert isinstance(enc, VaeEncoder) and isinstance(dec, VaeDecoder)
        self.enc, self.dec = enc, dec
        assert enc.latent_dim == dec.latent_dim
        self.n_latents = enc.latent_dim
        self.pz, self.px_z, self.qz_x = prior_dist, likelihood_dist, post_dist

    def encode(self, inp):
        return self.enc(inp)

    def decode(self, inp):
        return self.dec(inp)


class VAE(BaseVae):
    """vae.py:121-204"""

    def __init__(self, enc, dec, feature_dim, n_latents, ltype, private_latents=None, llik_scaling=1,
                 prior_dist="normal", likelihood_dist="normal", post_dist="normal", obj_fn=None, beta=1,
                 id_name="mod_1", enc_mu_logvar=True):
        dist_map = {"normal": dist.Normal, "categorical": dist.Categorical, "laplace": dist.Laplace,
                    "gumbel": dist.Gumbel, "gaussian": dist.Normal}
        self.prior_str = prior_dist.lower()
        enc_net, dec_net = DencoderFactory().get_nework_classes(enc, dec, n_latents, private_latents, feature_dim,
                                                                enc_mu_logvar)
        super().__init__(enc_net, dec_net, dist_map[prior_dist.lower()], dist_map[likelihood_dist.lower()],
                         dist_map[post_dist.lower()])
        self.llik_scaling = llik_scaling
        self.data_dim = feature_dim
        self.private_latents = private_latents
        self.n_latents = n_latents
        self.post_dist, self.likelihood_dist, self.prior_dist = self.qz_x, self.px_z, self.pz
        self.total_latents = n_latents + private_latents if private_latents is not None else n_latents
        self._pz_params = nn.ParameterList([
            nn.Parameter(torch.zeros(1, self.total_latents), requires_grad=False),
            nn.Parameter(torch.ones(1, self.total_latents), requires_grad=False)])
        self._pz_params_private = None
        if private_latents is not None:
            self._pz_params_private = nn.ParameterList([
                nn.Parameter(torch.zeros(1, private_latents), requires_grad=False),
                nn.Parameter(torch.ones(1, private_latents), requires_grad=False)])
        self.modelName = id_name
        self.ltype = ltype
        self.obj_name = obj_fn
        self.beta = beta
        # unimodal use (models/trainer.py:112-113): noise generator state and the N(0, 1) prior of UnimodalObjective.elbo
        self.eps_override = None
        self.register_buffer("_rng_state", torch.tensor([torch.initial_seed() & 0x7FFFFFFF, 0, 0], dtype=torch.int32),
                             persistent=False)
        self.register_buffer("_theta0", torch.zeros(1, self.total_latents), persistent=False)
        for part in ("enc", "dec"):
            st = getattr(getattr(self, part), "drop_state", None)
            if st is not None and not st.prefix:
                st.prefix = part

    @property
    def pz_params(self):
        return self._pz_params[0], F.softmax(self._pz_params[1], dim=1) * self._pz_params[1].size(-1)

    @property
    def pz_params_private(self):
        return self._pz_params_private[0], \
            F.softmax(self._pz_params_private[1], dim=1) * self._pz_params_private[1].size(-1)

    # ---- the unimodal case: `self.model = vaes["mod_1"]` (models/trainer.py:112-113) ------------------------------
    def objective(self, data):
        """VAE.forward + objective with UnimodalObjective.elbo (models/vae.py:92-119,268-282, models/objectives.py:233-247):
        q = Normal(mu, lv as sigma), one rsample, recon = dec(z), KL(q || N(0, 1)) against the raw `_pz_params`;
        loss = -(lpx_z.sum(-1) - beta * kld.sum()).sum() = sum_b recon_b + B * beta * sum_b kld_b (kld.sum() is the batch
        total and is subtracted from every row).  Returns per-sample sums for "kld" (B,) and "reconstruction_loss"
        (B,) -- the reference returns the (B, D) / (B, F) element tensors; the logged `.sum()`s are the same."""
        from .mmvae_base import packed_head
        from .objectives import recon_rowsum
        if self.obj_name != "elbo":
            raise NotImplementedError(f"unimodal objective '{self.obj_name}': elbo is on the MI355X path")
        if self.prior_str not in ("normal", "gaussian"):
            raise NotImplementedError(f"unimodal VAE with prior '{self.prior_str}'")
        x = data["mod_1"]
        ops.GradReducer.begin_step(self._rng_state.device)
        for part in (self.enc, self.dec):
            st = getattr(part, "drop_state", None)
            if st is not None:
                st.reset_calls()
        packed = packed_head(*self.enc(x))
        B, D = packed.shape[0], self.total_latents
        if self.eps_override is not None:
            eps = self.eps_override.pop(0).reshape(B, D).to(device=packed.device, dtype=torch.float32).contiguous()
        else:
            eps = ops.randn((B, D), self._rng_state)
        _, kl, z = ops.poe_reparam_kl(self._theta0, [packed], [eps], 2, 0b10)
        out, _ = self.dec({"latents": z[0].unsqueeze(0), "masks": x["masks"]})
        rec = recon_rowsum(self.ltype, out, x)
        loss = ops.lincomb_rows([rec, kl[1]], [[1.0, float(B) * float(self.beta)]])
        return {"loss": loss[0], "kld": kl[1], "reconstruction_loss": -rec}
