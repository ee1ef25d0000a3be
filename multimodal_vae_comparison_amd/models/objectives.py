"""Objective layer behind the reference's names (reference: models/objectives.py).

`MultimodalObjective(obj, beta)` / `ReconLoss.<name>(output_dist, target, bs) -> (bs, -1)` keep the reference's
plugin contract.  The mixers' `objective()` use the fused row-sum kernels directly (`recon_rowsum`), which
compute `ReconLoss.<name>(...).sum(-1)` without materialising the (bs, F) tensor."""
import torch

from .. import ops


class ReconLoss:
    """models/objectives.py:389-509.  Positive losses of shape (bs, -1)."""

    @staticmethod
    def bce(output, target, bs):
        """objectives.py:392-406 (the reference computes this on the CPU; here it never leaves the GPU)"""
        x_hat = output.loc
        return ops.bce_elem(x_hat, target.float().reshape(x_hat.shape).detach()).reshape(bs, -1)

    @staticmethod
    def category_ce(output, target, bs):
        """objectives.py:486-500: CrossEntropyLoss over dim 1 = TIME, probability targets; (B,T,V) -> (B,V)"""
        return ops.ce_over_time(output.loc, target.float().detach(), per_v=True).reshape(bs, -1)


    @staticmethod
    def lprob(output, target, bs):
        """objectives.py:409-424: -output.log_prob(target) as float64 (bs, -1), NaN -> 0 (those elements carry no
        gradient).  `output`: torch.distributions Normal / Laplace -- loc / scale as the mixers build them: a scalar
        0.75 scale, or scale := loc after BaseObjective.recon_loss_fn's masked-modality assignment (:43-45)."""
        loc = output.loc
        lap = isinstance(output, torch.distributions.Laplace)
        sc = output.scale
        if torch.is_tensor(sc) and sc.numel() > 1:
            if sc.data_ptr() == loc.data_ptr() or torch.equal(sc, loc):
                scale = None                                # the scale := loc quirk
            else:
                s0 = sc.reshape(-1)[0]
                assert bool((sc == s0).all()), "ReconLoss.lprob on the MI355X path: a constant scale, or scale := loc"
                scale = float(s0)
        else:
            scale = float(sc)
        return ops.lprob_elem(loc, target.float().reshape(loc.shape).detach(), scale, lap).reshape(bs, -1)

    @staticmethod
    def l1(output, target, bs):
        """objectives.py:427-442 (the reference computes it on the CPU and returns a CPU tensor; here it stays on the GPU)"""
        x = output.loc
        return ops.pointwise_elem(x, target.float().reshape(x.shape).detach(), ops.PW_L1).reshape(bs, -1)

    @staticmethod
    def mse(output, target, bs):
        """objectives.py:444-459"""
        x = output.loc
        return ops.pointwise_elem(x, target.float().reshape(x.shape).detach(), ops.PW_MSE).reshape(bs, -1)

    @staticmethod
    def optimal_sigma(output, target, bs):
        """objectives.py:503-509 (sigma-VAE): ONE log sigma = softclip(log sqrt(mean (t - x)^2), -6) per call; the squared
        term is detached, the only gradient path is log sigma"""
        x = output.loc
        return ops.optimal_sigma_elem(x, target.float().reshape(x.shape).detach()).reshape(bs, -1)

    @staticmethod
    def feature_loss(output, target, bs):
        """objectives.py:461-484: VGG19 feature loss with weights downloaded at call time (nn_modules.py:1103) -- needs a
        network and an un-vendored weight file: OUT OF SCOPE on this path, raises."""
        raise NotImplementedError("recon_loss feature_loss needs the downloaded VGG19 weights (reference "
                                  "nn_modules.py:1103); not on the MI355X path")


PX_SCALE = 0.75     # the decoders return (recon, 0.75): px_z = Normal / Laplace(recon, 0.75)


# per-sample sums  sum_f ReconLoss.<ltype>(...)[b, f]  in one kernel
def recon_rowsum(ltype, out, target, laplace=False):
    """`out`: decoder output tensor; `target`: {"data", "masks"} (BaseObjective.recon_loss_fn, objectives.py:30-52:
    slice to the mask length -- and then the likelihood's scale := its loc --, reshape target like the output)."""
    if getattr(out, "_bce_rows", False):      # Dec_CNN already produced the bce row sums with its last layer (ops.convT3_bce)
        return out
    masked = target["masks"] is not None
    if masked and out.shape[1] != target["masks"].shape[1]:      # (a decoder told `keep_steps` has sliced already)
        out = out[:, : target["masks"].shape[1]]
    data = target["data"]
    if ltype == "lprob":
        # a K-sample decoder output (K,B,...) is compared with the target repeated K times (reshape_for_loss,
        # objectives.py:118-120): the kernel indexes the target row as (output row) % B instead of materialising it
        tgt = data.float().reshape(out.shape) if out.numel() == data.numel() else data.float().reshape(data.shape[0], -1)
        src = getattr(out, "_lprob_src", None)      # Dec_MNIST / Dec_SVHN: sigmoid-epilogue output, logits' gradient
        if src is not None and not masked and src[0].numel() == out.numel():
            return ops.lprob_rowsum(src[0], tgt.reshape(tgt.shape[0], -1), PX_SCALE, laplace, perm_c=src[1],
                                    logit_grad=True)
        return ops.lprob_rowsum(out, tgt, None if masked else PX_SCALE, laplace)
    if ltype == "optimal_sigma":
        tgt = data.float()
        if out.numel() != tgt.numel():      # K-sample output (K*B rows): the target repeated K times, as reshape_for_loss
            tgt = tgt.reshape(data.shape[0], -1).repeat(out.numel() // tgt.numel(), 1)      # does (objectives.py:118-120)
        return ops.optimal_sigma_rowsum(out, tgt.reshape(out.shape))
    if ltype == "bce":
        raw = getattr(out, "_bce_src", None)       # Dec_CNN: the producing layer's raw output (gradient = d logits)
        if raw is not None and raw.numel() == out.numel():
            # K-sample output (K*B rows): the kernel pairs row r with target row r % B (no materialised repeat)
            return ops.bce_sigmoid_rowsum(raw, data.float().reshape(data.shape[0], -1) if raw.numel() != data.numel()
                                          else data.float().reshape(raw.shape))
        return ops.bce_rowsum(out, data.float().reshape(out.shape))
    if ltype == "category_ce":
        return ops.ce_over_time(out, data.float(), per_v=False)          # (K*B,T,V) logits against (B,T,V): row % B
    if ltype in ("l1", "mse"):
        return ops.pointwise_rowsum(out, data.float().reshape(data.shape[0], -1), ops.PW_L1 if ltype == "l1" else ops.PW_MSE)
    raise NotImplementedError(f"recon_loss {ltype} is not on the MI355X hot path (bce, category_ce, lprob, "
                              f"optimal_sigma, l1, mse are)")


class BaseObjective:
    """models/objectives.py:14-201"""

    def __init__(self):
        self.ltype = None
        self.beta = 1

    def set_ltype(self, ltype):
        self.ltype = ltype
        assert hasattr(ReconLoss, self.ltype), "Loss function {} is not implemented. Choose from: {}".format(
            self.ltype, [f for f in dir(ReconLoss) if callable(getattr(ReconLoss, f)) and not f.startswith("_")])

    def reshape_for_loss(self, output, target, K=1):
        """objectives.py:103-125"""
        target = torch.stack(target).float() if isinstance(target, list) else target
        target = target.repeat(K, *([1] * (len(target.shape) - 1))).reshape(*output.loc.shape)
        return output, target

    def recon_loss_fn(self, output, target, K=1):
        """objectives.py:30-52: returns -ReconLoss.<ltype>(...) of shape (bs, -1)"""
        if target["masks"] is not None:
            output.loc = output.loc[:, :target["masks"].shape[1]]
            output.scale = output.loc[:, :target["masks"].shape[1]]
        target = target["data"]
        output, target = self.reshape_for_loss(output, target, K)
        bs = target.shape[0]
        return -getattr(ReconLoss, self.ltype)(output, target, bs)

    def elbo(self, lpx_z, kld, beta=1):
        """objectives.py:54-67"""
        return -(lpx_z.sum(-1) - beta * kld.sum()).sum()


class MultimodalObjective(BaseObjective):
    """models/objectives.py:305-387: `elbo`, `dreg` and `iwae` (the reference's iwae needs its `.cuda()`-on-a-tuple at
    objectives.py:353 neutralised to run at all: see `iwae`)."""

    def __init__(self, obj: str, beta=1):
        super().__init__()
        assert hasattr(self, obj), "Objective {} is not implemented in multimodal scenario".format(obj)
        self.beta = beta
        self.obj_name = obj
        self.objective = getattr(self, obj)

    def calculate_loss(self, data):
        assert self.ltype is not None, "loss type is not set, please call set_ltype first"
        output = self.objective(data)
        assert isinstance(output, dict), "Objective function must return a dictionary"
        return output

    def elbo(self, data):
        loss = BaseObjective.elbo(self, data["lpx_z"], data["kld"], self.beta)
        return {"loss": loss, "reconstruction_loss": data["lpx_z"], "kld": data["kld"]}

    def dreg(self, data):
        """objectives.py:375-387 on the fused kernels (csrc/moe.hip).  data: {"lat": (M,K,B) latent part of the
        importance weights (ops.moe_ksample), "rows": [own_0, cross_0, own_1, ...] positive per-sample reconstruction
        sums (K*B), "lam": llik_scaling per modality}.  The z hook of :382-383 never fires in the reference (it is
        registered on a fresh torch.stack nothing consumes), so gradients flow through lw unweighted."""
        loss, rec = ops.dreg_loss(data["lat"], data["lam"], data["rows"])
        return {"loss": loss, "kld": torch.zeros((), dtype=torch.int64), "reconstruction_loss": rec}

    def iwae(self, data):
        """objectives.py:342-359, literally (pinned by tests/golden/moe_*_iwae_*.npz, generated from the reference under
        the tuple-`.cuda()` shim of tests/golden/ref_harness.py), on the fused kernels (csrc/moe.hip): same `data` as
        dreg, with `lat` = log p(z_r) - beta * log-mean-exp_m log q_m(z_r) (:353-356),
            lw_r[k,b] = lat_r[k,b] + lpx_own_r[k,b] + lpx_cross_r[k,b];   loss = - sum_b log-mean-exp_{(r,k)} lw_r[k,b]
        (`torch.cat(lws)` stacks the M (K,B) blocks along dim 0, log_mean_exp reduces dim 0), fp64 sums.
        The reference's reshape `lpx_z.reshape(*lpz.shape)` (:356) only succeeds when the decoders' leading axis has
        K*B rows: B = 1 on K-preserving towers, or K = 1.  Beyond those two edges -- K > 1 AND B > 1 -- the per-(k,b)
        reconstruction sums used here are a DEFINED EXTENSION (parity unpinned; oracle: moe_iwae_objective).
        Returns reconstruction_loss (M, 2, K*B) [own, cross] and kld = tensor(0) as the reference does."""
        loss, rec = ops.iwae_loss(data["lat"], data["lam"], data["rows"])
        return {"loss": loss, "kld": torch.zeros((), dtype=torch.int64), "reconstruction_loss": rec}
