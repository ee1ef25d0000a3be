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
        """objectives.py:409-424: -log_prob, NaN -> 0.  Per-sample sums only on the hot path (recon_rowsum); the
        element-wise fp64 tensor is evaluation API and not built."""
        raise NotImplementedError("ReconLoss.lprob: use objectives.recon_rowsum (per-sample sums)")

    @staticmethod
    def optimal_sigma(output, target, bs):
        """objectives.py:503-509 (sigma-VAE): see recon_rowsum"""
        raise NotImplementedError("ReconLoss.optimal_sigma: use objectives.recon_rowsum (per-sample sums)")


PX_SCALE = 0.75     # the decoders return (recon, 0.75): px_z = Normal / Laplace(recon, 0.75)


# per-sample sums  sum_f ReconLoss.<ltype>(...)[b, f]  in one kernel
def recon_rowsum(ltype, out, target, laplace=False):
    """`out`: decoder output tensor; `target`: {"data", "masks"} (BaseObjective.recon_loss_fn, objectives.py:30-52:
    slice to the mask length -- and then the likelihood's scale := its loc --, reshape target like the output)."""
    masked = target["masks"] is not None
    if masked:
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
        return ops.optimal_sigma_rowsum(out, data.float().reshape(out.shape))
    if ltype == "bce":
        raw = getattr(out, "_bce_src", None)       # Dec_CNN: the producing layer's raw output (gradient = d logits)
        if raw is not None and raw.numel() == out.numel():
            return ops.bce_sigmoid_rowsum(raw, data.float().reshape(raw.shape))
        return ops.bce_rowsum(out, data.float().reshape(out.shape))
    if ltype == "category_ce":
        return ops.ce_over_time(out, data.float(), per_v=False)
    raise NotImplementedError(f"recon_loss {ltype} is not on the MI355X hot path (bce, category_ce, lprob, "
                              f"optimal_sigma are)")


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
    """models/objectives.py:305-387: `elbo` and `dreg`.  `iwae` crashes in the reference (objectives.py:353: `.cuda()`
    on a tuple), so there is nothing to be faithful to: selecting it raises."""

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
        """objectives.py:342-359 as its formula is INTENDED (SURVEY 8(a) flagged rows): the reference's own iwae crashes
        (`data["pz_params"].cuda()` on a list, and (K,) reconstruction sums reshaped to (K,B)), so there is no behaviour
        to pin -- PARITY UNPINNED, restated in oracle/mmvae_oracle.py: moe_iwae_objective.  Same ingredients as dreg
        (`data` of MOE._objective_dreg), per SAMPLE instead of summed over the batch:
            lw_r[k,b] = log p(z_r) - log-mean-exp_m log q_m(z_r) + lpx_own_r[k,b] + lpx_cross_r[k,b]      (beta = 1)
            loss = - sum_b log-mean-exp_{(r,k)} lw_r[k,b]        (fp64, as dreg)
        A handful of small torch ops on the (M,K,B) tensors: the K-sample latent kernel and the row sums do the work."""
        import math
        if data is None or float(self.beta) != 1.0:
            raise NotImplementedError("iwae: beta = 1 only (the latent kernel folds log p(z) - log q(z) into one term)")
        lat, rows, lam = data["lat"], data["rows"], data["lam"]
        M, K, B = lat.shape
        lpx = [[-float(lam[r]) * rows[2 * r + j].reshape(K, B).double() for j in (0, 1)] for r in range(M)]
        lw = torch.stack([lat[r].double() + lpx[r][0] + lpx[r][1] for r in range(M)])             # (M,K,B)
        loss = -(torch.logsumexp(lw.reshape(M * K, B), 0) - math.log(M * K)).sum()
        rec = torch.stack([torch.stack([lpx[r][0].sum(-1), lpx[r][1].sum(-1)]) for r in range(M)]).detach()   # (M,2,K)
        return {"loss": loss, "kld": torch.zeros((), dtype=torch.int64), "reconstruction_loss": rec}
