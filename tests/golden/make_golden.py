"""Generate golden input/output vectors by running the REFERENCE itself (build container only).

    PYTHONHASHSEED=0 python tests/golden/make_golden.py

Imports /root/reference/multimodal_compare under tests/golden/ref_harness.py, loads the deterministic
weights of oracle/golden_weights.py into the reference's own model classes, runs `objective()` +
`backward()` + one Adam(amsgrad) step with the noise draws recorded, and writes tests/golden/<case>.npz.
The fixtures are data only (inputs, noise, expected outputs / gradient summaries); no reference source
is copied.  The oracle (oracle/mmvae_oracle.py) and the HIP path are both tested against these files.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.nn as nn

import ref_harness

ref_harness.install()
import models  # noqa: E402  (the reference package)
from models.vae import VAE  # noqa: E402

from oracle import golden_weights as gw  # noqa: E402
from oracle import mmvae_oracle as orc  # noqa: E402

MODS = [
    {"enc": "CNN2", "dec": "CNN", "data_dim": [64, 64, 3], "ltype": "bce", "llik_scaling": 1.0},
    {"enc": "TxtTransformer", "dec": "TxtTransformer", "data_dim": [45, 27, 1], "ltype": "category_ce",
     "llik_scaling": 1.0},
]

CASES = [
    # name, mixing, B, T, D, lengths (None = ragged random), mode, beta
    ("mopoe_b6_t5_d8", "mopoe", 6, 5, 8, [5, 3, 4, 2, 5, 1], "eval", 1.0),
    ("mopoe_b5_t5_d8_BeqT", "mopoe", 5, 5, 8, [5, 2, 4, 3, 1], "eval", 1.0),
    ("mopoe_b4_t6_d16_full", "mopoe", 4, 6, 16, [6, 6, 6, 6], "eval", 2.5),
    ("mopoe_b7_t9_d32_trainp0", "mopoe", 7, 9, 32, [9, 4, 7, 1, 3, 9, 5], "train_p0", 1.0),
    ("poe_b4_t5_d8", "poe", 4, 5, 8, [5, 2, 3, 4], "eval", 1.0),
    ("poe_b4_t4_d16_BeqT", "poe", 4, 4, 16, [4, 1, 3, 2], "eval", 0.5),
    ("moe_b5_t6_d8", "moe", 5, 6, 8, [6, 2, 4, 3, 1], "eval", 1.0),
    ("moe_b6_t6_d16_BeqT", "moe", 6, 6, 16, [6, 5, 1, 3, 2, 4], "eval", 2.0),
    ("dmvae_b5_t6_d8p4", "dmvae", 5, 6, 8, [6, 2, 4, 3, 1], "eval", 1.0),
    ("dmvae_b6_t6_d16p6_BeqT", "dmvae", 6, 6, 16, [6, 5, 1, 3, 2, 4], "eval", 0.5),
]
PRIVATE = {"dmvae_b5_t6_d8p4": 4, "dmvae_b6_t6_d16p6_BeqT": 6}

# cases with their own modality lists: the action Transformer towers, lprob (a21), optimal_sigma (a23)
ACTIONS = {"enc": "Transformer", "dec": "Transformer", "data_dim": [6, 4, 1], "llik_scaling": 1.0}
CASE_MODS = {
    "mopoe3_b5_t6_d8_actions_optsigma": [MODS[0], MODS[1], dict(ACTIONS, ltype="optimal_sigma")],
    "poe3_b4_t5_d8_actions_lprob": [dict(MODS[0], ltype="lprob"), MODS[1], dict(ACTIONS, ltype="lprob")],
}
# MNIST MLP towers + SVHN conv towers with lprob (the reference's mnist_svhn feature dims)
MS = [{"enc": "MNIST", "dec": "MNIST", "data_dim": [28, 28, 1], "ltype": "lprob", "llik_scaling": 1.0},
      {"enc": "SVHN", "dec": "SVHN", "data_dim": [32, 32, 3], "ltype": "lprob", "llik_scaling": 1.0}]
CASE_MODS["dmvae_ms_b5_d8p4_lprob"] = MS
CASE_MODS["mopoe_ms_b6_d8_lprob"] = MS
PRIVATE["dmvae_ms_b5_d8p4_lprob"] = 4
CASES += [
    ("mopoe3_b5_t6_d8_actions_optsigma", "mopoe", 5, 6, 8, [6, 2, 4, 3, 1], "eval", 1.0),
    ("poe3_b4_t5_d8_actions_lprob", "poe", 4, 5, 8, [5, 2, 3, 4], "eval", 1.0),
    ("dmvae_ms_b5_d8p4_lprob", "dmvae", 5, 0, 8, None, "eval", 1.0),
    ("mopoe_ms_b6_d8_lprob", "mopoe", 6, 0, 8, None, "eval", 2.0),
]


# the shipped configs/config_mnistsvhn.yml path: MoE with obj dreg, K > 1, prior laplace, llik_scaling auto
# (`prior` also selects the posterior and the cross-reconstruction likelihood, models/trainer.py:104)
MS_AUTO = [dict(m, llik_scaling="auto") for m in MS]
CASE_OPTS = {
    "moe_ms_b5_d8_dreg_k3_laplace": {"obj": "dreg", "K": 3, "prior": "laplace"},
    "moe_ms_b4_d20_dreg_k30_laplace": {"obj": "dreg", "K": 30, "prior": "laplace"},
    "moe_ms_b5_d8_dreg_k2_normal": {"obj": "dreg", "K": 2, "prior": "normal"},
    "moe_ms_b6_d8_elbo_laplace": {"obj": "elbo", "K": 1, "prior": "laplace"},
}
for _n in CASE_OPTS:
    CASE_MODS[_n] = MS_AUTO
CASES += [
    ("moe_ms_b5_d8_dreg_k3_laplace", "moe", 5, 0, 8, None, "eval", 1.0),
    ("moe_ms_b4_d20_dreg_k30_laplace", "moe", 4, 0, 20, None, "eval", 1.0),
    ("moe_ms_b5_d8_dreg_k2_normal", "moe", 5, 0, 8, None, "eval", 1.0),
    ("moe_ms_b6_d8_elbo_laplace", "moe", 6, 0, 8, None, "eval", 2.0),
]


# obj iwae, LITERAL (models/objectives.py:342-359) under ref_harness.install_tuple_cuda_shim(): the line
# `lw = lpz + lpx_z.reshape(*lpz.shape) - beta * lqz_x` (:356) only runs where the decoders' leading axis has K*B rows --
# K-preserving towers (MNIST / SVHN: leading axis K) at B = 1, any towers at K = 1 (leading axis B) -- and these cases pin
# it there: K = 3 / 8, both posterior families, beta != 1, the CdSprites+ towers with a ragged text mask.
CASE_OPTS.update({
    "moe_ms_b1_d8_iwae_k3_normal": {"obj": "iwae", "K": 3, "prior": "normal"},
    "moe_ms_b1_d8_iwae_k8_laplace": {"obj": "iwae", "K": 8, "prior": "laplace"},
    "moe_ms_b4_d8_iwae_k1_laplace": {"obj": "iwae", "K": 1, "prior": "laplace"},
    "moe_b5_t6_d8_iwae_k1": {"obj": "iwae", "K": 1, "prior": "normal"},
})
for _n in ("moe_ms_b1_d8_iwae_k3_normal", "moe_ms_b1_d8_iwae_k8_laplace", "moe_ms_b4_d8_iwae_k1_laplace"):
    CASE_MODS[_n] = MS_AUTO
CASES += [
    ("moe_ms_b1_d8_iwae_k3_normal", "moe", 1, 0, 8, None, "eval", 1.0),
    ("moe_ms_b1_d8_iwae_k8_laplace", "moe", 1, 0, 8, None, "eval", 2.0),
    ("moe_ms_b4_d8_iwae_k1_laplace", "moe", 4, 0, 8, None, "eval", 0.5),
    ("moe_b5_t6_d8_iwae_k1", "moe", 5, 6, 8, [6, 2, 4, 3, 1], "eval", 1.5),
]


# recon_loss l1 / mse (models/objectives.py:427-459) through a mixer: image mse, text l1, beta != 1
CASE_MODS["mopoe_b5_t6_d8_mse_l1"] = [dict(MODS[0], ltype="mse"), dict(MODS[1], ltype="l1")]
CASES += [("mopoe_b5_t6_d8_mse_l1", "mopoe", 5, 6, 8, [6, 2, 4, 3, 1], "eval", 1.5)]


# `prior: laplace` outside mixing moe: MoPoE / DMVAE build their posteriors with a hard-coded dist.Normal
# (mmvae_models.py:363-365,480-485) -- the config's family only reaches the LIKELIHOOD `vae.px_z` (:369,495), i.e.
# recon_loss lprob takes a Laplace log-prob.  (POE refuses non-gaussian priors itself, mmvae_models.py:152.)
CASE_OPTS.update({"mopoe_ms_b5_d8_lprob_laplace": {"obj": "elbo", "K": 1, "prior": "laplace"},
                  "dmvae_ms_b5_d8p4_lprob_laplace": {"obj": "elbo", "K": 1, "prior": "laplace"}})
CASE_MODS["mopoe_ms_b5_d8_lprob_laplace"] = MS
CASE_MODS["dmvae_ms_b5_d8p4_lprob_laplace"] = MS
PRIVATE["dmvae_ms_b5_d8p4_lprob_laplace"] = 4
CASES += [("mopoe_ms_b5_d8_lprob_laplace", "mopoe", 5, 0, 8, None, "eval", 1.5),
          ("dmvae_ms_b5_d8p4_lprob_laplace", "dmvae", 5, 0, 8, None, "eval", 0.5)]


# the unimodal case (models/trainer.py:112-113): one VAE trained with UnimodalObjective.elbo
CASE_MODS["vae_cnn2_b5_d8"] = [MODS[0]]
CASE_MODS["vae_txt_b5_t6_d8"] = [MODS[1]]
CASE_MODS["vae_mnist_b6_d8_lprob"] = [MS[0]]
CASES += [
    ("vae_cnn2_b5_d8", "vae", 5, 6, 8, [6, 2, 4, 3, 1], "eval", 1.5),
    ("vae_txt_b5_t6_d8", "vae", 5, 6, 8, [6, 2, 4, 3, 1], "eval", 1.0),
    ("vae_mnist_b6_d8_lprob", "vae", 6, 0, 8, None, "eval", 0.5),
]


# one small case per mixer also keeps EVERY element of every parameter gradient (tests/golden/full_grads/<case>.npz, keys
# = the reference's parameter names): the other fixtures hold 3 norms + 96 evenly spaced samples per tensor
FULL_GRADS = {"mopoe_b6_t5_d8", "poe_b4_t5_d8", "moe_b5_t6_d8", "dmvae_b5_t6_d8p4"}


def build_reference(mixing, D, beta, private=None, mods=None, obj="elbo", K=1, prior="normal"):
    if mixing == "vae":
        m = mods[0]
        return VAE(m["enc"], m["dec"], m["data_dim"], D, m["ltype"], None, obj_fn="elbo", beta=beta, id_name="mod_1",
                   llik_scaling=m["llik_scaling"])
    vaes = {}
    for i, m in enumerate(mods or MODS):
        # prior / posterior / likelihood all follow the config's `prior` key (models/trainer.py:100-105)
        vaes[f"mod_{i + 1}"] = VAE(m["enc"], m["dec"], m["data_dim"], D, m["ltype"], private, obj_fn=obj, beta=beta,
                                   id_name=f"mod_{i + 1}", llik_scaling=m["llik_scaling"], prior_dist=prior,
                                   post_dist=prior, likelihood_dist=prior)
    return getattr(models, mixing)(nn.ModuleDict(vaes), D, {"obj": obj, "beta": beta, "K": K}, {})


def make_batch(B, T, lengths, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, 64, 64, generator=g)
    ids = torch.randint(0, 27, (B, T), generator=g)
    lens = torch.tensor(lengths)
    mask = torch.arange(T)[None, :] < lens[:, None]
    onehot = torch.nn.functional.one_hot(ids, 27).float() * mask[..., None]
    return img, onehot, mask


def run_case(name, mixing, B, T, D, lengths, mode, beta, seed=0):
    private = PRIVATE.get(name)
    base = CASE_MODS.get(name, MODS)
    mods = [dict(m, private=private) for m in base] if private else base
    opts = CASE_OPTS.get(name, {})
    model = build_reference(mixing, D, beta, private, mods, **opts)
    shapes = orc.vae_param_shapes(mods[0], D) if mixing == "vae" else orc.model_param_shapes(mods, D)
    ref_sd = model.state_dict()
    trainable = {k for k, p in model.named_parameters() if p.requires_grad}
    assert trainable == set(shapes), (trainable ^ set(shapes))
    for k, s in shapes.items():
        assert tuple(ref_sd[k].shape) == tuple(s), (k, ref_sd[k].shape, s)
    params = gw.make_params(shapes, seed)
    missing = model.load_state_dict(params, strict=False)
    assert not missing.unexpected_keys
    if mode == "eval":
        model.eval()
    else:                                   # train mode with every dropout p forced to 0
        model.train()
        for mod in model.modules():
            if isinstance(mod, nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, nn.MultiheadAttention):
                mod.dropout = 0.0
    if mixing == "vae":                 # ONE modality, keyed mod_1 as the reference's objective expects
        gm = torch.Generator().manual_seed(seed + 1)
        if mods[0]["enc"] == "MNIST":
            x = torch.rand(B, 1, 28, 28, generator=gm)
            batch = {"mod_1": {"data": x, "masks": None, "categorical": False}}
            out = {"mnist": x.numpy()}
        else:
            img, onehot, mask = make_batch(B, T, lengths, seed + 1)
            if mods[0]["enc"] == "CNN2":
                batch = {"mod_1": {"data": img, "masks": None, "categorical": False}}
                out = {"img": img.numpy()}
            else:
                batch = {"mod_1": {"data": onehot, "masks": mask, "categorical": True}}
                out = {"onehot": onehot.numpy(), "mask": mask.numpy()}
    elif mods[0]["enc"] == "MNIST":       # image pair (B,1,28,28) / (B,3,32,32) in [0,1], no masks
        gm = torch.Generator().manual_seed(seed + 1)
        mn, sv = torch.rand(B, 1, 28, 28, generator=gm), torch.rand(B, 3, 32, 32, generator=gm)
        batch = {"mod_1": {"data": mn, "masks": None, "categorical": False},
                 "mod_2": {"data": sv, "masks": None, "categorical": False}}
        out = {"mnist": mn.numpy(), "svhn": sv.numpy()}
    else:
        img, onehot, mask = make_batch(B, T, lengths, seed + 1)
        batch = {"mod_1": {"data": img, "masks": None, "categorical": False},
                 "mod_2": {"data": onehot, "masks": mask, "categorical": True}}
        out = {"img": img.numpy(), "onehot": onehot.numpy(), "mask": mask.numpy()}
    if len(mods) == 3:      # action sequences (B, Ta, joints, feats) with their own ragged lengths
        ga = torch.Generator().manual_seed(seed + 5)
        Ta, J, Fe = mods[2]["data_dim"]
        act = torch.randn(B, Ta, J, Fe, generator=ga)
        alen = torch.randint(1, Ta + 1, (B,), generator=ga)
        alen[0] = Ta
        amask = torch.arange(Ta)[None, :] < alen[:, None]
        act = act * amask[:, :, None, None]
        batch["mod_3"] = {"data": act, "masks": amask, "categorical": False}
        out["act"], out["amask"] = act.numpy(), amask.numpy()
    meta = {"name": name, "mixing": mixing, "B": B, "T": T, "D": D, "beta": beta, "seed": seed, "mode": mode,
            "mods": mods, "lr": 1e-4}
    meta.update(opts)
    if opts.get("obj") == "iwae":
        meta["shims"] = [ref_harness.install_tuple_cuda_shim()]
    if opts:
        meta["llik"] = [float(v.llik_scaling) for v in model.vaes.values()]     # "auto" resolved by the reference

    order = None
    if mixing == "poe":                     # record the hash-seed dependent subset order (utils.py:98)
        from utils import subsample_input_modalities
        subs = subsample_input_modalities(batch)
        order = [[i for i in range(len(mods)) if s[f"mod_{i + 1}"]["data"] is not None] for s in subs]
        meta["order"] = order

    torch.manual_seed(seed + 2)
    with ref_harness.EpsTape() as tape:
        res = model.objective(batch)
    res["loss"].backward()
    for i, e in enumerate(tape.draws):
        out[f"eps_{i}"] = e.numpy()
    meta["n_eps"] = len(tape.draws)
    out["loss"] = res["loss"].detach().numpy()
    out["kld"] = res["kld"].detach().numpy()
    if opts.get("obj") == "iwae":       # one (M, 2, rows) tensor [own, cross], rows = the decoders' leading axis
        res["reconstruction_loss"] = list(res["reconstruction_loss"])
    if mixing == "vae":      # (B, F) / (B, D) element tensors: keep the per-sample sums
        out["rec_0"] = res["reconstruction_loss"].detach().double().sum(-1).numpy()
        out["kld"] = res["kld"].detach().sum(-1).numpy()
    else:
        for i, r in enumerate(res["reconstruction_loss"]):
            out[f"rec_{i}"] = r.detach().numpy()
    if mixing == "moe" and opts.get("obj", "elbo") == "elbo":      # MoE elbo leaves the trainable model prior untouched: the reference has no gradient for it
        for k, q in model.named_parameters():
            if q.requires_grad and q.grad is None:
                q.grad = torch.zeros_like(q)

    # intermediates straight from the reference's own methods (deterministic in this mode)
    with torch.no_grad():
        if mixing == "mopoe":
            lat = model.modality_mixing(batch)
            for i in range(len(mods)):
                mu, lv = lat["modalities"][f"mod_{i + 1}"]["shared"]
                out[f"enc_mu_{i}"], out[f"enc_lv_{i}"] = mu.numpy(), lv.numpy()
            out["joint_mu"], out["joint_var"] = lat["joint"][0].numpy(), lat["joint"][1].numpy()
            for k, (mu, var) in lat["subsets"].items():
                out[f"subset_mu/{k}"], out[f"subset_var/{k}"] = mu.squeeze(0).numpy(), var.squeeze(0).numpy()
            with ref_harness.EpsTape(replay=tape.draws):
                fw = model.forward(batch)
            for i in range(len(mods)):
                o = fw.mods[f"mod_{i + 1}"]
                out[f"z_{i}"] = o.latent_samples["latents"].numpy()
                out[f"recon_{i}"] = gw.summarize(o.decoder_dist.loc, 256)
        elif mixing == "vae":
            mu, lv = model.enc(batch["mod_1"])
            out["enc_mu_0"], out["enc_lv_0"] = mu.numpy(), lv.numpy()
        else:
            for i in range(len(mods)):
                mu, lv = model.vaes[f"mod_{i + 1}"].enc(batch[f"mod_{i + 1}"])
                out[f"enc_mu_{i}"], out[f"enc_lv_{i}"] = mu.numpy(), lv.numpy()

    for k, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None, k
            out[f"g/{k}"] = gw.summarize(p.grad)
    if name in FULL_GRADS:
        os.makedirs(os.path.join(HERE, "full_grads"), exist_ok=True)
        np.savez_compressed(os.path.join(HERE, "full_grads", name + ".npz"),
                            **{k: p.grad.detach().numpy() for k, p in model.named_parameters() if p.requires_grad})
    # one optimiser step exactly as models/trainer.py:79-81
    opt = torch.optim.Adam(filter(lambda q: q.requires_grad, model.parameters()), lr=meta["lr"], amsgrad=True)
    opt.step()
    for k, p in model.named_parameters():
        if p.requires_grad:
            out[f"a/{k}"] = gw.summarize(p.data)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: loss={float(res['loss']):.6f} kld={float(res['kld'].sum()):.6f} order={order} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    assert os.environ.get("PYTHONHASHSEED") == "0", "run with PYTHONHASHSEED=0 (PoE subset order, utils.py:98)"
    only = sys.argv[1:]
    for c in CASES:
        if not only or any(o in c[0] for o in only):
            run_case(*c)
