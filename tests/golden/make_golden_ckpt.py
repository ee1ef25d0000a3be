"""Checkpoint layout of the REFERENCE as data (build container only):  python tests/golden/make_golden_ckpt.py

For two reference models (MoPoE on the CdSprites+ towers, MoE on the MNIST / SVHN towers) this records what a
checkpoint written by the reference's Lightning trainer holds (main.py:46 ModelCheckpoint -> `state_dict` of the
LightningModule = `model.` + TorchMMVAE.state_dict(), `optimizer_states` = [torch.optim.Adam(amsgrad).state_dict()]):
  * every state_dict key in order, its shape / dtype, whether it is a parameter (and trainable) or a buffer
    (the PositionalEncoding `pe` tables), including the `.module.` segments of the nn.DataParallel wrappers;
  * the `pe` buffer contents (summaries: l2 / sum / abs-sum + 96 samples);
  * torch's Adam state_dict after one step on the golden weights: param_groups verbatim, the parameter order behind
    the integer ids, per-entry keys / shapes / dtypes, the step value.
-> tests/golden/ckpt_layout.json.  tests/test_checkpoint_layout.py holds the package's checkpoints to it.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import torch
import torch.nn as nn

import ref_harness

ref_harness.install()
import models  # noqa: E402
from models.vae import VAE  # noqa: E402

from oracle import golden_weights as gw  # noqa: E402
from oracle import mmvae_oracle as orc  # noqa: E402

CASES = {
    "mopoe_cdsprites_d8": ("mopoe", 8, "elbo", 1, "normal",
                           [{"enc": "CNN2", "dec": "CNN", "data_dim": [64, 64, 3], "ltype": "bce"},
                            {"enc": "TxtTransformer", "dec": "TxtTransformer", "data_dim": [45, 27, 1],
                             "ltype": "category_ce"}]),
    # MoE with obj elbo never touches the model-level prior `_pz_params.1` (requires_grad, mmvae_base.py:37): torch's Adam
    # creates no state entry for it -- a genuine reference checkpoint of this config has one entry fewer than parameters
    "moe_cdsprites_d8_elbo": ("moe", 8, "elbo", 1, "normal",
                              [{"enc": "CNN2", "dec": "CNN", "data_dim": [64, 64, 3], "ltype": "bce"},
                               {"enc": "TxtTransformer", "dec": "TxtTransformer", "data_dim": [45, 27, 1],
                                "ltype": "category_ce"}]),
    "moe_mnistsvhn_d8": ("moe", 8, "dreg", 2, "laplace",
                         [{"enc": "MNIST", "dec": "MNIST", "data_dim": [28, 28, 1], "ltype": "lprob"},
                          {"enc": "SVHN", "dec": "SVHN", "data_dim": [32, 32, 3], "ltype": "lprob"}]),
}

out = {}
for name, (mixing, D, obj, K, prior, mods) in CASES.items():
    vaes = {f"mod_{i + 1}": VAE(m["enc"], m["dec"], m["data_dim"], D, m["ltype"], None, obj_fn=obj, beta=1,
                                id_name=f"mod_{i + 1}", llik_scaling=1, prior_dist=prior, post_dist=prior,
                                likelihood_dist=prior) for i, m in enumerate(mods)}
    model = getattr(models, mixing)(nn.ModuleDict(vaes), D, {"obj": obj, "beta": 1, "K": K}, {})
    model.load_state_dict(gw.make_params(orc.model_param_shapes(mods, D), 0), strict=False)
    model.eval()
    pnames = {id(p): k for k, p in model.named_parameters()}
    params = dict(model.named_parameters())
    sd = []
    for k, v in model.state_dict().items():
        kind = "param" if k in params else "buffer"
        sd.append({"key": k, "shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", ""), "kind": kind,
                   "requires_grad": bool(params[k].requires_grad) if kind == "param" else False})
    pe = {k: [float(x) for x in gw.summarize(v)] for k, v in model.state_dict().items() if k.endswith(".pe")}
    # one optimisation step exactly as models/trainer.py:79-81 + Lightning would run it
    g = torch.Generator().manual_seed(5)
    B = 3
    if mods[0]["enc"] == "MNIST":
        batch = {"mod_1": {"data": torch.rand(B, 1, 28, 28, generator=g), "masks": None, "categorical": False},
                 "mod_2": {"data": torch.rand(B, 3, 32, 32, generator=g), "masks": None, "categorical": False}}
    else:
        T = 4
        ids = torch.randint(0, 27, (B, T), generator=g)
        mask = torch.arange(T)[None, :] < torch.tensor([4, 2, 3])[:, None]
        batch = {"mod_1": {"data": torch.rand(B, 3, 64, 64, generator=g), "masks": None, "categorical": False},
                 "mod_2": {"data": torch.nn.functional.one_hot(ids, 27).float() * mask[..., None], "masks": mask,
                           "categorical": True}}
    torch.manual_seed(7)
    opt = torch.optim.Adam(filter(lambda q: q.requires_grad, model.parameters()), lr=1e-4, amsgrad=True)
    model.objective(batch)["loss"].backward()
    opt.step()
    osd = opt.state_dict()
    order = [pnames[id(p)] for p in opt.param_groups[0]["params"]]
    entries = {}
    for i, st in osd["state"].items():
        entries[str(i)] = {k: {"shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", ""),
                               **({"value": float(v)} if k == "step" else {})} for k, v in st.items()}
    groups = [{k: (list(v) if isinstance(v, tuple) else v) for k, v in gdict.items()} for gdict in osd["param_groups"]]
    out[name] = {"mixing": mixing, "D": D, "obj": obj, "K": K, "prior": prior, "mods": mods, "state_dict": sd, "pe": pe,
                 "adam": {"param_groups": groups, "param_order": order, "state": entries}}
    print(name, len(sd), "state_dict entries,", len(order), "optimizer parameters,", len(entries), "with state")
with open(os.path.join(HERE, "ckpt_layout.json"), "w") as f:
    json.dump(out, f, indent=0, sort_keys=True)
