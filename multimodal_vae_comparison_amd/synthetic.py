"""Synthetic CdSprites+-shaped batches: image U[0,1) (B,3,64,64); text token ids U{0..26} ->
one-hot (B,T,27) fp32, lengths U{3..T} with sample 0 at full length, mask = arange(T) < len, one-hot zeroed at
padding -- the format DataModule.collate_fn produces (reference: models/dataloader.py:104-120,
models/datasets.py:251-282)."""
import torch


def cdsprites_batch(B, T, seed=1, device="cpu", V=27):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, 64, 64, generator=g)
    ids = torch.randint(0, V, (B, T), generator=g)
    lens = torch.randint(min(3, T), T + 1, (B,), generator=g)
    lens[0] = T
    mask = torch.arange(T)[None, :] < lens[:, None]
    onehot = torch.nn.functional.one_hot(ids, V).float() * mask[..., None]
    return {"mod_1": {"data": img.to(device), "masks": None, "categorical": False},
            "mod_2": {"data": onehot.to(device), "masks": mask.to(device), "categorical": True}}


def cdsprites_config(mixing="mopoe", n_latents=32, batch_size=128, beta=1, lr=1e-4, private=None):
    """the reference's configs/config_cdspritesplus.yml schema with the CNN2 image tower"""
    return {"batch_size": batch_size, "beta": beta, "dataset_name": "cdspritesplus", "lr": lr, "mixing": mixing,
            "n_latents": n_latents, "obj": "elbo", "optimizer": "adam", "K": 1,
            "modality_1": {"decoder": "CNN", "encoder": "CNN2", "mod_type": "image", "recon_loss": "bce",
                           "prior": "normal", "private_latents": private},
            "modality_2": {"decoder": "TxtTransformer", "encoder": "TxtTransformer", "mod_type": "text",
                           "recon_loss": "category_ce", "prior": "normal", "private_latents": private}}


def config_from_mods(mixing, mods, n_latents, batch_size=128, beta=1, lr=1e-4, obj="elbo", K=1, prior="normal"):
    """(config dict, feature_dims) for an arbitrary modality list [{"enc", "dec", "data_dim", "ltype", "private"?,
    "llik_scaling"?}] in the reference's YAML schema; the parity tests build their models from fixture metadata"""
    cfg = {"batch_size": batch_size, "beta": beta, "dataset_name": "synthetic", "lr": lr, "mixing": mixing,
           "n_latents": n_latents, "obj": obj, "optimizer": "adam", "K": K}
    dims = {}
    for i, m in enumerate(mods):
        cfg[f"modality_{i + 1}"] = {"decoder": m["dec"], "encoder": m["enc"], "mod_type": f"m{i + 1}",
                                    "recon_loss": m["ltype"], "prior": m.get("prior", prior),
                                    "private_latents": m.get("private"), "llik_scaling": m.get("llik_scaling", 1)}
        dims[f"m{i + 1}"] = list(m["data_dim"])
    return cfg, dims
