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
           "n_latents": n_latents, "obj": obj, "optimizer": "adam", "K": K}     # one modality: the unimodal VAE
    dims = {}
    for i, m in enumerate(mods):
        cfg[f"modality_{i + 1}"] = {"decoder": m["dec"], "encoder": m["enc"], "mod_type": f"m{i + 1}",
                                    "recon_loss": m["ltype"], "prior": m.get("prior", prior),
                                    "private_latents": m.get("private"), "llik_scaling": m.get("llik_scaling", 1)}
        dims[f"m{i + 1}"] = list(m["data_dim"])
    return cfg, dims


# ---- the BASELINE.json workloads (SURVEY 8(d) "Synthetic inputs") ---------------------------------------------------
CD_MODS = [{"enc": "CNN2", "dec": "CNN", "data_dim": [64, 64, 3], "ltype": "bce"},
           {"enc": "TxtTransformer", "dec": "TxtTransformer", "data_dim": [45, 27, 1], "ltype": "category_ce"}]
MS_MODS = [{"enc": "MNIST", "dec": "MNIST", "data_dim": [28, 28, 1], "ltype": "lprob"},
           {"enc": "SVHN", "dec": "SVHN", "data_dim": [32, 32, 3], "ltype": "lprob"}]


def mnist_svhn_batch(B, seed=1, device="cpu"):
    """MNIST (B,1,28,28) + SVHN (B,3,32,32) images, U[0,1), no masks (models/datasets.py:416-495)"""
    g = torch.Generator().manual_seed(seed)
    return {"mod_1": {"data": torch.rand(B, 1, 28, 28, generator=g).to(device), "masks": None, "categorical": False},
            "mod_2": {"data": torch.rand(B, 3, 32, 32, generator=g).to(device), "masks": None, "categorical": False}}


def vilanro_batch(B, T, Ta, seed=1, device="cpu"):
    """image + text as CdSprites+ and ragged action sequences (B, Ta, 4, 1) ~ N(0,1) with their masks (cfg5)"""
    b = cdsprites_batch(B, T, seed, device)
    g = torch.Generator().manual_seed(seed + 9)
    act = torch.randn(B, Ta, 4, 1, generator=g)
    alen = torch.randint(1, Ta + 1, (B,), generator=g)
    alen[0] = Ta
    amask = torch.arange(Ta)[None, :] < alen[:, None]
    b["mod_3"] = {"data": (act * amask[:, :, None, None]).to(device), "masks": amask.to(device), "categorical": False}
    return b


WORKLOADS = {
    # name: (description, mixing, mods, n_latents, per-GPU batch, T, extra config)
    "cfg1": ("configs[0]: MVAE (PoE), CdSprites+ L1 shapes, 64x64 image + 8-token text, n_latents=16, batch=32",
             "poe", CD_MODS, 16, 32, 8, {}),
    "cfg2": ("configs[1]: MoPoE, CdSprites+ L2 shapes, CNN2 image tower + TxtTransformer text tower, n_latents=32, "
             "batch=128/GPU, T=32", "mopoe", CD_MODS, 32, 128, 32, {}),
    "cfg3": ("configs[2] as stated: MMVAE (MoE), K-sample IWAE K=8, CdSprites+ L5 shapes, n_latents=32, batch=256, T=32 "
             "(every decoder decodes M*K*B = 4096 latent samples; K-preserving text decoder = defined extension, the "
             "reference's collapses K: parity unpinned for K>1, the iwae formula itself is pinned by reference fixtures)",
             "moe", CD_MODS, 32, 256, 32, {"obj": "iwae", "K": 8}),
    "cfg3_elbo": ("configs[2] with the objective the reference can run on these towers: MMVAE (MoE), obj elbo K=1, "
                  "CdSprites+ L5 shapes, n_latents=32, batch=256, T=32", "moe", CD_MODS, 32, 256, 32, {}),
    "cfg2_rnn": ("configs[1] with `encoder: TxtRNN` for the text modality (Embedding -> bi-GRU(512) -> Linear; a defined "
                 "path, the reference's crashes): MoPoE, CNN2 image tower, TxtRNN encoder + TxtTransformer decoder, "
                 "n_latents=32, batch=128, T=32", "mopoe", [CD_MODS[0], dict(CD_MODS[1], enc="TxtRNN")], 32, 128, 32, {}),
    "cfg4": ("configs[3]: DMVAE shared/private latents on MNIST-SVHN, MLP + conv towers, n_latents=20 + 10 private, "
             "lprob, batch=512", "dmvae", [dict(m, private=10) for m in MS_MODS], 20, 512, 0, {}),
    "cfg5": ("configs[4]: MoPoE on image + text + action sequences (Ta=100, 8/4-layer ff-1024 Transformer towers), "
             "optimal_sigma on the actions, n_latents=32, batch=128",
             "mopoe", CD_MODS + [{"enc": "Transformer", "dec": "Transformer", "data_dim": [100, 4, 1],
                                  "ltype": "optimal_sigma"}], 32, 128, 32, {}),
    "cdsprites_shipped": ("the shipped configs/config_cdspritesplus.yml: MoE elbo, `encoder: CNN` = ResNet-50 image "
                          "tower (random init; parity pinned to the oracle's restatement and through it to transformers' independent ResNet v1.5, UNPINNED vs torchvision itself) + "
                          "Dec_CNN, TxtTransformer text towers, n_latents=24, batch=24, T=32",
                          "moe", [dict(CD_MODS[0], enc="CNN"), CD_MODS[1]], 24, 24, 32, {}),
    "mnistsvhn": ("the shipped configs/config_mnistsvhn.yml: MoE, obj dreg, K=30, prior laplace, llik_scaling auto, "
                  "n_latents=20, batch=128", "moe", [dict(m, llik_scaling="auto") for m in MS_MODS], 20, 128, 0,
                  {"obj": "dreg", "K": 30, "prior": "laplace"}),
}


def workload(name, batch=None, device="cpu", seed=1):
    """(description, config dict, feature_dims, batch dict, meta) of a BASELINE workload"""
    desc, mixing, mods, D, B, T, extra = WORKLOADS[name]
    B = batch or B
    cfg, dims = config_from_mods(mixing, mods, D, batch_size=B, **extra)
    if mods[0]["enc"] == "MNIST":
        data = mnist_svhn_batch(B, seed, device)
    elif len(mods) == 3:
        data = vilanro_batch(B, T, mods[2]["data_dim"][0], seed, device)
    else:
        data = cdsprites_batch(B, T, seed, device)
    return desc, cfg, dims, data, {"mixing": mixing, "mods": mods, "D": D, "B": B, "T": T, **extra}


# forward multiply-accumulates per sample of one tower pass (SURVEY Appendix D for the CdSprites+ towers; the others
# counted the same way: conv = out positions x Cout x Cin x 16, linear = in x out)
def tower_macs(m, D, T=32):
    Dp = D + (m.get("private") or 0)
    enc = {"CNN": 335_000_000 + 1000 * 2 * Dp,            # ResNet-50 at 64x64 (~335 MMAC) + heads
           "CNN2": 7372800 - 32768 + 512 * 2 * Dp, "TxtTransformer": 929664 * T // 32,
           "TxtRNN": T * 3 * 512 * 512 + 3 * 512 * 512 + 512 * 2 * Dp,      # T recurrent steps (W_hh) + the reverse cell + o2p
           "MNIST": 784 * 400 + 400 * 400 + 400 * 2 * Dp,
           "SVHN": 393216 + 2097152 + 1048576 + 131072 + 128 * 2 * Dp}
    dec = {"CNN": 7618560 - 32 * 512 + Dp * 512, "TxtTransformer": 556032 * T // 32, "MNIST": Dp * 400 + 400 * 400 + 400 * 784,
           "SVHN": Dp * 128 + 131072 + 1048576 + 2097152 + 393216}
    if m["enc"] == "Transformer":     # per token: 8 x (4 d^2 + 2 d ff) + attention 2 T d ; decoder 4 x (8 d^2 + 2 d ff)
        Ta, d, ff = m["data_dim"][0], D, 1024
        return (Ta * (8 * (4 * d * d + 2 * d * ff + 2 * Ta * d) + 4 * d + d * d) + d * 2 * D,
                Ta * (4 * (8 * d * d + 2 * d * ff + 2 * Ta * d) + 4 * d))
    return enc[m["enc"]], dec[m["dec"]]


def step_flops_per_sample(meta):
    """algorithmic FLOPs per sample and training step: 2 x MACs x 3 (forward + data gradient + weight gradient),
    with the number of encoder / decoder passes each mixer makes"""
    M = len(meta["mods"])
    K = meta.get("K", 1)
    passes = {"mopoe": (1, 1), "poe": (2 ** (M - 1), 2 ** M - 1), "moe": (1, M * K), "dmvae": (1, 3)}[meta["mixing"]]
    macs = 0
    for m in meta["mods"]:
        e, d = tower_macs(m, meta["D"], meta["T"])
        macs += passes[0] * e + passes[1] * d
    return 6.0 * macs
