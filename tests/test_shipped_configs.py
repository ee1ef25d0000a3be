"""BASELINE north_star: "configs/config_*.yml and the models/ registry still select it unchanged".

Every configuration file the reference ships (configs/config_*.yml, captured as data by
tests/golden/make_config_fixture.py) is written back to disk unmodified and handed to MultimodalVAE exactly as
main.py:41-43 hands it to the reference: Config(path) + the dataset class's feature_dims.  A configuration either
constructs the model its keys describe, or fails with the documented error naming the out-of-scope tower
(SURVEY.md section 8 / DESIGN.md section 7).  CPU only: construction launches no kernel.
"""
import json
import os

import pytest
import torch
import yaml

from conftest import GOLDEN_DIR

with open(os.path.join(GOLDEN_DIR, "shipped_configs.json")) as f:
    SHIPPED = json.load(f)

# file -> what construction must do.  ("ok", mixer class, checks) or ("error", substring of the message)
EXPECT = {
    "config_mnistsvhn.yml": ("ok", "MOE"),
    # `encoder: CNN` = the ResNet-50 tower (models/resnet.py)
    "config_cdspritesplus.yml": ("ok", "MOE"),
    "config_cub.yml": ("ok", "MOE"),
    "config_vilanro.yml": ("ok", "POE"),
    "config_celeba.yml": ("error", "tower 'FNN'"),
    "config_fashionmnist.yml": ("error", "tower 'FNN'"),
    "config_polymnist.yml": ("error", "tower 'PolyMNIST'"),
    "config_sprites.yml": ("error", "tower 'VideoGPT'"),
}


def _construct(name, tmp_path):
    from multimodal_vae_comparison_amd.models.config_cls import Config
    from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
    path = tmp_path / name
    with open(path, "w") as f:
        yaml.safe_dump(SHIPPED[name]["config"], f)
    return MultimodalVAE(Config(str(path)), feature_dims=SHIPPED[name]["feature_dims"], device="cpu")


def test_fixture_covers_every_shipped_config():
    assert set(SHIPPED) == set(EXPECT) and len(SHIPPED) == 8


@pytest.mark.parametrize("name", sorted(EXPECT))
def test_shipped_config_selects_the_path(name, tmp_path):
    kind, what = EXPECT[name]
    if kind == "error":
        with pytest.raises(NotImplementedError, match=what):
            _construct(name, tmp_path)
        return
    tr = _construct(name, tmp_path)
    cfg = SHIPPED[name]["config"]
    assert type(tr.model).__name__ == what and tr.model.modelName == cfg["mixing"]
    assert tr.model.n_latents == cfg["n_latents"] and tr.model.obj_fn.obj_name == cfg["obj"]
    assert tr.model.K == cfg.get("K", 1)
    mods = [cfg[k] for k in sorted(k for k in cfg if k.startswith("modality_"))]
    for m, vae in zip(mods, tr.model.vaes.values()):
        assert type(vae.enc).__name__ == "Enc_" + m["encoder"] and type(vae.dec).__name__ == "Dec_" + m["decoder"]
        assert vae.ltype == m["recon_loss"] and vae.prior_str == m.get("prior", "normal")
    opt = tr.configure_optimizers()
    assert opt.param_groups[0]["lr"] == float(cfg["lr"]) and opt.param_groups[0]["amsgrad"]


def test_mnistsvhn_config_details(tmp_path):
    """configs/config_mnistsvhn.yml: moe, dreg, K 30, prior laplace on both modalities, llik_scaling auto"""
    tr = _construct("config_mnistsvhn.yml", tmp_path)
    m = tr.model
    assert m.K == 30 and m._laplace == [True, True]
    # set_likelihood_scales (mmvae_base.py:41-47): min prod(data_dim) / prod(data_dim_m) = 784 / {784, 3072}
    assert [float(v.llik_scaling) for v in m.vaes.values()] == pytest.approx([1.0, 784.0 / 3072.0], rel=1e-12)
    assert all(isinstance(v.post_dist, type) and v.post_dist is torch.distributions.Laplace for v in m.vaes.values())


def test_vilanro_config_with_the_plain_conv_encoder(tmp_path):
    """configs/config_vilanro.yml (PoE over language [4,9,1] + actions [100,4,1] + image, optimal_sigma everywhere)
    with `encoder: CNN` -> `CNN2` (the plain conv tower; CNN = ResNet-50 is SURVEY 8(f) rank 1): the 8 / 4-layer
    action Transformer towers at the reference's Ta = 100"""
    import copy
    from multimodal_vae_comparison_amd.models.config_cls import Config
    from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
    rec = copy.deepcopy(SHIPPED["config_vilanro.yml"])
    rec["config"]["modality_3"]["encoder"] = "CNN2"
    path = tmp_path / "config_vilanro_cnn2.yml"
    with open(path, "w") as f:
        yaml.safe_dump(rec["config"], f)
    tr = MultimodalVAE(Config(str(path)), feature_dims=rec["feature_dims"], device="cpu")
    m = tr.model
    assert type(m).__name__ == "POE" and len(m.vaes) == 3
    assert [type(v.enc).__name__ for v in m.vaes.values()] == ["Enc_TxtTransformer", "Enc_Transformer", "Enc_CNN2"]
    assert m.vaes["mod_2"].enc.data_dim == [100, 4, 1] and len(m.vaes["mod_2"].enc.seqTransEncoder.layers) == 8
    assert all(v.ltype == "optimal_sigma" for v in m.vaes.values())


@pytest.mark.skipif(not os.path.isdir("/root/reference/multimodal_compare/configs"), reason="reference tree absent")
def test_fixture_matches_the_reference_files():
    """in the build container: the JSON fixture is exactly what the reference's files parse to"""
    for name, rec in SHIPPED.items():
        with open(os.path.join("/root/reference/multimodal_compare/configs", name)) as f:
            assert yaml.safe_load(f) == rec["config"], name
