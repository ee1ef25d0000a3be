"""YAML config with the reference's schema (reference: models/config_cls.py:8-138; keys of
configs/config_cdspritesplus.yml).  Only parsing: run-directory versioning / CLI plumbing are host-side
orchestration outside the hot path."""
import yaml

_MOD_DEFAULTS = {"private_latents": None, "llik_scaling": 1, "prior": "normal", "recon_loss": None,
                 "mod_type": None, "path": None, "test_datapath": None}
_DEFAULTS = {"beta": 1, "K": 1, "obj": "elbo", "optimizer": "adam", "lr": 1e-4, "batch_size": 32, "seed": 1,
             "pre_trained": None, "model_cfg": None, "labels": None}


class Config:
    def __init__(self, source, **overrides):
        if isinstance(source, str):
            with open(source) as f:
                cfg = yaml.safe_load(f)
        else:
            cfg = dict(source)
        cfg.update(overrides)
        for k, v in _DEFAULTS.items():
            cfg.setdefault(k, v)
        self.mods = []
        for k in sorted((k for k in cfg if k.startswith("modality_")), key=lambda s: int(s.split("_")[1])):
            m = dict(_MOD_DEFAULTS)
            m.update(cfg[k])
            self.mods.append(m)
        for k, v in cfg.items():
            setattr(self, k, v)
        self.params = cfg
