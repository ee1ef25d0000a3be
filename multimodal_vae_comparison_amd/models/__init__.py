"""Registry: lower-case names -> mixer classes, exactly as the reference's models/__init__.py:1-8 so that
`getattr(models, cfg.mixing.lower())` (models/trainer.py:108) keeps selecting the model from the YAML config."""
from .mmvae_models import MOE as moe
from .mmvae_models import POE as poe
from .mmvae_models import MoPOE as mopoe
from .mmvae_models import DMVAE as dmvae
from .vae import VAE

__all__ = ["moe", "poe", "mopoe", "dmvae", "VAE"]
