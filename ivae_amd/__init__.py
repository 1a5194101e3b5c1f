"""ivae_amd -- MI355X-native implementation of the iVAE training hot path.

Drop-in counterparts of the reference's ``models`` / ``metrics`` exports for that path:
``iVAE``, ``VAE``, ``cleanIVAE``, ``cleanVAE``, ``MLP``, ``Normal``, ``mean_corr_coef``.
"""
from .nets import MLP, Normal, VAE, cleanIVAE, cleanVAE, iVAE, weights_init  # noqa: F401
from .mcc import mean_corr_coef, mean_corr_coef_gpu  # noqa: F401
from .optim import FusedAdam  # noqa: F401

__all__ = ["MLP", "Normal", "VAE", "cleanIVAE", "cleanVAE", "iVAE", "weights_init", "mean_corr_coef",
           "mean_corr_coef_gpu", "FusedAdam"]
