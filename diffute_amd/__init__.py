"""diffute_amd - MI355X-native (gfx950) hot path of DiffUTE behind the reference's diffusers surface.

Public names mirror what train_diffute_v1.py / app.ipynb import from diffusers
(`AutoencoderKL, DDPMScheduler, UNet2DConditionModel`, train_diffute_v1.py:51) so the scripts can
switch with `from diffute_amd import ...`.
"""
from .models import (AutoencoderKL, UNet2DConditionModel, DiagonalGaussianDistribution, TrOCREncoder,
                     SD2_INPAINT_UNET_CONFIG, SD_VAE_CONFIG, TROCR_LARGE_VIT_CONFIG)
from .schedulers import DDIMScheduler, DDPMScheduler, SD2_SCHEDULER_CONFIG
from .pipeline import denoise, edit_latents, mask_to_latent
from .optim import FusedAdamW, GradScaler
from ._cabi import set_exclusive_device, synchronize

__all__ = ["AutoencoderKL", "UNet2DConditionModel", "DDPMScheduler", "DDIMScheduler", "denoise", "edit_latents",
           "mask_to_latent", "FusedAdamW", "GradScaler", "TrOCREncoder", "TROCR_LARGE_VIT_CONFIG", "DiagonalGaussianDistribution", "SD2_INPAINT_UNET_CONFIG", "SD_VAE_CONFIG",
           "SD2_SCHEDULER_CONFIG", "set_exclusive_device", "synchronize"]
__version__ = "0.1.0"
from . import prepost  # noqa: E402,F401  (on-device pre/post-processing)
