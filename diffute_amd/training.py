"""The DiffUTE training step (train_diffute_v1.py:859-935) on the product classes.

    latents / masked latents  : frozen VAE encode -> latent_dist.sample() * scaling_factor      (:875-887)
    mask                      : nearest downsample to the latent grid                             (:880-884)
    noisy latents             : scheduler.add_noise(latents, noise, timesteps)                    (:889-897)
    prediction                : unet(cat([noisy, mask, masked_latents], 1), timesteps, ocr_embeddings).sample   (:911-913)
    loss                      : mse_loss(pred.float(), target.float())                            (:918)
    backward / clip / AdamW   : loss.backward(); clip_grad_norm_(unet.parameters(), 1.0); optimizer.step()   (:925-930)

Every FLOP of the models runs in the HIP library; torch provides the autograd plumbing.  The optimizer is either
diffute_amd.FusedAdamW (clipping + AdamW + EMA as one HIP pass over the packed fp32 arenas) or any
torch optimizer over `unet.parameters()` (the backward then exports per-parameter gradients into `.grad`).  With
`dist` set, gradients are averaged across ranks inside the backward (bucketed RCCL all-reduce of the packed gradient
arena on a side stream, UNet2DConditionModel.set_gradient_sync)."""
import torch

from .models import mse_loss
from .pipeline import mask_to_latent


def encode_latents(vae, images, noise=None, generator=None):
    """vae.encode(x).latent_dist.sample() * scaling_factor, no gradient (the VAE is frozen, train_diffute_v1.py:639)."""
    with torch.no_grad():
        dist = vae.encode(images).latent_dist
        z = dist.sample(noise=noise) if noise is not None else dist.sample(generator=generator)
        return z * vae.config.scaling_factor


def training_target(scheduler, latents, noise, timesteps):
    pt = scheduler.config.prediction_type
    if pt == "epsilon":
        return noise
    if pt == "v_prediction":
        return scheduler.get_velocity(latents, noise, timesteps)
    raise ValueError(f"Unknown prediction type {pt}")                      # train_diffute_v1.py:909


def train_step(unet, vae, scheduler, optimizer, batch, *, noise=None, timesteps=None, enc_noise=None, enc_noise_masked=None,
               max_grad_norm=1.0, generator=None, scaler=None):
    """One optimizer step.  batch: dict with pixel_values [B,3,H,W], masked_images [B,3,H,W], masks [B,1,H,W] and
    ocr_embeddings [B,S,1024] (the frozen TrOCR encoder's output, train_diffute_v1.py:868-871).  The random draws can be
    injected (tests); otherwise they come from torch's device RNG like the reference.  Returns loss / grad_norm tensors.
    scaler: a diffute_amd.GradScaler (or torch.amp.GradScaler with a torch optimizer) - `--mixed_precision fp16` (train_diffute_v1.py:267,583):
    backward on the scaled loss, unscale before clipping, the step skipped and the scale halved when the gradient overflowed."""
    pv = batch["pixel_values"]
    # both frozen-VAE encodes (train_diffute_v1.py:875,886) as one batch of 2B images: same arithmetic per image, half the launches
    nz = None if enc_noise is None else torch.cat([enc_noise, enc_noise_masked], 0)
    both = encode_latents(vae, torch.cat([pv, batch["masked_images"]], 0), nz, generator)
    latents, masked_latents = both[:pv.shape[0]].contiguous(), both[pv.shape[0]:].contiguous()
    factor = 2 ** (len(vae.config.block_out_channels) - 1)
    mask = mask_to_latent(batch["masks"], factor).to(latents.dtype)
    B = latents.shape[0]
    if noise is None:
        noise = torch.randn(latents.shape, device=latents.device, dtype=latents.dtype, generator=generator)
    if timesteps is None:
        timesteps = torch.randint(0, scheduler.num_train_timesteps, (B,), device=latents.device, generator=generator).long()
    noisy = scheduler.add_noise(latents, noise, timesteps)
    target = training_target(scheduler, latents, noise, timesteps)
    pred = unet(torch.cat([noisy, mask, masked_latents], dim=1), timesteps, batch["ocr_embeddings"]).sample
    loss = mse_loss(pred.float(), target.float())
    (loss if scaler is None else scaler.scale(loss)).backward()
    if hasattr(optimizer, "masters"):                 # diffute_amd.optim.FusedAdamW: (unscaling,) clipping and the update are one HIP pass
        if scaler is None:
            optimizer.step()
        else:
            scaler.step(optimizer); scaler.update()
        grad_norm = optimizer.grad_norm
    else:
        if scaler is not None:
            scaler.unscale_(optimizer)
        grad_norm = torch.nn.utils.clip_grad_norm_(unet.parameters(), max_grad_norm) if max_grad_norm else None
        if scaler is None:
            optimizer.step()
        else:
            scaler.step(optimizer); scaler.update()
        optimizer.zero_grad(set_to_none=True)
    return dict(loss=loss.detach(), grad_norm=grad_norm)


def train_vae_step(vae, optimizer, images, target=None, max_grad_norm=None, scaler=None):
    """One optimizer step of train_vae.py:716-736: pred = vae(x)["sample"] (decode of the posterior mode),
    loss = mse_loss(pred.float(), target.float()), backward, optimizer step.  `target` defaults to the input images.
    scaler: loss scaling for the fp16 build (see train_step)."""
    pred = vae(images)["sample"]
    loss = mse_loss(pred.float(), (images if target is None else target).float())
    (loss if scaler is None else scaler.scale(loss)).backward()
    if scaler is not None:
        scaler.unscale_(optimizer)
    grad_norm = torch.nn.utils.clip_grad_norm_(vae.parameters(), max_grad_norm) if max_grad_norm else None
    if scaler is None:
        optimizer.step()
    else:
        scaler.step(optimizer); scaler.update()
    optimizer.zero_grad(set_to_none=True)
    return dict(loss=loss.detach(), grad_norm=grad_norm)
