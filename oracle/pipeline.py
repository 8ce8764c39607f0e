"""Oracle: the denoise loop and the training-step forward that drive the models.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
  * denoise()      restates app.ipynb:796-816 (no CFG, one UNet call per step,
                   9-channel cat([latents, mask, masked_latents]) input).
  * mask_to_latent restates F.interpolate(mask, size=(h/8, w/8)) default nearest
                   (app.ipynb:787-791, train_diffute_v1.py:880-884): src = floor(dst*8).
  * train_forward  restates train_diffute_v1.py:875-918 with every random draw injected.
"""
import numpy as np
import torch

from . import schedulers as S
from .unet import unet_forward
from .vae import vae_encode_moments, gaussian_sample


def mask_to_latent(mask, factor=8):
    """mask [B,1,H,W] -> [B,1,H/f,W/f] nearest: out[y,x] = in[floor(y*f), floor(x*f)]."""
    return mask[:, :, ::factor, ::factor].to(torch.float32).contiguous()


@torch.no_grad()
def denoise(P, cfg, latents, mask, masked_latents, ctx, steps, scheduler="ddim",
            noise=None, emulate_bf16=False, ac=None, N=1000, trace=None):
    """Returns final latents [B,4,h,w] fp32.  noise: [steps,B,4,h,w] for DDPM (injected
    in place of the device RNG draw inside DDPMScheduler.step, app.ipynb:816)."""
    if ac is None:
        ac = S.make_tables(N)[2]
    ts = S.timesteps_ddim(steps, N) if scheduler == "ddim" else S.timesteps_ddpm(steps, N)
    x = latents.to(torch.float32).clone()          # init_noise_sigma == 1.0 (app.ipynb:800)
    for i, t in enumerate(ts):
        inp = torch.cat([x, mask.to(torch.float32), masked_latents.to(torch.float32)], dim=1)
        eps = unet_forward(P, cfg, inp, torch.tensor(int(t)), ctx, emulate_bf16=emulate_bf16)
        if scheduler == "ddim":
            xn = S.ddim_step(ac, eps.numpy(), int(t), x.numpy(), steps, N)
        else:
            nz = None if noise is None else noise[i].numpy()
            xn = S.ddpm_step(ac, eps.numpy(), int(t), x.numpy(), steps, N, noise=nz)
        x = torch.from_numpy(xn)
        if trace is not None:
            trace.append((int(t), eps.clone(), x.clone()))
    return x


@torch.no_grad()
def train_forward(Pu, ucfg, Pv, vcfg, pixel_values, masked_images, masks, ctx, timesteps,
                  noise, enc_noise, enc_noise_masked, ac=None, prediction_type="epsilon"):
    """Loss of one training step (train_diffute_v1.py:875-918), all randomness injected."""
    if ac is None:
        ac = S.make_tables()[2]
    sf = vcfg["scaling_factor"]
    lat = gaussian_sample(vae_encode_moments(Pv, vcfg, pixel_values), enc_noise) * sf
    m = mask_to_latent(masks, 2 ** (len(vcfg["block_out_channels"]) - 1))
    mlat = gaussian_sample(vae_encode_moments(Pv, vcfg, masked_images), enc_noise_masked) * sf
    noisy = torch.from_numpy(S.add_noise(ac, lat.numpy(), noise.numpy(), timesteps.numpy()))
    if prediction_type == "epsilon":
        target = noise
    elif prediction_type == "v_prediction":
        target = torch.from_numpy(S.get_velocity(ac, lat.numpy(), noise.numpy(), timesteps.numpy()))
    else:
        raise ValueError(f"Unknown prediction type {prediction_type}")
    pred = unet_forward(Pu, ucfg, torch.cat([noisy, m, mlat], dim=1), timesteps, ctx)
    return torch.mean((pred.float() - target.float()) ** 2), pred


def unet_train_grads(P, cfg, inp, timesteps, ctx, target, emulate_bf16=False):
    """Loss and parameter gradients of one denoiser training step (train_diffute_v1.py:913-925:
    `model_pred = unet(...).sample; loss = F.mse_loss(model_pred.float(), target.float()); accelerator.backward(loss)`),
    by torch autograd over the restatement.  With emulate_bf16 the forward rounds where the HIP path materialises bf16
    (the casts are straight-through), the backward arithmetic stays fp32.  Returns (loss, pred, {name: grad})."""
    Pg = {k: v.detach().clone().to(torch.float32).requires_grad_(True) for k, v in P.items()}
    with torch.enable_grad():
        pred = unet_forward.__wrapped__(Pg, cfg, inp, timesteps, ctx, emulate_bf16=emulate_bf16)
        loss = torch.mean((pred.float() - target.float()) ** 2)
        loss.backward()
    return float(loss.detach()), pred.detach(), {k: p.grad for k, p in Pg.items()}


def vae_train_grads(P, cfg, x, target, emulate_bf16=False):
    """Loss and parameter gradients of one autoencoder training step (train_vae.py:716-736: `pred = vae(x)["sample"]` =
    decode(encode(x).latent_dist.mode()); `loss = F.mse_loss(pred.float(), target.float())`; backward), by torch autograd
    over the restatement.  Returns (loss, recon, {name: grad})."""
    from .vae import gaussian_mode, vae_decode, vae_encode_moments
    Pg = {k: v.detach().clone().to(torch.float32).requires_grad_(True) for k, v in P.items()}
    with torch.enable_grad():
        z = gaussian_mode(vae_encode_moments.__wrapped__(Pg, cfg, x, emulate_bf16=emulate_bf16))
        recon = vae_decode.__wrapped__(Pg, cfg, z, emulate_bf16=emulate_bf16)
        loss = torch.mean((recon.float() - target.float()) ** 2)
        loss.backward()
    return float(loss.detach()), recon.detach(), {k: p.grad for k, p in Pg.items()}


def initial_latents(shape, seed=0):
    """P2 (app.ipynb:796-801): `randn_tensor(shape, generator=torch.manual_seed(0), dtype=fp32)` - a CPU draw (the
    generator is a CPU generator, so diffusers' randn_tensor creates the tensor on the CPU and moves it afterwards)."""
    return torch.randn(tuple(shape), generator=torch.manual_seed(seed), dtype=torch.float32)


@torch.no_grad()
def edit_latents(Pu, ucfg, Pv, vcfg, masked_image, mask, ctx, steps, enc_noise, scheduler="ddim", noise=None,
                 emulate_bf16=False, init=None):
    """The model part of text_editing() (app.ipynb:779-819) with the one device-RNG draw injected (`enc_noise` stands for
    latent_dist.sample()'s randn, :793): masked-image latents = vae.encode(masked).sample() * sf, mask -> latent grid
    (nearest), initial latents = seed-0 CPU randn * init_noise_sigma (=1), denoise loop, vae.decode(latents / sf)."""
    from .vae import vae_decode
    sf = vcfg["scaling_factor"]
    f = 2 ** (len(vcfg["block_out_channels"]) - 1)
    m = mask_to_latent(mask, f)
    mlat = gaussian_sample(vae_encode_moments(Pv, vcfg, masked_image, emulate_bf16=emulate_bf16), enc_noise) * sf
    B, _, H, W = masked_image.shape
    lat0 = initial_latents((B, vcfg["latent_channels"], H // f, W // f)) if init is None else init
    lat = denoise(Pu, ucfg, lat0, m, mlat, ctx, steps, scheduler, noise=noise, emulate_bf16=emulate_bf16)
    return vae_decode(Pv, vcfg, lat / sf, emulate_bf16=emulate_bf16), lat, mlat
