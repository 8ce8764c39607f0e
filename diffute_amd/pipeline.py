"""The reference's denoise loop (app.ipynb:796-816) driven through the C-ABI.

`denoise()` is the hot path BASELINE.json names: per step one `dmx_unet_forward` (which fuses
torch.cat([latents, mask, masked_image_latents], 1)) and one scheduler-step kernel.  The glyph
context K/V are projected once per image, timesteps live on the device, nothing synchronises
with the host inside the loop.
"""
import torch

from . import _cabi
from .schedulers import DDIMScheduler, DDPMScheduler


_SIDE = {}


def _side_stream(device):
    key = str(device)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


def mask_to_latent(mask, vae_scale_factor=8):
    """F.interpolate(mask, size=(H/8, W/8)) with the default nearest mode (app.ipynb:787-791,
    train_diffute_v1.py:880-884): out[y, x] = in[floor(y*8), floor(x*8)]."""
    return mask[:, :, ::vae_scale_factor, ::vae_scale_factor].to(torch.float32).contiguous()


@torch.no_grad()
def denoise(unet, scheduler, latents, mask, masked_image_latents, encoder_hidden_states,
            num_inference_steps, variance_noise=None, eta=0.0, callback=None, use_graph=True):
    """latents/masked_image_latents [B,4,h,w], mask [B,1,h,w] (already at latent resolution), context
    [B,S,1024]; all on the GPU.  variance_noise: optional [steps,B,4,h,w] injected in place of the
    per-step device randn of DDPMScheduler.step (app.ipynb:816).  Returns the final latents (fp32)."""
    _cabi.require_cuda(latents, mask, masked_image_latents, encoder_hidden_states)
    lib = _cabi.lib()
    unet._ensure_packed()
    unet.set_context(encoder_hidden_states)
    scheduler.set_timesteps(int(num_inference_steps))
    ts_host = [int(t) for t in scheduler.timesteps]
    ts_dev = scheduler.timesteps.to(device=latents.device, dtype=torch.int64).contiguous()
    x = (latents.to(torch.float32) * scheduler.init_noise_sigma).contiguous()      # app.ipynb:800
    m = mask.to(torch.float32).contiguous()
    ml = masked_image_latents.to(torch.float32).contiguous()
    eps = torch.empty_like(x)
    t_cur = torch.empty(1, dtype=torch.int64, device=x.device)        # fixed address: the captured graph reads it
    is_ddim = isinstance(scheduler, DDIMScheduler)
    vpred = int(scheduler.config.prediction_type == "v_prediction")
    # the step loop runs on a side stream: hipGraph capture / replay needs a non-default stream
    main = torch.cuda.current_stream(x.device)
    side = _side_stream(x.device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        st = _cabi.current_stream()
        for i, t in enumerate(ts_host):
            t_cur.copy_(ts_dev[i:i + 1], non_blocking=True)
            unet.forward_parts([x, m, ml], t_cur, out=eps, graph=use_graph)
            # the update is elementwise, so prev_sample overwrites the sample in place (stable pointers for the graph)
            if is_ddim:
                sbt, sat, sap, dirc, std = scheduler.step_coefficients(t, eta)
                nz = None
                if eta > 0:
                    nz = (variance_noise[i] if variance_noise is not None else torch.randn_like(x)).to(torch.float32).contiguous()
                _cabi.check(lib.dmx_sched_step_ddim(_cabi.ptr(x), _cabi.ptr(eps), _cabi.ptr(nz), _cabi.ptr(x), x.numel(),
                                                    sbt, sat, sap, dirc, std, vpred, st), "sched_step_ddim")
            else:
                sbt, sat, c0, c1, sigma = scheduler.step_coefficients(t)
                nz = None
                if t > 0:
                    nz = (variance_noise[i] if variance_noise is not None else torch.randn_like(x)).to(torch.float32).contiguous()
                _cabi.check(lib.dmx_sched_step_ddpm(_cabi.ptr(x), _cabi.ptr(eps), _cabi.ptr(nz), _cabi.ptr(x), x.numel(),
                                                    sbt, sat, c0, c1, sigma, vpred, st), "sched_step_ddpm")
            if callback is not None:
                callback(i, t, x, eps)
    main.wait_stream(side)
    return x


@torch.no_grad()
def edit_latents(unet, vae, scheduler, image, masked_image, mask, encoder_hidden_states, num_inference_steps,
                 init_latents=None, generator=None):
    """The model part of text_editing() (app.ipynb:779-819): VAE-encode the masked crop, downsample the
    mask, denoise from seeded noise, VAE-decode.  Pre/post-processing (crop, resize, paste) is out of scope."""
    sf = vae.config.scaling_factor
    f = 2 ** (len(vae.config.block_out_channels) - 1)
    m = mask_to_latent(mask, f)
    mlat = vae.encode(masked_image).latent_dist.sample(generator=generator) * sf          # app.ipynb:793-794
    B, _, H, W = masked_image.shape
    if init_latents is None:
        init_latents = torch.randn((B, vae.config.latent_channels, H // f, W // f),
                                   generator=torch.manual_seed(0), dtype=torch.float32).to(masked_image.device)  # :798
    lat = denoise(unet, scheduler, init_latents, m, mlat, encoder_hidden_states, num_inference_steps)
    return vae.decode(lat / sf).sample                                                    # app.ipynb:818-819
