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
TEMB_TABLE = True      # denoise(): time-embedding projections of all steps in one batched pass (False: four small launches per step; A/B aid)


def _side_streams(device, n):
    key = str(device)
    pool = _SIDE.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool


def mask_to_latent(mask, vae_scale_factor=8):
    """F.interpolate(mask, size=(H/8, W/8)) with the default nearest mode (app.ipynb:787-791,
    train_diffute_v1.py:880-884): out[y, x] = in[floor(y*8), floor(x*8)]."""
    return mask[:, :, ::vae_scale_factor, ::vae_scale_factor].to(torch.float32).contiguous()


class _Run:
    """One micro-batch of the denoise loop: its own stream, UNet execution slot and fixed-address buffers."""

    def __init__(self, unet, scheduler, latents, mask, mlat, ctx, slot, stream, use_graph, temb_table=None):
        self.unet, self.sched, self.slot, self.stream, self.use_graph = unet, scheduler, slot, stream, use_graph
        self.temb_table = temb_table
        with torch.cuda.stream(stream):
            self.x = (latents.to(torch.float32) * scheduler.init_noise_sigma).contiguous()      # app.ipynb:800
            self.m = mask.to(torch.float32).contiguous()
            self.ml = mlat.to(torch.float32).contiguous()
            self.eps = torch.empty_like(self.x)
            self.t_cur = torch.empty(1, dtype=torch.int64, device=self.x.device)   # fixed address: the captured graph reads it
            self.step_idx = torch.zeros(1, dtype=torch.int32, device=self.x.device)   # likewise: the row of temb_table this step fetches
            unet.set_context(ctx, slot=slot)

    def step(self, i, t, ts_dev, coefs, noise, is_ddim, vpred):
        lib = _cabi.lib()
        x, eps = self.x, self.eps
        with torch.cuda.stream(self.stream):
            st = _cabi.current_stream()
            if self.temb_table is not None:
                self.step_idx.copy_(ts_dev[1][i:i + 1], non_blocking=True)
                self.unet.forward_parts([x, self.m, self.ml], self.t_cur, out=eps, graph=self.use_graph, slot=self.slot,
                                        temb=(self.temb_table, self.step_idx))
            else:
                self.t_cur.copy_(ts_dev[0][i:i + 1], non_blocking=True)
                self.unet.forward_parts([x, self.m, self.ml], self.t_cur, out=eps, graph=self.use_graph, slot=self.slot)
            # the update is elementwise, so prev_sample overwrites the sample in place (stable pointers for the graph)
            if is_ddim:
                sbt, sat, sap, dirc, std = coefs
                _cabi.check(lib.dmx_sched_step_ddim(_cabi.ptr(x), _cabi.ptr(eps), _cabi.ptr(noise), _cabi.ptr(x), x.numel(),
                                                    sbt, sat, sap, dirc, std, vpred, st), "sched_step_ddim")
            else:
                sbt, sat, c0, c1, sigma = coefs
                _cabi.check(lib.dmx_sched_step_ddpm(_cabi.ptr(x), _cabi.ptr(eps), _cabi.ptr(noise), _cabi.ptr(x), x.numel(),
                                                    sbt, sat, c0, c1, sigma, vpred, st), "sched_step_ddpm")


@torch.no_grad()
def denoise(unet, scheduler, latents, mask, masked_image_latents, encoder_hidden_states,
            num_inference_steps, variance_noise=None, eta=0.0, callback=None, use_graph=True, micro_batches=1):
    """latents/masked_image_latents [B,4,h,w], mask [B,1,h,w] (already at latent resolution), context
    [B,S,1024]; all on the GPU.  variance_noise: optional [steps,B,4,h,w] injected in place of the
    per-step device randn of DDPMScheduler.step (app.ipynb:816).  Returns the final latents (fp32).

    micro_batches=n splits the batch into n independent chains (images do not interact), each on its own stream
    with its own captured graph: one chain's kernels fill the CUs the other chain's small / draining kernels
    leave idle.  Results are identical to micro_batches=1 up to per-kernel tile-plan rounding."""
    _cabi.require_cuda(latents, mask, masked_image_latents, encoder_hidden_states)
    _cabi.poll_device_error()            # what a kernel of an EARLIER pass raised (no sync; include/diffute_hip.h dmx_device_error)
    unet._ensure_packed()
    scheduler.set_timesteps(int(num_inference_steps))
    ts_host = [int(t) for t in scheduler.timesteps]
    dev = latents.device
    ts_dev = scheduler.timesteps.to(device=dev, dtype=torch.int64).contiguous()
    is_ddim = isinstance(scheduler, DDIMScheduler)
    vpred = int(scheduler.config.prediction_type == "v_prediction")
    B = latents.shape[0]
    n = max(1, min(int(micro_batches), B))
    bounds = [(B * j // n, B * (j + 1) // n) for j in range(n)]
    main = torch.cuda.current_stream(dev)
    # the time-embedding MLP + every resnet's time_emb_proj depend on the timestep only: all steps' rows in one batched pass up
    # front (bit-identical rows), each step then fetches its row with one tiny launch instead of recomputing four small layers
    temb_table = unet.temb_table(ts_dev) if (ts_host and TEMB_TABLE) else None
    ts_dev = (ts_dev, torch.arange(len(ts_host), dtype=torch.int32, device=dev))
    streams = _side_streams(dev, n)             # the loop runs on side streams: graph capture needs a non-default stream
    # several chains at once share the CUs: plans whose blocks wait for co-resident peers are off while they are enqueued (the plans are
    # chosen - and baked into the captured graphs, which are keyed on the setting - at enqueue time)
    lib_ = unet._lib
    old_exclusive = lib_.dmx_set_exclusive_device(0) if n > 1 else None
    try:
        return _denoise_enqueue(unet, scheduler, bounds, streams, main, n, latents, mask, masked_image_latents, encoder_hidden_states,
                                ts_host, ts_dev, temb_table, is_ddim, vpred, eta, variance_noise, callback, use_graph)
    finally:
        if old_exclusive is not None:
            lib_.dmx_set_exclusive_device(old_exclusive)


def _denoise_enqueue(unet, scheduler, bounds, streams, main, n, latents, mask, masked_image_latents, encoder_hidden_states,
                     ts_host, ts_dev, temb_table, is_ddim, vpred, eta, variance_noise, callback, use_graph):
    runs = []
    for j, (lo, hi) in enumerate(bounds):
        streams[j].wait_stream(main)
        runs.append(_Run(unet, scheduler, latents[lo:hi], mask[lo:hi], masked_image_latents[lo:hi],
                         encoder_hidden_states[lo:hi].contiguous(), j, streams[j], use_graph, temb_table))
    for i, t in enumerate(ts_host):
        coefs = scheduler.step_coefficients(t, eta) if is_ddim else scheduler.step_coefficients(t)
        need_noise = (eta > 0) if is_ddim else (t > 0)
        for j, (lo, hi) in enumerate(bounds):
            nz = None
            if need_noise:
                with torch.cuda.stream(streams[j]):
                    nz = (variance_noise[i][lo:hi] if variance_noise is not None else torch.randn_like(runs[j].x)).to(torch.float32).contiguous()
            runs[j].step(i, t, ts_dev, coefs, nz, is_ddim, vpred)
        if callback is not None:
            for s_ in streams[:n]:
                main.wait_stream(s_)
            callback(i, t, runs[0].x if n == 1 else torch.cat([r.x for r in runs], 0),
                     runs[0].eps if n == 1 else torch.cat([r.eps for r in runs], 0))
            for s_ in streams[:n]:
                s_.wait_stream(main)            # the callback's reads finish before the next step overwrites x / eps
    for s_ in streams[:n]:
        main.wait_stream(s_)
    return runs[0].x if n == 1 else torch.cat([r.x for r in runs], 0)


@torch.no_grad()
def edit_latents(unet, vae, scheduler, image, masked_image, mask, encoder_hidden_states, num_inference_steps,
                 init_latents=None, generator=None, enc_noise=None, variance_noise=None):
    """The model part of text_editing() (app.ipynb:779-819): VAE-encode the masked crop, downsample the
    mask, denoise from seeded noise, VAE-decode.  `image` is unused by the arithmetic (the reference's encode of it,
    app.ipynb:781, is dead code: its result is overwritten at :798); enc_noise / variance_noise inject the two device-RNG
    draws (latent_dist.sample(), DDPMScheduler.step) for tests.  Crop / resize / paste: diffute_amd.prepost."""
    sf = vae.config.scaling_factor
    f = 2 ** (len(vae.config.block_out_channels) - 1)
    m = mask_to_latent(mask, f)
    dist = vae.encode(masked_image).latent_dist
    mlat = (dist.sample(noise=enc_noise) if enc_noise is not None else dist.sample(generator=generator)) * sf   # app.ipynb:793-794
    B, _, H, W = masked_image.shape
    if init_latents is None:
        init_latents = torch.randn((B, vae.config.latent_channels, H // f, W // f),
                                   generator=torch.manual_seed(0), dtype=torch.float32).to(masked_image.device)  # :798
    lat = denoise(unet, scheduler, init_latents, m, mlat, encoder_hidden_states, num_inference_steps, variance_noise=variance_noise)
    return vae.decode(lat / sf).sample                                                    # app.ipynb:818-819
