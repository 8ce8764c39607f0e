"""Oracle: AutoencoderKL encode / decode (SD VAE layout), torch-CPU fp32.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates the public diffusers
module behind `vae.encode(x).latent_dist.sample()` (app.ipynb:781,793;
train_diffute_v1.py:875,886), `vae.decode(z).sample` (app.ipynb:819) and
`vae(x)["sample"]` (train_vae.py:721-722) as specified in SURVEY.md Appendix A.2.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from .unet import _resnet_spec, _q, _gn, _conv, resnet

SD_VAE = dict(in_channels=3, out_channels=3, latent_channels=4,
              block_out_channels=(128, 256, 512, 512), layers_per_block=2,
              norm_num_groups=32, scaling_factor=0.18215)

TINY_VAE = dict(in_channels=3, out_channels=3, latent_channels=4,
                block_out_channels=(64, 128, 128, 128), layers_per_block=1,
                norm_num_groups=32, scaling_factor=0.18215)


def _attn_spec(spec, p, c):
    spec[p + "group_norm.weight"] = (c,); spec[p + "group_norm.bias"] = (c,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        spec[p + n + ".weight"] = (c, c); spec[p + n + ".bias"] = (c,)


def vae_param_spec(cfg):
    boc = cfg["block_out_channels"]; L = cfg["layers_per_block"]; lc = cfg["latent_channels"]
    spec = OrderedDict()
    e = "encoder."
    spec[e + "conv_in.weight"] = (boc[0], cfg["in_channels"], 3, 3); spec[e + "conv_in.bias"] = (boc[0],)
    cprev = boc[0]
    for i, c in enumerate(boc):
        for j in range(L):
            _resnet_spec(spec, e + f"down_blocks.{i}.resnets.{j}.", cprev, c, 0); cprev = c
        if i < len(boc) - 1:
            spec[e + f"down_blocks.{i}.downsamplers.0.conv.weight"] = (c, c, 3, 3)
            spec[e + f"down_blocks.{i}.downsamplers.0.conv.bias"] = (c,)
    _resnet_spec(spec, e + "mid_block.resnets.0.", cprev, cprev, 0)
    _attn_spec(spec, e + "mid_block.attentions.0.", cprev)
    _resnet_spec(spec, e + "mid_block.resnets.1.", cprev, cprev, 0)
    spec[e + "conv_norm_out.weight"] = (cprev,); spec[e + "conv_norm_out.bias"] = (cprev,)
    spec[e + "conv_out.weight"] = (2 * lc, cprev, 3, 3); spec[e + "conv_out.bias"] = (2 * lc,)
    spec["quant_conv.weight"] = (2 * lc, 2 * lc, 1, 1); spec["quant_conv.bias"] = (2 * lc,)
    spec["post_quant_conv.weight"] = (lc, lc, 1, 1); spec["post_quant_conv.bias"] = (lc,)
    d = "decoder."
    cprev = boc[-1]
    spec[d + "conv_in.weight"] = (cprev, lc, 3, 3); spec[d + "conv_in.bias"] = (cprev,)
    _resnet_spec(spec, d + "mid_block.resnets.0.", cprev, cprev, 0)
    _attn_spec(spec, d + "mid_block.attentions.0.", cprev)
    _resnet_spec(spec, d + "mid_block.resnets.1.", cprev, cprev, 0)
    for i, c in enumerate(reversed(boc)):
        for j in range(L + 1):
            _resnet_spec(spec, d + f"up_blocks.{i}.resnets.{j}.", cprev, c, 0); cprev = c
        if i < len(boc) - 1:
            spec[d + f"up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3)
            spec[d + f"up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
    spec[d + "conv_norm_out.weight"] = (cprev,); spec[d + "conv_norm_out.bias"] = (cprev,)
    spec[d + "conv_out.weight"] = (cfg["out_channels"], cprev, 3, 3); spec[d + "conv_out.bias"] = (cfg["out_channels"],)
    return spec


def _mid_attn(x, P, p, groups, em):
    """Single-head attention over H*W tokens, d=C, with bias on q/k/v/out (Appendix A.2)."""
    B, C, H, W = x.shape
    r = x
    h = _gn(x, P, p + "group_norm.", groups, 1e-6, False, em)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    q = _q(F.linear(h, _q(P[p + "to_q.weight"], em), P[p + "to_q.bias"]), em)
    k = _q(F.linear(h, _q(P[p + "to_k.weight"], em), P[p + "to_k.bias"]), em)
    v = _q(F.linear(h, _q(P[p + "to_v.weight"], em), P[p + "to_v.bias"]), em)
    s = torch.matmul(q, k.transpose(-1, -2)) * (C ** -0.5)
    a = _q(torch.matmul(torch.softmax(s, dim=-1), v), em)
    a = F.linear(a, _q(P[p + "to_out.0.weight"], em), P[p + "to_out.0.bias"])
    a = a.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return _q(a + r, em)


@torch.no_grad()
def vae_encode_moments(P, cfg, x, emulate_bf16=False):
    """x [B,3,H,W] -> moments [B,8,H/8,W/8] (mean | logvar), fp32."""
    em = emulate_bf16
    boc = cfg["block_out_channels"]; L = cfg["layers_per_block"]; G = cfg["norm_num_groups"]
    e = "encoder."
    h = _q(_conv(_q(x.to(torch.float32), em), P, e + "conv_in.", em), em)
    for i, c in enumerate(boc):
        for j in range(L):
            h = resnet(h, None, P, e + f"down_blocks.{i}.resnets.{j}.", G, 1e-6, em)
        if i < len(boc) - 1:
            h = F.pad(h, (0, 1, 0, 1))                       # asymmetric pad, then s2 p0
            h = _q(_conv(h, P, e + f"down_blocks.{i}.downsamplers.0.conv.", em, stride=2, padding=0), em)
    h = resnet(h, None, P, e + "mid_block.resnets.0.", G, 1e-6, em)
    h = _mid_attn(h, P, e + "mid_block.attentions.0.", G, em)
    h = resnet(h, None, P, e + "mid_block.resnets.1.", G, 1e-6, em)
    h = _gn(h, P, e + "conv_norm_out.", G, 1e-6, True, em)
    h = _q(_conv(h, P, e + "conv_out.", em), em)
    return F.conv2d(h, _q(P["quant_conv.weight"], em), P["quant_conv.bias"])


def gaussian_sample(moments, noise):
    """DiagonalGaussianDistribution.sample() with the randn injected (Appendix A.2)."""
    mean, logvar = moments.chunk(2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    std = torch.exp(0.5 * logvar)
    return mean + std * noise


def gaussian_mode(moments):
    return moments.chunk(2, dim=1)[0]


@torch.no_grad()
def vae_decode(P, cfg, z, emulate_bf16=False):
    """z [B,4,h,w] -> image [B,3,8h,8w], fp32."""
    em = emulate_bf16
    boc = cfg["block_out_channels"]; L = cfg["layers_per_block"]; G = cfg["norm_num_groups"]
    d = "decoder."
    z = _q(F.conv2d(_q(z.to(torch.float32), em), _q(P["post_quant_conv.weight"], em), P["post_quant_conv.bias"]), em)
    h = _q(_conv(z, P, d + "conv_in.", em), em)
    h = resnet(h, None, P, d + "mid_block.resnets.0.", G, 1e-6, em)
    h = _mid_attn(h, P, d + "mid_block.attentions.0.", G, em)
    h = resnet(h, None, P, d + "mid_block.resnets.1.", G, 1e-6, em)
    for i, c in enumerate(reversed(boc)):
        for j in range(L + 1):
            h = resnet(h, None, P, d + f"up_blocks.{i}.resnets.{j}.", G, 1e-6, em)
        if i < len(boc) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _q(_conv(h, P, d + f"up_blocks.{i}.upsamplers.0.conv.", em), em)
    h = _gn(h, P, d + "conv_norm_out.", G, 1e-6, True, em)
    return _conv(h, P, d + "conv_out.", em)


def vae_flops(cfg, B, H, W):
    """(encode, decode) algorithmic FLOPs."""
    boc = cfg["block_out_channels"]; L = cfg["layers_per_block"]; lc = cfg["latent_channels"]
    def conv(hw, cin, cout, k=3): return 2 * B * hw * cin * cout * k * k
    def res(hw, cin, cout):
        return conv(hw, cin, cout) + conv(hw, cout, cout) + (conv(hw, cin, cout, 1) if cin != cout else 0)
    def attn(hw, c): return 4 * 2 * B * hw * c * c + 4 * B * hw * hw * c
    hw = H * W
    e = conv(hw, cfg["in_channels"], boc[0]); cprev = boc[0]
    for i, c in enumerate(boc):
        for j in range(L):
            e += res(hw, cprev, c); cprev = c
        if i < len(boc) - 1:
            hw //= 4; e += conv(hw, c, c)
    e += 2 * res(hw, cprev, cprev) + attn(hw, cprev) + conv(hw, cprev, 2 * lc) + conv(hw, 2 * lc, 2 * lc, 1)
    d = conv(hw, lc, lc, 1) + conv(hw, lc, cprev) + 2 * res(hw, cprev, cprev) + attn(hw, cprev)
    for i, c in enumerate(reversed(boc)):
        for j in range(L + 1):
            d += res(hw, cprev, c); cprev = c
        if i < len(boc) - 1:
            hw *= 4; d += conv(hw, c, c)
    d += conv(hw, cprev, cfg["out_channels"])
    return e, d
