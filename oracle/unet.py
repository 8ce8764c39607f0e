"""Oracle: UNet2DConditionModel forward (SD2-inpainting layout), torch-CPU fp32.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates the public diffusers
>=0.15 module that the reference calls at app.ipynb:814 and
train_diffute_v1.py:913 (`unet(x, t, ctx).sample`), as specified in SURVEY.md
Appendix A.1.  Parameter names follow the diffusers state-dict keys that
`UNet2DConditionModel.from_pretrained(..., subfolder="unet")`
(train_diffute_v1.py:633-635, app.ipynb:551-553) loads.

`emulate_bf16=True` rounds weights and every tensor the HIP path materialises in
HBM to bf16 at the same points (fp32 math in between), so the HIP bf16 path can be
compared at tight tolerance.  With emulate_bf16=False it is the plain fp32 model.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import prng

SD2_INPAINT_UNET = dict(
    in_channels=9, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
    layers_per_block=2, attention_head_dim=(5, 10, 20, 20),  # = head COUNTS
    cross_attention_dim=1024, norm_num_groups=32,
    down_has_attn=(True, True, True, False), up_has_attn=(False, True, True, True),
)

TINY_UNET = dict(
    in_channels=9, out_channels=4, block_out_channels=(64, 128, 256, 256),
    layers_per_block=2, attention_head_dim=(1, 2, 4, 4),
    cross_attention_dim=128, norm_num_groups=32,
    down_has_attn=(True, True, True, False), up_has_attn=(False, True, True, True),
)


# ----------------------------------------------------------------------------- spec
def _resnet_spec(spec, p, cin, cout, temb_ch):
    spec[p + "norm1.weight"] = (cin,); spec[p + "norm1.bias"] = (cin,)
    spec[p + "conv1.weight"] = (cout, cin, 3, 3); spec[p + "conv1.bias"] = (cout,)
    if temb_ch:
        spec[p + "time_emb_proj.weight"] = (cout, temb_ch); spec[p + "time_emb_proj.bias"] = (cout,)
    spec[p + "norm2.weight"] = (cout,); spec[p + "norm2.bias"] = (cout,)
    spec[p + "conv2.weight"] = (cout, cout, 3, 3); spec[p + "conv2.bias"] = (cout,)
    if cin != cout:
        spec[p + "conv_shortcut.weight"] = (cout, cin, 1, 1); spec[p + "conv_shortcut.bias"] = (cout,)


def _xformer_spec(spec, p, c, ctx_dim):
    spec[p + "norm.weight"] = (c,); spec[p + "norm.bias"] = (c,)
    spec[p + "proj_in.weight"] = (c, c); spec[p + "proj_in.bias"] = (c,)
    t = p + "transformer_blocks.0."
    for i, kv in ((1, c), (2, ctx_dim)):
        spec[t + f"norm{i}.weight"] = (c,); spec[t + f"norm{i}.bias"] = (c,)
        spec[t + f"attn{i}.to_q.weight"] = (c, c)
        spec[t + f"attn{i}.to_k.weight"] = (c, kv)
        spec[t + f"attn{i}.to_v.weight"] = (c, kv)
        spec[t + f"attn{i}.to_out.0.weight"] = (c, c); spec[t + f"attn{i}.to_out.0.bias"] = (c,)
    spec[t + "norm3.weight"] = (c,); spec[t + "norm3.bias"] = (c,)
    spec[t + "ff.net.0.proj.weight"] = (8 * c, c); spec[t + "ff.net.0.proj.bias"] = (8 * c,)
    spec[t + "ff.net.2.weight"] = (c, 4 * c); spec[t + "ff.net.2.bias"] = (c,)
    spec[p + "proj_out.weight"] = (c, c); spec[p + "proj_out.bias"] = (c,)


def unet_param_spec(cfg):
    """Ordered {state-dict key: shape}."""
    boc = cfg["block_out_channels"]; L = cfg["layers_per_block"]; ctx = cfg["cross_attention_dim"]
    temb = boc[0] * 4
    spec = OrderedDict()
    spec["time_embedding.linear_1.weight"] = (temb, boc[0]); spec["time_embedding.linear_1.bias"] = (temb,)
    spec["time_embedding.linear_2.weight"] = (temb, temb); spec["time_embedding.linear_2.bias"] = (temb,)
    spec["conv_in.weight"] = (boc[0], cfg["in_channels"], 3, 3); spec["conv_in.bias"] = (boc[0],)
    skips = [boc[0]]
    cprev = boc[0]
    for i, c in enumerate(boc):
        for j in range(L):
            _resnet_spec(spec, f"down_blocks.{i}.resnets.{j}.", cprev, c, temb)
            if cfg["down_has_attn"][i]:
                _xformer_spec(spec, f"down_blocks.{i}.attentions.{j}.", c, ctx)
            cprev = c; skips.append(c)
        if i < len(boc) - 1:
            spec[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (c, c, 3, 3)
            spec[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (c,)
            skips.append(c)
    cm = boc[-1]
    _resnet_spec(spec, "mid_block.resnets.0.", cm, cm, temb)
    _xformer_spec(spec, "mid_block.attentions.0.", cm, ctx)
    _resnet_spec(spec, "mid_block.resnets.1.", cm, cm, temb)
    rev = list(reversed(boc))
    for i, c in enumerate(rev):
        for j in range(L + 1):
            cs = skips.pop()
            _resnet_spec(spec, f"up_blocks.{i}.resnets.{j}.", cprev + cs, c, temb)
            if cfg["up_has_attn"][i]:
                _xformer_spec(spec, f"up_blocks.{i}.attentions.{j}.", c, ctx)
            cprev = c
        if i < len(boc) - 1:
            spec[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3)
            spec[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
    spec["conv_norm_out.weight"] = (boc[0],); spec["conv_norm_out.bias"] = (boc[0],)
    spec["conv_out.weight"] = (cfg["out_channels"], boc[0], 3, 3); spec["conv_out.bias"] = (cfg["out_channels"],)
    assert not skips
    return spec


def count_params(spec):
    return int(sum(int(np.prod(s)) for s in spec.values()))


def make_params(spec, seed=1234, dtype=torch.float32):
    """Seeded synthetic weights: conv/linear ~ U(-a,a), a=sqrt(3/fan_in);
    norm gamma = 1+0.1u, beta = 0.1u, other biases 0.05u (u ~ U(-1,1))."""
    out = OrderedDict()
    for name, shape in spec.items():
        n = int(np.prod(shape))
        u = prng.uniform01(seed, prng.tensor_id(name), n) * np.float32(2.0) - np.float32(1.0)
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            v = u * np.float32(math.sqrt(3.0 / fan_in))
        elif name.endswith("weight"):      # norm gamma
            v = np.float32(1.0) + np.float32(0.1) * u
        elif ("norm" in name.rsplit(".", 2)[-2]):  # norm beta
            v = np.float32(0.1) * u
        else:
            v = np.float32(0.05) * u
        out[name] = torch.from_numpy(v.astype(np.float32).reshape(shape)).to(dtype)
    return out


# ----------------------------------------------------------------------------- math
def _q(x, on):
    """Round to the storage type and back (HBM materialisation point of the HIP path): `on` is False (plain fp32),
    True / "bf16" (the bf16 build) or "fp16" (the fp16 build selected by `.to(dtype=torch.float16)`,
    train_diffute_v1.py:789-797, BASELINE configs[4])."""
    if not on:
        return x
    return x.to(torch.float16 if on == "fp16" else torch.bfloat16).to(torch.float32)


def timestep_embedding(t, dim):
    """flip_sin_to_cos=True, freq_shift=0 (Appendix A.1 step 1): cat([cos, sin])."""
    half = dim // 2
    freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    arg = t[:, None].to(torch.float32) * freq[None, :]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


def _gn(x, P, p, groups, eps, silu, em):
    y = F.group_norm(x, groups, P[p + "weight"], P[p + "bias"], eps)
    if silu:
        y = F.silu(y)
    return _q(y, em)


def _conv(x, P, p, em, stride=1, padding=1):
    return F.conv2d(x, _q(P[p + "weight"], em), P[p + "bias"], stride=stride, padding=padding)


def resnet(x, emb_act, P, p, groups, eps, em):
    """ResnetBlock2D (Appendix A.1). emb_act = SiLU(emb) or None (VAE)."""
    h = _gn(x, P, p + "norm1.", groups, eps, True, em)
    h = _conv(h, P, p + "conv1.", em)
    if emb_act is not None:
        tp = F.linear(emb_act, _q(P[p + "time_emb_proj.weight"], em), P[p + "time_emb_proj.bias"])
        h = h + tp[:, :, None, None]
    h = _q(h, em)
    h = _gn(h, P, p + "norm2.", groups, eps, True, em)
    h = _conv(h, P, p + "conv2.", em)
    if (p + "conv_shortcut.weight") in P:
        x = F.conv2d(x, _q(P[p + "conv_shortcut.weight"], em), P[p + "conv_shortcut.bias"])
    return _q(x + h, em)


def _attention(xq, src, P, p, heads, em):
    B, S, C = xq.shape
    d = C // heads
    q = _q(F.linear(xq, _q(P[p + "to_q.weight"], em)), em)
    k = _q(F.linear(src, _q(P[p + "to_k.weight"], em)), em)
    v = _q(F.linear(src, _q(P[p + "to_v.weight"], em)), em)
    q = q.view(B, S, heads, d).transpose(1, 2)
    k = k.view(B, -1, heads, d).transpose(1, 2)
    v = v.view(B, -1, heads, d).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
    a = torch.matmul(torch.softmax(s, dim=-1), v)
    a = _q(a.transpose(1, 2).reshape(B, S, C), em)
    return F.linear(a, _q(P[p + "to_out.0.weight"], em), P[p + "to_out.0.bias"])


def transformer2d(x, ctx, P, p, heads, groups, em):
    """Transformer2DModel, use_linear_projection=True, one BasicTransformerBlock."""
    B, C, H, W = x.shape
    r = x
    h = _gn(x, P, p + "norm.", groups, 1e-6, False, em)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    h = _q(F.linear(h, _q(P[p + "proj_in.weight"], em), P[p + "proj_in.bias"]), em)
    t = p + "transformer_blocks.0."
    n = _q(F.layer_norm(h, (C,), P[t + "norm1.weight"], P[t + "norm1.bias"], 1e-5), em)
    h = _q(h + _attention(n, n, P, t + "attn1.", heads, em), em)
    n = _q(F.layer_norm(h, (C,), P[t + "norm2.weight"], P[t + "norm2.bias"], 1e-5), em)
    h = _q(h + _attention(n, ctx, P, t + "attn2.", heads, em), em)
    n = _q(F.layer_norm(h, (C,), P[t + "norm3.weight"], P[t + "norm3.bias"], 1e-5), em)
    g = F.linear(n, _q(P[t + "ff.net.0.proj.weight"], em), P[t + "ff.net.0.proj.bias"])
    a, b = g.chunk(2, dim=-1)
    g = _q(a * F.gelu(b), em)                      # exact (erf) GELU
    h = _q(h + F.linear(g, _q(P[t + "ff.net.2.weight"], em), P[t + "ff.net.2.bias"]), em)
    h = F.linear(h, _q(P[p + "proj_out.weight"], em), P[p + "proj_out.bias"])
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return _q(h + r, em)


@torch.no_grad()
def unet_forward(P, cfg, sample, timestep, ctx, emulate_bf16=False, taps=None):
    """eps = unet(sample[B,9,h,w], timestep (0-d or [B] int), ctx[B,577,1024]).  NCHW fp32."""
    em = emulate_bf16
    boc = cfg["block_out_channels"]; L = cfg["layers_per_block"]
    heads = cfg["attention_head_dim"]; G = cfg["norm_num_groups"]
    B = sample.shape[0]
    t = torch.as_tensor(timestep, dtype=torch.int64).reshape(-1)
    if t.numel() == 1:
        t = t.expand(B)
    sample = _q(sample.to(torch.float32), em)
    ctx = _q(ctx.to(torch.float32), em)
    temb = timestep_embedding(t, boc[0])
    emb = F.linear(temb, _q(P["time_embedding.linear_1.weight"], em), P["time_embedding.linear_1.bias"])
    emb = F.linear(F.silu(emb), _q(P["time_embedding.linear_2.weight"], em), P["time_embedding.linear_2.bias"])
    emb_act = F.silu(emb)
    h = _q(_conv(sample, P, "conv_in.", em), em)
    if taps is not None: taps["conv_in"] = h
    skips = [h]
    for i, c in enumerate(boc):
        for j in range(L):
            h = resnet(h, emb_act, P, f"down_blocks.{i}.resnets.{j}.", G, 1e-5, em)
            if cfg["down_has_attn"][i]:
                h = transformer2d(h, ctx, P, f"down_blocks.{i}.attentions.{j}.", heads[i], G, em)
            skips.append(h)
        if i < len(boc) - 1:
            h = _q(_conv(h, P, f"down_blocks.{i}.downsamplers.0.conv.", em, stride=2, padding=1), em)
            skips.append(h)
        if taps is not None: taps[f"down{i}"] = h
    h = resnet(h, emb_act, P, "mid_block.resnets.0.", G, 1e-5, em)
    h = transformer2d(h, ctx, P, "mid_block.attentions.0.", heads[-1], G, em)
    h = resnet(h, emb_act, P, "mid_block.resnets.1.", G, 1e-5, em)
    if taps is not None: taps["mid"] = h
    rheads = list(reversed(heads))
    for i, c in enumerate(reversed(boc)):
        for j in range(L + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet(h, emb_act, P, f"up_blocks.{i}.resnets.{j}.", G, 1e-5, em)
            if cfg["up_has_attn"][i]:
                h = transformer2d(h, ctx, P, f"up_blocks.{i}.attentions.{j}.", rheads[i], G, em)
        if i < len(boc) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _q(_conv(h, P, f"up_blocks.{i}.upsamplers.0.conv.", em), em)
        if taps is not None: taps[f"up{i}"] = h
    h = _gn(h, P, "conv_norm_out.", G, 1e-5, True, em)
    return _conv(h, P, "conv_out.", em)        # fp32 eps (kept fp32 by the HIP path too)


def unet_flops(cfg, B, h, w, ctx_len=577, cached_ctx_kv=False):
    """Algorithmic FLOPs (2*MACs of conv/linear/attention matmuls), SURVEY.md Appendix C."""
    boc = cfg["block_out_channels"]; L = cfg["layers_per_block"]; ctxd = cfg["cross_attention_dim"]
    f = 0
    def conv(hw, cin, cout, k=3): return 2 * B * hw * cin * cout * k * k
    def lin(m, cin, cout): return 2 * m * cin * cout
    def res(hw, cin, cout):
        x = conv(hw, cin, cout) + conv(hw, cout, cout) + lin(B, boc[0] * 4, cout)
        if cin != cout: x += conv(hw, cin, cout, 1)
        return x
    def xf(hw, c):
        m = B * hw
        x = 2 * lin(m, c, c)                              # proj_in/out
        x += 4 * lin(m, c, c) + 4 * B * hw * hw * c        # self: q,k,v,out + core
        x += 2 * lin(m, c, c) + 4 * B * hw * ctx_len * c   # cross: q,out + core
        if not cached_ctx_kv: x += 2 * lin(B * ctx_len, ctxd, c)
        x += lin(m, c, 8 * c) + lin(m, 4 * c, c)
        return x
    f += lin(B, boc[0], boc[0] * 4) + lin(B, boc[0] * 4, boc[0] * 4)
    hw = h * w
    f += conv(hw, cfg["in_channels"], boc[0])
    skips = [(boc[0], hw)]; cprev = boc[0]
    for i, c in enumerate(boc):
        for j in range(L):
            f += res(hw, cprev, c)
            if cfg["down_has_attn"][i]: f += xf(hw, c)
            cprev = c; skips.append((c, hw))
        if i < len(boc) - 1:
            hw //= 4; f += conv(hw, c, c); skips.append((c, hw))
    f += 2 * res(hw, cprev, cprev) + xf(hw, cprev)
    for i, c in enumerate(reversed(boc)):
        for j in range(L + 1):
            cs, _ = skips.pop()
            f += res(hw, cprev + cs, c)
            if cfg["up_has_attn"][i]: f += xf(hw, c)
            cprev = c
        if i < len(boc) - 1:
            hw *= 4; f += conv(hw, c, c)
    f += conv(hw, boc[0], cfg["out_channels"])
    return f
