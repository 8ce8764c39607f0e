"""Synthetic masked-text-crop inputs (no datasets or checkpoints exist offline).

Same counter-PRNG streams as scripts/make_golden.py, so GPU runs can be compared with the committed
golden vectors: N(0,1) initial latents (stands in for randn(generator=manual_seed(0)), app.ipynb:798),
a rectangular text-line mask at latent resolution, 0.18215*N(0,1) masked-image latents
(vae.encode(masked).sample()*scaling_factor, app.ipynb:793-794) and an N(0,1) glyph context
[B,577,1024] (TrOCR encoder output, app.ipynb:775-776).
"""
import torch

from .init import normal, uniform01


def synth_inputs(B, h, w, ctx_len=577, ctx_dim=1024, seed=0, device="cpu"):
    lat = normal(seed + 0, 11, B * 4 * h * w, device).reshape(B, 4, h, w)
    mask = torch.zeros(B, 1, h, w, device=device)
    mask[:, :, (3 * h) // 8:(5 * h) // 8, w // 8:(7 * w) // 8] = 1.0
    mlat = normal(seed + 1, 12, B * 4 * h * w, device).reshape(B, 4, h, w) * 0.18215
    ctx = normal(seed + 2, 13, B * ctx_len * ctx_dim, device).reshape(B, ctx_len, ctx_dim)
    return lat, mask, mlat, ctx


def synth_images(B, H, W, seed=5, device="cpu"):
    """U(-1,1) images with a white band and dark glyph-like strokes ("synthetic masked-text crops")."""
    img = uniform01(seed, 21, B * 3 * H * W, device).reshape(B, 3, H, W) * 2 - 1
    return img


def text_crop_images(B, H, W, seed=5, device="cpu"):
    img = synth_images(B, H, W, seed, device) * 0.25
    img[:, :, (3 * H) // 8:(5 * H) // 8, W // 8:(7 * W) // 8] = 0.9            # white text band
    for k in range(6):                                                         # dark vertical strokes
        x0 = W // 8 + (k * 2 + 1) * (6 * W // 8) // 13
        img[:, :, (3 * H) // 8 + H // 32:(5 * H) // 8 - H // 32, x0:x0 + max(W // 64, 1)] = -0.8
    return img
