"""Oracle: ViT encoder of TrOCR (the glyph encoder), torch-CPU fp32.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates the public transformers `ViTModel` forward that the
reference calls as `trocr_model(pixel_values).last_hidden_state` (app.ipynb:546-548,773-776;
train_diffute_v1.py:630-631,868-871; `trocr_model = VisionEncoderDecoderModel.from_pretrained(...).encoder`):
patch embedding (conv k=stride=patch), [CLS] + learned position embeddings, pre-LayerNorm transformer blocks
(x += dense(attn(LN_before x)); x += dense(gelu(dense(LN_after x)))), final LayerNorm.  The pooler of ViTModel does not
touch last_hidden_state and is not restated.  PINNED against the reference's real dependency: scripts/pin_vit_oracle.py runs
transformers' own `ViTModel` (installed in the build container) on seeded weights and commits its output as
tests/golden/vit_transformers.npz; tests/test_host.py holds this restatement to it (and to the live ViTModel whenever
transformers is importable).  Parameter names follow the checkpoint-era (transformers 4.x) ViTModel state dict that
microsoft/trocr-* ships, so real checkpoints load.

`emulate_bf16=True` rounds weights and every tensor the HIP path materialises to bf16 at the same points."""
from collections import OrderedDict

import torch
import torch.nn.functional as F

TROCR_LARGE_VIT = dict(image_size=384, patch_size=16, num_channels=3, hidden_size=1024, num_layers=24, num_heads=16,
                       intermediate_size=4096, qkv_bias=False, layer_norm_eps=1e-12)
TINY_VIT = dict(image_size=64, patch_size=16, num_channels=3, hidden_size=128, num_layers=2, num_heads=2,
                intermediate_size=256, qkv_bias=True, layer_norm_eps=1e-12)


def vit_param_spec(cfg):
    D, I, P, C = cfg["hidden_size"], cfg["intermediate_size"], cfg["patch_size"], cfg["num_channels"]
    n = (cfg["image_size"] // P) ** 2
    spec = OrderedDict()
    spec["embeddings.cls_token"] = (1, 1, D)
    spec["embeddings.position_embeddings"] = (1, n + 1, D)
    spec["embeddings.patch_embeddings.projection.weight"] = (D, C, P, P)
    spec["embeddings.patch_embeddings.projection.bias"] = (D,)
    for i in range(cfg["num_layers"]):
        p = f"encoder.layer.{i}."
        spec[p + "layernorm_before.weight"] = (D,); spec[p + "layernorm_before.bias"] = (D,)
        for nm in ("query", "key", "value"):
            spec[p + f"attention.attention.{nm}.weight"] = (D, D)
        if cfg["qkv_bias"]:
            for nm in ("query", "key", "value"):
                spec[p + f"attention.attention.{nm}.bias"] = (D,)
        spec[p + "attention.output.dense.weight"] = (D, D); spec[p + "attention.output.dense.bias"] = (D,)
        spec[p + "layernorm_after.weight"] = (D,); spec[p + "layernorm_after.bias"] = (D,)
        spec[p + "intermediate.dense.weight"] = (I, D); spec[p + "intermediate.dense.bias"] = (I,)
        spec[p + "output.dense.weight"] = (D, I); spec[p + "output.dense.bias"] = (D,)
    spec["layernorm.weight"] = (D,); spec["layernorm.bias"] = (D,)
    return spec


def _q(x, on):
    return x.to(torch.bfloat16).to(torch.float32) if on else x


@torch.no_grad()
def vit_forward(P, cfg, pixel_values, emulate_bf16=False):
    """last_hidden_state [B, N+1, D] fp32 from pixel_values [B, C, S, S]."""
    em = emulate_bf16
    D, H, eps, ps = cfg["hidden_size"], cfg["num_heads"], cfg["layer_norm_eps"], cfg["patch_size"]
    B = pixel_values.shape[0]
    x = _q(pixel_values.to(torch.float32), em)
    e = F.conv2d(x, _q(P["embeddings.patch_embeddings.projection.weight"], em), P["embeddings.patch_embeddings.projection.bias"], stride=ps)
    e = _q(e, em).flatten(2).transpose(1, 2)                                   # [B, N, D]
    x = torch.cat([P["embeddings.cls_token"].expand(B, -1, -1), e], dim=1) + P["embeddings.position_embeddings"]
    x = _q(x, em)
    d = D // H
    for i in range(cfg["num_layers"]):
        p = f"encoder.layer.{i}."
        n = _q(F.layer_norm(x, (D,), P[p + "layernorm_before.weight"], P[p + "layernorm_before.bias"], eps), em)
        def proj(nm):
            b = P.get(p + f"attention.attention.{nm}.bias")
            return _q(F.linear(n, _q(P[p + f"attention.attention.{nm}.weight"], em), b), em).view(B, -1, H, d).transpose(1, 2)
        q, k, v = proj("query"), proj("key"), proj("value")
        a = torch.matmul(torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5), dim=-1), v)
        a = _q(a.transpose(1, 2).reshape(B, -1, D), em)
        x = _q(x + F.linear(a, _q(P[p + "attention.output.dense.weight"], em), P[p + "attention.output.dense.bias"]), em)
        n = _q(F.layer_norm(x, (D,), P[p + "layernorm_after.weight"], P[p + "layernorm_after.bias"], eps), em)
        h = _q(F.gelu(F.linear(n, _q(P[p + "intermediate.dense.weight"], em), P[p + "intermediate.dense.bias"])), em)
        x = _q(x + F.linear(h, _q(P[p + "output.dense.weight"], em), P[p + "output.dense.bias"]), em)
    return _q(F.layer_norm(x, (D,), P["layernorm.weight"], P["layernorm.bias"], eps), em)


def vit_flops(cfg, B):
    D, I, L = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_layers"]
    n = (cfg["image_size"] // cfg["patch_size"]) ** 2; S = n + 1
    f = 2 * B * n * D * cfg["num_channels"] * cfg["patch_size"] ** 2
    f += L * (2 * B * S * D * 3 * D + 4 * B * S * S * D + 2 * B * S * D * D + 4 * B * S * D * I)
    return f
