"""Pins oracle/vit.py against the REAL dependency of the reference: transformers' `ViTModel`.

The reference builds its glyph encoder as `VisionEncoderDecoderModel.from_pretrained('microsoft/trocr-large-printed')
.encoder` (train_diffute_v1.py:630-631, app.ipynb:546-548) and calls `trocr_model(pixel_values).last_hidden_state`
(train_diffute_v1.py:868-871, app.ipynb:773-776).  That encoder class is transformers' ViT model; `transformers` IS
installed in the build container (5.x), so its forward can be run here on seeded weights and its output committed as a
golden vector.  Run from the repo root (build container only; nothing under tests/ imports transformers at test time
unless it is importable, and the fixture travels as data):

    python scripts/pin_vit_oracle.py

Writes tests/golden/vit_transformers.npz: for each case the config, the pixel input and ViTModel's last_hidden_state
(fp32, eager attention).  Weights are NOT stored: they are regenerated from the counter PRNG (diffute_amd.init.init_param,
seed 777, keyed by the transformers-4.x / checkpoint key names that oracle/vit.py and diffute_amd.TrOCREncoder use).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffute_amd.init import init_param  # noqa: E402
from oracle import vit as OVT  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "vit_transformers.npz")

# checkpoint-era (transformers 4.x, what microsoft/trocr-* ships) key -> transformers 5.x module key
_REN = (("encoder.layer.", "layers."), ("attention.attention.query", "attention.q_proj"), ("attention.attention.key", "attention.k_proj"),
        ("attention.attention.value", "attention.v_proj"), ("attention.output.dense", "attention.o_proj"),
        ("intermediate.dense", "mlp.fc1"), ("output.dense", "mlp.fc2"))

CASES = {
    # name: oracle cfg dict (oracle/vit.py keys), batch
    "tiny": (dict(OVT.TINY_VIT), 3),                              # 2 layers, 17 tokens, q/k/v biases (tests' TINY_VIT)
    "nobias": (dict(image_size=96, patch_size=16, num_channels=3, hidden_size=256, num_layers=3, num_heads=4,
                    intermediate_size=512, qkv_bias=False, layer_norm_eps=1e-12), 2),   # TrOCR-style: no q/k/v bias, 37 tokens
}
SEED = 777


def seeded_state(cfg):
    return {k: init_param(k, shp, seed=SEED) for k, shp in OVT.vit_param_spec(cfg).items()}


def to_v5(k, have):
    k5 = k
    for a, b in _REN:
        k5 = k5.replace(a, b)
    return k5 if k5 in have else k


def run_transformers(cfg, P, px):
    from transformers import ViTConfig, ViTModel
    hc = ViTConfig(image_size=cfg["image_size"], patch_size=cfg["patch_size"], num_channels=cfg["num_channels"], hidden_size=cfg["hidden_size"],
                   num_hidden_layers=cfg["num_layers"], num_attention_heads=cfg["num_heads"], intermediate_size=cfg["intermediate_size"],
                   qkv_bias=cfg["qkv_bias"], layer_norm_eps=cfg["layer_norm_eps"], hidden_act="gelu", hidden_dropout_prob=0.0,
                   attention_probs_dropout_prob=0.0)
    hc._attn_implementation = "eager"
    m = ViTModel(hc, add_pooling_layer=False).eval()
    have = set(m.state_dict().keys())
    sd = {to_v5(k, have): v for k, v in P.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    with torch.no_grad():
        return m(pixel_values=px).last_hidden_state


def main():
    import transformers
    torch.manual_seed(0)
    out = {"transformers_version": np.array(transformers.__version__), "seed": np.array(SEED)}
    for name, (cfg, B) in CASES.items():
        P = seeded_state(cfg)
        g = torch.Generator().manual_seed(31 + len(name))
        px = torch.randn(B, cfg["num_channels"], cfg["image_size"], cfg["image_size"], generator=g)
        ref = run_transformers(cfg, P, px)
        mine = OVT.vit_forward(P, cfg, px)
        err = float((mine - ref).norm() / ref.norm())
        print(f"{name}: oracle/vit.py vs transformers.ViTModel {transformers.__version__}: rel-L2 {err:.2e}, max abs {float((mine - ref).abs().max()):.2e}")
        assert err < 1e-5
        out[f"{name}_pixels"] = px.numpy()
        out[f"{name}_last_hidden_state"] = ref.numpy()
        out[f"{name}_cfg"] = np.array(repr(sorted(cfg.items())))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
