"""Secondary configurations of BASELINE.json (not the headline bench line): cfg3 VAE encode+decode at 512 px,
cfg5 768-px denoise loop, and the glyph encoder (TrOCR-large ViT).  Prints one JSON object per configuration."""
import argparse
import json
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffute_amd as D  # noqa: E402
from diffute_amd.flops import unet_flops  # noqa: E402
from diffute_amd.synthetic import synth_inputs, text_crop_images  # noqa: E402


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vae-batch", type=int, default=32)
    ap.add_argument("--skip-vae", action="store_true")
    ap.add_argument("--skip-768", action="store_true")
    ap.add_argument("--skip-vit", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda")
    if not args.skip_vae:
        vae = D.AutoencoderKL(device=dev).requires_grad_(False)
        B = args.vae_batch
        img = text_crop_images(B, 512, 512, device=dev)
        with torch.no_grad():
            t_enc, post = timed(lambda: vae.encode(img).latent_dist, 2)
            z = post.mode()
            t_dec, rec = timed(lambda: vae.decode(z).sample, 2)
        assert torch.isfinite(rec).all() and rec.shape == img.shape
        fe, fd = 1.1167e12 * B, 2.5145e12 * B
        print(json.dumps({"config": f"cfg3: AutoencoderKL encode+decode, 512x512, batch {B}, bf16", "encode_ms": round(t_enc * 1e3, 2),
                          "decode_ms": round(t_dec * 1e3, 2), "images_per_s": round(B / (t_enc + t_dec), 2),
                          "encode_tflops": round(fe / t_enc / 1e12, 1), "decode_tflops": round(fd / t_dec / 1e12, 1)}))
        del vae, img, rec, z
        torch.cuda.empty_cache()
    if not args.skip_vit:
        enc = D.TrOCREncoder(device=dev)
        fl1 = 2 * 576 * 1024 * 768 + 24 * (2 * 577 * 1024 * 3072 + 4 * 577 * 577 * 1024 + 2 * 577 * 1024 * 1024 + 4 * 577 * 1024 * 4096)
        for B in (1, 8):
            px = torch.randn(B, 3, 384, 384, device=dev)
            with torch.no_grad():
                t, y = timed(lambda: enc(px).last_hidden_state, 5)
            assert torch.isfinite(y).all() and y.shape == (B, 577, 1024)
            print(json.dumps({"config": f"glyph encoder: TrOCR-large ViT (24 layers, 577 tokens), batch {B}, bf16", "ms": round(t * 1e3, 2),
                              "images_per_s": round(B / t, 1), "tflops": round(fl1 * B / t / 1e12, 1)}))
        del enc
        torch.cuda.empty_cache()
    if not args.skip_768:
        unet = D.UNet2DConditionModel(device=dev).requires_grad_(False).to(dtype=torch.float16)      # the fp16 build of the library (BASELINE configs[4])
        lat, mask, mlat, ctx = synth_inputs(2, 96, 96, 577, 1024, device=dev)
        t, out = timed(lambda: D.denoise(unet, D.DDIMScheduler(), lat, mask, mlat, ctx, 50), 2)
        assert torch.isfinite(out).all()
        fl = 50 * unet_flops(unet.config, 2, 96, 96, 577, True)
        print(json.dumps({"config": "cfg5: 768x768 denoise loop, 50 DDIM steps, batch 2, fp16 (libdiffute_hip_f16.so)",
                          "ms_per_batch": round(t * 1e3, 1), "images_per_s": round(2 / t, 3), "loop_tflops": round(fl / t / 1e12, 1)}))


if __name__ == "__main__":
    main()
