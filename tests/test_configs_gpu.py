"""BASELINE.json configs 3, 4, 5 on the GPU (the three the round-1 suite never ran), plus the glyph encoder against the
fixture produced by transformers' own ViTModel.

cfg3: AutoencoderKL at the SD-VAE config (83,653,863 parameters; T2/T3/K6b at real shapes) - B=1, 256 px against the oracle
      run here; B=32, 512 px through size-independent properties.
cfg4: one B=8, 512 px training step of the full UNet - finite, bit-deterministic, and the gradient of ONE sample's loss
      taken inside the batch of 8 equals that sample's B=1 step (up to bf16 tile-plan noise).
cfg5: 768 px (latent 96) in FP16 as BASELINE names it - the fp16 build of the library (libdiffute_hip_f16.so, selected by
      `.to(dtype=torch.float16)`): B=1 against the fp16-emulating oracle run here, B=2 properties, a short 768-px denoise loop.
"""
import os

import numpy as np
import pytest
import torch

from util import assert_close, rel_l2

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
E2E_EMU = 2.5e-2      # whole-model bf16 bound vs the bf16-emulating oracle (see test_models_gpu.py)


# ------------------------------------------------------------------------------------------------ cfg3: full SD-VAE
@pytest.fixture(scope="module")
def sd_vae(cuda):
    import diffute_amd as D
    vae = D.AutoencoderKL(device=cuda).requires_grad_(False)
    assert sum(p.numel() for p in vae.parameters()) == 83_653_863
    return vae


def test_cfg3_sd_vae_256px_vs_oracle(cuda, sd_vae):
    """T2 / T3 / K6b at the SD-VAE widths (128..512 channels, d=512 single-head attention over 32x32 tokens): encode
    moments and decoded image of one 256-px crop against the bf16-emulating oracle and the fp32 oracle."""
    from diffute_amd.init import normal
    from diffute_amd.synthetic import text_crop_images
    from oracle import vae as OV
    img = text_crop_images(1, 256, 256, device=cuda)
    z = normal(5, 61, 4 * 32 * 32, cuda).reshape(1, 4, 32, 32)
    with torch.no_grad():
        mom = sd_vae.encode(img).latent_dist.parameters
        dec = sd_vae.decode(z).sample
    P = {k: v.detach().cpu().float() for k, v in sd_vae.state_dict().items()}
    e1 = assert_close(mom, OV.vae_encode_moments(P, OV.SD_VAE, img.cpu(), emulate_bf16=True), E2E_EMU, "SD-VAE encode 256 px vs bf16emu")
    e2 = assert_close(mom, OV.vae_encode_moments(P, OV.SD_VAE, img.cpu()), 5e-2, "SD-VAE encode 256 px vs fp32")
    e3 = assert_close(dec, OV.vae_decode(P, OV.SD_VAE, z.cpu(), emulate_bf16=True), E2E_EMU, "SD-VAE decode 256 px vs bf16emu")
    e4 = assert_close(dec, OV.vae_decode(P, OV.SD_VAE, z.cpu()), 5e-2, "SD-VAE decode 256 px vs fp32")
    print(f"cfg3 SD-VAE 256 px rel-L2: encode {e1:.2e} (bf16emu) {e2:.2e} (fp32); decode {e3:.2e} (bf16emu) {e4:.2e} (fp32)")
    # the fp32 validation instantiation of the same graphs (north_star: within 1e-3 rel of the fp32 reference path)
    f1 = assert_close(sd_vae.encode_fp32(img), OV.vae_encode_moments(P, OV.SD_VAE, img.cpu()), 1e-3, "SD-VAE encode 256 px, fp32 path vs fp32 oracle")
    f2 = assert_close(sd_vae.decode_fp32(z), OV.vae_decode(P, OV.SD_VAE, z.cpu()), 1e-3, "SD-VAE decode 256 px, fp32 path vs fp32 oracle")
    print(f"cfg3 SD-VAE 256 px fp32 validation path rel-L2: encode {f1:.2e}, decode {f2:.2e}")


def test_cfg3_sd_vae_batch32_512px_properties(cuda, sd_vae):
    """the cfg3 workload itself: encode + decode of 32 crops at 512 px.  Finite, deterministic (bit-equal repeat), batch
    independent (image 5 of the batch == image 5 alone, bf16 tile-plan noise only), encode->decode round trip shape."""
    from diffute_amd.synthetic import text_crop_images
    img = text_crop_images(32, 512, 512, device=cuda)
    with torch.no_grad():
        mom = sd_vae.encode(img).latent_dist.parameters.clone()
        assert mom.shape == (32, 8, 64, 64) and torch.isfinite(mom).all() and float(mom.std()) > 1e-3
        mom2 = sd_vae.encode(img).latent_dist.parameters
        assert torch.equal(mom, mom2)
        z = mom[:, :4].contiguous()
        dec = sd_vae.decode(z).sample.clone()
        assert dec.shape == (32, 3, 512, 512) and torch.isfinite(dec).all()
        assert torch.equal(dec, sd_vae.decode(z).sample)
        m1 = sd_vae.encode(img[5:6].contiguous()).latent_dist.parameters
        d1 = sd_vae.decode(z[5:6].contiguous()).sample
    assert rel_l2(m1, mom[5:6]) < 2e-2 and rel_l2(d1, dec[5:6]) < 2e-2
    # images are independent: sample 5 must not depend on what else is in the batch
    with torch.no_grad():
        img_b = img.clone(); img_b[6] = -img_b[6]
        mom_b = sd_vae.encode(img_b).latent_dist.parameters
    assert torch.equal(mom_b[5], mom[5]) and not torch.equal(mom_b[6], mom[6])


# ------------------------------------------------------------------------------------------------ cfg4: B=8, 512 px train step
def test_cfg4_train_step_batch8_512px(cuda):
    """P5 at the cfg4 per-GPU shape (8 x 512 px, 865.9 M parameters): finite loss / gradients, two identical steps give
    bit-identical gradients (no atomics anywhere in the backward), and d(loss of sample 0)/dW taken inside the batch of 8
    equals the B=1 step on that sample (gradient rel-L2 <= 6e-2 per tensor norm-weighted; B=1 and B=8 use different
    tile / split plans, so only bf16 rounding noise separates them)."""
    import diffute_amd as D
    from diffute_amd.models import mse_loss
    from diffute_amd.synthetic import synth_inputs
    from oracle import prng
    unet = D.UNet2DConditionModel(device=cuda)
    lat, mask, mlat, ctx = synth_inputs(8, 64, 64, 577, 1024, device=cuda)
    x = torch.cat([lat, mask, mlat], 1)
    t = torch.tensor([437, 12, 999, 650, 3, 800, 250, 501], device=cuda)
    tgt = torch.from_numpy(prng.normal(9, 43, 8 * 4 * 64 * 64).reshape(8, 4, 64, 64)).to(cuda)

    def step(xs, ts, cs, tg, sel=None):
        unet.zero_grad(set_to_none=True)
        pred = unet(xs, ts, cs).sample
        loss = mse_loss(pred if sel is None else pred[sel], tg if sel is None else tg[sel])
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), pred.detach().clone(), {k: p.grad.clone() for k, p in unet.named_parameters()}

    l_a, p_a, g_a = step(x, t, ctx, tgt)
    assert np.isfinite(l_a) and torch.isfinite(p_a).all()
    for k, g in g_a.items():
        assert torch.isfinite(g).all(), f"{k}: non-finite gradient"
    assert float(torch.sqrt(sum(g.float().pow(2).sum() for g in g_a.values()))) > 0
    l_b, p_b, g_b = step(x, t, ctx, tgt)
    assert l_a == l_b and torch.equal(p_a, p_b)
    for k in g_a:
        assert torch.equal(g_a[k], g_b[k]), f"{k}: gradients differ between identical steps"
    del g_b, p_b
    # gradient of sample 0's loss inside the batch vs the B=1 step on sample 0
    l_s, p_s, g_s = step(x, t, ctx, tgt, sel=slice(0, 1))
    assert torch.equal(p_s, p_a), "the sample-0 step runs the SAME forward as the full step (same inputs): its prediction must have the same bits"
    l_1, p_1, g_1 = step(x[:1].contiguous(), t[:1].contiguous(), ctx[:1].contiguous(), tgt[:1].contiguous())
    assert abs(l_s - l_1) <= 2e-2 * abs(l_1)
    assert rel_l2(p_s[:1], p_1) < 2e-2
    num = den = 0.0
    worst = (0.0, "")
    for k in g_1:
        a, b = g_s[k].float(), g_1[k].float()
        num += float((a - b).pow(2).sum()); den += float(b.pow(2).sum())
        r = abs(float(a.norm()) / max(float(b.norm()), 1e-30) - 1)
        if r > worst[0]:
            worst = (r, k)
    tot = (num / den) ** 0.5
    print(f"cfg4: loss {l_a:.5f}; sample-0 gradient inside B=8 vs alone: whole-gradient rel-L2 {tot:.2e}, worst norm ratio off by {worst[0]:.3f} ({worst[1]})")
    assert tot <= 6e-2, f"per-sample gradient mismatch: {tot:.3e}"
    assert worst[0] <= 0.05, f"gradient norm of {worst[1]} off by {worst[0]:.3f}"


def test_cfg4_fp16_mixed_precision_step(cuda):
    """`--mixed_precision fp16` (train_diffute_v1.py:267,583,790) at full size: the SD2-inpaint UNet on the fp16 build, batch 2 x 512 px, one step of
    diffute_amd.training's loop shape with GradScaler (accelerate's default scale 2^16) + FusedAdamW.  The scaled backward stays finite (no skipped
    step), loss and UNSCALED gradient norm agree with the bf16 build's step on the same inputs and weights (the two builds differ by rounding only),
    and the exported, unscaled gradients of the two builds agree to bf16 noise."""
    import diffute_amd as D
    from diffute_amd.models import mse_loss
    from diffute_amd.synthetic import synth_inputs
    from oracle import prng
    lat, mask, mlat, ctx = synth_inputs(2, 64, 64, 577, 1024, device=cuda)
    x = torch.cat([lat, mask, mlat], 1)
    t = torch.tensor([437, 12], device=cuda)
    tgt = torch.from_numpy(prng.normal(9, 43, 2 * 4 * 64 * 64).reshape(2, 4, 64, 64)).to(cuda)
    res = {}
    for dt in (torch.bfloat16, torch.float16):
        unet = D.UNet2DConditionModel(device=cuda).to(dtype=dt)
        S = 65536.0 if dt == torch.float16 else 1.0
        loss = mse_loss(unet(x, t, ctx).sample, tgt)
        (loss * S).backward()
        g = {k: (p.grad.float() / S) for k, p in unet.named_parameters()}
        unet.zero_grad(set_to_none=True)
        opt = D.FusedAdamW(unet, lr=1e-5, max_grad_norm=1.0)
        scaler = D.GradScaler(enabled=dt == torch.float16)
        loss2 = mse_loss(unet(x, t, ctx).sample, tgt)
        scaler.scale(loss2).backward()
        scaler.step(opt); scaler.update()
        assert not opt.found_inf and opt.t == 1, f"{dt}: the step was skipped (gradient overflow at scale {scaler.get_scale()})"
        assert scaler.get_scale() == (65536.0 if dt == torch.float16 else 1.0)
        assert float(loss2.detach()) == float(loss.detach())
        res[dt] = (float(loss.detach()), float(opt.grad_norm), g)
        gn = float(torch.sqrt(sum(v.pow(2).sum() for v in g.values())))
        assert all(torch.isfinite(v).all() for v in g.values())
        assert abs(gn - float(opt.grad_norm)) <= 1e-4 * gn, f"{dt}: fused unscaled norm {float(opt.grad_norm)} vs exported {gn}"
        del unet, opt
    (lb, nb, gb), (lh, nh, gh) = res[torch.bfloat16], res[torch.float16]
    num = sum(float((gh[k] - gb[k]).pow(2).sum()) for k in gb); den = sum(float(gb[k].pow(2).sum()) for k in gb)
    tot = (num / den) ** 0.5
    print(f"cfg4 fp16 mixed precision: loss {lh:.5f} (bf16 build {lb:.5f}); gradient norm {nh:.4f} ({nb:.4f}); gradients fp16 vs bf16 build rel-L2 {tot:.2e}")
    assert abs(lh - lb) <= 1e-2 * abs(lb) and abs(nh - nb) <= 5e-2 * nb
    assert tot <= 6e-2


# ------------------------------------------------------------------------------------------------ cfg5: 768 px, fp16
E2E_FP16 = 8e-3       # whole-model fp16 bound vs the fp16-emulating oracle (11 mantissa bits: ~8x tighter than bf16's 2.5e-2)


def test_cfg5_unet_768px_fp16(cuda):
    """BASELINE configs[4] as written: 768 px (latent 96; S = 9216 self-attention rows at the first level), batch 2, FP16.
    `.to(dtype=torch.float16)` (vae.to(device, dtype=weight_dtype), train_diffute_v1.py:789-797) moves the model to the fp16
    build of the library (fp16 storage, v_mfma_f32_32x32x16_f16, fp32 accumulation).  B=1 against the fp16-emulating oracle
    and the fp32 oracle run here; B=2 properties (finite - an fp16 overflow would show as inf / nan -, deterministic, batch
    independent); a 3-step 768-px denoise loop equal to its reference-shaped twin; and the bf16 build on the same input
    against the bf16-emulating oracle (the two builds must also agree with each other to bf16 noise)."""
    import diffute_amd as D
    from diffute_amd.synthetic import synth_inputs
    from oracle import unet as OU
    unet = D.UNet2DConditionModel(device=cuda).requires_grad_(False)
    lat, mask, mlat, ctx = synth_inputs(2, 96, 96, 577, 1024, device=cuda)
    t = torch.tensor([981], device=cuda)
    unet.set_context(ctx[1:2].contiguous())
    yb = unet.forward_parts([lat[1:2].contiguous(), mask[1:2].contiguous(), mlat[1:2].contiguous()], t).clone()      # bf16 build
    unet.to(dtype=torch.float16)
    assert unet.compute_dtype == torch.float16 and unet.dtype == torch.float16
    unet.set_context(ctx)
    y = unet.forward_parts([lat, mask, mlat], t).clone()
    assert y.shape == (2, 4, 96, 96) and torch.isfinite(y).all() and float(y.std()) > 1e-3
    assert torch.equal(unet.forward_parts([lat, mask, mlat], t), y)
    unet.set_context(ctx[1:2].contiguous())
    y1 = unet.forward_parts([lat[1:2].contiguous(), mask[1:2].contiguous(), mlat[1:2].contiguous()], t)
    assert rel_l2(y1, y[1:2]) < 5e-3
    P = {k: v.detach().cpu().float() for k, v in unet.state_dict().items()}
    x1 = torch.cat([lat[1:2], mask[1:2], mlat[1:2]], 1).cpu()
    ref16 = OU.unet_forward(P, OU.SD2_INPAINT_UNET, x1, torch.tensor(981), ctx[1:2].cpu(), emulate_bf16="fp16")
    ref32 = OU.unet_forward(P, OU.SD2_INPAINT_UNET, x1, torch.tensor(981), ctx[1:2].cpu())
    refb = OU.unet_forward(P, OU.SD2_INPAINT_UNET, x1, torch.tensor(981), ctx[1:2].cpu(), emulate_bf16=True)
    e16 = assert_close(y1, ref16, E2E_FP16, "cfg5 UNet forward (fp16 build), latent 96, vs fp16-emulating oracle")
    e32 = assert_close(y1, ref32, E2E_FP16, "cfg5 UNet forward (fp16 build), latent 96, vs fp32 oracle")
    eb = assert_close(yb, refb, E2E_EMU, "cfg5 UNet forward (bf16 build), latent 96, vs bf16-emulating oracle")
    print(f"cfg5 UNet 768 px rel-L2: fp16 build {e16:.2e} (fp16emu oracle) {e32:.2e} (fp32 oracle); bf16 build {eb:.2e} (bf16emu oracle); "
          f"fp16 vs bf16 build {rel_l2(y1, yb):.2e}; max |eps| {float(y.abs().max()):.2f}")
    unet.set_context(ctx)
    out = D.denoise(unet, D.DDIMScheduler(), lat, mask, mlat, ctx, 3)
    sch = D.DDIMScheduler(); sch.set_timesteps(3)
    xx = lat * sch.init_noise_sigma
    with torch.no_grad():
        for tt in sch.timesteps:
            eps = unet(torch.cat([sch.scale_model_input(xx, tt), mask, mlat], dim=1), tt, ctx).sample
            xx = sch.step(eps, tt, xx).prev_sample
    assert torch.isfinite(out).all() and torch.equal(xx, out)


def test_fp16_build_tiny_pipeline_vs_oracle(cuda):
    """the whole model path of text_editing() (app.ipynb:779-819) on the fp16 build: VAE encode -> 4-step DDIM loop -> VAE decode
    on tiny configs against the fp16-emulating oracle (the bf16 twin of this test is test_models_gpu.py::test_edit_latents...)."""
    import diffute_amd as D
    from diffute_amd.init import normal
    from diffute_amd.synthetic import synth_images
    from oracle import pipeline as OP, unet as OU, vae as OV
    unet = D.UNet2DConditionModel(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4), cross_attention_dim=128).to(cuda, dtype=torch.float16).requires_grad_(False)
    vae = D.AutoencoderKL(block_out_channels=(64, 128, 128, 128), layers_per_block=1).to(cuda, dtype=torch.float16).requires_grad_(False)
    assert unet.compute_dtype == torch.float16 and vae.compute_dtype == torch.float16
    img = synth_images(1, 128, 128, device=cuda)
    mask = torch.zeros(1, 1, 128, 128, device=cuda); mask[:, :, 48:80, 16:112] = 1.0
    ctx = normal(2, 13, 77 * 128, cuda).reshape(1, 77, 128)
    enc_noise = normal(8, 71, 4 * 16 * 16, cuda).reshape(1, 4, 16, 16)
    init = normal(0, 11, 4 * 16 * 16, cuda).reshape(1, 4, 16, 16)
    masked = img * (mask < 0.5)
    out = D.edit_latents(unet, vae, D.DDIMScheduler(), img, masked, mask, ctx, 4, init_latents=init, enc_noise=enc_noise)
    Pu = {k: v.detach().cpu().float() for k, v in unet.state_dict().items()}
    Pv = {k: v.detach().cpu().float() for k, v in vae.state_dict().items()}
    ref, _, _ = OP.edit_latents(Pu, OU.TINY_UNET, Pv, OV.TINY_VAE, masked.cpu(), mask.cpu(), ctx.cpu(), 4, enc_noise.cpu(), "ddim",
                                emulate_bf16="fp16", init=init.cpu())
    e = assert_close(out, ref, E2E_FP16, "tiny encode -> denoise -> decode on the fp16 build vs the fp16-emulating oracle")
    ref32, _, _ = OP.edit_latents(Pu, OU.TINY_UNET, Pv, OV.TINY_VAE, masked.cpu(), mask.cpu(), ctx.cpu(), 4, enc_noise.cpu(), "ddim", init=init.cpu())
    e32 = assert_close(out, ref32, E2E_FP16, "... vs the fp32 oracle")
    print(f"fp16 build, tiny text_editing() model path: rel-L2 {e:.2e} (fp16emu) {e32:.2e} (fp32)")


# ------------------------------------------------------------------------------------------------ N1 pinned: transformers fixture
def test_glyph_encoder_vs_transformers_fixture(cuda):
    """N1 against the reference's real dependency: the HIP ViT encoder on the seeded weights vs what transformers'
    ViTModel produced for the same weights and pixels (tests/golden/vit_transformers.npz, scripts/pin_vit_oracle.py).
    bf16 compute vs an fp32 reference: 5e-2 (measured ~6e-3)."""
    import sys
    import diffute_amd as D
    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), "..", "scripts"))
    import pin_vit_oracle as PV
    g = np.load(os.path.join(GOLD, "vit_transformers.npz"))
    for name, (cfg, B) in PV.CASES.items():
        enc = D.TrOCREncoder(image_size=cfg["image_size"], patch_size=cfg["patch_size"], hidden_size=cfg["hidden_size"],
                             num_hidden_layers=cfg["num_layers"], num_attention_heads=cfg["num_heads"],
                             intermediate_size=cfg["intermediate_size"], qkv_bias=cfg["qkv_bias"], seed=PV.SEED).cuda()
        P = PV.seeded_state(cfg)
        sd = enc.state_dict()
        assert set(sd) == set(P) and all(torch.equal(sd[k].cpu(), P[k]) for k in P), "product init differs from the fixture's seeded weights"
        with torch.no_grad():
            y = enc(torch.from_numpy(g[f"{name}_pixels"]).cuda()).last_hidden_state
        e = assert_close(y, torch.from_numpy(g[f"{name}_last_hidden_state"]), 5e-2, f"HIP ViT ({name}) vs transformers.ViTModel fixture")
        print(f"glyph encoder ({name}) vs transformers fixture: rel-L2 {e:.2e}")
        # the fp32 validation instantiation of the same graph against the same fixture: north_star's 1e-3 against the
        # reference's actual dependency
        yf = enc.forward_fp32(torch.from_numpy(g[f"{name}_pixels"]).cuda())
        ef = assert_close(yf, torch.from_numpy(g[f"{name}_last_hidden_state"]), 1e-3, f"HIP ViT fp32 path ({name}) vs transformers.ViTModel fixture")
        print(f"glyph encoder ({name}) fp32 validation path vs transformers fixture: rel-L2 {ef:.2e}")
