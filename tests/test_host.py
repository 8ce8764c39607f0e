"""CPU-side checks (no GPU): oracle pins, host logic of the product, C-ABI library surface."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_param_counts_match_public_models():
    """The only external pins available: SD2-inpaint UNet 865,925,124 / SD VAE 83,653,863."""
    from oracle import unet as OU, vae as OV
    assert OU.count_params(OU.unet_param_spec(OU.SD2_INPAINT_UNET)) == 865_925_124
    assert OU.count_params(OV.vae_param_spec(OV.SD_VAE)) == 83_653_863


def test_scheduler_index_math_bit_exact():
    """S2: integer timestep grids (known answers) - oracle, product and committed fixture agree exactly."""
    import diffute_amd as D
    from oracle import schedulers as OS
    g = np.load(os.path.join(GOLD, "sched.npz"))
    for n, first, last in ((10, 900, 0), (50, 980, 0), (150, 894, 0)):
        p = D.DDPMScheduler(); p.set_timesteps(n)
        assert p.timesteps.dtype == torch.int64
        assert np.array_equal(p.timesteps.numpy(), OS.timesteps_ddpm(n)) and np.array_equal(p.timesteps.numpy(), g[f"ddpm_{n}"])
        assert int(p.timesteps[0]) == first and int(p.timesteps[-1]) == last
        d = D.DDIMScheduler(); d.set_timesteps(n)
        assert np.array_equal(d.timesteps.numpy(), OS.timesteps_ddim(n)) and np.array_equal(d.timesteps.numpy(), g[f"ddim_{n}"])
        assert int(d.timesteps[0]) == first + 1 and int(d.timesteps[-1]) == 1
        assert all(p.previous_timestep(int(t)) == OS.prev_timestep(int(t), n) for t in p.timesteps)
    assert D.DDPMScheduler().init_noise_sigma == 1.0 and D.DDPMScheduler().num_train_timesteps == 1000
    assert D.DDPMScheduler().config.prediction_type == "epsilon"


def test_scheduler_tables():
    """alphas_cumprod: oracle's scalar restatement of torch.linspace/cumprod vs the product's torch expression
    (<= 1 ulp apart: torch's vectorised linspace kernel is SIMD-width dependent) and vs the fixture (exact)."""
    import diffute_amd as D
    from oracle import schedulers as OS
    betas, alphas, ac = OS.make_tables()
    g = np.load(os.path.join(GOLD, "sched.npz"))
    assert np.array_equal(ac, g["alphas_cumprod"]) and np.array_equal(betas, g["betas"])
    s = D.DDIMScheduler()
    assert np.abs(s.alphas_cumprod.numpy() - ac).max() <= 6e-8
    assert abs(float(ac[0]) - 0.99915) < 1e-6 and abs(float(ac[-1]) - 0.0046602) < 1e-6
    # step coefficients: product host math == oracle scalars when fed the same table
    s.alphas_cumprod = torch.from_numpy(ac.copy()); s.final_alpha_cumprod = s.alphas_cumprod[0]
    s.set_timesteps(50)
    x, e = g["x"], g["eps"]
    sbt, sat, sap, dirc, std = s.step_coefficients(981)
    ref = OS.ddim_step(ac, e, 981, x, 50)
    mine = (np.float32(sap) * ((x - np.float32(sbt) * e) / np.float32(sat)) + np.float32(dirc) * e).astype(np.float32)
    assert np.array_equal(mine, ref) and np.array_equal(ref, g["ddim_step_981_50"])


def test_oracle_golden_reproducible():
    """The committed tiny goldens are what the oracle computes today."""
    from oracle import prng, unet as OU
    g = np.load(os.path.join(GOLD, "tiny_unet.npz"))
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from make_golden import synth_inputs
    cfg = OU.TINY_UNET
    P = OU.make_params(OU.unet_param_spec(cfg), seed=1234)
    lat, mask, mlat, ctx = synth_inputs(2, 16, 16, 77, cfg["cross_attention_dim"])
    y = OU.unet_forward(P, cfg, torch.cat([lat, mask, mlat], 1), torch.tensor(981), ctx)
    assert float((y - torch.from_numpy(g["eps_fp32"])).abs().max()) < 1e-4


def test_product_param_table_matches_oracle_spec():
    """Two independent enumerations (C++ ParamTable vs oracle spec) agree on every key, shape and value."""
    import diffute_amd as D
    from oracle import unet as OU, vae as OV
    u = D.UNet2DConditionModel(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4), cross_attention_dim=128)
    spec = OU.unet_param_spec(OU.TINY_UNET)
    sd = u.state_dict()
    assert set(sd) == set(spec) and all(tuple(sd[k].shape) == tuple(spec[k]) for k in spec)
    P = OU.make_params(spec)
    assert all(torch.equal(sd[k], P[k]) for k in spec)
    v = D.AutoencoderKL(block_out_channels=(64, 128, 128, 128), layers_per_block=1)
    vs = OV.vae_param_spec(OV.TINY_VAE)
    assert set(v.state_dict()) == set(vs)
    assert v.config.scaling_factor == 0.18215 and 2 ** (len(v.config.block_out_channels) - 1) == 8


def test_full_config_param_table():
    from diffute_amd import _cabi
    import ctypes
    import diffute_amd as D
    c = _cabi.UNetConfig()
    cfg = D.SD2_INPAINT_UNET_CONFIG
    c.in_channels, c.out_channels, c.layers_per_block, c.cross_attention_dim, c.norm_num_groups = 9, 4, 2, 1024, 32
    for i in range(4):
        c.block_out_channels[i] = cfg["block_out_channels"][i]; c.heads[i] = cfg["attention_head_dim"][i]
        c.down_has_attn[i] = int(i < 3); c.up_has_attn[i] = int(i > 0)
    lib = _cabi.lib()
    h = lib.dmx_unet_create(ctypes.byref(c))
    assert h
    n = lib.dmx_unet_param_count(h)
    name = ctypes.c_char_p(); shape = (ctypes.c_int * 4)()
    total = 0
    for i in range(n):
        assert lib.dmx_unet_param_info(h, i, ctypes.byref(name), ctypes.byref(shape)) == 0
        total += int(np.prod([s for s in shape if s > 0]))
    assert n == 686 and total == 865_925_124
    assert lib.dmx_unet_arena_bytes(h) > 2 * 865_000_000 * 0.99
    # workspace plan is a pure host computation
    assert lib.dmx_unet_workspace_bytes(h, 4, 64, 64, 577) > 0
    lib.dmx_unet_destroy(h)


def test_cabi_exports_every_declared_symbol():
    from diffute_amd import _cabi
    hdr = open(os.path.join(ROOT, "include", "diffute_hip.h")).read()
    declared = set(re.findall(r"\b(dmx_[a-z0-9_]+)\s*\(", hdr))
    for elem in ("bf16", "fp16"):          # the two builds of the same sources (libdiffute_hip.so, libdiffute_hip_f16.so)
        lib = _cabi.lib(elem)
        for sym in declared:
            assert hasattr(lib, sym), f"{sym} declared in include/diffute_hip.h but not exported by the {elem} build"
        assert lib.dmx_element_type().decode() == elem
    assert declared == set(_cabi.exported_symbols())


def test_dtype_selects_the_build():
    """`.to(dtype=torch.float16)` (vae.to(device, dtype=weight_dtype), train_diffute_v1.py:789-797) moves a model to the fp16 build;
    bf16 / fp32 requests stay on the bf16 build; the Parameters (fp32 masters) are untouched by the switch."""
    import diffute_amd as D
    u = D.UNet2DConditionModel(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4), cross_attention_dim=128)
    before = {k: v.clone() for k, v in u.state_dict().items()}
    assert u.compute_dtype == torch.bfloat16
    u.to(dtype=torch.float16)
    assert u.compute_dtype == torch.float16 and u.dtype == torch.float16 and u._lib is _cabi_lib("fp16")
    u.to(dtype=torch.float32)
    assert u.compute_dtype == torch.bfloat16 and u._lib is _cabi_lib("bf16")
    after = u.state_dict()
    assert all(torch.equal(before[k], after[k]) and after[k].dtype == torch.float32 for k in before)
    v = D.AutoencoderKL(block_out_channels=(64, 128, 128, 128), layers_per_block=1).to("cpu", dtype=torch.float16)
    assert v.compute_dtype == torch.float16
    with pytest.raises(NotImplementedError):
        D.AutoencoderKL(block_out_channels=(64, 128, 128, 192), layers_per_block=1)      # mid-block width outside 128 / 256 / 512


def _cabi_lib(elem):
    from diffute_amd import _cabi
    return _cabi.lib(elem)


def test_error_paths_without_gpu():
    """Loud failures: CPU tensors are rejected, bad configs are rejected, errors carry a message."""
    import diffute_amd as D
    from diffute_amd import _cabi
    with pytest.raises(ValueError):
        D.UNet2DConditionModel(block_out_channels=(48, 96, 192, 192), attention_head_dim=(1, 2, 4, 4))
    s = D.DDIMScheduler(); s.set_timesteps(10)
    with pytest.raises(RuntimeError):
        s.step(torch.zeros(1, 4, 8, 8), 901, torch.zeros(1, 4, 8, 8))
    u = D.UNet2DConditionModel(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4), cross_attention_dim=128)
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            u(torch.zeros(1, 9, 8, 8), 10, torch.zeros(1, 7, 128))
    with pytest.raises(ValueError):
        D.DDPMScheduler().step_coefficients(10)        # set_timesteps not called
    assert _cabi.lib().dmx_version() >= 100


def test_mask_downsample_nearest():
    """P3: F.interpolate(mask, size=(h/8,w/8)) default nearest == src index floor(dst*8)."""
    import torch.nn.functional as F
    import diffute_amd as D
    m = torch.zeros(1, 1, 64, 64); m[:, :, 17:41, 9:50] = 1
    assert torch.equal(D.mask_to_latent(m), F.interpolate(m, size=(8, 8)))


def test_save_load_roundtrip(tmp_path):
    import diffute_amd as D
    v = D.AutoencoderKL(block_out_channels=(64, 128, 128, 128), layers_per_block=1)
    v.save_pretrained(str(tmp_path / "vae"))
    v2 = D.AutoencoderKL.from_pretrained(str(tmp_path), subfolder="vae")
    assert all(torch.equal(a, b) for a, b in zip(v.state_dict().values(), v2.state_dict().values()))
    s = D.DDPMScheduler(); s.save_pretrained(str(tmp_path / "scheduler"))
    s2 = D.DDPMScheduler.from_pretrained(str(tmp_path), subfolder="scheduler")
    assert torch.equal(s.alphas_cumprod, s2.alphas_cumprod)
    # legacy VAE attention key names (query/key/value/proj_attn) are accepted
    sd = {k.replace("to_q", "query").replace("to_k", "key").replace("to_v", "value").replace("to_out.0", "proj_attn"): t
          for k, t in v.state_dict().items()}
    v3 = D.AutoencoderKL(block_out_channels=(64, 128, 128, 128), layers_per_block=1, seed=1)
    v3.load_state_dict(D.AutoencoderKL._convert_legacy_keys(sd))
    assert all(torch.equal(a, b) for a, b in zip(v.state_dict().values(), v3.state_dict().values()))


def test_flops_formula_matches_oracle():
    import diffute_amd as D
    from diffute_amd.flops import unet_flops
    from oracle import unet as OU
    u_cfg = D.models._Config(**{**D.SD2_INPAINT_UNET_CONFIG})
    assert unet_flops(u_cfg, 4, 64, 64, 577, True) == OU.unet_flops(OU.SD2_INPAINT_UNET, 4, 64, 64, 577, True)
    assert abs(unet_flops(u_cfg, 1, 64, 64, 577, False) / 1e12 - 0.853) < 0.002


def test_oracle_training_step_matches_golden():
    """P5/P6 oracle (autograd over the restatement) against the committed fixture: loss, global gradient norm, the L2
    norm of every parameter gradient and a dozen gradients in full (regenerated from the same seeds)."""
    import os
    import numpy as np
    import torch
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts"))
    from make_golden import TRAIN_KEYS, synth_inputs
    from oracle import pipeline as OP, prng, unet as OU
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_train.npz"))
    cfg = OU.TINY_UNET
    P = OU.make_params(OU.unet_param_spec(cfg), seed=1234)
    lat, mask, mlat, ctx = synth_inputs(2, 16, 16, 77, cfg["cross_attention_dim"])
    inp = torch.cat([lat, mask, mlat], 1)
    tgt = torch.from_numpy(prng.normal(9, 41, 2 * 4 * 16 * 16).reshape(2, 4, 16, 16))
    loss, pred, grads = OP.unet_train_grads(P, cfg, inp, torch.tensor([981, 17]), ctx, tgt)
    assert abs(loss - float(g["loss_fp32"])) <= 1e-5 * abs(loss)
    assert np.allclose(pred.numpy(), g["pred_fp32"], rtol=1e-4, atol=1e-5)
    assert str(g["names"]).split("\n") == list(grads.keys())
    norms = np.array([float(v.double().pow(2).sum().sqrt()) for v in grads.values()])
    assert np.allclose(norms, g["gnorms_fp32"], rtol=1e-3)
    for k in TRAIN_KEYS:
        ref = g["g_fp32_" + k]
        assert np.linalg.norm(grads[k].numpy() - ref) <= 1e-3 * np.linalg.norm(ref), k


def test_glyph_encoder_param_table_matches_oracle_spec():
    """N1: the C++ parameter table of the TrOCR ViT encoder agrees key-by-key and shape-by-shape with the oracle's
    enumeration, for the tiny config (with q/k/v biases) and the TrOCR-large config (303,617,024 parameters without pooler)."""
    import ctypes
    from diffute_amd import _cabi
    from oracle import vit as OVT
    lib = _cabi.lib()
    for cfg in (OVT.TINY_VIT, OVT.TROCR_LARGE_VIT):
        c = _cabi.ViTConfig(cfg["image_size"], cfg["patch_size"], cfg["num_channels"], cfg["hidden_size"], cfg["num_layers"], cfg["num_heads"],
                            cfg["intermediate_size"], int(cfg["qkv_bias"]), cfg["layer_norm_eps"])
        h = lib.dmx_vit_create(ctypes.byref(c))
        assert h
        spec = OVT.vit_param_spec(cfg)
        name = ctypes.c_char_p(); shape = (ctypes.c_int * 4)()
        got = {}
        for i in range(lib.dmx_vit_param_count(h)):
            assert lib.dmx_vit_param_info(h, i, ctypes.byref(name), ctypes.byref(shape)) == 0
            got[name.value.decode()] = tuple(int(s) for s in shape if s > 0)
        lib.dmx_vit_destroy(h)
        assert got == dict(spec)
    n = sum(int(np.prod(s)) for s in OVT.vit_param_spec(OVT.TROCR_LARGE_VIT).values())
    assert n == 303_617_024


def test_crop_ladder_and_origin_match_the_notebook_logic():
    """N2 host logic (app.ipynb:674-720): product mirror vs the oracle's restatement over a sweep of boxes, and the
    ladder's documented break points"""
    import numpy as np
    from diffute_amd import prepost as P
    from oracle import prepost as OP
    assert P.crop_scale_for([0, 0, 50, 20], 1000, 1000) == 128            # 6*20 < 128
    assert P.crop_scale_for([0, 0, 50, 22], 1000, 1000) == 256            # 132 >= 128
    assert P.crop_scale_for([0, 0, 700, 22], 1000, 1200) == 1000          # text longer than the ladder value -> short side
    assert P.crop_scale_for([0, 0, 50, 200], 900, 1200) == 900            # 6*200 = 1200 > short side
    rng = np.random.RandomState(3)
    for _ in range(300):
        h, w = int(rng.randint(64, 1500)), int(rng.randint(64, 1500))
        x1 = int(rng.randint(0, w - 8)); y1 = int(rng.randint(0, h - 8))
        x2 = int(rng.randint(x1 + 4, w)); y2 = int(rng.randint(y1 + 2, h))
        loc = [x1, y1, x2, y2]
        cs = P.crop_scale_for(loc, h, w)
        assert cs == OP.crop_scale_for(loc, h, w) and 0 < cs <= min(h, w)
        try:
            want = OP.crop_origin(loc, cs, w, np.random.RandomState(9))
        except ValueError:                                                  # np.random.randint(low >= high): the reference raises too
            with pytest.raises(ValueError):
                P.crop_origin(loc, cs, w, np.random.RandomState(9))
            continue
        assert P.crop_origin(loc, cs, w, np.random.RandomState(9)) == want


def _vit_pin_cases():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import pin_vit_oracle as PV
    return PV


def test_vit_oracle_pinned_to_transformers_fixture():
    """N1 oracle pin: oracle/vit.py on the seeded weights reproduces what transformers' own ViTModel produced
    (tests/golden/vit_transformers.npz, generated by scripts/pin_vit_oracle.py in the build container)."""
    from oracle import vit as OVT
    PV = _vit_pin_cases()
    g = np.load(os.path.join(GOLD, "vit_transformers.npz"))
    for name, (cfg, B) in PV.CASES.items():
        assert str(g[f"{name}_cfg"]) == repr(sorted(cfg.items())), "fixture was generated for another config"
        P = PV.seeded_state(cfg)
        px = torch.from_numpy(g[f"{name}_pixels"])
        assert px.shape[0] == B
        ref = torch.from_numpy(g[f"{name}_last_hidden_state"])
        out = OVT.vit_forward(P, cfg, px)
        err = float((out - ref).norm() / ref.norm())
        assert err <= 1e-5, f"{name}: oracle/vit.py vs transformers.ViTModel fixture rel-L2 {err:.2e}"


def test_vit_oracle_against_live_transformers():
    """the same pin against the installed transformers (any version whose ViTModel loads the mapped keys); skipped where
    transformers is not importable"""
    pytest.importorskip("transformers")
    from oracle import vit as OVT
    PV = _vit_pin_cases()
    cfg, B = PV.CASES["tiny"]
    P = PV.seeded_state(cfg)
    px = torch.randn(B, 3, cfg["image_size"], cfg["image_size"], generator=torch.Generator().manual_seed(5))
    ref = PV.run_transformers(cfg, P, px)
    out = OVT.vit_forward(P, cfg, px)
    assert float((out - ref).norm() / ref.norm()) <= 1e-5


def test_config_edge_cases_of_from_pretrained():
    """a diffusers config.json may carry `attention_head_dim` as ONE int; scheduler options whose arithmetic the step
    kernels do not implement are refused instead of silently ignored (DDPMScheduler's own default is clip_sample=True)."""
    import diffute_amd as D
    u = D.UNet2DConditionModel(block_out_channels=(64, 64, 64, 64), attention_head_dim=1, cross_attention_dim=128)
    assert u.config.attention_head_dim == (1, 1, 1, 1)
    for bad in (dict(clip_sample=True), dict(thresholding=True), dict(variance_type="learned_range"),
                dict(timestep_spacing="trailing"), dict(prediction_type="sample"), dict(beta_schedule="squaredcos_cap_v2")):
        for cls in (D.DDPMScheduler, D.DDIMScheduler):
            with pytest.raises(NotImplementedError):
                cls(**bad)
    assert D.DDPMScheduler(clip_sample=False, variance_type="fixed_small").config.clip_sample is False


def test_initial_latents_are_the_seed0_cpu_draw():
    """P2 (app.ipynb:796-801): `randn_tensor(shape, generator=torch.manual_seed(0))` is a CPU draw; fixture captured once
    (tests/golden/p2_init_latents.npz).  Its head is the well-known manual_seed(0) sequence of torch's CPU generator."""
    from oracle import pipeline as OP
    g = np.load(os.path.join(GOLD, "p2_init_latents.npz"))
    x = OP.initial_latents((1, 4, 64, 64))
    assert np.array_equal(x.numpy(), g["latents"])
    assert np.allclose(g["latents"].reshape(-1)[:4], [-1.1258, -1.1524, -0.2506, -0.4339], atol=5e-5)
    # the product draws the same tensor when no init_latents are passed (diffute_amd/pipeline.py edit_latents)
    y = torch.randn((1, 4, 64, 64), generator=torch.manual_seed(0), dtype=torch.float32)
    assert torch.equal(x, y)


def test_from_pretrained_diffusers_directory_layout(tmp_path):
    """N5 (train_diffute_v1.py:628-635, app.ipynb:545-553): `X.from_pretrained(path, subfolder=...)` on a directory shaped
    like the public stable-diffusion-2-inpainting repo - the full config.json key sets (tests/golden/sd2_inpaint_layout/,
    null / default-valued extras included), a `.bin` (torch.save) state dict for the UNet and a legacy-named `.bin` for the VAE.
    Widths are shrunk so the files stay small; the key set is the public one."""
    import json
    import shutil
    import diffute_amd as D
    src = os.path.join(GOLD, "sd2_inpaint_layout")
    ucfg = json.load(open(os.path.join(src, "unet", "config.json")))
    vcfg = json.load(open(os.path.join(src, "vae", "config.json")))
    # the public widths validate as they are (constructing 866 M parameters is left to the GPU tests)
    assert ucfg["block_out_channels"] == [320, 640, 1280, 1280] and ucfg["attention_head_dim"] == [5, 10, 20, 20] and ucfg["in_channels"] == 9
    D.models._check_supported("UNet2DConditionModel", ucfg, D.models._UNET_ONLY_SUPPORTED, D.models._UNET_MUST_BE_NONE)
    ucfg.update(block_out_channels=[64, 128, 128, 128], attention_head_dim=[1, 2, 2, 2], cross_attention_dim=128)
    vcfg.update(block_out_channels=[64, 64, 128, 128], layers_per_block=1)
    root = tmp_path / "sd2-inp"
    for sub, cfg in (("unet", ucfg), ("vae", vcfg)):
        (root / sub).mkdir(parents=True)
        json.dump(cfg, open(root / sub / "config.json", "w"))
    shutil.copytree(os.path.join(src, "scheduler"), root / "scheduler")
    u0 = D.UNet2DConditionModel(**{k: v for k, v in ucfg.items() if not k.startswith("_")}, seed=5)
    sd = {k: v.clone() for k, v in u0.state_dict().items()}
    k_pi = "down_blocks.0.attentions.0.proj_in.weight"
    sd[k_pi] = sd[k_pi][:, :, None, None].clone()                 # an exporter that kept the 1x1-conv shape
    torch.save(sd, root / "unet" / "diffusion_pytorch_model.bin")
    v0 = D.AutoencoderKL(**{k: v for k, v in vcfg.items() if not k.startswith("_")}, seed=6)
    legacy = {}
    for k, t in v0.state_dict().items():                            # pre-0.15 attention names, 1x1-conv-shaped projections
        k2 = k.replace("to_q", "query").replace("to_k", "key").replace("to_v", "value").replace("to_out.0", "proj_attn")
        legacy[k2] = t[:, :, None, None].clone() if (k2 != k and t.ndim == 2) else t.clone()
    torch.save(legacy, root / "vae" / "diffusion_pytorch_model.bin")
    u = D.UNet2DConditionModel.from_pretrained(str(root), subfolder="unet", revision=None)
    v = D.AutoencoderKL.from_pretrained(str(root), subfolder="vae", revision=None)
    s = D.DDPMScheduler.from_pretrained(str(root), subfolder="scheduler")
    assert all(torch.equal(a, b) for a, b in zip(u0.state_dict().values(), u.state_dict().values()))
    assert all(torch.equal(a, b) for a, b in zip(v0.state_dict().values(), v.state_dict().values()))
    assert u.config.use_linear_projection is True and u.config.sample_size == 64 and u.config.norm_eps == 1e-5
    assert v.config.scaling_factor == 0.18215 and v.config.latent_channels == 4 and tuple(v.config.block_out_channels) == (64, 64, 128, 128)
    assert s.config.prediction_type == "epsilon" and s.num_train_timesteps == 1000 and s.config.steps_offset == 1
    # unsupported variants are refused, not mis-run
    for bad in (dict(use_linear_projection=False), dict(act_fn="gelu"), dict(class_embed_type="timestep"), dict(resnet_time_scale_shift="scale_shift"),
                dict(down_block_types=["DownBlock2D", "AttnDownBlock2D", "DownBlock2D", "DownBlock2D"])):
        with pytest.raises(NotImplementedError):
            D.UNet2DConditionModel(**{**{k: v for k, v in ucfg.items() if not k.startswith("_")}, **bad})
    # save_pretrained -> from_pretrained (safetensors) keeps the extra keys
    u.save_pretrained(str(tmp_path / "out" / "unet"))
    u2 = D.UNet2DConditionModel.from_pretrained(str(tmp_path / "out"), subfolder="unet")
    assert u2.config.mid_block_type == "UNetMidBlock2DCrossAttn" and all(torch.equal(a, b) for a, b in zip(u.state_dict().values(), u2.state_dict().values()))


def test_grad_scaler_follows_torch_amp_contract():
    """diffute_amd.GradScaler (the loss scaling of `--mixed_precision fp16`, train_diffute_v1.py:267,583) against torch.amp.GradScaler's rules on a CPU
    torch optimizer: scale / unscale_ / skipped step on inf / backoff / growth after growth_interval clean steps / state_dict, and the guard
    against a double unscale_."""
    import pytest
    import torch
    import diffute_amd as D
    w = torch.nn.Parameter(torch.tensor([1.0, -2.0, 3.0]))
    opt = torch.optim.SGD([w], lr=0.5)
    sc = D.GradScaler(init_scale=8.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2)
    assert sc.get_scale() == 8.0 and sc.is_enabled()
    with pytest.raises(RuntimeError):
        sc.update()                                                  # no inf check recorded yet (torch: the same error)
    # a clean step: gradient of sum(w * c) is c, scaled by 8 in .grad, unscaled before the step
    c = torch.tensor([0.5, 0.25, -1.0])
    sc.scale((w * c).sum()).backward()
    assert torch.equal(w.grad, 8.0 * c)
    sc.unscale_(opt)
    assert torch.equal(w.grad, c)
    with pytest.raises(RuntimeError):
        sc.unscale_(opt)
    sc.step(opt); sc.update(); opt.zero_grad()
    assert torch.allclose(w.detach(), torch.tensor([1.0, -2.0, 3.0]) - 0.5 * c) and sc.get_scale() == 8.0
    # an overflowed gradient: the step is skipped, the scale halves, the growth counter restarts
    before = w.detach().clone()
    sc.scale((w * torch.tensor([float("inf"), 1.0, 1.0])).sum()).backward()
    assert sc.step(opt) is None
    sc.update(); opt.zero_grad()
    assert torch.equal(w.detach(), before) and sc.get_scale() == 4.0
    # growth_interval = 2 clean steps double it
    for i in range(2):
        sc.scale((w * c).sum()).backward(); sc.step(opt); sc.update(); opt.zero_grad()
        assert sc.get_scale() == (4.0 if i == 0 else 8.0)
    st = sc.state_dict()
    sc2 = D.GradScaler(); sc2.load_state_dict(st)
    assert sc2.get_scale() == 8.0 and sc2.state_dict() == st
    sc.update(new_scale=128.0)
    assert sc.get_scale() == 128.0
    # disabled: a pass-through (mixed_precision "no" / "bf16")
    off = D.GradScaler(enabled=False)
    loss = (w * c).sum()
    assert off.scale(loss) is loss and off.get_scale() == 1.0 and off.state_dict() == {}
    loss.backward(); off.unscale_(opt); off.step(opt); off.update()
