"""Model-level parity on the GPU: UNet / VAE / scheduler / denoise loop through the product classes
(which call the C-ABI) vs the oracle and the committed golden vectors."""
import os

import numpy as np
import pytest
import torch

from util import assert_close, rel_l2

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# End-to-end bound vs the bf16-emulating oracle.  Two bf16 pipelines that differ only in fp32 accumulation order
# decorrelate to ~1 bf16 ulp (2^-8) of noise per materialised tensor after a few layers (a 1e-6 difference flips a
# rounding, the flip is a 4e-3 difference one layer later); over the ~100 rounding points of a UNet/VAE forward that
# random-walks to ~1e-2.  Per-kernel tests (test_ops_gpu.py) hold the 1e-3 bar; whole-model bounds are stated here.
E2E_EMU = 2.5e-2
TINY_UNET = dict(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4), cross_attention_dim=128)
TINY_VAE = dict(block_out_channels=(64, 128, 128, 128), layers_per_block=1)


@pytest.fixture(scope="module")
def tiny_unet(cuda):
    import diffute_amd as D
    return D.UNet2DConditionModel(**TINY_UNET).cuda().requires_grad_(False)


@pytest.fixture(scope="module")
def tiny_vae(cuda):
    import diffute_amd as D
    return D.AutoencoderKL(**TINY_VAE).cuda().requires_grad_(False)


def test_scheduler_steps_bit_exact(cuda):
    """S3/S4: the elementwise updates are bit-identical to the oracle's fp32 evaluation given the same scalar
    coefficients; the coefficients themselves (diffusers' `x ** 0.5` on 0-d tensors) agree with the oracle's
    sqrt restatement to 1 ulp (pow vs sqrt is libm-defined in the last bit)."""
    import diffute_amd as D
    from oracle import schedulers as OS
    g = np.load(os.path.join(GOLD, "sched.npz"))
    x = torch.from_numpy(g["x"]).to(cuda); e = torch.from_numpy(g["eps"]).to(cuda); nz = torch.from_numpy(g["noise"]).to(cuda)
    ddim = D.DDIMScheduler(); ddim.set_timesteps(50)
    ddpm = D.DDPMScheduler(); ddpm.set_timesteps(50)
    ac = ddim.alphas_cumprod.numpy()
    for t in (981, 501, 1):
        c = ddim.step_coefficients(t)
        assert np.allclose(c, OS.ddim_coefs(ac, t, 50), rtol=2e-7, atol=0)
        out = ddim.step(e, torch.tensor(t), x).prev_sample.cpu().numpy()
        assert np.array_equal(out, OS.ddim_apply(c, g["eps"], g["x"])), f"ddim step t={t}"
    for t in (980, 500, 0):
        c = ddpm.step_coefficients(t)
        assert np.allclose(c, OS.ddpm_coefs(ac, t, 50), rtol=2e-7, atol=0)
        out = ddpm.step(e, torch.tensor(t), x, variance_noise=nz).prev_sample.cpu().numpy()
        assert np.array_equal(out, OS.ddpm_apply(c, g["eps"], g["x"], g["noise"] if t > 0 else None)), f"ddpm step t={t}"
    # v-prediction branch
    dv = D.DDIMScheduler(prediction_type="v_prediction"); dv.set_timesteps(50)
    c = dv.step_coefficients(501)
    assert np.array_equal(dv.step(e, 501, x).prev_sample.cpu().numpy(), OS.ddim_apply(c, g["eps"], g["x"], None, "v_prediction"))
    # add_noise / get_velocity (train_diffute_v1.py:897,907): same gathered coefficients -> bit-identical
    ts = torch.tensor([500])
    sa = (ddpm.alphas_cumprod ** 0.5)[500].numpy(); sb = ((1 - ddpm.alphas_cumprod) ** 0.5)[500].numpy()
    an = ddpm.add_noise(x, nz, ts).cpu().numpy(); ve = ddpm.get_velocity(x, nz, ts).cpu().numpy()
    assert np.array_equal(an, (sa * g["x"] + sb * g["noise"]).astype(np.float32))
    assert np.array_equal(ve, (sa * g["noise"] - sb * g["x"]).astype(np.float32))
    assert np.allclose(an, g["add_noise"], rtol=1e-6, atol=1e-7) and np.allclose(ve, g["velocity"], rtol=1e-6, atol=1e-7)
    # against the committed golden (table differs from torch.linspace's SIMD path by <= 1 ulp)
    assert np.allclose(ddim.step(e, 981, x).prev_sample.cpu().numpy(), g["ddim_step_981_50"], rtol=1e-5, atol=1e-6)


def test_tiny_unet_forward(cuda, tiny_unet):
    """T1 on the tiny config: vs the bf16-emulating oracle (tight) and the fp32 oracle (bf16 budget)."""
    from diffute_amd.synthetic import synth_inputs
    g = np.load(os.path.join(GOLD, "tiny_unet.npz"))
    lat, mask, mlat, ctx = synth_inputs(2, 16, 16, 77, 128, device=cuda)
    x = torch.cat([lat, mask, mlat], 1)
    with torch.no_grad():
        y = tiny_unet(x, torch.tensor(981), ctx).sample
    e16 = assert_close(y, torch.from_numpy(g["eps_bf16emu"]), E2E_EMU, "tiny unet vs bf16-emulating oracle")
    e32 = assert_close(y, torch.from_numpy(g["eps_fp32"]), 5e-2, "tiny unet vs fp32 oracle")
    print(f"tiny unet rel-L2: vs bf16emu {e16:.2e}, vs fp32 {e32:.2e}")
    with torch.no_grad():
        yb = tiny_unet(x, torch.tensor([981, 3]), ctx).sample          # per-sample timesteps (train_diffute_v1.py:892-893)
    assert_close(yb, torch.from_numpy(g["eps_bf16emu_tvec"]), E2E_EMU, "tiny unet, LongTensor[B] timesteps")
    # forward_parts (fused concat) == concatenated input, bit for bit
    t = torch.tensor([981], device=cuda)
    yp = tiny_unet.forward_parts([lat, mask, mlat], t)
    assert torch.equal(yp, y)
    # determinism
    assert torch.equal(tiny_unet.forward_parts([lat, mask, mlat], t), yp)


def test_tiny_unet_nonsquare_odd_batch(cuda, tiny_unet):
    """shapes the goldens do not cover: batch 3, an 8 x 24 latent grid (64 x 192 px) and a 40-token context - every
    gather / GroupNorm slab / upsample-phase / attention tail path away from the square power-of-two case, against the
    oracle run here"""
    from oracle import unet as OU
    from diffute_amd.synthetic import synth_inputs
    lat, mask, mlat, ctx = synth_inputs(3, 8, 24, 40, 128, device=cuda, seed=11)
    x = torch.cat([lat, mask, mlat], 1)
    t = torch.tensor([981, 500, 3])
    with torch.no_grad():
        y = tiny_unet(x, t.to(cuda), ctx).sample
    P = {k: v.detach().cpu().float() for k, v in tiny_unet.state_dict().items()}
    ref = OU.unet_forward(P, OU.TINY_UNET, x.cpu().float(), t, ctx.cpu().float(), emulate_bf16=True)
    assert y.shape == (3, 4, 8, 24)
    assert_close(y, ref, E2E_EMU, "tiny unet, B=3, 8x24 latents, 40 context tokens")
    with torch.no_grad():                                   # sample 1 alone == sample 1 of the batch (batch independence)
        y1 = tiny_unet(x[1:2].contiguous(), t[1:2].to(cuda), ctx[1:2].contiguous()).sample
    assert rel_l2(y1, y[1:2]) < 2e-2


def test_tiny_unet_on_the_160_column_tiles(cuda):
    """the 128x160 / 128x320 tile instances with their own epilogue (folded LayerNorm + GEGLU, bias + row bias + residual)
    inside a whole UNet: every eligible GEMM signature of the tiny model is pinned to them through the run-time plan
    override and the result is held to the same bound against the golden as the default plans"""
    import diffute_amd as D
    from diffute_amd import _cabi
    from diffute_amd.synthetic import synth_inputs
    g = np.load(os.path.join(GOLD, "tiny_unet.npz"))
    lat, mask, mlat, ctx = synth_inputs(2, 16, 16, 77, 128, device=cuda)
    x = torch.cat([lat, mask, mlat], 1)
    lib = _cabi.lib()
    unet = D.UNet2DConditionModel(**TINY_UNET).cuda().requires_grad_(False)
    ref = None
    try:
        with torch.no_grad():
            ref = unet(x, torch.tensor(981), ctx).sample.clone()
        for (B_, hw, C) in ((2, 256, 64), (2, 64, 128), (2, 16, 256), (2, 4, 256)):
            M = B_ * hw
            lib.dmx_gemm_plan_override(M, 8 * C, C, 1, 0, 11, 1)            # FF1: folded LayerNorm + GEGLU on the 128x320 tile
            for (N, K) in ((C, C), (C, 4 * C), (3 * C, C), (C, 9 * C), (C, 18 * C), (C, 9 * C + C)):
                lib.dmx_gemm_plan_override(M, N, K, 1, 0, 10, 1)            # linears / convolutions of that level on the 128x160 tile
        unet2 = D.UNet2DConditionModel(**TINY_UNET).cuda().requires_grad_(False)   # fresh handle: no captured graphs, new workspace query
        with torch.no_grad():
            y = unet2(x, torch.tensor(981), ctx).sample
    finally:
        lib.dmx_gemm_plan_override(0, 0, 0, 0, 0, -1, 0)
    assert_close(y, torch.from_numpy(g["eps_bf16emu"]), E2E_EMU, "tiny unet on the 160-column tiles vs bf16-emulating oracle")
    assert rel_l2(y, ref) < 2e-2 and not torch.equal(y, ref), "the override must have changed at least one GEMM's tile plan"


def test_weight_prefetch_plan_changes_no_result(cuda):
    """The weight prefetch plan (every launch of dmx_unet_forward* touches the weights of the launches that follow it, Exec::note / peek)
    only moves data into the memory-side cache: eps with the plan on and off is bit-identical - at the tiny configuration and at a
    full-size one (256 px, every kernel family of the headline pass: chains off at this size, fused GroupNorm -> conv on), eager and
    through the captured graph.  The dry walk and the real walk must also agree on the launch list (the library checks that itself)."""
    import diffute_amd as D
    from diffute_amd import _cabi
    from diffute_amd.synthetic import synth_inputs
    lib = _cabi.lib()
    cases = [(D.UNet2DConditionModel(**TINY_UNET).cuda().requires_grad_(False), synth_inputs(2, 16, 32, 40, 128, device=cuda, seed=7)),
             (D.UNet2DConditionModel(device=cuda).requires_grad_(False), synth_inputs(2, 32, 32, 577, 1024, device=cuda, seed=8))]
    try:
        for unet, (lat, mask, mlat, ctx) in cases:
            t = torch.tensor([437], device=cuda)
            outs = {}
            for on in (1, 0, 1):
                lib.dmx_set_weight_prefetch(on)
                unet.set_context(ctx)
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    e = unet.forward_parts([lat, mask, mlat], t).clone()
                    g = unet.forward_parts([lat, mask, mlat], t, graph=True).clone()
                    _cabi.check(unet._lib.dmx_unet_refresh_derived(unet._h, None), "refresh")      # drop the captured graph: it embeds the setting
                torch.cuda.synchronize()
                assert torch.equal(e, g), "graph replay differs from the eager launch"
                outs.setdefault(on, []).append(e)
            assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][0], outs[1][1]), "the prefetch plan changed a result"
    finally:
        lib.dmx_set_weight_prefetch(1)


def test_fused_groupnorm_conv_inside_the_models(cuda):
    """conv_halo.hip inside whole models: the executors fuse GroupNorm -> SiLU -> conv3x3 of every ResnetBlock2D only at the levels
    where it pays at the bench batch (>= 32 x 32 pixels), which the tiny configurations barely reach - so here the fused launch is
    forced wherever the kernel takes the problem (dmx_set_halo_conv(2): 64 / 128-column tiles, 16 x 16 and 8 x 32 tile geometries, K
    splits, statistics records from convs, from the transformer's last GEMM and from dmx_colstats) and held to the oracle like the
    unfused path, for the tiny UNet (16 x 32 latents) and the tiny autoencoder (32 x 64 px)."""
    import diffute_amd as D
    from diffute_amd import _cabi
    from diffute_amd.synthetic import synth_inputs
    from oracle import unet as OU, vae as OV
    lib = _cabi.lib()
    lat, mask, mlat, ctx = synth_inputs(1, 16, 32, 40, 128, device=cuda, seed=5)
    x = torch.cat([lat, mask, mlat], 1)
    img = torch.rand(1, 3, 32, 64, device=cuda) * 2 - 1
    outs = {}
    try:
        for mode in (0, 2):
            lib.dmx_set_halo_conv(mode)
            torch.manual_seed(3)
            unet = D.UNet2DConditionModel(**TINY_UNET).cuda().requires_grad_(False)
            vae = D.AutoencoderKL(block_out_channels=(64, 128, 128, 128), layers_per_block=1).cuda().requires_grad_(False)
            with torch.no_grad():
                eps = unet(x, torch.tensor(500), ctx).sample
                z = vae.encode(img).latent_dist.mode()
                rec = vae.decode(z).sample
            outs[mode] = (eps, z, rec, unet, vae)
    finally:
        lib.dmx_set_halo_conv(1)
    eps0, z0, rec0, unet, vae = outs[0]
    eps2, z2, rec2, _, _ = outs[2]
    P = {k: v.detach().cpu().float() for k, v in unet.state_dict().items()}
    ref = OU.unet_forward(P, OU.TINY_UNET, x.cpu().float(), torch.tensor(500), ctx.cpu().float(), emulate_bf16=True)
    assert_close(eps2, ref, E2E_EMU, "tiny unet, fused GroupNorm -> conv everywhere vs bf16-emulating oracle")
    assert_close(eps0, ref, E2E_EMU, "tiny unet, unfused vs bf16-emulating oracle")
    assert rel_l2(eps2, eps0) < 2e-2 and not torch.equal(eps2, eps0), "fused vs unfused: two bf16 roundings apart, and the switch must have changed the launch sequence"
    PV = {k: v.detach().cpu().float() for k, v in vae.state_dict().items()}
    zref = OV.vae_encode_moments(PV, OV.TINY_VAE, img.cpu().float(), emulate_bf16=True)[:, :4] if hasattr(OV, "vae_encode_moments") else None
    if zref is not None:
        assert_close(z2, zref, E2E_EMU, "tiny vae encode, fused vs oracle")
    rref = OV.vae_decode(PV, OV.TINY_VAE, z2.cpu().float(), emulate_bf16=True)
    assert_close(rec2, rref, E2E_EMU, "tiny vae decode, fused GroupNorm -> conv everywhere vs bf16-emulating oracle")
    assert rel_l2(z2, z0) < 2e-2 and rel_l2(rec2, rec0) < 4e-2          # (the two decodes start from their own, slightly different latents)


def test_tiny_vae(cuda, tiny_vae):
    from diffute_amd.synthetic import synth_images
    from diffute_amd.init import normal
    g = np.load(os.path.join(GOLD, "tiny_vae.npz"))
    img = synth_images(2, 64, 64, device=cuda)
    with torch.no_grad():
        dist = tiny_vae.encode(img).latent_dist
        assert_close(dist.parameters, torch.from_numpy(g["moments_bf16emu"]), E2E_EMU, "tiny vae moments vs bf16emu")
        assert_close(dist.parameters, torch.from_numpy(g["moments_fp32"]), 5e-2, "tiny vae moments vs fp32")
        z = normal(5, 22, 2 * 4 * 8 * 8, cuda).reshape(2, 4, 8, 8)
        d = tiny_vae.decode(z).sample
        assert_close(d, torch.from_numpy(g["image_bf16emu"]), E2E_EMU, "tiny vae decode vs bf16emu")
        assert_close(d, torch.from_numpy(g["image_fp32"]), 5e-2, "tiny vae decode vs fp32")
        # latent_dist.sample() with injected noise == mean + exp(0.5*clamp(logvar))*noise (oracle)
        from oracle.vae import gaussian_sample
        nz = normal(9, 23, 2 * 4 * 8 * 8, cuda).reshape(2, 4, 8, 8)
        s = dist.sample(noise=nz)
        ref = gaussian_sample(dist.parameters.cpu(), nz.cpu())
        assert float((s.cpu() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
        assert torch.equal(dist.mode(), dist.parameters[:, :4])
        # vae(x)["sample"] (train_vae.py:721): decode(encode(x).mode())
        rec = tiny_vae(img)["sample"]
        assert rec.shape == img.shape and torch.isfinite(rec).all()


def test_tiny_vae_nonsquare(cuda, tiny_vae):
    """64 x 96 px, batch 3: the asymmetric stride-2 pad, the phase-decomposed decoder upsamplers and the d=C attention on a
    non-square grid, against the oracle run here"""
    from oracle import vae as OV
    from diffute_amd.synthetic import synth_images
    from diffute_amd.init import normal
    img = synth_images(3, 64, 96, device=cuda)
    P = {k: v.detach().cpu().float() for k, v in tiny_vae.state_dict().items()}
    with torch.no_grad():
        mom = tiny_vae.encode(img).latent_dist.parameters
        z = normal(6, 24, 3 * 4 * 8 * 12, cuda).reshape(3, 4, 8, 12)
        dec = tiny_vae.decode(z).sample
    assert mom.shape == (3, 8, 8, 12) and dec.shape == (3, 3, 64, 96)
    assert_close(mom, OV.vae_encode_moments(P, OV.TINY_VAE, img.cpu(), emulate_bf16=True), E2E_EMU, "tiny vae encode 64x96")
    assert_close(dec, OV.vae_decode(P, OV.TINY_VAE, z.cpu(), emulate_bf16=True), E2E_EMU, "tiny vae decode 64x96")


def test_tiny_denoise_loops(cuda, tiny_unet):
    """P1: the 4-step loop (app.ipynb:796-816) with DDIM and with DDPM + injected variance noise."""
    import diffute_amd as D
    from diffute_amd.init import normal
    from diffute_amd.synthetic import synth_inputs
    g = np.load(os.path.join(GOLD, "tiny_loop.npz"))
    lat, mask, mlat, ctx = synth_inputs(2, 16, 16, 77, 128, device=cuda)
    out = D.denoise(tiny_unet, D.DDIMScheduler(), lat, mask, mlat, ctx, 4)
    assert_close(out, torch.from_numpy(g["ddim_bf16emu"]), 2e-2, "tiny DDIM loop vs bf16emu")
    assert_close(out, torch.from_numpy(g["ddim_fp32"]), 5e-2, "tiny DDIM loop vs fp32")
    nz = normal(3, 31, 4 * 2 * 4 * 16 * 16, cuda).reshape(4, 2, 4, 16, 16)
    out2 = D.denoise(tiny_unet, D.DDPMScheduler(), lat, mask, mlat, ctx, 4, variance_noise=nz)
    assert_close(out2, torch.from_numpy(g["ddpm_bf16emu"]), 2e-2, "tiny DDPM loop vs bf16emu")
    # the reference-shaped loop (cat / unet(...).sample / scheduler.step(...).prev_sample) gives the same latents
    sch = D.DDIMScheduler(); sch.set_timesteps(4)
    x = lat * sch.init_noise_sigma
    with torch.no_grad():
        for t in sch.timesteps:
            inp = torch.cat([sch.scale_model_input(x, t), mask, mlat], dim=1)
            eps = tiny_unet(inp, t, ctx).sample
            x = sch.step(eps, t, x).prev_sample
    assert torch.equal(x, out)
    # two independent chains of the batch on two streams (each with its own graph, step index and slot of the context cache)
    out_mb = D.denoise(tiny_unet, D.DDIMScheduler(), lat, mask, mlat, ctx, 4, micro_batches=2)
    assert_close(out_mb, out.cpu(), 1e-2, "micro-batched DDIM loop vs the single chain")


def test_cfg1_full_golden(cuda):
    """BASELINE config 1 on the GPU: full SD2-inpaint UNet, B=1, 256 px, 10 DDIM steps, seeded weights,
    against the committed oracle output (fp32 oracle and bf16-emulating oracle)."""
    import diffute_amd as D
    from diffute_amd.synthetic import synth_inputs
    path = os.path.join(GOLD, "cfg1_full.npz")
    if not os.path.exists(path):
        pytest.skip("cfg1_full.npz not generated")
    g = np.load(path)
    unet = D.UNet2DConditionModel(device=cuda).requires_grad_(False)
    assert sum(p.numel() for p in unet.parameters()) == 865_925_124
    lat, mask, mlat, ctx = synth_inputs(1, 32, 32, 577, 1024, device=cuda)
    trace = []
    out = D.denoise(unet, D.DDIMScheduler(), lat, mask, mlat, ctx, 10,
                    callback=lambda i, t, x, eps: trace.append((t, eps.clone())) if i == 0 else None)
    e0 = assert_close(trace[0][1], torch.from_numpy(g["eps0_bf16emu"]), E2E_EMU, "cfg1 first-step eps vs bf16emu")
    e1 = assert_close(out, torch.from_numpy(g["final_bf16emu"]), 3e-2, "cfg1 final latents vs bf16emu")
    e2 = assert_close(out, torch.from_numpy(g["final_fp32"]), 5e-2, "cfg1 final latents vs fp32 oracle")
    print(f"cfg1 rel-L2: eps0 {e0:.2e}, final vs bf16emu {e1:.2e}, final vs fp32 {e2:.2e}")


def test_cfg2_headline_workload_vs_golden(cuda):
    """The workload bench.py times (BASELINE configs[1]: 512 px, 50 DDIM steps, batch 4, bf16) checked on its RESULT: sample 0
    of the B=4 `denoise()` against the committed oracle run of that sample (tests/golden/cfg2_b1.npz from
    scripts/make_golden.py --cfg2: eps at steps 0 / 25 / 49 and the final latents, fp32 oracle and bf16-emulating oracle), and
    the reference's own scheduler - DDPM, 50 steps, injected variance noise (app.ipynb:545,806-816) - likewise.
    Bounds: final latents <= 2e-2 vs the fp32 oracle, first-step eps <= 2.5e-2 vs the bf16-emulating oracle
    (one forward); later eps are compared on THEIR OWN trajectories, which have drifted apart by then - hence the wider bound."""
    import diffute_amd as D
    from diffute_amd.init import normal
    from diffute_amd.synthetic import synth_inputs
    path = os.path.join(GOLD, "cfg2_b1.npz")
    if not os.path.exists(path):
        pytest.skip("cfg2_b1.npz not generated")
    g = np.load(path)
    unet = D.UNet2DConditionModel(device=cuda).requires_grad_(False)
    lat, mask, mlat, ctx = synth_inputs(4, 64, 64, 577, 1024, device=cuda)
    for sched, cls, nz in (("ddim", D.DDIMScheduler, None),
                           ("ddpm", D.DDPMScheduler, normal(3, 31, 50 * 4 * 4 * 64 * 64, cuda).reshape(50, 4, 4, 64, 64))):
        if f"final_{sched}_fp32" not in g.files:
            continue
        trace = {}
        out = D.denoise(unet, cls(), lat, mask, mlat, ctx, 50, variance_noise=nz,
                        callback=lambda i, t, x, eps: trace.__setitem__(i, (t, eps[:1].clone(), x[:1].clone())) if i in (0, 25, 49) else None)
        assert [trace[i][0] for i in (0, 25, 49)] == [int(g[f"timesteps_{sched}"][i]) for i in (0, 25, 49)]
        e0 = assert_close(trace[0][1], torch.from_numpy(g[f"eps0_{sched}_bf16emu"]), E2E_EMU, f"cfg2 {sched}: first-step eps vs bf16emu oracle")
        e0f = assert_close(trace[0][1], torch.from_numpy(g[f"eps0_{sched}_fp32"]), 5e-2, f"cfg2 {sched}: first-step eps vs fp32 oracle")
        e25 = assert_close(trace[25][1], torch.from_numpy(g[f"eps25_{sched}_fp32"]), 8e-2, f"cfg2 {sched}: eps at step 25 vs fp32 oracle")
        e49 = assert_close(trace[49][1], torch.from_numpy(g[f"eps49_{sched}_fp32"]), 8e-2, f"cfg2 {sched}: eps at step 49 vs fp32 oracle")
        x25 = assert_close(trace[25][2], torch.from_numpy(g[f"x25_{sched}_fp32"]), 2e-2, f"cfg2 {sched}: latents after step 25 vs fp32 oracle")
        ef = assert_close(out[:1], torch.from_numpy(g[f"final_{sched}_fp32"]), 2e-2, f"cfg2 {sched}: final latents (50 steps) vs fp32 oracle")
        eb = assert_close(out[:1], torch.from_numpy(g[f"final_{sched}_bf16emu"]), 2e-2, f"cfg2 {sched}: final latents (50 steps) vs bf16emu oracle")
        print(f"cfg2 headline workload, sample 0 of B=4, 50 {sched.upper()} steps: eps0 {e0:.2e} (bf16emu) {e0f:.2e} (fp32), eps25 {e25:.2e}, eps49 {e49:.2e}, "
              f"x25 {x25:.2e}, final latents {ef:.2e} (fp32 oracle) {eb:.2e} (bf16emu oracle); "
              f"oracle bf16emu vs fp32 final {rel_l2(torch.from_numpy(g[f'final_{sched}_bf16emu']), torch.from_numpy(g[f'final_{sched}_fp32'])):.2e}")
        assert torch.isfinite(out).all()


@pytest.mark.parametrize("chains", [False, True], ids=["separate_gemms", "transformer_chains"])
def test_cfg1_full_size_block_taps(cuda, chains):
    """per-block checks at FULL size (SD2-inpaint config, cfg1's first UNet call): L2 norm, sum and a strided slice of every
    block output - conv_in, down0..3, mid, up0..3 - against the committed oracle values (tests/golden/cfg1_taps.npz).
    Run on both forms of the C = 320 transformer blocks: the separate GEMMs (what the executor picks at this size) and the
    chained kernels of xf_chain.hip forced on (what it picks from 192 row blocks up, e.g. the batch-4 512-px bench)."""
    import diffute_amd as D
    from diffute_amd import _cabi
    from diffute_amd.synthetic import synth_inputs
    path = os.path.join(GOLD, "cfg1_taps.npz")
    if not os.path.exists(path):
        pytest.skip("cfg1_taps.npz not generated")
    g = np.load(path)
    unet = D.UNet2DConditionModel(device=cuda).requires_grad_(False)
    lat, mask, mlat, ctx = synth_inputs(1, 32, 32, 577, 1024, device=cuda)
    old = _cabi.lib().dmx_set_xf_chain(2 if chains else 0)
    try:
        y, taps = unet.forward_taps(torch.cat([lat, mask, mlat], 1), torch.tensor(int(g["timestep"])), ctx)
    finally:
        _cabi.lib().dmx_set_xf_chain(old)
    rep = []
    for k, v in taps.items():
        v = v.float().cpu()
        sl = v.flatten()[::97][:4096]
        tol = 2e-3 if k == "conv_in" else 2.5e-2
        e = assert_close(sl, torch.from_numpy(g[f"slice_bf16emu_{k}"]), tol, f"cfg1 block {k}: strided slice vs bf16emu oracle")
        ef = assert_close(sl, torch.from_numpy(g[f"slice_fp32_{k}"]), 2 * tol, f"cfg1 block {k}: strided slice vs fp32 oracle")
        l2 = float(v.double().pow(2).sum().sqrt())
        assert abs(l2 / float(g[f"l2_bf16emu_{k}"]) - 1) < 5e-3, f"cfg1 block {k}: L2 norm {l2} vs {float(g[f'l2_bf16emu_{k}'])}"
        sm = float(v.double().sum())
        # per-element errors are ~1e-2 relative; a part of them is common-mode inside a channel (GroupNorm statistics, biases), so
        # the sum is held to 2e-3 of the absolute sum (measured worst: up1, 153 of 1.0e6)
        assert abs(sm - float(g[f"sum_bf16emu_{k}"])) <= 2e-3 * float(g[f"abs_bf16emu_{k}"]), f"cfg1 block {k}: sum {sm} vs {float(g[f'sum_bf16emu_{k}'])}"
        rep.append(f"{k} {e:.1e}/{ef:.1e}")
    print("cfg1 full-size per-block slices, rel-L2 vs bf16emu / fp32 oracle: " + ", ".join(rep))


def test_full_size_properties(cuda):
    """cfg2-sized call (B=4, 512 px): size-independent properties - finite output, batch independence
    (sample i of a batch == the same sample run alone), determinism."""
    import diffute_amd as D
    from diffute_amd.synthetic import synth_inputs
    unet = D.UNet2DConditionModel(device=cuda).requires_grad_(False)
    lat, mask, mlat, ctx = synth_inputs(4, 64, 64, 577, 1024, device=cuda)
    t = torch.tensor([981], device=cuda)
    unet.set_context(ctx)
    y = unet.forward_parts([lat, mask, mlat], t).clone()
    assert torch.isfinite(y).all() and float(y.std()) > 1e-3
    y2 = unet.forward_parts([lat, mask, mlat], t)
    assert torch.equal(y, y2)
    unet.set_context(ctx[2:3].contiguous())
    y1 = unet.forward_parts([lat[2:3].contiguous(), mask[2:3].contiguous(), mlat[2:3].contiguous()], t)
    assert rel_l2(y1, y[2:3]) < 2e-2        # different split-K / tile choices at B=1 change rounding only


# ------------------------------------------------------------------------------------------------ training (P5 / P6)
def _oracle_train_grads(model, cfg_oracle, x, t, ctx, target, emulate_bf16=False):
    """loss and parameter gradients of the oracle (torch autograd on the CPU restatement; with emulate_bf16 the forward
    rounds where the HIP path materialises bf16, the backward stays fp32 - casts are straight-through)"""
    from oracle import unet as OU
    P = {k: v.detach().cpu().float().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    with torch.enable_grad():
        pred = OU.unet_forward.__wrapped__(P, cfg_oracle, x.cpu(), t.cpu(), ctx.cpu(), emulate_bf16=emulate_bf16)
        loss = torch.mean((pred.float() - target.cpu().float()) ** 2)
        loss.backward()
    return float(loss.detach()), pred.detach(), {k: p.grad for k, p in P.items()}


def test_tiny_unet_train_step(cuda):
    """P5/P6: one training step of the tiny UNet through the product classes (HIP forward that keeps activations, HIP
    MSE loss, hand-written HIP backward) vs torch autograd on the oracle.  Tolerances (bf16 activations AND bf16
    activation gradients against an fp32 backward): loss within 2 %, the whole gradient vector within 4e-2 rel-L2 of the
    fp32 oracle and of the bf16-emulating oracle (measured 2.5e-2 / 2.4e-2), no single parameter worse than 1e-1 (5e-2)."""
    import diffute_amd as D
    from diffute_amd.models import mse_loss
    from diffute_amd.synthetic import synth_inputs
    from oracle import unet as OU
    model = D.UNet2DConditionModel(**TINY_UNET).cuda()
    lat, mask, mlat, ctx = synth_inputs(2, 16, 16, 77, 128, device=cuda)
    x = torch.cat([lat, mask, mlat], 1)
    t = torch.tensor([981, 17], device=cuda)
    g = torch.Generator().manual_seed(11)
    target = torch.randn(2, 4, 16, 16, generator=g).to(cuda)
    pred = model(x, t, ctx).sample
    loss = mse_loss(pred, target)
    loss.backward()
    for em in (False, True):
        ref_loss, ref_pred, ref_g = _oracle_train_grads(model, OU.TINY_UNET, x, t, ctx, target, emulate_bf16=em)
        assert_close(pred.detach(), ref_pred, 5e-2, "train forward vs oracle")
        assert abs(float(loss.detach()) - ref_loss) <= 2e-2 * abs(ref_loss), f"loss {float(loss)} vs oracle {ref_loss}"
        errs = []
        num = den = 0.0
        for k, p in model.named_parameters():
            assert p.grad is not None, f"no gradient for {k}"
            gh = p.grad.detach().float().cpu(); gr = ref_g[k]
            assert torch.isfinite(gh).all(), f"{k}: non-finite gradient"
            errs.append((rel_l2(gh, gr), k))
            num += float((gh - gr).pow(2).sum()); den += float(gr.pow(2).sum())
        tot = (num / den) ** 0.5
        errs.sort(reverse=True)
        print(f"tiny train step vs {'bf16-emulating' if em else 'fp32'} oracle: loss {float(loss):.6f} (oracle {ref_loss:.6f}); "
              f"whole-gradient rel-L2 {tot:.2e}; median {errs[len(errs) // 2][0]:.2e}; worst: " + ", ".join(f"{k} {e:.2e}" for e, k in errs[:4]))
        assert tot <= 4e-2, f"whole-gradient rel-L2 {tot:.3e}"
        assert errs[0][0] <= 1e-1, f"gradient of {errs[0][1]}: rel-L2 {errs[0][0]:.3e}"


def test_tiny_unet_train_step_fp16_build(cuda):
    """`--mixed_precision fp16` (train_diffute_v1.py:267,583,790): the fp16 build's training step - fp32 master parameters, fp16 compute copies,
    fp16 activations AND fp16 activation gradients - under a loss scale (diffute_amd.GradScaler.scale(loss).backward()), gradients unscaled and
    compared with torch autograd on the fp32 oracle.  Same bars as the bf16 build's test (4e-2 whole-gradient rel-L2, 1e-1 worst parameter;
    fp16 has three more mantissa bits, so it lands well inside).  WITHOUT the scale the small activation gradients go through fp16's subnormal range:
    finite, but measurably worse - which is why the reference's accelerate wraps the backward in a GradScaler."""
    import diffute_amd as D
    from diffute_amd.models import mse_loss
    from diffute_amd.synthetic import synth_inputs
    from oracle import unet as OU
    model = D.UNet2DConditionModel(**TINY_UNET).cuda().to(dtype=torch.float16)
    assert model.compute_dtype == torch.float16
    lat, mask, mlat, ctx = synth_inputs(2, 16, 16, 77, 128, device=cuda)
    x = torch.cat([lat, mask, mlat], 1)
    t = torch.tensor([981, 17], device=cuda)
    g = torch.Generator().manual_seed(11)
    target = torch.randn(2, 4, 16, 16, generator=g).to(cuda)
    ref_loss, ref_pred, ref_g = _oracle_train_grads(model, OU.TINY_UNET, x, t, ctx, target, emulate_bf16=False)
    res = {}
    for S in (1024.0, 1.0):
        scaler = D.GradScaler(init_scale=S)
        model.zero_grad(set_to_none=True)
        pred = model(x, t, ctx).sample
        loss = mse_loss(pred, target)
        scaler.scale(loss).backward()
        assert_close(pred.detach().float(), ref_pred, 5e-2, "fp16 train forward vs oracle")
        assert abs(float(loss.detach()) - ref_loss) <= 2e-2 * abs(ref_loss), f"loss {float(loss)} vs oracle {ref_loss}"
        errs = []
        num = den = 0.0
        for k, p in model.named_parameters():
            assert p.grad is not None and p.grad.dtype == torch.float32, f"no fp32 gradient for {k}"
            gh = p.grad.detach().float().cpu() / S; gr = ref_g[k]
            assert torch.isfinite(gh).all(), f"{k}: non-finite gradient at loss scale {S}"
            errs.append((rel_l2(gh, gr), k))
            num += float((gh - gr).pow(2).sum()); den += float(gr.pow(2).sum())
        tot = (num / den) ** 0.5
        errs.sort(reverse=True)
        res[S] = tot
        print(f"tiny fp16 train step, loss scale {S:g}, vs fp32 oracle: loss {float(loss):.6f} (oracle {ref_loss:.6f}); whole-gradient rel-L2 {tot:.2e}; "
              f"median {errs[len(errs) // 2][0]:.2e}; worst: " + ", ".join(f"{k} {e:.2e}" for e, k in errs[:3]))
        if S > 1.0:
            assert tot <= 4e-2, f"whole-gradient rel-L2 {tot:.3e}"
            assert errs[0][0] <= 1e-1, f"gradient of {errs[0][1]}: rel-L2 {errs[0][0]:.3e}"
    assert res[1024.0] <= res[1.0] * 1.05, f"the loss scale did not help: {res}"


def test_fp16_grad_scaler_with_fused_adamw(cuda):
    """GradScaler + FusedAdamW on the fp16 build (accelerate's fp16 loop: scale -> backward -> unscale_ -> clip -> step -> update):
    (a) a step at scale 2^10 lands where torch.optim.AdamW + clip_grad_norm_ puts a twin model fed the same (unscaled, exported) gradients;
    (b) an overflowing backward (scale 2^40: the scaled dL/dpred does not fit fp16) is SKIPPED - masters, moments, step count, compute copies
    untouched - and the scale backs off; (c) growth after `growth_interval` clean steps; (d) the torch-optimizer route through the same scaler."""
    import diffute_amd as D
    from diffute_amd.models import mse_loss
    from diffute_amd.synthetic import synth_inputs
    lat, mask, mlat, ctx = synth_inputs(1, 8, 8, 20, 128, device=cuda)
    x = torch.cat([lat, mask, mlat], 1); t = torch.tensor([500], device=cuda); target = torch.zeros(1, 4, 8, 8, device=cuda)
    hp = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    ref = D.UNet2DConditionModel(**TINY_UNET).cuda().to(dtype=torch.float16)
    opt_ref = torch.optim.AdamW(ref.parameters(), **hp)
    sc_ref = D.GradScaler(init_scale=1024.0, growth_interval=2)
    fus = D.UNet2DConditionModel(**TINY_UNET).cuda().to(dtype=torch.float16)
    opt_fus = D.FusedAdamW(fus, max_grad_norm=0.05, **hp)
    sc_fus = D.GradScaler(init_scale=1024.0, growth_interval=2)
    # (a) + (d)
    l_ref = mse_loss(ref(x, t, ctx).sample, target); sc_ref.scale(l_ref).backward()
    sc_ref.unscale_(opt_ref)
    gn_ref = torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.05)
    sc_ref.step(opt_ref); sc_ref.update(); opt_ref.zero_grad(set_to_none=True)
    l_fus = mse_loss(fus(x, t, ctx).sample, target); sc_fus.scale(l_fus).backward()
    sc_fus.unscale_(opt_fus); sc_fus.step(opt_fus); sc_fus.update()
    assert not opt_fus.found_inf and opt_fus.t == 1
    assert abs(float(l_ref.detach()) - float(l_fus.detach())) <= 1e-6 * abs(float(l_ref.detach()))
    assert abs(float(gn_ref) - float(opt_fus.grad_norm)) <= 1e-5 * float(gn_ref), f"unscaled gradient norm {float(opt_fus.grad_norm)} vs {float(gn_ref)}"
    sd_ref = {k: v.clone() for k, v in ref.state_dict().items()}; sd_fus = {k: v.clone() for k, v in fus.state_dict().items()}
    worst = max(float((sd_ref[k] - sd_fus[k]).abs().max() / (sd_ref[k].abs().max() + 1e-12)) for k in sd_ref)
    assert worst <= 2e-6, f"parameters after one scaled step differ: {worst:.2e}"
    # (b) overflow: skipped as a whole
    before = (opt_fus.masters.clone(), opt_fus.exp_avg.clone(), opt_fus.exp_avg_sq.clone())
    sc_fus.update(new_scale=2.0 ** 40)
    l2 = mse_loss(fus(x, t, ctx).sample, target); sc_fus.scale(l2).backward()
    sc_fus.step(opt_fus)
    assert opt_fus.found_inf and opt_fus.t == 1, "an overflowed gradient must skip the step"
    sc_fus.update()
    assert sc_fus.get_scale() == 2.0 ** 39
    for a, b, nm in zip(before, (opt_fus.masters, opt_fus.exp_avg, opt_fus.exp_avg_sq), ("masters", "exp_avg", "exp_avg_sq")):
        assert torch.equal(a, b), f"{nm} changed in a skipped step"
    l3 = mse_loss(fus(x, t, ctx).sample, target)
    assert torch.equal(l3.detach(), l2.detach()), "the compute copies changed in a skipped step"
    l3.backward()                                                   # (gradient dropped below)
    opt_fus.zero_grad()
    # the same overflow through the torch-optimizer route: skipped, scale halves
    sc_ref.update(new_scale=2.0 ** 40)
    p_before = {k: v.clone() for k, v in ref.state_dict().items()}
    sc_ref.scale(mse_loss(ref(x, t, ctx).sample, target)).backward()
    sc_ref.step(opt_ref); sc_ref.update(); opt_ref.zero_grad(set_to_none=True)
    assert sc_ref.get_scale() == 2.0 ** 39
    assert all(torch.equal(p_before[k], v) for k, v in ref.state_dict().items())
    # (c) growth: two clean steps at growth_interval = 2 double the scale
    sc_fus.update(new_scale=256.0)
    for i in range(2):
        sc_fus.scale(mse_loss(fus(x, t, ctx).sample, target)).backward()
        sc_fus.step(opt_fus); sc_fus.update()
    assert opt_fus.t == 3 and sc_fus.get_scale() == 512.0
    st = sc_fus.state_dict(); sc2 = D.GradScaler(); sc2.load_state_dict(st)
    assert sc2.get_scale() == 512.0
    with torch.no_grad():
        out = fus.requires_grad_(False)(x, t, ctx).sample
    assert torch.isfinite(out).all()


def test_train_step_deterministic_and_accumulates(cuda):
    """the backward has no atomics: two identical steps give bit-identical gradients; a second backward accumulates into .grad"""
    import diffute_amd as D
    from diffute_amd.models import mse_loss
    from diffute_amd.synthetic import synth_inputs
    model = D.UNet2DConditionModel(**TINY_UNET).cuda()
    lat, mask, mlat, ctx = synth_inputs(1, 8, 8, 20, 128, device=cuda)
    x = torch.cat([lat, mask, mlat], 1); t = torch.tensor([500], device=cuda); target = torch.zeros(1, 4, 8, 8, device=cuda)
    def grads():
        model.zero_grad(set_to_none=True)
        mse_loss(model(x, t, ctx).sample, target).backward()
        return {k: p.grad.clone() for k, p in model.named_parameters()}
    a = grads(); b = grads()
    for k in a:
        assert torch.equal(a[k], b[k]), f"{k}: gradients differ between identical steps"
    mse_loss(model(x, t, ctx).sample, target).backward()          # accumulate on top of b
    k = "mid_block.resnets.0.conv1.weight"
    assert torch.allclose(dict(model.named_parameters())[k].grad, 2 * a[k], rtol=1e-6, atol=0)


def test_tiny_train_step_vs_golden(cuda):
    """P5 against the committed fixture (tests/golden/tiny_train.npz, scripts/make_golden.py): loss within 2 %, the L2 norm
    of every parameter gradient within 10 % (median within 3 %), the dozen stored gradients within 8e-2 rel-L2."""
    import diffute_amd as D
    from diffute_amd.models import mse_loss
    from diffute_amd.synthetic import synth_inputs
    from oracle import prng
    g = np.load(os.path.join(GOLD, "tiny_train.npz"))
    model = D.UNet2DConditionModel(**TINY_UNET).cuda()
    lat, mask, mlat, ctx = synth_inputs(2, 16, 16, 77, 128, device=cuda)
    tgt = torch.from_numpy(prng.normal(9, 41, 2 * 4 * 16 * 16).reshape(2, 4, 16, 16)).to(cuda)
    pred = model(torch.cat([lat, mask, mlat], 1), torch.tensor([981, 17], device=cuda), ctx).sample
    loss = mse_loss(pred, tgt)
    loss.backward()
    assert abs(float(loss) - float(g["loss_bf16emu"])) <= 2e-2 * float(g["loss_bf16emu"])
    assert_close(pred.detach(), torch.from_numpy(g["pred_bf16emu"]), E2E_EMU, "train forward vs golden")
    names = str(g["names"]).split("\n")
    sd = dict(model.named_parameters())
    ratio = np.array([float(sd[k].grad.float().norm()) for k in names]) / g["gnorms_bf16emu"]
    assert np.all(np.abs(ratio - 1) <= 0.10), f"gradient norm off for {names[int(np.argmax(np.abs(ratio - 1)))]}: x{ratio[np.argmax(np.abs(ratio - 1))]:.3f}"
    assert abs(np.median(ratio) - 1) <= 0.03
    for k in [n[len("g_bf16emu_"):] for n in g.files if n.startswith("g_bf16emu_")]:
        assert_close(sd[k].grad, torch.from_numpy(g["g_bf16emu_" + k]), 8e-2, "grad " + k)


def test_gradient_sync_single_rank_matches_plain(cuda):
    """D1 plumbing on one GPU: with a 1-rank RCCL group the in-backward bucketed exchange (events, side stream) must leave the gradients exactly
    as without it - in BOTH schedules: "rs_ag" (the default; the in-place reduce-scatter + all-gather whose output aliases a slice of its input,
    dist.py reduce_buckets - on RCCL and GPU tensors here) and "all_reduce".  And accumulation x sync (`accelerator.accumulate`,
    train_diffute_v1.py:873,926): with accumulate_steps = 2 / inside `no_sync()` the first backward leaves `.grad` alone and exchanges nothing,
    the second delivers the window's sum - the same bits as two plain backwards accumulated by autograd."""
    import torch.distributed as dist
    import diffute_amd as D
    from diffute_amd import dist as DD
    from diffute_amd.models import mse_loss
    from diffute_amd.synthetic import synth_inputs
    model = D.UNet2DConditionModel(**TINY_UNET).cuda()
    ins = []
    for sd in (0, 7):
        lat, mask, mlat, ctx = synth_inputs(1, 8, 8, 20, 128, seed=sd, device=cuda)
        ins.append((torch.cat([lat, mask, mlat], 1), torch.tensor([321 + sd], device=cuda), ctx, torch.full((1, 4, 8, 8), 1.0 - 0.1 * sd, device=cuda)))

    def backward(i):
        x, t, ctx, target = ins[i]
        mse_loss(model(x, t, ctx).sample, target).backward()
        torch.cuda.synchronize()

    def grads(seq, wrap=None):
        model.zero_grad(set_to_none=True)
        for j, i in enumerate(seq):
            if wrap is not None and j < len(seq) - 1:
                with wrap():
                    backward(i)
            else:
                backward(i)
        return {k: p.grad.clone() for k, p in model.named_parameters()}
    plain, plain2 = grads([0]), grads([0, 1])
    import socket
    with socket.socket() as sk:                          # a free port of this box (a fixed one may be held by an earlier rendezvous)
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=cuda if cuda.index is not None else torch.device("cuda", 0))
    calls = {"n": 0}
    real = DD.reduce_buckets

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    DD.reduce_buckets = counting
    try:
        for mode in ("rs_ag", "all_reduce"):
            model.set_gradient_sync(dist, mode=mode)
            synced = grads([0])
            for k in plain:
                assert torch.equal(plain[k], synced[k]), (mode, k)
            # accumulation window of two micro-steps: one exchange, the window's sum
            model.set_gradient_sync(dist, mode=mode, accumulate_steps=2)
            n0 = calls["n"]
            model.zero_grad(set_to_none=True)
            backward(0)
            assert calls["n"] == n0 and all(p.grad is None for p in model.parameters()), "a non-boundary micro-step exchanged / delivered a gradient"
            backward(1)
            assert calls["n"] == n0 + 1
            for k, p in model.named_parameters():
                assert torch.equal(p.grad, plain2[k]), (mode, "accumulate_steps", k)
            # the same window written with no_sync()
            model.set_gradient_sync(dist, mode=mode)
            n0 = calls["n"]
            acc = grads([0, 1], wrap=model.no_sync)
            assert calls["n"] == n0 + 1
            for k in plain2:
                assert torch.equal(acc[k], plain2[k]), (mode, "no_sync", k)
    finally:
        DD.reduce_buckets = real
        model.set_gradient_sync(None)
        dist.destroy_process_group()
    # FusedAdamW in a window: the arena holds the window's sum at the boundary, one pending step
    model.set_gradient_sync(None)
    opt = D.FusedAdamW(model, lr=1e-3)
    model.zero_grad(set_to_none=True); opt.zero_grad()
    backward(0); backward(1)
    torch.cuda.synchronize()
    want = model._tb["grads"].clone()                    # FusedAdamW's own accumulation (no exchange)
    opt.zero_grad()

    class OneRank:                                        # a 1-rank "process group" whose collectives are the identity
        @staticmethod
        def get_world_size(group=None): return 1
        @staticmethod
        def get_rank(group=None): return 0
        @staticmethod
        def reduce_scatter_tensor(out, inp, group=None): assert out.data_ptr() == inp.data_ptr()
        @staticmethod
        def all_gather_into_tensor(out, inp, group=None): assert out.data_ptr() == inp.data_ptr()
        @staticmethod
        def all_reduce(t, group=None): pass
    model.set_gradient_sync(OneRank, accumulate_steps=2)
    try:
        backward(0); backward(1)
        torch.cuda.synchronize()
        got = model._tb["grads"]
        bad = (got != want).nonzero().flatten()
        assert opt._pending == 2 and bad.numel() == 0, (f"FusedAdamW + accumulation window: the arena is not the window's sum: {bad.numel()} of {got.numel()} slots differ, "
                                                        f"first at {bad[:4].tolist()}, got {got[bad[:4]].tolist()} want {want[bad[:4]].tolist()}")
        opt.step()
    finally:
        model.set_gradient_sync(None)


def test_cfg1_full_train_step_golden(cuda):
    """P5 at full size: SD2-inpaint UNet (865.9 M parameters), B=1, latent 32, one training step against the committed
    oracle fixture (tests/golden/cfg1_train.npz: bf16-emulating forward, fp32 autograd backward).  Loss within 2 %, the
    L2 norm of every one of the 686 parameter gradients within 3 % (measured: worst 0.3 %, median 0.1 %), three gradients in full."""
    import diffute_amd as D
    from diffute_amd.models import mse_loss
    from diffute_amd.synthetic import synth_inputs
    from oracle import prng
    path = os.path.join(GOLD, "cfg1_train.npz")
    if not os.path.exists(path):
        pytest.skip("cfg1_train.npz not generated")
    g = np.load(path)
    unet = D.UNet2DConditionModel(device=cuda)
    lat, mask, mlat, ctx = synth_inputs(1, 32, 32, 577, 1024, device=cuda)
    tgt = torch.from_numpy(prng.normal(9, 42, 4 * 32 * 32).reshape(1, 4, 32, 32)).to(cuda)
    pred = unet(torch.cat([lat, mask, mlat], 1), torch.tensor([437], device=cuda), ctx).sample
    loss = mse_loss(pred, tgt)
    loss.backward()
    assert_close(pred.detach(), torch.from_numpy(g["pred"]), E2E_EMU, "cfg1 train forward")
    assert abs(float(loss.detach()) - float(g["loss"])) <= 2e-2 * float(g["loss"])
    names = str(g["names"]).split("\n")
    sd = dict(unet.named_parameters())
    ratio = np.array([float(sd[k].grad.float().norm()) for k in names]) / g["gnorms"]
    worst = int(np.argmax(np.abs(ratio - 1)))
    print(f"cfg1 train: loss {float(loss.detach()):.5f} (oracle {float(g['loss']):.5f}); gradient-norm ratio median {np.median(ratio):.4f}, "
          f"worst {names[worst]} x{ratio[worst]:.3f}")
    assert np.all(np.abs(ratio - 1) <= 0.03), f"gradient norm off for {names[worst]}: x{ratio[worst]:.3f}"
    assert abs(np.median(ratio) - 1) <= 0.01
    assert_close(sd["conv_in.weight"].grad, torch.from_numpy(g["g_conv_in"]), 8e-2, "grad conv_in.weight")
    assert_close(sd["mid_block.attentions.0.transformer_blocks.0.attn1.to_q.weight"].grad[:64], torch.from_numpy(g["g_mid_to_q"]), 8e-2, "grad mid to_q")
    assert_close(sd["up_blocks.3.resnets.2.norm2.weight"].grad, torch.from_numpy(g["g_up3_norm2"]), 8e-2, "grad up3 norm2")


def test_fused_adamw_matches_torch(cuda):
    """N3: two steps of the fused HIP AdamW (+ global-norm clipping) against torch.optim.AdamW + clip_grad_norm_ fed with
    the SAME gradients (exported from the HIP backward): the first update agrees to fp32 rounding (same loss, same norm), the second within the
    noise of a few flipped bf16 roundings; the bf16 compute copies follow (next forward changes identically), state_dict() returns the updated masters."""
    import diffute_amd as D
    from diffute_amd.models import mse_loss
    from diffute_amd.synthetic import synth_inputs
    lat, mask, mlat, ctx = synth_inputs(1, 8, 8, 20, 128, device=cuda)
    x = torch.cat([lat, mask, mlat], 1); t = torch.tensor([500], device=cuda); target = torch.zeros(1, 4, 8, 8, device=cuda)
    hp = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    ref = D.UNet2DConditionModel(**TINY_UNET).cuda()
    opt_ref = torch.optim.AdamW(ref.parameters(), **hp)
    fus = D.UNet2DConditionModel(**TINY_UNET).cuda()
    opt_fus = D.FusedAdamW(fus, max_grad_norm=0.05, **hp)
    for step in range(2):
        l_ref = mse_loss(ref(x, t, ctx).sample, target); l_ref.backward()
        gn_ref = torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.05)
        opt_ref.step(); opt_ref.zero_grad(set_to_none=True)
        l_fus = mse_loss(fus(x, t, ctx).sample, target); l_fus.backward()
        opt_fus.step()
        # step 0: identical weights -> identical loss; later the fp32 masters agree to ~1e-7, which flips a few bf16 roundings
        # (measured over kernel revisions: 1e-4 .. 4.4e-4 of the loss at step 1 - which roundings flip changes with any change of a summation order)
        assert abs(float(l_ref.detach()) - float(l_fus.detach())) <= (1e-6 if step == 0 else 1e-3) * abs(float(l_ref.detach())), f"step {step}: losses diverged"
        assert abs(float(gn_ref) - float(opt_fus.grad_norm)) <= (1e-5 if step == 0 else 2e-3) * float(gn_ref)
        if step == 0:
            # identical weights and (deterministic) gradients went in: the updated masters must agree to fp32 rounding.
            # (Later steps cannot be compared element-wise: Adam moves every element by ~lr whatever its gradient's size,
            # so elements with noise-level gradients follow the sign of bf16 rounding noise.)
            sd_ref = {k: v.clone() for k, v in ref.state_dict().items()}; sd_fus = fus.state_dict()
            worst = max(float((sd_ref[k] - sd_fus[k]).abs().max() / (sd_ref[k].abs().max() + 1e-12)) for k in sd_ref)
            assert worst <= 2e-6, f"parameters after one step differ: max deviation relative to the tensor's max {worst:.2e}"
    with torch.no_grad():                       # inference path (folded LayerNorm copies refreshed) agrees as well
        a = ref.requires_grad_(False)(x, t, ctx).sample; b = fus.requires_grad_(False)(x, t, ctx).sample
    assert_close(b, a.cpu(), 2e-2, "forward after fused optimizer steps")


def test_fused_adamw_ema_shadow(cuda):
    """`--use_ema` (train_diffute_v1.py:642-646,934-935): the shadow arena updated inside the fused optimizer pass follows
    diffusers' EMAModel.step recurrence  s -= (1 - decay_k) * (s - p_k)  with decay_k = min(decay, (1 + k') / (10 + k')),
    k' = max(0, k - update_after_step - 1), applied to the parameters the optimizer just produced."""
    import diffute_amd as D
    from diffute_amd.models import mse_loss
    from diffute_amd.synthetic import synth_inputs
    lat, mask, mlat, ctx = synth_inputs(1, 8, 8, 20, 128, device=cuda)
    x = torch.cat([lat, mask, mlat], 1); t = torch.tensor([300], device=cuda); target = torch.zeros(1, 4, 8, 8, device=cuda)
    unet = D.UNet2DConditionModel(**TINY_UNET).cuda()
    opt = D.FusedAdamW(unet, lr=1e-3, max_grad_norm=1.0, ema_decay=0.9999)
    shadow = {k: v.detach().clone().float() for k, v in unet.state_dict().items()}
    for k in range(1, 5):
        mse_loss(unet(x, t, ctx).sample, target).backward()
        opt.step()
        step = max(0, k - 0 - 1)
        decay = 0.0 if step <= 0 else min(0.9999, (1 + step) / (10 + step))
        assert abs(opt.ema_decay_at(k) - decay) < 1e-12
        params = unet.state_dict()
        for name in shadow:
            shadow[name] = shadow[name] - (1 - decay) * (shadow[name] - params[name].float())
    ema = opt.ema_state_dict()
    assert set(ema) == set(shadow)
    worst = max(float((ema[n] - shadow[n]).abs().max() / (shadow[n].abs().max() + 1e-12)) for n in shadow)
    assert worst <= 1e-6, f"EMA shadow parameters deviate from the EMAModel recurrence: {worst:.2e}"
    moved = max(float((ema[n] - params[n].float()).abs().max()) for n in shadow)
    assert moved > 0, "the shadow copy must lag the parameters"
    with pytest.raises(RuntimeError):
        D.FusedAdamW(D.UNet2DConditionModel(**TINY_UNET).cuda()).ema_state_dict()


# ------------------------------------------------------------------------------------------------ glyph encoder (N1)
TINY_VIT = dict(image_size=64, patch_size=16, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, qkv_bias=True)


def test_tiny_glyph_encoder(cuda):
    """N1 on a tiny config (2 layers, 17 tokens, q/k/v biases): vs the bf16-emulating oracle and the fp32 oracle."""
    import diffute_amd as D
    from oracle import vit as OVT
    enc = D.TrOCREncoder(**TINY_VIT).cuda()
    g = torch.Generator().manual_seed(3)
    px = torch.randn(3, 3, 64, 64, generator=g)
    with torch.no_grad():
        y = enc(px.cuda()).last_hidden_state
    P = {k: v.detach().cpu() for k, v in enc.state_dict().items()}
    assert y.shape == (3, 17, 128)
    e16 = assert_close(y, OVT.vit_forward(P, OVT.TINY_VIT, px, emulate_bf16=True), E2E_EMU, "tiny ViT vs bf16-emulating oracle")
    e32 = assert_close(y, OVT.vit_forward(P, OVT.TINY_VIT, px), 5e-2, "tiny ViT vs fp32 oracle")
    print(f"tiny glyph encoder rel-L2: vs bf16emu {e16:.2e}, vs fp32 {e32:.2e}")


def test_full_glyph_encoder_properties(cuda):
    """TrOCR-large shapes (24 layers, 577 tokens of 1024): output shape, finiteness, batch independence, determinism,
    and one full-size oracle comparison at B=1."""
    import diffute_amd as D
    from oracle import vit as OVT
    enc = D.TrOCREncoder(device=cuda)
    assert sum(p.numel() for p in enc.parameters()) == 303_617_024
    g = torch.Generator().manual_seed(4)
    px = torch.randn(2, 3, 384, 384, generator=g).cuda()
    with torch.no_grad():
        y = enc(px).last_hidden_state
        y0 = enc(px[:1].contiguous()).last_hidden_state
        y2 = enc(px).last_hidden_state
    assert y.shape == (2, 577, 1024) and torch.isfinite(y).all()
    assert torch.equal(y, y2)
    assert_close(y[:1], y0.cpu(), E2E_EMU, "batch independence")     # B=1 and B=2 take different tile / split-K plans: bf16 noise over 24 layers
    P = {k: v.detach().cpu() for k, v in enc.state_dict().items()}
    ref = OVT.vit_forward(P, OVT.TROCR_LARGE_VIT, px[:1].cpu(), emulate_bf16=True)
    e = assert_close(y0, ref, E2E_EMU, "TrOCR-large encoder vs bf16-emulating oracle")
    print(f"full glyph encoder rel-L2 vs bf16emu oracle {e:.2e}")


# ------------------------------------------------------------------------------------------------ VAE training (N4)
def test_tiny_vae_train_step(cuda):
    """N4: one autoencoder training step (train_vae.py:721-724) on the tiny VAE: HIP forward that keeps activations + HIP
    backward vs torch autograd over the oracle.  Loss within 2 %, whole-gradient rel-L2 <= 5e-2, worst parameter <= 1.5e-1."""
    import diffute_amd as D
    from diffute_amd.models import mse_loss
    from oracle import pipeline as OP, vae as OV
    vae = D.AutoencoderKL(**TINY_VAE).cuda()
    g = torch.Generator().manual_seed(21)
    x = (torch.rand(2, 3, 64, 64, generator=g) * 2 - 1).cuda()
    tgt = (torch.rand(2, 3, 64, 64, generator=g) * 2 - 1).cuda()
    recon = vae(x)["sample"]
    loss = mse_loss(recon, tgt)
    loss.backward()
    P = {k: v.detach().cpu() for k, v in vae.state_dict().items()}
    for em in (True, False):
        rl, rrec, rg = OP.vae_train_grads(P, OV.TINY_VAE, x.cpu(), tgt.cpu(), emulate_bf16=em)
        assert_close(recon.detach(), rrec, 5e-2, "vae train forward")
        assert abs(float(loss.detach()) - rl) <= 2e-2 * abs(rl)
        errs = []; num = den = 0.0
        for k, p in vae.named_parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
            gh = p.grad.float().cpu(); gr = rg[k]
            num += float((gh - gr).pow(2).sum()); den += float(gr.pow(2).sum())
            if not k.endswith("to_k.bias"):      # d/d(b_k) is exactly 0 in exact arithmetic (softmax is shift-invariant per row): pure noise
                errs.append((rel_l2(gh, gr), k))
        errs.sort(reverse=True)
        tot = (num / den) ** 0.5
        print(f"tiny VAE train step vs {'bf16-emulating' if em else 'fp32'} oracle: loss {float(loss.detach()):.6f} ({rl:.6f}); whole-gradient rel-L2 {tot:.2e}; "
              "worst: " + ", ".join(f"{k} {e:.2e}" for e, k in errs[:4]))
        assert tot <= 5e-2 and errs[0][0] <= 1.5e-1, f"gradient mismatch: whole {tot:.3e}, worst {errs[0]}"


def test_tiny_vae_train_step_fp16_build(cuda):
    """train_vae.py's `--mixed_precision fp16`: the autoencoder training step on the fp16 build under torch's own GradScaler and a torch optimizer
    (the route accelerate takes); unscaled gradients vs the fp32 oracle at the bf16 test's bars, then one optimizer step through
    diffute_amd.training.train_vae_step(scaler=...)."""
    import diffute_amd as D
    from diffute_amd.models import mse_loss
    from diffute_amd.training import train_vae_step
    from oracle import pipeline as OP, vae as OV
    vae = D.AutoencoderKL(**TINY_VAE).cuda().to(dtype=torch.float16)
    g = torch.Generator().manual_seed(21)
    x = (torch.rand(2, 3, 64, 64, generator=g) * 2 - 1).cuda()
    tgt = (torch.rand(2, 3, 64, 64, generator=g) * 2 - 1).cuda()
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    opt = torch.optim.AdamW(vae.parameters(), lr=1e-4)
    recon = vae(x)["sample"]
    loss = mse_loss(recon, tgt)
    scaler.scale(loss).backward()
    scaler.unscale_(opt)
    P = {k: v.detach().cpu() for k, v in vae.state_dict().items()}
    rl, rrec, rg = OP.vae_train_grads(P, OV.TINY_VAE, x.cpu(), tgt.cpu(), emulate_bf16=False)
    assert_close(recon.detach().float(), rrec, 5e-2, "fp16 vae train forward")
    assert abs(float(loss.detach()) - rl) <= 2e-2 * abs(rl)
    errs = []; num = den = 0.0
    for k, p in vae.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        gh = p.grad.float().cpu(); gr = rg[k]
        num += float((gh - gr).pow(2).sum()); den += float(gr.pow(2).sum())
        if not k.endswith("to_k.bias"):
            errs.append((rel_l2(gh, gr), k))
    errs.sort(reverse=True)
    tot = (num / den) ** 0.5
    print(f"tiny fp16 VAE train step vs fp32 oracle: loss {float(loss.detach()):.6f} ({rl:.6f}); whole-gradient rel-L2 {tot:.2e}; worst: " +
          ", ".join(f"{k} {e:.2e}" for e, k in errs[:3]))
    assert tot <= 5e-2 and errs[0][0] <= 1.5e-1, f"gradient mismatch: whole {tot:.3e}, worst {errs[0]}"
    scaler.step(opt); scaler.update(); opt.zero_grad(set_to_none=True)
    sc = D.GradScaler(init_scale=1024.0)
    l0 = float(train_vae_step(vae, opt, x, tgt, scaler=sc)["loss"])
    l1 = float(train_vae_step(vae, opt, x, tgt, scaler=sc)["loss"])
    assert l1 < l0 < float(loss.detach()), f"loss does not fall over three fp16 steps: {float(loss.detach())} {l0} {l1}"


def test_training_loop_reduces_loss(cuda):
    """End-to-end sanity of forward + backward + fused optimizer: 25 steps on one fixed batch (tiny UNet, tiny frozen VAE,
    train_diffute_v1.py:859-935 through diffute_amd.training.train_step) must cut the loss by more than a third, with
    finite gradients throughout and parameters that actually moved."""
    import diffute_amd as D
    from diffute_amd.training import train_step
    unet = D.UNet2DConditionModel(**TINY_UNET).cuda()
    vae = D.AutoencoderKL(**TINY_VAE).cuda().requires_grad_(False)
    sched = D.DDPMScheduler()
    opt = D.FusedAdamW(unet, lr=2e-4, weight_decay=1e-2, max_grad_norm=1.0)
    g = torch.Generator(device=cuda).manual_seed(7)
    B = 2
    batch = dict(pixel_values=torch.rand(B, 3, 128, 128, device=cuda, generator=g) * 2 - 1,
                 masked_images=torch.rand(B, 3, 128, 128, device=cuda, generator=g) * 2 - 1,
                 masks=(torch.rand(B, 1, 128, 128, device=cuda, generator=g) > 0.6).float(),
                 ocr_embeddings=torch.randn(B, 77, 128, device=cuda, generator=g))
    noise = torch.randn(B, 4, 16, 16, device=cuda, generator=g); ts = torch.tensor([700, 150], device=cuda)
    en = torch.randn(B, 4, 16, 16, device=cuda, generator=g); en2 = torch.randn(B, 4, 16, 16, device=cuda, generator=g)
    w0 = unet.state_dict()["mid_block.resnets.0.conv1.weight"].clone()
    losses = []
    for _ in range(25):
        out = train_step(unet, vae, sched, opt, batch, noise=noise, timesteps=ts, enc_noise=en, enc_noise_masked=en2)
        losses.append(float(out["loss"])); assert torch.isfinite(out["grad_norm"])
    print("training loop losses:", " ".join(f"{l:.4f}" for l in losses[::4]))
    assert losses[-1] < 0.66 * losses[0], f"loss did not fall: {losses[0]:.4f} -> {losses[-1]:.4f}"
    w1 = unet.state_dict()["mid_block.resnets.0.conv1.weight"]
    assert float((w1 - w0).abs().max()) > 1e-4


# ------------------------------------------------------------------------------------------------ contract details (round-1 advisor findings)
def test_context_cache_is_keyed_on_the_tensor_object(cuda, tiny_unet):
    """unet(sample, t, ehs) caches the cross-attention K/V of `ehs`; a NEW tensor that the caching allocator places at the
    freed address of an earlier one (same shape, _version 0) must not hit the old entry."""
    from diffute_amd.synthetic import synth_inputs
    lat, mask, mlat, ctx = synth_inputs(1, 8, 8, 20, 128, device=cuda)
    x = torch.cat([lat, mask, mlat], 1); t = torch.tensor(500)
    with torch.no_grad():
        want_b = tiny_unet(x, t, (ctx * -0.5).contiguous()).sample.clone()
        e1 = ctx.clone(); p1 = e1.data_ptr()
        ya = tiny_unet(x, t, e1).sample.clone()
        del e1
        e2 = torch.empty_like(ctx); e2.copy_(ctx * -0.5)           # usually lands on e1's address
        same_addr = e2.data_ptr() == p1
        yb = tiny_unet(x, t, e2).sample
    assert torch.equal(yb, want_b) and not torch.equal(ya, yb), f"stale context K/V reused (same address: {same_addr})"
    # in-place edits of the same tensor object are seen through _version
    with torch.no_grad():
        e2.mul_(-2.0)                                              # now == ctx
        yc = tiny_unet(x, t, e2).sample
    assert torch.equal(yc, ya)


def test_fused_adamw_optimizer_contract(cuda):
    """FusedAdamW inside the reference's loop shape (train_diffute_v1.py:745-750, :873, :925-933): gradient accumulation over
    two micro-batches == one step on their summed loss; an LR scheduler drives param_groups; state_dict()/load_state_dict()
    round-trip the packed state; load_state_dict on the MODEL after the optimizer exists reaches the arena and the masters."""
    import diffute_amd as D
    from diffute_amd.models import mse_loss
    from diffute_amd.synthetic import synth_inputs
    lat, mask, mlat, ctx = synth_inputs(2, 8, 8, 20, 128, device=cuda)
    x = torch.cat([lat, mask, mlat], 1); t = torch.tensor([500, 40], device=cuda); target = torch.zeros(2, 4, 8, 8, device=cuda)
    hp = dict(lr=1e-3, weight_decay=1e-2, max_grad_norm=0.0)
    a = D.UNet2DConditionModel(**TINY_UNET).cuda(); oa = D.FusedAdamW(a, **hp)
    b = D.UNet2DConditionModel(**TINY_UNET).cuda(); ob = D.FusedAdamW(b, **hp)
    assert isinstance(oa, torch.optim.Optimizer)
    # a: two accumulated micro-batches (each sample's loss / 2); b: the batch of two in one backward (mean over both)
    for i in range(2):
        (mse_loss(a(x[i:i + 1].contiguous(), t[i:i + 1].contiguous(), ctx[i:i + 1].contiguous()).sample, target[i:i + 1]) * 0.5).backward()
    mse_loss(b(x, t, ctx).sample, target).backward()
    def flat_grads(m):                    # the arena also holds non-gradient regions (derived copies, row padding): export per parameter
        from diffute_amd import _cabi
        out = []
        for k, p in zip(m._keys, m._param_list()):
            g = torch.empty(p.shape, dtype=torch.float32, device=cuda)
            _cabi.check(_cabi.lib().dmx_unet_grad_export(m._h, _cabi.ptr(m._tb["grads"]), k.encode(), _cabi.ptr(g), _cabi.current_stream()), "grad_export")
            out.append(g.reshape(-1))
        return torch.cat(out)
    ga = flat_grads(a); gb = flat_grads(b)
    assert rel_l2(ga, gb) < 2e-2, f"accumulated gradient differs from the batched one: {rel_l2(ga, gb):.2e}"
    oa.step(); ob.step()
    with pytest.raises(RuntimeError):
        oa.step()                                                   # no backward since the last step
    # zero_grad drops what was accumulated
    (mse_loss(a(x, t, ctx).sample, target)).backward(); g1 = flat_grads(a)
    oa.zero_grad()
    (mse_loss(a(x, t, ctx).sample, target)).backward()
    assert torch.equal(flat_grads(a), g1)
    oa.step()
    # LR scheduler (get_scheduler("constant_with_warmup") is a LambdaLR)
    sch = torch.optim.lr_scheduler.LambdaLR(oa, lambda s: min(1.0, (s + 1) / 4))
    assert abs(oa.lr - 0.25e-3) < 1e-12
    mse_loss(a(x, t, ctx).sample, target).backward(); oa.step(); sch.step()
    assert abs(oa.lr - 0.5e-3) < 1e-12
    # optimizer checkpoint round trip: a fresh model + optimizer continue bit-identically
    sd_opt = oa.state_dict(); sd_model = {k: v.clone() for k, v in a.state_dict().items()}
    c = D.UNet2DConditionModel(**TINY_UNET, seed=99).cuda(); oc = D.FusedAdamW(c, **hp)
    c.load_state_dict(sd_model)                                     # AFTER the optimizer was built: must reach arena + masters
    with torch.no_grad():
        assert torch.equal(c.requires_grad_(False)(x, t, ctx).sample, a.requires_grad_(False)(x, t, ctx).sample)
    a.requires_grad_(True); c.requires_grad_(True)
    oc.load_state_dict(sd_opt)
    assert oc.t == oa.t and abs(oc.lr - oa.lr) < 1e-12
    for m, o in ((a, oa), (c, oc)):
        mse_loss(m(x, t, ctx).sample, target).backward(); o.step()
    sa, sc = a.state_dict(), c.state_dict()
    assert all(torch.equal(sa[k], sc[k]) for k in sa), "resumed run diverged from the original"


def test_edit_latents_end_to_end(cuda, tiny_unet, tiny_vae):
    """P1 + P2 + P3 + P4 + T2 + T3 in the order of text_editing() (app.ipynb:779-819): encode the masked crop (injected
    sampling noise), downsample the mask, start from the seed-0 CPU randn, denoise, decode - product (HIP) vs oracle."""
    import diffute_amd as D
    from diffute_amd.init import normal
    from diffute_amd.synthetic import text_crop_images
    from oracle import pipeline as OP, unet as OU, vae as OV
    img = text_crop_images(1, 128, 128, device=cuda)
    mask = torch.zeros(1, 1, 128, 128, device=cuda); mask[:, :, 48:80, 16:112] = 1.0
    masked = img * (mask < 0.5)
    ctx = normal(2, 13, 77 * 128, cuda).reshape(1, 77, 128)
    en = normal(4, 71, 4 * 16 * 16, cuda).reshape(1, 4, 16, 16)
    out = D.edit_latents(tiny_unet, tiny_vae, D.DDIMScheduler(), img, masked, mask, ctx, 3, enc_noise=en)
    Pu = {k: v.detach().cpu() for k, v in tiny_unet.state_dict().items()}
    Pv = {k: v.detach().cpu() for k, v in tiny_vae.state_dict().items()}
    ref, lat, mlat = OP.edit_latents(Pu, OU.TINY_UNET, Pv, OV.TINY_VAE, masked.cpu(), mask.cpu(), ctx.cpu(), 3, en.cpu(), emulate_bf16=True)
    assert out.shape == (1, 3, 128, 128)
    e = assert_close(out, ref, 4e-2, "edit_latents (encode -> 3 DDIM steps from the seed-0 latents -> decode) vs bf16-emulating oracle")
    print(f"edit_latents end to end rel-L2 {e:.2e}")
    # explicit init_latents == the default seed-0 draw
    init = OP.initial_latents((1, 4, 16, 16)).to(cuda)
    out2 = D.edit_latents(tiny_unet, tiny_vae, D.DDIMScheduler(), img, masked, mask, ctx, 3, enc_noise=en, init_latents=init)
    assert torch.equal(out, out2)


# ------------------------------------------------------------------------------------------------ per-block taps and the fp32 validation path
def _load_taps():
    g = np.load(os.path.join(GOLD, "tiny_unet_taps.npz"))
    f32 = {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("fp32_")}
    b16 = {k[8:]: (torch.from_numpy(g[k].astype(np.int32)) << 16).view(torch.float32) for k in g.files if k.startswith("bf16emu_")}
    return f32, b16


TAP_TOL = dict(conv_in=2e-3, down0=1e-2, down1=2e-2, down2=2e-2, down3=2e-2, mid=2e-2, up0=2e-2, up1=2e-2, up2=2e-2, up3=2e-2)


def test_tiny_unet_block_taps(cuda, tiny_unet):
    """every block output of the product (bf16) forward - conv_in, down0..3, mid, up0..3 - against the committed per-block
    tensors of the bf16-emulating oracle (tests/golden/tiny_unet_taps.npz): a wrong-but-small term inside one block cannot hide
    behind the end-to-end tolerance.  Bounds per block grow along the chain like the bf16 decorrelation does (TAP_TOL)."""
    from diffute_amd.synthetic import synth_inputs
    _, b16 = _load_taps()
    lat, mask, mlat, ctx = synth_inputs(2, 16, 16, 77, 128, device=cuda)
    x = torch.cat([lat, mask, mlat], 1)
    y, taps = tiny_unet.forward_taps(x, torch.tensor(981), ctx)
    assert list(taps) == list(tiny_unet.TAP_NAMES) and set(taps) == set(b16)
    errs = {k: assert_close(taps[k], b16[k], TAP_TOL[k], f"block tap {k} vs bf16-emulating oracle") for k in taps}
    print("block taps rel-L2:", " ".join(f"{k} {e:.1e}" for k, e in errs.items()))
    with torch.no_grad():
        assert torch.equal(y, tiny_unet(x, torch.tensor(981), ctx).sample)      # taps do not change the result


def test_tiny_unet_fp32_validation_path(cuda, tiny_unet):
    """north_star: "within 1e-3 rel fp32".  The fp32 instantiation of the SAME graph walker (fp32 activations, fp32 master
    weights, plain fp32 kernels) against the fp32 oracle: eps and every block output <= 1e-3 (measured ~1e-6: only the
    summation order differs), scalar and per-sample timesteps, fused 3-part input."""
    from diffute_amd.synthetic import synth_inputs
    g = np.load(os.path.join(GOLD, "tiny_unet.npz"))
    f32, _ = _load_taps()
    lat, mask, mlat, ctx = synth_inputs(2, 16, 16, 77, 128, device=cuda)
    y, taps = tiny_unet.forward_fp32([lat, mask, mlat], torch.tensor(981), ctx, taps=True)
    e = assert_close(y, torch.from_numpy(g["eps_fp32"]), 1e-3, "fp32 validation path: eps vs fp32 oracle")
    errs = {k: assert_close(taps[k], f32[k], 1e-3, f"fp32 validation path: block {k}") for k in taps}
    print(f"fp32 validation path rel-L2: eps {e:.1e}; blocks " + " ".join(f"{k} {v:.1e}" for k, v in errs.items()))
    # and the bf16 product path sits at its bf16 distance from the same fp32 truth, block by block
    _, tb = tiny_unet.forward_taps(torch.cat([lat, mask, mlat], 1), torch.tensor(981), ctx)
    for k in tb:
        assert rel_l2(tb[k], taps[k]) < 2e-2, k


def test_cfg1_fp32_validation_path(cuda):
    """BASELINE config 1 (full SD2-inpaint UNet, B=1, 256 px, 10 DDIM steps) on the fp32 validation path: first-step eps and
    the final latents within 1e-3 rel-L2 of the fp32 oracle's (tests/golden/cfg1_full.npz eps0_fp32 / final_fp32)."""
    import diffute_amd as D
    from diffute_amd.synthetic import synth_inputs
    path = os.path.join(GOLD, "cfg1_full.npz")
    if not os.path.exists(path):
        pytest.skip("cfg1_full.npz not generated")
    g = np.load(path)
    unet = D.UNet2DConditionModel(device=cuda).requires_grad_(False)
    lat, mask, mlat, ctx = synth_inputs(1, 32, 32, 577, 1024, device=cuda)
    sch = D.DDIMScheduler(); sch.set_timesteps(10)
    x = lat * sch.init_noise_sigma
    eps0 = None
    for t in sch.timesteps:
        eps = unet.forward_fp32([sch.scale_model_input(x, t), mask, mlat], t, ctx)
        if eps0 is None:
            eps0 = eps.clone()
        x = sch.step(eps, t, x).prev_sample
    e0 = assert_close(eps0, torch.from_numpy(g["eps0_fp32"]), 1e-3, "cfg1 fp32 path: first-step eps")
    e1 = assert_close(x, torch.from_numpy(g["final_fp32"]), 1e-3, "cfg1 fp32 path: final latents after 10 DDIM steps")
    print(f"cfg1 on the fp32 validation path: eps0 rel-L2 {e0:.2e}, final latents {e1:.2e}")


def test_tiny_vae_fp32_validation_path(cuda, tiny_vae):
    """north_star's "within 1e-3 rel fp32" at model level for the autoencoder: the fp32 instantiation of the encode / decode
    graphs (fp32 activations, fp32 master weights, plain FMA kernels; AutoencoderKL.encode_fp32 / decode_fp32) against
    the fp32 oracle's golden tensors."""
    from diffute_amd.synthetic import synth_images
    from diffute_amd.init import normal
    g = np.load(os.path.join(GOLD, "tiny_vae.npz"))
    img = synth_images(2, 64, 64, device=cuda)
    z = normal(5, 22, 2 * 4 * 8 * 8, cuda).reshape(2, 4, 8, 8)
    e1 = assert_close(tiny_vae.encode_fp32(img), torch.from_numpy(g["moments_fp32"]), 1e-3, "tiny vae fp32 path, moments")
    e2 = assert_close(tiny_vae.decode_fp32(z), torch.from_numpy(g["image_fp32"]), 1e-3, "tiny vae fp32 path, image")
    print(f"tiny vae fp32 validation path rel-L2: moments {e1:.2e}, image {e2:.2e}")


def test_profile_by_kernel_symbol(cuda):
    """bench.py's roofline leg: dmx_profile_symbols lists the bracketed launches by kernel SYMBOL (the launch helpers note the template instance they
    launch, in rocprofv3's spelling); the symbols of a class sum to the class total of dmx_profile_end."""
    import ctypes
    import diffute_amd as D
    from diffute_amd import _cabi
    from diffute_amd.synthetic import synth_inputs
    lib = _cabi.lib()
    unet = D.UNet2DConditionModel(**TINY_UNET).cuda().requires_grad_(False)
    lat, mask, mlat, ctx = synth_inputs(2, 16, 16, 77, 128, device=cuda)
    D.denoise(unet, D.DDIMScheduler(), lat, mask, mlat, ctx, 2)
    torch.cuda.synchronize()
    lib.dmx_profile_begin()
    D.denoise(unet, D.DDIMScheduler(), lat, mask, mlat, ctx, 2)
    buf = (ctypes.c_double * (4 * 28))()
    _cabi.check(lib.dmx_profile_end(buf, len(buf)), "profile_end")
    sbuf = ctypes.create_string_buffer(1 << 16)
    nb = lib.dmx_profile_symbols(sbuf, len(sbuf))
    assert nb > 0
    rows = [l.split("\t", 5) for l in sbuf.raw[:nb].decode().splitlines()]
    assert rows and all(len(r) == 6 for r in rows)
    syms = [r[5] for r in rows]
    assert any(s.startswith("void dmx_gemm_kernel<") and s.endswith("(GemmArgs)") for s in syms), syms
    assert any("dmx_attn_d64_kernel" in s for s in syms), syms
    by_class = {}
    for c, n, ms, fl, by, _ in rows:
        e = by_class.setdefault(int(c), [0.0, 0.0]); e[0] += float(n); e[1] += float(ms)
    for c, (n, ms) in by_class.items():
        if c in (3,) or c >= 10:                          # attention and the GEMM / chain / halo classes: every launch of the class notes its symbol
            assert n == buf[4 * c] and abs(ms - buf[4 * c + 1]) <= 1e-3 * max(1.0, buf[4 * c + 1]), (c, n, ms, buf[4 * c], buf[4 * c + 1])
    assert lib.dmx_profile_symbols(sbuf, 8) == 0           # buffer too small: nothing written, 0 returned


def test_deferred_splitk_reduce_is_bit_identical(cuda):
    """ConvOpts.defer (exec.hip): at the 16x16 / 8x8 levels conv1 of a resnet is a split-K GEMM; its reduce pass is left to norm2, whose slab kernel sums
    the partial planes in its load stage (norm.hip GroupNormArgs.red_*) with the arithmetic of dmx_splitk_reduce_kernel - the same bits, one launch
    less.  conv2 does the same ACROSS blocks: its reduce pass rides in the next block's norm1 (resnet_run sets ConvOpts.defer; every other first consumer
    flushes).  Full-size UNet, batch 4 and batch 1, and the tiny config, with the switch on and off; at full size the deferral must actually happen:
    fewer launches of the reduce kernel (profile class 2) with the switch on."""
    import ctypes
    import diffute_amd as D
    from diffute_amd import _cabi
    from diffute_amd.synthetic import synth_inputs
    lib = _cabi.lib()

    def fwd(unet, parts, t):
        for sl in unet._slots.values():
            sl["ws_need"] = None
        _cabi.check(lib.dmx_unet_refresh_derived(unet._h, None), "refresh")      # drops the captured graphs
        return unet.forward_parts(parts, t).clone()
    full = D.UNet2DConditionModel(device=cuda).requires_grad_(False)
    tiny = D.UNet2DConditionModel(**TINY_UNET).cuda().requires_grad_(False)
    t = torch.tensor([501], device=cuda)
    try:
        for unet, shp in ((full, (4, 64, 64, 577, 1024)), (full, (1, 64, 64, 577, 1024)), (tiny, (2, 16, 16, 77, 128)), (tiny, (3, 8, 24, 40, 128))):
            lat, mask, mlat, ctx = synth_inputs(*shp, device=cuda)
            unet.set_context(ctx)
            outs, reduces = {}, {}
            for on in (1, 0, 1):
                lib.dmx_set_defer_reduce(on)
                outs.setdefault(on, []).append(fwd(unet, [lat, mask, mlat], t))
                torch.cuda.synchronize()
                lib.dmx_profile_begin()                      # (bracketed walk: launches per kernel class)
                prof = unet.forward_parts([lat, mask, mlat], t).clone()
                buf = (ctypes.c_double * (4 * 32))()
                _cabi.check(lib.dmx_profile_end(buf, len(buf)), "profile_end")
                assert torch.equal(prof, outs[on][-1])
                reduces[on] = int(buf[4 * 2])
            torch.cuda.synchronize()
            assert torch.isfinite(outs[1][0]).all()
            assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][0], outs[1][1]), f"deferred reduce changes the result at {shp}"
            if unet is full:
                # split-K resnet convs exist at full size (8x8 level: 6 conv1 + 6 conv2 at least); with the switch on, only a conv whose first consumer
                # is not a GroupNorm keeps its own reduce launch
                print(f"splitk_reduce launches at {shp[:3]}: {reduces[0]} without deferral, {reduces[1]} with")
                assert reduces[0] >= 8 and reduces[1] <= reduces[0] - 8, reduces
    finally:
        lib.dmx_set_defer_reduce(1)
