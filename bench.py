#!/usr/bin/env python3
"""bench.py - BASELINE.json metric: 512x512 50-step denoise images/sec (whole job), MI355X.

One "step" = one pass of the hot path over one batch: the full 50-step DDIM denoise loop
(app.ipynb:796-816: per step cat([latents, mask, masked_latents]) -> UNet -> scheduler.step) for a
batch of 4 synthetic 512x512 masked-text crops (latents 4x64x64, glyph context [4,577,1024]) with
random-init SD2-inpainting-shaped weights, bf16 MFMA compute, inputs resident in HBM.

  python bench.py --gpus N --steps K --warmup W
N>1: launched by torch.distributed.run, one rank per GPU; images are independent so the batch is
replicated per rank (weak scaling, no data-path collective); RCCL is used only for the barrier and
the max-over-ranks of the elapsed time.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# profile classes of the library (diffute_amd/csrc/kernels.h ProfClass); classes 10.. = one per GEMM tile config, named
# after the rocprofv3 kernel name of that template instance dmx_gemm_kernel<WM, TN, BKT, NSTAGE, TM, NP>
# plan id -> (class name, template arguments of dmx_gemm_kernel<WM, TN, BKT, NSTAGE, TM, NP, NWN, PS, MF>); ids 3-5 and 13 are retired
GEMM_CFGS = [("gemm_128x128x32", "<2, 2, 32, 4, 2, 0, 2, false, 32>"), ("gemm_128x64x32", "<2, 1, 32, 4, 2, 0, 2, false, 32>"),
             ("gemm_256x128x64", "<4, 2, 64, 3, 2, 0, 2, false, 32>"), ("retired_3", ""), ("retired_4", ""), ("retired_5", ""),
             ("gemm_256x128x64_ws", "<2, 2, 64, 3, 4, 4, 2, false, 32>"), ("gemm_128x64x64_8w", "<4, 1, 64, 3, 1, 0, 2, false, 32>"),
             ("gemm_128x128x32_8w", "<4, 2, 32, 4, 1, 0, 2, false, 32>"), ("gemm_128x128x64_8w", "<4, 2, 64, 3, 1, 0, 2, false, 32>"),
             ("gemm_128x160x64", "<4, 5, 64, 2, 1, 0, 1, false, 32>"), ("gemm_128x320x64", "<4, 5, 64, 2, 1, 0, 2, false, 32>"),
             ("streamk_256x160x64", "<8, 5, 64, 3, 1, 0, 1, true, 32>"), ("retired_13", ""),
             ("streamk_256x128x64", "<8, 4, 64, 3, 1, 0, 1, true, 32>"), ("streamk_256x160x64_mf16", "<4, 5, 64, 3, 4, 0, 2, true, 16>")]
PROF_CLASSES = ["gemm_128x128_legacy", "gemm_128x64_legacy", "splitk_reduce", "attention_d64", "groupnorm", "layernorm", "other", "gemm_256x128_legacy", "wgrad",
                "gemm_256x128_ws_legacy"] + [n for n, _ in GEMM_CFGS] + ["xf_chain", "conv3x3_gn_halo", "skinny_conv"]
MFMA_BF16_PEAK_TFLOPS = 2500.0


def pmc_traffic(kernel_name):
    """bytes per launch of `kernel_name` from the committed PMC summary (scripts/rocprof_to_profiles.py), or None"""
    import csv
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    path = next((os.path.join(here, f) for f in ("r06_pmc_traffic.csv", "r05_pmc_traffic.csv", "r04_pmc_traffic.csv", "r03_pmc_traffic.csv", "r02_pmc_traffic.csv", "r01_pmc_traffic.csv") if os.path.exists(os.path.join(here, f))), None)
    if path is None:
        return None
    for r in csv.DictReader(open(path)):
        if r["kernel"] == kernel_name:
            return float(r["traffic_bytes_per_launch"])
    return None


def pmc_l2(kernel_name):
    """the committed L2-side / wave-state counter row of `kernel_name` (scripts/rocprof_to_profiles.py l2), or None"""
    import csv
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    path = next((os.path.join(here, f) for f in ("r06_pmc_l2.csv",) if os.path.exists(os.path.join(here, f))), None)
    if path is None:
        return None
    for r in csv.DictReader(open(path)):
        if r["kernel"] == kernel_name:
            return {k: (float(v) if k != "kernel" else v) for k, v in r.items()}
    return None


L2_PEAK_GBPS = 34500.0          # /opt/skills/guides/MI355X_MICROARCH.md, L2 section (fallback when the probe output below is absent)
HBM_ACHIEVABLE_GBPS = 6300.0    # same guide, HBM section


def l2_lds_ceiling():
    """the L2 -> LDS ceiling measured on this part: the best aggregate rate of an L2-resident ("shared") LDS-DMA stream over all 256 CUs, no compute -
    scripts/l2_lds_probe.hip, output committed as profiles/r06_l2_lds_probe.txt.  Returns (GB/s, source)."""
    import re
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r06_l2_lds_probe.txt")
    best = 0.0
    if os.path.exists(path):
        for line in open(path):
            m = re.search(r"shared\s+stream:.*?([0-9.]+) TB/s aggregate", line)
            if m:
                best = max(best, float(m.group(1)) * 1e3)
    return (best, "profiles/r06_l2_lds_probe.txt (scripts/l2_lds_probe.hip: L2-resident LDS-DMA stream on 256 CUs, best of the ring depths)") if best > 0 else \
           (L2_PEAK_GBPS, "MI355X_MICROARCH.md L2 section")


def secondary_bound(symbol, avg_launch_us, traffic_bytes):
    """What bounds the dominant kernel once the MFMA roof is ruled out (roofline.frac well under 1): the bytes it moves through the L2 -> LDS path per launch
    (TCC requests x 128 B) over its live launch time against the measured L2 -> LDS ceiling, the HBM traffic against the achievable HBM rate, and where the
    waves' cycles went (SQ counters).  The label is the largest of those fractions if it reaches one half; otherwise "latency" (waves parked on s_waitcnt /
    s_barrier: load round trips and block-level serialisation) or "issue" (the SIMD's issue slots: VALU / MFMA dependency stalls)."""
    row = pmc_l2(symbol)
    if row is None or avg_launch_us <= 0:
        return None
    ceil_gbps, ceil_src = l2_lds_ceiling()
    l2_gbps = row["l2_request_bytes_per_launch_at_128B"] / (avg_launch_us * 1e-6) / 1e9
    hbm_gbps = (traffic_bytes / (avg_launch_us * 1e-6) / 1e9) if traffic_bytes else None
    fr = {"l2": l2_gbps / ceil_gbps, "hbm": (hbm_gbps / HBM_ACHIEVABLE_GBPS) if hbm_gbps else 0.0,
          "lds": row["lds_issuing_frac_SQ_ACTIVE_INST_LDS"] + row["lds_issue_stall_frac_SQ_WAIT_INST_LDS"]}
    label = max(fr, key=fr.get)
    if fr[label] < 0.5:              # no pipe is half busy: the waves either sit parked (load round trips, block barriers) or the SIMD's issue slots are the limit
        label = "latency" if row["wave_parked_frac_SQ_WAIT_ANY"] >= 0.45 else "issue"
    return {"label": label, "l2_GBps": round(l2_gbps, 1), "l2_lds_ceiling_GBps": round(ceil_gbps, 1), "l2_frac_of_ceiling": round(fr["l2"], 4), "ceiling_source": ceil_src,
            "l2_hit_rate": row["l2_hit_rate"],
            "hbm_GBps": round(hbm_gbps, 1) if hbm_gbps else None, "hbm_frac_of_achievable": round(fr["hbm"], 4),
            "wave_parked_frac": row["wave_parked_frac_SQ_WAIT_ANY"], "issue_stall_frac": row["issue_stall_frac_SQ_WAIT_INST_ANY"],
            "issuing_frac": row["issuing_frac_SQ_ACTIVE_INST_ANY"], "lds_issuing_frac": row["lds_issuing_frac_SQ_ACTIVE_INST_LDS"],
            "lds_bank_conflict_per_lds_cycle": row["lds_bank_conflict_cycles_per_lds_active_cycle"],
            "source": "profiles/r06_pmc_l2.csv (separate rocprofv3 --pmc passes: TCC_* / TCP_TCC_READ_REQ, SQ_*), rates over this run's live avg_launch_us"}


def secondary_configs(dev, unet):
    """The other BASELINE configs and the reference's own operating point (each: 3 warm-ups, the MEDIAN of 5 timed repetitions -
    SURVEY 8d), so the driver sees them: cfg3 (AutoencoderKL encode + decode, 512 px, batch 32), the reference's loop (app.ipynb:545,806-816,914:
    DDPMScheduler, 150 steps, batch 1 - weight-bandwidth bound, priced against HBM), cfg5 (768 px, 50 steps, batch 2, FP16: the
    fp16 build of the library) and one cfg4 training step at the per-GPU shape (8 x 512 px, forward + backward + fused AdamW).
    Builder-side detail: scripts/bench_extra.py, scripts/bench_train.py."""
    import torch
    import diffute_amd as D
    from diffute_amd.flops import unet_flops
    from diffute_amd.synthetic import synth_inputs, text_crop_images
    out = {}

    def timed(fn, warm=3, reps=5):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2], r

    try:
        # ---- the reference's operating point: B = 1, 150 DDPM steps (injected variance noise = the device randn it draws)
        lat, mask, mlat, ctx = synth_inputs(1, 64, 64, 577, 1024, device=dev)
        nz = torch.randn(150, 1, 4, 64, 64, device=dev)
        t, o = timed(lambda: D.denoise(unet, D.DDPMScheduler(), lat, mask, mlat, ctx, 150, variance_noise=nz))
        wbytes = 2.0 * sum(p.numel() for p in unet.parameters())
        out["b1_ddpm150"] = {"config": "reference operating point: 512 px, batch 1, 150 DDPM steps, bf16 (app.ipynb:545,914)", "ms": round(t * 1e3, 1),
                             "images_per_s": round(1 / t, 3), "ms_per_unet_step": round(t * 1e3 / 150, 3),
                             "roofline": {"bound": "hbm", "achieved": round(wbytes * 150 / t / 1e9, 1), "peak": 6300.0, "unit": "GB/s",
                                          "frac": round(wbytes * 150 / t / 6.3e12, 4), "note": "algorithmic bytes = the 1.73 GB of bf16 weights every UNet call must stream; peak = 6.3 TB/s achievable HBM"},
                             "finite": bool(torch.isfinite(o).all())}
        # ---- diagnostic only (the headline stays batch 4): the same 50-step loop at batch 16 - separates "the kernels are slow" from
        # "batch 4 is small for 256 CUs"
        lat, mask, mlat, ctx = synth_inputs(16, 64, 64, 577, 1024, device=dev)
        t, o = timed(lambda: D.denoise(unet, D.DDIMScheduler(), lat, mask, mlat, ctx, 50), warm=1, reps=3)
        fl = 50 * unet_flops(unet.config, 16, 64, 64, 577, True, phase_upsample=True)
        out["b16"] = {"config": "diagnostic: the headline loop (512 px, 50 DDIM steps, bf16) at batch 16", "ms_per_batch": round(t * 1e3, 1), "images_per_s": round(16 / t, 3),
                      "loop_tflops": round(fl / t / 1e12, 1), "mfma_frac": round(fl / t / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "finite": bool(torch.isfinite(o).all())}
        del lat, mask, mlat, ctx, o
        unet._slots = {}
        torch.cuda.empty_cache()
        # ---- cfg5: 768 px, fp16 build
        unet.to(dtype=torch.float16)
        lat, mask, mlat, ctx = synth_inputs(2, 96, 96, 577, 1024, device=dev)
        t, o = timed(lambda: D.denoise(unet, D.DDIMScheduler(), lat, mask, mlat, ctx, 50))
        fl = 50 * unet_flops(unet.config, 2, 96, 96, 577, True, phase_upsample=True)
        out["cfg5_768px_fp16"] = {"config": "768x768, 50 DDIM steps, batch 2, fp16 (libdiffute_hip_f16.so)", "ms_per_batch": round(t * 1e3, 1), "images_per_s": round(2 / t, 3),
                                  "loop_tflops": round(fl / t / 1e12, 1), "mfma_frac": round(fl / t / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "finite": bool(torch.isfinite(o).all())}
        unet.to(dtype=torch.bfloat16)
        unet._slots = {}
        torch.cuda.empty_cache()
        # ---- cfg3: SD-VAE encode + decode, batch 32, 512 px
        vae = D.AutoencoderKL(device=dev).requires_grad_(False)
        img = text_crop_images(32, 512, 512, device=dev)
        with torch.no_grad():
            te, post = timed(lambda: vae.encode(img).latent_dist)
            z = post.mode()
            td, rec = timed(lambda: vae.decode(z).sample)
        fe, fd = 1.1167e12 * 32, 2.5145e12 * 32
        out["cfg3_vae_b32"] = {"config": "AutoencoderKL encode + decode, 512x512, batch 32, bf16", "encode_ms": round(te * 1e3, 1), "decode_ms": round(td * 1e3, 1),
                               "images_per_s": round(32 / (te + td), 1), "tflops": round((fe + fd) / (te + td) / 1e12, 1),
                               "mfma_frac": round((fe + fd) / (te + td) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "finite": bool(torch.isfinite(rec).all())}
        del img, rec, z, post
        torch.cuda.empty_cache()
        # ---- cfg4 per-GPU shape: one training step (VAE encodes + forward + backward + clip + AdamW), 8 x 512 px
        from diffute_amd.training import train_step
        unet.requires_grad_(True)
        opt = D.FusedAdamW(unet, lr=1e-4, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8, max_grad_norm=1.0)
        g = torch.Generator(device=dev).manual_seed(5)
        batch = dict(pixel_values=torch.rand(8, 3, 512, 512, device=dev, generator=g) * 2 - 1, masked_images=torch.rand(8, 3, 512, 512, device=dev, generator=g) * 2 - 1,
                     masks=(torch.rand(8, 1, 512, 512, device=dev, generator=g) > 0.7).float(), ocr_embeddings=torch.randn(8, 577, 1024, device=dev, generator=g))
        sched = D.DDPMScheduler()
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            t, r = timed(lambda: train_step(unet, vae, sched, opt, batch, generator=g), warm=3, reps=5)
        out["cfg4_train_step_b8"] = {"config": "one training step at the cfg4 per-GPU shape: 8 x 512 px, bf16, VAE encodes + UNet forward + backward + clip + fused AdamW (1 GPU, no exchange)",
                                     "ms_per_step": round(t * 1e3, 1), "images_per_s": round(8 / t, 2), "loss": float(r["loss"]),
                                     "max_mem_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1)}
    except Exception as e:                                     # noqa: BLE001 - the headline line must still print
        out["error"] = f"{type(e).__name__}: {e}"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--latent", type=int, default=64, help="latent side (64 = 512 px)")
    ap.add_argument("--denoise-steps", type=int, default=50)
    ap.add_argument("--micro-batches", type=int, default=1, help="independent chains of the batch run concurrently on separate streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the one-rep timings of the other BASELINE configs (cfg3, cfg5, train step, B=1 DDPM-150)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--profile-csv", default="", help="write one row per kernel launch of the profiled pass")
    ap.add_argument("--mode", choices=("denoise", "train"), default="denoise",
                    help="denoise = BASELINE configs[1] (the headline metric); train = configs[3] DDP training step (scripts/bench_train.py)")
    args = ap.parse_args()
    # ---- launcher contract: `--gpus N` is the number of ranks.  Under torchrun WORLD_SIZE must agree; a bare
    # `python bench.py --gpus N` (N > 1) starts the N ranks itself as a CHILD torchrun - before this process has touched
    # the GPU (never exec from a process that initialised HIP) - and forwards the child's output and exit code.
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    if int(env_world or "1") != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world or 1}; launch with "
                         f"python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...")
    if args.mode == "train":                  # same launcher contract (torchrun env), per-GPU batch 8 at 512 px
        import runpy
        sys.argv = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", "bench_train.py"),
                    "--steps", str(args.steps), "--warmup", str(args.warmup)] + (["--batch", str(args.batch)] if args.batch != 4 else [])
        runpy.run_path(sys.argv[0], run_name="__main__")
        return

    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    import diffute_amd as D
    from diffute_amd import dist as DD
    dist = DD.init_from_env("nccl")
    from diffute_amd import _cabi
    from diffute_amd.flops import context_kv_flops, unet_flops
    from diffute_amd.synthetic import synth_inputs

    B, hw, T = args.batch, args.latent, args.denoise_steps
    unet = D.UNet2DConditionModel(device=dev).requires_grad_(False)
    sched = D.DDIMScheduler()
    lat, mask, mlat, ctx = synth_inputs(B, hw, hw, 577, 1024, seed=100 * rank, device=dev)
    unet._ensure_packed()

    def one_pass():
        return D.denoise(unet, sched, lat, mask, mlat, ctx, T, micro_batches=args.micro_batches)

    def sync_all():
        DD.barrier_sync(dist, dev)
        _cabi.poll_device_error()           # the public sync point: a kernel that gave up on an in-kernel wait raises here, never a silently wrong pass

    # every pass is checked, on the device and without a host sync: element 0 counts the non-finite values of the pass's latents,
    # element 1 the values that differ from the first pass's (the loop is deterministic: same inputs, same bits)
    n_pass = args.warmup + args.steps
    health = torch.zeros(max(n_pass, 1), 2, dtype=torch.int64, device=dev)
    first = None

    def checked_pass(i):
        nonlocal first
        o = one_pass()
        health[i, 0] = (~torch.isfinite(o)).sum()
        if first is None:
            first = o.clone()
        else:
            health[i, 1] = (o != first).sum()
        return o

    out = None
    for i in range(args.warmup):
        out = checked_pass(i)
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = checked_pass(args.warmup + i)
    sync_all()
    elapsed = time.perf_counter() - t0
    elapsed, value = DD.whole_job_throughput(dist, elapsed, B * args.steps, dev)     # max time, total images / it
    if dist is not None:                        # a bad pass on ANY rank is the job's error: rank 0 reports the element-wise maximum over the ranks
        dist.all_reduce(health, op=dist.ReduceOp.MAX)
    hh = health.cpu()
    bad = [i for i in range(n_pass) if int(hh[i, 0]) or int(hh[i, 1])]
    if bad:
        # a loop that returns garbage is not a measurement: say which pass, print ONE JSON line with value null, exit non-zero
        i = bad[0]
        err = {"metric": "512x512 50-step denoise images/sec", "value": None, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "error": f"pass {i} of {n_pass} (warm-up passes first) returned {int(hh[i, 0])} non-finite and {int(hh[i, 1])} values that differ from pass 0; "
                        f"{len(bad)} bad passes: {bad[:16]}",
               "bad_passes": [{"pass": j, "nonfinite": int(hh[j, 0]), "differs_from_pass0": int(hh[j, 1])} for j in bad[:16]]}
        if rank == 0:
            print(json.dumps(err), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        raise SystemExit(1)
    ms_per_step = 1e3 * elapsed / args.steps

    # executed work: the upsampler convs run phase-decomposed (4/9 of the reference formulation's multiply-adds)
    loop_flops = T * unet_flops(unet.config, B, hw, hw, 577, True, phase_upsample=True) + context_kv_flops(unet.config, B, 577)
    result = {
        "metric": "512x512 50-step denoise images/sec", "value": round(value, 3), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"SD2-inpaint UNet {T}-step DDIM denoise loop, {hw * 8}x{hw * 8} px, batch {B} per GPU, "
                               "glyph context [B,577,1024], ctx K/V cached per image (BASELINE configs[1]); "
                               "VAE encode/decode not in the timed region",
                   "global_batch": world * B, "denoise_steps": T, "parallelism": f"replicas x{world} (no collective)",
                   "weights": "random-init SD2-inpainting shapes (865,925,124 params), bf16 packed"},
        "loop_tflops_per_gpu": round(loop_flops * args.steps / elapsed / 1e12, 2),
        "loop_mfma_frac": round(loop_flops * args.steps / elapsed / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
        "passes_checked": {"n": n_pass, "all_finite": True, "all_bit_equal_to_pass0": True,
                           "cost": "three tiny device kernels per pass (isfinite count, != count, index writes; 65 k elements) inside the timed region, no host sync"},
    }

    dom = None
    if rank == 0 and not args.no_profile:
        # ---- roofline leg: one more pass with every launch bracketed by hipEvents on its stream
        lib = _cabi.lib()
        torch.cuda.synchronize(dev)
        if args.profile_csv:
            lib.dmx_profile_dump_path(args.profile_csv.encode())
        lib.dmx_profile_begin()
        one_pass()
        buf = (ctypes.c_double * (4 * len(PROF_CLASSES)))()
        _cabi.check(lib.dmx_profile_end(buf, len(buf)), "profile_end")
        classes = {}
        for i, name in enumerate(PROF_CLASSES):
            n, ms, fl, by = buf[4 * i:4 * i + 4]
            if n > 0:
                classes[name] = {"launches": int(n), "total_ms": round(ms, 3), "avg_us": round(1e3 * ms / n, 2),
                                 "tflops": round(fl / (ms * 1e-3) / 1e12, 1) if fl > 0 and ms > 0 else None,
                                 "gbps": round(by / (ms * 1e-3) / 1e9, 1) if ms > 0 else None}
        # per kernel SYMBOL (the launch helpers note rocprofv3's spelling of the instance they launch): one class can span several
        # template instances (the halo conv: warp-specialised / ping-pong / narrow tiles; every GEMM tile config: with and without
        # the column-statistics epilogue), so the roofline is quoted for the dominant SYMBOL, with its class total beside it
        sbuf = ctypes.create_string_buffer(1 << 16)
        nb = lib.dmx_profile_symbols(sbuf, len(sbuf))
        if nb == 0:                                     # (0 = the buffer was too small for the symbol table)
            sbuf = ctypes.create_string_buffer(1 << 20)
            nb = lib.dmx_profile_symbols(sbuf, len(sbuf))
        symbols = []
        for line in sbuf.raw[:nb].decode().splitlines():
            c, n_, ms_, fl_, by_, sym = line.split("\t", 5)
            symbols.append({"symbol": sym, "class": PROF_CLASSES[int(c)], "launches": int(float(n_)), "ms": float(ms_), "flops": float(fl_), "bytes": float(by_)})
        profiled_total = sum(c["total_ms"] for c in classes.values())
        n_launch = sum(c["launches"] for c in classes.values())
        # hipEvent cost per bracketed launch, derived in this run: the bracketed times of the profiled pass sum to more than the timed
        # (un-profiled, graph-replayed) pass by the event records inside the brackets
        ev_us = max(0.0, 1e3 * (profiled_total - ms_per_step) / max(n_launch, 1))
        result["profile_overhead_ms"] = round(profiled_total - ms_per_step, 3)
        result["event_overhead_us_per_launch"] = round(ev_us, 3)
        mfma_syms = [q for q in symbols if q["flops"] > 0]
        for q in mfma_syms:
            q["ms_corrected"] = max(q["ms"] - q["launches"] * ev_us * 1e-3, 1e-6)
        if not mfma_syms:                               # no launch helper noted a symbol with FLOPs (an fp32 / non-MFMA path): the headline line still prints
            result["roofline"] = None
            result["kernel_classes"] = classes
        dom = max(mfma_syms, key=lambda q: q["ms_corrected"]) if mfma_syms else None
    if rank == 0 and not args.no_profile and dom is not None:
        ach = dom["flops"] / (dom["ms_corrected"] * 1e-3) / 1e12
        cls = classes[dom["class"]]
        cls_syms = [q for q in mfma_syms if q["class"] == dom["class"]]
        cls_ms = sum(q["ms_corrected"] for q in cls_syms)
        cls_tr = [pmc_traffic(q["symbol"]) for q in cls_syms]
        result["roofline"] = {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": pmc_traffic(dom["symbol"]),
                              "kernel": dom["symbol"],
                              "secondary_bound": secondary_bound(dom["symbol"], 1e3 * dom["ms_corrected"] / dom["launches"], pmc_traffic(dom["symbol"])),
                              "launches": dom["launches"], "avg_launch_us": round(1e3 * dom["ms_corrected"] / dom["launches"], 2),
                              "algorithmic_gflop_per_launch": round(dom["flops"] / dom["launches"] / 1e9, 3),
                              "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["launches"]),      # activations + weights + output of this symbol's launches: compare with `traffic`
                              "class": {"name": dom["class"], "symbols": len(cls_syms), "launches": cls["launches"], "total_ms": round(cls_ms, 3),
                                        "tflops": round(sum(q["flops"] for q in cls_syms) / (cls_ms * 1e-3) / 1e12, 1),
                                        "frac": round(sum(q["flops"] for q in cls_syms) / (cls_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                                        # launch-weighted PMC traffic over the symbols of the class that the committed profile lists
                                        "traffic_launch_weighted": (round(sum(t * q["launches"] for t, q in zip(cls_tr, cls_syms) if t is not None) /
                                                                          max(sum(q["launches"] for t, q in zip(cls_tr, cls_syms) if t is not None), 1))
                                                                    if any(t is not None for t in cls_tr) else None)},
                              "whole_loop_frac": result["loop_mfma_frac"],
                              "note": "dominant kernel SYMBOL of one hipEvent-bracketed 50-step pass (event cost per launch derived in-run: bracketed sum minus the timed pass, "
                                      "over the launch count); class = all template instances of that kernel family; whole_loop_frac = algorithmic FLOPs of the whole loop / timed pass / peak; "
                                      "traffic = HBM bytes per launch of this symbol from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes committed under profiles/ "
                                      "(FETCH x2 per the gfx950 note), null if that symbol is absent there"}
        result["kernel_symbols"] = sorted(({"symbol": q["symbol"], "class": q["class"], "launches": q["launches"], "total_ms": round(q["ms_corrected"], 3),
                                            "tflops": round(q["flops"] / (q["ms_corrected"] * 1e-3) / 1e12, 1)} for q in mfma_syms), key=lambda q: -q["total_ms"])[:12]
        result["kernel_classes"] = classes

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # ---- CPU baseline: the torch-CPU fp32 oracle (a port; the reference itself cannot be imported)
        from oracle import unet as OU
        P = {k: v.detach().cpu() for k, v in unet.state_dict().items()}
        l1, m1, ml1, c1 = [x[:1].cpu() for x in (lat, mask, mlat, ctx)]
        inp = torch.cat([l1, m1, ml1], 1)
        tt = torch.tensor(981)
        OU.unet_forward(P, OU.SD2_INPAINT_UNET, inp[:, :, :16, :16], tt, c1)        # warm-up (small)
        # the thread count is measured, not assumed: 16 and 32 threads, one forward each (a bounded sample: ~2 s per try), the
        # faster is the baseline.  More does not help on this pool's hosts - measured once on a 256-core box: 16 threads 2.23 s,
        # 64 threads 4.17 s, all 256 cores 137.7 s (the container's CPU quota makes wide thread pools thrash)
        ncpu = os.cpu_count() or 1
        tries = {}
        for nthreads in sorted({min(16, ncpu), min(32, ncpu)}):
            torch.set_num_threads(nthreads)
            c0 = time.perf_counter()
            eps_cpu = OU.unet_forward(P, OU.SD2_INPAINT_UNET, inp, tt, c1)
            tries[nthreads] = time.perf_counter() - c0
        nthreads = min(tries, key=tries.get)
        t_fwd = tries[nthreads]
        unet.set_context(ctx[:1].contiguous())
        eps_gpu = unet.forward_parts([lat[:1].contiguous(), mask[:1].contiguous(), mlat[:1].contiguous()],
                                     torch.tensor([981], device=dev))
        rel = float((eps_gpu.cpu() - eps_cpu).norm() / eps_cpu.norm())
        result["cpu_baseline"] = {"value": round(1.0 / (T * t_fwd), 5), "unit": "images/s", "cores": nthreads, "kind": "port",
                                  "host_cores_total": os.cpu_count(),
                                  "sample": f"1 UNet forward (B=1, {hw * 8} px, fp32 torch-CPU oracle, {nthreads} threads) = "
                                            f"{t_fwd:.2f} s, x{T} steps extrapolated linearly; seconds by thread count: "
                                            + ", ".join(f"{k}: {v:.2f}" for k, v in sorted(tries.items())),
                                  "gpu_vs_cpu_eps_rel_l2": round(rel, 5)}
    if rank == 0 and world == 1 and not args.no_secondary:
        result["secondary"] = secondary_configs(dev, unet)
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
