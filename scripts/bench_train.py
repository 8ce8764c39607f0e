"""Training-step timing for BASELINE configs[3] (train_diffute_v1.py DDP bf16, 512x512 synthetic text crops, 8 per GPU).

    python scripts/bench_train.py [--batch 8] [--steps 3] [--warmup 1] [--no-vae]
    python -m torch.distributed.run --nproc-per-node N scripts/bench_train.py ...     (one rank per GPU, RCCL)

Prints one JSON line: images/s over all ranks for forward + backward + gradient exchange + clip + AdamW, plus (N > 1) the
EXPOSED gradient-exchange time per step: how long the backward's stream waited for the side-stream exchange after its own
kernels had finished ("exposed (non-overlapped) all-reduce time")."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffute_amd as D                                  # noqa: E402
from diffute_amd import dist as DD                       # noqa: E402
from diffute_amd.training import train_step             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--px", type=int, default=512)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tiny", action="store_true", help="tiny UNet / VAE (plumbing check)")
    ap.add_argument("--exchange", choices=("rs_ag", "all_reduce"), default="rs_ag", help="gradient exchange schedule (SURVEY.md D1)")
    ap.add_argument("--torch-adamw", action="store_true", help="torch.optim.AdamW on exported gradients instead of the fused HIP optimizer")
    ap.add_argument("--mixed-precision", choices=("bf16", "fp16"), default="bf16", help="fp16: the fp16 build + diffute_amd.GradScaler (train_diffute_v1.py:267)")
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = DD.init_from_env("nccl")
    ucfg = dict(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4), cross_attention_dim=128) if a.tiny else {}
    vcfg = dict(block_out_channels=(64, 128, 128, 128), layers_per_block=1) if a.tiny else {}
    unet = D.UNet2DConditionModel(**ucfg).to(dev)
    vae = D.AutoencoderKL(**vcfg).to(dev).requires_grad_(False)
    sched = D.DDPMScheduler()
    scaler = None
    if a.mixed_precision == "fp16":
        unet.to(dtype=torch.float16); vae.to(dtype=torch.float16)
        scaler = D.GradScaler()
    if dist is not None:
        DD.broadcast_parameters(list(unet.parameters()), dist)                      # D3
        unet.set_gradient_sync(dist, mode=a.exchange)                              # D1
    if a.torch_adamw:
        opt = torch.optim.AdamW(unet.parameters(), lr=1e-4, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)   # train_diffute_v1.py:190-194
    else:
        opt = D.FusedAdamW(unet, lr=1e-4, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8, max_grad_norm=1.0)
    g = torch.Generator(device=dev).manual_seed(5 + rank)
    B, px = a.batch, a.px
    ctx_dim = unet.config.cross_attention_dim
    batch = dict(pixel_values=torch.rand(B, 3, px, px, device=dev, generator=g) * 2 - 1,
                 masked_images=torch.rand(B, 3, px, px, device=dev, generator=g) * 2 - 1,
                 masks=(torch.rand(B, 1, px, px, device=dev, generator=g) > 0.7).float(),
                 ocr_embeddings=torch.randn(B, 577, ctx_dim, device=dev, generator=g))
    stream = torch.cuda.Stream(device=dev)
    losses = []
    exposed = []
    with torch.cuda.stream(stream):
        for _ in range(a.warmup):
            train_step(unet, vae, sched, opt, batch, generator=g, scaler=scaler)
        DD.barrier_sync(dist, dev)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = train_step(unet, vae, sched, opt, batch, generator=g, scaler=scaler)
            losses.append(out["loss"])
        DD.barrier_sync(dist, dev)
        dt = time.perf_counter() - t0
        e = unet.exposed_exchange_ms()                     # last step: time the backward's stream waited for the exchange after its own
        if e is not None:                                  # kernels (read after the timed region: the query synchronises with the host)
            exposed.append(e)
    t, thr = DD.whole_job_throughput(dist, dt, B * a.steps, dev)
    if rank == 0:
        print(json.dumps({"metric": "DDP training images/sec (forward + backward + exchange + AdamW)", "value": round(thr, 3), "unit": "images/s",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(t / a.steps * 1e3, 2),
                          "optimizer": "torch.optim.AdamW" if a.torch_adamw else "FusedAdamW (HIP)", "per_gpu_batch": B, "px": px, "dtype": a.mixed_precision, "data": "synthetic", "loss_last": float(losses[-1]),
                          "loss_scale": (scaler.get_scale() if scaler else None),
                          "gradient_exchange": (a.exchange if dist is not None else None),
                          "exposed_exchange_ms_last_step": (round(exposed[-1], 3) if exposed else None),
                          "max_mem_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2)}))


if __name__ == "__main__":
    main()
