"""Per-kernel-class time of ONE UNet training forward + backward (hipEvent-bracketed launches, dmx_profile_*).
    python scripts/prof_train.py [--batch 8] [--latent 64] [--csv gpurun_out/train_launches.csv]"""
import argparse
import collections
import csv
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffute_amd as D                                  # noqa: E402
from diffute_amd import _cabi                            # noqa: E402
from diffute_amd.models import mse_loss                  # noqa: E402
from diffute_amd.synthetic import synth_inputs           # noqa: E402

CLASSES = ["gemm_128x128", "gemm_128x64", "splitk_reduce", "attention_d64", "groupnorm", "layernorm", "other", "gemm_256x128", "wgrad", "gemm_256x128_ws",
           "gemm_128x128x32", "gemm_128x64x32", "gemm_256x128x64", "gemm_128x64x64_deep", "gemm_128x128x64_deep", "gemm_256x256x32", "gemm_256x128x64_ws",
           "gemm_128x64x64_8w", "gemm_128x128x32_8w", "gemm_128x128x64_8w", "gemm_128x160x64", "gemm_128x320x64"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--csv", default="")
    a = ap.parse_args()
    dev = torch.device("cuda")
    unet = D.UNet2DConditionModel(device=dev)
    lat, mask, mlat, ctx = synth_inputs(a.batch, a.latent, a.latent, 577, 1024, device=dev)
    x = torch.cat([lat, mask, mlat], 1); t = torch.randint(0, 1000, (a.batch,), device=dev); tgt = torch.randn_like(lat)
    stream = torch.cuda.Stream()
    lib = _cabi.lib()
    with torch.cuda.stream(stream):
        for it in range(2):
            if it == 1:
                torch.cuda.synchronize()
                if a.csv:
                    lib.dmx_profile_dump_path(a.csv.encode())
                lib.dmx_profile_begin()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
                e0.record()
            pred = unet(x, t, ctx).sample
            if it == 1:
                e1.record()
            mse_loss(pred, tgt).backward()
            if it == 1:
                e2.record()
        buf = (ctypes.c_double * (4 * len(CLASSES)))()
        _cabi.check(lib.dmx_profile_end(buf, len(buf)), "profile_end")
    torch.cuda.synchronize()
    print(f"batch {a.batch} latent {a.latent}: forward {e0.elapsed_time(e1):.1f} ms, backward (+grad export) {e1.elapsed_time(e2):.1f} ms (profiled run: launches are event-bracketed)")
    tot = 0.0
    for i, name in enumerate(CLASSES):
        n, ms, fl, by = buf[4 * i:4 * i + 4]
        if n > 0:
            tot += ms
            print(f"  {name:14s} {int(n):5d} launches {ms:8.2f} ms  avg {1e3 * ms / n:7.1f} us" + (f"  {fl / ms / 1e9:7.1f} TF/s" if fl > 0 else "") +
                  (f"  {by / ms / 1e6:7.1f} GB/s" if by > 0 else ""))
    print(f"  sum {tot:.1f} ms")
    if a.csv and os.path.exists(a.csv):
        agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
        for r in csv.DictReader(open(a.csv)):
            k = (int(r["class"]), r["tag"].split(" M=")[0] if r["tag"].startswith("wgrad") else r["tag"][:40])
            agg[k][0] += 1; agg[k][1] += float(r["ms"]); agg[k][2] += float(r["flops"])
        print("top tags:")
        for (c, tag), (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
            print(f"  {CLASSES[c]:14s} {tag:42s} x{n:4d} {ms:8.2f} ms" + (f" {fl / ms / 1e9:7.1f} TF/s" if fl > 0 else ""))


if __name__ == "__main__":
    main()
