"""Times the persistent stream-K big-tile instances (force_tn 13..16) against the automatic plan on the conv / linear shapes of
the 512-px pass, inside a captured graph with rotating weights (same harness as scripts/tune_gemm.py), and checks each forced
result against the automatic plan's result."""
import math
import sys

import torch

sys.path.insert(0, ".")
from diffute_amd import ops  # noqa: E402
from scripts.tune_gemm import time_it, weights  # noqa: E402

B = 4
SHAPES = [  # (kind, H, Cin, N) conv3x3 at HxH / ("lin", M, K, N)
    ("conv", 64, 320, 320), ("conv", 64, 640, 320), ("conv", 64, 960, 320), ("lin", 16384, 1280, 320), ("lin", 16384, 320, 320),
    ("conv", 32, 640, 640), ("conv", 32, 1280, 640), ("conv", 32, 1920, 640), ("lin", 4096, 2560, 640),
    ("conv", 16, 1280, 1280), ("conv", 16, 2560, 1280), ("lin", 1024, 5120, 1280), ("lin", 16384, 320, 960), ("lin", 4096, 640, 1920),
]
if len(sys.argv) > 1 and sys.argv[1] == "vae":
    B = 8
    SHAPES = [("conv", 256, 128, 128), ("conv", 128, 256, 256), ("conv", 64, 512, 512), ("conv", 128, 128, 256), ("conv", 64, 256, 512)]
TNS = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [13, 15]

dev = torch.device("cuda")
for sh in SHAPES:
    if sh[0] == "conv":
        _, H, Cin, N = sh
        x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
        K = 9 * Cin; M = B * H * H
        ws = weights(N, K)
        run = lambda tn, i=0: ops.conv_gemm(x, ws[i % len(ws)], N, ksize=3, pad=1, force_tn=tn)
    else:
        _, M, K, N = sh
        x = torch.randn(1, 1, M, K, device=dev).to(torch.bfloat16)
        ws = weights(N, K)
        run = lambda tn, i=0: ops.conv_gemm(x, ws[i % len(ws)], N, ksize=1, pad=0, force_tn=tn)
    fl = 2.0 * M * N * K
    ref = run(0).float()
    t0 = time_it(lambda i: run(0, i))
    line = f"{sh[0]} M={M:6d} N={N:5d} K={K:6d}: auto {t0:7.1f} us ({fl / t0 / 1e6:6.0f} TF)"
    for tn in TNS:
        try:
            out = run(tn).float()
            err = float((out - ref).norm() / ref.norm())
            same = torch.equal(run(tn).float(), out)
            t = time_it(lambda i: run(tn, i))
            line += f" | tn{tn} {t:7.1f} us ({fl / t / 1e6:6.0f} TF) err {err:.1e}{'' if same else ' NONDET'}"
        except RuntimeError as e:
            line += f" | tn{tn} n/a"
    print(line, flush=True)
