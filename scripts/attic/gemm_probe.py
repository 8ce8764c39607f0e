"""Times the GEMM kernel on a few reference shapes (square GEMMs + UNet shapes) under forced plans."""
import math
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffute_amd import ops  # noqa: E402


def time_it(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


dev = torch.device("cuda")
for (M, N, K) in [(16384, 320, 320), (16384, 320, 2880), (16384, 1280, 320), (16384, 2560, 320),
                  (4096, 640, 640), (1024, 1280, 1280), (256, 1280, 1280), (16384, 640, 5760)]:
    x = torch.randn(1, 1, M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    for tn in (5, 4, 3, 2, 1):
        for gm in (8,):
            t = time_it(lambda: ops.conv_gemm(x, w, N, ksize=1, pad=0, force_tn=tn, force_splitk=1, group_m=gm))
            print(f"M={M} N={N} K={K} tn={tn} group_m={gm}: {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.0f} TF", flush=True)
