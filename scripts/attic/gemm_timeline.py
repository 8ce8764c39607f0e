"""Per-block timeline of the GEMM kernel (s_memrealtime, 10 ns ticks): when blocks start, how long the
prologue / K loop / epilogue take.  Measurement aid."""
import math
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffute_amd import ops  # noqa: E402

dev = torch.device("cuda")
import itertools
cases = [(M, N, K, tn, dbg) for (M, N, K) in [(16384, 320, 320), (4096, 640, 640), (1024, 1280, 1280), (16384, 2560, 320), (16384, 320, 2880)] for tn in (8, 9, 10, 11, 12) for dbg in (0,)]
for (M, N, K, tn, dbg) in cases:
    x = torch.randn(1, 1, M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    bm, bn, bk = {1: (128, 64, 32), 2: (128, 128, 32), 3: (256, 128, 64), 4: (128, 64, 64), 5: (128, 128, 64), 6: (256, 256, 32), 7: (256, 128, 64), 8: (128, 64, 64), 9: (128, 128, 32), 10: (128, 128, 64), 11: (128, 160, 64), 12: (128, 320, 64)}[tn]
    nb = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
    tim = torch.zeros(nb, 4, dtype=torch.int64, device=dev)
    for _ in range(3):
        ops.conv_gemm(x, w, N, ksize=1, pad=0, force_tn=tn, force_splitk=1, timing=tim, dbg=dbg)
    torch.cuda.synchronize()
    t = tim.cpu().double() * 0.01           # us
    t0 = t[:, 0].min()
    start = t[:, 0] - t0
    pro = t[:, 1] - t[:, 0]; loop = t[:, 2] - t[:, 1]; epi = t[:, 3] - t[:, 2]
    end = t[:, 3] - t0
    print(f"M={M} N={N} K={K} tn={tn} dbg={dbg} blocks={nb}: kernel span {end.max():.1f} us | block start p50 {start.median():.1f} max {start.max():.1f} | "
          f"prologue {pro.mean():.2f} | loop mean {loop.mean():.2f} max {loop.max():.2f} ({K // bk} tiles -> {loop.mean() / (K // bk):.3f} us/tile) | "
          f"epilogue mean {epi.mean():.2f} max {epi.max():.2f}", flush=True)
