"""Per-block timeline (s_memrealtime, 10 ns ticks) of the persistent stream-K GEMM instances with the ablation switches of
GemmDesc.dbg (bit 0: no MFMA phase, bit 1: no DMA refills; results invalid) - what the K loop costs per K-tile and what
bounds it.  Measurement aid."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffute_amd import ops  # noqa: E402

dev = torch.device("cuda")
TNS = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [13, 15]
GEO = {11: (128, 160, 64), 13: (256, 160, 64), 15: (256, 128, 64), 16: (256, 160, 64)}
for (M, N, K) in [(16384, 320, 2880), (16384, 320, 5760), (16384, 640, 2560)]:
    x = torch.randn(1, 1, M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    for tn in TNS:
        bm, bn, bk = GEO[tn]
        tiles = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
        nkt = K // bk
        grid = tiles if tn < 13 else min(256, tiles * nkt // 4) // 8 * 8
        for dbg in (0, 1, 2, 3):
            tim = torch.zeros(max(grid, tiles), 4, dtype=torch.int64, device=dev)
            for _ in range(3):
                ops.conv_gemm(x, w, N, ksize=1, pad=0, force_tn=tn, timing=tim, dbg=dbg)
            torch.cuda.synchronize()
            t = tim[:grid].cpu().double() * 0.01
            t0 = t[:, 0].min()
            span = (t[:, 3] - t0).max(); per = (t[:, 3] - t[:, 0])
            kt_per_block = tiles * nkt / grid
            print(f"M={M} N={N} K={K} tn={tn} dbg={dbg}: grid {grid}, {kt_per_block:.1f} K-tiles/block | kernel span {span:6.1f} us | block mean {per.mean():6.1f} max {per.max():6.1f} "
                  f"-> {per.mean() / kt_per_block:.3f} us per K-tile | start spread {(t[:, 0] - t0).max():.1f} | last item: prologue {(t[:, 1] - t[:, 0]).mean():.1f} loop {(t[:, 2] - t[:, 1]).mean():.1f} tail {(t[:, 3] - t[:, 2]).mean():.1f}", flush=True)
