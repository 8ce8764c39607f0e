"""Per-block timeline of the GEMM kernel (s_memrealtime, 10 ns ticks): when blocks start, how long the
prologue / K loop / epilogue take.  Measurement aid."""
import math
import sys

import torch

sys.path.insert(0, ".")
from diffute_amd import ops  # noqa: E402

dev = torch.device("cuda")
for (M, N, K, tn) in [(16384, 320, 320, 2), (16384, 320, 2880, 2), (4096, 640, 640, 2), (1024, 1280, 1280, 1), (16384, 2560, 320, 2), (4096, 4096, 4096, 2), (4096, 4096, 4096, 3), (16384, 640, 5760, 3), (16384, 2560, 320, 3)]:
    x = torch.randn(1, 1, M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    bm, bn, bk = (256, 128, 64) if tn == 3 else (128, 64 * tn, 32)
    nb = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
    tim = torch.zeros(nb, 4, dtype=torch.int64, device=dev)
    for _ in range(3):
        ops.conv_gemm(x, w, N, ksize=1, pad=0, force_tn=tn, force_splitk=1, timing=tim)
    torch.cuda.synchronize()
    t = tim.cpu().double() * 0.01           # us
    t0 = t[:, 0].min()
    start = t[:, 0] - t0
    pro = t[:, 1] - t[:, 0]; loop = t[:, 2] - t[:, 1]; epi = t[:, 3] - t[:, 2]
    end = t[:, 3] - t0
    print(f"M={M} N={N} K={K} tn={tn} blocks={nb}: kernel span {end.max():.1f} us | block start p50 {start.median():.1f} max {start.max():.1f} | "
          f"prologue {pro.mean():.2f} | loop mean {loop.mean():.2f} max {loop.max():.2f} ({K // bk} tiles -> {loop.mean() / (K // bk):.3f} us/tile) | "
          f"epilogue mean {epi.mean():.2f} max {epi.max():.2f}", flush=True)
