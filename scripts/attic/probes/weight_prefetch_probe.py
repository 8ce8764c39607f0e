"""Does a weight matrix that another kernel has just read (so that it sits in the memory-side Infinity Cache / some L2) make the GEMM that
streams it faster?  Per shape: the GEMM alone on COLD weights (rotating over > 1 GB of weight buffers), then the same with a small read-only
pass over the weights issued right before the GEMM (same stream).  Times are hipEvent brackets around the GEMM launch only.  Measurement aid
for the weight-prefetch stream (EXPERIMENTS.md round 4)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffute_amd import ops  # noqa: E402

dev = torch.device("cuda")
for (M, N, K, ks) in [(1024, 1280, 1280, 1), (4096, 640, 640, 1), (256, 1280, 1280, 1), (256, 1280, 11520, 3), (1024, 1280, 11520, 3), (64, 1280, 1280, 1), (64, 1280, 11520, 3)]:
    nbuf = max(8, int(1.5e9 / (N * K * 2)))
    ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16) for _ in range(min(nbuf, 400))]
    if ks == 1:
        x = torch.randn(1, 1, M, K, device=dev).to(torch.bfloat16)
    else:
        side = int(math.isqrt(M // 4)) if M >= 256 else 8
        B = M // (side * side)
        x = torch.randn(B, side, side, K // 9, device=dev).to(torch.bfloat16)
    res = {}
    for mode in ("cold", "prefetched", "hot"):
        ts = []
        for i in range(60):
            w = ws[0] if mode == "hot" else ws[i % len(ws)]
            if mode == "prefetched":
                w.view(torch.int16).sum()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(300000)                     # the GPU spins while the host queues event, GEMM, event: no submit latency in the bracket
            a.record()
            if ks == 1:
                ops.conv_gemm(x, w, N, ksize=1, pad=0)
            else:
                ops.conv_gemm(x, w, N, ksize=3, pad=1)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts = sorted(ts[10:])
        res[mode] = ts[len(ts) // 2]
    print(f"M={M} N={N} K={K}: cold {res['cold']:.1f} us, weights read just before {res['prefetched']:.1f} us, same weights every time {res['hot']:.1f} us", flush=True)
