"""Per-block timeline (s_memrealtime, 10 ns ticks) of the short-K GEMMs of the transformer blocks (K = C linears, FF2): where the
~17 us of a 16384x320x320 launch go - dispatch ramp, prologue (first DMA round trip), K loop, epilogue.  The input is rewritten by a
streaming kernel before every launch (as in the pass: X is the previous kernel's output, not an L2-resident buffer).  Measurement aid."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffute_amd import ops  # noqa: E402

dev = torch.device("cuda")
GEO = {1: (128, 64, 32), 2: (128, 128, 32), 8: (128, 64, 64), 9: (128, 128, 32), 10: (128, 128, 64), 11: (128, 160, 64)}
TNS = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 and sys.argv[1][0].isdigit() else [8, 10, 11]
SHAPES = [(4096, 640, 640), (1024, 1280, 1280), (256, 1280, 1280)] if '--low' in sys.argv else [(16384, 320, 320), (16384, 320, 1280), (16384, 960, 320), (4096, 640, 640)]
for (M, N, K) in SHAPES:
    src = torch.randn(1, 1, M, K, device=dev).to(torch.bfloat16)
    x = torch.empty_like(src)
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    for tn in TNS:
        bm, bn, bk = GEO[tn]
        tiles = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
        for dbg in (0, 1, 2, 3):
            tim = torch.zeros(tiles, 4, dtype=torch.int64, device=dev)
            for _ in range(3):
                x.copy_(src)
                ops.conv_gemm(x, w, N, ksize=1, pad=0, force_tn=tn, timing=tim, dbg=dbg)
            torch.cuda.synchronize()
            t = tim.cpu().double() * 0.01
            t0 = t[:, 0].min()
            life = t[:, 3] - t[:, 0]
            st = (t[:, 0] - t0)
            print(f"M={M} N={N} K={K} tn={tn} dbg={dbg}: {tiles} blocks | span {(t[:, 3] - t0).max():5.1f} us | start p50 {st.median():4.1f} p90 {st.quantile(0.9):4.1f} max {st.max():4.1f} | "
                  f"life mean {life.mean():4.1f} max {life.max():4.1f} = prologue {(t[:, 1] - t[:, 0]).mean():4.2f} + loop {(t[:, 2] - t[:, 1]).mean():4.2f} + epilogue {(t[:, 3] - t[:, 2]).mean():4.2f}", flush=True)
    # the same launches timed by events inside one graph (what the pass pays per launch, including the launch gap)
    for tn in TNS:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(2):
                x.copy_(src); ops.conv_gemm(x, w, N, ksize=1, pad=0, force_tn=tn)
            with torch.cuda.graph(g, stream=s):
                for _ in range(20):
                    x.copy_(src); ops.conv_gemm(x, w, N, ksize=1, pad=0, force_tn=tn)
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, stream=s):
                for _ in range(20):
                    x.copy_(src)
            res = []
            for gg in (g, g2):
                gg.replay(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s); [gg.replay() for _ in range(5)]; e1.record(s); torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) * 10.0)
        print(f"M={M} N={N} K={K} tn={tn}: copy+gemm {res[0]:.1f} us, copy alone {res[1]:.1f} us -> gemm {res[0] - res[1]:.1f} us per launch in a graph", flush=True)
