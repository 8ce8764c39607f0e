"""Times the linear-layer GEMM shapes of the UNet (B=4, 512 px) on the default plan and on the persistent kernel's instances
(force_tn 13 / 14 / 15), and prints the per-block timeline of the persistent kernel.  Measurement aid.

The timed loop rotates over several activation / weight / output buffers (total > the 256 MB Infinity Cache would be
needed to defeat it entirely; here 8 sets, which at least defeats L2) inside one captured hipGraph, like the in-situ pass."""
import math
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffute_amd import ops  # noqa: E402

dev = torch.device("cuda")
SHAPES = [  # (M, N, K, geglu, res)
    (16384, 320, 320, 0, 1), (4096, 640, 640, 0, 1), (1024, 1280, 1280, 0, 1), (256, 1280, 1280, 0, 1),
    (16384, 960, 320, 0, 0), (4096, 1920, 640, 0, 0), (1024, 3840, 1280, 0, 0),
    (16384, 2560, 320, 1, 0), (4096, 5120, 640, 1, 0), (1024, 10240, 1280, 1, 0),
    (16384, 320, 1280, 0, 1), (4096, 640, 2560, 0, 1), (1024, 1280, 5120, 0, 1),
    (16384, 320, 128, 0, 0), (2560, 640, 1024, 0, 0), (2560, 1280, 1024, 0, 0), (2560, 2560, 1024, 0, 0),
]
NSET = 8


def bench(fn_list, reps=6):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for f in fn_list:
            f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                for f in fn_list:
                    f()
        g.replay(); torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(s); g.replay(); g.replay(); b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / (2 * reps * len(fn_list)) * 1e3


which = sys.argv[1:] or ["time"]
for (M, N, K, geglu, res) in SHAPES:
    xs = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(NSET)]
    ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16) for _ in range(NSET)]
    bs = [torch.randn(N, device=dev) * 0.1 for _ in range(NSET)]
    rs = [torch.randn(M, N, device=dev).to(torch.bfloat16) for _ in range(NSET)] if res else [None] * NSET
    line = f"M={M:6d} N={N:6d} K={K:5d} geglu={geglu} res={res}:"
    if "time" in which:
        for tn in (0, 13, 14, 15):
            if geglu and tn in (13, 14):
                continue
            try:
                fns = [(lambda i=i: ops.linear(xs[i], ws[i], bias=bs[i], res=rs[i], geglu=bool(geglu), force_tn=tn, rowstats=bool(res))) for i in range(NSET)]
                t = bench(fns)
                line += f"  tn={tn}: {t:6.1f} us ({2.0 * M * N * K / t / 1e6:5.0f} TF)"
            except Exception as e:  # noqa: BLE001
                line += f"  tn={tn}: failed ({str(e)[:40]})"
        print(line, flush=True)
    if "timeline" in which:
        for tn in ((15,) if geglu else (13, 14, 15)):
            tim = torch.zeros(8192, 4, dtype=torch.int64, device=dev)
            for _ in range(3):
                ops.linear(xs[0], ws[0], bias=bs[0], res=rs[0], geglu=bool(geglu), force_tn=tn, timing=tim, dbg=8, rowstats=bool(res))
            torch.cuda.synchronize()
            tall = tim.cpu().double() * 0.01
            tt = tall[:4096]; ex = tall[4096:]
            live = tt[:, 0] > 0
            tt = tt[live]; ex = ex[live]
            t0 = tt[:, 0].min()
            print(f"   tn={tn}: epi-input issue {(ex[:, 0] - tt[:, 0]).mean():.2f} | tile setup {(ex[:, 1] - ex[:, 0]).mean():.2f} | DMA prologue issue {(tt[:, 1] - ex[:, 1]).mean():.2f} | "
                  f"first K-tile landed +{(ex[:, 2] - tt[:, 1]).mean():.2f} | loop rest {(tt[:, 2] - ex[:, 2]).mean():.2f} | first epilogue {(ex[:, 3] - tt[:, 2]).mean():.2f}")
            print(f"   tn={tn} blocks={len(tt)}: span {(tt[:, 3] - t0).max():.1f} us | start max {(tt[:, 0] - t0).max():.1f} | prologue {(tt[:, 1] - tt[:, 0]).mean():.2f} | "
                  f"first-tile loop {(tt[:, 2] - tt[:, 1]).mean():.2f} | rest (epilogue + later tiles) {(tt[:, 3] - tt[:, 2]).mean():.2f}", flush=True)
