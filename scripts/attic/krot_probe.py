"""NEEDS the krot patch of EXPERIMENTS.md round 5 item 3 (reverted in the tree: GemmArgs.krot, api.hip dbg >> 3).  K rotation of the GEMM walk (GemmArgs.krot: block t starts at K-tile (t krot) mod nkt and wraps): per-block timeline and in-graph time per
launch for the K = C linears and a few other unsplit shapes of the pass, against krot = 0.  Measurement aid (dbg bits 3.. carry krot)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffute_amd import ops
dev = torch.device("cuda")
GEO = {8: (128, 64, 64), 10: (128, 128, 64), 9: (128, 128, 32)}
SH = [(4096, 640, 640, 10), (1024, 1280, 1280, 8), (256, 1280, 1280, 8), (4096, 5120, 640, 9), (4096, 640, 2560, 10), (1024, 3840, 1280, 10), (4096, 1920, 640, 10)]
for (M, N, K, tn) in SH:
    src = torch.randn(1, 1, M, K, device=dev).to(torch.bfloat16)
    x = torch.empty_like(src)
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    bm, bn, bk = GEO[tn]
    tiles = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
    x.copy_(src)
    ref = ops.conv_gemm(x, w, N, ksize=1, pad=0, force_tn=tn).float()
    line = f"M={M} N={N} K={K} tn={tn} ({tiles} blocks):"
    for krot in (0, 1, 3, 5, 7):
        dbg = krot << 3
        tim = torch.zeros(tiles, 4, dtype=torch.int64, device=dev)
        for _ in range(3):
            x.copy_(src)
            y = ops.conv_gemm(x, w, N, ksize=1, pad=0, force_tn=tn, timing=tim, dbg=dbg)
        torch.cuda.synchronize()
        err = float((y.float() - ref).norm() / ref.norm())
        t = tim.cpu().double() * 0.01
        loop = float((t[:, 2] - t[:, 1]).mean()); span = float((t[:, 3] - t[:, 0].min()).max())
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(2):
                x.copy_(src); ops.conv_gemm(x, w, N, ksize=1, pad=0, force_tn=tn, dbg=dbg)
            with torch.cuda.graph(g, stream=s):
                for _ in range(20):
                    x.copy_(src); ops.conv_gemm(x, w, N, ksize=1, pad=0, force_tn=tn, dbg=dbg)
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); [g.replay() for _ in range(5)]; e1.record(s); torch.cuda.synchronize()
        per = e0.elapsed_time(e1) * 10.0
        line += f"  krot {krot}: loop {loop:5.2f} span {span:5.1f} copy+gemm {per:5.1f} us (rel diff {err:.1e})"
    print(line, flush=True)
