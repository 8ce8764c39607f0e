"""GROUP_M of the GEMM tile rasterisation (rows of a super-tile; the tiles of one XCD cover group_m row tiles x a run of column tiles): in-graph time per
launch for the unsplit GEMMs of the transformer blocks.  Measurement aid."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffute_amd import ops
dev = torch.device("cuda")
SH = [(4096, 640, 640, 10), (1024, 1280, 1280, 8), (256, 1280, 1280, 8), (4096, 5120, 640, 9), (4096, 640, 2560, 10), (1024, 3840, 1280, 10), (4096, 1920, 640, 10), (1024, 10240, 1280, 12), (16384, 320, 320, 8)]
for (M, N, K, tn) in SH:
    src = torch.randn(1, 1, M, K, device=dev).to(torch.bfloat16)
    x = torch.empty_like(src)
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    line = f"M={M} N={N} K={K} tn={tn}:"
    for gm in (1, 2, 4, 8, 16, 32):
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(2):
                x.copy_(src); ops.conv_gemm(x, w, N, ksize=1, pad=0, force_tn=tn, group_m=gm)
            with torch.cuda.graph(g, stream=s):
                for _ in range(20):
                    x.copy_(src); ops.conv_gemm(x, w, N, ksize=1, pad=0, force_tn=tn, group_m=gm)
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); [g.replay() for _ in range(5)]; e1.record(s); torch.cuda.synchronize()
        line += f"  gm {gm}: {e0.elapsed_time(e1) * 10.0:5.1f}"
    print(line, flush=True)
