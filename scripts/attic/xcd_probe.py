"""Sensitivity of the deep-K convolutions to the (m, n) patch an XCD works on (GemmArgs.group_m): does the traffic an XCD
pulls over the fabric (activation rows re-fetched per XCD vs weight rows re-fetched per XCD) set the time?  Measurement aid."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffute_amd import ops
dev = torch.device("cuda")
NSET = 6
def bench(fn_list, reps=4):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for f in fn_list: f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                for f in fn_list: f()
        g.replay(); torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(s); g.replay(); g.replay(); b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / (2 * reps * len(fn_list)) * 1e3
for (B, H, Ci, Co, tn, sk) in [(4, 16, 1280, 1280, 10, 3), (4, 16, 1280, 1280, 7, 6), (4, 16, 2560, 1280, 7, 6), (4, 32, 640, 640, 7, 3), (4, 32, 1280, 640, 7, 3), (4, 8, 1280, 1280, 9, 12), (4, 64, 320, 320, 9, 1)]:
    xs = [torch.randn(B, H, H, Ci, device=dev).to(torch.bfloat16) for _ in range(NSET)]
    ws = [(torch.randn(Co, 9 * Ci, device=dev) / math.sqrt(9 * Ci)).to(torch.bfloat16) for _ in range(NSET)]
    bs = [torch.randn(Co, device=dev) * 0.1 for _ in range(NSET)]
    line = f"M={B*H*H} N={Co} K={9*Ci} tn={tn} sk={sk}:"
    for gm in (1, 2, 4, 8, 16):
        fns = [(lambda i=i: ops.conv_gemm(xs[i], ws[i], Co, bias=bs[i], force_tn=tn, force_splitk=sk, group_m=gm)) for i in range(NSET)]
        line += f"  gm{gm} {bench(fns):6.1f}"
    print(line, flush=True)
