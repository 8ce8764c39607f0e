"""Cost of the epilogue inputs / outputs of the K = C linear layers: plain, + residual, + residual + row statistics.  Measurement aid."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffute_amd import ops
dev = torch.device("cuda")
NSET = 8
def bench(fn_list, reps=6):
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
for (M, N, K) in [(16384, 320, 320), (4096, 640, 640), (1024, 1280, 1280)]:
    xs = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(NSET)]
    ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16) for _ in range(NSET)]
    bs = [torch.randn(N, device=dev) * 0.1 for _ in range(NSET)]
    rs = [torch.randn(M, N, device=dev).to(torch.bfloat16) for _ in range(NSET)]
    line = f"M={M} N={N} K={K}:"
    for tn in (8, 10, 2):
        for mode in ("plain", "res", "res+stats"):
            fns = [(lambda i=i: ops.linear(xs[i], ws[i], bias=bs[i], res=(rs[i] if mode != "plain" else None), force_tn=tn, rowstats=(mode == "res+stats"))) for i in range(NSET)]
            line += f"  tn{tn} {mode} {bench(fns):5.1f}"
    print(line, flush=True)
