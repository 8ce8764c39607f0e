"""Times the d=64 attention kernel on every shape of the UNet pass (B=4, 512 px) - self and cross, all levels.  DMX_ATTN_ROWS
pins the rows per block (64 / 128 / 256).  Measurement aid."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffute_amd import ops
dev = torch.device("cuda")
def bench(f, reps=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): f()
        g.replay(); torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(s); g.replay(); g.replay(); b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / (2 * reps) * 1e3
line = "rows/block=" + os.environ.get("DMX_ATTN_ROWS", "128") + ":"
for (B, H, Sq, Skv) in [(4, 5, 4096, 4096), (4, 5, 4096, 577), (4, 10, 1024, 1024), (4, 10, 1024, 577), (4, 20, 256, 256), (4, 20, 256, 577), (4, 20, 64, 64), (4, 20, 64, 577)]:
    C = H * 64
    pad = (Skv + 63) // 64 * 64
    q = torch.randn(B * Sq, C, device=dev).to(torch.bfloat16)
    kv = torch.randn(B * pad, 2 * C, device=dev).to(torch.bfloat16)
    t = bench(lambda: ops.attention_v(q, kv[:, :C], kv[:, C:], B, H, Sq, Skv, 0.125, kv_rows=pad))
    line += f"  {Sq}x{Skv}: {t:6.1f}"
print(line, flush=True)
