"""Times the attention shapes of the UNet pass (B=4, 512 px; and the 768-px first level) on the row-major-V path, inside a
captured graph that rotates over several q/k/v sets.  Measurement aid."""
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffute_amd import ops  # noqa: E402

dev = torch.device("cuda")
SHAPES = [(4, 5, 4096, 4096), (4, 10, 1024, 1024), (4, 20, 256, 256), (4, 20, 64, 64),
          (4, 5, 4096, 577), (4, 10, 1024, 577), (4, 20, 256, 577), (4, 20, 64, 577), (2, 5, 9216, 9216), (8, 16, 577, 577)]
NSET = 4


def bench(fns, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for f in fns:
            f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                for f in fns:
                    f()
        g.replay(); torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(s); g.replay(); g.replay(); b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / (2 * reps * len(fns)) * 1e3


for (B, H, Sq, Skv) in SHAPES:
    C = H * 64
    pad = (Skv + 63) // 64 * 64
    sets = []
    for i in range(NSET):
        if Sq == Skv:
            qkv = torch.randn(B * Sq, 3 * C, device=dev).to(torch.bfloat16)
            sets.append((qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], Sq))
        else:
            q = torch.randn(B * Sq, C, device=dev).to(torch.bfloat16)
            kv = torch.randn(B * pad, 2 * C, device=dev).to(torch.bfloat16)
            sets.append((q, kv[:, :C], kv[:, C:], pad))
    t = bench([(lambda s=s: ops.attention_v(s[0], s[1], s[2], B, H, Sq, Skv, 0.125, kv_rows=s[3])) for s in sets])
    fl = 4.0 * B * H * Sq * Skv * 64
    print(f"B={B} H={H:2d} Sq={Sq:5d} Skv={Skv:5d}: {t:7.1f} us  {fl / t / 1e6:6.0f} TF/s", flush=True)
