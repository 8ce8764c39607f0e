"""How much of the 4096 x 4096 self-attention launch is load imbalance?  640 blocks (batch 4 x 5 heads x 32 query blocks of four waves) on 256 CUs at three
blocks per CU = half the SIMDs run three waves, half two.  Time the same kernel with the head count varied: 3 x 256 = 768 blocks fill every slot.
If time(H = 6) ~ time(H = 5), a balanced split of the 640 work units over 768 slots would be worth up to 1 - 640 / 768 = 17 %.
    python scripts/attn_balance_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffute_amd import ops  # noqa: E402

dev = torch.device("cuda")
S = 4096


def timed(B, H, reps=20):
    C = H * 64
    qkv = torch.randn(B * S, 3 * C, device=dev).to(torch.bfloat16)
    f = lambda: ops.attention_v(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, H, S, S, 0.125)
    for _ in range(3):
        f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps * 1e3)
    return best


print("# (batch, heads) -> blocks, us per launch, us per 256 blocks")
for B, H in ((4, 2), (4, 3), (4, 4), (4, 5), (4, 6), (4, 7), (4, 8), (4, 9), (4, 12), (2, 5), (1, 5)):
    blocks = B * H * (S // 128)
    us = timed(B, H)
    print(f"B={B} H={H:2d}: {blocks:5d} blocks  {us:7.1f} us   {us * 256 / blocks:6.1f} us per 256 blocks", flush=True)
