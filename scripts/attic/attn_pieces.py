"""Marginal cost of the pieces of the d=64 attention loop (library built with EXTRA=-DDMX_ATTN_PROBE; DMX_ATTN_PROBE_BITS picks
the variant - results of the probe variants are invalid by construction).  Measurement aid."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffute_amd import ops  # noqa: E402

dev = torch.device("cuda")
B, H, S = 4, 5, int(os.environ.get("S", "4096"))
C = H * 64
qkv = torch.randn(B * S, 3 * C, device=dev).to(torch.bfloat16)
f = lambda: ops.attention_v(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, H, S, S, 0.125)  # noqa: E731
for _ in range(3):
    f()
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    f()
b.record(); torch.cuda.synchronize()
print(os.environ.get("DMX_ATTN_PROBE_BITS", "0"), "%.1f us" % (a.elapsed_time(b) / 20 * 1e3), flush=True)
