"""Launches the S = 4096 self-attention and the three transformer chain kernels a few times each (for rocprofv3 --pmc passes)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffute_amd import ops  # noqa: E402

dev = torch.device("cuda")
M, C = 16384, 320
g = torch.Generator(device="cpu").manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
bfl = lambda v: v.to(torch.bfloat16).contiguous()
a, h0, xres = bfl(rn(M, C)), bfl(rn(M, C)), bfl(rn(M, C))
wo, wq, wp = (bfl(rn(C, C, sc=1 / math.sqrt(C))) for _ in range(3))
bo, c1, c2, b2, bp = rn(C, sc=0.1), rn(C, sc=0.1), rn(C, sc=0.1), rn(C, sc=0.1), rn(C, sc=0.1)
w1 = ops.pack_linear_weight(rn(8 * C, C, sc=1 / math.sqrt(C)), geglu=True)
c1f, c2f = rn(8 * C, sc=0.1), rn(8 * C, sc=0.1)
w2 = bfl(rn(C, 4 * C, sc=1 / math.sqrt(4 * C)))
wqkv = bfl(rn(3 * C, C, sc=1 / math.sqrt(C))); c1q, c2q = rn(3 * C, sc=0.1), rn(3 * C, sc=0.1)
qkv = bfl(rn(4 * 4096, 3 * C))
for _ in range(6):
    ops.xf_chain(2, a, None, wo, bo, c1q, c2q, w1=wqkv)
    ops.xf_chain(0, a, h0, wo, bo, c1, c2, w1=wq)
    ops.xf_chain(1, a, h0, wo, bo, c1f, c2f, wf1=w1, wf2=w2, bf2=b2, wpo=wp, bpo=bp, xres=xres)
    ops.attention_v(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], 4, 5, 4096, 4096, 0.125)
torch.cuda.synchronize()
print("done")
