"""one skinny conv launch set (for rocprofv3 --pmc): b4 8x8 1280->1280, plain and with GroupNorm"""
import math, sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffute_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
B, H, W, C0, N = 4, 8, 8, 1280, 1280
x0 = (torch.randn(B, H, W, C0, device=dev, generator=g) * 1.5).to(ops.h16())
Wt = (torch.randn(N, 9 * C0, device=dev, generator=g) / math.sqrt(9 * C0)).to(ops.h16())
WP = ops.skinny_pack(Wt, [(C0, 9, C0, 0)])
bias = torch.randn(N, device=dev, generator=g)
dbg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for _ in range(5):
    ops.skinny_conv([dict(x=x0, taps=9)], WP, N, bias=bias, dbg=dbg)
torch.cuda.synchronize()
