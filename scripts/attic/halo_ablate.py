"""Ablation timing of the halo conv K loop (HaloConvArgs.dbg; results invalid for dbg != 0): whole-kernel time per variant.
    python scripts/halo_ablate.py [B H W C0 C1 N Csc]"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffute_amd import ops
sh = [int(v) for v in sys.argv[1:8]] if len(sys.argv) >= 8 else [4, 64, 64, 320, 0, 320, 0]
B, H, W, C0, C1, N, Csc = sh
dev = torch.device("cuda:0")
Cin = C0 + C1; K = 9 * Cin + Csc
x0 = torch.randn(B, H, W, C0, device=dev).to(torch.bfloat16)
x1 = torch.randn(B, H, W, C1, device=dev).to(torch.bfloat16) if C1 else None
sc = torch.randn(B, H, W, Csc, device=dev).to(torch.bfloat16) if Csc else None
ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16) for _ in range(4)]
b = torch.randn(N, device=dev); te = torch.randn(B, N, device=dev)
r = None if Csc else torch.randn(B, H, W, N, device=dev).to(torch.bfloat16)
g = torch.ones(Cin, device=dev); be = torch.zeros(Cin, device=dev)
st0 = ops.colstats(x0); st1 = ops.colstats(x1) if C1 else None
names = {0: "full", 1: "no MFMA", 2: "no weight DMA", 4: "no normalise", 8: "no patch DMA", 3: "no MFMA, no weight DMA", 7: "no MFMA / W DMA / normalise", 15: "skeleton (barriers only)", 31: "no barriers either",
         14: "MFMA only (+barriers)", 30: "MFMA only, no barriers", 16: "full, no barriers"}
for dbg in (0, 1, 2, 4, 8, 3, 7, 15, 31, 14, 30, 16):
    def fn(i):
        return ops.conv3x3_gn(x0, ws[i % 4], N, x1=x1, gn=(g, be, 32, 1e-5, True), st0=st0, st1=st1, sc0=sc, bias=b, rowbias=te, res=r, out_stats=True, dbg=dbg)
    for i in range(2): fn(i)
    torch.cuda.synchronize()
    ts = []
    for i in range(6):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); fn(i); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"dbg {dbg:2d} {names[dbg]:32s}: min {min(ts):7.1f} us (call incl. memset + launch)", flush=True)
