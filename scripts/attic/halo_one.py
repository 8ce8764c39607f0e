"""One halo-conv shape, N launches (for rocprofv3 --pmc / --kernel-trace).   python scripts/halo_one.py B H W C0 C1 N Csc [reps] [split] [old]"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffute_amd import ops
B, H, W, C0, C1, N, Csc = [int(v) for v in sys.argv[1:8]]
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 10
split = int(sys.argv[9]) if len(sys.argv) > 9 else 0
bn = int(sys.argv[10]) if len(sys.argv) > 10 else 0
old = len(sys.argv) > 11
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
for i in range(reps):
    if old:
        t = ops.groupnorm(x0, g, be, 32, 1e-5, True, x1=x1)
        ops.conv_gemm(t, ws[i % 4], N, sc0=sc, bias=b, rowbias=te, res=r)
    else:
        ops.conv3x3_gn(x0, ws[i % 4], N, x1=x1, gn=(g, be, 32, 1e-5, True), st0=st0, st1=st1, sc0=sc, bias=b, rowbias=te, res=r, out_stats=True, force_split=split, force_bn=bn)
torch.cuda.synchronize()
