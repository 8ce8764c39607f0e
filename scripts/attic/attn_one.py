import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffute_amd import ops
dev = torch.device("cuda")
B, H, S = 4, 5, 4096
C = H * 64
qkv = torch.randn(B * S, 3 * C, device=dev).to(torch.bfloat16)
for _ in range(5):
    ops.attention_v(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, H, S, S, 0.125)
torch.cuda.synchronize()
