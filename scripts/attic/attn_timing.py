"""Per-wave cycle accounting of the software-pipelined attention loop (build with -DDMX_ATTN_TIMING -DDMX_ATTN_PIPE_MIN=1: the kernel then
writes, per wave, the shader cycles spent in [top: DMA issue + mask] [slots: MFMAs + softmax] [wait: vmcnt] [barrier] into the lse buffer).
Measurement aid."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffute_amd import ops
dev = torch.device("cuda")
for (B, H, Sq, Skv) in [(4, 5, 4096, 4096), (1, 5, 4096, 4096), (4, 10, 1024, 1024), (4, 5, 4096, 577)]:
    C = H * 64
    pad = (Skv + 63) // 64 * 64
    q = torch.randn(B * Sq, C, device=dev).to(torch.bfloat16)
    kv = torch.randn(B * pad, 2 * C, device=dev).to(torch.bfloat16)
    for _ in range(3):
        o, lse = ops.attention_train(q, kv[:, :C], kv[:, C:], B, H, Sq, Skv, 0.125, kv_rows=pad)
    torch.cuda.synchronize()
    nblk = (Sq // 128) * H * B
    d = lse.flatten()[:nblk * 16].reshape(nblk, 4, 4).double().cpu()
    nt = (Skv + 63) // 64
    m = d.mean((0, 1)) / nt
    print(f"B{B} H{H} {Sq}x{Skv}: cycles per tile and wave: top {m[0]:.0f}  slots {m[1]:.0f}  vmcnt wait {m[2]:.0f}  barrier {m[3]:.0f}  total {m.sum():.0f}   (slowest wave total {d.sum(2).max() / nt:.0f})")
