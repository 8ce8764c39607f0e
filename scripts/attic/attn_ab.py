"""A/B of two builds of the d=64 attention kernel (DIFFUTE_HIP_LIB): times every shape of the UNet pass and writes / compares the outputs
(python scripts/attn_ab.py dump file.pt | cmp file.pt).  Measurement aid."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffute_amd import ops
dev = torch.device("cuda")
mode, path = sys.argv[1], sys.argv[2]
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
outs, line = {}, os.environ.get("DIFFUTE_HIP_LIB", "in-tree") + ":"
SH = [(4, 5, 4096, 4096), (4, 5, 4096, 577), (4, 10, 1024, 1024), (4, 10, 1024, 577), (4, 20, 256, 256), (4, 20, 256, 577), (4, 20, 64, 64), (4, 20, 64, 577),
      (2, 5, 9216, 9216), (1, 5, 4096, 4096), (3, 5, 100, 130), (2, 5, 4096, 4000)]
for (B, H, Sq, Skv) in SH:
    C = H * 64
    pad = (Skv + 63) // 64 * 64
    g = torch.Generator(device=dev).manual_seed(Sq * 7 + Skv)
    q = (torch.randn(B * Sq, C, device=dev, generator=g) * 1.5).to(torch.bfloat16)
    kv = torch.randn(B * pad, 2 * C, device=dev, generator=g).to(torch.bfloat16)
    kv[:, :C] *= 2.0                                   # spread of scores ~ 3 after the 1/8 scale: rescales happen
    o = ops.attention_v(q, kv[:, :C], kv[:, C:], B, H, Sq, Skv, 0.125, kv_rows=pad)
    outs[(B, H, Sq, Skv)] = o.cpu()
    t = bench(lambda: ops.attention_v(q, kv[:, :C], kv[:, C:], B, H, Sq, Skv, 0.125, kv_rows=pad))
    line += f"  {Sq}x{Skv}(B{B}H{H}): {t:6.1f}"
print(line, flush=True)
if mode == "dump":
    torch.save(outs, path)
else:
    ref = torch.load(path)
    for k, o in outs.items():
        r = ref[k]
        fin = bool(torch.isfinite(o.float()).all())
        print(k, "bit-equal" if torch.equal(o, r) else f"DIFFERENT: rel-L2 {float((o.float() - r.float()).norm() / r.float().norm()):.3e}", "" if fin else "NON-FINITE")
