"""Per-block phase timeline of the halo conv (HaloConvArgs.timing): prologue / K loop / publish / wait / epilogue, in us.
    python scripts/halo_timeline.py [B H W C0 C1 N Csc] [--split S] [--dbg D]"""
import argparse, math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffute_amd import ops, _cabi
ap = argparse.ArgumentParser(); ap.add_argument("shape", nargs="*", type=int, default=[4, 64, 64, 320, 0, 320, 0])
ap.add_argument("--split", type=int, default=0); ap.add_argument("--bn", type=int, default=0); ap.add_argument("--dbg", type=int, default=0); ap.add_argument("--waves", type=int, default=0); ap.add_argument("--nogn", action="store_true"); ap.add_argument("--peers", type=int, default=1)
a = ap.parse_args()
B, H, W, C0, C1, N, Csc = a.shape
dev = torch.device("cuda:0")
_cabi.lib().dmx_set_halo_peers(a.peers)
Cin = C0 + C1; K = 9 * Cin + Csc
x0 = torch.randn(B, H, W, C0, device=dev).to(torch.bfloat16)
x1 = torch.randn(B, H, W, C1, device=dev).to(torch.bfloat16) if C1 else None
sc = torch.randn(B, H, W, Csc, device=dev).to(torch.bfloat16) if Csc else None
w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
b = torch.randn(N, device=dev); te = torch.randn(B, N, device=dev)
r = None if Csc else torch.randn(B, H, W, N, device=dev).to(torch.bfloat16)
g = torch.ones(Cin, device=dev); be = torch.zeros(Cin, device=dev)
st0 = ops.colstats(x0); st1 = ops.colstats(x1) if C1 else None
tm = torch.zeros(4096 * 3, 8, dtype=torch.int64, device=dev)
kw = dict(x1=x1, sc0=sc, bias=b, rowbias=te, res=r, out_stats=True, force_split=a.split, force_bn=a.bn, force_waves=a.waves, dbg=a.dbg)
if not a.nogn: kw.update(gn=(g, be, 32, 1e-5, True), st0=st0, st1=st1)
for i in range(3): ops.conv3x3_gn(x0, w, N, **kw)
ops.conv3x3_gn(x0, w, N, timing=tm, **kw)
torch.cuda.synchronize()
tall = tm.cpu()
word7 = tall[:4096, 7].clone()
tall[:4096, 7] = word7 & 0xffffffff                 # taps of the block's slice; above: exchange through L2 | XCC id + 1 | slice
live = tall[:4096, 0] > 0
print(f"peers={a.peers}: blocks exchanging through their XCD's L2: {int(((word7[live] >> 32) & 1).sum())} of {int(live.sum())}; XCC ids seen: {sorted(set((((word7[live] >> 40) & 0xff) - 1).tolist()))}")
ph8 = tall[4096:].reshape(-1, 2, 8)
ph = ph8[:, :, :5].double()
t = tall[:4096]
t = t[t[:, 0] > 0]
ph = ph[: len(t)]
for gi, gname in ((0, "wave 0 (group A)"), (1, "wave 4 (group B)")):
    v = ph[:, gi]
    print(f"  {gname}: shader cycles per pipeline step (mean over blocks): DMA phase %.0f  barrier X %.0f  MFMA phase %.0f  barrier Y %.0f   (sum %.0f)" %
          tuple((v[:, k] / 27.0).mean().item() if False else (v[:, k].mean().item() / max(1.0, 1.0)) for k in range(4)) + (0,)) if False else None
    steps = (v[:, 4] - 0).clamp_min(1)
    per = v[:, :4] / tall[:4096][tall[:4096][:, 0] > 0][: len(v), 7:8].double().clamp_min(1)      # per tap of the block's own slice
    print(f"  {gname}: cycles per pipeline step: DMA phase {per[:,0].mean():.0f}  barrier X {per[:,1].mean():.0f}  MFMA phase {per[:,2].mean():.0f}  barrier Y {per[:,3].mean():.0f}  sum {per.sum(1).mean():.0f}   (waves 0 / 4 of the warp-specialised instances: compute / loader - stream | barrier ; issue | barrier | normalise | end waits)")
t0 = t[:, 0].min()
t = t[:, [0, 1, 2, 3, 4, 6, 5, 7]]          # (stamp 6 = items done, before the statistics fold; 5 = end)
us = (t[:, :7] - t0).double() / 100.0
print(f"{len(t)} blocks; steps per block {sorted(set(t[:, 7].tolist()))}")
pro = (ph8[: len(t), 0, 5:8] - t[:, 0:1]).double() / 100.0
print("  prologue (us from block start, mean): requests issued %.2f  statistics done %.2f  patch landed + barrier %.2f" % tuple(pro.mean(0).tolist()))
names = ["start", "prologue done", "K loop done", "published", "peers arrived", "items done", "end"]
for i, n in enumerate(names):
    print(f"  {n:15s}: mean {us[:, i].mean():7.2f}  min {us[:, i].min():7.2f}  max {us[:, i].max():7.2f} us")
d = us[:, 1:] - us[:, :-1]
print("  phase durations (mean): prologue %.2f  K loop %.2f  publish %.2f  wait+stage %.2f  items %.2f  statistics %.2f" % tuple(d.mean(0).tolist()))
steps = t[:, 7].double()
print("  K loop per step: mean %.3f us (min %.3f, max %.3f)" % (float((d[:, 1] / steps).mean()), float((d[:, 1] / steps).min()), float((d[:, 1] / steps).max())))
