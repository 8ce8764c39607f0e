import sys, math, torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from util import bf, seeded
from diffute_amd import ops
import torch.nn.functional as F
dev = torch.device("cuda:0")
def run(B, H, W, C0, N, fS, gn=True, dbg=0):
    x0 = bf(seeded((B, C0, H, W), 1) * 1.5 + 0.3)
    w = bf(seeded((N, C0, 3, 3), 3, 1 / math.sqrt(9 * C0))); b = seeded((N,), 4, 0.1)
    g = 1 + 0.1 * seeded((C0,), 10); be = 0.1 * seeded((C0,), 11)
    X0 = ops.nchw_to_nhwc_bf16(x0.to(dev))
    W_ = ops.pack_conv_weight(w.to(dev))
    WP = ops.skinny_pack(W_, [(C0, 9, C0, 0)])
    sg = dict(x=X0, taps=9)
    if gn:
        sg.update(st=ops.colstats(X0), gamma=g.to(dev), beta=be.to(dev), gn_c0=0)
    out = ops.skinny_conv([sg], WP, N, gn=(32, C0, 1e-5, True) if gn else None, bias=b.to(dev), force_S=fS, dbg=dbg)
    torch.cuda.synchronize()
    o = out.float().reshape(B * H * W, N)
    bad = ~torch.isfinite(o)
    h = F.silu(F.group_norm(x0, 32, g, be, 1e-5)) if gn else x0
    ref = F.conv2d(bf(h), w, b, padding=1).permute(0, 2, 3, 1).reshape(B * H * W, N)
    err = (o.cpu() - ref).norm() / ref.norm() if not bad.any() else float('nan')
    print(f"dbg{dbg} B{B} {H}x{W} C{C0} N{N} S{fS} gn{gn}: nonfinite {int(bad.sum())} rows {bad.any(1).nonzero().flatten()[:10].tolist()} cols {bad.any(0).nonzero().flatten()[:10].tolist()} err {err}")
for dbg in (0, 64, 8, 16, 32, 16 + 32):
    for rep in range(2):
        run(4, 8, 8, 1280, 1280, 0, gn=True, dbg=dbg)
