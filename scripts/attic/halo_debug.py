"""Locate errors of the halo conv: error maps over pixels / channels for a few small cases.   python scripts/halo_debug.py"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch.nn.functional as F
from diffute_amd import ops
from util import bf, seeded
dev = torch.device("cuda:0")
def run(B, H, W, Cin, N, split, gn=False, bn=0):
    x = bf(seeded((B, Cin, H, W), 1)); w = bf(seeded((N, Cin, 3, 3), 3, 1 / math.sqrt(9 * Cin)))
    X = ops.nchw_to_nhwc_bf16(x.to(dev))
    kw = {}
    h = x
    if gn:
        g = 1 + 0.1 * seeded((Cin,), 10); be = 0.1 * seeded((Cin,), 11)
        kw = dict(gn=(g.to(dev), be.to(dev), 32, 1e-5, True), st0=ops.colstats(X))
        h = bf(F.silu(F.group_norm(x, 32, g, be, 1e-5)))
    ref = bf(F.conv2d(h, w, None, padding=1))
    out = ops.nhwc_bf16_to_nchw(ops.conv3x3_gn(X, ops.pack_conv_weight(w.to(dev)), N, force_split=split, force_bn=bn, **kw)).cpu()
    err = (out - ref).abs()
    bad = err > 0.05
    print(f"B={B} {H}x{W} Cin={Cin} N={N} split={split} gn={gn} bn={bn}: rel-L2 {float((out-ref).norm()/ref.norm()):.3e}, bad {int(bad.sum())} of {bad.numel()}")
    if bad.any():
        print("  bad per sample:", bad.sum((1, 2, 3)).tolist())
        print("  bad per channel block of 16:", bad.sum((0, 2, 3)).reshape(-1, 16).sum(1).tolist())
        print("  bad per row y:", bad.sum((0, 1, 3)).tolist())
        print("  bad per col x:", bad.sum((0, 1, 2)).tolist())
for args in [(1, 8, 32, 64, 160, 1), (1, 8, 32, 192, 160, 2), (1, 8, 32, 192, 160, 1, False, 80), (1, 8, 32, 192, 160, 2, False, 80), (1, 8, 32, 192, 160, 2, True, 80), (2, 32, 32, 320, 640, 2, True, 80), (1, 16, 16, 640, 320, 4, True, 80), (1, 32, 32, 128, 128, 2, False, 64)]:
    run(*args)
