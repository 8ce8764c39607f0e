"""Stand-alone A/B of the halo conv's K-split slab exchange: peers of a tile on one XCD exchanging through its L2 (dmx_set_halo_peers(1)) against the
round-5 dealing with write-through slabs (0).  us per launch inside a captured graph of `reps` launches, both settings alternating; with --once it
launches each setting `reps` times eagerly (for scripts/pmc_kernel.sh: FETCH_SIZE / WRITE_SIZE per launch, first half = peers off, second half = on).

    python scripts/attic/halo_peers_probe.py [--reps 20] [--once]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffute_amd import ops, _cabi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--once", action="store_true")
ap.add_argument("--peers", type=int, default=-1, help="with --once: only this setting")
args = ap.parse_args()
dev = torch.device("cuda:0")
lib = _cabi.lib()
g = torch.Generator(device=dev).manual_seed(1)
for (B, H, W, C, N, split) in ((4, 64, 64, 320, 320, 2), (4, 64, 64, 640, 320, 2), (4, 32, 32, 640, 640, 4), (4, 32, 32, 1280, 640, 4), (4, 16, 16, 1280, 1280, 8)):
    x = (torch.randn(B, H, W, C, device=dev, generator=g) * 1.2).to(ops.h16())
    w = (torch.randn(N, C, 3, 3, device=dev, generator=g) / math.sqrt(9 * C))
    W_ = ops.pack_conv_weight(w)
    bias = torch.randn(N, device=dev, generator=g)
    res = torch.randn(B, H, W, N, device=dev, generator=g).to(ops.h16())
    gam = 1 + 0.1 * torch.randn(C, device=dev, generator=g); bet = 0.1 * torch.randn(C, device=dev, generator=g)
    kw = dict(bias=bias, res=res, force_split=split, out_stats=True, gn=(gam, bet, 32, 1e-5, True), st0=ops.colstats(x))
    outs = {}
    line = f"{B}x{H}x{W} Cin {C} N {N} split {split}:"
    for peers in ((args.peers,) if args.peers >= 0 else (0, 1, 0, 1)):
        lib.dmx_set_halo_peers(peers)
        out, _ = ops.conv3x3_gn(x, W_, N, **kw)
        torch.cuda.synchronize()
        if args.once:
            for _ in range(args.reps):
                ops.conv3x3_gn(x, W_, N, **kw)
            torch.cuda.synchronize()
            continue
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            ops.conv3x3_gn(x, W_, N, **kw)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for _ in range(args.reps):
                    o2, _ = ops.conv3x3_gn(x, W_, N, **kw)
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); gr.replay(); gr.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (3 * args.reps) * 1e3
        outs.setdefault(peers, out.clone())
        assert torch.equal(outs[peers], out) and torch.equal(outs[min(outs)], out)
        line += f"  peers={peers} {us:6.1f} us"
    print(line, flush=True)
lib.dmx_set_halo_peers(1)
