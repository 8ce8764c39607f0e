"""The balanced attention schedule (attention_sk.hip) against the plain grid, us per launch inside a captured graph of `reps` launches, alternating.
    python scripts/attic/attn_balanced_probe.py [--reps 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffute_amd import ops, _cabi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda:0")
lib = _cabi.lib()
g = torch.Generator(device=dev).manual_seed(1)
for (B, H, S, Skv, mode) in ((1, 5, 4096, 4096, 2), (1, 10, 1024, 1024, 2), (2, 10, 1024, 1024, 2), (4, 5, 4096, 4096, 2), (16, 5, 4096, 4096, 2), (8, 5, 4096, 4096, 2), (3, 5, 4096, 4096, 1), (2, 5, 4096, 4096, 1), (4, 5, 4096, 577, 2), (4, 10, 1024, 1024, 2), (2, 5, 9216, 9216, 2)):
    C = H * 64
    pad = (Skv + 63) // 64 * 64
    q = torch.randn(B * S, C, device=dev, generator=g).to(ops.h16())
    kv = torch.randn(B * pad, 2 * C, device=dev, generator=g).to(ops.h16())
    lib.dmx_set_attn_balanced(mode)
    wsb = lib.dmx_attention_fwd_v_balanced_workspace_bytes(B, H, S, Skv)
    line = f"B={B} H={H} {S}x{Skv} (mode {mode}, workspace {wsb >> 20} MB):"
    outs = {}
    for which in ("plain", "balanced", "plain", "balanced"):
        if which == "balanced" and not wsb:
            continue
        fn = (lambda: ops.attention_v(q, kv[:, :C], kv[:, C:], B, H, S, Skv, 0.125, kv_rows=pad)) if which == "plain" else \
             (lambda: ops.attention_v_balanced(q, kv[:, :C], kv[:, C:], B, H, S, Skv, 0.125, kv_rows=pad))
        out = fn(); torch.cuda.synchronize()
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            fn()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for _ in range(args.reps):
                    fn()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); gr.replay(); gr.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (3 * args.reps) * 1e3
        if which in outs:
            assert torch.equal(outs[which], out), f"{which}: not bit-repeatable"
        outs[which] = out.clone()
        line += f"  {which} {us:6.1f} us"
    if "balanced" in outs:
        d = (outs["balanced"].float() - outs["plain"].float())
        line += f"   rel-L2 between them {float(d.norm() / outs['plain'].float().norm()):.2e}"
    print(line, flush=True)
lib.dmx_set_attn_balanced(1)
