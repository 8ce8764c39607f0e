"""Chain mode 1 (xf_chain.hip) with hot operands (same buffers every launch), with COLD weights / activations (rotating over more than the
Infinity Cache holds) and with cold operands whose weights another kernel has just read: what the in-situ launch (108 us against 85 us
stand-alone) pays for cold weights, and what a weight prefetch could give back.  Measurement aid (EXPERIMENTS.md round 4)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffute_amd import ops  # noqa: E402

dev = torch.device("cuda")
M, C = 16384, 320
g = torch.Generator(device="cpu").manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
bfl = lambda v: v.to(torch.bfloat16).contiguous()
NW, NA = 96, 12                                        # weight sets (2.9 MB each), activation sets (31 MB each)
W = []
for i in range(NW):
    W.append(dict(wo=bfl(rn(C, C, sc=1 / math.sqrt(C))), wp=bfl(rn(C, C, sc=1 / math.sqrt(C))),
                  w1=ops.pack_linear_weight(rn(8 * C, C, sc=1 / math.sqrt(C)), geglu=True), w2=bfl(rn(C, 4 * C, sc=1 / math.sqrt(4 * C)))))
A = [dict(a=bfl(rn(M, C)), h0=bfl(rn(M, C)), xres=bfl(rn(M, C))) for _ in range(NA)]
bo, b2, bp = rn(C, sc=0.1), rn(C, sc=0.1), rn(C, sc=0.1)
c1f, c2f = rn(8 * C, sc=0.1), rn(8 * C, sc=0.1)


def run(i, cold_w, cold_a, prefetch):
    w = W[i % NW] if cold_w else W[0]
    x = A[i % NA] if cold_a else A[0]
    if prefetch:
        for k in ("wo", "wp", "w1", "w2"):
            w[k].view(torch.int16).sum()
    return ops.xf_chain(1, x["a"], x["h0"], w["wo"], bo, c1f, c2f, wf1=w["w1"], wf2=w["w2"], bf2=b2, wpo=w["wp"], bpo=bp, xres=x["xres"])


def timed(cold_w, cold_a, prefetch, reps=48):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(3):
            run(i, cold_w, cold_a, prefetch)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for i in range(reps):
                run(i, cold_w, cold_a, prefetch)
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); gr.replay(); gr.replay(); e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (2 * reps)


base_pf = None
for (cw, ca, pf, name) in [(False, False, False, "hot weights, hot activations"), (True, False, False, "cold weights, hot activations"),
                           (False, True, False, "hot weights, cold activations"), (True, True, False, "cold weights, cold activations"),
                           (True, True, True, "cold + the weights read by four small kernels just before (their time included)")]:
    print(f"chain mode 1, M = {M}: {name}: {timed(cw, ca, pf):.1f} us", flush=True)
# the four small reads alone
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    gr = torch.cuda.CUDAGraph()
    for k in ("wo", "wp", "w1", "w2"):
        W[0][k].view(torch.int16).sum()
    with torch.cuda.graph(gr, stream=s):
        for i in range(48):
            for k in ("wo", "wp", "w1", "w2"):
                W[i % NW][k].view(torch.int16).sum()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s); gr.replay(); gr.replay(); e1.record(s); torch.cuda.synchronize()
print(f"the four weight reads alone: {e0.elapsed_time(e1) * 1e3 / 96:.1f} us per set")
