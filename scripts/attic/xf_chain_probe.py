"""Stand-alone timing of the transformer chain kernels (xf_chain.hip) at the 64x64-level shape, graph-captured, against the
separate GEMMs they replace.  Measurement aid."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffute_amd import ops  # noqa: E402

dev = torch.device("cuda")
M = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16384
C = 320
g = torch.Generator(device="cpu").manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
bfl = lambda v: v.to(torch.bfloat16).contiguous()
a, h0, xres = bfl(rn(M, C)), bfl(rn(M, C)), bfl(rn(M, C))
wo, wq, wp = (bfl(rn(C, C, sc=1 / math.sqrt(C))) for _ in range(3))
bo, c1, c2, b2, bp = rn(C, sc=0.1), rn(C, sc=0.1), rn(C, sc=0.1), rn(C, sc=0.1), rn(C, sc=0.1)
w1 = ops.pack_linear_weight(rn(8 * C, C, sc=1 / math.sqrt(C)), geglu=True)
c1f, c2f = rn(8 * C, sc=0.1), rn(8 * C, sc=0.1)
w2 = bfl(rn(C, 4 * C, sc=1 / math.sqrt(4 * C)))


def graph_time(fn, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(reps):
                fn()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); [gr.replay() for _ in range(5)]; e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


wqkv = bfl(rn(3 * C, C, sc=1 / math.sqrt(C))); c1q, c2q = rn(3 * C, sc=0.1), rn(3 * C, sc=0.1)
t2 = graph_time(lambda: ops.xf_chain(2, a, None, wo, bo, c1q, c2q, w1=wqkv))


def separate2():
    h, st = ops.linear(a, wo, bias=bo, rowstats=True)
    return ops.linear(h, wqkv, ln=(st, c1q, c2q, 1e-5))


print(f"M={M}: chain mode 2 (proj_in -> LN -> qkv) {t2:.1f} us (separate {graph_time(separate2):.1f})")
t0 = graph_time(lambda: ops.xf_chain(0, a, h0, wo, bo, c1, c2, w1=wq))
t1 = graph_time(lambda: ops.xf_chain(1, a, h0, wo, bo, c1f, c2f, wf1=w1, wf2=w2, bf2=b2, wpo=wp, bpo=bp, xres=xres))


def separate0():
    h, st = ops.linear(a, wo, bias=bo, res=h0, rowstats=True)
    return ops.linear(h, wq, ln=(st, c1, c2, 1e-5))


def separate1():
    h, st = ops.linear(a, wo, bias=bo, res=h0, rowstats=True)
    gg = ops.linear(h, w1, geglu=True, ln=(st, c1f, c2f, 1e-5))
    h3 = ops.linear(gg, w2, bias=b2, res=h)
    return ops.linear(h3, wp, bias=bp, res=xres)


try:
    s0, s1 = graph_time(separate0), graph_time(separate1)
except Exception as e:          # (the ln= tuple protocol of ops.linear may differ: the chain numbers are what this script is for)
    print("separate path not timed:", e); s0 = s1 = float("nan")
print(f"M={M}: chain mode 0 {t0:.1f} us (separate {s0:.1f}), chain mode 1 {t1:.1f} us (separate {s1:.1f})")
fl1 = 2.0 * M * C * C * 2 + 2.0 * M * C * 8 * C + 2.0 * M * 4 * C * C
print(f"mode 1: {fl1 / t1 * 1e-6:.0f} TFLOP/s; weight stream per block 2.96 MB -> {2.96e6 / (t1 * 1e-6) * 1e-9:.0f} GB/s per CU if one round")

# ---- per-block phase timeline (s_memrealtime stamps) with the ablation bits of dmx_xf_chain_desc.dbg
nb = M // 64
for mode, names in ((0, ["load x", "gemm1", "epi1+exchange", "gemm2", "", "", "epi2+store"]),
                    (1, ["load x", "gemm1", "epi1+exchange", "ff loop (64 tiles)", "epi3+exchange", "proj_out", "epi4+store"])):
    for dbg in (0, 1, 2, 3):
        if dbg and '--ablate' not in sys.argv:          # (the ablation variants exist in -DDMX_PROBES builds only)
            continue
        tim = torch.zeros(nb, 8, dtype=torch.int64, device=dev)
        for _ in range(3):
            if mode == 0:
                ops.xf_chain(0, a, h0, wo, bo, c1, c2, w1=wq, dbg=dbg, timing=tim)
            else:
                ops.xf_chain(1, a, h0, wo, bo, c1f, c2f, wf1=w1, wf2=w2, bf2=b2, wpo=wp, bpo=bp, xres=xres, dbg=dbg, timing=tim)
        torch.cuda.synchronize()
        tt = tim.cpu().double() * 0.01
        t0 = tt[:, 0].min()
        seg = [(tt[:, i + 1] - tt[:, i]).mean().item() for i in range(7)]
        if mode == 0:
            seg = seg[:4] + [0, 0] + [(tt[:, 7] - tt[:, 4]).mean().item()]
        print(f"mode {mode} dbg {dbg}: span {(tt[:, 7] - t0).max():6.1f} us, start spread {(tt[:, 0] - t0).max():4.1f} | " +
              ", ".join(f"{n} {v:.1f}" for n, v in zip(names, seg) if n), flush=True)
