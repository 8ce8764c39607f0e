"""Stand-alone timing of the weight-streaming conv (skinny.hip) against the path it replaces (GroupNorm launch + tiled implicit-GEMM conv with its
split-K reduce), on the shapes of the 8x8 / 16x16 levels.  Weights rotate over enough copies to defeat the 256-MB memory-side cache (in the pass
every layer's weights are cold).  us per call from hipEvents around `reps` back-to-back calls.

    python scripts/skinny_probe.py [--reps 30] [--force-S n]"""
import argparse
import json
import math
import sys

import torch

sys.path.insert(0, ".")
from diffute_amd import ops, _cabi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--force-S", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    shapes = [  # name, B, H, W, C0, C1, N, sc
        ("b4 8x8 1280->1280 K=11520", 4, 8, 8, 1280, 0, 1280, 0),
        ("b4 8x8 2560->1280 K=23040", 4, 8, 8, 1280, 1280, 1280, 0),
        ("b4 8x8 conv2+sc K=14080", 4, 8, 8, 1280, 0, 1280, 2560),
        ("b1 8x8 1280->1280 K=11520", 1, 8, 8, 1280, 0, 1280, 0),
        ("b1 16x16 1280->1280 K=11520", 1, 16, 16, 1280, 0, 1280, 0),
        ("b1 16x16 2560->1280 K=23040", 1, 16, 16, 1280, 1280, 1280, 0),
        ("b2 8x8 1280->1280", 2, 8, 8, 1280, 0, 1280, 0),
    ]
    res = []
    for name, B, H, W, C0, C1, N, SC in shapes:
        Cin = C0 + C1
        K = 9 * Cin + SC
        ncopy = max(2, int(math.ceil(300e6 / (N * K * 2))))
        x0 = (torch.randn(B, H, W, C0, device=dev, generator=g) * 1.5).to(ops.h16())
        x1 = (torch.randn(B, H, W, C1, device=dev, generator=g)).to(ops.h16()) if C1 else None
        sc = (torch.randn(B, H, W, SC, device=dev, generator=g)).to(ops.h16()) if SC else None
        Ws = [(torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)).to(ops.h16()) for _ in range(ncopy)]
        pk = [(C0, 9, Cin, 0)] + ([(C1, 9, Cin, C0)] if C1 else []) + ([(SC, 1, 0, 9 * Cin)] if SC else [])
        WPs = [ops.skinny_pack(w, pk) for w in Ws]
        bias = torch.randn(N, device=dev, generator=g); temb = torch.randn(B, N, device=dev, generator=g)
        gam = 1 + 0.1 * torch.randn(Cin, device=dev, generator=g); bet = 0.1 * torch.randn(Cin, device=dev, generator=g)
        st0 = ops.colstats(x0); st1 = ops.colstats(x1) if C1 else None

        def segs(gn):
            s = [dict(x=x0, taps=9)]
            if gn:
                s[0].update(st=st0, gamma=gam[:C0].contiguous(), beta=bet[:C0].contiguous(), gn_c0=0)
            if C1:
                s.append(dict(x=x1, taps=9))
                if gn:
                    s[1].update(st=st1, gamma=gam[C0:].contiguous(), beta=bet[C0:].contiguous(), gn_c0=C0)
            if SC:
                s.append(dict(x=sc, taps=1))
            return s
        sg_plain, sg_gn = segs(False), segs(True)

        def t_of(fn):
            """us per call inside a captured graph of `reps` calls (no host launch overhead), best of 3 replays"""
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for i in range(3):
                    fn(i)
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=st):
                    for i in range(args.reps):
                        fn(i)
                best = 1e9
                for _ in range(3):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(st); gr.replay(); b.record(st); torch.cuda.synchronize()
                    best = min(best, a.elapsed_time(b) * 1e3 / args.reps)
            _cabi.poll_device_error()
            return best
        kw = dict(bias=bias, rowbias=temb, out_stats=True, force_S=args.force_S)
        t_sk = t_of(lambda i: ops.skinny_conv(sg_plain, WPs[i % ncopy], N, **kw))
        t_skgn = t_of(lambda i: ops.skinny_conv(sg_gn, WPs[i % ncopy], N, gn=(32, Cin, 1e-5, True), **kw))
        t_gemm = t_of(lambda i: ops.conv_gemm(x0, Ws[i % ncopy], N, x1=x1, bias=bias, rowbias=temb, sc0=sc))
        t_gn = t_of(lambda i: ops.groupnorm(x0, gam, bet, 32, 1e-5, True, x1=x1))
        # phase timeline of one launch (10 ns ticks -> us): start .. prologue | K loop | publish | wait | finish
        tim = torch.zeros(2048 * 6, dtype=torch.int64, device=dev)
        ops.skinny_conv(sg_gn, WPs[0], N, gn=(32, Cin, 1e-5, True), timing=tim, **kw)
        torch.cuda.synchronize()
        T = tim.view(-1, 6).cpu().double()
        T = T[T[:, 0] > 0]
        t0 = T[:, 0].min()
        ph = dict(blocks=int(T.shape[0]), start_spread=round(float((T[:, 0] - t0).max()) / 100, 2), prologue=round(float((T[:, 1] - T[:, 0]).mean()) / 100, 2),
                  kloop=round(float((T[:, 2] - T[:, 1]).mean()) / 100, 2), kloop_max=round(float((T[:, 2] - T[:, 1]).max()) / 100, 2),
                  publish=round(float((T[:, 3] - T[:, 2]).mean()) / 100, 2), wait=round(float((T[:, 4] - T[:, 3]).mean()) / 100, 2),
                  finish=round(float((T[:, 5] - T[:, 4]).mean()) / 100, 2), span=round(float((T[:, 5].max() - t0)) / 100, 2))
        def phases(sg, gn, dbg):
            tim.zero_()
            ops.skinny_conv(sg, WPs[1], N, gn=gn, timing=tim, dbg=dbg, **kw)
            torch.cuda.synchronize()
            T_ = tim.view(-1, 6).cpu().double(); T_ = T_[T_[:, 0] > 0]
            return [round(float((T_[:, i + 1] - T_[:, i]).mean()) / 100, 2) for i in range(5)]
        ph["plain"] = phases(sg_plain, None, 0)
        ph["plain_nomfma"] = phases(sg_plain, None, 1)
        ph["plain_norefill"] = phases(sg_plain, None, 2)
        ph["plain_nolds"] = phases(sg_plain, None, 4)
        ph["plain_only_loads"] = phases(sg_plain, None, 5)
        wb = N * K * 2
        r = dict(shape=name, skinny_us=round(t_sk, 1), skinny_gn_us=round(t_skgn, 1), gemm_us=round(t_gemm, 1), groupnorm_us=round(t_gn, 1),
                 weight_MB=round(wb / 1e6, 1), skinny_TBps=round(wb / t_sk / 1e6, 2), hbm_floor_us=round(wb / 6.3e6, 1), gn_phases_us=ph)
        print(json.dumps(r), flush=True)
        res.append(r)
        del Ws, WPs


if __name__ == "__main__":
    main()
