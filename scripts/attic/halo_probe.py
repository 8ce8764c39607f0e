"""Stand-alone timing of the halo conv (conv_halo.hip) against GroupNorm + the implicit-GEMM conv on the UNet / VAE shapes.
    python scripts/halo_probe.py [--reps 30]"""
import argparse, math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffute_amd import ops

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=30); ap.add_argument("--split", type=int, default=0); ap.add_argument("--bn", type=int, default=0); ap.add_argument("--waves", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    shapes = [  # B, H, W, C0, C1, N, Csc
        (4, 64, 64, 320, 0, 320, 0), (4, 64, 64, 320, 320, 320, 0), (4, 64, 64, 640, 320, 320, 0), (4, 64, 64, 320, 0, 320, 640),
        (4, 32, 32, 640, 0, 640, 0), (4, 32, 32, 320, 0, 640, 0), (4, 32, 32, 1280, 640, 640, 0), (4, 32, 32, 640, 0, 640, 1280),
        (4, 16, 16, 1280, 0, 1280, 0), (4, 16, 16, 1280, 1280, 1280, 0), (4, 16, 16, 640, 0, 1280, 0),
        (8, 128, 128, 256, 0, 256, 0), (8, 256, 256, 128, 0, 128, 0),
    ]
    for (B, H, W, C0, C1, N, Csc) in shapes:
        Cin = C0 + C1
        x0 = torch.randn(B, H, W, C0, device=dev).to(torch.bfloat16)
        x1 = torch.randn(B, H, W, C1, device=dev).to(torch.bfloat16) if C1 else None
        sc = torch.randn(B, H, W, Csc, device=dev).to(torch.bfloat16) if Csc else None
        K = 9 * Cin + Csc
        ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16) for _ in range(4)]      # rotate: weights do not stay in L2
        b = torch.randn(N, device=dev); te = torch.randn(B, N, device=dev)
        r = None if Csc else torch.randn(B, H, W, N, device=dev).to(torch.bfloat16)
        g = torch.ones(Cin, device=dev); be = torch.zeros(Cin, device=dev)
        st0 = ops.colstats(x0); st1 = ops.colstats(x1) if C1 else None
        def halo(i):
            return ops.conv3x3_gn(x0, ws[i % 4], N, x1=x1, gn=(g, be, 32, 1e-5, True), st0=st0, st1=st1, sc0=sc, bias=b, rowbias=te, res=r, out_stats=True, force_split=a.split, force_bn=a.bn, force_waves=a.waves)
        def old(i):
            t = ops.groupnorm(x0, g, be, 32, 1e-5, True, x1=x1)
            return ops.conv_gemm(t, ws[i % 4], N, sc0=sc, bias=b, rowbias=te, res=r)
        res = {}
        try:
            halo(0)
        except RuntimeError:
            continue                                   # (forced tile / wave layout does not take this shape)
        for name, fn in (("halo", halo), ("gn+gemm", old)):
            for i in range(3): fn(i)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.reps): fn(i)
            e1.record(); torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) / a.reps * 1e3
        fl = 2.0 * B * H * W * N * K
        print(f"B={B} {H}x{W} Cin={Cin} N={N} Csc={Csc} K={K}: halo {res['halo']:.1f} us ({fl / res['halo'] / 1e6:.0f} TF/s)   gn+gemm {res['gn+gemm']:.1f} us  (eager launches incl. host overhead)", flush=True)

if __name__ == "__main__":
    main()
