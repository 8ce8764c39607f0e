"""Reference point for the GEMM plans: every distinct conv/linear GEMM shape of a profiled UNet pass, timed (a) through
this library's automatic plan (implicit-GEMM gather included) and (b) as a plain [M,K]x[K,N] bf16 torch.matmul (the
vendor BLAS: hipBLASLt / rocBLAS) on a pre-materialised matrix of the same size -- i.e. the vendor number has no im2col,
bias, residual or type conversion in it.  Input: the CSV written by `bench.py --profile-csv`.  Development tool."""
import csv
import math
import re
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "scripts")
from diffute_amd import ops  # noqa: E402
from tune_gemm import time_it, weights  # noqa: E402


def main(path):
    shapes = {}
    for r in csv.DictReader(open(path)):
        if int(r["class"]) not in (0, 1, 7, 9) and not 10 <= int(r["class"]) < 22:
            continue
        m = dict(re.findall(r"(\w+)=(\d+)", r["tag"]))
        key = tuple(int(m[k]) for k in ("M", "N", "K", "ks", "st", "ups"))
        shapes[key] = shapes.get(key, 0) + 1
    dev = torch.device("cuda")
    tot_mine = tot_vendor = tot_fl = 0.0
    for (M, N, K, ks, st, ups), cnt in sorted(shapes.items(), key=lambda kv: -kv[1] * kv[0][0] * kv[0][1] * kv[0][2]):
        B = 4 if M % 4 == 0 else 1
        if ks == 3:
            ohw = M // B; OH = int(round(math.sqrt(ohw))); Cin = K // 9
            if Cin * 9 != K:
                ks = 1
        ws = weights(N, K)
        if ks == 3:
            H = OH // 2 if ups else (OH * 2 if st == 2 else OH)
            x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
            mine = lambda i: ops.conv_gemm(x, ws[i % len(ws)], N, ksize=3, stride=st, pad=1, ups=bool(ups))
        else:
            x = torch.randn(1, 1, M, K, device=dev).to(torch.bfloat16)
            mine = lambda i: ops.conv_gemm(x, ws[i % len(ws)], N, ksize=1, pad=0)
        a2d = torch.randn(M, K, device=dev).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        wts = [w.t() for w in ws]                               # [K, N] views of the [N, K] row-major weights (NT GEMM, as ours)
        vendor = lambda i: torch.matmul(a2d, wts[i % len(wts)], out=out)
        t_m = time_it(mine); t_v = time_it(vendor)
        fl = 2.0 * M * N * K
        tot_mine += t_m * cnt; tot_vendor += t_v * cnt; tot_fl += fl * cnt
        print(f"M={M:6d} N={N:5d} K={K:6d} ks={ks} st={st} ups={ups} x{cnt:3d}: ours {t_m:7.1f}us ({fl / t_m / 1e6:6.0f} TF) | "
              f"vendor plain GEMM {t_v:7.1f}us ({fl / t_v / 1e6:6.0f} TF) | ratio {t_v / t_m:5.2f}", flush=True)
    print(f"sum over one UNet forward: ours {tot_mine / 1e3:.2f} ms ({tot_fl / tot_mine / 1e6:.0f} TF/s), vendor plain GEMMs {tot_vendor / 1e3:.2f} ms ({tot_fl / tot_vendor / 1e6:.0f} TF/s)")
    # the vendor library's own ceiling on this box, for scale
    for n in (4096, 8192):
        a = torch.randn(n, n, device=dev).to(torch.bfloat16); b = torch.randn(n, n, device=dev).to(torch.bfloat16); o = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
        t = time_it(lambda i: torch.matmul(a, b.t(), out=o))
        print(f"vendor square {n}^3: {t:.1f} us = {2.0 * n ** 3 / t / 1e6:.0f} TF/s")


if __name__ == "__main__":
    main(sys.argv[1])
