"""Times every distinct conv/linear GEMM shape of a profiled UNet pass under forced (tile, split-K) plans and
prints the best next to the automatic plan's choice.  Input: the CSV written by bench.py --profile-csv."""
import csv
import math
import re
import sys

import torch

sys.path.insert(0, ".")
from diffute_amd import ops  # noqa: E402


TUNE_B = int(__import__('os').environ.get('TUNE_B', '4'))     # batch the profiled pass ran at (8 for the training step)


def time_it(fn, reps=12, replays=3):
    """us per call of fn(i).  The calls are captured into one graph (no host launch overhead: the python/ctypes path
    costs ~25 us per call, more than the small GEMMs) and fn rotates its weights (i) so they stream from HBM as in the
    UNet pass instead of sitting in L2 / MALL."""
    fn(0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for i in range(reps):
                fn(i)
    g.replay(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(replays):
        g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * replays) * 1e3     # us


def weights(N, K, total=384 << 20):
    n = max(2, min(12, total // (N * K * 2)))
    return [(torch.randn(N, K, device='cuda') / math.sqrt(K)).to(torch.bfloat16) for _ in range(n)]


def main(path, only=""):
    shapes = {}
    for r in csv.DictReader(open(path)):
        if int(r["class"]) not in (0, 1, 7, 9) and not 10 <= int(r["class"]) < 22:
            continue
        if only and only not in r["tag"]:
            continue
        m = dict(re.findall(r"(\w+)=(\d+)", r["tag"]))
        key = tuple(int(m[k]) for k in ("M", "N", "K", "ks", "st", "ups"))
        shapes[key] = shapes.get(key, 0) + 1
    dev = torch.device("cuda")
    tot_auto = tot_best = 0.0
    zx = torch.zeros(1, 1, 128, 64, device=dev, dtype=torch.bfloat16); zw = torch.zeros(64, 64, device=dev, dtype=torch.bfloat16)
    floor = time_it(lambda i: ops.conv_gemm(zx, zw, 64, ksize=1, pad=0))
    print(f'per-launch floor of this harness: {floor:.1f} us')
    for (M, N, K, ks, st, ups), cnt in sorted(shapes.items(), key=lambda kv: -kv[1] * kv[0][0] * kv[0][1] * kv[0][2]):
        B = TUNE_B if M % TUNE_B == 0 else 1
        if ks == 3 and ups != 2:
            ohw = M // B; OH = int(round(math.sqrt(ohw))); Cin = K // 9
            if Cin * 9 != K:      # fused shortcut: treat as plain K for timing
                ks = 1
        if ups == 2:                                   # phase-decomposed upsample conv: M = 4*B*H*W output rows, K = 4*Cin
            Cin = K // 4; H = int(round(math.sqrt(M // 4 // B)))
            x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
            ws = [w.view(4, N, K) for w in weights(4 * N, K)]
            run = lambda tn, sk, i=0: ops.conv_ups2x(x, ws[i % len(ws)], N, force_tn=tn, force_splitk=sk)
        elif ks == 3:
            H = OH // 2 if ups else (OH * 2 if st == 2 else OH)
            x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
            ws = weights(N, K)
            run = lambda tn, sk, i=0: ops.conv_gemm(x, ws[i % len(ws)], N, ksize=3, stride=st, pad=1, ups=bool(ups), force_tn=tn, force_splitk=sk)
        else:
            x = torch.randn(1, 1, M, K, device=dev).to(torch.bfloat16)
            ws = weights(N, K)
            run = lambda tn, sk, i=0: ops.conv_gemm(x, ws[i % len(ws)], N, ksize=1, pad=0, force_tn=tn, force_splitk=sk)
        t_auto = time_it(lambda i: run(0, 0, i))
        res = []
        for tn in (10, 9, 8, 7, 3, 2, 1):
            for sk in (1, 2, 3, 4, 6, 8, 12, 16):
                if K % (64 if tn in (3, 4, 5, 7, 8, 10) else 32):
                    continue
                if sk > 1 and (K // (64 if tn in (3, 4, 5, 7, 8, 10) else 32)) // sk < (4 if tn in (3, 4, 5, 7, 8, 10) else 8):
                    continue
                try:
                    res.append((time_it(lambda i: run(tn, sk, i)), tn, sk))
                except RuntimeError:
                    pass
        res.sort()
        bt, btn, bsk = res[0]
        fl = 2.0 * M * N * K
        tot_auto += t_auto * cnt; tot_best += bt * cnt
        print(f"M={M:6d} N={N:5d} K={K:6d} ks={ks} st={st} ups={ups} x{cnt:3d}: auto {t_auto:7.1f}us ({fl / t_auto / 1e6:6.0f} TF) | "
              f"best tn={btn} sk={bsk:2d} {bt:7.1f}us ({fl / bt / 1e6:6.0f} TF) | next {res[1][1]}/{res[1][2]} {res[1][0]:.1f}", flush=True)
    print(f"sum over pass: auto {tot_auto / 1e3:.2f} ms, best {tot_best / 1e3:.2f} ms")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
