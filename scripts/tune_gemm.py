"""Times every distinct conv/linear GEMM shape of a profiled UNet pass under forced (tile, split-K) plans and
prints the best next to the automatic plan's choice.  Input: the CSV written by bench.py --profile-csv."""
import csv
import math
import re
import sys

import torch

sys.path.insert(0, ".")
from diffute_amd import ops  # noqa: E402


def time_it(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3     # us


def main(path):
    shapes = {}
    for r in csv.DictReader(open(path)):
        if int(r["class"]) not in (0, 1, 7):
            continue
        m = dict(re.findall(r"(\w+)=(\d+)", r["tag"]))
        key = tuple(int(m[k]) for k in ("M", "N", "K", "ks", "st", "ups"))
        shapes[key] = shapes.get(key, 0) + 1
    dev = torch.device("cuda")
    tot_auto = tot_best = 0.0
    floor = time_it(lambda: ops.conv_gemm(torch.zeros(1, 1, 128, 64, device=dev, dtype=torch.bfloat16), torch.zeros(64, 64, device=dev, dtype=torch.bfloat16), 64, ksize=1, pad=0))
    print(f'host/launch floor of this harness: {floor:.1f} us per call')
    for (M, N, K, ks, st, ups), cnt in sorted(shapes.items(), key=lambda kv: -kv[1] * kv[0][0] * kv[0][1] * kv[0][2]):
        B = 4 if M % 4 == 0 else 1
        if ks == 3:
            ohw = M // B; OH = int(round(math.sqrt(ohw))); Cin = K // 9
            if Cin * 9 != K:      # fused shortcut: treat as plain K for timing
                ks = 1
        if ks == 3:
            H = OH // 2 if ups else (OH * 2 if st == 2 else OH)
            x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16)
            w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
            run = lambda tn, sk: ops.conv_gemm(x, w, N, ksize=3, stride=st, pad=1, ups=bool(ups), force_tn=tn, force_splitk=sk)
        else:
            x = torch.randn(1, 1, M, K, device=dev).to(torch.bfloat16)
            w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
            run = lambda tn, sk: ops.conv_gemm(x, w, N, ksize=1, pad=0, force_tn=tn, force_splitk=sk)
        t_auto = time_it(lambda: run(0, 0))
        res = []
        for tn in (3, 2, 1):
            for sk in (1, 2, 3, 4, 6, 8, 12, 16):
                if sk > 1 and (K // (64 if tn == 3 else 32)) // sk < (4 if tn == 3 else 8):
                    continue
                try:
                    res.append((time_it(lambda: run(tn, sk)), tn, sk))
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
    main(sys.argv[1])
