"""Times the big AutoencoderKL convolutions (cfg3: batch 32, 512 px) on every GEMM tile configuration (force_tn).  Measurement aid."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffute_amd import ops  # noqa: E402

dev = torch.device("cuda")
B = int(os.environ.get("B", "32"))
SHAPES = [  # (H, W, Cin, Cout, res)
    (512, 512, 128, 128, 1), (256, 256, 256, 256, 1), (128, 128, 512, 512, 1), (64, 64, 512, 512, 1), (256, 256, 128, 256, 0),
]
TNS = [int(t) for t in sys.argv[1:]] or [0, 3, 7, 9, 10, 11]


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for (H, W, Ci, Co, res) in SHAPES:
    x = torch.randn(B, H, W, Ci, device=dev).to(torch.bfloat16)
    w = (torch.randn(Co, 9 * Ci, device=dev) / math.sqrt(9 * Ci)).to(torch.bfloat16)
    b = torch.randn(Co, device=dev) * 0.1
    r = torch.randn(B, H, W, Co, device=dev).to(torch.bfloat16) if res and Ci == Co else None
    fl = 2.0 * B * H * W * Co * 9 * Ci
    line = f"B={B} {H}x{W} {Ci}->{Co}:"
    for tn in TNS:
        try:
            t = timeit(lambda: ops.conv_gemm(x, w, Co, bias=b, res=r, force_tn=tn))
            line += f"  tn={tn}: {t:6.2f} ms ({fl / t / 1e9:5.0f} TF)"
        except Exception as e:  # noqa: BLE001
            line += f"  tn={tn}: failed ({str(e)[:30]})"
    print(line, flush=True)
    del x, w, r
