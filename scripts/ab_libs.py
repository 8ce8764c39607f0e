"""Times the 50-step pass with an alternative build of the library (path as argv[1], default: the in-tree build): for A/Bs of
compile-time variants on one box (run twice in one gpurun call).  Measurement aid."""
import sys
import time

import torch

sys.path.insert(0, ".")
from diffute_amd import _cabi  # noqa: E402
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _cabi._LIB_PATH = sys.argv[1]
import diffute_amd as D  # noqa: E402
from diffute_amd.synthetic import synth_inputs  # noqa: E402

dev = torch.device("cuda")
unet = D.UNet2DConditionModel(device=dev).requires_grad_(False)
lat, mask, mlat, ctx = synth_inputs(4, 64, 64, 577, 1024, device=dev)
ts = []
for r in range(4):
    D.denoise(unet, D.DDIMScheduler(), lat, mask, mlat, ctx, 50)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    D.denoise(unet, D.DDIMScheduler(), lat, mask, mlat, ctx, 50)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"{_cabi._LIB_PATH}: min {min(ts):.1f} median {sorted(ts)[len(ts) // 2]:.1f} ms per pass")
