"""Bit-equality of the 50-step pass between two builds of the library (argv[1], argv[2]; "-" = the in-tree build), each in its own process.
    python scripts/ab_equal.py save <lib|-> <file>     then     python scripts/ab_equal.py cmp <lib|-> <file>
Measurement aid for changes that must not change a result (register allocation, address arithmetic)."""
import sys

import torch

sys.path.insert(0, ".")
from diffute_amd import _cabi  # noqa: E402
mode, libp, path = sys.argv[1], sys.argv[2], sys.argv[3]
if libp != "-":
    _cabi._LIB_PATH = libp
import diffute_amd as D  # noqa: E402
from diffute_amd.synthetic import synth_inputs  # noqa: E402

dev = torch.device("cuda")
unet = D.UNet2DConditionModel(device=dev).requires_grad_(False)
outs = {}
for B, lt in ((4, 64), (1, 64), (2, 96)):
    lat, mask, mlat, ctx = synth_inputs(B, lt, lt, 577, 1024, device=dev)
    outs[f"b{B}_l{lt}"] = D.denoise(unet, D.DDIMScheduler(), lat, mask, mlat, ctx, 10).cpu()
if mode == "save":
    torch.save(outs, path)
    print("saved", {k: float(v.abs().mean()) for k, v in outs.items()})
else:
    ref = torch.load(path)
    for k in outs:
        print(k, "bit-equal" if torch.equal(ref[k], outs[k]) else f"DIFFERS: max abs {float((ref[k] - outs[k]).abs().max()):.3e}")
