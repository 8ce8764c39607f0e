"""In-situ time of every fused GroupNorm -> conv3x3 launch shape of the 50-step pass with the warp-specialised instances on / off
(one process, profiled passes: every launch bracketed by hipEvents; the (class, shape) tags of the library's profiler).
    python scripts/halo_ws_by_shape.py [--fp16] [--batch 4] [--latent 64]"""
import collections
import csv
import os
import sys
import tempfile

import torch

sys.path.insert(0, ".")
import diffute_amd as D  # noqa: E402
from diffute_amd import _cabi  # noqa: E402
from diffute_amd.synthetic import synth_inputs  # noqa: E402

batch = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 4
latent = int(sys.argv[sys.argv.index("--latent") + 1]) if "--latent" in sys.argv else 64
dev = torch.device("cuda")
unet = D.UNet2DConditionModel(device=dev).requires_grad_(False)
if "--fp16" in sys.argv:
    unet.to(dtype=torch.float16)
lib = unet._lib
lat, mask, mlat, ctx = synth_inputs(batch, latent, latent, 577, 1024, device=dev)


def profiled(on):
    lib.dmx_set_halo_ws(int(on))
    for sl in unet._slots.values():
        sl["ws_need"] = None
    unet._ensure_packed()
    _cabi.check(lib.dmx_unet_refresh_derived(unet._h, None), "refresh")
    D.denoise(unet, D.DDIMScheduler(), lat, mask, mlat, ctx, 50)
    torch.cuda.synchronize()
    path = os.path.join(tempfile.mkdtemp(), "p.csv")
    lib.dmx_profile_dump_path(path.encode())
    lib.dmx_profile_begin()
    D.denoise(unet, D.DDIMScheduler(), lat, mask, mlat, ctx, 50)
    import ctypes
    buf = (ctypes.c_double * 256)()
    lib.dmx_profile_end(buf, len(buf))
    agg = collections.OrderedDict()
    tot = 0.0
    for r in csv.DictReader(open(path)):
        tot += float(r["ms"])
        if "halo" in r["tag"]:
            a = agg.setdefault(r["tag"], [0, 0.0]); a[0] += 1; a[1] += float(r["ms"])
    return agg, tot


res = {}
for rnd in range(2):
    for on in (1, 0):
        agg, tot = profiled(on)
        res.setdefault(on, []).append((agg, tot))
print(f"# halo launches by shape, us per launch in the profiled pass (two rounds each): warp-specialised on | off   [batch {batch}, latent {latent}, {unet._elem}]")
for tag in res[1][0][0]:
    on = [1e3 * r[0][tag][1] / r[0][tag][0] for r in res[1]]
    off = [1e3 * r[0][tag][1] / r[0][tag][0] for r in res[0] if tag in r[0]]
    n = res[1][0][0][tag][0]
    print(f"{n:5d} x  on {on[0]:7.1f} {on[1]:7.1f}   off {off[0]:7.1f} {off[1]:7.1f}   {tag}")
print("# profiled pass totals, ms: on", " ".join(f"{r[1]:.1f}" for r in res[1]), "| off", " ".join(f"{r[1]:.1f}" for r in res[0]))
