"""split-K reduce / GroupNorm launch counts and times of one profiled 50-step pass with dmx_set_defer_reduce on and off.  Measurement aid."""
import ctypes, sys
import torch
sys.path.insert(0, ".")
import diffute_amd as D
from diffute_amd import _cabi
from diffute_amd.synthetic import synth_inputs
dev = torch.device("cuda"); lib = _cabi.lib()
unet = D.UNet2DConditionModel(device=dev).requires_grad_(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lat, mask, mlat, ctx = synth_inputs(B, 64, 64, 577, 1024, device=dev)
unet._ensure_packed()
for on in (1, 0):
    lib.dmx_set_defer_reduce(on)
    for sl in unet._slots.values(): sl["ws_need"] = None
    _cabi.check(lib.dmx_unet_refresh_derived(unet._h, None), "refresh")
    D.denoise(unet, D.DDIMScheduler(), lat, mask, mlat, ctx, 50); torch.cuda.synchronize()
    lib.dmx_profile_dump_path(f"/tmp/defer_{on}.csv".encode())
    lib.dmx_profile_begin()
    D.denoise(unet, D.DDIMScheduler(), lat, mask, mlat, ctx, 50)
    buf = (ctypes.c_double * (4 * 28))()
    _cabi.check(lib.dmx_profile_end(buf, len(buf)), "profile_end")
    print(f"defer={on}: split-K reduce {int(buf[4*2])} launches {buf[4*2+1]:.2f} ms | GroupNorm {int(buf[4*4])} launches {buf[4*4+1]:.2f} ms | total launches {int(sum(buf[4*i] for i in range(28)))} total ms {sum(buf[4*i+1] for i in range(28)):.1f}")
    import collections
    agg = collections.defaultdict(lambda: [0, 0.0])
    for l in open(f"/tmp/defer_{on}.csv").read().splitlines()[1:]:
        c, ms, fl, by, rest = l.split(",", 4)
        if int(c) in (2, 4):
            tag = rest.rsplit(',"', 1)[0]
            agg[(int(c), tag)][0] += 1; agg[(int(c), tag)][1] += float(ms)
    for (c, tag), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"    class {c} {tag:60s} {n:5d} x {1e3*ms/n:6.1f} us = {ms:6.2f} ms")
lib.dmx_set_defer_reduce(1)
