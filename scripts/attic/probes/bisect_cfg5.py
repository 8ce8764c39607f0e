import sys, torch
sys.path.insert(0, '.')
import diffute_amd as D
from diffute_amd import _cabi
from diffute_amd.synthetic import synth_inputs
cuda = torch.device('cuda')
unet = D.UNet2DConditionModel(device=cuda).requires_grad_(False)
lat, mask, mlat, ctx = synth_inputs(2, 96, 96, 577, 1024, device=cuda)
t = torch.tensor([981], device=cuda)
for elem in ('bf16', 'fp16'):
    if elem == 'fp16':
        unet.to(dtype=torch.float16)
    lib = unet._lib
    for (pf, ws, halo) in [(1, 1, 1), (0, 1, 1), (1, 0, 1), (0, 0, 1), (0, 0, 0)]:
        lib.dmx_set_weight_prefetch(pf); lib.dmx_set_halo_ws(ws); lib.dmx_set_halo_conv(halo)
        for sl in unet._slots.values(): sl["ws_need"] = None
        unet.set_context(ctx)
        y = unet.forward_parts([lat, mask, mlat], t).clone()
        bad = 0
        for _ in range(4):
            y2 = unet.forward_parts([lat, mask, mlat], t)
            bad += int(not torch.equal(y, y2))
        print(elem, f"prefetch={pf} ws={ws} halo={halo}: {bad} of 4 repeats differ, max diff {float((y2 - y).abs().max()):.3e}", flush=True)
