"""A/B of the fused GroupNorm -> conv launch on BASELINE configs[2] (AutoencoderKL encode + decode, 512 px, batch 32) in one process.
    python scripts/ab_vae.py [--rounds 3] [--batch 32]"""
import sys, time
import torch
sys.path.insert(0, ".")
import diffute_amd as D
from diffute_amd import _cabi
from diffute_amd.synthetic import text_crop_images
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
batch = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 32
dev = torch.device("cuda"); lib = _cabi.lib()
vae = D.AutoencoderKL(device=dev).requires_grad_(False)
img = text_crop_images(batch, 512, 512, device=dev)
def timed():
    with torch.no_grad():
        z = vae.encode(img).latent_dist.mode(); vae.decode(z); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            z = vae.encode(img).latent_dist.mode(); torch.cuda.synchronize(); t1 = time.perf_counter()
            vae.decode(z); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 3 * 1e3
res = {1: [], 0: []}
for r in range(rounds):
    for on in (1, 0):
        if "--ws" in sys.argv: lib.dmx_set_halo_ws(on)       # the warp-specialised halo instances vs the ping-pong
        else: lib.dmx_set_halo_conv(on)
        for attr in ("_slots", "_ws"):
            if hasattr(vae, attr) and isinstance(getattr(vae, attr), dict): getattr(vae, attr).clear()
        res[on].append(timed())
    print(f"round {r}: on {res[1][-1]:.1f} ms, off {res[0][-1]:.1f} ms", flush=True)
print(f"vae b{batch}: halo on median {sorted(res[1])[len(res[1]) // 2]:.1f} ms, off median {sorted(res[0])[len(res[0]) // 2]:.1f} ms (encode + decode)")
