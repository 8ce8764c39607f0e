"""In-situ A/B of the deep-ring GEMM instances (plan ids 3 / 4) on the K = C linears of the 16x16 / 32x32 levels via dmx_gemm_plan_override."""
import sys, time, ctypes, torch
sys.path.insert(0, ".")
import diffute_amd as D
from diffute_amd import _cabi
from diffute_amd.synthetic import synth_inputs
dev = torch.device("cuda"); lib = _cabi.lib()
unet = D.UNet2DConditionModel(device=dev).requires_grad_(False)
lat, mask, mlat, ctx = synth_inputs(4, 64, 64, 577, 1024, device=dev)
lib.dmx_gemm_plan_override.argtypes = [ctypes.c_int] * 7
def setting(plans):
    lib.dmx_gemm_plan_override(0, 0, 0, 0, 0, -1, 0)
    for (M, N, K, cfg, sk) in plans:
        lib.dmx_gemm_plan_override(M, N, K, 1, 0, cfg, sk)
    for sl in unet._slots.values(): sl["ws_need"] = None
    unet._ensure_packed()
    _cabi.check(lib.dmx_unet_refresh_derived(unet._h, None), "refresh")
    torch.cuda.synchronize()
def timed():
    D.denoise(unet, D.DDIMScheduler(), lat, mask, mlat, ctx, 50); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): D.denoise(unet, D.DDIMScheduler(), lat, mask, mlat, ctx, 50)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 3 * 1e3
variants = {"base": [], "deep 16x16 (1024x1280x1280 -> cfg 3)": [(1024, 1280, 1280, 3, 1)], "deep 32x32 (4096x640x640 -> cfg 4)": [(4096, 640, 640, 4, 1)],
            "deep 32x32 cfg 3": [(4096, 640, 640, 3, 1)], "both": [(1024, 1280, 1280, 3, 1), (4096, 640, 640, 4, 1)]}
for r in range(2):
    for name, plans in variants.items():
        setting(plans); print(f"{name}: {timed():.1f} ms", flush=True)
