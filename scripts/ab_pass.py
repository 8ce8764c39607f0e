"""A/B of a library switch inside ONE process and on one box: alternating timed 50-step passes with the switch on / off.
    python scripts/ab_pass.py gn_stats|xf_chain|xf_gn_fold|halo_ws|weight_prefetch|temb_table|halo|halo_all|halo_peers|attn_balanced|defer_reduce|defer_cross [--rounds 4] [--batch 4]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import diffute_amd as D  # noqa: E402
from diffute_amd import _cabi  # noqa: E402
from diffute_amd.synthetic import synth_inputs  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "gn_stats"
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 4
batch = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 4
latent = int(sys.argv[sys.argv.index("--latent") + 1]) if "--latent" in sys.argv else 64
dev = torch.device("cuda")
lib = _cabi.lib()
unet = D.UNet2DConditionModel(device=dev).requires_grad_(False)
if "--fp16" in sys.argv:
    unet.to(dtype=torch.float16)
    lib = unet._lib
lat, mask, mlat, ctx = synth_inputs(batch, latent, latent, 577, 1024, device=dev)


def setting(on):
    if what == "gn_stats":
        lib.dmx_set_gn_producer_stats(int(on))
    elif what == "temb_table":
        import diffute_amd.pipeline as P
        P.TEMB_TABLE = bool(on)
    elif what == "halo":
        lib.dmx_set_halo_conv(1 if on else 0)         # 1: the fused GroupNorm -> conv launch where it pays (the default)
    elif what == "halo_all":
        lib.dmx_set_halo_conv(2 if on else 1)         # 2: wherever the kernel takes the problem vs the default rule
    elif what == "halo_peers":
        lib.dmx_set_halo_peers(1 if on else 0)        # a tile's K-split peers on one XCD, slabs through its L2, vs the round-5 dealing with write-through slabs
    elif what == "attn_balanced":
        lib.dmx_set_attn_balanced(1 if on else 0)     # stream-K attention (attention_sk.hip) where its plan takes the launch vs the plain grid everywhere
    elif what == "weight_prefetch":
        lib.dmx_set_weight_prefetch(1 if on else 0)     # every launch touches the weights of the launches that follow it (Exec::note / peek)
    elif what == "halo_ws":
        lib.dmx_set_halo_ws(1 if on else 0)           # the warp-specialised halo instances vs the two-group ping-pong
    elif what == "defer_reduce":
        lib.dmx_set_defer_reduce(1 if on else 0)      # split-K conv1 of a resnet leaves its reduce pass to norm2's slab kernel
    elif what == "defer_cross":
        lib.dmx_set_defer_reduce(1 if on else 2)      # conv2's reduce in the NEXT block's norm1 (round 6) vs only the in-block deferral of round 5
    elif what == "xf_gn_fold":
        lib.dmx_set_xf_chain(1 if on else 5)          # bit 2: chains without the folded entry GroupNorm
    elif what == "xf_chain":
        lib.dmx_set_xf_chain(2 if on else 0)          # 2: at every supported size (the executor's own rule needs >= 192 row blocks)
    for sl in unet._slots.values():
        sl["ws_need"] = None
    unet._ensure_packed()
    _cabi.check(lib.dmx_unet_refresh_derived(unet._h, None), "refresh")     # drops the captured graphs (they embed the old setting)
    torch.cuda.synchronize()


def timed():
    D.denoise(unet, D.DDIMScheduler(), lat, mask, mlat, ctx, 50)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = D.denoise(unet, D.DDIMScheduler(), lat, mask, mlat, ctx, 50)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 3 * 1e3, out.clone()


# results are compared, not only times: every pass finite, every pass of one setting bit-equal to that setting's first pass,
# and the two settings against each other (rel-L2; 0 = the switch does not change a bit)
res = {True: [], False: []}
first = {}
for r in range(rounds):
    for on in (True, False):
        setting(on)
        ms, out = timed()
        res[on].append(ms)
        assert torch.isfinite(out).all(), f"{what}={on}: non-finite latents in round {r}"
        if on not in first:
            first[on] = out
        else:
            assert torch.equal(out, first[on]), f"{what}={on}: round {r} differs from round 0 (max abs {float((out - first[on]).abs().max()):.3e})"
    print(f"round {r}: on {res[True][-1]:.1f} ms, off {res[False][-1]:.1f} ms", flush=True)
print(f"{what}: on  min {min(res[True]):.1f} median {sorted(res[True])[len(res[True]) // 2]:.1f} ms per pass")
print(f"{what}: off min {min(res[False]):.1f} median {sorted(res[False])[len(res[False]) // 2]:.1f} ms per pass")
d = float((first[True].float() - first[False].float()).norm() / first[False].float().norm())
print(f"{what}: results finite and bit-repeatable in both settings; on vs off rel-L2 {d:.3e}" + (" (bit-equal)" if torch.equal(first[True], first[False]) else ""))
