"""In-situ GEMM plan tuning: every distinct GEMM signature of the UNet denoise pass is re-timed under candidate
(tile instance, split-K) plans INSIDE a real pass - real epilogue (bias / time embedding / residual / folded LayerNorm /
GEGLU / row statistics), real neighbours and cache state - using the per-launch hipEvent profile (dmx_profile_*) and the
run-time plan override (dmx_gemm_plan_override).  The split-K reduce pass is charged to its GEMM.  Prints gemm_tuned.h lines
for the plans that beat the current one by more than the threshold.

    python scripts/tune_in_situ.py [--batch 4] [--latent 64] [--steps 2] [--min-gain 0.03] [--top 60]
"""
import argparse
import collections
import csv
import ctypes
import os
import re
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffute_amd as D                                  # noqa: E402
from diffute_amd import _cabi                            # noqa: E402
from diffute_amd.synthetic import synth_inputs           # noqa: E402

CFG_BK = {0: 32, 1: 32, 2: 64, 6: 64, 7: 64, 8: 32, 9: 64, 10: 64, 11: 64, 12: 64, 14: 64, 15: 64}   # template instance -> K-tile (12, 14, 15: persistent stream-K)
TN_TO_CFG = {2: 0, 1: 1, 3: 2, 7: 6, 8: 7, 9: 8, 10: 9, 11: 10, 12: 11, 13: 12, 15: 14, 16: 15}
NCLASS = 26


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--min-gain", type=float, default=0.03)
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--only-cfgs", default="", help="comma-separated template instances to try (default: all)")
    ap.add_argument("--vae", action="store_true", help="tune the GEMMs of AutoencoderKL encode + decode of --batch images of 8*--latent px")
    ap.add_argument("--train", action="store_true", help="tune the GEMMs of one training forward + backward (use --batch 8) instead of the denoise pass")
    a = ap.parse_args()
    dev = torch.device("cuda")
    lib = _cabi.lib()
    unet = D.UNet2DConditionModel(device=dev)
    if not a.train:
        unet.requires_grad_(False)
    sched = D.DDIMScheduler()
    lat, mask, mlat, ctx = synth_inputs(a.batch, a.latent, a.latent, 577, 1024, device=dev)
    unet._ensure_packed()
    xin = torch.cat([lat, mask, mlat], 1); tt = torch.randint(0, 1000, (a.batch,), device=dev); tgt = torch.randn_like(lat)
    side = torch.cuda.Stream()

    vae = D.AutoencoderKL(device=dev).requires_grad_(False) if a.vae else None
    if a.vae:
        from diffute_amd.synthetic import synth_images
        img = synth_images(a.batch, 8 * a.latent, 8 * a.latent, device=dev)
        zlat = torch.randn(a.batch, 4, a.latent, a.latent, device=dev)

    state = {"out": None}                               # the latents of the last denoise pass: candidates are checked, not only timed

    def result_ok(ref):
        """a candidate plan's pass against the current plan's: finite and within the rounding noise of a different summation order"""
        out = state["out"]
        if ref is None or out is None:
            return True
        if not torch.isfinite(out).all():
            return False
        return float((out.float() - ref.float()).norm() / ref.float().norm()) <= 2e-2

    def run_pass():
        if a.vae:
            with torch.no_grad(), torch.cuda.stream(side):
                for _ in range(a.steps):
                    vae.encode(img); vae.decode(zlat)
            torch.cuda.synchronize()
            return
        if not a.train:
            state["out"] = D.denoise(unet, sched, lat, mask, mlat, ctx, a.steps).clone()
            return
        from diffute_amd.models import mse_loss
        with torch.cuda.stream(side):
            for _ in range(a.steps):
                mse_loss(unet(xin, tt, ctx).sample, tgt).backward()
        torch.cuda.synchronize()
    path = os.path.join(tempfile.gettempdir(), "in_situ_launches.csv")

    def profiled_pass():
        """{(M, N, K, st, ups): [launches, ms, tn, sk]} over one short pass; reduce launches are charged to their GEMM"""
        torch.cuda.synchronize()
        lib.dmx_profile_dump_path(path.encode())
        lib.dmx_profile_begin()
        buf = (ctypes.c_double * (4 * NCLASS))()
        try:
            run_pass()
        except RuntimeError:                               # e.g. a split-K candidate that needs more workspace than the model reserved
            torch.cuda.synchronize()
            lib.dmx_profile_end(buf, len(buf))
            return {}
        _cabi.check(lib.dmx_profile_end(buf, len(buf)), "profile_end")
        out = collections.OrderedDict()
        for r in csv.DictReader(open(path)):
            m = dict(re.findall(r"(\w+)=(\d+)", r["tag"]))
            if "M" not in m or "tn" not in m:
                continue
            key = tuple(int(m[k]) for k in ("M", "N", "K", "st", "ups"))
            e = out.setdefault(key, [0, 0.0, int(m["tn"]), int(m["sk"])])
            if int(r["class"]) != 2:                       # 2 = split-K reduce: time only
                e[0] += 1
            e[1] += float(r["ms"])
        return out

    _raw_override = lib.dmx_gemm_plan_override

    def override(*args):
        """plan override + forget the cached workspace size: a persistent plan needs slab workspace the classic plan did not"""
        _raw_override(*args)
        for sl in getattr(unet, "_slots", {}).values():
            sl["ws_need"] = None
    lib_override = override
    run_pass()                                                 # warm-up (one-time function attributes, context K/V)
    base = profiled_pass()
    ref_out = state["out"]
    base2 = profiled_pass()
    assert ref_out is None or torch.equal(ref_out, state["out"]), "two passes of the current plans differ"
    for k in base:
        base[k][1] = min(base[k][1], base2[k][1])
    order = sorted(base, key=lambda k: -base[k][1])[:a.top]
    total = sum(v[1] for v in base.values())
    print(f"{len(base)} GEMM signatures, {total:.2f} ms per {a.steps}-step pass (profiled); tuning the top {len(order)}", flush=True)
    lines = []
    for key in order:
        M, N, K, st, ups = key
        n0, ms0, tn0, sk0 = base[key]
        cfg0 = TN_TO_CFG[tn0]
        cands = []
        only = [int(c) for c in a.only_cfgs.split(",")] if a.only_cfgs else None
        for cfg, bk in CFG_BK.items():
            if K % bk or (only is not None and cfg not in only):
                continue
            nkt = K // bk
            for sk in (1, 2, 3, 4, 6, 8, 12, 16):
                if sk > 1 and cfg >= 12:
                    continue                               # the persistent stream-K instances take no split-K
                if sk > 1 and nkt // sk < (4 if bk == 64 else 8):
                    continue
                if (cfg, sk) != (cfg0, sk0):
                    cands.append((cfg, sk))
        res = []
        for cfg, sk in cands:
            lib_override(M, N, K, st, ups, cfg, sk)
            got = profiled_pass().get(key)
            ok = result_ok(ref_out)
            lib_override(M, N, K, st, ups, cfg0, sk0)
            if not ok:
                print(f"  cfg {cfg}/sk {sk} on M={M} N={N} K={K}: REJECTED - the pass's result is non-finite or > 2e-2 from the current plans'", flush=True)
                continue
            if got is None or TN_TO_CFG[got[2]] != cfg or (got[3] != sk and not (sk > 1 and got[3] > 1)):
                continue                                   # override not applicable to this GEMM (epilogue / alignment)
            res.append((got[1], cfg, got[3]))
        res.sort()
        # confirm the finalists against a fresh measurement of the current plan
        best = None
        if res and res[0][0] < (1 - a.min_gain) * ms0:
            ref = min(profiled_pass()[key][1], ms0)
            for ms, cfg, sk in res[:2]:
                lib_override(M, N, K, st, ups, cfg, sk)
                ms2 = min(ms, profiled_pass()[key][1])
                lib_override(M, N, K, st, ups, cfg0, sk0)
                if ms2 < (1 - a.min_gain) * ref and (best is None or ms2 < best[0]):
                    best = (ms2, cfg, sk, ref)
        if best:
            ms2, cfg, sk, ref = best
            lib_override(M, N, K, st, ups, cfg, sk)         # keep it: later signatures are tuned next to the better plan
            lines.append(f"    {{{M}, {N}, {K}, 0, {st}, {ups}, {cfg}, {sk}}},   // in situ: {1e3 * ref / n0:.1f} -> {1e3 * ms2 / n0:.1f} us x{n0 // a.steps} per step")
            print(f"M={M} N={N} K={K} st={st} ups={ups} x{n0}: cfg {cfg0}/sk {sk0} {1e3 * ref / n0:7.1f} us -> cfg {cfg}/sk {sk} {1e3 * ms2 / n0:7.1f} us", flush=True)
        else:
            lib_override(M, N, K, st, ups, cfg0, sk0)
            second = f"(best other: cfg {res[0][1]}/sk {res[0][2]} {1e3 * res[0][0] / n0:.1f} us)" if res else ""
            print(f"M={M} N={N} K={K} st={st} ups={ups} x{n0}: cfg {cfg0}/sk {sk0} {1e3 * ms0 / n0:7.1f} us kept {second}", flush=True)
    final = profiled_pass()
    print(f"pass GEMM time (profiled, {a.steps} steps): {total:.2f} -> {sum(v[1] for v in final.values()):.2f} ms")
    print("gemm_tuned.h lines:")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
