"""Soak / repeat-determinism harness of the headline loop (what bench.py times and what a server calling text_editing()
(app.ipynb:653) repeatedly does): N back-to-back `denoise()` passes on ONE model with NO host synchronisation between them,
every result cloned on the loop's stream; afterwards every pass must be finite and bit-equal to pass 0.

    python scripts/soak.py [--passes 25] [--batch 4] [--latent 64] [--steps 50] [--fp16] [--ddpm]
                           [--no-graph] [--prefetch 0|1] [--halo 0|1|2] [--halo-ws 0|1] [--xf-chain N] [--gn-stats 0|1]
                           [--memset-nodes] [--pool-counts]  (probe builds: the round-4 memset-node behaviour and its direct evidence)
                           [--no-hold] [--sync]  (host sync after every pass: the regime the tests always ran)
Prints one JSON line; exit code 1 when a pass is non-finite or differs from pass 0."""
import argparse
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import diffute_amd as D  # noqa: E402
from diffute_amd import _cabi  # noqa: E402
from diffute_amd.synthetic import synth_inputs  # noqa: E402


def soak(unet, sched_cls, inputs, passes, steps, use_graph=True, sync=False, variance_noise=None, hold=True, step_report=None, pool_counts=None):
    """-> (list of per-pass dicts, outputs).  No host sync inside the loop unless `sync`.  hold=True keeps the previous pass's result
    tensor alive while the next pass runs (what `out = one_pass()` in bench.py does: the loop's buffers then alternate between
    addresses, i.e. between captured graphs); hold=False releases it first (every pass gets the same addresses)."""
    lat, mask, mlat, ctx = inputs
    outs = []
    o = None
    for _ in range(passes):
        if not hold:
            o = None
        cb = None
        if step_report is not None:
            flags = torch.zeros(steps, 2, dtype=torch.int32, device=lat.device)
            step_report.append(flags)

            def cb(i, t, x, eps, flags=flags):                      # device-side only: no host sync
                flags[i, 0] = (~torch.isfinite(eps)).sum()
                flags[i, 1] = (~torch.isfinite(x)).sum()
        o = D.denoise(unet, sched_cls(), lat, mask, mlat, ctx, steps, use_graph=use_graph, variance_noise=variance_noise, callback=cb)
        outs.append(o.clone())
        if pool_counts is not None:
            import ctypes
            buf = (ctypes.c_ulonglong * 8)()
            unet._lib.dmx_probe_pool_counts(buf, 1)
            pool_counts.append([int(v) for v in buf[:4]] + [hex(v) for v in buf[4:]])
        if sync:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    rep = []
    for i, o in enumerate(outs):
        fin = bool(torch.isfinite(o).all())
        bad = int((~torch.isfinite(o)).sum())
        same = bool(torch.equal(o, outs[0]))
        d = float((o - outs[0]).abs().max()) if fin and bool(torch.isfinite(outs[0]).all()) else float("nan")
        rep.append({"pass": i, "finite": fin, "nonfinite_elems": bad, "equal_pass0": same, "max_abs_diff_vs_pass0": d})
    return rep, outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", type=int, default=25)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--fp16", action="store_true")
    ap.add_argument("--ddpm", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--sync", action="store_true")
    ap.add_argument("--no-hold", action="store_true")
    ap.add_argument("--step-report", action="store_true", help="per-step non-finite counts of eps / x through denoise(callback=)")
    ap.add_argument("--prefetch", type=int, default=None)
    ap.add_argument("--halo", type=int, default=None)
    ap.add_argument("--halo-ws", type=int, default=None)
    ap.add_argument("--xf-chain", type=int, default=None)
    ap.add_argument("--gn-stats", type=int, default=None)
    ap.add_argument("--memset-nodes", action="store_true", help="-DDMX_PROBES builds only: zero the per-forward pools with memset nodes again (the round-4 behaviour)")
    ap.add_argument("--pool-counts", action="store_true", help="-DDMX_PROBES builds only (implies --memset-nodes): non-zero 16-byte units of the statistics / "
                    "flag pools before and after their memset node, per pass, and a sample of what the node wrote")
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    dev = torch.device("cuda")
    unet = D.UNet2DConditionModel(device=dev).requires_grad_(False)
    if args.fp16:
        unet.to(dtype=torch.float16)
    lib = unet._lib
    if args.prefetch is not None:
        lib.dmx_set_weight_prefetch(args.prefetch)
    if args.halo is not None:
        lib.dmx_set_halo_conv(args.halo)
    if args.halo_ws is not None:
        lib.dmx_set_halo_ws(args.halo_ws)
    if args.xf_chain is not None:
        lib.dmx_set_xf_chain(args.xf_chain)
    if args.gn_stats is not None:
        lib.dmx_set_gn_producer_stats(args.gn_stats)
    if args.memset_nodes or args.pool_counts:
        if not hasattr(lib, "dmx_set_pool_memset_nodes"):
            raise SystemExit("--memset-nodes / --pool-counts need a probe build: make -C diffute_amd/csrc EXTRA=-DDMX_PROBES")
        lib.dmx_set_pool_memset_nodes(1)
    unet._ensure_packed()
    inputs = synth_inputs(args.batch, args.latent, args.latent, 577, 1024, device=dev)
    nz = torch.randn(args.steps, args.batch, 4, args.latent, args.latent, device=dev) if args.ddpm else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sr = [] if args.step_report else None
    pc = None
    if args.pool_counts:
        pc = []
        lib.dmx_probe_pool_counts(None, 1)
        lib.dmx_set_pool_memset_nodes(3)
    rep, outs = soak(unet, D.DDPMScheduler if args.ddpm else D.DDIMScheduler, inputs, args.passes, args.steps,
                     use_graph=not args.no_graph, sync=args.sync, variance_noise=nz, hold=not args.no_hold, step_report=sr, pool_counts=pc)
    el = time.perf_counter() - t0
    bad = [r for r in rep if not (r["finite"] and r["equal_pass0"])]
    line = {"tag": args.tag, "argv": sys.argv[1:], "passes": args.passes, "ms_per_pass": round(1e3 * el / args.passes, 1),
            "all_finite": all(r["finite"] for r in rep), "all_equal_pass0": all(r["equal_pass0"] for r in rep),
            "bad_passes": bad[:4], "n_bad": len(bad), "first_bad_pass": bad[0]["pass"] if bad else None}
    if pc is not None:
        line["pool_nonzero_units_per_pass[stats_before,stats_after,flags_before,flags_after]"] = pc
    if sr is not None:
        first = {}
        for pi, fl in enumerate(sr):
            f = fl.cpu()
            nz_ = (f.sum(1) > 0).nonzero()
            if len(nz_):
                st = int(nz_[0])
                first[pi] = {"step": st, "eps_nonfinite": int(f[st, 0]), "x_nonfinite": int(f[st, 1])}
        line["first_bad_step_by_pass"] = dict(list(first.items())[:6])
    print(json.dumps(line), flush=True)
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
