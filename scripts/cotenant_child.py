"""One process of the co-tenancy reproduction (scripts/cotenant_repro.py; VERDICT r5 item 1, EXPERIMENTS.md round 5 item 6:
`test_cfg4_train_step_batch8_512px` once produced a silently different sample-0 gradient on a lease where other processes may have shared the GPU).

    train         the cfg4 step of tests/test_configs_gpu.py (SD2-inpaint UNet, 8 x 512 px, train_diffute_v1.py:913-925): --steps full B = 8 steps,
                  then the "loss of sample 0 only" step inside the batch and the B = 1 step on sample 0.  After every step: host sync, device-error
                  poll, and a checksum (sum of the int32 bit patterns, sum of squares in double) of the prediction and of EVERY exported gradient
                  tensor -> JSON.  --occupy N keeps N CUs busy on three side streams (N / 3 each) while each step runs (dmx_test_occupy_cus).
    denoise-loop  a co-tenant: the headline denoise loop (batch 4, 512 px, --dsteps DDIM steps) repeated until <ctl>.stop exists; the first pass runs
                  BEFORE <ctl>.ready is written (quiet reference), every later pass is classified equal / raised / SILENTLY DIFFERENT.
    tiny-train-loop  a co-tenant like tests/d1_world2_worker.py: tiny UNet, forward + backward with the world-2 gloo all-reduce inside the backward, looped.
    idle          a co-tenant that only holds a GPU context (one tiny allocation) until <ctl>.stop exists.
    churn         a co-tenant that keeps allocating, filling and FREEING 2 GB blocks (hipMalloc / hipFree through torch with empty_cache: page-table updates
                  and TLB shoot-downs next to the running step - what a process that is starting up on the same GPU does).
--wait-go: do not touch the GPU (not even import torch) until <ctl>.go exists (tests/conftest.py starts co-tenants before pytest initialises HIP).
"""
import argparse
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TINY_UNET = dict(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4), cross_attention_dim=128)


def checksum(t):
    import torch
    t = t.detach().contiguous()
    bits = t.view(torch.int32) if t.dtype == torch.float32 else t.view(torch.int16).to(torch.int32)
    return [int(bits.to(torch.int64).sum()), float(t.double().pow(2).sum())]


def train(args):
    import torch
    import diffute_amd as D
    from diffute_amd import _cabi
    from diffute_amd.models import mse_loss
    from diffute_amd.synthetic import synth_inputs
    from oracle import prng
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    lib = _cabi.lib()
    unet = D.UNet2DConditionModel(device=dev)
    lat, mask, mlat, ctx = synth_inputs(8, 64, 64, 577, 1024, device=dev)
    x = torch.cat([lat, mask, mlat], 1)
    t = torch.tensor([437, 12, 999, 650, 3, 800, 250, 501], device=dev)
    tgt = torch.from_numpy(prng.normal(9, 43, 8 * 4 * 64 * 64).reshape(8, 4, 64, 64)).to(dev)
    sides = [torch.cuda.Stream(device=dev) for _ in range(3)]       # (streams share a few hardware queues: a hog on ONE side stream may just serialise with the step)
    res = {"mode": "train", "tag": args.tag, "occupy": args.occupy, "steps": [], "env": {k: os.environ[k] for k in os.environ if k.startswith(("AMD_", "HIP_", "HSA_"))}}

    def step(kind, xs, ts, cs, tg, sel=None):
        rec = {"kind": kind, "device_error": None}
        try:
            unet.zero_grad(set_to_none=True)
            if args.occupy:
                for si, side in enumerate(sides):
                    with torch.cuda.stream(side):
                        _cabi.check(lib.dmx_test_occupy_cus((args.occupy + 2 - si) // 3, 45_000_000, _cabi.current_stream()), "occupy")      # 450 ms: covers the step
                time.sleep(0.02)
            t0 = time.perf_counter()
            pred = unet(xs, ts, cs).sample
            loss = mse_loss(pred if sel is None else pred[sel], tg if sel is None else tg[sel])
            loss.backward()
            torch.cuda.synchronize()
            rec["wall_ms"] = (time.perf_counter() - t0) * 1e3
            _cabi.poll_device_error()
            rec["loss"] = float(loss.detach())
            rec["pred"] = checksum(pred)
            rec["grads"] = {k: checksum(p.grad) for k, p in unet.named_parameters()}
        except RuntimeError as e:
            torch.cuda.synchronize()
            rec["device_error"] = str(e)[:400]
            try:
                _cabi.poll_device_error()
            except RuntimeError as e2:
                rec["device_error"] += " | " + str(e2)[:200]
        res["steps"].append(rec)
        return rec

    for i in range(args.steps):
        step(f"full{i}", x, t, ctx, tgt)
    step("sel0", x, t, ctx, tgt, sel=slice(0, 1))
    step("b1", x[:1].contiguous(), t[:1].contiguous(), ctx[:1].contiguous(), tgt[:1].contiguous())
    with open(args.out, "w") as f:
        json.dump(res, f)


def wait_stop(ctl):
    return os.path.exists(ctl + ".stop")


def denoise_loop(args):
    import torch
    import diffute_amd as D
    from diffute_amd.synthetic import synth_inputs
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    unet = D.UNet2DConditionModel(device=dev).requires_grad_(False)
    lat, mask, mlat, ctx = synth_inputs(4, 64, 64, 577, 1024, device=dev)
    ref = D.denoise(unet, D.DDIMScheduler(), lat, mask, mlat, ctx, args.dsteps).clone()
    torch.cuda.synchronize()
    res = {"mode": "denoise-loop", "passes": 0, "equal": 0, "raised": 0, "silently_different": 0, "errors": []}
    open(args.ctl + ".ready", "w").close()
    while not wait_stop(args.ctl):
        try:
            out = D.denoise(unet, D.DDIMScheduler(), lat, mask, mlat, ctx, args.dsteps)
            torch.cuda.synchronize()
            from diffute_amd import _cabi
            _cabi.poll_device_error()
            res["equal" if torch.equal(out, ref) else "silently_different"] += 1
        except RuntimeError as e:
            torch.cuda.synchronize()
            res["raised"] += 1
            if len(res["errors"]) < 5:
                res["errors"].append(str(e)[:300])
        res["passes"] += 1
    with open(args.out, "w") as f:
        json.dump(res, f)


def tiny_train_loop(args):
    os.environ.update(RANK=str(args.rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(args.port))
    import torch
    import torch.distributed as dist
    import diffute_amd as D
    from diffute_amd.models import mse_loss
    from diffute_amd.synthetic import synth_inputs
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=args.rank, world_size=2)
    model = D.UNet2DConditionModel(**TINY_UNET).cuda()
    model.set_gradient_sync(dist, mode="all_reduce")
    lat, mask, mlat, ctx = synth_inputs(1, 8, 8, 20, 128, seed=10 * args.rank, device=dev)
    x, t, tg = torch.cat([lat, mask, mlat], 1), torch.tensor([321 + 111 * args.rank], device=dev), torch.full((1, 4, 8, 8), 1.0 - 0.5 * args.rank, device=dev)
    res = {"mode": "tiny-train-loop", "rank": args.rank, "passes": 0}
    ready = False
    stop = torch.zeros(1)
    while True:
        model.zero_grad(set_to_none=True)
        mse_loss(model(x, t, ctx).sample, tg).backward()
        torch.cuda.synchronize()
        res["passes"] += 1
        if not ready:
            open(args.ctl + f".ready{args.rank}", "w").close()
            ready = True
        stop[0] = 1.0 if wait_stop(args.ctl) else 0.0
        dist.all_reduce(stop)                      # both ranks leave in the same iteration
        if stop[0] > 0:
            break
    dist.destroy_process_group()
    with open(args.out, "w") as f:
        json.dump(res, f)


def churn(args):
    import torch
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    n = 0
    open(args.ctl + f".ready{args.rank}", "w").close()
    while not wait_stop(args.ctl):
        a = torch.randn(512 << 20, device=dev)          # 2 GB
        b = a * 2.0
        torch.cuda.synchronize()
        del a, b
        torch.cuda.empty_cache()
        n += 1
    with open(args.out, "w") as f:
        json.dump({"mode": "churn", "rounds": n}, f)


def idle(args):
    import torch
    torch.zeros(16, device="cuda:0")
    torch.cuda.synchronize()
    open(args.ctl + f".ready{args.rank}", "w").close()
    while not wait_stop(args.ctl):
        time.sleep(0.1)
    with open(args.out, "w") as f:
        json.dump({"mode": "idle"}, f)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["train", "denoise-loop", "tiny-train-loop", "idle", "churn"])
    ap.add_argument("--wait-go", action="store_true")
    ap.add_argument("--out", required=True)
    ap.add_argument("--tag", default="")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--occupy", type=int, default=0)
    ap.add_argument("--ctl", default="")
    ap.add_argument("--dsteps", type=int, default=10)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--port", type=int, default=29533)
    args = ap.parse_args()
    if args.wait_go:
        t0 = time.time()
        while not os.path.exists(args.ctl + ".go"):
            if wait_stop(args.ctl) or time.time() - t0 > 3600:
                with open(args.out, "w") as f:
                    json.dump({"mode": args.mode, "never_started": True}, f)
                return
            time.sleep(0.2)
    try:
        {"train": train, "denoise-loop": denoise_loop, "tiny-train-loop": tiny_train_loop, "idle": idle, "churn": churn}[args.mode](args)
    except Exception as e:                             # noqa: BLE001 - the orchestrator reads the file
        with open(args.out, "w") as f:
            json.dump({"mode": args.mode, "fatal": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()[-2000:]}, f)
        sys.exit(1)


if __name__ == "__main__":
    main()
